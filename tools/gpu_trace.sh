# kernel timeline of one bench step (rocprofv3 --kernel-trace), raw csv kept under gpurun_out/trace_<tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-t}
mkdir -p gpurun_out/trace_$T
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_$T -o tr -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/trace_$T.log 2>&1
find gpurun_out/trace_$T -name "*kernel_trace.csv" | head -n 1
