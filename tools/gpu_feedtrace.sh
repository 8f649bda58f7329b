# timeline of the GPU feed: rocprofv3 kernel + memory-copy trace of tools/gpu_feedtrace.py, csv kept under gpurun_out/feedtrace_<tag>
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-t}
mkdir -p gpurun_out/feedtrace_$T
python3 tools/gpu_feedtrace.py write ${2:-4000000} > gpurun_out/feedtrace_$T.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/feedtrace_$T -o tr -- python3 tools/gpu_feedtrace.py run 3 >> gpurun_out/feedtrace_$T.log 2>&1
tail -n 8 gpurun_out/feedtrace_$T.log
find gpurun_out/feedtrace_$T -name "*.csv"
