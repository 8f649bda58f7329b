# sort-emulation check on the GPU box: unit cases, big goldens, fuzz, then a debug bench and a plain bench
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-x}
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "std_sort or two_lanes" > gpurun_out/sc_${T}_units.log 2>&1
python -m pytest tests/test_gpu_big_golden.py -x -q -m gpu > gpurun_out/sc_${T}_big.log 2>&1
timeout -k 10 300 python tools/gpu_sortfuzz.py 60 11 > gpurun_out/sc_${T}_fuzz.log 2>&1
BK_DEBUG=sort BREAKID_GROUP_LANES=1 timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-sample 0 > gpurun_out/sc_${T}_dbg.json 2> gpurun_out/sc_${T}_dbg.err
timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/sc_${T}_bench.json 2> gpurun_out/sc_${T}_bench.err
for f in units big fuzz; do tail -n 2 gpurun_out/sc_${T}_$f.log; done
python - <<PY
import json
d=json.loads(open("gpurun_out/sc_${T}_bench.json").read().strip().splitlines()[-1])
print("ms_per_step",d["ms_per_step"],d["stage_ms_per_step"])
PY
