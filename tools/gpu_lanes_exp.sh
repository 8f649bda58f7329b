# lanes experiment on the GPU box: parity of the K-lane split, then bench lines for several (lanes, solo) settings
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes_of_groups" > gpurun_out/lanes_parity.log 2>&1 || { tail -n 30 gpurun_out/lanes_parity.log; exit 1; }
tail -n 2 gpurun_out/lanes_parity.log
for cfg in "2 0" "2 1" "3 1" "3 2" "4 2" "4 1"; do
  set -- $cfg
  BREAKID_GROUP_LANES=$1 BREAKID_LANE_SOLO=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/lanes_$1_$2.json 2> gpurun_out/lanes_$1_$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lanes_$1_$2.json").read().strip().splitlines()[-1])
print("lanes $1 solo $2: ms_per_step",d["ms_per_step"],{k:v for k,v in d["stage_ms_per_step"].items() if "mask" in k})
PY
done
