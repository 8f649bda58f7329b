# lanes experiment on the GPU box: bench lines for several (lanes, solo, weight exponent) settings
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in ${LANE_CFGS:-"1 0 2" "2 0 2" "3 0 2" "4 0 2" "2 0 1" "3 0 1"}; do
  set -- $cfg
  BREAKID_GROUP_LANES=$1 BREAKID_LANE_SOLO=$2 BREAKID_LANE_WEIGHT_EXP=$3 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/lanes_$1_$2_$3.json 2> gpurun_out/lanes_$1_$2_$3.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lanes_$1_$2_$3.json").read().strip().splitlines()[-1])
print("lanes $1 solo $2 wexp $3: ms_per_step",d["ms_per_step"],{k:v for k,v in d["stage_ms_per_step"].items() if "mask" in k or "isolated" in k or "fast" in k})
PY
done
