set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "2 0 16" "3 0 16" "4 0 16" "3 1 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 BREAKID_GROUP_LANES=$1 BREAKID_LANE_ADAPT=$2 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/lanesA_$1_$2.json 2> gpurun_out/lanesA_$1_$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/lanesA_$1_$2.json").read().strip().splitlines()[-1])
print("lanes $1 adapt $2 queues $3: ms_per_step",d["ms_per_step"],{k:v for k,v in d["stage_ms_per_step"].items() if "mask" in k or "isolated" in k or "fast" in k})
PY
done
