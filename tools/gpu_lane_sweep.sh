# lanes experiment: bench step time for several lane plans (BREAKID_GROUP_LANES / BREAKID_LANE_BULK / GPU_MAX_HW_QUEUES)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
for cfg in ${CFGS:-"2 0 16" "3 1 16" "5 1 32" "7 1 32" "7 1 64" "9 1 64"}; do
  set -- $cfg
  echo "== lanes $1 bulk $2 queues $3"
  BK_DEBUG_LANES=1 BREAKID_GROUP_LANES=$1 BREAKID_LANE_BULK=$2 GPU_MAX_HW_QUEUES=$3 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/lanes/l$1b$2q$3.log 2> gpurun_out/lanes/l$1b$2q$3.err || exit 1
  python3 -c "
import json,sys
l=json.loads(open('gpurun_out/lanes/l$1b$2q$3.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step']['mask_and_cluster_lanes'], l['config']['valid_clusters'])"
  grep "done after" gpurun_out/lanes/l$1b$2q$3.err | tail -$((2*$1)) | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
done
