# lanes x tail: "lanes bulk tail_level"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
IFS="|"
for cfg in ${CFGS:-"2 0 4"}; do
  IFS=" "
  set -- $cfg
  echo "== lanes $1 bulk $2 tail_level $3"
  BK_SORT_TAIL_LEVEL=$3 BK_DEBUG_LANES=1 BREAKID_GROUP_LANES=$1 BREAKID_LANE_BULK=$2 GPU_MAX_HW_QUEUES=${Q:-16} timeout -k 10 200 python3 bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 > gpurun_out/lanes/x.log 2> gpurun_out/lanes/x.err || exit 1
  python3 -c "
import json,sys
l=json.loads(open('gpurun_out/lanes/x.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step']['mask_and_cluster_lanes'], l['config']['valid_clusters'])"
  grep "done after" gpurun_out/lanes/x.err | tail -$((2*$1)) | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
  IFS="|"
done
