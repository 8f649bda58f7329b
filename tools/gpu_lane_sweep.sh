# lanes experiment: bench step time for several lane plans: "lanes bulk queues persist_tiles" (0 = per-level launches only)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
IFS="|"
for cfg in ${CFGS:-"2 0 16 0|2 0 16 256"}; do
  IFS=" "
  set -- $cfg
  echo "== lanes $1 bulk $2 queues $3 persist_tiles $4"
  tag=l$1b$2q$3p$4
  if [ "$4" = "0" ]; then export BK_SORT_NO_PERSIST=1; else unset BK_SORT_NO_PERSIST; export BK_SORT_PERSIST_TILES=$4; fi
  BK_DEBUG_LANES=1 BREAKID_GROUP_LANES=$1 BREAKID_LANE_BULK=$2 GPU_MAX_HW_QUEUES=$3 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/lanes/$tag.log 2> gpurun_out/lanes/$tag.err || exit 1
  python3 -c "
import json,sys
l=json.loads(open('gpurun_out/lanes/$tag.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step']['mask_and_cluster_lanes'], l['config']['valid_clusters'])"
  grep "done after" gpurun_out/lanes/$tag.err | tail -$((2*$1)) | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
  IFS="|"
done
