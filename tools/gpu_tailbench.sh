cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sortcheck
for v in ${VARIANTS:-"BK_SORT_NO_TAIL=1" "BK_SORT_TAIL_LEVEL=0" "BK_SORT_TAIL_LEVEL=4"} ; do
  echo "== $v"
  env $v BK_DEBUG_LANES=1 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/sortcheck/b.log 2> gpurun_out/sortcheck/b.err || { tail -5 gpurun_out/sortcheck/b.err; exit 1; }
  python3 -c "
import json
l=json.loads(open('gpurun_out/sortcheck/b.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step'].get('mask_and_cluster_lanes'), l['config']['valid_clusters'])"
  grep "done after" gpurun_out/sortcheck/b.err | tail -4 | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
done
