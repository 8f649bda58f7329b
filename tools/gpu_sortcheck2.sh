# exactness of the sort emulation after a change (unit cases, fuzz, heap-branch goldens, lanes), then the bench step with and without it
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sortcheck
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "std_sort or heap_beyond or lanes_of_groups or earlier_statements or wgs_shape_device" > gpurun_out/sortcheck/parity.log 2>&1 || { tail -30 gpurun_out/sortcheck/parity.log; exit 1; }
tail -2 gpurun_out/sortcheck/parity.log
timeout -k 10 300 python tools/gpu_sortfuzz.py 400 11 2>&1 | tail -2 || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_big_golden.py -x -q -k "stages" 2>&1 | tail -2 || exit 1
for v in "BK_SORT_NO_TAIL=1" "BK_SORT_TAIL_LEVEL=0" "BK_SORT_TAIL_LEVEL=2" "BK_SORT_TAIL_LEVEL=4" ; do
  echo "== $v"
  env $v BK_DEBUG_LANES=1 timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/sortcheck/b.log 2> gpurun_out/sortcheck/b.err || { tail -5 gpurun_out/sortcheck/b.err; exit 1; }
  python3 -c "
import json
l=json.loads(open('gpurun_out/sortcheck/b.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step'].get('mask_and_cluster_lanes'), l['config']['valid_clusters'])"
  grep "done after" gpurun_out/sortcheck/b.err | tail -4 | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
done
