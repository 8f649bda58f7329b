# A/B of environment switches on the bench step: VARIANTS="A=1|B=2 C=3|..." (one bench run each, STEPS steps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
IFS="|"
for v in ${VARIANTS:-"BK_NOP=1"}; do
  IFS=" "
  echo "== $v"
  env $v BK_DEBUG=lanes timeout -k 10 300 python3 bench.py --steps ${STEPS:-10} --warmup 1 --cpu-sample 0 --from-bam 0 > gpurun_out/ab/b.log 2> gpurun_out/ab/b.err || { tail -5 gpurun_out/ab/b.err; exit 1; }
  python3 -c "
import json
l=json.loads(open('gpurun_out/ab/b.log').read().strip().split('\n')[-1])
print(l['ms_per_step'], l['stage_ms_per_step'].get('mask_and_cluster_lanes'), l['config']['valid_clusters'])"
  grep "done after" gpurun_out/ab/b.err | tail -8 | sed 's/\[lanes\] lane //; s/ done after//' | tr '\n' ' '; echo
  IFS="|"
done
