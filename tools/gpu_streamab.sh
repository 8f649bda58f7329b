# k_stream with and without the side rows on one box: HIP-event time of the bench line + FETCH_SIZE / WRITE_SIZE of the launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/streamab
mkdir -p $D
for v in side noside; do
  if [ $v = noside ]; then export BREAKID_NO_SIDE=1; else unset BREAKID_NO_SIDE; fi
  timeout -k 10 200 python3 bench.py --steps 5 --warmup 1 --cpu-sample 0 --from-bam 0 > $D/b_$v.log 2> $D/b_$v.err || exit 1
  python3 -c "
import json
l=json.loads(open('$D/b_$v.log').read().strip().split('\n')[-1])
print('$v', l['ms_per_step'], l['roofline']['avg_launch_ms'], l['roofline']['frac'])"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D/raw_${v}_$c -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --from-bam 0 > $D/p.log 2>&1 || exit 1
    f=$(find $D/raw_${v}_$c -name "*counter_collection.csv" | head -n 1)
    python3 -c "
import csv
t=0;n=0
for r in csv.DictReader(open('$f')):
    if 'k_stream' in r['Kernel_Name'] and r['Counter_Name']=='$c':
        t+=float(r['Counter_Value']); n+=1
print('$v $c KiB per launch', t/max(1,n), 'launches', n)"
    rm -rf $D/raw_${v}_$c
  done
done
