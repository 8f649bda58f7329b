#!/bin/bash
# default-mode (AHC = src/util_cluster.cc) bench lines: configs[0] size, configs[3] (panel, 6.8 M records), a 20 M-record WGS shape and
# configs[1] itself; kernel summary of the 20 M run
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=16 BK_DEBUG=ahc
O=gpurun_out/ahc
mkdir -p $O
timeout -k 10 200 python bench.py --mode ahc --records 1000000 --steps 5 --warmup 1 --from-bam 0 --cpu-sample 0 > $O/bench_1M_ahc.json 2> $O/bench_1M_ahc.err; echo "1M rc=$?"
timeout -k 10 300 python bench.py --mode ahc --workload panel --steps 3 --warmup 1 --from-bam 0 --cpu-sample 0 > $O/bench_panel_ahc.json 2> $O/bench_panel_ahc.err; echo "panel rc=$?"
timeout -k 10 300 python bench.py --mode ahc --records 20000000 --steps 3 --warmup 1 --from-bam 0 --cpu-sample 0 > $O/bench_20M_ahc.json 2> $O/bench_20M_ahc.err; echo "20M rc=$?"
timeout -k 10 400 python bench.py --mode ahc --records 620000000 --steps 1 --warmup 1 --from-bam 0 --cpu-sample 0 > $O/bench_620M_ahc.json 2> $O/bench_620M_ahc.err; echo "620M rc=$?"
for f in 1M panel 20M 620M; do echo "== $f: $(grep -o 'ms_per_step": [0-9.]*' $O/bench_${f}_ahc.json | head -1) $(grep -o '"stage": "ahc_cluster[a-z_]*", "ms": [0-9.]*' $O/bench_${f}_ahc.json) $(grep -o '"stage": "mask_and_cluster[a-z_]*", "ms": [0-9.]*' $O/bench_${f}_ahc.json) $(tail -1 $O/bench_${f}_ahc.err | cut -c1-300)"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ahc20 -- python3 bench.py --mode ahc --records 20000000 --steps 3 --warmup 1 --from-bam 0 --cpu-sample 0 > $O/prof_bench.log 2> $O/prof_bench.err; echo "prof rc=$?"
cp $(find $O/prof -name "*kernel_stats.csv" | head -n 1) $O/kernel_stats_20M_ahc.csv && rm -rf $O/prof
head -12 $O/kernel_stats_20M_ahc.csv | cut -c1-160
