set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
D=gpurun_out/prof_r04
mkdir -p $D
export GPU_MAX_HW_QUEUES=16
python bench.py > $D/bench_final.log 2> $D/bench_final.err
echo bench done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/ks_raw -o ks -- python3 bench.py --steps 3 --warmup 1 --cpu-sample 0 --from-bam 0 > $D/ks.log 2>&1
cp $(find $D/ks_raw -name "*kernel_stats.csv" | head -n 1) $D/ks_kernel_stats.csv
echo stats done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/f_raw -o fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --from-bam 0 > $D/fetch.log 2>&1
cp $(find $D/f_raw -name "*counter_collection.csv" | head -n 1) $D/fetch_counter_collection.csv
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/w_raw -o write -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --from-bam 0 > $D/write.log 2>&1
cp $(find $D/w_raw -name "*counter_collection.csv" | head -n 1) $D/write_counter_collection.csv
echo write done
# the feed: file -> calls on the 8 M-record BAM, kernel totals of the GPU feed, inflate alone
timeout -k 10 600 python3 tools/gpu_feedbench.py 4000000 > $D/feedbench.log 2>&1
python3 tools/gpu_feedtrace.py write 4000000 > $D/feedtrace.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/ft_raw -o ft -- python3 tools/gpu_feedtrace.py run 3 >> $D/feedtrace.log 2>&1
cp $(find $D/ft_raw -name "*kernel_stats.csv" | head -n 1) $D/feed_kernel_stats.csv
INFLATE_LEVELS=1,6,0 timeout -k 10 300 python3 tools/gpu_inflatebench.py 1000000 > $D/inflatebench.log 2>&1
echo feed done
rm -rf $D/ks_raw $D/f_raw $D/w_raw $D/ft_raw
ls -la $D
