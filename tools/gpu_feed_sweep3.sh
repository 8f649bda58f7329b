# feed: staging threads sweep (8 slots, lag 6, 32 MiB chunks, 16 hardware queues)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/gpu_feedtrace.py write 4000000 > gpurun_out/feedsweep3.log 2>&1
export GPU_MAX_HW_QUEUES=16 BREAKID_FEED_SLOTS=8 BREAKID_FEED_LAG=6 BREAKID_FEED_CHUNK_MB=32
for th in 4 8 12 16; do
  echo "== threads $th" >> gpurun_out/feedsweep3.log
  BREAKID_THREADS=$th timeout -k 10 120 python3 tools/gpu_feedtrace.py run 4 2>&1 | grep "feed/gpu\|rep " | tail -n 4 | cut -c1-600 >> gpurun_out/feedsweep3.log
done
grep "==\|rep 3\|rep 2" gpurun_out/feedsweep3.log
