# feed: slots / lag / chunk size sweep on one box (tools/gpu_feedtrace.py run, BREAKID_FEED_STATS lines)
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/gpu_feedtrace.py write 4000000 > gpurun_out/feedsweep2.log 2>&1
for cfg in "4 2 64" "8 6 64" "8 6 32" "12 10 32" "12 10 16"; do
  set -- $cfg
  echo "== slots $1 lag $2 chunk $3 MB" >> gpurun_out/feedsweep2.log
  BREAKID_FEED_SLOTS=$1 BREAKID_FEED_LAG=$2 BREAKID_FEED_CHUNK_MB=$3 timeout -k 10 120 python3 tools/gpu_feedtrace.py run 3 2>&1 | grep "feed/gpu\|rep " | tail -n 4 | cut -c1-600 >> gpurun_out/feedsweep2.log
done
grep "==\|rep 2\|rep 1" gpurun_out/feedsweep2.log
