# the feed part of the round-3 soak again, at the last commit
cd $GRAFT_REPO_ROOT
L=gpurun_out/soak_r03b.log
echo "# (at the last commit: the feed part again)" > $L
run() { echo "\$ $*" >> $L; timeout -k 10 ${T:-400} "$@" 2>&1 | grep -v amdgpu.ids | tail -n 1 >> $L; echo "done: $*"; }
run python tools/gpu_inflatefuzz.py 300 3
run python tools/gpu_feedfuzz.py 200 2720
BREAKID_FEED_PACKED_CHUNKS=1 run python tools/gpu_feedfuzz.py 80 33
run python tools/gpu_feedsoak.py 2000000 10
run python tools/gpu_fuzz.py 150 82
cat $L
