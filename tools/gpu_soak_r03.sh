# round-3 soak at the final code (one gpurun call, ~15 minutes): commands and their last lines into gpurun_out/soak_r03.log
cd $GRAFT_REPO_ROOT
L=gpurun_out/soak_r03.log
echo "# Round-3 soak runs on the GPU box (final code); commands and their last lines" > $L
run() { echo "\$ $*" >> $L; timeout -k 10 ${T:-600} "$@" 2>&1 | grep -v amdgpu.ids | tail -n ${N:-1} >> $L; echo "done: $*"; }
run python tools/gpu_inflatefuzz.py 400 1
run python tools/gpu_inflatefuzz.py 400 2
run python tools/gpu_feedfuzz.py 300 2719
BREAKID_FEED_PACKED_CHUNKS=1 run python tools/gpu_feedfuzz.py 100 32
run python tools/gpu_feedsoak.py 2000000 15
run python tools/gpu_sortfuzz.py 800 41
run python tools/gpu_sortfuzz.py 800 42
run python tools/gpu_sortsoak.py 80
run python tools/gpu_fuzz.py 300 81
run python tools/gpu_determinism.py 620000000 40
cat $L
