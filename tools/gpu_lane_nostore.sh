# experiment: what the stores of the writing walk cost (results are wrong without them; only the times are read)
cd $GRAFT_REPO_ROOT/breakid_amd/csrc
for v in "-DBGZF_STATS" "-DBGZF_STATS -DLANE_EXPERIMENT_NO_TOKENS" "-DBGZF_STATS -DLANE_EXPERIMENT_NO_LITERALS" "-DBGZF_STATS -DLANE_EXPERIMENT_NO_LITERALS -DLANE_EXPERIMENT_NO_TOKENS"; do
  echo "== $v"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math $v -c bgzf_gpu.hip -o build/bgzf_gpu.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbreakid_hip.so build/*.o -lz -pthread || exit 1
  (cd $GRAFT_REPO_ROOT && BK_BGZF_STATS=1 INFLATE_LEVELS=6 timeout -k 10 300 python3 tools/gpu_inflatebench.py ${PAIRS:-300000} 2>&1 | grep "level\|bgzf lanes" | tail -2 | cut -c1-330)
done
