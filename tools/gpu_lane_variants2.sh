# lanes decoder inside the streaming feed: waves per SIMD of k_bgzf_decode_shared (a rebuild each), file -> calls of the 8 M-record BAM
cd $GRAFT_REPO_ROOT/breakid_amd/csrc
for v in ${VARIANTS:-2 3 4}; do
  echo "== shared decoder: $v waves per SIMD"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -DBGZF_SHARED_WAVES=$v ${EXTRA} -c bgzf_gpu.hip -o build/bgzf_gpu.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbreakid_hip.so build/*.o -lz -pthread || exit 1
  (cd $GRAFT_REPO_ROOT && BREAKID_FEED_COPY_THREADS=${CT:-12} BREAKID_FEED_STATS=1 timeout -k 10 300 python3 tools/gpu_feedbench.py 4000000 2>&1 | grep "file -> calls\|chunks: file" | cut -c1-330 | sed -n 6,9p)
done
