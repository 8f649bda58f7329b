# lanes decoder: window / checkpoint variants, each a rebuild of bgzf_gpu.hip on the box; 570 MB inflate at zlib levels 1 and 6
cd $GRAFT_REPO_ROOT/breakid_amd/csrc
for v in ${VARIANTS:-"1024 384" "512 256" "512 192" "2048 512" "1024 256"}; do
  set -- $v
  echo "== part $1 check $2"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -DLANE_PART_BITS=$1 -DLANE_CHECK_BITS=$2 -c bgzf_gpu.hip -o build/bgzf_gpu.o || exit 1
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbreakid_hip.so build/*.o -lz -pthread || exit 1
  (cd $GRAFT_REPO_ROOT && INFLATE_LEVELS=1,6 timeout -k 10 300 python3 tools/gpu_inflatebench.py ${PAIRS:-1000000} 2>&1 | grep level)
done
