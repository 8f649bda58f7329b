# decoder statistics (symbols, passes, steps per block): a -DBGZF_STATS build of bgzf_gpu.hip on the box, then the normal one again
set -e
cd $GRAFT_REPO_ROOT/breakid_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fno-fast-math -DBGZF_STATS -c bgzf_gpu.hip -o build/bgzf_gpu.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbreakid_hip.so build/*.o -lz -pthread
cd $GRAFT_REPO_ROOT
BK_DEBUG=bgzf timeout -k 10 300 python3 tools/gpu_inflatebench.py ${1:-300000} 2>&1 | grep -v "amdgpu.ids\|^\[bgzf\] "
