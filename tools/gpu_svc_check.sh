#!/bin/bash
# first checks of the resident sort service on the GPU box: unit sorts vs libstdc++, fuzz, lanes, a short bench
set -o pipefail
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=16 BK_DEBUG_SVC=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -s -k "std_sort_emulation or heap_beyond or wgs_shape_device_resident" > gpurun_out/svc_units.log 2>&1
echo "units rc=$?" | tee -a gpurun_out/svc_units.log
tail -5 gpurun_out/svc_units.log
timeout -k 10 400 python tools/gpu_sortfuzz.py ${1:-60} 11 > gpurun_out/svc_fuzz.log 2>&1
echo "fuzz rc=$?"; tail -3 gpurun_out/svc_fuzz.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lanes_of_groups" > gpurun_out/svc_lanes.log 2>&1
echo "lanes rc=$?"; tail -5 gpurun_out/svc_lanes.log
BK_DEBUG_LANES=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --from-bam 0 --cpu-sample 0 > gpurun_out/svc_bench.log 2> gpurun_out/svc_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/svc_bench.log | cut -c1-1500; tail -5 gpurun_out/svc_bench.err
