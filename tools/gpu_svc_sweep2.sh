#!/bin/bash
# resident sort service: streams shared by the lanes x lanes against the step / stage time
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
mkdir -p gpurun_out
for cfg in "$@"; do
  IFS=, read s l <<< "$cfg"
  BREAKID_LANE_STREAMS=$s BREAKID_GROUP_LANES=$l timeout -k 10 200 python bench.py --steps 4 --warmup 1 --from-bam 0 --cpu-sample 0 > gpurun_out/sweep.log 2> gpurun_out/sweep.err || { echo "cfg $cfg failed"; tail -3 gpurun_out/sweep.err; exit 1; }
  echo "streams=$s lanes=$l: $(grep -o 'ms_per_step": [0-9.]*' gpurun_out/sweep.log | head -1) $(grep -o 'mask_and_cluster[a-z_]*", "ms": [0-9.]*' gpurun_out/sweep.log)"
done
