#!/bin/bash
# resident sort service: workgroup counts and lanes of groups against the stage time (bench, 3 steps each)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
mkdir -p gpurun_out
for cfg in "$@"; do
  IFS=, read w n l <<< "$cfg"
  BREAKID_SVC_WIDE=$w BREAKID_SVC_NARROW=$n BREAKID_GROUP_LANES=$l BK_DEBUG_SVC=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --from-bam 0 --cpu-sample 0 > gpurun_out/sweep.log 2> gpurun_out/sweep.err || { echo "cfg $cfg failed"; tail -3 gpurun_out/sweep.err; exit 1; }
  echo "wide=$w narrow=$n lanes=$l: $(grep -o 'ms_per_step": [0-9.]*' gpurun_out/sweep.log) $(grep -o 'mask_and_cluster[a-z_]*", "ms": [0-9.]*' gpurun_out/sweep.log)"
  grep "svc\]   [wn]" gpurun_out/sweep.err | grep -v "by last state" | tail -2 | cut -c1-330
done
