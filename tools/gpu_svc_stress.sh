#!/bin/bash
# the bench step with the resident sort service N times in fresh processes: does a task ever fail?
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-16}
mkdir -p gpurun_out
if [ -n "$DBG" ]; then export BK_DEBUG=svc; fi
for i in $(seq 1 ${1:-6}); do
  BREAKID_GROUP_LANES=${2:-12} timeout -k 10 200 python bench.py --steps 6 --warmup 1 --from-bam 0 --cpu-sample 0 > gpurun_out/stress.log 2> gpurun_out/stress.err
  rc=$?
  echo "run $i rc=$rc $(grep -o 'ms_per_step": [0-9.]*' gpurun_out/stress.log | head -1) $(grep -o 'valid_clusters": [0-9]*' gpurun_out/stress.log)"
  if [ $rc -ne 0 ]; then grep "failing task\|error -\|svc\] tasks\|svc\]   wide workgroups (\|svc\]   narrow workgroups (" gpurun_out/stress.err | cut -c1-700 | tail -8; cp gpurun_out/stress.err gpurun_out/stress_fail_$i.err; fi
done
