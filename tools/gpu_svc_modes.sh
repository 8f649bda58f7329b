#!/bin/bash
export GPU_MAX_HW_QUEUES=16 BK_DEBUG_SVC=1 BREAKID_SVC_TIMEOUT_MS=1500
for m in 0; do
  echo "== mode $m"
  BREAKID_SVC_MODE=$m timeout -k 10 120 python -m pytest tests/test_gpu_parity.py -q -s -k "std_sort_emulation and (random_ties or all_equal or killer_lds_large)" > gpurun_out/svc_mode_$m.log 2>&1
  grep "svc\]\|passed\|failed" gpurun_out/svc_mode_$m.log | cut -c1-700
done
