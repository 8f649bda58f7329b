#!/bin/bash
export GPU_MAX_HW_QUEUES=16 BK_DEBUG_SVC=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -s -k "std_sort_emulation" > gpurun_out/svc_units.log 2>&1
echo "units rc=$?"; grep "svc\]\|passed\|failed" gpurun_out/svc_units.log | cut -c1-400
