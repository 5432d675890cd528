#!/bin/bash
# Round 6, call (t): runtime variables against the single-stream K loop (N = 1, no collective): does any of them move the launch cost?
set -u
O=gpurun_out/r6t; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
    print("== %-34s %.1f steps/s %.4f ms iter %.2f us outside %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["outside_k_loop_ms"]))
except Exception as e:
    print("==", sys.argv[1], "unreadable", e)
PY
}
run() { tag=$1; shift; timeout 300 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
run baseline_1
HSA_ENABLE_INTERRUPT=0 run hsa_enable_interrupt_0
AMD_DIRECT_DISPATCH=0 run amd_direct_dispatch_0
HIP_FORCE_DEV_KERNARG=0 run hip_force_dev_kernarg_0
GPU_STREAMOPS_CP_WAIT=1 run gpu_streamops_cp_wait_1
ROC_AQL_QUEUE_SIZE=1024 run roc_aql_queue_size_1024
HSA_ENABLE_SDMA=0 run hsa_enable_sdma_0
DEBUG_CLR_USE_STDMUTEX_IN_AMD_MONITOR=1 run clr_stdmutex
run baseline_2
