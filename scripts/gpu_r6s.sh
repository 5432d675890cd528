#!/bin/bash
# Round 6, call (s): bench.py's own handling of GPU_MAX_HW_QUEUES (set in-process before the HIP runtime comes up) and RCCL / gloo under one
# hardware queue: the emulated collective with and without it, the self-check at N = 1 (RCCL) and N = 2 (gloo), the global mode over RCCL at
# world size 1, two gloo ranks of the bench.
set -u
O=gpurun_out/r6s; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("== %-30s %.1f steps/s %.4f ms iter %s us | hwq %s | ranks %s" % (sys.argv[1], d["value"], d["ms_per_step"], ("%.2f" % r["avg_launch_us"]) if r.get("avg_launch_us") else None, d["config"]["gpu_max_hw_queues"], d.get("ranks_seen")))
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
run n1_default
run emulated_blocking_default --emulate-collective blocking
run emulated_blocking_hwq0 --emulate-collective blocking --hw-queues 0
run emulated_deferred_default --emulate-collective deferred
run global_ws1_rccl_hwq1 --mode global --hw-queues 1
run global_ws1_rccl_hwq0 --mode global --hw-queues 0
GPU_MAX_HW_QUEUES=1 timeout 600 python scripts/multi_gpu_selfcheck.py > $O/selfcheck_n1_rccl_hwq1.txt 2>&1; echo "selfcheck n1 rc=$?"; grep SELFCHECK $O/selfcheck_n1_rccl_hwq1.txt
GPU_MAX_HW_QUEUES=1 timeout 600 python scripts/multi_gpu_selfcheck.py --gpus 2 --backend gloo --all-on-gpu0 > $O/selfcheck_n2_gloo_hwq1.txt 2>&1; echo "selfcheck n2 rc=$?"; grep SELFCHECK $O/selfcheck_n2_gloo_hwq1.txt
BHG_ALL_RANKS_ON_GPU0=1 timeout 400 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_2ranks_gloo_default.err > $O/bench_2ranks_gloo_default.json; echo "2 ranks rc=$?"; line two_ranks_gloo_default $O/bench_2ranks_gloo_default.json
BHG_ALL_RANKS_ON_GPU0=1 timeout 400 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 --hw-queues 0 2> $O/bench_2ranks_gloo_hwq0.err > $O/bench_2ranks_gloo_hwq0.json; echo "2 ranks rc=$?"; line two_ranks_gloo_hwq0 $O/bench_2ranks_gloo_hwq0.json
timeout 600 python -m pytest tests/test_engine_shim.py tests/test_gpu_global.py -m gpu -q 2>&1 | tail -2
