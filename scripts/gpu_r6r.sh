#!/bin/bash
# Round 6, call (r): the second-queue penalty of call (q) under GPU_MAX_HW_QUEUES (how many hardware queues ROCclr maps the streams onto).
set -u
O=gpurun_out/r6r; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("== %-34s %.1f steps/s %.4f ms iter %.2f us outside %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["outside_k_loop_ms"]))
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
for q in 1 2 4; do
  export GPU_MAX_HW_QUEUES=$q
  run hwq${q}_none
  run hwq${q}_blocking --emulate-collective blocking
  run hwq${q}_deferred --emulate-collective deferred
done
unset GPU_MAX_HW_QUEUES
export HIP_FORCE_DEV_KERNARG=1
run devkernarg_none
run devkernarg_blocking --emulate-collective blocking
unset HIP_FORCE_DEV_KERNARG
export ROC_ACTIVE_WAIT_TIMEOUT=0
run activewait0_blocking --emulate-collective blocking
