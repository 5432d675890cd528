#!/bin/bash
# Round 6, call (q): what a per-step collective's second hardware queue costs the K loop, emulated at N = 1 (a tiny kernel on another
# stream after every step, fenced at once / at the next hop).
set -u
O=gpurun_out/r6q; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("== %-22s %.1f steps/s %.4f ms iter %.2f us outside %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["outside_k_loop_ms"]))
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
for rep in 1 2; do
  run none_$rep
  run blocking_$rep --emulate-collective blocking
  run deferred_$rep --emulate-collective deferred
done
