#!/bin/bash
# Round 6, call (p): the step on the process's default (null) stream vs on a stream of its own — same box, twice each.
set -u
O=gpurun_out/r6p; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("== %-22s %.1f steps/s %.4f ms iter %.2f us outside %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], r["avg_launch_us"], d["outside_k_loop_ms"]))
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
for rep in 1 2; do
  run default_stream_$rep
  run own_stream_$rep --own-stream
done
