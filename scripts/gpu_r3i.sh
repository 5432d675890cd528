#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "projected or hoisted or fused" 2>&1 | tail -2
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3i_bench_$tag.err > $O/r3i_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3i_bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-22s value %.1f steps/s ms/step %.3f iter_us %.1f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3i_bench_$tag.err").read()[-1500:])
PY
}
for rep in a b; do
BHG_PROJ_GRAW_SPLIT=0 run graw_one_wg_$rep
run graw_split_$rep
BHG_PROJ_GRAW_SPLIT=0 run neumann_graw_one_wg_$rep --algo neumann --cg-iters 10
run neumann_graw_split_$rep --algo neumann --cg-iters 10
done
