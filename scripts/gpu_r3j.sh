#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "projected or hoisted or fused or structured" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_cfg2_goldens.py -m gpu -q 2>&1 | tail -2
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3j_bench_$tag.err > $O/r3j_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3j_bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-22s value %.1f steps/s ms/step %.3f iter_us %.1f outside_ms %.3f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3j_bench_$tag.err").read()[-1500:])
PY
}
run cg_a; run cg_b
run neumann_a --algo neumann --cg-iters 10
