#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "persistent_graphs or hip_graph" > $O/r3h_tests.log 2>&1; echo "rc=$?"; grep -vE "^Extension" $O/r3h_tests.log | tail -25
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3h_bench_$tag.err > $O/r3h_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3h_bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-26s value %.1f steps/s ms/step %.3f" % ("$tag", d["value"], d["ms_per_step"]))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3h_bench_$tag.err").read()[-2500:])
PY
}
run autograd_persistent --hvp autograd --steps 60
run autograd_per_solve --hvp autograd --steps 60 --hvp-graph solve
run autograd_eager --hvp autograd --steps 60 --no-hvp-graph
run neumann_autograd_persistent --hvp autograd --steps 60 --algo neumann --cg-iters 10
run neumann_autograd_eager --hvp autograd --steps 60 --algo neumann --cg-iters 10 --no-hvp-graph
