#!/bin/bash
# A/B of the k_outer_all instances (BHG_OUTER_NO_PRE) under the one-pass solver: parity first, then bench lines.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or structured_goldens or cfg2" 2>&1 | tail -3
run() { tag=$1; shift
  timeout 300 python bench.py --steps 100 --cpu-steps 0 --no-kernel-timing "$@" 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-28s value %.1f steps/s  ms/step %.3f" % ("$tag", d["value"], d["ms_per_step"]))
except Exception as e:
    print("== $tag bench failed:", e); print(open("gpurun_out/bench_$tag.err").read()[-800:])
PY
}
BHG_OUTER_NO_PRE=1 run nopre
run pre
BHG_OUTER_NO_PRE=1 run nopre_again
run pre_again
BHG_OUTER_NO_PRE=1 run nopre_neumann --algo neumann
run pre_neumann --algo neumann
BHG_OUTER_NO_PRE=1 run nopre_neumann_again --algo neumann
run pre_neumann_again --algo neumann
for v in nopre pre; do
  rm -rf /tmp/tr_$v
  if [ $v = nopre ]; then export BHG_OUTER_NO_PRE=1; else unset BHG_OUTER_NO_PRE; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_$v.log 2>&1)
  f=$(find /tmp/tr_$v -name '*kernel_trace.csv' | head -1)
  echo "--- timeline $v"
  if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | grep -E "k_outer_all|iteration span"; else tail -3 /tmp/tr_$v.log; fi
done
