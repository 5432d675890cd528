#!/bin/bash
# A/B of k_outer_all start staggering (BHG_OUTER_STAGGER) under the one-pass solver: parity first, then bench lines.
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
run base
BHG_OUTER_STAGGER=1 run oprio1
BHG_OUTER_STAGGER=2 run oprio2
BHG_OUTER_STAGGER=1 BHG_GEMM_PRIO=1 run oprio1_gprio1
BHG_OUTER_STAGGER=1 BHG_GEMM_PRIO=2 run oprio1_gprio2
run base_again
BHG_OUTER_STAGGER=1 run oprio1_again
BHG_OUTER_STAGGER=2 run oprio2_again
BHG_OUTER_STAGGER=1 BHG_GEMM_PRIO=1 run oprio1_gprio1_again
BHG_OUTER_STAGGER=1 BHG_GEMM_PRIO=2 run oprio1_gprio2_again
for v in 0 1 2; do
  rm -rf /tmp/tr_$v
  (cd /tmp && BHG_OUTER_STAGGER=1 BHG_GEMM_PRIO=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_$v.log 2>&1)
  f=$(find /tmp/tr_$v -name '*kernel_trace.csv' | head -1)
  echo "--- timeline gemm prio=$v"
  if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | grep -E "k_gemm|k_outer_all|iteration span"; else tail -3 /tmp/tr_$v.log; fi
done
