#!/bin/bash
# Round 5, call 4 — the whole GPU suite on the product libbhg.so + the measurement build libbhg_ab.so (every-arm tests), then the
# product's bench lines next to the measurement build at its defaults (same form, same speed) and the Neumann line.
set -u
O=gpurun_out/r5d; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -30
grep -E "withheld beta|packed prepare|update inside k_graw|cfg5 as named" $O/pytest_gpu_full.log | head -20
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('== %-28s %.1f steps/s  %.3f ms/step (min-max %s)  iter_us %s (%s)  outside %.3f ms  parity %s  %s' % ('$tag', d['value'], d['ms_per_step'], ['%.3f'%v for v in d['regions']['ms_per_step_min_max']], '%.2f'%r['avg_launch_us'] if r else None, ['%.2f'%v for v in (r.get('avg_launch_us_min_max') or [])], d['outside_k_loop_ms'] or 0, (d.get('parity') or {}).get('well_conditioned_variant',{}).get('vs_reference_cpu_fp32'), d['config']['lib'][:12]))
except Exception as e:
    print('== $tag unreadable', e, open('$O/bench_$tag.err').read()[-1500:])
PY
}
run cg_product_20 --steps 20 --warmup 5
run cg_product_200
run cg_ab_defaults_200 --ab-lib
run cg_product_200_again
run neumann_product --algo neumann --cg-iters 10
run neumann_ab_defaults --algo neumann --cg-iters 10 --ab-lib
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee $O/smoke.log
