#!/bin/bash
# Round 5, call 16 — column tiles in XCD pairs (strip tiles find their weight columns in the L2 of their neighbours), head-first order
# of k_headu as an arm: solver-form tests + every-arm goldens, same-box A/B lines, FETCH/WRITE_SIZE of the CG iteration.
set -u
O=gpurun_out/r5n; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cfg2_goldens.py -m gpu -x -q -rP --durations=5 -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or every_arm or metric_configuration or withheld or cfg2_metric_workload or fused_solver_full_size or packed" > $O/pytest_forms.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_forms.log | tail -6
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('== %-28s %.1f steps/s  %.3f ms/step  iter_us %s (%s) frac %.3f outside %.3f ms  parity %s  %s' % ('$tag', d['value'], d['ms_per_step'], '%.2f'%r['avg_launch_us'], ['%.2f'%v for v in (r.get('avg_launch_us_min_max') or [])], r['frac'], d['outside_k_loop_ms'] or 0, (d.get('parity') or {}).get('well_conditioned_variant',{}).get('vs_reference_cpu_fp32'), d['config']['lib'][:12]))
except Exception as e:
    print('== $tag unreadable', e, open('$O/bench_$tag.err').read()[-1500:])
PY
}
run cg_product_20 --steps 20 --warmup 5
run cg_product_200
run cg_ab_defaults --ab-lib
run cg_unpaired --debug xcd_pairs=0
run cg_head_first --debug headu_head_first=1
run cg_product_200_again
run cg_unpaired_again --debug xcd_pairs=0
run neumann_product --algo neumann --cg-iters 10
run neumann_unpaired --algo neumann --cg-iters 10 --debug xcd_pairs=0
ALGOS=cg bash scripts/gpu_pmc5.sh > $O/pmc.log 2>&1; tail -12 $O/pmc.log; cp gpurun_out/pmc/r05_pmc_traffic.json $O/ 2>/dev/null
