#!/bin/bash
# Round 5, call 14 — the update blocks in the head launch (k_headu): solver-form tests, every-arm goldens, then same-box A/B lines.
set -u
O=gpurun_out/r5m; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cfg2_goldens.py -m gpu -x -q -rP --durations=5 -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or every_arm or metric_configuration or withheld or cfg2_metric_workload or fused_solver_full_size" > $O/pytest_forms.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error" $O/pytest_forms.log | tail -6
grep -E "\[3072, 2048, 1536, 384, 10\] K=20 full( |-upd)" $O/pytest_forms.log | head
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
run cg_upd_in_prehead --debug lin_update_in_head=0
run cg_product_200_again
run cg_upd_in_prehead_again --debug lin_update_in_head=0
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_iter_timeline.py $f k_wskpl | tee $O/timeline_cg.txt
