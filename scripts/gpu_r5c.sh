#!/bin/bash
# Round 5, call 3 — the whole GPU suite on the round's device code so far (closed-form meta-weight-net, bounded poll_beta, Neumann
# update inside k_graw, right-hand side read in place, once-per-step passes on packed operands), then same-box A/B bench lines and
# the per-dispatch list of what runs outside the K loop.
set -u
O=gpurun_out/r5c; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 1500 python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -30
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('== %-28s %.1f steps/s  %.3f ms/step (min-max %s)  iter_us %s (%s)  outside %.3f ms  parity %s' % ('$tag', d['value'], d['ms_per_step'], ['%.3f'%v for v in d['regions']['ms_per_step_min_max']], '%.2f'%r['avg_launch_us'] if r else None, ['%.2f'%v for v in (r.get('avg_launch_us_min_max') or [])], d['outside_k_loop_ms'] or 0, (d.get('parity') or {}).get('well_conditioned_variant',{}).get('vs_reference_cpu_fp32')))
except Exception as e:
    print('== $tag unreadable', e, open('$O/bench_$tag.err').read()[-1500:])
PY
}
run cg_default_20 --steps 20 --warmup 5
run cg_default_200
run cg_split_k_prepare --debug packed_prepare=0
run cg_rhs_copied --debug cg_rhs_direct=0
run cg_round4_outside --debug packed_prepare=0 --debug cg_rhs_direct=0 --upper autograd
run cg_default_200_again
run cg_default_20_again --steps 20 --warmup 5
run neumann_default --algo neumann --cg-iters 10
run neumann_round4 --algo neumann --cg-iters 10 --debug packed_prepare=0 --debug neumann_vnew=0 --upper autograd
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_iter_timeline.py $f k_wskpl | tee $O/timeline_default.txt
[ -n "$f" ] && python scripts/print_step_outside.py $f | tee $O/outside_fused.txt | tail -40
