#!/bin/bash
# Round 5, call 2 — first GPU run of the round's new device code: closed-form meta-weight-net (bhg_mwn), bounded poll_beta,
# the projected Neumann solver's update inside k_graw (six launches), bench.py with repeated regions; cfg-5 forward-over-reverse probe.
set -u
O=gpurun_out/r5b; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cfg2_goldens.py -m gpu -x -q --durations=8 \
  -k "mwn or weight_net or withheld or hoisted_and_projected or projected_solvers_edge or structured or every_arm or fused_solver_matches or metric_workload" \
  > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_new.log | tail -25
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
run cg_default_20_again --steps 20 --warmup 5
run cg_default_200
run cg_upper_autograd_200 --upper autograd
run neumann_default --algo neumann --cg-iters 10
run neumann_update_launch --algo neumann --cg-iters 10 --debug neumann_vnew=0
run neumann_upper_autograd --algo neumann --cg-iters 10 --upper autograd --debug neumann_vnew=0
run cg_default_200_again
timeout 200 python scripts/cfg5_modes.py fwdrev 2 2>&1 | grep "^\[" | tee $O/cfg5_fwdrev.txt; echo "cfg5 fwdrev rc=$?"
