#!/bin/bash
# round 4, call c: step length requested before the small slices' loop (k_graw), fatter update blocks (k_pstep<U>),
# output stores nt / sc1 / sc0 sc1 (libbhg_wt{1,2,3}.so) — parity subset, then same-box A/B, then stamps
set -u
O=gpurun_out/r4c; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or without_a_solution" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_subset.log | tail -5
timeout 600 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -x > $O/pytest_goldens.log 2>&1; echo "pytest goldens rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_goldens.log | tail -4
run() { tag=$1; shift; timeout 300 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-22s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
L=$GRAFT_REPO_ROOT/betty_amd/csrc
run default
run pstep_u1 --debug pstep_unroll=1
run pstep_u2 --debug pstep_unroll=2
BHG_LIB=$L/libbhg_wt1.so run wt1_nt
BHG_LIB=$L/libbhg_wt2.so run wt2_sc1
BHG_LIB=$L/libbhg_wt3.so run wt3_sc0sc1
run default_again
run neumann --algo neumann --cg-iters 10
BHG_LIB=$L/libbhg_wt2.so run neumann_wt2 --algo neumann --cg-iters 10
BHG_LIB=$L/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_default.txt
BHG_LIB=$L/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py pstep_unroll=1 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_u1.txt
