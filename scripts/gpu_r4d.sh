#!/bin/bash
# round 4, call d: one-burst small-slice blocks in k_graw, the residual step applied by k_graw's tiles (rnew), k_pstep<U>
set -u
O=gpurun_out/r4d; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or without_a_solution" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_subset.log | tail -5
timeout 600 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -x > $O/pytest_goldens.log 2>&1; echo "pytest goldens rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_goldens.log | tail -4
run() { tag=$1; shift; timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-22s %.1f steps/s  %.3f ms/step  iter_us %s  parity %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us'), (d.get('parity') or {}).get('ok')))" 2>&1 | tail -1; }
L=$GRAFT_REPO_ROOT/betty_amd/csrc
run default
run rnew0 --no-parity --debug rnew_in_graw=0
run pstep_u2 --no-parity --debug pstep_unroll=2
run pstep_u1 --no-parity --debug pstep_unroll=1
run default_again --no-parity
run rnew0_again --no-parity --debug rnew_in_graw=0
run neumann --no-parity --algo neumann --cg-iters 10
BHG_LIB=$L/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_default.txt
