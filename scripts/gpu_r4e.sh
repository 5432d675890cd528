#!/bin/bash
# round 4, call e: k_graw with 64-column tiles (k_graw64), tiles first
set -u
O=gpurun_out/r4e; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or without_a_solution" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_subset.log | tail -5
timeout 600 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -x > $O/pytest_goldens.log 2>&1; echo "pytest goldens rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_goldens.log | tail -4
run() { tag=$1; shift; timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-22s %.1f steps/s  %.3f ms/step  iter_us %s  parity %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us'), (d.get('parity') or {}).get('ok')))" 2>&1 | tail -1; }
L=$GRAFT_REPO_ROOT/betty_amd/csrc
run default
run cols32 --no-parity --debug graw_cols=32
run rnew0 --no-parity --debug rnew_in_graw=0
run cols32_rnew0 --no-parity --debug graw_cols=32 --debug rnew_in_graw=0
run default_again --no-parity
run cols32_again --no-parity --debug graw_cols=32
run neumann --no-parity --algo neumann --cg-iters 10
run neumann_cols32 --no-parity --algo neumann --cg-iters 10 --debug graw_cols=32
BHG_LIB=$L/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_default.txt
