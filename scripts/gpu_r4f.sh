#!/bin/bash
# round 4, call f: the chain's first product by linearity with the update launch inside it (k_wskpl, debug key lin_first)
set -u
O=gpurun_out/r4f; mkdir -p $O; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 120 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-22s %.1f steps/s  %.3f ms/step  iter_us %s  parity %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us'), (d.get('parity') or {}).get('ok')))" 2>&1 | tail -1; tail -2 $O/bench_$tag.err | grep -v amdgpu.ids; }
L=$GRAFT_REPO_ROOT/betty_amd/csrc
run default
run upd_in_first --no-parity --debug lin_update_next=0
run lin0 --no-parity --debug lin_first=0
run default_again --no-parity
run upd_in_first_again --no-parity --debug lin_update_next=0
run neumann --no-parity --algo neumann --cg-iters 10
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or without_a_solution" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_subset.log | tail -4
timeout 300 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -x > $O/pytest_goldens.log 2>&1; echo "pytest goldens rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_goldens.log | tail -4
cd /tmp && rm -rf /tmp/tr_d && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_d -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_d.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_d/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_iter_timeline.py $f k_wskpl | tee $O/timeline_default.txt
BHG_LIB=$L/libbhg_stamps.so timeout 100 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | grep -v "k_wskpl" | tee $O/stamps_default.txt
