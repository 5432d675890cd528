#!/bin/bash
# round 4, call a: packed chain (k_wskp) — parity of every arm, same-box A/B of the packed / unpacked forms, one timeline
set -u
mkdir -p gpurun_out/r4a; export TMPDIR=/tmp
[ -x build_probes/chain_probe ] && timeout 120 ./build_probes/chain_probe > gpurun_out/r4a/chain_probe.txt 2>&1
O=gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or without_a_solution or wsk" > $O/pytest_subset.log 2>&1; echo "pytest subset rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_subset.log | tail -30
timeout 900 python -m pytest tests/test_cfg2_goldens.py tests/test_gpu_global.py -m gpu -q -x -s > $O/pytest_goldens.log 2>&1; echo "pytest goldens rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_goldens.log | tail -12
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-22s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run default
run unpacked --debug packed_chain=0
run gram_launch --debug packed_gram=0
run depth3 --debug packed_depth=3
run default_again
run graw_v1 --debug graw_v2=0
run pstep_v1 --debug pstep_v2=0
run ragged_always --debug wskp_ragged=2
run ragged_never --debug wskp_ragged=0

run neumann --algo neumann --cg-iters 10
run neumann_unpacked --algo neumann --cg-iters 10 --debug packed_chain=0
BHG_LIB=$GRAFT_REPO_ROOT/betty_amd/csrc/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_default.txt
for arm in default unpacked; do
  extra=""; [ $arm = unpacked ] && extra="packed_chain=0"
  cd /tmp && rm -rf /tmp/tr_$arm && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$arm -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused $extra > /tmp/tr_$arm.log 2>&1; echo "trace $arm rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$arm/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && M=k_pstep; [ $arm = unpacked ] && M=k_proj_step; python scripts/print_iter_timeline.py $f "$M" | tee $O/timeline_$arm.txt
done
