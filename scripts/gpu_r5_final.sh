#!/bin/bash
# Round 5 — the refresh of profiles/r05_* on the library that ships (libbhg.so sha in $O/lib.sha; bench lines carry it):
# GPU suite (result lines printed, skip reasons listed), smoke, the driver's command (20 steps) twice and the 200-step line with the
# CPU baseline, rocprofv3 --kernel-trace --stats of the default command, one-iteration timelines (CG, Neumann), what runs outside the K
# loop, FETCH_SIZE / WRITE_SIZE traffic (CG and Neumann; stamped with the sha256), SQ counters, same-box A/B lines on the measurement
# build, secondary lines.
set -u
O=gpurun_out/r5f; mkdir -p $O gpurun_out/pmc; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 1500 python -m pytest tests -m gpu -q -rPs --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
line() { python - "$1" "$2" <<'PY'
import json, sys
tag, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print('== %-30s %.1f steps/s  %.3f ms/step (min-max %s)  iter_us %s (%s) frac %s  outside %s ms  %s' % (tag, d['value'], d['ms_per_step'], ['%.3f' % v for v in d['regions']['ms_per_step_min_max']],
          '%.2f' % r['avg_launch_us'] if r.get('avg_launch_us') else None, ['%.2f' % v for v in (r.get('avg_launch_us_min_max') or [])], '%.3f' % r['frac'] if r.get('frac') else None,
          '%.3f' % d['outside_k_loop_ms'] if d.get('outside_k_loop_ms') else None, d['config']['lib'][:12]))
except Exception as e:
    print('==', tag, 'unreadable', e)
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
for i in a b; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd_20steps_$i.err > $O/bench_driver_cmd_20steps_$i.json; line driver_cmd_20steps_$i $O/bench_driver_cmd_20steps_$i.json; done
timeout 400 python bench.py 2> $O/bench_default.err > $O/bench_default.json; line default_200steps $O/bench_default.json
cd /tmp && rm -rf /tmp/prof_default && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-parity --reps 2 > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p $O/prof_default; cp /tmp/prof_default/*kernel_stats*.csv $O/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > $O/prof_default/bench_line_under_rocprof.json
for algo in cg neumann; do
  M=k_wskpl; [ $algo = neumann ] && M=k_graw
  cd /tmp && rm -rf /tmp/tr_$algo && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$algo -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 $algo fused > /tmp/tr_$algo.log 2>&1; echo "trace $algo rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$algo/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/print_iter_timeline.py $f $M | tee $O/timeline_$algo.txt
  [ -n "$f" ] && [ $algo = cg ] && python scripts/print_step_outside.py $f > $O/outside_fused.txt 2>&1
done
bash scripts/gpu_pmc5.sh 2>&1 | tail -30 | tee $O/pmc.log
cp gpurun_out/pmc/r05_pmc_traffic.json $O/ 2>/dev/null
bash scripts/gpu_pmc_sq5.sh 2>&1 | tail -14 | tee $O/pmc_sq.log
cp gpurun_out/pmc/r05_pmc_sq_default.json $O/pmc_sq_default.json 2>/dev/null
run cg_product_again
run cg_ab_defaults --ab-lib
run cg_round4_outside_of_loop --debug packed_prepare=0 --debug cg_rhs_direct=0 --upper autograd
run cg_split_k_prepare --debug packed_prepare=0
run cg_rhs_copied --debug cg_rhs_direct=0
run cg_upper_autograd --upper autograd
run cg_kpstep_launch --debug lin_first=0
run cg_upd_in_prehead --debug lin_update_in_head=0
run cg_head_last --debug headu_head_first=0
run cg_unpaired --debug xcd_pairs=0
run cg_round5_start_loop --debug lin_update_in_head=0 --debug xcd_pairs=0
run neumann_default --algo neumann --cg-iters 10
run neumann_update_launch --algo neumann --cg-iters 10 --debug neumann_vnew=0
run neumann_round4_form --algo neumann --cg-iters 10 --debug neumann_vnew=0 --debug packed_prepare=0 --upper autograd
run cg_keep_solution --keep-solution
run cg_nofuse --no-fuse
run cg_autograd_eager --hvp autograd --steps 60 --no-hvp-graph
run cg_autograd_graph_persistent --hvp autograd --steps 60
run darts --algo darts
run cg_global_ws1 --mode global
BHG_ALL_RANKS_ON_GPU0=1 timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_selflaunch_2ranks_one_gpu_gloo.err > $O/bench_selflaunch_2ranks_one_gpu_gloo.json; echo "self-launch --gpus 2 rc=$?"
timeout 400 python bench.py --algo neumann --cg-iters 10 2> $O/bench_neumann_default_full.err > $O/bench_neumann_default_full.json; line neumann_with_parity_and_cpu $O/bench_neumann_default_full.json
