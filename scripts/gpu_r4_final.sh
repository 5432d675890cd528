#!/bin/bash
# Round 4, final refresh of profiles/r04_* for the shipped libbhg.so: GPU suite, smoke,
# default bench line (parity + CPU baseline), rocprofv3 kernel stats of the same command, one-iteration timeline, same-box A/B lines,
# stamps, PMC traffic and SQ counters (stamped with the library's sha256).
set -u
mkdir -p gpurun_out/r4 gpurun_out/pmc; export TMPDIR=/tmp
O=gpurun_out/r4
timeout 1200 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee $O/smoke.log
timeout 600 python bench.py 2> $O/bench_default.err > $O/bench_default.json; tail -c 600 $O/bench_default.json; echo
cd /tmp && rm -rf /tmp/prof_default && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-parity > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p $O/prof_default; cp /tmp/prof_default/*kernel_stats*.csv $O/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > $O/prof_default/bench_line_under_rocprof.json
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r4/prof_default/*kernel_stats*.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    for r in rows[:12]:
        print(f'{r["Name"][:84]:84s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} pct={r["Percentage"]}')
PY
for arm in default lin0 unpacked; do
  extra=""; M=k_wskpl; [ $arm = lin0 ] && extra="lin_first=0" && M=k_pstep; [ $arm = unpacked ] && extra="packed_chain=0" && M=k_proj_step
  cd /tmp && rm -rf /tmp/tr_$arm && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$arm -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused $extra > /tmp/tr_$arm.log 2>&1; echo "trace $arm rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$arm/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/print_iter_timeline.py $f "$M" | tee $O/timeline_$arm.txt
  [ -n "$f" ] && [ $arm = default ] && python scripts/print_step_outside.py $f > $O/outside_fused.txt 2>&1
done
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-26s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run cg_default_again
run cg_update_in_first_product --debug lin_update_next=0
run cg_kpstep_launch --debug lin_first=0
run cg_graw_stores_raw --debug rnew_in_graw=0
run cg_graw_cols64 --debug graw_cols=64
run cg_round3_product --debug packed_chain=0
run cg_default_third
run neumann_fused --algo neumann --cg-iters 10
run neumann_fused_again --algo neumann --cg-iters 10
run neumann_round3_product --algo neumann --cg-iters 10 --debug packed_chain=0
run cg_keep_solution --keep-solution
BHG_ALL_RANKS_ON_GPU0=1 timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_selflaunch_2ranks_one_gpu_gloo.err > $O/bench_selflaunch_2ranks_one_gpu_gloo.json; echo "self-launch --gpus 2 rc=$?"; tail -c 300 $O/bench_selflaunch_2ranks_one_gpu_gloo.json; echo
BHG_LIB=$GRAFT_REPO_ROOT/betty_amd/csrc/libbhg_stamps.so timeout 200 python scripts/stamp_trace.py 2>&1 | grep -vE "Warning|warn" | tee $O/stamps_default.txt
bash scripts/gpu_pmc4.sh 2>&1 | tail -24
bash scripts/gpu_pmc_sq4.sh 2>&1 | tail -12
