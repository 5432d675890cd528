#!/bin/bash
# Refresh of profiles/r03_* for the shipped libbhg.so: GPU suite, smoke, default bench line, rocprofv3 kernel stats of the same
# command, one-iteration timeline, secondary lines, PMC traffic passes (stamped with the library's sha256).
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -16
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee $O/smoke.log
timeout 600 python bench.py 2> $O/bench_default.err > $O/bench_default.json; tail -c 2500 $O/bench_default.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p $O/prof_default; cp /tmp/prof_default/*kernel_stats*.csv $O/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > $O/prof_default/bench_line_under_rocprof.json
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_default/*kernel_stats*.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms", tot / 1e6)
    for r in rows[:16]:
        print(f'{r["Name"][:84]:84s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:8.2f} pct={r["Percentage"]}')
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_f -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_f.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_f/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f "k_proj_step" | tee $O/timeline_fused.txt; python scripts/print_step_outside.py $f > $O/outside_fused.txt 2>&1; tail -1 $O/outside_fused.txt; fi
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python -c "
import json
d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print('== %-18s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run neumann_fused --algo neumann --cg-iters 10
BHG_MLP_PROJ=0 run neumann_classic_chain --algo neumann --cg-iters 10
run cg_nofuse --no-fuse
BHG_MLP_HOIST=0 run cg_classic_chain
BHG_MLP_PROJ=0 run cg_hoisted_not_projected
run cg_keep_solution --keep-solution
BHG_PROJ_STEP_ALONE=1 run cg_proj_two_launches
run cg_autograd_graph_persistent --hvp autograd --steps 60
run cg_autograd_graph_per_solve --hvp autograd --steps 60 --hvp-graph solve
run cg_autograd_eager --hvp autograd --steps 60 --no-hvp-graph
run cg_autograd_tunableop --hvp autograd --steps 60 --tunableop
run neumann_autograd_graph_persistent --hvp autograd --steps 60 --algo neumann --cg-iters 10
run darts --algo darts
run cg_global_ws1 --mode global
BHG_MLP_HOIST=0 run cg_global_ws1_classic_chain --mode global
bash scripts/gpu_pmc3.sh 2>&1 | tail -40
bash scripts/gpu_pmc3_neumann.sh 2>&1 | tail -6
