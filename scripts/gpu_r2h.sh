#!/bin/bash
# Lean refresh of profiles/ for the shipped libbhg.so: GPU suite, smoke, default bench line, rocprofv3 kernel stats of the
# same command, one-iteration timeline, PMC traffic passes (stamped with the library's sha256).
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2> gpurun_out/bench_default.err > gpurun_out/bench_default.json; tail -c 3000 gpurun_out/bench_default.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_default; cp /tmp/prof_default/*kernel_stats*.csv gpurun_out/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > gpurun_out/prof_default/bench_line_under_rocprof.json
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
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | tee gpurun_out/timeline_fused.txt; python scripts/print_step_outside.py $f > gpurun_out/outside_fused.txt; tail -1 gpurun_out/outside_fused.txt; fi
bash scripts/gpu_pmc2.sh 2>&1 | tail -45
