#!/bin/bash
# Round-end style check + everything profiles/ needs: all GPU tests, smoke, the default bench line (with the CPU
# baseline), rocprofv3 kernel stats of the same command, PMC traffic / matrix-pipe passes, timelines, A/B bench lines.
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/smoke.log
timeout 600 python bench.py 2> gpurun_out/bench_default.err > gpurun_out/bench_default.json; tail -c 3000 gpurun_out/bench_default.json
run() { tag=$1; shift
  timeout 300 python bench.py --steps 100 --cpu-steps 0 "$@" 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("== %-22s value %.1f (timed %.1f) steps/s  ms/step %.3f  iter_us %.1f  roof_frac %.3f  hvp_us %.1f hvp_frac %.3f outside_ms %.3f" % ("$tag", d["value"], d.get("value_with_kernel_timing") or 0, d["ms_per_step"], d.get("per_iteration_us") or 0, r.get("frac") or 0, h.get("avg_call_us") or 0, h.get("frac") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("gpurun_out/bench_$tag.err").read()[-1500:])
PY
}
run cg_nofuse --algo cg --no-fuse
run cg_keep_solution --algo cg --keep-solution
run neumann_keep_solution --algo neumann --cg-iters 10 --keep-solution
run cg_autograd --algo cg --hvp autograd --steps 20
run neumann_fused --algo neumann --cg-iters 10
run neumann_nofuse --algo neumann --cg-iters 10 --no-fuse
run darts --algo darts
run cg_global_ws1 --algo cg --mode global --steps 20
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
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_n -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 neumann fused > /tmp/tr_n.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_n/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_outer_all | tee gpurun_out/timeline_neumann_fused.txt; fi
bash scripts/gpu_pmc2.sh 2>&1 | tail -45
bash scripts/gpu_pmc_sq.sh 2>&1 | tail -28
timeout 300 python scripts/bench_kernels.py --scale 1 --extra 5000000 --iters 40 2>/dev/null > gpurun_out/bench_kernels_N15M.json; python -c "import json; d=json.load(open(\"gpurun_out/bench_kernels_N15M.json\")); print(\"N15M\", {k:(round(v[\"us\"],1), round(v[\"GBps\"])) for k,v in d[\"kernels\"].items()})"
kb() { tag=$1; shift; timeout 300 python scripts/bench_kernels.py --iters 40 "$@" 2>/dev/null > gpurun_out/bench_kernels_$tag.json; python -c "import json; d=json.load(open(\"gpurun_out/bench_kernels_$tag.json\")); print(\"$tag\", {k:(round(v[\"us\"],1), round(v[\"GBps\"])) for k,v in d[\"kernels\"].items()})"; }
kb N10M_cached --scale 1
kb N10M_cache_defeated --scale 1 --scrub-mb 512
kb N20M --scale 2
kb N120M --scale 12
kb N120M_cache_defeated --scale 12 --scrub-mb 512
