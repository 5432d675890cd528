#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "projected_solvers_edge or hoisted_and_projected" 2>&1 | grep -E "projected vs classic|passed|failed|Error|assert|rror" | tail -30 | tee $O/r3g_tests.log
BHG_WSL_DEPTH=3 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "hoisted_and_projected or fused_solver" 2>&1 | tail -2
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3g_bench_$tag.err > $O/r3g_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3g_bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("== %-22s value %.1f steps/s ms/step %.3f iter_us %.1f (events %.1f) frac %.3f hvp_frac %.3f outside_ms %.3f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0, r.get("avg_launch_us_hip_events") or 0, r.get("frac") or 0, h.get("frac") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3g_bench_$tag.err").read()[-1500:])
PY
}
for rep in a b; do
run depth2_$rep
BHG_WSL_DEPTH=3 run depth3_$rep
done
BHG_WSL_DEPTH=3 run neumann_depth3 --algo neumann --cg-iters 10
run neumann_depth2 --algo neumann --cg-iters 10
cd /tmp && BHG_WSL_DEPTH=3 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_p -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_p.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_p/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f "k_proj_update" | tee $O/r3g_timeline_depth3.txt; fi
