#!/bin/bash
# Round 3, call C: Neumann solver on the hoisted chain, graph opt-in test, A/B lines, cfg 5 HVP profile.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "hoisted or fused or wsk or token or hip_graph or structured or wide_head" 2>&1 | grep -E "hoisted vs|passed|failed|Error|assert|rror" | tee $O/r3c_tests.log
timeout 600 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -s -k "well and neumann10" 2>&1 | grep -E "fused-default|passed|failed|Error|assert" | tail -8 | tee $O/r3c_cfg2_neumann.log
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3c_bench_$tag.err > $O/r3c_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3c_bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("== %-22s value %.1f steps/s ms/step %.3f iter_us %.1f (events %.1f) frac %.3f hvp_frac %.3f outside_ms %.3f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0, r.get("avg_launch_us_hip_events") or 0, r.get("frac") or 0, h.get("frac") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3c_bench_$tag.err").read()[-1500:])
PY
}
for rep in a b; do
BHG_MLP_HOIST=0 run neumann_classic_$rep --algo neumann --cg-iters 10
run neumann_hoist_$rep --algo neumann --cg-iters 10
BHG_HOIST_STAGED_MINK=1024 run cg_staged1024_$rep
run cg_default_$rep
done
run autograd_graph --hvp autograd --steps 60
run autograd_eager --hvp autograd --steps 60 --no-hvp-graph
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cfg5prof -o c5 -- python $GRAFT_REPO_ROOT/scripts/cfg5_hvp_profile.py > $GRAFT_REPO_ROOT/$O/r3c_cfg5_hvp_profile.log 2>&1; echo "cfg5 profile rc=$?"
cd $GRAFT_REPO_ROOT
grep "^rep" $O/r3c_cfg5_hvp_profile.log
f=$(find /tmp/cfg5prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -25 $f > $O/r3c_cfg5_hvp_kernel_stats_top25.csv; head -12 $f | cut -c1-160; python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("cfg5 profile: %d kernel launches, total kernel time %.3f s over the whole script (2 reps)" % (calls, tot/1e9))
PY
fi
