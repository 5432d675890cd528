#!/bin/bash
# Round 3, call B: the hoisted chain (k_hoist) — parity tests, A/B against the classic chain, split / form sweeps, timelines.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
python -X faulthandler -m pytest tests/test_dropin_reference.py -m gpu -x -q -s > $O/r3b_dropin.log 2>&1; echo "dropin rc=$?"; grep -vE "^Extension modules" $O/r3b_dropin.log | tail -30
BHG_HVP_GRAPH=0 python -X faulthandler -m pytest tests/test_dropin_reference.py -m gpu -x -q > $O/r3b_dropin_nograph.log 2>&1; echo "dropin (no graph) rc=$?"; grep -vE "^Extension modules" $O/r3b_dropin_nograph.log | tail -5
for c in cfg2 cfg3; do timeout 300 python scripts/hvp_graph_probe.py $c 5 > $O/r3b_graph_$c.log 2>&1; echo "probe $c rc=$?"; grep -E "BHG_HVP|graph vs|Warn" $O/r3b_graph_$c.log; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -x -k "hoisted or fused_solver or fused_cg or wsk or token or structured_goldens or structured_hip" 2>&1 | grep -E "hoisted vs|passed|failed|Error|assert|rror" | tee $O/r3b_tests.log
timeout 900 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -s -k "well and cg20" 2>&1 | grep -E "fused|passed|failed|Error|assert" > $O/r3b_cfg2_goldens.log; tail -4 $O/r3b_cfg2_goldens.log
run() { tag=$1; shift
  timeout 300 python bench.py --cpu-steps 0 "$@" 2> $O/r3b_bench_$tag.err > $O/r3b_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3b_bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("== %-22s value %.1f steps/s ms/step %.3f iter_us %.1f (events %.1f) frac28N %.3f hvp_frac %.3f outside_ms %.3f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0, r.get("avg_launch_us_hip_events") or 0, r.get("frac") or 0, h.get("frac") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3b_bench_$tag.err").read()[-1500:])
PY
}
for rep in a b; do
BHG_MLP_HOIST=0 run classic_$rep
run hoist_$rep
done
BHG_HOIST_WGS=1024 run hoist_wgs1024
BHG_HOIST_WGS=640 run hoist_wgs640
BHG_HOIST_WGS=512 run hoist_wgs512
BHG_HOIST_STAGED_MINK=100000 run hoist_direct_wsk
BHG_HOIST_STAGED_MINK=256 run hoist_all_staged
BHG_MLP_WSK_DEPTH=2 run hoist_depth2
trace() { tag=$1; mark=$2; shift; shift
  cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_$tag.log 2>&1; echo "trace $tag rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$tag/*kernel_trace.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f "$mark" | tee $O/r3b_timeline_$tag.txt; else tail -5 /tmp/tr_$tag.log; fi
}
trace hoist "k_hoist(" BHG_MLP_HOIST=1
trace classic k_cg_beta BHG_MLP_HOIST=0
