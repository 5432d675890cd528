#!/bin/bash
# A/B of the in-workgroup split-K GEMM (BHG_MLP_WSK=1): MLP tests under the switch, bench lines of both arms, timeline.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
BHG_MLP_WSK=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mlp or fused or structured or cfg2" 2>&1 | tail -8
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
run wsk0_a
BHG_MLP_WSK=1 run wsk1_a
BHG_MLP_WSK=1 BHG_MLP_WSK_DEPTH=2 run wsk1_d2
run wsk0_b
BHG_MLP_WSK=1 run wsk1_b
BHG_MLP_WSK=1 run wsk1_neumann --algo neumann --cg-iters 10
run wsk0_neumann --algo neumann --cg-iters 10
cd /tmp && BHG_MLP_WSK=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_f -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_f.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_f/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | tee gpurun_out/timeline_fused_wsk.txt; fi
