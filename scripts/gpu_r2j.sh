#!/bin/bash
# A/B: M-side row clamp of k_gemm (BHG_GEMM_NO_ROW_CLAMP=1 restores the plain loads), in-workgroup split-K for short
# reductions only (BHG_MLP_WSK=2).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
BHG_MLP_WSK=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mlp or fused or structured or cfg2" 2>&1 | tail -8
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
for rep in a b; do
BHG_GEMM_NO_ROW_CLAMP=1 run base_$rep
run clamp_$rep
BHG_MLP_WSK=2 run clamp_wsk2_$rep
BHG_GEMM_NO_ROW_CLAMP=1 BHG_MLP_WSK=2 run wsk2_$rep
done
BHG_MLP_WSK=2 run neumann_clamp_wsk2 --algo neumann --cg-iters 10
run neumann_clamp --algo neumann --cg-iters 10
BHG_GEMM_NO_ROW_CLAMP=1 run neumann_base --algo neumann --cg-iters 10
cd /tmp && BHG_MLP_WSK=2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_f -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_f.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr_f/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | tee gpurun_out/timeline_fused_wsk2.txt; fi
