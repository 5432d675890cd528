#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 -k "fused or structured or mlp or cfg2 or alternate or resident or deterministic or kernels_vs_oracle or diagonal or falls_back" 2>&1 | tail -12 | tee gpurun_out/pytest_gpu_f.log
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
run cg_fused --algo cg
run cg_nofuse --algo cg --no-fuse
run neumann_fused --algo neumann --cg-iters 10
BHG_NEUMANN_P_EVERY_ITER=1 run neumann_fused_pevery --algo neumann --cg-iters 10
timeout 300 python scripts/bench_kernels.py --scale 1 --iters 40 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print({k:(round(v['us'],1), round(v['GBps'])) for k,v in d['kernels'].items()})"
