#!/bin/bash
# Generic same-box A/B of env-switched arms: gpu_ab.sh "<ENV=.. ENV=..>" "<ENV=..>" ... ; each arm twice, CG and Neumann.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/bench_{tag}.json").read().strip().splitlines()[-1])
    r = d["roofline"] or {}; h = d["hvp_roofline"] or {}
    print("== %-34s %.1f steps/s (timed %.1f) ms/step %.3f iter_us %.1f hvp_us %.1f outside_ms %.3f" % (tag, d["value"], d.get("value_with_kernel_timing") or 0, d["ms_per_step"], d.get("per_iteration_us") or 0, h.get("avg_call_us") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("==", tag, "bench failed:", e); print(open(f"gpurun_out/bench_{tag}.err").read()[-1500:])
PY
}
i=0
for rep in a b; do
  i=0
  for arm in "$@"; do
    tag="arm${i}_${rep}"
    env $arm timeout 300 python bench.py --steps 100 --cpu-steps 0 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json; echo "[$arm]"; run $tag
    if [ "${AB_NEUMANN:-0}" = 1 ]; then
      tag="arm${i}_${rep}_neumann"
      env $arm timeout 300 python bench.py --steps 100 --cpu-steps 0 --algo neumann --cg-iters 10 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json; run $tag
    fi
    i=$((i+1))
  done
done
