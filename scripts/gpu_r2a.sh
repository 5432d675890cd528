#!/bin/bash
# Round-2 run A: GPU tests, fused vs un-fused bench lines, one-iteration timeline of the fused solver.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
for cfg in "cg_fused --algo cg" "cg_nofuse --algo cg --no-fuse" "neumann_fused --algo neumann --cg-iters 10" "neumann_nofuse --algo neumann --cg-iters 10 --no-fuse"; do
  set -- $cfg; tag=$1; shift
  timeout 300 python bench.py --steps 100 --cpu-steps 0 "$@" 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
  echo "== $tag"; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("value %.1f steps/s  ms/step %.3f  iter_us %s  roof_frac %s  hvp_us %s hvp_frac %s outside_ms %s" % (d["value"], d["ms_per_step"], d.get("per_iteration_us"), r.get("frac"), h.get("avg_call_us"), h.get("frac"), d.get("outside_k_loop_ms")))
except Exception as e:
    print("bench failed:", e); print(open("gpurun_out/bench_$tag.err").read()[-1500:])
PY
done
for arm in fused nofuse; do
  cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$arm -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 2 cg $arm > /tmp/tr_$arm.log 2>&1; echo "trace $arm rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$arm/*kernel_trace.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    if [ $arm = fused ]; then python scripts/print_iter_timeline.py $f k_cg_pdir | tee gpurun_out/timeline_$arm.txt; else python scripts/print_iter_timeline.py $f k_cg_resident | tee gpurun_out/timeline_$arm.txt; fi
  else tail -5 /tmp/tr_$arm.log; fi
done
