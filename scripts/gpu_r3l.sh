#!/bin/bash
# round 3: the global-batch one-pass CG solver (tests + --mode global at world size 1)
mkdir -p gpurun_out/r3l
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_global.py -x -q 2>&1 | tail -15 > gpurun_out/r3l/pytest_global.log
cat gpurun_out/r3l/pytest_global.log
for arm in default hoist0; do
  if [ $arm = hoist0 ]; then export BHG_MLP_HOIST=0; else unset BHG_MLP_HOIST; fi
  timeout 300 python bench.py --mode global --gpus 1 --steps 200 --warmup 20 --no-slope > gpurun_out/r3l/bench_global_$arm.json 2> gpurun_out/r3l/bench_global_$arm.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3l/bench_global_$arm.json").read().strip().splitlines()[-1])
    print("$arm", d["value"], d["ms_per_step"], d["config"].get("parallelism"))
except Exception as e:
    print("$arm failed", e); print(open("gpurun_out/r3l/bench_global_$arm.err").read()[-1500:])
PY
done
unset BHG_MLP_HOIST
timeout 300 python bench.py --mode global --gpus 1 --steps 200 --warmup 20 --no-slope --keep-solution > gpurun_out/r3l/bench_global_keep.json 2>&1
tail -c 600 gpurun_out/r3l/bench_global_keep.json
