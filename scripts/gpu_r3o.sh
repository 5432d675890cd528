#!/bin/bash
# round 3: k_proj_step (scalars of iteration k-1 + recurrences of iteration k in one launch) — parity, then A/B
mkdir -p gpurun_out/r3o
cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_cfg2_goldens.py -x -q -k "hoisted or projected or solution or fused_solver or cfg2 or token or deterministic" 2>&1 | tail -8 | tee gpurun_out/r3o/pytest.log
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> gpurun_out/r3o/bench_$tag.err > gpurun_out/r3o/bench_$tag.json; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3o/bench_$tag.json') if l.startswith('{')][-1]; print('== %-14s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run merged
BHG_PROJ_STEP_ALONE=1 run alone
run merged2
BHG_PROJ_STEP_ALONE=1 run alone2
