#!/bin/bash
# round 3: per-iteration Gram products K-split over workgroups (last arriver sums the slabs)
mkdir -p gpurun_out/r3s
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_cfg2_goldens.py tests/test_gpu_parity.py -m gpu -x -q -s -k "cfg2 or hoisted or projected or solution or deterministic" 2>&1 | grep -E "fused-default|passed|failed|Error|error" | tail -14 | tee gpurun_out/r3s/pytest.log
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> gpurun_out/r3s/bench_$tag.err > gpurun_out/r3s/bench_$tag.json; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3s/bench_$tag.json') if l.startswith('{')][-1]; print('== %-14s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run ksplit
BHG_GRAM_KSPLIT=0 run nosplit
run ksplit2
run neumann --algo neumann --cg-iters 10
BHG_GRAM_KSPLIT=0 run neumann_nosplit --algo neumann --cg-iters 10
