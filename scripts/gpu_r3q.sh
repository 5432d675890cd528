#!/bin/bash
# round 3: dot blocks of k_hoist take 4 x 256 float4 each (fewer partials for k_proj_step's blocks to sum again)
mkdir -p gpurun_out/r3q
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_cfg2_goldens.py tests/test_gpu_parity.py -m gpu -x -q -s -k "cfg2 or hoisted or projected or solution" 2>&1 | grep -E "fused-default|passed|failed|Error|error" | tail -14 | tee gpurun_out/r3q/pytest.log
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> gpurun_out/r3q/bench_$tag.err > gpurun_out/r3q/bench_$tag.json; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3q/bench_$tag.json') if l.startswith('{')][-1]; print('== %-14s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run a
run b
run tunableop --hvp autograd --steps 60 --tunableop --no-slope
