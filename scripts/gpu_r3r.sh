#!/bin/bash
# round 3: raw.raw from the G(raw) tiles themselves (no Q / P Gram products)
mkdir -p gpurun_out/r3r
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_cfg2_goldens.py tests/test_gpu_parity.py -m gpu -x -q -s -k "cfg2 or hoisted or projected or solution" 2>&1 | grep -E "fused-default|passed|failed|Error|error" | tail -14 | tee gpurun_out/r3r/pytest.log
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> gpurun_out/r3r/bench_$tag.err > gpurun_out/r3r/bench_$tag.json; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3r/bench_$tag.json') if l.startswith('{')][-1]; print('== %-14s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
run a
run b

