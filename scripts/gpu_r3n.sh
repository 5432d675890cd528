#!/bin/bash
# round 3: two-process gloo test of the global one-pass solver; BLAS back ends under the opaque (autograd) HVP path
mkdir -p gpurun_out/r3n
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_global.py -q -k two_processes 2>&1 | tail -8
run() { tag=$1; shift; timeout 500 python bench.py --hvp autograd --steps 60 --cpu-steps 0 --no-slope > gpurun_out/r3n/bench_$tag.json 2> gpurun_out/r3n/bench_$tag.err; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3n/bench_$tag.json') if l.startswith('{')][-1]; print('== %-12s %.1f steps/s %.3f ms/step' % ('$tag', d['value'], d['ms_per_step']))" 2>&1 | tail -1; }
run default
TORCH_BLAS_PREFER_HIPBLASLT=0 run rocblas
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_VERBOSE=0 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv run tunable
tail -3 gpurun_out/r3n/bench_tunable.err
