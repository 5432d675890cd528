#!/bin/bash
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=10 -k "resident or hybrid or falls_back or deterministic or kernels_vs_oracle or diagonal or variants or roberta" 2>&1 | tail -8
timeout 300 python scripts/bench_kernels.py --scale 2 --iters 30 2>/dev/null > gpurun_out/bench_kernels_N20M.json; python -c "
import json; d=json.load(open('gpurun_out/bench_kernels_N20M.json')); print('N20M', {k:(round(v['us'],1), round(v['GBps'])) for k,v in d['kernels'].items()})"
timeout 300 python scripts/bench_kernels.py --scale 1 --iters 30 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('N10M', {k:(round(v['us'],1), round(v['GBps'])) for k,v in d['kernels'].items()})"
