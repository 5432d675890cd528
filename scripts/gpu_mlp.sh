#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mlp or structured" 2>&1 | tail -5
timeout 300 python bench.py --cpu-steps 0 2>&1 | tail -1 | cut -c1-330
bash scripts/gpu_trace.sh
