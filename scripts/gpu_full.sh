#!/bin/bash
# Full round-end style check: all GPU tests, smoke, default bench (+rocprof of the same command).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -6 | tee gpurun_out/smoke.log
bash scripts/gpu_bench.sh default 2>&1 | cut -c1-2500
