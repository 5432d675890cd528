#!/bin/bash
mkdir -p gpurun_out/r3m
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_global.py -q 2>&1 | tail -25 > gpurun_out/r3m/pytest_global.log
cat gpurun_out/r3m/pytest_global.log
