#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$1" 2>&1 | tail -25
