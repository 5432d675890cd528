#!/bin/bash
set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "power_of_two or mlp_hvp_full_size" 2>&1 | tail -15
