#!/bin/bash
# Round 6, call (j): shapes outside the benchmark's family against the reference's CPU goldens (tests/golden/shapes.npz), the example test.
set -u
O=gpurun_out/r6j; mkdir -p $O
timeout 900 python -m pytest tests/test_shapes_goldens.py -m gpu -q -rP > $O/pytest_shapes.log 2>&1; echo "rc=$?"; grep -E "form '|passed|failed|Error" $O/pytest_shapes.log | tail -14
