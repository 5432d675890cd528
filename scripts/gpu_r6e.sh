#!/bin/bash
# Round 6, call (e): the zero-padded twin (widths that are not multiples of 32), native prepare for wide heads, the restored cfg-3
# declared-batch-norm test, the cfg-5 test with its K = 3 same-GPU assertion, then the whole GPU suite as the driver runs it.
set -u
O=gpurun_out/r6e; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "widths_that_are_not_multiples" > $O/pytest_padded.log 2>&1; echo "pytest padded rc=$?"; grep -E "padded twin|passed|failed|Error" $O/pytest_padded.log | tail -20
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "cfg3_resnet12 or cfg5_reference_network" > $O/pytest_cfg35.log 2>&1; echo "pytest cfg3/5 rc=$?"; grep -E "^resnet12|^cfg5|passed|failed" $O/pytest_cfg35.log | tail -8
timeout 2400 python -m pytest tests -m gpu -q -x -rs --durations=12 > $O/pytest_gpu_full.log 2>&1; echo "pytest full rc=$?"; tail -25 $O/pytest_gpu_full.log
