#!/bin/bash
# Round 6, call (f): the narrow-end probe (VERDICT r5 #1b: one launch with two row-group rendezvous vs three launches), the padded-twin
# tests after the Neumann-accumulator fix, and the whole GPU suite without -x.
set -u
O=gpurun_out/r6f; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do timeout 120 ./build_probes/narrow_probe > $O/narrow_probe_$i.txt 2>&1; echo "probe rc=$?"; cat $O/narrow_probe_$i.txt; done
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -rP -k "widths_that_are_not_multiples or wide_head" > $O/pytest_padded.log 2>&1; echo "pytest padded rc=$?"; grep -E "padded twin|^wide head|passed|failed|Error" $O/pytest_padded.log | tail -30
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest full rc=$?"; tail -22 $O/pytest_gpu_full.log
