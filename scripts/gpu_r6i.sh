#!/bin/bash
# Round 6, call (i): heads of up to 256 classes in the fused solvers (class-chunk loop in head_body.inc): the wide-head tests, the
# solver-form tests incl. a 100-class head, the whole suite, and the driver's bench command (did k_headu's code change cost anything?).
set -u
O=gpurun_out/r6i; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -rP -k "wide_head" > $O/pytest_wide.log 2>&1; echo "pytest wide rc=$?"; grep -E "^wide head|passed|failed|Error" $O/pytest_wide.log | tail -16
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected_chain_match_classic_chain and 100]" > $O/pytest_forms.log 2>&1; echo "pytest forms rc=$?"; tail -3 $O/pytest_forms.log
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=5 > $O/pytest_gpu_full.log 2>&1; echo "pytest full rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest_gpu_full.log | tail -8
for i in a b; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --cpu-steps 0 2> $O/bench_$i.err > $O/bench_$i.json; python -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
