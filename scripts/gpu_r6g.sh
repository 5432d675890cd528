#!/bin/bash
# Round 6, call (g): padded-twin and plan-selection tests, the batch-norm kernels in isolation (bandwidth on 32 B / element) with their
# rocprofv3 kernel stats, the driver's bench command on the current library.
set -u
O=gpurun_out/r6g; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_plan_selection.py -m gpu -q -rP -k "widths_that_are_not_multiples or description_is_what" > $O/pytest_padded.log 2>&1; echo "pytest padded rc=$?"; grep -E "padded twin|passed|failed|Error" $O/pytest_padded.log | tail -24
timeout 600 python scripts/bench_bn.py > $O/bench_bn.txt 2>&1; echo "bench_bn rc=$?"; grep -v amdgpu $O/bench_bn.txt | head -8
cd /tmp && rm -rf /tmp/bnprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnprof -o t -- python $GRAFT_REPO_ROOT/scripts/bench_bn.py > /tmp/bnprof.log 2>&1; echo "bn rocprof rc=$?"
cd $GRAFT_REPO_ROOT; cp /tmp/bnprof/*kernel_stats.csv $O/bench_bn_kernel_stats.csv 2>/dev/null; grep -E "k_bn_vjp" $O/bench_bn_kernel_stats.csv | head -6
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd.err > $O/bench_driver_cmd.json; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6g/bench_driver_cmd.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'iter_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('neumann10', d['secondary']['neumann10']['value'], d['secondary']['neumann10']['per_iteration_us'], 'cg_resident', d['secondary']['cg_resident']['avg_launch_us'], d['secondary']['cg_resident']['frac'])
PY
