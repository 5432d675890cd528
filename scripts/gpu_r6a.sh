#!/bin/bash
# Round 6, call (a): where cfg 3's second goes (rocprofv3 kernel trace of one ResNet-12 CG-20 step, classified), the driver's bench
# command on the library as the round began + the new `secondary` / `devices` objects, the new GPU tests of this commit.
set -u
O=gpurun_out/r6a; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
cd /tmp && rm -rf /tmp/cfg3 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3 -o t -- python $GRAFT_REPO_ROOT/scripts/cfg3_profile.py 2 > /tmp/cfg3.log 2>&1; echo "cfg3 rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -2 /tmp/cfg3.log | tee $O/cfg3_step.txt
f=$(ls /tmp/cfg3/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_cfg3_breakdown.py $f | tee $O/cfg3_breakdown.txt
cp /tmp/cfg3/*kernel_stats.csv $O/cfg3_kernel_stats.csv 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd.err > $O/bench_driver_cmd.json; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6a/bench_driver_cmd.json').read().strip().splitlines()[-1])
print('value', d['value'], 'iter_us', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'])
print('secondary', json.dumps(d.get('secondary'))[:1500])
print('devices', d.get('devices'), 'ranks_seen', d.get('ranks_seen'))
PY
timeout 1200 python -m pytest tests/test_gpu_global.py tests/test_cfg2_goldens.py -m gpu -q -x -rs -k "shipped or ddp_wrapper or averages or metric_workload" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_new.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cfg2_metric_workload" > $O/pytest_cfg2.log 2>&1; echo "pytest2 rc=$?"; tail -3 $O/pytest_cfg2.log
