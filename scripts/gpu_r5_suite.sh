#!/bin/bash
# Round 5, last call — the default GPU suite as the driver runs it (with -rP for the printed lines), then the driver's bench command with
# roofline.traffic replayed from the committed, sha-matched PMC pass.
set -u
O=gpurun_out/r5s; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 1200 python -m pytest tests -m gpu -x -q -rPs --durations=10 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -3
grep -E "resnet12 cg20" $O/pytest_gpu_full.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_20steps_traffic_replayed.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5s/bench_driver_cmd_20steps_traffic_replayed.json').read().strip().splitlines()[-1]); r = d['roofline']
print('== driver cmd: %.1f steps/s  %.3f ms  iter %.2f us  frac %.3f  traffic %s  own %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('traffic'), (r.get('own') or {}).get('frac_of_own_floor')))
PY
