#!/bin/bash
# Round 5 — the driver's bench command on whatever box this call lands on (box-to-box spread of the shipped library's line).
set -u
O=gpurun_out/r5x; mkdir -p $O
T=$(date +%s)
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$T.json 2> $O/bench_$T.err; echo "rc=$?"
python - <<PY
import json
d = json.loads(open('$O/bench_$T.json').read().strip().splitlines()[-1]); r = d['roofline']
print('== box $T: %.1f steps/s  %.3f ms  iter %.2f us (%s)  frac %.3f  traffic %s  cpu %s  %s' % (d['value'], d['ms_per_step'], r['avg_launch_us'], ['%.2f' % v for v in r['avg_launch_us_min_max']], r['frac'], r.get('traffic'), d['cpu_baseline']['value'], d['config']['lib_sha256'][:8]))
PY
