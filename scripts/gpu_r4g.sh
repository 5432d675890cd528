#!/bin/bash
# round 4, call g: the whole GPU suite on the linear-first-product default, smoke, default bench line with its parity object
set -u
O=gpurun_out/r4g; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/ -m gpu -q -x --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -16
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['per_iteration_us'], d['roofline']['frac'], d['parity'] and d['parity']['ok'], d['cpu_baseline'])"
