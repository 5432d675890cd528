#!/bin/bash
# Round 5, call 9 — the GPU suite once more with the printed result lines AND the skip reasons (-rPs) on the library that ships, the new
# two-process test of the closed-form upper net's data-parallel mean, smoke; then the cfg-5 evidence run (product vs the reference's
# algorithm on the same GPU, K = 20).
set -u
O=gpurun_out/r5h; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 1500 python -m pytest tests -m gpu -q -rPs --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2
grep -E "cfg5 as named|withheld beta|averages" $O/pytest_gpu_full.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd.err > $O/bench_driver_cmd.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5h/bench_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print('== driver cmd: %.1f steps/s %.3f ms iter %.2f us frac %.3f traffic %s (%s) own %.3f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r['traffic'], r['traffic_source'][:40], r['own']['frac_of_own_floor']))
PY
timeout 900 python scripts/cfg5_oracle_on_gpu.py 2>&1 | grep -vE "Warning|warn" | tee $O/cfg5_oracle_on_gpu.txt
