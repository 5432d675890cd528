#!/bin/bash
# Round 5, call 10 — last check of what ships: GPU suite (printed lines + skip reasons), smoke, the driver's command, two ranks on the
# one GPU over gloo (rank-consistent settle phase, closed-form upper net with average_over).
set -u
O=gpurun_out/r5i; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 1500 python -m pytest tests -m gpu -q -rPs --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2
grep -E "cfg5 as named" $O/pytest_gpu_full.log | grep -v print | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd.err > $O/bench_driver_cmd.json; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5i/bench_driver_cmd.json').read().strip().splitlines()[-1]); r=d['roofline']
print('== driver cmd: %.1f steps/s %.3f ms iter %.2f us frac %.3f settle %d traffic %s own %.3f cpu %.2f' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d['settle_steps'], r['traffic'], r['own']['frac_of_own_floor'], d['cpu_baseline']['value']))
PY
for i in 1 2; do BHG_ALL_RANKS_ON_GPU0=1 timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 5 --cpu-steps 0 2> $O/bench_2ranks_$i.err > $O/bench_2ranks_$i.json; echo "self-launch --gpus 2 ($i) rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_2ranks_$i.json').read().strip().splitlines()[-1]); print('   2 ranks: %.1f steps/s, settle %d, n_gpus %d' % (d['value'], d['settle_steps'], d['n_gpus']))
except Exception as e: print('   unreadable', e)
PY
done
