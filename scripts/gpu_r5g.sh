#!/bin/bash
# Round 5, call 7 — the K-looped closing launch (k_grawk) and the six-launch form for batches beyond 128: the solver-form tests, then
# the default bench line (must not move: the 128-row instance is untouched).
set -u
O=gpurun_out/r5g; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -rP --durations=5 -k "hoisted_and_projected or projected_solvers_edge or fused_solver_matches or packed_prepare or projection_is_gated" > $O/pytest_forms.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $O/pytest_forms.log | tail -12
grep -E "\[512, 256, 256, 64, 10\] K=4 full|\[256, 256, 128, 64, 10\]|\[256, 192, 128, 64, 32, 10\]" $O/pytest_forms.log | head -40
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 0 2> $O/bench_20.err > $O/bench_20.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5g/bench_20.json').read().strip().splitlines()[-1]); r=d['roofline']
print('== default 20 steps: %.1f steps/s %.3f ms iter %.2f us frac %.3f outside %.3f ms' % (d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], d['outside_k_loop_ms']))
PY
