#!/bin/bash
# Round 6, call (k): the whole GPU suite once more on the final tree (new: goldens of shapes outside the benchmark's family, the example test),
# the driver's command, and the self-launched 2-rank line over gloo after the fence moved inside the timed regions.
set -u
O=gpurun_out/r6k; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 2400 python -m pytest tests -m gpu -q -rPs --durations=10 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd.err > $O/bench_driver_cmd.json; echo "driver cmd rc=$?"
BHG_ALL_RANKS_ON_GPU0=1 timeout 400 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_2ranks_gloo.err > $O/bench_2ranks_gloo.json; echo "2 ranks rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6k/bench_driver_cmd.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("driver cmd:", d["value"], r["avg_launch_us"], r["frac"], r["traffic"], d["secondary"]["neumann10"]["value"], d["secondary"]["cg_resident"]["frac"])
d = json.loads(open("gpurun_out/r6k/bench_2ranks_gloo.json").read().strip().splitlines()[-1])
print("2 ranks gloo:", d["value"], d["ranks_seen"], d["devices"]["distinct_devices"], d["devices"]["allreduce_M_floats_us"])
PY
