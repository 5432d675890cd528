#!/bin/bash
# Round 5, call 1 — measure what was SHIPPED at the end of round 4 (library 3d11e78e) before any device-code edit:
# the driver's own command twice + the 200-step line, rocprofv3 --kernel-trace --stats of the default command, one-iteration
# timeline, FETCH_SIZE / WRITE_SIZE passes, SQ counters; then the cfg-5 HVP probe in three PyTorch convolution modes.
set -u
O=gpurun_out/r5a; mkdir -p $O gpurun_out/pmc; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-steps 0 2> $O/bench_20_$i.err > $O/bench_20_$i.json; echo "bench20 rc=$?"; done
timeout 400 python bench.py 2> $O/bench_default.err > $O/bench_default.json; echo "bench200 rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5a/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "%.1f steps/s %.3f ms iter %.2f us (events %.2f) frac %.3f outside %.3f ms" % (d["value"], d["ms_per_step"], r["avg_launch_us"], r["avg_launch_us_hip_events"] or 0, r["frac"], d["outside_k_loop_ms"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-parity > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p $O/prof_default; cp /tmp/prof_default/*kernel_stats*.csv $O/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > $O/prof_default/bench_line_under_rocprof.json
cd /tmp && rm -rf /tmp/tr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr.log 2>&1; echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls /tmp/tr/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_iter_timeline.py $f k_wskpl | tee $O/timeline_default.txt
[ -n "$f" ] && python scripts/print_step_outside.py $f > $O/outside_fused.txt 2>&1
bash scripts/gpu_pmc4.sh 2>&1 | tail -40 | tee $O/pmc.log
bash scripts/gpu_pmc_sq4.sh 2>&1 | tail -16 | tee $O/pmc_sq.log
cp gpurun_out/pmc/r04_pmc_traffic.json $O/pmc_traffic.json; cp gpurun_out/pmc/r04_pmc_sq_default.json $O/pmc_sq_default.json
for m in nocudnn default benchfast; do timeout 240 python scripts/cfg5_modes.py $m 2 2>&1 | grep "^\[" | tee -a $O/cfg5_modes.txt; echo "cfg5 $m rc=$?"; done
