#!/bin/bash
# Round 6 — the refresh of profiles/r06_* on the library that ships (sha in $O/lib.sha; bench lines carry it): the GPU suite with every
# test's printed lines and skip reasons, smoke, the driver's command twice + the 200-step line (with `secondary`, `devices`, CPU baseline),
# rocprofv3 --kernel-trace --stats of the default command, FETCH_SIZE / WRITE_SIZE traffic of the CG / Neumann iteration and of the
# batch-norm kernels (stamped with the sha256), the batch-norm kernels in isolation with their kernel stats, one cfg-3 step with declared
# batch norm classified by kernel, cfg-3 speed, the multi-GPU self-check.
set -u
O=gpurun_out/r6final; mkdir -p $O gpurun_out/pmc; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
timeout 2400 python -m pytest tests -m gpu -q -rPs --durations=10 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu_full.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
for i in a b; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd_$i.err > $O/bench_driver_cmd_$i.json; echo "driver cmd $i rc=$?"; done
timeout 500 python bench.py 2> $O/bench_default.err > $O/bench_default.json; echo "default rc=$?"
python - <<'PY'
import json
for tag in ("driver_cmd_a", "driver_cmd_b", "default"):
    try:
        d = json.loads(open(f"gpurun_out/r6final/bench_{tag}.json").read().strip().splitlines()[-1])
        s = d.get("secondary") or {}
        print(f"== {tag:14s} {d['value']:.1f} steps/s {d['ms_per_step']:.3f} ms iter {d['roofline']['avg_launch_us']:.2f} us frac {d['roofline']['frac']:.3f} traffic {d['roofline']['traffic']} "
              f"| neumann10 {s.get('neumann10', {}).get('value', 0):.1f} {s.get('neumann10', {}).get('per_iteration_us', 0):.2f} us | cg_resident {s.get('cg_resident', {}).get('avg_launch_us', 0):.2f} us "
              f"{s.get('cg_resident', {}).get('frac', 0):.3f} | cpu {((d.get('cpu_baseline') or {}).get('value') or 0):.3f} | lib {d['config']['lib_sha256']}")
    except Exception as e:
        print("==", tag, "unreadable", e)
PY
cd /tmp && rm -rf /tmp/prof_default && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --reps 2 > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; cp /tmp/prof_default/*kernel_stats*.csv $O/bench_default_kernel_stats.csv 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > $O/bench_line_under_rocprof.json
bash scripts/gpu_pmc6.sh 2>&1 | tail -40 | tee $O/pmc.log
cp gpurun_out/pmc/r06_pmc_traffic.json gpurun_out/pmc/r06_pmc_bn_traffic.json $O/ 2>/dev/null
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_driver_cmd_traffic_replayed.err > $O/bench_driver_cmd_traffic_replayed.json; echo "replayed rc=$?"
timeout 600 python scripts/bench_bn.py > $O/bench_bn.txt 2>&1; echo "bench_bn rc=$?"; grep -v amdgpu $O/bench_bn.txt | head -6
cd /tmp && rm -rf /tmp/bnprof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bnprof -o t -- python $GRAFT_REPO_ROOT/scripts/bench_bn.py > /tmp/bnprof.log 2>&1; echo "bn rocprof rc=$?"
cd $GRAFT_REPO_ROOT; cp /tmp/bnprof/*kernel_stats.csv $O/bench_bn_kernel_stats.csv 2>/dev/null
cd /tmp && rm -rf /tmp/cfg3f && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3f -o t -- python $GRAFT_REPO_ROOT/scripts/cfg3_profile.py 2 fused-bn > /tmp/cfg3f.log 2>&1; echo "cfg3 rocprof rc=$?"
cd $GRAFT_REPO_ROOT; f=$(ls /tmp/cfg3f/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_cfg3_breakdown.py $f > $O/cfg3_declared_breakdown.txt; head -2 $O/cfg3_declared_breakdown.txt
timeout 900 python scripts/cfg3_resnet12_compare.py > $O/cfg3_compare.txt 2>&1; echo "compare rc=$?"; grep -v amdgpu.ids $O/cfg3_compare.txt | tail -2
timeout 600 python scripts/multi_gpu_selfcheck.py > $O/selfcheck_n1_rccl.txt 2>&1; echo "selfcheck n1 rc=$?"; grep SELFCHECK $O/selfcheck_n1_rccl.txt
timeout 600 python scripts/multi_gpu_selfcheck.py --gpus 2 --backend gloo --all-on-gpu0 > $O/selfcheck_n2_gloo.txt 2>&1; echo "selfcheck n2 rc=$?"; grep SELFCHECK $O/selfcheck_n2_gloo.txt
BHG_ALL_RANKS_ON_GPU0=1 timeout 400 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_selflaunch_2ranks_one_gpu_gloo.err > $O/bench_selflaunch_2ranks_one_gpu_gloo.json; echo "self-launch --gpus 2 rc=$?"
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r6final/bench_selflaunch_2ranks_one_gpu_gloo.json").read().strip().splitlines()[-1])
    print("2 ranks on one GPU over gloo:", d["value"], "steps/s; ranks_seen", d["ranks_seen"], "distinct_devices", d["devices"]["distinct_devices"], "allreduce us", d["devices"]["allreduce_M_floats_us"])
except Exception as e:
    print("2-rank line unreadable", e)
PY
