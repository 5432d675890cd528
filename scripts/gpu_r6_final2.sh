#!/bin/bash
# Round 6, second session — the refresh of profiles/r06_* on the library that ships after the factor-exchange form went in (sha in
# $O/lib.sha; the bench lines carry it).  Everything scripts/gpu_r6_final.sh collects (it is called first, into gpurun_out/r6final/), then the
# global-batch lines: factor exchange x 2 and the one-pass form on the same box, the kernel table of the factor-exchange line under
# rocprofv3, the per-rank compute at emulated worlds 1 / 2 / 4 / 8, what a collective costs at world size 1, the host's enqueue cost.
set -u
bash scripts/gpu_r6_final.sh
O=gpurun_out/r6final; export TMPDIR=/tmp
for i in a b; do timeout 400 python bench.py --mode global --steps 20 --warmup 5 2> $O/bench_global_fx_$i.err > $O/bench_global_fx_$i.json; echo "global fx $i rc=$?"; done
timeout 400 python bench.py --mode global --global-form one_pass --steps 20 --warmup 5 2> $O/bench_global_onepass.err > $O/bench_global_onepass.json; echo "global one-pass rc=$?"
for v in plain gather; do extra=""; [ $v = gather ] && extra="--fx-always-gather"
  timeout 300 python bench.py --mode global --steps 20 --warmup 5 --no-parity --cpu-steps 0 $extra 2> $O/bench_global_coll_$v.err > $O/bench_global_coll_$v.json; echo "collectives $v rc=$?"; done
timeout 300 python scripts/fx_host_cost.py 2>&1 | grep always_gather > $O/fx_host_cost.txt; cat $O/fx_host_cost.txt
timeout 600 python scripts/fx_emulated_world.py > $O/fx_emulated_world.txt 2>&1; echo "emulated worlds rc=$?"; grep -v amdgpu $O/fx_emulated_world.txt | head -4 | cut -c1-200
cd /tmp && rm -rf /tmp/prof_fx && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_fx -o fxg -- python $GRAFT_REPO_ROOT/bench.py --mode global --steps 20 --warmup 5 --no-kernel-timing --cpu-steps 0 --no-parity > /tmp/prof_fx.log 2>&1; echo "fx rocprof rc=$?"
cd $GRAFT_REPO_ROOT; python scripts/kstats.py /tmp/prof_fx/*results.db 30 > $O/bench_global_fx_kernel_stats.txt; head -12 $O/bench_global_fx_kernel_stats.txt | cut -c1-160
python - <<'PY'
import json
for tag in ("global_fx_a", "global_fx_b", "global_onepass", "global_coll_plain", "global_coll_gather"):
    try:
        d = json.loads(open(f"gpurun_out/r6final/bench_{tag}.json").read().strip().splitlines()[-1])
        p = d.get("parity") or {}
        print(f"== {tag:20s} {d['value']:.1f} steps/s {d['ms_per_step']:.3f} ms | parity well {((p.get('well_conditioned_variant') or {}).get('vs_reference_cpu_fp32'))} metric-vs-fp64 {((p.get('metric_instance') or {}).get('vs_reference_fp64'))} | lib {d['config']['lib_sha256']}")
    except Exception as e:
        print("==", tag, "unreadable", e)
PY
