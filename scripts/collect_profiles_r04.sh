#!/bin/bash
# Copy what scripts/gpu_r4_full.sh left under gpurun_out/ (scratch) into profiles/ (tracked), named for round 4.
set -u
c() { [ -f "$1" ] && cp "$1" "$2"; }
g=gpurun_out/r4; p=profiles
c $g/bench_default.json $p/r04_bench_default.json
c $g/prof_default/bench_kernel_stats.csv $p/r04_bench_default_kernel_stats.csv
c $g/prof_default/bench_line_under_rocprof.json $p/r04_bench_line_under_rocprof.json
grep -vE "^Extension|Warning|warn" $g/pytest_gpu_full.log | grep -E "cfg2|hoisted|projected|classic|full|reference Engine|gated|Roberta|supernet|resnet|passed|failed|skipped|durations|s call|s setup" > $p/r04_pytest_gpu.log
c $g/smoke.log $p/r04_smoke.log
c $g/timeline_default.txt $p/r04_timeline_fused_fully_projected.txt
c $g/timeline_unpacked.txt $p/r04_timeline_round3_product_same_box.txt
c $g/outside_fused.txt $p/r04_outside_the_k_loop.txt
c $g/stamps_default.txt $p/r04_stamps_k_graw_k_pstep.txt
c $g/chain_probe.txt $p/r04_chain_probe_k_wskp.txt
c $g/opaque_product_vs_reference_on_gpu.txt $p/r04_opaque_product_vs_reference_on_gpu.txt
c gpurun_out/pmc/r04_pmc_traffic.json $p/r04_pmc_traffic.json
c gpurun_out/pmc/r04_pmc_sq_default.json $p/r04_pmc_sq_default.json
c gpurun_out/r4b/wskp_probe.txt $p/r04_wskp_probe_waves_layout_depth.txt
for t in cg_default_again cg_round3_product cg_gram_launch cg_graw_v1 cg_pstep_v1 cg_depth3 neumann_fused neumann_round3_product cg_keep_solution cg_nofuse cg_autograd_eager cg_autograd_graph_persistent darts cg_global_ws1 selflaunch_2ranks_one_gpu_gloo; do c $g/bench_$t.json $p/r04_bench_$t.json; done
python - <<'PY'
import json, glob
rows = []
for f in sorted(glob.glob("gpurun_out/r4/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows.append("%-44s %8.1f steps/s  %7.3f ms/step  iteration %s us  arms=%s" % (f.split("bench_")[-1][:-5], d["value"], d["ms_per_step"],
                ("%.1f" % d["per_iteration_us"]) if d.get("per_iteration_us") else "-", d["config"].get("debug_arms")))
open("profiles/r04_ab_same_box_lines.txt", "w").write("# scripts/gpu_r4_full.sh, one box, one libbhg.so: every line is bench.py with the arms named (bhg_debug_set keys)\n" + "\n".join(rows) + "\n")
print("\n".join(rows))
PY
