#!/bin/bash
# Copy what scripts/gpu_r3_full.sh left under gpurun_out/ (scratch) into profiles/ (tracked), named for round 3.
set -u
c() { [ -f "$1" ] && cp "$1" "$2"; }
g=gpurun_out; p=profiles
c $g/bench_default.json $p/r03_bench_default.json
c $g/prof_default/bench_kernel_stats.csv $p/r03_bench_default_kernel_stats.csv
c $g/prof_default/bench_line_under_rocprof.json $p/r03_bench_line_under_rocprof.json
grep -vE "^Extension|Warning|warn" $g/pytest_gpu_full.log | grep -E "cfg2|hoisted|projected|classic|full  |wide head|Roberta|supernet|resnet|passed|failed|skipped|durations|s call|s setup" > $p/r03_pytest_gpu.log
c $g/smoke.log $p/r03_smoke.log
c $g/timeline_fused.txt $p/r03_timeline_fused_fully_projected.txt
c $g/outside_fused.txt $p/r03_outside_the_k_loop.txt
c $g/pmc/r03_pmc_traffic.json $p/r03_pmc_traffic.json
for t in neumann_fused neumann_classic_chain cg_nofuse cg_classic_chain cg_hoisted_not_projected cg_keep_solution cg_proj_two_launches cg_autograd_tunableop cg_global_ws1_classic_chain cg_autograd_graph_persistent cg_autograd_graph_per_solve cg_autograd_eager neumann_autograd_graph_persistent darts cg_global_ws1; do c $g/bench_$t.json $p/r03_bench_$t.json; done
