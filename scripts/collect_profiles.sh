#!/bin/bash
# Copy what scripts/gpu_full.sh left under gpurun_out/ (scratch) into profiles/ (tracked), named for round 2.
set -u
c() { [ -f "$1" ] && cp "$1" "$2"; }
g=gpurun_out; p=profiles
c $g/bench_default.json $p/r02_bench_default.json
c $g/prof_default/bench_kernel_stats.csv $p/r02_bench_default_kernel_stats.csv
c $g/prof_default/bench_line_under_rocprof.json $p/r02_bench_line_under_rocprof.json
c $g/pytest_gpu.log $p/r02_pytest_gpu.log
c $g/smoke.log $p/r02_smoke.log
c $g/timeline_fused.txt $p/r02_timeline_fused.txt
c $g/timeline_neumann_fused.txt $p/r02_timeline_neumann_fused.txt
c $g/outside_fused.txt $p/r02_outside_the_k_loop.txt
c $g/pmc/r02_pmc_traffic.json $p/r02_pmc_traffic.json
c $g/pmc/r02_mfma_busy.json $p/r02_mfma_busy.json
for t in cg_nofuse cg_keep_solution neumann_keep_solution cg_autograd neumann_fused neumann_nofuse darts cg_global_ws1; do c $g/bench_$t.json $p/r02_bench_$t.json; done
c $g/bench_kernels_N10M_cached.json $p/r02_bench_kernels_N10M_cached.json
c $g/bench_kernels_N10M_cache_defeated.json $p/r02_bench_kernels_N10M_cache_defeated.json
c $g/bench_kernels_N120M.json $p/r02_bench_kernels_N120M.json
c $g/bench_kernels_N120M_cache_defeated.json $p/r02_bench_kernels_N120M_cache_defeated.json
c $g/bench_kernels_N15M.json $p/r02_bench_kernels_N15M_lds_assisted.json
c $g/bench_kernels_N20M.json $p/r02_bench_kernels_N20M_hybrid_resident.json
