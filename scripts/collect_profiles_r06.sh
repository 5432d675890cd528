#!/bin/bash
# Copies what scripts/gpu_r6_final.sh left under gpurun_out/r6final/ (and the earlier calls of the round under gpurun_out/r6?/) into
# profiles/r06_* (tracked).  Run locally after the GPU call.
set -u
S=gpurun_out/r6final; P=profiles
cp $S/lib.sha $P/r06_lib.sha
grep -v "amdgpu.ids" $S/pytest_gpu_full.log > $P/r06_pytest_gpu.log
grep -v "amdgpu.ids" $S/smoke.log > $P/r06_smoke.log
for t in driver_cmd_a driver_cmd_b default driver_cmd_traffic_replayed selflaunch_2ranks_one_gpu_gloo; do
  [ -s $S/bench_$t.json ] && python -c "
import json,sys
d=json.loads(open('$S/bench_$t.json').read().strip().splitlines()[-1]); json.dump(d, open('$P/r06_bench_$t.json','w'), indent=1)"
done
cp $S/bench_default_kernel_stats.csv $P/r06_bench_default_kernel_stats.csv 2>/dev/null
[ -s $S/bench_line_under_rocprof.json ] && cp $S/bench_line_under_rocprof.json $P/r06_bench_line_under_rocprof.json
cp $S/r06_pmc_traffic.json $P/r06_pmc_traffic.json 2>/dev/null
cp $S/r06_pmc_bn_traffic.json $P/r06_pmc_bn_traffic.json 2>/dev/null
grep -v "amdgpu.ids" $S/bench_bn.txt > $P/r06_bench_bn.txt
cp $S/bench_bn_kernel_stats.csv $P/r06_bench_bn_kernel_stats.csv 2>/dev/null
cp $S/cfg3_declared_breakdown.txt $P/r06_cfg3_step_kernel_breakdown_declared_batchnorm.txt 2>/dev/null
grep -v "amdgpu.ids" $S/cfg3_compare.txt > $P/r06_cfg3_declared_batchnorm_speed.txt
grep -v "amdgpu.ids\|Warning: find_unused" $S/selfcheck_n1_rccl.txt > $P/r06_multi_gpu_selfcheck_n1_rccl.txt
grep -v "amdgpu.ids\|Warning: find_unused" $S/selfcheck_n2_gloo.txt > $P/r06_multi_gpu_selfcheck_n2_gloo_one_gpu.txt
ls -la $P/r06_* | wc -l
