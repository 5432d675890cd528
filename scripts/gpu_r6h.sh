#!/bin/bash
# Round 6, call (h): the batch-norm kernels after the re-slicing (no division in the loops): tests, bandwidth, cfg-3 speed.
set -u
O=gpurun_out/r6h; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_fused_batchnorm.py -m gpu -q > $O/pytest_bn.log 2>&1; echo "pytest bn rc=$?"; tail -2 $O/pytest_bn.log
timeout 600 python scripts/bench_bn.py > $O/bench_bn.txt 2>&1; echo "bench_bn rc=$?"; grep -v amdgpu $O/bench_bn.txt | head -7
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -rP -k "cfg3_resnet12_cg20_with_declared" > $O/pytest_cfg3.log 2>&1; echo "pytest cfg3 rc=$?"; grep -E "^resnet12|passed|failed" $O/pytest_cfg3.log | tail -3
timeout 900 python scripts/cfg3_resnet12_compare.py > $O/cfg3_compare.txt 2>&1; echo "compare rc=$?"; grep -v amdgpu.ids $O/cfg3_compare.txt | tail -3
cd /tmp && rm -rf /tmp/cfg3f && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3f -o t -- python $GRAFT_REPO_ROOT/scripts/cfg3_profile.py 2 fused-bn > /tmp/cfg3f.log 2>&1; echo "cfg3 fused rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -1 /tmp/cfg3f.log | tee $O/cfg3_fused_step.txt
f=$(ls /tmp/cfg3f/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_cfg3_breakdown.py $f > $O/cfg3_fused_breakdown.txt; head -3 $O/cfg3_fused_breakdown.txt; grep -A7 "batch-norm" $O/cfg3_fused_breakdown.txt
