#!/bin/bash
# Round 6, call (b): fused batch-norm double backward (csrc/bhg_bn.hip, betty_amd/nn.py): kernel tests, the cfg-3 solve with declared
# layers, speed against the reference's algorithm on the same GPU, and the kernel breakdown of a declared-layers step.
set -u
O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_fused_batchnorm.py -m gpu -q -x -rP > $O/pytest_bn.log 2>&1; echo "pytest bn rc=$?"; tail -4 $O/pytest_bn.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "cfg3_resnet12" > $O/pytest_cfg3.log 2>&1; echo "pytest cfg3 rc=$?"; grep -E "resnet12 cg20|passed|failed" $O/pytest_cfg3.log | tail -6
timeout 900 python scripts/cfg3_resnet12_compare.py > $O/cfg3_compare.txt 2>&1; echo "compare rc=$?"; tail -4 $O/cfg3_compare.txt
cd /tmp && rm -rf /tmp/cfg3f && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3f -o t -- python $GRAFT_REPO_ROOT/scripts/cfg3_profile.py 2 fused-bn > /tmp/cfg3f.log 2>&1; echo "cfg3 fused rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -2 /tmp/cfg3f.log | tee $O/cfg3_fused_step.txt
f=$(ls /tmp/cfg3f/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_cfg3_breakdown.py $f | tee $O/cfg3_fused_breakdown.txt
