#!/bin/bash
# Round 6, call (l): cfg 3 with the 1 x 1 shortcut projections declared as matrix products on top of the declared batch norm.
set -u
O=gpurun_out/r6l; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fused_batchnorm.py -m gpu -q > $O/pytest_bn.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_bn.log
timeout 900 python scripts/cfg3_resnet12_compare.py > $O/cfg3_compare.txt 2>&1; echo "compare rc=$?"; grep -v amdgpu.ids $O/cfg3_compare.txt | tail -4
cd /tmp && rm -rf /tmp/cfg3p && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3p -o t -- python $GRAFT_REPO_ROOT/scripts/cfg3_profile.py 2 fused-bn pointwise > /tmp/cfg3p.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -1 /tmp/cfg3p.log | tee $O/cfg3_step.txt
f=$(ls /tmp/cfg3p/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python scripts/print_cfg3_breakdown.py $f > $O/cfg3_breakdown_bn_pointwise.txt; head -14 $O/cfg3_breakdown_bn_pointwise.txt | cut -c1-170
