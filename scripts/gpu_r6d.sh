#!/bin/bash
# Round 6, call (d): fused batch norm after the fp64 element-wise change and F.batch_norm's own dispatch (tests, speed, one product vs
# float64), the undeclared drop-in (auto_structure), the multi-GPU self-check at N = 1 (RCCL) and N = 2 (gloo, one GPU), cfg 5 at K = 3.
set -u
O=gpurun_out/r6d; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
timeout 900 python -m pytest tests/test_fused_batchnorm.py -m gpu -q -rP > $O/pytest_bn.log 2>&1; echo "pytest bn rc=$?"; grep -E "FusedBatchNorm2d vs|passed|failed" $O/pytest_bn.log | tail -4
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "cfg3_resnet12" > $O/pytest_cfg3.log 2>&1; echo "pytest cfg3 rc=$?"; grep -E "^resnet12|passed|failed" $O/pytest_cfg3.log | tail -6
timeout 900 python -m pytest tests/test_dropin_reference.py -m gpu -q -x -rP > $O/pytest_dropin.log 2>&1; echo "pytest dropin rc=$?"; grep -E "^reference Engine|passed|failed|Error" $O/pytest_dropin.log | tail -8
timeout 600 python scripts/multi_gpu_selfcheck.py > $O/selfcheck_n1_rccl.txt 2>&1; echo "selfcheck n1 rc=$?"; grep SELFCHECK $O/selfcheck_n1_rccl.txt
timeout 600 python scripts/multi_gpu_selfcheck.py --gpus 2 --backend gloo --all-on-gpu0 > $O/selfcheck_n2_gloo_one_gpu.txt 2>&1; echo "selfcheck n2 rc=$?"; grep SELFCHECK $O/selfcheck_n2_gloo_one_gpu.txt
timeout 900 python scripts/cfg3_resnet12_compare.py > $O/cfg3_compare.txt 2>&1; echo "compare rc=$?"; grep -v amdgpu.ids $O/cfg3_compare.txt | tail -4
timeout 900 python scripts/cfg3_bn_validate.py fp64 > $O/cfg3_bn_validate.txt 2>&1; echo "validate rc=$?"; grep -v amdgpu.ids $O/cfg3_bn_validate.txt
timeout 900 python scripts/cfg5_oracle_on_gpu.py 3 deterministic > $O/cfg5_k3_same_gpu.txt 2>&1; echo "cfg5 k3 rc=$?"; grep -v amdgpu.ids $O/cfg5_k3_same_gpu.txt | tail -4
