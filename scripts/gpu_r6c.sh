#!/bin/bash
# Round 6, call (c): is the declared-batch-norm solve of cfg 3 wrong, or is CG-20 on this instance a noise amplifier?
set -u
O=gpurun_out/r6c; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python scripts/cfg3_bn_validate.py fp64 > $O/cfg3_bn_validate.txt 2>&1; echo "rc=$?"; cat $O/cfg3_bn_validate.txt | grep -v amdgpu.ids
