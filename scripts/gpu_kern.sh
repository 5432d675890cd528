#!/bin/bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python scripts/bench_kernels.py 2>&1 | tee gpurun_out/bench_kernels.json | tail -60
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k -o k -- python $GRAFT_REPO_ROOT/scripts/bench_kernels.py --iters 30 > /tmp/prof_k.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_k; cp /tmp/prof_k/*stats*.csv gpurun_out/prof_k/ 2>/dev/null
grep -E "bhg::|Name" gpurun_out/prof_k/k_kernel_stats.csv | cut -c1-200
