#!/bin/bash
# One gpurun call: GPU tests, smoke, bench, rocprofv3 kernel trace.  Outputs under gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo/nproc"; nproc; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee gpurun_out/smoke.log
echo "== bench (auto)"
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | tail -3 | tee gpurun_out/bench_auto.log
echo "== bench (stream)"
timeout 600 python bench.py --steps 20 --warmup 3 --variant stream --cpu-steps 0 2>&1 | tail -3 | tee gpurun_out/bench_stream.log
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_auto -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-steps 0 --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/prof_auto.log 2>&1
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_stream -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-steps 0 --no-kernel-timing --variant stream > $GRAFT_REPO_ROOT/gpurun_out/prof_stream.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out -name "*kernel_stats*" | head; 
for f in $(find gpurun_out -name "*kernel_stats.csv"); do echo "--- $f"; head -25 $f; done
# drop the big traces, keep stats
find gpurun_out -name "*kernel_trace.csv" -size +20M -delete
