#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== resident full-size test"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "full_size or kernels_vs_oracle or deterministic or darts_eps or refuses" 2>&1 | tail -30
echo "rc=$?"
echo "== bench resident"
timeout 200 python bench.py --steps 2 --warmup 1 --variant resident --cpu-steps 0 > gpurun_out/bench_res.log 2>&1; echo "rc=$?"; tail -20 gpurun_out/bench_res.log
echo "== rocprofv3 stream"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stream -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-steps 0 --no-kernel-timing --variant stream > /tmp/prof_stream.log 2>&1; echo "rc=$?"
tail -3 /tmp/prof_stream.log
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_stream
find /tmp/prof_stream -name "*stats*.csv" -exec cp {} gpurun_out/prof_stream/ \;
ls -la /tmp/prof_stream/* | head -20
for f in gpurun_out/prof_stream/*kernel_stats.csv; do echo "--- $f"; head -30 $f; done
du -sh gpurun_out
