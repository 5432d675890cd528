#!/bin/bash
# N > 1 code path of bench.py on a 1-GPU box: 2 ranks share the GPU (gloo), replica and global modes.
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out
export BHG_ALL_RANKS_ON_GPU0=1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 2 --dist-backend gloo --cpu-steps 0 > gpurun_out/bench_2ranks_replica.json 2> gpurun_out/bench_2ranks_replica.err; echo "replica rc=$?"; cut -c1-600 gpurun_out/bench_2ranks_replica.json; tail -3 gpurun_out/bench_2ranks_replica.err | cut -c1-300
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --cpu-steps 0 --mode global > gpurun_out/bench_2ranks_global.json 2> gpurun_out/bench_2ranks_global.err; echo "global rc=$?"; cut -c1-600 gpurun_out/bench_2ranks_global.json; tail -3 gpurun_out/bench_2ranks_global.err | cut -c1-300
