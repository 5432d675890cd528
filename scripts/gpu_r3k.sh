#!/bin/bash
# N > 1 code path of bench.py on a 1-GPU box: two ranks share GPU 0, gloo (a CODE-PATH check, not a scaling number).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
for mode in replica global; do
BHG_ALL_RANKS_ON_GPU0=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 30 --warmup 3 --cpu-steps 0 --dist-backend gloo --mode $mode > $O/r3k_bench_2ranks_$mode.json 2> $O/r3k_bench_2ranks_$mode.err; echo "$mode rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$O/r3k_bench_2ranks_$mode.json").read().strip().splitlines()[-1])
    print("== 2 ranks on one GPU (gloo) $mode: value %.1f steps/s n_gpus %d ms/step %.3f parallelism: %s" % (d["value"], d["n_gpus"], d["ms_per_step"], d["config"]["parallelism"][:60]))
except Exception as e:
    print("failed:", e); print(open("$O/r3k_bench_2ranks_$mode.err").read()[-1500:])
PY
done
