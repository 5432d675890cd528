#!/bin/bash
# A/B sweep of the split-K sizing / tile-width switches under the one-pass solver (bench lines only).
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift
  timeout 300 python bench.py --steps 100 --cpu-steps 0 --no-kernel-timing "$@" 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-28s value %.1f steps/s  ms/step %.3f" % ("$tag", d["value"], d["ms_per_step"]))
except Exception as e:
    print("== $tag bench failed:", e); print(open("gpurun_out/bench_$tag.err").read()[-800:])
PY
}
run base
BHG_SPLIT_TARGET=512 run split512
BHG_SPLIT_TARGET=640 run split640
BHG_SPLIT_TARGET=1024 BHG_SPLIT_CAP=24 run split1024
BHG_MLP_TN=64 BHG_SPLIT_TARGET=512 BHG_SPLIT_CAP=24 run tn64_512
BHG_MLP_TN=64 BHG_SPLIT_TARGET=768 BHG_SPLIT_CAP=32 run tn64_768
BHG_NO_NT_SLABS=1 run no_nt
run base_again
