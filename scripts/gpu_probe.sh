#!/bin/bash
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "fused or structured or mlp or cfg2" 2>&1 | tail -3
for arm in "w4" "w3 BHG_LIB=$GRAFT_REPO_ROOT/betty_amd/csrc/libbhg_w3.so" "w4b" "w3b BHG_LIB=$GRAFT_REPO_ROOT/betty_amd/csrc/libbhg_w3.so"; do
  set -- $arm; tag=$1; shift
  env "$@" timeout 300 python bench.py --steps 150 --cpu-steps 0 --no-kernel-timing 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('== $tag', round(d['value'],1), round(d['ms_per_step'],3))"
  env "$@" timeout 300 python bench.py --steps 150 --cpu-steps 0 --no-kernel-timing --algo neumann --cg-iters 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('== $tag neumann', round(d['value'],1), round(d['ms_per_step'],3))"
done
