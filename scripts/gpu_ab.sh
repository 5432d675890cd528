#!/bin/bash
export TMPDIR=/tmp
for V in "BHG_MLP_TN=64" "BHG_MLP_TN=32" "BHG_MLP_TN=32 BHG_SPLIT_TARGET=768" "BHG_MLP_TN=32 BHG_SPLIT_TARGET=384"; do
  echo "== [$V]"; env $V timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mlp_hvp" 2>&1 | tail -1
  for i in 1 2; do env $V timeout 300 python bench.py --cpu-steps 0 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hvp_roofline']['avg_call_us'])"; done
done
