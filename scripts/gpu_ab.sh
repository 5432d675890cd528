#!/bin/bash
export TMPDIR=/tmp
for i in 1 2; do
for V in "BHG_SPLIT_TARGET=512 BHG_SPLIT_CAP=16" "BHG_SPLIT_TARGET=768 BHG_SPLIT_CAP=48" "BHG_SPLIT_TARGET=1024 BHG_SPLIT_CAP=48" "BHG_SPLIT_TARGET=256 BHG_SPLIT_CAP=8"; do
  echo "== [$V]"; env $V timeout 300 python bench.py --cpu-steps 0 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hvp_roofline']['avg_call_us'])"
done; done
