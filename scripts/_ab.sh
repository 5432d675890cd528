#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mlp or structured" 2>&1 | tail -2
for i in 1 2 3; do
for V in "BHG_X=new" "BHG_LIB=$PWD/betty_amd/csrc/libbhg_base.so"; do
  echo "== [$V]"; env $V timeout 300 python bench.py --cpu-steps 0 --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['hvp_roofline']['avg_call_us'], d['roofline']['avg_launch_us'], d['config']['finite'])"
done; done
