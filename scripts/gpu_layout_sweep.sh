#!/bin/bash
# Code placement of the K-loop kernels (csrc/bhg_mlp.hip: k_layout_anchor, BHG_LAYOUT_PAD).  Build the sixteen variants first, locally:
#   for k in $(seq 1 16); do hipcc <CXXFLAGS of the Makefile> -DBHG_LAYOUT_PAD=$k -c bhg_mlp.hip -o /tmp/pad_$k.o; hipcc -shared ... -o betty_amd/csrc/ab_pad/$k/libbhg.so; done
# (betty_amd/csrc/ab_pad/ is git-ignored through *.so and travels to the GPU box), then on the box: the driver's command per variant, twice.
set -u
mkdir -p gpurun_out/r6pad
for pass in 1 2; do for k in $(seq 1 16); do
  export BHG_LIB=$GRAFT_REPO_ROOT/betty_amd/csrc/ab_pad/$k/libbhg.so
  timeout 300 python bench.py --gpus 1 --steps 40 --warmup 5 --cpu-steps 0 --no-parity --no-secondary --reps 3 > /tmp/b.json 2>/tmp/b.err
  python -c "
import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('PAD $k pass $pass', round(d['value'],1),round(d['roofline']['avg_launch_us'],2),round(d['roofline']['frac'],4))"
done; done | tee gpurun_out/r6pad/sweep.txt
