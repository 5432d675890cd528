#!/usr/bin/env python
"""One full bench step under rocprofv3 --kernel-trace: what runs outside the K loop."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from betty_amd import hypergradient as hg
curr, prev, vector = bench.build(torch.device("cuda:0"), 0)
bench.declare_structure(curr, "hip")
for _ in range(3):
    for p in prev.parameters(): p.grad = None
    hg.cg(vector, curr, prev, True)
torch.cuda.synchronize()
