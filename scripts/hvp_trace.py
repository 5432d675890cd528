#!/usr/bin/env python
"""Run a few analytic MLP HVPs at the cfg-2 shapes (for rocprofv3 --kernel-trace)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
curr, prev, vector = bench.build(torch.device("cuda:0"), 0)
bench.declare_structure(curr, "hip")
prov = curr.hypergradient_structure(prev)
hvp = prov.prepare()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    hvp(vector)
torch.cuda.synchronize()
