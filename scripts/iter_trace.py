#!/usr/bin/env python
"""Run a few hypergradient steps of the metric workload (for rocprofv3 --kernel-trace timelines).
usage: iter_trace.py [steps] [cg|neumann] [fused|nofuse] [debug_key=int ...]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from betty_amd import hypergradient as hg
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
algo = sys.argv[2] if len(sys.argv) > 2 else "cg"
fused = (sys.argv[3] if len(sys.argv) > 3 else "fused") == "fused"
from betty_amd import _native
if sys.argv[4:]:
    _native.use_ab(True)   # measurement arms live in libbhg_ab.so
for kv in sys.argv[4:]:   # measurement arms: key=int (bhg_debug_set)
    k, _, v = kv.partition("=")
    _native.debug_set(k, int(v))
curr, prev, vector = bench.build(torch.device("cuda:0"), 0, K=20 if algo == "cg" else 10, algo=algo)
bench.declare_structure(curr, "hip", fused=fused)
for _ in range(steps):
    for p in prev.parameters():
        p.grad = None
    hg.jvp_fn_mapping[algo](vector, curr, prev, True)
torch.cuda.synchronize()
