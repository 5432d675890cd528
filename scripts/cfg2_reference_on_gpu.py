#!/usr/bin/env python
"""The metric workload with the REFERENCE's algorithm (oracle restatement: opaque double backward + per-tensor ATen
recurrence) running on the MI355X through PyTorch-ROCm — "what you get by just moving Betty to the GPU" — next to
this package on the same inputs.  Prints hypergradient steps/s."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hypergrad_oracle as horc
import bench
from betty_amd import hypergradient as hg

def rate(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)

for algo, K in (("cg", 20), ("neumann", 10), ("darts", 0)):
    curr, prev, vector = bench.build(torch.device("cuda:0"), seed=0, K=max(K, 1), algo=algo)
    ref = rate(lambda: getattr(horc, algo)(vector, curr, prev, False))
    opaque = rate(lambda: hg.jvp_fn_mapping[algo](vector, curr, prev, False))
    bench.declare_structure(curr, "hip")
    ours = rate(lambda: hg.jvp_fn_mapping[algo](vector, curr, prev, False), n=30)
    print(f"{algo:8s} K={K:2d}: reference algorithm on the GPU {ref:7.1f} steps/s | betty_amd opaque HVP {opaque:7.1f} | betty_amd analytic HVP {ours:7.1f}")
