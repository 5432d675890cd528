#!/usr/bin/env python
"""Host-side enqueue cost per CG iteration (bench workload): Python + ctypes + HIP launches, no device sync."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from betty_amd.backend import get_backend
curr, prev, vector = bench.build(torch.device("cuda:0"), 0)
bench.declare_structure(curr, "hip")
be = get_backend()
provider = curr.hypergradient_structure(prev)
hvp_fn = provider.prepare()
layout = be.layout(vector); x, r, p = layout.state(3)
be.cg_init(layout, vector, x, r, p)
pv = layout.views(p, vector)
for rep in range(3):
    torch.cuda.synchronize()
    th = tc = 0.0
    for k in range(20):
        t0 = time.perf_counter(); hv = hvp_fn(pv); t1 = time.perf_counter()
        be.cg_step(layout, hv, x, r, p, 1.0, k % 4, 0.0, hvp_shift=provider.hvp_shift); t2 = time.perf_counter()
        th += t1 - t0; tc += t2 - t1
    torch.cuda.synchronize()
    be.cg_init(layout, vector, x, r, p)
    print(f"per iteration: hvp_fn {1e6*th/20:.1f} us, cg_step {1e6*tc/20:.1f} us (host enqueue only)")
