#!/usr/bin/env python
"""Wall/GPU time of the phases of one structured CG hypergradient step (bench workload)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from betty_amd.backend import get_backend
curr, prev, vector = bench.build(torch.device("cuda:0"), 0)
bench.declare_structure(curr, "hip")
be = get_backend()
def ev(): 
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
acc = {}
for it in range(8):
    for p in prev.parameters(): p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = [("start", ev())]
    provider = curr.hypergradient_structure(prev)
    hvp_fn = provider.prepare(); marks.append(("prepare", ev()))
    layout = be.layout(vector); x, r, p = layout.state(3)
    be.cg_init(layout, vector, x, r, p); marks.append(("cg_init", ev()))
    pv = layout.views(p, vector)
    for k in range(20):
        hv = hvp_fn(pv)
        be.cg_step(layout, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == 19 else 0.0), hvp_shift=provider.hvp_shift)
    marks.append(("loop", ev()))
    provider.mixed_vjp(layout.views(x, vector), True); marks.append(("mixed_vjp", ev()))
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    if it >= 3:
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc.setdefault(n1, []).append(e0.elapsed_time(e1))
        acc.setdefault("host_enqueue_ms", []).append(1e3 * t_host)
        acc.setdefault("wall_ms", []).append(1e3 * t_all)
print({k: round(sum(v) / len(v), 3) for k, v in acc.items()})
