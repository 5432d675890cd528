#!/usr/bin/env python
"""hipGraph replay of the opaque (autograd double-backward) HVP: A/B against eager launches on the three BASELINE
configurations whose inner problem has no declared structure, same inputs, results compared.

    python scripts/hvp_graph_probe.py cfg2|cfg3|cfg3prox|cfg5 [steps]

Prints one line per arm: steps/s, captures / replays / fallbacks, and the relative difference of the two arms' results
(the replayed launches are the very kernels eager autograd runs, so the difference is ATen's own run-to-run noise)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import zoo  # noqa: E402
from betty_amd import Config, hypergradient as hg  # noqa: E402
from betty_amd.hypergradient import _common  # noqa: E402

dev = "cuda:0"
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def build():
    if which == "cfg2":
        import bench

        curr, prev, vector = bench.build(torch.device(dev), seed=0, K=20, algo="cg")
        return "cg", curr, prev, vector
    if which in ("cfg3", "cfg3prox"):
        g = torch.Generator().manual_seed(77)
        torch.manual_seed(77)
        inner, upper = zoo.ResNet12().to(dev), zoo.ResNet12().to(dev)
        for p, q in zip(inner.parameters(), upper.parameters()):
            q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g).to(dev))
        x = torch.randn(25, 3, 84, 84, generator=g).to(dev)
        y = torch.arange(5).repeat_interleave(5).to(dev)
        vector = [0.01 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
        prev = zoo.StubProblem("upper", upper, config=Config())
        curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=20), loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))
        if which == "cfg3prox":
            zoo.attach_prox_structure(curr)
        return "cg", curr, prev, vector
    if which == "cfg5":
        g = torch.Generator().manual_seed(55)
        torch.manual_seed(55)
        inner, upper = zoo.Supernet(c=16, cells=10).to(dev), zoo.ArchParams(cells=10).to(dev)
        x = torch.randn(64, 3, 32, 32, generator=g).to(dev)
        y = torch.randint(0, 10, (64,), generator=g).to(dev)
        vector = [1e-2 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
        prev = zoo.StubProblem("upper", upper, config=Config())
        curr = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=20, neumann_alpha=0.1),
                               loss_fn=zoo.make_supernet_loss(prev, 0.1), batch=(x, y))
        return "neumann", curr, prev, vector
    raise SystemExit("cfg2 | cfg3 | cfg3prox | cfg5")


algo, curr, prev, vector = build()
print(f"{which}: {algo}, T = {len(vector)} tensors, N = {sum(v.numel() for v in vector):,}", flush=True)
results = {}
for arm in ("0", "1"):
    os.environ["BHG_HVP_GRAPH"] = arm
    for k in _common.GRAPH_STATS:
        _common.GRAPH_STATS[k] = 0
    fn = lambda: hg.jvp_fn_mapping[algo](vector, curr, prev, False)  # noqa: E731
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    results[arm] = torch.cat([o.reshape(-1).double() for o in out])
    print(f"  BHG_HVP_GRAPH={arm}: {1.0 / dt:8.3f} steps/s ({1e3 * dt:9.2f} ms/step; first call {first:.2f} s)  {_common.GRAPH_STATS}", flush=True)
a, b = results["0"], results["1"]
print(f"  graph vs eager: rel diff {((a - b).norm() / a.norm()).item():.2e}, finite: {bool(torch.isfinite(b).all())}")
