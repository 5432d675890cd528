#!/usr/bin/env python
"""BASELINE cfg 5 at the example's scale (mixed-op supernet, width 16, 10 cells = 1,545 tensors, batch 64 x 3 x 32 x 32):
ONE opaque Hessian-vector product (Neumann K = 1) under `rocprofv3 --kernel-trace --stats`, to say where its ~4.7 s go —
host launches or device kernels — and which kernels.  Prints wall time and the GPU time the torch profiler-free events see."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zoo  # noqa: E402
from betty_amd import Config, hypergradient as hg  # noqa: E402

dev = "cuda:0"
g = torch.Generator().manual_seed(55)
torch.manual_seed(55)
inner, upper = zoo.Supernet(c=16, cells=10).to(dev), zoo.ArchParams(cells=10).to(dev)
x = torch.randn(64, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (64,), generator=g).to(dev)
vector = [1e-2 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
prev = zoo.StubProblem("upper", upper, config=Config())
curr = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=1, neumann_alpha=0.1),
                       loss_fn=zoo.make_supernet_loss(prev, 0.1), batch=(x, y))
for rep in range(2):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    hg.neumann(vector, curr, prev, False)
    e1.record()
    t_host = time.perf_counter() - t0      # host returns when everything is enqueued
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"rep {rep}: neumann K=1 (gradient with graph + 1 HVP + mixed VJP): enqueue {t_host:.2f} s, complete {t_all:.2f} s, "
          f"GPU span {e0.elapsed_time(e1) / 1e3:.2f} s", flush=True)
