#!/usr/bin/env python
"""cfg 3 (ResNet-12, CG) with DECLARED batch-norm layers (betty_amd.nn) against the undeclared network on the same GPU:
  (1) one Hessian-vector product, both ways, and each way twice (run-to-run noise);
  (2) the CG solve at K = 1, 2, 5, 10, 20: how a rounding-level difference between two valid products grows with the horizon;
  (3) [fp64] the same solves in float64 (ATen's own convolutions; slow) — the truth both fp32 solves are measured against."""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import hypergrad_oracle as horc  # noqa: E402
import zoo  # noqa: E402
from betty_amd import Config, hypergradient as hg, nn as bnn  # noqa: E402

dev = "cuda:0"
want64 = "fp64" in sys.argv[1:]
torch.backends.cudnn.deterministic = True


def case(dtype=torch.float32, fused=False, K=20):
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    inner, upper = zoo.ResNet12(), zoo.ResNet12()
    for p, q in zip(inner.parameters(), upper.parameters()):
        q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g))
    inner, upper = inner.to(dev, dtype), upper.to(dev, dtype)
    x = torch.randn(25, 3, 84, 84, generator=g).to(dev, dtype)
    y = torch.arange(5).repeat_interleave(5).to(dev)
    vector = [(0.01 * torch.randn(p.shape, generator=g)).to(dev, dtype) for p in inner.parameters()]
    if fused:
        bnn.fuse_batchnorm_(inner)
    if fused == "pointwise":   # ... and the 1 x 1 shortcut projections as matrix products
        bnn.declare_pointwise_convs_(inner)
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=K, cg_alpha=1.0), loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))
    return curr, prev, vector


def flat(ts):
    return torch.cat([t.reshape(-1).double() for t in ts])


def rel(a, b):
    return float((a - b).norm() / b.norm())


def hvp(curr, vector):
    params = list(curr.module.parameters())
    loss = curr.training_step_exec(curr.cur_batch)
    grads = torch.autograd.grad(loss, params, create_graph=True)
    return flat(torch.autograd.grad(grads, params, grad_outputs=vector))


res = {}
for fused in (False, True):
    curr, prev, vector = case(fused=fused)
    h1, h2 = hvp(curr, vector), hvp(curr, vector)
    res[fused] = h1
    print(f"one H v, {'declared' if fused else 'undeclared'} batch norm: run-to-run {rel(h2, h1):.2e}", flush=True)
print(f"one H v: declared vs undeclared batch norm: rel {rel(res[True], res[False]):.2e}", flush=True)
if want64:
    curr, prev, vector = case(torch.float64)
    t0 = time.time()
    h64 = hvp(curr, vector)
    print(f"one H v in float64 ({time.time() - t0:.1f} s): undeclared fp32 {rel(res[False], h64):.2e}, declared fp32 {rel(res[True], h64):.2e} from it", flush=True)

curr, prev, vector = case(fused="pointwise")
hpw = hvp(curr, vector)
print(f"one H v: declared batch norm + 1x1 convolutions as matrix products vs undeclared: rel {rel(hpw, res[False]):.2e}" +
      (f"; from the float64 product {rel(hpw, h64):.2e}" if want64 else ""), flush=True)
for K in (1, 2, 5, 10, 20):
    out = {}
    for fused in (False, True, "pointwise"):
        curr, prev, vector = case(fused=fused, K=K)
        out[fused] = flat(hg.cg(vector, curr, prev, False))
    curr, prev, vector = case(K=K)
    chk = flat(horc.cg(vector, curr, prev, False))
    line = f"CG K={K:2d}: declared vs undeclared {rel(out[True], out[False]):.2e} (+ pointwise: {rel(out['pointwise'], out[False]):.2e}); undeclared vs the reference's algorithm on this GPU {rel(out[False], chk):.2e}, declared vs it {rel(out[True], chk):.2e}"
    if want64 and K in (5, 20):
        curr, prev, vector = case(torch.float64, K=K)
        t0 = time.time()
        truth = flat(horc.cg(vector, curr, prev, False))
        line += (f" | float64 truth ({time.time() - t0:.0f} s): reference algorithm fp32 {rel(chk, truth):.2e}, undeclared {rel(out[False], truth):.2e}, declared {rel(out[True], truth):.2e}, "
                 f"declared + pointwise {rel(out['pointwise'], truth):.2e}")
    print(line, flush=True)
