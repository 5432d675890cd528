#!/usr/bin/env python
"""BASELINE cfg 3 shape (ResNet12(5, 32)-shaped inner, 10.43 M parameters in 122 tensors, 84 x 84 support images, prox to the upper copy, CG K=20) with
an OPAQUE inner loss (autograd double backward): the fused recurrence of this package vs the reference's
per-tensor recurrence (oracle restatement) on the same GPU.  Prints hypergradient steps/s of both."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hypergrad_oracle as horc
import zoo
from betty_amd import Config, hypergradient as hg

dev = "cuda:0"
g = torch.Generator().manual_seed(77); torch.manual_seed(77)
inner, upper = zoo.ResNet12().to(dev), zoo.ResNet12().to(dev)
for p, q in zip(inner.parameters(), upper.parameters()):
    q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g).to(dev))
x = torch.randn(25, 3, 84, 84, generator=g).to(dev); y = torch.arange(5).repeat_interleave(5).to(dev)
vector = [0.01 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
prev = zoo.StubProblem("upper", upper, config=Config())
curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=20), loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))

def bench(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)

ours = bench(lambda: hg.cg(vector, curr, prev, False))
ref = bench(lambda: horc.cg(vector, curr, prev, False))
# round 6: the batch-norm layers declared (betty_amd.nn.fuse_batchnorm_): their double backward is bhg_bn_backward_vjp
from betty_amd import nn as bnn
want = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)]).double()
n_bn = bnn.fuse_batchnorm_(inner)
got = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)]).double()
ours_bn = bench(lambda: hg.cg(vector, curr, prev, False))
print(f"ResNet-12 cfg3 CG-20 with {n_bn} declared batch-norm layers: betty_amd {ours_bn:.2f} steps/s (x{ours_bn / ref:.2f} over the reference's algorithm on "
      f"this GPU, x{ours_bn / ours:.2f} over the undeclared product); result vs the undeclared product: rel {float((got - want).norm() / want.norm()):.2e}")
# ... and the four 1 x 1 shortcut projections as matrix products (betty_amd.nn.declare_pointwise_convs_)
n_pw = bnn.declare_pointwise_convs_(inner)
got2 = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)]).double()
ours_pw = bench(lambda: hg.cg(vector, curr, prev, False))
print(f"ResNet-12 cfg3 CG-20 with {n_bn} declared batch-norm layers and {n_pw} declared 1x1 convolutions: betty_amd {ours_pw:.2f} steps/s (x{ours_pw / ref:.2f} over the "
      f"reference's algorithm on this GPU); result vs the undeclared product: rel {float((got2 - want).norm() / want.norm()):.2e}")
zoo.attach_prox_structure(curr)
ours_struct = bench(lambda: hg.cg(vector, curr, prev, False))
print(f"ResNet-12 cfg3 CG-20, opaque HVP: betty_amd {ours:.2f} steps/s | reference algorithm on the same GPU {ref:.2f} steps/s"
      f" | betty_amd with the proximal structure (closed-form mixed term) {ours_struct:.2f} steps/s")
