#!/usr/bin/env python
"""BASELINE cfg 3 (ResNet-12 inner, 10.43 M parameters / 122 tensors, 25 x 3 x 84 x 84, CG K = 20, opaque double backward): one
hypergradient step under `rocprofv3 --kernel-trace --stats` so that the time of a step can be split into MIOpen convolutions,
batch-norm / element-wise chains of the double backward, GEMMs, this library's kernels and idle gaps.

    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg3 -o t -- python scripts/cfg3_profile.py [steps]
    python scripts/print_cfg3_breakdown.py /tmp/cfg3/*kernel_trace.csv

The timed steps are separated from the warm-up (MIOpen's solver search) by a one-second host sleep: the breakdown script cuts the
trace at the last idle gap longer than 0.5 s."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import zoo  # noqa: E402
from betty_amd import Config, hypergradient as hg  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fused_bn = len(sys.argv) > 2 and sys.argv[2] == "fused-bn"   # declare the batch-norm layers (betty_amd.nn.fuse_batchnorm_)
dev = "cuda:0"
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
if fused_bn:
    from betty_amd import nn as bnn

    print("declared batch-norm layers:", bnn.fuse_batchnorm_(inner))
    if len(sys.argv) > 3 and sys.argv[3] == "pointwise":
        print("declared 1x1 convolutions:", bnn.declare_pointwise_convs_(inner))
for _ in range(2):
    hg.cg(vector, curr, prev, False)
torch.cuda.synchronize()
time.sleep(1.0)
t0 = time.perf_counter()
for _ in range(steps):
    hg.cg(vector, curr, prev, False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"cfg3 ResNet-12 CG-20 opaque{' (declared batch norm)' if fused_bn else ''}: {steps} steps in {dt:.3f} s = {steps / dt:.3f} steps/s ({1e3 * dt / steps / 21:.1f} ms per (HVP + recurrence), 21 double-backward-sized passes per step)")
