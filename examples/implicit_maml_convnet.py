#!/usr/bin/env python
"""Implicit MAML on a batch-norm ConvNet (BASELINE cfg 3; the structure of the reference's examples/implicit_maml/) on synthetic
few-shot tasks.

Inner problem: a task learner (conv 3x3 + BatchNorm2d + LeakyReLU blocks, linear head) adapted on a support set with a proximal
term to the meta-initialisation.  Upper problem: the meta-initialisation, trained through the implicit hypergradient (CG, K steps).
The inner training_step stays OPAQUE — the Hessian-vector products are autograd's double backward, as in the reference (cg.py:39-41) —
but the network's batch-norm layers are DECLARED: `betty_amd.nn.fuse_batchnorm_(net)` ties them into the graph so that their share of
every product is ONE fused call (csrc/bhg_bn.hip: two launches per layer) instead of ATen's ~340-launch decomposition.

    python examples/implicit_maml_convnet.py --k 5 --iters 40 [--no-fuse]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betty_amd import Config  # noqa: E402
from betty_amd import nn as bnn  # noqa: E402
from betty_amd.engine import Engine, EngineConfig  # noqa: E402
from betty_amd.problems import ImplicitProblem  # noqa: E402


class ConvNet(nn.Module):
    def __init__(self, ways=5, width=32, blocks=4):
        super().__init__()
        chans = [3] + [width] * blocks
        self.convs = nn.ModuleList([nn.Conv2d(a, b, 3, padding=1, bias=False) for a, b in zip(chans[:-1], chans[1:])])
        self.norms = nn.ModuleList([nn.BatchNorm2d(width, track_running_stats=False) for _ in range(blocks)])
        self.fc = nn.Linear(width, ways)

    def forward(self, x):
        for conv, norm in zip(self.convs, self.norms):
            x = F.max_pool2d(F.leaky_relu(norm(conv(x)), 0.1), 2)
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))


PROX = 0.5


class Meta(ImplicitProblem):   # upper: the meta-initialisation, judged on the query set with the adapted learner
    def training_step(self, batch):
        x, y = batch
        return F.cross_entropy(self.learner.module(x), y)


class Learner(ImplicitProblem):   # inner: adaptation on the support set, proximal to the meta-initialisation
    def training_step(self, batch):
        x, y = batch
        prox = sum(((p - q) ** 2).sum() for p, q in zip(self.module.parameters(), self.meta.module.parameters()))
        return F.cross_entropy(self.module(x), y) + 0.5 * PROX * prox


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--no-fuse", action="store_true", help="leave the batch-norm layers undeclared (the A/B arm)")
    args = ap.parse_args()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    ways, shots = 5, 5
    proto = torch.randn(ways, 3, args.size, args.size, generator=g)

    def task(n):
        y = torch.arange(ways).repeat_interleave(n)
        return proto[y] + 0.7 * torch.randn(len(y), 3, args.size, args.size, generator=g), y

    support = [task(shots) for _ in range(8)]
    query = [task(shots) for _ in range(8)]
    meta_net, learner_net = ConvNet(ways), ConvNet(ways)
    learner_net.load_state_dict(meta_net.state_dict())
    n_bn = 0 if args.no_fuse else bnn.fuse_batchnorm_(learner_net)
    cfg = Config(type="cg", unroll_steps=3, cg_iterations=args.k, cg_alpha=1.0)
    upper = Meta(name="meta", module=meta_net, optimizer=torch.optim.Adam(meta_net.parameters(), lr=1e-3), train_data_loader=query, config=Config())
    inner = Learner(name="learner", module=learner_net, optimizer=torch.optim.SGD(learner_net.parameters(), lr=0.05), train_data_loader=support, config=cfg)
    engine = Engine(config=EngineConfig(train_iters=args.iters), problems=[upper, inner],
                    dependencies={"u2l": {upper: [inner]}, "l2u": {inner: [upper]}})
    calls0 = bnn.fused_batchnorm_calls()
    t0 = time.perf_counter()
    engine.run()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    calls = {k: v - calls0[k] for k, v in bnn.fused_batchnorm_calls().items()}
    with torch.no_grad():
        x, y = query[0]
        acc = (learner_net(x.to(engine.device)).argmax(1).cpu() == y).float().mean().item()
    finite = all(bool(torch.isfinite(p).all()) for p in meta_net.parameters())
    print(f"implicit MAML, ConvNet with {len(learner_net.norms)} batch-norm layers ({n_bn} declared), cg K={args.k}: {upper.count} upper steps in {dt:.2f} s, "
          f"query acc {acc:.2f}, fused double-backward calls {calls['backward_vjp']}, finite {finite}")


if __name__ == "__main__":
    main()
