#!/usr/bin/env python
"""Learning to reweight noisy samples with a meta-weight-net (BASELINE cfg 2; the structure of the
reference's examples/learning_to_reweight/) on synthetic CIFAR-shaped data.

Inner problem: ReLU-MLP classifier trained on noisy labels with per-sample weights s_i = MWN(CE_i).
Upper problem: the MWN, trained on a small clean validation set through the implicit hypergradient.
The inner problem declares its structure, so the K Hessian-vector products of every hypergradient run
on the fp32 matrix-core kernels (betty_amd/csrc/bhg_mlp.hip) instead of PyTorch's double backward.

    python examples/learning_to_reweight_mlp.py --algo neumann --k 10 --iters 300
"""
import argparse
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betty_amd import Config  # noqa: E402
from betty_amd.engine import Engine, EngineConfig  # noqa: E402
from betty_amd.hypergradient.structured import SigmoidMLPWeightNet, WeightedCEMLP  # noqa: E402
from betty_amd.problems import ImplicitProblem  # noqa: E402


class MLP(nn.Module):
    def __init__(self, sizes):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(sizes[:-1], sizes[1:])])

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i + 1 < len(self.layers):
                x = F.relu(x)
        return x


class MWN(nn.Module):
    def __init__(self, hidden=100):
        super().__init__()
        self.l1, self.l2 = nn.Linear(1, hidden), nn.Linear(hidden, 1)

    def forward(self, x):
        return torch.sigmoid(self.l2(F.relu(self.l1(x))))


RIDGE = 1e-3


class Reweight(ImplicitProblem):  # upper
    def training_step(self, batch):
        x, y = batch
        return F.cross_entropy(self.classifier.module(x), y)


class Classifier(ImplicitProblem):  # inner
    def training_step(self, batch):
        x, y = batch
        ce = F.cross_entropy(self.module(x), y, reduction="none")
        w = self.reweight(ce.detach().reshape(-1, 1)).reshape(-1)
        return torch.mean(w * ce) + RIDGE * sum((p * p).sum() for p in self.module.parameters())

    def hypergradient_structure(self, prev):
        # the inner MLP and — optionally — the meta-weight-net are DECLARED: the K Hessian-vector products run on the matrix-core kernels,
        # the sample weights and their VJP to the MWN's parameters in closed form (csrc/bhg_mwn.hip); both declarations are checked against
        # autograd on first use.  (Under a data-parallel strategy add average_over=True: the mean DDP's reducer would have taken.)
        return WeightedCEMLP(self, prev, layers=list(self.module.layers),
                             weight_fn=lambda ce: prev(ce.reshape(-1, 1)), ridge=RIDGE,
                             weight_net=SigmoidMLPWeightNet(prev.module.l1, prev.module.l2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="neumann", choices=["cg", "neumann"])
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--sizes", default="3072,512,256,10")
    args = ap.parse_args()
    torch.manual_seed(0)
    sizes = [int(s) for s in args.sizes.split(",")]
    teacher = MLP([sizes[0], 64, sizes[-1]])
    g = torch.Generator().manual_seed(1)

    def make(n, noise):
        x = torch.randn(n, sizes[0], generator=g)
        y = teacher(x).argmax(1)
        flip = torch.rand(n, generator=g) < noise
        return x, torch.where(flip, torch.randint(0, sizes[-1], (n,), generator=g), y)

    train = [make(100, 0.4) for _ in range(20)]   # 40 % label noise, batch 100
    clean = [make(100, 0.0) for _ in range(5)]
    cfg = Config(type=args.algo, unroll_steps=5, cg_iterations=args.k, cg_alpha=1.0, neumann_iterations=args.k, neumann_alpha=0.1)
    net, mwn = MLP(sizes), MWN(100)
    upper = Reweight(name="reweight", module=mwn, optimizer=torch.optim.Adam(mwn.parameters(), lr=1e-3),
                     train_data_loader=clean, config=Config())
    inner = Classifier(name="classifier", module=net, optimizer=torch.optim.SGD(net.parameters(), lr=0.05),
                       train_data_loader=train, config=cfg)
    engine = Engine(config=EngineConfig(train_iters=args.iters), problems=[upper, inner],
                    dependencies={"u2l": {upper: [inner]}, "l2u": {inner: [upper]}})
    engine.run()
    with torch.no_grad():
        x, y = clean[0]
        acc = (net(x.to(engine.device)).argmax(1).cpu() == y).float().mean().item()
        w_clean = mwn(torch.tensor([[0.1]], device=engine.device)).item()
        w_noisy = mwn(torch.tensor([[3.0]], device=engine.device)).item()
    print(f"algo={args.algo} K={args.k}: {upper.count} upper steps, clean acc {acc:.2f}, "
          f"MWN weight at CE=0.1: {w_clean:.3f}, at CE=3.0: {w_noisy:.3f}")


if __name__ == "__main__":
    main()
