#!/usr/bin/env python
"""Logistic-regression hyper-parameter optimisation (BASELINE cfg 1; the scenario of the reference's
examples/logistic_regression_hpo/ and test/test_regression.py) on the MI355X backend.

Two-level problem: the inner problem fits weights w with a per-weight L2 penalty lam; the outer problem
tunes lam on validation data through the implicit hypergradient (cg / neumann / darts / sama).

    python examples/logistic_regression_hpo.py --algo cg --analytic
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betty_amd import Config  # noqa: E402
from betty_amd.engine import Engine, EngineConfig  # noqa: E402
from betty_amd.problems import ImplicitProblem  # noqa: E402


class Weights(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(d))

    def forward(self, x):
        return x @ self.w, self.w


class Penalty(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.lam = torch.nn.Parameter(torch.ones(d))

    def forward(self):
        return self.lam * 1.0


class Outer(ImplicitProblem):
    def training_step(self, batch):
        x, y = batch
        return F.binary_cross_entropy_with_logits(self.inner(x)[0], y)

    def param_callback(self):
        for p in self.trainable_parameters():
            p.data.clamp_(min=1e-8)


class Inner(ImplicitProblem):
    def training_step(self, batch):
        x, y = batch
        logits, w = self.module(x)
        return F.binary_cross_entropy_with_logits(logits, y) + 0.5 * (self.outer() * w * w).sum()

    def on_inner_loop_start(self):
        self.module.w.data.zero_()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--algo", default="cg", choices=["cg", "neumann", "darts", "sama"])
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--analytic", action="store_true", help="use the closed-form logistic HVP kernels")
    args = ap.parse_args()
    rng = np.random.RandomState(0)
    torch.manual_seed(0)
    d = args.dim
    w_gt = rng.randn(d)
    x = rng.randn(1000, d)
    y = ((x @ w_gt + 0.1 * rng.randn(1000)) > 0).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    cfg = {
        "cg": Config(type="cg", cg_iterations=5, cg_alpha=1.0, unroll_steps=100),
        "neumann": Config(type="neumann", neumann_iterations=5, neumann_alpha=0.5, unroll_steps=100),
        "darts": Config(type="darts", unroll_steps=100),
        "sama": Config(type="sama", unroll_steps=100),
    }[args.algo]
    penalty, weights = Penalty(d), Weights(d)
    outer = Outer(name="outer", module=penalty, optimizer=torch.optim.SGD(penalty.parameters(), lr=1.0, momentum=0.9),
                  train_data_loader=[(t(x[500:]), t(y[500:]))], config=Config())
    inner = Inner(name="inner", module=weights, optimizer=torch.optim.SGD(weights.parameters(), lr=0.1),
                  train_data_loader=[(t(x[:500]), t(y[:500]))], config=cfg)
    if args.analytic and args.algo in ("cg", "neumann"):
        from betty_amd.hypergradient.structured import LogisticRegressionL2

        inner.hypergradient_structure = lambda prev: LogisticRegressionL2(inner, prev, weights.w, lam_fn=lambda: prev())
    engine = Engine(config=EngineConfig(train_iters=args.iters), problems=[outer, inner],
                    dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}})
    engine.run()
    val = outer.training_step(outer.cur_batch).item()
    print(f"algo={args.algo} analytic={args.analytic}  final validation loss {val:.4f}  (reference's test threshold: < 0.48 at dim 20)")


if __name__ == "__main__":
    main()
