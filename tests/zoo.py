"""Seeded synthetic bilevel problems shared by the golden generator, the oracle tests and the
GPU parity tests.

Every case is a pair of duck-typed problems exposing exactly the surface the reference's
``cg`` / ``neumann`` / ``darts`` touch (SURVEY.md §8b "Object surface"): ``paths``, ``config``,
``cur_batch``, ``training_step_exec``, ``trainable_parameters``, ``parameters``,
``meta_trainable_parameters``, ``_strategy``, ``set_grads``.  The same objects run through
  * the real reference (tests/golden/make_golden.py, in the build container only),
  * the CPU oracle (oracle/hypergrad_oracle.py),
  * the HIP path (betty_amd.hypergradient) on the GPU box.
Inputs (weights, data, direction vector) are stored in the golden files so all three see
bit-identical numbers.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Callable, Dict, List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class StubProblem:
    """Minimal stand-in for betty's ``ImplicitProblem`` (implicit_problem.py:80-84,
    problem.py:327-332,583-597,850-854)."""

    def __init__(self, name, module, config=None, loss_fn=None, batch=None):
        self.name = name
        self.module = module
        self.config = config
        self._loss_fn = loss_fn
        self.cur_batch = batch
        self.paths = []
        self._strategy = "default"
        self.fwd = module  # what `training_step` calls; a DDP wrapper in the distributed tests
        self.optimizer = None

    # SAMA's preconditioner reads the inner optimizer's state (problem.py:697-722)
    def get_opt_param_group_for_param(self, param):
        for group in self.optimizer.param_groups:
            for p in group["params"]:
                if param is p:
                    return group

    def get_opt_state_for_param(self, param):
        return self.optimizer.state[param]

    def synchronize_params(self, params, all_reduce=False):
        pass

    def training_step_exec(self, batch):
        return self._loss_fn(self, batch)

    def parameters(self):
        return list(self.module.parameters())

    def trainable_parameters(self):
        return list(self.module.parameters())

    def meta_trainable_parameters(self):
        return self.trainable_parameters()

    def set_grads(self, params, grads):
        for param, grad in zip(params, grads):
            if grad is not None:
                if getattr(param, "grad", None) is not None:
                    param.grad = param.grad + grad
                else:
                    param.grad = grad


# ---------------------------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------------------------
class Vec(nn.Module):
    """A bare parameter vector; forward returns a NON-leaf (``w * 1``) so the module also works
    under DistributedDataParallel (SURVEY.md §8e caveat)."""

    def __init__(self, n, fill=0.0):
        super().__init__()
        self.w = nn.Parameter(torch.full((n,), float(fill)))

    def forward(self):
        return self.w * 1.0


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
    """Meta-weight-net 1 -> h -> 1 with sigmoid output (examples/learning_to_reweight/model.py:98-111)."""

    def __init__(self, hidden):
        super().__init__()
        self.l1 = nn.Linear(1, hidden)
        self.l2 = nn.Linear(hidden, 1)

    def forward(self, x):
        return torch.sigmoid(self.l2(F.relu(self.l1(x))))


class SmallConv(nn.Module):
    def __init__(self, ways=5):
        super().__init__()
        self.c1 = nn.Conv2d(3, 8, 3, padding=1)
        self.c2 = nn.Conv2d(8, 8, 3, padding=1)
        self.fc = nn.Linear(8 * 4 * 4, ways)

    def forward(self, x):
        x = F.max_pool2d(torch.tanh(self.c1(x)), 2)
        x = F.max_pool2d(torch.tanh(self.c2(x)), 2)
        return self.fc(x.flatten(1))


class _Res12Unit(nn.Module):
    """Three 3x3 conv + BN (batch statistics) + LeakyReLU; the first unit of a stage widens (1x1 projection + BN on
    the shortcut) and halves the resolution with a 2x2 max-pool."""

    def __init__(self, cin, cout, first):
        super().__init__()
        chans = [cin, cout, cout, cout]
        self.convs = nn.ModuleList([nn.Conv2d(a, b, 3, padding=1, bias=False) for a, b in zip(chans[:-1], chans[1:])])
        self.norms = nn.ModuleList([nn.BatchNorm2d(cout, track_running_stats=False) for _ in range(3)])
        self.first = first
        if first:
            self.proj = nn.Conv2d(cin, cout, 1, bias=False)
            self.proj_norm = nn.BatchNorm2d(cout, track_running_stats=False)

    def forward(self, x):
        y = x
        for i, (conv, norm) in enumerate(zip(self.convs, self.norms)):
            y = norm(conv(y))
            if i < 2:
                y = F.leaky_relu(y, 0.1)
        short = self.proj_norm(self.proj(x)) if self.first else x
        y = F.leaky_relu(y + short, 0.1)
        return F.max_pool2d(y, 2) if self.first else y


class ResNet12(nn.Module):
    """The few-shot ResNet-12 of BASELINE.json cfg 3 at the example's own size, ``ResNet12(ways, 32)`` of
    examples/implicit_maml/models.py:411-483: 4 stages x 3 units x 3 convs, widths 32-80-160-320 ("wider"), a 1x1
    projection per stage, 5x5 average pool, linear classifier = 10,430,533 parameters in 122 tensors, for 84 x 84
    inputs — full-size GPU parity only, no golden file.  (Own restatement of the shape; dropout / DropBlock are off in
    the example's configuration and omitted.)"""

    def __init__(self, ways=5, hidden=32):
        super().__init__()
        widths = [hidden, int(hidden * 2.5), hidden * 5, hidden * 10]
        units, cin = [], 3
        for c in widths:
            for u in range(3):
                units.append(_Res12Unit(cin if u == 0 else c, c, first=(u == 0)))
            cin = c
        self.units = nn.ModuleList(units)
        self.fc = nn.Linear(widths[-1], ways)

    def forward(self, x):
        for unit in self.units:
            x = unit(x)
        return self.fc(F.adaptive_avg_pool2d(x, 1).flatten(1))


class TokenClassifier(nn.Module):
    """RoBERTa-base-shaped encoder (BASELINE.json cfg 4): vocab 50265, 514 positions, 12 pre-norm layers
    of width 768 / 12 heads / FFN 3072, 2-way head = 124.1 M parameters.  Synthetic tokens, no dropout."""

    def __init__(self, vocab=50265, width=768, layers=12, heads=12, ffn=3072, positions=514, classes=2):
        super().__init__()
        self.tok = nn.Embedding(vocab, width)
        self.pos = nn.Embedding(positions, width)
        self.norm = nn.LayerNorm(width)
        layer = nn.TransformerEncoderLayer(width, heads, ffn, dropout=0.0, activation="gelu", batch_first=True, norm_first=True)
        self.encoder = nn.TransformerEncoder(layer, layers, enable_nested_tensor=False)
        self.head = nn.Linear(width, classes)

    def forward(self, tokens):
        pos = torch.arange(tokens.shape[1], device=tokens.device)
        h = self.norm(self.tok(tokens) + self.pos(pos)[None])
        return self.head(self.encoder(h)[:, 0])


def roberta_seqcls(layers=12, hidden=768, heads=12, ffn=3072, vocab=50265, positions=514):
    """BASELINE.json cfg 4's inner model AS NAMED: transformers' ``RobertaForSequenceClassification`` built offline from a
    config (examples/bert_data_reweighting/model.py:11-32 loads the same class from a checkpoint):
    roberta-base = 124,647,170 parameters in 201 tensors.  Dropout is switched off — a finite-difference hypergradient
    evaluates the loss three times and needs the same function each time.  Reduced ``layers`` / sizes give the CPU-sized
    variant of the distributed tests."""
    from transformers import RobertaConfig, RobertaForSequenceClassification

    cfg = RobertaConfig(vocab_size=vocab, max_position_embeddings=positions, type_vocab_size=1, num_labels=2, hidden_size=hidden,
                        num_hidden_layers=layers, num_attention_heads=heads, intermediate_size=ffn,
                        hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    return RobertaForSequenceClassification(cfg)


class RobertaInner(nn.Module):
    """forward(batch) -> logits, as BertModel.forward does with (seqs, masks, segments) (model.py:22-32)."""

    def __init__(self, **kw):
        super().__init__()
        self.bert = roberta_seqcls(**kw)

    def forward(self, batch):
        seqs, masks, segments = batch
        return self.bert(input_ids=seqs, attention_mask=masks, token_type_ids=segments).logits


def make_roberta_reweight_loss(upper):
    """examples/bert_data_reweighting/main.py:118-128: per-sample CE weighted by the reweighting net of its detached value."""

    def loss(self, batch):
        seqs, masks, segments, labels = batch
        lv = F.cross_entropy(self.fwd((seqs, masks, segments)).view(-1, 2), labels, reduction="none").reshape(-1, 1)
        w = upper.fwd(lv.detach())
        return torch.mean(w * lv)

    return loss


class _MixedOp(nn.Module):
    """Softmax(alpha)-weighted sum of candidate ops (DARTS search space, reduced)."""

    def __init__(self, c):
        super().__init__()
        self.ops = nn.ModuleList([
            nn.Sequential(nn.Conv2d(c, c, 3, padding=1, bias=False), nn.BatchNorm2d(c, track_running_stats=False)),
            nn.Sequential(nn.Conv2d(c, c, 3, padding=1, groups=c, bias=False), nn.Conv2d(c, c, 1, bias=False),
                          nn.BatchNorm2d(c, track_running_stats=False)),
            nn.Sequential(nn.Conv2d(c, c, 5, padding=2, groups=c, bias=False), nn.Conv2d(c, c, 1, bias=False),
                          nn.BatchNorm2d(c, track_running_stats=False)),
            nn.AvgPool2d(3, stride=1, padding=1),
            nn.Identity(),
        ])

    def forward(self, x, w):
        return sum(w[i] * op(x) for i, op in enumerate(self.ops))


class Supernet(nn.Module):
    """Mixed-op supernet (BASELINE.json cfg 5 shape): `cells` cells of 4 nodes, every node sums mixed ops
    over all earlier nodes (14 edges x 5 candidate ops per cell).  forward(x, alphas[cells][14][5])."""

    N_OPS, N_EDGES = 5, 14

    def __init__(self, c=32, cells=4, classes=10):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, c, 3, padding=1, bias=False), nn.BatchNorm2d(c, track_running_stats=False))
        self.cells = nn.ModuleList([nn.ModuleList([_MixedOp(c) for _ in range(self.N_EDGES)]) for _ in range(cells)])
        self.fc = nn.Linear(c, classes)

    def forward(self, x, alphas):
        s0 = s1 = self.stem(x)
        for cell, a in zip(self.cells, alphas):
            w = torch.softmax(a, dim=-1)
            states, e = [s0, s1], 0
            for _ in range(4):
                nxt = 0
                for h in states:
                    nxt = nxt + cell[e](h, w[e])
                    e += 1
                states.append(nxt)
            s0, s1 = s1, sum(states[2:]) / 4.0
        return self.fc(s1.mean(dim=(2, 3)))


class ArchParams(nn.Module):
    def __init__(self, cells=4):
        super().__init__()
        self.alpha = nn.Parameter(1e-3 * torch.randn(cells, Supernet.N_EDGES, Supernet.N_OPS))

    def forward(self):
        return self.alpha * 1.0


# ---------------------------------------------------------------------------------------------
# inner losses (what the user writes in ``training_step``)
# ---------------------------------------------------------------------------------------------
def make_logreg_loss(upper):
    # examples/logistic_regression_hpo/logistic_regression_implicit.py:80-91
    def loss(self, batch):
        x, y = batch
        w = self.module.w
        lam = upper.fwd()
        return F.binary_cross_entropy_with_logits(x @ w, y) + 0.5 * (lam * w * w).sum()

    return loss


def make_reweight_loss(upper, ridge):
    # examples/learning_to_reweight/main.py:117-127 (+ ridge keeps H positive definite for CG)
    def loss(self, batch):
        x, y = batch
        logits = self.module(x)
        ce = F.cross_entropy(logits, y, reduction="none")
        weight = upper.fwd(ce.detach().reshape(-1, 1))
        out = torch.mean(weight.reshape(-1) * ce)
        if ridge:
            out = out + ridge * sum((p * p).sum() for p in self.module.parameters())
        return out

    return loss


def make_imaml_loss(upper, reg):
    # examples/implicit_maml/main.py:87-92,122-129: CE + reg * ||w - theta||^2
    def loss(self, batch):
        x, y = batch
        out = F.cross_entropy(self.module(x), y)
        prox = sum(((p - q) ** 2).sum() for p, q in zip(self.module.parameters(), upper.module.parameters()))
        return out + reg * prox

    return loss


def make_supernet_loss(upper, ridge):
    # examples/neural_architecture_search: the inner loss is the supernet's CE at the current architecture
    def loss(self, batch):
        x, y = batch
        out = F.cross_entropy(self.module(x, upper.fwd()), y)
        return out + ridge * sum((p * p).sum() for p in self.module.parameters())

    return loss


# ---------------------------------------------------------------------------------------------
# case registry
# ---------------------------------------------------------------------------------------------
@dataclasses.dataclass
class Case:
    name: str
    family: str  # logreg | reweight | imaml | deep
    algo: str  # cg | neumann | darts
    cfg: Dict  # Config kwargs
    # parity tolerance of the HIP path against the fp32 reference golden: north_star's 1e-4, or — for the finite-difference
    # algorithms (darts, sama), whose fp32 result is noise-limited — 5x the reference's OWN fp32-vs-fp64 spread on that case
    # when that is larger (tests/test_oracle.py::test_case_tolerances_follow_the_golden_spread ties the numbers to the goldens)
    rtol: float = 1e-4


CASES: List[Case] = [
    # cfg 1 of BASELINE.json (plumbing case) and its variants
    Case("logreg_cg5", "logreg", "cg", dict(type="cg", cg_iterations=5, cg_alpha=1.0)),
    Case("logreg_cg3_a01", "logreg", "cg", dict(type="cg", cg_iterations=3, cg_alpha=0.1)),
    Case("logreg_cg0", "logreg", "cg", dict(type="cg", cg_iterations=0, cg_alpha=1.0)),
    Case("logreg_neumann5", "logreg", "neumann", dict(type="neumann", neumann_iterations=5, neumann_alpha=0.5)),
    Case("logreg_neumann0", "logreg", "neumann", dict(type="neumann", neumann_iterations=0, neumann_alpha=0.5)),
    Case("logreg_darts", "logreg", "darts", dict(type="darts", darts_alpha=0.01), rtol=1e-4),
    # cfg 2 shape (reduced): ReLU-MLP inner with MWN-weighted CE
    Case("reweight_neumann10", "reweight", "neumann", dict(type="neumann", neumann_iterations=10, neumann_alpha=0.1)),
    Case("reweight_cg20", "reweight", "cg", dict(type="cg", cg_iterations=20, cg_alpha=1.0)),
    Case("reweight_darts", "reweight", "darts", dict(type="darts", darts_alpha=0.1), rtol=4e-4),
    # cfg 3 shape (reduced): conv net inner, prox-regularised to the upper copy (M = N)
    Case("imaml_cg10", "imaml", "cg", dict(type="cg", cg_iterations=10, cg_alpha=1.0)),
    Case("imaml_neumann6", "imaml", "neumann", dict(type="neumann", neumann_iterations=6, neumann_alpha=0.3)),
    Case("imaml_darts", "imaml", "darts", dict(type="darts", darts_alpha=0.05), rtol=1e-4),
    # SAMA: Adam-preconditioned finite difference; SGD = identity preconditioner
    Case("reweight_sama_adam", "reweight", "sama", dict(type="sama", sama_adam_alpha=1.0), rtol=1e-4),
    Case("logreg_sama_sgd", "logreg", "sama", dict(type="sama", sama_adam_alpha=0.01), rtol=1e-4),
    # *_multitask=True: the perturbed inner weights are NOT restored (darts.py:61-63, sama.py:51-53)
    Case("reweight_darts_multitask", "reweight", "darts", dict(type="darts", darts_alpha=0.1, darts_multitask=True), rtol=4e-4),
    Case("logreg_sama_multitask", "logreg", "sama", dict(type="sama", sama_adam_alpha=0.01, sama_multitask=True), rtol=1e-4),
    # many small tensors (T = 48 > 32): exercises the device pointer-table path (cfg 5 shape)
    Case("deep_neumann6", "deep", "neumann", dict(type="neumann", neumann_iterations=6, neumann_alpha=0.2)),
    Case("deep_cg6", "deep", "cg", dict(type="cg", cg_iterations=6, cg_alpha=1.0)),
    Case("deep_darts", "deep", "darts", dict(type="darts", darts_alpha=0.1), rtol=1.5e-3),
    Case("deep_sama_adam", "deep", "sama", dict(type="sama", sama_adam_alpha=1.0), rtol=2.6e-4),
]
CASE_BY_NAME = {c.name: c for c in CASES}


def _family_modules(family):
    """Fresh (randomly initialised) inner/upper modules of a family; weights are overwritten
    from the golden file afterwards."""
    if family == "logreg":
        return Vec(100, 0.0), Vec(100, 1.0)
    if family == "reweight":
        return MLP([48, 64, 32, 10]), MWN(16)
    if family == "imaml":
        return SmallConv(5), SmallConv(5)
    if family == "deep":
        return MLP([12] + [12] * 23 + [4]), MWN(8)  # 24 Linear layers = 48 tensors
    raise KeyError(family)


def seed_family_inputs(family, seed=0):
    """Create the seeded inputs of a family (run ONCE by make_golden.py; everyone else loads
    the stored arrays)."""
    g = torch.Generator().manual_seed(1000 + seed)
    torch.manual_seed(2000 + seed)
    inner, upper = _family_modules(family)
    arrays = {}
    if family == "logreg":
        # BASELINE.md cfg 1: w_gt~N(0,1)^100, X~N(0,1)^{1000x100}, y=(X w_gt + 0.1 eps > 0), 500 train
        w_gt = torch.randn(100, generator=g)
        X = torch.randn(1000, 100, generator=g)
        y = ((X @ w_gt + 0.1 * torch.randn(1000, generator=g)) > 0).float()
        xtr, ytr = X[:500].clone(), y[:500].clone()
        w = torch.zeros(100)
        for _ in range(100):  # 100 SGD steps, lr 0.1, lambda = 1
            z = xtr @ w
            gw = xtr.t() @ (torch.sigmoid(z) - ytr) / 500 + w
            w = w - 0.1 * gw
        inner.w.data.copy_(w)
        arrays["batch_x"], arrays["batch_y"] = xtr, ytr
    elif family in ("reweight", "deep"):
        d_in = 48 if family == "reweight" else 12
        n_cls = 10 if family == "reweight" else 4
        B = 40
        arrays["batch_x"] = torch.randn(B, d_in, generator=g)
        yy = torch.randint(0, n_cls, (B,), generator=g)
        flip = torch.rand(B, generator=g) < 0.4
        yy = torch.where(flip, torch.randint(0, n_cls, (B,), generator=g), yy)
        arrays["batch_y"] = yy
    elif family == "imaml":
        arrays["batch_x"] = torch.randn(25, 3, 16, 16, generator=g)
        arrays["batch_y"] = torch.arange(5).repeat_interleave(5)
        # outer copy = inner + small perturbation, like a few adaptation steps apart
        for p, q in zip(inner.parameters(), upper.parameters()):
            q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g))
    for i, p in enumerate(inner.parameters()):
        arrays[f"inner_{i}"] = p.data.clone()
        arrays[f"vec_{i}"] = 0.01 * torch.randn(p.shape, generator=g)
    for i, p in enumerate(upper.parameters()):
        arrays[f"upper_{i}"] = p.data.clone()
    # a plausible Adam state for the SAMA cases (drawn last so the older inputs keep their values)
    for i, p in enumerate(inner.parameters()):
        gl = 0.01 * torch.randn(p.shape, generator=g)
        arrays[f"opt_{i}_last_grad"] = gl
        arrays[f"opt_{i}_exp_avg"] = 0.1 * gl + 0.005 * torch.randn(p.shape, generator=g)
        arrays[f"opt_{i}_exp_avg_sq"] = 0.001 * gl * gl + 1e-5 * (1.0 + torch.rand(p.shape, generator=g))
    return {k: v.numpy() for k, v in arrays.items()}


FAMILY_LOSS: Dict[str, Callable] = {
    "logreg": lambda upper: make_logreg_loss(upper),
    "reweight": lambda upper: make_reweight_loss(upper, 0.5),  # keep in sync with RIDGE below
    "imaml": lambda upper: make_imaml_loss(upper, 0.5),
    "deep": lambda upper: make_reweight_loss(upper, 0.5),
}


def build_case(case: Case, inputs: Dict[str, np.ndarray], config_cls, device="cpu", dtype=torch.float32):
    """Instantiate (curr, prev, vector) of a case from stored inputs."""
    inner, upper = _family_modules(case.family)

    def T(a):
        t = torch.from_numpy(np.asarray(a))
        if t.is_floating_point():
            t = t.to(dtype)
        return t.to(device)

    inner, upper = inner.to(device=device, dtype=dtype), upper.to(device=device, dtype=dtype)
    for i, p in enumerate(inner.parameters()):
        p.data.copy_(T(inputs[f"inner_{i}"]))
    for i, p in enumerate(upper.parameters()):
        p.data.copy_(T(inputs[f"upper_{i}"]))
    vector = [T(inputs[f"vec_{i}"]) for i in range(len(list(inner.parameters())))]
    batch = (T(inputs["batch_x"]), T(inputs["batch_y"]))
    prev = StubProblem("upper", upper, config=config_cls())
    curr = StubProblem("inner", inner, config=config_cls(**case.cfg), loss_fn=FAMILY_LOSS[case.family](prev), batch=batch)
    if case.algo == "sama":
        if case.family == "logreg":
            curr.optimizer = torch.optim.SGD(inner.parameters(), lr=0.1)
        else:
            curr.optimizer = torch.optim.Adam(inner.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8)
            for i, p in enumerate(inner.parameters()):
                curr.optimizer.state[p] = {
                    "step": torch.tensor(3.0),
                    "exp_avg": T(inputs[f"opt_{i}_exp_avg"]),
                    "exp_avg_sq": T(inputs[f"opt_{i}_exp_avg_sq"]),
                    "last_grad": T(inputs[f"opt_{i}_last_grad"]),
                }
    return curr, prev, vector


RIDGE = {"reweight": 0.5, "deep": 0.5}


def attach_mlp_structure(curr, family, impl=None, fused=True, weight_net=False, average_over=None, overlap=False):
    """Opt the inner problem into the analytic HVP (betty_amd.hypergradient.structured).  weight_net=True also declares the upper
    module (zoo.MWN) as the closed-form meta-weight-net (SigmoidMLPWeightNet)."""
    from betty_amd.hypergradient.structured import SigmoidMLPWeightNet, WeightedCEMLP

    def structure(prev):
        return WeightedCEMLP(
            curr, prev, layers=list(curr.module.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)),
            ridge=RIDGE[family], impl=impl, fused=fused,
            weight_net=SigmoidMLPWeightNet(prev.module.l1, prev.module.l2, average_over=average_over, overlap=overlap) if weight_net else None,
        )

    curr.hypergradient_structure = structure
    return curr


def attach_logreg_structure(curr):
    from betty_amd.hypergradient.structured import LogisticRegressionL2

    curr.hypergradient_structure = lambda prev: LogisticRegressionL2(curr, prev, curr.module.w, lam_fn=lambda: prev.fwd())
    return curr


def attach_prox_structure(curr):
    """iMAML structure: data loss = CE only, prox coefficient of make_imaml_loss (0.5)."""
    from betty_amd.hypergradient.structured import ProximalRegularized

    def structure(prev):
        return ProximalRegularized(curr, prev, data_loss=lambda batch: F.cross_entropy(curr.module(batch[0]), batch[1]), reg=0.5)

    curr.hypergradient_structure = structure
    return curr



# ------------------------------------------------------------------------------------------------------------------------------------
# BASELINE.json cfg 5 AS NAMED: the reference's own DARTS supernet `Network(16, 10, 8)` (1,930,618 parameters in 1,399 tensors) and
# `Architecture(4)` (2 x 14 x 8 = 224) — examples/neural_architecture_search/model_search.py:129-234,302-317 — on the example's batch
# 64 x 3 x 32 x 32 (train_search.py:24), Neumann K = 20.  The model files are the reference's own: read from the checkout in the build
# container (tests/golden/make_cfg5_golden.py) or from the staged, git-ignored, test-only copy oracle/_ref/examples_nas on the GPU box.
# ------------------------------------------------------------------------------------------------------------------------------------
NAS_DIRS = (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "examples_nas"),
            "/root/reference/examples/neural_architecture_search")
CFG5_SEED, CFG5_K, CFG5_ALPHA, CFG5_BATCH = 5, 20, 0.01, 64


def nas_dir():
    for d in NAS_DIRS:
        if os.path.isfile(os.path.join(d, "model_search.py")):
            return d
    return None


def reference_nas_modules():
    import importlib
    import sys
    import types

    if "utils" not in sys.modules or not hasattr(sys.modules["utils"], "accuracy"):
        # model_search.py does `from utils import accuracy`; the example's utils.py needs torchvision (absent here) for its data
        # pipeline only — a stand-in module carries the one function the model file names (it is never called by `loss`)
        stub = types.ModuleType("utils")
        stub.accuracy = lambda output, target, topk=(1,): [torch.zeros(()) for _ in topk]
        sys.modules["utils"] = stub
    d = nas_dir()
    sys.path.insert(0, d)
    try:
        return importlib.import_module("model_search")
    finally:
        sys.path.remove(d)


def cfg5_as_named_case(Config, device, dtype=torch.float32, K=CFG5_K):
    """(curr, prev, vector): every tensor is drawn by the CPU generator in fp32 and then moved / cast, so the build container (CPU,
    fp32 and fp64) and the GPU box see the same numbers."""
    ms = reference_nas_modules()
    g = torch.Generator().manual_seed(CFG5_SEED)
    torch.manual_seed(CFG5_SEED)
    inner = ms.Network(16, 10, 8, torch.nn.CrossEntropyLoss()).to(device=device, dtype=dtype)
    upper = ms.Architecture(4).to(device=device, dtype=dtype)
    x = torch.randn(CFG5_BATCH, 3, 32, 32, generator=g).to(device=device, dtype=dtype)
    y = torch.randint(0, 10, (CFG5_BATCH,), generator=g).to(device)
    vector = [(1e-2 * torch.randn(p.shape, generator=g)).to(device=device, dtype=dtype) for p in inner.parameters()]
    prev = StubProblem("arch", upper, config=Config())

    def loss_fn(self, batch):   # train_search.py:124-129 (Classifier.training_step)
        xb, tb = batch
        return self.module.loss(xb, prev.module(), tb)

    curr = StubProblem("classifier", inner, config=Config(type="neumann", neumann_iterations=K, neumann_alpha=CFG5_ALPHA),
                       loss_fn=loss_fn, batch=(x, y))
    return curr, prev, vector


def cfg5_checksums(curr, prev, vector):
    """fp64 sums (and absolute sums) of every input tensor: the GPU box proves it rebuilt the very problem the golden was made from."""
    import numpy as np

    x, y = curr.cur_batch
    ts = list(curr.parameters()) + list(prev.parameters()) + [x, y.double()] + list(vector)
    return np.array([t.detach().double().sum().item() for t in ts] + [t.detach().double().abs().sum().item() for t in ts])


# ---- round 6: shapes OUTSIDE the benchmark's family (widths that are not multiples of 32, heads wider than 32 classes) ---------------
# (dims, batch): the reweighting problem of examples/learning_to_reweight/main.py:117-127 on a ReLU-MLP of these widths; goldens from the
# reference's own cg / neumann on the CPU in tests/golden/shapes.npz (tests/golden/make_shapes_golden.py)
SHAPE_CASES = {
    "mnist_784_512_256_128_10": ([784, 512, 256, 128, 10], 100),
    "ragged_784_500_250_100_10": ([784, 500, 250, 100, 10], 128),
    "head100_256_384_128_100": ([256, 384, 128, 100], 100),
    "head100_512_256_256_64_100": ([512, 256, 256, 64, 100], 100),
    "head1000_64_96_1000": ([64, 96, 1000], 130),
}
SHAPE_RIDGE, SHAPE_K = 0.3, 10


def shape_case(name, Config, device, algo, seed=0, dtype=torch.float32):
    """(curr, prev, vector): every tensor drawn by the CPU generator in fp32 and then moved / cast (the build container and the GPU box
    see the same numbers)."""
    dims, B = SHAPE_CASES[name]
    g = torch.Generator().manual_seed(9000 + seed)
    inner, upper = MLP(dims), MWN(16)
    with torch.no_grad():
        for p in list(inner.parameters()) + list(upper.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(p.shape[-1], 4) ** 0.5))
    x = torch.randn(B, dims[0], generator=g)
    y = torch.randint(0, dims[-1], (B,), generator=g)
    vector = [0.1 * torch.randn(p.shape, generator=g) for p in inner.parameters()]
    inner, upper = inner.to(device=device, dtype=dtype), upper.to(device=device, dtype=dtype)
    x, y = x.to(device=device, dtype=dtype), y.to(device)
    vector = [v.to(device=device, dtype=dtype) for v in vector]
    prev = StubProblem("upper", upper, config=Config())
    cfg = Config(type="cg", cg_iterations=SHAPE_K, cg_alpha=1.0) if algo == "cg" else Config(type="neumann", neumann_iterations=SHAPE_K, neumann_alpha=0.1)
    curr = StubProblem("inner", inner, config=cfg, loss_fn=make_reweight_loss(prev, SHAPE_RIDGE), batch=(x, y))
    return curr, prev, vector


def shape_checksums(name, Config, seed=0):
    import numpy as np

    curr, prev, vector = shape_case(name, Config, "cpu", "cg", seed=seed)
    x, y = curr.cur_batch
    ts = list(curr.module.parameters()) + list(prev.module.parameters()) + [x, y.double()] + list(vector)
    return np.array([t.detach().double().sum().item() for t in ts] + [t.detach().double().abs().sum().item() for t in ts])
