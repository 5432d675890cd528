#!/usr/bin/env python
"""Generate tests/golden/cfg2_full.npz: outputs of the REAL reference (leopard-ai/betty at /root/reference) on the
metric workload at FULL size — ``bench.build(seed)``: MLP 3072-2048-1536-384-10 (N = 10,034,826), MWN 1-100-1, batch 100
— for ``cg`` K = 20 (betty/hypergradient/cg.py:8-70) and ``neumann`` K = 10, alpha = 0.1 (neumann.py:8-66), in fp32
(what the HIP path must match) and in fp64 (the truth, and the reference's own rounding spread).

Two variants of the same shapes:
  metric  ridge = bench.RIDGE = 1e-2, seeds 0-4 — the configuration the metric is quoted on.  Twenty un-preconditioned CG
          iterations on this (indefinite-plus-small-ridge) Hessian are not a contraction in fp32: the reference's own fp32
          answer sits 1.7e-3 ... 8.2e-2 from its fp64 answer on one thread (5e-5 ... 5e-3 on eight: the GEMM summation
          order alone moves it; printed below, stored as ``ref_spread``; log: profiles/r03_cfg2_reference_cpu_goldens.log),
          so rtol 1e-4 against it is not decidable there — the GPU test holds the product to the reference's own spread.
  well    ridge = RIDGE_WELL = 0.3 (2*ridge = 0.6 dominates the negative curvature of the CE Hessian; the Krylov
          iterations still matter: K = 1 and K = 20 differ by 13-80 %).  Seeds are the first five of range(16) whose
          fp64 pre-activations keep a relative distance >= KINK_MARGIN from the ReLU kink (inner net AND meta-weight-net):
          a pre-activation within fp32 summation noise of zero flips its mask between two correct fp32 implementations
          (seed 4: margin 2.4e-7, the reference's fp32 and fp64 answers differ by 5e-3 at EVERY K for that reason alone)
          — a property of the instance, not of an implementation.  On the selected seeds the reference's fp32-vs-fp64
          spread is <= 1e-5 (asserted here), so rtol 1e-4 has resolving power and every kernel arm is held to it.

The inputs are not stored (120 MB per seed): ``bench.build`` regenerates them from the seed with the CPU generator; the
file carries fp64 checksums of every input tensor so the GPU test can prove it rebuilt the very same problem.

Usage:  python tests/golden/make_cfg2_golden.py        (build container only: needs /root/reference; ~10 min, 1 thread)
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

RIDGE_WELL = 0.3
KINK_MARGIN = 1.5e-6
ALGOS = {"cg20": ("cg", 20), "neumann10": ("neumann", 10)}


def reference_fn(algo):
    """The reference's own function (imported lazily: tests import this module for `checksums` on boxes without it)."""
    sys.path.insert(0, "/root/reference")
    import betty.hypergradient  # noqa: F401

    return sys.modules[f"betty.hypergradient.{algo}"].__dict__[algo]


def run(seed, ridge, dtype, algo, K):
    curr, prev, vector = bench.build(torch.device("cpu"), seed, dtype=dtype, K=K, algo=algo, ridge=ridge)
    out = reference_fn(algo)(vector, curr, prev, False)
    return torch.cat([o.detach().reshape(-1) for o in out])


def checksums(seed):
    """fp64 sums of every input tensor of bench.build(seed) (weights, MWN, batch, labels, direction)."""
    curr, prev, vector = bench.build(torch.device("cpu"), seed)
    x, y = curr.cur_batch
    ts = list(curr.parameters()) + list(prev.parameters()) + [x, y.double()] + list(vector)
    return np.array([t.detach().double().sum().item() for t in ts] + [t.detach().double().abs().sum().item() for t in ts])


def kink_margin(seed):
    """min |pre-activation| / mean |pre-activation| over every ReLU of the inner net and of the MWN, in fp64."""
    curr, prev, _ = bench.build(torch.device("cpu"), seed, dtype=torch.float64)
    x, y = curr.cur_batch
    h, m = x, float("inf")
    for lin in curr.module.layers[:-1]:
        a = lin(h)
        m = min(m, (a.abs().min() / a.abs().mean()).item())
        h = F.relu(a)
    ce = F.cross_entropy(curr.module.layers[-1](h), y, reduction="none")
    a = prev.module.l1(ce.reshape(-1, 1))
    return min(m, (a.abs().min() / a.abs().mean()).item())


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def main():
    torch.set_num_threads(1)  # fixed reduction order of the CPU GEMMs / dots
    blob = {}
    margins = {s: kink_margin(s) for s in range(16)}
    well_seeds = [s for s in range(16) if margins[s] >= KINK_MARGIN][:5]
    assert len(well_seeds) == 5
    print("kink margins:", {s: f"{m:.1e}" for s, m in margins.items()})
    print("well-conditioned variant uses seeds", well_seeds)
    variants = {"metric": (bench.RIDGE, [0, 1, 2, 3, 4]), "well": (RIDGE_WELL, well_seeds)}
    for vname, (ridge, seeds) in variants.items():
        blob[f"{vname}/ridge"] = np.array(ridge)
        blob[f"{vname}/seeds"] = np.array(seeds)
        for seed in seeds:
            blob[f"{vname}/{seed}/checksum"] = checksums(seed)
            blob[f"{vname}/{seed}/kink_margin"] = np.array(margins[seed])
            for aname, (algo, K) in ALGOS.items():
                t0 = time.time()
                r32 = run(seed, ridge, torch.float32, algo, K)
                r64 = run(seed, ridge, torch.float64, algo, K)
                spread = rel(r32, r64)
                blob[f"{vname}/{seed}/{aname}/fp32"] = r32.numpy()
                blob[f"{vname}/{seed}/{aname}/fp64"] = r64.numpy()
                blob[f"{vname}/{seed}/{aname}/ref_spread"] = np.array(spread)
                print(f"{vname:6s} ridge={ridge:g} seed={seed} {aname:9s} |out|={r64.norm().item():.4e} "
                      f"reference fp32-vs-fp64 = {spread:.2e}   ({time.time() - t0:.0f}s)", flush=True)
                if vname == "well":
                    assert spread <= 1e-5, "the well-conditioned variant must keep the reference's own spread <= 1e-5"
    np.savez_compressed(os.path.join(HERE, "cfg2_full.npz"), **blob)
    print(f"wrote cfg2_full.npz ({os.path.getsize(os.path.join(HERE, 'cfg2_full.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
