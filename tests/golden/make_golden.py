#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (leopard-ai/betty v0.2.1 at
/root/reference) on the seeded problems of tests/zoo.py.

Runs only in the build container (the reference does not exist on the GPU box).  The outputs are
numbers, not code: per family one ``<family>.npz`` holding the inputs (weights, batch, direction
vector) and, per case, the reference's results
    out/<case>/fp32/<i>       sync=False, fp32            (what the HIP path must match, rtol per case)
    out/<case>/fp64/<i>       sync=False, everything in fp64  (conditioning check)
    out/<case>/sync32/<i>     sync=True, fp32: prev.grad after the call (the call returns None)
    out/<case>/w32/<i>        darts only: inner weights after the call (perturb/restore drift)

Usage:  PYTHONPATH=/root/reference python tests/golden/make_golden.py [--with-cfg2]
        --with-cfg2 also regenerates cfg2_full.npz (the metric workload at FULL size, N = 10,034,826, cg K = 20 and
        neumann K = 10 on five seeds of two conditioning variants; ~10 min) through make_cfg2_golden.py.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))  # tests/
sys.path.insert(0, "/root/reference")

import zoo  # noqa: E402

import betty.hypergradient  # noqa: E402,F401
from betty.configs import Config as RefConfig  # noqa: E402

REF = {
    "cg": sys.modules["betty.hypergradient.cg"].cg,
    "neumann": sys.modules["betty.hypergradient.neumann"].neumann,
    "darts": sys.modules["betty.hypergradient.darts"].darts,
    "sama": sys.modules["betty.hypergradient.sama"].sama,
}


def run(case, inputs, dtype, sync):
    curr, prev, vector = zoo.build_case(case, inputs, RefConfig, device="cpu", dtype=dtype)
    out = REF[case.algo](vector, curr, prev, sync)
    if sync:
        assert out is None
        res = [p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p) for p in prev.trainable_parameters()]
    else:
        res = [o.detach().clone() for o in out]
    weights = [p.data.clone() for p in curr.trainable_parameters()]
    return res, weights


def main():
    torch.set_num_threads(1)  # fixed reduction order for torch.dot / mm
    families = sorted({c.family for c in zoo.CASES})
    for fam in families:
        inputs = zoo.seed_family_inputs(fam)
        blob = {f"in/{k}": v for k, v in inputs.items()}
        for case in [c for c in zoo.CASES if c.family == fam]:
            r32, w32 = run(case, inputs, torch.float32, False)
            r64, _ = run(case, inputs, torch.float64, False)
            s32, _ = run(case, inputs, torch.float32, True)
            for i, t in enumerate(r32):
                blob[f"out/{case.name}/fp32/{i}"] = t.numpy()
            for i, t in enumerate(r64):
                blob[f"out/{case.name}/fp64/{i}"] = t.numpy()
            for i, t in enumerate(s32):
                blob[f"out/{case.name}/sync32/{i}"] = t.numpy()
            if case.algo in ("darts", "sama"):
                for i, t in enumerate(w32):
                    blob[f"out/{case.name}/w32/{i}"] = t.numpy()
            a = torch.cat([t.reshape(-1).double() for t in r32])
            b = torch.cat([t.reshape(-1) for t in r64])
            rel = ((a - b).norm() / b.norm()).item()
            worst = ((a - b).abs().max() / b.abs().max()).item()
            print(f"{case.name:22s} |out|={b.norm().item():.4e}  fp32-vs-fp64 rel={rel:.2e} max/max={worst:.2e}")
        np.savez_compressed(os.path.join(HERE, f"{fam}.npz"), **blob)
        print(f"wrote {fam}.npz ({os.path.getsize(os.path.join(HERE, fam + '.npz')) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
    if "--with-cfg2" in sys.argv:
        import make_cfg2_golden

        make_cfg2_golden.main()
