#!/usr/bin/env python
"""Per-rank COMPUTE time of one factor-exchange CG iteration at world sizes 1, 2, 4, 8 — emulated on ONE GPU (gpurun boxes have one):
`world` copies of the cfg-2 inner network with their own batches, states and buffers live in this process, every all-gather is a set of
device copies between their buffers (NOT timed), and HIP events on the launch stream bracket rank 0's two phases of every iteration
(CHAIN: step + R-chain; GRAM: pack + rectangular Gram products + G(raw) + inner products).  What it shows: how the per-rank work grows
with the world size (Gram blocks are [Bp x world * Bp]) — the compute side of DESIGN 4b's table.  It is NOT a scaling measurement: no
link carried a byte."""
import argparse
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import zoo  # noqa: E402
from betty_amd import Config, _native  # noqa: E402
from betty_amd.backend import get_backend  # noqa: E402
from betty_amd.flat import FlatLayout  # noqa: E402
from betty_amd.hypergradient.structured import WeightedCEMLP  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="3072,2048,1536,384,10")
    ap.add_argument("--batch", type=int, default=100)
    ap.add_argument("--K", type=int, default=20)
    ap.add_argument("--solves", type=int, default=6)
    ap.add_argument("--worlds", default="1,2,4,8")
    args = ap.parse_args()
    dims = [int(v) for v in args.dims.split(",")]
    B, K, ridge, alpha = args.batch, args.K, 0.3, 1.0
    be = get_backend()
    out = {"dims": dims, "batch_per_rank": B, "K": K, "note": __doc__.split("\n\n")[0][:0] or "emulated ranks on one GPU; gathers are untimed device copies", "worlds": {}}
    for G in [int(v) for v in args.worlds.split(",")]:
        g = torch.Generator().manual_seed(7)
        inner, upper = zoo.MLP(dims), zoo.MWN(16)
        inner, upper = inner.to(DEV), upper.to(DEV)
        prev = zoo.StubProblem("upper", upper, config=Config())
        x = torch.randn(G * B, dims[0], generator=g).to(DEV)
        y = torch.randint(0, dims[-1], (G * B,), generator=g).to(DEV)
        inners = [inner] + [copy.deepcopy(inner) for _ in range(G - 1)]
        vec = [0.01 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]
        provs, lays, rhss, bufs = [], [], [], []
        for r in range(G):
            c = zoo.StubProblem("inner", inners[r], config=Config(type="cg", cg_iterations=K, cg_alpha=alpha),
                                loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x[r * B:(r + 1) * B], y[r * B:(r + 1) * B]))
            prov = WeightedCEMLP(c, prev, layers=list(inners[r].layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)), ridge=ridge,
                                 impl="hip", fused=True, verify=False)
            prov.pad_widths = False
            prov.prepare()
            lay = FlatLayout([t.numel() for t in vec], DEV)
            assert prov.fused_cg_fx_ready(lay, K, G)
            v = lay.new_flat()
            be.flatten(lay, vec, v, 1.0)
            provs.append(prov); lays.append(lay); rhss.append(lay.views(v, vec)); bufs.append(prov._state.fx_buffers(G))

        def gather(name):
            for r in range(G):
                for j in range(G):
                    if j != r:
                        bufs[j][name][r].copy_(bufs[r][name][r])

        ev = lambda: torch.cuda.Event(enable_timing=True)
        chain_us, gram_us = [], []
        for solve in range(args.solves):
            for r in range(G):
                provs[r].cg_fx_phase(rhss[r], 0, K, _native.BHG_CG_FX_BEGIN, G, r, alpha)
            gather("const")
            marks = []
            for k in range(K):
                e0, e1, e2, e3 = ev(), ev(), ev(), ev()
                e0.record()
                provs[0].cg_fx_phase(rhss[0], k, K, _native.BHG_CG_FX_CHAIN, G, 0, alpha)
                e1.record()
                for r in range(1, G):
                    provs[r].cg_fx_phase(rhss[r], k, K, _native.BHG_CG_FX_CHAIN, G, r, alpha)
                gather("slab")
                e2.record()
                provs[0].cg_fx_phase(rhss[0], k, K, _native.BHG_CG_FX_GRAM, G, 0, alpha)
                e3.record()
                for r in range(1, G):
                    provs[r].cg_fx_phase(rhss[r], k, K, _native.BHG_CG_FX_GRAM, G, r, alpha)
                gather("scal")
                marks.append((e0, e1, e2, e3))
            for r in range(G):
                provs[r].cg_fx_phase(rhss[r], K - 1, K, _native.BHG_CG_FX_END, G, r, alpha)
            torch.cuda.synchronize()
            if solve >= 2:   # (two warm-up solves)
                for k, (e0, e1, e2, e3) in enumerate(marks):
                    if k >= 1:   # iteration 0 carries the once-per-solve work
                        chain_us.append(e0.elapsed_time(e1) * 1e3)
                        gram_us.append(e2.elapsed_time(e3) * 1e3)
        rzx_ok = all(bool(torch.isfinite(p._state.buf.fws.view(torch.uint8).float()).all()) for p in provs[:1])
        med = lambda v: sorted(v)[len(v) // 2]
        b0 = bufs[0]
        out["worlds"][G] = {"chain_phase_us": round(med(chain_us), 2), "gram_phase_us": round(med(gram_us), 2),
                            "iteration_compute_us": round(med(chain_us) + med(gram_us), 2),
                            "gathered_per_iteration_bytes_per_rank": int(b0["slab"].shape[1] * 4 + b0["scal"].shape[1] * 8),
                            "received_per_iteration_bytes": int((G - 1) * (b0["slab"].shape[1] * 4 + b0["scal"].shape[1] * 8)),
                            "once_per_solve_bytes_per_rank": int(b0["const"].shape[1] * 4),
                            "workspace_mb": round(b0["xws"].numel() / 2 ** 20, 1), "finite": rzx_ok}
        print(json.dumps({G: out["worlds"][G]}), flush=True)
        del provs, bufs, inners
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
