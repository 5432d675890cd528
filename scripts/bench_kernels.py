#!/usr/bin/env python
"""Micro-benchmark of the recurrence kernels in isolation at the BASELINE size (N = 10,034,826 in
the 8 tensors of the cfg-2 MLP).  The HVP producer is emulated by a diagonal multiply that rewrites
the 8 HVP tensors before every launch (so Hp is as cache-cold/warm as behind a real producer).
Prints one line per kernel: median / min launch time from HIP events and achieved algorithmic GB/s.

--scrub-mb M (cache-defeated mode): before every timed launch a device copy of 2 x M MiB (default off; use >= 384) runs
through the memory system, so the 256 MiB Infinity Cache no longer holds the state vectors x, r, p / v, p when the kernel
starts (the HVP tensors are re-written AFTER the scrub: they are as warm as behind a real producer).  Without it the
N = 10 M working set (160 MB) stays in the cache from launch to launch and the "GB/s" are fabric-side, not HBM, numbers.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betty_amd import _native  # noqa: E402
from betty_amd.backend import get_backend  # noqa: E402

SIZES = [3072 * 2048, 2048, 2048 * 1536, 1536, 1536 * 384, 384, 384 * 10, 10]


def timeit(fn, pre, n):
    ts = []
    for _ in range(n):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ts)
    return 1e3 * ms[len(ms) // 2], 1e3 * ms[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--scale", type=int, default=1, help="replicate the tensor list (bigger N)")
    ap.add_argument("--extra", type=int, default=0, help="append one tensor of this many elements (e.g. 5000000 -> N = 15 M: "
                    "beyond the register-only resident capacity, inside the LDS-assisted one)")
    ap.add_argument("--scrub-mb", type=int, default=0, help="cache-defeated mode: copy 2 x this many MiB before every launch")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    be = get_backend()
    sizes = SIZES * args.scale + ([args.extra] if args.extra > 0 else [])
    N = sum(sizes)
    gen = torch.Generator().manual_seed(0)
    vec = [torch.randn(n, generator=gen).to(dev) for n in sizes]
    diag = [1.0 + 0.5 * torch.rand(n, generator=gen).to(dev) for n in sizes]
    hv = [torch.empty_like(v) for v in vec]
    lay = be.layout(vec)
    x, r, p = lay.state(3)
    out = {}
    scrub = None
    if args.scrub_mb > 0:
        sa = torch.zeros(args.scrub_mb * (1 << 20) // 4, device=dev)
        sb = torch.empty_like(sa)
        scrub = lambda: sb.copy_(sa)

    def with_scrub(pre):
        if scrub is None:
            return pre

        def f():
            scrub()
            pre()
        return f

    # calibration: plain device copy of 3 x 40 MB in, 3 x 40 MB out
    src = torch.randn(3 * N, device=dev)
    dst = torch.empty_like(src)
    med, mn = timeit(lambda: dst.copy_(src), with_scrub(lambda: None), args.iters)
    out["torch_copy_24N"] = dict(us=med, min_us=mn, GBps=24.0 * N / med / 1e3)

    def refresh(views):
        def f():
            torch._foreach_mul_(hv, 0.0)
            torch._foreach_addcmul_(hv, diag, views)
        return f

    for name, variant in (("cg_stream", _native.BHG_CG_STREAM), ("cg_resident", _native.BHG_CG_RESIDENT)):
        if variant == _native.BHG_CG_RESIDENT and lay.n_chunks > be.lib.bhg_cg_resident_capacity_chunks():
            continue
        be.cg_init(lay, vec, x, r, p)
        pv = lay.views(p, vec)
        k = [0]

        def step():
            be.cg_step(lay, hv, x, r, p, 1.0, k[0], 0.0, variant=variant)
            k[0] += 1
            if k[0] % 6 == 0:  # restart before the recurrence under/overflows
                be.cg_init(lay, vec, x, r, p)
                k[0] = 0

        med, mn = timeit(step, with_scrub(refresh(pv)), args.iters)
        out[name] = dict(us=med, min_us=mn, GBps=28.0 * N / med / 1e3, frac_of_8TBps=28.0 * N / med / 1e3 / 8000)
        assert not be.cg_barrier_timed_out(lay), "grid barrier timed out"

    v, pp = lay.state(2)
    be.neumann_init(lay, vec, v, pp)
    vv = lay.views(v, vec)
    med, mn = timeit(lambda: be.neumann_step(lay, hv, v, pp, 0.01, 0.0), with_scrub(refresh(vv)), args.iters)
    out["neumann_step"] = dict(us=med, min_us=mn, GBps=20.0 * N / med / 1e3, frac_of_8TBps=20.0 * N / med / 1e3 / 8000)

    w = [t.clone() for t in vec]
    coef = torch.tensor([1e-3], device=dev)
    med, mn = timeit(lambda: be.axpy_multi(lay, w, vec, coef[0], 1.0), with_scrub(lambda: None), args.iters)
    out["axpy_multi"] = dict(us=med, min_us=mn, GBps=12.0 * N / med / 1e3)
    med, mn = timeit(lambda: be.darts_eps(lay, vec, 0.01), with_scrub(lambda: None), args.iters)
    out["darts_eps(sqnorm)"] = dict(us=med, min_us=mn, GBps=4.0 * N / med / 1e3)
    med, mn = timeit(lambda: be.cg_init(lay, vec, x, r, p), with_scrub(lambda: None), args.iters)
    out["cg_init"] = dict(us=med, min_us=mn, GBps=16.0 * N / med / 1e3)
    print(json.dumps({"N": N, "T": len(sizes), "scrub_mb_before_each_launch": args.scrub_mb, "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
