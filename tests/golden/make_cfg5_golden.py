#!/usr/bin/env python
"""Generate tests/golden/cfg5_as_named.npz: outputs of the REAL reference (leopard-ai/betty at /root/reference) on BASELINE cfg 5
as named — its own DARTS supernet ``Network(16, 10, 8)`` (examples/neural_architecture_search/model_search.py:129-234; 1,930,618
parameters in 1,399 tensors) and ``Architecture(4)`` (model_search.py:302-317; 2 x 14 x 8 = 224), batch 64 x 3 x 32 x 32
(train_search.py:24), ``neumann`` K = 20, alpha = 0.01 (betty/hypergradient/neumann.py:8-66) — on the CPU, in fp32 (what the HIP
path is compared with) and in fp64 (the truth; |fp32 - fp64| is the reference's OWN rounding spread on this instance, stored as
``ref_spread`` and used by the GPU test as its noise floor).

The inputs are not stored: ``tests/zoo.cfg5_as_named_case`` regenerates them from the seed with the CPU generator; the file carries
fp64 checksums of every input tensor so the GPU test proves it rebuilt the very same problem.

Usage:  python tests/golden/make_cfg5_golden.py        (build container only: needs /root/reference; 8 threads)
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import zoo  # noqa: E402


def reference():
    sys.path.insert(0, "/root/reference")
    import betty.hypergradient  # noqa: F401
    from betty.configs import Config

    return sys.modules["betty.hypergradient.neumann"].neumann, Config


def run(dtype, K):
    fn, Config = reference()
    curr, prev, vector = zoo.cfg5_as_named_case(Config, torch.device("cpu"), dtype=dtype, K=K)
    out = fn(vector, curr, prev, False)
    return torch.cat([o.detach().reshape(-1) for o in out]), zoo.cfg5_checksums(curr, prev, vector) if dtype == torch.float32 else None


def main():
    assert zoo.nas_dir() is not None
    blob = {}
    for K in (zoo.CFG5_K,):
        t0 = time.time()
        r32, cs = run(torch.float32, K)
        t1 = time.time()
        r64, _ = run(torch.float64, K)
        spread = ((r32.double() - r64).norm() / r64.norm()).item()
        blob[f"neumann{K}/fp32"], blob[f"neumann{K}/fp64"] = r32.numpy(), r64.numpy()
        blob[f"neumann{K}/ref_spread"] = np.array(spread)
        blob["checksum"] = cs
        print(f"cfg 5 as named, neumann K={K} alpha={zoo.CFG5_ALPHA}: |out| = {r64.norm().item():.6e}, {r32.numel()} floats; reference fp32-vs-fp64 "
              f"= {spread:.2e}   (fp32 {t1 - t0:.0f} s, fp64 {time.time() - t1:.0f} s, {torch.get_num_threads()} threads)", flush=True)
    np.savez_compressed(os.path.join(HERE, "cfg5_as_named.npz"), **blob)
    print(f"wrote cfg5_as_named.npz ({os.path.getsize(os.path.join(HERE, 'cfg5_as_named.npz')) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main()
