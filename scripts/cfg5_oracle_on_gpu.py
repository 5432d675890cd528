#!/usr/bin/env python
"""cfg 5 as named on the MI355X, K = 20: the product path (k_neumann_step on 1,399 tensors + forward-over-reverse HVP passes) against
the oracle's restatement of neumann.py on the SAME device tensors — i.e. the reference's algorithm with autograd's double backward on
the same GPU (~6 minutes: its 20 products are 18 s each) — and both against the reference's own CPU run (tests/golden/cfg5_as_named.npz).
Evidence run, not a test: the default GPU suite checks the product against the CPU golden."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import hypergrad_oracle as horc  # noqa: E402
import zoo  # noqa: E402
from betty_amd import Config  # noqa: E402
from betty_amd import hypergradient as hg  # noqa: E402

dev = torch.device("cuda:0")
gold = np.load(os.path.join(ROOT, "tests", "golden", "cfg5_as_named.npz"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else zoo.CFG5_K   # 3: the short-horizon golden of round 6
if len(sys.argv) > 2 and sys.argv[2] == "deterministic":
    torch.backends.cudnn.deterministic = True


def flat(ts):
    return np.concatenate([t.detach().double().cpu().numpy().ravel() for t in ts])


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


curr, prev, vector = zoo.cfg5_as_named_case(Config, dev, K=K)
curr.hypergradient_hvp = "forward_over_reverse"
torch.cuda.synchronize()
t0 = time.perf_counter()
got = flat(hg.jvp_fn_mapping["neumann"](vector, curr, prev, False))
torch.cuda.synchronize()
t_got = time.perf_counter() - t0
print(f"product (forward-over-reverse HVP): {t_got:.1f} s/step; vs reference-CPU fp32 {rel(got, gold[f'neumann{K}/fp32']):.2e}, "
      f"vs reference fp64 {rel(got, gold[f'neumann{K}/fp64']):.2e} (reference's own fp32-vs-fp64 {float(gold[f'neumann{K}/ref_spread']):.2e})", flush=True)
curr2, prev2, vector2 = zoo.cfg5_as_named_case(Config, dev, K=K)
torch.cuda.synchronize()
t0 = time.perf_counter()
want = flat(horc.neumann(vector2, curr2, prev2, False))
torch.cuda.synchronize()
t_ref = time.perf_counter() - t0
print(f"reference algorithm on this GPU (oracle restatement, double backward): {t_ref:.1f} s/step; vs reference-CPU fp32 "
      f"{rel(want, gold[f'neumann{K}/fp32']):.2e}, vs reference fp64 {rel(want, gold[f'neumann{K}/fp64']):.2e}", flush=True)
print(f"product vs reference algorithm on the same GPU: rel {rel(got, want):.2e}, max/max {float(np.abs(got - want).max() / np.abs(want).max()):.2e}; "
      f"speed-up x{t_ref / t_got:.1f}", flush=True)
