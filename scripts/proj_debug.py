#!/usr/bin/env python
"""Debug: CG scalars {rr_old, den, alpha, rr_new, beta, pp} after K iterations, projection level 1 (N-sized r / p) vs level 2
(fully projected), K = 1..6, small MLP and the cfg-2 shapes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
from betty_amd.backend import get_backend
be = get_backend()
for dims, B, ridge in (([256, 384, 128, 10], 100, 0.05), ([512, 256, 256, 64, 10], 128, 0.05)):
    for K in (1, 2, 3, 4, 6):
        rows = {}
        for name, env, keep in (("level1", "9", False), ("level2", "1", False), ("level2-again", "1", False)):
            os.environ["BHG_MLP_PROJ"] = env
            out, _ = T._run_solver("cg", dims, B, ridge, K, 7, True, keep=keep)
            curr, prev, direction, provider = T._mlp_problem(dims, B, ridge=ridge, seed=7)
            lay = be.layout(direction)
            rows[name] = (be.cg_scalars(lay).cpu().numpy()[:6], np.concatenate([o.ravel() for o in out]))
        a, b, c = rows["level1"], rows["level2"], rows["level2-again"]
        rel = np.linalg.norm(a[1] - b[1]) / np.linalg.norm(a[1])
        print(f"{dims} K={K}: out rel {rel:.2e}; reproducible {np.array_equal(b[1], c[1])}")
        print("   level1", " ".join(f"{v:.8e}" for v in a[0]))
        print("   level2", " ".join(f"{v:.8e}" for v in b[0]))
        print("   again ", " ".join(f"{v:.8e}" for v in c[0]))
