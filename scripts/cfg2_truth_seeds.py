"""The metric workload (cfg 2, CG K = 20) on several seeded problems: distance to the fp64 truth of the product's default
path and of the reference's algorithm in fp32 on the same device (oracle restatement on ATen).  Shows what north_star's
rtol 1e-4 means at K = 20 on this problem: both sit in the same fp32 noise band."""
import os, sys
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "oracle"))
import bench
import hypergrad_oracle as horc
from betty_amd import hypergradient as hg


def flat(ts):
    return np.concatenate([t.detach().double().cpu().numpy().ravel() for t in ts])


dev = torch.device("cuda:0")
K, algo = 20, "cg"
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    c64, p64, v64 = bench.build(dev, seed=seed, dtype=torch.float64, K=K, algo=algo)
    truth = flat(getattr(horc, algo)(v64, c64, p64, False))
    del c64, p64, v64
    curr, prev, vector = bench.build(dev, seed=seed, K=K, algo=algo)
    ref = flat(getattr(horc, algo)(vector, curr, prev, False))
    bench.declare_structure(curr, "hip")
    got = flat(hg.jvp_fn_mapping[algo](vector, curr, prev, False))
    n = np.linalg.norm(truth)
    print(f"seed {seed}: product (default path) {np.linalg.norm(got - truth) / n:.2e} | reference algorithm, fp32 on ATen {np.linalg.norm(ref - truth) / n:.2e}"
          f" | product vs reference-fp32 {np.linalg.norm(got - ref) / np.linalg.norm(ref):.2e}", flush=True)
