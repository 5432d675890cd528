"""Distance to the fp64 truth of the metric workload's hypergradient (cfg 2, CG K = 20) for the product's arms:
fused / un-fused solver x BHG_MLP_WSK = 0 / 2.  Prints one line per arm."""
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
c64, p64, v64 = bench.build(dev, seed=0, dtype=torch.float64, K=K, algo=algo)
truth = flat(getattr(horc, algo)(v64, c64, p64, False))
del c64, p64, v64
res = {}
for wsk in ("0", "2", "1"):
    os.environ["BHG_MLP_WSK"] = wsk
    for fused in (True, False):
        curr, prev, vector = bench.build(dev, seed=0, K=K, algo=algo)
        bench.declare_structure(curr, "hip", fused=fused)
        out = flat(hg.jvp_fn_mapping[algo](vector, curr, prev, False))
        res[(wsk, fused)] = out
        print(f"wsk={wsk} fused={fused}: rel err vs fp64 truth {np.linalg.norm(out - truth) / np.linalg.norm(truth):.3e}")
    a, b = res[(wsk, True)], res[(wsk, False)]
    print(f"wsk={wsk}: fused vs un-fused {np.linalg.norm(a - b) / np.linalg.norm(b):.3e}")
