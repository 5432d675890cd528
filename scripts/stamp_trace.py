#!/usr/bin/env python
"""Per-workgroup time stamps of the projected iteration's closing launches (k_graw, k_proj_step) from the measurement build
libbhg_stamps.so (make -C betty_amd/csrc stamps).  usage: BHG_LIB=betty_amd/csrc/libbhg_stamps.so python scripts/stamp_trace.py [key=int ...]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BHG_LIB", os.path.join(ROOT, "betty_amd", "csrc", "libbhg_stamps.so"))
import bench
from betty_amd import _native
from betty_amd import hypergradient as hg
lib = _native.load()
for kv in sys.argv[1:]:
    k, _, v = kv.partition("=")
    _native.debug_set(k, int(v))
curr, prev, vector = bench.build(torch.device("cuda:0"), 0, K=20, algo="cg")
bench.declare_structure(curr, "hip", fused=True)
def step():
    for p in prev.parameters():
        p.grad = None
    hg.jvp_fn_mapping["cg"](vector, curr, prev, True)
for _ in range(3):
    step()
torch.cuda.synchronize()
lib.bhg_debug_stamps_enable.argtypes = [ctypes.c_int]
lib.bhg_debug_stamps_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.bhg_debug_stamps_enable(1) == 0
step()
torch.cuda.synchronize()
KN, BN, SN = 4, 4096, 8
buf = np.zeros(KN * BN * SN, dtype=np.uint64)
assert lib.bhg_debug_stamps_read(buf.ctypes.data, buf.size) == 0
st = buf.reshape(KN, BN, SN).astype(np.int64)
def show(name, kid, classes):
    a = st[kid]
    live = a[:, 0] > 0
    if not live.any():
        print(name, "no stamps"); return
    t0 = a[live, 0].min()
    end = a[live, 4].max()
    print(f"{name}: {int(live.sum())} workgroups, span {(end - t0) * 0.01:.2f} us")
    for cname, pick in classes:
        sel = a[live][pick(a[live])]
        if len(sel) == 0:
            continue
        rel = (sel - t0) * 0.01
        cols = []
        for s in range(5):
            v = rel[:, s][sel[:, s] > 0]
            cols.append("   -   " if len(v) == 0 else f"{np.median(v):5.2f}/{v.max():5.2f}")
        print(f"   {cname:22s} n={len(sel):4d}  entry {cols[0]}  s1 {cols[1]}  s2 {cols[2]}  s3 {cols[3]}  exit {cols[4]}   (median/max us since the launch's first stamp)")
# classes by which slots a workgroup stamped: k_graw's tile workgroups stamp s2 (MFMAs done) and s3, its small-slice blocks only
# s1 (step length known); k_pstep's update blocks stamp s1 (beta known), its small blocks do not
show("k_graw", 0, [("small slices (alpha)", lambda a: (a[:, 2] == 0) & (a[:, 4] > 0)), ("tiles", lambda a: a[:, 2] > 0)])
show("k_pstep / update blocks of k_wskpl (s2: operands requested, s3: partial sums in LDS, s1: beta known)", 1,
     [("update blocks", lambda a: a[:, 1] > 0), ("small blocks", lambda a: (a[:, 1] == 0) & (a[:, 4] > 0))])
show("k_wskpl tiles (entry: before the K loop, s1: after it, s2: partial tiles in LDS, s3: beta polled)", 2, [("tiles", lambda a: a[:, 4] > 0)])
# the three share one clock: offsets of the classes of k_wskpl relative to its first update block
a1, a2 = st[1], st[2]
if (a1[:, 0] > 0).any() and (a2[:, 0] > 0).any():
    print("k_wskpl: first tile stamp - first update-block stamp = %.2f us" % ((a2[a2[:, 0] > 0, 0].min() - a1[a1[:, 0] > 0, 0].min()) * 0.01))
