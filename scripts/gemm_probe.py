#!/usr/bin/env python
"""Skinny-GEMM probe: bhg_mlp_forward on a [K -> 2048 -> 10] net (batch 100 / 128 rows) for several K, so the R-forward
GEMM kernel (128 x 2048 x K, split-K) can be timed from a rocprofv3 kernel trace as a function of its K-loop length.
usage: gemm_probe.py K1 K2 ...   (run under rocprofv3 --kernel-trace; scripts/print_gemm_probe.py prints the table)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from betty_amd import _native
from betty_amd.hypergradient._mlp_hip import _Buffers

lib = _native.load()
dev = torch.device("cuda:0")
Ks = [int(a) for a in sys.argv[1:]] or [1536, 3072, 6144]
for K in Ks:
    dims = (K, 2048, 10)
    buf = _Buffers(dims, 128, dev, lib)
    W = [torch.randn(2048, K, device=dev) * 0.01, torch.randn(10, 2048, device=dev) * 0.01]
    b = [torch.zeros(2048, device=dev), torch.zeros(10, device=dev)]
    d = buf.desc
    d.B = 100
    for l in range(2):
        d.W[l] = W[l].data_ptr()
    buf.h[0][:100].copy_(torch.randn(100, K, device=dev))
    tab, keep = _native.ptr_array([t.data_ptr() for t in b])
    st = int(torch.cuda.current_stream().cuda_stream)
    for _ in range(12):
        _native.check(lib.bhg_mlp_forward(ctypes.byref(d), tab, buf.labels.data_ptr(), buf.ce.data_ptr(), st), "fwd")
    torch.cuda.synchronize()
