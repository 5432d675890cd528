#!/usr/bin/env python
"""bhg_bn_backward_vjp (csrc/bhg_bn.hip) in isolation: microseconds per call and achieved bandwidth on its ALGORITHMIC bytes — the two
launches read a, gy, x twice and write dx, dgy once: 32 bytes per element — against the 8 TB/s HBM peak, on the batch-norm shapes of
BASELINE cfg 3 (ResNet-12, batch 25) and on shapes past the 256 MiB Infinity Cache.  ATen's decomposed double backward of the same layer
(what an undeclared nn.BatchNorm2d costs per Hessian-vector product) is timed beside it."""
import json
import sys
import os
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from betty_amd import nn as bnn  # noqa: E402

dev = "cuda:0"
shapes = [(25, 32, 84, 84), (25, 80, 42, 42), (25, 160, 21, 21), (25, 320, 10, 10), (64, 256, 56, 56), (128, 64, 112, 112)]
rows = []
for shape in shapes:
    N, C, H, W = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g).to(dev)
    gy = torch.randn(shape, generator=g).to(dev)
    a = torch.randn(shape, generator=g).to(dev)
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    b, c = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    mean = x.mean((0, 2, 3))
    invstd = (x.var((0, 2, 3), unbiased=False) + 1e-5).rsqrt()
    for _ in range(3):
        bnn._vjp_hip(x, gy, a, gamma, mean, invstd, b, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        bnn._vjp_hip(x, gy, a, gamma, mean, invstd, b, c)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    nbytes = 32.0 * x.numel()
    # ATen's own double backward of the same layer (one Hessian-vector-product's worth: the VJP of batch norm's backward)
    xr, gyr, gr = x.clone().requires_grad_(True), gy.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    beta = torch.zeros(C, device=dev, requires_grad=True)

    def aten():
        y = F.batch_norm(xr, None, None, gr, beta, True, 0.1, 1e-5)
        first = torch.autograd.grad(y, (xr, gr, beta), gyr, create_graph=True)
        phi = (first[0] * a).sum() + (first[1] * b).sum() + (first[2] * c).sum()
        return torch.autograd.grad(phi, (xr, gyr, gr))

    for _ in range(2):
        aten()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n2 = 10
    for _ in range(n2):
        aten()
    torch.cuda.synchronize()
    us_aten = 1e6 * (time.perf_counter() - t0) / n2
    rows.append({"shape": list(shape), "elements": x.numel(), "working_set_MB": 5 * 4 * x.numel() / 1e6, "us_per_call": us,
                 "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_8TBps": nbytes / (us * 1e-6) / 1e9 / 8000.0,
                 "aten_forward_backward_double_backward_us": us_aten})
    print(f"{str(shape):22s} {x.numel() / 1e6:7.2f} M elements: {us:8.1f} us per call = {nbytes / (us * 1e-6) / 1e12:.2f} TB/s on 32 B/element "
          f"({100 * nbytes / (us * 1e-6) / 8e12:.0f} % of 8 TB/s) | ATen forward + backward + double backward of the layer: {us_aten:9.1f} us", flush=True)
print(json.dumps({"kernel": "bhg_bn_backward_vjp (k_bn_vjp_stats + k_bn_vjp_apply)", "rows": rows}))
