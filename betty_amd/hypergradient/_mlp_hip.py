"""HIP (MFMA) evaluation of the analytic MLP HVP: per-step cache + ``bhg_mlp_hvp`` calls.

The once-per-step quantities (forward activations, ReLU masks, softmax, back-propagated deltas)
are 1/K of the work and are computed with ATen ops into 128-row padded buffers; the K HVPs of a
hypergradient step run on libbhg's fp32 matrix-core kernels (csrc/bhg_mlp.hip).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn.functional as F

from .. import _native

BP = 128  # batch rows of every activation buffer (one 128-row MFMA workgroup tile)


def _pad_rows(t: torch.Tensor) -> torch.Tensor:
    out = torch.zeros((BP,) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
    out[: t.shape[0]] = t
    return out


class HipMLPState:
    def __init__(self, spec, x, y):
        if not x.is_cuda:
            raise _native.NativeLibraryError("WeightedCEMLP(impl='hip') needs CUDA/HIP tensors; there is no CPU fallback")
        self.lib = _native.load()
        self.spec = spec
        Ws = [lin.weight.detach() for lin in spec.layers]
        bs = [lin.bias.detach() for lin in spec.layers]
        L, B = len(Ws), x.shape[0]
        if B > BP:
            raise ValueError(f"WeightedCEMLP(impl='hip') supports batches up to {BP} rows, got {B}")
        if L > _native.BHG_MLP_MAX_LAYERS:
            raise ValueError("too many layers")
        for W in Ws:
            if W.dtype != torch.float32 or not W.is_contiguous():
                raise ValueError("weights must be contiguous fp32")
        # ---- once per step (ATen) -------------------------------------------------------------------
        hs, masks = [x.detach().to(torch.float32)], []
        h = hs[0]
        for l in range(L):
            a = torch.addmm(bs[l], h, Ws[l].t())
            if l + 1 < L:
                m = (a > 0).to(torch.float32)
                h = a * m
                masks.append(m)
                hs.append(h)
            else:
                z = a
        logp = F.log_softmax(z, dim=1)
        p = logp.exp()
        ce = -logp.gather(1, y.reshape(-1, 1)).reshape(-1)
        self.sample_weight = spec.weight_fn(ce.detach())  # graph to prev's parameters
        sd = self.sample_weight.detach().reshape(-1).to(torch.float32) / B
        self.err = p - F.one_hot(y, z.shape[1]).to(torch.float32)
        deltas = [None] * L
        deltas[L - 1] = sd[:, None] * self.err
        for l in range(L - 1, 0, -1):
            deltas[l - 1] = masks[l - 1] * (deltas[l] @ Ws[l])
        self.B, self.L = B, L
        self.Ws, self.hs_raw, self.masks_raw = Ws, hs, masks
        # ---- padded device buffers + descriptor ---------------------------------------------------------
        dev = x.device
        self.h = [_pad_rows(t) for t in hs]
        self.mask = [_pad_rows(t) for t in masks]
        self.delta = [_pad_rows(t) for t in deltas]
        self.prob = _pad_rows(p)
        self.sd = _pad_rows(sd)
        dims = [Ws[0].shape[1]] + [W.shape[0] for W in Ws]
        self.Rh = [torch.zeros(BP, dims[l + 1], device=dev) for l in range(L - 1)]
        self.Rd = [torch.zeros(BP, dims[l + 1], device=dev) for l in range(L)]
        d = _native.Mlp()
        d.L, d.B, d.Bp = L, B, BP
        for i, v in enumerate(dims):
            d.dims[i] = v
        for l in range(L):
            d.W[l] = Ws[l].data_ptr()
            d.h[l] = self.h[l].data_ptr()
            d.delta[l] = self.delta[l].data_ptr()
            d.Rd[l] = self.Rd[l].data_ptr()
            if l + 1 < L:
                d.mask[l] = self.mask[l].data_ptr()
                d.Rh[l] = self.Rh[l].data_ptr()
        d.prob, d.sd = self.prob.data_ptr(), self.sd.data_ptr()
        d.ridge2 = 0.0  # the ridge's 2*ridge*I is applied by the recurrence kernel (spec.hvp_shift)
        n_part = int(self.lib.bhg_mlp_partial_floats(ctypes.byref(d)))
        self.partial = torch.empty(max(n_part, 1), device=dev)
        d.partial, d.partial_floats = self.partial.data_ptr(), n_part
        self.desc = d
        # HVP outputs are consumed by the recurrence kernel on the same stream before the next
        # HVP is launched, so one set of output tensors serves all K iterations.
        self.out = []
        for lin in spec.layers:
            self.out += [torch.empty_like(lin.weight), torch.empty_like(lin.bias)]
        self._out_tab, self._out_keep = _native.ptr_array([t.data_ptr() for t in self.out])

    def hvp(self, direction_views):
        dirs = []
        for t in direction_views:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
                t = t.detach().to(torch.float32).contiguous().clone()
            dirs.append(t)
        tab, _keep = _native.ptr_array([t.data_ptr() for t in dirs])
        _native.check(
            self.lib.bhg_mlp_hvp(ctypes.byref(self.desc), tab, self._out_tab, int(torch.cuda.current_stream().cuda_stream)),
            "bhg_mlp_hvp",
        )
        return self.out

    def mixed_coeff(self, dir_views):
        """c_i = (p_i - onehot_i) . Rz_i(direction) / B — one R-forward, once per step (ATen)."""
        Vs, cs = dir_views[0::2], dir_views[1::2]
        Rh = None
        for l in range(self.L):
            Ra = torch.addmm(cs[l], self.hs_raw[l], Vs[l].t())
            if Rh is not None:
                Ra = Ra + Rh @ self.Ws[l].t()
            if l + 1 < self.L:
                Rh = self.masks_raw[l] * Ra
        return (self.err * Ra).sum(1) / self.B
