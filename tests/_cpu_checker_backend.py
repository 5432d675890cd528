"""CPU checker backend — TEST INFRASTRUCTURE.

Implements the interface of ``betty_amd.backend.HipBackend`` on CPU tensors by calling the C
oracle (oracle/liborc.so).  It lets the ``-m "not gpu"`` suite exercise the *host orchestration*
of betty_amd.hypergradient (flat state, views handed to autograd, sync/DDP semantics) on a box
without a GPU.  It lives under tests/ on purpose: the package never imports it and has no CPU
path of its own.
"""
import ctypes
import os
import subprocess

import torch

from betty_amd.flat import layout_for

_ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


def load_oracle_lib():
    so = os.path.join(_ORACLE_DIR, "liborc.so")
    src = os.path.join(_ORACLE_DIR, "recurrence.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    f, d, i64, vp = ctypes.c_float, ctypes.c_double, ctypes.c_int64, ctypes.c_void_p
    lib.orc_dot_scaled.restype = d
    lib.orc_dot_scaled.argtypes = [vp, vp, i64, f]
    lib.orc_sqnorm.restype = d
    lib.orc_sqnorm.argtypes = [vp, i64]
    lib.orc_cg_init.restype = None
    lib.orc_cg_init.argtypes = [vp, vp, vp, vp, i64]
    lib.orc_cg_resid.restype = d
    lib.orc_cg_resid.argtypes = [vp, vp, i64, f]
    lib.orc_cg_dir.restype = None
    lib.orc_cg_dir.argtypes = [vp, vp, vp, i64, f, f, f]
    lib.orc_neumann_step.restype = None
    lib.orc_neumann_step.argtypes = [vp, vp, vp, i64, f, f]
    lib.orc_darts_eps.restype = d
    lib.orc_darts_eps.argtypes = [d, d]
    lib.orc_axpy.restype = None
    lib.orc_axpy.argtypes = [vp, vp, i64, f]
    lib.orc_scale_copy.restype = None
    lib.orc_scale_copy.argtypes = [vp, vp, i64, f]
    return lib


def _f32(x):
    return ctypes.c_float(x).value


class CpuCheckerBackend:
    name = "cpu-checker"

    def __init__(self):
        self.orc = load_oracle_lib()
        self._rr = {}
        self._g = {}

    def layout(self, tensors):
        return layout_for(tensors)

    @staticmethod
    def _prep(tensors):
        return [t.detach().to(torch.float32).contiguous() for t in tensors]

    @staticmethod
    def _slices(layout, flat):
        return [flat[s : s + n] for s, n in zip(layout.starts, layout.numels)]

    def flatten(self, layout, tensors, flat, scale=1.0):
        for t, dst in zip(self._prep(tensors), self._slices(layout, flat)):
            self.orc.orc_scale_copy(dst.data_ptr(), t.data_ptr(), t.numel(), scale)

    def scatter(self, layout, flat, tensors, scale=1.0):
        for t, src in zip(tensors, self._slices(layout, flat)):
            self.orc.orc_scale_copy(t.data_ptr(), src.data_ptr(), t.numel(), scale)

    def after_cg(self, layout):
        pass

    def scale_flat(self, flat, scale):
        self.orc.orc_scale_copy(flat.data_ptr(), flat.data_ptr(), flat.numel(), scale)

    def neumann_init(self, layout, vector, v, p):
        for t, a, b in zip(self._prep(vector), self._slices(layout, v), self._slices(layout, p)):
            a.copy_(t.reshape(-1))
            b.copy_(t.reshape(-1))

    def neumann_step(self, layout, hvp, v, p, alpha, out_scale=0.0, hvp_shift=0.0):
        for h, a, b in zip(self._prep(hvp), self._slices(layout, v), self._slices(layout, p)):
            if hvp_shift:
                h = h.reshape(-1) + torch.tensor(hvp_shift, dtype=torch.float32) * a
            self.orc.orc_neumann_step(h.data_ptr(), a.data_ptr(), b.data_ptr(), h.numel(), alpha, out_scale)

    def cg_init(self, layout, vector, x, r, p):
        rr = 0.0
        for t, a, b, c in zip(self._prep(vector), self._slices(layout, x), self._slices(layout, r), self._slices(layout, p)):
            self.orc.orc_cg_init(t.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), t.numel())
            rr += self.orc.orc_sqnorm(t.data_ptr(), t.numel())
        self._rr[id(layout)] = rr
        self._g.pop(id(layout), None)

    def cg_step(self, layout, hvp, x, r, p, cg_alpha, it, out_scale=0.0, variant=None, hvp_shift=0.0):
        hs = self._prep(hvp)
        xs, rs, ps = self._slices(layout, x), self._slices(layout, r), self._slices(layout, p)
        if hvp_shift:
            hs = [h.reshape(-1) + torch.tensor(hvp_shift, dtype=torch.float32) * q for h, q in zip(hs, ps)]
        rr = self._rr[id(layout)]
        den = sum(self.orc.orc_dot_scaled(h.data_ptr(), q.data_ptr(), h.numel(), cg_alpha) for h, q in zip(hs, ps))
        a = _f32(_f32(rr) / _f32(den))
        rr_new = sum(self.orc.orc_cg_resid(h.data_ptr(), q.data_ptr(), h.numel(), a) for h, q in zip(hs, rs))
        b = _f32(_f32(rr_new) / _f32(rr))
        for xx, q, pp in zip(xs, rs, ps):
            self.orc.orc_cg_dir(xx.data_ptr(), q.data_ptr(), pp.data_ptr(), xx.numel(), a, b, out_scale)
        self._rr[id(layout)] = rr_new
        self.last_scalars = (rr, den, a, rr_new, b)

    # phased CG for sharded state: one-element "partial arrays" the caller all-reduces between the phases
    def cg_partials(self, layout, which, it):
        st = self._g.setdefault(id(layout), {})
        if which == 2 and "R" not in st:
            st["R"] = torch.tensor([self._rr[id(layout)]], dtype=torch.float64)
        return st[{0: "P", 1: "Rn", 2: "R"}[which]]

    def cg_phase(self, phase, layout, hvp, x, r, p, cg_alpha, it, out_scale=0.0, hvp_shift=0.0):
        st = self._g.setdefault(id(layout), {})
        hs = self._prep(hvp)
        xs, rs, ps = self._slices(layout, x), self._slices(layout, r), self._slices(layout, p)
        if hvp_shift:
            hs = [h.reshape(-1) + torch.tensor(hvp_shift, dtype=torch.float32) * q for h, q in zip(hs, ps)]
        if phase == 0:
            if it == 0 and "R" not in st:
                st["R"] = torch.tensor([self._rr[id(layout)]], dtype=torch.float64)
            st["P"] = torch.tensor([sum(self.orc.orc_dot_scaled(h.data_ptr(), q.data_ptr(), h.numel(), cg_alpha)
                                        for h, q in zip(hs, ps))], dtype=torch.float64)
        elif phase == 1:
            a = _f32(_f32(float(st["R"][0])) / _f32(float(st["P"][0])))
            st["a"] = a
            st["Rn"] = torch.tensor([sum(self.orc.orc_cg_resid(h.data_ptr(), q.data_ptr(), h.numel(), a)
                                         for h, q in zip(hs, rs))], dtype=torch.float64)
        else:
            b = _f32(_f32(float(st["Rn"][0])) / _f32(float(st["R"][0])))
            for xx, q, pp in zip(xs, rs, ps):
                self.orc.orc_cg_dir(xx.data_ptr(), q.data_ptr(), pp.data_ptr(), xx.numel(), st["a"], b, out_scale)
            st["R"] = st["Rn"].clone()

    def darts_eps(self, layout, vector, R):
        ss = sum(self.orc.orc_sqnorm(t.data_ptr(), t.numel()) for t in self._prep(vector))
        eps = self.orc.orc_darts_eps(ss, R)
        return (torch.tensor(eps, dtype=torch.float64).to(torch.float32), torch.tensor(eps, dtype=torch.float64),
                torch.tensor(ss, dtype=torch.float64))

    def axpy_multi(self, layout, dst, src, coef, mul):
        a = _f32(mul * float(coef)) if coef is not None else _f32(mul)
        for d_, s_ in zip(dst, self._prep(src)):
            assert d_.is_contiguous() and d_.dtype == torch.float32
            self.orc.orc_axpy(d_.data_ptr(), s_.data_ptr(), d_.numel(), a)

    def sama_adam_precondition(self, layout, vector, last_grad, exp_avg, exp_avg_sq, out_flat, beta1, beta2, eps, lr):
        # fp32 ATen evaluation of betty/hypergradient/utils.py:37-63 (the checker for the fused kernel)
        for v, g, m, u, dst in zip(self._prep(vector), self._prep(last_grad), self._prep(exp_avg),
                                   self._prep(exp_avg_sq), self._slices(layout, out_flat)):
            m_old = (m - (1 - beta1) * g) / beta1 if beta1 != 0 else 0
            u_old = (u - (1 - beta2) * g * g) / beta2
            scale = (1 - beta1) * beta2 * u_old - beta1 * (1 - beta2) * g * m_old
            scale = scale / (torch.sqrt(u) + eps) ** 3
            dst.copy_((v * scale * lr).reshape(-1))

