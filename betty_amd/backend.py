"""Vector backend: the only place Python calls into libbhg's kernels.

There is exactly one product backend, :class:`HipBackend` (gfx950 kernels through the C ABI of
``include/bhg.h``).  It refuses to run when the library is missing, when no GPU is visible or
when a tensor is not a contiguous fp32 device tensor — there is no CPU path in the product.

``use_backend`` exists so that the *tests* can drive the host orchestration (sync semantics,
DDP/gloo behaviour) with a checker backend living under ``tests/``; nothing in the package ever
selects anything but :class:`HipBackend`.
"""
from __future__ import annotations

import contextlib
import ctypes
import weakref
from typing import List, Optional, Sequence

import torch

from . import _native
from .flat import FlatLayout, layout_for


def _stream_ptr() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class HipBackend:
    name = "hip"

    def __init__(self):
        _native.load()   # fail loudly now if the extension has not been built
        if not torch.cuda.is_available():
            raise _native.NativeLibraryError(
                "betty_amd needs a visible MI355X (torch.cuda.is_available() is False); "
                "there is no CPU fallback."
            )
        self.cg_variant = _native.BHG_CG_AUTO
        # collectives this process has in flight on a communication stream (betty_amd.distributed): while > 0 the
        # resident CG kernel is not eligible — it needs every CU for its grid barrier and RCCL's channel kernels
        # hold some (the barrier would spin until the collective ends, or time out)
        self.collectives_in_flight = 0
        self._health = []  # (event, pinned flag copy): non-blocking time-out checks of finished resident solves
        self._fused_flags = weakref.WeakValueDictionary()   # id -> 1-element int32 view of a fused workspace's time-out word

    # -- helpers ---------------------------------------------------------------------------
    def layout(self, tensors: Sequence[torch.Tensor]) -> FlatLayout:
        return layout_for(tensors)

    @staticmethod
    def _prep(tensors: Sequence[torch.Tensor], layout: FlatLayout, writable: bool = False) -> List[torch.Tensor]:
        """Validate / normalise a tensor list for the kernels: fp32, contiguous, 16-B aligned,
        on the layout's device, sizes matching the layout.  Read-only inputs are converted when
        needed; writable ones must already comply."""
        if len(tensors) != layout.T:
            raise ValueError(f"expected {layout.T} tensors, got {len(tensors)}")
        out = []
        for t, n in zip(tensors, layout.numels):
            if t.numel() != n:
                raise ValueError("tensor sizes do not match the flat layout")
            if not t.is_cuda:
                raise _native.NativeLibraryError("HipBackend got a CPU tensor; there is no CPU fallback")
            ok = t.dtype == torch.float32 and t.is_contiguous() and (t.data_ptr() % 16 == 0 or n == 0)
            if not ok:
                if writable:
                    raise ValueError("in-place target must be a contiguous, 16-byte aligned fp32 tensor")
                t = t.detach().to(torch.float32).contiguous()
                if t.data_ptr() % 16 != 0:
                    t = t.clone(memory_format=torch.contiguous_format)
            out.append(t)
        return out

    @staticmethod
    def _table(tensors: Sequence[torch.Tensor]):
        return _native.ptr_array([t.data_ptr() for t in tensors])

    # -- multi-tensor <-> flat ----------------------------------------------------------------
    def flatten(self, layout: FlatLayout, tensors, flat: torch.Tensor, scale: float = 1.0) -> None:
        ts = self._prep(tensors, layout)
        tab, _keep = self._table(ts)
        _native.check(
            self.lib.bhg_flatten(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, flat.data_ptr(),
                                 scale, layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_flatten",
        )

    def scatter(self, layout: FlatLayout, flat: torch.Tensor, tensors, scale: float = 1.0) -> None:
        ts = self._prep(tensors, layout, writable=True)
        tab, _keep = self._table(ts)
        _native.check(
            self.lib.bhg_scatter(flat.data_ptr(), tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks,
                                 scale, layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_scatter",
        )

    def scale_flat(self, flat: torch.Tensor, scale: float) -> None:
        _native.check(self.lib.bhg_scale_flat(flat.data_ptr(), flat.numel(), scale, _stream_ptr()), "bhg_scale_flat")

    # -- Neumann ---------------------------------------------------------------------------------
    def neumann_init(self, layout, vector, v, p) -> None:
        ts = self._prep(vector, layout)
        tab, _keep = self._table(ts)
        _native.check(
            self.lib.bhg_neumann_init(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, v.data_ptr(),
                                      None if p is None else p.data_ptr(), layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_neumann_init",
        )

    def neumann_step(self, layout, hvp, v, p, alpha: float, out_scale: float = 0.0, hvp_shift: float = 0.0) -> None:
        tab, _keep = self._cached_table(hvp, layout)
        _native.check(
            self.lib.bhg_neumann_step(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, v.data_ptr(),
                                      p.data_ptr(), alpha, out_scale, hvp_shift, layout.workspace.data_ptr(),
                                      _stream_ptr()),
            "bhg_neumann_step",
        )

    # -- CG ------------------------------------------------------------------------------------------
    def cg_init(self, layout, vector, x, r, p, keep_mask: Optional[int] = None):
        """x = 0 (x None: not materialised), r = p = vector, r.r partials (cg.py:34-36).  keep_mask (bhg_cg_init_masked): only the
        tensors whose bit is set have their slices of r / p written — for a solver that reads the others from the right-hand side's
        own tensors.  Returns the tensors the kernel read (the caller's, or fp32 / contiguous copies of them)."""
        ts = self._prep(vector, layout)
        tab, _keep = self._table(ts)
        mask = (1 << 64) - 1 if keep_mask is None else int(keep_mask) & ((1 << 64) - 1)
        _native.check(
            self.lib.bhg_cg_init_masked(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, None if x is None else x.data_ptr(),
                                        r.data_ptr(), p.data_ptr(), mask, layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_cg_init_masked",
        )
        return ts

    def _cached_table(self, tensors, layout):
        """Pointer table of an HVP tensor list; a provider that hands back the SAME output tensors every
        iteration (the analytic HVPs do) is validated once.  Weak references prove the ids still name the same
        live objects without keeping an autograd HVP (fresh tensors every call) alive longer than the caller does."""
        key = (id(layout),) + tuple(map(id, tensors))
        cached = getattr(self, "_tab_cache", None)
        if cached is not None and cached[0] == key and all(r() is t for r, t in zip(cached[3], tensors)):
            return cached[1], cached[2]
        ts = self._prep(tensors, layout)
        tab, keep = self._table(ts)
        self._tab_cache = None
        if all(a is b for a, b in zip(ts, tensors)):   # nothing had to be converted: the table names the caller's tensors
            self._tab_cache = (key, tab, keep, [weakref.ref(t) for t in tensors])   # `keep` owns the host pointer array
        return tab, keep

    def _pick_variant(self, layout, it: int, variant: Optional[int]) -> int:
        """The variant is chosen ONCE per solve (iteration 0) and kept: the resident kernel's barrier targets count
        two arrivals per workgroup and completed iteration, so the two variants must not alternate inside a solve."""
        if it > 0 and getattr(layout, "_cg_variant", None) is not None:
            return layout._cg_variant
        v = self.cg_variant if variant is None else variant
        if v == _native.BHG_CG_AUTO:
            # the library's own AUTO predicate (capacity, residency census, LDS grant for the larger instances)
            resident = self.collectives_in_flight == 0 and bool(self.lib.bhg_cg_resident_usable(int(layout.n_chunks)))
            v = _native.BHG_CG_RESIDENT if resident else _native.BHG_CG_STREAM
        layout._cg_variant = v
        return v

    def cg_step(self, layout, hvp, x, r, p, cg_alpha: float, it: int, out_scale: float = 0.0,
                variant: Optional[int] = None, hvp_shift: float = 0.0) -> None:
        tab, _keep = self._cached_table(hvp, layout)
        _native.check(
            self.lib.bhg_cg_step(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, x.data_ptr(),
                                 r.data_ptr(), p.data_ptr(), cg_alpha, it, out_scale, hvp_shift,
                                 self._pick_variant(layout, it, variant),
                                 layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_cg_step",
        )

    # -- phased CG for sharded state (betty_amd/global_hvp.py) ----------------------------------------------------
    def cg_phase(self, phase: int, layout, hvp, x, r, p, cg_alpha: float, it: int, out_scale: float = 0.0,
                 hvp_shift: float = 0.0) -> None:
        tab, _keep = self._cached_table(hvp, layout)
        _native.check(
            self.lib.bhg_cg_phase(int(phase), tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, x.data_ptr(),
                                  r.data_ptr(), p.data_ptr(), cg_alpha, it, out_scale, hvp_shift,
                                  layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_cg_phase",
        )

    def cg_partials(self, layout, which: int, it: int) -> torch.Tensor:
        """The per-block partial sums (float64 view into the layout's workspace) a caller with sharded state must
        all-reduce(SUM) after bhg_cg_init (which=2), phase 0 (which=0) and phase 1 (which=1)."""
        ws = layout.workspace
        off = int(self.lib.bhg_cg_partials_dev(ws.data_ptr(), int(which), int(it))) - ws.data_ptr()
        n = int(self.lib.bhg_cg_partials_count())
        return ws[off:off + 8 * n].view(torch.float64)

    def after_cg(self, layout) -> None:
        """Health check of the resident kernel's grid barrier WITHOUT stalling the host: the time-out word of the
        solve that was just enqueued is copied to pinned memory behind it, and the copies of EARLIER solves that
        have completed by now are inspected.  A barrier that gave up (the GPU became shared mid-run) has already
        NaN-poisoned that solve's result; here the cause is raised instead of leaving a silent NaN hypergradient."""
        self.check_health(block=False)
        if getattr(layout, "_cg_variant", None) != _native.BHG_CG_RESIDENT:
            return
        ws = layout.workspace
        off = int(self.lib.bhg_cg_timeout_flag_dev(ws.data_ptr())) - ws.data_ptr()
        host = torch.empty(1, dtype=torch.int32).pin_memory()
        host.copy_(ws[off:off + 4].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._health.append((ev, host))

    @property
    def lib(self):
        """The library of the moment: the product, or the measurement build while _native.use_ab() is on."""
        return _native.load()

    def watch_fused_workspace(self, flag_view: torch.Tensor) -> None:
        """Register the time-out word of a fused MLP solver's workspace (bhg_mlp_timeout_flag_dev): read — and cleared — by
        check_health(block=True).  No per-step cost: the pollers poison the result with NaN themselves, this is the diagnosis."""
        self._fused_flags[id(flag_view)] = flag_view

    def check_health(self, block: bool = True) -> None:
        """Raise if a resident-CG grid barrier timed out in a solve that has finished (block=True: wait for all), or — block=True
        only, it synchronises — if the in-launch beta exchange of a fused MLP solve gave up (csrc/mlp/wskp.inc poll_beta)."""
        if block:
            for flag in list(self._fused_flags.values()):
                if int(flag.item()) != 0:
                    flag.zero_()
                    raise _native.NativeLibraryError(
                        "bhg_mlp_cg_solve: workgroups of the first chain launch waited ~1 s for the step's beta, which another "
                        "workgroup of the same launch publishes, and gave up; the hypergradient of that call is NaN.  The launch "
                        "was built without its publisher (a library bug) or the device could not keep the launch resident.")
        keep = []
        for ev, host in self._health:
            if block:
                ev.synchronize()
            if ev.query():
                if int(host.item()) != 0:
                    self._health = []
                    raise _native.NativeLibraryError(
                        "k_cg_resident: a grid barrier timed out (the GPU is shared, partitioned, or another stream held "
                        "CUs during the solve); the hypergradient of that call is NaN. Set HipBackend.cg_variant = "
                        "BHG_CG_STREAM on shared devices.")
            else:
                keep.append((ev, host))
        self._health = keep

    def cg_barrier_timed_out(self, layout) -> bool:
        """True if a grid barrier of the resident CG kernel gave up polling since the last cg_init
        (the GPU was shared or partitioned mid-run).  The kernel also poisons its result with NaN, so
        this is the diagnosis, not the alarm.  Synchronises the host (debug / health checks only)."""
        ws = layout.workspace
        off = int(self.lib.bhg_cg_timeout_flag_dev(ws.data_ptr())) - ws.data_ptr()
        return bool(ws[off:off + 4].view(torch.int32).item())

    def cg_scalars(self, layout) -> torch.Tensor:
        """{rr_old, pHp, alpha, rr_new, beta} of the last CG step (device -> host copy; debug)."""
        ws = layout.workspace
        return ws[:64].view(torch.float64)[:5].clone()

    # -- DARTS ---------------------------------------------------------------------------------------
    def darts_eps(self, layout, vector, R: float):
        """Returns (eps_f32, eps_f64, sum_of_squares_f64) as 0-dim device tensors; no host synchronisation."""
        ts = self._prep(vector, layout)
        tab, _keep = self._table(ts)
        out = torch.empty(2, dtype=torch.float64, device=layout.device)
        eps32 = torch.empty(1, dtype=torch.float32, device=layout.device)
        _native.check(
            self.lib.bhg_darts_eps(tab, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks, float(R),
                                   out.data_ptr(), eps32.data_ptr(), layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_darts_eps",
        )
        return eps32[0], out[1], out[0]

    def sama_adam_precondition(self, layout, vector, last_grad, exp_avg, exp_avg_sq, out_flat, beta1, beta2, eps, lr) -> None:
        tabs, keep = [], []
        for lst in (vector, last_grad, exp_avg, exp_avg_sq):
            prepared = self._prep(lst, layout)  # kept alive until the launch below is enqueued
            tab, k = self._table(prepared)
            tabs.append(tab)
            keep.append((prepared, k))
        _native.check(
            self.lib.bhg_sama_adam_precondition(tabs[0], tabs[1], tabs[2], tabs[3], layout.T, layout.chunks_dev.data_ptr(),
                                                layout.n_chunks, out_flat.data_ptr(), float(beta1), float(beta2), float(eps),
                                                float(lr), layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_sama_adam_precondition",
        )

    def axpy_multi(self, layout, dst, src, coef: Optional[torch.Tensor], mul: float) -> None:
        d = self._prep(dst, layout, writable=True)
        s = self._prep(src, layout)
        td, _k1 = self._table(d)
        ts, _k2 = self._table(s)
        _native.check(
            self.lib.bhg_axpy_multi(td, ts, layout.T, layout.chunks_dev.data_ptr(), layout.n_chunks,
                                    coef.data_ptr() if coef is not None else None, mul,
                                    layout.workspace.data_ptr(), _stream_ptr()),
            "bhg_axpy_multi",
        )


_backend = None
_override = None


def get_backend():
    """The active vector backend: :class:`HipBackend` unless a test installed a checker."""
    global _backend
    if _override is not None:
        return _override
    if _backend is None:
        _backend = HipBackend()
    return _backend


@contextlib.contextmanager
def use_backend(backend):
    """TEST HOOK ONLY: run host orchestration against another object with HipBackend's
    interface (see tests/_cpu_checker_backend.py).  Never used inside the package."""
    global _override
    prev = _override
    _override = backend
    try:
        yield backend
    finally:
        _override = prev
