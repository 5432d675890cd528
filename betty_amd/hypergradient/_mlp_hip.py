"""HIP (MFMA) evaluation of the analytic MLP HVP: per-step cache + ``bhg_mlp_*`` calls.

Per hypergradient step:
  1. ``bhg_mlp_forward``   activations, ReLU masks, softmax, per-sample CE           (native, ~8 launches)
  2. the sample weights ``s = weight_fn(ce)`` come from the UPPER problem's module      (PyTorch: it is
     an arbitrary user module whose graph the final mixed VJP needs)
  3. ``bhg_mlp_backward``  back-propagated deltas                                      (native)
  4. K x ``bhg_mlp_hvp``   the Hessian-vector products on the fp32 matrix cores        (native)
  5. ``bhg_mlp_mixed_coeff`` one R-forward for the mixed second derivative, then a backward through
     ``s``'s graph                                                                    (native + PyTorch)
All [Bp = 128, d] activation buffers are allocated once per (shapes, device) and reused across steps.
Round 6: the head kernels take up to 256 classes (chunks of 4 * JMAX classes, csrc/mlp/head_body.inc; rounds 1-5: 32), so such heads
have the fused solvers too; beyond that — or with a feature width that is not a multiple of 4 — steps 1, 3 and 5 still run natively (the
output layer as one more split-K product + a row kernel: k_softmax_ce_rows / k_coeff_rows in csrc/bhg_mlp.hip) and the K loop is
K x (HVP kernels + recurrence kernel).  Networks whose input / hidden widths are not multiples of 32 run on a zero-padded twin (PaddedHipMLPState below).
There is no ATen arithmetic in this file.
"""
from __future__ import annotations

import ctypes
import weakref

import torch

from .. import _native

TILE_M = 128  # activation buffers hold a multiple of 128 rows (the MFMA workgroup tile height)

_BUFFERS = weakref.WeakKeyDictionary()   # first nn.Linear of an inner network -> {(dims, BP, device): _Buffers}



class _Buffers:
    """Device buffers + descriptor for one (dims, device); contents are rewritten every step."""

    def __init__(self, dims, BP, device, lib):
        L = len(dims) - 1
        self.BP = BP
        self.dims, self.L = dims, L
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        self.h = [z(BP, dims[l]) for l in range(L)]
        self.mask = [z(BP, dims[l + 1]) for l in range(L - 1)]
        self.delta = [z(BP, dims[l + 1]) for l in range(L)]
        self.prob = z(BP, dims[L])
        self.sd = z(BP)
        self.ce = z(BP)
        self.coeff = z(BP)
        self.labels = torch.zeros(BP, dtype=torch.int64, device=device)
        self.Rh = [z(BP, dims[l + 1]) for l in range(L - 1)]
        self.Rd = [z(BP, dims[l + 1]) for l in range(L)]
        d = _native.Mlp()
        d.L, d.Bp = L, BP
        for i, v in enumerate(dims):
            d.dims[i] = v
        for l in range(L):
            d.h[l] = self.h[l].data_ptr()
            d.delta[l] = self.delta[l].data_ptr()
            d.Rd[l] = self.Rd[l].data_ptr()
            if l + 1 < L:
                d.mask[l] = self.mask[l].data_ptr()
                d.Rh[l] = self.Rh[l].data_ptr()
        d.prob, d.sd = self.prob.data_ptr(), self.sd.data_ptr()
        d.ridge2 = 0.0  # the ridge's 2*ridge*I is applied by the recurrence kernel (spec.hvp_shift)
        d.B = 1
        n_part = int(lib.bhg_mlp_partial_floats(ctypes.byref(d)))
        self.partial = torch.empty(max(n_part, 1), dtype=torch.float32, device=device)
        d.partial, d.partial_floats = self.partial.data_ptr(), n_part
        self.desc = d
        self.out = None   # HVP output tensors, allocated with the first problem that uses these buffers
        self.native_prepare = bool(lib.bhg_mlp_supports_native_prepare(ctypes.byref(d)))


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


class FusedSolve:
    """Token of ONE run of a fused solver: returned by ``cg_solve`` / ``neumann_solve``, handed back to ``mixed_coeff``
    so the state knows — by object identity, not by comparing addresses of views — that the direction it is asked about
    is the solution of exactly that run (whose Rz the solver accumulated on its way).  A hand-driven ``hvp()`` or another
    solve on the same state retires the token."""

    __slots__ = ("kind", "alpha", "K", "layout", "v_last", "materialised", "projected")

    def __init__(self, kind, alpha, K, layout, v_last=None, materialised=True, projected=0):
        self.kind, self.alpha, self.K, self.layout, self.v_last, self.materialised = kind, alpha, K, layout, v_last, materialised
        self.projected = projected   # Neumann: what bhg_mlp_neumann_solve reported (its closing pass already summed Rz(v_K))

    def __bool__(self):
        return True


class HipMLPState:
    def __init__(self, spec, x, y):
        if not x.is_cuda:
            raise _native.NativeLibraryError("WeightedCEMLP(impl='hip') needs CUDA/HIP tensors; there is no CPU fallback")
        self.lib = lib = _native.load()
        self.spec = spec
        Ws = [lin.weight.detach() for lin in spec.layers]
        bs = [lin.bias.detach() for lin in spec.layers]
        L, B = len(Ws), x.shape[0]
        BP = (B + TILE_M - 1) // TILE_M * TILE_M
        if L > _native.BHG_MLP_MAX_LAYERS:
            raise ValueError("too many layers")
        for t in Ws + bs:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.data_ptr() % 16 != 0:
                raise ValueError("weights and biases must be contiguous, 16-byte aligned fp32 tensors")
        dims = tuple([Ws[0].shape[1]] + [W.shape[0] for W in Ws])
        # one set of buffers per inner network and batch tile count, in a WEAK-keyed map on the network's first layer: the buffers
        # die with the module (round 3 kept them in a plain dict keyed by id(), never evicted, and an id recycled by a new module
        # would have found a dead network's buffers); nothing is attached to the module itself (deepcopy / pickling stay clean)
        key = (dims, BP, str(x.device))
        owner = _BUFFERS.setdefault(spec.layers[0], {})
        buf = owner.get(key)
        if buf is None:
            buf = owner[key] = _Buffers(dims, BP, x.device, lib)
        self.buf, self.B, self.L, self.Ws = buf, B, L, Ws
        d = buf.desc
        d.B = B
        for l in range(L):
            d.W[l] = Ws[l].data_ptr()
        self.desc = d
        # padded input batch and labels (rows >= B of h[0] stay zero from allocation)
        if getattr(buf, "last_B", B) > B:  # a smaller batch than last time: clear the now-unused rows
            buf.h[0][B:].zero_()
        buf.last_B = B
        xs, ys = x.detach().reshape(B, -1), y.detach().reshape(-1)
        # raw pointers go into a kernel launched on the CURRENT device: the batch must live on the buffers' device and that device must
        # be the current one — anything else takes copy_(), which handles cross-device batches
        same_dev = (xs.device == buf.h[0].device and ys.device == xs.device and xs.is_cuda
                    and torch.cuda.current_device() == (xs.device.index if xs.device.index is not None else torch.cuda.current_device()))
        if (same_dev and xs.dtype == torch.float32 and xs.is_contiguous() and xs.data_ptr() % 16 == 0 and dims[0] % 4 == 0
                and ys.dtype == torch.int64 and ys.is_contiguous()):
            _native.check(lib.bhg_mlp_stage_batch(ctypes.byref(d), xs.data_ptr(), ys.data_ptr(), buf.labels.data_ptr(), _stream()),
                          "bhg_mlp_stage_batch")   # one launch for both
        else:
            buf.h[0][:B].copy_(xs.to(torch.float32))
            buf.labels[:B].copy_(ys)

        # the once-per-step passes on the chain's packed operands when the network takes that form (include/bhg.h: the weights are
        # packed first, the hidden layers behind the first run as one launch each) — decided per call: an A/B key may switch it
        d.prepacked = 0
        # (fused or not: every arm of a problem shares ONE set of activations, masks and deltas)
        self.packed_prepare = bool(buf.native_prepare and lib.bhg_mlp_supports_packed_prepare(ctypes.byref(d)))
        if self.packed_prepare:
            fws = self._fused_ws(x.device)
            bias_tab, self._bias_keep = _native.ptr_array([b.data_ptr() for b in bs])
            _native.check(lib.bhg_mlp_forward_packed(ctypes.byref(d), bias_tab, buf.labels.data_ptr(), buf.ce.data_ptr(), fws.data_ptr(),
                                                     fws.numel(), _stream()), "bhg_mlp_forward_packed")
            ce = buf.ce[:B]
        elif buf.native_prepare:
            bias_tab, self._bias_keep = _native.ptr_array([b.data_ptr() for b in bs])
            _native.check(lib.bhg_mlp_forward(ctypes.byref(d), bias_tab, buf.labels.data_ptr(), buf.ce.data_ptr(), _stream()),
                          "bhg_mlp_forward")
            ce = buf.ce[:B]
        else:
            raise _native.NativeLibraryError("bhg_mlp_supports_native_prepare refused this network (layer count / batch padding): there is no ATen path")
        # sample weights: the declared closed form of the meta-weight-net (one launch, writes s / B where bhg_mlp_backward reads it),
        # else the upper problem's module through autograd (keeps the graph to prev's parameters)
        self.native_upper, self.sample_weight = False, None
        wn = getattr(spec, "weight_net", None)
        if wn is not None and buf.native_prepare:
            self._upper_slots = wn.slots(list(spec.prev.trainable_parameters()))
            ts = [t.detach() for t in wn.tensors()]
            self.native_upper = (self._upper_slots is not None and ts[0].shape[0] <= int(lib.bhg_mwn_max_hidden()) and
                                 all(t.is_cuda and t.device == x.device and t.dtype == torch.float32 and t.is_contiguous() for t in ts))
        if self.native_upper:
            self._wn = ts
            _native.check(lib.bhg_mwn_forward(buf.ce.data_ptr(), B, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr(),
                                              ts[0].shape[0], None, buf.sd.data_ptr(), _stream()), "bhg_mwn_forward")
        else:
            self.sample_weight = spec.weight_fn(ce.detach().clone())
            buf.sd[:B].copy_(self.sample_weight.detach().reshape(-1).to(torch.float32) / B)
        if self.packed_prepare:
            _native.check(lib.bhg_mlp_backward_packed(ctypes.byref(d), buf.labels.data_ptr(), buf.fws.data_ptr(), buf.fws.numel(), _stream()),
                          "bhg_mlp_backward_packed")
            d.prepacked = 1   # the solves of this step find h_l, delta_l and the chain's weights packed
        else:
            _native.check(lib.bhg_mlp_backward(ctypes.byref(d), buf.labels.data_ptr(), _stream()), "bhg_mlp_backward")
        # HVP outputs are consumed by the recurrence kernel on the same stream before the next HVP is
        # launched, so ONE set of output tensors serves all K iterations — and all steps (no allocator
        # traffic per step, stable addresses for the recurrence kernel's pointer-table cache).
        if buf.out is None:
            buf.out = []
            for lin in spec.layers:
                buf.out += [torch.empty_like(lin.weight), torch.empty_like(lin.bias)]
            buf.out_tab, buf.out_keep = _native.ptr_array([t.data_ptr() for t in buf.out])
        self.out = buf.out
        self._out_tab = buf.out_tab

    def upper_vjp(self, coeff, upper, scale: float = 1.0, retain_graph: bool = False, with_flat: bool = False):
        """d(sum_i coeff_i s_i)/d(upper parameters), aligned with ``upper``: closed form (views of ONE fresh flat buffer in
        ``upper``'s order, times ``scale``) when the weight net is declared, else autograd through ``sample_weight``'s graph."""
        if not self.native_upper:
            sw = self.sample_weight
            out = list(torch.autograd.grad(sw, upper, grad_outputs=coeff.reshape(sw.shape), retain_graph=retain_graph, allow_unused=retain_graph))
            return (out, None) if with_flat else out
        ts, B = self._wn, self.B
        H = ts[0].shape[0]
        sizes = [p.numel() for p in upper]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=ts[0].device)   # fresh: .grad may keep its views
        views, off = [], 0
        for p, n in zip(upper, sizes):
            views.append(flat[off: off + n].view(p.shape))
            off += n
        g = [views[i] for i in self._upper_slots]   # (w1, b1, w2, b2) -> their slots in `upper`
        coeff = coeff if (coeff.dtype == torch.float32 and coeff.is_contiguous()) else coeff.to(torch.float32).contiguous()
        _native.check(
            self.lib.bhg_mwn_backward(self.buf.ce.data_ptr(), coeff.data_ptr(), B, ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(),
                                      ts[3].data_ptr(), H, float(scale), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                      g[3].data_ptr(), _stream()), "bhg_mwn_backward")
        return (views, flat) if with_flat else views

    # ---- per iteration -----------------------------------------------------------------------------------------------
    @staticmethod
    def _dir_table(direction_views):
        dirs = []
        for t in direction_views:
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0):
                t = t.detach().to(torch.float32).contiguous().clone()
            dirs.append(t)
        tab, keep = _native.ptr_array([t.data_ptr() for t in dirs])
        return tab, (dirs, keep)

    def hvp(self, direction_views):
        # CG / Neumann pass the SAME view tensors every iteration (views of the persistent flat direction):
        # validate and build the pointer table once per distinct tensor set (the cache holds the tensors, so
        # their ids cannot be recycled while it is alive).
        self._solve = None   # a hand-driven HVP overwrites the workspace an earlier fused solve's token refers to
        key = tuple(map(id, direction_views))
        cached = getattr(self, "_dir_cache", None)
        if cached is None or cached[0] != key:
            tab, keep = self._dir_table(direction_views)
            cached = (key, tab, keep, list(direction_views))
            # only a table that names the CALLER's tensors may be reused: a converted clone goes stale as soon as
            # the caller changes the direction in place
            self._dir_cache = cached if all(a is b for a, b in zip(keep[0], direction_views)) else None
        # the un-fused Neumann loop asks for the GEMM forms the fused Neumann solver uses (bitwise-equal arms)
        mode = 3 if getattr(getattr(self.spec.curr, "config", None), "type", None) == "neumann" else 0
        _native.check(self.lib.bhg_mlp_hvp_mode(ctypes.byref(self.desc), cached[1], self._out_tab, mode, _stream()), "bhg_mlp_hvp_mode")
        return self.out

    # ---- fused solvers (csrc/bhg_mlp.hip: bhg_mlp_cg_solve / bhg_mlp_neumann_solve) -----------------------------------
    def fused_supported(self, layout) -> bool:
        """True when the whole K loop can run natively with the recurrence fused into the HVP's output kernels:
        narrow classifier head, and ``layout`` is the flat layout of exactly [W1, b1, W2, b2, ...]."""
        if not self.lib.bhg_mlp_supports_fused_solve(ctypes.byref(self.desc)):
            return False
        want = []
        for W in self.Ws:
            want += [W.numel(), W.shape[0]]
        return tuple(want) == tuple(layout.numels) and str(layout.device) == str(self.Ws[0].device)

    def _fused_ws(self, device):
        buf = self.buf
        if getattr(buf, "fws", None) is None:
            n = int(self.lib.bhg_mlp_fused_ws_bytes(ctypes.byref(self.desc)))
            buf.fws = torch.zeros(max(n, 256), dtype=torch.uint8, device=device)
            # the solver's time-out word (bounded in-launch beta exchange): diagnosed by HipBackend.check_health
            addr = self.lib.bhg_mlp_timeout_flag_dev(ctypes.byref(self.desc), buf.fws.data_ptr())
            if addr:
                off = int(addr) - buf.fws.data_ptr()
                buf.fws_flag = buf.fws[off: off + 4].view(torch.int32)
                from ..backend import get_backend  # noqa: PLC0415

                get_backend().watch_fused_workspace(buf.fws_flag)
        return buf.fws

    def _fused_args(self, layout):
        fws = self._fused_ws(layout.device)
        starts = (ctypes.c_int64 * len(layout.starts))(*layout.starts)
        return fws, starts

    def cg_state_mask(self):
        """Which tensors' slices of r / p the coming solve (without a solution vector) reads or writes: None = all of them, else the
        bit mask of bhg_mlp_cg_state_mask — the fully projected solver keeps only the biases and the head weight N-sized."""
        mask = int(self.lib.bhg_mlp_cg_state_mask(ctypes.byref(self.desc), 0))
        return None if mask == (1 << 64) - 1 else mask

    def cg_solve(self, layout, x, r, p, K: int, cg_alpha: float, shift: float, keep_x: bool = True, rhs=None) -> FusedSolve:
        """cg.py:38-56 for this structure: K x (HVP chain with fused r/x update + direction update).
        keep_x=False: the library gets x = NULL and never reads or writes the solution vector.  Returns the token that
        lets mixed_coeff() work from the Rz(x) the solver accumulated.  rhs: the right-hand side's own tensors (what cg_init read),
        for a solve whose r / p were initialised through cg_state_mask()."""
        fws, starts = self._fused_args(layout)
        rhs_tab, _rhs_keep = (None, None) if rhs is None else _native.ptr_array([t.data_ptr() for t in rhs])
        _native.check(
            self.lib.bhg_mlp_cg_solve_rhs(ctypes.byref(self.desc), x.data_ptr() if keep_x else None, r.data_ptr(), p.data_ptr(), starts,
                                          layout.chunks_dev.data_ptr(), layout.n_chunks, int(K), float(cg_alpha), float(shift),
                                          layout.workspace.data_ptr(), fws.data_ptr(), fws.numel(), rhs_tab, _stream()),
            "bhg_mlp_cg_solve_rhs",
        )
        self._rhs_keep = (rhs, _rhs_keep)   # the solve is asynchronous: its first iteration reads these tensors
        # the solver accumulated Rz(x) on its way: mixed_coeff(solve=token) of exactly this solution needs no R-forward pass
        self._solve = FusedSolve("cg", float(cg_alpha), int(K), layout, materialised=keep_x)
        return self._solve

    # ---- global-batch CG (csrc/bhg_mlp.hip: bhg_mlp_cg_global_phase; driven by betty_amd/global_hvp.py) --------------------
    solution_free = True   # keep_x=False is honoured: the mixed coefficient then comes from the accumulated Rz(x)

    def cg_global_phase(self, layout, x, r, p, k: int, K: int, phase: int, world: int, php, cg_alpha: float, shift: float,
                        keep_x: bool = True) -> None:
        """One phase of one iteration of the global-batch CG solver on THIS rank's share of the batch (include/bhg.h);
        ``php`` is a one-element float64 device tensor the caller all-reduces between CHAIN and UPDATE."""
        fws, starts = self._fused_args(layout)
        if k == 0 and phase == _native.BHG_CG_GLOBAL_CHAIN:
            self._solve = None   # the workspace an earlier solve's token refers to is being rewritten
        _native.check(
            self.lib.bhg_mlp_cg_global_phase(ctypes.byref(self.desc), x.data_ptr() if keep_x else None, r.data_ptr(), p.data_ptr(),
                                             starts, layout.chunks_dev.data_ptr(), layout.n_chunks, int(k), int(K), int(phase),
                                             int(world), php.data_ptr(), float(cg_alpha), float(shift),
                                             layout.workspace.data_ptr(), fws.data_ptr(), fws.numel(), _stream()),
            "bhg_mlp_cg_global_phase",
        )

    def cg_global_finish(self, layout, K: int, cg_alpha: float, keep_x: bool = True) -> FusedSolve:
        """Token of the global solve that just ran its K iterations through cg_global_phase."""
        self._solve = FusedSolve("cg", float(cg_alpha), int(K), layout, materialised=keep_x)
        return self._solve

    # ---- global-batch CG, factor-exchange form (csrc/mlp/fx.inc: bhg_mlp_cg_fx_phase; driven by betty_amd/global_hvp.py) ---------------
    def fx_supported(self, layout, world: int) -> bool:
        """The fully projected solver on sample-partitioned data: needs the projected plan (>= 3 layers, widths % 32 == 0, narrow head)."""
        return bool(self.fused_supported(layout) and self.lib.bhg_mlp_fx_supported(ctypes.byref(self.desc), int(world)))

    def fx_buffers(self, world: int):
        """The three gathered buffers ([world][...]: constants of a solve, factors of an iteration, fp64 partials) and the solver's own
        workspace — per (buffers, world), allocated once."""
        buf = self.buf
        held = getattr(buf, "fx", None)
        if held is None:
            held = buf.fx = {}
        got = held.get(world)
        if got is None:
            d, lib, dev = ctypes.byref(self.desc), self.lib, buf.h[0].device
            z = lambda n, dt: torch.zeros(int(n), dtype=dt, device=dev)
            got = held[world] = {
                "const": z(world * lib.bhg_mlp_fx_const_floats(d), torch.float32).view(world, -1),
                "slab": z(world * lib.bhg_mlp_fx_slab_floats(d), torch.float32).view(world, -1),
                "scal": z(world * lib.bhg_mlp_fx_scal_doubles(d), torch.float64).view(world, -1),
                "xws": z(max(int(lib.bhg_mlp_fx_ws_bytes(d, world)), 256), torch.uint8),
            }
        return got

    def cg_fx_phase(self, rhs, k: int, K: int, phase: int, world: int, rank: int, cg_alpha: float, shift: float) -> None:
        """One phase of the factor-exchange solver on THIS rank's share of the batch (include/bhg.h); the caller all-gathers the buffer
        the phase wrote its slot of.  ``rhs``: the replicated right-hand side's tensors (read in CHAIN of iteration 0)."""
        b = self.fx_buffers(world)
        fws = self._fused_ws(self.buf.h[0].device)
        if phase == _native.BHG_CG_FX_BEGIN:
            self._solve = None   # the workspace an earlier solve's token refers to is being rewritten
            self._fx_rhs = _native.ptr_array([t.data_ptr() for t in rhs]) + (list(rhs),)
        _native.check(
            self.lib.bhg_mlp_cg_fx_phase(ctypes.byref(self.desc), self._fx_rhs[0], int(k), int(K), int(phase), int(world), int(rank),
                                         b["const"].data_ptr(), b["slab"].data_ptr(), b["scal"].data_ptr(), float(cg_alpha), float(shift),
                                         fws.data_ptr(), fws.numel(), b["xws"].data_ptr(), b["xws"].numel(), _stream()),
            "bhg_mlp_cg_fx_phase",
        )

    def neumann_fx_phase(self, rhs, k: int, K: int, phase: int, world: int, rank: int, alpha: float, shift: float) -> None:
        """The same for the Neumann series (bhg_mlp_neumann_fx_phase): CHAIN for k = 0 .. K (the last one is the closing half pass),
        GRAM for k = 0 .. K-1, END at k = K; the only buffer the caller gathers per iteration is the factor slab."""
        b = self.fx_buffers(world)
        fws = self._fused_ws(self.buf.h[0].device)
        if phase == _native.BHG_CG_FX_BEGIN:
            self._solve = None
            self._fx_rhs = _native.ptr_array([t.data_ptr() for t in rhs]) + (list(rhs),)
        _native.check(
            self.lib.bhg_mlp_neumann_fx_phase(ctypes.byref(self.desc), self._fx_rhs[0], int(k), int(K), int(phase), int(world), int(rank),
                                              b["const"].data_ptr(), b["slab"].data_ptr(), b["scal"].data_ptr(), float(alpha), float(shift),
                                              fws.data_ptr(), fws.numel(), b["xws"].data_ptr(), b["xws"].numel(), _stream()),
            "bhg_mlp_neumann_fx_phase",
        )

    def neumann_fx_finish(self, layout, K: int, alpha: float) -> FusedSolve:
        """Token of the factor-exchange Neumann solve: sum_{k <= K} Rz(v_k) sits where bhg_mlp_neumann_mixed_coeff reads it (projected)."""
        self._solve = FusedSolve("neumann", float(alpha), int(K), layout, v_last=None, materialised=False, projected=1)
        return self._solve

    def cg_fx_finish(self, layout, K: int, cg_alpha: float) -> FusedSolve:
        """Token of the factor-exchange solve that just ran (its Rz(x) sits where bhg_mlp_cg_mixed_coeff reads it; x never existed)."""
        self._solve = FusedSolve("cg", float(cg_alpha), int(K), layout, materialised=False)
        return self._solve

    def neumann_solve(self, layout, v0, v1, p, K: int, alpha: float, shift: float, keep_p: bool = True) -> FusedSolve:
        """neumann.py:61-66 for this structure: K HVP chains whose output kernels apply v' = v - a*Hv, p += v'.
        keep_p=False: the library gets p = NULL; mixed_coeff() of this solve is then formed from the Rz sums the head
        kernel collected plus one R-forward of the last direction."""
        fws, starts = self._fused_args(layout)
        projected = ctypes.c_int(0)
        _native.check(
            self.lib.bhg_mlp_neumann_solve(ctypes.byref(self.desc), v0.data_ptr(), v1.data_ptr(), p.data_ptr() if keep_p else None,
                                           starts, int(K), float(alpha), float(shift), fws.data_ptr(), fws.numel(), ctypes.byref(projected),
                                           _stream()),
            "bhg_mlp_neumann_solve",
        )
        # v_K sits in v0 after an even number of iterations, in v1 after an odd one
        self._solve = FusedSolve("neumann", float(alpha), int(K), layout, v_last=v0 if K % 2 == 0 else v1, materialised=keep_p)
        self._solve.projected = int(projected.value)   # which form ran: travels with the token to bhg_mlp_neumann_mixed_coeff
        return self._solve

    def mixed_coeff(self, dir_views, solve: FusedSolve = None):
        """c_i = (p_i - onehot_i) . Rz_i(direction) / B.
        ``solve`` = the token of the fused solve whose solution is meant: the coefficient then comes from the Rz the solver
        accumulated (no R-forward pass; the only way to ask about a solution that was never materialised).  Without a token
        ``dir_views`` are read like any direction — one R-forward, whatever memory they live in."""
        buf, B = self.buf, self.B
        if solve is not None:
            if solve is not getattr(self, "_solve", None):
                raise RuntimeError("stale fused-solve token: another solve or a hand-driven HVP has reused this state's workspace")
            if solve.kind == "neumann" and not solve.materialised:
                lay, v_last = solve.layout, solve.v_last
                if v_last is None:   # the factor-exchange solve: v_K never existed N-sized; its Rz is already in the sum (projected = 1)
                    tab, _keep = None, None
                else:
                    views = [v_last[s: s + n] for s, n in zip(lay.starts, lay.numels)]
                    tab, _keep = self._dir_table(views)
                _native.check(
                    self.lib.bhg_mlp_neumann_mixed_coeff(ctypes.byref(self.desc), tab, buf.labels.data_ptr(), buf.coeff.data_ptr(),
                                                         solve.alpha, solve.K, int(getattr(solve, "projected", 0)), buf.fws.data_ptr(),
                                                         buf.fws.numel(), _stream()),
                    "bhg_mlp_neumann_mixed_coeff",
                )
                return buf.coeff[:B] if self.native_upper else buf.coeff[:B].clone()   # (closed-form upper VJP: consumed on this stream at once)
            if solve.kind == "cg":
                _native.check(
                    self.lib.bhg_mlp_cg_mixed_coeff(ctypes.byref(self.desc), buf.labels.data_ptr(), buf.coeff.data_ptr(), solve.alpha,
                                                    buf.fws.data_ptr(), buf.fws.numel(), _stream()),
                    "bhg_mlp_cg_mixed_coeff",
                )
                return buf.coeff[:B] if self.native_upper else buf.coeff[:B].clone()   # (closed-form upper VJP: consumed on this stream at once)
            # a materialised Neumann accumulator: read it like any direction (below)
        tab, _keep = self._dir_table(dir_views)
        _native.check(
            self.lib.bhg_mlp_mixed_coeff(ctypes.byref(self.desc), tab, buf.labels.data_ptr(), buf.coeff.data_ptr(), _stream()),
            "bhg_mlp_mixed_coeff",
        )
        return buf.coeff[:B] if self.native_upper else buf.coeff[:B].clone()   # (closed-form upper VJP: consumed on this stream at once)


# ---- networks whose widths are not multiples of 32: the zero-padded twin (round 6) ---------------------------------------------------
# The reference's cg / neumann are shape-agnostic (betty/hypergradient/cg.py:8-70); the fused solvers' chain works on 32-wide tiles
# (csrc/bhg_mlp.hip: hoist_plan).  A network like 784-512-250-100-10 therefore gets a TWIN with its input and hidden widths rounded up to
# multiples of 32, zero-filled: a padded unit has zero weights and a zero bias, so its pre-activation is exactly 0, its ReLU mask 0, its
# activation, delta, Rh, Rd 0 — every inner product of the solve sees extra zeros and nothing else, and the twin's Hessian-vector
# products, step lengths and hypergradient are the network's own.  The twin's weights are refreshed from the real ones every step (one
# strided device copy per tensor, bhg_copy2d), right-hand sides are padded on the way in and results un-padded on the way out; the
# kernels never learn about it.
PAD_WIDTHS_TO_32 = True     # False: such networks keep the classic chain (the A/B arm of the tests)
_TWINS = weakref.WeakKeyDictionary()   # first nn.Linear of a network -> {(padded dims, device): _Twin}


def _pad32(n: int) -> int:
    return (int(n) + 31) // 32 * 32


def padded_dims(dims):
    """Input and hidden widths rounded up to 32; the class count stays (the head kernels take any count up to their own limit)."""
    return tuple(_pad32(d) for d in dims[:-1]) + (int(dims[-1]),)


class _TwinLayer:
    """What HipMLPState reads of an nn.Linear (weight, bias), on zero-padded storage; weak-referenceable (buffer maps are keyed by it)."""

    __slots__ = ("weight", "bias", "__weakref__")

    def __init__(self, out_f, in_f, device):
        self.weight = torch.zeros(out_f, in_f, dtype=torch.float32, device=device)
        self.bias = torch.zeros(out_f, dtype=torch.float32, device=device)


class _Twin:
    def __init__(self, dims, pdims, device):
        self.layers = [_TwinLayer(pdims[l + 1], pdims[l], device) for l in range(len(dims) - 1)]
        self.x = None            # [B, pdims[0]] zero-padded input batch
        self.dirs = [t for lay in self.layers for t in (torch.zeros_like(lay.weight), torch.zeros_like(lay.bias))]   # padded directions
        self.rhs = [torch.zeros_like(t) for t in self.dirs]                                                         # padded right-hand side


def _copy2d(lib, dst, ldd, src, lds, rows, cols):
    _native.check(lib.bhg_copy2d(dst.data_ptr(), int(ldd), src.data_ptr(), int(lds), int(rows), int(cols), _stream()), "bhg_copy2d")


class _TwinSpec:
    """The slice of WeightedCEMLP that HipMLPState reads, with the twin's layers in place of the real ones."""

    def __init__(self, spec, twin):
        self._spec, self.layers = spec, twin.layers

    def __getattr__(self, name):
        return getattr(self._spec, name)


class PaddedHipMLPState:
    """HipMLPState's interface for a network whose widths are not multiples of 32, served by a HipMLPState on its zero-padded twin."""

    solution_free = True

    def __init__(self, spec, x, y):
        if not x.is_cuda:
            raise _native.NativeLibraryError("WeightedCEMLP(impl='hip') needs CUDA/HIP tensors; there is no CPU fallback")
        self.lib = lib = _native.load()
        self.spec = spec
        real = [lin for lin in spec.layers]
        self.real_Ws = [lin.weight.detach() for lin in real]
        self.real_bs = [lin.bias.detach() for lin in real]
        for t in self.real_Ws + self.real_bs:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("weights and biases must be contiguous fp32 tensors")
        dims = tuple([self.real_Ws[0].shape[1]] + [W.shape[0] for W in self.real_Ws])
        self.dims, self.pdims = dims, padded_dims(dims)
        self.shapes = [s for W in self.real_Ws for s in ((W.shape[0], W.shape[1]), (1, W.shape[0]))]        # (rows, cols) per real tensor
        self.pcols = [c for l in range(len(dims) - 1) for c in (self.pdims[l], self.pdims[l + 1])]         # leading dimension of its twin
        owner = _TWINS.setdefault(spec.layers[0], {})
        key = (self.pdims, str(x.device))
        twin = owner.get(key)
        if twin is None:
            twin = owner[key] = _Twin(dims, self.pdims, x.device)
        self.twin = twin
        # refresh the twin from the real weights (the padding stays zero: only the real block is ever written)
        for l, (W, b) in enumerate(zip(self.real_Ws, self.real_bs)):
            _copy2d(lib, twin.layers[l].weight, self.pdims[l], W, dims[l], W.shape[0], W.shape[1])
            _copy2d(lib, twin.layers[l].bias, self.pdims[l + 1], b, dims[l + 1], 1, b.shape[0])
        B = x.shape[0]
        xs = x.detach().reshape(B, -1)
        xs = xs if (xs.dtype == torch.float32 and xs.is_contiguous()) else xs.to(torch.float32).contiguous()
        if twin.x is None or twin.x.shape[0] != B:
            twin.x = torch.zeros(B, self.pdims[0], dtype=torch.float32, device=x.device)
        _copy2d(lib, twin.x, self.pdims[0], xs, dims[0], B, dims[0])
        self.inner = HipMLPState(_TwinSpec(spec, twin), twin.x, y)
        self.B = B
        self.out = None

    # what structured.py reads off the state
    @property
    def native_upper(self):
        return self.inner.native_upper

    @property
    def sample_weight(self):
        return self.inner.sample_weight

    def upper_vjp(self, *a, **k):
        return self.inner.upper_vjp(*a, **k)

    # ---- padding / un-padding of tensor lists shaped like the REAL parameters ------------------------------------------------------
    def _pad(self, tensors, into):
        for t, dst, (rows, cols), ldd in zip(tensors, into, self.shapes, self.pcols):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                t = t.detach().to(torch.float32).contiguous()
            _copy2d(self.lib, dst, ldd, t, cols, rows, cols)
        return into

    def _unpad(self, padded, into):
        for src, t, (rows, cols), lds in zip(padded, into, self.shapes, self.pcols):
            _copy2d(self.lib, t, cols, src, lds, rows, cols)
        return into

    def _real_views(self, layout, flat):
        like = [t for W, b in zip(self.real_Ws, self.real_bs) for t in (W, b)]
        return layout.views(flat, like)

    def _twin_layout(self):
        from ..backend import get_backend  # noqa: PLC0415

        return get_backend().layout(self.twin.dirs)

    # ---- per iteration (un-fused arm, the structure guard) ---------------------------------------------------------------------------
    def hvp(self, direction_views):
        out_p = self.inner.hvp(self._pad(direction_views, self.twin.dirs))
        if self.out is None:
            self.out = [torch.empty_like(t) for W, b in zip(self.real_Ws, self.real_bs) for t in (W, b)]
        return self._unpad(out_p, self.out)

    # ---- fused solvers -----------------------------------------------------------------------------------------------------------------
    def fused_supported(self, layout) -> bool:
        want = [n for W in self.real_Ws for n in (W.numel(), W.shape[0])]
        return (tuple(want) == tuple(layout.numels) and str(layout.device) == str(self.real_Ws[0].device)
                and self.inner.fused_supported(self._twin_layout()))

    def cg_solve(self, layout, x, r, p, K: int, cg_alpha: float, shift: float, keep_x: bool = True, rhs=None) -> FusedSolve:
        """The caller's r holds the right-hand side (cg_init: r = p = vector); the solve runs on the twin's own flat state."""
        from ..backend import get_backend  # noqa: PLC0415

        be, play = get_backend(), self._twin_layout()
        self._pad(self._real_views(layout, r), self.twin.rhs)
        px, pr, pp = play.state(3)
        skip_x = not keep_x
        mask = self.inner.cg_state_mask() if skip_x else None
        ts = be.cg_init(play, self.twin.rhs, None if skip_x else px, pr, pp, keep_mask=mask) if mask is not None else \
            be.cg_init(play, self.twin.rhs, None if skip_x else px, pr, pp)
        token = self.inner.cg_solve(play, px, pr, pp, K, cg_alpha, shift, keep_x=keep_x, rhs=ts if mask is not None else None)
        if keep_x:
            self._unpad(play.views(px, self.twin.dirs), self._real_views(layout, x))
        return token

    def neumann_solve(self, layout, v0, v1, p, K: int, alpha: float, shift: float, keep_p: bool = True) -> FusedSolve:
        """The caller's v0 holds the right-hand side (neumann_init: v = vector, p = vector)."""
        from ..backend import get_backend  # noqa: PLC0415

        be, play = get_backend(), self._twin_layout()
        self._pad(self._real_views(layout, v0), self.twin.rhs)
        pv0, pv1, pp = play.state(3)
        be.neumann_init(play, self.twin.rhs, pv0, pp if keep_p else None)
        token = self.inner.neumann_solve(play, pv0, pv1, pp, K, alpha, shift, keep_p=keep_p)
        self._twin_p = (token, play, pp) if keep_p else None   # a materialised accumulator is read like a direction by mixed_coeff
        if keep_p:
            self._unpad(play.views(pp, self.twin.dirs), self._real_views(layout, p))
        return token

    def mixed_coeff(self, dir_views, solve: FusedSolve = None):
        if solve is not None:
            held = getattr(self, "_twin_p", None)
            if solve.kind == "neumann" and solve.materialised and held is not None and held[0] is solve:
                return self.inner.mixed_coeff(held[1].views(held[2], self.twin.dirs), solve)   # the twin's own accumulator
            return self.inner.mixed_coeff(None, solve)
        return self.inner.mixed_coeff(self._pad(dir_views, self.twin.dirs))


def make_state(spec, x, y):
    """HipMLPState, or PaddedHipMLPState when an input / hidden width is not a multiple of 32 and the padded twin takes the fused form.
    ``spec.pad_widths = False`` (set by the global-batch mode, whose ranks exchange the N-sized state of the REAL network between the
    phases of an iteration) keeps the network as it is."""
    Ws = [lin.weight for lin in spec.layers]
    dims = tuple([Ws[0].shape[1]] + [W.shape[0] for W in Ws])
    # (L >= 3 and a head the head kernels take — <= 256 classes: the conditions under which the twin takes the fused form at all)
    if (PAD_WIDTHS_TO_32 and getattr(spec, "pad_widths", True) and len(Ws) >= 3 and dims[-1] <= 256 and padded_dims(dims) != dims and x.is_cuda):
        return PaddedHipMLPState(spec, x, y)
    return HipMLPState(spec, x, y)
