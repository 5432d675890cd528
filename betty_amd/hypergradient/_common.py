"""Shared pieces of cg/neumann: inner-gradient graph, HVP providers, final mixed VJP."""
from __future__ import annotations

import contextlib
import os
import warnings
from typing import List, Sequence

import torch


def inner_gradient(curr):
    """``g = d L_in / d w`` with a graph (cg.py:27-32, neumann.py:31-36): re-evaluates the inner
    loss on the inner problem's last batch at the current inner weights."""
    in_loss = curr.training_step_exec(curr.cur_batch)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        in_grad = torch.autograd.grad(in_loss, curr.trainable_parameters(), create_graph=True)
    return in_grad


class AutogradHVP:
    """Default Hessian-vector product: PyTorch-ROCm double backward through the user's opaque
    ``training_step`` (cg.py:39-41, neumann.py:62)."""

    def __init__(self, in_grad, params):
        self.in_grad = in_grad
        self.params = list(params)

    def __call__(self, direction_views: Sequence[torch.Tensor]):
        return torch.autograd.grad(self.in_grad, self.params, grad_outputs=direction_views, retain_graph=True)


# ---- forward-over-reverse Hessian-vector product (opt-in, round 5) ------------------------------------------------------------
# H v = d/d eps [ grad_w L(w + eps v) ]: forward-mode tangents (torch.autograd.forward_ad) carried through ONE evaluation of the
# loss and its ordinary backward — the same quantity as the double backward of cg.py:39-41 / neumann.py:62, to rounding.  Why it
# exists: ATen's `_convolution_double_backward` handles a GROUPED convolution by a Python-visible loop over the groups — 16 to 64
# tiny convolutions per depthwise layer and term — so the double backward of a DARTS supernet (BASELINE cfg 5: the reference's
# `Network(16, 10, 8)`, eight cells x 14 edges x 4 separable / dilated ops) is ~300 k launches enqueued by the host:
# 18.4 s per product on the MI355X box against 1.9 s for the whole forward-over-reverse pass, whose derivative formulas are
# whole-tensor convolutions (profiles/r05_cfg5_hvp_conv_modes.txt).  K + 1 passes per solve: K products and the mixed second
# derivative d/d eps grad_lambda L(w + eps (-alpha x)); no `in_grad` graph is built at all.
# OPT-IN: `inner_problem.hypergradient_hvp = "forward_over_reverse"`.  The problem promises that
#   * its trainable parameters are exactly parameters of `problem.module` (they are swapped for dual tensors by name);
#   * training_step is a deterministic function of (parameters, batch) given the RNG state: the state at the start of the solve is
#     restored before every pass, so dropout draws the same masks K + 1 times (the reference's ONE graph has one set of masks);
#   * module buffers (batch-norm running statistics) may be updated in place: the first pass updates the real ones — once, as the
#     reference's single training_step call does — the others work on clones.
# Operators without a forward-mode formula raise NotImplementedError on the first pass: the caller falls back to the double backward.
def forward_over_reverse_wanted(curr) -> bool:
    return getattr(curr, "hypergradient_hvp", None) == "forward_over_reverse"


class ForwardOverReverseHVP:
    def __init__(self, curr, prev):
        import torch.autograd.forward_ad as fwAD  # noqa: PLC0415
        from torch.nn.utils import stateless  # noqa: PLC0415

        self.fwAD, self._swap = fwAD, stateless._reparametrize_module
        self.curr, self.prev = curr, prev
        self.module = curr.module
        self.params = list(curr.trainable_parameters())
        by_id = {id(p): n for n, p in self.module.named_parameters()}
        missing = [i for i, p in enumerate(self.params) if id(p) not in by_id]
        if missing:
            raise ValueError("hypergradient_hvp = 'forward_over_reverse': trainable parameters must be parameters of problem.module")
        self.names = [by_id[id(p)] for p in self.params]
        dev = self.params[0].device
        self._cuda = dev.type == "cuda"
        self._rng_cpu = torch.get_rng_state()
        self._rng_dev = torch.cuda.get_rng_state(dev) if self._cuda else None
        self._dev = dev
        self.passes = 0
        self.fallback = self.in_grad = None
        # the first pass updates module buffers in place (batch-norm statistics): kept so that a FAILED first pass can be undone before
        # the double backward runs training_step again — buffers advance once per solve either way
        self._buffers0 = {n: b.detach().clone() for n, b in self.module.named_buffers()}
        self._warn_train_mode_batchnorm()

    def _warn_train_mode_batchnorm(self):
        """on the dense-convolution ResNet-12 of BASELINE cfg 3 (train-mode batch norm) this method measured 0.57x the double
        backward's speed and a solve 4.4e-3 away from it (profiles/r05_opaque_product_vs_reference_on_gpu.txt) — it pays only where the
        double backward is host-bound (grouped convolutions: cfg 5, x 9.9, 8.7e-5).  Said loudly, once per problem, where the module
        normalises with batch statistics; `problem.hypergradient_hvp_ack_batchnorm = True` acknowledges it."""
        bn = [n for n, m in self.module.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training]
        if not bn or getattr(self.curr, "hypergradient_hvp_ack_batchnorm", False) or getattr(self.curr, "_bhg_for_bn_warned", False):
            return
        grouped = any(isinstance(m, torch.nn.modules.conv._ConvNd) and m.groups > 1 for m in self.module.modules())
        warnings.warn(
            f"betty_amd: hypergradient_hvp = 'forward_over_reverse' on a module with {len(bn)} train-mode batch-norm layer(s) "
            f"(first: {bn[0]!r}){'' if grouped else ' and NO grouped convolution'}: forward-mode tangents through batch statistics measured "
            "4.4e-3 from the double backward's solve on a dense-convolution ResNet-12 at 0.57x its speed; the method pays where the "
            "double backward is host-bound (grouped / depthwise convolutions).  Compare both on your problem; set "
            "problem.hypergradient_hvp_ack_batchnorm = True to silence this.", RuntimeWarning, stacklevel=3)
        try:
            self.curr._bhg_for_bn_warned = True
        except AttributeError:
            pass

    @staticmethod
    def _is_missing_forward_formula(exc) -> bool:
        """Only a missing forward-mode derivative may trigger the fall-back — not an out-of-memory error or a shape bug in the user's
        training_step, which must surface as they are."""
        if isinstance(exc, NotImplementedError):
            return True
        if isinstance(exc, torch.cuda.OutOfMemoryError):
            return False
        msg = str(exc).lower()
        return any(k in msg for k in ("forward ad", "forward-mode", "forward mode", "jvp", "dual level", "dual tensor", "fwad"))

    def _pass(self, tangents, wrt_upper: bool):
        fwAD = self.fwAD
        if self.passes > 0:   # the same random draws as the first pass
            torch.set_rng_state(self._rng_cpu)
            if self._cuda:
                torch.cuda.set_rng_state(self._rng_dev, self._dev)
        swap = {}
        if self.passes > 0:   # running statistics were updated by the first pass: later passes must not move them again
            swap.update({n: b.detach().clone() for n, b in self.module.named_buffers()})
        self.passes += 1
        with fwAD.dual_level():
            duals = [fwAD.make_dual(p.detach().requires_grad_(True), t.detach().reshape(p.shape)) for p, t in zip(self.params, tangents)]
            swap.update(dict(zip(self.names, duals)))
            with self._swap(self.module, swap):
                loss = self.curr.training_step_exec(self.curr.cur_batch)
            wrt = list(self.prev.trainable_parameters()) if wrt_upper else duals
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                grads = torch.autograd.grad(loss, wrt, allow_unused=wrt_upper)
            out = []
            for g, w in zip(grads, wrt):
                t = None if g is None else fwAD.unpack_dual(g).tangent
                out.append(torch.zeros_like(w) if t is None else t.detach())
        return out

    def __call__(self, direction_views: Sequence[torch.Tensor]):
        if self.fallback is not None:
            return self.fallback(direction_views)
        try:
            return self._pass(direction_views, wrt_upper=False)
        except (NotImplementedError, RuntimeError) as exc:
            if self.passes > 1 or not self._is_missing_forward_formula(exc):
                raise
            # undo what the failed pass did to the module's buffers: inner_gradient() below runs training_step once more, and the
            # statistics must have advanced exactly once when the solve returns
            with torch.no_grad():
                for n, b in self.module.named_buffers():
                    if n in self._buffers0 and b.shape == self._buffers0[n].shape:
                        b.copy_(self._buffers0[n])
            # an operator of this training_step has no forward-mode formula: the reference's double backward takes over for the solve
            warnings.warn(f"betty_amd: forward-over-reverse HVP not available for this training_step ({type(exc).__name__}: {exc}); "
                          "using the double backward", RuntimeWarning)
            torch.set_rng_state(self._rng_cpu)
            if self._cuda:
                torch.cuda.set_rng_state(self._rng_dev, self._dev)
            self.in_grad = inner_gradient(self.curr)
            self.fallback = AutogradHVP(self.in_grad, self.params)
            return self.fallback(direction_views)

    def mixed(self, neg_x_views, sync: bool):
        """cg.py:58-68 / neumann.py:44-54: d(g . neg_x)/d lambda = tangent of grad_lambda L in direction neg_x on the inner weights."""
        if self.fallback is not None:
            return mixed_vjp(self.in_grad, self.prev, neg_x_views, sync)
        grads = self._pass(neg_x_views, wrt_upper=True)
        upper = list(self.prev.trainable_parameters())
        if sync:
            # accumulate THROUGH autograd (AccumulateGrad nodes), like the reference's backward(..., inputs=upper): hooks fire
            torch.autograd.backward(upper, grad_tensors=grads)
            return None
        return grads


# ---- hipGraph replay of an opaque Hessian-vector product ----------------------------------------------------------------
# The double backward of a user's training_step is hundreds to thousands of small ATen launches (cfg 2's MLP: ~1.1 ms per
# HVP, launch-bound): the K HVPs of one solve run the SAME launch sequence on the SAME addresses — the autograd graph of
# `in_grad` is fixed for the step and the direction lives in one persistent flat vector — so the sequence is captured once
# (HIP graph) and replayed K - 2 times.
#   call 1: eager (settles library handles, workspaces, autotuning on the capture stream)
#   call 2: captured (hipStreamBeginCapture .. EndCapture around torch.autograd.grad), then replayed
#   call 3..K: replayed
# Autograd runs a backward node on the stream its forward ran on, so the graph of `in_grad` must be BUILT on the stream
# that later captures: `solve_stream` moves the whole solve to a library-owned side stream when the caller sits on the
# default (legacy) stream, which HIP cannot capture.  A capture that raises (a host sync inside the double backward, ...)
# leaves the wrapper eager for the rest of the solve.
# OPT-IN (measured on the MI355X, round 3): cfg 2 with the opaque HVP goes from 43.9 to 72.4 steps/s and seven solves of
# seven captures each replay cleanly, a ResNet-12 (GPU-bound double backward) gains nothing — and hipStreamEndCapture
# SEGFAULTS inside ROCm 7.2 on the tiny logistic-regression graph of the reference's regression scenario, which no
# try / except can contain.  So nothing is captured unless the user says so: `curr.hypergradient_graph = True` on the inner
# problem, or BHG_HVP_GRAPH=1 in the environment (=0 forces it off).
_SOLVE_STREAMS = {}
GRAPH_STATS = {"captures": 0, "replays": 0, "fallbacks": 0}


def hvp_graph_wanted(K: int, tensors, curr=None) -> bool:
    mode = os.environ.get("BHG_HVP_GRAPH", "")
    if mode == "0" or K < 3 or not tensors or not tensors[0].is_cuda:
        return False
    return mode == "1" or bool(getattr(curr, "hypergradient_graph", False))


@contextlib.contextmanager
def solve_stream(device, enabled: bool):
    """Run the body on a capturable stream (see above); joins the caller's stream on both sides."""
    if not enabled:
        yield None
        return
    cur = torch.cuda.current_stream(device)
    if cur != torch.cuda.default_stream(device):
        yield cur       # the caller already works on a stream of its own
        return
    side = _SOLVE_STREAMS.get(device.index)
    if side is None:
        side = _SOLVE_STREAMS[device.index] = torch.cuda.Stream(device)
    side.wait_stream(cur)
    try:
        with torch.cuda.stream(side):
            yield side
    finally:
        cur.wait_stream(side)


class GraphedHVP:
    """Wraps an eager HVP callable ``fn(direction_views) -> tuple of tensors`` (pure device work on fixed addresses)."""

    def __init__(self, fn):
        self.fn, self.calls, self.graph, self.out, self.key, self.dead = fn, 0, None, None, None, False

    @staticmethod
    def _key(views):
        return tuple((v.data_ptr(), tuple(v.shape), v.dtype) for v in views)

    def __call__(self, views):
        self.calls += 1
        if self.dead:
            return self.fn(views)
        key = self._key(views)
        if self.graph is not None:
            if key == self.key:
                self.graph.replay()
                GRAPH_STATS["replays"] += 1
                return self.out
            self.dead = True            # the caller moved the direction: the captured addresses are stale
            return self.fn(views)
        dev = views[0].device
        if self.calls == 1 or torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
            self.key = key
            return self.fn(views)
        if key != self.key:
            self.dead = True
            return self.fn(views)
        graph = torch.cuda.CUDAGraph()
        try:
            # a private memory pool per graph (released with the graph at the end of the solve): a pool handle shared
            # across solves is dropped by the caching allocator when the previous graph dies (round-3 measurement: the second
            # capture then trips an internal assert of HIPCachingAllocator)
            graph.capture_begin(capture_error_mode="thread_local")
            try:
                out = self.fn(views)
            finally:
                graph.capture_end()
            graph.replay()
        except Exception as exc:   # not capturable on this stack: eager for the rest of the solve
            self.dead = True
            GRAPH_STATS["fallbacks"] += 1
            warnings.warn(f"betty_amd: hipGraph capture of the Hessian-vector product failed ({type(exc).__name__}: {exc}); "
                          "continuing with eager launches", RuntimeWarning)
            torch.cuda.synchronize(dev)
            return self.fn(views)
        GRAPH_STATS["captures"] += 1
        GRAPH_STATS["replays"] += 1
        self.graph, self.out = graph, tuple(out)
        return self.out


class PersistentOpaqueGraphs:
    """Opt-in `inner_problem.hypergradient_graph = "persistent"`: the inner problem promises that its ``training_step`` is a
    STATIC function of (parameters, upper parameters, batch) — same shapes, no data-dependent Python control flow, no host
    side effects that matter.  Then not only the K HVPs of one solve but the steps of a whole run share two HIP graphs:

        G1   loss = training_step(batch*) ; in_grad = autograd.grad(loss, params, create_graph=True)   (batch* = static copies)
        G2   hvp  = autograd.grad(in_grad, params, grad_outputs=direction views, retain_graph=True)

    A step is then: copy the batch into batch*, replay G1 (the autograd graph OBJECT of `in_grad` is the one built at capture
    time; its saved tensors live at fixed addresses inside G1's pool and are refreshed by the replay), K x replay G2, and the
    usual EAGER mixed second derivative through that autograd graph (with retain_graph=True, so the graph survives the step;
    DistributedDataParallel hooks fire as always).  First step of a signature: eager (warm-up); second: capture; from the
    third on: replay.  Anything that changes the signature (parameter storage, batch shapes, direction buffer) recaptures.
    Not taken under DistributedDataParallel-wrapped upper modules reached inside the capture (their forward does host
    bookkeeping).  Measured (MI355X, cfg 2, opaque HVP, CG K = 20): see DESIGN.md section 5.

    STATIC-CLOSURE REQUIREMENT: the eager backward through the captured autograd graph is made to pass autograd's in-place check
    by setting the version counters of the TRACKED tensors (parameters, buffers, the static batch copies) back to their
    capture-time values (`saved_versions`, through the private `torch._C._autograd._unsafe_set_version_counter`).  Any other
    tensor `training_step` closes over and that is saved for backward is either caught by that check (an error) or, if it is
    rewritten in place between steps without autograd noticing, silently stale: a `training_step` that uses such tensors must
    not opt in.  `persistent_graphs_for` refuses (returns None) when the private API is absent."""

    def __init__(self):
        self.sig = None
        self.state = 0          # 0: nothing seen | 1: warmed up eagerly | 2: captured | -1: gave up
        self.g1 = self.g2 = None
        self.static_leaves = self.spec = None
        self.in_grad = self.out = None
        self.tracked, self.versions = [], []

    @contextlib.contextmanager
    def saved_versions(self):
        """The eager backward through the CAPTURED autograd graph checks that the tensors it saved (weights, upper weights,
        the static batch) still carry the version numbers of capture time — but optimizers and the batch refresh have written
        to them in place since (that is the point: same storage, new values, refreshed by the replay of G1).  Around that one
        backward the version counters are set to their capture-time values and restored afterwards, so every OTHER autograd
        graph that holds these tensors (the upper loss of the step, say) sees the numbers it expects."""
        if self.state != 2 or not self.tracked:
            yield
            return
        now = [t._version for t in self.tracked]
        torch._C._autograd._unsafe_set_version_counter(self.tracked, self.versions)
        try:
            yield
        finally:
            torch._C._autograd._unsafe_set_version_counter(self.tracked, now)

    @staticmethod
    def _signature(curr, params, batch_leaves, views):
        leaves = tuple((tuple(t.shape), t.dtype, str(t.device)) if torch.is_tensor(t) else ("leaf", repr(t)) for t in batch_leaves)
        return (tuple((id(p), p.data_ptr(), tuple(p.shape)) for p in params), leaves, GraphedHVP._key(views))

    def begin_step(self, curr, params, views, prev=None):
        """-> (in_grad, hvp_fn, persistent): `persistent` tells mixed_vjp to keep the autograd graph alive."""
        from torch.utils import _pytree as pytree

        leaves, spec = pytree.tree_flatten(curr.cur_batch)
        sig = self._signature(curr, params, leaves, views)
        dev = views[0].device
        on_default = torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev)
        if self.state == -1 or on_default:
            in_grad = inner_gradient(curr)
            return in_grad, AutogradHVP(in_grad, params), False
        if sig != self.sig or self.state == 0:      # new signature: an eager step first (warm-up on the capture stream)
            self.sig, self.state = sig, 1
            self.g1 = self.g2 = self.in_grad = self.out = None
            in_grad = inner_gradient(curr)
            return in_grad, AutogradHVP(in_grad, params), False
        if self.state == 1:                         # second step of this signature: capture G1 and G2
            try:
                self.static_leaves = [t.clone() if torch.is_tensor(t) else t for t in leaves]
                self.spec = spec
                static_batch = pytree.tree_unflatten(self.static_leaves, spec)
                g1 = torch.cuda.CUDAGraph()
                g1.capture_begin(capture_error_mode="thread_local")
                try:
                    loss = curr.training_step_exec(static_batch)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        in_grad = torch.autograd.grad(loss, curr.trainable_parameters(), create_graph=True)
                finally:
                    g1.capture_end()
                g1.replay()
                g2 = torch.cuda.CUDAGraph()
                g2.capture_begin(capture_error_mode="thread_local")
                try:
                    out = torch.autograd.grad(in_grad, list(params), grad_outputs=views, retain_graph=True)
                finally:
                    g2.capture_end()
                self.g1, self.g2, self.in_grad, self.out, self.state = g1, g2, in_grad, tuple(out), 2
                tracked = list(params) + [t for t in self.static_leaves if torch.is_tensor(t)]
                if prev is not None:
                    mod = getattr(prev, "module", None)
                    tracked += list(prev.trainable_parameters()) + (list(mod.buffers()) if mod is not None else [])
                cmod = getattr(curr, "module", None)
                tracked += list(cmod.buffers()) if cmod is not None else []
                seen, uniq = set(), []
                for t in tracked:
                    if id(t) not in seen:
                        seen.add(id(t))
                        uniq.append(t)
                self.tracked, self.versions = uniq, [t._version for t in uniq]
                GRAPH_STATS["captures"] += 2
            except Exception as exc:
                self.state = -1
                GRAPH_STATS["fallbacks"] += 1
                warnings.warn(f"betty_amd: persistent hipGraph capture of the inner gradient / HVP failed ({type(exc).__name__}: {exc}); "
                              "continuing with eager launches", RuntimeWarning)
                torch.cuda.synchronize(dev)
                in_grad = inner_gradient(curr)
                return in_grad, AutogradHVP(in_grad, params), False
        else:                                       # replay: refresh the static batch, recompute loss / in_grad in place
            for st, t in zip(self.static_leaves, leaves):
                if torch.is_tensor(st):
                    st.copy_(t)
            self.g1.replay()
            GRAPH_STATS["replays"] += 1

        def hvp(direction_views):
            if GraphedHVP._key(direction_views) != self.sig[2]:
                raise RuntimeError("persistent HVP graph: the direction moved inside a solve")
            self.g2.replay()
            GRAPH_STATS["replays"] += 1
            return self.out

        return self.in_grad, hvp, True


def _persistent_mixed(self, prev, neg_x_views, sync: bool):
    """Final hop of a step whose `in_grad` lives in the persistent graph G1: the mixed second derivative (cg.py:58-68) as a THIRD
    captured graph, G3 = autograd.grad(in_grad, upper, grad_outputs=views) — captured on the first replayed step, replayed
    afterwards.  `sync=True` accumulates the replayed result into `.grad` (the upper modules are not DDP-wrapped here —
    persistent graphs are not taken under DDP — so there is no reducer hook a `backward` would have to fire)."""
    upper = list(prev.trainable_parameters())
    key = (tuple((id(p), p.data_ptr()) for p in upper), GraphedHVP._key(neg_x_views))
    if getattr(self, "g3_dead", False):   # a capture of this hop failed once: stay eager for good (no retry, no warning per step)
        return mixed_vjp(self.in_grad, prev, neg_x_views, sync, retain_graph=True)
    if getattr(self, "g3", None) is None or self.g3_key != key:
        with self.saved_versions():
            try:
                g3 = torch.cuda.CUDAGraph()
                g3.capture_begin(capture_error_mode="thread_local")
                try:
                    outs = torch.autograd.grad(self.in_grad, upper, grad_outputs=neg_x_views, retain_graph=True)
                finally:
                    g3.capture_end()
                self.g3, self.g3_key, self.g3_out = g3, key, tuple(outs)
                GRAPH_STATS["captures"] += 1
            except Exception as exc:   # stay eager for this hop
                self.g3 = None
                self.g3_dead = True
                warnings.warn(f"betty_amd: hipGraph capture of the mixed second derivative failed ({type(exc).__name__}: {exc}); "
                              "this hop stays eager for the rest of the run", RuntimeWarning)
                torch.cuda.synchronize(neg_x_views[0].device)
                return mixed_vjp(self.in_grad, prev, neg_x_views, sync, retain_graph=True)
    self.g3.replay()
    GRAPH_STATS["replays"] += 1
    grads = [g.clone() for g in self.g3_out]
    if sync:
        for p, g in zip(upper, grads):
            p.grad = g if p.grad is None else p.grad + g
        return None
    return grads


PersistentOpaqueGraphs.mixed = _persistent_mixed


def _uses_ddp(problem) -> bool:
    from torch.nn.parallel import DistributedDataParallel as DDP

    return any(isinstance(getattr(problem, name, None), DDP) for name in ("fwd", "module"))


def persistent_graphs_for(curr, K: int, tensors, prev=None):
    """The inner problem's PersistentOpaqueGraphs when it opted in with hypergradient_graph = "persistent", else None."""
    if os.environ.get("BHG_HVP_GRAPH", "") == "0" or K < 1 or not tensors or not tensors[0].is_cuda:
        return None
    if _uses_ddp(curr) or (prev is not None and _uses_ddp(prev)):   # a DDP forward does host bookkeeping: not capturable
        return None
    if getattr(curr, "hypergradient_graph", False) != "persistent":
        return None
    if not hasattr(getattr(torch._C, "_autograd", None), "_unsafe_set_version_counter"):   # the private API saved_versions() needs
        warnings.warn("betty_amd: this PyTorch has no torch._C._autograd._unsafe_set_version_counter — persistent HVP graphs are off",
                      RuntimeWarning)
        return None
    cache = getattr(curr, "_bhg_persistent_graphs", None)
    if cache is None:
        cache = curr._bhg_persistent_graphs = PersistentOpaqueGraphs()
    return cache


def mixed_vjp(in_grad, prev, neg_x_views: List[torch.Tensor], sync: bool, retain_graph: bool = False):
    """Final hop to the upper parameters (cg.py:58-68, neumann.py:44-54).

    ``neg_x_views`` already holds ``-(alpha * x)``; by linearity of the VJP
    ``-(d(g.x)/d lambda) == d(g.(-x))/d lambda`` bit for bit, so no extra negation pass is needed.
    ``sync=True`` accumulates into ``.grad`` through ``backward`` (DDP reducer hooks fire) and
    returns None; ``sync=False`` returns the list for ``Problem.set_grads``."""
    upper = prev.trainable_parameters()
    if sync:
        torch.autograd.backward(in_grad, inputs=upper, grad_tensors=neg_x_views, retain_graph=retain_graph)
        return None
    return list(torch.autograd.grad(in_grad, upper, grad_outputs=neg_x_views, retain_graph=retain_graph))
