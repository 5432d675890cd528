"""Opt-in analytic Hessian-vector products for structured inner problems.

The reference always differentiates twice through the user's opaque ``training_step``
(cg.py:39-41, neumann.py:62).  When the inner problem *declares* a structure this backend knows,
the HVP and the mixed second derivative are computed in closed form instead (SURVEY.md
Appendix A), with everything that does not depend on the direction vector cached across the K
iterations of one hypergradient step.  A problem opts in by exposing::

    def hypergradient_structure(self, prev):     # on the INNER problem
        return WeightedCEMLP(self, prev, layers=[...], weight_fn=..., ridge=...)

Problems that do not are handled by autograd exactly as in the reference.

Protocol (what cg/neumann call)::

    hvp_fn = provider.prepare()        # once per hypergradient step
    hvp    = hvp_fn(direction_views)   # K times; list aligned with curr.parameters()
    out    = provider.mixed_vjp(neg_x_views, sync)   # final hop to prev's parameters

Optional "one pass" extension: ``token = provider.fused_cg(layout, x, r, p, K, alpha)`` (``fused_neumann`` alike) runs the
whole K loop natively and returns a truthy value — ``True``, or a token object that cg/neumann hand back as
``provider.mixed_vjp(views, sync, solve=token)`` so the provider knows the views are the solution of exactly that run
(a provider that never materialises the solution must be told; it cannot read the views).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import warnings

import torch
import torch.nn.functional as F


class StructureMismatchError(RuntimeError):
    """The declared closed form does not describe the problem's actual ``training_step``."""


# A declared structure is a promise about the user's own loss; a broken promise (label smoothing, dropout, another reduction, a
# layer missing from `layers`) would silently give wrong hypergradients.  So the FIRST time a provider is prepared for a problem
# (per parameter shapes / batch size / ridge) one analytic Hessian-vector product and one mixed second derivative are compared
# with autograd's double backward through the problem's real training_step on a random direction, and a mismatch raises.  One
# double backward, once per problem; ``verify=False`` on the provider or VERIFY_STRUCTURE = False opts out (production runs that
# have been checked once).
VERIFY_STRUCTURE = True
VERIFY_RTOL = 1e-3
# ... unless the instance sits ON a ReLU kink: a hidden pre-activation within fp32 summation noise of zero (relative distance below
# VERIFY_KINK_MARGIN of the layer's mean magnitude) gets its mask from the summation ORDER, and two correct fp32 evaluations — ATen's
# GEMM, the split-K kernels, the packed kernels — disagree on it; one flipped unit moves the Hessian-vector product by a few 1e-3
# (seed 4 of the metric workload: margin 2.4e-7, the reference's own fp32 and fp64 runs differ by 5e-3 at every K).
# The check still separates a wrong declaration (label smoothing, dropout, another reduction: 1e-1 .. 1e+2) from a right one there.
VERIFY_KINK_MARGIN = 1e-6
VERIFY_RTOL_ON_A_KINK = 3e-2
VERIFY_KINK_RETRIES = 3   # relaxed passes (one per prepare) before a kink-ridden problem's verdict is cached after all


# betty_amd.install(auto_structure=True): an inner problem WITHOUT a declaration is looked at once — a Linear / ReLU stack with a
# per-sample-weighted cross-entropy is the shape of the reference's data-reweighting example (examples/learning_to_reweight/
# main.py:117-127), whose training_step is opaque to the reference (betty/problems/problem.py:327-332) — and, when the closed form of
# WeightedCEMLP survives the SAME check a declaration gets (one double backward through the problem's real training_step), the
# fused solver takes it; anything else stays on the opaque path, silently.  Off by default: the look costs a double backward at the
# first hypergradient of every inner problem, and a structure is a promise the user has not made.
AUTO_STRUCTURE = False
AUTO_IMPL = None          # TEST HOOK ONLY: "torch" lets the CPU suite exercise the recognition with the ATen closed form
AUTO_STATS = {"looked": 0, "accepted": 0, "rejected": 0}


def structured_hvp_for(curr, prev):
    hook = getattr(curr, "hypergradient_structure", None)
    if hook is not None:
        return hook(prev)
    if AUTO_STRUCTURE:
        return _auto_structure(curr, prev)
    return None


def _linear_stack(module):
    """The nn.Linear layers of ``module`` in registration order when they carry ALL of its parameters, chain (out_l = in_{l+1}) and all
    have a bias; else None.  (What sits between them is not visible from here: the guard decides.)"""
    if module is None:
        return None
    inner = getattr(module, "module", None)
    if isinstance(module, torch.nn.parallel.DistributedDataParallel) and inner is not None:
        module = inner
    layers = [m for m in module.modules() if isinstance(m, torch.nn.Linear)]
    if not layers or any(lin.bias is None for lin in layers):
        return None
    owned = {id(t) for lin in layers for t in (lin.weight, lin.bias)}
    if any(id(p) not in owned for p in module.parameters()) or len(owned) != 2 * len(layers):
        return None
    if any(a.out_features != b.in_features for a, b in zip(layers[:-1], layers[1:])):
        return None
    for m in module.modules():   # randomness / batch statistics inside the step: not this closed form
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training:
            return None
        if isinstance(m, torch.nn.modules.dropout._DropoutNd) and m.training and m.p > 0:
            return None
    return layers


def _auto_structure(curr, prev):
    """None, or a WeightedCEMLP for this (curr, prev) whose closed form was checked against the problem's own training_step.  The
    verdict is cached on the problem (keyed by the parameter shapes and the batch size)."""
    if len(getattr(curr, "paths", []) or []) > 0:
        return None
    store = curr.__dict__ if hasattr(curr, "__dict__") else None
    batch = getattr(curr, "cur_batch", None)
    if store is None or not (isinstance(batch, (tuple, list)) and len(batch) == 2 and all(torch.is_tensor(t) for t in batch)):
        return None
    x, y = batch
    params = list(curr.trainable_parameters()) if hasattr(curr, "trainable_parameters") else list(curr.parameters())
    key = (tuple(tuple(p.shape) for p in params), int(x.shape[0]))
    cache = store.setdefault("_bhg_auto_structure", {})

    def build(verdict, verify):
        layers = verdict["layers"]
        fwd = getattr(prev, "fwd", None) or prev.module
        wn = SigmoidMLPWeightNet(*verdict["weight_net"]) if verdict["weight_net"] is not None else None
        return WeightedCEMLP(curr, prev, layers=layers, weight_fn=lambda ce: fwd(ce.reshape(-1, 1)), ridge=verdict["ridge"],
                             impl=AUTO_IMPL, verify=verify, weight_net=wn)

    if key in cache:
        return build(cache[key], False) if cache[key] is not None else None
    AUTO_STATS["looked"] += 1
    cache[key] = None
    layers = _linear_stack(getattr(curr, "module", None))
    ok = (layers is not None and len(params) == 2 * len(layers)
          and all(a is b for a, b in zip(params, [t for lin in layers for t in (lin.weight, lin.bias)]))
          and y.dtype in (torch.int64, torch.int32) and y.dim() == 1 and x.shape[0] == y.shape[0] and x.is_floating_point()
          and x[0].numel() == layers[0].in_features and (AUTO_IMPL == "torch" or (x.is_cuda and params[0].dtype == torch.float32)))
    if ok:
        upper_layers = _linear_stack(getattr(prev, "module", None))
        wn = None
        if (upper_layers is not None and len(upper_layers) == 2 and upper_layers[0].in_features == 1 and upper_layers[1].out_features == 1):
            wn = (upper_layers[0], upper_layers[1])
        for cand in ([wn, None] if wn is not None else [None]):
            verdict = {"layers": layers, "weight_net": cand, "ridge": 0.0}
            try:
                # the one quantity a declaration states that cannot be read off the modules: `ridge * sum(w^2)` in the loss shows up as
                # 2 ridge v in H v — estimated from one product, then the estimate goes through the ordinary guard like a declared value
                probe = build(verdict, False)
                probe.prepare()
                verdict["ridge"] = probe._estimate_ridge()
                build(verdict, True).prepare()           # raises StructureMismatchError when the closed form does not describe the step
            except (StructureMismatchError, ValueError, NotImplementedError):
                continue
            cache[key] = verdict
            break
    if cache[key] is None:
        AUTO_STATS["rejected"] += 1
        return None
    AUTO_STATS["accepted"] += 1
    return build(cache[key], False)


class SigmoidMLPWeightNet:
    """Declared structure of the sample-weight function of :class:`WeightedCEMLP`: the meta-weight-net of data reweighting
    (examples/learning_to_reweight/model.py:98-111, ``MLP(hidden_size, num_layers=1)``; called at main.py:123-125)::

        s = sigmoid(out(relu(hidden(ce.reshape(-1, 1)))))          hidden: nn.Linear(1, H), out: nn.Linear(H, 1)

    With it the HIP implementation evaluates the weights and their VJP to (hidden, out)'s parameters in closed form
    (csrc/bhg_mwn.hip: one launch each way) instead of ~15 ATen launches through autograd per step — five rocBLAS GEMMs for a
    100 x 100 problem.  It is a promise like the inner structure and is checked the same way (the first prepare() compares the
    mixed second derivative with autograd through the problem's real ``training_step``).  ``weight_fn`` stays the definition:
    it is what runs whenever the closed form does not apply (``impl="torch"``, a CPU tensor, H > 2048, upper parameters
    that are not exactly these four tensors).

    ``average_over``: the reference's ``sync=True`` hop accumulates through autograd so that a DistributedDataParallel wrapper
    of the upper module all-reduces (mean) the hypergradient (betty/problems/problem.py:220-224, cg.py:58-63).  The closed form
    does not pass through that wrapper, so a data-parallel caller says so here: ``True`` (the default process group) or a process
    group -> the M-sized flat result is all-reduced (RCCL) and averaged before it is accumulated into ``.grad``; ``None`` (the
    default) -> what the reference would have done: when the upper problem's forward module (``prev.fwd`` / ``prev.module``) IS a
    DistributedDataParallel wrapper, the mean over that wrapper's process group (its reducer's job in the reference; ADVICE r5:
    never a silently rank-local gradient under DDP), otherwise no collective, as for an un-wrapped module in the reference.

    ``overlap=True`` (with a mean to take): the all-reduce is issued asynchronously and the compute stream is NOT made to wait for it
    at once — the next step's solve runs under it; see betty_amd/distributed.py (``fence_grads``) for where the fence falls and the
    contract (nobody reads these ``.grad`` tensors before the upper optimizer's step, the next hop, or an explicit fence).
    """

    def __init__(self, hidden: torch.nn.Linear, out: torch.nn.Linear, average_over=None, overlap: bool = False):
        if hidden.in_features != 1 or out.out_features != 1 or hidden.out_features != out.in_features:
            raise ValueError("SigmoidMLPWeightNet: hidden must be Linear(1, H) and out Linear(H, 1)")
        if hidden.bias is None or out.bias is None:
            raise ValueError("SigmoidMLPWeightNet: both layers carry a bias in the reference's meta-weight-net")
        self.hidden, self.out, self.average_over, self.overlap = hidden, out, average_over, bool(overlap)

    def tensors(self):
        return [self.hidden.weight, self.hidden.bias, self.out.weight, self.out.bias]

    def slots(self, upper_params):
        """Index of (w1, b1, w2, b2) inside ``upper_params``, or None when the upper problem's trainable parameters are not
        exactly these four tensors (then the closed form does not describe d/d(upper) and autograd keeps the job)."""
        mine = self.tensors()
        if len(upper_params) != 4:
            return None
        idx = []
        for t in mine:
            hit = [i for i, p in enumerate(upper_params) if p is t]
            if len(hit) != 1:
                return None
            idx.append(hit[0])
        return idx if sorted(idx) == [0, 1, 2, 3] else None


class WeightedCEMLP:
    """ReLU-MLP with per-sample-weighted cross-entropy (+ optional ridge):

        L_in(w, lam) = (1/B) sum_i s_i(lam) * CE(f_w(x_i), y_i) + ridge * ||w||^2,
        s_i = weight_fn(CE_i.detach())          (e.g. a meta-weight-net of ``prev``)

    — the inner problem of examples/learning_to_reweight/main.py:117-127 (BASELINE cfg 2 / the
    metric's 10 M-parameter problem).  ``layers`` are the ``nn.Linear`` modules in forward order;
    ``curr.parameters()`` must enumerate them as [W1, b1, W2, b2, ...].

    Full Hessian (not Gauss-Newton), SURVEY.md Appendix A.3:
      cached per step:  h_l, masks m_l, softmax p, delta_l
      per direction (V_l, c_l):
        R-forward   Ra_l = Rh_{l-1} W_l^T + h_{l-1} V_l^T + c_l,  Rh_l = m_l * Ra_l
        top         Rdelta_L = (s/B) * (p*Rz - p (p.Rz))
        R-backward  Rdelta_{l-1} = m_{l-1} * (delta_l V_l + Rdelta_l W_l)
        outputs     H(W_l) = Rdelta_l^T h_{l-1} + delta_l^T Rh_{l-1} + 2 ridge V_l,
                    H(b_l) = sum_batch Rdelta_l + 2 ridge c_l
      mixed VJP to lam: c_i = (p_i - onehot_i).Rz_i(x) / B, then backward of sum_i c_i s_i(lam).

    With the HIP implementation and a narrow classifier head the whole K loop runs natively and "one pass"
    (``fused_cg`` / ``fused_neumann``, csrc/bhg_mlp.hip ``bhg_mlp_cg_solve`` / ``bhg_mlp_neumann_solve``): the
    kernels that produce H(W_l), H(b_l) apply the CG / Neumann update to the flat state vectors while the tile is
    on chip, so no N-sized H*direction vector exists.

    The HVP callable returns the Hessian WITHOUT its ``2*ridge*I`` part; ``hvp_shift = 2*ridge`` tells
    cg/neumann to add ``hvp_shift * direction`` inside the fused recurrence kernel.

    ``impl="hip"`` (the default, and the only path the package ever takes on its own) runs the GEMM
    chain on the MFMA kernels of libbhg and raises on CPU tensors; ``impl="torch"`` must be requested
    explicitly and evaluates the same formulas with ATen ops — it exists for the tests (math vs
    autograd, cross-check of the kernels) and for ``bench.py --hvp analytic-aten``.
    """

    def __init__(self, curr, prev, layers: Sequence[torch.nn.Linear], weight_fn: Callable, ridge: float = 0.0,
                 batch=None, impl: Optional[str] = None, fused: bool = True, keep_solution: bool = False, verify: bool = True,
                 weight_net: Optional["SigmoidMLPWeightNet"] = None):
        self.curr, self.prev = curr, prev
        # optional declared structure of weight_fn itself (closed-form sample weights and upper VJP on the HIP path)
        self.weight_net = weight_net
        self.verify = bool(verify)
        self.layers = list(layers)
        self.weight_fn = weight_fn
        self.ridge = float(ridge)
        # the ridge's Hessian 2*ridge*I is applied inside the CG/Neumann recurrence kernel (it holds the
        # direction in registers anyway), so the HVP kernels neither read V again nor add it
        self.hvp_shift = 2.0 * self.ridge
        self.batch = batch
        self.impl = impl
        # fused=False keeps the K loop as K x (HVP kernels + recurrence kernel) — the A/B arm of the tests and of
        # ``bench.py --no-fuse``; the product default lets the HVP's output kernels apply the recurrence themselves
        self.fused = bool(fused)
        # The fused CG solver does not need the N-sized solution x to produce the hypergradient: the mixed second
        # derivative of this structure only needs Rz(x) = sum_k alpha_k Rz(p_k), a batch x classes array the solver
        # accumulates from the Rz every iteration's head kernel computes anyway.  keep_solution=True materialises x
        # all the same (x <- x + alpha p inside the output kernels' epilogues, cg.py:49) for callers that want to read it.
        self.keep_solution = bool(keep_solution)
        params = list(curr.parameters())
        expect = []
        for lin in self.layers:
            expect += [lin.weight, lin.bias]
        if len(params) != len(expect) or any(a is not b for a, b in zip(params, expect)):
            raise ValueError("WeightedCEMLP: curr.parameters() must be [W1, b1, W2, b2, ...] of `layers`")

    # ---------------------------------------------------------------------------------------------
    def prepare(self):
        x, y = self.batch if self.batch is not None else self.curr.cur_batch
        impl = self.impl or "hip"  # the product path; it raises on CPU tensors (no CPU fallback)
        if impl == "hip":
            from ._mlp_hip import make_state  # noqa: PLC0415

            self._state = make_state(self, x, y)   # (a zero-padded twin for widths that are not multiples of 32)
        elif impl == "torch":
            self._state = _TorchMLPState(self, x, y)
        else:
            raise ValueError(f"unknown impl {impl!r}")
        if self.verify and VERIFY_STRUCTURE:
            self._verify_against_autograd(x, y)
        return self._state.hvp

    def _autograd_second_order(self, params, upper):
        """(direction, H direction, d(g . direction)/d upper) through the problem's REAL training_step: one double backward."""
        gen = torch.Generator(device="cpu").manual_seed(20240926)
        direction = [torch.randn(p.shape, generator=gen).to(device=p.device, dtype=p.dtype) for p in params]
        # the extra training_step of the check must not be seen by the run: the RNG streams are forked around it (dropout elsewhere in
        # the user's step keeps its sequence) — module buffers a training_step updates in place are the user's to exclude
        devs = [params[0].device] if params and params[0].is_cuda else []
        with torch.random.fork_rng(devices=devs), torch.enable_grad():
            loss = self.curr.training_step_exec(self.batch if self.batch is not None else self.curr.cur_batch)
            grads = torch.autograd.grad(loss, params, create_graph=True)
            dot = sum((g * d).sum() for g, d in zip(grads, direction))
            second = torch.autograd.grad(dot, params + upper, allow_unused=True)
        return direction, second[:len(params)], second[len(params):]

    def _estimate_ridge(self) -> float:
        """auto-structure only (after prepare()): the coefficient rho of a `rho * sum(w^2)` term in the problem's loss, read off ONE
        Hessian-vector product — H_autograd v - H_closed-form(ridge = 0) v = 2 rho v when that term is all the closed form misses.
        0.0 when the difference is not a multiple of v (the guard then judges the structure with ridge 0, and rejects it)."""
        params = list(self.curr.parameters())
        direction, hv_auto, _ = self._autograd_second_order(params, list(self.prev.trainable_parameters()))
        hv0 = [h.detach().clone() for h in self._state.hvp(direction)]
        num = sum(float(((a.double() - b.double()) * d.double()).sum()) for a, b, d in zip(hv_auto, hv0, direction))
        den = sum(float((d.double() ** 2).sum()) for d in direction)
        c = num / den
        res = sum(float(((a.double() - b.double() - c * d.double()) ** 2).sum()) for a, b, d in zip(hv_auto, hv0, direction)) ** 0.5
        ref = sum(float((a.double() ** 2).sum()) for a in hv_auto) ** 0.5
        # (a multiple of v below the product's own fp32 noise is no ridge: 1e-5 of |H v|)
        if c <= 0.0 or res > VERIFY_RTOL * max(ref, 1e-300) or c * den ** 0.5 <= 1e-5 * ref:
            return 0.0
        rho = 0.5 * c
        # a coefficient written as a short decimal in the user's code is recovered exactly when the estimate sits within fp32 noise of one
        for digits in range(1, 7):
            r = round(rho, digits)
            if r > 0 and abs(r - rho) <= 2e-4 * rho:
                return float(r)
        return float(rho)

    def _verify_against_autograd(self, x, y):
        """See VERIFY_STRUCTURE.  The verdict is cached on the problem object, keyed by what the closed form depends on."""
        params = list(self.curr.parameters())
        key = (tuple(tuple(p.shape) for p in params), int(x.shape[0]), self.ridge, type(self._state).__name__)
        done = self.curr.__dict__.setdefault("_bhg_structure_verified", set()) if hasattr(self.curr, "__dict__") else set()
        if key in done:
            return
        upper = list(self.prev.trainable_parameters())
        direction, hv_auto, mixed_auto = self._autograd_second_order(params, upper)
        hv = [h.detach().clone() + self.hvp_shift * d for h, d in zip(self._state.hvp(direction), direction)]
        coeff = self._state.mixed_coeff(direction)   # (the graph of the sample weights is kept: the real mixed_vjp comes later)
        mixed = self._state.upper_vjp(coeff, upper, retain_graph=True)   # closed-form weight net: its kernels are what is checked
        mixed = [torch.zeros_like(p) if g is None else g for g, p in zip(mixed, upper)]   # (a parameter the weights do not depend on)

        def rel(got, want):
            num = sum(float(((a.double() - (b.double() if b is not None else 0.0)) ** 2).sum()) for a, b in zip(got, want)) ** 0.5
            den = sum(float((b.double() ** 2).sum()) for b in want if b is not None) ** 0.5
            return num / den if den > 0 else num

        e_hvp, e_mix = rel(hv, hv_auto), rel(mixed, mixed_auto)
        tol, relaxed = VERIFY_RTOL, False
        # Config.precision fp16 / bf16: training_step_exec ran the autograd side under autocast (problem.py:327-332) while the closed
        # form is fp32 — a correct declaration then differs by the reduced precision's own error
        if str(getattr(getattr(self.curr, "config", None), "precision", "fp32")) in ("fp16", "bf16"):
            tol = max(tol, 5e-2)
        if not (e_hvp <= tol and e_mix <= tol):
            with torch.no_grad():   # how close does this instance sit to a ReLU kink?  (one fp32 forward through the declared layers)
                h, margin = x.detach().reshape(x.shape[0], -1).to(self.layers[0].weight.dtype), float("inf")
                for lin in self.layers[:-1]:
                    a = torch.addmm(lin.bias, h, lin.weight.t())
                    margin = min(margin, float(a.abs().min() / a.abs().mean().clamp_min(1e-30)))
                    h = torch.relu(a)
            if margin < VERIFY_KINK_MARGIN:
                tol = VERIFY_RTOL_ON_A_KINK
                relaxed = True
        bad = not (e_hvp <= tol and e_mix <= tol)
        # data-parallel runs: every rank raises or none does (a rank that raised alone would leave the others in a collective).  A
        # collective that FAILS is not swallowed: the ranks' verdicts would be unknown.
        import torch.distributed as dist  # noqa: PLC0415

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([1.0 if bad else 0.0, 1.0 if relaxed else 0.0], device=params[0].device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            bad, relaxed = bool(flag[0].item() > 0), bool(flag[1].item() > 0)
        if bad:
            raise StructureMismatchError(
                f"hypergradient_structure of problem {getattr(self.curr, 'name', '?')!r} declares {type(self).__name__}, but its "
                f"training_step disagrees with that closed form on a random direction: Hessian-vector product off by {e_hvp:.2e}, "
                f"mixed second derivative by {e_mix:.2e} (tolerance {tol:g}).  Typical causes: label smoothing, dropout, a "
                f"reduction other than the batch mean, a ridge that is not `ridge * sum(w^2)`, layers missing from `layers`.")
        if relaxed:
            # a pass under the kink tolerance is NOT cached at once: a slightly wrong declaration (label smoothing ~0.01) can hide under
            # 3e-2 on such a batch, so the next prepare() — normally another batch — is checked again at the strict tolerance
            # (ADVICE r5).  A problem that only ever shows such batches (one fixed batch: bench.py's seed 4) is accepted after
            # VERIFY_KINK_RETRIES relaxed passes, said once.
            tries = self.curr.__dict__.setdefault("_bhg_structure_relaxed", {}) if hasattr(self.curr, "__dict__") else {}
            tries[key] = tries.get(key, 0) + 1
            final = tries[key] >= VERIFY_KINK_RETRIES
            if tries[key] == 1 or final:
                warnings.warn(
                    f"betty_amd: structure check of problem {getattr(self.curr, 'name', '?')!r} passed only under the ReLU-kink tolerance "
                    f"({tol:g}; Hessian-vector product off by {e_hvp:.2e}, mixed second derivative by {e_mix:.2e}: some hidden "
                    "pre-activation of this batch sits within 1e-6 of zero) — " +
                    (f"accepted after {tries[key]} such batches" if final else "not cached, the next batch is checked again"),
                    RuntimeWarning, stacklevel=3)
            if not final:
                return
        done.add(key)

    # Optional protocol extension: a provider whose HVP kernels can apply the recurrence themselves runs the whole K
    # loop ("one pass": no N-sized H*direction vector).  Both return False when the fused path does not apply and the
    # caller falls back to K x (hvp_fn + recurrence kernel).
    def fused_cg_ready(self, layout, K: int) -> bool:
        st = self._state
        return K > 0 and self.fused and hasattr(st, "cg_solve") and st.fused_supported(layout)

    def fused_cg_skips_solution(self, layout, K: int) -> bool:
        """True when fused_cg will run AND leaves x untouched (the caller may then skip zeroing it)."""
        return (not self.keep_solution) and self.fused_cg_ready(layout, K)

    def fused_cg_state_mask(self, layout, K: int):
        """None, or the bit mask (bit t: tensor t of the layout) of the state slices fused_cg reads / writes — the others of r and p
        need not be initialised when ``rhs`` (the right-hand side's own tensors) is handed to fused_cg."""
        st = self._state
        if not self.fused_cg_skips_solution(layout, K) or not hasattr(st, "cg_state_mask"):
            return None
        return st.cg_state_mask()

    def fused_cg(self, layout, x, r, p, K: int, cg_alpha: float, rhs=None):
        """False, or the token of the solve (hand it to mixed_vjp(..., solve=token))."""
        if not self.fused_cg_ready(layout, K):
            return False
        return self._state.cg_solve(layout, x, r, p, K, cg_alpha, self.hvp_shift, keep_x=self.keep_solution, rhs=rhs)

    # Global-batch CG (betty_amd/global_hvp.py): the same one-pass iteration, cut where the ranks must talk.
    def fused_cg_global_ready(self, layout, K: int) -> bool:
        st = self._state
        return K > 0 and self.fused and hasattr(st, "cg_global_phase") and st.fused_supported(layout)

    def fused_cg_global_skips_solution(self, layout, K: int) -> bool:
        return (not self.keep_solution) and self.fused_cg_global_ready(layout, K) and bool(getattr(self._state, "solution_free", False))

    def cg_global_phase(self, layout, x, r, p, k: int, K: int, phase: int, world: int, php, cg_alpha: float) -> None:
        keep = not self.fused_cg_global_skips_solution(layout, K)
        self._state.cg_global_phase(layout, x, r, p, k, K, phase, world, php, cg_alpha, self.hvp_shift, keep_x=keep)

    def cg_global_finish(self, layout, K: int, cg_alpha: float):
        """The token of the solve (hand it to mixed_vjp(..., solve=token)), or True."""
        keep = not self.fused_cg_global_skips_solution(layout, K)
        return self._state.cg_global_finish(layout, K, cg_alpha, keep_x=keep)

    # Global-batch CG, factor-exchange form: the fully projected solver on sample-partitioned data (no N-sized exchange, no x).
    def fused_cg_fx_ready(self, layout, K: int, world: int) -> bool:
        st = self._state
        return (K > 0 and self.fused and not self.keep_solution and hasattr(st, "cg_fx_phase") and st.fx_supported(layout, world))

    def cg_fx_phase(self, rhs, k: int, K: int, phase: int, world: int, rank: int, cg_alpha: float) -> None:
        self._state.cg_fx_phase(rhs, k, K, phase, world, rank, cg_alpha, self.hvp_shift)

    def cg_fx_finish(self, layout, K: int, cg_alpha: float):
        return self._state.cg_fx_finish(layout, K, cg_alpha)

    def fused_neumann_fx_ready(self, layout, K: int, world: int) -> bool:
        st = self._state
        return (K > 0 and self.fused and not self.keep_solution and hasattr(st, "neumann_fx_phase") and st.fx_supported(layout, world))

    def neumann_fx_phase(self, rhs, k: int, K: int, phase: int, world: int, rank: int, alpha: float) -> None:
        self._state.neumann_fx_phase(rhs, k, K, phase, world, rank, alpha, self.hvp_shift)

    def neumann_fx_finish(self, layout, K: int, alpha: float):
        return self._state.neumann_fx_finish(layout, K, alpha)

    def fused_neumann_ready(self, layout, K: int) -> bool:
        st = self._state
        return K > 0 and self.fused and hasattr(st, "neumann_solve") and st.fused_supported(layout)

    def fused_neumann_skips_solution(self, layout, K: int) -> bool:
        """True when fused_neumann will run AND leaves the accumulator p untouched (keep_solution=False): the mixed
        derivative then comes from sum_k Rz(v_k), collected by the head kernel, plus one R-forward of the last v."""
        return (not self.keep_solution) and self.fused_neumann_ready(layout, K)

    def fused_neumann(self, layout, v, p, K: int, alpha: float):
        """False, or the token of the solve (hand it to mixed_vjp(..., solve=token))."""
        if not self.fused_neumann_ready(layout, K):
            return False
        # second direction buffer: the R-backward GEMMs of an HVP still read v while its epilogues write v'
        v_alt = next(t for t in layout.state(3) if t is not v and t is not p)
        return self._state.neumann_solve(layout, v, v_alt, p, K, alpha, self.hvp_shift, keep_p=self.keep_solution)

    def mixed_vjp(self, neg_x_views, sync: bool, solve=None):
        """``solve``: token of the fused solve whose solution ``neg_x_views`` name (required when that solution was never
        materialised); without it the views are read like any direction."""
        st = self._state
        coeff = st.mixed_coeff(neg_x_views, solve) if solve is not None else st.mixed_coeff(neg_x_views)  # [B]: d(g.(-x))/d s_i
        upper = self.prev.trainable_parameters()
        if getattr(st, "native_upper", False):
            # closed-form weight net (csrc/bhg_mwn.hip): the M-sized result lands in ONE fresh flat buffer; sync=True accumulates it
            # into .grad like Problem.set_grads (problem.py:583-597) after the data-parallel mean the declaration asks for
            wn = self.weight_net
            world, group = 1, None
            if sync:
                import torch.distributed as dist  # noqa: PLC0415

                if dist.is_available() and dist.is_initialized():
                    from ..distributed import ddp_process_group_of  # noqa: PLC0415

                    if wn.average_over is not None:
                        group = None if wn.average_over is True else wn.average_over
                        world = dist.get_world_size(group)
                    else:
                        # no declaration: do what the reference's backward() through the wrapper would have done
                        wrapped, group = ddp_process_group_of(getattr(self.prev, "fwd", None), getattr(self.prev, "module", None))
                        if wrapped:
                            world = dist.get_world_size(group)
                        elif dist.get_world_size() > 1 and getattr(self, "expects_data_parallel_mean", False):
                            # the global-batch mode promises a GLOBAL hypergradient: a rank-local accumulation would silently diverge
                            raise RuntimeError(
                                "cg_global(sync=True) with a closed-form weight net: declare SigmoidMLPWeightNet(average_over=True) (or a "
                                "process group) or wrap the upper module in DistributedDataParallel — otherwise every rank would "
                                "accumulate the hypergradient of its own batch share only")
            grads, flat = st.upper_vjp(coeff, upper, scale=1.0 / world, with_flat=True)
            if sync:
                from ..distributed import defer_grad_sync, fence_grads, install_optimizer_fence  # noqa: PLC0415

                fence_grads()   # an earlier deferred mean into these parameters: order this stream behind it before .grad is touched
            if world > 1:
                if wn.overlap and sync:
                    # pre-scaled by 1 / world: the mean (DDP's reduction).  Issued asynchronously: the views of `flat` below are
                    # accumulated / assigned only AFTER the collective in stream order wherever they are read (fence_grads)
                    if all(p_.grad is None for p_ in upper):
                        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
                        from ..backend import get_backend  # noqa: PLC0415

                        defer_grad_sync(work, flat, get_backend())
                        install_optimizer_fence(getattr(self.prev, "optimizer", None))
                    else:   # accumulation onto an existing .grad reads the reduced values right away: nothing to overlap with
                        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if sync:
                for p_, g_ in zip(upper, grads):
                    p_.grad = g_ if p_.grad is None else p_.grad + g_
                return None
            return grads
        if sync:
            torch.autograd.backward(st.sample_weight, grad_tensors=coeff.reshape(st.sample_weight.shape), inputs=upper)
            return None
        return list(torch.autograd.grad(st.sample_weight, upper, grad_outputs=coeff.reshape(st.sample_weight.shape)))


class _TorchMLPState:
    """ATen evaluation of the closed form (device agnostic; the math reference for the kernels)."""

    def __init__(self, spec: WeightedCEMLP, x, y):
        self.spec = spec
        Ws = [lin.weight.detach() for lin in spec.layers]
        bs = [lin.bias.detach() for lin in spec.layers]
        B = x.shape[0]
        hs, masks = [x.detach()], []
        h = hs[0]
        for l, (W, b) in enumerate(zip(Ws, bs)):
            a = torch.addmm(b, h, W.t())
            if l + 1 < len(Ws):
                m = (a > 0).to(a.dtype)
                h = a * m
                masks.append(m)
                hs.append(h)
            else:
                z = a
        logp = F.log_softmax(z, dim=1)
        p = logp.exp()
        ce = -logp.gather(1, y.reshape(-1, 1)).reshape(-1)
        self.sample_weight = spec.weight_fn(ce.detach())  # keeps the graph to prev's parameters
        sd = self.sample_weight.detach().reshape(-1) / B
        onehot = F.one_hot(y, z.shape[1]).to(z.dtype)
        self.err = p - onehot  # [B, C]
        deltas = [None] * len(Ws)
        deltas[-1] = sd[:, None] * self.err
        for l in range(len(Ws) - 1, 0, -1):
            deltas[l - 1] = masks[l - 1] * (deltas[l] @ Ws[l])
        self.Ws, self.hs, self.masks, self.p, self.sd, self.deltas, self.B = Ws, hs, masks, p, sd, deltas, B

    native_upper = False

    def upper_vjp(self, coeff, upper, retain_graph=False):
        return list(torch.autograd.grad(self.sample_weight, upper, grad_outputs=coeff.reshape(self.sample_weight.shape),
                                        retain_graph=retain_graph, allow_unused=retain_graph))   # (retain_graph: the structure check)

    def _r_forward(self, Vs, cs):
        Rh, Rhs = None, [None]
        for l, (W, V, c) in enumerate(zip(self.Ws, Vs, cs)):
            Ra = torch.addmm(c, self.hs[l], V.t())
            if Rh is not None:
                Ra = Ra + Rh @ W.t()
            if l + 1 < len(self.Ws):
                Rh = self.masks[l] * Ra
                Rhs.append(Rh)
        return Ra, Rhs  # Rz, [None, Rh_1, ..]

    def hvp(self, direction_views):
        Vs, cs = direction_views[0::2], direction_views[1::2]
        rho2 = 0.0  # ridge part handled by the recurrence kernel (spec.hvp_shift)
        Rz, Rhs = self._r_forward(Vs, cs)
        Rd = self.sd[:, None] * (self.p * Rz - self.p * (self.p * Rz).sum(1, keepdim=True))
        out = [None] * (2 * len(self.Ws))
        for l in range(len(self.Ws) - 1, -1, -1):
            HW = Rd.t() @ self.hs[l]
            if Rhs[l] is not None:
                HW = HW + self.deltas[l].t() @ Rhs[l]
            out[2 * l] = HW + rho2 * Vs[l] if rho2 else HW
            Hb = Rd.sum(0)
            out[2 * l + 1] = Hb + rho2 * cs[l] if rho2 else Hb
            if l > 0:
                Rd = self.masks[l - 1] * (self.deltas[l] @ Vs[l] + Rd @ self.Ws[l])
        return out

    def mixed_coeff(self, dir_views, solve=None):
        if solve is not None:   # the factor-exchange solve: Rz(x) was accumulated, x = -cg_alpha * sum_k alpha_k p_k never existed
            if solve is not getattr(self, "_fx_token", None):
                raise RuntimeError("stale fused-solve token")
            Rz = (-solve[1]) * self._fx["Rzx"].to(self.err.dtype)
            return (self.err * Rz).sum(1) / self.B
        Rz, _ = self._r_forward(dir_views[0::2], dir_views[1::2])
        return (self.err * Rz).sum(1) / self.B

    # ---- global-batch CG, phase by phase: the math bhg_mlp_cg_global_phase implements, in ATen (tests: gloo, CPU) -----------------
    def fused_supported(self, layout) -> bool:
        want = []
        for W in self.Ws:
            want += [W.numel(), W.shape[0]]
        return tuple(want) == tuple(layout.numels)

    def cg_global_phase(self, layout, x, r, p, k, K, phase, world, php, cg_alpha, shift, keep_x=True):
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        shapes = []
        for W in self.Ws:
            shapes += [W.shape, (W.shape[0],)]
        views = lambda flat: [flat[s: s + n].view(sh) for s, n, sh in zip(layout.starts, layout.numels, shapes)]
        dot = lambda a, b: float((a.double() * b.double()).sum())
        if phase == 0:      # BHG_CG_GLOBAL_CHAIN
            if k == 0:
                self._g = {"rr": dot(r, r)}
                self._g["pp"] = self._g["rr"]
            else:
                g = self._g
                beta = f32(g["rr_new"]) / f32(g["rr"])
                p.mul_(beta).add_(r)                    # cg.py:53 (the kernels form it lazily where they read it)
                b = float(beta)
                g["pp"] = g["rr_new"] + 2.0 * b * g["rp"] + b * b * g["pp_old"]
                g["rr"] = g["rr_new"]
            hv = self.hvp(views(p))
            self._hv = torch.zeros_like(p)
            for dst, h in zip(views(self._hv), hv):
                dst.copy_(h)
            php[0] = dot(p, self._hv)                   # this rank's p . H_data p
        elif phase == 1:    # BHG_CG_GLOBAL_UPDATE
            g = self._g
            den = float(cg_alpha) * (float(php[0]) / world + float(shift) * g["pp"])
            alpha = f32(g["rr"]) / f32(den)             # cg.py:47
            hp = self._hv + f32(shift) * p
            r.sub_(alpha * hp)                          # cg.py:50 on the local Hessian: the ranks' mean is the global r'
            x.add_(alpha * p)                           # cg.py:49
            if k == K - 1:
                x.mul_(-float(cg_alpha))                # cg.py:56 and the negation of cg.py:59/68
        else:               # BHG_CG_GLOBAL_DOTS, after the residual's exchange
            g = self._g
            g["rr_new"], g["rp"], g["pp_old"] = dot(r, r), dot(r, p), dot(p, p)

    def cg_global_finish(self, layout, K, cg_alpha, keep_x=True):
        return True

    # ---- global-batch CG, factor-exchange form: the math of csrc/mlp/fx.inc phase by phase, in ATen (tests: gloo, CPU).  Same slot
    # conventions as HipMLPState: every phase writes THIS rank's row of the buffer the caller gathers after it.
    def fx_supported(self, layout, world: int) -> bool:
        return self.fused_supported(layout) and len(self.Ws) >= 3

    def fx_buffers(self, world: int):
        held = self.__dict__.setdefault("_fx_bufs", {})
        if world not in held:
            L, B = len(self.Ws), self.B
            cf = sum(h.shape[1] for h in self.hs) * B + sum(d.shape[1] for d in self.deltas[1:]) * B
            sf = sum(W.shape[0] for W in self.Ws) * B + sum(W.shape[0] for W in self.Ws[:-1]) * B
            dev, dt = self.hs[0].device, self.hs[0].dtype
            held[world] = {"const": torch.zeros(world, cf, dtype=dt, device=dev), "slab": torch.zeros(world, sf, dtype=dt, device=dev),
                           "scal": torch.zeros(world, 3, dtype=torch.float64, device=dev), "xws": torch.zeros(1, dtype=torch.uint8, device=dev)}
        return held[world]

    def _fx_chain(self, st):
        """The R-chain on products-with-the-batch (Gf_p, Gb_p) and the narrow slices of the direction; returns Rz, [Rh_l], [Rd_l]."""
        L, Ws = len(self.Ws), self.Ws
        Rhs, Rh = [], None
        for l in range(L - 1):
            Ra = st["Gf_p"][l] + st["p_c"][l]
            if Rh is not None:
                Ra = Ra + Rh @ Ws[l].t()
            Rh = self.masks[l] * Ra
            Rhs.append(Rh)
        Rz = self.hs[L - 1] @ st["p_V"].t() + st["p_c"][L - 1] + Rh @ Ws[L - 1].t()
        Rd = self.sd[:, None] * (self.p * Rz - self.p * (self.p * Rz).sum(1, keepdim=True))
        Rds = [None] * L
        Rds[L - 1] = Rd
        for l in range(L - 1, 0, -1):
            Gb = self.deltas[l] @ st["p_V"] if l == L - 1 else st["Gb_p"][l]
            Rd = self.masks[l - 1] * (Gb + Rd @ Ws[l])
            Rds[l - 1] = Rd
        return Rz, Rhs, Rds

    def cg_fx_phase(self, rhs, k, K, phase, world, rank, cg_alpha, shift):
        L, B, G = len(self.Ws), self.B, world
        bufs = self.fx_buffers(world)
        dd = lambda a, b: float((a.double() * b.double()).sum())
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        wide = range(L - 1)
        split_rows = lambda buf, widths: [list(torch.split(buf[g].view(B, -1), widths, 1)) for g in range(G)]
        if phase == 0:      # BEGIN
            bufs["const"][rank].copy_(torch.cat(self.hs + self.deltas[1:], 1).reshape(-1))
            self._fx = {}
            return
        st = self._fx
        if phase == 1:      # CHAIN
            if k == 0:
                widths = [h.shape[1] for h in self.hs] + [d.shape[1] for d in self.deltas[1:]]
                rows = split_rows(bufs["const"], widths)
                st["h_all"] = [torch.cat([rows[g][l] for g in range(G)], 0) for l in range(L)]
                st["d_all"] = [None] + [torch.cat([rows[g][L + l - 1] for g in range(G)], 0) for l in range(1, L)]
                vec = [t.detach() for t in rhs]
                st["r_c"] = [vec[2 * l + 1].clone() for l in range(L)]
                st["p_c"] = [t.clone() for t in st["r_c"]]
                st["r_V"] = vec[2 * (L - 1)].clone()
                st["p_V"] = st["r_V"].clone()
                st["Gf_r"] = [self.hs[l] @ vec[2 * l].t() for l in wide]
                st["Gb_r"] = [None] + [self.deltas[l] @ vec[2 * l] for l in range(1, L - 1)]
                st["Gf_p"] = [t.clone() for t in st["Gf_r"]]
                st["Gb_p"] = [None] + [t.clone() for t in st["Gb_r"][1:]]
                st["rr_w"] = sum(dd(vec[2 * l], vec[2 * l]) for l in wide)
                st["rp_w"] = st["pp_w"] = st["rr_w"]
                st["Rzx"] = torch.zeros(B, self.Ws[-1].shape[0], dtype=torch.float64, device=self.hs[0].device)
            else:
                self._fx_step(bufs, G, cg_alpha, shift, last=False)
            st["Rz"], st["Rhs"], st["Rds"] = self._fx_chain(st)
            bufs["slab"][rank].copy_(torch.cat(st["Rds"] + st["Rhs"], 1).reshape(-1))
            return
        if phase == 2:      # GRAM
            widths = [t.shape[1] for t in st["Rds"]] + [t.shape[1] for t in st["Rhs"]]
            rows = split_rows(bufs["slab"], widths)
            Rd_all = [torch.cat([rows[g][l] for g in range(G)], 0) for l in range(L)]
            Rh_all = [torch.cat([rows[g][L + l] for g in range(G)], 0) for l in range(L - 1)]
            h_all, d_all = st["h_all"], st["d_all"]
            Gf_raw, Gb_raw = [], [None]
            for l in wide:
                t = (self.hs[l] @ h_all[l].t()) @ Rd_all[l]
                if l >= 1:
                    t = t + (self.hs[l] @ Rh_all[l - 1].t()) @ d_all[l]
                Gf_raw.append(t / G)
            for l in range(1, L - 1):
                Gb_raw.append(((self.deltas[l] @ Rd_all[l].t()) @ h_all[l] + (self.deltas[l] @ d_all[l].t()) @ Rh_all[l - 1]) / G)
            st["Gf_raw"], st["Gb_raw"] = Gf_raw, Gb_raw
            st["raw_c"] = [Rd_all[l].sum(0) / G for l in range(L)]
            st["raw_V"] = (Rd_all[L - 1].t() @ h_all[L - 1] + d_all[L - 1].t() @ Rh_all[L - 2]) / G

            def share(Gf_u, Gb_u):
                return sum(dd(st["Rds"][l], Gf_u[l]) for l in wide) + sum(dd(st["Rhs"][l - 1], Gb_u[l]) for l in range(1, L - 1))

            bufs["scal"][rank].copy_(torch.tensor([share(st["Gf_r"], st["Gb_r"]), share(st["Gf_p"], st["Gb_p"]), share(Gf_raw, Gb_raw)],
                                                  dtype=torch.float64))
            return
        self._fx_step(bufs, G, cg_alpha, shift, last=True)   # END

    def _fx_step(self, bufs, G, cg_alpha, shift, last):
        """k_fx_step: alpha from the gathered shares and the narrow slices; recurrences; beta (cg.py:42-53 on batch-sized quantities)."""
        st, L = self._fx, len(self.Ws)
        dd = lambda a, b: float((a.double() * b.double()).sum())
        f32 = lambda v: torch.tensor(float(v), dtype=torch.float32)
        wide = range(L - 1)
        tot = bufs["scal"].sum(0)
        r_raw, p_raw, raw_raw = float(tot[0]) / G, float(tot[1]) / G, float(tot[2]) / G
        p_raw_n = sum(dd(st["p_c"][l], st["raw_c"][l]) for l in range(L)) + dd(st["p_V"], st["raw_V"])
        pp_n = sum(dd(t, t) for t in st["p_c"]) + dd(st["p_V"], st["p_V"])
        rr_n = sum(dd(t, t) for t in st["r_c"]) + dd(st["r_V"], st["r_V"])
        rr = st["rr_w"] + rr_n
        den = float(cg_alpha) * ((p_raw + p_raw_n) + float(shift) * (st["pp_w"] + pp_n))
        alpha = float(f32(rr) / f32(den))
        st["Rzx"] += alpha * st["Rz"].double()
        if last:
            return
        for l in range(L):
            st["r_c"][l] = st["r_c"][l] - alpha * (st["raw_c"][l] + shift * st["p_c"][l])
        st["r_V"] = st["r_V"] - alpha * (st["raw_V"] + shift * st["p_V"])
        for l in wide:
            st["Gf_r"][l] = st["Gf_r"][l] - alpha * (st["Gf_raw"][l] + shift * st["Gf_p"][l])
        for l in range(1, L - 1):
            st["Gb_r"][l] = st["Gb_r"][l] - alpha * (st["Gb_raw"][l] + shift * st["Gb_p"][l])
        rHp = r_raw + shift * st["rp_w"]
        pHp = p_raw + shift * st["pp_w"]
        HpHp = raw_raw + 2.0 * shift * p_raw + shift * shift * st["pp_w"]
        rr_w1 = st["rr_w"] - 2.0 * alpha * rHp + alpha * alpha * HpHp
        rp_w1 = st["rp_w"] - alpha * pHp
        rr_new = rr_w1 + sum(dd(t, t) for t in st["r_c"]) + dd(st["r_V"], st["r_V"])
        beta = float(f32(rr_new) / f32(rr))
        for l in range(L):
            st["p_c"][l] = st["r_c"][l] + beta * st["p_c"][l]
        st["p_V"] = st["r_V"] + beta * st["p_V"]
        for l in wide:
            st["Gf_p"][l] = st["Gf_r"][l] + beta * st["Gf_p"][l]
        for l in range(1, L - 1):
            st["Gb_p"][l] = st["Gb_r"][l] + beta * st["Gb_p"][l]
        st["pp_w"] = rr_w1 + 2.0 * beta * rp_w1 + beta * beta * st["pp_w"]
        st["rp_w"] = rr_w1 + beta * rp_w1
        st["rr_w"] = rr_w1

    def cg_fx_finish(self, layout, K, cg_alpha):
        self._fx_token = ("cg_fx", float(cg_alpha))
        return self._fx_token

    def neumann_fx_phase(self, rhs, k, K, phase, world, rank, alpha, shift):
        """neumann.py:59-66 on the global batch, factor-exchange form (bhg_mlp_neumann_fx_phase): no scalars; the direction lives in the
        p slots of the state; Rzx = sum_{k <= K} Rz(v_k)."""
        L = len(self.Ws)
        if phase in (0, 2) or (phase == 1 and k == 0):      # BEGIN, GRAM, and the first CHAIN are the CG form's
            return self.cg_fx_phase(rhs, k, K, phase, world, rank, alpha, shift)
        st = self._fx
        st["Rzx"] += st["Rz"].double()                      # Rz(v_{k-1}) (END: Rz(v_K))
        if phase == 3:
            return
        for l in range(L):
            st["p_c"][l] = st["p_c"][l] - alpha * (st["raw_c"][l] + shift * st["p_c"][l])
        st["p_V"] = st["p_V"] - alpha * (st["raw_V"] + shift * st["p_V"])
        for l in range(L - 1):
            st["Gf_p"][l] = st["Gf_p"][l] - alpha * (st["Gf_raw"][l] + shift * st["Gf_p"][l])
        for l in range(1, L - 1):
            st["Gb_p"][l] = st["Gb_p"][l] - alpha * (st["Gb_raw"][l] + shift * st["Gb_p"][l])
        st["Rz"], st["Rhs"], st["Rds"] = self._fx_chain(st)
        self.fx_buffers(world)["slab"][rank].copy_(torch.cat(st["Rds"] + st["Rhs"], 1).reshape(-1))

    def neumann_fx_finish(self, layout, K, alpha):
        self._fx_token = ("neumann_fx", float(alpha))
        return self._fx_token


class LogisticRegressionL2:
    """Logistic regression with a per-weight L2 penalty (the inner problem
    of examples/logistic_regression_hpo/logistic_regression_implicit.py:80-91 and
    test/test_regression.py:47-59):

        L_in(w, lam) = mean_i BCE(x_i.w, y_i) + 1/2 sum_j lam_j w_j^2

    H p = X^T( s * (X p) ) + lam * p with s_i = sigma_i (1 - sigma_i) / n  — two GEMV passes over X
    (csrc/bhg_logreg.hip); the mixed derivative of g.x w.r.t. the lam TENSOR is w * x, pushed into
    ``prev``'s parameters through ``lam``'s own graph.  ``lam_fn()`` must return lam (shape [d]) as
    a function of ``prev``'s parameters.
    """

    def __init__(self, curr, prev, weight: torch.nn.Parameter, lam_fn: Callable, batch=None):
        self.curr, self.prev, self.weight, self.lam_fn, self.batch = curr, prev, weight, lam_fn, batch
        params = list(curr.parameters())
        if len(params) != 1 or params[0] is not weight:
            raise ValueError("LogisticRegressionL2: curr.parameters() must be [weight]")

    def prepare(self):
        from .. import _native  # noqa: PLC0415

        x, _y = self.batch if self.batch is not None else self.curr.cur_batch
        if not x.is_cuda:
            raise _native.NativeLibraryError("LogisticRegressionL2 needs CUDA/HIP tensors; there is no CPU fallback")
        self.lib = _native.load()
        self.X = x.detach().to(torch.float32).contiguous()
        n, d = self.X.shape
        self.n, self.d = n, d
        self.lam = self.lam_fn()  # keeps the graph to prev's parameters
        self.lam_d = self.lam.detach().to(torch.float32).contiguous()
        self.w = self.weight.detach().to(torch.float32).contiguous()
        self.s = torch.empty(n, device=x.device)
        self.tmp = torch.empty(int(self.lib.bhg_logreg_tmp_floats(n, d)), device=x.device)
        self.out = torch.empty(d, device=x.device)
        self._stream = lambda: int(torch.cuda.current_stream().cuda_stream)
        _native.check(self.lib.bhg_logreg_prepare(self.X.data_ptr(), self.w.data_ptr(), self.s.data_ptr(), n, d, self._stream()),
                      "bhg_logreg_prepare")
        self._native = _native
        return self.hvp

    def hvp(self, direction_views):
        (p,) = direction_views
        p = p.detach().to(torch.float32).contiguous()
        self._native.check(
            self.lib.bhg_logreg_hvp(self.X.data_ptr(), self.s.data_ptr(), self.lam_d.data_ptr(), p.data_ptr(),
                                    self.out.data_ptr(), self.tmp.data_ptr(), self.n, self.d, self._stream()),
            "bhg_logreg_hvp",
        )
        return [self.out.view(self.weight.shape)]

    def mixed_vjp(self, neg_x_views, sync: bool):
        coeff = (self.w * neg_x_views[0].reshape(-1)).reshape(self.lam.shape)  # d(g.(-x))/d lam
        upper = self.prev.trainable_parameters()
        if sync:
            torch.autograd.backward(self.lam, grad_tensors=coeff, inputs=upper)
            return None
        return list(torch.autograd.grad(self.lam, upper, grad_outputs=coeff))


class ProximalRegularized:
    """Inner loss = data loss + reg * ||w - theta||^2 with theta the UPPER problem's parameters, one per
    inner parameter — implicit MAML (examples/implicit_maml/main.py:87-92,122-129).

    The proximal term's Hessian is 2*reg*I and its mixed derivative is -2*reg*I, so
      * the HVP callback differentiates the DATA loss only (PyTorch double backward through the user's
        network) and ``hvp_shift = 2*reg`` lets the CG/Neumann kernel add 2*reg*p from the direction it
        already holds in registers — the T elementwise double-backward kernels of the prox term disappear;
      * the final hop is closed form: hypergradient = +2*reg*x, i.e. ``-2*reg * (-alpha*x)`` applied to the
        flat result — no second-order autograd call at all.
    ``data_loss(batch)`` must return the loss WITHOUT the proximal term; ``prev.trainable_parameters()`` must
    align one-to-one with ``curr.parameters()``.
    """

    hvp_is_autograd = True   # the HVP callback is an opaque double backward: cg/neumann may replay it as a HIP graph

    def __init__(self, curr, prev, data_loss: Callable, reg: float, batch=None):
        self.curr, self.prev, self.data_loss, self.reg, self.batch = curr, prev, data_loss, float(reg), batch
        self.hvp_shift = 2.0 * self.reg
        inner, upper = list(curr.parameters()), list(prev.trainable_parameters())
        if len(inner) != len(upper) or any(a.shape != b.shape for a, b in zip(inner, upper)):
            raise ValueError("ProximalRegularized: upper and inner parameters must match one-to-one")

    def prepare(self):
        import warnings  # noqa: PLC0415

        batch = self.batch if self.batch is not None else self.curr.cur_batch
        loss = self.data_loss(batch)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self._in_grad = torch.autograd.grad(loss, self.curr.trainable_parameters(), create_graph=True)
        params = list(self.curr.parameters())

        def hvp(direction_views):
            return torch.autograd.grad(self._in_grad, params, grad_outputs=direction_views, retain_graph=True)

        return hvp

    def mixed_vjp(self, neg_x_views, sync: bool):
        self._in_grad = None  # release the double-backward graph
        grads = [(-2.0 * self.reg) * t for t in neg_x_views]  # = +2*reg*(alpha*x)
        if sync:
            # accumulate into .grad THROUGH autograd so DistributedDataParallel's reducer hooks fire
            torch.autograd.backward(list(self.prev.trainable_parameters()), grad_tensors=grads)
            return None
        return grads

