"""Conjugate-gradient best-response-Jacobian product (iMAML style) on fused gfx950 kernels.

Behavioural twin of /root/reference betty/hypergradient/cg.py:8-70 — same signature, same
iteration, including the reference's ``cg_alpha`` quirk (the step length is computed with
``cg_alpha * Hp`` but the residual is updated with the un-scaled ``Hp``, cg.py:42-50) and the
absence of any convergence test or breakdown guard.  What changes is where the arithmetic runs:
``x, r, p`` are flat HBM vectors, the HVP tensors autograd returns are consumed in place through a
pointer table, and each iteration's dots/AXPYs are one persistent kernel (28*N bytes) or three
streaming kernels (40*N bytes) instead of ~10*T ATen launches moving ~140*N bytes.
"""
from __future__ import annotations

import torch

from ..backend import get_backend
from ._common import (AutogradHVP, ForwardOverReverseHVP, GraphedHVP, forward_over_reverse_wanted, hvp_graph_wanted, inner_gradient,
                      mixed_vjp, persistent_graphs_for, solve_stream)
from .structured import structured_hvp_for


def cg(vector, curr, prev, sync):
    """``vector`` (list aligned with ``curr``'s parameters) -> best-response-Jacobian^T product
    in ``prev``'s parameter space.  ``sync=True``: accumulate into ``prev`` ``.grad`` and return
    None; ``sync=False``: return the list (cg.py:58-68)."""
    assert len(curr.paths) == 0, "cg method is not supported for higher order MLO!"
    vector = list(vector)
    provider = structured_hvp_for(curr, prev)
    K = int(curr.config.cg_iterations)
    # opaque double backward (no structure, or a structure whose HVP is an autograd callback): replayed as a HIP graph
    graphed = (provider is None or getattr(provider, "hvp_is_autograd", False)) and hvp_graph_wanted(K, vector, curr)
    # hypergradient_graph = "persistent": loss / gradient-with-graph AND the HVP captured once for the whole run
    persist = persistent_graphs_for(curr, K, vector, prev) if provider is None else None
    with solve_stream(vector[0].device if vector else None, graphed or persist is not None):
        return _cg(vector, curr, prev, sync, provider, K, graphed, persist)


def _cg(vector, curr, prev, sync, provider, K, graphed, persist=None):
    config = curr.config
    be = get_backend()
    layout = be.layout(vector)
    x, r, p = layout.state(3)
    keep_graph = for_hvp = False
    if provider is None:
        if persist is not None:
            in_grad, hvp_fn, keep_graph = persist.begin_step(curr, list(curr.parameters()), layout.views(p, vector), prev)
        elif forward_over_reverse_wanted(curr):
            # opt-in: H p by forward-over-reverse passes (no double-backward graph; _common.ForwardOverReverseHVP)
            in_grad, hvp_fn, for_hvp = None, ForwardOverReverseHVP(curr, prev), True
        else:
            in_grad = inner_gradient(curr)
            hvp_fn = AutogradHVP(in_grad, curr.parameters())
    else:
        in_grad = None
        hvp_fn = provider.prepare()
    if graphed and persist is None and not for_hvp:
        hvp_fn = GraphedHVP(hvp_fn)

    alpha = float(config.cg_alpha)
    fused = getattr(provider, "fused_cg", None)
    # a provider whose fused solver derives the mixed derivative from batch-sized factors never touches x (see
    # WeightedCEMLP.keep_solution): then x is not even zeroed
    skips = getattr(provider, "fused_cg_skips_solution", None)
    skip_x = bool(fused is not None and alpha != 0.0 and skips is not None and skips(layout, K))
    # ... and a fused solver that works from batch-sized projections reads the N-sized right-hand side once, in its first iteration:
    # it says which tensors' slices of r / p it needs at all (fused_cg_state_mask) and reads the others from `vector` itself
    state_mask = getattr(provider, "fused_cg_state_mask", None)
    keep_mask = state_mask(layout, K) if (skip_x and state_mask is not None) else None
    # the solver reads the un-masked tensors' values straight from `vector` with 16-byte loads: a right-hand side that is a slice of
    # some flat buffer at an odd offset (or another dtype / layout) takes the copying initialisation instead
    if keep_mask is not None and not all(t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0 for t in vector):
        keep_mask = None
    # x = 0, r = p = vector, rr = r.r   (cg.py:34-36)
    rhs = be.cg_init(layout, vector, None if skip_x else x, r, p, keep_mask=keep_mask) if keep_mask is not None else \
        be.cg_init(layout, vector, None if skip_x else x, r, p)
    p_views = layout.views(p, vector)

    # a structured provider may leave a diagonal part of the Hessian (ridge) to the recurrence kernel
    shift = float(getattr(provider, "hvp_shift", 0.0)) if provider is not None else 0.0
    if fused is not None and alpha != 0.0:
        solve = fused(layout, x, r, p, K, alpha, rhs=rhs) if keep_mask is not None else fused(layout, x, r, p, K, alpha)
        if keep_mask is not None and not solve:   # (r / p were initialised for THAT solver only: the generic loop below must not run on them)
            raise RuntimeError("a provider that announced a state mask must run its fused solver")
    else:
        solve = False
    if solve:
        pass  # the provider's own kernels ran all K iterations (HVP outputs consumed on chip, no N-sized H p)
    else:
        for k in range(K):
            hvp = hvp_fn(p_views)  # H p   (cg.py:39-41)
            # cg.py:42-55 in one launch group; the last one also applies cg.py:56 and the negation
            last = k == K - 1 and alpha != 0.0
            be.cg_step(layout, hvp, x, r, p, alpha, k, out_scale=(-alpha if last else 0.0), hvp_shift=shift)
        if K > 0 and alpha == 0.0:
            be.scale_flat(x, -alpha)  # out_scale = 0 means "no final scaling" to the kernel: do cg.py:56 explicitly
        be.after_cg(layout)
    # K == 0: x is identically zero, -alpha * 0 needs no pass.

    neg_x = layout.views(x, vector)
    if provider is not None:
        if solve and solve is not True:   # a token: the provider is told WHICH solve these views name (see structured.py)
            return provider.mixed_vjp(neg_x, sync, solve=solve)
        return provider.mixed_vjp(neg_x, sync)
    if for_hvp:   # the mixed second derivative is one more forward-over-reverse pass (or the fallback's double backward)
        return hvp_fn.mixed(neg_x, sync)
    if keep_graph:   # the captured autograd graph of `in_grad` outlives the step (see PersistentOpaqueGraphs.saved_versions)
        return persist.mixed(prev, neg_x, sync)
    return mixed_vjp(in_grad, prev, neg_x, sync)
