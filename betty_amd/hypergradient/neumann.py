"""Neumann-series best-response-Jacobian product on fused gfx950 kernels.

Behavioural twin of /root/reference betty/hypergradient/neumann.py:8-66: ``p = v``; K times
``v <- v - alpha*Hv ; p <- p + v``; result ``alpha*p`` pushed through the mixed second derivative.
Each iteration is ONE streaming kernel (20*N bytes: read Hv, v, p; write v, p) instead of 4*T
ATen launches; the final ``alpha*p`` and the negation are folded into the last iteration.
"""
from __future__ import annotations

from ..backend import get_backend
from ._common import (AutogradHVP, ForwardOverReverseHVP, GraphedHVP, forward_over_reverse_wanted, hvp_graph_wanted, inner_gradient,
                      mixed_vjp, persistent_graphs_for, solve_stream)
from .structured import structured_hvp_for


def neumann(vector, curr, prev, sync):
    assert len(curr.paths) == 0, "neumann method is not supported for higher order MLO!"
    vector = list(vector)
    provider = structured_hvp_for(curr, prev)
    K = int(curr.config.neumann_iterations)
    # opaque double backward: captured once per solve, replayed as a HIP graph (see _common.GraphedHVP)
    graphed = (provider is None or getattr(provider, "hvp_is_autograd", False)) and hvp_graph_wanted(K, vector, curr)
    persist = persistent_graphs_for(curr, K, vector, prev) if provider is None else None   # see cg.py
    with solve_stream(vector[0].device if vector else None, graphed or persist is not None):
        return _neumann(vector, curr, prev, sync, provider, K, graphed, persist)


def _neumann(vector, curr, prev, sync, provider, K, graphed, persist=None):
    config = curr.config
    be = get_backend()
    layout = be.layout(vector)
    v, p = layout.state(2)
    keep_graph = for_hvp = False
    if provider is None:
        # neumann.py:39 differentiates w.r.t. trainable_parameters() (cg uses parameters())
        if persist is not None:
            in_grad, hvp_fn, keep_graph = persist.begin_step(curr, list(curr.trainable_parameters()), layout.views(v, vector), prev)
        elif forward_over_reverse_wanted(curr):
            # opt-in: H v by forward-over-reverse passes (no double-backward graph; _common.ForwardOverReverseHVP)
            in_grad, hvp_fn, for_hvp = None, ForwardOverReverseHVP(curr, prev), True
        else:
            in_grad = inner_gradient(curr)
            hvp_fn = AutogradHVP(in_grad, curr.trainable_parameters())
    else:
        in_grad = None
        hvp_fn = provider.prepare()
    if graphed and persist is None and not for_hvp:
        hvp_fn = GraphedHVP(hvp_fn)

    alpha = float(config.neumann_alpha)
    fused = getattr(provider, "fused_neumann", None)
    # a provider whose fused solver derives the mixed derivative from batch-sized factors never touches the accumulator
    skips = getattr(provider, "fused_neumann_skips_solution", None)
    skip_p = bool(fused is not None and alpha != 0.0 and skips is not None and skips(layout, K))
    be.neumann_init(layout, vector, v, None if skip_p else p)  # p = v   (neumann.py:60)
    v_views = layout.views(v, vector)

    shift = float(getattr(provider, "hvp_shift", 0.0)) if provider is not None else 0.0
    solve = fused(layout, v, p, K, alpha) if (fused is not None and alpha != 0.0) else False
    if solve:
        pass  # the provider's own kernels ran all K iterations (v ping-pongs with a third flat vector of the layout)
    else:
        for k in range(K):
            hvp = hvp_fn(v_views)  # neumann.py:62
            last = k == K - 1 and alpha != 0.0
            be.neumann_step(layout, hvp, v, p, alpha, out_scale=(-alpha if last else 0.0), hvp_shift=shift)  # 63-64 (+66)
        if K == 0 or alpha == 0.0:
            be.scale_flat(p, -alpha)  # alpha * p (with p = v when K == 0)   (neumann.py:66)

    neg_p = layout.views(p, vector)
    if provider is not None:
        if solve and solve is not True:   # a token: the provider is told WHICH solve these views name (see structured.py)
            return provider.mixed_vjp(neg_p, sync, solve=solve)
        return provider.mixed_vjp(neg_p, sync)
    if for_hvp:   # the mixed second derivative is one more forward-over-reverse pass (or the fallback's double backward)
        return hvp_fn.mixed(neg_p, sync)
    if keep_graph:   # the captured autograd graph of `in_grad` outlives the step (see PersistentOpaqueGraphs.saved_versions)
        return persist.mixed(prev, neg_p, sync)
    return mixed_vjp(in_grad, prev, neg_p, sync)
