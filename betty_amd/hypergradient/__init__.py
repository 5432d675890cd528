"""``betty_amd.hypergradient`` — drop-in for ``betty.hypergradient`` on MI355X.

Same surface as /root/reference betty/hypergradient/__init__.py:13-39: a ``jvp_fn_mapping``
registry keyed by ``Config.type`` and ``get_grads(loss, path, retain_graph, do_sync)``.
"""
from __future__ import annotations

from .cg import cg
from .darts import darts
from .neumann import neumann
from .sama import sama
from .utils import grad, replace_none_with_zero

def reinforce(vector, curr, prev, sync):
    """The reference registers ``reinforce`` (__init__.py:18) but its function is a stub with the wrong arity
    (reinforce.py:6 takes no ``sync``) and returns None: any run that selects it fails inside get_grads.  The key is
    kept so that ``Config(type="reinforce")`` is accepted where the reference accepts it, and fails with a message."""
    raise NotImplementedError("Config.type='reinforce' is an unimplemented stub in the reference "
                              "(betty/hypergradient/reinforce.py:6-7); use darts | sama | neumann | cg")


def cg_global(vector, curr, prev, sync):
    """Extension key (not in the reference): ``Config(type="cg_global")`` solves ONE inner problem whose batch is spread
    over the default process group — data-parallel HVP, sharded CG state (betty_amd/global_hvp.py)."""
    from ..global_hvp import cg_global as impl  # noqa: PLC0415

    return impl(vector, curr, prev, sync)


def neumann_global(vector, curr, prev, sync):
    """Extension key (not in the reference): ``Config(type="neumann_global")`` — the Neumann series on ONE inner problem whose batch is
    spread over the default process group, in the factor-exchange form (one all-gather of batch-sized factors per iteration, no reduction;
    betty_amd/global_hvp.py)."""
    from ..global_hvp import neumann_global as impl  # noqa: PLC0415

    return impl(vector, curr, prev, sync)


jvp_fn_mapping = {
    "darts": darts,
    "sama": sama,
    "neumann": neumann,
    "cg": cg,
    "reinforce": reinforce,
    "cg_global": cg_global,
    "neumann_global": neumann_global,
}


def get_grads(loss, path, retain_graph, do_sync):
    """Hypergradient of ``loss`` along ``path`` = [upper, lower_1, ..., upper] (__init__.py:22-39).

    The direct gradient w.r.t. the first lower problem's parameters is pushed through the path one hop at a
    time: hop ``(lower, nxt)`` multiplies it with the best-response Jacobian of ``lower`` w.r.t. ``nxt`` using
    the approximation named by ``lower.config.type``.  Only the LAST hop may synchronise (``do_sync``): it then
    accumulates into ``nxt``'s ``.grad`` through ``backward`` (DDP reducer) and the function returns None."""
    upper, first_lower = path[0], path[1]
    targets = first_lower.meta_trainable_parameters()
    vector = grad(loss, targets, retain_graph=retain_graph, allow_unused=True,
                  is_fsdp=getattr(upper, "_strategy", "default") == "fsdp")
    vector = replace_none_with_zero(vector, targets)
    hops = list(zip(path[1:-1], path[2:]))          # (lower, next-on-the-way-up)
    for n, (lower, nxt) in enumerate(hops):
        kind = lower.config.type
        assert kind in jvp_fn_mapping
        last = n == len(hops) - 1
        vector = jvp_fn_mapping[kind](vector, lower, nxt, bool(do_sync and last))
    return vector
