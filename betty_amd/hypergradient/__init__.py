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

jvp_fn_mapping = {
    "darts": darts,
    "sama": sama,
    "neumann": neumann,
    "cg": cg,
}


def get_grads(loss, path, retain_graph, do_sync):
    """Hypergradient of ``loss`` along ``path`` = [upper, lower_1, ..., upper]
    (__init__.py:22-39): direct gradient w.r.t. the first lower problem's parameters, then one
    best-response-Jacobian product per hop, right to left; only the last hop may sync."""
    is_fsdp = getattr(path[0], "_strategy", "default") == "fsdp"   # __init__.py:23
    lower = path[1].meta_trainable_parameters()
    jvp = grad(loss, lower, retain_graph=retain_graph, allow_unused=True, is_fsdp=is_fsdp)
    jvp = replace_none_with_zero(jvp, lower)
    for i in range(1, len(path) - 1):
        jvp_fn_type = path[i].config.type
        assert jvp_fn_type in jvp_fn_mapping
        jvp_fn = jvp_fn_mapping[jvp_fn_type]
        sync = bool(do_sync and i == len(path) - 2)
        jvp = jvp_fn(jvp, path[i], path[i + 1], sync)
    return jvp
