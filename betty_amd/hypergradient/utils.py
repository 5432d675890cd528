"""Small helpers of the hypergradient path (reference: betty/hypergradient/utils.py:5-21,
betty/utils.py:132-137)."""
from __future__ import annotations

import torch


def _grad_or_zero(p):
    return p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)


def grad(loss, parameters, retain_graph=False, allow_unused=False, is_fsdp=False):
    """First-order gradient of ``loss`` w.r.t. ``parameters`` (entries may be None with ``allow_unused``),
    hypergradient/utils.py:5-21.  Under FSDP (``is_fsdp``) the parameters are flat shards whose gradient
    only materialises through ``backward``: snapshot ``.grad``, backward into it, return the difference and
    put the snapshot back (utils.py:9-17)."""
    if is_fsdp:
        parameters = list(parameters)
        before = [_grad_or_zero(p) for p in parameters]
        torch.autograd.backward(loss, retain_graph=retain_graph, inputs=parameters)
        grads = []
        for p, g0 in zip(parameters, before):
            grads.append(_grad_or_zero(p) - g0)
            if p.grad is not None:
                p.grad.copy_(g0)
        return grads
    return torch.autograd.grad(loss, parameters, retain_graph=retain_graph, allow_unused=allow_unused)


def replace_none_with_zero(tensor_list, reference):
    """betty/utils.py:132-137."""
    return tuple(t if t is not None else torch.zeros_like(r) for t, r in zip(tensor_list, reference))
