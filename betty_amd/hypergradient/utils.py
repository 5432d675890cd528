"""Small helpers of the hypergradient path (reference: betty/hypergradient/utils.py:5-21,
betty/utils.py:125-137)."""
from __future__ import annotations

import torch


def grad(loss, parameters, retain_graph=False, allow_unused=False, is_fsdp=False):
    """First-order gradient of ``loss`` w.r.t. ``parameters`` as a tuple (entries may be None
    with ``allow_unused``).  Mirrors hypergradient/utils.py:18-21; the FSDP branch (9-17) is
    out of scope (the reference calls FSDP experimental) and raises."""
    if is_fsdp:
        raise NotImplementedError("betty_amd: the FSDP strategy is out of scope for the MI355X backend")
    return torch.autograd.grad(loss, parameters, retain_graph=retain_graph, allow_unused=allow_unused)


def replace_none_with_zero(tensor_list, reference):
    """betty/utils.py:132-137."""
    return tuple(t if t is not None else torch.zeros_like(r) for t, r in zip(tensor_list, reference))


def neg_with_none(a):
    """betty/utils.py:125-129."""
    return None if a is None else -a
