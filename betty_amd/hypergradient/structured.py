"""Opt-in analytic HVP providers for structured inner problems.

The reference always differentiates twice through the user's opaque ``training_step``.  When the
inner problem *declares* a structure the backend knows (logistic regression; ReLU-MLP with
weighted cross-entropy), the HVP and the mixed second derivative are computed by dedicated HIP
kernels instead.  A problem opts in by exposing ``hypergradient_structure(prev)`` returning a
provider; problems that do not are handled by autograd exactly as in the reference.
"""
from __future__ import annotations


def structured_hvp_for(curr, prev):
    hook = getattr(curr, "hypergradient_structure", None)
    if hook is None:
        return None
    return hook(prev)
