"""Shared pieces of cg/neumann: inner-gradient graph, HVP providers, final mixed VJP."""
from __future__ import annotations

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


def mixed_vjp(in_grad, prev, neg_x_views: List[torch.Tensor], sync: bool):
    """Final hop to the upper parameters (cg.py:58-68, neumann.py:44-54).

    ``neg_x_views`` already holds ``-(alpha * x)``; by linearity of the VJP
    ``-(d(g.x)/d lambda) == d(g.(-x))/d lambda`` bit for bit, so no extra negation pass is needed.
    ``sync=True`` accumulates into ``.grad`` through ``backward`` (DDP reducer hooks fire) and
    returns None; ``sync=False`` returns the list for ``Problem.set_grads``."""
    upper = prev.trainable_parameters()
    if sync:
        torch.autograd.backward(in_grad, inputs=upper, grad_tensors=neg_x_views)
        return None
    return list(torch.autograd.grad(in_grad, upper, grad_outputs=neg_x_views))
