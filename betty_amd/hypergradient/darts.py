"""Central finite-difference best-response-Jacobian product ("darts") on gfx950 kernels.

Behavioural twin of /root/reference betty/hypergradient/darts.py:8-69.  The vector work —
``||v||``, the three in-place perturbations of the live inner weights — is four multi-tensor
launches (40*N bytes) instead of ``cat`` + ``norm`` + 3*T ``add_`` calls, and ``eps`` stays on
the device (the reference synchronises the host with ``.item()``, darts.py:35).
"""
from __future__ import annotations

import torch

from ..backend import get_backend
from .utils import grad, replace_none_with_zero


def darts(vector, curr, prev, sync):
    config = curr.config
    is_fsdp = getattr(curr, "_strategy", "default") == "fsdp"
    be = get_backend()
    vector = list(vector)
    weights = [w.data for w in curr.meta_trainable_parameters()]
    upper = prev.trainable_parameters()

    layout = be.layout(vector)
    # eps = R / (||v|| + 1e-15)   (darts.py:29-35), 0-dim device tensors
    eps32, eps64, sumsq = be.darts_eps(layout, vector, float(config.darts_alpha))
    if is_fsdp:
        # darts.py:31-34: every rank holds a shard of v; eps uses the norm of the whole vector.  Same fp32
        # steps as the reference: local norm -> square -> all-reduce(SUM) -> sqrt -> + 1e-15 -> R / .
        import torch.distributed as dist

        sq = sumsq.sqrt().to(torch.float32).pow(2)
        dist.all_reduce(sq, op=dist.ReduceOp.SUM)
        norm = sq.sqrt().add_(1e-15)
        eps64 = float(config.darts_alpha) / norm.to(torch.float64)
        eps32 = eps64.to(torch.float32)
    two_eps = (2.0 * eps64).to(torch.float32)  # the reference divides fp32 tensors by the Python float 2*eps

    # w <- w + eps*v   (darts.py:37-38)
    be.axpy_multi(layout, weights, vector, eps32, 1.0)
    loss_p = curr.training_step_exec(curr.cur_batch)
    # is_fsdp: the gradient of a flat shard only materialises through backward into .grad (darts.py:40-42, utils.py:9-17)
    grad_p = replace_none_with_zero(grad(loss_p, upper, allow_unused=True, is_fsdp=is_fsdp), upper)
    if sync:
        # darts.py:44-46: -g+/(2 eps) goes straight into .grad
        prev.set_grads(upper, [-(g / two_eps) for g in grad_p])

    # w <- w - 2*eps*v   (darts.py:49-50)
    be.axpy_multi(layout, weights, vector, eps32, -2.0)
    loss_n = curr.training_step_exec(curr.cur_batch)
    if sync:
        torch.autograd.backward(loss_n / two_eps, inputs=upper)  # darts.py:52-53 (DDP hooks fire)
        grad_n = None
    else:
        grad_n = replace_none_with_zero(grad(loss_n, upper, allow_unused=True, is_fsdp=is_fsdp), upper)  # darts.py:55-58

    # restore w   (darts.py:61-63)
    if not config.darts_multitask:
        be.axpy_multi(layout, weights, vector, eps32, 1.0)

    if sync:
        return None
    return [(gn - gp) / two_eps for gn, gp in zip(grad_n, grad_p)]  # darts.py:65-67
