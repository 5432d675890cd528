"""SAMA: Adam-preconditioned finite-difference best-response-Jacobian product on gfx950 kernels.

Behavioural twin of /root/reference betty/hypergradient/sama.py:7-61 with
``precondition`` from betty/hypergradient/utils.py:24-92: the direction is first multiplied by the
derivative of the inner optimizer's update w.r.t. the gradient (identity for SGD; the closed form
of utils.py:37-63 for Adam, built from ``exp_avg``, ``exp_avg_sq`` and the ``last_grad`` the problem
records in ``optimizer_step``), then the same central finite difference as ``darts`` is applied with
radius ``sama_adam_alpha``.  The preconditioner is ONE fused kernel over the four tensor lists
(20*N bytes) writing a flat vector; norm and weight perturbations reuse the darts kernels.
"""
from __future__ import annotations

import torch

from ..backend import get_backend
from .utils import grad, replace_none_with_zero


def _optimizer_kind(optimizer) -> str:
    """hypergradient/utils.py:24-30."""
    name = type(optimizer).__name__.lower()
    if "adam" in name:
        return "adam"
    if "rmsprop" in name:
        return "rmsprop"
    return "sgd"


def precondition(vector, problem, be):
    """Returns (layout, list of per-parameter tensors) of the preconditioned direction."""
    layout = be.layout(vector)
    kind = _optimizer_kind(problem.optimizer)
    if kind == "sgd":  # utils.py:33-34
        return layout, list(vector)
    if kind != "adam":
        raise NotImplementedError(f"SAMA preconditioning for {type(problem.optimizer).__name__} is not implemented!")
    params = problem.meta_trainable_parameters()
    opt = problem.optimizer
    group_of = {}
    for gi, group in enumerate(opt.param_groups):
        for p in group["params"]:
            group_of[id(p)] = gi
    out = layout.state(4)[3]  # a flat vector that does not alias the cg/neumann state
    out_views = layout.views(out, vector)
    # one launch per optimizer param group (hyper-parameters are per group); tensors of other groups
    # are masked out by giving them zero-length... simpler and exact: run the fused kernel per group
    # on the full list with that group's hyper-parameters and keep only that group's slices.
    groups = sorted({group_of[id(p)] for p in params})
    zeros = {}

    def state_or_zero(p, v, key):
        st = opt.state.get(p, {})
        t = st.get(key)
        if t is None:
            z = zeros.get(id(p))
            if z is None:
                z = torch.zeros_like(v)
                zeros[id(p)] = z
            return z
        return t

    last_grad = [state_or_zero(p, v, "last_grad") for p, v in zip(params, vector)]
    exp_avg = [state_or_zero(p, v, "exp_avg") for p, v in zip(params, vector)]
    exp_avg_sq = [state_or_zero(p, v, "exp_avg_sq") for p, v in zip(params, vector)]
    if len(groups) == 1:
        g = opt.param_groups[groups[0]]
        b1, b2 = g["betas"]
        be.sama_adam_precondition(layout, vector, last_grad, exp_avg, exp_avg_sq, out, b1, b2, g["eps"], g["lr"])
        return layout, out_views
    result = [None] * len(params)
    for gi in groups:
        g = opt.param_groups[gi]
        b1, b2 = g["betas"]
        be.sama_adam_precondition(layout, vector, last_grad, exp_avg, exp_avg_sq, out, b1, b2, g["eps"], g["lr"])
        for i, p in enumerate(params):
            if group_of[id(p)] == gi:
                result[i] = out_views[i].clone()
    return layout, result


def sama(vector, curr, prev, sync):
    config = curr.config
    be = get_backend()
    vector = list(vector)
    weights = [w.data for w in curr.meta_trainable_parameters()]
    upper = prev.trainable_parameters()

    layout, pv = precondition(vector, curr, be)  # sama.py:25
    eps32, eps64, _ = be.darts_eps(layout, pv, float(config.sama_adam_alpha))  # sama.py:26-27
    two_eps = (2.0 * eps64).to(torch.float32)

    be.axpy_multi(layout, weights, pv, eps32, 1.0)  # sama.py:29-30
    loss_p = curr.training_step_exec(curr.cur_batch)
    grad_p = replace_none_with_zero(grad(loss_p, upper, allow_unused=True), upper)
    if sync:
        prev.set_grads(upper, [-(g / two_eps) for g in grad_p])  # sama.py:34-36

    be.axpy_multi(layout, weights, pv, eps32, -2.0)  # sama.py:39-40
    loss_n = curr.training_step_exec(curr.cur_batch)
    if sync:
        torch.autograd.backward(loss_n / two_eps, inputs=upper)  # sama.py:42-43
        grad_n = None
    else:
        grad_n = replace_none_with_zero(grad(loss_n, upper, allow_unused=True), upper)

    if not config.sama_multitask:  # sama.py:51-53
        be.axpy_multi(layout, weights, pv, eps32, 1.0)
    else:  # sama.py:54-55: average the (deliberately un-restored) weights over ranks
        curr.synchronize_params(curr.meta_trainable_parameters(), all_reduce=True)

    if sync:
        return None
    return [(gn - gp) / two_eps for gn, gp in zip(grad_n, grad_p)]  # sama.py:57-59
