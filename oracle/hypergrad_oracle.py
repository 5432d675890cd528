"""CPU oracle for the hypergradient hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file, and only as the checker.  ``betty_amd`` never imports it and has no CPU fallback.

What it is: a restatement, in plain per-tensor PyTorch-CPU arithmetic, of the algorithm of
leopard-ai/betty v0.2.1's implicit-differentiation path.  The arithmetic of that path lives in a
third-party dependency that is not under /root/reference — **PyTorch autograd / ATen**
(``torch>=1.8.0``, requirements/requirements.txt:1; present here as torch 2.10.0) — so the
restatement calls the same autograd primitives at the same call sites and reproduces the
reference's op order exactly (scale-then-``cat``-then-``dot``; ``a*b`` rounded before ``+``/``-``).

Parity pin: the reference's own tests hold no golden vectors for this path (only the loose
``loss < 0.48`` thresholds of test/test_regression.py:126,151,176), so the oracle is pinned
against outputs of the reference itself: ``tests/golden/make_golden.py`` imports the real
``betty.hypergradient`` from /root/reference in the build container, runs it on the seeded
problems of ``tests/zoo.py`` and commits inputs + outputs under ``tests/golden/``;
``tests/test_oracle.py`` requires this file to reproduce them **bit for bit** in fp32 and fp64.

Each function cites the reference lines it follows.
"""
from __future__ import annotations

import warnings

import torch


# -- betty/utils.py:117-137 ---------------------------------------------------------------------
def flat_scaled(tensors, alpha=1.0):
    """``to_vec`` (utils.py:117-118): scale every tensor, reshape to 1-D, concatenate."""
    pieces = []
    for t in tensors:
        pieces.append(alpha * t.reshape(-1))
    return torch.cat(pieces)


def negate_or_none(t):
    """``neg_with_none`` (utils.py:125-129)."""
    if t is None:
        return None
    return -t


def zeros_for_missing(tensors, reference):
    """``replace_none_with_zero`` (utils.py:132-137)."""
    filled = []
    for t, ref in zip(tensors, reference):
        filled.append(torch.zeros_like(ref) if t is None else t)
    return tuple(filled)


# -- betty/hypergradient/utils.py:5-21 (non-FSDP branch, 18-21) -----------------------------------
def first_order_grad(loss, parameters, retain_graph=False, allow_unused=False, through_grad_field=False):
    """hypergradient/utils.py:5-21.  ``through_grad_field`` is the FSDP branch (9-17): the gradient is read
    off ``.grad`` around a ``backward`` and the previous ``.grad`` is put back."""
    if through_grad_field:
        def current(p):
            return p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)

        saved = [current(p) for p in parameters]
        torch.autograd.backward(loss, retain_graph=retain_graph, inputs=parameters)
        deltas = []
        for p, old in zip(parameters, saved):
            deltas.append(current(p) - old)
            p.grad.copy_(old.data)
        return deltas
    return torch.autograd.grad(loss, parameters, retain_graph=retain_graph, allow_unused=allow_unused)


def _inner_gradient(curr):
    """cg.py:27-32 / neumann.py:31-36: inner loss on the stored batch, gradient with graph."""
    loss = curr.training_step_exec(curr.cur_batch)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return torch.autograd.grad(loss, curr.trainable_parameters(), create_graph=True)


def _to_upper(in_grad, prev, direction, sync):
    """cg.py:58-68 / neumann.py:44-54: mixed second derivative applied to ``direction``."""
    if sync:
        negated = [negate_or_none(d) for d in direction]
        torch.autograd.backward(in_grad, inputs=prev.trainable_parameters(), grad_tensors=negated)
        return None
    out = torch.autograd.grad(in_grad, prev.trainable_parameters(), grad_outputs=direction)
    return [negate_or_none(o) for o in out]


# -- betty/hypergradient/cg.py:8-70 ------------------------------------------------------------------
def cg(vector, curr, prev, sync):
    if len(curr.paths) != 0:  # cg.py:25
        raise AssertionError("cg method is not supported for higher order MLO!")
    cfg = curr.config
    in_grad = _inner_gradient(curr)

    # cg.py:34-36
    sol = [torch.zeros_like(v) for v in vector]
    res = [torch.zeros_like(v).copy_(v) for v in vector]
    dirn = [torch.zeros_like(q).copy_(q) for q in res]

    for _ in range(cfg.cg_iterations):  # cg.py:38
        hd = torch.autograd.grad(in_grad, curr.parameters(), grad_outputs=dirn, retain_graph=True)  # 39-41
        hd_flat = flat_scaled(hd, alpha=cfg.cg_alpha)  # 42: scaled copy used ONLY in the denominator
        res_flat = flat_scaled(res)  # 43
        dir_flat = flat_scaled(dirn)  # 44
        num = torch.dot(res_flat, res_flat)  # 45
        den = torch.dot(hd_flat, dir_flat)  # 46
        step = num / den  # 47

        sol_next = [s + step * d for s, d in zip(sol, dirn)]  # 49
        res_next = [q - step * h for q, h in zip(res, hd)]  # 50: un-scaled Hp (reference quirk)
        res_next_flat = flat_scaled(res_next)  # 51
        ratio = torch.dot(res_next_flat, res_next_flat) / num  # 52
        dir_next = [q + ratio * d for q, d in zip(res_next, dirn)]  # 53

        sol, dirn, res = sol_next, dir_next, res_next  # 55
    sol = [cfg.cg_alpha * s for s in sol]  # 56
    return _to_upper(in_grad, prev, sol, sync)


# -- betty/hypergradient/neumann.py:8-66 ----------------------------------------------------------------
def neumann_series(v, in_grad, params, iterations, alpha):
    """``approx_inverse_hvp`` (neumann.py:59-66)."""
    acc = v  # 60
    for _ in range(iterations):
        hv = torch.autograd.grad(in_grad, params, grad_outputs=v, retain_graph=True)  # 62
        v = [vi - alpha * hi for vi, hi in zip(v, hv)]  # 63
        acc = [vi + ai for vi, ai in zip(v, acc)]  # 64
    return [alpha * ai for ai in acc]  # 66


def neumann(vector, curr, prev, sync):
    if len(curr.paths) != 0:  # neumann.py:29
        raise AssertionError("neumann method is not supported for higher order MLO!")
    cfg = curr.config
    in_grad = _inner_gradient(curr)
    series = neumann_series(vector, in_grad, curr.trainable_parameters(), cfg.neumann_iterations, cfg.neumann_alpha)
    return _to_upper(in_grad, prev, series, sync)


# -- betty/hypergradient/darts.py:8-69 (non-FSDP) ----------------------------------------------------------
def darts(vector, curr, prev, sync):
    cfg = curr.config
    radius = cfg.darts_alpha  # 29
    norm = flat_scaled(vector).norm()  # 30
    if getattr(curr, "_strategy", "default") == "fsdp":  # 31-34: the vector is sharded over the ranks
        import torch.distributed as dist

        total = norm.pow(2)
        dist.all_reduce(total, op=dist.ReduceOp.SUM)
        norm = total.sqrt()
    eps = radius / norm.add_(1e-15).item()  # 35

    for w, v in zip(curr.meta_trainable_parameters(), vector):  # 37-38
        w.data.add_(v.data, alpha=eps)
    loss_plus = curr.training_step_exec(curr.cur_batch)  # 39
    g_plus = first_order_grad(loss_plus, prev.trainable_parameters(), allow_unused=True)  # 40-42
    g_plus = zeros_for_missing(g_plus, prev.trainable_parameters())  # 43
    if sync:  # 44-46
        g_plus = [-g.div_(2 * eps) for g in g_plus]
        prev.set_grads(prev.trainable_parameters(), g_plus)

    for w, v in zip(curr.meta_trainable_parameters(), vector):  # 49-50
        w.data.sub_(v.data, alpha=2 * eps)
    loss_minus = curr.training_step_exec(curr.cur_batch)  # 51
    if sync:  # 52-53
        torch.autograd.backward(loss_minus / (2 * eps), inputs=prev.trainable_parameters())
        g_minus = None
    else:  # 55-58
        g_minus = first_order_grad(loss_minus, prev.trainable_parameters(), allow_unused=True)
        g_minus = zeros_for_missing(g_minus, prev.trainable_parameters())

    if not cfg.darts_multitask:  # 61-63
        for w, v in zip(curr.meta_trainable_parameters(), vector):
            w.data.add_(v.data, alpha=eps)

    if sync:
        return None
    return [(a - b).div_(2 * eps) for a, b in zip(g_minus, g_plus)]  # 65-67


# -- betty/hypergradient/utils.py:24-92 + sama.py:7-61 -------------------------------------------------------
def _optimizer_family(optimizer):
    """get_optimzer_type (utils.py:24-30)."""
    name = type(optimizer).__name__.lower()
    if "adam" in name:
        return "adam"
    if "rmsprop" in name:
        return "rmsprop"
    return "sgd"


def adam_preconditioned(vectors, problem):
    """precondition_adam (utils.py:37-63): d(Adam update)/d(gradient) applied to ``vectors``."""
    result = []
    for vec, param in zip(vectors, problem.meta_trainable_parameters()):
        group = problem.get_opt_param_group_for_param(param)
        state = problem.get_opt_state_for_param(param)
        with torch.no_grad():
            b1, b2 = group["betas"]
            eps = group["eps"]
            g = state.get("last_grad", torch.zeros_like(vec))
            m = state.get("exp_avg", torch.zeros_like(vec))
            u = state.get("exp_avg_sq", torch.zeros_like(vec))
            m_prev = (m - (1 - b1) * g) / b1 if b1 != 0 else 0  # 50-52
            u_prev = (u - (1 - b2) * g * g) / b2  # 53
            factor = (1 - b1) * b2 * u_prev - b1 * (1 - b2) * g * m_prev  # 55-57
            factor /= (torch.sqrt(u) + eps) ** 3  # 58
        result.append(vec * factor * group["lr"])  # 59
    return result


def preconditioned(vectors, problem):
    """precondition (utils.py:86-92)."""
    family = _optimizer_family(problem.optimizer)
    if family == "sgd":
        return vectors
    if family == "adam":
        return adam_preconditioned(vectors, problem)
    raise NotImplementedError(f"SAMA preconditioning for {family} is not implemented!")


def sama(vector, curr, prev, sync):
    cfg = curr.config
    radius = cfg.sama_adam_alpha  # 23
    vector = preconditioned(vector, curr)  # 25
    norm = flat_scaled(vector).norm()  # 26
    eps = radius / norm.add_(1e-15).item()  # 27

    for w, v in zip(curr.meta_trainable_parameters(), vector):  # 29-30
        w.data.add_(v.data, alpha=eps)
    loss_plus = curr.training_step_exec(curr.cur_batch)
    g_plus = torch.autograd.grad(loss_plus, prev.trainable_parameters(), allow_unused=True)  # 32
    g_plus = zeros_for_missing(g_plus, prev.trainable_parameters())
    if sync:  # 34-36
        g_plus = [-g.div_(2 * eps) for g in g_plus]
        prev.set_grads(prev.trainable_parameters(), g_plus)

    for w, v in zip(curr.meta_trainable_parameters(), vector):  # 39-40
        w.data.sub_(v.data, alpha=2 * eps)
    loss_minus = curr.training_step_exec(curr.cur_batch)
    if sync:  # 42-43
        torch.autograd.backward(loss_minus / (2 * eps), inputs=prev.trainable_parameters())
        g_minus = None
    else:  # 45-48
        g_minus = torch.autograd.grad(loss_minus, prev.trainable_parameters(), allow_unused=True)
        g_minus = zeros_for_missing(g_minus, prev.trainable_parameters())

    if not cfg.sama_multitask:  # 51-53
        for w, v in zip(curr.meta_trainable_parameters(), vector):
            w.data.add_(v.data, alpha=eps)
    else:  # 54-55
        curr.synchronize_params(curr.meta_trainable_parameters(), all_reduce=True)

    if sync:
        return None
    return [(a - b).div_(2 * eps) for a, b in zip(g_minus, g_plus)]  # 57-59


JVP_FNS = {"cg": cg, "neumann": neumann, "darts": darts, "sama": sama}


# -- betty/hypergradient/__init__.py:22-39 ------------------------------------------------------------------
def get_grads(loss, path, retain_graph, do_sync):
    lower = path[1].meta_trainable_parameters()
    sharded = getattr(path[0], "_strategy", "default") == "fsdp"  # 23
    jvp = first_order_grad(loss, lower, retain_graph=retain_graph, allow_unused=True, through_grad_field=sharded)  # 24-30
    jvp = zeros_for_missing(jvp, lower)  # 31
    hops = len(path) - 1
    for i in range(1, hops):  # 32
        fn = JVP_FNS[path[i].config.type]  # 33-35
        jvp = fn(jvp, path[i], path[i + 1], bool(do_sync and i == hops - 1))  # 36-37
    return jvp
