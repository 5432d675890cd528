"""The CPU oracle (oracle/hypergrad_oracle.py) against the goldens produced by the REAL
reference (tests/golden/make_golden.py) — bit for bit, fp32 and fp64, sync and non-sync."""
import os
import sys

import numpy as np
import pytest
import torch

import zoo
from conftest import golden_list, load_golden, rel_err

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import hypergrad_oracle as orc  # noqa: E402

from betty_amd import Config  # noqa: E402


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)  # goldens were generated single-threaded (fixed reduction order)
    yield
    torch.set_num_threads(n)


def _run(case, inputs, dtype, sync):
    curr, prev, vector = zoo.build_case(case, inputs, Config, device="cpu", dtype=dtype)
    out = orc.JVP_FNS[case.algo](vector, curr, prev, sync)
    if sync:
        assert out is None
        return [p.grad.detach().numpy() if p.grad is not None else np.zeros(p.shape) for p in prev.trainable_parameters()], curr
    return [o.detach().numpy() for o in out], curr


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_oracle_reproduces_reference_fp32(case):
    inputs, outputs = load_golden(case.family)
    got, curr = _run(case, inputs, torch.float32, False)
    want = golden_list(outputs, case.name, "fp32")
    assert len(got) == len(want)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    if case.algo in ("darts", "sama"):
        for p, w in zip(curr.trainable_parameters(), golden_list(outputs, case.name, "w32")):
            np.testing.assert_array_equal(p.data.numpy(), w)


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_oracle_reproduces_reference_fp64(case):
    inputs, outputs = load_golden(case.family)
    got, _ = _run(case, inputs, torch.float64, False)
    for g, w in zip(got, golden_list(outputs, case.name, "fp64")):
        np.testing.assert_array_equal(g, w)


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_oracle_sync_accumulates_into_grad(case):
    inputs, outputs = load_golden(case.family)
    got, _ = _run(case, inputs, torch.float32, True)
    for g, w in zip(got, golden_list(outputs, case.name, "sync32")):
        np.testing.assert_array_equal(g, w)


def test_golden_conditioning():
    """fp32 and fp64 reference runs agree far inside the parity tolerance, so rtol 1e-4 is a
    meaningful bar for the cg/neumann cases (finite differences are noisier, see Case.rtol)."""
    for case in zoo.CASES:
        _, outputs = load_golden(case.family)
        a, b = golden_list(outputs, case.name, "fp32"), golden_list(outputs, case.name, "fp64")
        if np.linalg.norm(np.concatenate([x.ravel() for x in b])) == 0:
            continue
        rel, _ = rel_err(a, b)
        assert rel < (5e-4 if case.algo in ("darts", "sama") else 5e-6), (case.name, rel)


def test_logreg_closed_form():
    """Known-answer check: for logistic regression the exact CG solution
    of (X^T S X + diag(lam)) x = v pushed through the mixed derivative is  -w .* x ."""
    case = zoo.CASE_BY_NAME["logreg_cg5"]
    inputs, outputs = load_golden("logreg")
    X = inputs["batch_x"].astype(np.float64)
    w = inputs["inner_0"].astype(np.float64)
    lam = inputs["upper_0"].astype(np.float64)
    v = inputs["vec_0"].astype(np.float64)
    z = X @ w
    s = 1 / (1 + np.exp(-z))
    H = X.T @ (X * (s * (1 - s) / len(z))[:, None]) + np.diag(lam)
    # 5 CG iterations in exact arithmetic (cg_alpha = 1 => textbook CG)
    x = np.zeros_like(v); r = v.copy(); p = v.copy()
    for _ in range(5):
        Hp = H @ p
        a = (r @ r) / (Hp @ p)
        x = x + a * p
        rn = r - a * Hp
        p = rn + (rn @ rn) / (r @ r) * p
        r = rn
    want = -(w * x)
    got = golden_list(outputs, case.name, "fp64")[0]
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-14)


@pytest.mark.skipif(not os.path.isdir("/root/reference/betty"), reason="reference checkout not present")
def test_oracle_matches_live_reference_on_fresh_seed():
    """Beyond the committed goldens: a freshly seeded problem through the live reference."""
    sys.path.insert(0, "/root/reference")
    try:
        import betty.hypergradient  # noqa: F401
        from betty.configs import Config as RefConfig
        ref = {k: getattr(sys.modules[f"betty.hypergradient.{k}"], k) for k in ("cg", "neumann", "darts", "sama")}
    finally:
        sys.path.remove("/root/reference")
    for case in zoo.CASES:
        if case.family != "reweight":
            continue
        inputs = zoo.seed_family_inputs(case.family, seed=7)
        c1, p1, v1 = zoo.build_case(case, inputs, RefConfig)
        c2, p2, v2 = zoo.build_case(case, inputs, Config)
        a = ref[case.algo](v1, c1, p1, False)
        b = orc.JVP_FNS[case.algo](v2, c2, p2, False)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.detach().numpy(), y.detach().numpy())


def _fsdp_grad_case():
    torch.manual_seed(3)
    lin = torch.nn.Linear(6, 4)
    x = torch.randn(5, 6)
    params = list(lin.parameters())
    # a pre-existing .grad that must survive untouched (utils.py:10,16)
    for p in params:
        p.grad = torch.randn_like(p)
    keep = [p.grad.clone() for p in params]
    loss = (lin(x) ** 2).sum()
    return params, keep, loss


def test_oracle_fsdp_grad_branch():
    """hypergradient/utils.py:9-17 (gradient read off `.grad` around a backward, `.grad` restored): the oracle's
    restatement equals the plain autograd gradient, leaves `.grad` as it was — and equals the live reference's
    `grad(..., is_fsdp=True)` bit for bit when the checkout is present."""
    params, keep, loss = _fsdp_grad_case()
    plain = torch.autograd.grad(loss, params, retain_graph=True)
    got = orc.first_order_grad(loss, params, retain_graph=True, through_grad_field=True)
    for g, w in zip(got, plain):
        np.testing.assert_allclose(g.numpy(), w.numpy(), rtol=1e-6, atol=1e-7)   # (grad + g) - grad rounds once
    for p, k in zip(params, keep):
        np.testing.assert_array_equal(p.grad.numpy(), k.numpy())
    if os.path.isdir("/root/reference/betty"):
        sys.path.insert(0, "/root/reference")
        try:
            from betty.hypergradient.utils import grad as ref_grad
        finally:
            sys.path.remove("/root/reference")
        params2, _, loss2 = _fsdp_grad_case()
        want = ref_grad(loss2, params2, retain_graph=True, is_fsdp=True)
        for g, w in zip(got, want):
            np.testing.assert_array_equal(g.numpy(), w.numpy())


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_case_tolerances_follow_the_golden_spread(case):
    """Case.rtol = max(1e-4, ~5 x the reference's own fp32-vs-fp64 distance on that case): no blanket tolerance."""
    _, outputs = load_golden(case.family)
    a, b = golden_list(outputs, case.name, "fp32"), golden_list(outputs, case.name, "fp64")
    if not b:
        pytest.skip("no fp64 golden")
    spread, _ = rel_err(a, b)
    want = max(1e-4, 5.0 * spread)
    assert want <= case.rtol <= 1.1 * want, (case.name, case.rtol, spread)
