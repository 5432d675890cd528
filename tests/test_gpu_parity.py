"""GPU parity tests proper: the HIP path (through the C ABI) against
  * the goldens produced by the real reference (fp32, tolerance per case, 1e-4 for cg/neumann),
  * the C oracle, kernel by kernel, on ragged multi-tensor inputs,
  * closed forms at BASELINE.json's full size (N = 10,034,826) where the oracle would be slow.
Tolerances are written next to each assertion.
"""
import contextlib
import ctypes
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import zoo
from conftest import golden_list, load_golden, rel_err

from betty_amd import Config, _native
from betty_amd import hypergradient as hg
from betty_amd.backend import get_backend

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def be():
    from betty_amd.backend import get_backend

    b = get_backend()
    assert b.name == "hip"
    return b


@pytest.fixture(scope="module")
def orc():
    from _cpu_checker_backend import load_oracle_lib

    return load_oracle_lib()


def _np(ts):
    return [t.detach().cpu().numpy() for t in ts]


# ------------------------------------------------------------------------------------------------
# end to end against the reference goldens
# ------------------------------------------------------------------------------------------------
VARIANTS = {"auto": _native.BHG_CG_AUTO, "stream": _native.BHG_CG_STREAM, "resident": _native.BHG_CG_RESIDENT}


@pytest.mark.parametrize("variant", ["stream", "resident"])
@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_hip_path_matches_reference(case, variant, be):
    if case.algo != "cg" and variant == "resident":
        pytest.skip("variant only affects cg")
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    v_before = [v.clone() for v in vector]
    w_before = [p.data.clone() for p in curr.trainable_parameters()]
    be.cg_variant = VARIANTS[variant]
    try:
        out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
    finally:
        be.cg_variant = _native.BHG_CG_AUTO
    want = golden_list(outputs, case.name, "fp32")
    assert len(out) == len(want)
    rel, mx = rel_err(_np(out), want)
    # rtol: 1e-4 (north_star) for cg/neumann; finite differences carry the fp32 noise the reference
    # itself shows between its fp32 and fp64 runs (Case.rtol)
    assert rel <= case.rtol and mx <= 10 * case.rtol, (case.name, variant, rel, mx)
    for v, vb in zip(vector, v_before):
        assert torch.equal(v, vb), "direction vector must not be mutated"
    if case.algo in ("darts", "sama"):
        for p, w in zip(curr.trainable_parameters(), golden_list(outputs, case.name, "w32")):
            np.testing.assert_allclose(p.data.cpu().numpy(), w, rtol=0, atol=2e-7)
    else:
        for p, w in zip(curr.trainable_parameters(), w_before):
            assert torch.equal(p.data, w)


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_hip_sync_accumulates_and_returns_none(case, be):
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    for p in prev.trainable_parameters():
        p.grad = torch.full_like(p, 0.25)
    assert hg.jvp_fn_mapping[case.algo](vector, curr, prev, True) is None
    got = [p.grad.detach().cpu().numpy().astype(np.float64) - 0.25 for p in prev.trainable_parameters()]
    want = golden_list(outputs, case.name, "sync32")
    rel, _ = rel_err(got, want)
    wmax = max(np.abs(np.concatenate([w.ravel() for w in want])).max(), 1e-30)
    scale = max(1.0, 0.25 / wmax)  # the 0.25 offset costs fp32 digits when the result is tiny
    assert rel <= case.rtol * scale + 2e-7 * scale, (case.name, rel)


def test_cg_is_bitwise_deterministic(be):
    """End to end with the analytic HVP (every kernel on the path is ours: fixed-order reductions, no atomics
    on data).  With an autograd HVP the run-to-run spread is ATen's: the same double backward through
    `nn.Linear` differs by ~5e-7 between calls in one process (hipBLASLt solution choice), and MIOpen's conv
    weight gradients use float atomics — so that variant is only checked to tolerance, against the goldens."""
    case = zoo.CASE_BY_NAME["reweight_cg20"]
    inputs, _ = load_golden(case.family)

    def run():
        curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
        zoo.attach_mlp_structure(curr, case.family)
        return _np(hg.cg(vector, curr, prev, False))

    outs = [run() for _ in range(3)]
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("variant", ["stream", "resident"])
def test_cg_recurrence_kernels_are_bitwise_deterministic(variant, be):
    """Same HVP tensors in -> same x, r, p and scalars out, bit for bit, run after run (N = 1.2 M in 7 tensors)."""
    sizes = [500000, 4097, 300001, 3, 250000, 77, 150000]
    gen = torch.Generator().manual_seed(5)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in sizes]
    diag = [(1.0 + 0.5 * torch.rand(n, generator=gen)).to(DEV) for n in sizes]
    lay = be.layout(vec)
    results = []
    for _ in range(3):
        x, r, p = lay.new_flat(), lay.new_flat(), lay.new_flat()
        be.cg_init(lay, vec, x, r, p)
        for k in range(6):
            hv = [d * t for d, t in zip(diag, lay.views(p, vec))]   # elementwise: deterministic
            be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == 5 else 0.0), variant=VARIANTS[variant])
        results.append((x.cpu().numpy(), r.cpu().numpy(), p.cpu().numpy(), be.cg_scalars(lay).cpu().numpy()))
    for other in results[1:]:
        for a, b in zip(results[0], other):
            np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# kernel level against the C oracle on ragged multi-tensor inputs
# ------------------------------------------------------------------------------------------------
RAGGED = [
    [1],
    [3, 5, 7],
    [4095, 4096, 4097],
    [10000, 1, 64, 63, 8193],
    [17] * 50,  # T = 50 > 32: device pointer table
    [4096 * 3 + 5, 2, 4096],
    [250000, 130001, 77],
]


def _fuzz_sizes(seed):
    """Seeded random tensor-size lists: tiny tensors, sizes straddling chunk (4096) and float4 boundaries,
    a few large ones, T on both sides of the 32-pointer inline table."""
    rs = np.random.RandomState(seed)
    T = int(rs.choice([2, 5, 31, 32, 33, 40, 70]))
    pool = [1, 2, 3, 4, 5, 63, 64, 65, 4093, 4094, 4095, 4096, 4097, 4099, 8191, 8192, 8193, 12288]
    out = []
    for _ in range(T):
        kind = rs.randint(0, 10)
        if kind < 5:
            out.append(int(rs.choice(pool)))
        elif kind < 8:
            out.append(int(rs.randint(1, 3000)))
        else:
            out.append(int(rs.randint(20000, 120000)))
    return out


RAGGED += [_fuzz_sizes(seed) for seed in range(8)]


def _rand_list(sizes, gen, scale=1.0):
    return [(scale * torch.randn(n, generator=gen)).to(DEV) for n in sizes]


def _flat_concat(layout, flat):
    return torch.cat([flat[s : s + n] for s, n in zip(layout.starts, layout.numels)]).cpu().numpy()


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("sizes", RAGGED, ids=lambda s: f"T{len(s)}_N{sum(s)}")
def test_flatten_scatter_axpy_bitwise(sizes, be):
    gen = torch.Generator().manual_seed(sum(sizes))
    src = _rand_list(sizes, gen)
    lay = be.layout(src)
    flat = lay.new_flat()
    be.flatten(lay, src, flat, 0.37)
    want = np.concatenate([(np.float32(0.37) * t.cpu().numpy()) for t in src])
    np.testing.assert_array_equal(_flat_concat(lay, flat), want)
    # padding untouched
    mask = torch.ones(lay.flat_size, dtype=torch.bool)
    for s, n in zip(lay.starts, lay.numels):
        mask[s : s + n] = False
    assert float(flat.cpu()[mask].abs().sum()) == 0.0
    dst = [torch.empty_like(t) for t in src]
    be.scatter(lay, flat, dst, -2.0)
    for d, t in zip(dst, src):
        np.testing.assert_array_equal(d.cpu().numpy(), np.float32(-2.0) * (np.float32(0.37) * t.cpu().numpy()))
    # axpy with a device coefficient
    w = _rand_list(sizes, gen)
    w0 = [t.clone() for t in w]
    coef = torch.tensor([0.123], device=DEV)
    be.axpy_multi(lay, w, src, coef[0], -2.0)
    a = np.float32(-2.0) * np.float32(0.123)
    for wi, w0i, si in zip(w, w0, src):
        np.testing.assert_array_equal(wi.cpu().numpy(), w0i.cpu().numpy() + a * si.cpu().numpy())


@pytest.mark.parametrize("sizes", RAGGED, ids=lambda s: f"T{len(s)}_N{sum(s)}")
def test_neumann_kernels_bitwise_vs_oracle(sizes, be, orc):
    gen = torch.Generator().manual_seed(7 + sum(sizes))
    vec = _rand_list(sizes, gen)
    lay = be.layout(vec)
    v, p = lay.new_flat(), lay.new_flat()
    be.neumann_init(lay, vec, v, p)
    v_ref = np.concatenate([t.cpu().numpy() for t in vec])
    p_ref = v_ref.copy()
    for k in range(3):
        hv = _rand_list(sizes, gen)
        out_scale = -0.3 if k == 2 else 0.0
        be.neumann_step(lay, hv, v, p, 0.3, out_scale)
        h = np.concatenate([t.cpu().numpy() for t in hv])
        orc.orc_neumann_step(_ptr(h), _ptr(v_ref), _ptr(p_ref), len(h), 0.3, out_scale)
    np.testing.assert_array_equal(_flat_concat(lay, v), v_ref)  # element-wise, no reductions: exact
    np.testing.assert_array_equal(_flat_concat(lay, p), p_ref)


def _oracle_cg(orc, vec_np, hvps, cg_alpha):
    n = len(vec_np)
    x, r, p = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
    orc.orc_cg_init(_ptr(vec_np), _ptr(x), _ptr(r), _ptr(p), n)
    rr = orc.orc_sqnorm(_ptr(vec_np), n)
    scal = []
    K = len(hvps)
    for k, h in enumerate(hvps):
        den = orc.orc_dot_scaled(_ptr(h), _ptr(p), n, cg_alpha)
        a = np.float32(np.float32(rr) / np.float32(den))
        rr_new = orc.orc_cg_resid(_ptr(h), _ptr(r), n, float(a))
        b = np.float32(np.float32(rr_new) / np.float32(rr))
        orc.orc_cg_dir(_ptr(x), _ptr(r), _ptr(p), n, float(a), float(b), -cg_alpha if k == K - 1 else 0.0)
        scal.append((rr, den, float(a), rr_new, float(b)))
        rr = rr_new
    return x, r, p, scal


@pytest.mark.parametrize("variant", ["stream", "resident"])
@pytest.mark.parametrize("sizes", RAGGED, ids=lambda s: f"T{len(s)}_N{sum(s)}")
def test_cg_kernels_vs_oracle(sizes, variant, be, orc):
    gen = torch.Generator().manual_seed(11 + sum(sizes))
    vec = _rand_list(sizes, gen)
    lay = be.layout(vec)
    x, r, p = lay.new_flat(), lay.new_flat(), lay.new_flat()
    be.cg_init(lay, vec, x, r, p)
    K = 4
    hv_lists = []
    for k in range(K):
        # H p with a fixed SPD diagonal so the recurrence stays well scaled
        pv = lay.views(p, vec)
        hv = [(1.0 + 0.5 * torch.sin(torch.arange(t.numel(), device=DEV, dtype=torch.float32))) * t for t in pv]
        hv_lists.append(np.concatenate([t.cpu().numpy() for t in hv]))
        be.cg_step(lay, hv, x, r, p, 0.7, k, out_scale=(-0.7 if k == K - 1 else 0.0), variant=VARIANTS[variant])
        got_scal = be.cg_scalars(lay).cpu().numpy()
        assert np.all(np.isfinite(got_scal))
    # the oracle consumes the very HVP vectors the GPU run produced, so the two recurrences can
    # only drift through the dot products' summation order (fp64 on both sides)
    xo, ro, po, scal = _oracle_cg(orc, np.concatenate([t.cpu().numpy() for t in vec]), hv_lists, 0.7)
    np.testing.assert_allclose(got_scal, np.array(scal[-1]), rtol=1e-6)
    for got, want in ((x, xo), (r, ro), (p, po)):
        g = _flat_concat(lay, got)
        scale = np.abs(want).max() + 1e-30
        # 4 iterations of fp32 recurrences whose alpha/beta may differ in the last ulp
        assert np.abs(g - want).max() <= 4e-6 * scale


def test_darts_eps_matches_oracle(be, orc):
    gen = torch.Generator().manual_seed(3)
    vec = _rand_list([1000, 33, 5000], gen, scale=0.01)
    lay = be.layout(vec)
    e32, e64, _ = be.darts_eps(lay, vec, 0.01)
    flat = np.concatenate([t.cpu().numpy() for t in vec])
    want = orc.orc_darts_eps(orc.orc_sqnorm(_ptr(flat), len(flat)), 0.01)
    assert abs(float(e64) - want) <= 1e-12 * want
    assert float(e32) == np.float32(float(e64))
    # against torch's own norm, as darts.py:30-35 computes it
    ref = 0.01 / (float(torch.cat([t.reshape(-1) for t in vec]).norm()) + 1e-15)
    assert abs(float(e64) - ref) <= 1e-6 * ref
    # all-zero vector: eps = R / 1e-15, finite
    z = [torch.zeros(10, device=DEV)]
    e32, e64, _ = be.darts_eps(be.layout(z), z, 0.01)
    assert np.isfinite(float(e64)) and float(e64) > 1e12


# ------------------------------------------------------------------------------------------------
# full BASELINE size (N = 10,034,826 in the 8 tensors of the cfg-2 MLP): closed forms
# ------------------------------------------------------------------------------------------------
CFG2_SIZES = [3072 * 2048, 2048, 2048 * 1536, 1536, 1536 * 384, 384, 384 * 10, 10]


@pytest.mark.parametrize("variant", ["stream", "resident"])
def test_cg_full_size_diagonal_hessian_converges_exactly(variant, be):
    """H = diag(d) with 4 distinct eigenvalues: CG solves H x = v exactly in 4 iterations
    (Krylov argument), independent of N.  Checks x = -cg_alpha * v / d at N = 10 M."""
    assert sum(CFG2_SIZES) == 10_034_826
    gen = torch.Generator().manual_seed(5)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in CFG2_SIZES]
    dvals = torch.tensor([0.5, 1.0, 2.0, 4.0], device=DEV)
    diag = [dvals[torch.arange(n, device=DEV) % 4] for n in CFG2_SIZES]
    lay = be.layout(vec)
    if variant == "resident" and lay.n_chunks > be.lib.bhg_cg_resident_capacity_chunks():
        pytest.skip("vector does not fit the resident kernel on this device")
    x, r, p = lay.state(3)
    be.cg_init(lay, vec, x, r, p)
    K = 4
    for k in range(K):
        hv = [d * t for d, t in zip(diag, lay.views(p, vec))]
        be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == K - 1 else 0.0), variant=VARIANTS[variant])
    for xv, v, d in zip(lay.views(x, vec), vec, diag):
        want = -(v / d)
        err = (xv - want).abs().max().item()
        assert err <= 2e-5 * want.abs().max().item(), err  # fp32 CG, 4 steps, kappa = 8
    ws_timeout = lay.workspace[24704 + 4 : 24704 + 8].view(torch.int32)  # barrier_words[1]
    assert int(ws_timeout.item()) == 0, "grid barrier timed out"


def test_cg_resident_lds_assisted_instance(be):
    """N = 15.1 M (1.5 x the cfg-2 tensor list) exceeds the register-only resident capacity (11.5 M) and fits the
    LDS-assisted instance (9 of 15 direction slices per workgroup parked in LDS): same closed form, bitwise equal to
    the streaming kernels iteration by iteration is not required (different summation grouping of the dots), but the
    solve must be exact to fp32 CG accuracy, bit-reproducible, and must not time out."""
    sizes = CFG2_SIZES + [5_000_000, 4097, 63]
    N = sum(sizes)
    gen = torch.Generator().manual_seed(11)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in sizes]
    dvals = torch.tensor([0.5, 1.0, 2.0, 4.0], device=DEV)
    diag = [dvals[torch.arange(n, device=DEV) % 4] for n in sizes]
    lay = be.layout(vec)
    G = be.lib.bhg_cg_resident_capacity_chunks() // 28
    if not be.lib.bhg_cg_resident_ok() or not (11 * G < lay.n_chunks <= 15 * G):
        pytest.skip("size does not select the LDS-assisted instance on this device")
    outs = []
    for variant in ("resident", "resident", "stream"):
        x, r, p = lay.state(3)
        be.cg_init(lay, vec, x, r, p)
        K = 4
        for k in range(K):
            hv = [d * t for d, t in zip(diag, lay.views(p, vec))]
            be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == K - 1 else 0.0), variant=VARIANTS[variant])
            lay._cg_variant = None
        assert not be.cg_barrier_timed_out(lay)
        outs.append(x.clone())
        for xv, v, d in zip(lay.views(x, vec), vec, diag):
            want = -(v / d)
            assert (xv - want).abs().max().item() <= 2e-5 * want.abs().max().item(), (variant, N)
    assert torch.equal(outs[0], outs[1]), "the resident kernel must be bit-reproducible"
    rel = (outs[0] - outs[2]).norm() / outs[2].norm()
    assert rel <= 1e-6, rel


def test_neumann_full_size_closed_form(be):
    """H = diag(d): p_K = sum_{j<=K} (1 - alpha d)^j v, result = -alpha p_K (fp32 geometric sum)."""
    gen = torch.Generator().manual_seed(6)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in CFG2_SIZES]
    diag = [0.5 + (torch.arange(n, device=DEV) % 7).float() * 0.25 for n in CFG2_SIZES]
    lay = be.layout(vec)
    v, p = lay.state(2)
    be.neumann_init(lay, vec, v, p)
    K, alpha = 10, 0.1
    for k in range(K):
        hv = [d * t for d, t in zip(diag, lay.views(v, vec))]
        be.neumann_step(lay, hv, v, p, alpha, out_scale=(-alpha if k == K - 1 else 0.0))
    for pv, vv, d in zip(lay.views(p, vec), vec, diag):
        q = (1.0 - alpha * d.double())
        geo = sum(q**j for j in range(K + 1))
        want = (-alpha * geo * vv.double()).float()
        assert (pv - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_recurrences_at_roberta_scale_closed_form(be):
    """Maximum size on the list (BASELINE cfg 4: 124 M elements in 12 x 16 + 5 tensors, far beyond the resident
    kernel's capacity): streaming CG on a 4-eigenvalue diagonal Hessian is exact after 4 iterations, Neumann
    matches the geometric-series closed form."""
    layer = [768 * 768, 768] * 4 + [3072 * 768, 3072, 768 * 3072, 768] + [768] * 4
    sizes = [50265 * 768, 514 * 768, 768, 768] + layer * 12 + [768 * 2, 2]
    assert sum(sizes) > 120_000_000
    gen = torch.Generator().manual_seed(8)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in sizes]
    lay = be.layout(vec)
    assert lay.n_chunks > be.lib.bhg_cg_resident_capacity_chunks()
    dvals = torch.tensor([0.5, 1.0, 2.0, 4.0], device=DEV)
    diag = [dvals[torch.arange(n, device=DEV) % 4] for n in sizes]
    x, r, p = lay.state(3)
    be.cg_init(lay, vec, x, r, p)
    for k in range(4):
        hv = [d * t for d, t in zip(diag, lay.views(p, vec))]
        be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == 3 else 0.0))   # AUTO -> stream
        del hv
    for xv, v, d in zip(lay.views(x, vec), vec, diag):
        want = -(v / d)
        assert (xv - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    v_, p_ = lay.state(2)
    be.neumann_init(lay, vec, v_, p_)
    K, alpha = 6, 0.1
    for k in range(K):
        hv = [d * t for d, t in zip(diag, lay.views(v_, vec))]
        be.neumann_step(lay, hv, v_, p_, alpha, out_scale=(-alpha if k == K - 1 else 0.0))
        del hv
    for pv, vv, d in zip(lay.views(p_, vec), vec, diag):
        q = 1.0 - alpha * d.double()
        geo = sum(q**j for j in range(K + 1))
        want = (-alpha * geo * vv.double()).float()
        assert (pv - want).abs().max().item() <= 2e-6 * want.abs().max().item()


def test_product_refuses_cpu_tensors(be):
    t = [torch.randn(10)]
    lay = be.layout([torch.randn(10, device=DEV)])
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        be.neumann_init(lay, t, lay.new_flat(), lay.new_flat())


# ------------------------------------------------------------------------------------------------
# analytic MLP HVP on the matrix cores (csrc/bhg_mlp.hip)
# ------------------------------------------------------------------------------------------------
def _mlp_problem(dims, B, ridge, seed):
    from betty_amd.hypergradient.structured import WeightedCEMLP

    g = torch.Generator().manual_seed(seed)
    inner = zoo.MLP(dims)
    upper = zoo.MWN(16)
    with torch.no_grad():
        for p in list(inner.parameters()) + list(upper.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(p.shape[-1], 4) ** 0.5))
    inner, upper = inner.to(DEV), upper.to(DEV)
    x = torch.randn(B, dims[0], generator=g).to(DEV)
    y = torch.randint(0, dims[-1], (B,), generator=g).to(DEV)
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type="cg"), loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x, y))
    direction = [torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]

    def provider(impl):
        # (verify=False: these helpers feed tests that count kernel launches; the structure guard has tests of its own and is on
        #  wherever a problem is declared through bench.declare_structure / zoo.attach_mlp_structure)
        return WeightedCEMLP(curr, prev, layers=list(inner.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)),
                             ridge=ridge, impl=impl, verify=False)

    return curr, prev, direction, provider


@pytest.mark.parametrize(
    "dims,B",
    [([48, 64, 32, 10], 40), ([70, 130, 33, 10], 100), ([256, 192, 10], 128), ([33, 7], 5), ([12] + [12] * 5 + [4], 17),
     ([70, 130, 36, 10], 200), ([64, 48, 10], 300), ([40, 52, 50], 129)],  # > 128 rows: several 128-row M tiles
    ids=lambda v: str(v),
)
def test_mlp_hvp_kernels_vs_aten_and_autograd(dims, B):
    curr, prev, direction, provider = _mlp_problem(dims, B, ridge=0.05, seed=sum(dims) + B)
    hip = provider("hip")
    got = [t.clone() for t in hip.prepare()(direction)]
    ref = provider("torch").prepare()(direction)
    assert hip.hvp_shift == pytest.approx(0.1)
    # same closed form, fp32 on both sides: only the GEMM summation order differs
    for a, b in zip(got, ref):
        scale = b.abs().max().item() + 1e-30
        assert (a - b).abs().max().item() <= 2e-5 * scale
    # and against double backward through the user's training_step
    loss = curr.training_step_exec(curr.cur_batch)
    g = torch.autograd.grad(loss, curr.parameters(), create_graph=True)
    want = torch.autograd.grad(g, curr.parameters(), grad_outputs=direction, retain_graph=True)
    full = [a + hip.hvp_shift * d for a, d in zip(got, direction)]  # + the ridge part the recurrence kernel adds
    rel, _ = rel_err(_np(full), _np(want))
    assert rel <= 2e-5, rel
    # mixed VJP
    mv = hip.mixed_vjp(direction, False)
    want_m = torch.autograd.grad(g, prev.trainable_parameters(), grad_outputs=direction)
    rel, _ = rel_err(_np(mv), _np(want_m))
    assert rel <= 2e-5, rel


@pytest.mark.parametrize("name", ["reweight_cg20", "reweight_neumann10", "deep_cg6", "deep_neumann6"])
@pytest.mark.parametrize("sync", [False, True])
def test_structured_hip_path_matches_reference(name, sync, be):
    case = zoo.CASE_BY_NAME[name]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    zoo.attach_mlp_structure(curr, case.family, impl="hip")
    out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
    if sync:
        assert out is None
        out = [p.grad for p in prev.trainable_parameters()]
    rel, mx = rel_err(_np(out), golden_list(outputs, case.name, "fp32"))
    assert rel <= 1e-4 and mx <= 1e-3, (rel, mx)  # north_star tolerance


# ------------------------------------------------------------------------------------------------
# round 5: the meta-weight-net in closed form (csrc/bhg_mwn.hip; SigmoidMLPWeightNet) — the upper problem of
# examples/learning_to_reweight/model.py:98-111, evaluated and differentiated in one launch each way
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H", [(100, 100), (37, 500), (1500, 64), (5, 2048), (1, 1), (2500, 300)])
def test_mwn_kernels_match_autograd(B, H, be):
    """bhg_mwn_forward / bhg_mwn_backward through the C ABI against the module itself under autograd (fp32 on the same GPU)."""
    import ctypes

    torch.manual_seed(B * 7 + H)
    net = zoo.MWN(H).to(DEV)
    ce = (2.5 * torch.rand(B, device=DEV)).contiguous()
    coeff = torch.randn(B, device=DEV).contiguous()
    s_ref = net(ce.reshape(-1, 1)).reshape(-1)
    g_ref = torch.autograd.grad(s_ref, list(net.parameters()), grad_outputs=coeff)
    lib = be.lib
    s, sd = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    w1, b1, w2, b2 = [t.detach().contiguous() for t in (net.l1.weight, net.l1.bias, net.l2.weight, net.l2.bias)]
    st = int(torch.cuda.current_stream().cuda_stream)
    _native.check(lib.bhg_mwn_forward(ce.data_ptr(), B, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), H, s.data_ptr(),
                                      sd.data_ptr(), st), "bhg_mwn_forward")
    np.testing.assert_allclose(s.cpu().numpy(), s_ref.detach().cpu().numpy(), rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(sd.cpu().numpy(), (s_ref.detach() / B).cpu().numpy(), rtol=2e-6, atol=2e-7 / B)
    outs = [torch.full_like(t, float("nan")) for t in (w1, b1, w2, b2)]
    for scale in (1.0, 0.25):
        _native.check(lib.bhg_mwn_backward(ce.data_ptr(), coeff.data_ptr(), B, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), H,
                                           ctypes.c_float(scale), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                                           st), "bhg_mwn_backward")
        rel, mx = rel_err(_np(outs), [scale * g.detach().cpu().numpy().astype(np.float64) for g in g_ref])
        assert rel <= 2e-6 and mx <= 2e-5, (B, H, scale, rel, mx)
    again = [o.clone() for o in outs]
    _native.check(lib.bhg_mwn_backward(ce.data_ptr(), coeff.data_ptr(), B, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), H,
                                       ctypes.c_float(0.25), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(), st),
                  "bhg_mwn_backward")
    for a, b in zip(again, outs):
        assert torch.equal(a, b), "fixed summation order: bitwise run-to-run"
    assert lib.bhg_mwn_forward(ce.data_ptr(), B, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 4096, s.data_ptr(), None, st) != 0


@pytest.mark.parametrize("name", ["reweight_cg20", "reweight_neumann10", "deep_cg6", "deep_neumann6"])
@pytest.mark.parametrize("sync", [False, True])
def test_closed_form_weight_net_matches_reference(name, sync, be):
    """The structured path with the upper module DECLARED (closed-form sample weights and upper VJP) against the reference's goldens,
    and against the same path with the upper module under autograd; sync=True accumulates into .grad like Problem.set_grads."""
    case = zoo.CASE_BY_NAME[name]
    inputs, outputs = load_golden(case.family)
    res = {}
    for declared in (True, False):
        curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
        zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=declared)
        if sync:
            for p in prev.trainable_parameters():
                p.grad = torch.full_like(p, 0.25)
        out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
        if sync:
            assert out is None
            out = [p.grad - 0.25 for p in prev.trainable_parameters()]
        res[declared] = _np(out)
        if declared:
            from betty_amd.hypergradient import structured

            prov = structured.structured_hvp_for(curr, prev)
            prov.prepare()
            assert prov._state.native_upper, "the closed form was declared but did not run"
    want = golden_list(outputs, case.name, "fp32")
    wmax = max(np.abs(np.concatenate([w.ravel() for w in want])).max(), 1e-30)
    scale = max(1.0, 0.25 / wmax) if sync else 1.0   # the 0.25 offset costs fp32 digits when the result is tiny
    rel, mx = rel_err(res[True], want)
    assert rel <= (1e-4 + 2e-7) * scale and mx <= 1e-3 * scale, (rel, mx)
    rel2, _ = rel_err(res[True], res[False])
    assert rel2 <= 2e-5 * scale, rel2


def test_closed_form_weight_net_declines_what_it_does_not_describe(be):
    """Upper parameters that are not exactly the four tensors of the declared net: autograd keeps the job (no wrong gradient)."""
    case = zoo.CASE_BY_NAME["reweight_cg20"]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True)
    only = list(prev.module.parameters())[:2]
    prev.trainable_parameters = lambda: only          # e.g. a frozen output layer
    from betty_amd.hypergradient import structured

    prov = structured.structured_hvp_for(curr, prev)
    prov.verify = False
    prov.prepare()
    assert not prov._state.native_upper
    out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
    rel, _ = rel_err(_np(out), golden_list(outputs, case.name, "fp32")[:2])
    assert rel <= 1e-4, rel


def test_a_wrong_weight_net_declaration_is_rejected(be):
    """A meta-weight-net with another activation than the declared one: the first prepare() raises StructureMismatchError."""
    from betty_amd.hypergradient import structured

    case = zoo.CASE_BY_NAME["reweight_cg20"]
    inputs, _ = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    fwd = prev.module.forward
    prev.module.forward = lambda x: torch.sigmoid(prev.module.l2(torch.tanh(prev.module.l1(x))))   # tanh, not relu
    try:
        zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True)
        with pytest.raises(structured.StructureMismatchError):
            hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
    finally:
        prev.module.forward = fwd


@pytest.mark.parametrize("name", ["reweight_cg20", "reweight_neumann10", "deep_cg6", "deep_neumann6"])
def test_structured_goldens_with_and_without_fusion(name, be):
    """Both arms of the structured path against the reference's goldens: fused (the product default: recurrence
    applied inside the HVP's output kernels, step length from the batch-sized factors) and un-fused (HVP kernels
    + recurrence kernel)."""
    case = zoo.CASE_BY_NAME[name]
    inputs, outputs = load_golden(case.family)
    res = {}
    for fused in (True, False):
        curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
        zoo.attach_mlp_structure(curr, case.family, impl="hip", fused=fused)
        out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
        rel, mx = rel_err(_np(out), golden_list(outputs, case.name, "fp32"))
        assert rel <= 1e-4 and mx <= 1e-3, (fused, rel, mx)
        res[fused] = _np(out)
    rel, _ = rel_err(res[True], res[False])
    assert rel <= 2e-5, rel   # same HVP tiles, same roundings; only alpha's reduction differs (factors vs N-sized dot)


def _run_solver(algo, dims, B, ridge, K, seed, fused, alpha=None, keep=True):
    curr, prev, direction, provider = _mlp_problem(dims, B, ridge=ridge, seed=seed)
    from betty_amd.hypergradient.structured import WeightedCEMLP

    curr.config = Config(type="cg", cg_iterations=K, cg_alpha=1.0 if alpha is None else alpha) if algo == "cg" else \
        Config(type="neumann", neumann_iterations=K, neumann_alpha=0.05 if alpha is None else alpha)
    curr.hypergradient_structure = lambda prev_: WeightedCEMLP(
        curr, prev_, layers=list(curr.module.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge,
        impl="hip", fused=fused, keep_solution=keep, verify=False)   # (most of these tests read the flat solution vector back)
    vec = [0.1 * d for d in direction]
    out = hg.jvp_fn_mapping[algo](vec, curr, prev, False)
    lay = get_backend().layout(vec)
    state = [t.clone() for t in lay.state(3)] if algo == "cg" else [lay.state(2)[1].clone()]
    return _np(out), [s.cpu().numpy() for s in state]


@pytest.mark.parametrize("algo", ["cg", "neumann"])
@pytest.mark.parametrize(
    "dims,B,K",
    [([48, 64, 32, 10], 40, 5), ([70, 130, 36, 10], 100, 4), ([256, 192, 10], 128, 6), ([36, 7], 5, 3),
     ([12] + [12] * 5 + [4], 17, 4), ([64, 48, 10], 300, 3), ([128, 128, 64, 10], 64, 1), ([256, 384, 128, 10], 100, 7)],
    ids=lambda v: str(v),
)
def test_fused_solver_matches_unfused(algo, dims, B, K):
    """One-pass solver vs K x (bhg_mlp_hvp + recurrence kernel) on the same inputs: ragged tiles (edge path of the
    fused epilogue), L = 1 (head only), L = 2, deep nets, several 128-row batch tiles, all-interior FAST tiles;
    the flat solution vector itself is compared too, not only the M-sized hypergradient."""
    lib = _native.load()
    n0 = lib.bhg_mlp_hoist_launches()
    got, st_f = _run_solver(algo, dims, B, 0.05, K, sum(dims) + B, True)
    hoisted = lib.bhg_mlp_hoist_launches() > n0     # shapes the hoisted chain takes (>= 3 layers, all widths % 32 == 0)
    want, st_u = _run_solver(algo, dims, B, 0.05, K, sum(dims) + B, False)
    rel, _ = rel_err(got, want)
    assert rel <= 5e-5, rel
    x_f, x_u = st_f[0].astype(np.float64), st_u[0].astype(np.float64)   # x (cg) / p (neumann): the solve's result
    assert np.linalg.norm(x_f - x_u) <= 5e-5 * np.linalg.norm(x_u)
    if algo == "neumann":
        if hoisted:   # same products, another summation tree (direction products summed on their own, added in the epilogue)
            assert np.linalg.norm(x_f - x_u) <= 2e-6 * np.linalg.norm(x_u)
        else:         # no reduction anywhere in the Neumann recurrence: identical tiles, identical roundings
            assert np.array_equal(st_f[0], st_u[0])


@pytest.mark.parametrize("dims,B,K", [([256, 384, 128, 10], 100, 5), ([70, 130, 36, 10], 100, 4), ([3072, 2048, 1536, 384, 10], 100, 20)],
                         ids=lambda v: str(v))
def test_fused_cg_without_a_solution_vector(dims, B, K, be, bhg_debug):
    """The product default: the fused CG solver is handed x = NULL (WeightedCEMLP.keep_solution=False) — the mixed
    second derivative comes from Rz(x) = sum_k alpha_k Rz(p_k), accumulated from batch-sized factors, so the N-sized
    solution is never zeroed, read or written, and the x buffer of the layout is provably untouched (filled with NaN before
    the call, still all NaN after it).  At the same projection level (BHG_MLP_PROJ=9: the N-sized residual and direction are
    kept in both runs) the hypergradient is BIT-identical to the run that materialises x — x is write-only; the default
    without x goes further (no N-sized state at all, tests/test_cfg2_goldens.py holds it to the reference)."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    bhg_debug.setenv("BHG_MLP_PROJ", "9")
    outs = {}
    for keep in (True, False):
        curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=sum(dims) + B + K)
        curr.config = Config(type="cg", cg_iterations=K, cg_alpha=1.0)
        curr.hypergradient_structure = lambda prev_, keep=keep, curr=curr: WeightedCEMLP(
            curr, prev_, layers=list(curr.module.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=0.05,
            impl="hip", fused=True, keep_solution=keep)
        vec = [0.1 * d for d in direction]
        lay = be.layout(vec)
        x = lay.state(3)[0]
        x.fill_(float("nan"))
        try:
            outs[keep] = [t.clone() for t in hg.jvp_fn_mapping["cg"](vec, curr, prev, False)]
            if keep:
                assert torch.isfinite(x[lay.starts[0]: lay.starts[0] + vec[0].numel()]).all()
            else:
                assert torch.isnan(x).all(), "x must not be touched"
        finally:
            x.zero_()   # layouts (and their state buffers, zero padding included) are cached per shape list
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)


@pytest.mark.parametrize("dims,B,K", [([256, 384, 128, 10], 100, 5), ([70, 130, 36, 10], 100, 4), ([48, 64, 32, 10], 40, 1),
                                      ([3072, 2048, 1536, 384, 10], 100, 10)], ids=lambda v: str(v))
def test_fused_neumann_without_an_accumulator_vector(dims, B, K, be):
    """The product default of the fused Neumann solver: p = NULL — the accumulator p = sum_k v_k (neumann.py:64) is never
    written; the head kernel sums Rz(v_k), k < K, and bhg_mlp_neumann_mixed_coeff adds the last direction's share with
    the one R-forward pass the mixed coefficient costs anyway.  By linearity it is the same hypergradient: equal to the
    run that materialises p to fp32 summation noise (the Rz of a sum vs the sum of the Rz's), and p provably untouched."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    outs = {}
    for keep in (True, False):
        curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=sum(dims) + B + K)
        curr.config = Config(type="neumann", neumann_iterations=K, neumann_alpha=0.05)
        curr.hypergradient_structure = lambda prev_, keep=keep, curr=curr: WeightedCEMLP(
            curr, prev_, layers=list(curr.module.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=0.05,
            impl="hip", fused=True, keep_solution=keep)
        vec = [0.1 * d for d in direction]
        lay = be.layout(vec)
        p = lay.state(2)[1]
        p.fill_(float("nan"))
        try:
            outs[keep] = _np(hg.jvp_fn_mapping["neumann"](vec, curr, prev, False))
            if not keep:
                assert torch.isnan(p).all(), "p must not be touched"
        finally:
            p.zero_()
    rel, _ = rel_err(outs[False], outs[True])
    assert rel <= 2e-6, rel


@pytest.mark.parametrize("algo", ["cg", "neumann"])
def test_mixed_coeff_identifies_the_solve_by_token_not_by_address(algo, be):
    """Round-2 finding: the state recognised "the solution the fused solver just produced" by pointer arithmetic on
    data_ptr().  Now the solver returns a token.  With the solution materialised (keep_solution=True):
      * mixed_vjp(views, solve=token)  -> coefficient from the accumulated Rz (no R-forward),
      * mixed_vjp(CLONED views)        -> no token, the clones are read like any direction (one R-forward) — same
        hypergradient to fp32 noise, wherever the caller's copies live,
      * a token from an earlier solve, or after a hand-driven HVP, is refused."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    dims, B, K = [256, 384, 128, 10], 100, 5
    curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=77)
    prov = WeightedCEMLP(curr, prev, layers=list(curr.module.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)), ridge=0.05,
                         impl="hip", fused=True, keep_solution=True)
    hvp_fn = prov.prepare()
    vec = [0.1 * d for d in direction]
    lay = be.layout(vec)
    if algo == "cg":
        x, r, p = lay.state(3)
        be.cg_init(lay, vec, x, r, p)
        token = prov.fused_cg(lay, x, r, p, K, 1.0)
        sol = x
    else:
        v, p = lay.state(2)
        be.neumann_init(lay, vec, v, p)
        token = prov.fused_neumann(lay, v, p, K, 0.05)
        sol = p
    assert token and token is not True
    views = lay.views(sol, vec)
    st = prov._state
    c_token = st.mixed_coeff(views, token).clone()            # from the Rz the solver accumulated
    clones = [t.clone() for t in views]                       # different addresses, same numbers
    c_clone = st.mixed_coeff(clones).clone()                  # no token: the clones are read like any direction
    assert (c_token - c_clone).norm().item() <= 1e-5 * c_clone.norm().item()   # Rz of a sum vs the sum of the Rz's
    by_clone = _np(prov.mixed_vjp(clones, False))
    # the un-fused loop on the same inputs
    curr.config = Config(type=algo, cg_iterations=K, cg_alpha=1.0, neumann_iterations=K, neumann_alpha=0.05)
    curr.hypergradient_structure = lambda prev_: WeightedCEMLP(curr, prev_, layers=list(curr.module.layers),
                                                                weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=0.05,
                                                                impl="hip", fused=False)
    want = _np(hg.jvp_fn_mapping[algo](vec, curr, prev, False))
    rel, _ = rel_err(by_clone, want)
    assert rel <= 5e-5, rel
    # stale tokens are refused: a hand-driven HVP reuses the workspace the token refers to
    hvp_fn(lay.views(lay.state(3)[2], vec))
    with pytest.raises(RuntimeError, match="stale"):
        st.mixed_coeff(views, token)


@pytest.mark.parametrize("algo,K", [("cg", 8), ("neumann", 6)])
def test_opaque_hvp_replayed_as_a_hip_graph(algo, K, be):
    """Opt-in `curr.hypergradient_graph = True`: the opaque double backward of an MLP (no declared structure) is captured on
    the second call of a solve and replayed for the rest (GraphedHVP); the replayed launches are the very kernels eager
    autograd runs, so the hypergradient equals the eager one to ATen's own run-to-run noise, solve after solve."""
    from betty_amd.hypergradient import _common

    dims, B = [256, 384, 128, 10], 100
    outs = {}
    for arm in (False, True):
        curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=21)
        curr.config = Config(type=algo, cg_iterations=K, cg_alpha=1.0, neumann_iterations=K, neumann_alpha=0.05)
        curr.hypergradient_graph = arm
        before = dict(_common.GRAPH_STATS)
        vec = [0.1 * d for d in direction]
        for _ in range(3):   # three solves: three captures, each on a fresh autograd graph
            outs[arm] = _np(hg.jvp_fn_mapping[algo](vec, curr, prev, False))
        d = {k: _common.GRAPH_STATS[k] - before[k] for k in before}
        assert d == ({"captures": 3, "replays": 3 * (K - 1), "fallbacks": 0} if arm else {"captures": 0, "replays": 0, "fallbacks": 0}), d
    rel, _ = rel_err(outs[True], outs[False])
    assert rel <= 2e-5, rel


@pytest.mark.parametrize("algo,K", [("cg", 6), ("neumann", 5)])
def test_opaque_solve_with_persistent_graphs(algo, K, be):
    """Opt-in `curr.hypergradient_graph = "persistent"`: loss + gradient-with-graph (G1) and the HVP (G2) are captured on the
    second step and replayed on every later one, while the BATCH and the inner WEIGHTS change from step to step — the
    hypergradient of every step equals the eager one on the same inputs (sync=False and sync=True), and the counters show the
    graphs really ran (2 captures; per replayed step 1 + K replays)."""
    from betty_amd.hypergradient import _common

    dims, B = [256, 384, 128, 10], 100
    g = torch.Generator().manual_seed(3)
    runs = {}
    for arm in ("eager", "persistent"):
        curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=31)
        curr.config = Config(type=algo, cg_iterations=K, cg_alpha=1.0, neumann_iterations=K, neumann_alpha=0.05)
        curr.hypergradient_graph = "persistent" if arm == "persistent" else False
        vec = [0.1 * d for d in direction]
        before = dict(_common.GRAPH_STATS)
        outs = []
        gg = torch.Generator().manual_seed(77)
        for step in range(5):
            x = torch.randn(B, dims[0], generator=gg).to(DEV)
            y = torch.randint(0, dims[-1], (B,), generator=gg).to(DEV)
            curr.cur_batch = (x, y)                                      # a NEW batch every step
            with torch.no_grad():
                for prm in curr.module.parameters():                     # the optimizer moved the weights in place
                    prm.add_(0.01 * torch.randn(prm.shape, generator=gg).to(DEV))
            if step == 4:
                for q in prev.trainable_parameters():
                    q.grad = None
                assert hg.jvp_fn_mapping[algo](vec, curr, prev, True) is None          # sync=True through the kept graph
                outs.append(_np([q.grad for q in prev.trainable_parameters()]))
            else:
                outs.append(_np(hg.jvp_fn_mapping[algo](vec, curr, prev, False)))
        runs[arm] = outs
        d = {k: _common.GRAPH_STATS[k] - before[k] for k in before}
        if arm == "persistent":
            # captures: G1, G2 (step 1) and G3, the mixed second derivative (end of step 1); replays: step 1: K + 1 (G3),
            # steps 2-4: 1 (G1) + K (G2) + 1 (G3)
            assert d["captures"] == 3 and d["fallbacks"] == 0 and d["replays"] == (K + 1) + 3 * (K + 2), d
        else:
            assert d == {"captures": 0, "replays": 0, "fallbacks": 0}, d
    for step, (a, b) in enumerate(zip(runs["persistent"], runs["eager"])):
        rel, _ = rel_err(a, b)
        assert rel <= 2e-5, (step, rel)


@pytest.mark.parametrize("dims,B", [([192, 256, 128, 48], 72), ([784, 300, 100], 100), ([64, 96, 1000], 130), ([256, 384, 128, 100], 100),
                                    ([512, 256, 256, 64, 100], 100), ([256, 256, 130, 40], 64)], ids=lambda v: str(v))
def test_wide_head_runs_natively_and_matches_autograd(dims, B, be):
    """A classifier head wider than 32 outputs (40, 48, 100, 1000 classes; ragged widths; more than one 128-row batch tile) against the
    opaque autograd path on the same inputs, CG and Neumann.  Rounds 1-5: once-per-step passes on ATen, K loop un-fused.  Round 6:
      * up to 256 classes with a feature width that is a multiple of 4 the head kernels take it (classes in chunks of 4 * JMAX,
        csrc/mlp/head_body.inc) — the FUSED solvers run, in whatever form the plan gives the shapes (projected for the three- and four-layer nets);
      * beyond that (1000 classes; a two-layer net whose feature width is not a multiple of 4) the once-per-step passes are native all the same — the output layer as one more
        split-K product + k_softmax_ce_rows / k_coeff_rows — and the K loop is K x (HVP kernels + recurrence kernel)."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    for algo, K in (("cg", 4), ("neumann", 4)):
        outs = {}
        for arm in ("hip", "autograd"):
            curr, prev, direction, _ = _mlp_problem(dims, B, ridge=0.05, seed=5)
            curr.config = Config(type=algo, cg_iterations=K, cg_alpha=1.0, neumann_iterations=K, neumann_alpha=0.05)
            if arm == "hip":
                prov = WeightedCEMLP(curr, prev, layers=list(curr.module.layers), weight_fn=lambda ce, prev=prev: prev.fwd(ce.reshape(-1, 1)),
                                     ridge=0.05, impl="hip", fused=True)
                curr.hypergradient_structure = lambda prev_, prov=prov: prov
            outs[arm] = _np(hg.jvp_fn_mapping[algo]([0.1 * d for d in direction], curr, prev, False))
            if arm == "hip":
                st = prov._state
                inner_state = getattr(st, "inner", st)
                twin = len(dims) - 1 >= 3 and dims[-1] <= 256      # (a ragged feature width is rounded up by the zero-padded twin)
                want_fused = dims[-1] <= 256 and (twin or dims[-2] % 4 == 0)
                assert inner_state.buf.native_prepare and st.fused_supported(be.layout(direction)) == want_fused, (dims, want_fused)
        rel, _ = rel_err(outs["hip"], outs["autograd"])
        print(f"wide head {dims} B={B} {algo}: analytic ({'fused solver' if want_fused else 'native prepare + MFMA HVPs'}) vs autograd {rel:.2e}")
        assert rel <= 1e-4, (algo, rel)


def test_fused_cg_scalars_match_unfused(be):
    """alpha from the batch-sized factors == alpha from the N-sized dot (to fp32 reduction noise), iteration by
    iteration: one fused iteration against one un-fused iteration started from the same state."""
    dims, B = [256, 384, 128, 10], 100
    for K in (1, 2, 3):
        sc = {}
        for fused in (True, False):
            _run_solver("cg", dims, B, 0.05, K, 7, fused)
            curr, prev, direction, provider = _mlp_problem(dims, B, ridge=0.05, seed=7)
            lay = be.layout(direction)
            sc[fused] = be.cg_scalars(lay).cpu().numpy()
        # {rr_old, den, alpha, rr_new, beta}; the last iteration of the fused solver skips the (unused) direction update
        np.testing.assert_allclose(sc[True][:3], sc[False][:3], rtol=2e-5)


def test_fused_solver_is_bitwise_deterministic():
    a, sa = _run_solver("cg", [256, 384, 128, 10], 100, 0.05, 6, 3, True)
    b, sb = _run_solver("cg", [256, 384, 128, 10], 100, 0.05, 6, 3, True)
    assert all(np.array_equal(u, v) for u, v in zip(sa, sb))
    assert len(a) == len(b) and all(np.array_equal(u, v) for u, v in zip(a, b))


@pytest.mark.parametrize("ridge", [0.3, 1e-2], ids=["well-conditioned", "metric"])
def test_fused_solver_full_size_cfg2(ridge):
    """BASELINE cfg-2 shapes (N = 10,034,826, batch 100), CG K = 20 and Neumann K = 10: fused vs un-fused.
    ridge 0.3 (the well-conditioned variant of tests/golden/cfg2_full.npz): held to 1e-5 — two orders inside north_star's
    rtol.  ridge 1e-2 (the metric configuration): CG-20 amplifies last-bit differences of the step length (batch-sized
    factors vs N-sized dot) into the 3rd-4th digit there — the reference does the same to itself (golden `ref_spread`
    1.85e-2 on this seed) — so the arms are only held to that spread."""
    import bench

    for algo, K in (("cg", 20), ("neumann", 10)):
        outs = {}
        for arm in ("fused", "fused+solution", "unfused"):
            curr, prev, vector = bench.build(torch.device(DEV), seed=0, K=K, algo=algo, ridge=ridge)
            bench.declare_structure(curr, "hip", fused=arm != "unfused", keep_solution=arm == "fused+solution")
            outs[arm] = _np(hg.jvp_fn_mapping[algo](vector, curr, prev, False))
        rel, _ = rel_err(outs["fused"], outs["unfused"])
        rel_k, _ = rel_err(outs["fused+solution"], outs["unfused"])
        print(f"cfg2 full size ridge={ridge:g} {algo} K={K}: fused vs un-fused rel {rel:.2e} (with the solution vector materialised {rel_k:.2e})")
        if algo == "cg":   # (without x: fully projected; with x: N-sized r / p kept — two forms of the same recurrences)
            bound = 1e-5 if ridge >= 0.1 else 5.6e-2
            assert np.isfinite(rel) and rel <= bound and rel_k <= bound, (algo, ridge, rel, rel_k)
        else:   # Neumann has no reduction and no division: the same number to fp32 summation noise at either ridge
            assert rel_k <= 1e-6 and rel <= 5e-6, (algo, rel, rel_k)


@pytest.mark.parametrize(
    "dims,B",
    [([256, 384, 128, 10], 100), ([512, 256, 256, 64, 10], 128), ([256, 128, 64, 10], 64), ([768, 64, 32, 10], 37),
     ([256, 256, 256, 256, 128, 10], 100), ([512, 256, 64, 10], 200)],
    ids=lambda v: str(v),
)
def test_wsk_gemm_arm_matches_split_k(dims, B, bhg_debug):
    """k_gemm_wsk (a final 32 x 32 tile per workgroup: K split over the workgroup's waves, sum + bias + mask [+ the T2
    partial of the fused CG step length] before anything leaves the chip) against the split-K launches + reduce
    kernels: the HVP's outputs (un-fused chain, K-contiguous and N-contiguous weight operands, one and two operand
    pairs) and a fused CG solve (lazy direction formed in the loaders, T2 from the tile epilogue).  The launch counter
    proves the arm under test really ran."""
    lib = _native.load()
    bhg_debug.setenv("BHG_MLP_HOIST", "0")   # the classic chain: BHG_MLP_WSK picks the form of ITS skinny GEMMs
    outs, sols = {}, {}
    for arm in ("0", "1"):
        bhg_debug.setenv("BHG_MLP_WSK", arm)
        n0 = lib.bhg_mlp_wsk_launches()
        curr, prev, direction, provider = _mlp_problem(dims, B, ridge=0.05, seed=sum(dims) + B)
        outs[arm] = [t.clone() for t in provider("hip").prepare()(direction)]
        sols[arm] = _run_solver("cg", dims, B, 0.05, 4, sum(dims) + B, True)
        n1 = lib.bhg_mlp_wsk_launches()
        assert (n1 > n0) == (arm == "1"), (arm, n0, n1)
    for a, b in zip(outs["1"], outs["0"]):
        scale = b.abs().max().item() + 1e-30
        assert (a - b).abs().max().item() <= 2e-5 * scale   # same products, different summation tree
    rel, _ = rel_err(sols["1"][0], sols["0"][0])
    assert rel <= 5e-5, rel
    x1, x0 = sols["1"][1][0].astype(np.float64), sols["0"][1][0].astype(np.float64)
    assert np.linalg.norm(x1 - x0) <= 5e-5 * np.linalg.norm(x0)
    # bit-reproducible (fixed-order sums in LDS, no atomics)
    bhg_debug.setenv("BHG_MLP_WSK", "1")
    again = _run_solver("cg", dims, B, 0.05, 4, sum(dims) + B, True)
    assert all(np.array_equal(u, v) for u, v in zip(again[1], sols["1"][1]))


@pytest.mark.parametrize("dims,B", [([256, 384, 128, 10], 100), ([512, 256, 256, 64, 10], 128), ([64, 96, 64, 32, 64, 96, 32, 64, 10], 50),
                                    ([3072, 2048, 1536, 384, 10], 100), ([256, 128, 64, 10], 17)], ids=lambda v: str(v))
def test_packed_prepare_matches_the_split_k_prepare(dims, B, bhg_debug):
    """Round 5: the once-per-step passes (activations, ReLU masks, softmax, deltas) with the hidden layers behind the first as ONE
    launch each on the chain's packed operands (bhg_mlp_forward_packed / _backward_packed) against round 4's split-K GEMM + reduce
    pairs: same arrays up to the summation order, masks identical away from the ReLU kink, the packed copies what k_pack would have
    produced — and the same hypergradient from the solver that finds its operands already packed."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    arrays, outs = {}, {}
    for arm in ("1", "0"):
        bhg_debug.reset()
        bhg_debug.setenv("BHG_PACKED_PREPARE", arm)
        curr, prev, direction, provider = _mlp_problem(dims, B, ridge=0.3, seed=sum(dims) + B)
        prov = provider("hip")
        prov.prepare()
        st = prov._state
        assert st.packed_prepare == (arm == "1"), "the arm that was asked for did not run"
        buf = st.buf
        arrays[arm] = ([h[:B].clone() for h in buf.h[1:]], [m[:B].clone() for m in buf.mask], [d[:B].clone() for d in buf.delta],
                       buf.ce[:B].clone(), [h[B:].abs().max().item() if h.shape[0] > B else 0.0 for h in buf.h[1:]])
        curr.config = Config(type="cg", cg_iterations=6, cg_alpha=1.0)
        curr.hypergradient_structure = lambda prev_, curr=curr: WeightedCEMLP(
            curr, prev_, layers=list(curr.module.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=0.3, impl="hip", verify=False)
        outs[arm] = _np(hg.cg([0.1 * d for d in direction], curr, prev, False))
    hp, mp, dp, cep, padp = arrays["1"]
    h0, m0, d0, ce0, _ = arrays["0"]
    assert all(v == 0.0 for v in padp), "padding rows of the activations must stay zero"
    for a, b in zip(hp, h0):
        assert (a - b).abs().max().item() <= 2e-5 * (b.abs().max().item() + 1e-30)
    for a, b, hh in zip(mp, m0, h0):
        flips = (a != b)
        assert flips.float().mean().item() <= 1e-4, "masks may differ only where the pre-activation sits on the kink"
        if flips.any():
            assert hh[flips].abs().max().item() <= 1e-5 * hh.abs().max().item()
    for a, b in zip(dp, d0):
        assert (a - b).abs().max().item() <= 5e-5 * (b.abs().max().item() + 1e-30)
    assert (cep - ce0).abs().max().item() <= 1e-5 * ce0.abs().max().item()
    rel, _ = rel_err(outs["1"], outs["0"])
    print(f"packed prepare {dims} B={B}: hypergradient vs split-K prepare {rel:.2e}")
    assert rel <= 5e-5, rel


@pytest.mark.parametrize("dims,B,K", [([512, 256, 256, 64, 10], 100, 4), ([3072, 2048, 1536, 384, 10], 100, 20)], ids=lambda v: str(v))
def test_right_hand_side_read_in_place_is_bit_identical(dims, B, K, bhg_debug):
    """Round 5: the fully projected solver reads the N-sized right-hand side ONCE (iteration 0) — from the caller's own tensors, with the
    MFMA layers' slices of r and p left unwritten by bhg_cg_init_masked — against round 4's copy into r and p: the same bits."""
    lib = _native.load()
    outs = {}
    for arm in ("1", "0"):
        bhg_debug.reset()
        bhg_debug.setenv("BHG_CG_RHS_DIRECT", arm)
        outs[arm] = _run_solver("cg", dims, B, 0.5, K, sum(dims) + B, True, keep=False)[0]
    assert all(np.array_equal(u, v) for u, v in zip(outs["1"], outs["0"]))
    bhg_debug.reset()
    curr, prev, direction, provider = _mlp_problem(dims, B, ridge=0.5, seed=1)
    prov = provider("hip")
    prov.prepare()
    mask = prov._state.cg_state_mask()
    L = len(dims) - 1
    assert mask == sum(1 << (2 * l + 1) for l in range(L)) | (1 << (2 * (L - 1))), "only the biases and the head weight stay N-sized"


def test_withheld_beta_times_out_poisons_the_result_and_is_diagnosed(be, bhg_debug):
    """The in-launch beta exchange of the six-launch CG iteration (poll_beta, csrc/mlp/wskp.inc) is BOUNDED: with the publisher made
    to keep beta to itself (fault injection, debug key lin_withhold_beta) the pollers give up after ~1 s per launch, the solve runs
    to its end with a NaN result — no hung GPU — and HipBackend.check_health names the cause; the next solve is clean again."""
    import time

    dims, B, K = [512, 256, 256, 64, 10], 100, 3   # FOUR layers, batch <= 128: the form with the in-launch exchange (hoist_plan: lin_ok)
    good = _run_solver("cg", dims, B, 0.05, K, 77, True, keep=False)
    be.check_health()
    bhg_debug.setenv("BHG_LIN_WITHHOLD_BETA", "1")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bad = _run_solver("cg", dims, B, 0.05, K, 77, True, keep=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(not np.isfinite(o).all() for o in bad[0]), "a solve whose pollers gave up must not look like a right answer"
    assert dt < 60.0, f"the bounded wait took {dt:.1f} s"
    with pytest.raises(_native.NativeLibraryError, match="gave up"):
        be.check_health()
    bhg_debug.reset()
    again = _run_solver("cg", dims, B, 0.05, K, 77, True, keep=False)
    be.check_health()   # (the word was cleared when it was reported)
    assert all(np.array_equal(u, v) for u, v in zip(again[0], good[0]))
    print(f"withheld beta: {K - 1} iterations x 2 polling launches gave up in {dt:.1f} s, NaN result, diagnosed by check_health")


@pytest.mark.parametrize(
    "dims,B,K",
    [([256, 384, 128, 10], 100, 5), ([512, 256, 256, 64, 10], 128, 4), ([256, 256, 256, 256, 128, 10], 100, 6), ([512, 1024, 64, 10], 200, 3),
     ([64, 96, 64, 32, 64, 96, 32, 64, 10], 50, 4),   # eight layers: the deepest net the hoisted / projected forms take
     ([512, 256, 256, 64, 10], 200, 4), ([256, 256, 128, 64, 10], 256, 5), ([256, 192, 128, 64, 32, 10], 300, 3),   # round 5: batches > 128
     ([256, 256, 192, 640, 10], 100, 5), ([256, 256, 128, 64, 24], 100, 5),   # four layers, a head k_headu does not take (K > 512; C > 12)
     ([256, 256, 128, 64, 100], 100, 4),                                        # round 6: a 100-class head (classes in chunks of 32)
     ([3072, 2048, 1536, 384, 10], 100, 20)],
    ids=lambda v: str(v),
)
@pytest.mark.parametrize("algo", ["cg", "neumann"])
def test_hoisted_and_projected_chain_match_classic_chain(algo, dims, B, K, bhg_debug):
    """Fused solvers, three forms of the R-chain on the same inputs:
      classic    BHG_MLP_HOIST=0: direction products inside the chain, lazy direction mixed in the GEMM loaders;
      hoisted    BHG_MLP_HOIST=1 BHG_MLP_PROJ=0: every direction product h V^T / delta V in ONE grouped launch on the
                 residual (k_hoist), G(p) = G(r) + beta G(p_old), the chain keeps the constant-weight products in the
                 in-workgroup split-K form with G as addend, k_cg_beta's work inside that launch;
      projected  the CG default: after the first iteration G(r) itself comes from batch-sized recurrences through B x B Gram
                 matrices (k_wsk_group, k_hoist with two operand pairs, k_proj_update) — nothing N-sized is read.
    Against each other and against the un-fused loop: same hypergradient; bit-reproducible; launch counters prove which arm
    ran.  (Neumann takes the hoisted form only when asked, BHG_MLP_HOIST=2, and has no projected form.)"""
    lib = _native.load()
    # ridge 0.5 at the full size: with 0.05 the Hessian of this random instance is indefinite and twenty CG iterations turn
    # ANY difference in summation order into an O(1) difference (measured: 1.97 between two correct arms)
    ridge = 2.0 if dims[0] >= 3072 else (0.5 if 640 in dims else 0.05)   # (the 640-wide instance: same remark, measured 3.9e-4 classic vs un-fused at 0.05)
    arms = {"classic": {"BHG_MLP_HOIST": "0"}, "hoisted": {"BHG_MLP_HOIST": "1" if algo == "cg" else "2", "BHG_MLP_PROJ": "0"}}
    if algo == "cg":
        # projected: G(r) by recurrence, r / p still N-sized (what a caller who wants x gets); full: NO N-sized state after the
        # first iteration — the default when the solution vector is not materialised
        arms["projected"] = {"BHG_MLP_HOIST": "1", "BHG_MLP_PROJ": "1"}
        arms["full"] = {"BHG_MLP_HOIST": "1", "BHG_MLP_PROJ": "1", "keep": "0"}
    else:
        # the Neumann default without an accumulator vector: G(v) by recurrence, nothing N-sized after the first iteration,
        # plus the closing half pass that adds Rz(v_K)
        arms["full"] = {"keep": "0"}
    # round 4: the chain through the constant weights runs on packed operands (k_wskp), the per-iteration Gram products as extra
    # workgroups of its launches; round 3's forms (LDS-staged row-major chain, Gram launch of its own) stay as arms
    arms["full-unpacked"] = dict(arms["full"], BHG_PACKED_CHAIN="0")
    arms["full-gramlaunch"] = dict(arms["full"], BHG_PACKED_GRAM="0")
    arms["full-grawv1"] = dict(arms["full"], BHG_GRAW_V2="0")        # Gram products in the chain launches, G(raw) by k_hoist (three slabs)
    arms["full-alphalaunch"] = dict(arms["full"], BHG_ALPHA_IN_HOIST="0")
    arms["full-pstepv1"] = dict(arms["full"], BHG_PSTEP_V2="0")         # k_proj_step instead of k_pstep (same work, lazily fetched arguments)
    arms["hoisted-unpacked"] = dict(arms["hoisted"], BHG_PACKED_CHAIN="0")
    if algo == "neumann":
        # round 5: the default's k_graw applies the update itself (six launches); the arm keeps round 4's update launch.  Same roundings
        # in the same order: the two are compared bit for bit below
        arms["full-updatelaunch"] = dict(arms["full"], BHG_NEUMANN_VNEW="0")
    if algo == "cg":
        # four-layer nets with a batch of <= 128: the chain's first product by linearity (k_wskpl), the recurrences riding in the
        # pre-head launch (k_wskpu) — the arms: update blocks inside k_wskpl; the k_pstep launch; k_graw storing G(raw) instead of
        # applying the residual step (on other nets the three arms run the default's launches)
        arms["full-updfirst"] = dict(arms["full"], BHG_LIN_UPDATE_NEXT="0")
        arms["full-upd-prehead"] = dict(arms["full"], BHG_LIN_UPDATE_IN_HEAD="0")   # round 4's place for the update blocks (k_wskpu)
        arms["full-head-last"] = dict(arms["full"], BHG_HEADU_HEAD_FIRST="0")        # k_headu with the update blocks leading the grid
        arms["full-deep-kpstep"] = dict(arms["full"], BHG_LIN_DEEP="0")   # nets deeper than four layers with round 4's k_pstep launch
        arms["full-kpstep"] = dict(arms["full"], BHG_LIN_FIRST="0")
        arms["full-grawraw"] = dict(arms["full"], BHG_RNEW_IN_GRAW="0")
    arms["full-unpaired"] = dict(arms["full"], BHG_XCD_PAIRS="0")       # column tiles of a strip's 64 columns on two XCDs (before round 5)
    arms["full-no-kloop"] = dict(arms["full"], BHG_GRAW_KLOOP="0")      # batches beyond 128 with round 3's closing launches
    out = {}
    for name, env in arms.items():
        bhg_debug.reset()
        for k in ("BHG_MLP_HOIST", "BHG_MLP_PROJ"):
            bhg_debug.delenv(k, raising=False)
        for k, v in env.items():
            if k != "keep":
                bhg_debug.setenv(k, v)
        # this test is about the FORMS: the cost model that gates projection by batch size / layer widths (its own test:
        # test_projection_is_gated_by_batch_size) must not pick the arm here — the eight-layer net is 32-96 wide
        bhg_debug.setenv("BHG_PROJ_MAX_RATIO", "100000000")
        keep = env.get("keep", "1") == "1"
        h0, p0, l0 = lib.bhg_mlp_hoist_launches(), lib.bhg_mlp_proj_iterations(), lib.bhg_mlp_lin_launches()
        out[name] = _run_solver(algo, dims, B, ridge, K, sum(dims) + B, True, keep=keep)
        dh, dp = lib.bhg_mlp_hoist_launches() - h0, lib.bhg_mlp_proj_iterations() - p0
        if algo == "cg" and name == "full":
            # round 5: every net of >= 4 layers takes the SIX-launch form (k_wskpl once per iteration) — the deeper ones with their update
            # blocks behind the publisher inside that launch, batches beyond 128 with the K-looped closing launch; 3-layer nets do not
            want_lin = K if len(dims) - 1 >= 4 else 0   # (any batch: k_graw's K-looped instance takes padded batches beyond 128)
            assert lib.bhg_mlp_lin_launches() - l0 == want_lin, (dims, B, lib.bhg_mlp_lin_launches() - l0, want_lin)
        want = {"classic": (0, 0), "hoisted": (K, 0), "projected": (1, K - 1), "full": (1, K - 1 if algo == "cg" else K)}[name.split("-")[0]]
        assert (dh, dp) == want, (name, dh, dp, want)
        again = _run_solver(algo, dims, B, ridge, K, sum(dims) + B, True, keep=keep)
        assert all(np.array_equal(u, v) for u, v in zip(again[0], out[name][0])), f"{name}: bit-reproducible"
    unf = _run_solver(algo, dims, B, ridge, K, sum(dims) + B, False)
    tol = 5e-6 if algo == "neumann" else (1e-4 if K >= 20 else 5e-5)
    for name in arms:
        rel_c, _ = rel_err(out[name][0], out["classic"][0])
        rel_u, _ = rel_err(out[name][0], unf[0])
        print(f"{algo} {dims} K={K} {name:9s}: vs classic chain {rel_c:.2e}, vs un-fused {rel_u:.2e}")
        assert rel_c <= tol and rel_u <= tol, (name, rel_c, rel_u)
    if algo == "neumann" and "full-updatelaunch" in out:
        same = all(np.array_equal(u, v) for u, v in zip(out["full"][0], out["full-updatelaunch"][0]))
        rel_v, _ = rel_err(out["full"][0], out["full-updatelaunch"][0])
        print(f"neumann {dims} K={K}: update inside k_graw vs update launch: {'bit-identical' if same else 'rel %.2e' % rel_v}")
        assert rel_v <= 1e-6, rel_v


@pytest.mark.parametrize("dims", [[256, 256, 192, 640, 10], [256, 256, 128, 64, 24], [256, 256, 192, 640, 24]], ids=lambda v: str(v))
def test_four_layer_nets_outside_the_head_launch_form(dims):
    """PRODUCT build, no key set: four-layer nets whose head k_headu does not take — a last hidden layer wider than 512, more than
    12 classes — keep their update blocks in the pre-head launch (k_wskpu; `lin_head` in cg_ctx_init).  Against the un-fused loop,
    bit-reproducible."""
    assert not _native.is_ab()
    lib = _native.load()
    B, K, ridge = 100, 5, 0.5   # (ridge 0.05 leaves the 640-wide random instance close to indefinite: CG turns summation order into 4e-4)
    l0 = lib.bhg_mlp_lin_launches()
    got, _ = _run_solver("cg", dims, B, ridge, K, sum(dims) + B, True, keep=False)
    lin = lib.bhg_mlp_lin_launches() - l0
    again, _ = _run_solver("cg", dims, B, ridge, K, sum(dims) + B, True, keep=False)
    unf, _ = _run_solver("cg", dims, B, ridge, K, sum(dims) + B, False)
    rel, _ = rel_err(got, unf)
    print(f"cg {dims} K={K}: k_wskpl launches {lin}, fused vs un-fused {rel:.2e}")
    assert all(np.array_equal(u, v) for u, v in zip(again, got)), "bit-reproducible"
    assert rel <= 5e-5, (dims, rel)


@pytest.mark.parametrize("algo", ["cg", "neumann"])
@pytest.mark.parametrize("ridge,alpha,K", [(0.05, 0.5, 4), (0.0, None, 3), (0.05, None, 1), (0.05, None, 2), (0.3, 0.25, 7)],
                         ids=lambda v: str(v))
def test_projected_solvers_edge_cases(algo, ridge, alpha, K, bhg_debug):
    """The default (fully projected CG / projected Neumann, no solution vector) against the classic chain where the recurrences
    could go wrong: cg_alpha != 1 (the reference's quirk: the step length uses cg_alpha * Hp, the residual update the
    un-scaled Hp, cg.py:42-50), no ridge (shift = 0), K = 1 (no recurrence at all), K = 2 (one), a batch that fills two
    128-row tiles, and an odd iteration count."""
    lib = _native.load()
    for dims, B in (([256, 384, 128, 10], 100), ([512, 256, 256, 64, 10], 200)):
        outs = {}
        for arm in ("0", "1"):
            bhg_debug.setenv("BHG_MLP_HOIST", arm)
            bhg_debug.setenv("BHG_PROJ_MAX_RATIO", "100000000")   # (the recurrences are under test, not the cost model that gates them)
            p0 = lib.bhg_mlp_proj_iterations()
            outs[arm], _ = _run_solver(algo, dims, B, ridge, K, 5 + K, True, alpha=alpha, keep=False)
            projected = lib.bhg_mlp_proj_iterations() - p0
            assert projected == (0 if arm == "0" else (K - 1 if algo == "cg" else K)), (arm, projected)
        rel, _ = rel_err(outs["1"], outs["0"])
        print(f"{algo} {dims} B={B} ridge={ridge} alpha={alpha} K={K}: projected vs classic {rel:.2e}")
        assert rel <= 5e-5, (algo, dims, B, ridge, alpha, K, rel)


def test_wsk_defaults_per_solver(bhg_debug):
    """Default (no BHG_MLP_WSK): the fused CG solver takes the in-workgroup form for reductions of <= 1024 k (mode 2); the
    un-fused CG chain never does; the Neumann solver uses mode 3 (short reductions direct, long R-backward LDS-staged)
    in BOTH arms — the un-fused loop asks for it through bhg_mlp_hvp_mode — so they stay bitwise equal."""
    bhg_debug.delenv("BHG_MLP_WSK", raising=False)
    bhg_debug.setenv("BHG_MLP_HOIST", "0")   # the classic chain (the hoisted form has its own test below)
    lib = _native.load()
    dims, B = [256, 384, 128, 10], 100     # per iteration: R-forward of layer 0 (256 k) and R-backward into it (2 x 128 k)
    n0 = lib.bhg_mlp_wsk_launches()
    _run_solver("cg", dims, B, 0.05, 3, 11, False)
    assert lib.bhg_mlp_wsk_launches() == n0
    _run_solver("cg", dims, B, 0.05, 3, 11, True)
    n1 = lib.bhg_mlp_wsk_launches()
    assert n1 == n0 + 2 * 3
    _, st_f = _run_solver("neumann", dims, B, 0.05, 3, 11, True)
    n2 = lib.bhg_mlp_wsk_launches()
    _, st_u = _run_solver("neumann", dims, B, 0.05, 3, 11, False)
    n3 = lib.bhg_mlp_wsk_launches()
    assert n2 - n1 == 2 * 3 and n3 - n2 == 2 * 3
    assert np.array_equal(st_f[0], st_u[0])
    # a long R-backward reduction (2 x 1024 k) in the Neumann solver: the LDS-staged form, still bitwise equal to the un-fused arm
    dims2 = [256, 1024, 1024, 10]
    _, s_f = _run_solver("neumann", dims2, B, 0.05, 2, 12, True)
    _, s_u = _run_solver("neumann", dims2, B, 0.05, 2, 12, False)
    assert np.array_equal(s_f[0], s_u[0])


def test_cg_variants_may_alternate_inside_a_solve(be):
    """A streamed iteration credits the resident kernel's arrival counter (k_cg_dir), so the public ABI's explicit
    variant argument may change between iterations: stream/resident alternating == all-stream."""
    if not be.lib.bhg_cg_resident_ok():
        pytest.skip("resident kernel not eligible on this device")
    g = torch.Generator().manual_seed(5)
    sizes = [100_000, 257, 40_000]
    vec = [torch.randn(n, generator=g).to(DEV) for n in sizes]
    diag = [(1.0 + torch.rand(n, generator=g)).to(DEV) for n in sizes]
    lay = be.layout(vec)
    res = []
    for pattern in ([1, 1, 1, 1, 1, 1], [1, 2, 1, 2, 2, 1], [2, 1, 1, 2, 1, 2]):
        x, r, p = lay.state(3)
        be.cg_init(lay, vec, x, r, p)
        for k, v in enumerate(pattern):
            hvp = [d * q for d, q in zip(diag, lay.views(p, vec))]
            be.cg_step(lay, hvp, x, r, p, 1.0, k, out_scale=(-1.0 if k == len(pattern) - 1 else 0.0), variant=v)
            lay._cg_variant = None   # the Python wrapper pins the variant per solve; this test drives the ABI directly
        assert not be.cg_barrier_timed_out(lay)
        res.append(x.clone())
    for other in res[1:]:
        rel = (other - res[0]).norm() / res[0].norm()
        assert torch.isfinite(other).all() and rel <= 1e-5, rel


def test_mlp_hvp_full_size_cfg2():
    """BASELINE cfg-2 shapes (N = 10,034,826, batch 100): MFMA HVP vs the ATen closed form, and
    linearity H(a u + b v) = a H u + b H v as a size-independent property."""
    dims, B = [3072, 2048, 1536, 384, 10], 100
    curr, prev, direction, provider = _mlp_problem(dims, B, ridge=1e-2, seed=1)
    hvp = provider("hip").prepare()
    got = [t.clone() for t in hvp(direction)]
    ref = provider("torch").prepare()(direction)
    rel, mx = rel_err(_np(got), _np(ref))
    assert rel <= 1e-5 and mx <= 1e-4, (rel, mx)
    g = torch.Generator().manual_seed(9)
    other = [torch.randn(d.shape, generator=g).to(DEV) for d in direction]
    h2 = [t.clone() for t in hvp(other)]
    comb = [t.clone() for t in hvp([0.5 * u - 2.0 * v for u, v in zip(direction, other)])]
    rel, _ = rel_err(_np(comb), _np([0.5 * a - 2.0 * b for a, b in zip(got, h2)]))
    assert rel <= 1e-5, rel
    # symmetry of the Hessian, <u, H v> = <v, H u>, dots in fp64: another property that needs no reference run
    uHv = sum((u.double() * hv.double()).sum().item() for u, hv in zip(other, got))
    vHu = sum((v.double() * hu.double()).sum().item() for v, hu in zip(direction, h2))
    assert abs(uHv - vHu) <= 1e-5 * max(abs(uHv), abs(vHu)), (uHv, vHu)


@pytest.mark.parametrize("algo,K", [("cg", 20), ("neumann", 10)])
def test_fused_solver_power_of_two_scaling_full_size(algo, K):
    """BASELINE cfg-2 shapes: the solvers are linear in the right-hand side and every operation on the way commutes
    with a power-of-two scale (products, sums, the fp32 roundings, the step-length quotients), so
    solve(4 v) == 4 solve(v) BIT FOR BIT — a full-size check that needs no reference and no tolerance."""
    import bench

    curr, prev, vector = bench.build(torch.device(DEV), seed=0, K=K, algo=algo)
    bench.declare_structure(curr, "hip", fused=True)
    one = [t.clone() for t in hg.jvp_fn_mapping[algo](vector, curr, prev, False)]
    four = hg.jvp_fn_mapping[algo]([4.0 * v for v in vector], curr, prev, False)
    for a, b in zip(one, four):
        assert torch.isfinite(b).all()
        assert torch.equal(4.0 * a, b), (algo, (4.0 * a - b).abs().max().item())


# ------------------------------------------------------------------------------------------------
# analytic logistic-regression HVP (csrc/bhg_logreg.hip), BASELINE cfg 1
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d", [(500, 100), (37, 5), (4096, 257), (1, 1)])
def test_logreg_hvp_kernels_vs_autograd(n, d):
    lib = _native.load()
    g = torch.Generator().manual_seed(n + d)
    X = torch.randn(n, d, generator=g).to(DEV)
    w = (0.3 * torch.randn(d, generator=g)).to(DEV).requires_grad_(True)
    y = (torch.rand(n, generator=g) < 0.5).float().to(DEV)
    lam = (0.5 + torch.rand(d, generator=g)).to(DEV)
    p = torch.randn(d, generator=g).to(DEV)
    s = torch.empty(n, device=DEV)
    tmp = torch.empty(int(lib.bhg_logreg_tmp_floats(n, d)), device=DEV)
    out = torch.empty(d, device=DEV)
    st = int(torch.cuda.current_stream().cuda_stream)
    _native.check(lib.bhg_logreg_prepare(X.data_ptr(), w.data_ptr(), s.data_ptr(), n, d, st), "prepare")
    _native.check(lib.bhg_logreg_hvp(X.data_ptr(), s.data_ptr(), lam.data_ptr(), p.data_ptr(), out.data_ptr(), tmp.data_ptr(), n, d, st), "hvp")
    loss = torch.nn.functional.binary_cross_entropy_with_logits(X @ w, y) + 0.5 * (lam * w * w).sum()
    (gr,) = torch.autograd.grad(loss, w, create_graph=True)
    (want,) = torch.autograd.grad(gr, w, grad_outputs=p)
    # fp32 GEMV pair vs autograd's fp32 mm chain; rtol 2e-5 of the largest entry
    assert (out - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("name", ["logreg_cg5", "logreg_cg3_a01", "logreg_neumann5", "logreg_cg0"])
@pytest.mark.parametrize("sync", [False, True])
def test_structured_logreg_path_matches_reference(name, sync, be):
    case = zoo.CASE_BY_NAME[name]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    zoo.attach_logreg_structure(curr)
    out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
    if sync:
        assert out is None
        out = [p.grad for p in prev.trainable_parameters()]
    rel, mx = rel_err(_np(out), golden_list(outputs, case.name, "fp32"))
    assert rel <= 1e-4 and mx <= 1e-3, (rel, mx)


# ------------------------------------------------------------------------------------------------
# robustness / edge cases
# ------------------------------------------------------------------------------------------------
def test_resident_census_passes_on_a_dedicated_gpu(be):
    assert be.lib.bhg_cg_resident_ok() == 1
    assert be.lib.bhg_cg_resident_capacity_chunks() >= 2453  # the BASELINE cfg-2 vector fits


def test_zero_sized_and_odd_inputs(be):
    """Empty tensors inside a list, non-contiguous and fp16 inputs (converted), 1-element tensors."""
    gen = torch.Generator().manual_seed(1)
    sizes = [0, 5, 0, 4097, 1]
    vec = _rand_list(sizes, gen)
    lay = be.layout(vec)
    assert lay.n_chunks == 4 and lay.total == sum(sizes)
    v, p = lay.new_flat(), lay.new_flat()
    be.neumann_init(lay, vec, v, p)
    hv = _rand_list(sizes, gen)
    hv[3] = hv[3].to(torch.float16)  # converted to fp32 by the backend
    base = torch.randn(10, generator=gen).to(DEV)
    hv[1] = base[::2]  # non-contiguous view of 5 elements
    be.neumann_step(lay, hv, v, p, 0.5)
    want_v = torch.cat([a - 0.5 * b.float().reshape(-1) for a, b in zip(vec, hv)])
    got_v = torch.cat([v[s : s + n] for s, n in zip(lay.starts, lay.numels)])
    assert torch.equal(got_v, want_v)
    # an entirely empty problem is a no-op
    empty = [torch.empty(0, device=DEV)]
    le = be.layout(empty)
    be.cg_init(le, empty, le.new_flat(), le.new_flat(), le.new_flat())
    be.cg_step(le, empty, le.new_flat(), le.new_flat(), le.new_flat(), 1.0, 0)


@pytest.mark.parametrize("sizes", RAGGED[:5], ids=lambda s: f"T{len(s)}_N{sum(s)}")
def test_sama_precondition_kernel_vs_aten(sizes, be):
    gen = torch.Generator().manual_seed(5 + sum(sizes))
    vec = _rand_list(sizes, gen)
    g = _rand_list(sizes, gen, scale=0.01)
    m = [0.1 * a + 0.005 * b for a, b in zip(g, _rand_list(sizes, gen))]
    u = [0.001 * a * a + 1e-5 * (1.0 + torch.rand(a.shape, generator=gen).to(DEV)) for a in g]
    lay = be.layout(vec)
    out = lay.new_flat()
    b1, b2, eps, lr = 0.9, 0.999, 1e-8, 1e-3
    be.sama_adam_precondition(lay, vec, g, m, u, out, b1, b2, eps, lr)
    got = _flat_concat(lay, out)
    # betty/hypergradient/utils.py:50-59 with ATen ops on the same device (fp32)
    want = []
    for v_, g_, m_, u_ in zip(vec, g, m, u):
        m_old = (m_ - (1 - b1) * g_) / b1
        u_old = (u_ - (1 - b2) * g_ * g_) / b2
        sc = (1 - b1) * b2 * u_old - b1 * (1 - b2) * g_ * m_old
        sc = sc / (torch.sqrt(u_) + eps) ** 3
        want.append((v_ * sc * lr).cpu().numpy())
    want = np.concatenate(want)
    # same formula and rounding sequence; sqrt/div/pow may differ in the last ulp between ATen and the kernel
    np.testing.assert_allclose(got, want, rtol=3e-6, atol=1e-12)


def test_set_grads_multi_tensor_accumulate(be):
    """Problem.set_grads (problem.py:583-597) on many tensors = one multi-tensor accumulate."""
    from betty_amd.problems import ImplicitProblem

    module = zoo.MLP([12] + [12] * 5 + [4]).to(DEV)  # 12 tensors
    prob = ImplicitProblem(name="p", module=module, config=Config())
    params = prob.trainable_parameters()
    gen = torch.Generator().manual_seed(0)
    g1 = [torch.randn(p.shape, generator=gen).to(DEV) for p in params]
    g2 = [torch.randn(p.shape, generator=gen).to(DEV) for p in params]
    g2[3] = None  # skipped (allow_unused)
    g1_copy = [t.clone() for t in g1]
    prob.set_grads(params, g1)  # assigns (p.grad IS g1[i], as in the reference)
    prob.set_grads(params, g2)  # accumulates: two multi-tensor launches, out of place
    for p, a, a0, b in zip(params, g1, g1_copy, g2):
        assert torch.equal(a, a0), "set_grads must not mutate the caller's tensors (problem.py:594 is out of place)"
        want = a if b is None else a + b
        assert torch.equal(p.grad, want)


def test_cg_hybrid_resident_instance_at_20M(be):
    """2 x the cfg-2 tensor list (N = 20 M) exceeds every all-on-chip instance: BHG_CG_AUTO takes the HYBRID resident
    kernel (14 resident slots per workgroup + the remaining chunks streamed inside the same launch); it must solve
    the 4-eigenvalue diagonal system exactly, bit-reproducibly, agree with the 3-kernel streaming form, never time out."""
    sizes = CFG2_SIZES * 2
    gen = torch.Generator().manual_seed(8)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in sizes]
    dvals = torch.tensor([0.5, 1.0, 2.0, 4.0], device=DEV)
    diag = [dvals[torch.arange(n, device=DEV) % 4] for n in sizes]
    lay = be.layout(vec)
    cap = be.lib.bhg_cg_resident_capacity_chunks()
    assert cap >= 5000 or cap == 0
    if not be.lib.bhg_cg_resident_ok() or lay.n_chunks > cap:
        pytest.skip("hybrid resident instance not eligible on this device")
    outs = []
    for variant in (_native.BHG_CG_AUTO, _native.BHG_CG_RESIDENT, _native.BHG_CG_STREAM):
        x, r, p = lay.state(3)
        be.cg_init(lay, vec, x, r, p)
        K = 4
        for k in range(K):
            hv = [d * t for d, t in zip(diag, lay.views(p, vec))]
            be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == K - 1 else 0.0), variant=variant)
            if k == 0 and variant == _native.BHG_CG_AUTO:
                assert lay._cg_variant == _native.BHG_CG_RESIDENT
            lay._cg_variant = None if variant != _native.BHG_CG_AUTO else lay._cg_variant
        assert not be.cg_barrier_timed_out(lay)
        outs.append(x.clone())
        for xv, v, d in zip(lay.views(x, vec), vec, diag):
            want = -(v / d)
            assert (xv - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    assert torch.equal(outs[0], outs[1]), "the hybrid resident kernel must be bit-reproducible"
    rel = (outs[0] - outs[2]).norm() / outs[2].norm()
    assert rel <= 1e-6, rel


def test_cg_auto_falls_back_to_stream_beyond_resident_capacity(be):
    """4 x the cfg-2 tensor list (N = 40 M): less than half of the vector would be resident, BHG_CG_AUTO must take the
    3-kernel streaming form and still solve the 4-eigenvalue diagonal system exactly; forcing the resident variant
    must be refused with BHG_ERR_CAPACITY, not mis-computed."""
    sizes = CFG2_SIZES * 4
    gen = torch.Generator().manual_seed(8)
    vec = [torch.randn(n, generator=gen).to(DEV) for n in sizes]
    dvals = torch.tensor([0.5, 1.0, 2.0, 4.0], device=DEV)
    diag = [dvals[torch.arange(n, device=DEV) % 4] for n in sizes]
    lay = be.layout(vec)
    assert lay.n_chunks > be.lib.bhg_cg_resident_capacity_chunks()
    x, r, p = lay.state(3)
    be.cg_init(lay, vec, x, r, p)
    with pytest.raises(_native.NativeLibraryError, match="capacity"):
        be.cg_step(lay, [d * t for d, t in zip(diag, lay.views(p, vec))], x, r, p, 1.0, 0, variant=_native.BHG_CG_RESIDENT)
    lay._cg_variant = None
    K = 4
    for k in range(K):
        hv = [d * t for d, t in zip(diag, lay.views(p, vec))]
        be.cg_step(lay, hv, x, r, p, 1.0, k, out_scale=(-1.0 if k == K - 1 else 0.0), variant=_native.BHG_CG_AUTO)
    assert lay._cg_variant == _native.BHG_CG_STREAM
    for xv, v, d in zip(lay.views(x, vec), vec, diag):
        want = -(v / d)
        assert (xv - want).abs().max().item() <= 2e-5 * want.abs().max().item()


@pytest.mark.parametrize("sync", [False, True])
def test_structured_proximal_path_matches_reference(sync, be):
    case = zoo.CASE_BY_NAME["imaml_cg10"]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
    zoo.attach_prox_structure(curr)
    out = hg.cg(vector, curr, prev, sync)
    if sync:
        assert out is None
        out = [p.grad for p in prev.trainable_parameters()]
    rel, mx = rel_err(_np(out), golden_list(outputs, case.name, "fp32"))
    assert rel <= 1e-4 and mx <= 1e-3, (rel, mx)



_TIMEOUT_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from betty_amd import _native
from betty_amd.backend import get_backend
be = get_backend()
if {limit!r} is not None:
    _native.use_ab(True)   # (fault injection lives in the measurement build)
    _native.debug_set("cg_spin_limit", int({limit!r}))
dev = torch.device("cuda:0")
vec = [torch.randn(600 * 4096, device=dev)]          # every one of the 256 workgroups owns chunks
hv = [2.0 * vec[0]]
lay = be.layout(vec)
x, r, p = lay.state(3)
be.cg_init(lay, vec, x, r, p)
be.cg_step(lay, hv, x, r, p, 1.0, 0, 0.0, variant=_native.BHG_CG_RESIDENT)
torch.cuda.synchronize()
print("TIMED_OUT", int(be.cg_barrier_timed_out(lay)), "NAN", int(torch.isnan(x).any().item()))
"""


@pytest.mark.gpu
@pytest.mark.parametrize("limit,expect", [("0", (1, 1)), (None, (0, 0))])
def test_resident_barrier_timeout_poisons_the_result(limit, expect):
    """A grid barrier that gives up (GPU shared mid-run) must not return a plausible wrong answer:
    the flag is raised AND the iterate is NaN.  bhg_debug_set("cg_spin_limit", 0) forces the time-out (in a process of its
    own: a timed-out barrier leaves the resident kernel's counters in an unusable state)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    out = subprocess.run([sys.executable, "-c", _TIMEOUT_SCRIPT.format(root=root, limit=limit)], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("TIMED_OUT")][-1].split()
    assert (int(line[1]), int(line[3])) == expect, out.stdout


# ------------------------------------------------------------------------------------------------
# BASELINE.json cfg 3 at full size: ResNet-12 inner (8.0 M parameters, 50 tensors), prox-regularised to
# the upper copy (M = N), CG K = 20 — checker = the oracle's restatement of cg.py running on the same
# device tensors (per-tensor ATen, i.e. what the reference itself would launch on this GPU)
# ------------------------------------------------------------------------------------------------
def _resnet12_case(cfg, dtype=torch.float32):
    """(Every tensor is drawn in fp32 and then cast: the float64 case is the same problem.)"""
    g = torch.Generator().manual_seed(77)
    torch.manual_seed(77)
    inner, upper = zoo.ResNet12().to(DEV), zoo.ResNet12().to(DEV)
    for p, q in zip(inner.parameters(), upper.parameters()):
        q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g).to(DEV))
    inner, upper = inner.to(dtype), upper.to(dtype)
    x = torch.randn(25, 3, 84, 84, generator=g).to(DEV, dtype)   # 5-way 5-shot support set at the example's 84 x 84
    y = torch.arange(5).repeat_interleave(5).to(DEV)
    vector = [(0.01 * torch.randn(p.shape, generator=g)).to(DEV, dtype) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(**cfg), loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))
    return curr, prev, vector


@contextlib.contextmanager
def _deterministic_convolutions():
    """cfg 3 runs under MIOpen's deterministic solver filter.  With the default solvers the REFERENCE'S OWN algorithm has two outcomes
    on this instance: about one run in sixteen — of the checker (pure PyTorch: no kernel of this package runs) as of the product, which
    share autograd's convolution double backward — lands 4.4e-3 from the others, always the same second answer, while one
    Hessian-vector product repeats to 1.4e-6 (tests/probe_cfg3_flake.py; profiles/r05_cfg3_two_outcomes_probe.log: checker 1 / 17,
    resident 2 / 16, stream 0 / 16).  Under the filter checker, resident and stream kernels are bit-reproducible run to run
    (profiles/r05_cfg3_two_outcomes_probe_deterministic_convs.log: 66 solves, no second outcome) at the same speed."""
    was = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        yield
    finally:
        torch.backends.cudnn.deterministic = was


@pytest.fixture(scope="module")
def resnet12_checker():
    """The checker's answer for cfg 3 (oracle restatement on the device: opaque double backward + per-tensor ATen
    recurrence), computed ONCE for both kernel variants, and its own run-to-run spread."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    with _deterministic_convolutions():
        curr, prev, vector = _resnet12_case(dict(type="cg", cg_iterations=20, cg_alpha=1.0))
        want = _np(horc.cg(vector, curr, prev, False))
        again = _np(horc.cg(vector, curr, prev, False))
    noise, _ = rel_err(again, want)
    return want, noise


@pytest.mark.parametrize("variant", ["resident", "stream"])
def test_cfg3_resnet12_cg20(variant, be, resnet12_checker):
    """BASELINE cfg 3 at the example's own shape: ResNet12(5, 32) = 10,430,533 parameters in 122 tensors, 25 support
    images of 84 x 84, proximal term to the upper copy, CG K = 20 (examples/implicit_maml/main.py:87-92,122-129)."""
    want, noise = resnet12_checker
    curr, prev, vector = _resnet12_case(dict(type="cg", cg_iterations=20, cg_alpha=1.0))
    assert len(vector) == 122 and sum(v.numel() for v in vector) == 10_430_533
    be.cg_variant = VARIANTS[variant]
    try:
        with _deterministic_convolutions():
            got = hg.jvp_fn_mapping["cg"](vector, curr, prev, False)
    finally:
        be.cg_variant = _native.BHG_CG_AUTO
    rel, mx = rel_err(_np(got), want)
    # rtol 1e-4 (north_star) unless the convolution double backward is itself noisier than that on this box
    tol = max(1e-4, 20 * noise)
    print(f"resnet12 cg20 [{variant}]: rel={rel:.2e} max/max={mx:.2e} checker-noise={noise:.2e}")
    assert rel <= tol and mx <= 10 * tol, (variant, rel, mx, noise)
    assert not be.cg_barrier_timed_out(be.layout(vector))


def test_cfg3_resnet12_cg20_with_declared_batchnorm_layers(be, resnet12_checker):
    """Round 6: the same solve with the inner network's 40 batch-norm layers DECLARED (betty_amd.nn.fuse_batchnorm_): their share of every
    Hessian-vector product is bhg_bn_backward_vjp (two launches per layer, fp64 arithmetic inside) instead of ATen's ~340-launch
    decomposition of batch norm's double backward.  Same checker (the reference's algorithm on plain nn.BatchNorm2d), same tolerance as
    the undeclared variants above; the counters prove the fused node ran once per layer and product.  (Measured: one product 2.1e-6 from
    ATen's, the K = 20 solve 5.3e-6 from the checker — profiles/r06_cfg3_declared_batchnorm_vs_float64.txt, which also shows that BOTH
    fp32 products sit 7.8e-3 from the float64 one: the instance's own fp32 floor, shared by every implementation.)"""
    from betty_amd import nn as bnn

    want, noise = resnet12_checker
    K = 20
    curr, prev, vector = _resnet12_case(dict(type="cg", cg_iterations=K, cg_alpha=1.0))
    n_bn = bnn.fuse_batchnorm_(curr.module)
    assert n_bn == 40 == sum(isinstance(m, torch.nn.BatchNorm2d) for m in curr.module.modules())
    c0 = bnn.fused_batchnorm_calls()
    with _deterministic_convolutions():
        got = hg.jvp_fn_mapping["cg"](vector, curr, prev, False)
    calls = {k: v - c0[k] for k, v in bnn.fused_batchnorm_calls().items()}
    # K products through every declared layer (the mixed second derivative of this loss — a proximal term — does not pass through them)
    assert calls["forward"] == n_bn and calls["backward_vjp"] == K * n_bn, (calls, n_bn)
    rel, mx = rel_err(_np(got), want)
    tol = max(1e-4, 20 * noise)
    print(f"resnet12 cg20 [declared batch norm, {n_bn} layers]: rel={rel:.2e} max/max={mx:.2e} checker-noise={noise:.2e}")
    assert rel <= tol and mx <= 10 * tol, (rel, mx, noise)


@pytest.mark.parametrize("radius", [0.01, 1.0])   # 0.01 = Config's default darts_alpha; 1.0 = well above fp32 resolution
def test_cfg4_roberta_scale_darts(radius, be):
    import copy

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    g = torch.Generator().manual_seed(91)
    torch.manual_seed(91)
    inner, upper = zoo.TokenClassifier().to(DEV), zoo.MWN(100).to(DEV)
    tokens = torch.randint(0, 50265, (8, 128), generator=g).to(DEV)
    labels = torch.randint(0, 2, (8,), generator=g).to(DEV)
    vector = [1e-3 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]
    assert sum(v.numel() for v in vector) == 124_055_810
    cfg = dict(type="darts", darts_alpha=radius)

    def problems(inner_m, upper_m):
        prev = zoo.StubProblem("upper", upper_m, config=Config())
        curr = zoo.StubProblem("inner", inner_m, config=Config(**cfg), loss_fn=zoo.make_reweight_loss(prev, 0.0),
                               batch=(tokens, labels))
        return curr, prev

    # truth: the same algorithm in fp64 (the finite difference perturbs 124 M weights by ~1e-6 each, a few
    # ulps of an fp32 weight, so the fp32 result is noise-limited for ANY implementation)
    curr64, prev64 = problems(copy.deepcopy(inner).double(), copy.deepcopy(upper).double())
    truth = horc.darts([v.double() for v in vector], curr64, prev64, False)
    del curr64, prev64
    # the reference's algorithm in fp32 on this device (per-tensor ATen) ...
    w_before = [p.data.clone() for p in inner.parameters()]
    curr, prev = problems(inner, upper)
    want = horc.darts(vector, curr, prev, False)
    for p, w in zip(inner.parameters(), w_before):      # start both runs from identical weights
        p.data.copy_(w)
    # ... and the HIP path
    got = hg.jvp_fn_mapping["darts"](vector, curr, prev, False)
    e_ref, _ = rel_err(_np(want), _np(truth))
    e_got, _ = rel_err(_np(got), _np(truth))
    rel, mx = rel_err(_np(got), _np(want))
    print(f"roberta-scale darts R={radius}: vs fp64 truth: reference-fp32 {e_ref:.2e}, hip {e_got:.2e}; "
          f"hip vs reference-fp32 {rel:.2e}")
    # the HIP result differs from the reference's fp32 result by no more than that differs from the truth,
    # and is within 2x of its distance to the truth (2e-3 floor = the golden darts cases' Case.rtol)
    assert rel <= max(2e-3, e_ref), (rel, e_ref)
    assert e_got <= max(2e-3, 2.0 * e_ref), (e_got, e_ref)
    # perturb / restore drift of the 124 M weights: three roundings, each <= 1/2 ulp of the weight
    drift = max(((p.data - w).abs() / w.abs().clamp_min(1e-3)).max().item() for p, w in zip(inner.parameters(), w_before))
    assert drift <= 2.4e-7, drift


def test_cfg4_roberta_base_as_named_darts(be):
    """BASELINE cfg 4 AS NAMED: ``RobertaForSequenceClassification`` (roberta-base from config: 124,647,170 parameters in
    201 tensors — examples/bert_data_reweighting/model.py:11-32), reweighting net 1-500-1 (main.py:96-100), 16 x 50 tokens
    (+ mask, segments, labels), finite-difference DARTS hypergradient.  Same three-way comparison as the stand-in above."""
    import copy

    pytest.importorskip("transformers")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    g = torch.Generator().manual_seed(17)
    torch.manual_seed(17)
    inner, upper = zoo.RobertaInner().to(DEV), zoo.MWN(500).to(DEV)
    n_par = sum(p.numel() for p in inner.parameters())
    assert n_par == 124_647_170 and len(list(inner.parameters())) == 201
    assert sum(p.numel() for p in upper.parameters()) == 1_501
    B, S = 16, 50
    batch = (torch.randint(3, 50264, (B, S), generator=g).to(DEV), torch.ones(B, S, dtype=torch.long, device=DEV),
             torch.zeros(B, S, dtype=torch.long, device=DEV), torch.randint(0, 2, (B,), generator=g).to(DEV))
    vector = [1e-3 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]
    radius = 1.0   # well above fp32 resolution of a 124 M-element perturbation (see the stand-in test for the default 0.01)

    def problems(inner_m, upper_m):
        prev = zoo.StubProblem("upper", upper_m, config=Config())
        curr = zoo.StubProblem("inner", inner_m, config=Config(type="darts", darts_alpha=radius),
                               loss_fn=zoo.make_roberta_reweight_loss(prev), batch=batch)
        return curr, prev

    curr64, prev64 = problems(copy.deepcopy(inner).double(), copy.deepcopy(upper).double())
    truth = horc.darts([v.double() for v in vector], curr64, prev64, False)
    del curr64, prev64
    w_before = [p.data.clone() for p in inner.parameters()]
    curr, prev = problems(inner, upper)
    want = horc.darts(vector, curr, prev, False)
    for p, w in zip(inner.parameters(), w_before):
        p.data.copy_(w)
    got = hg.jvp_fn_mapping["darts"](vector, curr, prev, False)
    e_ref, _ = rel_err(_np(want), _np(truth))
    e_got, _ = rel_err(_np(got), _np(truth))
    rel, _ = rel_err(_np(got), _np(want))
    print(f"RobertaForSequenceClassification (124,647,170 params / 201 tensors) darts R={radius}: vs fp64 truth: reference-fp32 "
          f"{e_ref:.2e}, hip {e_got:.2e}; hip vs reference-fp32 {rel:.2e}")
    assert rel <= max(2e-3, e_ref), (rel, e_ref)
    assert e_got <= max(2e-3, 2.0 * e_ref), (e_got, e_ref)
    drift = max(((p.data - w).abs() / w.abs().clamp_min(1e-3)).max().item() for p, w in zip(inner.parameters(), w_before))
    assert drift <= 2.4e-7, drift


# ------------------------------------------------------------------------------------------------
# BASELINE.json cfg 5 shape: mixed-op supernet inner (621 tensors), architecture parameters upper,
# Neumann K = 20
# ------------------------------------------------------------------------------------------------
def _supernet_case(c, cells, batch, hw, K):
    g = torch.Generator().manual_seed(55)
    torch.manual_seed(55)
    inner, upper = zoo.Supernet(c=c, cells=cells).to(DEV), zoo.ArchParams(cells=cells).to(DEV)
    x = torch.randn(batch, 3, hw, hw, generator=g).to(DEV)
    y = torch.randint(0, 10, (batch,), generator=g).to(DEV)
    vector = [1e-2 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=K, neumann_alpha=0.1),
                           loss_fn=zoo.make_supernet_loss(prev, 0.1), batch=(x, y))
    return curr, prev, vector


def test_cfg5_supernet_neumann20(be):
    """BASELINE cfg 5's algorithm (Neumann K = 20) on a 621-tensor mixed-op supernet (reduced: 4 cells, 16 x 16 images)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    curr, prev, vector = _supernet_case(32, 4, 16, 16, 20)
    assert len(vector) == 621
    want = horc.neumann(vector, curr, prev, False)
    again = horc.neumann(vector, curr, prev, False)
    noise, _ = rel_err(_np(again), _np(want))
    got = hg.jvp_fn_mapping["neumann"](vector, curr, prev, False)
    rel, mx = rel_err(_np(got), _np(want))
    print(f"supernet neumann20: rel={rel:.2e} max/max={mx:.2e} checker-noise={noise:.2e}")
    tol = max(1e-4, 20 * noise)
    assert rel <= tol and mx <= 10 * tol, (rel, mx, noise)


# ------------------------------------------------------------------------------------------------
# BASELINE.json cfg 5 AS NAMED: the reference's own DARTS supernet `Network(16, 10, 8)` (1,930,618 parameters in 1,399 tensors)
# and `Architecture(4)` (2 x 14 x 8 = 224), built from the reference's files (staged test-only under oracle/_ref/examples_nas by
# `make -C oracle ref`; examples/neural_architecture_search/model_search.py:129-234,302-317), Neumann K = 20 on the example's
# batch 64 x 3 x 32 x 32 (train_search.py:24) — against the REFERENCE'S OWN CPU RUN (tests/golden/cfg5_as_named.npz, generated by
# tests/golden/make_cfg5_golden.py from /root/reference; zoo.cfg5_as_named_case rebuilds the inputs from the seed, checksums prove it).
# ------------------------------------------------------------------------------------------------
_CFG5_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg5_as_named.npz")


@pytest.mark.skipif(zoo.nas_dir() is None, reason="reference NAS example not staged (make -C oracle ref)")
@pytest.mark.skipif(not os.path.isfile(_CFG5_GOLD), reason="tests/golden/cfg5_as_named.npz not generated")
def test_cfg5_reference_network_16_10_8_neumann20_batch64(be):
    """Product path for this opaque inner problem: k_neumann_step on 1,399 tensors through the device pointer table, the K = 20
    Hessian-vector products and the mixed second derivative by forward-over-reverse passes (inner_problem.hypergradient_hvp, round 5:
    ATen's double backward of a grouped convolution loops over the groups on the host — 18.4 s per product on the MI355X box against
    1.9 s per pass, profiles/r05_cfg5_hvp_conv_modes.txt; round 4 ran this test in 12.5 minutes and kept it opt-in for that reason).
    Tolerance: north_star's rtol 1e-4, or 5x the reference's OWN fp32-vs-fp64 distance on this instance where that is larger (the
    rule of the darts / sama goldens).  The K = 3 solve is also run by the reference's algorithm on this GPU: rtol 1e-4 holds there."""
    import time

    gold = np.load(_CFG5_GOLD)
    curr, prev, vector = zoo.cfg5_as_named_case(Config, DEV)
    n_params, n_tensors = sum(p.numel() for p in curr.parameters()), len(list(curr.parameters()))
    assert (n_params, n_tensors) == (1_930_618, 1_399), (n_params, n_tensors)
    assert sum(p.numel() for p in prev.parameters()) == 224
    np.testing.assert_allclose(zoo.cfg5_checksums(curr, prev, vector), gold["checksum"], rtol=1e-9, atol=1e-9,
                               err_msg="the GPU box did not rebuild the problem the golden was generated from")
    K = zoo.CFG5_K
    want32, want64, spread = gold[f"neumann{K}/fp32"], gold[f"neumann{K}/fp64"], float(gold[f"neumann{K}/ref_spread"])
    curr.hypergradient_hvp = "forward_over_reverse"
    # (929 train-mode batch-norm layers: the package says so once per problem — round 6 — and this network, all grouped convolutions, is
    #  the case the method is for: acknowledged, and checked against the double backward's solve below)
    curr.hypergradient_hvp_ack_batchnorm = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = hg.jvp_fn_mapping["neumann"](vector, curr, prev, False)
    torch.cuda.synchronize()
    t_got = time.perf_counter() - t0
    flat = np.concatenate([g.ravel() for g in _np(got)])
    rel32 = float(np.linalg.norm(flat - want32) / np.linalg.norm(want32))
    rel64 = float(np.linalg.norm(flat - want64) / np.linalg.norm(want64))
    tol = max(1e-4, 5.0 * spread)
    print(f"cfg5 as named: Network(16,10,8) {n_params:,} params / {n_tensors} tensors, Architecture 224, batch 64, neumann K={K}: "
          f"vs reference-CPU fp32 {rel32:.2e}, vs reference fp64 {rel64:.2e} (reference fp32 vs fp64 {spread:.2e}; tolerance {tol:.1e}); "
          f"product {t_got:.1f} s/step")
    assert rel32 <= tol, (rel32, rel64, spread)
    # Round 6 — where north_star's rtol 1e-4 IS decidable for this network.  Not against the CPU output at a shorter
    # horizon: the reference's own fp32-vs-fp64 distance on this instance does not shrink with K (K = 3: 1.25e-3, K = 20: 9.9e-4 —
    # tests/golden/make_cfg5_golden.py 3; it is the fp32 forward / first gradient through eight cells of batch statistics, not the
    # Neumann horizon), so the K = 3 golden is held to the same 5x-spread rule.  But against the reference's ALGORITHM on the same
    # device (oracle restatement of neumann.py:29-66 with autograd's double backward, ~95 s for K = 3) the whole solve agrees to
    # rounding: measured 4.5e-7 under MIOpen's deterministic solver filter (profiles/r06_cfg5_k3_product_vs_reference_algorithm_on_gpu.txt;
    # K = 20 without the filter: 8.7e-5) — asserted at 1e-4, two orders of margin.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    K3 = 3
    assert f"neumann{K3}/fp32" in gold.files, "tests/golden/make_cfg5_golden.py 3 adds the short-horizon golden"
    with _deterministic_convolutions():
        curr3, prev3, vector3 = zoo.cfg5_as_named_case(Config, DEV, K=K3)
        curr3.hypergradient_hvp, curr3.hypergradient_hvp_ack_batchnorm = "forward_over_reverse", True
        t0 = time.perf_counter()
        got3 = np.concatenate([g.ravel() for g in _np(hg.jvp_fn_mapping["neumann"](vector3, curr3, prev3, False))])
        torch.cuda.synchronize()
        t_p3 = time.perf_counter() - t0
        curr3r, prev3r, vector3r = zoo.cfg5_as_named_case(Config, DEV, K=K3)
        t0 = time.perf_counter()
        ref3 = np.concatenate([g.ravel() for g in _np(horc.neumann(vector3r, curr3r, prev3r, False))])
        torch.cuda.synchronize()
        t_r3 = time.perf_counter() - t0
    spread3 = float(gold[f"neumann{K3}/ref_spread"])
    rel3_cpu = float(np.linalg.norm(got3 - gold[f"neumann{K3}/fp32"]) / np.linalg.norm(gold[f"neumann{K3}/fp32"]))
    rel3_gpu = float(np.linalg.norm(got3.astype(np.float64) - ref3) / np.linalg.norm(ref3))
    print(f"cfg5 as named, neumann K={K3}: product vs the reference's algorithm on this GPU {rel3_gpu:.2e} (product {t_p3:.1f} s, reference "
          f"algorithm {t_r3:.1f} s: x{t_r3 / t_p3:.1f}); vs reference-CPU fp32 {rel3_cpu:.2e} (reference fp32 vs fp64 at this K: {spread3:.2e})")
    assert rel3_gpu <= 1e-4, rel3_gpu
    assert rel3_cpu <= max(1e-4, 5.0 * spread3), (rel3_cpu, spread3)


# ------------------------------------------------------------------------------------------------
# BASELINE.json's metric workload at full size, end to end: the product path (analytic MFMA HVP + fused
# recurrence) against the reference's algorithm on the same device tensors (oracle restatement: opaque
# double backward + per-tensor ATen recurrence)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ridge", [0.3, 1e-2], ids=["well-conditioned", "metric"])
@pytest.mark.parametrize("algo,K", [("cg", 20), ("neumann", 10)])
def test_cfg2_metric_workload_end_to_end(algo, K, ridge, be):
    """Three runs of the same algorithm on the same inputs ON THE DEVICE: fp64 (the truth), the reference's algorithm in
    fp32 (oracle restatement on ATen: opaque double backward + per-tensor recurrence), and the product path.
    (The comparison against the reference's CPU run itself lives in tests/test_cfg2_goldens.py.)
      ridge 0.3  — the well-conditioned variant: all three agree to north_star's rtol 1e-4 with two orders to spare;
      ridge 1e-2 — the configuration the metric is quoted on: twenty fp32 CG iterations are not a contraction on it (the
                   reference's CPU run sits 1.85e-2 from its own fp64 run on this seed, ATen's double backward is not even
                   reproducible between calls), so the product is held to 3x the reference's measured distance to the
                   truth (floor 1e-4) instead of to one lucky draw — round 2 asserted 1e-4 here and the kernel choice was
                   being made by that lottery.  Neumann (no division, no chaos) is held to 1e-4 at either ridge."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    import bench

    dev = torch.device(DEV)
    curr64, prev64, vec64 = bench.build(dev, seed=0, dtype=torch.float64, K=K, algo=algo, ridge=ridge)
    truth = [t.detach().clone() for t in getattr(horc, algo)(vec64, curr64, prev64, False)]
    del curr64, prev64, vec64
    curr, prev, vector = bench.build(dev, seed=0, K=K, algo=algo, ridge=ridge)
    assert sum(p.numel() for p in curr.parameters()) == 10_034_826
    want = getattr(horc, algo)(vector, curr, prev, False)
    bench.declare_structure(curr, "hip")
    got = hg.jvp_fn_mapping[algo](vector, curr, prev, False)
    again = hg.jvp_fn_mapping[algo](vector, curr, prev, False)
    for a, b in zip(got, again):
        assert torch.equal(a, b), "the product path must be bit-reproducible"
    e_ref, _ = rel_err(_np(want), _np(truth))
    e_got, _ = rel_err(_np(got), _np(truth))
    rel, _ = rel_err(_np(got), _np(want))
    print(f"cfg2 full size ridge={ridge:g} {algo} K={K}: vs fp64 truth: reference-fp32 {e_ref:.2e}, hip {e_got:.2e}; hip vs reference-fp32 {rel:.2e}")
    if algo == "neumann" or ridge >= 0.1:
        assert e_got <= 1e-4 and rel <= 1e-4, (algo, ridge, e_got, rel, e_ref)
    else:
        # the distance to the TRUTH is held to rtol 1e-4 (measured 8e-7 on this seed); only the distance to the reference's own fp32
        # answer keeps the loose bound — 1.85e-2 is the reference's CPU fp32-vs-fp64 distance on this seed (cfg2_full.npz)
        bound = max(1e-4, 3.0 * max(e_ref, 1.85e-2))
        assert e_got <= 1e-4 and rel <= bound + e_ref, (algo, ridge, e_got, rel, e_ref)


@pytest.mark.parametrize("algo", ["cg", "neumann"])
@pytest.mark.parametrize("keep", [False, True], ids=["solution-free", "x-materialised"])
@pytest.mark.parametrize("dims,B,K", [([784, 512, 256, 128, 10], 100, 5), ([100, 70, 50, 10], 64, 4), ([784, 500, 250, 100, 10], 128, 6),
                                      ([33, 65, 31, 17, 9, 5], 40, 3)], ids=lambda v: str(v))
def test_widths_that_are_not_multiples_of_32_take_the_fused_projected_form(algo, keep, dims, B, K, be):
    """the reference's cg / neumann are shape-agnostic (cg.py:8-70); round 5's fast form asked for every width % 32 == 0
    (a 784-wide input fell to the classic chain).  Round 6: such a network runs on its ZERO-PADDED TWIN (betty_amd/hypergradient/
    _mlp_hip.py: PaddedHipMLPState) — same kernels, the launch counters prove the form: one hoisted N-sized pass, K - 1 (cg) / K
    (neumann) projected iterations and, for nets of >= 4 layers, k_wskpl once per cg iteration (the six-launch form).  Product library.
    Against (a) the same network with the twin switched off (classic chain on the ragged shapes), (b) the un-fused loop, (c) the
    reference's algorithm on this GPU (oracle restatement: opaque double backward), ridge 0.5 so that rtol 1e-4 has resolving power."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as horc

    from betty_amd.hypergradient import _mlp_hip

    assert not _native.is_ab()
    lib = _native.load()
    ridge, seed = 0.5, sum(dims) + B
    assert _mlp_hip.PAD_WIDTHS_TO_32 and _mlp_hip.padded_dims(dims) != tuple(dims)
    h0, p0, l0 = lib.bhg_mlp_hoist_launches(), lib.bhg_mlp_proj_iterations(), lib.bhg_mlp_lin_launches()
    got, st = _run_solver(algo, dims, B, ridge, K, seed, True, keep=keep)
    dh, dp, dl = lib.bhg_mlp_hoist_launches() - h0, lib.bhg_mlp_proj_iterations() - p0, lib.bhg_mlp_lin_launches() - l0
    # what the plan says for the TWIN's shapes is what ran (tests/test_plan_selection.py pins the shape -> form map on the CPU box); the
    # three wide nets take the projected forms, the 33-65-31-17-9-5 net — 64-96-32-32-32 as a twin — is declined by the cost model
    plan = _native.plan_describe(_mlp_hip.padded_dims(dims), B, algo, keep)
    if algo == "cg":
        want = (0 if not plan["hoist"] else (1 if plan["proj_level"] >= 1 else K), {0: 0, 1: K - 1, 2: K - 1}[plan["proj_level"]], K if plan["lin"] else 0)
    else:
        want = (0 if not plan["hoist"] else (1 if plan["proj_level"] >= 1 else K), K if plan["proj_level"] >= 1 else 0, 0)
    assert (dh, dp, dl) == want, (dims, plan["form"], dh, dp, dl, want)
    if max(dims) >= 100 and not keep:
        assert plan["proj_level"] >= 1 and plan["lin"] == (1 if (algo == "cg" and len(dims) - 1 >= 4) else 0), plan
    again, _ = _run_solver(algo, dims, B, ridge, K, seed, True, keep=keep)
    assert all(np.array_equal(u, v) for u, v in zip(again, got)), "bit-reproducible"
    unf, st_u = _run_solver(algo, dims, B, ridge, K, seed, False)
    _mlp_hip.PAD_WIDTHS_TO_32 = False
    try:
        h1 = lib.bhg_mlp_hoist_launches()
        plain, st_p = _run_solver(algo, dims, B, ridge, K, seed, True, keep=keep)
        assert lib.bhg_mlp_hoist_launches() == h1, "without the twin these shapes keep the classic chain"
    finally:
        _mlp_hip.PAD_WIDTHS_TO_32 = True
    curr, prev, direction, _ = _mlp_problem(dims, B, ridge=ridge, seed=seed)
    curr.config = Config(type="cg", cg_iterations=K, cg_alpha=1.0) if algo == "cg" else Config(type="neumann", neumann_iterations=K, neumann_alpha=0.05)
    want = _np(getattr(horc, algo)([0.1 * d for d in direction], curr, prev, False))
    e_unf, e_plain, e_ref = rel_err(got, unf)[0], rel_err(got, plain)[0], rel_err(got, want)[0]
    print(f"{algo} {dims} B={B} K={K} keep={keep}: padded twin: hoist {dh} proj {dp} lin {dl}; vs un-fused {e_unf:.2e}, vs classic chain on the "
          f"ragged shapes {e_plain:.2e}, vs the reference's algorithm on this GPU {e_ref:.2e}")
    assert e_unf <= 5e-5 and e_plain <= 5e-5 and e_ref <= 1e-4, (e_unf, e_plain, e_ref)
    if keep:   # the materialised solution (cg: x; neumann: the accumulator), un-padded into the caller's flat vector
        a, b = st[0].astype(np.float64), st_p[0].astype(np.float64)
        assert np.linalg.norm(a - b) <= 5e-5 * np.linalg.norm(b)


def test_structure_guard_on_the_hip_path(be):
    """The declared WeightedCEMLP is checked against the problem's real training_step on first use (one double backward): a loss
    with label smoothing is rejected, the matching loss passes and the verdict is cached on the problem."""
    from betty_amd.hypergradient.structured import StructureMismatchError, WeightedCEMLP

    dims, B, ridge = [256, 384, 128, 10], 100, 0.05
    for smoothing in (0.0, 0.1):
        curr, prev, direction, _ = _mlp_problem(dims, B, ridge=ridge, seed=5)

        def loss_fn(self, batch, smoothing=smoothing, prev=prev):
            x, y = batch
            logits = self.fwd(x)
            ce = F.cross_entropy(logits, y, reduction="none", label_smoothing=smoothing)
            w = prev.fwd(F.cross_entropy(logits, y, reduction="none").detach().reshape(-1, 1)).reshape(-1)
            return torch.mean(w * ce) + ridge * sum((p * p).sum() for p in self.module.parameters())

        curr._loss_fn = loss_fn
        curr.config = Config(type="cg", cg_iterations=3, cg_alpha=1.0)
        curr.hypergradient_structure = lambda prev_, curr=curr: WeightedCEMLP(
            curr, prev_, layers=list(curr.module.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="hip")
        vec = [0.1 * d for d in direction]
        if smoothing:
            with pytest.raises(StructureMismatchError):
                hg.cg(vec, curr, prev, False)
        else:
            out = hg.cg(vec, curr, prev, False)
            assert all(bool(torch.isfinite(t).all()) for t in out) and len(curr._bhg_structure_verified) == 1


@pytest.mark.parametrize("algo", ["cg", "neumann"])
def test_projection_is_gated_by_batch_size(algo):
    """the projected solvers' Gram work grows with B^2 — with a batch wider than the narrowest hidden layer
    (here B = 1024 against 128) the plan keeps the hoisted chain on the N-sized residual (no projected iteration, no Gram region
    in the workspace) and the result still matches the un-fused loop."""
    lib = _native.load()
    dims, B, K = [256, 384, 128, 10], 1024, 3
    h0, p0 = lib.bhg_mlp_hoist_launches(), lib.bhg_mlp_proj_iterations()
    got, _ = _run_solver(algo, dims, B, 0.05, K, 77, True, keep=False)
    dh, dp = lib.bhg_mlp_hoist_launches() - h0, lib.bhg_mlp_proj_iterations() - p0
    assert dp == 0, (dh, dp)
    want, _ = _run_solver(algo, dims, B, 0.05, K, 77, False)
    rel, _ = rel_err(got, want)
    print(f"{algo} {dims} B={B}: projection gated off (hoist launches {dh}), vs un-fused {rel:.2e}")
    assert rel <= 5e-5, rel


def test_example_learning_to_reweight_runs_with_both_structures_declared():
    """examples/learning_to_reweight_mlp.py end to end on the GPU (own Engine shim, Adam on the meta-weight-net, SGD on the classifier):
    the inner MLP and the meta-weight-net are both declared, so every hypergradient of the run goes through the fused solver AND the
    closed-form upper kernels — with the first-use checks of both declarations against autograd on the way."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for algo in ("cg", "neumann"):
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "learning_to_reweight_mlp.py"), "--algo", algo, "--k", "4",
                            "--iters", "30", "--sizes", "256,128,64,10"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(f"algo={algo}")]
        assert line and "upper steps" in line[0], r.stdout[-500:]
        print(line[0])


def test_example_implicit_maml_convnet_runs_with_declared_batchnorm():
    """examples/implicit_maml_convnet.py end to end on the GPU (own Engine shim): the learner's batch-norm layers are declared
    (betty_amd.nn.fuse_batchnorm_), so every Hessian-vector product of every hypergradient runs bhg_bn_backward_vjp once per layer — the
    printed counter proves it — and the undeclared arm of the same script reaches the same query accuracy."""
    import re
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for arm in ([], ["--no-fuse"]):
        r = subprocess.run([sys.executable, os.path.join(root, "examples", "implicit_maml_convnet.py"), "--k", "3", "--iters", "12", "--size", "16"] + arm,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("implicit MAML")]
        assert line and "finite True" in line[0], r.stdout[-500:]
        print(line[0])
        out[bool(arm)] = (int(re.search(r"fused double-backward calls (\d+)", line[0]).group(1)), int(re.search(r"(\d+) upper steps", line[0]).group(1)),
                          float(re.search(r"query acc ([0-9.]+)", line[0]).group(1)))
    calls, steps, acc = out[False]
    assert steps == out[True][1] > 0 and calls == steps * 3 * 4, (calls, steps)     # K = 3 products x 4 declared layers per hypergradient
    assert out[True][0] == 0 and abs(acc - out[True][2]) <= 0.21, out
