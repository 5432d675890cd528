"""GPU tests of the global-batch one-pass CG solver (bhg_mlp_cg_global_phase through betty_amd/global_hvp.py).

gpurun boxes have ONE GPU, so the multi-rank protocol is exercised two ways:
  * world size 1 through ``cg_global`` itself (a one-rank process group): every phase kernel runs, no collective;
  * TWO ranks emulated in one process on one GPU — two copies of the inner network, each with half of the batch, their
    own flat state / workspaces, the two all-reduces done by hand between the phase calls — against the ONE-rank solver on
    the concatenated batch (itself held to the reference by tests/test_gpu_parity.py and tests/test_cfg2_goldens.py).
The real collectives (gloo, world size 2 and 4) run the same protocol on CPU in tests/test_distributed_cpu.py.
"""
import copy
import os
import socket

import numpy as np
import pytest
import torch

import zoo
from conftest import rel_err

from betty_amd import Config, _native
from betty_amd import hypergradient as hg
from betty_amd.backend import get_backend
from betty_amd.flat import FlatLayout
from betty_amd.hypergradient.structured import WeightedCEMLP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture(scope="module")
def one_rank_group():
    import torch.distributed as dist

    if dist.is_initialized():
        yield None
        return
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        yield None
    finally:
        dist.destroy_process_group()


def _problem(dims, B, ridge, seed, K, keep):
    """Inner MLP + meta-weight-net on the GPU with a batch of B samples; the structure hook is attached."""
    g = torch.Generator().manual_seed(seed)
    inner, upper = zoo.MLP(dims), zoo.MWN(16)
    with torch.no_grad():
        for p in list(inner.parameters()) + list(upper.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(p.shape[-1], 4) ** 0.5))
    inner, upper = inner.to(DEV), upper.to(DEV)
    x = torch.randn(B, dims[0], generator=g).to(DEV)
    y = torch.randint(0, dims[-1], (B,), generator=g).to(DEV)
    vec = [0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    return inner, prev, x, y, vec


def _attach(inner, prev, x, y, ridge, K, keep, alpha=1.0):
    curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=K, cg_alpha=alpha),
                           loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x, y))
    curr.hypergradient_structure = lambda prev_: WeightedCEMLP(
        curr, prev_, layers=list(inner.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="hip",
        fused=True, keep_solution=keep)
    return curr


@pytest.mark.parametrize("keep", [True, False], ids=["x-materialised", "solution-free"])
@pytest.mark.parametrize("dims,B,K", [([256, 384, 128, 10], 100, 5), ([70, 130, 36, 10], 100, 4), ([64, 10], 50, 3),
                                      ([128, 96, 64, 32, 10], 128, 1)], ids=lambda v: str(v))
def test_world_size_one_is_the_one_rank_solver(dims, B, K, keep, one_rank_group, global_form):
    """cg_global at world size 1: the phase-cut iteration (chain | step length + outputs | dots of the residual) gives what
    bhg_mlp_cg_solve gives — the dot products of the residual are taken per chunk instead of per output tile, so equality is to
    fp32 summation noise, not bitwise."""
    from betty_amd.global_hvp import ONE_PASS_STATS

    global_form("one_pass")
    inner, prev, x, y, vec = _problem(dims, B, 0.05, sum(dims) + B + K, K, keep)
    want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vec, _attach(inner, prev, x, y, 0.05, K, keep), prev, False)]
    n0 = ONE_PASS_STATS["solves"]
    got = [t.clone() for t in hg.jvp_fn_mapping["cg_global"](vec, _attach(inner, prev, x, y, 0.05, K, keep), prev, False)]
    assert ONE_PASS_STATS["solves"] == n0 + 1, "the one-pass form must be the one that ran"
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 5e-5, rel   # (the one-rank solver projects its direction products when it can; this one never does)


def _emulate(parts, prev, vecs, K, alpha, keep):
    """Drive the protocol of betty_amd/global_hvp.py::_cg_global_one_pass for len(parts) ranks living in this process."""
    be = get_backend()
    G = len(parts)
    lays, provs, st = [], [], []
    for curr, vec in zip(parts, vecs):
        lay = FlatLayout([t.numel() for t in vec], vec[0].device)     # NOT the cached layout: every "rank" owns its state
        lays.append(lay)
        prov = curr.hypergradient_structure(prev)
        prov.pad_widths = False   # as cg_global does: the protocol exchanges the real network's state
        prov.prepare()
        assert prov.fused_cg_global_ready(lay, K)
        provs.append(prov)
        st.append(dict(v=lay.new_flat(), php=torch.zeros(1, dtype=torch.float64, device=DEV), xrp=lay.state(3)))
    skip_x = [bool(p.fused_cg_global_skips_solution(l, K)) for p, l in zip(provs, lays)]
    assert all(s == (not keep) for s in skip_x)
    # right-hand side: mean of the local vectors ("all-reduce" by hand)
    for lay, vec, s in zip(lays, vecs, st):
        be.flatten(lay, vec, s["v"], 1.0 / G)
    total = sum(s["v"] for s in st)
    for lay, vec, s in zip(lays, vecs, st):
        s["v"].copy_(total)
        x, r, p = s["xrp"]
        if not keep:
            x.fill_(float("nan"))
        be.cg_init(lay, lay.views(s["v"], vec), x if keep else None, r, p)
    for k in range(K):
        for prov, lay, s in zip(provs, lays, st):
            prov.cg_global_phase(lay, *s["xrp"], k, K, _native.BHG_CG_GLOBAL_CHAIN, G, s["php"], alpha)
        tot = sum(s["php"] for s in st)
        for s in st:
            s["php"].copy_(tot)
        for prov, lay, s in zip(provs, lays, st):
            prov.cg_global_phase(lay, *s["xrp"], k, K, _native.BHG_CG_GLOBAL_UPDATE, G, s["php"], alpha)
        if k + 1 < K:
            mean_r = sum(s["xrp"][1] for s in st) * (1.0 / G)
            for s in st:
                s["xrp"][1].copy_(mean_r)
            for prov, lay, s in zip(provs, lays, st):
                prov.cg_global_phase(lay, *s["xrp"], k, K, _native.BHG_CG_GLOBAL_DOTS, G, s["php"], alpha)
    outs = []
    for prov, lay, vec, s in zip(provs, lays, vecs, st):
        token = prov.cg_global_finish(lay, K, alpha)
        outs.append([t.clone() for t in prov.mixed_vjp(lay.views(s["xrp"][0], vec), False, solve=token)])
    xs = [s["xrp"][0].clone() for s in st]
    if not keep:
        for xx in xs:
            assert torch.isnan(xx).all(), "the solution vector must not be touched"
    hyper = [sum(o[i] for o in outs) / G for i in range(len(outs[0]))]
    return hyper, xs


@pytest.mark.parametrize("hoist", [None, "0"], ids=["hoisted-default", "classic-chain"])
@pytest.mark.parametrize("keep", [True, False], ids=["x-materialised", "solution-free"])
@pytest.mark.parametrize("dims,B,K,alpha", [([256, 384, 128, 10], 100, 6, 1.0), ([70, 130, 36, 10], 64, 4, 0.5),
                                            ([512, 256, 128, 64, 10], 128, 8, 1.0)], ids=lambda v: str(v))
def test_two_emulated_ranks_match_the_one_rank_solver_on_the_concatenated_batch(dims, B, K, alpha, keep, hoist, bhg_debug):
    """mean_g (r - alpha H_g p) = r - alpha H p: two half batches, two states, the 8-byte and the N-sized exchange done by hand,
    against bhg_mlp_cg_solve on the whole batch (2B rows: other tile counts, other summation trees — fp32 noise, tolerance
    north_star's 1e-4 with ridge 0.05 keeping the K-step recurrence well inside it)."""
    if hoist is None:
        bhg_debug.delenv("BHG_MLP_HOIST", raising=False)
    else:
        bhg_debug.setenv("BHG_MLP_HOIST", hoist)
    ridge = 0.05
    inner, prev, x, y, _ = _problem(dims, 2 * B, ridge, 7 * sum(dims) + B + K, K, keep)
    g = torch.Generator().manual_seed(99)
    vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(2)]
    vmean = [0.5 * (a + b) for a, b in zip(*vecs)]
    # the one-rank solver on the concatenated batch, right-hand side = mean of the ranks' vectors
    want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vmean, _attach(inner, prev, x, y, ridge, K, True, alpha), prev, False)]
    want_x = get_backend().layout(vmean).state(3)[0].clone()
    # two ranks: the same weights (separate module objects => separate activation buffers), half of the batch each
    inner2 = copy.deepcopy(inner)
    parts = [_attach(inner, prev, x[:B], y[:B], ridge, K, keep, alpha), _attach(inner2, prev, x[B:], y[B:], ridge, K, keep, alpha)]
    lib = _native.load()
    h0 = lib.bhg_mlp_hoist_launches()
    got, xs = _emulate(parts, prev, vecs, K, alpha, keep)
    hoisted = lib.bhg_mlp_hoist_launches() > h0
    assert hoisted == (hoist is None and all(d % 32 == 0 for d in dims[:-1]) and len(dims) >= 4), (hoisted, dims)
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 1e-4, rel
    if keep:
        assert torch.equal(xs[0], xs[1]), "replicated state: both ranks must hold the same bits"
        a, b = xs[0].double().cpu().numpy(), want_x.double().cpu().numpy()
        assert np.linalg.norm(a - b) <= 1e-4 * np.linalg.norm(b)


# ---- round 6: the FACTOR-EXCHANGE form (csrc/mlp/fx.inc) ----------------------------------------------------------------------------
@pytest.fixture
def global_form():
    import betty_amd.global_hvp as gh

    old = gh.GLOBAL_FORM

    def set_(v):
        gh.GLOBAL_FORM = v

    yield set_
    gh.GLOBAL_FORM = old


FX_SHAPES = [([256, 384, 128, 10], 100, 5, 1.0), ([128, 96, 64, 32, 10], 128, 3, 0.5), ([512, 256, 128, 64, 10], 77, 8, 1.0),
             ([256, 384, 128, 100], 100, 4, 1.0), ([512, 384, 256, 256, 128, 10], 60, 4, 1.0)]


@pytest.mark.parametrize("dims,B,K,alpha", FX_SHAPES, ids=lambda v: str(v))
def test_factor_exchange_at_world_size_one_is_the_one_rank_solver(dims, B, K, alpha, one_rank_group, global_form):
    """cg_global at world size 1 takes the factor-exchange form (solution-free callers): the fully projected solver with its Gram and
    G(raw) products as launches of their own and its inner products / scalars in kernels of their own — against bhg_mlp_cg_solve."""
    from betty_amd.global_hvp import FX_STATS, ONE_PASS_STATS

    inner, prev, x, y, vec = _problem(dims, B, 0.05, sum(dims) + B + K, K, False)
    want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vec, _attach(inner, prev, x, y, 0.05, K, False, alpha), prev, False)]
    n0, o0 = FX_STATS["solves"], ONE_PASS_STATS["solves"]
    lib = _native.load()
    p0 = lib.bhg_mlp_proj_iterations()
    got = [t.clone() for t in hg.jvp_fn_mapping["cg_global"](vec, _attach(inner, prev, x, y, 0.05, K, False, alpha), prev, False)]
    assert FX_STATS["solves"] == n0 + 1 and ONE_PASS_STATS["solves"] == o0, "the factor-exchange form must be the one that ran"
    assert lib.bhg_mlp_proj_iterations() == p0 + K, "K recurrence steps (K - 1 between iterations + the closing one)"
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 2e-5, rel
    # the caller who wants x keeps the one-pass form
    got_x = hg.jvp_fn_mapping["cg_global"](vec, _attach(inner, prev, x, y, 0.05, K, True, alpha), prev, False)
    assert FX_STATS["solves"] == n0 + 1 and ONE_PASS_STATS["solves"] == o0 + 1
    rel, _ = rel_err([t.cpu().numpy() for t in got_x], [t.cpu().numpy() for t in want])
    assert rel <= 5e-5, rel


@pytest.mark.parametrize("K,ridge", [(1, 0.05), (2, 0.0), (3, 0.0)])
def test_factor_exchange_edge_cases_one_iteration_and_no_ridge(K, ridge, one_rank_group):
    """K = 1 (BEGIN, one CHAIN / GRAM pair, END: the closing step is also the first) and a loss without ridge (the shift terms of every
    recurrence vanish): against the one-rank solver."""
    from betty_amd.global_hvp import FX_STATS

    dims, B = [256, 384, 128, 10], 100
    inner, prev, x, y, vec = _problem(dims, B, ridge, 4711 + K, K, False)
    want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vec, _attach(inner, prev, x, y, ridge, K, False), prev, False)]
    n0 = FX_STATS["solves"]
    got = [t.clone() for t in hg.jvp_fn_mapping["cg_global"](vec, _attach(inner, prev, x, y, ridge, K, False), prev, False)]
    assert FX_STATS["solves"] == n0 + 1
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 2e-5, rel


def _emulate_fx(parts, prev, vecs, K, alpha):
    """Drive betty_amd/global_hvp.py::_cg_global_factor_exchange for len(parts) ranks living in this process: every all-gather is a
    copy of rank r's row into the other ranks' buffers."""
    be = get_backend()
    G = len(parts)
    lays, provs, rhss, bufs = [], [], [], []
    for curr, vec in zip(parts, vecs):
        lay = FlatLayout([t.numel() for t in vec], vec[0].device)
        prov = curr.hypergradient_structure(prev)
        prov.pad_widths = False
        prov.prepare()
        assert prov.fused_cg_fx_ready(lay, K, G)
        lays.append(lay)
        provs.append(prov)
        bufs.append(prov._state.fx_buffers(G))
    assert len({id(b["slab"]) for b in bufs}) == G, "every emulated rank needs buffers of its own"
    flats = []
    for lay, vec in zip(lays, vecs):
        v = lay.new_flat()
        be.flatten(lay, vec, v, 1.0 / G)
        flats.append(v)
    total = sum(flats)
    for lay, vec, v in zip(lays, vecs, flats):
        v.copy_(total)
        rhss.append(lay.views(v, vec))

    def gather(name):
        for r in range(G):
            for j in range(G):
                if j != r:
                    bufs[j][name][r].copy_(bufs[r][name][r])

    for g, (prov, rhs) in enumerate(zip(provs, rhss)):
        prov.cg_fx_phase(rhs, 0, K, _native.BHG_CG_FX_BEGIN, G, g, alpha)
    gather("const")
    for k in range(K):
        for g, (prov, rhs) in enumerate(zip(provs, rhss)):
            prov.cg_fx_phase(rhs, k, K, _native.BHG_CG_FX_CHAIN, G, g, alpha)
        gather("slab")
        for g, (prov, rhs) in enumerate(zip(provs, rhss)):
            prov.cg_fx_phase(rhs, k, K, _native.BHG_CG_FX_GRAM, G, g, alpha)
        gather("scal")
    outs = []
    for g, (prov, lay, rhs) in enumerate(zip(provs, lays, rhss)):
        prov.cg_fx_phase(rhs, K - 1, K, _native.BHG_CG_FX_END, G, g, alpha)
        token = prov.cg_fx_finish(lay, K, alpha)
        outs.append([t.clone() for t in prov.mixed_vjp(None, False, solve=token)])
    for o in outs[1:]:   # the narrow state and every scalar are replicated: NaN or Inf anywhere would show on every rank alike
        assert all(bool(torch.isfinite(t).all()) for t in o)
    return [sum(o[i] for o in outs) / G for i in range(len(outs[0]))], provs


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("dims,B,K,alpha", [([256, 384, 128, 10], 100, 6, 1.0), ([128, 96, 64, 32, 10], 64, 4, 0.5),
                                            ([512, 256, 128, 64, 10], 128, 8, 1.0)], ids=lambda v: str(v))
def test_emulated_ranks_exchanging_factors_match_the_one_rank_solver_on_the_concatenated_batch(dims, B, K, alpha, world):
    """`world` shares of one batch, `world` states; the three all-gathers done by hand; against bhg_mlp_cg_solve on the whole batch
    (world * B rows) with the mean of the ranks' right-hand sides.  Rectangular Gram blocks [Bp x world * Bp] for world > 1."""
    import warnings

    ridge = 0.05
    for attempt in range(4):   # an instance with a hidden pre-activation within 1e-6 of zero is not a test of the solver: the two sides
        # round their forward passes differently (other tile counts), a ReLU mask flips, the Hessians differ — draw another one
        inner, prev, x, y, _ = _problem(dims, world * B, ridge, 7 * sum(dims) + B + K + 1000 * attempt, K, False)
        g = torch.Generator().manual_seed(99)
        vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(world)]
        vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vmean, _attach(inner, prev, x, y, ridge, K, True, alpha), prev, False)]
            inners = [inner] + [copy.deepcopy(inner) for _ in range(world - 1)]
            parts = [_attach(inners[r], prev, x[r * B:(r + 1) * B], y[r * B:(r + 1) * B], ridge, K, False, alpha) for r in range(world)]
            got, _ = _emulate_fx(parts, prev, vecs, K, alpha)
        if not any("ReLU-kink" in str(w.message) for w in caught):
            break
    else:
        pytest.fail("four instances in a row sat on a ReLU kink")
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 1e-4, rel


# ---- the Neumann series in the factor-exchange form (Config(type="neumann_global"); bhg_mlp_neumann_fx_phase) ---------------------------
def _attach_neumann(inner, prev, x, y, ridge, K, alpha):
    curr = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=K, neumann_alpha=alpha),
                           loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x, y))
    curr.hypergradient_structure = lambda prev_: WeightedCEMLP(
        curr, prev_, layers=list(inner.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="hip", fused=True)
    return curr


def _emulate_neumann_fx(parts, prev, vecs, K, alpha):
    """betty_amd/global_hvp.py::neumann_global for len(parts) ranks living in this process: ONE gather per iteration (the factor slab)."""
    be = get_backend()
    G = len(parts)
    provs, lays, rhss, bufs, flats = [], [], [], [], []
    for curr, vec in zip(parts, vecs):
        lay = FlatLayout([t.numel() for t in vec], vec[0].device)
        prov = curr.hypergradient_structure(prev)
        prov.pad_widths = False
        prov.prepare()
        assert prov.fused_neumann_fx_ready(lay, K, G)
        v = lay.new_flat()
        be.flatten(lay, vec, v, 1.0 / G)
        provs.append(prov); lays.append(lay); flats.append(v); bufs.append(prov._state.fx_buffers(G))
    total = sum(flats)
    for lay, vec, v in zip(lays, vecs, flats):
        v.copy_(total)
        rhss.append(lay.views(v, vec))

    def gather(name):
        for r in range(G):
            for j in range(G):
                if j != r:
                    bufs[j][name][r].copy_(bufs[r][name][r])

    def phase(k, ph):
        for g, (prov, rhs) in enumerate(zip(provs, rhss)):
            prov.neumann_fx_phase(rhs, k, K, ph, G, g, alpha)

    phase(0, _native.BHG_CG_FX_BEGIN); gather("const")
    for k in range(K):
        phase(k, _native.BHG_CG_FX_CHAIN); gather("slab")
        phase(k, _native.BHG_CG_FX_GRAM)
    phase(K, _native.BHG_CG_FX_CHAIN)
    phase(K, _native.BHG_CG_FX_END)
    outs = [[t.clone() for t in prov.mixed_vjp(None, False, solve=prov.neumann_fx_finish(lay, K, alpha))] for prov, lay in zip(provs, lays)]
    return [sum(o[i] for o in outs) / G for i in range(len(outs[0]))]


@pytest.mark.parametrize("dims,B,K,alpha,world", [([256, 384, 128, 10], 100, 5, 0.1, 1), ([256, 384, 128, 10], 100, 10, 0.1, 2),
                                                  ([512, 256, 128, 64, 10], 77, 6, 0.05, 3), ([512, 384, 256, 256, 128, 10], 60, 4, 0.1, 2),
                                                  ([256, 384, 128, 100], 130, 3, 0.1, 2), ([256, 384, 128, 10], 64, 1, 0.2, 4)], ids=lambda v: str(v))
def test_neumann_in_the_factor_exchange_form_matches_the_one_rank_solver(dims, B, K, alpha, world, one_rank_group):
    """neumann.py:59-66 on the global batch: emulated ranks exchanging NOTHING but the factor slab (one gather per iteration, no scalars)
    against the one-rank Neumann solver on the concatenated batch; at world size 1 also through Config(type="neumann_global") itself."""
    import warnings

    ridge = 0.05
    for attempt in range(4):
        inner, prev, x, y, _ = _problem(dims, world * B, ridge, 13 * sum(dims) + B + K + 1000 * attempt, K, False)
        g = torch.Generator().manual_seed(5)
        vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(world)]
        vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            want = [t.clone() for t in hg.jvp_fn_mapping["neumann"](vmean, _attach_neumann(inner, prev, x, y, ridge, K, alpha), prev, False)]
            inners = [inner] + [copy.deepcopy(inner) for _ in range(world - 1)]
            parts = [_attach_neumann(inners[r], prev, x[r * B:(r + 1) * B], y[r * B:(r + 1) * B], ridge, K, alpha) for r in range(world)]
            got = _emulate_neumann_fx(parts, prev, vecs, K, alpha)
        if not any("ReLU-kink" in str(w.message) for w in caught):
            break
    else:
        pytest.fail("four instances in a row sat on a ReLU kink")
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 1e-4, rel
    if world == 1:
        from betty_amd.global_hvp import FX_STATS

        n0 = FX_STATS.get("neumann_solves", 0)
        c = _attach_neumann(inner, prev, x, y, ridge, K, alpha)
        c.config.type = "neumann_global"
        got2 = [t.clone() for t in hg.jvp_fn_mapping["neumann_global"](vmean, c, prev, False)]
        assert FX_STATS.get("neumann_solves", 0) == n0 + 1
        rel, _ = rel_err([t.cpu().numpy() for t in got2], [t.cpu().numpy() for t in want])
        assert rel <= 1e-4, rel


SWEEP = [  # dims, per-rank batch, K, cg_alpha, world — odd world sizes (1 / G is not a power of two), batches beyond one row tile of 128
    # (Bp = 256, 384), ragged batches, three to five layers, heads of 3 .. 100 classes
    ([256, 384, 128, 10], 37, 5, 1.0, 3), ([256, 384, 128, 3], 130, 4, 1.0, 2), ([512, 256, 128, 64, 10], 200, 5, 0.5, 2),
    ([384, 256, 256, 64, 100], 64, 4, 1.0, 3), ([512, 384, 256, 256, 128, 10], 60, 4, 1.0, 2), ([512, 512, 256, 256, 128, 10], 300, 3, 1.0, 1),
    ([1024, 512, 256, 16], 96, 6, 0.25, 5), ([256, 384, 128, 10], 128, 7, 1.0, 6),
]


@pytest.mark.parametrize("dims,B,K,alpha,world", SWEEP, ids=lambda v: str(v))
def test_factor_exchange_sweep_of_shapes_batches_and_world_sizes(dims, B, K, alpha, world):
    import warnings

    ridge = 0.05
    lib = _native.load()
    for attempt in range(4):
        inner, prev, x, y, _ = _problem(dims, world * B, ridge, 97 * sum(dims) + B + K + 1000 * attempt, K, False)
        g = torch.Generator().manual_seed(5)
        vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(world)]
        vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vmean, _attach(inner, prev, x, y, ridge, K, True, alpha), prev, False)]
            inners = [inner] + [copy.deepcopy(inner) for _ in range(world - 1)]
            parts = [_attach(inners[r], prev, x[r * B:(r + 1) * B], y[r * B:(r + 1) * B], ridge, K, False, alpha) for r in range(world)]
            p0 = lib.bhg_mlp_proj_iterations()
            got, _ = _emulate_fx(parts, prev, vecs, K, alpha)
            assert lib.bhg_mlp_proj_iterations() - p0 == world * K
        if not any("ReLU-kink" in str(w.message) for w in caught):
            break
    else:
        pytest.fail("four instances in a row sat on a ReLU kink")
    rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
    assert rel <= 1e-4, rel


@pytest.mark.parametrize("ks", [2, 4])
def test_k_split_gram_slabs_are_consumed_as_a_concatenated_reduction(ks, bhg_debug):
    """Key fx_ksplit: T_l / E_l leave ks K-split slabs, never summed — the G(raw) product runs over [S | T_0 | .. | T_{ks-1}]
    against [Rd ; delta ; .. ; delta] (csrc/mlp/fx.inc: fx_plan; the product takes ks = 2 while the Gram launch would leave three quarters of
    the chip idle — world size 1 at these shapes — and 1 otherwise).  Two emulated ranks (default there: 1) against the one-rank solver; and
    against the default arm to summation noise."""
    dims, B, K, alpha, world, ridge = [256, 384, 128, 10], 100, 6, 1.0, 2, 0.05
    inner, prev, x, y, _ = _problem(dims, world * B, ridge, 31337, K, False)
    g = torch.Generator().manual_seed(99)
    vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(world)]
    vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
    want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vmean, _attach(inner, prev, x, y, ridge, K, True, alpha), prev, False)]
    outs = {}
    for arm in (None, ks):
        if arm is None:
            bhg_debug.delenv("fx_ksplit")
        else:
            bhg_debug.setenv("fx_ksplit", str(arm))
        inners = [copy.deepcopy(inner) for _ in range(world)]   # fresh modules: fresh buffers, sized for this arm
        parts = [_attach(inners[r], prev, x[r * B:(r + 1) * B], y[r * B:(r + 1) * B], ridge, K, False, alpha) for r in range(world)]
        outs[arm], _ = _emulate_fx(parts, prev, vecs, K, alpha)
        rel, _ = rel_err([t.cpu().numpy() for t in outs[arm]], [t.cpu().numpy() for t in want])
        assert rel <= 1e-4, (arm, rel)
    rel, _ = rel_err([t.cpu().numpy() for t in outs[ks]], [t.cpu().numpy() for t in outs[None]])
    assert rel <= 2e-5, rel


# ---- two real processes, real collectives (gloo stages the device buffers through the host), one GPU ------------------------------
def _two_process_worker(rank, world, port, q, form="one_pass"):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import betty_amd.global_hvp as gh
        from betty_amd.global_hvp import FX_STATS, ONE_PASS_STATS, cg_global

        gh.GLOBAL_FORM = form
        torch.cuda.set_device(0)
        dims, B, K, ridge = [256, 384, 128, 10], 100, 6, 0.05
        inner, prev, x, y, _ = _problem(dims, world * B, ridge, 4242, K, False)       # same seed: same weights, same full batch
        g = torch.Generator().manual_seed(99)
        vecs = [[0.1 * torch.randn(p.shape, generator=g).to(DEV) for p in inner.parameters()] for _ in range(world)]
        vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
        want = [t.clone() for t in hg.jvp_fn_mapping["cg"](vmean, _attach(inner, prev, x, y, ridge, K, True), prev, False)]
        mine = _attach(inner, prev, x[rank * B:(rank + 1) * B], y[rank * B:(rank + 1) * B], ridge, K, False)
        got = cg_global(vecs[rank], mine, prev, False)
        rel, _ = rel_err([t.cpu().numpy() for t in got], [t.cpu().numpy() for t in want])
        flat = torch.cat([t.reshape(-1) for t in got]).cpu()
        others = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(others, flat)
        same = all(torch.equal(o, flat) for o in others)
        if form == "auto":
            q.put((rank, rel, same, FX_STATS["solves"], FX_STATS["slab_gathers"], FX_STATS["scal_gathers"], K, FX_STATS["const_gathers"],
                   ONE_PASS_STATS["solves"]))
        else:
            q.put((rank, rel, same, ONE_PASS_STATS["solves"], ONE_PASS_STATS["scalar_all_reduces"], ONE_PASS_STATS["residual_all_reduces"], K))
    finally:
        dist.destroy_process_group()


def test_two_processes_share_one_gpu_over_gloo():
    """The whole of cg_global — right-hand-side exchange, K x (8-byte SUM, N-sized MEAN), final M-sized exchange — between two
    processes (both on cuda:0; gloo moves the device buffers through the host): every rank returns the same bits, equal to the
    one-rank solver on the concatenated batch to north_star's tolerance."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_two_process_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, same, solves, n_scalar, n_resid, K in sorted(q.get(timeout=5) for _ in range(world)):
        assert rel <= 1e-4, (rank, rel)
        assert same
        assert (solves, n_scalar, n_resid) == (1, K, K - 1)


def test_two_processes_exchange_factors_over_gloo_on_one_gpu():
    """cg_global in its factor-exchange form between two processes on cuda:0: the right-hand side's mean (one N-sized all-reduce), one
    gather of the constants, K gathers of the factor slab, K gathers of the fp64 partials, the M-sized exchange — the same bits on both
    ranks, the one-rank solver's answer on the concatenated batch to north_star's tolerance."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_two_process_worker, args=(r, world, port, q, "auto")) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, same, solves, n_slab, n_scal, K, n_const, one_pass in sorted(q.get(timeout=5) for _ in range(world)):
        assert rel <= 1e-4, (rank, rel)
        assert same
        assert (solves, n_slab, n_scal, n_const, one_pass) == (1, K, K, 1, 0)


# ---- round 5: the closed-form upper net under data parallelism (SigmoidMLPWeightNet.average_over) ---------------------------------
def _average_over_worker(rank, world, port, q):
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zoo
        from betty_amd import Config

        torch.cuda.set_device(0)
        case = zoo.CASE_BY_NAME["reweight_cg20"]
        inputs = zoo.seed_family_inputs(case.family, seed=rank)      # the ranks' own batches / directions, ...
        shared = zoo.seed_family_inputs(case.family, seed=0)
        for k in inputs:                                                # ... the same inner and upper weights (replicas)
            if not k.startswith(("batch", "vec")):
                inputs[k] = shared[k]
        res = {}
        for sync in (False, True):
            curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
            zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True, average_over=True)
            for p in prev.trainable_parameters():
                p.grad = torch.zeros_like(p)          # (accumulated into, not replaced: problem.py:592-597)
            out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
            res[sync] = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()]) if sync else \
                torch.cat([t.reshape(-1) for t in out])
        local = res[False].cpu()
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        mean = sum(every) / world
        differ = float((every[0] - every[-1]).norm() / every[0].norm())
        rel = float((res[True].cpu() - mean).norm() / mean.norm())
        q.put((rank, rel, differ))
    finally:
        dist.destroy_process_group()


def _overlap_or_ddp_worker(rank, world, port, q, mode):
    """mode "ddp": average_over left unset, the upper module wrapped in DistributedDataParallel (the closed form must take the
    mean the wrapper's reducer would have taken, never a rank-local gradient).  mode "overlap": average_over=True, overlap=True — the
    all-reduce is deferred (betty_amd/distributed.py), .grad holds the mean after the fence / the optimizer's step."""
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zoo
        from betty_amd import Config
        from betty_amd import distributed as bd

        torch.cuda.set_device(0)
        case = zoo.CASE_BY_NAME["reweight_cg20"]
        inputs = zoo.seed_family_inputs(case.family, seed=rank)
        shared = zoo.seed_family_inputs(case.family, seed=0)
        for k in inputs:
            if not k.startswith(("batch", "vec")):
                inputs[k] = shared[k]
        res, pending, fenced_by_step = {}, None, None
        for sync in (False, True):
            curr, prev, vector = zoo.build_case(case, inputs, Config, device=DEV)
            if mode == "ddp":
                from torch.nn.parallel import DistributedDataParallel as DDP

                prev.fwd = DDP(prev.module, device_ids=[0], gradient_as_bucket_view=True, find_unused_parameters=True)   # as bench.py / problem.py:220-224
                zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True, average_over=None)
            else:
                zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True, average_over=True, overlap=True)
                prev.optimizer = torch.optim.SGD(prev.trainable_parameters(), lr=0.0)
            out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
            if sync and mode == "overlap":
                pending = bd.pending_grad_syncs()
                prev.optimizer.step()                      # lr = 0: reads .grad, changes nothing; its pre-hook is the fence
                fenced_by_step = bd.pending_grad_syncs() == 0
                # a second hop accumulates ONTO the fenced .grad: twice the mean
                hg.jvp_fn_mapping[case.algo](vector, curr, prev, True)
                bd.fence_grads()
                res["twice"] = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()])
                for p in prev.trainable_parameters():
                    p.grad = None
                hg.jvp_fn_mapping[case.algo](vector, curr, prev, True)
                bd.fence_grads()
            res[sync] = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()]) if sync else \
                torch.cat([t.reshape(-1) for t in out])
        local = res[False].cpu()
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        mean = sum(every) / world
        differ = float((every[0] - every[-1]).norm() / every[0].norm())
        rel = float((res[True].cpu() - mean).norm() / mean.norm())
        rel2 = float((res["twice"].cpu() - 2 * mean).norm() / mean.norm()) if "twice" in res else 0.0
        q.put((rank, rel, differ, rel2, pending, fenced_by_step))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ddp", "overlap"])
def test_closed_form_upper_net_under_ddp_wrapper_and_with_the_deferred_all_reduce(mode):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_overlap_or_ddp_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, differ, rel2, pending, fenced_by_step in sorted(q.get(timeout=5) for _ in range(world)):
        assert differ > 1e-3, "the ranks must hold different local results for the test to mean anything"
        assert rel <= 1e-5 and rel2 <= 1e-5, (mode, rank, rel, rel2)
        if mode == "overlap":
            assert pending == 1 and fenced_by_step, (pending, fenced_by_step)


def test_closed_form_upper_net_averages_over_ranks():
    """sync=True with the upper module declared in closed form and average_over=True: what lands in .grad is the MEAN over the ranks of
    their local hypergradients (sync=False does no collective) — the reduction DistributedDataParallel's reducer performs for the
    reference's backward (problem.py:220-224, cg.py:58-63) — between two processes on the one GPU over gloo."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_average_over_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, differ in sorted(q.get(timeout=5) for _ in range(world)):
        assert differ > 1e-3, "the ranks must hold different local results for the test to mean anything"
        assert rel <= 1e-5, (rank, rel)
