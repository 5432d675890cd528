"""Host orchestration of betty_amd.hypergradient (flat state, autograd views, sync semantics,
registry / get_grads) on CPU, with the kernels replaced by the C oracle through the test-only
checker backend.  The GPU suite (test_gpu_parity.py) runs the same cases through the HIP kernels."""
import os
import sys
import numpy as np
import pytest
import torch

import zoo
from _cpu_checker_backend import CpuCheckerBackend
from conftest import golden_list, load_golden, rel_err

import betty_amd
from betty_amd import Config
from betty_amd import hypergradient as hg
from betty_amd.backend import use_backend


@pytest.fixture()
def checker():
    with use_backend(CpuCheckerBackend()) as b:
        yield b


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_host_path_matches_reference(case, checker):
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    w_before = [p.data.clone() for p in curr.trainable_parameters()]
    v_before = [v.clone() for v in vector]
    out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
    want = golden_list(outputs, case.name, "fp32")
    assert len(out) == len(want)
    rel, mx = rel_err([o.detach().numpy() for o in out], want)
    assert rel <= case.rtol and mx <= 10 * case.rtol, (rel, mx)
    for v, vb in zip(vector, v_before):  # the direction vector is never mutated (cg.py:35 copies)
        assert torch.equal(v, vb)
    if case.algo in ("darts", "sama"):  # weights restored up to the reference's own drift
        for p, w in zip(curr.trainable_parameters(), golden_list(outputs, case.name, "w32")):
            np.testing.assert_allclose(p.data.numpy(), w, rtol=0, atol=1e-7)
    else:
        for p, w in zip(curr.trainable_parameters(), w_before):
            assert torch.equal(p.data, w)


@pytest.mark.parametrize("case", zoo.CASES, ids=lambda c: c.name)
def test_sync_accumulates_and_returns_none(case, checker):
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    # pre-existing gradient must be accumulated into, not overwritten (problem.py:592-597)
    for p in prev.trainable_parameters():
        p.grad = torch.full_like(p, 0.25)
    ret = hg.jvp_fn_mapping[case.algo](vector, curr, prev, True)
    assert ret is None
    got = [p.grad.detach().numpy() - 0.25 for p in prev.trainable_parameters()]
    want = golden_list(outputs, case.name, "sync32")
    rel, _ = rel_err(got, want)
    scale = max(1.0, 0.25 / max(np.abs(np.concatenate([w.ravel() for w in want])).max(), 1e-30))
    assert rel <= case.rtol * scale + 1e-6 * scale


def test_registry_and_get_grads(checker):
    # the reference's keys (betty/hypergradient/__init__.py:13-19); `reinforce` is a stub there and fails loudly here
    assert set(hg.jvp_fn_mapping) == {"cg", "neumann", "darts", "sama", "reinforce", "cg_global", "neumann_global"}   # *_global: extensions
    with pytest.raises(NotImplementedError, match="reinforce"):
        hg.jvp_fn_mapping["reinforce"]([], None, None, False)
    case = zoo.CASE_BY_NAME["logreg_cg5"]
    inputs, _ = load_golden("logreg")
    curr, prev, _ = zoo.build_case(case, inputs, Config)
    # upper loss = validation BCE through the inner weights; path = [upper, inner, upper]
    xv = torch.from_numpy(inputs["batch_x"][:64])
    yv = torch.from_numpy(inputs["batch_y"][:64])
    loss = torch.nn.functional.binary_cross_entropy_with_logits(xv @ curr.module.w, yv)
    out = hg.get_grads(loss, [prev, curr, prev], retain_graph=False, do_sync=False)
    assert len(out) == 1 and out[0].shape == (100,)
    # against the oracle's get_grads on an identical fresh problem
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as orc

    curr2, prev2, _ = zoo.build_case(case, inputs, Config)
    loss2 = torch.nn.functional.binary_cross_entropy_with_logits(xv @ curr2.module.w, yv)
    want = orc.get_grads(loss2, [prev2, curr2, prev2], False, False)
    rel, _ = rel_err([out[0].detach().numpy()], [want[0].detach().numpy()])
    assert rel < 1e-5
    # do_sync=True: result lands in .grad, returns None
    curr3, prev3, _ = zoo.build_case(case, inputs, Config)
    loss3 = torch.nn.functional.binary_cross_entropy_with_logits(xv @ curr3.module.w, yv)
    assert hg.get_grads(loss3, [prev3, curr3, prev3], False, True) is None
    rel, _ = rel_err([prev3.module.w.grad.numpy()], [want[0].detach().numpy()])
    assert rel < 1e-5


def test_higher_order_paths_rejected(checker):
    case = zoo.CASE_BY_NAME["logreg_cg5"]
    inputs, _ = load_golden("logreg")
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    curr.paths = [[curr, prev, curr]]
    with pytest.raises(AssertionError, match="higher order"):
        hg.cg(vector, curr, prev, False)
    curr.config = Config(type="neumann")
    with pytest.raises(AssertionError, match="higher order"):
        hg.neumann(vector, curr, prev, False)


def test_config_matches_reference_defaults():
    c = Config()
    assert (c.type, c.cg_iterations, c.cg_alpha, c.neumann_iterations, c.neumann_alpha) == ("darts", 1, 1.0, 1, 1.0)
    assert (c.darts_alpha, c.darts_multitask, c.first_order, c.retain_graph, c.allow_unused) == (0.01, False, True, False, True)
    assert (c.gradient_accumulation, c.precision, c.unroll_steps) == (1, "fp32", 1)


def test_install_mutates_registry_in_place():
    class FakeRef:
        jvp_fn_mapping = {"cg": 1, "neumann": 2, "darts": 3, "sama": 4, "reinforce": 5}

    mapping = FakeRef.jvp_fn_mapping
    betty_amd.install(FakeRef)
    assert FakeRef.jvp_fn_mapping is mapping
    assert mapping["cg"] is hg.cg and mapping["neumann"] is hg.neumann and mapping["darts"] is hg.darts
    assert mapping["sama"] is hg.sama and mapping["reinforce"] == 5 and mapping["cg_global"] is hg.cg_global and mapping["neumann_global"] is hg.neumann_global


@pytest.mark.parametrize("name", ["reweight_cg20", "reweight_neumann10", "deep_cg6", "deep_neumann6"])
@pytest.mark.parametrize("sync", [False, True])
def test_analytic_mlp_hvp_matches_reference(name, sync, checker):
    """The closed-form R-op HVP + mixed VJP, evaluated with ATen ops, through
    the same cg/neumann host path: must reproduce the reference's autograd result."""
    case = zoo.CASE_BY_NAME[name]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    zoo.attach_mlp_structure(curr, case.family, impl="torch")
    out = hg.jvp_fn_mapping[case.algo](vector, curr, prev, sync)
    if sync:
        assert out is None
        out = [p.grad for p in prev.trainable_parameters()]
    rel, mx = rel_err([o.detach().numpy() for o in out], golden_list(outputs, case.name, "fp32"))
    assert rel <= 1e-4 and mx <= 1e-3, (rel, mx)


def test_analytic_mlp_hvp_equals_autograd_hvp_fp64():
    """HVP and mixed VJP against double backward, fp64, one random direction."""
    case = zoo.CASE_BY_NAME["reweight_cg20"]
    inputs, _ = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config, dtype=torch.float64)
    zoo.attach_mlp_structure(curr, case.family, impl="torch")
    prov = curr.hypergradient_structure(prev)
    hvp_fn = prov.prepare()
    got = [h + prov.hvp_shift * v for h, v in zip(hvp_fn(vector), vector)]  # + ridge part (applied by the recurrence kernel)
    loss = curr.training_step_exec(curr.cur_batch)
    g = torch.autograd.grad(loss, curr.parameters(), create_graph=True)
    want = torch.autograd.grad(g, curr.parameters(), grad_outputs=vector, retain_graph=True)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-10, atol=1e-14)
    mixed = prov.mixed_vjp(vector, False)
    want_m = torch.autograd.grad(g, prev.trainable_parameters(), grad_outputs=vector)
    for a, b in zip(mixed, want_m):
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-9, atol=1e-14)


def test_structured_provider_has_no_silent_cpu_path():
    """Default impl is the HIP one: CPU tensors raise instead of falling back to ATen."""
    from betty_amd import NativeLibraryError

    case = zoo.CASE_BY_NAME["reweight_cg20"]
    inputs, _ = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    zoo.attach_mlp_structure(curr, case.family, impl=None)
    with pytest.raises(NativeLibraryError, match="no CPU fallback"):
        curr.hypergradient_structure(prev).prepare()


@pytest.mark.parametrize("sync", [False, True])
def test_proximal_structure_matches_reference(sync, checker):
    """iMAML closed form: data-loss HVP + shift 2*reg, mixed VJP = +2*reg*x."""
    case = zoo.CASE_BY_NAME["imaml_cg10"]
    inputs, outputs = load_golden(case.family)
    curr, prev, vector = zoo.build_case(case, inputs, Config)
    zoo.attach_prox_structure(curr)
    out = hg.cg(vector, curr, prev, sync)
    if sync:
        assert out is None
        out = [p.grad for p in prev.trainable_parameters()]
    rel, mx = rel_err([o.detach().numpy() for o in out], golden_list(outputs, case.name, "fp32"))
    assert rel <= 1e-4 and mx <= 1e-3, (rel, mx)



def test_fsdp_first_hop_gradient_matches_oracle(checker):
    """`get_grads` with an FSDP upper problem (__init__.py:23-30): the first hop goes through `.grad`."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import hypergrad_oracle as orc
    from betty_amd.hypergradient.utils import grad

    torch.manual_seed(3)
    lin = torch.nn.Linear(6, 4)
    x = torch.randn(5, 6)
    params = list(lin.parameters())
    for p in params:
        p.grad = torch.randn_like(p)
    keep = [p.grad.clone() for p in params]
    loss = (lin(x) ** 2).sum()
    want = orc.first_order_grad(loss, params, retain_graph=True, through_grad_field=True)
    got = grad(loss, params, retain_graph=True, is_fsdp=True)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g.numpy(), w.numpy())
    for p, k in zip(params, keep):
        np.testing.assert_array_equal(p.grad.numpy(), k.numpy())
    # an unused parameter has no .grad to restore: zeros, no exception
    extra = torch.nn.Parameter(torch.ones(3))
    out = grad((lin(x) ** 2).sum(), params + [extra], is_fsdp=True)
    assert float(out[-1].abs().sum()) == 0.0 and extra.grad is None


@pytest.mark.parametrize("dims,B", [([48, 64, 32, 24, 10], 40), ([20, 7], 9), ([30, 16, 5], 12)])
def test_step_length_identity_of_the_fused_cg_solver_fp64(dims, B):
    """The fused solver (csrc/bhg_mlp.hip, k_cg_alpha) never forms the N-sized H p; its step length uses
        p.Hp = sum_b Rz_b.Rd_L,b + 2 sum_{l>=1} <delta_l V_l, Rh_{l-1}> + shift * p.p
    from batch-sized factors of the R-chain.  Checked here in fp64 against p . (autograd double backward)."""
    from betty_amd.hypergradient.structured import WeightedCEMLP

    g = torch.Generator().manual_seed(sum(dims) + B)
    inner, upper = zoo.MLP(dims).double(), zoo.MWN(8).double()
    x = torch.randn(B, dims[0], generator=g, dtype=torch.float64)
    y = torch.randint(0, dims[-1], (B,), generator=g)
    ridge = 0.05
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type="cg"), loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x, y))
    prov = WeightedCEMLP(curr, prev, layers=list(inner.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)),
                         ridge=ridge, impl="torch")
    prov.prepare()
    st = prov._state
    p = [torch.randn(q.shape, generator=g, dtype=torch.float64) for q in inner.parameters()]
    loss = curr.training_step_exec(curr.cur_batch)
    grad = torch.autograd.grad(loss, curr.parameters(), create_graph=True)
    Hp = torch.autograd.grad(grad, curr.parameters(), grad_outputs=p)
    direct = sum((a * b).sum() for a, b in zip(Hp, p))
    Vs, cs = p[0::2], p[1::2]
    Rz, Rhs = st._r_forward(Vs, cs)
    Rd_top = st.sd[:, None] * (st.p * Rz - st.p * (st.p * Rz).sum(1, keepdim=True))
    t1 = (Rz * Rd_top).sum()
    t2 = sum(2.0 * ((st.deltas[l] @ Vs[l]) * Rhs[l]).sum() for l in range(1, len(Vs)))
    pp = sum((q * q).sum() for q in p)
    factored = t1 + t2 + prov.hvp_shift * pp
    assert abs(float(direct - factored)) <= 1e-12 * abs(float(direct))


# ------------------------------------------------------------------------------------------------
# solution-free protocol of the structured providers (host orchestration only; the kernels behind it are
# tested on the GPU: test_fused_cg_without_a_solution_vector, test_fused_neumann_without_an_accumulator_vector)
# ------------------------------------------------------------------------------------------------
class _RecordingBackend:
    """Records what cg() / neumann() hand to the backend; the state vectors are plain CPU tensors."""

    def __init__(self):
        self.calls = []

    def layout(self, tensors):
        from betty_amd.flat import layout_for

        return layout_for(tensors)

    def cg_init(self, layout, vector, x, r, p):
        self.calls.append(("cg_init", x is None))

    def neumann_init(self, layout, vector, v, p):
        self.calls.append(("neumann_init", p is None))

    def after_cg(self, layout):
        self.calls.append(("after_cg",))


class _FakeFusedProvider:
    """A structured provider whose fused solvers exist and, depending on `free`, leave the solution untouched."""

    hvp_shift = 0.0

    def __init__(self, prev, free, ready=True):
        self.prev, self.free, self.ready, self.seen = prev, free, ready, []

    def prepare(self):
        return lambda views: (_ for _ in ()).throw(AssertionError("the un-fused loop must not run"))

    def fused_cg_skips_solution(self, layout, K):
        return self.free and self.ready

    def fused_neumann_skips_solution(self, layout, K):
        return self.free and self.ready

    def fused_cg(self, layout, x, r, p, K, alpha):
        self.seen.append(("fused_cg", K, alpha))
        return self.ready

    def fused_neumann(self, layout, v, p, K, alpha):
        self.seen.append(("fused_neumann", K, alpha))
        return self.ready

    def mixed_vjp(self, views, sync):
        self.seen.append(("mixed_vjp", [tuple(v.shape) for v in views], sync))
        return [torch.zeros_like(q) for q in self.prev.trainable_parameters()]


class _TokenProvider(_FakeFusedProvider):
    """A provider whose fused solvers return a token object: cg() / neumann() must hand exactly that object back."""

    class Token:
        pass

    def fused_cg(self, layout, x, r, p, K, alpha):
        self.token = self.Token()
        return self.token

    fused_neumann = lambda self, layout, v, p, K, alpha: self.fused_cg(layout, None, None, p, K, alpha)  # noqa: E731

    def mixed_vjp(self, views, sync, solve=None):
        self.seen.append(("mixed_vjp", solve))
        return [torch.zeros_like(q) for q in self.prev.trainable_parameters()]


@pytest.mark.parametrize("algo", ["cg", "neumann"])
def test_fused_solve_token_travels_back_to_mixed_vjp(algo):
    """The solution of a fused solve is identified by the token the solver returned (object identity), never by the
    addresses of the views (round-2 finding: `data_ptr()` arithmetic)."""
    inner, upper = zoo.MLP([6, 5, 3]), zoo.MWN(4)
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type=algo, cg_iterations=2, neumann_iterations=2, neumann_alpha=0.1))
    provider = _TokenProvider(prev, free=True)
    curr.hypergradient_structure = lambda prev_: provider
    vector = [torch.randn_like(p) for p in inner.parameters()]
    with use_backend(_RecordingBackend()):
        hg.jvp_fn_mapping[algo](vector, curr, prev, False)
    assert provider.seen[-1] == ("mixed_vjp", provider.token)


@pytest.mark.parametrize("algo", ["cg", "neumann"])
@pytest.mark.parametrize("free", [True, False])
def test_solution_free_protocol_of_structured_providers(algo, free):
    """cg() / neumann() skip zeroing (and never pass) the solution / accumulator vector exactly when the provider's fused
    solver declares it leaves that vector untouched; the provider's mixed_vjp still receives views shaped like the
    inner parameters (they NAME the solve; a solution-free provider does not read them)."""
    inner, upper = zoo.MLP([6, 5, 3]), zoo.MWN(4)
    prev = zoo.StubProblem("upper", upper, config=Config())
    cfg = Config(type=algo, cg_iterations=3, neumann_iterations=3, cg_alpha=1.0, neumann_alpha=0.1)
    curr = zoo.StubProblem("inner", inner, config=cfg)
    provider = _FakeFusedProvider(prev, free)
    curr.hypergradient_structure = lambda prev_: provider
    vector = [torch.randn_like(p) for p in inner.parameters()]
    be = _RecordingBackend()
    with use_backend(be):
        out = hg.jvp_fn_mapping[algo](vector, curr, prev, False)
    assert len(out) == len(list(prev.trainable_parameters()))
    assert be.calls[0] == (f"{algo}_init", free)
    assert provider.seen[0][0] == f"fused_{algo}" and provider.seen[0][1] == 3
    kind, shapes, sync = provider.seen[-1]
    assert kind == "mixed_vjp" and sync is False and shapes == [tuple(p.shape) for p in inner.parameters()]


# ---- the structure guard: a declared closed form is checked against the problem's real training_step once ------------------------
def _guard_case(smoothing):
    import torch.nn.functional as F

    import zoo
    from betty_amd import Config
    from betty_amd.hypergradient.structured import WeightedCEMLP

    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    dims, B, ridge = [24, 32, 16, 5], 12, 0.05
    inner = zoo.MLP(dims).double() if hasattr(zoo, "MLP") else None
    if inner is None:
        class _M(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.layers = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

            def forward(self, x):
                for i, lin in enumerate(self.layers):
                    x = lin(x)
                    if i + 1 < len(self.layers):
                        x = F.relu(x)
                return x

        inner = _M().double()
    upper = torch.nn.Sequential(torch.nn.Linear(1, 8), torch.nn.ReLU(), torch.nn.Linear(8, 1), torch.nn.Sigmoid()).double()
    x = torch.randn(B, dims[0], generator=g, dtype=torch.float64)
    y = torch.randint(0, dims[-1], (B,), generator=g)
    prev = zoo.StubProblem("upper", upper, config=Config())

    def loss_fn(self, batch):
        xb, yb = batch
        logits = self.module(xb)
        ce = F.cross_entropy(logits, yb, reduction="none", label_smoothing=smoothing)
        w = prev.fwd(F.cross_entropy(logits, yb, reduction="none").detach().reshape(-1, 1)).reshape(-1)
        return torch.mean(w * ce) + ridge * sum((p * p).sum() for p in self.module.parameters())

    curr = zoo.StubProblem("inner", inner, config=Config(type="cg"), loss_fn=loss_fn, batch=(x, y))
    return curr, WeightedCEMLP(curr, prev, layers=list(inner.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="torch")


def test_structure_guard_accepts_the_declared_loss_and_caches_the_verdict():
    curr, prov = _guard_case(0.0)
    prov.prepare()
    assert len(curr._bhg_structure_verified) == 1
    from betty_amd.hypergradient import structured

    calls = []
    orig = structured.WeightedCEMLP._verify_against_autograd

    def spy(self, x, y):
        before = len(self.curr._bhg_structure_verified)
        orig(self, x, y)
        calls.append(len(self.curr._bhg_structure_verified) - before)

    structured.WeightedCEMLP._verify_against_autograd = spy
    try:
        prov.prepare()
    finally:
        structured.WeightedCEMLP._verify_against_autograd = orig
    assert calls == [0]   # second prepare: the cached verdict, no second double backward needed


def test_structure_guard_rejects_a_training_step_the_closed_form_does_not_describe():
    """Label smoothing in the user's loss: the declared WeightedCEMLP would silently give wrong hypergradients."""
    from betty_amd.hypergradient.structured import StructureMismatchError

    curr, prov = _guard_case(0.2)
    with pytest.raises(StructureMismatchError, match="label smoothing"):
        prov.prepare()
    prov.verify = False          # the documented opt-out
    prov.prepare()


def test_cfg5_named_network_is_the_references_own():
    """BASELINE cfg 5 names `Network(16, 10, 8)` + `Architecture` of examples/neural_architecture_search/model_search.py:129-234,
    302-317.  The GPU test builds exactly that model from the reference's files (staged test-only by `make -C oracle ref`): here,
    without a GPU, its size is pinned — 1,930,618 parameters in 1,399 tensors, 2 x 14 x 8 = 224 architecture parameters — and the
    seeded problem zoo.cfg5_as_named_case builds is the one the committed reference-CPU golden was generated from (checksums)."""
    if zoo.nas_dir() is None:
        pytest.skip("reference NAS example not staged (make -C oracle ref)")
    curr, prev, vector = zoo.cfg5_as_named_case(Config, torch.device("cpu"))
    assert (sum(p.numel() for p in curr.parameters()), len(list(curr.parameters()))) == (1_930_618, 1_399)
    assert sum(p.numel() for p in prev.parameters()) == 224
    assert curr.config.type == "neumann" and curr.config.neumann_iterations == 20
    gold_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg5_as_named.npz")
    if os.path.isfile(gold_path):
        gold = np.load(gold_path)
        np.testing.assert_array_equal(zoo.cfg5_checksums(curr, prev, vector), gold["checksum"])
        assert gold["neumann20/fp32"].shape == (224,) and np.isfinite(gold["neumann20/fp32"]).all()
        # the reference's own fp32 run is the reference's own fp64 run to its stated spread
        spread = np.linalg.norm(gold["neumann20/fp32"] - gold["neumann20/fp64"]) / np.linalg.norm(gold["neumann20/fp64"])
        assert abs(spread - float(gold["neumann20/ref_spread"])) <= 1e-12 + 1e-6 * spread


# ---- round 5: forward-over-reverse Hessian-vector products (opt-in: inner_problem.hypergradient_hvp) ------------------------------
class _ConvBNDrop(torch.nn.Module):
    """Depthwise + pointwise convolution, batch norm with running statistics, dropout: the operators whose double backward the
    forward-over-reverse pass replaces, and the two kinds of state a repeated training_step could disturb."""

    def __init__(self):
        super().__init__()
        self.dw = torch.nn.Conv2d(4, 4, 3, padding=1, groups=4, bias=False)
        self.pw = torch.nn.Conv2d(4, 6, 1, bias=False)
        self.bn = torch.nn.BatchNorm2d(6, affine=False)
        self.drop = torch.nn.Dropout(0.25)
        self.fc = torch.nn.Linear(6, 3)

    def forward(self, x, gate):
        h = torch.tanh(self.bn(self.pw(self.dw(x)))) * gate.reshape(1, -1, 1, 1)
        return self.fc(self.drop(h.mean((2, 3))))


class _Gate(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.g = torch.nn.Parameter(torch.linspace(0.5, 1.5, 6))

    def forward(self):
        return torch.sigmoid(self.g)


def _convbn_case(algo, K, seed=3):
    torch.manual_seed(seed)
    inner, upper = _ConvBNDrop(), _Gate()
    g = torch.Generator().manual_seed(seed)
    x, y = torch.randn(8, 4, 6, 6, generator=g), torch.randint(0, 3, (8,), generator=g)
    vector = [0.1 * torch.randn(p.shape, generator=g) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    cfg = Config(type="cg", cg_iterations=K, cg_alpha=1.0) if algo == "cg" else Config(type="neumann", neumann_iterations=K, neumann_alpha=0.05)

    def loss_fn(self, batch):
        xb, yb = batch
        out = self.module(xb, prev.module())
        return torch.nn.functional.cross_entropy(out, yb) + 0.5 * sum((p * p).sum() for p in self.module.parameters())

    curr = zoo.StubProblem("inner", inner, config=cfg, loss_fn=loss_fn, batch=(x, y))
    return curr, prev, vector


@pytest.mark.parametrize("algo,K", [("cg", 4), ("neumann", 5)])
@pytest.mark.parametrize("sync", [False, True])
def test_forward_over_reverse_hvp_matches_the_double_backward(algo, K, sync, checker):
    """Same solve twice on the same seeded problem — the reference's double backward (default) and the opt-in forward-over-reverse
    passes — through the host path: same hypergradient to rounding, same RNG state afterwards (dropout drew ONE set of masks, as in the
    reference's single graph), batch-norm running statistics advanced exactly once in both."""
    res = {}
    for mode in (None, "forward_over_reverse"):
        curr, prev, vector = _convbn_case(algo, K)
        if mode:
            curr.hypergradient_hvp = mode
        torch.manual_seed(11)
        out = hg.jvp_fn_mapping[algo](vector, curr, prev, sync)
        if sync:
            assert out is None
            out = [p.grad for p in prev.trainable_parameters()]
        res[mode] = ([o.detach().numpy().astype(np.float64) for o in out], torch.get_rng_state().clone(),
                     curr.module.bn.running_mean.clone(), int(curr.module.bn.num_batches_tracked))
    rel, _ = rel_err(res["forward_over_reverse"][0], res[None][0])
    assert rel <= 2e-5, rel
    assert torch.equal(res["forward_over_reverse"][1], res[None][1]), "the K + 1 passes must leave the RNG where ONE training_step leaves it"
    assert res["forward_over_reverse"][3] == res[None][3] == 1
    torch.testing.assert_close(res["forward_over_reverse"][2], res[None][2], rtol=1e-6, atol=1e-7)


def test_forward_over_reverse_falls_back_when_an_operator_has_no_forward_formula(checker):
    """A training_step through a custom autograd.Function without a jvp: the first pass raises, the solve continues on the double
    backward (with a warning) and gives the default's answer."""

    class Square(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            ctx.save_for_backward(x)
            return x * x

        @staticmethod
        def backward(ctx, g):
            (x,) = ctx.saved_tensors
            return 2.0 * x * g

    res = {}
    for mode in (None, "forward_over_reverse"):
        torch.manual_seed(5)
        inner = torch.nn.Linear(5, 1, bias=False)
        upper = _Gate()
        g = torch.Generator().manual_seed(5)
        x = torch.randn(16, 5, generator=g)
        prev = zoo.StubProblem("upper", upper, config=Config())

        def loss_fn(self, batch, prev=prev):
            return Square.apply(self.module(batch)).mean() * prev.module().sum() + (self.module.weight ** 2).sum()

        curr = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=3, neumann_alpha=0.05), loss_fn=loss_fn, batch=x)
        if mode:
            curr.hypergradient_hvp = mode
        vector = [0.1 * torch.randn(p.shape, generator=g) for p in inner.parameters()]
        if mode:
            with pytest.warns(RuntimeWarning, match="forward-over-reverse HVP not available"):
                out = hg.neumann(vector, curr, prev, False)
        else:
            out = hg.neumann(vector, curr, prev, False)
        res[mode] = [o.detach().numpy().astype(np.float64) for o in out]
    rel, _ = rel_err(res["forward_over_reverse"], res[None])
    assert rel <= 1e-6, rel


# ---- round 6: the deferred data-parallel mean of an accumulated hypergradient (betty_amd/distributed.py) ---------------------------
class _FakeWork:
    def __init__(self):
        self.waits = 0

    def wait(self):
        self.waits += 1


def test_deferred_grad_sync_is_fenced_once_and_by_the_optimizer_step():
    from betty_amd import distributed as bd

    assert bd.fence_grads() == 0
    w1, w2 = _FakeWork(), _FakeWork()
    bd.defer_grad_sync(w1, torch.zeros(4))
    bd.defer_grad_sync(w2, torch.zeros(4))
    assert bd.pending_grad_syncs() == 2
    assert bd.fence_grads() == 2 and (w1.waits, w2.waits) == (1, 1) and bd.pending_grad_syncs() == 0
    assert bd.fence_grads() == 0 and (w1.waits, w2.waits) == (1, 1)          # a second fence does nothing
    # optimizer.step() reads .grad: the pre-hook fences first (idempotent installation)
    p = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.SGD([p], lr=0.1)
    assert bd.install_optimizer_fence(opt) and bd.install_optimizer_fence(opt)
    p.grad = torch.ones(3)
    w3 = _FakeWork()
    bd.defer_grad_sync(w3, p.grad)
    opt.step()
    assert w3.waits == 1 and bd.pending_grad_syncs() == 0
    opt.step()
    assert w3.waits == 1
    assert bd.install_optimizer_fence(None) is False


def test_ddp_wrapper_names_the_group_whose_mean_the_reference_would_take():
    from betty_amd import distributed as bd

    lin = torch.nn.Linear(2, 2)
    assert bd.ddp_process_group_of(lin, None) == (False, None)

    class _Fake(torch.nn.parallel.DistributedDataParallel):   # (constructing a real wrapper needs a process group: tests/test_distributed_cpu.py)
        def __init__(self):   # noqa: D107
            torch.nn.Module.__init__(self)
            self.process_group = "the-group"

    assert bd.ddp_process_group_of(None, _Fake()) == (True, "the-group")


def test_forward_over_reverse_says_so_on_train_mode_batchnorm_and_can_be_acknowledged(checker):
    """the opt-in passes measured 4.4e-3 off on a dense-convolution batch-norm net; a user who sets the flag on such a module
    is told once per problem, and can acknowledge it."""
    import warnings as _w

    curr, prev, vector = _convbn_case("neumann", 2)
    curr.hypergradient_hvp = "forward_over_reverse"
    with pytest.warns(RuntimeWarning, match="train-mode batch-norm"):
        hg.neumann(vector, curr, prev, False)
    with _w.catch_warnings():
        _w.simplefilter("error")
        hg.neumann(vector, curr, prev, False)              # once per problem
        curr2, prev2, vector2 = _convbn_case("neumann", 2)
        curr2.hypergradient_hvp, curr2.hypergradient_hvp_ack_batchnorm = "forward_over_reverse", True
        hg.neumann(vector2, curr2, prev2, False)           # acknowledged
        curr3, prev3, vector3 = _convbn_case("neumann", 2)
        curr3.module.eval()                                 # eval-mode statistics: nothing to say
        curr3.hypergradient_hvp = "forward_over_reverse"
        hg.neumann(vector3, curr3, prev3, False)


def test_forward_over_reverse_fallback_is_only_for_a_missing_formula_and_leaves_buffers_advanced_once(checker):
    """(a) a RuntimeError that is NOT a missing forward-mode formula (here: a shape bug in training_step) surfaces as it is;
    (b) when the fall-back does run on a module with batch-norm statistics, they advance once per solve, as in the default path."""
    curr, prev, vector = _convbn_case("neumann", 2)
    curr.hypergradient_hvp, curr.hypergradient_hvp_ack_batchnorm = "forward_over_reverse", True

    def broken(self, batch):
        xb, _ = batch
        return (self.module(xb, prev.module()) @ torch.ones(7)).sum()      # 3 classes against 7: a shape error, not a missing jvp

    curr._loss_fn = broken
    with pytest.raises(RuntimeError, match="shape|size|mat"):
        hg.neumann(vector, curr, prev, False)

    class NoJvp(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            return g

    res = {}
    for mode in (None, "forward_over_reverse"):
        curr, prev, vector = _convbn_case("neumann", 2)

        def loss_fn(self, batch, prev=prev):
            xb, yb = batch
            return torch.nn.functional.cross_entropy(NoJvp.apply(self.module(xb, prev.module())), yb) + 0.5 * sum((p * p).sum() for p in self.module.parameters())

        curr._loss_fn = loss_fn
        if mode:
            curr.hypergradient_hvp, curr.hypergradient_hvp_ack_batchnorm = mode, True
            with pytest.warns(RuntimeWarning, match="forward-over-reverse HVP not available"):
                torch.manual_seed(11)
                out = hg.neumann(vector, curr, prev, False)
        else:
            torch.manual_seed(11)
            out = hg.neumann(vector, curr, prev, False)
        res[mode] = ([o.detach().numpy().astype(np.float64) for o in out], curr.module.bn.running_mean.clone(), int(curr.module.bn.num_batches_tracked))
    assert res["forward_over_reverse"][2] == res[None][2] == 1
    torch.testing.assert_close(res["forward_over_reverse"][1], res[None][1], rtol=1e-6, atol=1e-7)
    rel, _ = rel_err(res["forward_over_reverse"][0], res[None][0])
    assert rel <= 1e-6, rel


def test_structure_guard_does_not_cache_a_pass_under_the_kink_tolerance_at_once():
    """a batch with a hidden pre-activation on the ReLU kink relaxes the check 30x — such a pass is said aloud and NOT cached
    (the next batch is checked strictly again); a problem that only ever shows such a batch is accepted after VERIFY_KINK_RETRIES."""
    import warnings as _w

    from betty_amd.hypergradient import structured

    smoothing = 0.004
    curr, prov = _guard_case(smoothing)
    x, _ = curr.cur_batch
    lin = prov.layers[0]
    with torch.no_grad():   # put sample 0's first hidden unit exactly on the kink
        lin.bias[0] -= (lin.weight[0] @ x[0].reshape(-1) + lin.bias[0])
    # the mild smoothing must sit between the two tolerances for the test to mean anything
    structured_tol = (structured.VERIFY_RTOL, structured.VERIFY_RTOL_ON_A_KINK)
    for n in range(1, structured.VERIFY_KINK_RETRIES + 1):
        with _w.catch_warnings(record=True) as rec:
            _w.simplefilter("always")
            prov.prepare()
        msgs = [str(r.message) for r in rec if "ReLU-kink tolerance" in str(r.message)]
        cached = len(curr.__dict__.get("_bhg_structure_verified", ()))
        if n < structured.VERIFY_KINK_RETRIES:
            assert cached == 0, (n, structured_tol)
            assert (len(msgs) == 1 and "not cached" in msgs[0]) if n == 1 else not msgs
        else:
            assert cached == 1 and len(msgs) == 1 and "accepted after" in msgs[0]
    # the same declaration on a batch clear of the kink is rejected at the strict tolerance
    curr2, prov2 = _guard_case(smoothing)
    with pytest.raises(structured.StructureMismatchError):
        prov2.prepare()


# ---- round 6: betty_amd.install(auto_structure=True) — recognising the Linear / ReLU stack without a declaration -----------------
@pytest.fixture()
def auto_structure():
    from betty_amd.hypergradient import structured

    saved = (structured.AUTO_STRUCTURE, structured.AUTO_IMPL)
    structured.AUTO_STRUCTURE, structured.AUTO_IMPL = True, "torch"   # (the ATen closed form: the CPU suite has no GPU)
    try:
        yield structured
    finally:
        structured.AUTO_STRUCTURE, structured.AUTO_IMPL = saved


def _undeclared_case(smoothing=0.0, ridge=0.05, upper_act="relu", dropout=0.0, dtype=torch.float32):
    torch.manual_seed(21)
    g = torch.Generator().manual_seed(21)
    dims, B = [12, 16, 8, 4], 10

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
            self.drop = torch.nn.Dropout(dropout)

        def forward(self, x):
            for i, lin in enumerate(self.layers):
                x = lin(x)
                if i + 1 < len(self.layers):
                    x = self.drop(torch.relu(x))
            return x

    class MWN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2 = torch.nn.Linear(1, 8), torch.nn.Linear(8, 1)

        def forward(self, c):
            h = self.l1(c)
            return torch.sigmoid(self.l2(torch.relu(h) if upper_act == "relu" else torch.tanh(h)))

    inner, upper = Net().to(dtype), MWN().to(dtype)
    x, y = torch.randn(B, dims[0], generator=g, dtype=dtype), torch.randint(0, dims[-1], (B,), generator=g)
    prev = zoo.StubProblem("upper", upper, config=Config())

    def loss_fn(self, batch):
        xb, yb = batch
        logits = self.module(xb)
        ce = torch.nn.functional.cross_entropy(logits, yb, reduction="none", label_smoothing=smoothing)
        w = prev.fwd(torch.nn.functional.cross_entropy(logits, yb, reduction="none").detach().reshape(-1, 1)).reshape(-1)
        return torch.mean(w * ce) + ridge * sum((p * p).sum() for p in self.module.parameters())

    curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=4, cg_alpha=1.0), loss_fn=loss_fn, batch=(x, y))
    vector = [0.1 * torch.randn(p.shape, generator=g, dtype=dtype) for p in inner.parameters()]
    return curr, prev, vector


def _opaque(curr, prev, vector, structured):
    structured.AUTO_STRUCTURE = False
    try:
        return [t.detach().numpy() for t in hg.cg(vector, curr, prev, False)]
    finally:
        structured.AUTO_STRUCTURE = True


def test_auto_structure_recognises_the_reweighting_mlp_and_its_ridge(auto_structure, checker):
    curr, prev, vector = _undeclared_case(ridge=0.05)
    assert not hasattr(curr, "hypergradient_structure")
    want = _opaque(curr, prev, vector, auto_structure)
    s0 = dict(auto_structure.AUTO_STATS)
    prov = auto_structure.structured_hvp_for(curr, prev)
    assert isinstance(prov, auto_structure.WeightedCEMLP) and prov.ridge == 0.05 and prov.weight_net is not None
    assert auto_structure.AUTO_STATS["accepted"] == s0["accepted"] + 1
    got = [t.detach().numpy() for t in hg.cg(vector, curr, prev, False)]
    assert auto_structure.AUTO_STATS["looked"] == s0["looked"] + 1, "the verdict is cached on the problem: one look"
    rel, _ = rel_err(got, want)
    assert rel <= 1e-4, rel
    # no ridge in the loss: recognised with ridge 0
    curr0, prev0, _ = _undeclared_case(ridge=0.0)
    assert auto_structure.structured_hvp_for(curr0, prev0).ridge == 0.0


@pytest.mark.parametrize("why", ["label_smoothing", "dropout", "no_tuple_batch", "extra_parameter", "declared_wins"])
def test_auto_structure_leaves_everything_else_on_the_opaque_path(why, auto_structure, checker):
    kw = {"label_smoothing": dict(smoothing=0.1), "dropout": dict(dropout=0.3)}.get(why, {})
    curr, prev, vector = _undeclared_case(**kw)
    if why == "no_tuple_batch":
        x, y = curr.cur_batch
        curr.cur_batch = {"x": x, "y": y}
        curr._loss_fn = (lambda f: lambda self, batch: f(self, (batch["x"], batch["y"])))(curr._loss_fn)
    if why == "extra_parameter":
        curr.module.scale = torch.nn.Parameter(torch.ones(()))
    if why == "declared_wins":
        sentinel = object()
        curr.hypergradient_structure = lambda prev_: sentinel
        assert auto_structure.structured_hvp_for(curr, prev) is sentinel
        return
    s0 = dict(auto_structure.AUTO_STATS)
    assert auto_structure.structured_hvp_for(curr, prev) is None
    if why in ("label_smoothing",):
        assert auto_structure.AUTO_STATS["rejected"] == s0["rejected"] + 1
    if why != "extra_parameter":   # (the stub's parameter list no longer matches its vector there)
        torch.manual_seed(5)
        got = [t.detach().numpy() for t in hg.cg(vector, curr, prev, False)]   # silently opaque: still the reference's answer
        torch.manual_seed(5)
        want = _opaque(curr, prev, vector, auto_structure)
        rel, _ = rel_err(got, want)
        assert rel <= 1e-12, rel


def test_auto_structure_keeps_a_wrong_weight_net_guess_out_but_the_inner_form_in(auto_structure, checker):
    """The upper module LOOKS like the meta-weight-net (Linear(1, H), Linear(H, 1)) but uses tanh: the closed-form weight net is
    rejected by the guard's mixed-derivative check, the inner closed form with weight_fn under autograd is accepted."""
    curr, prev, vector = _undeclared_case(upper_act="tanh", ridge=0.5)   # (2 ridge dominates: four fp32 CG steps stay comparable)
    prov = auto_structure.structured_hvp_for(curr, prev)
    # (impl="torch" never uses the closed-form weight net — both candidates pass there; on the HIP path the first is rejected:
    #  tests/test_dropin_reference.py)
    assert isinstance(prov, auto_structure.WeightedCEMLP)
    want = _opaque(curr, prev, vector, auto_structure)
    got = [t.detach().numpy() for t in hg.cg(vector, curr, prev, False)]
    rel, _ = rel_err(got, want)
    assert rel <= 1e-4, rel


def test_install_switches_auto_structure():
    import betty_amd
    from betty_amd.hypergradient import structured

    class _Reg:
        jvp_fn_mapping = {}

    saved = structured.AUTO_STRUCTURE
    try:
        betty_amd.install(_Reg, auto_structure=True)
        assert structured.AUTO_STRUCTURE is True
        betty_amd.install(_Reg)
        assert structured.AUTO_STRUCTURE is True          # None leaves it
        betty_amd.install(_Reg, auto_structure=False)
        assert structured.AUTO_STRUCTURE is False
    finally:
        structured.AUTO_STRUCTURE = saved


def test_bench_sets_one_hardware_queue_only_for_runs_with_a_collective_stream():
    """bench.py --hw-queues (round 6: a collective's stream is a second hardware queue — 6 us per iteration of the K loop)."""
    import bench

    assert bench.resolve_hw_queues(-1, 1, False, {}) == (None, "runtime default")            # the driver's N = 1 line: untouched
    env = {}
    assert bench.resolve_hw_queues(-1, 8, False, env) == ("1", "set by bench.py") and env == {"GPU_MAX_HW_QUEUES": "1"}
    env = {}
    assert bench.resolve_hw_queues(-1, 1, True, env) == ("1", "set by bench.py")             # --emulate-collective at N = 1
    env = {"GPU_MAX_HW_QUEUES": "4"}
    assert bench.resolve_hw_queues(-1, 8, False, env) == ("4", "from the environment") and env["GPU_MAX_HW_QUEUES"] == "4"   # the user's choice wins
    env = {}
    assert bench.resolve_hw_queues(0, 8, False, env) == (None, "runtime default") and env == {}
    env = {"GPU_MAX_HW_QUEUES": "4"}
    assert bench.resolve_hw_queues(2, 1, False, env) == ("2", "set by bench.py")
