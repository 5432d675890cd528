"""Drop-in under the REAL reference: `betty_amd.install()` replaces the registry entries of a live
`betty.hypergradient` and the reference's own Engine/Problem (unmodified, imported from the checkout named by
$BETTY_REF; default: oracle/_ref — the git-ignored staging copy `make -C oracle ref` / `__graft_entry__.build()` takes
in the build container and gpurun ships to the GPU box — else /root/reference) drives our cg/neumann/darts through
`Problem.backward -> get_grads`.
  * in the build container (no GPU) the kernels are the C oracle via the test-only checker backend;
  * `-m gpu`: the SAME scenario with the reference's Engine over the HIP kernels (`HipBackend`) — the two halves of the
    drop-in claim in one run."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

_STAGED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
REF = os.environ.get("BETTY_REF") or (_STAGED if os.path.isdir(os.path.join(_STAGED, "betty")) else "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "betty")), reason="reference checkout not present")


@pytest.fixture()
def ref():
    sys.path.insert(0, REF)
    try:
        import betty  # noqa: F401
        import betty.hypergradient as bh
        from betty.configs import Config, EngineConfig
        from betty.engine import Engine
        from betty.problems import ImplicitProblem
    finally:
        sys.path.remove(REF)
    saved = dict(bh.jvp_fn_mapping)
    yield dict(bh=bh, Config=Config, EngineConfig=EngineConfig, Engine=Engine, ImplicitProblem=ImplicitProblem, saved_mapping=dict(saved))
    bh.jvp_fn_mapping.clear()
    bh.jvp_fn_mapping.update(saved)


@pytest.mark.parametrize("algo", ["cg", "neumann", "darts"])
def test_reference_engine_runs_on_our_functions(ref, algo):
    from _cpu_checker_backend import CpuCheckerBackend
    from betty_amd.backend import use_backend

    _reference_engine_scenario(ref, algo, use_backend(CpuCheckerBackend()))


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["cg", "neumann", "darts"])
def test_reference_engine_runs_on_the_hip_backend(ref, algo):
    """The reference's unmodified Engine / ImplicitProblem on the MI355X, hypergradients by libbhg's kernels."""
    from betty_amd.backend import get_backend

    assert get_backend().name == "hip"
    _reference_engine_scenario(ref, algo, contextlib.nullcontext())


def _reference_engine_scenario(ref, algo, backend_ctx):
    import betty_amd
    from betty_amd import hypergradient as hg

    Config, EngineConfig, Engine, ImplicitProblem = ref["Config"], ref["EngineConfig"], ref["Engine"], ref["ImplicitProblem"]
    mapping = betty_amd.install(ref["bh"])
    assert mapping is ref["bh"].jvp_fn_mapping and mapping[algo] is hg.jvp_fn_mapping[algo]
    calls = []
    orig = mapping[algo]

    def spy(vector, curr, prev, sync):
        calls.append((curr.name, prev.name, sync, len(vector)))
        return orig(vector, curr, prev, sync)

    mapping[algo] = spy

    class Child(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(20))

        def forward(self, x):
            return x @ self.w, self.w

    class Parent(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.ones(20))

        def forward(self):
            return self.w

    class Outer(ImplicitProblem):
        def training_step(self, batch):
            x, y = batch
            return F.binary_cross_entropy_with_logits(self.inner(x)[0], y)

        def param_callback(self):
            for p in self.trainable_parameters():
                p.data.clamp_(min=1e-8)

    class Inner(ImplicitProblem):
        def training_step(self, batch):
            x, y = batch
            outs, w = self.module(x)
            return F.binary_cross_entropy_with_logits(outs, y) + 0.5 * (self.outer() * w * w).sum()

        def on_inner_loop_start(self):
            self.module.w.data.zero_()

    rng = np.random.RandomState(0)
    torch.manual_seed(0)
    w_gt = rng.randn(20)
    x = rng.randn(1000, 20)
    y = ((x @ w_gt + 0.1 * rng.randn(1000)) > 0).astype(np.float32)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    parent, child = Parent(), Child()
    cfgs = {
        "darts": Config(unroll_steps=100),
        "cg": Config(type="cg", cg_iterations=3, cg_alpha=0.1, unroll_steps=100),
        "neumann": Config(type="neumann", neumann_iterations=5, unroll_steps=100),
    }
    outer = Outer(name="outer", module=parent, optimizer=torch.optim.SGD(parent.parameters(), lr=1.0, momentum=0.9),
                  train_data_loader=[(t(x[500:]), t(y[500:]))], config=Config())
    inner = Inner(name="inner", module=child, optimizer=torch.optim.SGD(child.parameters(), lr=0.1),
                  train_data_loader=[(t(x[:500]), t(y[:500]))], config=cfgs[algo])
    engine = Engine(config=EngineConfig(train_iters=2000), problems=[outer, inner],
                    dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}})
    with backend_ctx:
        engine.run()
        loss = outer.training_step(outer.cur_batch)
    assert len(calls) == 20  # 2000 iterations / unroll 100
    assert all(c == ("inner", "outer", True, 1) for c in calls)  # the reference calls with sync=True here
    assert float(loss.detach()) < 0.48  # test_regression.py:126,151,176


@pytest.mark.gpu
def test_reference_engine_drives_the_fused_mlp_solver(ref):
    """The headline path under the reference's OWN Engine (missing #2): the learning_to_reweight scenario
    (examples/learning_to_reweight/main.py:117-127 — MWN-weighted cross-entropy, here on a small ReLU-MLP) run twice from the same
    seeds with the reference's unmodified Engine / ImplicitProblem (`Problem.backward -> get_grads`, problem.py:573-581,
    hypergradient/__init__.py:22-39):
      (1) registry untouched: the reference's own `cg` (autograd double backward, cg.py:8-70) on the GPU;
      (2) `betty_amd.install()` + the inner problem declares `hypergradient_structure` -> WeightedCEMLP: every hypergradient is
          ONE `bhg_mlp_cg_solve` (fully projected solver; the launch counters prove it).
    The meta-weight-net's parameters after six outer steps must agree."""
    import betty_amd
    from betty_amd import _native
    from betty_amd.backend import get_backend
    from betty_amd.hypergradient.structured import WeightedCEMLP

    assert get_backend().name == "hip"
    Config, EngineConfig, Engine, ImplicitProblem = ref["Config"], ref["EngineConfig"], ref["Engine"], ref["ImplicitProblem"]
    dims, B, K, steps, ridge = [256, 384, 128, 10], 100, 5, 6, 0.05

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

        def forward(self, x):
            for i, lin in enumerate(self.layers):
                x = lin(x)
                if i + 1 < len(self.layers):
                    x = F.relu(x)
            return x

    class MWN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2 = torch.nn.Linear(1, 32), torch.nn.Linear(32, 1)

        def forward(self, x):
            return torch.sigmoid(self.l2(F.relu(self.l1(x))))

    def run(structured):
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(12)
        net, mwn = MLP(), MWN()
        xt, xv = torch.randn(B, dims[0], generator=g), torch.randn(B, dims[0], generator=g)
        yt, yv = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)

        class Reweight(ImplicitProblem):
            def training_step(self, batch):
                x, y = batch
                return F.cross_entropy(self.classifier(x), y)

        class Classifier(ImplicitProblem):
            def training_step(self, batch):
                x, y = batch
                ce = F.cross_entropy(self.module(x), y, reduction="none")
                w = self.reweight(ce.detach().reshape(-1, 1)).reshape(-1)
                return torch.mean(w * ce) + ridge * sum((p * p).sum() for p in self.module.parameters())

        if structured:
            Classifier.hypergradient_structure = lambda self, prev: WeightedCEMLP(
                self, prev, layers=list(self.module.layers), weight_fn=lambda ce: prev.module(ce.reshape(-1, 1)), ridge=ridge, impl="hip")
        outer = Reweight(name="reweight", module=mwn, optimizer=torch.optim.SGD(mwn.parameters(), lr=0.1),
                         train_data_loader=[(xv, yv)], config=Config())
        inner = Classifier(name="classifier", module=net, optimizer=torch.optim.SGD(net.parameters(), lr=0.05),
                           train_data_loader=[(xt, yt)], config=Config(type="cg", cg_iterations=K, cg_alpha=1.0, unroll_steps=1))
        engine = Engine(config=EngineConfig(train_iters=steps), problems=[outer, inner],
                        dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}})
        engine.run()
        return [p.detach().double().cpu().numpy().copy() for p in mwn.parameters()], [p.detach().double().cpu().numpy().copy() for p in net.parameters()]

    saved = dict(ref["bh"].jvp_fn_mapping)
    ref_upper, ref_inner = run(False)                     # (1) the reference's own cg
    assert ref["bh"].jvp_fn_mapping == saved
    lib = _native.load()
    betty_amd.install(ref["bh"])
    p0 = lib.bhg_mlp_proj_iterations()
    got_upper, got_inner = run(True)                      # (2) libbhg's fused solver under the same Engine
    assert lib.bhg_mlp_proj_iterations() - p0 == steps * (K - 1), "every hypergradient must be one fully projected bhg_mlp_cg_solve"
    num = sum(float(((a - b) ** 2).sum()) for a, b in zip(got_upper, ref_upper)) ** 0.5
    den = sum(float((b ** 2).sum()) for b in ref_upper) ** 0.5
    moved = sum(float(((a - b) ** 2).sum()) for a, b in zip(ref_inner, got_inner)) ** 0.5
    print(f"reference Engine, reweighting MLP {dims}, cg K={K}, {steps} outer steps: meta-weight-net after the run: "
          f"|hip - reference| / |reference| = {num / den:.2e}; inner nets differ by {moved:.2e}")
    assert np.isfinite(num) and num / den <= 1e-4, (num, den)


@pytest.mark.gpu
def test_reference_engine_takes_the_fused_solver_without_a_declaration(ref):
    """the headline path for an UNMODIFIED Betty user.  Same scenario as above, but the inner problem declares nothing:
    `betty_amd.install(auto_structure=True)` looks at it once (Linear / ReLU stack, (x, y) batch, meta-weight-net of two Linear layers),
    estimates the ridge from one Hessian-vector product, puts the closed form through the declaration's own check against autograd —
    and every hypergradient is one fully projected bhg_mlp_cg_solve.  The same problem with label smoothing in its loss is looked at,
    REJECTED by that check, and runs on the opaque path (no projected iteration), silently, with the reference's result."""
    import betty_amd
    from betty_amd import _native
    from betty_amd.backend import get_backend
    from betty_amd.hypergradient import structured

    assert get_backend().name == "hip"
    Config, EngineConfig, Engine, ImplicitProblem = ref["Config"], ref["EngineConfig"], ref["Engine"], ref["ImplicitProblem"]
    dims, B, K, steps, ridge = [256, 384, 128, 10], 100, 5, 6, 0.05

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

        def forward(self, x):
            for i, lin in enumerate(self.layers):
                x = lin(x)
                if i + 1 < len(self.layers):
                    x = F.relu(x)
            return x

    class MWN(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l1, self.l2 = torch.nn.Linear(1, 32), torch.nn.Linear(32, 1)

        def forward(self, x):
            return torch.sigmoid(self.l2(F.relu(self.l1(x))))

    def run(smoothing):
        torch.manual_seed(11)
        g = torch.Generator().manual_seed(12)
        net, mwn = MLP(), MWN()
        xt, xv = torch.randn(B, dims[0], generator=g), torch.randn(B, dims[0], generator=g)
        yt, yv = torch.randint(0, 10, (B,), generator=g), torch.randint(0, 10, (B,), generator=g)

        class Reweight(ImplicitProblem):
            def training_step(self, batch):
                x, y = batch
                return F.cross_entropy(self.classifier(x), y)

        class Classifier(ImplicitProblem):       # nothing declared: an ordinary Betty problem
            def training_step(self, batch):
                x, y = batch
                logits = self.module(x)
                ce = F.cross_entropy(logits, y, reduction="none", label_smoothing=smoothing)
                w = self.reweight(F.cross_entropy(logits, y, reduction="none").detach().reshape(-1, 1)).reshape(-1)
                return torch.mean(w * ce) + ridge * sum((p * p).sum() for p in self.module.parameters())

        outer = Reweight(name="reweight", module=mwn, optimizer=torch.optim.SGD(mwn.parameters(), lr=0.1),
                         train_data_loader=[(xv, yv)], config=Config())
        inner = Classifier(name="classifier", module=net, optimizer=torch.optim.SGD(net.parameters(), lr=0.05),
                           train_data_loader=[(xt, yt)], config=Config(type="cg", cg_iterations=K, cg_alpha=1.0, unroll_steps=1))
        engine = Engine(config=EngineConfig(train_iters=steps), problems=[outer, inner],
                        dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}})
        engine.run()
        return [p.detach().double().cpu().numpy().copy() for p in mwn.parameters()]

    def dist(a, b):
        num = sum(float(((x - y) ** 2).sum()) for x, y in zip(a, b)) ** 0.5
        return num / sum(float((y ** 2).sum()) for y in b) ** 0.5

    lib = _native.load()
    saved_flag = structured.AUTO_STRUCTURE
    try:
        for smoothing in (0.0, 0.1):
            ref["bh"].jvp_fn_mapping.clear()
            ref["bh"].jvp_fn_mapping.update(ref["saved_mapping"])   # the reference's own functions for the baseline run
            want = run(smoothing)                                    # the reference's own cg on this GPU
            betty_amd.install(ref["bh"], auto_structure=True)
            p0, s0 = lib.bhg_mlp_proj_iterations(), dict(structured.AUTO_STATS)
            got = run(smoothing)
            d_proj = lib.bhg_mlp_proj_iterations() - p0
            looked = structured.AUTO_STATS["looked"] - s0["looked"]
            acc, rej = structured.AUTO_STATS["accepted"] - s0["accepted"], structured.AUTO_STATS["rejected"] - s0["rejected"]
            e = dist(got, want)
            print(f"reference Engine, UNDECLARED reweighting MLP {dims}, label smoothing {smoothing}: looked {looked}, accepted {acc}, rejected {rej}, "
                  f"projected iterations {d_proj}; meta-weight-net after {steps} outer steps vs the reference's own cg: {e:.2e}")
            assert looked == 1, "one look per problem: the verdict is cached"
            if smoothing == 0.0:
                assert (acc, rej) == (1, 0) and d_proj == steps * (K - 1), (acc, rej, d_proj)
            else:
                assert (acc, rej) == (0, 1) and d_proj == 0, (acc, rej, d_proj)
            assert np.isfinite(e) and e <= 1e-4, e
            betty_amd.install(ref["bh"], auto_structure=False)
    finally:
        structured.AUTO_STRUCTURE = saved_flag
