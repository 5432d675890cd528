"""The caller slice (betty_amd.problems / betty_amd.engine) around the hot path:
  * path finding matches the reference's contract ([upper, lower..., upper]; test/test_engine.py:124-130
    and the 3-level example of examples/learning_by_ignoring/main.py:327-328),
  * the scenario of the reference's only hot-path test (test/test_regression.py:105-176: logistic
    regression HPO, 2000 iterations, unroll 100, final outer loss < 0.48) passes with cg / neumann /
    darts — on CPU through the test-only checker backend, on the GPU through the HIP kernels.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))

from betty_amd import Config
from betty_amd.engine import Engine, EngineConfig
from betty_amd.problems import ImplicitProblem


class ChildNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(20))

    def forward(self, inputs):
        return inputs @ self.w, self.w


class ParentNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(20))

    def forward(self):
        return self.w * 1.0


class Outer(ImplicitProblem):
    def training_step(self, batch):
        inputs, targets = batch
        return F.binary_cross_entropy_with_logits(self.inner(inputs)[0], targets)

    def param_callback(self):
        for p in self.trainable_parameters():
            p.data.clamp_(min=1e-8)


class Inner(ImplicitProblem):
    def training_step(self, batch):
        inputs, targets = batch
        outs, w = self.module(inputs)
        return F.binary_cross_entropy_with_logits(outs, targets) + 0.5 * (self.outer() * w * w).sum()

    def on_inner_loop_start(self):
        self.module.w.data.zero_()


def _scenario(inner_config, device, structured=False):
    rng = np.random.RandomState(0)
    torch.manual_seed(0)
    w_gt = rng.randn(20)
    x = rng.randn(1000, 20)
    y = ((x @ w_gt + 0.1 * rng.randn(1000)) > 0).astype(np.float32)
    perm = rng.permutation(1000)
    tr, va = perm[:500], perm[500:]
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))
    train_loader = [(t(x[tr]), t(y[tr]))]
    valid_loader = [(t(x[va]), t(y[va]))]
    parent, child = ParentNet(), ChildNet()
    outer = Outer(name="outer", module=parent, optimizer=torch.optim.SGD(parent.parameters(), lr=1.0, momentum=0.9),
                  train_data_loader=valid_loader, config=Config())
    inner = Inner(name="inner", module=child, optimizer=torch.optim.SGD(child.parameters(), lr=0.1),
                  train_data_loader=train_loader, config=inner_config)
    if structured:
        from betty_amd.hypergradient.structured import LogisticRegressionL2

        inner.hypergradient_structure = lambda prev: LogisticRegressionL2(inner, prev, child.w, lam_fn=lambda: prev())
    engine = Engine(config=EngineConfig(train_iters=2000), problems=[outer, inner],
                    dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}}, device=device)
    return engine, outer, inner


CONFIGS = {
    "darts": Config(unroll_steps=100),
    "cg": Config(type="cg", cg_iterations=3, cg_alpha=0.1, unroll_steps=100),
    "neumann": Config(type="neumann", neumann_iterations=5, unroll_steps=100),
}


def test_paths_and_leaves():
    engine, outer, inner = _scenario(CONFIGS["darts"], torch.device("cpu"))
    assert outer.paths == [[outer, inner, outer]]
    assert inner.paths == []
    assert engine.leaves == [inner]
    assert outer.children == [inner] and inner.parents == [outer]
    assert inner.outer is outer and outer.inner is inner


def test_three_level_paths():
    mk = lambda n: ImplicitProblem(name=n, module=torch.nn.Linear(2, 2), config=Config())
    reweight, finetune, pretrain = mk("reweight"), mk("finetune"), mk("pretrain")
    deps = {
        "u2l": {reweight: [pretrain]},
        "l2u": {pretrain: [finetune, reweight], finetune: [reweight]},
    }
    engine = Engine(problems=[reweight, finetune, pretrain], dependencies=deps, device=torch.device("cpu"))
    got = sorted([p.name for p in path] for path in reweight.paths)
    assert got == sorted([["reweight", "finetune", "pretrain", "reweight"], ["reweight", "pretrain", "reweight"]])
    assert engine.leaves == [pretrain]


def test_step_counting():
    engine, outer, inner = _scenario(Config(unroll_steps=10), torch.device("cpu"))
    from _cpu_checker_backend import CpuCheckerBackend
    from betty_amd.backend import use_backend

    engine.config.train_iters = 30
    with use_backend(CpuCheckerBackend()):
        engine.run()
    assert inner.count == 30 and outer.count == 3  # 10 inner steps -> 1 outer step (test_engine.py:146-152)


@pytest.mark.parametrize("algo", ["darts", "cg", "neumann"])
def test_regression_scenario_cpu_checker(algo):
    from _cpu_checker_backend import CpuCheckerBackend
    from betty_amd.backend import use_backend

    engine, outer, inner = _scenario(CONFIGS[algo], torch.device("cpu"))
    with use_backend(CpuCheckerBackend()):
        engine.run()
        loss = outer.training_step(outer.cur_batch)
    assert float(loss.detach()) < 0.48, float(loss.detach())  # test_regression.py:126,151,176


@pytest.mark.gpu
@pytest.mark.parametrize("structured", [False, True], ids=["autograd-hvp", "analytic-hvp"])
@pytest.mark.parametrize("algo", ["darts", "cg", "neumann"])
def test_regression_scenario_gpu(algo, structured):
    if algo == "darts" and structured:
        pytest.skip("darts needs no HVP")
    engine, outer, inner = _scenario(CONFIGS[algo], torch.device("cuda:0"), structured=structured)
    engine.run()
    loss = outer.training_step(outer.cur_batch)
    assert float(loss.detach()) < 0.48, float(loss.detach())


def test_roll_back_snapshot_and_flow():
    """cache_states/recover_states through the flat snapshot, and the roll-back step flow
    (problem.py:379-381,417-436): after every unroll window the inner problem sits at
    (state at loop start) + ONE step on the last batch."""
    from _cpu_checker_backend import CpuCheckerBackend
    from betty_amd.backend import use_backend

    with use_backend(CpuCheckerBackend()):
        # (1) snapshot round trip incl. optimizer state born after the snapshot
        engine, outer, inner = _scenario(Config(type="darts", unroll_steps=5), torch.device("cpu"))
        inner.optimizer = torch.optim.Adam(inner.module.parameters(), lr=0.05)
        w0 = inner.module.w.data.clone()
        inner.cache_states()
        loss = inner.training_step(inner.get_batch())
        loss.backward()
        inner.optimizer.step()  # creates exp_avg / exp_avg_sq, moves w
        assert not torch.equal(inner.module.w.data, w0) and len(inner.optimizer.state[inner.module.w]) > 0
        inner.recover_states()
        assert torch.equal(inner.module.w.data, w0)
        assert len(inner.optimizer.state[inner.module.w]) == 0  # back to "not initialised"
        # with existing state: moments restored bit for bit
        inner.zero_grad()
        inner.training_step(inner.get_batch()).backward()
        inner.optimizer.step()
        m0 = inner.optimizer.state[inner.module.w]["exp_avg"].clone()
        inner.cache_states()
        inner.zero_grad()
        inner.training_step(inner.get_batch()).backward()
        inner.optimizer.step()
        assert not torch.equal(inner.optimizer.state[inner.module.w]["exp_avg"], m0)
        inner.recover_states()
        assert torch.equal(inner.optimizer.state[inner.module.w]["exp_avg"], m0)

        # (2) flow with plain SGD: w_end = w_start - lr * grad(w_start)
        engine, outer, inner = _scenario(Config(type="darts", unroll_steps=5), torch.device("cpu"))
        engine.config.roll_back = True
        engine._parse_dependency()
        assert inner._roll_back and not outer._roll_back
        w_start = torch.zeros(20)  # on_inner_loop_start zeroes the weights
        x, y = inner.train_data_loader[0]
        z = x @ w_start
        lam = outer.module.w.data.clone()
        g = x.t() @ (torch.sigmoid(z) - y) / x.shape[0] + lam * w_start
        engine.config.train_iters = 5
        engine.run()
        assert inner.count == 5 and outer.count == 1
        torch.testing.assert_close(inner.module.w.data, w_start - 0.1 * g, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("algo", ["cg", "neumann", "darts"])
def test_autocast_precision_runs_on_gpu(algo):
    """Config(precision="bf16") wraps training_step in autocast (problem.py:327-332); parameters and the
    hypergradient vectors stay fp32, so the kernels see fp32 tensors (SURVEY §7 'AMP')."""
    cfg = {
        "cg": Config(type="cg", cg_iterations=3, cg_alpha=0.1, unroll_steps=20, precision="bf16"),
        "neumann": Config(type="neumann", neumann_iterations=3, unroll_steps=20, precision="bf16"),
        "darts": Config(type="darts", unroll_steps=20, precision="bf16"),
    }[algo]
    engine, outer, inner = _scenario(cfg, torch.device("cuda:0"))
    engine.config.train_iters = 100
    engine.run()
    assert outer.count == 5
    lam = outer.module.w.detach()
    assert torch.isfinite(lam).all() and (lam - 1.0).abs().max() > 1e-4


_RCCL_WORLD1 = r"""
import os, sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))   # RCCL
import zoo
from conftest import load_golden, golden_list, rel_err
from torch.nn.parallel import DistributedDataParallel as DDP
from betty_amd import Config, hypergradient as hg
from betty_amd.distributed import exchange_async
case = zoo.CASE_BY_NAME["reweight_cg20"]
inputs, outputs = load_golden(case.family)
curr, prev, vector = zoo.build_case(case, inputs, Config, device="cuda:0")
prev.fwd = DDP(prev.module, device_ids=[0], gradient_as_bucket_view=True, find_unused_parameters=True)  # problem.py:220-224
assert hg.cg(vector, curr, prev, True) is None          # sync=True: backward -> DDP reducer -> RCCL all-reduce
got = [p.grad.detach().cpu().numpy() for p in prev.trainable_parameters()]
rel, _ = rel_err(got, golden_list(outputs, case.name, "sync32"))
local = hg.cg(vector, curr, prev, False)
h = exchange_async(local)                               # one flat asynchronous all-reduce on the comm stream
avg = h.wait()
rel2, _ = rel_err([t.cpu().numpy() for t in avg], [t.cpu().numpy() for t in local])
# --- Problem.synchronize_params (problem.py:599-609) as ONE flat collective over RCCL; the early return at world
# size 1 is bypassed by telling the problem it has 2 replicas: broadcast = identity, all-reduce(mean) = w / 2
from betty_amd.problems import ImplicitProblem
mod = zoo.MLP([33, 17, 5]).to("cuda:0")
prob = ImplicitProblem("p", module=mod)
prob._world_size = 2
before = [p.detach().clone() for p in mod.parameters()]
prob.synchronize_params(prob.trainable_parameters())
same = all(torch.equal(a, b) for a, b in zip(before, mod.parameters()))
prob.synchronize_params(prob.trainable_parameters(), all_reduce=True)
half = all(torch.equal(a / 2, b) for a, b in zip(before, mod.parameters()))
# --- a collective in flight makes BHG_CG_AUTO take the streaming CG kernels (the resident kernel's grid barrier needs
# every CU; RCCL's channel kernels hold some): same result as with an idle device, no barrier time-out
from betty_amd.backend import get_backend
from betty_amd import _native
be = get_backend()
big = [torch.randn(30_000_000, device="cuda:0")]
ref = [t.cpu().numpy() for t in hg.cg(vector, curr, prev, False)]
# (exchange_async issues no collective at world size 1, so the in-flight state is produced by hand around a real
#  asynchronous RCCL all-reduce)
w2 = dist.all_reduce(big[0], async_op=True)
be.collectives_in_flight += 1
busy = [t.cpu().numpy() for t in hg.cg(vector, curr, prev, False)]
lay = be.layout(vector)
picked = lay._cg_variant
w2.wait()
be.collectives_in_flight -= 1
be.check_health()
rel3, _ = rel_err(busy, ref)
# the resident kernel forced WHILE a collective runs must still terminate with the right answer or poison it — never hang
w3 = dist.all_reduce(big[0], async_op=True)
be.cg_variant = _native.BHG_CG_RESIDENT
forced = [t.cpu().numpy() for t in hg.cg(vector, curr, prev, False)]
be.cg_variant = _native.BHG_CG_AUTO
w3.wait()
import numpy as np
forced_ok = all(np.isfinite(t).all() for t in forced)
rel4 = rel_err(forced, ref)[0] if forced_ok else -1.0
torch.cuda.synchronize(); dist.barrier(); dist.destroy_process_group()
print("RCCL_OK", rel, rel2, int(same), int(half), picked, rel3, rel4)
"""


@pytest.mark.gpu
def test_rccl_backend_world_size_1():
    """The N > 1 code path on the real collective library: process group "nccl" (= RCCL on ROCm), DDP-wrapped upper
    module, sync=True hop and the flat asynchronous exchange — with one rank, which is all a 1-GPU box can hold."""
    import subprocess
    import sys

    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", _RCCL_WORLD1.format(root=root, tests=HERE)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RCCL_OK")][-1].split()
    assert float(line[1]) <= 1e-4, line       # vs the reference's sync=True golden
    assert float(line[2]) <= 1e-7, line       # mean over one rank = the local result
    assert line[3] == "1" and line[4] == "1", line   # synchronize_params: broadcast keeps, all-reduce(mean over "2") halves
    assert int(line[5]) == 1, line            # BHG_CG_STREAM picked while the exchange was in flight
    assert float(line[6]) <= 1e-5, line       # ... with the idle-device result
    assert float(line[7]) <= 1e-5, line       # forced resident kernel beside a collective: right answer (or -1 = poisoned)


@pytest.mark.gpu
def test_roll_back_snapshot_and_flow_on_gpu():
    """SURVEY §8 f.4 on the MI355X: cache_states / recover_states through ONE flat HBM snapshot written / restored by
    the multi-tensor kernels (bhg_flatten / bhg_scatter), Adam state included, and the roll-back step flow
    (problem.py:379-381,417-436; implicit_problem.py:67-78)."""
    dev = torch.device("cuda:0")
    engine, outer, inner = _scenario(Config(type="darts", unroll_steps=5), dev)
    inner.optimizer = torch.optim.Adam(inner.module.parameters(), lr=0.05)
    w0 = inner.module.w.data.clone()
    inner.cache_states()
    inner.training_step(inner.get_batch()).backward()
    inner.optimizer.step()
    assert not torch.equal(inner.module.w.data, w0) and len(inner.optimizer.state[inner.module.w]) > 0
    inner.recover_states()
    assert torch.equal(inner.module.w.data, w0)
    assert len(inner.optimizer.state[inner.module.w]) == 0
    inner.zero_grad()
    inner.training_step(inner.get_batch()).backward()
    inner.optimizer.step()
    st = inner.optimizer.state[inner.module.w]
    m0, v0, w1 = st["exp_avg"].clone(), st["exp_avg_sq"].clone(), inner.module.w.data.clone()
    inner.cache_states()
    for _ in range(3):
        inner.zero_grad()
        inner.training_step(inner.get_batch()).backward()
        inner.optimizer.step()
    assert not torch.equal(st["exp_avg"], m0)
    inner.recover_states()
    st = inner.optimizer.state[inner.module.w]
    assert torch.equal(st["exp_avg"], m0) and torch.equal(st["exp_avg_sq"], v0) and torch.equal(inner.module.w.data, w1)
    assert st["exp_avg"].is_cuda

    # flow with plain SGD: w_end = w_start - lr * grad(w_start)
    engine, outer, inner = _scenario(Config(type="darts", unroll_steps=5), dev)
    engine.config.roll_back = True
    engine._parse_dependency()
    assert inner._roll_back and not outer._roll_back
    w_start = torch.zeros(20, device=dev)
    x, y = (t.to(dev) for t in inner.train_data_loader[0])
    lam = outer.module.w.data.clone()
    g = x.t() @ (torch.sigmoid(x @ w_start) - y) / x.shape[0] + lam * w_start
    engine.config.train_iters = 5
    engine.run()
    assert inner.count == 5 and outer.count == 1
    torch.testing.assert_close(inner.module.w.data, w_start - 0.1 * g, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_fp16_precision_uses_the_dynamic_loss_scaler_on_gpu():
    """Config(precision="fp16"): autocast + GradScaler built from initial_dynamic_scale / scale_factor
    (problem.py:165-174,508-509; implicit_problem.py:45-57): the loss handed to backward is scaled, the optimizer
    step un-scales, and the scale grows by scale_factor after the growth interval."""
    cfg = Config(type="darts", unroll_steps=10, precision="fp16", initial_dynamic_scale=1024.0, scale_factor=4.0)
    engine, outer, inner = _scenario(cfg, torch.device("cuda:0"))
    engine.config.train_iters = 40
    batch = inner.get_batch()
    raw = inner.training_step_exec(batch)
    scaled = inner.get_loss(batch)
    assert inner.scaler is not None and outer.scaler is None   # only the fp16 problem gets one
    assert float(inner.scaler.get_scale()) == 1024.0
    torch.testing.assert_close(scaled.float(), raw.float() * 1024.0, rtol=1e-3, atol=0.0)
    inner.scaler.set_growth_interval(8)
    engine.run()
    assert inner.count == 40
    assert float(inner.scaler.get_scale()) >= 4096.0            # grew by scale_factor = 4 at least once
    w = inner.module.w.detach()
    assert torch.isfinite(w).all() and w.abs().max() > 1e-3     # the un-scaled updates moved the weights sanely
