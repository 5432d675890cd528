"""N > 1 path on CPU: world size 2, gloo.  The path shards by replica: each rank
solves on its own batch with no collective inside the loop; the `sync=True` hop ends in
`torch.autograd.backward`, so DistributedDataParallel's reducer all-reduces (mean) the upper-level
hypergradient.  Checked: after the synced call `prev.grad` equals the mean over ranks of the
un-synced local results, and the call returns None — for cg, neumann and darts.

Kernels are replaced by the C oracle through the test-only checker backend (no GPU here); the
GPU suite covers the kernels themselves."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case_name, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zoo
        from _cpu_checker_backend import CpuCheckerBackend
        from conftest import load_golden
        from torch.nn.parallel import DistributedDataParallel as DDP

        from betty_amd import Config
        from betty_amd import hypergradient as hg
        from betty_amd.backend import use_backend

        case = zoo.CASE_BY_NAME[case_name]
        inputs, _ = load_golden(case.family)
        inputs = dict(inputs)
        # different data per rank: each rank keeps its half of the batch
        half = inputs["batch_x"].shape[0] // world
        inputs["batch_x"] = inputs["batch_x"][rank * half : (rank + 1) * half]
        inputs["batch_y"] = inputs["batch_y"][rank * half : (rank + 1) * half]

        with use_backend(CpuCheckerBackend()):
            # local, un-synced result
            curr, prev, vector = zoo.build_case(case, inputs, Config)
            local = hg.jvp_fn_mapping[case.algo](vector, curr, prev, False)
            local = torch.cat([t.reshape(-1) for t in local]).detach()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            want = torch.stack(gathered).mean(0)

            # synced call through DDP (the wrapper the reference uses, problem.py:220-224)
            curr, prev, vector = zoo.build_case(case, inputs, Config)
            prev.fwd = DDP(prev.module, gradient_as_bucket_view=True, find_unused_parameters=True)
            ret = hg.jvp_fn_mapping[case.algo](vector, curr, prev, True)
            got = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()]).detach()
        err = float((got - want).abs().max() / want.abs().max())
        differs = float((local - want).abs().max() / want.abs().max())
        q.put((rank, ret is None, err, differs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case_name", ["reweight_cg20", "reweight_neumann10", "reweight_darts"])
def test_sync_hop_averages_over_ranks(case_name):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    results = sorted(q.get(timeout=5) for _ in range(world))
    for rank, returned_none, err, differs in results:
        assert returned_none
        # DDP's mean of per-rank results; 1e-6 = fp32 all-reduce rounding (darts: finite differences)
        assert err < (2e-4 if "darts" in case_name else 2e-6), (rank, err)
        assert differs > 1e-3, "ranks must see different data for the test to mean anything"


def _roberta_ddp_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import zoo
        from _cpu_checker_backend import CpuCheckerBackend
        from torch.nn.parallel import DistributedDataParallel as DDP

        from betty_amd import Config
        from betty_amd import hypergradient as hg
        from betty_amd.backend import use_backend

        def build():
            torch.manual_seed(3)    # the same weights on every rank (what DDP's constructor would broadcast)
            inner = zoo.RobertaInner(layers=2, hidden=64, heads=4, ffn=128, vocab=1000, positions=66)
            upper = zoo.MWN(50)
            g = torch.Generator().manual_seed(100 + rank)   # a different token batch per rank
            B, S = 6, 24
            batch = (torch.randint(3, 999, (B, S), generator=g), torch.ones(B, S, dtype=torch.long),
                     torch.zeros(B, S, dtype=torch.long), torch.randint(0, 2, (B,), generator=g))
            gv = torch.Generator().manual_seed(9)
            vector = [0.05 * torch.randn(p.shape, generator=gv) for p in inner.parameters()]
            prev = zoo.StubProblem("upper", upper, config=Config())
            # radius 1.0: the +/- perturbations differ by enough that the fp32 difference of the two gradients keeps 4 digits
            curr = zoo.StubProblem("inner", inner, config=Config(type="darts", darts_alpha=1.0),
                                   loss_fn=zoo.make_roberta_reweight_loss(prev), batch=batch)
            return curr, prev, vector

        with use_backend(CpuCheckerBackend()):
            curr, prev, vector = build()
            local = hg.jvp_fn_mapping["darts"](vector, curr, prev, False)
            local = torch.cat([t.reshape(-1) for t in local]).detach()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            want = torch.stack(gathered).mean(0)
            curr, prev, vector = build()
            prev.fwd = DDP(prev.module, gradient_as_bucket_view=True, find_unused_parameters=True)   # problem.py:220-224
            ret = hg.jvp_fn_mapping["darts"](vector, curr, prev, True)
            got = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()]).detach()
        err = float((got - want).abs().max() / want.abs().max())
        differs = float((local - want).abs().max() / want.abs().max())
        q.put((rank, ret is None, err, differs, len(vector)))
    finally:
        dist.destroy_process_group()


def test_cfg4_roberta_darts_sync_hop_under_ddp():
    """BASELINE cfg 4's shape of problem at CPU size: transformers' RobertaForSequenceClassification (2 layers, width 64 —
    the class the example uses, reduced in depth and width), reweighting net upper, DARTS finite difference, DDP over 2
    ranks with different token batches: the synced hop leaves the MEAN of the per-rank hypergradients in prev.grad
    (darts.py:44-46,52-53 through DistributedDataParallel's reducer) and returns None."""
    pytest.importorskip("transformers")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_roberta_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, returned_none, err, differs, T in sorted(q.get(timeout=5) for _ in range(world)):
        assert returned_none and T > 30
        assert err < 2e-4, (rank, err)       # finite differences + fp32 all-reduce rounding (as reweight_darts above)
        assert differs > 1e-3, "ranks must see different data for the test to mean anything"


def _sync_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _cpu_checker_backend import CpuCheckerBackend

        from betty_amd import Config
        from betty_amd.backend import use_backend
        from betty_amd.problems import ImplicitProblem

        torch.manual_seed(100 + rank)
        module = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))  # 4 tensors, ragged sizes
        prob = ImplicitProblem(name="p", module=module, config=Config())
        prob._world_size = world
        mine = torch.cat([p.data.reshape(-1) for p in module.parameters()]).clone()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        with use_backend(CpuCheckerBackend()):
            prob.synchronize_params(prob.trainable_parameters(), all_reduce=True)  # problem.py:607-609
            avg = torch.cat([p.data.reshape(-1) for p in module.parameters()]).clone()
            with torch.no_grad():
                for p in module.parameters():
                    p.add_(float(rank))  # diverge again
            prob.synchronize_params(prob.trainable_parameters())  # broadcast from rank 0 (problem.py:605-606)
            bcast = torch.cat([p.data.reshape(-1) for p in module.parameters()]).clone()
        want_avg = torch.stack(gathered).mean(0)
        q.put((rank, float((avg - want_avg).abs().max()), float((bcast - want_avg).abs().max())))
    finally:
        dist.destroy_process_group()


def test_synchronize_params_is_one_flat_collective():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err_avg, err_bcast in sorted(q.get(timeout=5) for _ in range(world)):
        assert err_avg < 1e-6  # all ranks hold the mean
        assert err_bcast < 1e-6  # rank 0 added 0.0, so after the broadcast everyone holds the mean again


def _exchange_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from _cpu_checker_backend import CpuCheckerBackend

        from betty_amd.backend import use_backend
        from betty_amd.distributed import exchange_async

        g = torch.Generator().manual_seed(10 + rank)
        grads = [torch.randn(s, generator=g) for s in ((7, 5), (5,), (4097,), (1,))]
        flat_mine = torch.cat([t.reshape(-1) for t in grads])
        gathered = [torch.zeros_like(flat_mine) for _ in range(world)]
        dist.all_gather(gathered, flat_mine)
        with use_backend(CpuCheckerBackend()):
            handle = exchange_async(grads)
            out = handle.wait()
        got = torch.cat([t.reshape(-1) for t in out])
        want = torch.stack(gathered).mean(0)
        shapes_ok = all(a.shape == b.shape for a, b in zip(out, grads))
        q.put((rank, float((got - want).abs().max()), shapes_ok))
    finally:
        dist.destroy_process_group()


def test_flat_async_exchange_of_a_hypergradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, shapes_ok in sorted(q.get(timeout=5) for _ in range(world)):
        assert shapes_ok and err < 1e-6


def _engine_worker(rank, world, port, q, flat=False, compare=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import test_engine_shim as tes
        from _cpu_checker_backend import CpuCheckerBackend

        from betty_amd import Config
        from betty_amd.backend import use_backend
        from betty_amd.engine import Engine, EngineConfig

        # same model init on every rank (seeded inside _scenario), different data order per rank
        engine, outer, inner = tes._scenario(Config(type="cg", cg_iterations=3, cg_alpha=0.1, unroll_steps=20),
                                             torch.device("cpu"))
        x, y = inner.train_data_loader[0]
        half = x.shape[0] // world
        inner.train_data_loader = [(x[rank * half:(rank + 1) * half], y[rank * half:(rank + 1) * half])]
        with use_backend(CpuCheckerBackend()):
            engine = Engine(config=EngineConfig(train_iters=100, strategy="distributed", backend="gloo",
                                                flat_exchange_min_params=1 if flat else 0),
                            problems=[outer, inner], dependencies={"u2l": {outer: [inner]}, "l2u": {inner: [outer]}},
                            device=torch.device("cpu"))
            # the scenario's Inner.training_step calls self.module directly (like the reference's test), so under DDP
            # its own gradients are never reduced; keep that in the flat arm: only the UPPER problem exchanges
            inner._flat_exchange = False
            if compare:   # exact A/B of the UPPER exchange only: no reducer on the inner module in either arm (under DDP a
                inner.fwd = inner.module   # forward through the wrapper arms the reducer for the next direct-call step)
            engine.run()
        lam = outer.module.w.detach().clone()
        gathered = [torch.zeros_like(lam) for _ in range(world)]
        dist.all_gather(gathered, lam)
        assert outer._flat_exchange == flat
        q.put((rank, outer.count, float((gathered[0] - gathered[1]).abs().max()), float((lam - 1.0).abs().max()), lam.numpy()))
    finally:
        dist.destroy_process_group()


def _run_engine(world, flat, compare=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, q, flat, compare)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    return sorted((q.get(timeout=5) for _ in range(world)), key=lambda t: t[0])


def test_engine_distributed_strategy_keeps_upper_parameters_in_sync():
    """Engine(strategy="distributed"): the upper module is DDP-wrapped, the synced hypergradient hop
    averages over ranks, so after 5 upper steps on different data every rank holds the same lambda."""
    for rank, upper_steps, diff, moved, _lam in _run_engine(2, flat=False):
        assert upper_steps == 5
        assert diff < 1e-6, "ranks diverged: the hypergradient was not averaged"
        assert moved > 1e-3, "lambda did not move at all"


def test_engine_flat_async_exchange_equals_ddp():
    """EngineConfig.flat_exchange_min_params: direct gradient and per-path hypergradients averaged by flat asynchronous
    all-reduces (Problem.backward -> exchange_async, waited for before the optimizer step) instead of the DDP reducer:
    same parameters after 5 upper steps, on every rank."""
    ddp = _run_engine(2, flat=False, compare=True)
    flat = _run_engine(2, flat=True, compare=True)
    for (rank, steps, diff, moved, lam_f), (_, _, _, _, lam_d) in zip(flat, ddp):
        assert steps == 5 and diff < 1e-6 and moved > 1e-3
        np.testing.assert_allclose(lam_f, lam_d, rtol=2e-5, atol=1e-7)


def _fsdp_darts_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hypergrad_oracle as horc
        import zoo
        from _cpu_checker_backend import CpuCheckerBackend
        from conftest import load_golden

        from betty_amd import Config
        from betty_amd import hypergradient as hg
        from betty_amd.backend import use_backend

        case = zoo.CASE_BY_NAME["reweight_darts"]
        inputs, _ = load_golden(case.family)

        def build(cfg_cls):
            curr, prev, vector = zoo.build_case(case, inputs, cfg_cls)
            curr._strategy = "fsdp"
            # each rank holds a different "shard" of the direction: scale it per rank so the norms differ
            vector = [(rank + 1.0) * v for v in vector]
            return curr, prev, vector

        # what eps must be: R / sqrt(sum over ranks of the local squared norms)
        _, _, vec = build(Config)
        local_sq = sum(float((v.double() ** 2).sum()) for v in vec)
        all_sq = torch.tensor([local_sq], dtype=torch.float64)
        dist.all_reduce(all_sq)

        curr, prev, vector = build(Config)
        want = horc.darts(vector, curr, prev, False)            # oracle restatement of darts.py incl. 31-34
        with use_backend(CpuCheckerBackend()):
            curr, prev, vector = build(Config)
            w0 = [p.data.clone() for p in curr.trainable_parameters()]
            got = hg.darts(vector, curr, prev, False)
            drift = max(float((p.data - w).abs().max()) for p, w in zip(curr.trainable_parameters(), w0))
        err = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(got, want))
        ref_err = -1.0
        if os.path.isdir("/root/reference/betty"):              # the live reference, same two ranks
            sys.path.insert(0, "/root/reference")
            import betty.hypergradient  # noqa: F401
            from betty.configs import Config as RefConfig

            ref_darts = sys.modules["betty.hypergradient.darts"].darts
            curr, prev, vector = build(RefConfig)
            ref = ref_darts(vector, curr, prev, False)
            ref_err = max(float((a - b).abs().max()) for a, b in zip(want, ref))   # oracle == reference, bitwise
        q.put((rank, err, drift, ref_err, local_sq, float(all_sq)))
    finally:
        dist.destroy_process_group()


def test_darts_fsdp_branch_uses_the_norm_of_the_whole_sharded_vector():
    """darts.py:31-34: under FSDP every rank holds a shard of the direction, eps = R / ||whole vector||.
    Two gloo ranks with different shards: the product (checker backend) matches the oracle's restatement, the
    oracle matches the live reference bit for bit (when the checkout is present), weights are restored."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fsdp_darts_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    results = sorted(q.get(timeout=5) for _ in range(world))
    assert results[0][4] != results[1][4], "ranks must hold different shards"
    for rank, err, drift, ref_err, local_sq, all_sq in results:
        assert all_sq > local_sq
        assert err < 2e-3, (rank, err)          # finite differences: the darts tolerance of the golden cases
        assert drift < 2e-7, (rank, drift)
        assert ref_err in (-1.0, 0.0), (rank, ref_err)


# ------------------------------------------------------------------------------------------------
# global-HVP mode (betty_amd/global_hvp.py): ONE inner problem, batch spread over the ranks,
# sharded CG state.  Oracle = the reference algorithm in ONE process on the concatenated batch.
# ------------------------------------------------------------------------------------------------
def _global_worker(rank, world, port, case_name, structured, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hypergrad_oracle as orc
        import zoo
        from _cpu_checker_backend import CpuCheckerBackend
        from conftest import load_golden

        from betty_amd import Config
        from betty_amd.backend import use_backend
        import betty_amd.global_hvp as gh
        from betty_amd.global_hvp import cg_global

        gh.GLOBAL_FORM = {"onepass": "one_pass", "sharded": "sharded", "factor": "auto"}.get(structured, "auto")
        case = zoo.CASE_BY_NAME[case_name]
        inputs, _ = load_golden(case.family)
        full_inputs = dict(inputs)
        n = (inputs["batch_x"].shape[0] // world) * world      # equal shares
        full_inputs["batch_x"], full_inputs["batch_y"] = inputs["batch_x"][:n], inputs["batch_y"][:n]
        share = n // world
        mine = dict(full_inputs)
        mine["batch_x"] = full_inputs["batch_x"][rank * share:(rank + 1) * share]
        mine["batch_y"] = full_inputs["batch_y"][rank * share:(rank + 1) * share]

        # single-process oracle on the concatenated batch (every rank computes it: no communication involved)
        curr, prev, vector = zoo.build_case(case, full_inputs, Config)
        want = torch.cat([t.reshape(-1) for t in orc.cg(vector, curr, prev, False)]).detach()

        with use_backend(CpuCheckerBackend()):
            curr, prev, vector = zoo.build_case(case, mine, Config)
            if structured:
                zoo.attach_mlp_structure(curr, case.family, impl="torch", fused=structured in ("onepass", "factor"))
            got = cg_global(vector, curr, prev, False)
            got = torch.cat([t.reshape(-1) for t in got]).detach()
            # sync=True: lands in .grad through backward (no DDP wrapper here: the local share of the mean)
            curr2, prev2, vector2 = zoo.build_case(case, mine, Config)
            if structured:
                zoo.attach_mlp_structure(curr2, case.family, impl="torch", fused=structured in ("onepass", "factor"))
            ret = cg_global(vector2, curr2, prev2, True)
            local = torch.cat([p.grad.reshape(-1) for p in prev2.trainable_parameters()]).detach()
            gathered = [torch.zeros_like(local) for _ in range(world)]
            dist.all_gather(gathered, local)
            synced = torch.stack(gathered).mean(0)
        rel = float((got - want).norm() / want.norm())
        rel_sync = float((synced - want).norm() / want.norm())
        # every rank must hold the same global answer
        others = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(others, got)
        same = all(torch.equal(o, got) for o in others)
        from betty_amd.global_hvp import ONE_PASS_STATS
        K = int(curr.config.cg_iterations)
        stats = (ONE_PASS_STATS["solves"], ONE_PASS_STATS["scalar_all_reduces"], ONE_PASS_STATS["residual_all_reduces"], K)
        if structured == "factor":
            fx = gh.FX_STATS
            stats = (fx["solves"], fx["const_gathers"], fx["slab_gathers"], fx["scal_gathers"], fx["rhs_all_reduces"], ONE_PASS_STATS["solves"], K)
        q.put((rank, rel, rel_sync, ret is None, same, stats))
    finally:
        dist.destroy_process_group()


def _neumann_global_worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import hypergrad_oracle as orc
        import zoo
        from _cpu_checker_backend import CpuCheckerBackend
        from conftest import load_golden

        import betty_amd.global_hvp as gh
        from betty_amd import Config
        from betty_amd.backend import use_backend

        case = zoo.CASE_BY_NAME["reweight_neumann10"]
        inputs, _ = load_golden(case.family)
        full_inputs = dict(inputs)
        n = (inputs["batch_x"].shape[0] // world) * world
        full_inputs["batch_x"], full_inputs["batch_y"] = inputs["batch_x"][:n], inputs["batch_y"][:n]
        share = n // world
        mine = dict(full_inputs)
        mine["batch_x"] = full_inputs["batch_x"][rank * share:(rank + 1) * share]
        mine["batch_y"] = full_inputs["batch_y"][rank * share:(rank + 1) * share]
        curr, prev, vector = zoo.build_case(case, full_inputs, Config)
        want = torch.cat([t.reshape(-1) for t in orc.neumann(vector, curr, prev, False)]).detach()
        with use_backend(CpuCheckerBackend()):
            curr, prev, vector = zoo.build_case(case, mine, Config)
            zoo.attach_mlp_structure(curr, case.family, impl="torch", fused=True)
            got = torch.cat([t.reshape(-1) for t in gh.neumann_global(vector, curr, prev, False)]).detach()
        rel = float((got - want).norm() / want.norm())
        others = [torch.zeros_like(got) for _ in range(world)]
        dist.all_gather(others, got)
        fx = gh.FX_STATS
        q.put((rank, rel, all(torch.equal(o, got) for o in others),
               (fx["solves"], fx["const_gathers"], fx["slab_gathers"], fx["scal_gathers"], int(curr.config.neumann_iterations))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_neumann_global_exchanges_one_factor_slab_per_iteration_and_matches_the_oracle(world):
    """Config(type="neumann_global") over real gloo collectives (the phases' math in ATen): against the reference's neumann run in ONE
    process on the concatenated batch; per solve one gather of the constants, K gathers of the factors, NO scalar exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_neumann_global_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, same, (solves, n_const, n_slab, n_scal, K) in sorted(q.get(timeout=5) for _ in range(world)):
        assert rel <= 1e-4, (rank, rel)
        assert same
        assert (solves, n_const, n_slab, n_scal) == (1, 1, K, 0)


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("case_name,structured", [("reweight_cg20", False), ("reweight_cg20", "sharded"), ("reweight_cg20", "onepass"),
                                                  ("reweight_cg20", "factor"), ("logreg_cg5", False), ("logreg_cg3_a01", False)])
def test_global_hvp_cg_matches_single_process_oracle(world, case_name, structured):
    """Sharded x, r, p; reduce-scatter of the data-parallel HVPs; partial-sum all-reduces between the CG phases;
    all-gather of the direction: the result equals the reference's cg on the concatenated batch.
    "onepass": the replicated-state form (bhg_mlp_cg_global_phase's protocol, its math in ATen here): two collectives per
    iteration — 8 bytes before the step length, the residual after the outputs — and only the 8 bytes in the last one."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_global_worker, args=(r, world, port, case_name, structured, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    for rank, rel, rel_sync, returned_none, same, stats in sorted(q.get(timeout=5) for _ in range(world)):
        assert rel <= 1e-4, (rank, rel)            # north_star tolerance vs the single-process reference algorithm
        assert rel_sync <= 1e-4, (rank, rel_sync)  # sync=True: mean over ranks of what landed in .grad
        assert returned_none and same
        if structured == "factor":
            # round 6, the factor-exchange form (csrc/mlp/fx.inc's protocol, its math in ATen here): per solve ONE gather of the constants,
            # K gathers of the batch-sized factors, K gathers of three fp64 shares, the right-hand side's mean — nothing else; two solves ran
            solves, n_const, n_slab, n_scal, n_rhs, one_pass, K = stats
            assert (solves, n_const, n_slab, n_scal, n_rhs, one_pass) == (2, 2, 2 * K, 2 * K, 2, 0), stats
            continue
        solves, n_scalar, n_resid, K = stats
        if structured == "onepass":                # two solves ran (sync False / True)
            assert (solves, n_scalar, n_resid) == (2, 2 * K, 2 * (K - 1)), stats
        else:
            assert solves == 0, stats
