#!/usr/bin/env python
"""Falsifiable N > 1 check: run on whatever GPUs are visible, exit 0 with `SELFCHECK OK ...` or non-zero with
ONE line `SELFCHECK FAIL: <reason>`.

    python scripts/multi_gpu_selfcheck.py                    # N = torch.cuda.device_count() ranks, one per GPU, RCCL (backend nccl)
    python scripts/multi_gpu_selfcheck.py --gpus 2 --backend gloo --all-on-gpu0     # two ranks sharing GPU 0 (a 1-GPU box)

Every rank checks, against what the same algorithm gives in ONE process (the expectations the gloo tests of tests/test_gpu_global.py /
tests/test_distributed_cpu.py pin):
  census     the process group spans N ranks (all-reduce of ones) on N distinct devices (PCI bus ids / uuids)
  replica    the reference's DDP mode (problem.py:220-224, cg.py:58-63): every rank solves ITS problem; .grad after the sync=True hop
             = the mean over ranks of the local hypergradients — closed-form upper net with the deferred all-reduce
             (SigmoidMLPWeightNet(average_over=True, overlap=True)) and, second, the DDP wrapper with nothing declared
  global     cg_global, factor-exchange form, one-pass fused form and sharded form (opaque HVP): equal on every rank, equal to the one-rank solve of the
             concatenated batch
  exchange   betty_amd.distributed.exchange_async: one flat all-reduce = the mean
and reports the measured latency of the M-float all-reduce.  Nothing here is a throughput claim."""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


class Fail(Exception):
    pass


def worker(args):
    import torch
    import torch.distributed as dist

    import bench
    import zoo
    from betty_amd import Config
    from betty_amd import distributed as bd
    from betty_amd import hypergradient as hg
    from betty_amd.global_hvp import cg_global
    from betty_amd.hypergradient.structured import SigmoidMLPWeightNet, WeightedCEMLP

    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    dev_index = 0 if args.all_on_gpu0 else local
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(args.backend)
    verdict, reason = 1.0, ""
    try:
        census = bench.device_census(dist, world, rank, device, 301)
        if census["ranks_seen"] != world:
            raise Fail(f"the process group spans {census['ranks_seen']} ranks, expected {world}")
        if not args.all_on_gpu0 and world > 1 and census["distinct_devices"] != world:
            raise Fail(f"{world} ranks on {census['distinct_devices']} distinct devices: {census['devices']}")

        def rel(a, b):
            a = torch.cat([t.reshape(-1).double() for t in a])
            b = torch.cat([t.reshape(-1).double() for t in b])
            return float((a - b).norm() / b.norm().clamp_min(1e-300))

        def gathered_mean(flat):
            every = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(every, flat)
            return sum(every) / world

        # ---- replica mode -------------------------------------------------------------------------------------------------------------
        case = zoo.CASE_BY_NAME["reweight_cg20"]
        inputs = zoo.seed_family_inputs(case.family, seed=rank)
        shared = zoo.seed_family_inputs(case.family, seed=0)
        for k in inputs:
            if not k.startswith(("batch", "vec")):
                inputs[k] = shared[k]
        curr, prev, vector = zoo.build_case(case, inputs, Config, device=str(device))
        zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True)
        local_h = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)])
        want = gathered_mean(local_h)
        for mode in ("overlap", "ddp"):
            curr, prev, vector = zoo.build_case(case, inputs, Config, device=str(device))
            if mode == "ddp":
                from torch.nn.parallel import DistributedDataParallel as DDP

                prev.fwd = DDP(prev.module, device_ids=[dev_index], gradient_as_bucket_view=True, find_unused_parameters=True)
                zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True, average_over=None)
            else:
                zoo.attach_mlp_structure(curr, case.family, impl="hip", weight_net=True, average_over=True, overlap=True)
            assert hg.cg(vector, curr, prev, True) is None
            if mode == "overlap" and world > 1 and bd.pending_grad_syncs() != 1:
                raise Fail("the deferred all-reduce was not registered")
            bd.fence_grads()
            got = torch.cat([p.grad.reshape(-1) for p in prev.trainable_parameters()])
            e = rel([got], [want])
            if not e <= 1e-5:
                raise Fail(f"replica mode ({mode}): .grad is {e:.2e} from the mean over ranks of the local hypergradients")

        # ---- global-batch mode --------------------------------------------------------------------------------------------------------
        dims, B, K, ridge = [256, 384, 128, 10], 100, 6, 0.05
        g = torch.Generator().manual_seed(4242)
        inner, upper = zoo.MLP(dims), zoo.MWN(16)
        with torch.no_grad():
            for p in list(inner.parameters()) + list(upper.parameters()):
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / max(p.shape[-1], 4) ** 0.5))
        inner, upper = inner.to(device), upper.to(device)
        x = torch.randn(world * B, dims[0], generator=g).to(device)
        y = torch.randint(0, dims[-1], (world * B,), generator=g).to(device)
        vecs = [[0.1 * torch.randn(p.shape, generator=g).to(device) for p in inner.parameters()] for _ in range(world)]
        vmean = [sum(v[i] for v in vecs) / world for i in range(len(vecs[0]))]
        prevg = zoo.StubProblem("upper", upper, config=Config())

        def attach(xb, yb, structured):
            c = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=K, cg_alpha=1.0),
                                loss_fn=zoo.make_reweight_loss(prevg, ridge), batch=(xb, yb))
            if structured:
                c.hypergradient_structure = lambda prev_: WeightedCEMLP(
                    c, prev_, layers=list(inner.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="hip", fused=True)
            return c

        want_g = [t.clone() for t in hg.cg(vmean, attach(x, y, True), prevg, False)]
        sl = slice(rank * B, (rank + 1) * B)
        import betty_amd.global_hvp as gh

        for form, structured in (("factor-exchange", True), ("one-pass", True), ("sharded", False)):
            gh.GLOBAL_FORM = {"factor-exchange": "auto", "one-pass": "one_pass", "sharded": "sharded"}[form]
            gh.FX_ALWAYS_GATHER = True   # (world size 1 over RCCL: the in-place all-gathers are issued all the same)
            n_fx, n_op = gh.FX_STATS["solves"], gh.ONE_PASS_STATS["solves"]
            got = cg_global(vecs[rank], attach(x[sl], y[sl], structured), prevg, False)
            took = (gh.FX_STATS["solves"] - n_fx, gh.ONE_PASS_STATS["solves"] - n_op)
            if took != {"factor-exchange": (1, 0), "one-pass": (0, 1), "sharded": (0, 0)}[form]:
                raise Fail(f"global mode ({form}): another form ran (factor-exchange, one-pass solves = {took})")
            e = rel(got, want_g)
            if not e <= 1e-4:
                raise Fail(f"global mode ({form}): {e:.2e} from the one-rank solve of the concatenated batch")
            flat = torch.cat([t.reshape(-1) for t in got])
            every = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(every, flat)
            if not all(torch.equal(o, flat) for o in every):
                raise Fail(f"global mode ({form}): the ranks returned different bits")

        # the Neumann series on the global batch (factor-exchange form: one gather per iteration, no reduction)
        from betty_amd.global_hvp import neumann_global

        def attach_n(xb, yb):
            c = zoo.StubProblem("inner", inner, config=Config(type="neumann", neumann_iterations=K, neumann_alpha=0.1),
                                loss_fn=zoo.make_reweight_loss(prevg, ridge), batch=(xb, yb))
            c.hypergradient_structure = lambda prev_: WeightedCEMLP(
                c, prev_, layers=list(inner.layers), weight_fn=lambda ce: prev_.fwd(ce.reshape(-1, 1)), ridge=ridge, impl="hip", fused=True)
            return c

        want_n = [t.clone() for t in hg.neumann(vmean, attach_n(x, y), prevg, False)]
        got = neumann_global(vecs[rank], attach_n(x[sl], y[sl]), prevg, False)
        e = rel(got, want_n)
        if not e <= 1e-4:
            raise Fail(f"global mode (neumann, factor-exchange): {e:.2e} from the one-rank solve of the concatenated batch")
        flat = torch.cat([t.reshape(-1) for t in got])
        every = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(every, flat)
        if not all(torch.equal(o, flat) for o in every):
            raise Fail("global mode (neumann, factor-exchange): the ranks returned different bits")

        # ---- flat asynchronous exchange -----------------------------------------------------------------------------------------------
        gl = torch.Generator().manual_seed(1000 + rank)
        mine = [torch.randn(257, 33, generator=gl).to(device), torch.randn(1001, generator=gl).to(device)]
        avg = bd.exchange_async(mine).wait()
        e = rel([torch.cat([t.reshape(-1) for t in avg])], [gathered_mean(torch.cat([t.reshape(-1) for t in mine]))])
        if not e <= 1e-6:
            raise Fail(f"exchange_async: {e:.2e} from the mean")
        from betty_amd.backend import get_backend

        get_backend().check_health()
        lat = census["allreduce_M_floats_us"]
        if rank == 0:
            print(f"SELFCHECK OK: world {world} over {args.backend}, devices " +
                  ", ".join(f"{d['name']}@{d['pci']}" for d in census["devices"]) +
                  (f"; all-reduce of 301 floats {lat:.1f} us" if lat else "") + "; replica (deferred all-reduce, DDP wrapper), global "
                  "(cg: factor-exchange, one-pass, sharded; neumann: factor-exchange), flat exchange: all equal to the one-process expectations", flush=True)
        verdict = 0.0
    except Fail as f:
        reason = str(f)
    except Exception as exc:   # noqa: BLE001 - anything else is a failure with its one-line reason too
        reason = f"{type(exc).__name__}: {exc}".splitlines()[0]
    flag = torch.tensor([verdict], device=device if args.backend == "nccl" else "cpu")
    try:
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    except Exception:   # noqa: BLE001
        pass
    if reason:
        print(f"SELFCHECK FAIL: rank {rank}: {reason}", flush=True)
    dist.destroy_process_group()
    return 1 if (reason or flag.item() > 0) else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="ranks (default: every visible GPU)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL; default) | gloo")
    ap.add_argument("--all-on-gpu0", action="store_true", help="debug: every rank on cuda:0 (use with --backend gloo on a 1-GPU box)")
    args = ap.parse_args()
    if "WORLD_SIZE" in os.environ:
        sys.exit(worker(args))
    import torch

    if not torch.cuda.is_available():
        print("SELFCHECK FAIL: no GPU visible")
        sys.exit(2)
    n = args.gpus or torch.cuda.device_count()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


if __name__ == "__main__":
    main()
