"""Round-3 VERDICT item 8, protocol level: a global-batch fully projected CG solve whose ranks exchange BATCH-SIZED FACTORS
(one all-gather + one scalar all-reduce per iteration) instead of the N-sized residual.  tests/proj_global_ref.py restates the
protocol in ATen; here it is held against the reference's algorithm (cg.py:34-56) run in ONE process on the concatenated batch —
emulated ranks (threads) and real gloo ranks, world size 2 / 4.  CPU only; the HIP kernels of this form are future work."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import proj_global_ref as P  # noqa: E402


def _errs(out, ref, world):
    rz, xc, xV, _ = ref
    errs = []
    for g in range(world):
        errs.append(float((out[g][0] - rz[g]).norm() / rz[g].norm()))
        errs += [float((a - b).norm() / b.norm()) for a, b in zip(out[g][1], xc)]
        errs.append(float((out[g][2] - xV).norm() / xV.norm()))
    return max(errs)


@pytest.mark.parametrize("dims,B,world,K,shift", [([24, 32, 16, 12, 5], 24, 1, 6, 0.3), ([24, 32, 16, 12, 5], 24, 2, 6, 0.3),
                                                  ([24, 32, 16, 12, 5], 24, 4, 8, 0.05), ([20, 16, 12, 4], 18, 3, 5, 0.0),
                                                  ([20, 16, 14, 12, 10, 6], 20, 2, 7, 0.2)])
def test_factor_exchange_equals_single_process_cg_fp64(dims, B, world, K, shift):
    out, ref, comms = P.run_emulated(dims, B, world, K, 0.8, shift, seed=1)
    assert _errs(out, ref, world) <= 1e-10
    L = len(dims) - 1
    for c in comms:   # the communication pattern: 2 L gathers of constants per solve, then ONE gather + ONE scalar reduce per iteration
        assert (c.gathers, c.scalar_reduces) == (2 * L + K, K)
    # every rank ends with the same replicated narrow slices
    for g in range(1, world):
        assert all(torch.equal(a, b) for a, b in zip(out[g][1], out[0][1])) and torch.equal(out[g][2], out[0][2])


def test_factor_exchange_fp32_within_the_north_star_tolerance():
    """fp32 state and products (fp64 only in the inner products, as the kernels do) on a well-conditioned instance: rtol 1e-4
    against the fp64 single-process answer, and no worse than a few times what the reference's algorithm itself loses in fp32
    (with shift 0.3 this very instance is ill-conditioned: plain fp32 CG and the protocol are both 0.6 away after 20 iterations)."""
    dims, B, world, K, shift = [64, 96, 48, 32, 10], 40, 4, 20, 1.0
    out, _, _ = P.run_emulated(dims, B, world, K, 1.0, shift, seed=3, dtype=torch.float32)
    share = B // world
    parts = lambda t: [t[g * share:(g + 1) * share] for g in range(world)]
    Ws, bs, x, y, w, vec = P.make_problem(dims, B, 3, torch.float64)
    ref = P.oracle(Ws, bs, parts(x), parts(y), parts(w), vec, K, 1.0, shift)
    Ws32, bs32, x32, _, w32, vec32 = P.make_problem(dims, B, 3, torch.float32)
    ref32 = P.oracle(Ws32, bs32, parts(x32), parts(y), parts(w32), vec32, K, 1.0, shift)
    up = lambda o: (o[0].double(), [t.double() for t in o[1]], o[2].double())
    err = _errs([up(o) for o in out], ref, world)
    err_plain = _errs([([r.double() for r in ref32[0]][g], [t.double() for t in ref32[1]], ref32[2].double()) for g in range(world)], ref, world)
    assert err <= 1e-4 and err <= 4.0 * err_plain + 1e-6, (err, err_plain)


def test_bytes_per_iteration_at_the_benchmark_shapes():
    """What DESIGN 4b quotes: per rank and iteration the factors are B x (sum of Rd widths + sum of Rh widths) floats."""
    dims, B = [3072, 2048, 1536, 384, 10], 100
    floats = B * (sum(dims[1:]) + sum(dims[1:-1]))
    N = sum(dims[l] * dims[l + 1] + dims[l + 1] for l in range(len(dims) - 1))
    assert N == 10_034_826
    assert floats * 4 == 3_178_400          # 3.2 MB gathered per rank and iteration
    assert 4 * N / (floats * 4) > 12.6      # against the 40.1 MB residual of the one-pass form


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        dims, B, K, shift = [24, 32, 16, 12, 5], 24, 6, 0.3
        Ws, bs, x, y, w, vec = P.make_problem(dims, B, 1)
        share = B // world
        parts = lambda t: [t[g * share:(g + 1) * share] for g in range(world)]
        comm = P.DistComm()
        got = P.solve(Ws, bs, parts(x)[rank], parts(y)[rank], parts(w)[rank], vec, K, 0.8, shift, comm)
        rz, xc, xV, _ = P.oracle(Ws, bs, parts(x), parts(y), parts(w), vec, K, 0.8, shift)
        err = max(float((got[0] - rz[rank]).norm() / rz[rank].norm()), float((got[2] - xV).norm() / xV.norm()),
                  max(float((a - b).norm() / b.norm()) for a, b in zip(got[1], xc)))
        q.put((rank, err, comm.gathers, comm.scalar_reduces, comm.bytes_gathered_per_rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_factor_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    res = sorted(q.get(timeout=5) for _ in range(world))
    L, K = 4, 6
    for rank, err, gathers, reduces, nbytes in res:
        assert err <= 1e-10, (rank, err)
        assert (gathers, reduces) == (2 * L + K, K)
