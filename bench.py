#!/usr/bin/env python
"""bench.py — hypergradient-steps/sec (CG K=20, 10 M inner params) on N MI355X.

One "step" = one call of the hot path ``cg(vector, curr, prev, sync=True)``: inner-loss
re-evaluation + gradient with graph, K=20 x (Hessian-vector product + fused CG recurrence), and
the final mixed-derivative VJP accumulated into the upper parameters' ``.grad`` (the path
``Problem.backward -> get_grads -> cg`` of the reference, betty/hypergradient/cg.py:8-70).

Workload = BASELINE.json configs[1]'s inner problem at the size the metric is quoted on
(BASELINE.md cfg 2 / metric): ReLU-MLP 3072-2048-1536-384-10 (N = 10,034,826 in 8 tensors) with
meta-weight-net-weighted cross-entropy (+1e-2 ||w||^2 so H is positive definite), upper problem
MWN 1-100-1 (M = 301), batch 100 x 3072, synthetic seeded data already resident in HBM.

Multi-GPU (``--gpus N`` under torch.distributed.run): the path shards by replica exactly as the
reference's DDP mode does — every rank solves on its local batch with no collective inside the CG
loop; the M-sized hypergradient is averaged over the ranks when the ``sync=True`` hop completes: by
ONE flat all-reduce of the M floats over RCCL issued by the hop itself (the upper module is declared in
closed form, SigmoidMLPWeightNet(average_over=True): csrc/bhg_mwn.hip), or — ``--upper autograd`` — by
the DDP reducer when the backward through the wrapped module fires, as in the reference.  Weak scaling:
value = N * steps / max-over-ranks time.  A run with more than one rank sets GPU_MAX_HW_QUEUES=1 for its processes unless the
variable is already set (--hw-queues): a collective's stream is a second hardware queue, and that costs every dependent launch
of the solver ~1 us on this runtime (--emulate-collective shows it at N = 1).

Timing: W warm-up steps, an untimed settle phase (--settle-ms), then --reps pairs of timed regions of
--steps steps (full K, then K / 2, interleaved), each bracketed by barrier + synchronize; the line
reports the MEDIAN full region (value, ms_per_step) and every region, and the per-iteration time as the
median of (t(K) - t(K/2)) / (K/2) over the pairs.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from betty_amd import Config  # noqa: E402
from betty_amd import hypergradient as hg  # noqa: E402

SIZES = [3072, 2048, 1536, 384, 10]
BATCH = 100
RIDGE = 1e-2
HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s float4 copy)


class InnerMLP(nn.Module):
    def __init__(self):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(SIZES[:-1], SIZES[1:])])

    def forward(self, x):
        for i, lin in enumerate(self.layers):
            x = lin(x)
            if i + 1 < len(self.layers):
                x = F.relu(x)
        return x


class MWN(nn.Module):
    def __init__(self, hidden=100):
        super().__init__()
        self.l1 = nn.Linear(1, hidden)
        self.l2 = nn.Linear(hidden, 1)

    def forward(self, x):
        return torch.sigmoid(self.l2(F.relu(self.l1(x))))


class Problem:
    """The slice of betty's ImplicitProblem the hot path touches."""

    def __init__(self, name, module, config, loss_fn=None, batch=None, forward_module=None):
        self.name, self.module, self.config = name, module, config
        self._loss_fn, self.cur_batch = loss_fn, batch
        self.paths, self._strategy = [], "default"
        self.fwd = forward_module if forward_module is not None else module

    def training_step_exec(self, batch):
        return self._loss_fn(self, batch)

    def parameters(self):
        return list(self.module.parameters())

    trainable_parameters = parameters
    meta_trainable_parameters = parameters

    def set_grads(self, params, grads):
        for p, g in zip(params, grads):
            if g is not None:
                p.grad = g if p.grad is None else p.grad + g


def declare_structure(curr, impl, fused=True, keep_solution=False, native_upper=True):
    """Opt the inner problem into the analytic MFMA HVP (betty_amd/hypergradient/structured.py)."""
    from betty_amd.hypergradient.structured import SigmoidMLPWeightNet, WeightedCEMLP

    # the meta-weight-net's own closed form (csrc/bhg_mwn.hip) unless --upper autograd; under DDP the declaration asks for the
    # data-parallel mean that the wrapper's reducer would have taken (SigmoidMLPWeightNet.average_over)
    def wnet(prev):
        if not native_upper:
            return None
        # overlap=True: the M-float all-reduce of step i is asynchronous and the compute stream is fenced behind it only at step
        # i + 1's hop (or the upper optimizer's step): it runs under the next solve's CG iterations (betty_amd/distributed.py)
        ddp = prev.fwd is not prev.module
        return SigmoidMLPWeightNet(prev.module.l1, prev.module.l2, average_over=True if ddp else None, overlap=ddp)

    curr.hypergradient_structure = lambda prev: WeightedCEMLP(
        curr, prev, layers=list(curr.module.layers), weight_fn=lambda ce: prev.fwd(ce.reshape(-1, 1)), ridge=curr.ridge, impl=impl,
        fused=fused, keep_solution=keep_solution, weight_net=wnet(prev),
    )


def make_loss(upper, ridge=RIDGE):
    def loss(self, batch):
        x, y = batch
        ce = F.cross_entropy(self.fwd(x), y, reduction="none")
        w = upper.fwd(ce.detach().reshape(-1, 1)).reshape(-1)
        out = torch.mean(w * ce)
        return out + ridge * sum((p * p).sum() for p in self.module.parameters())

    return loss


def build(device, seed, dtype=torch.float32, ddp=False, K=20, algo="cg", data_seed=None, ridge=RIDGE):
    """ridge = RIDGE is the metric workload; tests/golden/cfg2_full.npz also pins a well-conditioned
    variant of the same shapes (ridge = RIDGE_WELL, see tests/golden/make_cfg2_golden.py)."""
    torch.manual_seed(seed)
    inner = InnerMLP().to(device=device, dtype=dtype)
    mwn = MWN(100).to(device=device, dtype=dtype)
    g = torch.Generator().manual_seed(1234 + (seed if data_seed is None else data_seed))
    x = torch.randn(BATCH, SIZES[0], generator=g).to(device=device, dtype=dtype)
    y = torch.randint(0, 10, (BATCH,), generator=g)
    flip = torch.rand(BATCH, generator=g) < 0.4
    y = torch.where(flip, torch.randint(0, 10, (BATCH,), generator=g), y).to(device)
    vector = [(0.01 * torch.randn(p.shape, generator=g)).to(device=device, dtype=dtype) for p in inner.parameters()]
    fwd = mwn
    if ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP

        # the wrapper the reference uses, problem.py:220-224
        fwd = DDP(mwn, device_ids=[device.index], gradient_as_bucket_view=True, find_unused_parameters=True)
    prev = Problem("upper", mwn, Config(), forward_module=fwd)
    cfg = {
        "cg": Config(type="cg", cg_iterations=K, cg_alpha=1.0),
        "neumann": Config(type="neumann", neumann_iterations=K, neumann_alpha=0.1),
        "darts": Config(type="darts", darts_alpha=0.01),
    }[algo]
    curr = Problem("inner", inner, cfg, loss_fn=make_loss(prev, ridge), batch=(x, y))
    curr.ridge = ridge
    return curr, prev, vector


def lib_sha256():
    import hashlib

    from betty_amd import _native

    with open(_native.current_lib_path(), "rb") as f:   # the library this process actually calls (product, or libbhg_ab.so under --debug)
        return hashlib.sha256(f.read()).hexdigest()


PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")


def pmc_traffic(key, N):
    """(bytes, source) — fabric-side bytes per iteration from the committed PMC passes (profiles/rNN_pmc_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this very command, corrected per MI355X_MICROARCH.md
    §HBM; written by scripts/gpu_pmc2.sh).  PMC collection needs rocprofv3 around the process, so bench.py can only
    REPLAY a committed measurement — and does so ONLY when that file is stamped with the sha256 of the very libbhg.so
    that is loaded now (same kernels); otherwise (None, reason).  The source string says so in the JSON line."""
    sha = lib_sha256()
    for name in PMC_FILES:
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        if d.get("lib_sha256") == sha and d.get("workload_N") == N and d.get("traffic_bytes", {}).get(key) is not None:
            return d["traffic_bytes"][key], f"replayed from profiles/{name} (not measured in this run; libbhg.so sha256-matched {sha[:16]})"
    return None, "no committed PMC pass carries the sha256 of the loaded libbhg.so"


def host_info():
    model, phys = None, set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, (len(phys) or None), os.cpu_count()


def reference_cg():
    """(fn, kind, where): the reference's OWN cg (betty/hypergradient/cg.py:8-70) when its package is importable — the staged,
    git-ignored copy oracle/_ref/betty that `make -C oracle ref` takes from the checkout, or /root/reference itself — else the
    line-for-line restatement oracle/hypergrad_oracle.py (pinned bit-for-bit against it, tests/test_oracle.py)."""
    for where in (os.path.join(ROOT, "oracle", "_ref"), "/root/reference"):
        if os.path.isdir(os.path.join(where, "betty", "hypergradient")):
            try:
                if where not in sys.path:
                    sys.path.insert(0, where)
                import betty.hypergradient  # noqa: F401

                return sys.modules["betty.hypergradient.cg"].cg, "reference", where
            except Exception as exc:  # a missing optional dependency of the reference package
                print(f"bench.py: the reference at {where} does not import ({exc!r}); falling back to the restatement", file=sys.stderr)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hypergrad_oracle as orc

    return orc.cg, "port", os.path.join(ROOT, "oracle", "hypergrad_oracle.py")


def cpu_baseline(steps, K):
    """The reference's own CPU path (unmodified functions; the restatement only when the package is absent) timed on this
    box's host cores, after the GPU regions."""
    ref_cg, kind, where = reference_cg()

    class orc:   # noqa: N801 - keeps the call sites below as they were
        cg = staticmethod(ref_cg)

    # torch's CPU kernels stop scaling far below the 256 hardware threads of the GPU box's host (EPYC 9575F:
    # 8 thr 0.68 s, 16 thr 0.55 s, 32 thr 0.98 s, 64 thr 1.8 s, 128 thr 5.3 s per step): time a few thread
    # counts and report the BEST one, so the baseline is the reference path at its fastest on this host.
    curr, prev, vector = build(torch.device("cpu"), seed=0, K=K)
    best = None
    per_setting = max(2, steps)
    for threads in (8, 16, 32):
        if threads > (os.cpu_count() or 1):
            continue
        torch.set_num_threads(threads)
        orc.cg(vector, curr, prev, False)  # warm-up
        times = []
        budget_t0 = time.perf_counter()
        for _ in range(per_setting):
            t0 = time.perf_counter()
            orc.cg(vector, curr, prev, False)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - budget_t0 > 10.0:  # bounded sample
                break
        times.sort()
        med_t = times[len(times) // 2]
        if best is None or med_t < best[0]:
            best = (med_t, threads, times)
    med, used_threads, times = best
    steps = len(times)
    torch.set_num_threads(used_threads)
    model, phys, logical = host_info()
    return {
        "value": 1.0 / med,
        "unit": "hypergradient-steps/sec",
        "cores": used_threads,
        "host_model": model,
        "physical_cores": phys,
        "logical_cpus": logical,
        "kind": kind,
        "source": ("betty.hypergradient.cg.cg imported from " + where) if kind == "reference" else where,
        "sample": f"{steps} steps of the same workload (cg K={K}, N=10,034,826, batch {BATCH}), median, at the best of "
        f"8/16/32 torch threads ({used_threads}); " + ("the reference's own cg() on torch CPU fp32" if kind == "reference" else
                                                       "oracle/hypergrad_oracle.py on torch CPU fp32") +
        f", min {times[0]:.3f}s max {times[-1]:.3f}s",
    }


def projected_iteration_work():
    """What ONE iteration of the projected solvers executes / must move (useful batch rows only), from the shapes alone:
    flops on the matrix pipe — the chain through the constant weights, the B x B Gram products, the batch-deep G(raw) products —
    and the analytic minimum of its own traffic: every chain weight read once forward and once backward, the batch-sized
    recurrence arrays (G(r), G(p) read + written, G(raw) written + read) and the chain's activations written + read once."""
    d, B = SIZES, BATCH
    L = len(d) - 1
    chain = sum(2.0 * B * d[l] * d[l + 1] * 2 for l in range(1, L - 1)) + 2.0 * B * d[-2] * d[-1] * 4
    gram = sum(2.0 * B * B * (d[l] + d[l + 1]) for l in range(1, L - 1))
    graw = sum(2.0 * B * B * d[l + 1] * (1 if l == 0 else 2) for l in range(L - 1)) + sum(2.0 * B * B * d[l] * 2 for l in range(1, L - 1))
    weights = sum(2.0 * 4 * d[l] * d[l + 1] for l in range(1, L - 1))
    gwidth = sum(d[l + 1] for l in range(L - 1)) + sum(d[l] for l in range(1, L - 1))
    batch = 6.0 * 4 * B * gwidth + 2.0 * 2 * 4 * B * sum(d[l + 1] for l in range(L - 1))
    return {"flops": chain + gram + graw, "flops_chain": chain, "flops_gram": gram, "flops_graw": graw,
            "bytes_min": weights + batch, "bytes_weights": weights, "bytes_batch_sized": batch}


GOLDEN = os.path.join(ROOT, "tests", "golden", "cfg2_full.npz")
RIDGE_WELL = 0.3   # tests/golden/make_cfg2_golden.py


def parity_check(args, device, jvp_fn, curr, prev, vector, K):
    """Ties the number to a checked answer: the very solver that was just timed (same library, same arms) is run once more
    with sync=False and compared with the committed outputs of the REFERENCE's CPU run (tests/golden/cfg2_full.npz, generated by
    tests/golden/make_cfg2_golden.py from /root/reference).
      metric instance (ridge 1e-2, seed 0 — what was timed): the reference's own fp32 answer sits `reference_own_spread` from its
        fp64 answer (20 un-preconditioned CG iterations on an indefinite-plus-small-ridge Hessian), so rtol 1e-4 is undecidable
        there against the fp32 output: the product is held to rtol 1e-4 against the reference's fp64 output (the truth), and to 3x
        the reference's own spread against its fp32 output;
      well-conditioned variant (ridge 0.3, same shapes, same kernels, same K): rtol 1e-4 against the reference's fp32 output,
        ASSERTED — bench.py refuses to print a throughput when it fails."""
    import numpy as np

    key = {("cg", 20): "cg20", ("neumann", 10): "neumann10"}.get((args.algo, K))
    # (--mode global at world size 1 solves the very instance the goldens hold — seed 0, the whole batch on one rank: the call site runs
    #  this check at world size 1 only, so the global-batch forms are held to the same goldens)
    if key is None or not os.path.exists(GOLDEN):
        return None
    gold = np.load(GOLDEN)

    def rel(got, want):
        g = np.concatenate([t.detach().double().cpu().numpy().ravel() for t in got])
        w = np.asarray(want, dtype=np.float64).ravel()
        return float(np.linalg.norm(g - w) / np.linalg.norm(w)) if np.all(np.isfinite(g)) else float("inf")

    out = jvp_fn(vector, curr, prev, False)
    spread = float(gold[f"metric/0/{key}/ref_spread"])
    res = {
        "golden": f"tests/golden/cfg2_full.npz:metric/0/{key} + well/{int(gold['well/seeds'][0])}/{key} (outputs of the reference's "
                  "own functions on the CPU, tests/golden/make_cfg2_golden.py)",
        "metric_instance": {
            "vs_reference_cpu_fp32": rel(out, gold[f"metric/0/{key}/fp32"]),
            "vs_reference_fp64": rel(out, gold[f"metric/0/{key}/fp64"]),
            "reference_own_spread": spread,
        },
    }
    mi = res["metric_instance"]
    # Two gates of different width: the distance to the fp64 TRUTH is held to north_star's rtol 1e-4 (the product
    # delivers ~1e-6 on this seed); only the distance to the reference's fp32 CPU output — which itself sits `spread` from that
    # truth — keeps the bound of 3x the reference's own spread.
    mi["rtol_vs_fp64"] = 1e-4
    mi["bound_vs_reference_cpu_fp32"] = max(1e-4, 3.0 * spread)
    mi["ok"] = bool(mi["vs_reference_fp64"] <= mi["rtol_vs_fp64"] and mi["vs_reference_cpu_fp32"] <= mi["bound_vs_reference_cpu_fp32"])
    wseed = int(gold["well/seeds"][0])
    curr_w, prev_w, vector_w = build(device, seed=wseed, K=K, algo=args.algo, ridge=RIDGE_WELL)
    if args.hvp == "analytic":
        declare_structure(curr_w, "hip", fused=not args.no_fuse, keep_solution=args.keep_solution, native_upper=args.upper == "closed-form")
    elif args.hvp == "analytic-aten":
        declare_structure(curr_w, "torch")
    out_w = jvp_fn(vector_w, curr_w, prev_w, False)
    rw = rel(out_w, gold[f"well/{wseed}/{key}/fp32"])
    res["well_conditioned_variant"] = {"ridge": RIDGE_WELL, "seed": wseed, "vs_reference_cpu_fp32": rw,
                                       "vs_reference_fp64": rel(out_w, gold[f"well/{wseed}/{key}/fp64"]),
                                       "reference_own_spread": float(gold[f"well/{wseed}/{key}/ref_spread"]),
                                       "rtol": 1e-4, "ok": bool(rw <= 1e-4)}
    if not (res["well_conditioned_variant"]["ok"] and mi["ok"]):
        raise SystemExit("bench.py: the timed solver does not reproduce the reference's CPU output — refusing to report a "
                         "throughput for a wrong result: " + json.dumps(res))
    return res


def secondary_lines(args, device, be, N):
    """Driver-visible evidence beside the headline, ~2 s, AFTER every timed region of the headline:
      neumann10     — BASELINE cfg 2's OWN algorithm (betty/hypergradient/neumann.py:59-66, K = 10) on the same workload: steps/s,
                      the event-free per-iteration time from interleaved (K, K/2) regions, SURVEY 8(d)'s 20*N-byte yardstick against
                      8 TB/s, and the in-run parity against the reference-CPU goldens (tests/golden/cfg2_full.npz: neumann10);
      cg_resident   — the opaque path's streaming recurrence kernel k_cg_resident (one launch per CG iteration after PyTorch's double
                      backward; what an un-annotated Betty problem runs): HIP-event time per launch on the launch stream and the
                      28*N bytes it really moves against 8 TB/s."""
    import argparse
    import ctypes

    from betty_amd import _native

    out = {}
    steps = max(5, min(args.steps, 100))

    def region(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    # ---- Neumann K = 10 --------------------------------------------------------------------------------------------------
    K = 10
    curr, prev, vector = build(device, seed=0, K=K, algo="neumann")
    declare_structure(curr, "hip", native_upper=args.upper == "closed-form")
    fn = hg.jvp_fn_mapping["neumann"]

    def step():
        for p in prev.parameters():
            p.grad = None
        assert fn(vector, curr, prev, True) is None

    p0 = int(be.lib.bhg_mlp_proj_iterations())
    region(step, 8)
    t_full, t_half = [], []
    for _ in range(3):
        t_full.append(region(step, steps))
        curr.config.neumann_iterations = K // 2
        step()
        t_half.append(region(step, steps))
        curr.config.neumann_iterations = K
        step()
    projected = int(be.lib.bhg_mlp_proj_iterations()) > p0
    pair_us = [1e6 * (a - b) / (steps * (K - K // 2)) for a, b in zip(t_full, t_half)]
    it_us = med(pair_us)
    nargs = argparse.Namespace(**{**vars(args), "algo": "neumann"})
    par = parity_check(nargs, device, fn, curr, prev, vector, K)
    finite = all(bool(torch.isfinite(p.grad).all()) for p in prev.parameters())
    out["neumann10"] = {
        "metric": "hypergradient-steps/sec (neumann K=10, 10M inner params) — BASELINE cfg 2's own algorithm on the metric workload",
        "value": steps / med(t_full), "unit": "hypergradient-steps/sec", "ms_per_step": 1e3 * med(t_full) / steps,
        "steps_each": steps, "pairs": 3, "per_iteration_us": it_us, "per_iteration_us_pairs": pair_us,
        "solver_form": "projected Neumann (six launches per iteration)" if projected else "classic chain",
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_launch": 20.0 * N, "achieved": 20.0 * N / (it_us * 1e-6) / 1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": 20.0 * N / (it_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                     "note": "SURVEY 8(d)'s yardstick: the 20*N bytes the reference's Neumann recurrence moves per iteration over this "
                             "solver's event-free iteration time"},
        "finite": finite, "parity": par,
    }
    del curr, prev, vector
    # ---- the opaque path's recurrence kernel -----------------------------------------------------------------------------
    K = 20
    curr, prev, vector = build(device, seed=0, K=K, algo="cg")
    curr.hypergradient_graph = False
    fn = hg.jvp_fn_mapping["cg"]
    layout = be.layout(vector)
    resident = bool(be.lib.bhg_cg_resident_usable(int(layout.n_chunks)))
    for _ in range(2):
        fn(vector, curr, prev, False)
    torch.cuda.synchronize()
    be.lib.bhg_timing_enable(1)
    n_op = 4
    t_op = region(lambda: fn(vector, curr, prev, False), n_op)
    tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
    _native.check(be.lib.bhg_timing_read(0, ctypes.byref(tot), ctypes.byref(cnt)), "bhg_timing_read")
    be.lib.bhg_timing_enable(0)
    if cnt.value:
        us = 1e3 * tot.value / cnt.value
        out["cg_resident"] = {
            "kernel": ("k_cg_resident (1 launch per CG iteration)" if resident else "k_cg_dot + k_cg_resid + k_cg_dir (3 launches per CG iteration)") +
                      " behind PyTorch's double backward: the path of an inner problem WITHOUT a declared structure",
            "avg_launch_us": us, "launches_timed": cnt.value, "timer": "HIP events recorded inside libbhg on the launch stream (bhg_timing)",
            "algorithmic_bytes_per_launch": 28.0 * N, "achieved": 28.0 * N / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": 28.0 * N / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
            "traffic": pmc_traffic("k_cg_resident" if resident else "cg_stream", N)[0],
            "opaque_steps_per_sec": n_op / t_op,
            "note": "this kernel really moves its 28*N bytes (read Hp, p, r, x; write x, r, p); the state fits the 256 MiB Infinity Cache, "
                    "cache-defeated figure: profiles/r02_bench_kernels_N10M_cache_defeated.json",
        }
    return out


def device_census(dist, world, rank, device, M):
    """Makes an N > 1 line falsifiable: how many ranks the process group REALLY spans (an all-reduce of ones), which
    physical device each rank holds (name, PCI bus id, uuid when the runtime exposes them: N distinct entries or the run is not N GPUs), and
    the measured latency of the one collective a replica-mode step contains — the all-reduce of the M-sized hypergradient."""
    props = torch.cuda.get_device_properties(device)
    mine = {"rank": rank, "device_index": device.index, "name": props.name,
            "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xFF, getattr(props, "pci_device_id", -1) & 0xFF)
            if hasattr(props, "pci_bus_id") else None,
            "uuid": str(getattr(props, "uuid", "")) or None, "pid": os.getpid()}
    if dist is None or world == 1:
        return {"ranks_seen": 1, "devices": [mine], "distinct_devices": 1, "backend": None, "allreduce_M_floats_us": None}
    ones = torch.ones(1, device=device, dtype=torch.float32)
    dist.all_reduce(ones)
    devices = [None] * world
    dist.all_gather_object(devices, mine)
    buf = torch.zeros(M, device=device, dtype=torch.float32)
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    lat = torch.tensor([1e6 * (time.perf_counter() - t0) / n], device=device, dtype=torch.float64)
    dist.all_reduce(lat, op=dist.ReduceOp.MAX)
    ids = {(d.get("pci"), d.get("uuid"), d.get("device_index")) for d in devices}
    return {"ranks_seen": int(round(float(ones.item()))), "devices": devices, "distinct_devices": len(ids), "backend": dist.get_backend(),
            "allreduce_M_floats_us": float(lat.item()),
            "allreduce_note": "%d back-to-back all-reduces of the M = %d float hypergradient, max over ranks: the only collective of a replica-mode step" % (n, M)}


def resolve_hw_queues(flag, world, has_collective_stream, env):
    """--hw-queues: (value of GPU_MAX_HW_QUEUES for this process or None, where it came from); writes `env` when bench.py decides.
      flag > 0   that value, always;
      flag == -1 1 when a step of the run involves a second stream (more than one rank, or an emulated collective) and the variable is not
                 already set — a second active hardware queue costs every dependent launch of the solver ~1 us on this runtime;
      flag == 0  never touched."""
    if flag > 0 or (flag == -1 and (world > 1 or has_collective_stream) and "GPU_MAX_HW_QUEUES" not in env):
        env["GPU_MAX_HW_QUEUES"] = str(flag if flag > 0 else 1)
        return env["GPU_MAX_HW_QUEUES"], "set by bench.py"
    if "GPU_MAX_HW_QUEUES" in env:
        return env["GPU_MAX_HW_QUEUES"], "from the environment"
    return None, "runtime default"


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU over RCCL."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # ~1 s of GPU work: visible to coarse (SMI) utilisation sampling
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cg-iters", type=int, default=20, help="K: CG / Neumann iterations")
    ap.add_argument("--algo", choices=["cg", "neumann", "darts"], default="cg",
                    help="cg = the BASELINE metric; neumann / darts = secondary lines (BASELINE cfg 2 uses neumann K=10)")
    ap.add_argument("--global-form", choices=["auto", "one_pass", "sharded"], default="auto",
                    help="--mode global: auto = the factor-exchange form (batch-sized all-gathers, fully projected solver) when the structure "
                    "takes it, else one-pass, else sharded; one_pass / sharded pin the older forms")
    ap.add_argument("--fx-always-gather", action="store_true",
                    help="--mode global at world size 1: issue the factor-exchange form's all-gathers all the same (what the collectives cost "
                    "an iteration before a byte crosses a link: host calls, the collective's stream / hardware queue)")
    ap.add_argument("--mode", choices=["replica", "global"], default="replica",
                    help="replica = the reference's DDP mode (every rank solves its own problem; default).  global = ONE inner "
                    "problem whose batch is spread over the ranks: data-parallel HVP, sharded CG state (betty_amd/global_hvp.py)")
    ap.add_argument("--variant", choices=["auto", "stream", "resident"], default="auto")
    ap.add_argument("--hvp", choices=["analytic", "analytic-aten", "autograd"], default="analytic",
                    help="analytic = MFMA R-op kernels for the declared MLP structure; autograd = opaque double backward")
    ap.add_argument("--upper", choices=["closed-form", "autograd"], default="closed-form",
                    help="analytic HVP only: the meta-weight-net's sample weights and their VJP in closed form (csrc/bhg_mwn.hip, declared "
                         "through SigmoidMLPWeightNet) or through PyTorch autograd (the round-4 path: A/B)")
    ap.add_argument("--no-fuse", action="store_true",
                    help="analytic HVP only: K x (HVP kernels + recurrence kernel) instead of the one-pass fused solver (A/B)")
    ap.add_argument("--keep-solution", action="store_true",
                    help="fused CG solver: materialise the N-sized solution vector x (default: the hypergradient comes "
                         "from the accumulated Rz(x), x is never written)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="steps of the CPU baseline (0 = skip)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip per-launch HIP events")
    ap.add_argument("--no-hvp-graph", action="store_true",
                    help="--hvp autograd only: eager launches of the double backward instead of the (opt-in) hipGraph replay")
    ap.add_argument("--hvp-graph", choices=["solve", "persistent"], default="persistent",
                    help="--hvp autograd only: capture the K HVPs once per solve, or (default) loss / gradient-with-graph and the "
                         "HVP once for the whole run (the inner training_step is a static function here)")
    ap.add_argument("--tunableop", action="store_true",
                    help="--hvp autograd only: let PyTorch's TunableOp pick the GEMM kernel of every shape of the double backward "
                         "during the warm-up (the GEMMs of the opaque path are PyTorch's, not libbhg's)")
    ap.add_argument("--no-slope", action="store_true", help="skip the K/2 regions (event-free per-iteration time)")
    ap.add_argument("--settle-ms", type=float, default=300.0,
                    help="after the --warmup steps, keep stepping (untimed) until this much wall-clock has passed: clocks at their "
                         "sustained state before the first timed region (0 = off)")
    ap.add_argument("--reps", type=int, default=5,
                    help="timed regions of `--steps` steps each (full and, interleaved, K/2): the line reports the MEDIAN region and the spread")
    ap.add_argument("--hw-queues", type=int, default=-1,
                    help="GPU_MAX_HW_QUEUES for this process (how many hardware queues ROCclr maps the HIP streams onto), set before the first HIP "
                         "call.  -1 (default): 1 when the run has more than one rank (or emulates a collective) and the variable is not set in the "
                         "environment, else untouched; 0: never touch it.  Why: a process-group collective runs on a stream of its own, and once a "
                         "SECOND hardware queue has been active in a step every dependent launch of the solver costs ~1 us more (measured at N = 1 with "
                         "an emulated per-step collective: 678 vs 755 steps/s, DESIGN 4b)")
    ap.add_argument("--emulate-collective", choices=["none", "blocking", "deferred"], default="none",
                    help="A/B at N = 1: after every step touch the M-float result on ANOTHER stream the way a process-group collective does "
                         "(blocking: the main stream waits at once, like dist.all_reduce; deferred: it waits at the next step's hop) — what a second "
                         "hardware queue costs the K loop")
    ap.add_argument("--own-stream", action="store_true",
                    help="A/B: run every step on a stream created for the run instead of the process's default (null) stream")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object (Neumann K = 10 and the opaque path's k_cg_resident, ~2 s after the timed regions)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the in-run check of the timed solver against the committed reference-CPU goldens (A/B sweeps)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; default) | gloo (debug only)")
    ap.add_argument("--ab-lib", action="store_true",
                    help="run on the measurement build libbhg_ab.so with every arm at its default (what --debug implies): the A/B partner "
                         "of a --debug line, and the check that the two builds run the same default form")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=INT",
                    help="select a measurement arm of libbhg through bhg_debug_set (the library reads no environment variable); "
                         "repeatable, e.g. --debug packed_chain=0 --debug mlp_proj=0.  Echoed in config.debug_arms")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # ONE JSON line on stdout, nothing else: RCCL prints a version banner on the C-level stdout when a process group
    # comes up, so file descriptor 1 is pointed at stderr for the whole run and the line is written to the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one hardware queue for a run whose steps contain a collective (see --hw-queues); must happen before the HIP runtime comes up
    gpu_max_hw_queues, hwq_note = resolve_hw_queues(args.hw_queues, world, args.emulate_collective != "none", os.environ)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; betty_amd has no CPU path"
    if os.environ.get("BHG_ALL_RANKS_ON_GPU0") == "1":  # debug: exercise the N>1 code path on a 1-GPU box (gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.mode == "global":
        import torch.distributed as dist

        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from betty_amd import _native
    from betty_amd.backend import get_backend
    from betty_amd.distributed import fence_grads

    be = get_backend()
    if args.debug or args.ab_lib:   # measurement arms live in the measurement build (libbhg_ab.so: same sources, -DBHG_AB)
        _native.use_ab(True)
    for kv in args.debug:
        key, _, val = kv.partition("=")
        _native.debug_set(key, int(val))
    be.cg_variant = {"auto": _native.BHG_CG_AUTO, "stream": _native.BHG_CG_STREAM, "resident": _native.BHG_CG_RESIDENT}[args.variant]
    K = args.cg_iters
    if args.mode == "global":
        assert args.algo in ("cg", "neumann"), "--mode global: cg (three forms) or neumann (the factor-exchange form)"
        # the same inner / upper weights on every rank, a different batch per rank
        curr, prev, vector = build(device, seed=0, ddp=world > 1, K=K, algo=args.algo, data_seed=rank)
        jvp_fn = hg.jvp_fn_mapping["cg_global" if args.algo == "cg" else "neumann_global"]
        import betty_amd.global_hvp as _ghvp

        _ghvp.GLOBAL_FORM = args.global_form
        _ghvp.FX_ALWAYS_GATHER = bool(args.fx_always_gather)
    else:
        curr, prev, vector = build(device, seed=rank, ddp=world > 1, K=K, algo=args.algo)
        jvp_fn = hg.jvp_fn_mapping[args.algo]
    if args.hvp == "analytic":
        declare_structure(curr, "hip", fused=not args.no_fuse, keep_solution=args.keep_solution, native_upper=args.upper == "closed-form")
    elif args.hvp == "analytic-aten":  # same closed form on rocBLAS/ATen ops (A/B reference for the MFMA kernels)
        declare_structure(curr, "torch")
    else:   # opaque double backward: opt into the hipGraph replay of the K HVPs (betty_amd/hypergradient/_common.py)
        if args.tunableop:
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
            if hasattr(torch.cuda.tunable, "write_file_on_exit"):
                torch.cuda.tunable.write_file_on_exit(False)
            else:   # keep the results file out of the working tree
                torch.cuda.tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "bhg_tunableop_results.csv"))
        curr.hypergradient_graph = False if args.no_hvp_graph else (True if args.hvp_graph == "solve" else "persistent")
    N = sum(p.numel() for p in curr.parameters())
    M = sum(p.numel() for p in prev.parameters())
    layout = be.layout(vector)
    fused = args.hvp == "analytic" and not args.no_fuse and args.algo in ("cg", "neumann") and args.mode == "replica"
    # the predicate bhg_cg_step itself uses (capacity AND the residency census)
    resident = (not fused) and (args.variant == "resident" or (
        args.variant == "auto" and bool(be.lib.bhg_cg_resident_usable(int(layout.n_chunks)))))

    # per-launch timing: HIP events attached to the kernels / launch groups themselves on the launch stream
    # (hipExtLaunchKernelGGL / hipEventRecord inside libbhg; bhg_timing_enable/read in include/bhg.h)
    timing_on = not args.no_kernel_timing

    own_stream = torch.cuda.Stream(device) if args.own_stream else None
    if own_stream is not None:
        own_stream.wait_stream(torch.cuda.current_stream(device))

    coll_stream = torch.cuda.Stream(device) if args.emulate_collective != "none" else None
    pending = []

    def step():
        if pending:   # deferred: the fence falls where .grad is next touched
            torch.cuda.current_stream(device).wait_stream(coll_stream)
            pending.clear()
        for p in prev.parameters():
            p.grad = None
        if own_stream is not None:
            with torch.cuda.stream(own_stream):
                out = jvp_fn(vector, curr, prev, True)
        else:
            out = jvp_fn(vector, curr, prev, True)
        assert out is None
        if coll_stream is not None:
            g = next(iter(prev.parameters())).grad
            coll_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(coll_stream):
                g.mul_(1.0)   # stands in for the M-float all-reduce
            if args.emulate_collective == "blocking":
                torch.cuda.current_stream(device).wait_stream(coll_stream)
            else:
                pending.append(1)

    def timed_region(n):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        fence_grads()   # the deferred all-reduce of the region's LAST step belongs to the region (steps 1 .. n-1 were fenced by their successors)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    # Settle (untimed, reported as `settle_steps`): W = 5 warm-up steps are 7 ms of GPU work — the first timed region of a 20-step run
    # then still sees the clocks ramp (round 5, two boxes: its first (full, half) pair read -74 and +75 us per iteration next to
    # 60.0-60.7 for the other four).  Steps are run for about --settle-ms of wall-clock (default 300; 0 = off).
    # The COUNT is what every rank must agree on (a step of a multi-rank run contains a collective): eight probe steps are timed,
    # the number of further steps follows from that time, and the ranks take the maximum of their counts.
    settle_steps = 0
    if args.settle_ms > 0:
        probe = 8
        torch.cuda.synchronize()
        t_settle = time.perf_counter()
        for _ in range(probe):
            step()
        torch.cuda.synchronize()
        per_step = max((time.perf_counter() - t_settle) / probe, 1e-6)
        more = max(0, min(4096, int(1e-3 * args.settle_ms / per_step + 0.999) - probe))
        if dist is not None:
            t = torch.tensor([more], device=device, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            more = int(t.item())
        for i in range(more):
            step()
            if i % 8 == 7:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        settle_steps = probe + more
    # Region 1 — the headline: exactly `steps` steps, nothing but the product path on the stream, bracketed by barrier +
    # synchronize on both sides.  Repeated `--reps` times (default 5): `value` / `ms_per_step` are the MEDIAN region, every region
    # and the min-max spread are in the line (`regions`).  One region of 20 steps is 30 ms: a single sample of it moves by
    # several percent with the clock state of the box (round-4 verdict).
    # Region 1b — the same `steps` steps with HALF the iterations, equally free of events, INTERLEAVED with the full regions
    # (full, half, full, half, ...: slow drifts of the box cancel inside a pair): the difference of a pair is K/2 full iterations
    # per step on the SAME clock as the headline (no per-launch events, no profiler):  iteration = (t(K) - t(K/2)) / (K/2) per
    # pair, reported as the median over the pairs with its spread;  outside the K loop = t(K) - K * iteration (it absorbs what
    # the short last iteration saves).
    reps = max(1, args.reps)
    slope = args.algo in ("cg", "neumann") and K >= 2 and not args.no_slope
    key = "cg_iterations" if args.algo == "cg" else "neumann_iterations"
    h0, p0 = int(be.lib.bhg_mlp_hoist_launches()), int(be.lib.bhg_mlp_proj_iterations())
    t_full, t_half = [], []
    n_hoist = n_proj = 0
    for rep in range(reps):
        c0, c1 = int(be.lib.bhg_mlp_hoist_launches()), int(be.lib.bhg_mlp_proj_iterations())
        t_full.append(timed_region(args.steps))
        if rep == 0:
            n_hoist, n_proj = int(be.lib.bhg_mlp_hoist_launches()) - c0, int(be.lib.bhg_mlp_proj_iterations()) - c1
        if slope:
            setattr(curr.config, key, K // 2)
            step()
            t_half.append(timed_region(args.steps))
            setattr(curr.config, key, K)
            step()
    if dist is not None:   # every region: the max over the ranks
        t = torch.tensor(t_full + t_half, device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t = [float(v) for v in t.tolist()]
        t_full, t_half = t[:len(t_full)], t[len(t_full):]

    def med(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    elapsed = med(t_full)
    elapsed_half = med(t_half) if t_half else None
    pair_iter_us = [1e6 * (a - b) / (args.steps * (K - K // 2)) for a, b in zip(t_full, t_half)]
    solver_form = None
    if fused and args.algo == "cg":
        if n_proj == args.steps * (K - 1) and n_hoist == args.steps:
            solver_form = ("fully projected CG (libbhg default without a solution vector): after the first iteration the direction "
                           "products, r.r, p.p and p.Hp all come from batch-sized recurrences through B x B Gram matrices — no N-sized "
                           "state vector is read or written; bhg_mlp_proj_iterations / bhg_mlp_hoist_launches = %d / %d" % (n_proj, n_hoist)
                           if not args.keep_solution else
                           "projected CG with the N-sized r / p kept (the caller asked for x): %d projected iterations" % n_proj)
        elif n_hoist == args.steps * K:
            solver_form = "hoisted chain on the N-sized residual every iteration (BHG_MLP_PROJ=0)"
        elif n_hoist == 0:
            solver_form = "classic chain (BHG_MLP_HOIST=0)"
    elif fused and args.algo == "neumann":
        solver_form = ("projected Neumann solver (libbhg default without an accumulator vector): G(v') = G(v) - alpha (G(raw) + shift G(v)) "
                       "through B x B Gram matrices, nothing N-sized after the first iteration, closing half pass for Rz(v_K); "
                       "bhg_mlp_proj_iterations = %d" % n_proj) if n_proj == args.steps * K else "classic chain"
    if args.mode == "global" and args.hvp == "analytic":
        try:   # what the library's plan says this descriptor takes under Config(type="cg_global") (host logic only)
            pd = _native.plan_describe([p_.shape[1] for p_ in list(curr.parameters())[0::2]] + [list(curr.parameters())[-1].shape[0]],
                                       int(curr.cur_batch[0].shape[0]), "cg", args.keep_solution)
            solver_form = "global-batch CG, %s form (bhg_mlp_plan_describe); requested --global-form %s" % (pd["global_form"], args.global_form)
        except Exception as exc:   # noqa: BLE001
            solver_form = f"global-batch CG (plan description unavailable: {type(exc).__name__})"
    # Region 2 — the same `steps` steps again with HIP events around the launch groups (recorded inside libbhg on the
    # launch stream) for the roofline objects.  Kept out of region 1 because every event record costs the stream a
    # ~4 us bubble (measured: 210 vs 192 steps/s with 4 records per CG iteration); its throughput is reported as
    # `value_with_kernel_timing`.
    spans = {}
    elapsed_timed = None
    if timing_on:
        import ctypes

        be.lib.bhg_timing_enable(1)
        elapsed_timed = timed_region(args.steps)
        for name, kind in (("cg_step", 0), ("neumann_step", 1), ("hvp", 2), ("cg_iter", 3)):
            tot, cnt = ctypes.c_double(0.0), ctypes.c_int(0)
            _native.check(be.lib.bhg_timing_read(kind, ctypes.byref(tot), ctypes.byref(cnt)), "bhg_timing_read")
            if cnt.value:
                spans[name] = (1e3 * tot.value / cnt.value, cnt.value)   # (average us, launches)
        be.lib.bhg_timing_enable(0)
    if dist is not None and elapsed_timed is not None:
        t = torch.tensor([elapsed_timed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_timed = float(t.item())
    be.check_health()

    fence_grads()
    finite = all(bool(torch.isfinite(p.grad).all()) for p in prev.parameters())
    if not finite:
        raise SystemExit("bench.py: non-finite hypergradient — refusing to report a throughput for a wrong result")
    parity = parity_check(args, device, jvp_fn, curr, prev, vector, K) if (world == 1 and not args.no_parity) else None
    census = device_census(dist, world, rank, device, M)
    secondary = None
    if (world == 1 and args.algo == "cg" and K == 20 and fused and not args.no_secondary and not args.no_parity and not args.debug
            and not args.keep_solution):
        try:   # evidence BESIDE the headline: a failure here is reported in the line, it does not take the headline with it
            secondary = secondary_lines(args, device, be, N)
        except (Exception, SystemExit) as exc:   # noqa: BLE001
            secondary = {"error": f"{type(exc).__name__}: {exc}"[:2000]}
            print(f"bench.py: the secondary lines failed: {secondary['error']}", file=sys.stderr)
    out = None
    one_pass_solves = 0
    fx_stats = None
    if args.mode == "global":
        from betty_amd.global_hvp import FX_STATS, ONE_PASS_STATS

        one_pass_solves = ONE_PASS_STATS["solves"]
        fx_stats = dict(FX_STATS)
    if rank == 0:
        # replica mode: every rank completes `steps` independent hypergradient steps; global mode: the ranks share ONE
        # problem (global batch = world x local batch) and complete `steps` steps together
        value = (world if args.mode == "replica" else 1) * args.steps / elapsed
        # CG iteration 28*N B (read Hp,p,r,x; write x,r,p), Neumann iteration 20*N B; analytic HVP:
        # >= 20*N B of weight traffic (read W, V twice, write the HVP once) — with the fused solver the HVP output is
        # not written and the recurrence does not read it, but the algorithmic figure is kept (it is the yardstick).
        rec_bytes = (28.0 if args.algo == "cg" else 20.0) * N
        roof = None
        mall_note = ("working set (x, r, p, W, activations = %.0f MB) is smaller than the 256 MiB Infinity Cache: this is "
                     "fabric-side bandwidth on cache-resident data, quoted against the 8 TB/s HBM peak as north_star asks; "
                     "cache-defeated figures: profiles/r02_bench_kernels_N10M_cache_defeated.json" % ((3 * 4 * N + 4 * N + 8e6) / 1e6))
        # event-free iteration time (region 1 vs region 1b): the span every fused roofline figure below is quoted on
        iter_us = med(pair_iter_us) if pair_iter_us else None   # median over the (full, half) pairs
        if fused and args.algo in ("cg", "neumann") and (iter_us is not None or ("cg_iter" if args.algo == "cg" else "hvp") in spans):
            ev_us, ev_n = spans.get("cg_iter" if args.algo == "cg" else "hvp", (None, 0))
            us = iter_us if iter_us is not None else ev_us
            traffic, traffic_src = pmc_traffic("cg_iter_fused" if args.algo == "cg" else "neumann_iter_fused", N)
            alg = rec_bytes                     # 28*N per CG iteration, 20*N per Neumann iteration
            comp = rec_bytes + 20.0 * N         # + Appendix A.3's HVP weight traffic (read W, V twice, write H*dir once)
            roof = {"bound": "hbm",
                    "kernel": ("bhg_mlp_cg_solve: one whole CG-HVP iteration — fully projected form, SIX dependent launches: the R-chain "
                               "through the constant weights on packed operands (k_wskpl: first product by linearity on Rh_0(r'), beta computed "
                               "and published inside the launch; k_wskpc: pre-head product; k_headu: the head with the recurrences G(p') = G(r') + "
                               "beta G(p) as a second block class; k_wskpc x2; the B x B Gram products ride in them), then k_graw (G(raw) products with the inner "
                               "products, the residual step G(r') = G(r) - alpha (G(raw) + shift G(p)) and Rh_0(r') in their epilogue, small "
                               "slices' outputs, step length)" if (args.algo == "cg" and solver_form and solver_form.startswith("fully")) else
                               "bhg_mlp_cg_solve: one whole fused CG-HVP iteration (R-chain + k_cg_alpha + k_outer_all, whose "
                               "epilogue carries the r/p update)" if args.algo == "cg" else
                               "bhg_mlp_neumann_solve: one whole Neumann-HVP iteration (" + (
                                   "projected form, SIX dependent launches since round 5: the R-chain through the constant weights on packed "
                                   "operands (k_wskpc x2 forward with the Gram rider, k_head_forward, k_wskpc x2 backward), then k_graw — G(raw) products, "
                                   "the update G(v') = G(v) - alpha (G(raw) + shift G(v)) and Rh_0(v') in its epilogue, small slices' outputs; no update launch"
                                   if (solver_form or "").startswith("projected") else "classic chain") + ")"),
                    "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "algorithmic_bytes_per_launch": alg,
                    "avg_launch_us": us,
                    "avg_launch_us_source": ("event-free: (t(K) - t(K/2)) / (K/2) per interleaved pair of timed regions of %d steps (the "
                                             "headline's own clock), median of %d pairs" % (args.steps, len(pair_iter_us)))
                    if iter_us is not None else "HIP events around the iteration (bhg_timing)",
                    "avg_launch_us_pairs": pair_iter_us or None,
                    "avg_launch_us_min_max": [min(pair_iter_us), max(pair_iter_us)] if pair_iter_us else None,
                    "avg_launch_us_hip_events": ev_us, "launches_timed": ev_n,
                    "hip_events_note": "the event figure brackets every iteration with FOUR hipEventRecord calls on the launch stream (iteration "
                                       "span + HVP span); each record is a ~3-4 us bubble in a chain of dependent 6-14 us launches, so it reads "
                                       "~12 us above the event-free slope by construction — it is kept as the live HIP-event cross-check, the "
                                       "slope (no events, no profiler) is the figure that agrees with rocprofv3's per-kernel sums (profiles/)",
                    "traffic": traffic, "traffic_source": traffic_src,
                    "achieved_on_traffic_GBps": (traffic / (us * 1e-6) / 1e9) if traffic else None,
                    "own": None,
                    "composite_recurrence_plus_hvp_weights": {"algorithmic_bytes": comp, "achieved": comp / (us * 1e-6) / 1e9,
                                                              "frac": comp / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                                                              "note": "round-2's figure (28N + 20N); the fused path neither writes nor re-reads H*p, "
                                                                      "so this counts bytes the kernels do not move — yardstick only"},
                    "note": mall_note + "; `achieved` is SURVEY 8(d)'s yardstick — the 28*N bytes the REFERENCE's recurrence moves per "
                            "iteration divided by this solver's iteration time; the projected solver itself moves far fewer bytes (`traffic`), "
                            "its iteration is a chain of 6 dependent launches on batch-sized data plus two passes over the constant weights of the chain"}
            if (solver_form or "").startswith(("fully", "projected")):
                # the solver's OWN roofline next to the 8(d) yardstick: what it executes and must move, not what the reference moves
                w = projected_iteration_work()
                own_bytes = traffic if traffic else w["bytes_min"]
                floor_us = 1e6 * max(own_bytes / (HBM_PEAK_GBPS * 1e9), w["flops"] / 157.3e12)
                roof["own"] = {
                    "bytes_min_analytic": w["bytes_min"], "bytes_weights": w["bytes_weights"], "bytes_batch_sized": w["bytes_batch_sized"],
                    "traffic_measured": traffic, "flops_executed": w["flops"],
                    "floor_us": floor_us, "floor_uses": "measured traffic" if traffic else "analytic minimum traffic",
                    "frac_of_own_floor": floor_us / us,
                    "frac_hbm_on_own_bytes": own_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                    "frac_mfma_on_executed_flops": w["flops"] / (us * 1e-6) / 157.3e12,
                    "note": "floor_us = max(own bytes / 8 TB/s, executed flops / 157.3 TFLOP/s fp32 MFMA); the iteration is a chain of "
                            "dependent launches on batch-sized data, bound by neither roof — this fraction says how far from them",
                }
        elif ("cg_step" if args.algo == "cg" else "neumann_step") in spans:
            us, n = spans["cg_step" if args.algo == "cg" else "neumann_step"]
            roof = {
                "bound": "hbm",
                "kernel": ("k_neumann_step (1 launch/iter)" if args.algo == "neumann" else
                           "k_cg_resident (1 launch/iter)" if resident else "k_cg_dot+k_cg_resid+k_cg_dir (3 launches/iter)"),
                "achieved": rec_bytes / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": rec_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "traffic": pmc_traffic("k_cg_resident" if resident else "cg_stream", N)[0] if args.algo == "cg" else None,
                "traffic_source": pmc_traffic("k_cg_resident" if resident else "cg_stream", N)[1] if args.algo == "cg" else None,
                "algorithmic_bytes_per_launch": rec_bytes, "avg_launch_us": us, "launches_timed": n, "note": mall_note,
            }
        hvp_roof = None
        if "hvp" in spans:
            # useful flops of one analytic HVP (B valid rows, no padding): R-forward + R-backward + weight-shaped outputs
            d = SIZES
            fwd = sum(2.0 * BATCH * d[l] * d[l + 1] * (1 if l == 0 else 2) for l in range(len(d) - 1))
            bwd = sum(2.0 * BATCH * d[l] * d[l + 1] * 2 for l in range(1, len(d) - 1))
            outer = sum(2.0 * BATCH * d[l] * d[l + 1] * (1 if l == 0 else 2) for l in range(len(d) - 1))
            flops = fwd + bwd + outer
            us, n = spans["hvp"]
            src = "HIP events around the HVP chain (bhg_timing)"
            if fused and iter_us is not None:   # the whole iteration IS the HVP chain plus two ~5 us scalar launches
                us, src = iter_us, "event-free whole-iteration time (see roofline.avg_launch_us_source)"
            # what the projected iteration actually executes on the matrix pipe (useful rows only): the constant-weight chain,
            # the B x B Gram products and the batch-deep G(raw) products
            chain = sum(2.0 * BATCH * d[l] * d[l + 1] * 2 for l in range(1, len(d) - 2)) + 2.0 * BATCH * d[-2] * d[-1] * 4
            gram = sum(2.0 * BATCH * BATCH * (d[l] + d[l + 1]) for l in range(1, len(d) - 2))   # T_l, E_l per MFMA layer behind the first
            graw = sum(2.0 * BATCH * BATCH * d[l + 1] * (1 if l == 0 else 2) for l in range(len(d) - 2)) + \
                sum(2.0 * BATCH * BATCH * d[l] * 2 for l in range(1, len(d) - 2))
            hvp_roof = {
                "bound": "mfma",
                "kernel": "one H*p of the reference (R-forward, R-backward, weight-shaped outputs: the yardstick's 7.0 GFLOP)" +
                          ("; the fully projected solver obtains the same iteration from %.2f GFLOP (constant-weight chain %.2f + Gram "
                           "products %.2f + G(raw) products %.2f): the weight-shaped outputs are never formed"
                           % ((chain + gram + graw) / 1e9, chain / 1e9, gram / 1e9, graw / 1e9)
                           if (solver_form or "").startswith("fully") else
                           "; fused: its output kernels also carry the recurrence's r/x (or v/p) update" if fused else ""),
                "executed_flops_per_iteration": (chain + gram + graw) if (solver_form or "").startswith(("fully", "projected")) else None,
                # achieved / frac count what the matrix pipe EXECUTES; the reference HVP's flops over this solver's time is a
                # speed-up yardstick, kept under its own name
                "achieved": ((chain + gram + graw) if (solver_form or "").startswith(("fully", "projected")) else flops) / (us * 1e-6) / 1e12,
                "peak": 157.3, "unit": "TFLOP/s",
                "frac": ((chain + gram + graw) if (solver_form or "").startswith(("fully", "projected")) else flops) / (us * 1e-6) / 1e12 / 157.3,
                "yardstick_reference_hvp": {"flops": flops, "over_this_solvers_time_TFLOPs": flops / (us * 1e-6) / 1e12,
                                            "frac_of_peak": flops / (us * 1e-6) / 1e12 / 157.3,
                                            "note": "the reference's HVP flops divided by this solver's time: not a roofline"},
                "flops_per_call": flops, "avg_call_us": us, "avg_call_us_source": src,
                "avg_call_us_hip_events": spans["hvp"][0], "calls_timed": n,
            }
        K_eff = 0 if args.algo == "darts" else K
        per_iter_us = None
        if fused and iter_us is not None:
            per_iter_us = iter_us
        elif fused and args.algo == "cg" and "cg_iter" in spans:
            per_iter_us = spans["cg_iter"][0]
        elif "hvp" in spans:
            per_iter_us = spans["hvp"][0] + (spans.get("cg_step") or spans.get("neumann_step") or (0.0, 0))[0]
        out = {
            "metric": "hypergradient-steps/sec (CG K=20, 10M inner params)" if (args.algo == "cg" and K == 20)
            else f"hypergradient-steps/sec ({args.algo} K={K}, 10M inner params)",
            "value": value,
            "unit": "hypergradient-steps/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": settle_steps,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "regions": {"reps": reps, "steps_each": args.steps, "statistic": "median region (value, ms_per_step); every region is bracketed by "
                        "barrier + synchronize and is the max over ranks",
                        "ms_per_step_each": [1e3 * t / args.steps for t in t_full],
                        "ms_per_step_min_max": [1e3 * min(t_full) / args.steps, 1e3 * max(t_full) / args.steps],
                        "half_K_ms_per_step_each": [1e3 * t / args.steps for t in t_half] or None},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE cfg2/metric: MLP 3072-2048-1536-384-10 inner (N=%d, 8 tensors), MWN 1-100-1 upper (M=%d), "
                "batch %d, %s K=%d, sync=True" % (N, M, BATCH, args.algo, K),
                "hvp": ("analytic R-op HVP on fp32 MFMA, recurrence fused into its output kernels (one pass)" if fused else
                        "analytic R-op HVP on fp32 MFMA (bhg_mlp_hvp) + recurrence kernel" if args.hvp == "analytic" else
                        "analytic closed form on ATen/rocBLAS" if args.hvp == "analytic-aten" else
                        "pytorch-rocm autograd double backward" + ("" if args.no_hvp_graph else
                                                                   ", K HVPs captured once per solve and replayed as a HIP graph (opt-in)" if args.hvp_graph == "solve" else
                                                                   ", loss / gradient-with-graph and HVP captured ONCE for the run and replayed as two HIP graphs (opt-in)")),
                "blas_of_the_opaque_hvp": ("PyTorch TunableOp (tuned during the warm-up)" if args.tunableop else "PyTorch default") if args.hvp == "autograd" else None,
                "upper": ("meta-weight-net in closed form (bhg_mwn_forward / bhg_mwn_backward: one launch each way)" if args.upper == "closed-form"
                          else "meta-weight-net through PyTorch autograd") if args.hvp == "analytic" else "PyTorch autograd",
                "once_per_step_passes": ("hidden layers behind the first as one launch each on the chain's packed operands (bhg_mlp_forward_packed / "
                                         "_backward_packed); right-hand side read in place (bhg_cg_init_masked + bhg_mlp_cg_solve_rhs)")
                if (fused and not args.debug) else None,
                "cg_variant": "fused-solver" if fused else ("resident" if resident else "stream"),
                "solver_form": solver_form,
                "solution_vector": ("materialised" if (args.keep_solution or not fused or args.algo != "cg") else
                                    "not materialised: the mixed second derivative comes from Rz(x) = sum_k alpha_k Rz(p_k), "
                                    "accumulated from batch-sized factors"),
                "parallelism": (("global batch over %d rank(s), FACTOR-EXCHANGE form: the fully projected solver on sample-partitioned data; per "
                                 "iteration ONE all-gather of batch-sized factors (%d bytes per rank) and ONE of fp64 partials (%d bytes per rank), h_l / "
                                 "delta_l once per solve (%d bytes per rank); nothing N-sized exchanged but the right-hand side's mean; %d solves took it"
                                 % (world, fx_stats["slab_bytes_per_rank"], fx_stats["scal_bytes_per_rank"], fx_stats["const_bytes_per_rank"],
                                    fx_stats["solves"]))
                                if (fx_stats and fx_stats["solves"]) else
                                ("global batch over %d rank(s), one-pass form: replicated x / r / p, per iteration ONE 8-byte all-reduce (p.H_data p) "
                                 "and ONE 4N-byte all-reduce (mean of the locally updated residuals); %d solves took it" % (world, one_pass_solves))
                                if one_pass_solves else
                                ("global-HVP: data-parallel HVP, CG state sharded over %d rank(s), reduce-scatter / all-gather per iteration" % world))
                if args.mode == "global" else ("replicas + DDP all-reduce of the M-sized hypergradient" if world > 1 else "single GPU"),
                "finite": finite,
                "debug_arms": args.debug or None,
                "lib": "libbhg_ab.so (measurement build: A/B table compiled in)" if _native.is_ab() else "libbhg.so (product: no measurement arm in the code object)",
                "lib_sha256": lib_sha256()[:16],
                "gpu_max_hw_queues": f"{gpu_max_hw_queues or 'unset'} ({hwq_note})",
                "emulated_collective": args.emulate_collective if args.emulate_collective != "none" else None,
            },
            "parity": parity,
            "secondary": secondary,
            "ranks_seen": census["ranks_seen"],
            "devices": census,
            "roofline": roof,
            "hvp_roofline": hvp_roof,
            "value_with_kernel_timing": ((world if args.mode == "replica" else 1) * args.steps / elapsed_timed) if elapsed_timed else None,
            "per_iteration_us": per_iter_us,
            "outside_k_loop_ms": ((1e3 * elapsed / args.steps - K_eff * iter_us * 1e-3) if (fused and iter_us is not None) else
                                  (1e3 * elapsed_timed / args.steps - K_eff * per_iter_us * 1e-3) if per_iter_us and elapsed_timed else None),
            "cpu_baseline": None,
        }
        if world == 1 and args.cpu_steps > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps, K)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
