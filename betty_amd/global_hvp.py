"""Global-HVP conjugate gradient: ONE inner problem whose batch is spread over the ranks of a process group.

The reference's distributed mode replicates the whole solve: every rank runs ``cg`` on its own batch and only the
M-sized hypergradient is averaged (that is what ``cg(..., sync=True)`` still does).  This module
is the extension SURVEY.md §8(e)(2) / north_star describe — not in the reference; its oracle is the reference's ``cg``
run in ONE process on the concatenated batch (valid when the inner loss is a batch mean without batch-statistics
layers and all ranks hold equally sized batches):

  * the Hessian-vector product is data parallel: each rank differentiates its own batch,
  * the CG state ``x, r, p`` is SHARDED: rank g owns elements [g*S, (g+1)*S) of the flat vectors (S a multiple of
    4096, so every rank has the same chunk count),
  * per iteration: reduce-scatter(SUM) of the local HVPs (pre-scaled by 1/G: each rank receives its slice of the mean
    HVP — one collective of N/G per peer instead of an all-reduce of N), the three streaming CG kernels on the slice
    with the two dot products completed by all-reducing their per-block partials (bhg_cg_phase), all-gather of the
    new direction,
  * the mixed second derivative is taken on every rank's own batch in direction of the all-gathered solution and
    averaged (by the DDP reducer under ``sync=True``, by one flat all-reduce otherwise).

Collectives run on torch.distributed's stream for the group (RCCL over xGMI with backend "nccl"); the next HVP cannot
start before the direction is gathered, so the overlap available inside one solve is RCCL's own pipelining; the final
M-sized exchange overlaps with the caller's next kernels like any asynchronous collective.

ONE-PASS form (round 3; taken whenever the inner problem's structure has a fused solver — ``provider.fused_cg_global_ready``):
the state is REPLICATED instead of sharded, and the one-pass iteration of ``bhg_mlp_cg_solve`` (recurrence in the epilogues
of the weight-shaped outputs, no N-sized H p) runs on every rank's batch unchanged.  With identical r, p and step length,

    r - alpha * mean_g(H_g) p  ==  mean_g (r - alpha * H_g p),

so per iteration the ranks exchange 8 bytes (all-reduce SUM of p.H_data p, which is a sum over samples of batch-sized
factors) before the step length and 4*N bytes (all-reduce MEAN of the locally updated residual) after the outputs — two
collectives, none in the last iteration but the 8-byte one; x += alpha p and the dot products of the exchanged residual are
replicated work on identical data (include/bhg.h: bhg_mlp_cg_global_phase).  CG is a chain — the next matvec needs the
whole exchanged residual's beta — so inside a solve the large collective has nothing to hide behind; what it buys is one
collective of N instead of reduce-scatter + all-gather + two partial all-reduces, and the fused kernels.
"""
from __future__ import annotations

from typing import Optional

import weakref

import torch
import torch.distributed as dist

from .backend import get_backend
from .hypergradient._common import AutogradHVP, inner_gradient, mixed_vjp
from .hypergradient.structured import structured_hvp_for

SHARD_ALIGN = 4096   # = BHG_CHUNK_ELEMS: equal chunk counts on all ranks


def _backend_name(group) -> str:
    try:
        return str(dist.get_backend(group))
    except Exception:  # pragma: no cover
        return "unknown"


def reduce_scatter_sum(full: torch.Tensor, shard: torch.Tensor, rank: int, group) -> None:
    """shard <- slice ``rank`` of sum_over_ranks(full).  RCCL: one reduce-scatter; gloo (CPU tests) has none."""
    if _backend_name(group) == "gloo":
        tmp = full.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        S = shard.numel()
        shard.copy_(tmp[rank * S:(rank + 1) * S])
    else:
        dist.reduce_scatter_tensor(shard, full, op=dist.ReduceOp.SUM, group=group)


def all_gather_flat(shard: torch.Tensor, full: torch.Tensor, group) -> None:
    dist.all_gather_into_tensor(full, shard, group=group)


class _State:
    """Buffers of one (layout, world size): sharded x, r, p and the Hp slice; full-length direction / HVP / solution."""

    def __init__(self, be, full_layout, world: int):
        per = world * SHARD_ALIGN
        self.Np = (full_layout.flat_size + per - 1) // per * per
        self.S = self.Np // world
        dev = full_layout.device
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.p_full, self.h_full, self.x_full = z(self.Np), z(self.Np), z(self.Np)
        self.x, self.r, self.p, self.hs = z(self.S), z(self.S), z(self.S), z(self.S)
        self.shard_layout = be.layout([self.hs])


_STATES = {}


def all_reduce_mean(flat: torch.Tensor, world: int, group) -> None:
    """flat <- mean over ranks, bit-identical on every rank.  RCCL: one all-reduce with the AVG operator; gloo (CPU tests)
    has no AVG."""
    if _backend_name(group) == "nccl":
        dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=group)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.mul_(1.0 / world)


class _OnePassState:
    def __init__(self, full_layout):
        self.v = full_layout.new_flat()
        self.php = torch.zeros(1, dtype=torch.float64, device=full_layout.device)


_ONE_PASS = weakref.WeakKeyDictionary()
ONE_PASS_STATS = {"solves": 0, "scalar_all_reduces": 0, "residual_all_reduces": 0}   # test / measurement hook


def _cg_global_one_pass(vector, prev, sync, provider, be, full, K: int, alpha: float, G: int, group):
    """The replicated-state form described in the module docstring; returns what cg_global returns."""
    from . import _native  # noqa: PLC0415

    st = _ONE_PASS.get(full)   # (weak-keyed by the layout object: dies with it; round 3 keyed a plain dict by id())
    if st is None:
        st = _ONE_PASS[full] = _OnePassState(full)
    x, r, p = full.state(3)
    skip_x = bool(provider.fused_cg_global_skips_solution(full, K))
    # right-hand side: the mean over the ranks of the local gradients of the upper loss
    be.flatten(full, vector, st.v, 1.0 / G)
    if G > 1:
        dist.all_reduce(st.v, op=dist.ReduceOp.SUM, group=group)
    be.cg_init(full, full.views(st.v, vector), None if skip_x else x, r, p)     # x = 0, r = p = v, partials of r.r
    ONE_PASS_STATS["solves"] += 1
    for k in range(K):
        provider.cg_global_phase(full, x, r, p, k, K, _native.BHG_CG_GLOBAL_CHAIN, G, st.php, alpha)
        if G > 1:
            dist.all_reduce(st.php, op=dist.ReduceOp.SUM, group=group)          # 8 bytes
            ONE_PASS_STATS["scalar_all_reduces"] += 1
        provider.cg_global_phase(full, x, r, p, k, K, _native.BHG_CG_GLOBAL_UPDATE, G, st.php, alpha)
        if k + 1 < K:
            if G > 1:
                all_reduce_mean(r, G, group)                                    # 4*N bytes
                ONE_PASS_STATS["residual_all_reduces"] += 1
            provider.cg_global_phase(full, x, r, p, k, K, _native.BHG_CG_GLOBAL_DOTS, G, st.php, alpha)
    solve = provider.cg_global_finish(full, K, alpha)
    neg_x = full.views(x, vector)
    provider.expects_data_parallel_mean = G > 1   # a closed-form upper hop refuses to accumulate a rank-local share
    if solve is not True:
        out = provider.mixed_vjp(neg_x, sync, solve=solve)
    else:
        out = provider.mixed_vjp(neg_x, sync)
    if sync:
        return None
    from .distributed import exchange_async  # noqa: PLC0415

    return [t.clone() for t in exchange_async(out, group).wait()]


# ---- factor-exchange form (round 6): the fully projected solver on sample-partitioned data ------------------------------------------------
# GLOBAL_FORM: "auto" = factor exchange whenever the structure takes it and the caller does not ask for x (else one-pass, else sharded);
# "one_pass" / "sharded" pin the older forms (tests, bench.py --global-form).
GLOBAL_FORM = "auto"
FX_ALWAYS_GATHER = False   # True: issue the all-gathers at world size 1 as well (scripts/multi_gpu_selfcheck.py: RCCL's in-place
                           # all_gather_into_tensor on the very buffers, where a one-GPU box can reach it)
FX_STATS = {"solves": 0, "const_gathers": 0, "slab_gathers": 0, "scal_gathers": 0, "rhs_all_reduces": 0,
            "slab_bytes_per_rank": 0, "const_bytes_per_rank": 0, "scal_bytes_per_rank": 0}   # test / measurement hook


def all_gather_slots(buf: torch.Tensor, rank: int, group) -> None:
    """buf [world][n]: every rank wrote its own row; afterwards every rank holds all rows.  RCCL: one in-place all-gather."""
    if _backend_name(group) == "gloo":
        rows = [buf[i] for i in range(buf.shape[0])]
        dist.all_gather(rows, buf[rank].clone(), group=group)
    else:
        dist.all_gather_into_tensor(buf.view(-1), buf[rank], group=group)


class _FxState:
    def __init__(self, full_layout):
        self.v = full_layout.new_flat()


_FX = weakref.WeakKeyDictionary()


def _cg_global_factor_exchange(vector, prev, sync, provider, be, full, K: int, alpha: float, G: int, g: int, group):
    """Per iteration ONE all-gather of batch-sized factors (Rd_l, Rh_l: B x sum of widths floats per rank) and ONE of a few KB of fp64
    partials; h_l, delta_l once per solve; the right-hand side's mean once (the only N-sized collective, as in every form).  The ranks
    never exchange — and after iteration 0 never read — anything N-sized (csrc/mlp/fx.inc; protocol: tests/proj_global_ref.py)."""
    from . import _native  # noqa: PLC0415

    st = _FX.get(full)
    if st is None:
        st = _FX[full] = _FxState(full)
    be.flatten(full, vector, st.v, 1.0 / G)
    if G > 1:
        dist.all_reduce(st.v, op=dist.ReduceOp.SUM, group=group)
        FX_STATS["rhs_all_reduces"] += 1
    rhs = full.views(st.v, vector)
    state = provider._state
    bufs = state.fx_buffers(G)
    if G > 1 and not bufs.get("checked"):
        # the Gram blocks are [Bp x G Bp] with the SAME B valid rows in every rank's slot: checked once per set of buffers
        nb = torch.tensor([float(state.B), -float(state.B)], device=st.v.device if _backend_name(group) == "nccl" else "cpu")
        dist.all_reduce(nb, op=dist.ReduceOp.MAX, group=group)
        if float(nb[0]) != -float(nb[1]):
            raise ValueError(f"cg_global (factor-exchange form): every rank must hold the same batch size; got between {-int(nb[1])} and {int(nb[0])}")
        bufs["checked"] = True
    FX_STATS["solves"] += 1
    FX_STATS["slab_bytes_per_rank"] = bufs["slab"].shape[1] * 4
    FX_STATS["const_bytes_per_rank"] = bufs["const"].shape[1] * 4
    FX_STATS["scal_bytes_per_rank"] = bufs["scal"].shape[1] * 8
    gather = G > 1 or FX_ALWAYS_GATHER
    provider.cg_fx_phase(rhs, 0, K, _native.BHG_CG_FX_BEGIN, G, g, alpha)
    if gather:
        all_gather_slots(bufs["const"], g, group)
        FX_STATS["const_gathers"] += 1
    for k in range(K):
        provider.cg_fx_phase(rhs, k, K, _native.BHG_CG_FX_CHAIN, G, g, alpha)
        if gather:
            all_gather_slots(bufs["slab"], g, group)
            FX_STATS["slab_gathers"] += 1
        provider.cg_fx_phase(rhs, k, K, _native.BHG_CG_FX_GRAM, G, g, alpha)
        if gather:
            all_gather_slots(bufs["scal"], g, group)
            FX_STATS["scal_gathers"] += 1
    provider.cg_fx_phase(rhs, K - 1, K, _native.BHG_CG_FX_END, G, g, alpha)
    solve = provider.cg_fx_finish(full, K, alpha)
    provider.expects_data_parallel_mean = G > 1
    out = provider.mixed_vjp(None, sync, solve=solve)
    if sync:
        return None
    from .distributed import exchange_async  # noqa: PLC0415

    return [t.clone() for t in exchange_async(out, group).wait()]


def neumann_global(vector, curr, prev, sync, group: Optional[dist.ProcessGroup] = None):
    """``neumann`` (betty/hypergradient/neumann.py:8-66) for ONE inner problem whose batch is spread over the ranks — extension key
    ``Config(type="neumann_global")``, oracle = the reference's neumann in one process on the concatenated batch.  The series has no scalars,
    so in the factor-exchange form the ranks exchange NOTHING but the batch-sized factor slab: one all-gather per iteration, no reduction
    (csrc/mlp/fx.inc, bhg_mlp_neumann_fx_phase).  Needs a structure that takes that form (the weighted-CE MLP on the projected plan)."""
    assert len(curr.paths) == 0, "neumann method is not supported for higher order MLO!"
    assert dist.is_available() and dist.is_initialized(), "neumann_global needs an initialised process group"
    from . import _native  # noqa: PLC0415

    config = curr.config
    be = get_backend()
    vector = list(vector)
    G, g = dist.get_world_size(group), dist.get_rank(group)
    provider = structured_hvp_for(curr, prev)
    if provider is None:
        raise NotImplementedError("neumann_global needs a declared structure (hypergradient_structure): its form exchanges batch-sized factors")
    provider.pad_widths = False
    provider.prepare()
    full = be.layout(vector)
    K, alpha = int(config.neumann_iterations), float(config.neumann_alpha)
    ready = getattr(provider, "fused_neumann_fx_ready", None)
    if ready is None or K <= 0 or not ready(full, K, G):
        raise NotImplementedError("neumann_global: this structure / shape does not take the factor-exchange form (>= 3 layers, widths % 32 == 0, "
                                  "a head of <= 256 classes, no accumulator vector asked for)")
    st = _FX.get(full)
    if st is None:
        st = _FX[full] = _FxState(full)
    be.flatten(full, vector, st.v, 1.0 / G)
    if G > 1:
        dist.all_reduce(st.v, op=dist.ReduceOp.SUM, group=group)
        FX_STATS["rhs_all_reduces"] += 1
    rhs = full.views(st.v, vector)
    bufs = provider._state.fx_buffers(G)
    FX_STATS["solves"] += 1
    FX_STATS["neumann_solves"] = FX_STATS.get("neumann_solves", 0) + 1
    gather = G > 1 or FX_ALWAYS_GATHER
    provider.neumann_fx_phase(rhs, 0, K, _native.BHG_CG_FX_BEGIN, G, g, alpha)
    if gather:
        all_gather_slots(bufs["const"], g, group)
        FX_STATS["const_gathers"] += 1
    for k in range(K):
        provider.neumann_fx_phase(rhs, k, K, _native.BHG_CG_FX_CHAIN, G, g, alpha)
        if gather:
            all_gather_slots(bufs["slab"], g, group)
            FX_STATS["slab_gathers"] += 1
        provider.neumann_fx_phase(rhs, k, K, _native.BHG_CG_FX_GRAM, G, g, alpha)
    provider.neumann_fx_phase(rhs, K, K, _native.BHG_CG_FX_CHAIN, G, g, alpha)   # the closing half pass: Rz(v_K)
    provider.neumann_fx_phase(rhs, K, K, _native.BHG_CG_FX_END, G, g, alpha)
    solve = provider.neumann_fx_finish(full, K, alpha)
    provider.expects_data_parallel_mean = G > 1
    out = provider.mixed_vjp(None, sync, solve=solve)
    if sync:
        return None
    from .distributed import exchange_async  # noqa: PLC0415

    return [t.clone() for t in exchange_async(out, group).wait()]


def cg_global(vector, curr, prev, sync, group: Optional[dist.ProcessGroup] = None):
    """Same signature and result convention as ``cg`` (betty/hypergradient/cg.py:8-70); ``vector`` is this rank's
    gradient of ITS share of the upper loss (the global one is the mean over ranks).  Returns the GLOBAL hypergradient
    (identical on all ranks) when ``sync`` is False."""
    assert len(curr.paths) == 0, "cg method is not supported for higher order MLO!"
    assert dist.is_available() and dist.is_initialized(), "cg_global needs an initialised process group"
    config = curr.config
    be = get_backend()
    vector = list(vector)
    G, g = dist.get_world_size(group), dist.get_rank(group)

    provider = structured_hvp_for(curr, prev)
    if provider is not None:
        provider.pad_widths = False   # the ranks exchange the real network's N-sized state between phases: no padded twin here
    if provider is None:
        in_grad = inner_gradient(curr)
        hvp_fn = AutogradHVP(in_grad, curr.parameters())
    else:
        in_grad = None
        hvp_fn = provider.prepare()
    shift = float(getattr(provider, "hvp_shift", 0.0)) if provider is not None else 0.0

    full = be.layout(vector)
    K = int(config.cg_iterations)
    alpha = float(config.cg_alpha)
    fx_ready = getattr(provider, "fused_cg_fx_ready", None)
    if GLOBAL_FORM == "auto" and fx_ready is not None and K > 0 and alpha != 0.0 and fx_ready(full, K, G):
        return _cg_global_factor_exchange(vector, prev, sync, provider, be, full, K, alpha, G, g, group)
    ready = getattr(provider, "fused_cg_global_ready", None)
    if GLOBAL_FORM != "sharded" and ready is not None and K > 0 and alpha != 0.0 and ready(full, K):
        return _cg_global_one_pass(vector, prev, sync, provider, be, full, K, alpha, G, group)
    key = (id(full), G)
    st = _STATES.get(key)
    if st is None:
        st = _STATES[key] = _State(be, full, G)
    sl = st.shard_layout

    # v = mean over ranks of the local vectors; every rank keeps its slice
    be.flatten(full, vector, st.h_full, 1.0 / G)
    reduce_scatter_sum(st.h_full, st.hs, g, group)
    be.cg_init(sl, [st.hs], st.x, st.r, st.p)                      # x = 0, r = p = v_slice, partials of r.r
    dist.all_reduce(be.cg_partials(sl, 2, 0), op=dist.ReduceOp.SUM, group=group)
    all_gather_flat(st.p, st.p_full, group)
    p_views = full.views(st.p_full, vector)

    for k in range(K):
        hvp = hvp_fn(p_views)                                        # H_local p (cg.py:39-41), this rank's batch
        be.flatten(full, hvp, st.h_full, 1.0 / G)
        reduce_scatter_sum(st.h_full, st.hs, g, group)               # slice of the MEAN Hessian-vector product
        last = k == K - 1 and alpha != 0.0
        args = (sl, [st.hs], st.x, st.r, st.p, alpha, k, (-alpha if last else 0.0), shift)
        be.cg_phase(0, *args)                                        # partials of (cg_alpha*Hp).p over the slice
        dist.all_reduce(be.cg_partials(sl, 0, k), op=dist.ReduceOp.SUM, group=group)
        be.cg_phase(1, *args)                                        # alpha; r' = r - alpha*Hp; partials of r'.r'
        dist.all_reduce(be.cg_partials(sl, 1, k), op=dist.ReduceOp.SUM, group=group)
        be.cg_phase(2, *args)                                        # beta; x += alpha*p; p = r' + beta*p
        if k + 1 < K:
            all_gather_flat(st.p, st.p_full, group)
    if K > 0 and alpha == 0.0:
        be.scale_flat(st.x, -alpha)

    all_gather_flat(st.x, st.x_full, group)
    neg_x = full.views(st.x_full, vector)
    if provider is not None:
        provider.expects_data_parallel_mean = G > 1
    out = provider.mixed_vjp(neg_x, sync) if provider is not None else mixed_vjp(in_grad, prev, neg_x, sync)
    if sync:
        return None          # accumulated through backward(): the DDP reducer of prev's module averaged it
    from .distributed import exchange_async  # noqa: PLC0415

    return [t.clone() for t in exchange_async(out, group).wait()]
