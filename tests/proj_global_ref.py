"""Test infrastructure (NOT product code; nothing under betty_amd/ imports it): the protocol of a GLOBAL-batch fully projected CG
solve for the weighted-CE ReLU-MLP structure in which the ranks exchange BATCH-SIZED FACTORS instead of the N-sized residual
(round-3 VERDICT item 8), restated in ATen so that its math and its communication pattern can be checked on CPU — gloo, world
size 2 / 4 — against the reference's cg() run in ONE process on the concatenated batch (cg.py:25-68).  The HIP kernels of this
form exist since round 6 (csrc/mlp/fx.inc, bhg_mlp_cg_fx_phase; DESIGN 3.22 — tested in tests/test_gpu_global.py); the kernels of the
one-rank projected solver implement the same recurrences with G = 1.

Setting.  G ranks, rank g holds B_g samples; the inner loss is the mean over the ranks of the local weighted-CE means, so
H = mean_g H_g (+ shift I), exactly the global-batch mode of betty_amd/global_hvp.py.  Every weight-shaped output of H_g v is an
outer product of batch-sized factors:

    (H_g v)(W_l) = Rd_l^T h_l + delta_l^T Rh_{l-1}        (H_g v)(b_l) = colsum_b Rd_l            (all factors: rank g's samples)

so a vector u of the Krylov space is known to the chain of rank g through its PRODUCTS WITH RANK g's BATCH only,

    Gf_l(u) = h_l U_l^T   (B_g x d_{l+1})        Gb_l(u) = delta_l U_l   (B_g x d_l)        l over the wide ("MFMA") layers,

and these obey recurrences whose operator part needs the OTHER ranks' factors, not their N-sized outputs:

    Gf_l^{(g)}(H v) = mean_g' [ (h_l^{(g)} h_l^{(g')T}) Rd_l^{(g')} + (h_l^{(g)} Rh_{l-1}^{(g')T}) delta_l^{(g')} ]
    Gb_l^{(g)}(H v) = mean_g' [ (delta_l^{(g)} Rd_l^{(g')T}) h_l^{(g')} + (delta_l^{(g)} delta_l^{(g')T}) Rh_{l-1}^{(g')} ]

Per iteration the ranks ALL-GATHER their factors (Rd_l, Rh_l: B_g x (sum of widths) floats — ~3 MB per rank at cfg 2, against the
40 MB residual of the one-pass form) and ALL-REDUCE a handful of fp64 scalars (their shares of p.Hp, r.Hp, Hp.Hp on the wide
layers); h_l and delta_l of all ranks are gathered ONCE per solve.  The narrow slices (biases, head weight) stay explicit and
replicated: their outputs are sums over all samples of the gathered factors, computed redundantly by every rank.  No rank reads
or writes an N-sized vector after the projections of the right-hand side in iteration 0.
"""
import torch


class Comm:
    """all_gather along dim 0 / all_reduce(SUM) — a torch.distributed group, or a list of emulated ranks in one process."""

    def __init__(self, world):
        self.world = world
        self.bytes_gathered_per_rank = 0
        self.gathers = 0
        self.scalar_reduces = 0

    def all_gather(self, t):
        raise NotImplementedError

    def all_reduce_sum(self, t):
        raise NotImplementedError


class DistComm(Comm):
    def __init__(self):
        import torch.distributed as dist

        super().__init__(dist.get_world_size())
        self.dist = dist

    def all_gather(self, t):
        t = t.contiguous()
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        self.gathers += 1
        self.bytes_gathered_per_rank += t.numel() * t.element_size()
        return torch.cat(out, 0)

    def all_reduce_sum(self, t):
        t = t.clone()
        self.dist.all_reduce(t)
        self.scalar_reduces += 1
        return t


def local_state(Ws, bs, x, y, w):
    """Forward / first backward of rank g's share (cg.py:27-32 for this structure): h_l, masks, softmax, sd = w / B_g, delta_l."""
    B = x.shape[0]
    L = len(Ws)
    hs, masks, h = [x], [], x
    for l in range(L):
        a = h @ Ws[l].t() + bs[l]
        if l + 1 < L:
            m = (a > 0).to(a.dtype)
            h = a * m
            masks.append(m)
            hs.append(h)
        else:
            z = a
    p = torch.softmax(z, 1)
    onehot = torch.nn.functional.one_hot(y, z.shape[1]).to(z.dtype)
    sd = w / B
    deltas = [None] * L
    deltas[-1] = sd[:, None] * (p - onehot)
    for l in range(L - 1, 0, -1):
        deltas[l - 1] = masks[l - 1] * (deltas[l] @ Ws[l])
    return {"hs": hs, "masks": masks, "p": p, "sd": sd, "deltas": deltas, "err": p - onehot, "B": B, "L": L}


def r_chain(st, Ws, Gf, Gb, cs, V_head):
    """The R-chain of rank g on products-with-the-batch: Gf[l] = h_l U_l^T (l < L-1), Gb[l] = delta_l U_l (1 <= l < L-1), cs = the
    direction's bias slices, V_head = its head weight.  Returns Rz, [Rh_0 .. Rh_{L-2}], [Rd_0 .. Rd_{L-1}]."""
    L = st["L"]
    Rhs, Rh = [], None
    for l in range(L - 1):
        Ra = Gf[l] + cs[l]
        if Rh is not None:
            Ra = Ra + Rh @ Ws[l].t()
        Rh = st["masks"][l] * Ra
        Rhs.append(Rh)
    Rz = st["hs"][L - 1] @ V_head.t() + cs[L - 1] + Rh @ Ws[L - 1].t()
    p = st["p"]
    Rd = st["sd"][:, None] * (p * Rz - p * (p * Rz).sum(1, keepdim=True))
    Rds = [None] * L
    Rds[L - 1] = Rd
    for l in range(L - 1, 0, -1):
        G = st["deltas"][l] @ V_head if l == L - 1 else Gb[l]
        Rd = st["masks"][l - 1] * (G + Rd @ Ws[l])
        Rds[l - 1] = Rd
    return Rz, Rhs, Rds


def solve(Ws, bs, x, y, w, vec, K, cg_alpha, shift, comm, rank_rows=None):
    """Fully projected global-batch CG on rank g's samples.  vec: the right-hand side as [V_0, c_0, V_1, c_1, ...] (replicated).
    Returns Rz(x) on rank g's samples for x = -cg_alpha * (CG solution) — what the mixed second derivative needs (cg.py:58-68 for
    this structure: coefficient_b = err_b . Rz(x)_b / B) — and the narrow slices of x."""
    st = local_state(Ws, bs, x, y, w)
    L, G = st["L"], comm.world
    wide = range(L - 1)
    # ---- once per solve: every rank's activations and back-propagated errors (constants of the solve)
    h_all = [comm.all_gather(st["hs"][l]) for l in range(L)]
    d_all = [comm.all_gather(st["deltas"][l]) for l in range(L)]
    # narrow slices, explicit and replicated: biases of all layers, head weight
    r_c = [vec[2 * l + 1].clone() for l in range(L)]
    r_V = vec[2 * (L - 1)].clone()
    p_c = [t.clone() for t in r_c]
    p_V = r_V.clone()
    x_c = [torch.zeros_like(t) for t in r_c]
    x_V = torch.zeros_like(r_V)
    # ---- iteration 0: the ONLY N-sized reads — projections of the right-hand side on this rank's batch (p = r)
    Gf_r = [st["hs"][l] @ vec[2 * l].t() for l in wide]
    Gb_r = [None] + [st["deltas"][l] @ vec[2 * l] for l in range(1, L - 1)]
    Gf_p = [t.clone() for t in Gf_r]
    Gb_p = [None] + [t.clone() for t in Gb_r[1:]]
    dd = lambda a, b: (a.double() * b.double()).sum()
    rr = sum(dd(vec[i], vec[i]) for i in range(2 * L))   # r.r of the replicated right-hand side (computed identically everywhere)
    rr_w = sum(dd(vec[2 * l], vec[2 * l]) for l in wide)  # ... its share on the wide layers: carried by scalar recurrences below
    rp_w, pp_w = rr_w.clone(), rr_w.clone()
    Rzx = torch.zeros(st["B"], Ws[-1].shape[0], dtype=x.dtype)
    for k in range(K):
        Rz, Rhs, Rds = r_chain(st, Ws, Gf_p, Gb_p, p_c, p_V)
        # ---- the exchange: batch-sized factors of every rank (ONE all-gather per iteration)
        widths = [t.shape[1] for t in Rds] + [t.shape[1] for t in Rhs]
        packed = comm.all_gather(torch.cat(Rds + Rhs, 1))
        parts = list(torch.split(packed, widths, 1))
        Rd_all, Rh_all = parts[:L], parts[L:]
        # ---- G(raw) on this rank's samples: Gram blocks (this rank's rows x all samples) times the gathered factors; mean over ranks
        Gf_raw, Gb_raw = [], [None]
        for l in wide:
            t = (st["hs"][l] @ h_all[l].t()) @ Rd_all[l]
            if l >= 1:
                t = t + (st["hs"][l] @ Rh_all[l - 1].t()) @ d_all[l]
            Gf_raw.append(t / G)
        for l in range(1, L - 1):
            t = (st["deltas"][l] @ Rd_all[l].t()) @ h_all[l] + (st["deltas"][l] @ d_all[l].t()) @ Rh_all[l - 1]
            Gb_raw.append(t / G)
        # ---- narrow slices: outputs from ALL samples' factors, identical on every rank
        raw_c = [Rd_all[l].sum(0) / G for l in range(L)]
        raw_V = (Rd_all[L - 1].t() @ h_all[L - 1] + d_all[L - 1].t() @ Rh_all[L - 2]) / G
        # ---- inner products on the wide layers: this rank's share (its samples), summed over the ranks (ONE small all-reduce)
        #      u . raw = mean_g' sum_l <Rd_l^{(g')}, Gf_l^{(g')}(u)> + <Rh_{l-1}^{(g')}, Gb_l^{(g')}(u)>
        def share(Gf_u, Gb_u):
            s = sum(dd(Rds[l], Gf_u[l]) for l in wide)
            return s + sum(dd(Rhs[l - 1], Gb_u[l]) for l in range(1, L - 1))
        loc = torch.stack([share(Gf_r, Gb_r), share(Gf_p, Gb_p), share(Gf_raw, Gb_raw)])
        r_raw, p_raw, raw_raw = (comm.all_reduce_sum(loc) / G).tolist()
        # narrow slices' inner products (replicated data: no communication)
        p_raw_n = sum(dd(p_c[l], raw_c[l]) for l in range(L)) + dd(p_V, raw_V)
        pp_n = sum(dd(p_c[l], p_c[l]) for l in range(L)) + dd(p_V, p_V)
        pp = pp_w + pp_n
        den = cg_alpha * ((p_raw + p_raw_n) + shift * pp)      # cg.py:42-46 with the cg_alpha quirk
        alpha = float(rr / den)
        # ---- cg.py:49-50: x += alpha p (only Rz(x) and the narrow slices are kept); r -= alpha H p
        Rzx += alpha * Rz
        for l in range(L):
            x_c[l] += alpha * p_c[l]
            r_c[l] -= alpha * (raw_c[l] + shift * p_c[l])
        x_V += alpha * p_V
        r_V -= alpha * (raw_V + shift * p_V)
        for l in wide:
            Gf_r[l] = Gf_r[l] - alpha * (Gf_raw[l] + shift * Gf_p[l])
        for l in range(1, L - 1):
            Gb_r[l] = Gb_r[l] - alpha * (Gb_raw[l] + shift * Gb_p[l])
        # scalar recurrences on the wide layers: r'.r', r'.p
        rHp = r_raw + shift * rp_w
        pHp = p_raw + shift * pp_w
        HpHp = raw_raw + 2.0 * shift * p_raw + shift * shift * pp_w
        rr_w1 = rr_w - 2.0 * alpha * rHp + alpha * alpha * HpHp
        rp_w1 = rp_w - alpha * pHp
        rr_n1 = sum(dd(r_c[l], r_c[l]) for l in range(L)) + dd(r_V, r_V)
        rr_new = rr_w1 + rr_n1
        if k == K - 1:
            break
        beta = float(rr_new / rr)                               # cg.py:51-53
        for l in range(L):
            p_c[l] = r_c[l] + beta * p_c[l]
        p_V = r_V + beta * p_V
        for l in wide:
            Gf_p[l] = Gf_r[l] + beta * Gf_p[l]
        for l in range(1, L - 1):
            Gb_p[l] = Gb_r[l] + beta * Gb_p[l]
        pp_w = rr_w1 + 2.0 * beta * rp_w1 + beta * beta * pp_w
        rp_w = rr_w1 + beta * rp_w1
        rr_w, rr = rr_w1, rr_new
    scale = -cg_alpha                                            # cg.py:56 and the negation of cg.py:59/68
    return Rzx * scale, [t * scale for t in x_c], x_V * scale


def oracle(Ws, bs, xs, ys, ws, vec, K, cg_alpha, shift):
    """The reference's algorithm in ONE process (cg.py:34-56) on the global-batch operator H = mean_g H_g + shift I with explicit
    N-sized vectors; returns Rz(x) per rank's samples, the narrow slices of x, and x itself."""
    sts = [local_state(Ws, bs, x, y, w) for x, y, w in zip(xs, ys, ws)]
    G, L = len(sts), len(Ws)

    def hvp(v):
        out = [torch.zeros_like(t) for t in v]
        for st in sts:
            Gf = [st["hs"][l] @ v[2 * l].t() for l in range(L - 1)]
            Gb = [None] + [st["deltas"][l] @ v[2 * l] for l in range(1, L - 1)]
            Rz, Rhs, Rds = r_chain(st, Ws, Gf, Gb, v[1::2], v[2 * (L - 1)])
            for l in range(L):
                HW = Rds[l].t() @ st["hs"][l]
                if l >= 1:
                    HW = HW + st["deltas"][l].t() @ Rhs[l - 1]
                out[2 * l] += HW / G
                out[2 * l + 1] += Rds[l].sum(0) / G
        return [o + shift * t for o, t in zip(out, v)]

    dot = lambda a, b: sum((s.double() * t.double()).sum() for s, t in zip(a, b))
    x = [torch.zeros_like(t) for t in vec]
    r = [t.clone() for t in vec]
    p = [t.clone() for t in vec]
    rr = dot(r, r)
    for k in range(K):
        Hp = hvp(p)
        alpha = float(rr / (cg_alpha * dot(p, Hp)))
        x = [a + alpha * b for a, b in zip(x, p)]
        r = [a - alpha * b for a, b in zip(r, Hp)]
        rr_new = dot(r, r)
        beta = float(rr_new / rr)
        p = [a + beta * b for a, b in zip(r, p)]
        rr = rr_new
    x = [-cg_alpha * t for t in x]
    rz = []
    for st in sts:
        Gf = [st["hs"][l] @ x[2 * l].t() for l in range(L - 1)]
        Gb = [None] + [st["deltas"][l] @ x[2 * l] for l in range(1, L - 1)]
        rz.append(r_chain(st, Ws, Gf, Gb, x[1::2], x[2 * (L - 1)])[0])
    return rz, x[1::2], x[2 * (L - 1)], x


def make_problem(dims, B_total, seed, dtype=torch.float64):
    """Drawn in fp64 whatever `dtype` (the same instance in both precisions), then cast."""
    g = torch.Generator().manual_seed(seed)
    L = len(dims) - 1
    f64 = torch.float64
    Ws = [torch.randn(dims[l + 1], dims[l], generator=g, dtype=f64) / dims[l] ** 0.5 for l in range(L)]
    bs = [0.1 * torch.randn(dims[l + 1], generator=g, dtype=f64) for l in range(L)]
    x = torch.randn(B_total, dims[0], generator=g, dtype=f64)
    y = torch.randint(0, dims[-1], (B_total,), generator=g)
    w = torch.rand(B_total, generator=g, dtype=f64) + 0.5
    vec = []
    for l in range(L):
        vec += [torch.randn(dims[l + 1], dims[l], generator=g, dtype=f64), torch.randn(dims[l + 1], generator=g, dtype=f64)]
    c = lambda t: t.to(dtype)
    return [c(t) for t in Ws], [c(t) for t in bs], c(x), y, c(w), [c(t) for t in vec]

class ThreadComm(Comm):
    """Emulated ranks: one thread per rank of ONE process, rendezvous on a barrier (no torch.distributed)."""

    class Shared:
        def __init__(self, world):
            import threading

            self.world = world
            self.slots = [None] * world
            self.barrier = threading.Barrier(world)

    def __init__(self, shared, rank):
        super().__init__(shared.world)
        self.sh, self.rank = shared, rank

    def _exchange(self, t):
        self.sh.slots[self.rank] = t
        self.sh.barrier.wait()
        got = list(self.sh.slots)
        self.sh.barrier.wait()
        return got

    def all_gather(self, t):
        self.gathers += 1
        self.bytes_gathered_per_rank += t.numel() * t.element_size()
        return torch.cat(self._exchange(t.contiguous()), 0)

    def all_reduce_sum(self, t):
        self.scalar_reduces += 1
        return torch.stack(self._exchange(t.clone())).sum(0)


def run_emulated(dims, B_total, world, K, cg_alpha, shift, seed, dtype=torch.float64):
    """world emulated ranks on equal shares of one problem; returns (per-rank results, oracle results, per-rank Comm)."""
    import threading

    Ws, bs, x, y, w, vec = make_problem(dims, B_total, seed, dtype)
    share = B_total // world
    xs = [x[g * share:(g + 1) * share] for g in range(world)]
    ys = [y[g * share:(g + 1) * share] for g in range(world)]
    ws = [w[g * share:(g + 1) * share] for g in range(world)]
    shared = ThreadComm.Shared(world)
    comms = [ThreadComm(shared, g) for g in range(world)]
    out = [None] * world

    def work(g):
        out[g] = solve(Ws, bs, xs[g], ys[g], ws[g], vec, K, cg_alpha, shift, comms[g])

    ts = [threading.Thread(target=work, args=(g,)) for g in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return out, oracle(Ws, bs, xs, ys, ws, vec, K, cg_alpha, shift), comms
