"""The algebra behind libbhg's projected solvers (csrc/bhg_mlp.hip: k_hoist / k_wsk_group / k_proj_update / k_proj_scalars), checked
on the CPU in fp64 where it must hold to rounding: for a ReLU-MLP with per-sample-weighted cross-entropy (+ ridge),

  * the direction products G(.) = h V^T / delta V of the residual obey batch-sized recurrences through B x B Gram matrices,
        Gf_l(raw) = (h_l h_l^T) Rd_l + (h_l Rh_{l-1}^T) delta_l ,   Gb_l(raw) = (delta_l Rd_l^T) h_l + (delta_l delta_l^T) Rh_{l-1}
    because the weight-shaped outputs of H p are outer products of batch-sized factors, raw(W_l) = Rd_l^T h_l + delta_l^T Rh_{l-1};
  * r'.r', r'.p and p'.p' follow from  r.raw = sum <Rd_l, Gf_l(r)> + <Rh_{l-1}, Gb_l(r)>,  p.raw alike, and
    raw.raw = sum <Rd_l, Gf_l(raw)> + <Rh_{l-1}, Gb_l(raw)>  (= sum <Rd_l Rd_l^T, S_l> + 2 <E_l^T, T_l> + <D_l, Rh_{l-1} Rh_{l-1}^T>; small slices explicit);
  * so CG (cg.py:34-56, including the cg_alpha quirk) and the Neumann series (neumann.py:59-66) can run WITHOUT the N-sized
    residual / direction after the first iteration and give the reference's hypergradient.

The reference here is the oracle's restatement of the reference's functions (bit-pinned to the reference elsewhere)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))

import hypergrad_oracle as orc  # noqa: E402
import zoo  # noqa: E402
from betty_amd import Config  # noqa: E402


def _problem(dims, B, ridge, algo, K, alpha, seed=0):
    g = torch.Generator().manual_seed(seed)
    inner, upper = zoo.MLP(dims).double(), zoo.MWN(6).double()
    with torch.no_grad():
        for p in list(inner.parameters()) + list(upper.parameters()):
            p.copy_(torch.randn(p.shape, generator=g, dtype=torch.float64) / max(p.shape[-1], 4) ** 0.5)
    x = torch.randn(B, dims[0], generator=g, dtype=torch.float64)
    y = torch.randint(0, dims[-1], (B,), generator=g)
    prev = zoo.StubProblem("upper", upper, config=Config())
    cfg = Config(type=algo, cg_iterations=K, cg_alpha=alpha, neumann_iterations=K, neumann_alpha=alpha)
    curr = zoo.StubProblem("inner", inner, config=cfg, loss_fn=zoo.make_reweight_loss(prev, ridge), batch=(x, y))
    vector = [0.1 * torch.randn(p.shape, generator=g, dtype=torch.float64) for p in inner.parameters()]
    return curr, prev, vector


def _projected(curr, prev, vector, ridge, algo, K, alpha):
    """The projected solver, line by line what the kernels do (fp64 here)."""
    x, y = curr.cur_batch
    lins = list(curr.module.layers)
    Ws, bs = [l.weight.detach() for l in lins], [l.bias.detach() for l in lins]
    L, B = len(Ws), x.shape[0]
    hs, masks, h = [x], [], x
    for l in range(L):
        a = torch.addmm(bs[l], h, Ws[l].t())
        if l + 1 < L:
            m = (a > 0).double(); h = a * m; masks.append(m); hs.append(h)
    prob = F.softmax(a, 1)
    ce = -F.log_softmax(a, 1).gather(1, y.reshape(-1, 1)).reshape(-1)
    sw = prev.fwd(ce.detach().reshape(-1, 1)).reshape(-1)
    sd = sw.detach() / B
    err = prob - F.one_hot(y, a.shape[1]).double()
    dl = [None] * L
    dl[-1] = sd[:, None] * err
    for l in range(L - 1, 0, -1):
        dl[l - 1] = masks[l - 1] * (dl[l] @ Ws[l])
    shift = 2 * ridge
    V0, c0 = vector[0::2], vector[1::2]
    dd = lambda a_, b_: (a_ * b_).sum()
    # explicit small slices (biases, head weight); projections of the MFMA layers' weight slices
    cr, cp = [v.clone() for v in c0], [v.clone() for v in c0]
    Vhr, Vhp = V0[L - 1].clone(), V0[L - 1].clone()
    Gfr = [hs[l] @ V0[l].t() for l in range(L - 1)]
    Gbr = [None] + [dl[l] @ V0[l] for l in range(1, L - 1)]
    Gfp, Gbp = [g.clone() for g in Gfr], [None] + [g.clone() for g in Gbr[1:]]
    S = [hs[l] @ hs[l].t() for l in range(L - 1)]
    D = [None] + [dl[l] @ dl[l].t() for l in range(1, L - 1)]
    rr = sum(dd(v, v) for v in vector)
    rp, pp = rr.clone(), rr.clone()
    Rzx = torch.zeros(B, a.shape[1], dtype=torch.float64)
    n_pass = K if algo == "cg" else K + 1          # Neumann: the closing half pass adds Rz(v_K)
    for k in range(n_pass):
        Rh = [None] * L
        Rh[0] = masks[0] * (Gfp[0] + cp[0])
        for l in range(1, L - 1):
            Rh[l] = masks[l] * (Rh[l - 1] @ Ws[l].t() + Gfp[l] + cp[l])
        Rz = Rh[L - 2] @ Ws[L - 1].t() + hs[L - 1] @ Vhp.t() + cp[L - 1]
        if algo == "neumann":
            Rzx += Rz
            if k == K:
                break
        Rd = [None] * L
        Rd[L - 1] = sd[:, None] * (prob * Rz - prob * (prob * Rz).sum(1, keepdim=True))
        Rd[L - 2] = masks[L - 2] * (dl[L - 1] @ Vhp + Rd[L - 1] @ Ws[L - 1])
        for l in range(L - 2, 0, -1):
            Rd[l - 1] = masks[l - 1] * (Rd[l] @ Ws[l] + Gbp[l])
        raw_c = [Rd[l].sum(0) for l in range(L)]
        raw_Vh = Rd[L - 1].t() @ hs[L - 1] + dl[L - 1].t() @ Rh[L - 2]
        Tm = [None] + [hs[l] @ Rh[l - 1].t() for l in range(1, L - 1)]
        Em = [None] + [dl[l] @ Rd[l].t() for l in range(1, L - 1)]
        graw_f = [S[l] @ Rd[l] + (Tm[l] @ dl[l] if l > 0 else 0) for l in range(L - 1)]
        graw_b = [None] + [Em[l] @ hs[l] + D[l] @ Rh[l - 1] for l in range(1, L - 1)]
        if algo == "cg":
            T1 = dd(Rz, Rd[L - 1])
            T2 = 2 * dd(dl[L - 1] @ Vhp, Rh[L - 2]) + sum(2 * dd(Gbp[l], Rh[l - 1]) for l in range(1, L - 1))
            pHp = T1 + T2 + shift * pp
            a_k = rr / (alpha * pHp)                        # cg.py:42-47 (step length uses cg_alpha * Hp ...)
            Rzx += a_k * Rz
            r_raw = sum(dd(Rd[l], Gfr[l]) for l in range(L - 1)) + sum(dd(Rh[l - 1], Gbr[l]) for l in range(1, L - 1)) \
                + sum(dd(cr[l], raw_c[l]) for l in range(L)) + dd(Vhr, raw_Vh)
            # <raw_l, raw_l> = <raw_l, Rd_l^T h_l> + <raw_l, delta_l^T Rh_{l-1}> = <Rd_l, Gf_l(raw)> + <Rh_{l-1}, Gb_l(raw)>:
            # the same form as r.raw, with the products the iteration forms anyway (the tiles of the G(raw) launch emit it)
            raw_raw = sum(dd(Rd[l], graw_f[l]) for l in range(L - 1)) + sum(dd(Rh[l - 1], graw_b[l]) for l in range(1, L - 1)) \
                + sum(dd(raw_c[l], raw_c[l]) for l in range(L)) + dd(raw_Vh, raw_Vh)
            gram_route = sum(dd(Rd[l] @ Rd[l].t(), S[l]) for l in range(L - 1)) \
                + sum(2 * dd(Em[l].t(), Tm[l]) + dd(D[l], Rh[l - 1] @ Rh[l - 1].t()) for l in range(1, L - 1)) \
                + sum(dd(raw_c[l], raw_c[l]) for l in range(L)) + dd(raw_Vh, raw_Vh)
            assert abs(float(raw_raw - gram_route)) <= 1e-9 * abs(float(gram_route)) + 1e-300   # round 3's first form: Gram matrices
            p_raw = pHp - shift * pp
            r_Hp, Hp_Hp = r_raw + shift * rp, raw_raw + 2 * shift * p_raw + shift * shift * pp
            rr_new = rr - 2 * a_k * r_Hp + a_k * a_k * Hp_Hp          # (... the residual update the un-scaled Hp: cg.py:50)
            rp_mid = rp - a_k * pHp
            beta = rr_new / rr
        else:
            a_k, beta = alpha, None
        for l in range(L - 1):
            Gfr[l] = Gfr[l] - a_k * (graw_f[l] + shift * Gfp[l])
        for l in range(1, L - 1):
            Gbr[l] = Gbr[l] - a_k * (graw_b[l] + shift * Gbp[l])
        for l in range(L):
            cr[l] = cr[l] - a_k * (raw_c[l] + shift * cp[l])
        Vhr = Vhr - a_k * (raw_Vh + shift * Vhp)
        if algo == "cg":
            pp = rr_new + 2 * beta * rp_mid + beta * beta * pp
            rp = rr_new + beta * rp_mid
            rr = rr_new
            Gfp = [Gfr[l] + beta * Gfp[l] for l in range(L - 1)]
            Gbp = [None] + [Gbr[l] + beta * Gbp[l] for l in range(1, L - 1)]
            cp = [cr[l] + beta * cp[l] for l in range(L)]
            Vhp = Vhr + beta * Vhp
        else:   # the Neumann direction IS the updated vector
            Gfp, Gbp, cp, Vhp = [g.clone() for g in Gfr], [None] + [g.clone() for g in Gbr[1:]], [c.clone() for c in cr], Vhr.clone()
    coeff = -alpha * ((err * Rzx).sum(1) / B)       # x_final = -alpha * sum_k a_k p_k (cg.py:56,59) / -alpha * sum_k v_k (neumann.py:66)
    return list(torch.autograd.grad(sw, list(prev.trainable_parameters()), grad_outputs=coeff))


@pytest.mark.parametrize("algo,K,alpha", [("cg", 1, 1.0), ("cg", 6, 1.0), ("cg", 5, 0.5), ("neumann", 1, 0.05), ("neumann", 7, 0.05)])
@pytest.mark.parametrize("dims,B,ridge", [([12, 16, 8, 5], 7, 0.3), ([9, 10, 11, 12, 4], 13, 0.5), ([8, 9, 7, 3], 6, 0.1)])
def test_projected_recurrences_reproduce_the_reference_algorithm(dims, B, ridge, algo, K, alpha):
    curr, prev, vector = _problem(dims, B, ridge, algo, K, alpha)
    want = orc.JVP_FNS[algo](vector, curr, prev, False)
    got = _projected(curr, prev, vector, ridge, algo, K, alpha)
    num = sum(((a - b) ** 2).sum() for a, b in zip(got, want)) ** 0.5
    den = sum((b ** 2).sum() for b in want) ** 0.5
    assert float(num / den) <= 1e-9, float(num / den)
