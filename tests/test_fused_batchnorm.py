"""betty_amd.nn.FusedBatchNorm2d: batch norm whose share of a Hessian-vector product (cg.py:39-41, neumann.py:62 — the double backward)
is one fused call (csrc/bhg_bn.hip) instead of ATen's decomposition.

CPU: the closed form (tests/bn_ref.py, the checker) against autograd's own double backward in float64; the two autograd.Function
nodes wired as the package wires them, with the checker standing in for the HIP call — gradient, Hessian-vector product and a whole CG
hypergradient equal to nn.BatchNorm2d's; the module's bookkeeping (running statistics, eval mode, state_dict, affine = False).
GPU: the HIP kernels against the checker on the layer shapes of ResNet-12 (vector and scalar access paths, NULL cotangents), bit
reproducibility, and the BASELINE cfg 3 solve with declared layers against the same solve with nn.BatchNorm2d."""
import contextlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import bn_ref  # noqa: E402
import zoo  # noqa: E402
from conftest import rel_err  # noqa: E402

from betty_amd import Config  # noqa: E402
from betty_amd import hypergradient as hg  # noqa: E402
from betty_amd import nn as bnn  # noqa: E402


@pytest.fixture()
def checker():
    """Host orchestration against the CPU checker backend (tests/_cpu_checker_backend.py), as tests/test_host_logic.py does."""
    from _cpu_checker_backend import CpuCheckerBackend

    from betty_amd.backend import use_backend

    with use_backend(CpuCheckerBackend()) as b:
        yield b


@contextlib.contextmanager
def checker_vjp():
    prev = bnn._VJP_IMPL[0]
    bnn._VJP_IMPL[0] = bn_ref.bn_backward_vjp
    try:
        yield
    finally:
        bnn._VJP_IMPL[0] = prev


def _stats(x, eps=1e-5):
    dims = [0, 2, 3]
    mean = x.mean(dims)
    return mean, (x.var(dims, unbiased=False) + eps).rsqrt()


@pytest.mark.parametrize("affine", [True, False])
def test_closed_form_matches_autograds_double_backward_fp64(affine):
    torch.manual_seed(0)
    N, C, H, W = 5, 4, 3, 7
    x = torch.randn(N, C, H, W, dtype=torch.float64, requires_grad=True)
    gy = torch.randn(N, C, H, W, dtype=torch.float64, requires_grad=True)
    gamma = torch.randn(C, dtype=torch.float64, requires_grad=True) if affine else None
    beta = torch.zeros(C, dtype=torch.float64, requires_grad=True) if affine else None
    y = F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5)
    ins = (x, gamma, beta) if affine else (x,)
    first = torch.autograd.grad(y, ins, gy, create_graph=True)
    a = torch.randn_like(x)
    b, c = (torch.randn(C, dtype=torch.float64), torch.randn(C, dtype=torch.float64)) if affine else (None, None)
    phi = (first[0] * a).sum() + ((first[1] * b).sum() + (first[2] * c).sum() if affine else 0.0)
    want = torch.autograd.grad(phi, (x, gy) + ((gamma,) if affine else ()))
    mean, invstd = _stats(x.detach())
    got = bn_ref.bn_backward_vjp(x.detach(), gy.detach(), a, gamma.detach() if affine else None, mean, invstd, b, c)
    for w, g in zip(want, got):
        assert float((w - g).abs().max()) <= 1e-12 * max(1.0, float(w.abs().max()))
    assert affine or got[2] is None


class _Net(torch.nn.Module):
    def __init__(self, bn_cls, affine=True):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 6, 3, padding=1, bias=False)
        self.b1 = bn_cls(6, affine=affine)
        self.c2 = torch.nn.Conv2d(6, 4, 3, padding=1, bias=False)
        self.b2 = bn_cls(4, affine=affine)
        self.fc = torch.nn.Linear(4, 3)

    def forward(self, x):
        h = F.leaky_relu(self.b1(self.c1(x)), 0.1)
        h = F.leaky_relu(self.b2(self.c2(h)) + 0.5 * self.c2(h), 0.1)
        return self.fc(h.mean((2, 3)))


class _AlwaysFused(bnn.FusedBatchNorm2d):
    """The package's two autograd nodes on CPU float64 tensors (the module itself only takes that route for CUDA fp32 inputs)."""

    def _fusable(self, x):
        return self.training


def _problem(bn_cls, affine, algo="cg", K=4):
    torch.manual_seed(3)
    inner = _Net(bn_cls, affine).double()
    upper = _Net(torch.nn.BatchNorm2d, affine).double()
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(6, 3, 5, 5, generator=g, dtype=torch.float64), torch.randint(0, 3, (6,), generator=g)
    vector = [0.1 * torch.randn(p.shape, generator=g, dtype=torch.float64) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    cfg = Config(type="cg", cg_iterations=K, cg_alpha=1.0) if algo == "cg" else Config(type="neumann", neumann_iterations=K, neumann_alpha=0.05)
    curr = zoo.StubProblem("inner", inner, config=cfg, loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))
    return curr, prev, vector


@pytest.mark.parametrize("affine", [True, False])
def test_declared_layers_give_the_same_gradient_and_hessian_vector_product(affine):
    res = {}
    for cls in (torch.nn.BatchNorm2d, _AlwaysFused):
        curr, prev, vector = _problem(cls, affine)
        params = list(curr.module.parameters())
        with checker_vjp():
            loss = curr.training_step_exec(curr.cur_batch)
            grads = torch.autograd.grad(loss, params, create_graph=True)
            hv = torch.autograd.grad(grads, params, grad_outputs=vector)
        res[cls] = ([g.detach().numpy() for g in grads], [h.numpy() for h in hv], curr.module.b1.running_mean.clone(), int(curr.module.b1.num_batches_tracked))
    e_g, _ = rel_err(res[_AlwaysFused][0], res[torch.nn.BatchNorm2d][0])
    e_h, _ = rel_err(res[_AlwaysFused][1], res[torch.nn.BatchNorm2d][1])
    assert e_g <= 1e-12 and e_h <= 1e-10, (e_g, e_h)
    assert res[_AlwaysFused][3] == res[torch.nn.BatchNorm2d][3] == 1
    torch.testing.assert_close(res[_AlwaysFused][2], res[torch.nn.BatchNorm2d][2])


@pytest.mark.parametrize("algo", ["cg", "neumann"])
def test_hypergradient_through_declared_layers_matches_plain_batchnorm(algo, checker):
    out = {}
    for cls in (torch.nn.BatchNorm2d, _AlwaysFused):
        curr, prev, vector = _problem(cls, True, algo)
        calls0 = bnn.fused_batchnorm_calls()
        with checker_vjp():
            res = hg.jvp_fn_mapping[algo](vector, curr, prev, False)
        out[cls] = [t.detach().numpy() for t in res]
        calls = {k: v - calls0[k] for k, v in bnn.fused_batchnorm_calls().items()}
        if cls is _AlwaysFused:
            K = 4
            assert calls["forward"] == 2 and calls["backward_vjp"] == 2 * K, calls   # two layers: one fused call per layer and product
        else:
            assert calls["backward_vjp"] == 0
    rel, _ = rel_err(out[_AlwaysFused], out[torch.nn.BatchNorm2d])
    assert rel <= 1e-9, rel


def test_module_is_nn_batchnorm_wherever_the_fused_route_does_not_apply():
    torch.manual_seed(0)
    net = _Net(torch.nn.BatchNorm2d)
    assert bnn.fuse_batchnorm_(net) == 2 and type(net.b1) is bnn.FusedBatchNorm2d and bnn.fuse_batchnorm_(net) == 0
    ref = _Net(torch.nn.BatchNorm2d)
    ref.load_state_dict(net.state_dict())          # same keys: the declaration changes no parameter or buffer
    x = torch.randn(4, 3, 5, 5)
    for mode in ("train", "eval"):
        getattr(net, mode)(), getattr(ref, mode)()
        torch.testing.assert_close(net(x), ref(x))          # CPU tensors: nn.BatchNorm2d's own forward
    torch.testing.assert_close(net.b1.running_var, ref.b1.running_var)
    assert int(net.b1.num_batches_tracked) == int(ref.b1.num_batches_tracked) == 1
    # the product's double backward has no CPU route
    from betty_amd._native import NativeLibraryError

    with pytest.raises(NativeLibraryError):
        bnn._vjp_hip(x, x, x, None, torch.zeros(3), torch.ones(3), None, None)


# ---- GPU ---------------------------------------------------------------------------------------------------------------------------
DEV = "cuda:0"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(25, 64, 84, 84), (25, 128, 42, 42), (25, 256, 21, 21), (25, 512, 10, 10), (3, 5, 7, 3), (2, 1, 1, 1), (64, 16, 32, 32)])
@pytest.mark.parametrize("nulls", ["none", "affine_false", "no_a", "no_bc"])
def test_hip_kernels_match_the_checker(shape, nulls):
    from betty_amd.backend import get_backend

    assert get_backend().name == "hip"
    g = torch.Generator().manual_seed(sum(shape))
    N, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).to(DEV)
    gy = torch.randn(shape, generator=g).to(DEV)
    a = torch.randn(shape, generator=g).to(DEV) if nulls != "no_a" else None
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if nulls != "affine_false" else None
    b = torch.randn(C, generator=g).to(DEV) if nulls not in ("no_bc", "affine_false") else None
    c = torch.randn(C, generator=g).to(DEV) if nulls not in ("no_bc", "affine_false") else None
    mean, invstd = _stats(x.double())
    mean32, inv32 = mean.float(), invstd.float()
    got = bnn._vjp_hip(x, gy, a, gamma, mean32, inv32, b, c)
    again = bnn._vjp_hip(x, gy, a, gamma, mean32, inv32, b, c)
    d = lambda t: None if t is None else t.double()   # noqa: E731
    want = bn_ref.bn_backward_vjp(x.double(), gy.double(), d(a), d(gamma), mean32.double(), inv32.double(), d(b), d(c))
    for i, (w, gt, ag) in enumerate(zip(want, got, again)):
        if w is None:
            assert gt is None
            continue
        assert torch.equal(gt, ag), "bit-reproducible"
        rel = float((gt.double() - w).norm() / w.norm().clamp_min(1e-300))
        # (two elements per channel: xh = +-1 up to the rounding of invstd, everything else is cancellation — a looser bound there)
        assert rel <= (2e-6 if N * H * W >= 16 else 1e-3), (shape, nulls, i, rel)


@pytest.mark.gpu
def test_module_on_the_gpu_matches_nn_batchnorm_gradient_and_hvp():
    res = {}
    for fused in (False, True):
        torch.manual_seed(1)
        net = _Net(torch.nn.BatchNorm2d).to(DEV)
        if fused:
            assert bnn.fuse_batchnorm_(net) == 2
        g = torch.Generator().manual_seed(1)
        x, y = torch.randn(16, 3, 12, 12, generator=g).to(DEV), torch.randint(0, 3, (16,), generator=g).to(DEV)
        params = list(net.parameters())
        vec = [torch.randn(p.shape, generator=g).to(DEV) for p in params]
        c0 = bnn.fused_batchnorm_calls()
        loss = F.cross_entropy(net(x), y)
        grads = torch.autograd.grad(loss, params, create_graph=True)
        hv = torch.autograd.grad(grads, params, grad_outputs=vec)
        calls = {k: v - c0[k] for k, v in bnn.fused_batchnorm_calls().items()}
        assert calls["backward_vjp"] == (2 if fused else 0), calls
        res[fused] = ([t.detach().cpu().numpy() for t in grads], [t.cpu().numpy() for t in hv], net.b2.running_var.cpu())
    e_g, _ = rel_err(res[True][0], res[False][0])
    e_h, _ = rel_err(res[True][1], res[False][1])
    print(f"FusedBatchNorm2d vs nn.BatchNorm2d on the GPU: gradient {e_g:.2e}, Hessian-vector product {e_h:.2e}")
    assert e_g <= 1e-5 and e_h <= 1e-4, (e_g, e_h)
    torch.testing.assert_close(res[True][2], res[False][2])


# ---- PointwiseConv2d: a 1 x 1 convolution as the matrix product it is ---------------------------------------------------------------
@pytest.mark.parametrize("bias", [False, True])
def test_pointwise_conv_matches_conv2d_value_gradient_and_hessian_vector_product(bias):
    torch.manual_seed(2)
    ref = torch.nn.Sequential(torch.nn.Conv2d(5, 7, 1, bias=bias), torch.nn.Tanh(), torch.nn.Conv2d(7, 3, 3, padding=1), torch.nn.Conv2d(3, 4, 1, stride=2)).double()
    net = torch.nn.Sequential(torch.nn.Conv2d(5, 7, 1, bias=bias), torch.nn.Tanh(), torch.nn.Conv2d(7, 3, 3, padding=1), torch.nn.Conv2d(3, 4, 1, stride=2)).double()
    net.load_state_dict(ref.state_dict())
    assert bnn.declare_pointwise_convs_(net) == 1 and type(net[0]) is bnn.PointwiseConv2d and type(net[3]) is torch.nn.Conv2d   # (stride 2: left alone)
    assert bnn.declare_layers_(net) == {"batchnorm": 0, "pointwise_conv": 0}
    x = torch.randn(3, 5, 6, 4, dtype=torch.float64)
    res = {}
    for name, m in (("ref", ref), ("net", net)):
        params = list(m.parameters())
        g = torch.Generator().manual_seed(5)
        vec = [torch.randn(p.shape, generator=g, dtype=torch.float64) for p in params]
        y = m(x)
        grads = torch.autograd.grad((y ** 2).sum(), params, create_graph=True)
        hv = torch.autograd.grad(grads, params, grad_outputs=vec)
        res[name] = (y.detach(), [t.detach() for t in grads], list(hv))
    torch.testing.assert_close(res["net"][0], res["ref"][0], rtol=1e-12, atol=1e-12)
    for a, b in zip(res["net"][1] + res["net"][2], res["ref"][1] + res["ref"][2]):
        torch.testing.assert_close(a, b, rtol=1e-10, atol=1e-12)
    torch.nn.Conv2d(5, 7, 1, bias=bias).load_state_dict(bnn.PointwiseConv2d(5, 7, 1, bias=bias).state_dict())   # same keys


@pytest.mark.gpu
def test_pointwise_conv_on_the_gpu_matches_conv2d_gradient_and_hvp():
    res = {}
    for declared in (False, True):
        torch.manual_seed(4)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(16, 24, 1, bias=False), torch.nn.Tanh(),
                                  torch.nn.Conv2d(24, 8, 1), torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(), torch.nn.Linear(8, 3)).to(DEV)
        if declared:
            assert bnn.declare_pointwise_convs_(net) == 2
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(8, 3, 10, 10, generator=g).to(DEV), torch.randint(0, 3, (8,), generator=g).to(DEV)
        params = list(net.parameters())
        vec = [torch.randn(p.shape, generator=g).to(DEV) for p in params]
        grads = torch.autograd.grad(F.cross_entropy(net(x), y), params, create_graph=True)
        hv = torch.autograd.grad(grads, params, grad_outputs=vec)
        res[declared] = ([t.detach().cpu().numpy() for t in grads], [t.cpu().numpy() for t in hv])
    e_g, _ = rel_err(res[True][0], res[False][0])
    e_h, _ = rel_err(res[True][1], res[False][1])
    print(f"PointwiseConv2d vs nn.Conv2d on the GPU: gradient {e_g:.2e}, Hessian-vector product {e_h:.2e}")
    assert e_g <= 1e-5 and e_h <= 1e-4, (e_g, e_h)
