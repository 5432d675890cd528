"""Declared layer structure for OPAQUE inner problems: batch normalisation whose share of a Hessian-vector product is fused.

The reference takes ``H p`` as the double backward ``torch.autograd.grad(in_grad, params, grad_outputs=p)``
(betty/hypergradient/cg.py:39-41, neumann.py:62).  For a convolutional inner network with training-mode batch norm — BASELINE cfg 3,
the ResNet-12 of examples/implicit_maml/models.py:278-483 — ATen differentiates ``native_batch_norm_backward`` by decomposing it into
~340 element-wise and per-channel reduction launches per layer and product: 42 % of the kernel time of a hypergradient step
(profiles/r06_cfg3_step_kernel_breakdown.txt), all of it streaming over three activation-sized tensors.

``FusedBatchNorm2d`` / ``fuse_batchnorm_(module)`` declare the layer instead, the way ``hypergradient_structure`` declares an inner
loss: forward and first backward stay ATen's own kernels (``native_batch_norm`` / ``native_batch_norm_backward`` — MIOpen), but they
are tied into the autograd graph as two ``autograd.Function`` nodes, and the backward OF the backward node — what an HVP's double
backward runs — is ``bhg_bn_backward_vjp`` (csrc/bhg_bn.hip): two launches, 32 bytes per element, deterministic.

Parameters, buffers, ``state_dict`` keys, running-statistics updates and eval-mode behaviour are ``torch.nn.BatchNorm2d``'s own (the
class is a subclass that only overrides ``forward`` for CUDA fp32 training-mode inputs); on anything else — CPU tensors, eval mode,
other dtypes, channels-last — it IS ``nn.BatchNorm2d``.  Third derivatives are not provided (the backward of the backward node is
``once_differentiable``): implicit differentiation needs second order only.
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import _native

__all__ = ["FusedBatchNorm2d", "fuse_batchnorm_", "fused_batchnorm_calls", "PointwiseConv2d", "declare_pointwise_convs_", "declare_layers_"]

_CALLS = {"forward": 0, "backward": 0, "backward_vjp": 0}   # test / measurement hook: which nodes ran


def fused_batchnorm_calls() -> dict:
    return dict(_CALLS)


def _stream() -> int:
    return int(torch.cuda.current_stream().cuda_stream)


_WS = {}   # (device, C) -> scratch for the per-slice sums (stream-ordered reuse: one HVP's layers run back to back on one stream)


def _workspace(device, C, lib):
    key = (str(device), int(torch.cuda.current_stream(device).cuda_stream))
    n = int(lib.bhg_bn_ws_bytes(int(C)))
    ws = _WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _WS[key] = torch.empty(max(n, 1 << 16), dtype=torch.uint8, device=device)
    return ws


class _BNTrainBackward(torch.autograd.Function):
    """(x, gy, gamma) -> (gx, ggamma, gbeta): batch norm's backward as a graph node whose own backward is ONE fused call."""

    @staticmethod
    def forward(ctx, x, gy, gamma, mean, invstd, eps, impl, reserve):
        _CALLS["backward"] += 1
        gy = gy.contiguous()
        # the backward nn.BatchNorm2d itself would have run (MIOpen's when the forward chose MIOpen: same kernels as an undeclared layer)
        gx, gg, gb = torch.ops.aten._batch_norm_impl_index_backward(impl, x, gy, gamma, None, None, mean, invstd, True, eps,
                                                                    [True, gamma is not None, gamma is not None], reserve)
        ctx.save_for_backward(x, gy, gamma, mean, invstd)
        ctx.set_materialize_grads(False)
        return gx, gg, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, a, b, c):
        x, gy, gamma, mean, invstd = ctx.saved_tensors
        _CALLS["backward_vjp"] += 1
        dx, dgy, dgamma = _VJP_IMPL[0](x, gy, a, gamma, mean, invstd, b, c)
        return dx, dgy, dgamma, None, None, None, None, None


def _vjp_hip(x, gy, a, gamma, mean, invstd, b, c):
    """(dx, dgy, dgamma) through bhg_bn_backward_vjp.  The product's only implementation: it raises when the HIP extension is missing or
    the tensors are not on the GPU."""
    if not x.is_cuda:
        raise _native.NativeLibraryError("FusedBatchNorm2d's double backward needs CUDA/HIP tensors; there is no CPU fallback")
    lib = _native.load()
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)

    def prep(t):
        if t is None:
            return None
        t = t.detach()
        return t if (t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0) else t.to(torch.float32).contiguous().clone()

    a, b, c = prep(a), prep(b), prep(c)
    dx, dgy = torch.empty_like(x), torch.empty_like(x)
    dgamma = torch.empty_like(gamma) if gamma is not None else None
    ws = _workspace(x.device, C, lib)
    ptr = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    _native.check(
        lib.bhg_bn_backward_vjp(x.data_ptr(), gy.data_ptr(), ptr(a), ptr(gamma), mean.data_ptr(), invstd.data_ptr(), ptr(b), ptr(c),
                                int(N), int(C), int(HW), dx.data_ptr(), dgy.data_ptr(), ptr(dgamma), ws.data_ptr(), ws.numel(), _stream()),
        "bhg_bn_backward_vjp")
    return dx, dgy, dgamma


_VJP_IMPL = [_vjp_hip]   # TEST HOOK ONLY (tests/test_fused_batchnorm.py swaps in the float64 restatement to check the graph plumbing on CPU)


class _BNTrain(torch.autograd.Function):
    """Training-mode batch norm on ATen's own kernel; its backward is the node above (so create_graph=True records THAT node)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps):
        _CALLS["forward"] += 1
        # F.batch_norm's own dispatch (MIOpen / native): (output, batch mean, batch 1 / sqrt(var + eps), reserve space, implementation)
        y, mean, invstd, reserve, impl = torch._batch_norm_impl_index(x, gamma, beta, running_mean, running_var, True, momentum, eps,
                                                                      torch.backends.cudnn.enabled)
        ctx.save_for_backward(x, gamma, mean, invstd, reserve)
        ctx.eps, ctx.impl = eps, impl
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, mean, invstd, reserve = ctx.saved_tensors
        gx, gg, gb = _BNTrainBackward.apply(x, gy, gamma, mean, invstd, ctx.eps, ctx.impl, reserve)
        return gx, gg, gb, None, None, None, None


class FusedBatchNorm2d(torch.nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose double backward is fused (module docstring)."""

    def _fusable(self, x) -> bool:
        return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.data_ptr() % 16 == 0
                and (self.training or (self.running_mean is None and self.running_var is None))
                and (self.weight is None) == (self.bias is None)
                and (self.weight is None or (self.weight.dtype == torch.float32 and self.weight.is_contiguous())))

    def forward(self, x):
        if not self._fusable(x) or not torch.is_grad_enabled():
            return super().forward(x)
        self._check_input_dim(x)
        # nn.BatchNorm2d.forward's bookkeeping of the running statistics, restated (torch/nn/modules/batchnorm.py)
        factor = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            factor = 1.0 / float(self.num_batches_tracked) if self.momentum is None else self.momentum
        rm = self.running_mean if (not self.training or self.track_running_stats) else None
        rv = self.running_var if (not self.training or self.track_running_stats) else None
        return _BNTrain.apply(x, self.weight, self.bias, rm, rv, factor, self.eps)


def fuse_batchnorm_(module: torch.nn.Module) -> int:
    """Declare every ``nn.BatchNorm2d`` of ``module`` (exact class; subclasses are left alone) as :class:`FusedBatchNorm2d`, in place:
    same parameters, buffers and hooks, only the class changes.  Returns how many layers were declared."""
    n = 0
    for m in module.modules():
        if type(m) is torch.nn.BatchNorm2d:
            m.__class__ = FusedBatchNorm2d
            n += 1
    return n


class PointwiseConv2d(torch.nn.Conv2d):
    """``nn.Conv2d`` with a 1 x 1 kernel (stride 1, no padding, one group) evaluated as what it is — a matrix product over the channels,
    ``y[n] = W x[n]`` — so that autograd's double backward of it (cg.py:39-41, neumann.py:62) is three more matrix products on
    rocBLAS / hipBLASLt instead of MIOpen's convolution double backward, which for these shapes can fall to a PER-SAMPLE im2col + GEMM
    loop: on BASELINE cfg 3's ResNet-12 the four 1 x 1 shortcut projections cost 25 launches per sample-loop, 12.6 k ``Im2d2Col`` + 16.6 k
    small GEMM launches and ~220 ms of a 900 ms step (profiles/r06_cfg3_step_kernel_breakdown_declared_batchnorm.txt).  Same parameters,
    ``state_dict`` and results to fp32 rounding; any other configuration of the layer IS ``nn.Conv2d``."""

    def _pointwise(self) -> bool:
        return (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding in ((0, 0), "valid") and self.dilation == (1, 1)
                and self.groups == 1 and self.padding_mode == "zeros")

    def forward(self, x):
        if not self._pointwise() or x.dim() != 4:
            return super().forward(x)
        n, _, h, w = x.shape
        y = torch.matmul(self.weight.reshape(self.out_channels, self.in_channels), x.reshape(n, self.in_channels, h * w))
        if self.bias is not None:
            y = y + self.bias.reshape(1, -1, 1)
        return y.reshape(n, self.out_channels, h, w)


def declare_pointwise_convs_(module: torch.nn.Module) -> int:
    """Re-class every 1 x 1 / stride 1 / ungrouped ``nn.Conv2d`` of ``module`` (exact class) as :class:`PointwiseConv2d`, in place."""
    n = 0
    for m in module.modules():
        if type(m) is torch.nn.Conv2d and PointwiseConv2d._pointwise(m):
            m.__class__ = PointwiseConv2d
            n += 1
    return n


def declare_layers_(module: torch.nn.Module) -> dict:
    """Every layer declaration this module offers, in place: ``{"batchnorm": n, "pointwise_conv": m}``."""
    return {"batchnorm": fuse_batchnorm_(module), "pointwise_conv": declare_pointwise_convs_(module)}
