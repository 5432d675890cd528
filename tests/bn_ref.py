"""CHECKER (test infrastructure, never imported by the package): ATen restatement of csrc/bhg_bn.hip — the vector-Jacobian product of
training-mode batch norm's backward, F : (x, gy, gamma) -> (gx, ggamma, gbeta), in whatever dtype the inputs have.  Checked against
autograd's own double backward in tests/test_fused_batchnorm.py; the HIP kernels are checked against THIS on the GPU."""
import torch


def bn_backward_vjp(x, gy, a, gamma, mean, invstd, b, c):
    C = x.shape[1]
    dims = [0] + list(range(2, x.dim()))
    r = lambda t: t.reshape(1, C, *([1] * (x.dim() - 2)))   # noqa: E731
    M = x.numel() // C
    one = torch.ones(C, dtype=x.dtype, device=x.device)
    g = gamma if gamma is not None else one
    a = a if a is not None else torch.zeros_like(x)
    b = b if b is not None else torch.zeros_like(one)
    c = c if c is not None else torch.zeros_like(one)
    xh = (x - r(mean)) * r(invstd)
    Sa, Sg, P, Q, A = a.sum(dims), gy.sum(dims), (a * xh).sum(dims), (gy * xh).sum(dims), (a * gy).sum(dims)
    k = g * invstd / M
    Psi = M * A - Sa * Sg - P * Q
    dgy = r(g * invstd) * a + r(b - k * P) * xh + r(c - k * Sa)
    dgamma = invstd * Psi / M
    d3 = invstd * (2 * k * P * Q - b * Q) / M - g * Psi * invstd * invstd / (M * M)
    d4 = invstd * (k * (Sa * Q + Sg * P) - b * Sg) / M
    dx = r(-invstd * k * Q) * a + r(invstd * (b - k * P)) * gy + r(d3) * xh + r(d4)
    return dx, dgy, (dgamma if gamma is not None else None)
