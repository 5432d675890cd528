// bhg_logreg.hip — analytic Hessian-vector product of L2-regularised logistic regression.
//
// Inner problem of examples/logistic_regression_hpo/logistic_regression_implicit.py:80-91
// (and test/test_regression.py:47-59):  L(w) = mean_i BCE(x_i.w, y_i) + 1/2 sum_j lam_j w_j^2.
//   H p = X^T( s .* (X p) ) + lam .* p,   s_i = sigma_i (1 - sigma_i) / n      
// Replaces the double-backward `torch.autograd.grad(in_grad, params, grad_outputs=p)` of
// cg.py:39-41 / neumann.py:62 for this structure with two HBM-bound GEMV passes over X.
#include "bhg_common.hpp"

namespace bhg {
namespace {

constexpr int kRowBlocksMax = 256;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One wave per row: z = x_i . vec.  mode 0: out_i = sigma(z)(1-sigma(z))/n ; mode 1: out_i = s_i * z.
__global__ __launch_bounds__(kThreads) void k_rowdot(const float* __restrict__ X, const float* __restrict__ vec,
                                                     const float* __restrict__ s, float* __restrict__ out, int n,
                                                     int d, int mode) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kWaves + (threadIdx.x >> 6);
  const int nw = gridDim.x * kWaves;
  for (int i = wave; i < n; i += nw) {
    const float* row = X + (int64_t)i * d;
    float acc = 0.f;
    for (int j = lane; j < d; j += 64) acc = fmaf(row[j], vec[j], acc);
    acc = wave_sum_f(acc);
    if (lane == 0) {
      if (mode == 0) {
        const float sg = 1.f / (1.f + __expf(-acc));
        out[i] = sg * (1.f - sg) / (float)n;
      } else {
        out[i] = s[i] * acc;
      }
    }
  }
}

// Stage 1 of X^T u: block (bx, by) sums rows {by, by+RB, ...} of column tile bx (64 columns),
// 4 waves split the rows, fixed-order LDS combine.  part[by][j].
__global__ __launch_bounds__(kThreads) void k_colsum_part(const float* __restrict__ X, const float* __restrict__ u,
                                                          float* __restrict__ part, int n, int d) {
  __shared__ float red[kWaves][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (j < d) {
    for (int i = blockIdx.y * kWaves + w; i < n; i += gridDim.y * kWaves) acc = fmaf(X[(int64_t)i * d + j], u[i], acc);
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && j < d) part[(int64_t)blockIdx.y * d + j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Stage 2: out_j = sum_by part[by][j] + lam_j p_j  (fixed order => deterministic).
__global__ __launch_bounds__(kThreads) void k_colsum_final(const float* __restrict__ part, int rb,
                                                           const float* __restrict__ lam, const float* __restrict__ p,
                                                           float* __restrict__ out, int d) {
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= d) return;
  double a = 0.0;
  for (int b = 0; b < rb; ++b) a += (double)part[(int64_t)b * d + j];
  out[j] = (float)a + (lam ? lam[j] * p[j] : 0.f);
}

}  // namespace
}  // namespace bhg

using namespace bhg;

extern "C" {

int bhg_logreg_prepare(const float* X, const float* w, float* s, int n, int d, void* stream) {
  BHG_REQUIRE(X && w && s, "NULL pointer");
  BHG_REQUIRE(n > 0 && d > 0, "empty problem");
  hipStream_t st = static_cast<hipStream_t>(stream);
  int blocks = (n + kWaves - 1) / kWaves;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_rowdot, dim3(blocks), dim3(kThreads), 0, st, X, w, (const float*)nullptr, s, n, d, 0);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

size_t bhg_logreg_tmp_floats(int n, int d) {
  if (n <= 0 || d <= 0) return 0;
  return (size_t)n + (size_t)kRowBlocksMax * (size_t)d;
}

int bhg_logreg_hvp(const float* X, const float* s, const float* lam, const float* p, float* out, float* tmp,
                   int n, int d, void* stream) {
  BHG_REQUIRE(X && s && p && out && tmp, "NULL pointer");
  BHG_REQUIRE(n > 0 && d > 0, "empty problem");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* u = tmp;
  float* part = tmp + n;
  int blocks = (n + kWaves - 1) / kWaves;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_rowdot, dim3(blocks), dim3(kThreads), 0, st, X, p, s, u, n, d, 1);
  int rb = (n + 255) / 256;
  if (rb > kRowBlocksMax) rb = kRowBlocksMax;
  if (rb < 1) rb = 1;
  hipLaunchKernelGGL(k_colsum_part, dim3((d + 63) / 64, rb), dim3(kThreads), 0, st, X, (const float*)u, part, n, d);
  hipLaunchKernelGGL(k_colsum_final, dim3((d + kThreads - 1) / kThreads), dim3(kThreads), 0, st,
                     (const float*)part, rb, lam, p, out, d);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

}  // extern "C"
