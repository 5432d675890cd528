// bhg_mwn.hip — the meta-weight-net of data reweighting, closed form: sample weights and their VJP to the net's parameters.
//
// Upper problem of examples/learning_to_reweight (model.py:98-111, `MLP(hidden_size, num_layers = 1)`; main.py:117-127):
//     s_i = sigmoid( w2 . relu(w1 * ce_i + b1) + b2 )                ce_i: detached per-sample loss, w1, b1, w2 in R^H, b2 in R
// On the hypergradient path it is touched twice per step (cg.py:27-32 evaluates the inner loss, cg.py:58-68 differentiates
// g . x w.r.t. the upper parameters): through autograd that is ~15 ATen launches — five rocBLAS GEMMs of 8-14 us each for a
// 100 x 100 problem — ~100 us of a 1.4 ms step (profiles/r04_outside_the_k_loop.txt).  Here: one launch each way.
//   forward   sd_i = s_i / B                (what bhg_mlp_backward reads as the per-sample scale of delta_L)
//   backward  grads of  sum_i coeff_i s_i   (coeff_i = d(g . x)/d s_i from bhg_mlp_*_mixed_coeff), times `scale`
// Deterministic: one workgroup, fixed summation order (hidden unit j sums its samples i = 0 .. B-1 in order).
#include "bhg_common.hpp"

namespace bhg {
namespace {

constexpr int kMwnMaxH = 2048;      // hidden width: 3 H floats of parameters live in LDS
constexpr int kMwnChunk = 1024;     // samples per pass of the backward workgroup

__device__ __forceinline__ float mwn_sigmoid(float z) { return 1.f / (1.f + expf(-z)); }

// one thread per sample; parameters staged in LDS
__global__ __launch_bounds__(kThreads) void k_mwn_forward(const float* __restrict__ ce, int B, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ w2,
                                                          const float* __restrict__ b2, int H, float* __restrict__ s,
                                                          float* __restrict__ sd) {
  extern __shared__ float sp[];   // [3][H]
  for (int j = threadIdx.x; j < H; j += kThreads) { sp[j] = w1[j]; sp[H + j] = b1[j]; sp[2 * H + j] = w2[j]; }
  __syncthreads();
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= B) return;
  const float c = ce[i];
  float z = b2[0];
  for (int j = 0; j < H; ++j) {
    const float a = fmaf(sp[j], c, sp[H + j]);
    z = fmaf(sp[2 * H + j], a > 0.f ? a : 0.f, z);
  }
  const float v = mwn_sigmoid(z);
  if (s) s[i] = v;
  if (sd) sd[i] = v / (float)B;
}

// ONE workgroup.  Pass over the samples in chunks: (a) thread = sample: dz2_i = coeff_i s_i (1 - s_i) into LDS;
// (b) thread = hidden unit: gw2_j += dz2_i a_ij, gb1_j += dz2_i w2_j [z1_ij > 0], gw1_j += (the same) ce_i.
__global__ __launch_bounds__(kThreads) void k_mwn_backward(const float* __restrict__ ce, const float* __restrict__ coeff, int B,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, int H,
                                                           float scale, float* __restrict__ gw1, float* __restrict__ gb1,
                                                           float* __restrict__ gw2, float* __restrict__ gb2) {
  extern __shared__ float sp[];   // [3][H] parameters, then [2][kMwnChunk]: dz2, ce of the chunk
  __shared__ float red[kWaves];
  float* sdz = sp + 3 * H;
  float* sce = sdz + kMwnChunk;
  for (int j = threadIdx.x; j < H; j += kThreads) { sp[j] = w1[j]; sp[H + j] = b1[j]; sp[2 * H + j] = w2[j]; }
  const float bias2 = b2[0];
  constexpr int kPer = kMwnMaxH / kThreads;   // hidden units per thread
  float a_w1[kPer], a_b1[kPer], a_w2[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) a_w1[u] = a_b1[u] = a_w2[u] = 0.f;
  float a_b2 = 0.f;   // (thread 0 .. : partial of sum_i dz2_i over the samples this thread owns in phase a)
  // the first chunk's samples are requested together with the parameters (one round trip to memory, not two: the launch is latency, not work)
  constexpr int kSPer = kMwnChunk / kThreads;
  float c_pre[kSPer], k_pre[kSPer];
#pragma unroll
  for (int u = 0; u < kSPer; ++u) {
    const int t = threadIdx.x + kThreads * u;
    c_pre[u] = ce[t < B ? t : 0];
    k_pre[u] = coeff[t < B ? t : 0];
  }
  __syncthreads();
  for (int i0 = 0; i0 < B; i0 += kMwnChunk) {
    const int n = B - i0 < kMwnChunk ? B - i0 : kMwnChunk;
#pragma unroll
    for (int u = 0; u < kSPer; ++u) {
      const int t = threadIdx.x + kThreads * u;
      if (t >= n) continue;
      const float c = i0 == 0 ? c_pre[u] : ce[i0 + t];
      const float kc = i0 == 0 ? k_pre[u] : coeff[i0 + t];
      float z = bias2;
      for (int j = 0; j < H; ++j) {
        const float a = fmaf(sp[j], c, sp[H + j]);
        z = fmaf(sp[2 * H + j], a > 0.f ? a : 0.f, z);
      }
      const float v = mwn_sigmoid(z);
      const float dz = kc * (v * (1.f - v));
      sdz[t] = dz;
      sce[t] = c;
      a_b2 += dz;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int j = threadIdx.x + kThreads * u;
      if (j < H) {
        const float wj = sp[j], bj = sp[H + j], vj = sp[2 * H + j];
        for (int t = 0; t < n; ++t) {
          const float c = sce[t], dz = sdz[t];
          const float a = fmaf(wj, c, bj);
          const bool on = a > 0.f;
          a_w2[u] = fmaf(dz, on ? a : 0.f, a_w2[u]);
          const float g1 = on ? dz * vj : 0.f;
          a_b1[u] += g1;
          a_w1[u] = fmaf(g1, c, a_w1[u]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int j = threadIdx.x + kThreads * u;
    if (j < H) { gw1[j] = scale * a_w1[u]; gb1[j] = scale * a_b1[u]; gw2[j] = scale * a_w2[u]; }
  }
  // gb2 = sum_i dz2_i: wave sums in lane order, then the four waves in order
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a_b2 += __shfl_down(a_b2, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a_b2;
  __syncthreads();
  if (threadIdx.x == 0) gb2[0] = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

int check_mwn(const float* ce, int B, const float* w1, const float* b1, const float* w2, const float* b2, int H) {
  BHG_REQUIRE(ce && w1 && b1 && w2 && b2, "NULL argument");
  BHG_REQUIRE(B >= 1 && H >= 1 && H <= kMwnMaxH, "the closed-form meta-weight-net takes 1 <= hidden width <= 2048");
  return BHG_OK;
}

}  // namespace
}  // namespace bhg

using namespace bhg;

extern "C" {

int bhg_mwn_max_hidden(void) { return kMwnMaxH; }

int bhg_mwn_forward(const float* ce, int B, const float* w1, const float* b1, const float* w2, const float* b2, int H, float* s,
                    float* sd, void* stream) {
  if (int rc = check_mwn(ce, B, w1, b1, w2, b2, H)) return rc;
  BHG_REQUIRE(s || sd, "no output requested");
  hipLaunchKernelGGL(k_mwn_forward, dim3((B + kThreads - 1) / kThreads), dim3(kThreads), sizeof(float) * 3 * (size_t)H,
                     static_cast<hipStream_t>(stream), ce, B, w1, b1, w2, b2, H, s, sd);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_mwn_backward(const float* ce, const float* coeff, int B, const float* w1, const float* b1, const float* w2, const float* b2,
                     int H, float scale, float* gw1, float* gb1, float* gw2, float* gb2, void* stream) {
  if (int rc = check_mwn(ce, B, w1, b1, w2, b2, H)) return rc;
  BHG_REQUIRE(coeff && gw1 && gb1 && gw2 && gb2, "NULL argument");
  hipLaunchKernelGGL(k_mwn_backward, dim3(1), dim3(kThreads), sizeof(float) * (3 * (size_t)H + 2 * kMwnChunk),
                     static_cast<hipStream_t>(stream), ce, coeff, B, w1, b1, w2, b2, H, scale, gw1, gb1, gw2, gb2);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

}  // extern "C"
