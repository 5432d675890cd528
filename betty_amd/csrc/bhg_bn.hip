// bhg_bn.hip — the Hessian-vector product's share of a training-mode batch-normalisation layer, fused.
//
// Where it sits on the path: betty/hypergradient/cg.py:39-41 / neumann.py:62 take H p as the DOUBLE BACKWARD
// `torch.autograd.grad(in_grad, params, grad_outputs=p)`.  For a convolutional inner problem with batch norm (BASELINE cfg 3: ResNet-12,
// examples/implicit_maml/models.py:278-483) ATen differentiates batch norm's backward by DECOMPOSING it into element-wise and
// per-channel reduction launches: ~340 launches of 3-6 us per layer and product, 5.4 k launches per product, 42 % of the kernel time
// of a hypergradient step (profiles/r06_cfg3_step_kernel_breakdown.txt) — all of it HBM-bound streaming over three tensors.
//
// The backward of batch norm (batch statistics, biased variance; M = N * H * W elements per channel) is the map
//     F : (x, gy, gamma) -> (gx, ggamma, gbeta)
//         xh = (x - mean) * invstd          gbeta = sum gy          ggamma = sum gy * xh
//         gx = (gamma * invstd / M) * (M * gy - gbeta - xh * ggamma)
// and one Hessian-vector product needs its VECTOR-JACOBIAN product: given cotangents (a, b, c) of (gx, ggamma, gbeta), the gradients of
//     Phi = <a, gx> + b * ggamma + c * gbeta
// with respect to x, gy and gamma — mean and invstd being functions of x.  With the five per-channel sums
//     Sa = sum a    Sg = sum gy    P = sum a * xh    Q = sum gy * xh    A = sum a * gy          k = gamma * invstd / M
//     Psi = M * A - Sa * Sg - P * Q
// they are (derivation and a float64 check against autograd: tests/test_fused_batchnorm.py)
//     dgy    = gamma * invstd * a + (b - k P) * xh + (c - k Sa)
//     dgamma = invstd * Psi / M
//     dx     = invstd * [ -k Q * a + (b - k P) * gy ]  +  d3 * xh  +  d4
//              d3 = invstd * (2 k P Q - b Q) / M - gamma * Psi * invstd^2 / M^2          d4 = invstd * (k (Sa Q + Sg P) - b Sg) / M
// TWO launches: k_bn_vjp_stats (the five sums: every element of a, gy, x read once; fp64 accumulation; per-slice partials) and
// k_bn_vjp_apply (every block of a channel adds that channel's partials in slice order — the same bits everywhere — and streams
// a, gy, x once more, writing dx and dgy).  No atomics: bitwise run-to-run deterministic, like the rest of the library.
// Algorithmic bytes per call: (3 reads + 3 reads + 2 writes) * 4 * N * C * H * W = 32 bytes per element; HBM-bound.
//
// Layout: NCHW contiguous fp32 (PyTorch's default): channel c of sample n is the run of HW floats at ((n * C + c) * HW).  When
// HW % 4 == 0 every access is a 16-byte vector (dwordx4); otherwise scalar accesses (the 21 x 21 maps of ResNet-12).
#include "bhg_common.hpp"

namespace bhg {
namespace {

constexpr int kBnMaxSlices = 512;   // slices of one channel's M elements (workgroups per channel): sample groups x chunks of the H*W run
constexpr int kBnSums = 5;

struct BnArgs {
  const float* x; const float* gy; const float* a;      // a may be NULL (cotangent of gx not defined: treated as zero)
  const float* gamma;                                   // may be NULL (affine = False: gamma = 1, no dgamma)
  const float* mean; const float* invstd;
  const float* b; const float* c;                       // [C], may be NULL (zero)
  float* dx; float* dgy; float* dgamma;                 // dgamma may be NULL
  double* part;                                         // [C][slices][5]
  int N, C, HW;
  int groups, ng;                                       // sample groups and samples per group
  int tpr;                                              // threads per run (power of two <= 256): 256 / tpr samples side by side
  int chunks, chunk;                                    // chunks of one sample's H*W run and their length (a multiple of 4)
  int slices;                                           // groups * chunks
  long long M;                                          // N * HW elements per channel
};

template <int VEC>
__device__ __forceinline__ void bn_load(const float* __restrict__ p, float (&v)[VEC]) {
  if (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}

// grid = (slices, C).  Slice s = (sample group, chunk of the H*W run): no division inside the loops — a sample's run of channel c is
// contiguous, the workgroup walks its samples and, inside each, its chunk with a stride of 256 vectors.  (The first version cut the
// channel's N * HW elements into equal ranges and found (n, position) by a 64-bit division per vector: 3.7 TB/s on the ResNet-12
// shapes; round 6, profiles/r06_bench_bn_*.txt.)
template <int VEC>
__global__ __launch_bounds__(kThreads) void k_bn_vjp_stats(BnArgs q) {
  __shared__ double red[kWaves];
  const int c = blockIdx.y, s = blockIdx.x;
  const int grp = s / q.chunks, ck = s - grp * q.chunks;
  const int n0 = grp * q.ng, n1 = n0 + q.ng < q.N ? n0 + q.ng : q.N;
  const int lo = ck * q.chunk, hi = lo + q.chunk < q.HW ? lo + q.chunk : q.HW;
  const double mu = (double)q.mean[c], inv = (double)q.invstd[c];
  double sa = 0.0, sg = 0.0, sp = 0.0, sq = 0.0, sA = 0.0;
  // short runs (10 x 10 maps: 25 vectors): `tpr` threads walk one sample's chunk, 256 / tpr samples of the group side by side
  const int tv = (int)threadIdx.x & (q.tpr - 1), tr = (int)threadIdx.x / q.tpr, rows = kThreads / q.tpr;
  for (int n = n0 + tr; n < n1; n += rows) {
    const long long base = ((long long)n * q.C + c) * q.HW;
    for (int e = lo + VEC * tv; e < hi; e += VEC * q.tpr) {
      const long long off = base + e;
      float xv[VEC], gv[VEC], av[VEC];
      bn_load<VEC>(q.x + off, xv);
      bn_load<VEC>(q.gy + off, gv);
      if (q.a) bn_load<VEC>(q.a + off, av);
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const double xh = ((double)xv[u] - mu) * inv;
        const double aa = q.a ? (double)av[u] : 0.0;
        sa += aa;
        sg += (double)gv[u];
        sp += aa * xh;
        sq += (double)gv[u] * xh;
        sA += aa * (double)gv[u];
      }
    }
  }
  double* out = q.part + ((long long)c * q.slices + s) * kBnSums;
  const double r0 = block_sum(sa, red), r1 = block_sum(sg, red), r2 = block_sum(sp, red), r3 = block_sum(sq, red), r4 = block_sum(sA, red);
  if (threadIdx.x == 0) { out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3; out[4] = r4; }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void k_bn_vjp_apply(BnArgs q) {
  __shared__ double coef[8];
  __shared__ double red[kWaves];
  const int c = blockIdx.y, s = blockIdx.x;
  // the channel's five sums: every block of the channel adds the same partials in the same (fixed) tree — the same bits everywhere
  double Sa = 0.0, Sg = 0.0, P = 0.0, Q = 0.0, A = 0.0;
  {
    const double* pp = q.part + (long long)c * q.slices * kBnSums;
    for (int i = threadIdx.x; i < q.slices; i += kThreads) {
      Sa += pp[i * kBnSums + 0]; Sg += pp[i * kBnSums + 1]; P += pp[i * kBnSums + 2]; Q += pp[i * kBnSums + 3]; A += pp[i * kBnSums + 4];
    }
    Sa = block_sum(Sa, red); Sg = block_sum(Sg, red); P = block_sum(P, red); Q = block_sum(Q, red); A = block_sum(A, red);
  }
  if (threadIdx.x == 0) {
    const double M = (double)q.M, inv = (double)q.invstd[c], g = q.gamma ? (double)q.gamma[c] : 1.0;
    const double b = q.b ? (double)q.b[c] : 0.0, cc = q.c ? (double)q.c[c] : 0.0;
    const double k = g * inv / M;
    const double Psi = M * A - Sa * Sg - P * Q;
    coef[0] = g * inv;                  // dgy: * a
    coef[1] = b - k * P;                // dgy: * xh          (and dx: invstd * this * gy)
    coef[2] = cc - k * Sa;              // dgy: constant
    coef[3] = -inv * k * Q;             // dx: * a
    coef[4] = inv * (b - k * P);        // dx: * gy
    coef[5] = inv * (2.0 * k * P * Q - b * Q) / M - g * Psi * inv * inv / (M * M);   // dx: * xh
    coef[6] = inv * (k * (Sa * Q + Sg * P) - b * Sg) / M;                              // dx: constant
    if (s == 0 && q.dgamma) q.dgamma[c] = (float)(inv * Psi / M);
  }
  __syncthreads();
  // The element-wise part in fp64 as well (one rounding, at the store): these are sums of terms that cancel — M gy - Sg - xh Q and
  // the like — and the vector fp64 rate is far above what 32 bytes per element of HBM traffic can feed.
  const double c0 = coef[0], c1 = coef[1], c2 = coef[2], d0 = coef[3], d1 = coef[4], d2 = coef[5], d3 = coef[6];
  const double mu = (double)q.mean[c], inv = (double)q.invstd[c];
  const int grp = s / q.chunks, ck = s - grp * q.chunks;
  const int n0 = grp * q.ng, n1 = n0 + q.ng < q.N ? n0 + q.ng : q.N;
  const int lo = ck * q.chunk, hi = lo + q.chunk < q.HW ? lo + q.chunk : q.HW;
  // short runs (10 x 10 maps: 25 vectors): `tpr` threads walk one sample's chunk, 256 / tpr samples of the group side by side
  const int tv = (int)threadIdx.x & (q.tpr - 1), tr = (int)threadIdx.x / q.tpr, rows = kThreads / q.tpr;
  for (int n = n0 + tr; n < n1; n += rows) {
    const long long base = ((long long)n * q.C + c) * q.HW;
    for (int e = lo + VEC * tv; e < hi; e += VEC * q.tpr) {
      const long long off = base + e;
      float xv[VEC], gv[VEC], av[VEC], ox[VEC], og[VEC];
      bn_load<VEC>(q.x + off, xv);
      bn_load<VEC>(q.gy + off, gv);
      if (q.a) bn_load<VEC>(q.a + off, av);
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        const double xh = ((double)xv[u] - mu) * inv;
        const double aa = q.a ? (double)av[u] : 0.0;
        og[u] = (float)(c0 * aa + c1 * xh + c2);
        ox[u] = (float)(d0 * aa + d1 * (double)gv[u] + d2 * xh + d3);
      }
      if (VEC == 4) {
        *reinterpret_cast<float4*>(q.dgy + off) = make_float4(og[0], og[1], og[2], og[3]);
        *reinterpret_cast<float4*>(q.dx + off) = make_float4(ox[0], ox[1], ox[2], ox[3]);
      } else {
        q.dgy[off] = og[0];
        q.dx[off] = ox[0];
      }
    }
  }
}

// The slicing of one channel's N x HW elements: sample groups (<= 256, whole samples) x chunks of the H*W run (multiples of 4 elements, so
// that 16-byte accesses stay aligned): about 2048 workgroups per launch, at least ~1024 elements per chunk.
inline void bn_slicing(BnArgs* q, int vec) {
  const int N = q->N, C = q->C, HW = q->HW;
  const int target = (2048 + C - 1) / C;                        // workgroups per channel
  int chunks = (target + N - 1) / N;                            // more than one chunk per run only when there are few samples
  const int max_chunks = (HW + 1023) / 1024;
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  const int chunk = (((HW + chunks - 1) / chunks) + 3) & ~3;
  chunks = (HW + chunk - 1) / chunk;
  // samples per group: about 1024 elements per workgroup at least, at most 256 groups.  (NOT "no more workgroups than the target": for a
  // large layer 16 k workgroups of 3 k elements stream faster than 2 k workgroups of 25 k — 292 vs 330 us at 64 x 256 x 56 x 56.)
  int ng = (1024 + chunk - 1) / chunk;
  if ((N + 255) / 256 > ng) ng = (N + 255) / 256;
  if (ng > N) ng = N;
  if (ng < 1) ng = 1;
  int groups = (N + ng - 1) / ng;
  while (groups * chunks > kBnMaxSlices) { ++ng; groups = (N + ng - 1) / ng; }
  const int vecs = (chunk + vec - 1) / vec;
  int tpr = kThreads;
  while (tpr > 16 && tpr / 2 >= vecs) tpr /= 2;
  q->ng = ng; q->groups = groups; q->chunks = chunks; q->chunk = chunk; q->slices = groups * chunks; q->tpr = tpr;
}

}  // namespace
}  // namespace bhg

using namespace bhg;

extern "C" {

size_t bhg_bn_ws_bytes(int C) { return C > 0 ? sizeof(double) * kBnSums * kBnMaxSlices * (size_t)C : 0; }

int bhg_bn_backward_vjp(const float* x, const float* gy, const float* a, const float* gamma, const float* mean, const float* invstd,
                        const float* b, const float* c, int N, int C, int HW, float* dx, float* dgy, float* dgamma, void* ws,
                        size_t ws_bytes, void* stream) {
  BHG_REQUIRE(x && gy && mean && invstd && dx && dgy && ws, "null pointer");
  BHG_REQUIRE(N > 0 && C > 0 && HW > 0, "empty tensor");
  BHG_REQUIRE(C <= 65535, "more than 65535 channels");
  BHG_REQUIRE(ws_bytes >= bhg_bn_ws_bytes(C), "workspace smaller than bhg_bn_ws_bytes(C)");
  const bool vec = HW % 4 == 0;
  if (vec)
    for (const void* p : {(const void*)x, (const void*)gy, (const void*)a, (const void*)dx, (const void*)dgy})
      BHG_REQUIRE(((uintptr_t)p & 15) == 0, "tensors must be 16-byte aligned when H * W is a multiple of 4");
  BnArgs q{};
  q.x = x; q.gy = gy; q.a = a; q.gamma = gamma; q.mean = mean; q.invstd = invstd; q.b = b; q.c = c;
  q.dx = dx; q.dgy = dgy; q.dgamma = dgamma; q.part = static_cast<double*>(ws);
  q.N = N; q.C = C; q.HW = HW;
  q.M = (long long)N * HW;
  bn_slicing(&q, vec ? 4 : 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid(q.slices, C);
  if (vec) {
    hipLaunchKernelGGL(k_bn_vjp_stats<4>, grid, dim3(kThreads), 0, st, q);
    hipLaunchKernelGGL(k_bn_vjp_apply<4>, grid, dim3(kThreads), 0, st, q);
  } else {
    hipLaunchKernelGGL(k_bn_vjp_stats<1>, grid, dim3(kThreads), 0, st, q);
    hipLaunchKernelGGL(k_bn_vjp_apply<1>, grid, dim3(kThreads), 0, st, q);
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

}  // extern "C"
