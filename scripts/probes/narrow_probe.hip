// narrow_probe.hip — round 6, VERDICT r5 #1(b): the narrow end of a CG-HVP iteration (pre-head product K-split over 240 workgroups ->
// head, one workgroup per batch row -> backward product through W_2) costs THREE dependent launches, 20 us for ~2 us of roofline work.
// Would ONE launch with a PARTIAL rendezvous — the <= 60 workgroups that share a 32-row group meet on a counter; data handed over with
// 16-byte sc1 stores / sc1 loads, no fence — be shorter than the two launch boundaries it removes?
//   version A: three launches (plain loads / stores), the shapes and tilings of the shipped chain at cfg 2 (Bp 128, 1536 -> 384 -> 10 -> 384 -> 1536)
//   version B: one launch of 240 workgroups, phases separated by per-row-group counters (monotonic, target = 60 * generation / 32 * generation)
// Same arithmetic in the same order: the outputs must be bit-identical (checked).  Timing: 200 iterations of [touch, A1, A2, A3] and of
// [touch, B] minus [touch] alone (k_touch dirties the L2s like the other launches of an iteration do).
// Build: hipcc --offload-arch=gfx950 -O3 -o build_probes/narrow_probe scripts/probes/narrow_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include <cmath>

using f32x4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int Bp = 128, D1 = 1536, D2 = 384, C = 10, NS = 5;   // batch rows, K of the pre-head product, head width, classes, K splits
constexpr unsigned kSpinLimit = 1u << 22;

struct Args {
  const float* A1p;     // packed Rh_1            [D1/16][Bp][16]
  const float* W2f;     // packed W_2 (forward)   [D1/16][D2][16]
  float* slab;          // [NS][Bp][D2]
  const float* bias; const float* mask2; const float* addend;   // [D2], [Bp][D2], [Bp][D2]
  const float* h2;      // [Bp][D2]  (the direction's share needs h . V)
  const float* W3; const float* V3;   // [C][D2]
  const float* prob; const float* sd; const float* dtop;   // [Bp][C], [Bp], [Bp][C]
  float* rh2;           // [Bp][D2] row-major output of the head's combine
  float* rd2p;          // packed Rd_2            [D2/16][Bp][16]
  const float* W2b;     // packed W_2 (backward)  [D2/16][D1][16]
  const float* mask1; const float* gb1;   // [Bp][D1]
  float* out;           // [Bp][D1]
  unsigned* cnt;        // [8]: cnt[rg] phase 1 -> 2, cnt[4 + rg] phase 2 -> 3   (monotonic)
  unsigned* timeout;
  unsigned gen;
};

__device__ __forceinline__ void st16(float* p, float4 v, bool sc1) {
  if (sc1) { const f32x4 w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory"); }
  else *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 ld16_sc1(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return make_float4(v[0], v[1], v[2], v[3]);
}
// four 16-byte sc1 loads in flight together
__device__ __forceinline__ void ld16x4_sc1(const float* p0, const float* p1, const float* p2, const float* p3, float4& a, float4& b, float4& c, float4& d) {
  f32x4 va, vb, vc, vd;
  asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
               "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(va), "=&v"(vb), "=&v"(vc), "=&v"(vd) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  a = make_float4(va[0], va[1], va[2], va[3]); b = make_float4(vb[0], vb[1], vb[2], vb[3]);
  c = make_float4(vc[0], vc[1], vc[2], vc[3]); d = make_float4(vd[0], vd[1], vd[2], vd[3]);
}
__device__ __forceinline__ bool wait_count(unsigned* c, unsigned target, unsigned* timeout) {
  // thread 0 polls (relaxed agent-scope loads), bounded; the workgroup follows through the barrier
  __shared__ int ok;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { good = 0; __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
__device__ __forceinline__ void arrive(unsigned* c) {
  // every thread's sc1 stores have been acknowledged (vmcnt), then one increment
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- a 32 x 32 tile of A[.., K-range] B[.., K-range]^T on packed operands: 4 waves on disjoint K ranges, partials meet in LDS ----------
// A_SC1: the A operand was written by another workgroup of this launch (sc1 loads, all of this wave's share up front)
template <bool A_SC1>
__device__ __forceinline__ void tile_32x32(const float* Ap, int RA, const float* Bq, int RB, int m0, int n0, int cb0, int ce, float (*sP)[32][33],
                                           float out4[4]) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int c0 = cb0 + (wave * (ce - cb0)) / 4, c1 = cb0 + ((wave + 1) * (ce - cb0)) / 4;
  f32x4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  const int64_t sA = 16LL * RA, sB = 16LL * RB;
  const float* gA = Ap + (m0 + li) * 16 + 4 * lk;
  const float* gB = Bq + (n0 + li) * 16 + 4 * lk;
  for (int ch = c0; ch < c1; ++ch) {
    f32x4 sa[2][2], sb[2][2];
    float4 t[4];
    if (A_SC1) {
      ld16x4_sc1(gA + (2 * ch) * sA, gA + (2 * ch) * sA + 256, gA + (2 * ch + 1) * sA, gA + (2 * ch + 1) * sA + 256, t[0], t[1], t[2], t[3]);
      for (int h = 0; h < 2; ++h) for (int rb = 0; rb < 2; ++rb) { const float4 v = t[2 * h + rb]; sa[rb][h] = f32x4{v.x, v.y, v.z, v.w}; }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        if (!A_SC1) sa[rb][h] = *reinterpret_cast<const f32x4*>(gA + (int64_t)(2 * ch + h) * sA + rb * 256);
        sb[rb][h] = *reinterpret_cast<const f32x4*>(gB + (int64_t)(2 * ch + h) * sB + rb * 256);
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][h][c], sb[0][h][c], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[0][h][c], sb[1][h][c], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][h][c], sb[0][h][c], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(sa[1][h][c], sb[1][h][c], acc[1][1], 0, 0, 0);
      }
  }
  for (int rb = 0; rb < 2; ++rb) for (int cb = 0; cb < 2; ++cb) for (int r = 0; r < 4; ++r) sP[wave][16 * rb + 4 * lk + r][16 * cb + li] = acc[rb][cb][r];
  __syncthreads();
  const int row = threadIdx.x >> 3, col = (threadIdx.x & 7) * 4;
#pragma unroll
  for (int u = 0; u < 4; ++u) out4[u] = ((sP[0][row][col + u] + sP[1][row][col + u]) + sP[2][row][col + u]) + sP[3][row][col + u];
  __syncthreads();
}

// phase 1: pre-head product, tile (rg, tn) split sp -> slab
template <bool FUSED>
__device__ __forceinline__ void phase1(const Args& q, int rg, int tn, int sp, float (*sP)[32][33]) {
  const int nct = D1 / 32;
  float o[4];
  tile_32x32<false>(q.A1p, Bp, q.W2f, D2, 32 * rg, 32 * tn, (sp * nct) / NS, ((sp + 1) * nct) / NS, sP, o);
  const int row = threadIdx.x >> 3, col = (threadIdx.x & 7) * 4;
  st16(q.slab + ((int64_t)sp * Bp + 32 * rg + row) * D2 + 32 * tn + col, make_float4(o[0], o[1], o[2], o[3]), FUSED);
}

// phase 2: the head for batch row b (one workgroup): combine the slabs, logits, softmax R-op, back through W_3 into Rd_2 (packed)
template <bool FUSED>
__device__ __forceinline__ void phase2(const Args& q, int b, float* srow) {
  __shared__ float rz[16], rdl[16], dtl[16], pz[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t < D2 / 4) {
    const int k = 4 * t;
    const float* p0 = q.slab + (int64_t)b * D2 + k;
    float4 s0, s1, s2, s3, s4;
    if (FUSED) { ld16x4_sc1(p0, p0 + (int64_t)Bp * D2, p0 + 2LL * Bp * D2, p0 + 3LL * Bp * D2, s0, s1, s2, s3); s4 = ld16_sc1(p0 + 4LL * Bp * D2); }
    else { s0 = *(const float4*)p0; s1 = *(const float4*)(p0 + (int64_t)Bp * D2); s2 = *(const float4*)(p0 + 2LL * Bp * D2); s3 = *(const float4*)(p0 + 3LL * Bp * D2); s4 = *(const float4*)(p0 + 4LL * Bp * D2); }
    const float4 bv = *(const float4*)(q.bias + k), mv = *(const float4*)(q.mask2 + (int64_t)b * D2 + k), ad = *(const float4*)(q.addend + (int64_t)b * D2 + k);
    float4 v;
    v.x = ((((s0.x + s1.x) + s2.x) + s3.x) + s4.x + ad.x + bv.x) * mv.x; v.y = ((((s0.y + s1.y) + s2.y) + s3.y) + s4.y + ad.y + bv.y) * mv.y;
    v.z = ((((s0.z + s1.z) + s2.z) + s3.z) + s4.z + ad.z + bv.z) * mv.z; v.w = ((((s0.w + s1.w) + s2.w) + s3.w) + s4.w + ad.w + bv.w) * mv.w;
    *(float4*)(srow + k) = v;
    *(float4*)(q.rh2 + (int64_t)b * D2 + k) = v;
  }
  __syncthreads();
  for (int c = wave; c < C; c += 4) {
    float a = 0.f;
    for (int k = 4 * lane; k < D2; k += 256) {
      const float4 r = *(const float4*)(srow + k), hv = *(const float4*)(q.h2 + (int64_t)b * D2 + k);
      const float4 w = *(const float4*)(q.W3 + c * D2 + k), v = *(const float4*)(q.V3 + c * D2 + k);
      a += r.x * w.x + r.y * w.y + r.z * w.z + r.w * w.w + hv.x * v.x + hv.y * v.y + hv.z * v.z + hv.w * v.w;
    }
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) rz[c] = a;
  }
  __syncthreads();
  if (t < C) pz[t] = q.prob[b * C + t] * rz[t];
  __syncthreads();
  if (t < C) {
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += pz[c];
    const float p = q.prob[b * C + t];
    rdl[t] = q.sd[b] * (p * rz[t] - p * dot);
    dtl[t] = q.dtop[b * C + t];
  }
  __syncthreads();
  if (t < D2 / 4) {
    const int k = 4 * t;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 w = *(const float4*)(q.W3 + c * D2 + k), v = *(const float4*)(q.V3 + c * D2 + k);
      const float r = rdl[c], d = dtl[c];
      acc.x += d * v.x + r * w.x; acc.y += d * v.y + r * w.y; acc.z += d * v.z + r * w.z; acc.w += d * v.w + r * w.w;
    }
    const float4 mk = *(const float4*)(q.mask2 + (int64_t)b * D2 + k);
    acc.x *= mk.x; acc.y *= mk.y; acc.z *= mk.z; acc.w *= mk.w;
    st16(q.rd2p + ((int64_t)(k >> 4) * Bp + b) * 16 + (k & 15), acc, FUSED);
  }
}

// phase 3: backward product through W_2, tile (rg, tn) of [Bp x D1], K = D2
template <bool FUSED>
__device__ __forceinline__ void phase3(const Args& q, int rg, int tn, float (*sP)[32][33]) {
  float o[4];
  tile_32x32<FUSED>(q.rd2p, Bp, q.W2b, D1, 32 * rg, 32 * tn, 0, D2 / 32, sP, o);
  const int row = threadIdx.x >> 3, col = (threadIdx.x & 7) * 4;
  const int64_t idx = (int64_t)(32 * rg + row) * D1 + 32 * tn + col;
  const float4 mk = *(const float4*)(q.mask1 + idx), g = *(const float4*)(q.gb1 + idx);
  *(float4*)(q.out + idx) = make_float4((o[0] + g.x) * mk.x, (o[1] + g.y) * mk.y, (o[2] + g.z) * mk.z, (o[3] + g.w) * mk.w);
}

__global__ __launch_bounds__(256) void kA1(Args q) {
  __shared__ float sP[4][32][33];
  const int rg = blockIdx.x / 60, j = blockIdx.x % 60;
  phase1<false>(q, rg, j / NS, j % NS, sP);
}
__global__ __launch_bounds__(256) void kA2(Args q) {
  __shared__ __attribute__((aligned(16))) float srow[D2];
  phase2<false>(q, blockIdx.x, srow);
}
__global__ __launch_bounds__(256) void kA3(Args q) {
  __shared__ float sP[4][32][33];
  phase3<false>(q, blockIdx.x / 48, blockIdx.x % 48, sP);
}
__global__ __launch_bounds__(256) void kB(Args q) {
  __shared__ float sP[4][32][33];
  __shared__ __attribute__((aligned(16))) float srow[D2];
  const int rg = blockIdx.x / 60, j = blockIdx.x % 60;
  phase1<true>(q, rg, j / NS, j % NS, sP);
  arrive(q.cnt + rg);
  if (j < 32) {
    if (!wait_count(q.cnt + rg, 60u * q.gen, q.timeout)) return;
    phase2<true>(q, 32 * rg + j, srow);
    arrive(q.cnt + 4 + rg);
  }
  if (j < 48) {
    if (!wait_count(q.cnt + 4 + rg, 32u * q.gen, q.timeout)) return;
    phase3<true>(q, rg, j, sP);
  }
}
__global__ void k_touch(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

static float* dev_rand(size_t n, unsigned seed, float scale = 1.f, bool binary = false) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const float u = (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; h[i] = binary ? (u > -0.1f ? 1.f : 0.f) : u * scale; }
  float* d; CK(hipMalloc(&d, sizeof(float) * n)); CK(hipMemcpy(d, h.data(), sizeof(float) * n, hipMemcpyHostToDevice));
  return d;
}

int main() {
  Args q{};
  q.A1p = dev_rand((size_t)Bp * D1, 1, 0.5f); q.W2f = dev_rand((size_t)D1 * D2, 2, 0.05f);
  float* slab; CK(hipMalloc(&slab, sizeof(float) * NS * Bp * D2)); q.slab = slab;
  q.bias = dev_rand(D2, 3, 0.1f); q.mask2 = dev_rand((size_t)Bp * D2, 4, 1.f, true); q.addend = dev_rand((size_t)Bp * D2, 5, 0.1f);
  q.h2 = dev_rand((size_t)Bp * D2, 6, 0.5f); q.W3 = dev_rand((size_t)C * D2, 7, 0.1f); q.V3 = dev_rand((size_t)C * D2, 8, 0.1f);
  q.prob = dev_rand((size_t)Bp * C, 9, 0.2f); q.sd = dev_rand(Bp, 10, 0.02f); q.dtop = dev_rand((size_t)Bp * C, 11, 0.01f);
  float *rh2, *rd2p, *out; CK(hipMalloc(&rh2, sizeof(float) * Bp * D2)); CK(hipMalloc(&rd2p, sizeof(float) * Bp * D2)); CK(hipMalloc(&out, sizeof(float) * Bp * D1));
  q.rh2 = rh2; q.rd2p = rd2p; q.out = out;
  q.W2b = dev_rand((size_t)D2 * D1, 12, 0.05f); q.mask1 = dev_rand((size_t)Bp * D1, 13, 1.f, true); q.gb1 = dev_rand((size_t)Bp * D1, 14, 0.1f);
  unsigned* cnt; CK(hipMalloc(&cnt, 64 * sizeof(unsigned))); CK(hipMemset(cnt, 0, 64 * sizeof(unsigned))); q.cnt = cnt; q.timeout = cnt + 32;
  float* junk; const int junk_n = 8 << 20; CK(hipMalloc(&junk, sizeof(float) * junk_n)); CK(hipMemset(junk, 0, sizeof(float) * junk_n));

  // correctness: two right-hand sides, alternating — a hand-over that reads a stale line of the PREVIOUS generation gives the other
  // input's values — version A's outputs for each are the reference, version B must reproduce them bit for bit
  const float* A1p_alt[2] = {q.A1p, dev_rand((size_t)Bp * D1, 101, 0.5f)};
  std::vector<float> oa[2], ob((size_t)Bp * D1);
  for (int v = 0; v < 2; ++v) {
    q.A1p = A1p_alt[v];
    hipLaunchKernelGGL(kA1, dim3(240), dim3(256), 0, 0, q); hipLaunchKernelGGL(kA2, dim3(Bp), dim3(256), 0, 0, q); hipLaunchKernelGGL(kA3, dim3(192), dim3(256), 0, 0, q);
    oa[v].resize((size_t)Bp * D1);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(oa[v].data(), out, sizeof(float) * ob.size(), hipMemcpyDeviceToHost));
  }
  {
    size_t d01 = 0;
    for (size_t i = 0; i < ob.size(); ++i) d01 += oa[0][i] != oa[1][i];
    printf("the two inputs' outputs differ in %zu of %zu entries (they must, for the staleness check to mean anything)\n", d01, ob.size());
  }
  unsigned gen = 0;
  int bad_runs = 0;
  for (int rep = 0; rep < 40; ++rep) {
    q.gen = ++gen;
    q.A1p = A1p_alt[rep & 1];
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
    hipLaunchKernelGGL(kB, dim3(240), dim3(256), 0, 0, q);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(ob.data(), out, sizeof(float) * ob.size(), hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < ob.size(); ++i) diff += oa[rep & 1][i] != ob[i];
    if (diff) { ++bad_runs; if (bad_runs < 4) printf("rep %d: %zu of %zu outputs differ from the three-launch result\n", rep, diff, ob.size()); }
  }
  q.A1p = A1p_alt[0];
  unsigned to = 0; CK(hipMemcpy(&to, q.timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("one-launch form: %d of 40 runs differ from the three-launch result; time-out flag %u\n", bad_runs, to);

  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 200;
  auto time_loop = [&](int which) {
    float best = 1e30f;
    for (int trial = 0; trial < 5; ++trial) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) {
        hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
        if (which == 1) { hipLaunchKernelGGL(kA1, dim3(240), dim3(256), 0, 0, q); hipLaunchKernelGGL(kA2, dim3(Bp), dim3(256), 0, 0, q); hipLaunchKernelGGL(kA3, dim3(192), dim3(256), 0, 0, q); }
        if (which == 2) { q.gen = ++gen; hipLaunchKernelGGL(kB, dim3(240), dim3(256), 0, 0, q); }
        if (which == 3) { hipLaunchKernelGGL(kA1, dim3(240), dim3(256), 0, 0, q); }
        if (which == 4) { hipLaunchKernelGGL(kA2, dim3(Bp), dim3(256), 0, 0, q); }
        if (which == 5) { hipLaunchKernelGGL(kA3, dim3(192), dim3(256), 0, 0, q); }
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, 1e3f * ms / reps);
    }
    return best;
  };
  (void)time_loop(1); (void)time_loop(2);
  const float t0 = time_loop(0), tA = time_loop(1), tB = time_loop(2), t1 = time_loop(3), t2 = time_loop(4), t3 = time_loop(5);
  CK(hipMemcpy(&to, q.timeout, sizeof(unsigned), hipMemcpyDeviceToHost));
  printf("touch alone %.2f us | three launches %.2f us (pre-head %.2f + head %.2f + backward W2 %.2f alone) | one launch with two row-group rendezvous %.2f us"
         " | saved %.2f us per iteration; time-out flag %u\n", t0, tA - t0, t1 - t0, t2 - t0, t3 - t0, tB - t0, tA - tB, to);
  return 0;
}
