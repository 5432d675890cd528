// bhg_vector.hip — fused multi-tensor vector recurrences of the hypergradient hot path.
//
// Replaces the per-tensor ATen launches of betty/hypergradient/cg.py:34-56,
// neumann.py:59-66 and darts.py:29-38,49-50,62-63 (plus betty/utils.py:117-118 to_vec)
// with a handful of HBM-streaming kernels over flat fp32 state vectors.  Written for
// gfx950 only: wave64 reductions, 16-B (dwordx4) coalesced accesses, fp64 partial sums,
// fixed-order two-stage reductions (no float atomics => run-to-run bitwise determinism).
//
// Arithmetic follows the reference's rounding sequence (a*b rounded, then +/-; never
// contracted into an fma) so that the only differences to the CPU autograd reference are
// the reduction order of the dot products (we accumulate in fp64) and the HVP itself.
#include <hip/hip_ext.h>
#include <stdlib.h>

#include <vector>

#include <mutex>

#include "bhg_common.hpp"

namespace bhg {
namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }

#define BHG_FOR4(stmt_x, stmt_y, stmt_z, stmt_w) \
  do {                                           \
    stmt_x;                                      \
    stmt_y;                                      \
    stmt_z;                                      \
    stmt_w;                                      \
  } while (0)

// ---- pointer-table writer (T > kInlineT) -----------------------------------------------
struct WriterArgs {
  const void* p[kWriterT];
};
__global__ void k_write_table(const void** dst, WriterArgs a, int count) {
  const int i = threadIdx.x;
  if (i < count) dst[i] = a.p[i];
}

// ---- flatten / scatter -------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_flatten(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                      int n_chunks, float* __restrict__ flat,
                                                      float scale) {
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* src = tab_ptr(tab, ck.tensor) + ck.src_off;
    float* dst = flat + ck.flat_off;
    float4 v[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) v[i] = ld4(src, 4 * (threadIdx.x + kThreads * i), ck.len);
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      float4 o = v[i];
      o.x = mul_rn(scale, o.x); o.y = mul_rn(scale, o.y); o.z = mul_rn(scale, o.z); o.w = mul_rn(scale, o.w);
      st4(dst, 4 * (threadIdx.x + kThreads * i), ck.len, o);
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_scatter(const float* __restrict__ flat, PtrTab tab,
                                                      const bhg_chunk* __restrict__ chunks,
                                                      int n_chunks, float scale) {
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    float* dst = tab_ptr(tab, ck.tensor) + ck.src_off;
    const float* src = flat + ck.flat_off;
    float4 v[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) v[i] = ld4(src, 4 * (threadIdx.x + kThreads * i), ck.len);
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      float4 o = v[i];
      o.x = mul_rn(scale, o.x); o.y = mul_rn(scale, o.y); o.z = mul_rn(scale, o.z); o.w = mul_rn(scale, o.w);
      st4(dst, 4 * (threadIdx.x + kThreads * i), ck.len, o);
    }
  }
}

__global__ __launch_bounds__(kThreads) void k_scale_flat(float* __restrict__ flat, int64_t n4, int64_t n,
                                                         float scale) {
  // n4 = number of whole float4; tail handled by the last thread range
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    float4 v = reinterpret_cast<float4*>(flat)[i];
    v.x = mul_rn(scale, v.x); v.y = mul_rn(scale, v.y); v.z = mul_rn(scale, v.z); v.w = mul_rn(scale, v.w);
    reinterpret_cast<float4*>(flat)[i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n - 4 * n4)) {
    const int64_t j = 4 * n4 + threadIdx.x;
    flat[j] = mul_rn(scale, flat[j]);
  }
}

// ---- Neumann ----------------------------------------------------------------------------------
// neumann.py:60  p = v (the reference aliases; we keep two flat buffers)
__global__ __launch_bounds__(kThreads) void k_neumann_init(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                           int n_chunks, float* __restrict__ v,
                                                           float* __restrict__ p) {
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* src = tab_ptr(tab, ck.tensor) + ck.src_off;
    float4 t[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) t[i] = ld4(src, 4 * (threadIdx.x + kThreads * i), ck.len);
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      st4(v + ck.flat_off, 4 * (threadIdx.x + kThreads * i), ck.len, t[i]);
      if (p) st4(p + ck.flat_off, 4 * (threadIdx.x + kThreads * i), ck.len, t[i]);   // p == NULL: see bhg_mlp_neumann_solve
    }
  }
}

// neumann.py:62-64 (+66 and the negation of 45/54 folded in when out_scale != 0):
//   v <- v - alpha*Hv ; p <- p + v ; [p <- out_scale * p]
// 20*N algorithmic bytes: read Hv, v, p; write v, p.
__global__ __launch_bounds__(kThreads) void k_neumann_step(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                           int n_chunks, float* __restrict__ v,
                                                           float* __restrict__ p, float alpha,
                                                           float out_scale, float shift) {
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* hsrc = tab_ptr(tab, ck.tensor) + ck.src_off;
    float* vv = v + ck.flat_off;
    float* pp = p + ck.flat_off;
    float4 h[kVecPerThread], a[kVecPerThread], b[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      h[i] = ld4(hsrc, e, ck.len);
      a[i] = ld4(vv, e, ck.len);
      b[i] = ld4(pp, e, ck.len);
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      if (shift != 0.f) {  // H v = (raw HVP) + shift * v  (diagonal part of the Hessian kept out of the producer)
        h[i].x = add_rn(h[i].x, mul_rn(shift, a[i].x)); h[i].y = add_rn(h[i].y, mul_rn(shift, a[i].y));
        h[i].z = add_rn(h[i].z, mul_rn(shift, a[i].z)); h[i].w = add_rn(h[i].w, mul_rn(shift, a[i].w));
      }
      float4 nv, np;
      nv.x = sub_rn(a[i].x, mul_rn(alpha, h[i].x)); nv.y = sub_rn(a[i].y, mul_rn(alpha, h[i].y));
      nv.z = sub_rn(a[i].z, mul_rn(alpha, h[i].z)); nv.w = sub_rn(a[i].w, mul_rn(alpha, h[i].w));
      np.x = add_rn(nv.x, b[i].x); np.y = add_rn(nv.y, b[i].y);
      np.z = add_rn(nv.z, b[i].z); np.w = add_rn(nv.w, b[i].w);
      if (out_scale != 0.f) {
        np.x = mul_rn(out_scale, np.x); np.y = mul_rn(out_scale, np.y);
        np.z = mul_rn(out_scale, np.z); np.w = mul_rn(out_scale, np.w);
      }
      st4(vv, e, ck.len, nv);
      st4(pp, e, ck.len, np);
    }
  }
}

// ---- CG: streaming variant (3 kernels per iteration, 40*N bytes) -------------------------------
// cg.py:34-36: x = 0, r = p = vector; also the first numerator r.r (cg.py:45).
__global__ __launch_bounds__(kThreads) void k_cg_init(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                      int n_chunks, float* __restrict__ x,
                                                      float* __restrict__ r, float* __restrict__ p,
                                                      double* __restrict__ partR0,
                                                      unsigned* __restrict__ barrier_words,
                                                      double* __restrict__ scal, unsigned long long keep_mask) {
  // keep_mask (bhg_cg_init_masked): bit t clear -> tensor t's slices of r and p are NOT written (only its share of r.r is taken):
  // a solver that reads those slices of the right-hand side from the caller's tensors and never touches the direction's
  __shared__ double red[kWaves];
  double acc = 0.0;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* src = tab_ptr(tab, ck.tensor) + ck.src_off;
    const bool keep = ck.tensor >= 64 || ((keep_mask >> ck.tensor) & 1ull) != 0ull;   // (workgroup-uniform)
    float4 t[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) t[i] = ld4(src, 4 * (threadIdx.x + kThreads * i), ck.len);
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      if (x) st4(x + ck.flat_off, e, ck.len, zero);   // x == NULL: the caller does not materialise the solution vector
      if (keep) {
        st4(r + ck.flat_off, e, ck.len, t[i]);
        st4(p + ck.flat_off, e, ck.len, t[i]);
      }
      acc += (double)t[i].x * t[i].x + (double)t[i].y * t[i].y + (double)t[i].z * t[i].z +
             (double)t[i].w * t[i].w;
    }
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partR0[blockIdx.x] = s;
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) barrier_words[threadIdx.x] = 0u;  // grid-barrier state of the resident kernel
    if (threadIdx.x < 16) scal[threadIdx.x] = (threadIdx.x == S_NPART0) ? (double)gridDim.x : 0.0;
  }
}

// K1: den = (cg_alpha*Hp).p   (cg.py:42,44,46) — reads Hp, p: 8*N bytes.
__global__ __launch_bounds__(kThreads) void k_cg_dot(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                     int n_chunks, const float* __restrict__ p,
                                                     float cg_alpha, float shift, double* __restrict__ partP) {
  __shared__ double red[kWaves];
  double acc = 0.0;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* hsrc = tab_ptr(tab, ck.tensor) + ck.src_off;
    float4 h[kVecPerThread], q[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      h[i] = ld4(hsrc, e, ck.len);
      q[i] = ld4(p + ck.flat_off, e, ck.len);
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      if (shift != 0.f) {
        h[i].x = add_rn(h[i].x, mul_rn(shift, q[i].x)); h[i].y = add_rn(h[i].y, mul_rn(shift, q[i].y));
        h[i].z = add_rn(h[i].z, mul_rn(shift, q[i].z)); h[i].w = add_rn(h[i].w, mul_rn(shift, q[i].w));
      }
      acc += (double)mul_rn(cg_alpha, h[i].x) * q[i].x + (double)mul_rn(cg_alpha, h[i].y) * q[i].y +
             (double)mul_rn(cg_alpha, h[i].z) * q[i].z + (double)mul_rn(cg_alpha, h[i].w) * q[i].w;
    }
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partP[blockIdx.x] = s;
}

// K2: a = rr/den ; r' = r - a*Hp (cg.py:47,50) ; partial r'.r' (cg.py:51-52)
// reads Hp, r; writes r: 12*N bytes.  x += a*p is deferred to K3, which reads p anyway.
__global__ __launch_bounds__(kThreads) void k_cg_resid(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                       int n_chunks, float* __restrict__ r,
                                                       const float* __restrict__ p, float shift,
                                                       const double* __restrict__ partP,
                                                       const double* __restrict__ partR_old,
                                                       double* __restrict__ partR_new, int n_part,
                                                       int iter, double* __restrict__ scal) {
  __shared__ double red[kWaves];
  const double rr = sum_partials(partR_old, (int)scal[S_NPART0 + (iter & 1)], red);
  const double den = sum_partials(partP, n_part, red);
  const float alpha = (float)rr / (float)den;  // fp32 divide of fp32 dots, as torch does
  double acc = 0.0;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* hsrc = tab_ptr(tab, ck.tensor) + ck.src_off;
    float* rrp = r + ck.flat_off;
    float4 h[kVecPerThread], a[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      h[i] = ld4(hsrc, e, ck.len);
      a[i] = ld4(rrp, e, ck.len);
      if (shift != 0.f) {  // (12+4)*N bytes in this case: the direction is re-read for the diagonal term
        const float4 q = ld4(p + ck.flat_off, e, ck.len);
        h[i].x = add_rn(h[i].x, mul_rn(shift, q.x)); h[i].y = add_rn(h[i].y, mul_rn(shift, q.y));
        h[i].z = add_rn(h[i].z, mul_rn(shift, q.z)); h[i].w = add_rn(h[i].w, mul_rn(shift, q.w));
      }
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      float4 nr;
      nr.x = sub_rn(a[i].x, mul_rn(alpha, h[i].x)); nr.y = sub_rn(a[i].y, mul_rn(alpha, h[i].y));
      nr.z = sub_rn(a[i].z, mul_rn(alpha, h[i].z)); nr.w = sub_rn(a[i].w, mul_rn(alpha, h[i].w));
      st4(rrp, e, ck.len, nr);
      acc += (double)nr.x * nr.x + (double)nr.y * nr.y + (double)nr.z * nr.z + (double)nr.w * nr.w;
    }
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    partR_new[blockIdx.x] = s;
    if (blockIdx.x == 0) {
      scal[S_RR_OLD] = rr;
      scal[S_PHP] = den;
      scal[S_ALPHA] = (double)alpha;
      scal[S_NPART0 + ((iter + 1) & 1)] = (double)gridDim.x;
    }
  }
}

// K3: b = r'.r'/rr ; x += a*p ; p = r' + b*p (cg.py:49,52,53) [x <- out_scale*x on the last step]
// reads r', p, x; writes x, p: 20*N bytes.
__global__ __launch_bounds__(kThreads) void k_cg_dir(const bhg_chunk* __restrict__ chunks, int n_chunks,
                                                     float* __restrict__ x, const float* __restrict__ r,
                                                     float* __restrict__ p,
                                                     const double* __restrict__ partR_new, int n_part,
                                                     float out_scale, double* __restrict__ scal,
                                                     unsigned* __restrict__ barrier_words, unsigned resident_arrivals) {
  __shared__ double red[kWaves];
  const double rr_new = sum_partials(partR_new, n_part, red);
  const double rr_old = scal[S_RR_OLD];
  const float alpha = (float)scal[S_ALPHA];
  const float beta = (float)rr_new / (float)rr_old;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    float4 a[kVecPerThread], q[kVecPerThread], xx[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      a[i] = ld4(r + ck.flat_off, e, ck.len);
      q[i] = ld4(p + ck.flat_off, e, ck.len);
      xx[i] = ld4(x + ck.flat_off, e, ck.len);
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      float4 nx, np;
      nx.x = add_rn(xx[i].x, mul_rn(alpha, q[i].x)); nx.y = add_rn(xx[i].y, mul_rn(alpha, q[i].y));
      nx.z = add_rn(xx[i].z, mul_rn(alpha, q[i].z)); nx.w = add_rn(xx[i].w, mul_rn(alpha, q[i].w));
      if (out_scale != 0.f) {
        nx.x = mul_rn(out_scale, nx.x); nx.y = mul_rn(out_scale, nx.y);
        nx.z = mul_rn(out_scale, nx.z); nx.w = mul_rn(out_scale, nx.w);
      }
      np.x = add_rn(a[i].x, mul_rn(beta, q[i].x)); np.y = add_rn(a[i].y, mul_rn(beta, q[i].y));
      np.z = add_rn(a[i].z, mul_rn(beta, q[i].z)); np.w = add_rn(a[i].w, mul_rn(beta, q[i].w));
      st4(x + ck.flat_off, e, ck.len, nx);
      st4(p + ck.flat_off, e, ck.len, np);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[S_RR_NEW] = rr_new;
    scal[S_BETA] = (double)beta;
    // the resident kernel's barrier targets count 2 arrivals per workgroup and COMPLETED iteration: credit this
    // streamed iteration, so a later iteration of the same solve may use the resident kernel
    barrier_words[0] += resident_arrivals;
  }
}

// ---- CG: register-resident variant (1 persistent kernel per iteration, 28*N bytes) --------------
// One 512-thread workgroup per CU (2 waves per SIMD => 256 VGPRs per lane); every thread keeps
// its float4s of Hp (later r') and of p in VGPRs across two grid-wide barriers, so Hp, p, r, x
// are each read exactly once and x, r, p written exactly once per iteration.
// Holds N <= gridDim * kResMax * 4096 elements (11.5 M on a 256-CU MI355X).
constexpr int kResThreads = 512;
constexpr int kResWaves = kResThreads / 64;
constexpr int kResV = kChunk / (kResThreads * 4);  // float4 per thread per chunk = 2
constexpr int kResMax = 11;   // chunks per workgroup, register-only instance; 2 x 11 x 2 float4 = 176 VGPRs of state (252 total, no spill)
// LDS-assisted instance (round 2): the direction slices of the first kResLds slots are parked in the CU's otherwise
// unused LDS (9 x 16 KiB = 144 KiB) instead of registers, which frees registers for more Hp/r' slots:
// 15 slots = 15 x 8 (Hp) + 6 x 8 (p in registers) = 168 VGPRs of state -> 15.7 M elements at 28*N bytes.
constexpr int kResMaxLds = 15;
constexpr int kResLds = 9;
// Hybrid instance (HYBRID = true): kResHyb resident slots (fewer than kResMaxLds: the streamed loops need registers of
// their own) + every further chunk streamed inside the same launch.  BHG_CG_AUTO uses it while at least half of the
// vector is resident (N <= 2 x kResHyb x G chunks = 29 M elements); beyond that the dedicated streaming kernels, with
// four 256-thread workgroups per CU, have the better memory parallelism.
constexpr int kResHyb = 14;

__device__ __forceinline__ double block_sum_res(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kResWaves; ++i) s += red[i];
  return s;
}

using gu32 = __attribute__((address_space(1))) unsigned;
using gf64 = __attribute__((address_space(1))) double;

// Split grid barrier + all-reduce of one double per workgroup: `grid_arrive` publishes this
// workgroup's partial and bumps the arrival counter, `grid_wait_sum` (later, after independent
// work has been issued) waits for all arrivals and returns the fixed-order sum, identical in
// every workgroup.  Partials travel as 8-byte agent-scope atomics on both sides (write-through
// store, L1-bypassing load: MI355X_MICROARCH.md "8-B agent atomics both sides"); the counter is
// monotonic and zeroed by k_cg_init, so targets depend only on `iter`.
__device__ __forceinline__ void grid_arrive(double block_value, double* part, unsigned* counter) {
  if (threadIdx.x == 0) {
    __hip_atomic_store((gf64*)(part + blockIdx.x), block_value, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add((gu32*)counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ double grid_wait_sum(const double* part, unsigned* counter, unsigned target,
                                                double* red, unsigned* timeout_word, unsigned spin_limit) {
  double a = 0.0;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load((gu32*)counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) {  // bounded spin: flag and fall through instead of hanging the GPU ...
        __hip_atomic_store((gu32*)timeout_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = __builtin_nan("");      // ... and poison the sum: a wrong answer must not look like a right one
        break;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)gridDim.x; i += kResThreads)
    a += __hip_atomic_load((gf64*)(part + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return block_sum_res(a, red);
}

// Streams one flat vector through the resident chunks with a kDepth-deep software prefetch:
// f(i, j, loaded_float4) -> float4 to store back.  The compiler cannot hoist the loads of chunk
// i+1 above the stores of chunk i to the same array itself, so the pipeline is explicit.
constexpr int kResDepth = 3;
constexpr unsigned kSpinLimit = 1u << 22;  // ~4 s of polling before a grid barrier gives up
template <int NSLOT, typename F>
__device__ __forceinline__ void resident_stream(float* __restrict__ vec, const bhg_chunk* __restrict__ chunks,
                                                int n_chunks, F f) {
  const int G = gridDim.x;
  float4 buf[kResDepth][kResV];
#pragma unroll
  for (int d = 0; d < kResDepth; ++d) {
    const int c = blockIdx.x + d * G;
#pragma unroll
    for (int j = 0; j < kResV; ++j) buf[d][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < NSLOT && c < n_chunks) {
      const bhg_chunk ck = chunks[c];
#pragma unroll
      for (int j = 0; j < kResV; ++j) buf[d][j] = ld4(vec + ck.flat_off, 4 * (threadIdx.x + kResThreads * j), ck.len);
    }
  }
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int c = blockIdx.x + i * G;
    float4 cur[kResV];
#pragma unroll
    for (int j = 0; j < kResV; ++j) cur[j] = buf[i % kResDepth][j];
    const int cn = c + kResDepth * G;
    if (i + kResDepth < NSLOT && cn < n_chunks) {
      const bhg_chunk ckn = chunks[cn];
#pragma unroll
      for (int j = 0; j < kResV; ++j)
        buf[i % kResDepth][j] = ld4(vec + ckn.flat_off, 4 * (threadIdx.x + kResThreads * j), ckn.len);
    }
    if (c < n_chunks) {
      const bhg_chunk ck = chunks[c];
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        st4(vec + ck.flat_off, e, ck.len, f(i, j, cur[j]));
      }
    }
  }
}

// HYBRID (round 2): chunks beyond the NSLOT x G resident ones are STREAMED inside the same launch — they take part in
// the same two grid-wide dots and are re-read like in the 3-kernel form (40 bytes per element instead of 28), so the
// kernel has no size limit any more: N = 20 M costs 28*15.7 M + 40*4.3 M bytes in ONE launch instead of 40*20 M in three.
template <int NSLOT, int NLDS, bool HYBRID>
__global__ __launch_bounds__(kResThreads, 2) void k_cg_resident(
    PtrTab tab, const bhg_chunk* __restrict__ chunks, int n_chunks, float* __restrict__ x,
    float* __restrict__ r, float* __restrict__ p, float cg_alpha, int iter, float out_scale, float shift,
    const double* __restrict__ partR_old, double* __restrict__ partR_new,
    double* __restrict__ partP, unsigned* __restrict__ barrier_words, double* __restrict__ scal,
    unsigned spin_limit) {
  __shared__ double red[kResWaves];
  extern __shared__ __attribute__((aligned(16))) float4 sq[];   // NLDS x kResV x 512 float4: parked direction slices
  const int G = gridDim.x;

  float4 h[NSLOT][kResV], q[NSLOT - NLDS][kResV];
  // direction slice of slot i: LDS for the first NLDS slots (conflict-free: consecutive lanes, consecutive 16 B), else
  // registers; i is a compile-time constant in every (fully unrolled) use
  auto Q = [&](int i, int j) -> float4 { return i < NLDS ? sq[(i * kResV + j) * kResThreads + threadIdx.x] : q[i - NLDS < 0 ? 0 : i - NLDS][j]; };
  // ---- phase 1: load Hp, p once (all loads in flight together); den = (cg_alpha*Hp).p
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int c = blockIdx.x + i * G;
    float4 qv[kResV];
#pragma unroll
    for (int j = 0; j < kResV; ++j) {
      h[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      qv[j] = h[i][j];
    }
    if (c < n_chunks) {
      const bhg_chunk ck = chunks[c];
      const float* hs = tab_ptr(tab, ck.tensor) + ck.src_off;
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        h[i][j] = ld4(hs, e, ck.len);
        qv[j] = ld4(p + ck.flat_off, e, ck.len);
      }
    }
#pragma unroll
    for (int j = 0; j < kResV; ++j) {
      if (i < NLDS) sq[(i * kResV + j) * kResThreads + threadIdx.x] = qv[j];   // each thread reads back only its own entries
      else q[i - NLDS < 0 ? 0 : i - NLDS][j] = qv[j];
    }
  }
  // numerator r.r from the previous producer (k_cg_init or the previous iteration); these
  // loads overlap the big ones above.
  double rr;
  {
    const int n_part_old = (int)scal[S_NPART0 + (iter & 1)];
    double a = 0.0;
    for (int i = threadIdx.x; i < n_part_old; i += kResThreads) a += partR_old[i];
    rr = block_sum_res(a, red);
  }
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
#pragma unroll
    for (int j = 0; j < kResV; ++j) {
      const float4 qq = Q(i, j);
      if (shift != 0.f) {  // H p = (raw HVP) + shift * p: the Hessian's diagonal part never round-trips through HBM
        h[i][j].x = add_rn(h[i][j].x, mul_rn(shift, qq.x)); h[i][j].y = add_rn(h[i][j].y, mul_rn(shift, qq.y));
        h[i][j].z = add_rn(h[i][j].z, mul_rn(shift, qq.z)); h[i][j].w = add_rn(h[i][j].w, mul_rn(shift, qq.w));
      }
      acc += (double)mul_rn(cg_alpha, h[i][j].x) * qq.x + (double)mul_rn(cg_alpha, h[i][j].y) * qq.y +
             (double)mul_rn(cg_alpha, h[i][j].z) * qq.z + (double)mul_rn(cg_alpha, h[i][j].w) * qq.w;
    }
  }
  if (HYBRID) {   // streamed chunks: (cg_alpha*Hp).p without keeping anything
    for (int c = NSLOT * G + blockIdx.x; c < n_chunks; c += G) {
      const bhg_chunk ck = chunks[c];
      const float* hs = tab_ptr(tab, ck.tensor) + ck.src_off;
      float4 hv[kResV], qv[kResV];
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        hv[j] = ld4(hs, e, ck.len);
        qv[j] = ld4(p + ck.flat_off, e, ck.len);
      }
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        if (shift != 0.f) {
          hv[j].x = add_rn(hv[j].x, mul_rn(shift, qv[j].x)); hv[j].y = add_rn(hv[j].y, mul_rn(shift, qv[j].y));
          hv[j].z = add_rn(hv[j].z, mul_rn(shift, qv[j].z)); hv[j].w = add_rn(hv[j].w, mul_rn(shift, qv[j].w));
        }
        acc += (double)mul_rn(cg_alpha, hv[j].x) * qv[j].x + (double)mul_rn(cg_alpha, hv[j].y) * qv[j].y +
               (double)mul_rn(cg_alpha, hv[j].z) * qv[j].z + (double)mul_rn(cg_alpha, hv[j].w) * qv[j].w;
      }
    }
  }
  grid_arrive(block_sum_res(acc, red), partP, barrier_words);
  const double den = grid_wait_sum(partP, barrier_words, (unsigned)G * (2u * iter + 1u), red, barrier_words + 1,
                                   spin_limit);
  const float alpha = (float)rr / (float)den;

  // ---- phase 2a: r' = r - a*Hp (kept in h, stored once) ; partial r'.r' ; ARRIVE
  acc = 0.0;
  resident_stream<NSLOT>(r, chunks, n_chunks, [&](int i, int j, float4 rv) {
    float4 nr;
    nr.x = sub_rn(rv.x, mul_rn(alpha, h[i][j].x)); nr.y = sub_rn(rv.y, mul_rn(alpha, h[i][j].y));
    nr.z = sub_rn(rv.z, mul_rn(alpha, h[i][j].z)); nr.w = sub_rn(rv.w, mul_rn(alpha, h[i][j].w));
    h[i][j] = nr;
    acc += (double)nr.x * nr.x + (double)nr.y * nr.y + (double)nr.z * nr.z + (double)nr.w * nr.w;
    return nr;
  });
  if (HYBRID) {   // streamed chunks: r' = r - a*Hp with Hp (and p for the shift) read again
    for (int c = NSLOT * G + blockIdx.x; c < n_chunks; c += G) {
      const bhg_chunk ck = chunks[c];
      const float* hs = tab_ptr(tab, ck.tensor) + ck.src_off;
      float4 hv[kResV], rv[kResV];
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        hv[j] = ld4(hs, e, ck.len);
        rv[j] = ld4(r + ck.flat_off, e, ck.len);
        if (shift != 0.f) {
          const float4 qq = ld4(p + ck.flat_off, e, ck.len);
          hv[j].x = add_rn(hv[j].x, mul_rn(shift, qq.x)); hv[j].y = add_rn(hv[j].y, mul_rn(shift, qq.y));
          hv[j].z = add_rn(hv[j].z, mul_rn(shift, qq.z)); hv[j].w = add_rn(hv[j].w, mul_rn(shift, qq.w));
        }
      }
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        float4 nr;
        nr.x = sub_rn(rv[j].x, mul_rn(alpha, hv[j].x)); nr.y = sub_rn(rv[j].y, mul_rn(alpha, hv[j].y));
        nr.z = sub_rn(rv[j].z, mul_rn(alpha, hv[j].z)); nr.w = sub_rn(rv[j].w, mul_rn(alpha, hv[j].w));
        st4(r + ck.flat_off, e, ck.len, nr);
        acc += (double)nr.x * nr.x + (double)nr.y * nr.y + (double)nr.z * nr.z + (double)nr.w * nr.w;
      }
    }
  }
  grid_arrive(block_sum_res(acc, red), partR_new, barrier_words);

  // ---- phase 2b (hides the second barrier): x += a*p [x <- out_scale*x on the last step]
  resident_stream<NSLOT>(x, chunks, n_chunks, [&](int i, int j, float4 xv) {
    const float4 qq = Q(i, j);
    float4 nx;
    nx.x = add_rn(xv.x, mul_rn(alpha, qq.x)); nx.y = add_rn(xv.y, mul_rn(alpha, qq.y));
    nx.z = add_rn(xv.z, mul_rn(alpha, qq.z)); nx.w = add_rn(xv.w, mul_rn(alpha, qq.w));
    if (out_scale != 0.f) {
      nx.x = mul_rn(out_scale, nx.x); nx.y = mul_rn(out_scale, nx.y);
      nx.z = mul_rn(out_scale, nx.z); nx.w = mul_rn(out_scale, nx.w);
    }
    return nx;
  });
  const double rr_new = grid_wait_sum(partR_new, barrier_words, (unsigned)G * (2u * iter + 2u), red,
                                      barrier_words + 1, spin_limit);
  const float beta = (float)rr_new / (float)rr;

  // ---- phase 3: p = r' + b*p (registers only -> store)
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int c = blockIdx.x + i * G;
    if (c < n_chunks) {
      const bhg_chunk ck = chunks[c];
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        const float4 qq = Q(i, j);
        float4 np;
        np.x = add_rn(h[i][j].x, mul_rn(beta, qq.x)); np.y = add_rn(h[i][j].y, mul_rn(beta, qq.y));
        np.z = add_rn(h[i][j].z, mul_rn(beta, qq.z)); np.w = add_rn(h[i][j].w, mul_rn(beta, qq.w));
        st4(p + ck.flat_off, e, ck.len, np);
      }
    }
  }
  if (HYBRID) {   // streamed chunks: x += a*p and p = r' + b*p in one pass (r', p, x read; x, p written), like k_cg_dir
    for (int c = NSLOT * G + blockIdx.x; c < n_chunks; c += G) {
      const bhg_chunk ck = chunks[c];
      float4 rv[kResV], qv[kResV], xv[kResV];
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        rv[j] = ld4(r + ck.flat_off, e, ck.len);
        qv[j] = ld4(p + ck.flat_off, e, ck.len);
        xv[j] = ld4(x + ck.flat_off, e, ck.len);
      }
#pragma unroll
      for (int j = 0; j < kResV; ++j) {
        const int e = 4 * (threadIdx.x + kResThreads * j);
        float4 nx, np;
        nx.x = add_rn(xv[j].x, mul_rn(alpha, qv[j].x)); nx.y = add_rn(xv[j].y, mul_rn(alpha, qv[j].y));
        nx.z = add_rn(xv[j].z, mul_rn(alpha, qv[j].z)); nx.w = add_rn(xv[j].w, mul_rn(alpha, qv[j].w));
        if (out_scale != 0.f) {
          nx.x = mul_rn(out_scale, nx.x); nx.y = mul_rn(out_scale, nx.y);
          nx.z = mul_rn(out_scale, nx.z); nx.w = mul_rn(out_scale, nx.w);
        }
        np.x = add_rn(rv[j].x, mul_rn(beta, qv[j].x)); np.y = add_rn(rv[j].y, mul_rn(beta, qv[j].y));
        np.z = add_rn(rv[j].z, mul_rn(beta, qv[j].z)); np.w = add_rn(rv[j].w, mul_rn(beta, qv[j].w));
        st4(x + ck.flat_off, e, ck.len, nx);
        st4(p + ck.flat_off, e, ck.len, np);
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    scal[S_RR_OLD] = rr;
    scal[S_PHP] = den;
    scal[S_ALPHA] = (double)alpha;
    scal[S_RR_NEW] = rr_new;
    scal[S_BETA] = (double)beta;
    scal[S_NPART0 + ((iter + 1) & 1)] = (double)gridDim.x;
  }
}

// ---- DARTS vector ops ------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_sqnorm(PtrTab tab, const bhg_chunk* __restrict__ chunks,
                                                     int n_chunks, double* __restrict__ part) {
  __shared__ double red[kWaves];
  double acc = 0.0;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* src = tab_ptr(tab, ck.tensor) + ck.src_off;
    float4 t[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) t[i] = ld4(src, 4 * (threadIdx.x + kThreads * i), ck.len);
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i)
      acc += (double)t[i].x * t[i].x + (double)t[i].y * t[i].y + (double)t[i].z * t[i].z +
             (double)t[i].w * t[i].w;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// darts.py:29-35: eps = R / (norm + 1e-15); the norm is an fp32 tensor in the reference and the
// division happens in Python floats (double), then eps is used as an fp32 `alpha`.
__global__ __launch_bounds__(kThreads) void k_darts_eps(const double* __restrict__ part, int n_part, double R,
                                                        double* __restrict__ out, float* __restrict__ eps_f32) {
  __shared__ double red[kWaves];
  const double ss = sum_partials(part, n_part, red);
  if (threadIdx.x == 0) {
    float nf = (float)sqrt(ss);
    nf = add_rn(nf, 1e-15f);
    const double eps = R / (double)nf;
    out[0] = ss;
    out[1] = eps;
    if (eps_f32) *eps_f32 = (float)eps;
  }
}

// darts.py:37-38,49-50,62-63: w_t += (mul * coef) * v_t, in place on the live weights.
__global__ __launch_bounds__(kThreads) void k_axpy_multi(PtrTab dst, PtrTab src,
                                                         const bhg_chunk* __restrict__ chunks, int n_chunks,
                                                         const float* __restrict__ coef_dev, float mul) {
  const float a = coef_dev ? mul_rn(mul, *coef_dev) : mul;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    float* d = tab_ptr(dst, ck.tensor) + ck.src_off;
    const float* s = tab_ptr(src, ck.tensor) + ck.src_off;
    float4 dv[kVecPerThread], sv[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      dv[i] = ld4(d, e, ck.len);
      sv[i] = ld4(s, e, ck.len);
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      float4 o;
      o.x = add_rn(dv[i].x, mul_rn(a, sv[i].x)); o.y = add_rn(dv[i].y, mul_rn(a, sv[i].y));
      o.z = add_rn(dv[i].z, mul_rn(a, sv[i].z)); o.w = add_rn(dv[i].w, mul_rn(a, sv[i].w));
      st4(d, e, ck.len, o);
    }
  }
}

// ---- SAMA: Adam-preconditioned direction (betty/hypergradient/utils.py:37-63) ---------------------------
// out = v * scale * lr,  scale = ((1-b1) b2 u_old - b1 (1-b2) g m_old) / (sqrt(u) + eps)^3,
// m_old = (m - (1-b1) g)/b1 (0 when b1 == 0),  u_old = (u - (1-b2) g g)/b2.
// One pass: read v, g, m, u (4 tensors lists), write the flat preconditioned vector: 20*N bytes.
struct SamaCoef { float one_m_b1, b1, one_m_b2, b2, c1 /*(1-b1)*b2*/, c2 /*b1*(1-b2)*/, eps, lr; };
__global__ __launch_bounds__(kThreads) void k_sama_adam(PtrTab tv, PtrTab tg, PtrTab tm, PtrTab tu,
                                                        const bhg_chunk* __restrict__ chunks, int n_chunks,
                                                        float* __restrict__ out, SamaCoef k) {
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const bhg_chunk ck = chunks[c];
    const float* v = tab_ptr(tv, ck.tensor) + ck.src_off;
    const float* g = tab_ptr(tg, ck.tensor) + ck.src_off;
    const float* m = tab_ptr(tm, ck.tensor) + ck.src_off;
    const float* u = tab_ptr(tu, ck.tensor) + ck.src_off;
    float4 av[kVecPerThread], ag[kVecPerThread], am[kVecPerThread], au[kVecPerThread];
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      av[i] = ld4(v, e, ck.len); ag[i] = ld4(g, e, ck.len); am[i] = ld4(m, e, ck.len); au[i] = ld4(u, e, ck.len);
    }
#pragma unroll
    for (int i = 0; i < kVecPerThread; ++i) {
      const int e = 4 * (threadIdx.x + kThreads * i);
      const float vv[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
      const float gg[4] = {ag[i].x, ag[i].y, ag[i].z, ag[i].w};
      const float mm[4] = {am[i].x, am[i].y, am[i].z, am[i].w};
      const float uu[4] = {au[i].x, au[i].y, au[i].z, au[i].w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // same op order / roundings as the reference's ATen expression
        const float m_old = k.b1 != 0.f ? sub_rn(mm[j], mul_rn(k.one_m_b1, gg[j])) / k.b1 : 0.f;
        const float u_old = sub_rn(uu[j], mul_rn(mul_rn(k.one_m_b2, gg[j]), gg[j])) / k.b2;
        float sc = sub_rn(mul_rn(k.c1, u_old), mul_rn(mul_rn(k.c2, gg[j]), m_old));
        const float d = add_rn(sqrtf(uu[j]), k.eps);
        sc = sc / mul_rn(mul_rn(d, d), d);
        o[j] = mul_rn(mul_rn(vv[j], sc), k.lr);
      }
      st4(out + ck.flat_off, e, ck.len, make_float4(o[0], o[1], o[2], o[3]));
    }
  }
}

inline int grid_for(int n_chunks) { return n_chunks < kMaxBlocks ? (n_chunks > 0 ? n_chunks : 1) : kMaxBlocks; }

// Per-device caches (a process may drive several GPUs): index = current HIP device, -1 = not probed yet.
constexpr int kMaxDevices = 64;
int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return dev >= 0 && dev < kMaxDevices ? dev : -1;
}
int num_cus() {
  static int cus[kMaxDevices];
  static bool init = false;
  if (!init) { for (int i = 0; i < kMaxDevices; ++i) cus[i] = -1; init = true; }
  const int dev = current_device();
  if (dev < 0) return 0;
  if (cus[dev] >= 0) return cus[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cus[dev] = 0;
  }
  return cus[dev] = prop.multiProcessorCount;
}

}  // namespace

// ---- optional per-launch timing (bench.py's roofline leg) ---------------------------------------------
// When enabled, the recurrence launches carry start/stop events attached to the kernels themselves
// (hipExtLaunchKernelGGL), i.e. kernel begin -> kernel end on the launch stream, the same interval
// rocprofv3 --kernel-trace reports.  Off by default: no events, no overhead.
namespace {
struct TimedSpan { hipEvent_t a, b; int kind; };
bool g_timing = false;
std::vector<TimedSpan> g_spans;
std::vector<hipEvent_t> g_free_events;
constexpr size_t kMaxSpans = 16384;
hipEvent_t take_event() {
  if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace
// Returns true (and fills a/b) when this launch group should be timed.
bool span_begin(int kind, hipEvent_t* a, hipEvent_t* b) {  // declared in bhg_common.hpp
  *a = *b = nullptr;
  if (!g_timing || g_spans.size() >= kMaxSpans) return false;
  *a = take_event(); *b = take_event();
  if (!*a || !*b) return false;
  g_spans.push_back({*a, *b, kind});
  return true;
}


int make_table(PtrTab* out, const void* const* host_ptrs, int T, void* ws, int slot, hipStream_t stream) {
  memset(out, 0, sizeof(*out));
  if (T <= kInlineT) {
    for (int i = 0; i < T; ++i) out->inl[i] = host_ptrs[i];
    out->dev = nullptr;
    return BHG_OK;
  }
  if (!ws) {
    set_error("make_table: workspace required for T=%d > %d", T, kInlineT);
    return BHG_ERR_WS;
  }
  const void** dev = reinterpret_cast<const void**>(static_cast<char*>(ws) + kWsTables) + (size_t)slot * T;
  for (int off = 0; off < T; off += kWriterT) {
    WriterArgs a;
    const int cnt = (T - off) < kWriterT ? (T - off) : kWriterT;
    for (int i = 0; i < cnt; ++i) a.p[i] = host_ptrs[off + i];
    for (int i = cnt; i < kWriterT; ++i) a.p[i] = nullptr;
    hipLaunchKernelGGL(k_write_table, dim3(1), dim3(kWriterT), 0, stream, dev + off, a, cnt);
  }
  BHG_HIP_CHECK(hipGetLastError());
  out->dev = dev;
  return BHG_OK;
}

}  // namespace bhg

using namespace bhg;

#define BHG_COMMON_CHECKS(tabptr)                                          \
  BHG_REQUIRE(tabptr != nullptr || T == 0, "tensor table is NULL");        \
  BHG_REQUIRE(T >= 0 && n_chunks >= 0, "negative size");                   \
  BHG_REQUIRE(chunks_dev != nullptr || n_chunks == 0, "chunk table is NULL")

extern "C" {

int bhg_flatten(const void* const* src, int T, const bhg_chunk* chunks_dev, int n_chunks, float* flat,
                float scale, void* ws, void* stream) {
  BHG_COMMON_CHECKS(src);
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(flat, "flat is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tab;
  if (int rc = make_table(&tab, src, T, ws, 0, st)) return rc;
  hipLaunchKernelGGL(k_flatten, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks,
                     flat, scale);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_scatter(const float* flat, void* const* dst, int T, const bhg_chunk* chunks_dev, int n_chunks,
                float scale, void* ws, void* stream) {
  BHG_COMMON_CHECKS(dst);
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(flat, "flat is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tab;
  if (int rc = make_table(&tab, const_cast<const void* const*>(dst), T, ws, 0, st)) return rc;
  hipLaunchKernelGGL(k_scatter, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, flat, tab, chunks_dev,
                     n_chunks, scale);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_scale_flat(float* flat, int64_t n, float scale, void* stream) {
  BHG_REQUIRE(n >= 0, "negative size");
  if (n == 0) return BHG_OK;
  BHG_REQUIRE(flat, "flat is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + kThreads - 1) / kThreads;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_scale_flat, dim3((unsigned)blocks), dim3(kThreads), 0, st, flat, n4, n, scale);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_neumann_init(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks, float* v,
                     float* p, void* ws, void* stream) {
  BHG_COMMON_CHECKS(vec);
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(v, "state vector is NULL");   // p may be NULL (accumulator-free fused solve)
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tab;
  if (int rc = make_table(&tab, vec, T, ws, 0, st)) return rc;
  hipLaunchKernelGGL(k_neumann_init, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, tab, chunks_dev,
                     n_chunks, v, p);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_neumann_step(const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks, float* v,
                     float* p, float alpha, float out_scale, float hvp_shift, void* ws, void* stream) {
  BHG_COMMON_CHECKS(hvp);
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(v && p, "state vector is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tab;
  if (int rc = make_table(&tab, hvp, T, ws, 0, st)) return rc;
  hipEvent_t ea, eb;
  const bool timed = span_begin(BHG_TIMING_NEUMANN_STEP, &ea, &eb);
  hipExtLaunchKernelGGL(k_neumann_step, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, timed ? ea : nullptr,
                        timed ? eb : nullptr, 0, tab, chunks_dev, n_chunks, v, p, alpha, out_scale, hvp_shift);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_cg_init(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks, float* x, float* r,
                float* p, void* ws, void* stream) {
  return bhg_cg_init_masked(vec, T, chunks_dev, n_chunks, x, r, p, ~0ull, ws, stream);
}

int bhg_cg_init_masked(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks, float* x, float* r,
                       float* p, unsigned long long keep_mask, void* ws, void* stream) {
  BHG_COMMON_CHECKS(vec);
  BHG_REQUIRE(ws, "workspace is NULL");
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(r && p, "state vector is NULL");   // x may be NULL (see bhg_mlp_cg_solve)
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tab;
  if (int rc = make_table(&tab, vec, T, ws, 0, st)) return rc;
  char* w = static_cast<char*>(ws);
  hipLaunchKernelGGL(k_cg_init, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks, x,
                     r, p, reinterpret_cast<double*>(w + kWsPartR),
                     reinterpret_cast<unsigned*>(w + kWsBarrier), reinterpret_cast<double*>(w + kWsScal), keep_mask);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// bhg_debug_set("cg_spin_limit", n) overrides the barrier's polling bound (tests force a time-out with 0).
static unsigned spin_limit() { return dbg_is_set(DBG_cg_spin_limit) ? (unsigned)dbg(DBG_cg_spin_limit, 0) : kSpinLimit; }

// The LDS-assisted and hybrid resident instances park direction slices in 144 KiB of dynamic LDS per workgroup.  Probed
// once per device (mutex-guarded: the first CG steps of two host threads may race): the attribute must be granted for
// both instances AND the device must report that much LDS per workgroup.
static bool resident_lds_instances_ok() {
  static std::mutex mu;
  static signed char per_device[kMaxDevices];   // 0 unknown, 1 ok, -1 unavailable
  const int dev = current_device();
  if (dev < 0 || dev >= kMaxDevices) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (per_device[dev] == 0) {
    constexpr size_t lds = (size_t)kResLds * kResV * kResThreads * sizeof(float4);
    int max_lds = 0;
    bool ok = hipDeviceGetAttribute(&max_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && optin > max_lds) max_lds = optin;
    ok = ok && (size_t)max_lds >= lds;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_cg_resident<kResMaxLds, kResLds, false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(k_cg_resident<kResHyb, kResLds, true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    per_device[dev] = ok ? 1 : -1;
  }
  return per_device[dev] == 1;
}

int bhg_cg_resident_capacity_chunks(void) { return num_cus() * 2 * kResHyb; }

// One-time residency census (MI355X_MICROARCH.md "Residency and cooperative launch": size grid-barrier
// grids from measured residency, never from the occupancy API alone): launch the real resident kernel on
// an EMPTY problem — every workgroup only walks the two grid barriers — with a short spin limit.  If all
// `num_cus()` workgroups are not co-resident (CU masking, a shared GPU, a partitioned device) the barrier
// times out, the flag is read back and BHG_CG_AUTO never picks the resident variant in this process.
int bhg_cg_resident_ok(void) {
  static int per_device[kMaxDevices];
  static bool init = false;
  if (!init) { for (int i = 0; i < kMaxDevices; ++i) per_device[i] = -1; init = true; }
  const int dev = current_device();
  if (dev < 0) return 0;
  int& cached = per_device[dev];
  if (cached >= 0) return cached;
  const int G = num_cus();
  if (G <= 0) return cached = 0;
  void* ws = nullptr;
  const size_t bytes = ws_bytes(0);
  if (hipMalloc(&ws, bytes) != hipSuccess) { (void)hipGetLastError(); return cached = 0; }
  int ok = 0;
  do {
    if (hipMemset(ws, 0, bytes) != hipSuccess) break;
    char* w = static_cast<char*>(ws);
    double* partR = reinterpret_cast<double*>(w + kWsPartR);
    PtrTab tab;
    memset(&tab, 0, sizeof(tab));
    hipLaunchKernelGGL((k_cg_resident<kResMax, 0, false>), dim3(G), dim3(kResThreads), 0, nullptr, tab, (const bhg_chunk*)nullptr, 0,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, 1.0f, 0, 0.0f, 0.0f,
                       (const double*)partR, partR + kMaxBlocks, reinterpret_cast<double*>(w + kWsPartP),
                       reinterpret_cast<unsigned*>(w + kWsBarrier), reinterpret_cast<double*>(w + kWsScal),
                       1u << 16);
    if (hipGetLastError() != hipSuccess) break;
    unsigned words[2] = {0, 1};
    if (hipMemcpy(words, w + kWsBarrier, sizeof(words), hipMemcpyDeviceToHost) != hipSuccess) break;
    ok = (words[1] == 0 && words[0] == 2u * (unsigned)G) ? 1 : 0;
  } while (0);
  (void)hipFree(ws);
  (void)hipGetLastError();
  return cached = ok;
}

int bhg_cg_resident_usable(int n_chunks) {
  const int cap = bhg_cg_resident_capacity_chunks();
  if (n_chunks <= 0 || cap <= 0 || n_chunks > cap || !bhg_cg_resident_ok()) return 0;
  // beyond the register-only size the resident instances need 144 KiB of dynamic LDS per workgroup: on a device or
  // partition that does not grant it, AUTO streams instead of failing the step
  if (n_chunks > num_cus() * kResMax && !resident_lds_instances_ok()) return 0;
  return 1;
}

const double* bhg_cg_scalars_dev(const void* ws) {
  return reinterpret_cast<const double*>(static_cast<const char*>(ws) + kWsScal);
}

const unsigned* bhg_cg_timeout_flag_dev(const void* ws) {
  return reinterpret_cast<const unsigned*>(static_cast<const char*>(ws) + kWsBarrier) + 1;
}

int bhg_cg_step(const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks, float* x, float* r,
                float* p, float cg_alpha, int iter, float out_scale, float hvp_shift, int variant, void* ws,
                void* stream) {
  BHG_COMMON_CHECKS(hvp);
  BHG_REQUIRE(ws, "workspace is NULL");
  BHG_REQUIRE(iter >= 0, "negative iteration index");
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(x && r && p, "state vector is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int cap = bhg_cg_resident_capacity_chunks();
  if (variant == BHG_CG_AUTO) variant = bhg_cg_resident_usable(n_chunks) ? BHG_CG_RESIDENT : BHG_CG_STREAM;
  if (variant == BHG_CG_RESIDENT && n_chunks > cap) {
    set_error("bhg_cg_step: %d chunks exceed the resident capacity of %d", n_chunks, cap);
    return BHG_ERR_CAPACITY;
  }
  BHG_REQUIRE(variant == BHG_CG_STREAM || variant == BHG_CG_RESIDENT, "unknown variant");
  PtrTab tab;
  if (int rc = make_table(&tab, hvp, T, ws, 0, st)) return rc;
  char* w = static_cast<char*>(ws);
  double* scal = reinterpret_cast<double*>(w + kWsScal);
  double* partP = reinterpret_cast<double*>(w + kWsPartP);
  double* partR = reinterpret_cast<double*>(w + kWsPartR);
  // The producer of r.r for iteration `iter` wrote partR[iter & 1]; this iteration writes the other.
  double* partR_old = partR + (size_t)(iter & 1) * kMaxBlocks;
  double* partR_new = partR + (size_t)((iter + 1) & 1) * kMaxBlocks;
  // How many partials the previous producer wrote travels in scal[S_NPART0 + parity], and a streamed
  // iteration credits the resident kernel's arrival counter (k_cg_dir), so the stream and resident
  // variants may be mixed between iterations of one solve.
  const int n_stream = grid_for(n_chunks);
  hipEvent_t ea, eb;
  const bool timed = span_begin(BHG_TIMING_CG_STEP, &ea, &eb);
  if (variant == BHG_CG_STREAM) {
    // start event rides on the first kernel, stop event on the last: the span is the whole
    // iteration's recurrence including the two inter-kernel boundaries.
    hipExtLaunchKernelGGL(k_cg_dot, dim3(n_stream), dim3(kThreads), 0, st, timed ? ea : nullptr, nullptr, 0, tab,
                          chunks_dev, n_chunks, (const float*)p, cg_alpha, hvp_shift, partP);
    hipLaunchKernelGGL(k_cg_resid, dim3(n_stream), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks, r,
                       (const float*)p, hvp_shift, (const double*)partP, (const double*)partR_old, partR_new, n_stream, iter, scal);
    hipExtLaunchKernelGGL(k_cg_dir, dim3(n_stream), dim3(kThreads), 0, st, nullptr, timed ? eb : nullptr, 0,
                          chunks_dev, n_chunks, x, (const float*)r, p, (const double*)partR_new, n_stream,
                          out_scale, scal, reinterpret_cast<unsigned*>(w + kWsBarrier), 2u * (unsigned)num_cus());
  } else {
    const int G = num_cus();
    if (n_chunks <= G * kResMax) {   // register-only instance: fastest while it fits (11.5 M elements)
      hipExtLaunchKernelGGL((k_cg_resident<kResMax, 0, false>), dim3(G), dim3(kResThreads), 0, st, timed ? ea : nullptr,
                            timed ? eb : nullptr, 0, tab, chunks_dev, n_chunks, x, r, p, cg_alpha, iter, out_scale,
                            hvp_shift, (const double*)partR_old, partR_new, partP,
                            reinterpret_cast<unsigned*>(w + kWsBarrier), scal, spin_limit());
    } else {
      constexpr size_t lds = (size_t)kResLds * kResV * kResThreads * sizeof(float4);
      BHG_REQUIRE(resident_lds_instances_ok(), "this device does not grant the resident kernel's 144 KiB of dynamic LDS");
      if (n_chunks <= G * kResMaxLds)   // LDS-assisted instance: 9 direction slices per workgroup parked in LDS (15.7 M elements)
        hipExtLaunchKernelGGL((k_cg_resident<kResMaxLds, kResLds, false>), dim3(G), dim3(kResThreads), lds, st, timed ? ea : nullptr,
                              timed ? eb : nullptr, 0, tab, chunks_dev, n_chunks, x, r, p, cg_alpha, iter, out_scale,
                              hvp_shift, (const double*)partR_old, partR_new, partP,
                              reinterpret_cast<unsigned*>(w + kWsBarrier), scal, spin_limit());
      else                              // hybrid instance: 14 resident slots per workgroup, the rest streamed in the same launch
        hipExtLaunchKernelGGL((k_cg_resident<kResHyb, kResLds, true>), dim3(G), dim3(kResThreads), lds, st, timed ? ea : nullptr,
                              timed ? eb : nullptr, 0, tab, chunks_dev, n_chunks, x, r, p, cg_alpha, iter, out_scale,
                              hvp_shift, (const double*)partR_old, partR_new, partP,
                              reinterpret_cast<unsigned*>(w + kWsBarrier), scal, spin_limit());
    }
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- phased streaming CG iteration: the same three kernels as BHG_CG_STREAM, one call per phase, so that a caller
// whose state vectors are SHARDED over ranks (global-HVP mode, betty_amd/global_hvp.py) can all-reduce(SUM) the
// per-block partials between the phases: the consumer kernels then sum the all-reduced partials = the global dots,
// identically on every rank.  All ranks must use equally sized shards (same chunk count => same partial count).
//   phase 0: partP  <- partials of (cg_alpha*Hp).p            -> all-reduce bhg_cg_partials_dev(ws, 0, iter)
//   phase 1: alpha, r' = r - alpha*Hp, partR <- partials r'.r' -> all-reduce bhg_cg_partials_dev(ws, 1, iter)
//   phase 2: beta, x += alpha*p, p = r' + beta*p
// After bhg_cg_init the r.r partials are bhg_cg_partials_dev(ws, 2, 0) (all-reduce them as well).
int bhg_cg_phase(int phase, const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks, float* x, float* r,
                 float* p, float cg_alpha, int iter, float out_scale, float hvp_shift, void* ws, void* stream) {
  BHG_COMMON_CHECKS(hvp);
  BHG_REQUIRE(ws, "workspace is NULL");
  BHG_REQUIRE(iter >= 0 && phase >= 0 && phase <= 2, "bad phase / iteration index");
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(x && r && p, "state vector is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  double* scal = reinterpret_cast<double*>(w + kWsScal);
  double* partP = reinterpret_cast<double*>(w + kWsPartP);
  double* partR = reinterpret_cast<double*>(w + kWsPartR);
  double* partR_old = partR + (size_t)(iter & 1) * kMaxBlocks;
  double* partR_new = partR + (size_t)((iter + 1) & 1) * kMaxBlocks;
  const int n_stream = grid_for(n_chunks);
  if (phase == 2) {
    hipLaunchKernelGGL(k_cg_dir, dim3(n_stream), dim3(kThreads), 0, st, chunks_dev, n_chunks, x, (const float*)r, p,
                       (const double*)partR_new, n_stream, out_scale, scal, reinterpret_cast<unsigned*>(w + kWsBarrier),
                       2u * (unsigned)num_cus());
    BHG_HIP_CHECK(hipGetLastError());
    return BHG_OK;
  }
  PtrTab tab;
  if (int rc = make_table(&tab, hvp, T, ws, 0, st)) return rc;
  if (phase == 0)
    hipLaunchKernelGGL(k_cg_dot, dim3(n_stream), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks, (const float*)p, cg_alpha,
                       hvp_shift, partP);
  else
    hipLaunchKernelGGL(k_cg_resid, dim3(n_stream), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks, r, (const float*)p,
                       hvp_shift, (const double*)partP, (const double*)partR_old, partR_new, n_stream, iter, scal);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

double* bhg_cg_partials_dev(void* ws, int which, int iter) {
  char* w = static_cast<char*>(ws);
  if (which == 0) return reinterpret_cast<double*>(w + kWsPartP);
  double* partR = reinterpret_cast<double*>(w + kWsPartR);
  if (which == 1) return partR + (size_t)((iter + 1) & 1) * kMaxBlocks;
  return partR;   // which == 2: the r.r partials bhg_cg_init wrote (consumed by iteration 0)
}

int bhg_cg_partials_count(void) { return kMaxBlocks; }

int bhg_darts_eps(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks, double R,
                  double* out_dev, float* eps_f32_dev, void* ws, void* stream) {
  BHG_COMMON_CHECKS(vec);
  BHG_REQUIRE(ws && out_dev, "workspace/out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* w = static_cast<char*>(ws);
  double* part = reinterpret_cast<double*>(w + kWsPartP);
  int n_part = 0;
  if (n_chunks > 0) {
    PtrTab tab;
    if (int rc = make_table(&tab, vec, T, ws, 0, st)) return rc;
    n_part = grid_for(n_chunks);
    hipLaunchKernelGGL(k_sqnorm, dim3(n_part), dim3(kThreads), 0, st, tab, chunks_dev, n_chunks, part);
  }
  hipLaunchKernelGGL(k_darts_eps, dim3(1), dim3(kThreads), 0, st, (const double*)part, n_part, R, out_dev,
                     eps_f32_dev);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_axpy_multi(void* const* dst, const void* const* src, int T, const bhg_chunk* chunks_dev, int n_chunks,
                   const float* coef_dev, float mul, void* ws, void* stream) {
  BHG_COMMON_CHECKS(dst);
  BHG_REQUIRE(src != nullptr || T == 0, "src table is NULL");
  if (n_chunks == 0) return BHG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab td, ts;
  if (int rc = make_table(&td, const_cast<const void* const*>(dst), T, ws, 0, st)) return rc;
  if (int rc = make_table(&ts, src, T, ws, 1, st)) return rc;
  hipLaunchKernelGGL(k_axpy_multi, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, td, ts, chunks_dev,
                     n_chunks, coef_dev, mul);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_timing_enable(int on) {
  // (re)start or stop collecting; starting drops any spans not yet read
  for (auto& sp : g_spans) { g_free_events.push_back(sp.a); g_free_events.push_back(sp.b); }
  g_spans.clear();
  g_timing = on != 0;
  return BHG_OK;
}

int bhg_timing_read(int kind, double* total_ms, int* launches) {
  BHG_REQUIRE(total_ms && launches, "NULL output");
  double tot = 0.0;
  int n = 0;
  for (auto& sp : g_spans) {
    if (sp.kind != kind) continue;
    BHG_HIP_CHECK(hipEventSynchronize(sp.b));
    float ms = 0.f;
    BHG_HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
    tot += (double)ms;
    ++n;
  }
  *total_ms = tot;
  *launches = n;
  return BHG_OK;
}

int bhg_sama_adam_precondition(const void* const* vec, const void* const* last_grad, const void* const* exp_avg,
                               const void* const* exp_avg_sq, int T, const bhg_chunk* chunks_dev, int n_chunks,
                               float* out_flat, double beta1, double beta2, double eps, double lr, void* ws,
                               void* stream) {
  BHG_COMMON_CHECKS(vec);
  BHG_REQUIRE(last_grad && exp_avg && exp_avg_sq, "NULL state table");
  BHG_REQUIRE(beta2 != 0.0, "beta2 must be non-zero");
  if (n_chunks == 0) return BHG_OK;
  BHG_REQUIRE(out_flat, "out is NULL");
  hipStream_t st = static_cast<hipStream_t>(stream);
  PtrTab tv, tg, tm, tu;
  if (int rc = make_table(&tv, vec, T, ws, 0, st)) return rc;
  if (int rc = make_table(&tg, last_grad, T, ws, 1, st)) return rc;
  if (int rc = make_table(&tm, exp_avg, T, ws, 2, st)) return rc;
  if (int rc = make_table(&tu, exp_avg_sq, T, ws, 3, st)) return rc;
  // Python-float (double) products first, then one rounding to fp32 — what `python_scalar * tensor` does
  SamaCoef k;
  k.one_m_b1 = (float)(1.0 - beta1); k.b1 = (float)beta1;
  k.one_m_b2 = (float)(1.0 - beta2); k.b2 = (float)beta2;
  k.c1 = (float)((1.0 - beta1) * beta2); k.c2 = (float)(beta1 * (1.0 - beta2));
  k.eps = (float)eps; k.lr = (float)lr;
  hipLaunchKernelGGL(k_sama_adam, dim3(grid_for(n_chunks)), dim3(kThreads), 0, st, tv, tg, tm, tu, chunks_dev,
                     n_chunks, out_flat, k);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

}  // extern "C"
