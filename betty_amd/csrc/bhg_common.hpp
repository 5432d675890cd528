// bhg_common.hpp — shared device/host helpers for libbhg (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "bhg.h"

namespace bhg {

constexpr int kThreads = 256;             // 4 waves of 64
constexpr int kWaves = kThreads / 64;
constexpr int kChunk = BHG_CHUNK_ELEMS;   // 4096 fp32 = 16 KiB per vector per chunk
constexpr int kVecPerThread = kChunk / (kThreads * 4);  // float4 per thread per chunk = 4
constexpr int kMaxBlocks = 1024;          // 4 blocks per CU x 256 CUs
constexpr int kInlineT = 32;              // tensor pointers passed inline in kernargs
constexpr int kWriterT = 448;             // pointers per table-writer launch (< 4 KiB kernarg)

// ---- workspace layout (byte offsets) -------------------------------------------
constexpr size_t kWsScal = 0;                                 // 16 doubles
constexpr size_t kWsPartP = 128;                              // kMaxBlocks doubles
constexpr size_t kWsPartR = kWsPartP + 8 * kMaxBlocks;        // 2 x kMaxBlocks doubles
constexpr size_t kWsBarrier = kWsPartR + 16 * kMaxBlocks;     // 64 x u32
constexpr size_t kWsTables = kWsBarrier + 256;                // 4 x T pointers
inline size_t ws_bytes(int T) {
  size_t b = kWsTables + 4 * sizeof(void*) * (size_t)(T > 0 ? T : 0);
  return (b + 255) & ~(size_t)255;
}

// scalar slots
enum { S_RR_OLD = 0, S_PHP = 1, S_ALPHA = 2, S_RR_NEW = 3, S_BETA = 4, S_PP = 5 /* fused solver: p.p of the coming direction */, S_ALPHA_RING = 10 /* and 11: fused solver, alpha of iteration k in slot 10 + (k & 1) */,
       S_NPART0 = 8 /* and 9 */ };

// ---- debug / A-B switches ----------------------------------------------------------
// The library reads NO environment variable.  Every measurement arm and every test-only behaviour is an int in this table,
// unset by default, changed only through bhg_debug_set() (include/bhg.h) — so the product's behaviour cannot depend on a stray
// variable in a user's shell.  dbg(key, dflt): the value, or `dflt` while the key is unset.  Read on every call.
#define BHG_DBG_KEYS(X)                                                                                                      \
  X(mlp_wsk) X(mlp_wsk_maxk) X(mlp_wsk_depth) X(mlp_wsk_lds) X(wsl_depth) X(no_nt_slabs) X(gemm_no_xpose) X(mlp_no_fast)      \
  X(mlp_tn) X(split_target) X(split_cap) X(mlp_no_head) X(gram_ksplit) X(gram_kchunk) X(proj_graw_split) X(mlp_proj)         \
  X(mlp_hoist) X(hoist_wgs) X(proj_step_alone) X(mlp_no_side) X(mlp_no_fuse) X(mlp_no_outer_all) X(neumann_side)             \
  X(hoist_staged_mink) X(proj_alpha_alone) X(proj_small_alone) X(outer_order_by_work) X(outer_no_pre) X(outer_stagger)       \
  X(mlp_no_fused_solve) X(cg_eager_p) X(cg_x_every_iter) X(neumann_p_every_iter) X(head_no_prefetch) X(cg_spin_limit)        \
  X(packed_chain) X(packed_depth) X(packed_gram) X(proj_max_ratio) X(proj_ws_cap_mb) X(alpha_in_hoist) X(graw_v2) X(wskp_ragged) X(pstep_v2) X(pstep_unroll) X(rnew_in_graw) X(graw_cols) X(lin_first) X(lin_prio) X(lin_nub) X(lin_order) X(lin_update_next) X(lin_withhold_beta) X(neumann_vnew) X(cg_rhs_direct) X(packed_prepare) X(lin_deep) X(graw_kloop) X(lin_update_in_head) X(xcd_pairs) X(headu_head_first) X(fx_ksplit)
enum DbgKey : int {
#define BHG_DBG_ENUM(n) DBG_##n,
  BHG_DBG_KEYS(BHG_DBG_ENUM)
#undef BHG_DBG_ENUM
  DBG_COUNT
};
constexpr int kDbgUnset = INT32_MIN;
#ifdef BHG_AB
// measurement build (make ab -> libbhg_ab.so): the table is live; every-arm tests and `bench.py --debug` load THIS library
extern int g_dbg[DBG_COUNT];
inline bool dbg_is_set(DbgKey k) { return g_dbg[k] != kDbgUnset; }
inline int dbg(DbgKey k, int dflt) { return g_dbg[k] == kDbgUnset ? dflt : g_dbg[k]; }
#else
// PRODUCT build (libbhg.so, round 5): no table, no switch — every dbg(key, dflt) is the constant `dflt`, the arms behind the other
// values are dead code the compiler drops (kernels nobody launches leave the code object), bhg_debug_set() fails and
// bhg_debug_key_count() is 0.  One form per solver ships; the arms are measured and tested on libbhg_ab.so, built from the same sources.
constexpr bool dbg_is_set(DbgKey) { return false; }
constexpr int dbg(DbgKey, int dflt) { return dflt; }
#endif

// ---- in-kernel time stamps (measurement builds only: make stamps -> libbhg_stamps.so, -DBHG_STAMPS) ------------------------------
// BHG_STAMP(kernel_id, slot): thread 0 of the workgroup stores s_memrealtime (100 MHz, one clock for the whole chip) at
// [kernel_id][blockIdx.x][slot].  The shipped library compiles every stamp to nothing.
constexpr int kStampKernels = 4, kStampBlocks = 4096, kStampSlots = 8;
#ifdef BHG_STAMPS
#ifdef __HIPCC__
// (d_stamps is defined by the translation unit that stamps: bhg_mlp.hip)
#define BHG_STAMP(kid, slot)                                                                                            \
  do {                                                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < ::bhg::kStampBlocks) {                                                         \
      unsigned long long* sp_ = ::bhg::d_stamps;                                                                        \
      if (sp_) sp_[((kid) * ::bhg::kStampBlocks + blockIdx.x) * ::bhg::kStampSlots + (slot)] = wall_clock64();          \
    }                                                                                                                   \
  } while (0)
#endif
#else
#define BHG_STAMP(kid, slot) do { } while (0)
#endif

// ---- error plumbing -------------------------------------------------------------
void set_error(const char* fmt, ...);
#define BHG_HIP_CHECK(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      ::bhg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                       __LINE__);                                                  \
      return BHG_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)
#define BHG_REQUIRE(cond, msg)                                                     \
  do {                                                                             \
    if (!(cond)) {                                                                 \
      ::bhg::set_error("%s: %s", __func__, msg);                                   \
      return BHG_ERR_ARG;                                                          \
    }                                                                              \
  } while (0)

// ---- tensor pointer table ---------------------------------------------------------
// T <= kInlineT: pointers travel in the kernel arguments (no copy, no extra launch).
// Otherwise they are written into the workspace by k_write_table launches (stream
// ordered, no host sync, no pinned staging) and read through `dev`.
struct PtrTab {
  const void* const* dev;
  const void* inl[kInlineT];
};

int make_table(PtrTab* out, const void* const* host_ptrs, int T, void* ws, int slot,
               hipStream_t stream);

// measurement hook (bhg_timing_enable): true + two fresh events when this launch group is to be timed
bool span_begin(int kind, hipEvent_t* a, hipEvent_t* b);

#ifdef __HIPCC__
__device__ __forceinline__ float* tab_ptr(const PtrTab& t, int i) {
  const void* p = t.dev ? t.dev[i] : t.inl[i];
  return (float*)p;
}

// ---- vector access within a chunk ---------------------------------------------------
// `e` is the element offset inside the chunk (multiple of 4); base is 16-B aligned at
// e = 0 because chunk starts are multiples of kChunk elements from a 16-B aligned tensor
// base (checked on the host) and flat starts are multiples of BHG_FLAT_ALIGN.
__device__ __forceinline__ float4 ld4(const float* __restrict__ base, int e, int len) {
  if (e + 4 <= len) return *reinterpret_cast<const float4*>(base + e);
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < len) r.x = base[e];
  if (e + 1 < len) r.y = base[e + 1];
  if (e + 2 < len) r.z = base[e + 2];
  return r;
}
__device__ __forceinline__ void st4(float* __restrict__ base, int e, int len, float4 v) {
  if (e + 4 <= len) {
    *reinterpret_cast<float4*>(base + e) = v;
    return;
  }
  if (e < len) base[e] = v.x;
  if (e + 1 < len) base[e + 1] = v.y;
  if (e + 2 < len) base[e + 2] = v.z;
}

// ---- kernel arguments: one latency for all of them ------------------------------------------------------------------
// A miss of the scalar cache on the kernarg segment costs ~1.4 us on this system (stamps of k_graw: a tile whose descriptor shares
// the first lines of the argument struct reaches its first load 1.8 us after entry, one whose descriptor sits further back 4.6 us —
// two more DEPENDENT misses).  kernarg_warm<BYTES>() touches every 64-byte line of the struct with independent scalar loads and
// waits once: what the kernel then reads through run-time indices (block tables, per-problem descriptors) hits the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
  typedef const __attribute__((address_space(4))) unsigned* KP;
  KP kp = (KP)__builtin_amdgcn_kernarg_segment_ptr();
  unsigned acc = 0;
#pragma unroll
  for (int o = 0; o < BYTES; o += 64) acc |= kp[o / 4];
  asm volatile("" ::"s"(acc));
}
// Strided share of a partial array, summed in index order like `for (i = t; i < n; i += stride) acc += p[i]` — but U clamped
// loads are in flight together instead of one round trip per element (a loop-carried add behind every load: 7 trips for 448
// partials on one wave).
template <int U>
__device__ __forceinline__ double sum_strided(const double* __restrict__ p, const int n, const int t, const int stride) {
  double acc = 0.0;
  for (int i0 = t; i0 < n; i0 += U * stride) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * stride;
      v[u] = p[i < n ? i : n - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (i0 + u * stride < n) acc += v[u];
  }
  return acc;
}

// The same in two phases, for SEVERAL arrays at once: strided_load() only issues the (clamped, unconditional) loads, strided_sum()
// adds the live ones in index order.  A kernel that sums five partial arrays with five loops pays five dependent trips to the
// Infinity Cache (stamps: 7-9 us for the step length's 650 partials); with every array's loads issued before the first add, one.
// Requires n <= U * stride (callers fall back to sum_strided otherwise); p may be NULL when n == 0 (`dummy` is read instead).
template <int U>
struct StridedRegs { double v[U]; };
template <int U>
__device__ __forceinline__ void strided_load(StridedRegs<U>& r, const double* __restrict__ p, const int n, const int t, const int stride,
                                             const double* __restrict__ dummy) {
  const double* q = (p && n > 0) ? p : dummy;
  const int last = (p && n > 0) ? n - 1 : 0;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int i = t + u * stride;
    r.v[u] = q[i < n && i <= last ? i : last];
  }
}
template <int U>
__device__ __forceinline__ double strided_sum(const StridedRegs<U>& r, const int n, const int t, const int stride) {
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (t + u * stride < n) acc += r.v[u];
  return acc;
}

// ---- deterministic block reductions (fp64) --------------------------------------------
// Sum over the 64 lanes of a wave, fixed order, result in EVERY lane.  Data-parallel-primitive moves inside the 16-lane rows
// (xor 1, xor 2, half-row mirror, row mirror: every lane of a row ends with the row's sum — each step adds the two partners in
// both orders, which is the same bits), then the four row sums, (r0 + r1) + (r2 + r3), through v_readlane.  ~40 instructions
// without a trip through the LDS crossbar; the __shfl_down ladder it replaces is six DEPENDENT ds_bpermute pairs per double
// (~150 cycles each): 0.4 us per call, 2.6 us for the six sums of k_proj_step's scalar phase (its ISA: bpermute, wait, bpermute).
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(const double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ double readlane_f64(const double v, const int lane) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | (unsigned long long)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov_f64<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_mov_f64<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_mov_f64<0x141>(v);   // row_half_mirror
  v += dpp_mov_f64<0x140>(v);   // row_mirror
  const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
  return (r0 + r1) + (r2 + r3);  // valid in every lane
}
// The same for a float (the head kernel's class dot products).
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_f32(float v) {
  v += dpp_mov_f32<0xB1>(v);
  v += dpp_mov_f32<0x4E>(v);
  v += dpp_mov_f32<0x141>(v);
  v += dpp_mov_f32<0x140>(v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return (r0 + r1) + (r2 + r3);
}
// Three block sums with ONE pair of barriers (the fused CG epilogues emit r'.r', r'.p, p.p per workgroup: three block_sum calls
// were six barriers).  Same values as three block_sum calls.  `red`: 3 * kWaves doubles of LDS.
__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* red) {
  const double wa = wave_sum(a), wb = wave_sum(b), wc = wave_sum(c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` against the previous use
  if (lane == 0) { red[w] = wa; red[kWaves + w] = wb; red[2 * kWaves + w] = wc; }
  __syncthreads();
  double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) { sa += red[i]; sb += red[kWaves + i]; sc += red[2 * kWaves + i]; }
  a = sa; b = sb; c = sc;
}
// Sum over the block; result valid in every thread.  `red` is kWaves doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` against the previous use
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kWaves; ++i) s += red[i];
  return s;
}
// Fixed-order sum of n (<= kMaxBlocks) per-block partials written by an EARLIER kernel.
// Every block computes bit-identical results.
__device__ __forceinline__ double sum_partials(const double* __restrict__ part, int n,
                                               double* red) {
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += kThreads) a += part[i];
  return block_sum(a, red);
}
#endif  // __HIPCC__

}  // namespace bhg
