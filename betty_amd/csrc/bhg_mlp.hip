// bhg_mlp.hip — analytic Hessian-vector product of a ReLU-MLP with per-sample-weighted
// cross-entropy (+ ridge) on the gfx950 matrix cores.
//
// Replaces the double backward `torch.autograd.grad(in_grad, params, grad_outputs=p)` of
// betty/hypergradient/cg.py:39-41 / neumann.py:62 for the inner problem of
// examples/learning_to_reweight/main.py:117-127.  Everything that does
// not depend on the direction (activations h_l, ReLU masks m_l, softmax p, back-propagated
// delta_l) is computed once per hypergradient step by the caller and cached; one HVP is then
//   R-forward   Ra_l = Rh_{l-1} W_l^T + h_{l-1} V_l^T + c_l ,  Rh_l = m_l * Ra_l       (NT GEMMs)
//   top         Rd_L = sd * (p*Rz - p (p.Rz))
//   R-backward  Rd_{l-1} = m_{l-1} * (delta_l V_l + Rd_l W_l)                           (NN GEMMs)
//   outputs     H(W_l) = Rd_l^T h_{l-1} + delta_l^T Rh_{l-1} + 2 rho V_l                (TN GEMMs)
//               H(b_l) = colsum(Rd_l) + 2 rho c_l
// All GEMMs have one skinny dimension (the batch, padded to 128 rows) and stream a large weight
// or produce a large weight-shaped output: k_gemm tiles 128 x 32 per workgroup (4 waves of 32 x 32;
// 128 x 64 selectable), k_outer 128 x 64, both on v_mfma_f32_32x32x2_f32 (exact fp32), staged through
// LDS with coalesced 16-B global loads; the skinny GEMMs are split along K across ~768 workgroups to
// fill the 256 CUs (deterministic: partial slabs are summed in fixed order, no atomics).
//
// Layout of this translation unit: the device code lives in the fragments mlp/*.inc, #included below inside the anonymous
// namespace in dependency order (one TU: every kernel sees the same inlined helpers); this file keeps the host side — launch
// helpers, the plan of the hoisted / projected forms (hoist_plan), the workspace carve-up, run_chain (one HVP chain with its
// outputs stored or consumed), and the C entry points bhg_mlp_* of include/bhg.h.
#include <stdlib.h>
#include <vector>


#include "bhg_common.hpp"

#ifdef BHG_STAMPS
namespace bhg { __device__ unsigned long long* d_stamps = nullptr; }
#endif

namespace bhg {
namespace {
#include "mlp/gemm.inc"   // tile constants, GemmArgs, LDS tile loaders, gemm_body / k_gemm (split-K skinny GEMM on v_mfma_f32_32x32x2_f32)

#include "mlp/outer.inc"   // fused CG / Neumann epilogues (fuse_elem), k_outer (weight-shaped outputs), split-K reduce kernels

#include "mlp/scalars.inc"   // step length from batch-sized factors (alpha_compute), global-batch scalars, T2 / softmax reduces, bias outputs

#include "mlp/head.inc"   // narrow classifier head kernels and k_outer_all (every weight-shaped output in one launch)

#include "mlp/wsk.inc"   // skinny GEMMs with the K split inside the workgroup (register and LDS-staged forms), k_wsk_group

#include "mlp/wskp.inc"   // round 4: the same products on PACKED operands (no LDS in the K loop), several problems per launch; k_pack

// BHG_MLP_WSK: 0 = split-K launches + reduce everywhere | 1 = in-workgroup split wherever the shape allows | 2 = only
// for short reductions (pairs * K <= BHG_MLP_WSK_MAXK, default 1024), where the launch and the slab round trip weigh more
// than the operand re-reads.  Measured on the cfg-2 shapes (rocprofv3 timeline, MI355X): K = 2 x 384 -> 10.8 us against
// 9.6 + 5.0 us (GEMM + reduce); K = 3072 / 2 x 2048 / 2 x 1536 -> 37.3 / 40.4 / 30.2 us against 27.5 / 27.4 / 26.9 us:
// a lane's 16-B loads in MFMA layout touch 16 cache lines per instruction (64 B of each), and the vector cache's
// address path retires about one line per 4 clocks — 8 waves x 12 loads x 16 lines x 4 clk per 32-k chunk is three times
// the chunk's MFMA time.  Read on every call so a test can compare both arms in one process.
inline int wsk_mode(int chain_mode, int unfused_hint) {
  if (dbg_is_set(DBG_mlp_wsk)) return dbg(DBG_mlp_wsk, 0);   // an explicit value applies everywhere
  // defaults: fused CG solver 2 (short reductions; the staged form of mode 3 is +0.75 % there but draws 2.3e-4 from the fp64
  // truth in the CG-20 noise lottery); the Neumann solver — no reduction, no chaos — 3 in BOTH of its arms
  // (the un-fused arm asks for it through bhg_mlp_hvp_mode, so fused and un-fused stay bitwise equal); everything else 0
  if (chain_mode == FUSE_CG) return 2;
  if (chain_mode == FUSE_NEUMANN) return 3;
  return unfused_hint;
}
inline bool wsk_wanted(int mode, int pairs, int K) {
  if (mode == 1) return true;
  if (mode != 2 && mode != 3) return false;
  return pairs * K <= dbg(DBG_mlp_wsk_maxk, 1024);
}
inline int wsk_depth() {
  const int d = dbg(DBG_mlp_wsk_depth, 3);
  return d == 2 ? 2 : 3;
}
inline bool wsk_eligible(const WskArgs& a) {
  if (a.pairs < 1 || a.pairs > 2) return false;
  bool ok = a.M % 32 == 0 && a.N % 32 == 0 && a.K % 32 == 0 && a.K >= 32;
  for (int i = 0; i < a.pairs; ++i) ok = ok && (a.pr[i].lda & 3) == 0 && (a.pr[i].ldb & 3) == 0;
  return ok;
}
int64_t g_wsk_launches = 0;   // bhg_mlp_wsk_launches()
int64_t g_hoist_launches = 0; // bhg_mlp_hoist_launches()
int64_t g_proj_iterations = 0; // bhg_mlp_proj_iterations()
int64_t g_lin_launches = 0;    // bhg_mlp_lin_launches()
template <int LB>
void launch_gemm_wsk(const WskArgs& a_in, hipStream_t st, bool staged = false) {
  WskArgs a = a_in;
  ++g_wsk_launches;
  a.ntm = a.M / 32;
  a.ntn = a.N / 32;
  bool bf = false;
  for (int i = 0; i < a.pairs; ++i) bf = bf || a.pr[i].mix != 0;
  const dim3 grid(a.ntm * a.ntn), block(64 * kWskWaves);
  const int d = wsk_depth();
  const bool force_staged = dbg(DBG_mlp_wsk_lds, 0) != 0;
  // LDS-staged form (two register stages).  Only WITHOUT the lazy direction: the mixing instance (a third staged operand)
  // does not fit the register file (256 VGPRs + 42-44 spilled, round-2 verdict) and is not built — callers with a lazy
  // direction get the direct form or split-K (run_chain never asks for the staged form then); scripts/check_spills.py
  // keeps every shipped kernel free of spills.
  if ((staged || force_staged) && !bf) {
    const int lds = (int)(sizeof(float) * kWslWaveFloats * kWskWaves);
    static bool attr_done = false;
    if (!attr_done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_wsk<LB, false, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr_done = true;
    }
#ifdef BHG_AB   // (arm-only instances are not in the product's code object: bhg_common.hpp)
    const int wsl_depth = dbg(DBG_wsl_depth, 2);   // A/B: register stages of the staged form
    if (wsl_depth == 3) {
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_wsk<LB, false, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr3 = true;
      }
      hipLaunchKernelGGL((k_gemm_wsk<LB, false, 3, true>), grid, block, lds, st, a);
      return;
    }
#endif
    hipLaunchKernelGGL((k_gemm_wsk<LB, false, 2, true>), grid, block, lds, st, a);
    return;
  }
#define BHG_WSK(BFV, DV) hipLaunchKernelGGL((k_gemm_wsk<LB, BFV, DV>), grid, block, 0, st, a)
  if (bf) {
    // (N-contiguous B with the lazy direction holds 48 registers per stage: three stages would spill)
    if (d == 2 || LB == LAYOUT_RC) BHG_WSK(true, 2); else BHG_WSK(true, LB == LAYOUT_RC ? 2 : 3);
  } else {
#ifdef BHG_AB
    if (d == 2) BHG_WSK(false, 2); else
#endif
    BHG_WSK(false, 3);
  }
#undef BHG_WSK
}

#ifdef BHG_AB
void launch_wsk_group(const WskGroupArgs& g, int blocks, hipStream_t st) {   // blocks: the tiles (+ 1 with do_alpha)
  const int lds = (int)(sizeof(float) * kWslWaveFloats * kWskWaves);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wsk_group<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#ifdef BHG_AB
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wsk_group<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
#endif
    attr_done = true;
  }
#ifdef BHG_AB
  const int wsl_depth = dbg(DBG_wsl_depth, 2);
  if (wsl_depth == 3) { hipLaunchKernelGGL(k_wsk_group<3>, dim3(blocks), dim3(64 * kWskWaves), lds, st, g); return; }
#endif
  hipLaunchKernelGGL(k_wsk_group<2>, dim3(blocks), dim3(64 * kWskWaves), lds, st, g);
}
#endif

// Grouped launch on packed operands (wskp.inc).  Block tables are rounded up to multiples of 8 so that a problem's tile t keeps
// t % 8 == blockIdx.x % 8 (the XCD it runs on).
// CUs of the current device (per-device cache; 256 on an MI355X): grouped launches are sized to ONE resident round of workgroups
int chip_cus() {
  static int cus[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  if (cus[dev] > 0) return cus[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
  return cus[dev] = n;
}
// Row tiling of one problem (see wskp_body): chain products cover the B valid rows with 32-row tiles and, over a remainder of at
// most 16 rows, 16 x 64 tiles; K-split / raw problems keep every row tile (their consumers read all RA rows or none of the padding).
inline void wskp_tiling(WskpProb* q, bool ragged) {
  q->nfull = q->RA / 32;
  q->nstrip = 0;
  q->pairs = dbg(DBG_xcd_pairs, 1);
  if (!ragged || q->raw || q->nsplit != 1 || q->RB % 64 != 0) return;
  int nf = q->B / 32, rem = q->B % 32;
  if (rem > 16) { ++nf; rem = 0; }
  if (nf == 0) return;   // (a batch of <= 16 rows: one row of full tiles)
  q->nfull = nf;
  q->nstrip = rem > 0 ? q->RB / 64 : 0;
}
struct WskpBuilder {
  WskpArgs g{};
  int blk = 0;
  bool add(const WskpProb& q_in) {
    if (g.n >= kWskpMax) return false;
    g.p[g.n++] = q_in;
    return true;
  }
  // Row tiling of the launch: every row tile (4 x N/32 tiles at B = 100) while that fits ONE resident round of workgroups — a
  // 16 x 64 remainder tile loads 25 % more per MFMA and ends ~1 us after the 32 x 32 tiles of its launch (probe: 11.8 vs 10.9 us
  // for the product through W_1) — and the ragged tiling only where it makes the round fit (the product through W_1^T with its two
  // Gram riders: 224 + 16 + 16 workgroups on 256 CUs instead of 256 + 32).  Debug key wskp_ragged: 0 never, 2 always.
  void tile(bool ragged) {
    blk = 0;
    for (int i = 0; i < g.n; ++i) {
      WskpProb& q = g.p[i];
      wskp_tiling(&q, ragged);
      g.blk0[i] = blk;
      blk += (((q.nfull * (q.RB / 32) + q.nstrip) * q.nsplit) + 7) & ~7;
    }
  }
  void launch(hipStream_t st) {
    if (!g.n) return;
    const int mode = dbg(DBG_wskp_ragged, 1);
    tile(mode == 2);
    if (mode == 1 && blk > chip_cus()) tile(true);
    g.blk0[g.n] = blk;
    const int d = dbg(DBG_packed_depth, 2);   // register stages of the K loop (A/B: 2 is ~1 us per iteration ahead of 3 in six same-box pairs)
    // the per-iteration launches: a chain product + up to two Gram riders, named kernel arguments (no dependent scalar loads)
    bool riders = g.n <= 3;
    for (int i = 1; i < g.n && riders; ++i) {
      const WskpProb& r = g.p[i];
      riders = r.raw && r.nsplit == 1 && r.RA == g.p[0].RA && r.RB == g.p[0].RA && r.B == g.p[0].B && !r.bias && !r.mask && !r.addend &&
               !r.partT2 && r.nstrip == 0 && r.nfull == r.RA / 32;
    }
    if (riders) {
      WskpcArgs c{};
      c.a = g.p[0];
      c.na = g.n >= 2 ? g.blk0[1] : blk;
      if (g.n >= 2) { c.r0 = {g.p[1].Ap, g.p[1].Bq, g.p[1].out, g.p[1].outp, g.p[1].K, 0}; c.nr0 = (g.n == 3 ? g.blk0[2] : blk) - g.blk0[1]; }
      if (g.n == 3) c.r1 = {g.p[2].Ap, g.p[2].Bq, g.p[2].out, g.p[2].outp, g.p[2].K, 0};
#ifdef BHG_AB
      if (d != 2) hipLaunchKernelGGL(k_wskpc<3>, dim3(blk), dim3(64 * kWskpWaves), 0, st, c); else
#endif
      hipLaunchKernelGGL(k_wskpc<2>, dim3(blk), dim3(64 * kWskpWaves), 0, st, c);
    } else {
#ifdef BHG_AB
      if (d != 2) hipLaunchKernelGGL(k_wskp<3>, dim3(blk), dim3(64 * kWskpWaves), 0, st, g); else
#endif
      hipLaunchKernelGGL(k_wskp<2>, dim3(blk), dim3(64 * kWskpWaves), 0, st, g);
    }
    (void)d;
    g = WskpArgs{};
    blk = 0;
  }
};

template <int LA, int LB>
void launch_gemm(const GemmArgs& a_in, int tn, hipStream_t st) {
  GemmArgs a = a_in;
  // split-K slabs leave with non-temporal stores: they stream out while the kernel runs instead of sitting dirty in L2
  // until its end (same-box A/B: 3.707 vs 3.778 ms per step; BHG_NO_NT_SLABS restores plain stores)
  const bool nt_slabs = dbg(DBG_no_nt_slabs, 0) == 0;
  a.nt_out = (nt_slabs && (a.splits > 1 || a.out_rows > 0)) ? 1 : 0;
  // slab tiles of all-interior 128 x 32 instances leave through LDS as 16-B stores (11.7 vs 12.6 us for a 2-step
  // launch, 276.5 vs 270.5 steps/s, two same-box repetitions; BHG_GEMM_NO_XPOSE restores the direct 4-B stores)
  const bool xpose = dbg(DBG_gemm_no_xpose, 0) == 0;
  a.xpose_out = xpose ? 1 : 0;
  dim3 grid((a.N + tn - 1) / tn, (a.M + kTM - 1) / kTM, a.splits);
  bool fast = a.M % kTM == 0 && a.N % tn == 0 && a.K % kTK == 0;
  bool bf = false;
  for (int i = 0; i < a.pairs; ++i) {
    fast = fast && (a.pr[i].lda & 3) == 0 && (a.pr[i].ldb & 3) == 0;
    bf = bf || a.pr[i].mix != 0;
  }
  if (bf)   // a pair without mixing reads its own operand twice with weight 0 (B + 0 * B = B exactly; the loop stays branch-free)
    for (int i = 0; i < a.pairs; ++i)
      if (!a.pr[i].mix) a.pr[i].B2 = a.pr[i].B;
  const bool no_fast = dbg(DBG_mlp_no_fast, 0) != 0;   // A/B switch (debug)
  if (no_fast) fast = false;
#define BHG_GEMM(TNV, F, BFV) hipLaunchKernelGGL((k_gemm<LA, LB, TNV, F, BFV>), grid, dim3(256), 0, st, a)
#ifdef BHG_AB   // (debug key mlp_tn = 64)
  if (tn == 64) {
    if (bf) { if (fast) BHG_GEMM(64, true, true); else BHG_GEMM(64, false, true); }
    else    { if (fast) BHG_GEMM(64, true, false); else BHG_GEMM(64, false, false); }
  } else
#endif
  {
    if (bf) { if (fast) BHG_GEMM(32, true, true); else BHG_GEMM(32, false, true); }
    else    { if (fast) BHG_GEMM(32, true, false); else BHG_GEMM(32, false, false); }
  }
#undef BHG_GEMM
}
inline int skinny_tile_n() {
  const int tn = dbg(DBG_mlp_tn, 32);  // 32 measured +2 % over 64
  return tn == 64 ? 64 : 32;
}

void launch_reduce_mask(hipStream_t st, const float* part, int splits, int slab, const float* bias,
                        const float* mask, float* out, int rows, int N, int B, float* relu_mask_out = nullptr,
                        float* outp = nullptr) {
  if ((N & 3) == 0) {
    int blocks = (slab / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_reduce_mask<4>, dim3(blocks), dim3(256), 0, st, part, splits, slab, bias, mask, out, rows, N, B,
                       relu_mask_out, outp);
  } else {
    int blocks = (slab + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_reduce_mask<1>, dim3(blocks), dim3(256), 0, st, part, splits, slab, bias, mask, out, rows, N, B,
                       relu_mask_out, (float*)nullptr);
  }
}

int pick_splits(int tiles, int K, int pairs) {
  // fill 3 workgroups per CU (measured on the cfg-2 shapes: 512 -> 171 us, 768 -> 165 us, 896 -> 169 us per HVP);
  // never split below one K step; prefer a split count that divides the K steps evenly (no short last slice)
  const int ksteps = (K + kTK - 1) / kTK;
  const int target = dbg(DBG_split_target, 768);
  int s = (target + tiles - 1) / tiles;
  if (s > ksteps) s = ksteps;
  const int cap = dbg(DBG_split_cap, 16);
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (ksteps % s != 0) {
    for (int d = 1; d <= 2; ++d) {
      if (s - d >= 1 && ksteps % (s - d) == 0) { s -= d; break; }
      if (s + d <= cap && s + d <= ksteps && ksteps % (s + d) == 0) { s += d; break; }
    }
  }
  return s;
}

#include "mlp/recurrence.inc"   // mixed coefficient from Rz(x), direction update kernels (k_cg_pdir, k_cg_beta)

#include "mlp/proj.inc"   // hoisted direction products (k_hoist, k_hoist_reduce) and the projected solvers' kernels (k_proj_scalars / _update / _step)

#include "mlp/graw.inc"   // round 4: the closing launch of a projected iteration on packed Gram matrices (k_graw)

#include "mlp/pstep.inc"   // round 4: k_proj_step with its kernel arguments in two scalar-memory round trips (k_pstep)
#include "mlp/wskpl.inc"   // round 4: the chain's first product by linearity, with the update launch riding in it (k_wskpl)
#include "mlp/headu.inc"   // round 5: the head launch with the update blocks riding in it (k_headu)

// One chain product (no riders) with `nu` update blocks leading its grid (k_wskpu).  Tiling as WskpBuilder::launch does it.
void launch_wskpu(const WskpProb& q_in, const PstepArgs& ps, int nu, hipStream_t st) {
  WskpuArgs u{};
  WskpProb q = q_in;
  const int mode = dbg(DBG_wskp_ragged, 1);
  wskp_tiling(&q, mode == 2);
  int blk = (((q.nfull * (q.RB / 32) + q.nstrip) * q.nsplit) + 7) & ~7;
  if (mode == 1 && blk > chip_cus()) {
    wskp_tiling(&q, true);
    blk = (((q.nfull * (q.RB / 32) + q.nstrip) * q.nsplit) + 7) & ~7;
  }
  u.c.a = q; u.c.na = blk;
  u.nu = nu; u.ps = ps;
  hipLaunchKernelGGL((k_wskpu<2, 4>), dim3(nu + blk), dim3(64 * kWskpWaves), 0, st, u);
}

// ---- per-device side stream + events -------------------------------------------------------------------------------
struct SideState {
  hipStream_t side;
  hipEvent_t ev_rd[BHG_MLP_MAX_LAYERS];
  hipEvent_t ev_join;
};
int side_state(SideState** out) {
  static SideState per_device[64];
  int dev = 0;
  BHG_HIP_CHECK(hipGetDevice(&dev));
  BHG_REQUIRE(dev >= 0 && dev < 64, "device index out of range");
  SideState& ss = per_device[dev];
  if (!ss.side) {
    BHG_HIP_CHECK(hipStreamCreateWithFlags(&ss.side, hipStreamNonBlocking));
    const unsigned fl = hipEventDisableTiming | hipEventDisableSystemFence;
    for (int i = 0; i < BHG_MLP_MAX_LAYERS; ++i) BHG_HIP_CHECK(hipEventCreateWithFlags(&ss.ev_rd[i], fl));
    BHG_HIP_CHECK(hipEventCreateWithFlags(&ss.ev_join, fl));
    // k_outer's dynamic LDS is at most 64 K rows x (128 + 64) floats = 48 KiB; the limit is raised explicitly so a
    // larger kOK only needs this number changed (static LDS of the fused instances counts against the 160 KiB too)
#define BHG_OUTER_LDS(F, M) \
    BHG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer<F, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024))
    BHG_OUTER_LDS(true, FUSE_NONE); BHG_OUTER_LDS(false, FUSE_NONE);
    BHG_OUTER_LDS(true, FUSE_CG); BHG_OUTER_LDS(false, FUSE_CG);
    BHG_OUTER_LDS(true, FUSE_NEUMANN); BHG_OUTER_LDS(false, FUSE_NEUMANN);
#undef BHG_OUTER_LDS
#ifdef BHG_AB   // (debug key outer_no_pre)
    BHG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer_all<FUSE_CG, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    BHG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer_all<FUSE_NEUMANN, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
#endif
    BHG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer_all<FUSE_CG, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    BHG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_outer_all<FUSE_NEUMANN, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  }
  *out = &ss;
  return BHG_OK;
}

bool use_head(const bhg_mlp* m) {
  const bool no_head = dbg(DBG_mlp_no_head, 0) != 0;    // A/B switch (debug)
  return !no_head && m->L >= 1 && m->dims[m->L] <= kSmallC && (m->dims[m->L - 1] & 3) == 0;
}

// Blocks of the weight-shaped output of layer l (fused CG: one r'.r' partial per block)
int outer_blocks(const bhg_mlp* m, int l, bool head) {
  const int Mo = m->dims[l + 1], No = m->dims[l];
  if (head && l == m->L - 1) return ((No + 63) / 64) * Mo;
  return ((No + kTN - 1) / kTN) * ((Mo + kTM - 1) / kTM);
}
int bias_blocks(const bhg_mlp* m) {
  int blk = 0;
  for (int l = 0; l < m->L; ++l) blk += (m->dims[l + 1] + 63) / 64;
  return blk;
}
int reduce_blocks(int slab, int N) {
  int blocks = (N & 3) == 0 ? (slab / 4 + 255) / 256 : (slab + 255) / 256;
  return blocks > 2048 ? 2048 : blocks;
}

// Plan of the hoisted direction products (see k_hoist): which products, their split-K factors — the smallest K-steps-per-
// workgroup target whose workgroups all fit one resident wave of the chip (3 per CU) — and where their slabs and their
// persistent G arrays live inside the fused workspace.
// K split of the per-iteration Gram products over workgroups (k_wsk_group): about 512 k per workgroup (BHG_GRAM_KCHUNK), at most kGramSplitMax
// (BHG_GRAM_KSPLIT=0: one workgroup per tile, the A/B arm)
constexpr int kGramSplitMax = 8;
inline int gram_ksplit(int K) {
  if (dbg(DBG_gram_ksplit, 1) == 0) return 1;
  const int c = dbg(DBG_gram_kchunk, 512);   // k per workgroup (A/B)
  const int per = c >= 32 ? c : 512;
  const int s = K / per;
  return s < 1 ? 1 : (s > kGramSplitMax ? kGramSplitMax : s);
}
struct HoistPlan {
  bool ok;
  int n;
  int layer[kHoistMax], bwd[kHoistMax], K[kHoistMax], N[kHoistMax], splits[kHoistMax];
  int blk0[kHoistMax + 1];
  size_t slab_off[kHoistMax], g_off[kHoistMax];   // float offsets inside the hoist region
  size_t gr_off[kHoistMax], graw_off[kHoistMax];  // projected CG: G(r) and G(raw) of every product
  size_t s_off[BHG_MLP_MAX_LAYERS], d_off[BHG_MLP_MAX_LAYERS];   // B x B Gram matrices, constant over a solve
  int dot_blocks, raw_blocks;   // fully projected CG: dot blocks of r.raw / p.raw; tiles of the G(raw) launch (raw.raw partials)
  size_t tslab_off[BHG_MLP_MAX_LAYERS], eslab_off[BHG_MLP_MAX_LAYERS];   // K-split slabs of T_l / E_l (kGramSplitMax each)
  size_t sp_off[BHG_MLP_MAX_LAYERS], dp_off[BHG_MLP_MAX_LAYERS];         // the same four, PACKED ([Bp/16][Bp][16]: k_graw's M-side
  size_t tslabp_off[BHG_MLP_MAX_LAYERS], eslabp_off[BHG_MLP_MAX_LAYERS]; // operands; two slabs at most)
  int graw_tiles;                                                        // 64 x 32 tiles of the G(raw) launch (k_graw)
  // the chain's first product by linearity (k_wskpl; L >= 4 and the projected forms only): Z(p) = Rh_0(p) W_1^T in two slots by
  // iteration parity, packed Rh_0(r'), the second slot of Gf_1(p) (slot 0 is g_off[gf[1]]), a copy of r|b0
  size_t z1_off[2], rh0rp_off, gp1alt_off, rb0c_off; bool lin_ok;
  size_t rh0alt_off;   // projected Neumann with the update inside k_graw: the second slot of the ROW-MAJOR Rh_0 (slot 0 is m->Rh[0])
  size_t gp2alt_off;   // update blocks in the head launch (k_headu): the second slot of Gf_{L-2}(p) (slot 0 is g_off[gf[L-2]])
  bool proj_ok;                                                          // the projected solvers pay off and fit (cost model below)
  size_t floats;
  int gf[BHG_MLP_MAX_LAYERS], gb[BHG_MLP_MAX_LAYERS];   // index of the forward / backward product of layer l (-1: none)
};
inline bool graw_batch_ok(int Bp) { return Bp == 128 || (Bp % 128 == 0 && dbg(DBG_graw_kloop, 1) != 0); }
inline bool graw_split() {   // G(raw) products with two operand pairs: one workgroup per pair (A/B: BHG_PROJ_GRAW_SPLIT=0)
  return dbg(DBG_proj_graw_split, 1) != 0;
}
inline int proj_mode() {
  return dbg(DBG_mlp_proj, 1);   // read on every call (A/B in one process); default on
}
inline int hoist_mode() {
  return dbg(DBG_mlp_hoist, 1);   // read on every call so a test can compare both arms in one process
}
void hoist_plan(const bhg_mlp* m, HoistPlan* hp) {
  memset(hp, 0, sizeof(*hp));
  const int L = m->L, Bp = m->Bp;
  for (int l = 0; l < BHG_MLP_MAX_LAYERS; ++l) hp->gf[l] = hp->gb[l] = -1;
  if (!use_head(m) || L < 3 || L - 1 > kHoistMax / 2 || Bp % kTM != 0) return;
  for (int l = 0; l <= L - 1; ++l) if (m->dims[l] % 32 != 0) return;           // every MFMA layer: K and N multiples of 32
  const int Nh = m->dims[L - 1];
  if ((size_t)Nh * sizeof(float) > 64 * 1024) return;                         // the head kernel combines the last hidden layer
  for (int l = 1; l + 1 < L; ++l)                                              // T2 slots were sized for the reduce launch
    if ((Bp / 32) * (m->dims[l] / 32) != reduce_blocks(Bp * m->dims[l], m->dims[l])) return;
  int n = 0;
  for (int l = 0; l + 1 < L; ++l) { hp->layer[n] = l; hp->bwd[n] = 0; hp->K[n] = m->dims[l]; hp->N[n] = m->dims[l + 1]; hp->gf[l] = n; ++n; }
  for (int l = 1; l + 1 < L; ++l) { hp->layer[n] = l; hp->bwd[n] = 1; hp->K[n] = m->dims[l + 1]; hp->N[n] = m->dims[l]; hp->gb[l] = n; ++n; }
  hp->n = n;
  const int slots = dbg(DBG_hoist_wgs, 768);
  const int ntm = Bp / kTM;
  int tgt = 8;
  for (; tgt < 4096; ++tgt) {
    int wgs = 0;
    for (int i = 0; i < n; ++i) {
      const int ks = hp->K[i] / kTK;
      const int per = ks < tgt ? ks : tgt;
      wgs += (hp->N[i] / 32) * ntm * ((ks + per - 1) / per);
    }
    if (wgs <= slots) break;
  }
  size_t off = 0;
  int blk = 0;
  for (int i = 0; i < n; ++i) {
    const int ks = hp->K[i] / kTK;
    int sp = (ks + tgt - 1) / tgt;
    const int per = (ks + sp - 1) / sp;
    sp = (ks + per - 1) / per;               // no empty split
    hp->splits[i] = sp;
    hp->blk0[i] = blk;
    blk += (hp->N[i] / 32) * ntm * sp;
    hp->slab_off[i] = off; off += (size_t)sp * Bp * hp->N[i];
    hp->g_off[i] = off;    off += (size_t)Bp * hp->N[i];
    hp->gr_off[i] = off;   off += (size_t)Bp * hp->N[i];
    hp->graw_off[i] = off; off += (size_t)2 * Bp * hp->N[i];   // up to two slabs (one per operand pair)
  }
  // Cost model of the projected solvers: per iteration they trade the hoisted direction products on the N-sized
  // residual, F_h = 2 B (sum_{l<=L-2} d_l d_{l+1} + sum_{1<=l<=L-2} d_l d_{l+1}) flops (and its 4 N bytes), for B x B Gram products
  // and batch-deep G(raw) products, F_p = 2 B^2 (sum_{1<=l<=L-2} (d_l + d_{l+1}) + sum G(raw) widths) — quadratic in the batch,
  // like their Gram region ((2 + 16 + 6) Bp^2 floats per layer: 2.5 GB at Bp = 4096).  Projection is taken while
  // F_p <= proj_max_ratio % of F_h (default 400: the projected form also drops every N-sized read and launch) and the Gram region stays
  // under proj_ws_cap_mb (default 1024); otherwise the plan keeps the hoisted chain and carves no Gram region at all.
  {
    double fh = 0.0, fp = 0.0;
    for (int l = 0; l + 1 < L; ++l) fh += (double)m->dims[l] * m->dims[l + 1] * (l >= 1 ? 2.0 : 1.0);
    fh *= 2.0 * Bp;
    for (int l = 0; l + 1 < L; ++l) {
      fp += (double)m->dims[l + 1] * (l == 0 ? 1.0 : 2.0);                                 // Gf_l(raw)
      if (l >= 1) fp += 2.0 * m->dims[l] + (double)m->dims[l] + (double)m->dims[l + 1];     // Gb_l(raw); T_l, E_l
    }
    fp *= 2.0 * Bp * (double)Bp;
    size_t gram = 0;
    for (int l = 0; l + 1 < L; ++l) gram += (size_t)Bp * Bp * (l >= 1 ? (2 + 2 * kGramSplitMax + 6) : 2);
    // (and every batch-sized array small enough for proj_update_one's row index — (i + 0.5) * (1 / nv) in fp32 instead of an integer
    //  division — to be exact: float4 indices below 2^20; it first fails at 2.0 M, tests/test_proj_global_protocol.py)
    size_t widest = 0;
    for (int i = 0; i < n; ++i) widest = (size_t)hp->N[i] > widest ? (size_t)hp->N[i] : widest;
    hp->proj_ok = fp * 100.0 <= fh * (double)dbg(DBG_proj_max_ratio, 400) &&
                  gram * sizeof(float) <= (size_t)dbg(DBG_proj_ws_cap_mb, 1024) * 1024 * 1024 &&
                  (size_t)Bp * (widest / 4) <= ((size_t)1 << 20);
  }
  hp->dot_blocks = 0;
  for (int i = 0; i < n; ++i) hp->dot_blocks += dot_blocks_of(Bp * (hp->N[i] / 4));
  hp->raw_blocks = 0;
  for (int i = 0; i < n; ++i) hp->raw_blocks += (hp->N[i] / 32) * ntm * 2;   // at most two workgroups (operand pairs) per tile
  for (int l = 0; hp->proj_ok && l + 1 < L; ++l) {
    hp->s_off[l] = off; off += (size_t)Bp * Bp;
    if (l >= 1) {
      hp->d_off[l] = off; off += (size_t)Bp * Bp;
      hp->tslab_off[l] = off; off += (size_t)kGramSplitMax * Bp * Bp;   // T_l, E_l: up to kGramSplitMax K-split slabs each
      hp->eslab_off[l] = off; off += (size_t)kGramSplitMax * Bp * Bp;
    }
  }
  for (int l = 0; hp->proj_ok && l + 1 < L; ++l) {
    hp->sp_off[l] = off; off += (size_t)Bp * Bp;
    if (l >= 1) {
      hp->dp_off[l] = off; off += (size_t)Bp * Bp;
      hp->tslabp_off[l] = off; off += (size_t)2 * Bp * Bp;
      hp->eslabp_off[l] = off; off += (size_t)2 * Bp * Bp;
    }
  }
  // L == 4: the update blocks ride in the pre-head launch (k_wskpu) and find beta published.  In a deeper net the launch after the
  // first product reads what they write, so they stay in k_wskpl — dispatched BEHIND the small blocks, the first of which publishes
  // beta (round 5: WskplArgs.upd_first; round 4 had them ahead of the publisher, a deadlock for a wide net, and kept the k_pstep
  // launch for L > 4 instead).  The pollers' wait is bounded either way (poll_beta).
  hp->lin_ok = hp->proj_ok && L >= 4 && (L == 4 || dbg(DBG_lin_deep, 1) != 0);
  if (hp->lin_ok) {
    for (int i = 0; i < 2; ++i) { hp->z1_off[i] = off; off += (size_t)Bp * m->dims[2]; }
    hp->rh0rp_off = off; off += (size_t)Bp * m->dims[1];
    hp->gp1alt_off = off; off += (size_t)Bp * m->dims[2];
    hp->rb0c_off = off; off += (size_t)((m->dims[1] + 63) & ~63);
  }
  if (hp->proj_ok) { hp->rh0alt_off = off; off += (size_t)Bp * m->dims[1]; }
  if (hp->lin_ok) { hp->gp2alt_off = off; off += (size_t)Bp * m->dims[L - 1]; }
  hp->graw_tiles = 0;
  for (int i = 0; i < n; ++i) hp->graw_tiles += (Bp / 64) * (hp->N[i] / 32);
  hp->blk0[n] = blk;
  hp->floats = off;
  hp->ok = true;
}

// Workgroups (= raw.raw partials) of the G(raw) launch: a product with two operand pairs takes one workgroup per pair and tile
// unless BHG_PROJ_GRAW_SPLIT=0 (only the first layer's forward product has a single pair).
// column tile of k_graw: 32, or 64 (debug key graw_cols = 64; every product's width must allow it).  Same-box A/B at cfg 2, twice:
// 667.4 / 668.0 steps/s with 32 columns, 660.3 / 661.6 with 64 — the 64-column workgroup has its CU to itself and its phases
// (operands 4.6 us, MFMAs 4.1 us, epilogue 2.9 us; stamps) no longer overlap with another workgroup's.
int graw_cols(const HoistPlan* hp, int Bp = 128) {
  if (dbg(DBG_graw_cols, 32) != 64 || Bp != 128) return 32;
  for (int i = 0; i < hp->n; ++i)
    if (hp->N[i] % 64) return 32;
  return 64;
}
int graw_tile_count(const HoistPlan* hp, int Bp) {
  const int ct = graw_cols(hp, Bp);
  int t = 0;
  for (int i = 0; i < hp->n; ++i) t += (Bp / 64) * (hp->N[i] / ct);
  return t;
}
int graw_blocks(const HoistPlan* hp, int Bp) {
  int n = 0;
  for (int i = 0; i < hp->n; ++i) {
    const bool two = !(hp->bwd[i] == 0 && hp->layer[i] == 0);
    n += (hp->N[i] / 32) * (Bp / kTM) * ((two && graw_split()) ? 2 : 1);
  }
  return n;
}

// Fully projected CG: scalars of iteration k-1 + recurrences of iteration k in one launch (k_proj_step); BHG_PROJ_STEP_ALONE=1
// keeps them as two launches (k_proj_scalars at the end of an iteration, k_proj_update at the top of the next): the A/B arm.
bool proj_step_merged() {
  return dbg(DBG_proj_step_alone, 0) == 0;
}

// Fused-solver scratch (device), carved out of the caller's buffer by bhg_mlp_cg_solve.
struct FusedWs {
  double* partT1; double* partT2h; double* partT2; double* partPP; double* partRR[2];
  float* rz; double* rzx;           // [Bp][dims[L]]: Rz of the current direction / accumulated Rz(x)
  int t2_off[BHG_MLP_MAX_LAYERS];   // first T2 partial of the R-backward reduce INTO layer l-1 (l = 1 .. L-2)
  int nRR, nT2;
  float* hoist;                     // slabs + G arrays of the hoisted direction products (HoistPlan offsets)
  double* part_dot; double* pscal;  // fully projected CG: [2][dot_blocks] partials of r.raw / p.raw; {rr, rp, pp} over the MFMA layers
  double* part_raw;                 // fully projected CG: [raw_blocks] partials of raw.raw (tiles of the G(raw) launch)
  double* part_graw;                // k_graw: [3][graw_tiles] partials of r.raw, p.raw, raw.raw
  float* pb0[2];                    // fully projected CG: the first bias's slice of the direction, two slots by iteration parity (k_proj_step)
  float* pb1[2];                    // the second bias's slice likewise (k_wskpl)
  unsigned long long* gran;         // 64 granules of 64 bytes: beta, published inside k_wskpl for its own tiles
  // packed operands of the chain and of the Gram products (wskp.inc); NULL when the hoisted forms do not apply
  float* Wf[BHG_MLP_MAX_LAYERS];    // W_l as the N-side operand of the forward chain   [d_l / 16][d_{l+1}][16],  l = 1 .. L-2
  float* Wb[BHG_MLP_MAX_LAYERS];    // W_l^T as the N-side operand of the backward chain [d_{l+1} / 16][d_l][16]
  float* hpk[BHG_MLP_MAX_LAYERS];   // h_l      [d_l / 16][Bp][16],      l = 0 .. L-2
  float* dpk[BHG_MLP_MAX_LAYERS];   // delta_l  [d_{l+1} / 16][Bp][16],  l = 1 .. L-2
  float* Rhp[BHG_MLP_MAX_LAYERS];   // Rh_l     [d_{l+1} / 16][Bp][16],  l = 0 .. L-3   (written by the producer of Rh_l)
  float* Rdp[BHG_MLP_MAX_LAYERS];   // Rd_l     [d_{l+1} / 16][Bp][16],  l = 1 .. L-2   (written by the producer of Rd_l)
  size_t bytes;
};
void carve_fused_ws(const bhg_mlp* m, void* base, FusedWs* w) {
  memset(w, 0, sizeof(*w));
  const bool head = use_head(m);
  int nrr = bias_blocks(m);
  for (int l = 0; l < m->L; ++l) nrr += outer_blocks(m, l, head);
  w->nRR = nrr;
  size_t off = 0;
  auto take = [&](size_t bytes) { void* p = base ? static_cast<char*>(base) + off : nullptr; off += (bytes + 255) & ~(size_t)255; return p; };
  w->partT1 = static_cast<double*>(take(sizeof(double) * m->Bp));
  w->partT2h = static_cast<double*>(take(sizeof(double) * m->Bp));
  w->partPP = static_cast<double*>(take(sizeof(double) * kMaxBlocks));
  w->partRR[0] = static_cast<double*>(take(sizeof(double) * 3 * nrr));   // [3][nRR]: r'.r', r'.p, p.p per epilogue block
  w->partRR[1] = static_cast<double*>(take(sizeof(double) * 3 * nrr));
  int nt2 = 0;
  for (int l = 1; l + 1 < m->L; ++l) { w->t2_off[l] = nt2; nt2 += reduce_blocks(m->Bp * m->dims[l], m->dims[l]); }
  w->nT2 = nt2;
  w->partT2 = static_cast<double*>(take(sizeof(double) * (nt2 > 0 ? nt2 : 1)));
  w->rz = static_cast<float*>(take(sizeof(float) * (size_t)m->Bp * m->dims[m->L]));
  w->rzx = static_cast<double*>(take(sizeof(double) * (size_t)m->Bp * m->dims[m->L]));
  HoistPlan hp;
  hoist_plan(m, &hp);
  w->hoist = static_cast<float*>(take(sizeof(float) * (hp.ok ? hp.floats : 1)));
  w->part_dot = static_cast<double*>(take(sizeof(double) * 2 * (hp.ok ? hp.dot_blocks : 1)));
  w->part_raw = static_cast<double*>(take(sizeof(double) * (hp.ok ? hp.raw_blocks : 1)));
  w->part_graw = static_cast<double*>(take(sizeof(double) * 3 * (hp.ok ? hp.graw_tiles : 1)));
  w->pscal = static_cast<double*>(take(sizeof(double) * 8));
  for (int i = 0; i < 2; ++i) w->pb0[i] = static_cast<float*>(take(sizeof(float) * (size_t)m->dims[1]));
  for (int i = 0; i < 2; ++i) w->pb1[i] = static_cast<float*>(take(sizeof(float) * (size_t)(m->L >= 2 ? m->dims[2] : 1)));
  w->gran = static_cast<unsigned long long*>(take(64 * 64));
  if (hp.ok) {
    const int L = m->L;
    const size_t Bp = (size_t)m->Bp;
    for (int l = 0; l + 1 < L; ++l) {
      w->hpk[l] = static_cast<float*>(take(sizeof(float) * Bp * m->dims[l]));
      if (l + 2 < L) w->Rhp[l] = static_cast<float*>(take(sizeof(float) * Bp * m->dims[l + 1]));
      if (l >= 1) {
        w->Wf[l] = static_cast<float*>(take(sizeof(float) * (size_t)m->dims[l + 1] * m->dims[l]));
        w->Wb[l] = static_cast<float*>(take(sizeof(float) * (size_t)m->dims[l + 1] * m->dims[l]));
        w->dpk[l] = static_cast<float*>(take(sizeof(float) * Bp * m->dims[l + 1]));
        w->Rdp[l] = static_cast<float*>(take(sizeof(float) * Bp * m->dims[l + 1]));
      }
    }
  }
  w->bytes = off;
}

// Packed copies of the constants of a solve (wskp.inc): the chain's weights in both orientations, h_l and delta_l for the Gram
// products.  One launch, once per solve: 3 x 15 M floats moved at cfg 2, ~1 % of a CG-20 solve.
bool packed_chain_on(const FusedWs& w) { return w.Wf[1] != nullptr && dbg(DBG_packed_chain, 1) != 0; }
// weights_only: the chain's weights and the input batch h_0 — what bhg_mlp_forward_packed packs before the net's own forward pass,
// whose epilogues leave h_l and delta_l packed themselves
int pack_operands(const bhg_mlp* m, const FusedWs& w, hipStream_t st, bool weights_only = false) {
  PackArgs g{};
  int blk = 0;
  auto add = [&](const float* X, float* Xp, int R, int K, int kind) {
    if (g.n >= kPackMax) return false;
    g.p[g.n] = {X, Xp, R, K, kind};
    g.blk0[g.n] = blk;
    blk += pack_blocks(g.p[g.n]);
    ++g.n;
    return true;
  };
  const int L = m->L, Bp = m->Bp;
  bool ok = true;
  for (int l = 0; l + 1 < L && ok; ++l) {
    if (!weights_only || l == 0) ok = ok && add(m->h[l], w.hpk[l], Bp, m->dims[l], 0);
    if (l >= 1) {
      if (!weights_only) ok = ok && add(m->delta[l], w.dpk[l], Bp, m->dims[l + 1], 0);
      ok = ok && add(m->W[l], w.Wf[l], m->dims[l + 1], m->dims[l], 0);
      ok = ok && add(m->W[l], w.Wb[l], m->dims[l], m->dims[l + 1], 1);
    }
  }
  BHG_REQUIRE(ok, "too many layers for one packing launch");
  g.blk0[g.n] = blk;
  hipLaunchKernelGGL(k_pack, dim3(blk), dim3(256), 0, st, g);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// What one pass of the HVP chain does with its weight-shaped outputs.
struct ChainMode {
  int mode;                     // FUSE_NONE: store H*dir into out[] | FUSE_CG | FUSE_NEUMANN
  void* const* out;             // FUSE_NONE
  float* fa; float* fb; float* fd;   // fused: flat bases of FuseArgs a / b / d
  const int64_t* starts;        // fused: element offsets of the 2L tensors inside the flat vectors
  float alpha, shift, out_scale;
  int apply_out;
  // FUSE_CG
  FusedWs* ws;
  const double* partRR_old; int nRR_old;
  const double* partPP; int nPP;
  double* partRR_new;
  double* scal;
  float cg_alpha;
  int kpar;                     // iteration parity
  int x_mode;                   // see FuseArgs.x_mode (applies to the lazy slices only)
  int first;                    // first iteration of a solve (Rz(x) accumulator is set, not added to)
  int lazy;                     // the direction at fd is the previous one; this iteration's is fa + beta * fd
  double* rzx_acc;              // FUSE_NEUMANN without an accumulator vector: sum_k Rz(v_k) lands here (head kernel)
  int skip_outputs;             // FUSE_CG: stop after the step length (see bhg_mlp_cg_solve)
  int gemm_mode;                // FUSE_NONE: BHG_MLP_WSK-style mode asked for by the caller (bhg_mlp_hvp_mode)
  const HoistPlan* hoist;       // FUSE_CG + lazy: run the hoisted form of the chain (k_hoist); NULL = the classic chain
  const BetaArgs* beta; int beta_blocks;   // hoisted form: k_cg_beta's work rides in k_hoist's launch (iterations > 0)
  int proj;                     // hoisted: direction products from batch-sized recurrences (k_proj_update); CG: 1 / 2, Neumann: 1
  int stop_after_head;          // projected Neumann: the closing pass that only adds Rz(v_K) to the accumulated Rz sums
  // global-batch CG (bhg_mlp_cg_global_phase): the iteration is cut where the ranks must talk.
  //   gphase 1: the R-chain only; this rank's share of p.H_data p -> php[0] (k_php_local)
  //   gphase 2: step length from the all-reduced php[0] * inv_world, then the outputs with their epilogues
  int gphase; double* php; double inv_world;
  int second;                   // fully projected CG: iteration 1 (the scalars k_proj_step completes are those of the FIRST iteration)
  int lin;                      // fully projected CG: the chain's first product by linearity, update launch inside it (k_wskpl; cg_ctx_init decides)
  int nk;                       // projected Neumann: iteration index (the row-major Rh_0 lives in two slots by its parity, see vnew)
  const void* const* rhs;       // fully projected CG, first iteration: the right-hand side's own tensors (bhg_mlp_cg_solve_rhs) or NULL
  int lin_head;                 // lin on a four-layer net: the update blocks ride in the HEAD launch (k_headu), the pre-head launch is the plain product
};

// One Hessian-vector product of the MLP in direction `dir`, its weight-shaped outputs stored (FUSE_NONE) or consumed
// by the CG / Neumann recurrence while still on chip (fused modes).  On return everything is ordered on `st`.
//   FUSE_NONE / FUSE_NEUMANN: the outputs of layer l only need Rd_l and Rh_{l-1}, so they run on a library-owned side
//     stream beside the R-backward chain (event fork / join).
//   FUSE_CG: the fused epilogues need the step length, which needs the whole R-chain (T2 comes out of the R-backward
//     reduces) — so there is nothing to overlap: ONE stream, no events (an event record costs the stream a ~4 us
//     bubble, a cross-stream wait ~8 us: measured, rocprofv3 timelines in profiles/), and ONE launch for all
//     weight-shaped outputs.
int run_chain(const bhg_mlp* m, const void* const* dir, const ChainMode& cm, hipStream_t st) {
  const int L = m->L, Bp = m->Bp, B = m->B;
  const float rho2 = cm.mode == FUSE_NONE ? m->ridge2 : 0.f;
  const bool cg = cm.mode == FUSE_CG;
  const bool no_side_env = dbg(DBG_mlp_no_side, 0) != 0;    // A/B switches (debug)
  const bool no_fuse = dbg(DBG_mlp_no_fuse, 0) != 0;
  const bool no_outer_all = dbg(DBG_mlp_no_outer_all, 0) != 0;
  const bool neumann_side = dbg(DBG_neumann_side, 0) != 0;    // A/B: fused Neumann with side-stream outputs
  // `single`: one stream, no events, all weight-shaped outputs in one launch after the chain
  const bool single = cg || (cm.mode == FUSE_NEUMANN && !neumann_side && !no_outer_all);
  const bool no_side = no_side_env || single;
  const bool head = use_head(m);
  BHG_REQUIRE(!cg || head, "the fused CG solver needs the narrow-head kernels");
  SideState* ssp = nullptr;
  if (int rc = side_state(&ssp)) return rc;
  SideState& ss = *ssp;
  hipStream_t side = ss.side;
  const int tn = skinny_tile_n();
  const int wsk = wsk_mode(cm.mode, cm.gemm_mode);

  FuseArgs fbase{};
  fbase.scal = cm.scal; fbase.part = cm.partRR_new; fbase.alpha = cm.alpha; fbase.shift = cm.shift;
  fbase.out_scale = cm.out_scale; fbase.apply_out = cm.apply_out;
  fbase.part_stride = cg ? cm.ws->nRR : 0;
  fbase.kpar = cm.kpar;
  auto fuse_at = [&](int tensor, int part_base) {
    FuseArgs f = fbase;
    if (cm.mode != FUSE_NONE) {
      const int64_t o = cm.starts[tensor];
      f.a = cm.fa + o; f.b = cm.fb ? cm.fb + o : nullptr; f.d = cm.fd + o;
      // lazy direction: only the MFMA layers' weight slices (the small slices were updated by k_cg_beta)
      f.lazy = cm.lazy && (tensor & 1) == 0 && !(head && tensor == 2 * (L - 1));
      f.x_mode = (f.lazy || cm.mode == FUSE_NEUMANN) ? cm.x_mode : 0;
      if (!cm.fb) f.x_mode = 1;   // fused CG without a solution vector: nothing reads or writes x
    }
    f.part_base = part_base;
    return f;
  };
  // r'.r' partial slots of the fused CG epilogues: [W_0 tiles][W_1 tiles]...[bias blocks]
  int part_base_w[BHG_MLP_MAX_LAYERS], part_base_bias = 0;
  {
    int base = 0;
    for (int l = 0; l < L; ++l) { part_base_w[l] = base; base += outer_blocks(m, l, head); }
    part_base_bias = base;
  }

  // CG: needs the lazy direction (G(p) = G(r) + beta G(p_old)); Neumann: the direction v is explicit, G(v) directly
  const HoistPlan* hp = ((cg && cm.lazy) || (cm.mode == FUSE_NEUMANN && single)) ? cm.hoist : nullptr;
  // ---- hoisted form: every direction product in ONE grouped launch, then the chain with the constant weights only ---------
  HeadFuse head_fuse{};
  bool fuse_head = false;
  const bool do_chain = cm.gphase != 2;
  // projected CG, not the last iteration (projected Neumann: EVERY iteration): the iteration ends with the G(raw) products
  const bool proj_iter = hp && cm.proj && (cg ? (!cm.apply_out && !cm.skip_outputs) : true);
  // round 4: the chain through the constant weights on PACKED operands (wskp.inc), the per-iteration Gram products T_l / E_l as
  // extra workgroups of the chain launch that consumes the same packed activation (debug keys packed_chain / packed_gram: A/B)
  const bool packed = hp && packed_chain_on(*cm.ws);
  const bool gram_in_chain = packed && proj_iter && !cm.stop_after_head && dbg(DBG_packed_gram, 1) != 0;
  bool sd_in_chain = false;   // first iteration: S_l, D_l rode in the first chain launch as well
  // k_graw (graw.inc) closes the iteration when the Gram matrices arrive packed; its loaders sum two K-split slabs at most.
  // graw_single: what the NEXT iteration's recurrences are told about the layout of G(raw) (one slab per product, not one per pair)
  // (k_graw: K = the padded batch = 128 in registers; round 5: larger padded batches through the K-looped instance k_grawk)
  const bool graw_single = packed && dbg(DBG_packed_gram, 1) != 0 && dbg(DBG_graw_v2, 1) != 0 && graw_batch_ok(Bp);
  const bool graw2 = gram_in_chain && graw_single;
  // rnew: k_graw applies r' = r - alpha Hp to G(r) itself (GrawArgs.rnew) — like graw_single, what the NEXT iteration's recurrences
  // are told (G(r) is up to date, there is no G(raw)); the conditions are those of the step length computed inside k_graw
  const bool rnew = graw_single && cg && cm.proj >= 2 && cm.gphase == 0 && dbg(DBG_proj_small_alone, 0) == 0 &&
                    dbg(DBG_alpha_in_hoist, 1) != 0 && dbg(DBG_rnew_in_graw, 1) != 0;
  // vnew (round 5): the projected Neumann solver's k_graw applies v' = v - alpha (raw + shift v) to G(v) itself and leaves Rh_0(v') packed
  // and row-major — neumann.py:63 has no scalars to wait for — so the update launch at the top of the next iteration (k_proj_update) is
  // gone: SIX launches per iteration instead of seven.  Like rnew a property of the whole solve.  The row-major Rh_0 alternates between
  // m->Rh[0] and a second slot (iteration parity): the Gb_1 tiles of the launch that writes Rh_0(v') still read Rh_0(v).
  const bool vnew = cm.mode == FUSE_NEUMANN && hp && cm.proj && graw_single && L >= 3 && dbg(DBG_neumann_vnew, 1) != 0;
  auto rh0_slot = [&](int k) { return (k & 1) ? cm.ws->hoist + hp->rh0alt_off : m->Rh[0]; };   // Rh_0(v_k), row-major
  // lin: the chain's first product by linearity with the update launch riding in it (k_wskpl, wskpl.inc) — like rnew a property of
  // the whole solve: every iteration's first product, every k_graw (Rh_0(r') for the next one) and cg_iteration (the second bias's
  // direction in slots) follow it
  const bool lin = cm.lin != 0;
  BHG_REQUIRE(!lin || (rnew && hp && hp->lin_ok && cm.beta && cm.beta->nt <= 16 && proj_step_merged() && L >= 4),
              "the linear first product was planned for a solve that cannot run it");
  PstepArgs lin_ps{};   // the update blocks' arguments, built where k_pstep would be launched, used by the first product's launch
  int lin_nu = 0, lin_U = 4;
  bool lin_update_pending = false;   // the update blocks ride in the launch after the first product (k_wskpu)
  auto gp1 = [&](int par) { return cm.ws->hoist + (par ? hp->gp1alt_off : hp->g_off[hp->gf[1]]); };   // Gf_1(p): two slots (lin)
  const bool lin_head = lin && cm.lin_head != 0;
  auto gp2 = [&](int par) { return cm.ws->hoist + (par ? hp->gp2alt_off : hp->g_off[hp->gf[L - 2]]); };   // Gf_{L-2}(p): two slots (lin_head)
  // Gram products riding in chain launches: ONE K slab each — every rider sits in a launch whose tiles have the same K (T_1 with
  // the forward product through W_1; E_l and T_{l+1} with the backward product through W_l), so it ends when they do
  auto tsplit = [&](int K) { return gram_in_chain ? 1 : gram_ksplit(K); };
  auto esplit = [&](int l, int K) { (void)l; return gram_in_chain ? 1 : gram_ksplit(K); };
  (void)tsplit; (void)esplit; (void)sd_in_chain;   // (read by the measurement build's Gram / G(raw) launches only)
  if (hp && do_chain) {
    float* hbase = cm.ws->hoist;
    HoistArgs ha{};
    HoistRedArgs ra{};
    for (int i = 0; i < hp->n; ++i) {
      const int l = hp->layer[i];
      HoistProb& q = ha.p[i];
      q.A = hp->bwd[i] ? m->delta[l] : m->h[l];
      // CG: the RESIDUAL's slice — G(p) = G(r) + beta G(p_old) (k_hoist_reduce); Neumann: the direction itself
      q.Bm = cg ? ((cm.first && cm.rhs) ? static_cast<const float*>(cm.rhs[2 * l]) : cm.fa + cm.starts[2 * l]) : static_cast<const float*>(dir[2 * l]);
      q.slabs = hbase + hp->slab_off[i];
      q.K = hp->K[i]; q.N = hp->N[i]; q.splits = hp->splits[i]; q.rc = hp->bwd[i];
      q.lda = hp->K[i]; q.ldb = hp->bwd[i] ? hp->N[i] : hp->K[i];
      ha.blk0[i] = hp->blk0[i];
      HoistRedProb& rq = ra.p[i];
      rq.slabs = q.slabs; rq.G = hbase + hp->g_off[i]; rq.N = hp->N[i]; rq.splits = hp->splits[i];
      if (!hp->bwd[i] && l == 0) {
        rq.bias = static_cast<const float*>(dir[1]); rq.mask = m->mask[0]; rq.out = m->Rh[0];
        rq.outp = packed ? cm.ws->Rhp[0] : nullptr;
      }
    }
    ha.blk0[hp->n] = hp->blk0[hp->n];
    ha.n = hp->n; ha.Bp = Bp; ha.gemm_blocks = hp->blk0[hp->n];
    const bool proj = cm.proj != 0;
    int rblk = 0;
    for (int i = 0; i < hp->n; ++i) { ra.blk0[i] = rblk; rblk += (Bp * (hp->N[i] / 4) + 255) / 256; }
    ra.blk0[hp->n] = rblk;
    if (!proj || cm.first) {   // the N-sized pass over the residual: every iteration, or (projected CG) the first one only
      ha.do_beta = (cg && !cm.first && cm.beta && !proj) ? 1 : 0;
      ha.beta_blocks = ha.do_beta ? cm.beta_blocks : 0;
      if (ha.do_beta) ha.beta = *cm.beta;
      hipLaunchKernelGGL(k_hoist<FUSE_CG>, dim3(ha.gemm_blocks + (ha.do_beta ? cm.beta_blocks : 0)), dim3(256), 0, st, ha);
      ++g_hoist_launches;
      if (proj && cg) for (int i = 0; i < hp->n; ++i) ra.p[i].G2 = hbase + hp->gr_off[i];
      ra.n = hp->n; ra.Bp = Bp; ra.B = B; ra.first = cg ? cm.first : 1; ra.scal = cm.scal;
      hipLaunchKernelGGL(k_hoist_reduce, dim3(rblk), dim3(256), 0, st, ra);
      // (projected forms: the iteration-invariant Gram matrices S_l = h_l h_l^T, D_l = delta_l delta_l^T are formed once per
      //  solve — in the FIRST iteration's Gram launch, below, beside T_l and E_l: nothing needs them before its G(raw) products)
    } else if (vnew) {         // the last iteration's k_graw applied the update and left Rh_0(v'): nothing to launch
      ++g_proj_iterations;
    } else {                   // projected CG: G(r), G(p) from their batch-sized recurrences — nothing N-sized is read
      ProjArgs pa{};
      for (int i = 0; i < hp->n; ++i) {
        ProjProb& q = pa.p[i];
        q.Gr = hbase + (cg ? hp->gr_off[i] : hp->g_off[i]); q.Gp = hbase + hp->g_off[i]; q.Graw = hbase + hp->graw_off[i]; q.N = hp->N[i];
        // (products with two operand pairs leave one slab per pair, see the G(raw) launch)
        if (!graw_single && graw_split() && !(hp->bwd[i] == 0 && hp->layer[i] == 0)) q.Graw2 = q.Graw + (size_t)Bp * hp->N[i];
        if (rnew) q.Graw = nullptr;   // (the last iteration's k_graw left r' in G(r))
        if (!hp->bwd[i] && hp->layer[i] == 0) {
          q.bias = static_cast<const float*>(dir[1]); q.mask = m->mask[0]; q.out = m->Rh[0];
          q.outp = packed ? cm.ws->Rhp[0] : nullptr;
        }
        pa.blk0[i] = ra.blk0[i];
      }
      pa.blk0[hp->n] = rblk;
      pa.n = hp->n; pa.Bp = Bp; pa.B = B; pa.kpar_prev = cm.kpar ^ 1; pa.shift = cm.shift; pa.scal = cg ? cm.scal : nullptr; pa.alpha = cm.alpha;
      if (cg && cm.proj >= 2 && proj_step_merged()) {   // + the scalars and the small slices' direction update of the LAST iteration
        ProjStepArgs g{};
        g.pa = pa;
        ProjScalArgs& sa = g.sa;
        sa.part_dot = cm.ws->part_dot; sa.dot_blocks = hp->dot_blocks;
        sa.part_raw = cm.ws->part_raw; sa.raw_blocks = graw_blocks(hp, Bp);
        if (graw_single) {   // the last iteration closed with k_graw: three partials per tile workgroup
          const int gt = graw_tile_count(hp, Bp);
          sa.part_dot = cm.ws->part_graw; sa.dot_blocks = gt;
          sa.part_raw = cm.ws->part_graw + 2 * (size_t)gt; sa.raw_blocks = gt;
        }
        sa.part = cm.beta->part; sa.part_stride = cm.ws->nRR;   // the last iteration's epilogue partials (= its partRR_new)
        sa.off0 = part_base_w[L - 1]; sa.n0 = outer_blocks(m, L - 1, head); sa.off1 = part_base_bias; sa.n1 = bias_blocks(m);
        sa.r_small = cm.beta->r; sa.p_small = cm.beta->p; sa.snt = cm.beta->nt;
        int small_total = 0;
        for (int t = 0; t < cm.beta->nt; ++t) { sa.soff[t] = cm.beta->off[t]; sa.slen[t] = cm.beta->len[t]; small_total += cm.beta->len[t]; }
        sa.scal = cm.scal; sa.pscal = cm.ws->pscal; sa.shift = cm.shift; sa.first = cm.second; sa.kpar = cm.kpar ^ 1;
        const int sgrid = small_total > 0 ? (small_total + kThreads - 1) / kThreads : 1;
        g.update_blocks = rblk;
        g.r_b0 = cm.fa + cm.starts[1];
        g.p0_rd = cm.second ? cm.fd + cm.starts[1] : cm.ws->pb0[cm.kpar ^ 1];
        g.p0_wr = cm.ws->pb0[cm.kpar];
        if (graw_single && cm.beta->nt <= 16 && dbg(DBG_pstep_v2, 1) != 0) {   // the same work, arguments laid out for two round trips
          PstepArgs ps{};
          PstepHdr& h = ps.h;
          // U float4s per thread of an update block: every block repeats the scalar phase (~12 KB of partials), so fewer, fatter
          // blocks (944 -> 238 at cfg 2) repeat it less often
          const int pu = lin ? 4 : dbg(DBG_pstep_unroll, 4);
          const int U = pu >= 4 ? 4 : (pu >= 2 ? 2 : 1);
          int ublk = 0;
          for (int i = 0; i < hp->n; ++i) { h.blk0[i] = ublk; ublk += (Bp * (hp->N[i] / 4) + 256 * U - 1) / (256 * U); }
          h.blk0[hp->n] = ublk;
          const int rblk = ublk;   // (shadows the one-float4-per-thread count of k_proj_step / k_hoist_reduce)
          h.n = pa.n; h.Bp = pa.Bp; h.B = pa.B; h.kpar_prev = pa.kpar_prev; h.shift = pa.shift; h.update_blocks = rblk;
          h.scal = sa.scal; h.r_b0 = g.r_b0; h.p0_rd = g.p0_rd; h.p0_wr = g.p0_wr; h.r_small = sa.r_small; h.p_small = sa.p_small;
          h.part_dot = sa.part_dot; h.part_raw = sa.part_raw; h.part = sa.part; h.pscal = sa.pscal;
          h.dot_blocks = sa.dot_blocks; h.raw_blocks = sa.raw_blocks; h.part_stride = sa.part_stride;
          h.off0 = sa.off0; h.n0 = sa.n0; h.off1 = sa.off1; h.n1 = sa.n1; h.first = sa.first; h.kpar = sa.kpar; h.snt = sa.snt;
          for (int i = 0; i < hp->n; ++i) {
            const ProjProb& q = pa.p[i];
            ps.p[i] = {q.Gr, q.Gp, q.Graw, q.bias, q.mask, q.out, q.outp, q.N, q.Graw ? 1 : 0};
            if (lin && i == hp->gf[1]) { ps.p[i].Gp = gp1(cm.kpar ^ 1); ps.p[i].X = gp1(cm.kpar); ps.p[i].xkind = 2; }
            if (lin_head && i == hp->gf[L - 2]) { ps.p[i].Gp = gp2(cm.kpar ^ 1); ps.p[i].X = gp2(cm.kpar); ps.p[i].xkind = 2; }
            if (lin) ps.p[i].outp = nullptr;   // (nobody reads a packed Rh_0(p): the first product runs on Rh_0(r'))
          }
          if (lin) {
            h.p1_rd = cm.second ? cm.fd + cm.starts[3] : cm.ws->pb1[cm.kpar ^ 1];
            h.p1_wr = cm.ws->pb1[cm.kpar];
            h.gran = cm.ws->gran;
            h.rb0_copy = hbase + hp->rb0c_off;
            h.prio = dbg(DBG_lin_prio, 0);
            h.withhold = dbg(DBG_lin_withhold_beta, 0);   // (tests: the pollers' bounded wait, see poll_beta)
            // launched update blocks (debug key lin_nub; 0 = one per virtual block): few, fat blocks leave most CUs to the tiles
            // (1 = as many as leave the launch ONE workgroup per CU with the ragged row tiling: that instance needs 288 registers)
            int nub = dbg(DBG_lin_nub, 0);
            if (nub > 0) {
              WskpProb tq{};
              tq.RA = Bp; tq.RB = m->dims[2]; tq.B = B; tq.nsplit = 1;
              wskp_tiling(&tq, true);
              const int tiles = ((tq.nfull * (tq.RB / 32) + tq.nstrip + 7) & ~7) + (Bp / 32) * (Bp / 32);
              const int room = chip_cus() - tiles - sgrid;
              if (nub == 1 || nub > room) nub = room;
            }
            h.nub = (nub > 0 && nub < rblk) ? nub : 0;
          }
          for (int t = 0; t < sa.snt; ++t) { ps.t.slen[t] = sa.slen[t]; ps.t.soff[t] = sa.soff[t]; }
          if (lin) { lin_ps = ps; lin_nu = (h.nub > 0 ? h.nub : rblk) + sgrid; lin_U = U; }
#ifdef BHG_AB   // (debug key pstep_unroll)
          else if (U == 2) hipLaunchKernelGGL(k_pstep<2>, dim3(rblk + sgrid), dim3(256), 0, st, ps);
          else if (U == 1) hipLaunchKernelGGL(k_pstep<1>, dim3(rblk + sgrid), dim3(256), 0, st, ps);
#endif
          else hipLaunchKernelGGL(k_pstep<4>, dim3(rblk + sgrid), dim3(256), 0, st, ps);
        } else {
#ifdef BHG_AB
          hipLaunchKernelGGL(k_proj_step, dim3(rblk + sgrid), dim3(256), 0, st, g);
#else
          BHG_REQUIRE(false, "internal: every plan of the product takes k_pstep (<= 16 small tensors, the packed closing launch)");
#endif
        }
      } else {
        hipLaunchKernelGGL(k_proj_update, dim3(rblk), dim3(256), 0, st, pa);
      }
      ++g_proj_iterations;
    }
    const int staged_mink = dbg(DBG_hoist_staged_mink, 256);
    // forward chain: Rh_l = mask_l * (Rh_{l-1} W_l^T + Gf_l + c_l)
    for (int l = 1; l + 1 < L; ++l) {
      const int K = m->dims[l], N = m->dims[l + 1];
      const float* c = static_cast<const float*>(dir[2 * l + 1]);
      const float* Gf = hbase + hp->g_off[hp->gf[l]];
      if (packed) {
      if (lin && l == 1) {   // by linearity on Rh_0(r'), the update blocks in the same launch (wskpl.inc)
        BHG_REQUIRE(cm.first || lin_nu > 0, "no update blocks for the linear first product");
        WskplArgs la{};
        WskpProb& q = la.a;
        q.Ap = cm.first ? cm.ws->Rhp[0] : hbase + hp->rh0rp_off;   // (first iteration: p = r, Rh_0 from the N-sized pass)
        q.Bq = cm.ws->Wf[1]; q.RA = Bp; q.RB = N; q.K = K; q.B = B; q.nsplit = 1;
        q.mask = m->mask[1]; q.out = m->Rh[1]; q.outp = cm.ws->Rhp[1];
        q.znew = hbase + hp->z1_off[cm.kpar];
        if (cm.first) { q.addend = Gf; q.bias = c; }
        else {
          q.addend = hbase + hp->gr_off[hp->gf[1]]; q.addend2 = gp1(cm.kpar ^ 1);
          q.bias = cm.fa + cm.starts[3]; q.bias2 = lin_ps.h.p1_rd;
          q.zold = hbase + hp->z1_off[cm.kpar ^ 1];
          q.gran = cm.ws->gran;
        }
        const int rider_blocks = gram_in_chain ? (Bp / 32) * (Bp / 32) : 0;
        wskp_tiling(&q, false);
        int na = (q.nfull * (q.RB / 32) + q.nstrip + 7) & ~7;
        if (dbg(DBG_wskp_ragged, 1) == 2 || (dbg(DBG_wskp_ragged, 1) == 1 && na + rider_blocks > chip_cus()) ||
            (!cm.first && lin_ps.h.nub > 0)) {
          wskp_tiling(&q, true);
          na = (q.nfull * (q.RB / 32) + q.nstrip + 7) & ~7;
        }
        la.na = na;
        if (gram_in_chain) {   // T_1 = h_1 Rh_0^T, linear as well: T_1(p') = T_1(r') + beta T_1(p); two slots by parity
          float* tnew = hbase + hp->tslab_off[1] + (size_t)cm.kpar * Bp * Bp;
          float* tnewp = graw2 ? hbase + hp->tslabp_off[1] + (size_t)cm.kpar * Bp * Bp : nullptr;
          la.r0 = {cm.ws->hpk[1], q.Ap, tnew, tnewp, K, 0};
          la.r0_old = cm.first ? nullptr : hbase + hp->tslab_off[1] + (size_t)(cm.kpar ^ 1) * Bp * Bp;
        }
        la.nu = cm.first ? 0 : lin_nu;
        // (debug key lin_order = 1: small blocks, tiles, update blocks — measured 66 us per iteration against 59.7: the update waves,
        //  dispatched last, are the YOUNGEST on their SIMDs and lose every arbitration to the tiles' waves; dispatched first they win it)
        // update blocks in the NEXT launch — when that is the pre-head product (L = 4), whose tiles leave raw K-split slabs and read
        // nothing the update blocks write; a deeper net's second product adds Gf_2(p) in its epilogue, so its update blocks stay here
        const bool upd_next = !cm.first && L == 4 && dbg(DBG_lin_update_next, 1) != 0 && lin_ps.h.nub == 0;
        if (upd_next) {   // only the small slices' blocks (the first publishes beta) ride here, ahead of the tiles
          la.ns = lin_nu - lin_ps.h.update_blocks;
          la.nt = na + rider_blocks;
        } else if (!cm.first && L > 4 && lin_ps.h.nub == 0) {   // deeper nets: [small blocks][update blocks][tiles] — publisher first
          la.ns = lin_nu - lin_ps.h.update_blocks;
          la.nt = na + rider_blocks;
          la.upd_first = 1;
        } else
        if (!cm.first && dbg(DBG_lin_order, 0) != 0) {
          la.ns = lin_nu - (lin_ps.h.nub > 0 ? lin_ps.h.nub : lin_ps.h.update_blocks);
          la.nt = na + rider_blocks;
        }
        la.ps = lin_ps;
        // (measurement arm lin_update_next = 0 on a four-layer net: the update blocks are dispatched AHEAD of the block that publishes
        //  beta and poll for it — they must all fit on the chip beside it; the wait is bounded either way, this keeps the arm honest)
        BHG_REQUIRE(cm.first || la.ns > 0 || lin_ps.h.update_blocks <= 2 * chip_cus(),
                    "update blocks ahead of the publisher would not all be resident: use the default order");
        const int grid = (upd_next ? la.ns : la.nu) + na + rider_blocks;
        lin_update_pending = upd_next;
        BHG_REQUIRE(cm.first || lin_U == 4, "the update blocks inside k_wskpl are built for four float4 per thread");
#ifdef BHG_AB   // (debug key lin_nub)
        if (la.ps.h.nub > 0) hipLaunchKernelGGL((k_wskpl<2, 4, true>), dim3(grid), dim3(64 * kWskpWaves), 0, st, la); else
#endif
        hipLaunchKernelGGL((k_wskpl<2, 4, false>), dim3(grid), dim3(64 * kWskpWaves), 0, st, la);
        ++g_lin_launches;
        if (cm.first)   // r|b0 as k_graw's tiles will read it (its bias blocks update the slice in the same launch)
          BHG_HIP_CHECK(hipMemcpyAsync(hbase + hp->rb0c_off, cm.fa + cm.starts[1], sizeof(float) * (size_t)m->dims[1],
                                       hipMemcpyDeviceToDevice, st));
      } else {
        WskpBuilder wb;
        WskpProb q{};
        q.Ap = cm.ws->Rhp[l - 1]; q.Bq = cm.ws->Wf[l]; q.RA = Bp; q.RB = N; q.K = K; q.B = B; q.nsplit = 1;
        if (l == L - 2) {   // raw K-split slabs: the head kernel combines them itself (adds Gf and c_l, applies the mask)
          // one workgroup per CU including the Gram product riding along: a second round of workgroups would start when the
          // first ends (probe: 240 + 32 workgroups 9.1 us, 192 + 48 5.4 us)
          const int riders = (gram_in_chain && l == 1) ? (Bp / 32) * (Bp / 32) : 0;
          const int tiles_l = (Bp / 32) * (N / 32);
          int sp = (chip_cus() - riders) / tiles_l;
          const int cap = pick_splits((N + tn - 1) / tn, K, 1);                  // what m->partial was sized for
          if (sp > cap) sp = cap;
          if (sp > K / 64) sp = K / 64;
          if (sp < 1) sp = 1;
          q.nsplit = sp; q.raw = 1; q.out = m->partial;
          // (lin_head, past the first iteration: the head forms Gf(p') = Gf(r') + beta Gf(p) itself — k_headu, see there)
          head_fuse = {m->partial, sp, Bp * N, c, m->mask[l], m->Rh[l], (lin_head && !cm.first) ? hbase + hp->gr_off[hp->gf[l]] : Gf};
          fuse_head = true;
        } else {
          q.bias = c; q.mask = m->mask[l]; q.addend = Gf; q.out = m->Rh[l]; q.outp = cm.ws->Rhp[l];
        }
        wb.add(q);
        if (gram_in_chain && l == 1) {   // T_1 = h_1 Rh_0^T (same K as this launch's tiles)
          WskpProb t{};
          t.Ap = cm.ws->hpk[l]; t.Bq = cm.ws->Rhp[l - 1]; t.RA = Bp; t.RB = Bp; t.K = K; t.B = B;
          t.nsplit = 1; t.raw = 1; t.out = hbase + hp->tslab_off[l]; t.outp = graw2 ? hbase + hp->tslabp_off[l] : nullptr;
          wb.add(t);
        }
        if (lin_update_pending && wb.g.n == 1 && !lin_head) {   // (l = 2: nothing this product reads is written by the update blocks)
          launch_wskpu(wb.g.p[0], lin_ps, lin_ps.h.update_blocks, st);
          lin_update_pending = false;
        } else wb.launch(st);
      }
        if (gram_in_chain && cm.first && l == 1) {   // once per solve: S_l = h_l h_l^T, D_l = delta_l delta_l^T (launches of their own:
          WskpBuilder sb;                            // beside the first chain product they would stretch it — K up to d_0)
          for (int j = 0; j + 1 < L; ++j) {
            WskpProb u{};
            u.Ap = cm.ws->hpk[j]; u.Bq = cm.ws->hpk[j]; u.RA = Bp; u.RB = Bp; u.K = m->dims[j]; u.B = B; u.raw = 1;
            u.nsplit = 1;   // (constant over the solve: one slab)
            u.out = hbase + hp->s_off[j]; u.outp = hbase + hp->sp_off[j];
            if (!sb.add(u)) { sb.launch(st); sb.add(u); }
            if (j >= 1) {
              u.Ap = cm.ws->dpk[j]; u.Bq = cm.ws->dpk[j]; u.K = m->dims[j + 1]; u.out = hbase + hp->d_off[j]; u.outp = hbase + hp->dp_off[j];
              if (!sb.add(u)) { sb.launch(st); sb.add(u); }
            }
          }
          sb.launch(st);
          sd_in_chain = true;
        }
        continue;
      }
      if (l == L - 2) {   // the head kernel combines this layer's slabs itself (and adds Gf)
        GemmArgs a{};
        a.pr[0] = {m->Rh[l - 1], m->W[l], K, K};
        a.pairs = 1; a.M = Bp; a.N = N; a.K = K;
        a.splits = pick_splits((N + tn - 1) / tn, K, 1);
        a.out = m->partial; a.ldo = N; a.out_rows = Bp;
        launch_gemm<LAYOUT_KC, LAYOUT_KC>(a, tn, st);
        head_fuse = {m->partial, a.splits, Bp * N, c, m->mask[l], m->Rh[l], Gf};
        fuse_head = true;
      } else {
        WskArgs w{};
        w.pr[0] = {m->Rh[l - 1], m->W[l], K, K}; w.pairs = 1; w.M = Bp; w.N = N; w.K = K; w.B = B;
        w.bias = c; w.mask = m->mask[l]; w.out = m->Rh[l]; w.addend = Gf;
        launch_gemm_wsk<LAYOUT_KC>(w, st, K >= staged_mink);
      }
    }
    BHG_REQUIRE(!lin_update_pending || lin_head, "the update blocks found no launch to ride in");
    if (lin_update_pending) {   // the head launch with the update blocks behind the head's rows (k_headu)
      const int l = L - 1, K = m->dims[l], N = m->dims[l + 1];
      BHG_REQUIRE(fuse_head && cg && packed && N <= 12 && K <= 512, "k_headu was planned for a head it cannot run");
      HeaduArgs ha{};
      ha.Rh = (const float*)m->Rh[l - 1]; ha.h = m->h[l]; ha.W = m->W[l]; ha.V = static_cast<const float*>(dir[2 * l]);
      ha.cb = static_cast<const float*>(dir[2 * l + 1]); ha.prob = m->prob; ha.sd = m->sd; ha.rd = m->Rd[l];
      ha.K = K; ha.C = N; ha.B = B; ha.mode = HEAD_JVP;
      ha.delta_top = (const float*)m->delta[l]; ha.mask_prev = (const float*)m->mask[l - 1]; ha.rd_prev = m->Rd[l - 1];
      ha.fz = head_fuse;
      ha.partT1 = cm.ws->partT1; ha.partT2h = cm.ws->partT2h; ha.rz_out = cm.ws->rz; ha.rzx_acc = cm.rzx_acc; ha.rzx_first = cm.first;
      ha.rows = Bp; ha.rd_prev_p = cm.ws->Rdp[l - 1];
      ha.addend2 = gp2(cm.kpar ^ 1); ha.gran = cm.ws->gran;
      ha.nu = lin_ps.h.update_blocks; ha.ps = lin_ps; ha.head_first = dbg(DBG_headu_head_first, 1);
      hipLaunchKernelGGL(k_headu<4>, dim3(ha.nu + Bp), dim3(256), (size_t)K * sizeof(float), st, ha);
      lin_update_pending = false;
    } else {
      const int l = L - 1, K = m->dims[l], N = m->dims[l + 1];
      launch_head_forward(st, Bp, (const float*)m->Rh[l - 1], m->h[l], m->W[l], static_cast<const float*>(dir[2 * l]),
                          static_cast<const float*>(dir[2 * l + 1]), m->prob, m->sd, m->Rd[l], K, N, B, HEAD_JVP, nullptr, nullptr,
                          (const float*)m->delta[l], (const float*)m->mask[l - 1], m->Rd[l - 1], &head_fuse,
                          cg ? cm.ws->partT1 : nullptr, cg ? cm.ws->partT2h : nullptr, cg ? cm.ws->rz : nullptr, cm.rzx_acc, cm.first,
                          packed ? cm.ws->Rdp[l - 1] : nullptr);
    }
    if (cm.stop_after_head) {   // (projected Neumann's closing pass: Rz(v_K) is in the accumulator now)
      BHG_HIP_CHECK(hipGetLastError());
      return BHG_OK;
    }
    // backward chain: Rd_{l-1} = mask_{l-1} * (Rd_l W_l + Gb_l); T2_l = 2 <Gb_l, Rh_{l-1}> from the tile epilogue
    for (int l = L - 2; l >= 1; --l) {
      const int K = m->dims[l + 1], N = m->dims[l];
      if (packed) {
        WskpBuilder wb;
        WskpProb q{};
        q.Ap = cm.ws->Rdp[l]; q.Bq = cm.ws->Wb[l]; q.RA = Bp; q.RB = N; q.K = K; q.B = B; q.nsplit = 1;
        q.mask = m->mask[l - 1]; q.addend = hbase + hp->g_off[hp->gb[l]]; q.out = m->Rd[l - 1];
        q.outp = l >= 2 ? cm.ws->Rdp[l - 1] : nullptr;
        if (cg) { q.rh = m->Rh[l - 1]; q.partT2 = cm.ws->partT2 + cm.ws->t2_off[l]; }
        wb.add(q);
        if (gram_in_chain) {   // E_l = delta_l Rd_l^T, and T_{l+1} = h_{l+1} Rh_l^T: both reduce over d_{l+1}, like this launch's tiles
          WskpProb t{};
          t.Ap = cm.ws->dpk[l]; t.Bq = cm.ws->Rdp[l]; t.RA = Bp; t.RB = Bp; t.K = K; t.B = B;
          t.nsplit = 1; t.raw = 1; t.out = hbase + hp->eslab_off[l]; t.outp = graw2 ? hbase + hp->eslabp_off[l] : nullptr;
          wb.add(t);
          if (l + 1 <= L - 2) {
            t.Ap = cm.ws->hpk[l + 1]; t.Bq = cm.ws->Rhp[l];
            t.out = hbase + hp->tslab_off[l + 1]; t.outp = graw2 ? hbase + hp->tslabp_off[l + 1] : nullptr;
            wb.add(t);
          }
        }
        wb.launch(st);
        continue;
      }
      WskArgs w{};
      w.pr[0] = {m->Rd[l], m->W[l], K, N}; w.pairs = 1; w.M = Bp; w.N = N; w.K = K; w.B = B;
      w.mask = m->mask[l - 1]; w.out = m->Rd[l - 1]; w.addend = hbase + hp->g_off[hp->gb[l]];
      if (cg) { w.rh = m->Rh[l - 1]; w.partT2 = cm.ws->partT2 + cm.ws->t2_off[l]; }
      launch_gemm_wsk<LAYOUT_RC>(w, st, K >= staged_mink);
    }
  }
  // ---- R-forward ------------------------------------------------------------------------------------
  for (int l = 0; l < L && !hp && do_chain; ++l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    const float* V = static_cast<const float*>(dir[2 * l]);
    const float* c = static_cast<const float*>(dir[2 * l + 1]);
    if (head && l == L - 1) {
      launch_head_forward(st, Bp, l > 0 ? (const float*)m->Rh[l - 1] : nullptr, m->h[l], m->W[l], V, c, m->prob, m->sd,
                          m->Rd[l], K, N, B, HEAD_JVP, nullptr, nullptr, l > 0 ? (const float*)m->delta[l] : nullptr,
                          l > 0 ? (const float*)m->mask[l - 1] : nullptr, l > 0 ? m->Rd[l - 1] : nullptr,
                          fuse_head ? &head_fuse : nullptr, cg ? cm.ws->partT1 : nullptr, cg ? cm.ws->partT2h : nullptr,
                          cg ? cm.ws->rz : nullptr, cm.rzx_acc, cm.first);
      continue;
    }
    GemmArgs a{};
    a.pr[0] = {m->h[l], V, K, K};                       // h_{l-1} V_l^T
    if (cm.lazy) a.pr[0] = {m->h[l], cm.fa + cm.starts[2 * l], K, K, cm.fd + cm.starts[2 * l], 1};   // V_l = r_l + beta * p_l
    a.scal = cm.scal;
    a.pairs = 1;
    if (l > 0) { a.pr[1] = {m->Rh[l - 1], m->W[l], K, K}; a.pairs = 2; }  // Rh_{l-1} W_l^T
    a.M = Bp; a.N = N; a.K = K;
    const bool to_head = head && l == L - 2 && !no_fuse && (N & 3) == 0 && (size_t)N * sizeof(float) <= 64 * 1024;
    if (wsk_wanted(wsk, a.pairs, K) && !to_head && l + 1 < L) {
      // in-workgroup split-K: the 32 x 32 tile is summed, biased and masked before it leaves the chip (no reduce launch)
      WskArgs w{};
      w.pr[0] = a.pr[0]; w.pr[1] = a.pr[1]; w.pairs = a.pairs; w.M = Bp; w.N = N; w.K = K; w.B = B;
      w.bias = c; w.mask = m->mask[l]; w.out = m->Rh[l]; w.scal = cm.scal;
      if (wsk_eligible(w)) { launch_gemm_wsk<LAYOUT_KC>(w, st); continue; }
    }
    a.splits = pick_splits((N + tn - 1) / tn, K, a.pairs);
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_KC>(a, tn, st);
    const int slab = Bp * N;
    if (to_head) {
      // the head kernel of the next layer combines these slabs itself (one launch less on the chain)
      head_fuse = {m->partial, a.splits, slab, c, m->mask[l], m->Rh[l]};
      fuse_head = true;
    } else if (l + 1 < L) {
      launch_reduce_mask(st, m->partial, a.splits, slab, c, m->mask[l], m->Rh[l], Bp, N, B);
    } else {
      hipLaunchKernelGGL(k_reduce_softmax_jvp, dim3((Bp + 15) / 16), dim3(256), 0, st, (const float*)m->partial,
                         a.splits, slab, c, m->prob, m->sd, m->Rd[l], Bp, N, B);
    }
  }

  // ---- weight-shaped outputs: launch descriptions ----------------------------------------------------------
  auto outer_args = [&](int l, GemmArgs* ga, size_t* lds, bool* fast) {
    const int Mo = m->dims[l + 1], No = m->dims[l];
    const float* V = static_cast<const float*>(dir[2 * l]);
    GemmArgs a{};
    a.pr[0] = {m->Rd[l], m->h[l], Mo, No};              // Rd_l^T h_{l-1}
    a.pairs = 1;
    if (l > 0) { a.pr[1] = {m->delta[l], m->Rh[l - 1], Mo, No}; a.pairs = 2; }  // delta_l^T Rh_{l-1}
    a.M = Mo; a.N = No; a.K = B;                        // only the B valid batch rows contribute
    a.splits = 1;
    a.out = cm.mode == FUSE_NONE ? static_cast<float*>(cm.out[2 * l]) : nullptr; a.ldo = No; a.out_rows = 0;
    a.addend = rho2 != 0.f ? V : nullptr; a.addend_scale = rho2;
    a.kstages = (B + kOH - 1) / kOH < 2 ? 2 : (B + kOH - 1) / kOH;   // <= 64 K rows per pipeline stage
    const int Kh = (((B + a.kstages - 1) / a.kstages) + 1) & ~1;      // (see k_outer)
    *lds = (size_t)Kh * (kTM + kTN) * sizeof(float);
    const size_t lds_c = (size_t)kTM * kCPad * sizeof(float);
    if (*lds < lds_c) *lds = lds_c;
    bool f = Mo % kTM == 0 && No % kTN == 0 && (a.ldo & 3) == 0;
    for (int i = 0; i < a.pairs; ++i) f = f && (a.pr[i].lda & 3) == 0 && (a.pr[i].ldb & 3) == 0;
    if (cm.mode != FUSE_NONE) f = f && (cm.starts[2 * l] & 3) == 0;   // 16-B aligned state slices
    const bool no_fast = dbg(DBG_mlp_no_fast, 0) != 0;
    *fast = f && !no_fast;
    *ga = a;
  };
  auto head_outer_args = [&](int l) {
    HeadOuterArgs ha{};
    ha.rd = m->Rd[l]; ha.h = m->h[l]; ha.delta = m->delta[l]; ha.Rh = l > 0 ? m->Rh[l - 1] : nullptr;
    ha.V = static_cast<const float*>(dir[2 * l]); ha.rho2 = rho2;
    ha.out = cm.mode == FUSE_NONE ? static_cast<float*>(cm.out[2 * l]) : nullptr;
    ha.N = m->dims[l]; ha.C = m->dims[l + 1]; ha.B = B;
    return ha;
  };
  auto launch_outer = [&](int l, hipStream_t s) {
    const FuseArgs fz = fuse_at(2 * l, part_base_w[l]);
    if (head && l == L - 1) {
      const HeadOuterArgs ha = head_outer_args(l);
      const dim3 grid((ha.N + 63) / 64, ha.C);
#define BHG_HEAD_OUTER(RH, MODE) hipLaunchKernelGGL((k_head_outer<RH, MODE>), grid, dim3(256), 0, s, ha, fz)
      if (l > 0) {
        if (cm.mode == FUSE_CG) BHG_HEAD_OUTER(true, FUSE_CG);
        else if (cm.mode == FUSE_NEUMANN) BHG_HEAD_OUTER(true, FUSE_NEUMANN);
        else BHG_HEAD_OUTER(true, FUSE_NONE);
      } else {
        if (cm.mode == FUSE_CG) BHG_HEAD_OUTER(false, FUSE_CG);
        else if (cm.mode == FUSE_NEUMANN) BHG_HEAD_OUTER(false, FUSE_NEUMANN);
        else BHG_HEAD_OUTER(false, FUSE_NONE);
      }
#undef BHG_HEAD_OUTER
      return;
    }
    GemmArgs a; size_t lds; bool fast;
    outer_args(l, &a, &lds, &fast);
    dim3 grid((a.N + kTN - 1) / kTN, (a.M + kTM - 1) / kTM, 1);
#define BHG_OUTER(MODE)                                                                   \
  do {                                                                                    \
    if (fast) hipLaunchKernelGGL((k_outer<true, MODE>), grid, dim3(256), lds, s, a, fz);  \
    else hipLaunchKernelGGL((k_outer<false, MODE>), grid, dim3(256), lds, s, a, fz);      \
  } while (0)
    if (cm.mode == FUSE_CG) BHG_OUTER(FUSE_CG);
    else if (cm.mode == FUSE_NEUMANN) BHG_OUTER(FUSE_NEUMANN);
    else BHG_OUTER(FUSE_NONE);
#undef BHG_OUTER
  };
  BiasArgs ba{};
  int bias_blk = 0;
  {
    ba.L = L; ba.B = B; ba.rho2 = rho2;
    for (int l = 0; l < L; ++l) {
      ba.rd[l] = m->Rd[l];
      ba.c[l] = static_cast<const float*>(dir[2 * l + 1]);
      ba.out[l] = cm.mode == FUSE_NONE ? static_cast<float*>(cm.out[2 * l + 1]) : nullptr;
      ba.foff[l] = cm.mode == FUSE_NONE ? 0 : cm.starts[2 * l + 1];
      ba.n[l] = m->dims[l + 1];
      ba.blk0[l] = bias_blk;
      bias_blk += (m->dims[l + 1] + 63) / 64;
    }
    ba.blk0[L] = bias_blk;
    // fully projected CG (k_proj_step): the first bias's slice of the direction lives in the slot dir[1] names
    if (cg && cm.proj >= 2 && hp && proj_step_merged()) ba.d0 = static_cast<const float*>(dir[1]);
    if (lin) ba.d1 = static_cast<const float*>(dir[3]);
  }
  FuseArgs bias_fz = fbase;
  bias_fz.a = cm.fa; bias_fz.b = cm.fb; bias_fz.d = cm.fd; bias_fz.part_base = part_base_bias;   // offsets travel in ba.foff
  if (cm.mode != FUSE_NONE && !cm.fb) bias_fz.x_mode = 1;
  auto launch_bias = [&](hipStream_t s) {
    if (cm.mode == FUSE_CG) hipLaunchKernelGGL(k_bias_hvp<FUSE_CG>, dim3(bias_blk), dim3(256), 0, s, ba, bias_fz);
    else if (cm.mode == FUSE_NEUMANN) hipLaunchKernelGGL(k_bias_hvp<FUSE_NEUMANN>, dim3(bias_blk), dim3(256), 0, s, ba, bias_fz);
    else hipLaunchKernelGGL(k_bias_hvp<FUSE_NONE>, dim3(bias_blk), dim3(256), 0, s, ba, bias_fz);
  };

  AlphaArgs aa{};   // fused CG: the step length from the batch-sized factors (k_cg_alpha)
  if (cg) {
    aa.partT1 = cm.ws->partT1; aa.partT2h = L >= 2 ? cm.ws->partT2h : nullptr; aa.B = B;
    aa.partT2 = cm.ws->partT2; aa.nT2 = cm.ws->nT2;
    aa.partPP = cm.partPP; aa.nPP = cm.nPP;
    aa.partRR = cm.partRR_old; aa.nRR = cm.nRR_old;
    aa.cg_alpha = cm.cg_alpha; aa.shift = cm.shift; aa.scal = cm.scal;
    aa.rz = cm.ws->rz; aa.rzx = cm.ws->rzx; aa.nrz = B * m->dims[L]; aa.first = cm.first; aa.kpar = cm.kpar;
    if (cm.gphase == 2) { aa.php_ext = cm.php; aa.inv_world = cm.inv_world; }
  }
  // ---- R-backward (main stream) [overlapped with the weight-shaped outputs on the side stream unless FUSE_CG] ----------
  for (int l = L - 1; l >= 1 && !hp && do_chain; --l) {
    if (!single) {   // Rd_l is ready on the main stream here: hand H(W_l) to the side stream
      if (no_side) {
        launch_outer(l, st);
      } else {
        BHG_HIP_CHECK(hipEventRecord(ss.ev_rd[l], st));
        BHG_HIP_CHECK(hipStreamWaitEvent(side, ss.ev_rd[l], 0));
        launch_outer(l, side);
      }
    }
    const int K = m->dims[l + 1], N = m->dims[l];  // Rd_{l-1}[Bp][N] = delta_l[Bp][K] V_l[K][N] + Rd_l W_l
    const float* V = static_cast<const float*>(dir[2 * l]);
    if (head && l == L - 1) continue;  // Rd_{L-2} was produced by the fused k_head_forward
    GemmArgs a{};
    a.pr[0] = {m->delta[l], V, K, N};
    if (cm.lazy) a.pr[0] = {m->delta[l], cm.fa + cm.starts[2 * l], K, N, cm.fd + cm.starts[2 * l], 1};
    a.scal = cm.scal;
    a.pr[1] = {m->Rd[l], m->W[l], K, N};
    a.pairs = 2;
    a.M = Bp; a.N = N; a.K = K;
    const bool wsk_short = wsk_wanted(wsk, 2, K);
    if (wsk_short || (wsk == 3 && !cm.lazy)) {   // mode 3: long R-backward reductions in the LDS-staged form (no lazy direction)
      WskArgs w{};
      w.pr[0] = a.pr[0]; w.pr[1] = a.pr[1]; w.pairs = 2; w.M = Bp; w.N = N; w.K = K; w.B = B;
      w.mask = m->mask[l - 1]; w.out = m->Rd[l - 1]; w.scal = cm.scal;
      if (cg) { w.rh = m->Rh[l - 1]; w.partT2 = cm.ws->partT2 + cm.ws->t2_off[l]; }
      // (the T2 partial slots were carved for the reduce launch's block count: one per 1024 outputs, like the tiles here)
      if (wsk_eligible(w) && (!cg || (Bp / 32) * (N / 32) == reduce_blocks(Bp * N, N))) {
        launch_gemm_wsk<LAYOUT_RC>(w, st, !wsk_short);
        continue;
      }
    }
    a.splits = pick_splits((N + tn - 1) / tn, K, 2);
    if (cg) {
      // the two products land in separate slabs (each workgroup takes twice the K range of ONE pair: same count and
      // length of K loops), so the reduce can dot delta_l V_l with Rh_{l-1} on its way: T2_l
      if (a.splits < 2) a.splits = 2;
      a.pair_split = a.splits / 2;
    }
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_RC>(a, tn, st);
    const int slab = Bp * N;
    if (cg) {
      const int blocks = reduce_blocks(slab, N);
      double* pt2 = cm.ws->partT2 + cm.ws->t2_off[l];
      if ((N & 3) == 0)
        hipLaunchKernelGGL(k_reduce_mask_t2<4>, dim3(blocks), dim3(256), 0, st, (const float*)m->partial, a.pair_split, a.splits,
                           slab, (const float*)m->mask[l - 1], (const float*)m->Rh[l - 1], m->Rd[l - 1], Bp, N, B, pt2);
      else
        hipLaunchKernelGGL(k_reduce_mask_t2<1>, dim3(blocks), dim3(256), 0, st, (const float*)m->partial, a.pair_split, a.splits,
                           slab, (const float*)m->mask[l - 1], (const float*)m->Rh[l - 1], m->Rd[l - 1], Bp, N, B, pt2);
    } else {
      launch_reduce_mask(st, m->partial, a.splits, slab, nullptr, m->mask[l - 1], m->Rd[l - 1], Bp, N, B);
    }
  }

  if (cg && cm.gphase == 1) {   // global-batch CG: the chain is done; this rank's p.H_data p for the caller's all-reduce
    hipLaunchKernelGGL(k_php_local, dim3(1), dim3(kThreads), 0, st, aa, cm.php);
    BHG_HIP_CHECK(hipGetLastError());
    return BHG_OK;
  }
  if (single) {
    // ---- the step length, then every weight-shaped output with the recurrence in its epilogue
    // projected CG, not the last iteration: the step length rides in the launch of this iteration's Gram products
    // (projected Neumann: EVERY iteration — the closing pass needs G(raw) of the last one)
    const bool alpha_alone = dbg(DBG_proj_alpha_alone, 0) != 0 || gram_in_chain;   // A/B (packed Gram products: no Gram launch to ride in)
    const bool small_alone = dbg(DBG_proj_small_alone, 0) != 0;   // A/B
    const bool small_in_graw = proj_iter && (cm.proj >= 2 || !cg) && !small_alone;
    // fully projected CG with the Gram products in the chain launches: nothing is left between the chain and the G(raw) launch but
    // the step length — and its only readers inside that launch are the small slices' blocks, which recompute it from the same
    // partials (alpha_compute, bit-identical in every block); the first of them publishes it and completes Rz(x)
    const bool alpha_in_hoist = cg && proj_iter && cm.proj >= 2 && small_in_graw && gram_in_chain && cm.gphase == 0 &&
                                dbg(DBG_alpha_in_hoist, 1) != 0;
    const bool alpha_in_gram = cg && proj_iter && !alpha_alone;
    if (cg && !alpha_in_gram && !alpha_in_hoist) hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(kThreads), 0, st, aa);
    if (cg && cm.skip_outputs) {   // last iteration of a solve without a solution vector: r', p' and x are all dead
      BHG_HIP_CHECK(hipGetLastError());
      return BHG_OK;
    }
    if (proj_iter) {   // projected CG: G(raw) of this iteration for the next one's recurrences
      float* hbase = cm.ws->hoist;
#ifndef BHG_AB
      // (product: the Gram products rode in the chain launches and the closing launch is k_graw / k_grawk in EVERY plan — the hoist plan
      //  carves the packed operands whenever it is taken, the padded batch is a multiple of 128; round 3's Gram launch and its k_hoist
      //  form of the G(raw) products live on in the measurement build)
      BHG_REQUIRE(graw2, "internal: the packed closing launch must apply");
#else
      WskGroupArgs g{};
      int blk = 0;
      for (int l = 1; l + 1 < L && !gram_in_chain; ++l) {
        const int st_ = gram_ksplit(m->dims[l]), se_ = gram_ksplit(m->dims[l + 1]);
        g.p[g.n] = {m->h[l], m->Rh[l - 1], hbase + hp->tslab_off[l], Bp, Bp, m->dims[l], B, st_};       // T_l = h_l Rh_{l-1}^T
        g.blk0[g.n++] = blk; blk += (Bp / 32) * (Bp / 32) * st_;
        g.p[g.n] = {m->delta[l], m->Rd[l], hbase + hp->eslab_off[l], Bp, Bp, m->dims[l + 1], B, se_};   // E_l = delta_l Rd_l^T
        g.blk0[g.n++] = blk; blk += (Bp / 32) * (Bp / 32) * se_;
      }
      for (int l = 0; cm.first && !sd_in_chain && l + 1 < L; ++l) {   // once per solve: S_l, D_l
        g.p[g.n] = {m->h[l], m->h[l], hbase + hp->s_off[l], Bp, Bp, m->dims[l], B, 1};
        g.blk0[g.n++] = blk; blk += (Bp / 32) * (Bp / 32);
        if (l >= 1) {
          g.p[g.n] = {m->delta[l], m->delta[l], hbase + hp->d_off[l], Bp, Bp, m->dims[l + 1], B, 1};
          g.blk0[g.n++] = blk; blk += (Bp / 32) * (Bp / 32);
        }
      }
      g.blk0[g.n] = blk;
      if (alpha_in_gram) { g.do_alpha = 1; g.alpha = aa; }
      if (blk + (alpha_in_gram ? 1 : 0) > 0) launch_wsk_group(g, blk + (alpha_in_gram ? 1 : 0), st);
#endif
      const bool full = cg && cm.proj >= 2;   // fully projected CG: r.raw, p.raw, raw.raw from batch-sized arrays (k_pstep)
      SmallOutArgs so{};
      int small_blocks = 0;
      if (small_in_graw) {   // the small slices' outputs (head weight, biases) with their CG epilogue: block classes of the closing launch
        so.head = head_outer_args(L - 1);
        so.hf = fuse_at(2 * (L - 1), part_base_w[L - 1]);
        so.head_gx = (so.head.N + 63) / 64;
        so.head_blocks = so.head_gx * so.head.C;
        so.head_has_rh = L > 1;
        so.ba.L = ba.L; so.ba.B = ba.B; so.ba.rho2 = ba.rho2;
        for (int l = 0; l < L; ++l) {
          so.ba.rd[l] = ba.rd[l]; so.ba.c[l] = ba.c[l]; so.ba.out[l] = ba.out[l]; so.ba.n[l] = ba.n[l]; so.ba.blk0[l] = ba.blk0[l];
          so.ba.foff[l] = ba.foff[l];
        }
        so.ba.blk0[L] = ba.blk0[L];
        so.ba.d0 = ba.d0; so.ba.d1 = lin ? static_cast<const float*>(dir[3]) : nullptr;
        so.bf = bias_fz;
        so.bias_blocks = bias_blk;
        small_blocks = so.head_blocks + bias_blk;
      }
      if (graw2) {   // round 4: packed Gram matrices -> k_graw (64 x 32 tiles, inner products in the tile epilogue, one slab per product)
        GrawArgs ka{};
        const int ct = graw_cols(hp, Bp);
        // CT = 32: the small slices' blocks lead the grid; CT = 64: the tiles do (one CU each), the small blocks fill second slots
        const bool small_first = ct == 32;
        int gb = small_first ? (small_blocks + 15) & ~15 : 0;   // (every table entry a multiple of 16)
        int real_tiles = 0;
        for (int i = 0; i < hp->n; ++i) {
          const int l = hp->layer[i];
          GrawProb& q = ka.p[i];
          if (!hp->bwd[i]) {   // Gf_l(raw) = S_l Rd_l + T_l delta_l
            q.A1 = hbase + hp->sp_off[l]; q.B1 = m->Rd[l];
            if (l >= 1) { q.A2 = hbase + hp->tslabp_off[l] + ((lin && l == 1) ? (size_t)cm.kpar * Bp * Bp : 0); q.B2 = m->delta[l]; }
          } else {             // Gb_l(raw) = E_l h_l + D_l Rh_{l-1}
            q.A1 = hbase + hp->eslabp_off[l]; q.B1 = m->h[l];
            q.A2 = hbase + hp->dp_off[l]; q.B2 = (vnew && l == 1) ? rh0_slot(cm.nk) : m->Rh[l - 1];
          }
          q.Gr = q.Gp = q.B1;   // (always loadable)
          if (vnew) q.Gr = q.Gp = hbase + hp->g_off[i];   // G(v): updated in place by the tile that owns the element
          if (full) {           // the inner products' partner: Rd_l (= B1) forward, Rh_{l-1} (= B2) backward
            q.Gr = hbase + hp->gr_off[i];
            q.Gp = (lin && i == hp->gf[1]) ? gp1(cm.kpar) : ((lin_head && i == hp->gf[L - 2]) ? gp2(cm.kpar) : hbase + hp->g_off[i]);
            q.dots = hp->bwd[i] ? 2 : 1;
          }
          q.Graw = hbase + hp->graw_off[i]; q.N = hp->N[i];
          q.pb0 = real_tiles;
          real_tiles += (Bp / 64) * (hp->N[i] / ct);
          // every column tile with all its Bp / 64 row groups: groups of one tile 8 blocks apart (one XCD), column tiles padded to 8
          ka.blk0[i] = gb; gb += (Bp / 64) * (((hp->N[i] / ct) + 7) & ~7);
        }
        ka.blk0[hp->n] = gb; ka.tile_end = gb;
        BHG_REQUIRE(real_tiles == graw_tile_count(hp, Bp) && real_tiles <= hp->graw_tiles, "tile count of k_graw and of the plan disagree");
        ka.n = hp->n; ka.Bp = Bp; ka.B = B;
        ka.part = cm.ws->part_graw; ka.npart = real_tiles;
        ka.small_blocks = small_blocks; ka.so = so;
        ka.small0 = small_first ? 0 : gb;
        if (!small_first) gb += small_blocks;
        if (alpha_in_hoist) { ka.do_alpha = 1; ka.alpha = aa; }
        BHG_REQUIRE(!rnew || (alpha_in_hoist && full), "k_graw was to apply the residual step but has no step length");
        ka.rnew = (rnew || vnew) ? 1 : 0;
        if (vnew) {   // Rh_0(v') for the next iteration's first product (packed) and for its k_graw (row-major, the other slot)
          ka.nalpha = cm.alpha; ka.nshift = cm.shift;
          ka.rh0p = cm.ws->Rhp[0]; ka.rh0 = rh0_slot(cm.nk + 1); ka.mask0 = m->mask[0];
          ka.rb0 = ka.pb0s = static_cast<const float*>(dir[1]); ka.rh0_prod = hp->gf[0];
        }
        if (lin) {   // Rh_0(r') for the next iteration's first product; beta's granules cleared for its publication
          ka.rh0p = hbase + hp->rh0rp_off; ka.mask0 = m->mask[0]; ka.rb0 = hbase + hp->rb0c_off;
          ka.pb0s = static_cast<const float*>(dir[1]); ka.gran = cm.ws->gran; ka.rh0_prod = hp->gf[0];
        }
#ifdef BHG_AB   // (debug key graw_cols = 64)
        if (ct == 64) {
          if (cg) hipLaunchKernelGGL(k_graw64<FUSE_CG>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
          else hipLaunchKernelGGL(k_graw64<FUSE_NEUMANN>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
        } else
#endif
        if (Bp != 128) {   // the K-looped instance (32-column tiles)
          if (cg) hipLaunchKernelGGL(k_grawk<FUSE_CG>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
          else hipLaunchKernelGGL(k_grawk<FUSE_NEUMANN>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
        } else if (cg) hipLaunchKernelGGL(k_graw<FUSE_CG>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
        else hipLaunchKernelGGL(k_graw<FUSE_NEUMANN>, dim3(gb), dim3(64 * kGrawWaves), 0, st, ka);
      }
#ifdef BHG_AB
      HoistArgs ga{};
      int gblk = 0;
      const int ntm = Bp / kTM;
      for (int i = 0; i < hp->n && !graw2; ++i) {
        const int l = hp->layer[i];
        HoistProb& q = ga.p[i];
        if (!hp->bwd[i]) {   // Gf_l(raw) = S_l Rd_l + T_l delta_l
          q.A = hbase + hp->s_off[l]; q.Bm = m->Rd[l];
          if (l >= 1) { q.A2 = hbase + hp->tslab_off[l]; q.B2m = m->delta[l]; q.a2_slabs = gram_in_chain ? tsplit(m->dims[l]) : gram_ksplit(m->dims[l]); }
        } else {             // Gb_l(raw) = E_l h_l + D_l Rh_{l-1}
          q.A = hbase + hp->eslab_off[l]; q.Bm = m->h[l]; q.a_slabs = gram_in_chain ? esplit(l, m->dims[l + 1]) : gram_ksplit(m->dims[l + 1]);
          q.A2 = hbase + hp->d_off[l]; q.B2m = m->Rh[l - 1];
        }
        q.slabs = hbase + hp->graw_off[i];
        q.a_slab_stride = Bp * Bp;
        if (full) q.X = hp->bwd[i] ? (const float*)m->Rh[l - 1] : (const float*)m->Rd[l];   // raw.raw's share: <X, G(raw)>
        q.K = Bp; q.N = hp->N[i]; q.splits = (q.A2 && graw_split()) ? 2 : 1; q.rc = 1; q.lda = Bp; q.ldb = hp->N[i];
        ga.blk0[i] = gblk; gblk += (hp->N[i] / 32) * ntm * q.splits;
      }
      ga.blk0[hp->n] = gblk;
      ga.n = hp->n; ga.Bp = Bp; ga.gemm_blocks = gblk; ga.do_beta = 0;
      if (full && !graw2) {   // the projected inner products r.raw, p.raw ride behind the tiles
        int dblk = 0, nd = 0;
        for (int i = 0; i < hp->n; ++i) {
          const int l = hp->layer[i];
          ga.dp[nd] = {hbase + hp->gr_off[i], hbase + hp->g_off[i], hp->bwd[i] ? (const float*)m->Rh[l - 1] : (const float*)m->Rd[l], hp->N[i]};
          ga.dblk0[nd++] = dblk; dblk += dot_blocks_of(Bp * (hp->N[i] / 4));
        }
        ga.dblk0[nd] = dblk;
        ga.nd = nd; ga.dot_blocks = dblk; ga.B = B; ga.part_dot = cm.ws->part_dot; ga.part_raw = cm.ws->part_raw;
        BHG_REQUIRE(gblk == graw_blocks(hp, Bp), "tile count of the G(raw) launch and of its raw.raw partials disagree");
        BHG_REQUIRE(dblk == hp->dot_blocks, "dot block count of the plan and of the launch disagree");
      }
      if (small_in_graw) {
        ga.so = so;
        ga.small_blocks = small_blocks;
        if (alpha_in_hoist) { ga.do_alpha = 1; ga.alpha = aa; }
      }
      if (graw2) {
      } else if (cg) hipLaunchKernelGGL(k_hoist<FUSE_CG>, dim3(gblk + ga.dot_blocks + ga.small_blocks), dim3(256), 0, st, ga);
      else hipLaunchKernelGGL(k_hoist<FUSE_NEUMANN>, dim3(gblk + ga.dot_blocks + ga.small_blocks), dim3(256), 0, st, ga);
#endif
    }
    // one launch for all outputs when every MFMA layer is all-interior
    // (fully projected CG: the MFMA layers' slices of r and p are not materialised — only the small slices' blocks launch)
    const bool proj_full = hp && ((cg && cm.proj >= 2) || (!cg && cm.proj));   // no N-sized state: only the small slices' blocks
    const int n_mfma = proj_full ? 0 : (head ? L - 1 : L);
    OuterAllArgs oa{};
    bool all_fast = !no_outer_all && n_mfma <= kOuterAllMax && head;
    size_t lds_max = 0;
    int order[BHG_MLP_MAX_LAYERS];
    // dispatch order = tile order: the layer with the most tiles first (measured 260 vs 256 steps/s against
    // "two-pair tiles first"; BHG_OUTER_ORDER_BY_WORK selects the latter)
    const bool work_first = dbg(DBG_outer_order_by_work, 0) != 0;
    auto weight = [&](int l) { return !work_first ? (double)outer_blocks(m, l, head) : (l > 0 ? 2.0 : 1.0) * 1e9 + outer_blocks(m, l, head); };
    for (int i = 0; i < n_mfma; ++i) order[i] = i;
    for (int i = 1; i < n_mfma; ++i)   // insertion sort, descending
      for (int j = i; j > 0 && weight(order[j]) > weight(order[j - 1]); --j) { int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
    int blk = 0;
    for (int i = 0; i < n_mfma && all_fast; ++i) {
      const int l = order[i];
      size_t lds; bool fast;
      outer_args(l, &oa.g[i], &lds, &fast);
      all_fast = all_fast && fast;
      if (lds > lds_max) lds_max = lds;
      oa.f[i] = fuse_at(2 * l, part_base_w[l]);
      oa.gx[i] = (oa.g[i].N + kTN - 1) / kTN;
      oa.blk0[i] = blk;
      blk += outer_blocks(m, l, head);
    }
    if (small_in_graw) {
      // (the small slices' blocks already ran in k_hoist's launch)
    } else if (all_fast) {
      oa.n = n_mfma;
      oa.blk0[n_mfma] = blk;
      oa.head = head_outer_args(L - 1);
      oa.hf = fuse_at(2 * (L - 1), part_base_w[L - 1]);
      oa.head_gx = (oa.head.N + 63) / 64;
      oa.head_blocks = oa.head_gx * oa.head.C;
      oa.head_has_rh = L > 1;
      oa.bf = bias_fz;
      const int total = blk + oa.head_blocks + bias_blk;
      if (lds_max < (size_t)kTM * kCPad * sizeof(float)) lds_max = (size_t)kTM * kCPad * sizeof(float);
      const bool no_pre = dbg(DBG_outer_no_pre, 0) != 0;   // A/B runs
      (void)no_pre;
      const int stagger = dbg(DBG_outer_stagger, 1);
      oa.stagger = stagger;
#ifdef BHG_AB
      if (no_pre) {
        if (cg) hipLaunchKernelGGL((k_outer_all<FUSE_CG, false>), dim3(total), dim3(256), lds_max, st, oa, ba);
        else hipLaunchKernelGGL((k_outer_all<FUSE_NEUMANN, false>), dim3(total), dim3(256), lds_max, st, oa, ba);
      } else
#endif
      {
        if (cg) hipLaunchKernelGGL((k_outer_all<FUSE_CG, true>), dim3(total), dim3(256), lds_max, st, oa, ba);
        else hipLaunchKernelGGL((k_outer_all<FUSE_NEUMANN, true>), dim3(total), dim3(256), lds_max, st, oa, ba);
      }
    } else {
      for (int l = L - 1; l >= 0; --l) launch_outer(l, st);
      launch_bias(st);
    }
#ifdef BHG_AB
    if (proj_full && cg && !proj_step_merged()) {   // r'.r', beta, p'.p' of the iteration from batch-sized quantities (k_proj_scalars)
      BHG_REQUIRE(all_fast || small_in_graw, "the fully projected CG solver needs the single-launch output path");
      ProjScalArgs sa{};
      sa.part_dot = cm.ws->part_dot; sa.dot_blocks = hp->dot_blocks;
      sa.part_raw = cm.ws->part_raw; sa.raw_blocks = graw_blocks(hp, Bp);
      if (graw_single) {
        const int gt = graw_tile_count(hp, Bp);
        sa.part_dot = cm.ws->part_graw; sa.dot_blocks = gt;
        sa.part_raw = cm.ws->part_graw + 2 * (size_t)gt; sa.raw_blocks = gt;
      }
      sa.part = cm.partRR_new; sa.part_stride = cm.ws->nRR;
      sa.off0 = part_base_w[L - 1]; sa.n0 = outer_blocks(m, L - 1, head); sa.off1 = part_base_bias; sa.n1 = bias_blk;
      sa.r_small = cm.beta->r; sa.p_small = cm.beta->p; sa.snt = cm.beta->nt;
      for (int t = 0; t < cm.beta->nt; ++t) { sa.soff[t] = cm.beta->off[t]; sa.slen[t] = cm.beta->len[t]; }
      sa.scal = cm.scal; sa.pscal = cm.ws->pscal; sa.shift = cm.shift; sa.first = cm.first; sa.kpar = cm.kpar;
      int small_total = 0;
      for (int t = 0; t < cm.beta->nt; ++t) small_total += cm.beta->len[t];
      const int sgrid = small_total > 0 ? (small_total + kThreads - 1) / kThreads : 1;
      hipLaunchKernelGGL(k_proj_scalars, dim3(sgrid), dim3(kThreads), 0, st, sa);
    }
#endif
    BHG_HIP_CHECK(hipGetLastError());
    return BHG_OK;
  }

  if (L > 1 && !no_side) {  // the bias terms need every Rd_l (complete on the main stream now); they run beside H(W_0)
    BHG_HIP_CHECK(hipEventRecord(ss.ev_rd[0], st));
    BHG_HIP_CHECK(hipStreamWaitEvent(side, ss.ev_rd[0], 0));
    launch_bias(side);
  } else {
    launch_bias(st);
  }
  launch_outer(0, st);  // needs Rd_0, the end of the chain
  if (L > 1 && !no_side) {
    BHG_HIP_CHECK(hipEventRecord(ss.ev_join, side));
    BHG_HIP_CHECK(hipStreamWaitEvent(st, ss.ev_join, 0));
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int check_mlp(const bhg_mlp* m) {
  BHG_REQUIRE(m, "NULL descriptor");
  BHG_REQUIRE(m->L >= 1 && m->L <= BHG_MLP_MAX_LAYERS, "unsupported layer count");
  BHG_REQUIRE(m->Bp > 0 && m->Bp % kTM == 0 && m->B >= 1 && m->B <= m->Bp, "Bp must be a multiple of 128 rows >= B");
  return BHG_OK;
}

#include "mlp/fx.inc"   // round 6: the factor-exchange form of the global-batch CG solver (rectangular Gram blocks; kernels + host phases)

}  // namespace
}  // namespace bhg

using namespace bhg;

extern "C" {

size_t bhg_mlp_partial_floats(const bhg_mlp* m) {
  if (!m || m->L < 1 || m->L > BHG_MLP_MAX_LAYERS) return 0;
  size_t mx = 0;
  for (int l = 0; l < m->L; ++l) {
    // R-forward of layer l (N = dims[l+1], K = dims[l]) and R-backward into layer l (N = dims[l], K = dims[l+1])
    const int Nf = m->dims[l + 1], Kf = m->dims[l];
    const int sf = pick_splits((Nf + skinny_tile_n() - 1) / skinny_tile_n(), Kf, 2);
    mx = mx > (size_t)sf * m->Bp * Nf ? mx : (size_t)sf * m->Bp * Nf;
    int sb = pick_splits((Kf + skinny_tile_n() - 1) / skinny_tile_n(), Nf, 2);
    if (sb < 2) sb = 2;   // fused CG: the two operand pairs of the R-backward GEMM land in separate slabs
    mx = mx > (size_t)sb * m->Bp * Kf ? mx : (size_t)sb * m->Bp * Kf;
  }
  return mx;
}

int bhg_mlp_hvp(const bhg_mlp* m, const void* const* dir, void* const* out, void* stream) {
  return bhg_mlp_hvp_mode(m, dir, out, 0, stream);
}

int bhg_mlp_hvp_mode(const bhg_mlp* m, const void* const* dir, void* const* out, int gemm_mode, void* stream) {
  if (int rc = check_mlp(m)) return rc;
  BHG_REQUIRE(gemm_mode >= 0 && gemm_mode <= 3, "gemm_mode is 0 .. 3");
  BHG_REQUIRE(dir && out, "NULL argument");
  BHG_REQUIRE(m->partial && m->partial_floats >= bhg_mlp_partial_floats(m), "split-K scratch too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t t_a, t_b;
  const bool timed = span_begin(BHG_TIMING_MLP_HVP, &t_a, &t_b);
  if (timed) BHG_HIP_CHECK(hipEventRecord(t_a, st));
  ChainMode cm{};
  cm.mode = FUSE_NONE;
  cm.out = out;
  cm.gemm_mode = gemm_mode;
  if (int rc = run_chain(m, dir, cm, st)) return rc;
  if (timed) BHG_HIP_CHECK(hipEventRecord(t_b, st));
  return BHG_OK;
}

// ---- fused solvers: K iterations of HVP + recurrence without an N-sized H*direction vector ---------------------------
int64_t bhg_mlp_wsk_launches(void) { return bhg::g_wsk_launches; }
int64_t bhg_mlp_hoist_launches(void) { return bhg::g_hoist_launches; }
int64_t bhg_mlp_proj_iterations(void) { return bhg::g_proj_iterations; }
int64_t bhg_mlp_lin_launches(void) { return bhg::g_lin_launches; }

int bhg_mlp_neumann_mixed_coeff(const bhg_mlp* m, const void* const* v_last, const int64_t* labels, float* coeff, float alpha,
                                int K, int projected, void* fws, size_t fws_bytes, void* stream) {
  // coefficient of the accumulator-free Neumann solve: p_final = -alpha * sum_{k=0..K} v_k  (neumann.py:64,66 and the
  // negation of 45/54)  =>  coeff = -alpha * ( coeff(v_K)  +  (prob - onehot) . sum_{k<K} Rz(v_k) / B )
  BHG_REQUIRE(fws && fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  // projected (what bhg_mlp_neumann_solve reported for THIS solve): its closing pass already added Rz(v_K) to the sum
  if (!projected)
    if (int rc = bhg_mlp_mixed_coeff(m, v_last, labels, coeff, stream)) return rc;   // coeff(v_K): one R-forward
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  hipLaunchKernelGGL(k_coeff_from_rzx, dim3((m->Bp + kThreads - 1) / kThreads), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     K > 0 ? (const double*)w.rzx : (const double*)nullptr, m->prob, labels, coeff, m->Bp, m->dims[m->L], m->B, -alpha,
                     projected ? 0.f : -alpha);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_mlp_supports_fused_solve(const bhg_mlp* m) {
  const bool off = dbg(DBG_mlp_no_fused_solve, 0) != 0;   // A/B switch: callers fall back to HVP + recurrence kernel
  return !off && m && m->L >= 1 && m->L <= BHG_MLP_MAX_LAYERS && m->Bp > 0 && m->Bp % kTM == 0 && use_head(m);
}

// The time-out word of the in-launch beta exchange (poll_beta, wskp.inc): non-zero after a solve whose pollers gave up.
void* bhg_mlp_timeout_flag_dev(const bhg_mlp* m, void* fws) {
  if (!m || !fws || m->L < 1 || m->L > BHG_MLP_MAX_LAYERS || m->Bp <= 0) return nullptr;
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  return w.gran + 1;
}

size_t bhg_mlp_fused_ws_bytes(const bhg_mlp* m) {
  if (!m || m->L < 1 || m->L > BHG_MLP_MAX_LAYERS || m->Bp <= 0) return 0;
  FusedWs w;
  carve_fused_ws(m, nullptr, &w);
  return w.bytes;
}

static int solve_common_checks(const bhg_mlp* m, const int64_t* starts, const void* fws, size_t fws_bytes) {
  if (int rc = check_mlp(m)) return rc;
  BHG_REQUIRE(bhg_mlp_supports_fused_solve(m), "fused solve needs a narrow classifier head (<= 256 classes, feature width % 4 == 0)");
  BHG_REQUIRE(starts && fws, "NULL argument");
  BHG_REQUIRE(m->partial && m->partial_floats >= bhg_mlp_partial_floats(m), "split-K scratch too small");
  BHG_REQUIRE(fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  for (int l = 0; l < m->L; ++l) BHG_REQUIRE((starts[2 * l] & 3) == 0, "weight slices of the flat vectors must be 16-byte aligned");
  return BHG_OK;
}

// Everything one iteration of the fused CG solver needs besides its index (built per call: a few hundred bytes of host work).
struct CgCtx {
  const bhg_mlp* m; float* x; float* r; float* p; const int64_t* starts; const bhg_chunk* chunks_dev; int n_chunks, K;
  float cg_alpha, shift;
  FusedWs w; double* scal; const double* partR0; int n_init, pgrid, bgrid;
  bool lazy, hoist, lin, lin_head; int proj_level;
  const void* const* rhs;
  BetaArgs ba; HoistPlan hplan;
  const void* dir[2 * BHG_MLP_MAX_LAYERS];
};
// global: the global-batch solver (bhg_mlp_cg_global_phase) — lazy direction, N-sized residual (it is what the ranks exchange)
static void cg_ctx_init(CgCtx* c, const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts, const bhg_chunk* chunks_dev,
                        int n_chunks, int K, float cg_alpha, float hvp_shift, void* ws, void* fws, bool global) {
  c->m = m; c->x = x; c->r = r; c->p = p; c->starts = starts; c->chunks_dev = chunks_dev; c->n_chunks = n_chunks; c->K = K;
  c->cg_alpha = cg_alpha; c->shift = hvp_shift; c->rhs = nullptr;
  carve_fused_ws(m, fws, &c->w);
  char* wsb = static_cast<char*>(ws);
  c->scal = reinterpret_cast<double*>(wsb + kWsScal);
  c->partR0 = reinterpret_cast<const double*>(wsb + kWsPartR);   // r.r partials of bhg_cg_init (r = p there)
  c->n_init = n_chunks < kMaxBlocks ? n_chunks : kMaxBlocks;
  for (int i = 0; i < 2 * m->L; ++i) c->dir[i] = p + starts[i];
  c->pgrid = n_chunks < kMaxBlocks ? n_chunks : kMaxBlocks;
  // Direction update between two iterations: lazy (default; see k_cg_beta) or the 12*N-byte k_cg_pdir pass (A/B switch)
  const bool eager = dbg(DBG_cg_eager_p, 0) != 0;
  c->lazy = global || !eager;
  BetaArgs& ba = c->ba;
  ba = BetaArgs{};
  int small_total = 0;
  {
    const bool head = use_head(m);
    ba.stride = c->w.nRR; ba.n = c->w.nRR; ba.scal = c->scal; ba.r = r; ba.p = p;
    for (int l = 0; l < m->L; ++l) {   // biases
      ba.off[ba.nt] = starts[2 * l + 1]; ba.len[ba.nt] = m->dims[l + 1]; small_total += ba.len[ba.nt]; ++ba.nt;
    }
    if (head) {                        // narrow head weight
      ba.off[ba.nt] = starts[2 * (m->L - 1)]; ba.len[ba.nt] = m->dims[m->L] * m->dims[m->L - 1]; small_total += ba.len[ba.nt]; ++ba.nt;
    }
  }
  c->bgrid = small_total > 0 ? (small_total + kThreads - 1) / kThreads : 1;   // one element of the small slices per thread
  c->hplan.ok = false;
  if (c->lazy && hoist_mode() != 0) hoist_plan(m, &c->hplan);
  c->hoist = c->lazy && c->hplan.ok;
  // projection level: 1 = G(r) by recurrence, the N-sized r / p still updated by k_outer_all (needed when the caller wants x);
  // 2 = fully projected (default without a solution vector): no N-sized state after the first iteration
  // (BHG_MLP_PROJ: 0 off | 1 default | 9 level 1 even without a solution vector — the A/B arm of level 2)
  c->proj_level = (!c->hoist || !c->hplan.proj_ok || proj_mode() == 0 || global) ? 0 : ((proj_mode() == 9 || x) ? 1 : 2);
  // the chain's first product by linearity (k_wskpl): fully projected CG closing with k_graw that applies the residual step (the
  // conditions of run_chain's graw_single and rnew), a net with a product between the first and the pre-head one, few small tensors
  c->lin = c->proj_level == 2 && c->hplan.lin_ok && packed_chain_on(c->w) && dbg(DBG_packed_gram, 1) != 0 && dbg(DBG_graw_v2, 1) != 0 &&
           graw_batch_ok(m->Bp) && dbg(DBG_proj_small_alone, 0) == 0 && dbg(DBG_alpha_in_hoist, 1) != 0 && dbg(DBG_rnew_in_graw, 1) != 0 &&
           c->ba.nt <= 16 && proj_step_merged() && dbg(DBG_pstep_v2, 1) != 0 && dbg(DBG_lin_first, 1) != 0;
  // ... and on a four-layer net the update blocks ride in the head launch (k_headu, headu.inc) rather than in the pre-head one: the head's
  // prefetching instance must apply (<= 12 classes, last hidden width <= 512), and like lin it holds for the whole solve (slot parity)
  c->lin_head = c->lin && m->L == 4 && dbg(DBG_lin_update_next, 1) != 0 && dbg(DBG_lin_update_in_head, 1) != 0 && dbg(DBG_lin_nub, 0) == 0 &&
                m->dims[m->L] <= 12 && m->dims[m->L - 1] <= 512 && dbg(DBG_head_no_prefetch, 0) == 0;
}
// gphase 0: the whole iteration (one rank) | 1: up to this rank's p.H_data p | 2: from the step length on (see ChainMode)
static int cg_iteration(CgCtx* c, int k, int gphase, double* php, double inv_world, hipStream_t st) {
  const bhg_mlp* m = c->m;
  FusedWs& w = c->w;
  const int K = c->K;
  const bool lazy = c->lazy, hoist = c->hoist;
  hipEvent_t ta, tb, tc, td;
  const bool timed = gphase == 0 && span_begin(BHG_TIMING_MLP_HVP, &ta, &tb);
  const bool timed_it = gphase == 0 && span_begin(BHG_TIMING_MLP_CG_ITER, &tc, &td);
  if (timed_it) BHG_HIP_CHECK(hipEventRecord(tc, st));
  if (lazy && k > 0 && gphase != 2) {   // beta, p.p of the coming direction, direction update of the small slices
    c->ba.part = w.partRR[k & 1];
    // hoisted, not projected: inside k_hoist; fully projected: k_proj_scalars (end of the last iteration) + k_proj_update
    if (!hoist || c->proj_level == 1) hipLaunchKernelGGL(k_cg_beta, dim3(c->bgrid), dim3(kThreads), 0, st, c->ba);
  }
  if (timed) BHG_HIP_CHECK(hipEventRecord(ta, st));
  ChainMode cm{};
  cm.mode = FUSE_CG;
  cm.fa = c->r; cm.fb = c->x; cm.fd = c->p; cm.starts = c->starts;
  cm.shift = c->shift; cm.cg_alpha = c->cg_alpha;
  cm.apply_out = k == K - 1; cm.out_scale = -c->cg_alpha;   // cg.py:56 and the negation of cg.py:59/68
  cm.ws = &w; cm.scal = c->scal;
  cm.partRR_old = c->partR0;            // k > 0: r.r is the scalar scal[S_RR_NEW] (k_cg_beta / k_cg_pdir of the last iteration)
  cm.nRR_old = k == 0 ? c->n_init : 0;
  cm.partPP = k == 0 ? c->partR0 : w.partPP;   // p = r after the init, so p.p = r.r
  cm.nPP = k == 0 ? c->n_init : (lazy ? 0 : c->pgrid);
  cm.partRR_new = w.partRR[(k + 1) & 1];
  // iteration 0: beta = 0 (bhg_cg_init zeroes the scalars) and p = r, so "r + beta * p" is the initial direction
  cm.lazy = lazy;
  // x is read and written every other iteration (FuseArgs.x_mode): even iterations defer, odd ones catch up
  const bool x_every = dbg(DBG_cg_x_every_iter, 0) != 0;   // A/B switch
  cm.x_mode = (lazy && !x_every) ? ((k & 1) ? 2 : (k + 1 < K ? 1 : 0)) : 0;
  if (!c->x) cm.x_mode = 1;
  // Without a solution vector the LAST iteration ends with its step length: alpha_{K-1} completes Rz(x) (k_cg_alpha), and
  // nothing reads the residual, the direction or x of cg.py:49-53 after it — the weight-shaped outputs are not computed.
  cm.skip_outputs = (!c->x && k == K - 1) ? 1 : 0;
  cm.first = k == 0;
  cm.kpar = k & 1;
  cm.hoist = hoist ? &c->hplan : nullptr;
  cm.beta = &c->ba; cm.beta_blocks = c->bgrid;
  cm.proj = c->proj_level;
  cm.gphase = gphase; cm.php = php; cm.inv_world = inv_world;
  cm.second = k == 1;
  if (c->proj_level == 2 && proj_step_merged())   // the first bias's direction: flat p in iteration 0, then the slot of the parity
    c->dir[1] = k == 0 ? static_cast<const void*>(c->p + c->starts[1]) : static_cast<const void*>(w.pb0[k & 1]);
  if (c->lin) c->dir[3] = k == 0 ? static_cast<const void*>(c->p + c->starts[3]) : static_cast<const void*>(w.pb1[k & 1]);
  cm.lin = c->lin ? 1 : 0;
  cm.lin_head = c->lin_head ? 1 : 0;
  cm.rhs = c->rhs;
  if (int rc = run_chain(m, c->dir, cm, st)) return rc;
  if (timed) BHG_HIP_CHECK(hipEventRecord(tb, st));
#ifdef BHG_AB   // (key cg_eager_p; the product's direction is always lazy)
  if (!lazy && k + 1 < K)   // the direction is not used after the last iteration (the reference computes and drops it)
    hipLaunchKernelGGL(k_cg_pdir, dim3(c->pgrid), dim3(kThreads), 0, st, c->chunks_dev, c->n_chunks, (const float*)c->r, c->p,
                       (const double*)w.partRR[(k + 1) & 1], w.nRR, w.partPP, c->scal);
#endif
  if (timed_it) BHG_HIP_CHECK(hipEventRecord(td, st));
  return BHG_OK;
}

int bhg_mlp_cg_solve(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts,
                     const bhg_chunk* chunks_dev, int n_chunks, int K, float cg_alpha, float hvp_shift, void* ws,
                     void* fws, size_t fws_bytes, void* stream) {
  return bhg_mlp_cg_solve_rhs(m, x, r, p, starts, chunks_dev, n_chunks, K, cg_alpha, hvp_shift, ws, fws, fws_bytes, nullptr, stream);
}

unsigned long long bhg_mlp_cg_state_mask(const bhg_mlp* m, int has_x) {
  if (!m || m->L < 1 || m->L > BHG_MLP_MAX_LAYERS || 2 * m->L > 64 || has_x || !bhg_mlp_supports_fused_solve(m)) return ~0ull;
  // cg_ctx_init's decision with x == NULL (nothing in it depends on the state pointers)
  HoistPlan hp;
  hp.ok = false;
  if (dbg(DBG_cg_eager_p, 0) == 0 && hoist_mode() != 0) hoist_plan(m, &hp);
  const bool level2 = hp.ok && hp.proj_ok && proj_mode() != 0 && proj_mode() != 9;
  if (!level2 || dbg(DBG_cg_rhs_direct, 1) == 0) return ~0ull;
  unsigned long long mask = 0ull;
  for (int l = 0; l < m->L; ++l) mask |= 1ull << (2 * l + 1);   // biases
  mask |= 1ull << (2 * (m->L - 1));                              // the narrow head weight (use_head holds: the fused solve needs it)
  return mask;
}

int bhg_mlp_cg_solve_rhs(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts,
                         const bhg_chunk* chunks_dev, int n_chunks, int K, float cg_alpha, float hvp_shift, void* ws,
                         void* fws, size_t fws_bytes, const void* const* rhs, void* stream) {
  if (int rc = solve_common_checks(m, starts, fws, fws_bytes)) return rc;
  // x == NULL: the N-sized solution vector is not materialised.  For this structure the mixed second derivative only
  // needs Rz(x) = sum_k alpha_k Rz(p_k), which k_cg_alpha accumulates from the batch-sized Rz of every direction
  // (bhg_mlp_cg_mixed_coeff), so a caller that wants the hypergradient and not x itself saves x's share of the
  // recurrence traffic (8*N bytes every other iteration) and its zeroing in bhg_cg_init.
  BHG_REQUIRE(r && p && ws && chunks_dev, "NULL argument");
  BHG_REQUIRE(K >= 0 && n_chunks > 0, "bad size");
  if (K == 0) return BHG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  CgCtx c;
  cg_ctx_init(&c, m, x, r, p, starts, chunks_dev, n_chunks, K, cg_alpha, hvp_shift, ws, fws, false);
  BHG_REQUIRE(!rhs || c.proj_level == 2, "rhs names the right-hand side for the FULLY PROJECTED solver only (see bhg_mlp_cg_state_mask)");
  if (rhs)
    for (int l = 0; l + 1 < m->L; ++l) BHG_REQUIRE(rhs[2 * l] && ((uintptr_t)rhs[2 * l] & 15) == 0, "rhs tensors must be 16-byte aligned device pointers");
  c.rhs = rhs;
  if (c.hoist && packed_chain_on(c.w) && !m->prepacked)   // (prepacked: bhg_mlp_forward_packed / _backward_packed left the packed operands)
    if (int rc = pack_operands(m, c.w, st)) return rc;
  for (int k = 0; k < K; ++k)
    if (int rc = cg_iteration(&c, k, 0, nullptr, 1.0, st)) return rc;
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- global-batch CG: ONE inner problem whose batch is spread over `world` ranks (one process per GPU) --------------------------
// State x, r, p REPLICATED on every rank (bit-identical: every scalar below is computed from identical or all-reduced data);
// the Hessian is the mean over the ranks of the local ones (+ shift * I).  With the same r, p and step length on every rank,
//     r - alpha (mean_g H_g) p  =  mean_g ( r - alpha H_g p ):
// the one-pass chain with its fused epilogues runs on every rank AS IS on the local batch, and the MEAN of the locally updated
// residuals is the global one.  Per iteration the ranks exchange
//     8 bytes   this rank's p.H_data p (batch-sized factors of the R-chain)      SUM   before the step length
//     4*N bytes the locally updated residual                                     MEAN  after the outputs
// and nothing else: x += alpha p is replicated work on identical data; r'.r', r'.p, p.p come from one 8*N-byte pass over the
// exchanged residual (identical on every rank, so beta is too).  The last iteration exchanges the 8 bytes only (nothing reads its
// residual).  The caller drives, per iteration k:
//     phase BHG_CG_GLOBAL_CHAIN;  all-reduce(SUM) php[0];  phase BHG_CG_GLOBAL_UPDATE;
//     if (k + 1 < K) { all-reduce(MEAN) r;  phase BHG_CG_GLOBAL_DOTS; }
// on ONE stream (the collectives ordered with it).  world == 1 needs no collective and gives bhg_mlp_cg_solve's iteration with
// the residual's dot products taken in a pass of their own.
int bhg_mlp_cg_global_phase(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts, const bhg_chunk* chunks_dev,
                            int n_chunks, int k, int K, int phase, int world, double* php, float cg_alpha, float hvp_shift,
                            void* ws, void* fws, size_t fws_bytes, void* stream) {
  if (int rc = solve_common_checks(m, starts, fws, fws_bytes)) return rc;
  BHG_REQUIRE(r && p && ws && chunks_dev && php, "NULL argument");
  BHG_REQUIRE(K > 0 && k >= 0 && k < K && n_chunks > 0 && world >= 1, "bad size");
  BHG_REQUIRE(phase == BHG_CG_GLOBAL_CHAIN || phase == BHG_CG_GLOBAL_UPDATE || phase == BHG_CG_GLOBAL_DOTS, "unknown phase");
  hipStream_t st = static_cast<hipStream_t>(stream);
  CgCtx c;
  cg_ctx_init(&c, m, x, r, p, starts, chunks_dev, n_chunks, K, cg_alpha, hvp_shift, ws, fws, true);
  if (phase == BHG_CG_GLOBAL_DOTS) {
    BHG_REQUIRE(k + 1 < K, "the last iteration has no direction update");
    int grid = n_chunks < c.w.nRR ? n_chunks : c.w.nRR;
    if (grid > kMaxBlocks) grid = kMaxBlocks;
    hipLaunchKernelGGL(k_cg_global_dots, dim3(grid), dim3(kThreads), 0, st, chunks_dev, n_chunks, (const float*)r, (const float*)p,
                       c.w.partRR[(k + 1) & 1], c.w.nRR, c.w.nRR);
    BHG_HIP_CHECK(hipGetLastError());
    return BHG_OK;
  }
  if (k == 0 && phase == BHG_CG_GLOBAL_CHAIN && c.hoist && packed_chain_on(c.w) && !m->prepacked)
    if (int rc = pack_operands(m, c.w, st)) return rc;
  if (int rc = cg_iteration(&c, k, phase == BHG_CG_GLOBAL_CHAIN ? 1 : 2, php, 1.0 / (double)world, st)) return rc;
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- global-batch CG, FACTOR-EXCHANGE form (mlp/fx.inc; include/bhg.h) ---------------------------------------------------------------------
int bhg_mlp_fx_supported(const bhg_mlp* m, int world) {
  if (!m || m->L < 1 || m->L > BHG_MLP_MAX_LAYERS || m->Bp <= 0 || m->Bp % kTM != 0 || !bhg_mlp_supports_fused_solve(m)) return 0;
  FxPlan fp;
  fx_plan(m, world, &fp);
  return fp.ok && dbg(DBG_packed_chain, 1) != 0 ? 1 : 0;
}
size_t bhg_mlp_fx_ws_bytes(const bhg_mlp* m, int world) {
  if (!bhg_mlp_fx_supported(m, world)) return 0;
  FxPlan fp; fx_plan(m, world, &fp);
  FxWs x; fx_carve(m, fp, nullptr, &x);
  return x.bytes;
}
size_t bhg_mlp_fx_const_floats(const bhg_mlp* m) {
  if (!bhg_mlp_fx_supported(m, 1)) return 0;
  FxPlan fp; fx_plan(m, 1, &fp);
  return fp.const_floats;
}
size_t bhg_mlp_fx_slab_floats(const bhg_mlp* m) {
  if (!bhg_mlp_fx_supported(m, 1)) return 0;
  FxPlan fp; fx_plan(m, 1, &fp);
  return fp.slab_floats;
}
size_t bhg_mlp_fx_scal_doubles(const bhg_mlp* m) {
  if (!bhg_mlp_fx_supported(m, 1)) return 0;
  FxPlan fp; fx_plan(m, 1, &fp);
  return fp.scal_doubles;
}
// algo 0: CG (cg.py:34-56; k = 0 .. K-1, END at k = K-1) | 1: Neumann (neumann.py:59-66; CHAIN for k = 0 .. K — the last one is the closing half
// pass, forward chain + head only: Rz(v_K) — GRAM for k = 0 .. K-1, END at k = K)
static int fx_phase(int algo, const bhg_mlp* m, const void* const* rhs, int k, int K, int phase, int world, int rank, float* const_all,
                    float* slab_all, double* scal_all, float alpha, float hvp_shift, void* fws, size_t fws_bytes, void* xws, size_t xws_bytes,
                    void* stream) {
  if (int rc = check_mlp(m)) return rc;
  const bool neumann = algo == 1;
  BHG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad world size / rank");
  BHG_REQUIRE(bhg_mlp_fx_supported(m, world), "this network does not take the factor-exchange form (bhg_mlp_fx_supported)");
  BHG_REQUIRE(phase == BHG_CG_FX_BEGIN || phase == BHG_CG_FX_CHAIN || phase == BHG_CG_FX_GRAM || phase == BHG_CG_FX_END, "unknown phase");
  BHG_REQUIRE(K > 0 && k >= 0 && (k < K || (neumann && k == K && (phase == BHG_CG_FX_CHAIN || phase == BHG_CG_FX_END))), "bad iteration index");
  BHG_REQUIRE(const_all && slab_all && scal_all && fws && xws, "NULL argument");
  BHG_REQUIRE(m->partial && m->partial_floats >= bhg_mlp_partial_floats(m), "split-K scratch too small");
  BHG_REQUIRE(fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  BHG_REQUIRE(xws_bytes >= bhg_mlp_fx_ws_bytes(m, world), "factor-exchange workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FxPlan fp;   // (rebuilt on every call from the descriptor alone — a few microseconds of host work; no state between calls: reentrant)
  fx_plan(m, world, &fp);
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  FxWs x;
  fx_carve(m, fp, xws, &x);
  const bhg_mlp ml = fx_local(m, fp, slab_all, rank);
  double* scal_mine = scal_all + (size_t)rank * fp.scal_doubles;
  if (phase == BHG_CG_FX_BEGIN) {
    BHG_REQUIRE(k == 0, "BEGIN belongs to iteration 0");
    return fx_begin(m, fp, const_all, rank, st);
  }
  if (phase == BHG_CG_FX_CHAIN) {
    if (k == 0) {
      BHG_REQUIRE(rhs, "iteration 0 reads the right-hand side");
      for (int l = 0; l + 1 < m->L; ++l) BHG_REQUIRE(rhs[2 * l] && ((uintptr_t)rhs[2 * l] & 15) == 0, "rhs tensors must be 16-byte aligned device pointers");
      for (int i = 0; i < 2 * m->L; ++i) BHG_REQUIRE(rhs[i], "rhs holds 2 L device pointers");
      if (int rc = fx_first(m, &ml, fp, w, x, rhs, const_all, rank, st, neumann)) return rc;
    } else {
      if (int rc = fx_step(&ml, fp, w, x, scal_all, k - 1, false, alpha, hvp_shift, st, neumann)) return rc;
    }
    return fx_chain(&ml, fp, w, x, k & 1, st, neumann && k == K);
  }
  if (phase == BHG_CG_FX_GRAM) {
    BHG_REQUIRE(k < K, "GRAM belongs to iterations 0 .. K-1");
    return fx_gram(&ml, fp, w, x, slab_all, const_all, scal_mine, k & 1, hvp_shift, st);
  }
  BHG_REQUIRE(k == (neumann ? K : K - 1), "END belongs to the last iteration");
  return fx_step(&ml, fp, w, x, scal_all, k, true, alpha, hvp_shift, st, neumann);
}
int bhg_mlp_cg_fx_phase(const bhg_mlp* m, const void* const* rhs, int k, int K, int phase, int world, int rank, float* const_all,
                        float* slab_all, double* scal_all, float cg_alpha, float hvp_shift, void* fws, size_t fws_bytes, void* xws,
                        size_t xws_bytes, void* stream) {
  return fx_phase(0, m, rhs, k, K, phase, world, rank, const_all, slab_all, scal_all, cg_alpha, hvp_shift, fws, fws_bytes, xws, xws_bytes, stream);
}
int bhg_mlp_neumann_fx_phase(const bhg_mlp* m, const void* const* rhs, int k, int K, int phase, int world, int rank, float* const_all,
                             float* slab_all, double* scal_all, float alpha, float hvp_shift, void* fws, size_t fws_bytes, void* xws,
                             size_t xws_bytes, void* stream) {
  return fx_phase(1, m, rhs, k, K, phase, world, rank, const_all, slab_all, scal_all, alpha, hvp_shift, fws, fws_bytes, xws, xws_bytes, stream);
}

int bhg_mlp_neumann_solve(const bhg_mlp* m, float* v0, float* v1, float* p, const int64_t* starts, int K, float alpha,
                          float hvp_shift, void* fws, size_t fws_bytes, int* projected_out, void* stream) {
  if (int rc = solve_common_checks(m, starts, fws, fws_bytes)) return rc;
  // p == NULL: the N-sized accumulator is not materialised.  The mixed second derivative is linear in the direction and
  // only needs Rz(p_K) = sum_{k=0..K} Rz(v_k): the head kernel of iteration k leaves Rz(v_k) anyway (k < K, summed into the
  // workspace), and bhg_mlp_neumann_mixed_coeff adds the last term with the one R-forward pass the mixed coefficient
  // costs in any case (in direction v_K instead of p_K).
  BHG_REQUIRE(v0 && v1, "NULL argument");
  BHG_REQUIRE(K >= 0, "bad size");
  BHG_REQUIRE(p || use_head(m), "the accumulator-free Neumann solver needs the narrow-head kernels");
  hipStream_t st = static_cast<hipStream_t>(stream);
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  // Hoisting alone (direction products on the N-sized v every iteration, BHG_MLP_HOIST=2) neither gains nor loses for Neumann
  // (656 vs 656 steps/s at cfg 2: no step length, no lazy direction, no beta launch to save) and is an A/B arm only.  The
  // PROJECTED form (default without an accumulator vector) is the Neumann twin of the fully projected CG solver:
  //     G(v_{k+1}) = G(v_k) - alpha (G(raw_k) + shift G(v_k)),   G(raw) from B x B Gram matrices (see k_proj_update)
  // — no scalars at all, nothing N-sized after the first iteration; the small slices (biases, head weight) keep their
  // explicit epilogues.  The mixed coefficient needs Rz(sum_k v_k): the head kernel sums Rz(v_k), k < K, as before, and a
  // closing half pass (update + forward chain + head) adds Rz(v_K) — instead of bhg_mlp_neumann_mixed_coeff's R-forward over
  // the N-sized v_K, which no longer exists.
  HoistPlan hplan;
  hplan.ok = false;
  const bool want_proj = !p && K > 0 && proj_mode() != 0 && hoist_mode() != 0;
  if (hoist_mode() == 2 || want_proj) hoist_plan(m, &hplan);
  const bool proj = want_proj && hplan.ok && hplan.proj_ok && use_head(m);
  if (projected_out) *projected_out = proj ? 1 : 0;   // the caller hands it to bhg_mlp_neumann_mixed_coeff (no hidden per-workspace state)
  if (hplan.ok && K > 0 && packed_chain_on(w) && !m->prepacked)
    if (int rc = pack_operands(m, w, st)) return rc;
  for (int k = 0; k < K; ++k) {
    float* vin = (k & 1) ? v1 : v0;
    float* vout = (k & 1) ? v0 : v1;
    const void* dir[2 * BHG_MLP_MAX_LAYERS];
    for (int i = 0; i < 2 * m->L; ++i) dir[i] = vin + starts[i];
    hipEvent_t ta, tb;
    const bool timed = span_begin(BHG_TIMING_MLP_HVP, &ta, &tb);
    if (timed) BHG_HIP_CHECK(hipEventRecord(ta, st));
    ChainMode cm{};
    cm.mode = FUSE_NEUMANN;
    cm.fa = vout; cm.fb = p; cm.fd = vin; cm.starts = starts;
    cm.alpha = alpha; cm.shift = hvp_shift;
    cm.apply_out = k == K - 1; cm.out_scale = -alpha;   // neumann.py:66 and the negation of neumann.py:45/54
    // the accumulator p is read and written every OTHER iteration (FuseArgs.x_mode): even iterations defer, odd catch up
    const bool p_every = dbg(DBG_neumann_p_every_iter, 0) != 0;   // A/B switch
    cm.x_mode = p_every ? 0 : ((k & 1) ? 2 : (k + 1 < K ? 1 : 0));
    cm.first = k == 0;
    if (!p) { cm.x_mode = 1; cm.rzx_acc = w.rzx; }
    cm.ws = &w;
    cm.hoist = hplan.ok ? &hplan : nullptr;   // every direction product in one grouped launch (k_hoist), as in the CG solver
    cm.proj = proj ? 1 : 0;
    cm.nk = k;
    if (int rc = run_chain(m, dir, cm, st)) return rc;
    if (timed) BHG_HIP_CHECK(hipEventRecord(tb, st));
  }
  if (proj) {   // closing pass: Rz(v_K) joins the sum (G(v_K) by the recurrence, small slices of v_K from the last epilogues)
    float* vin = (K & 1) ? v1 : v0;
    const void* dir[2 * BHG_MLP_MAX_LAYERS];
    for (int i = 0; i < 2 * m->L; ++i) dir[i] = vin + starts[i];
    ChainMode cm{};
    cm.mode = FUSE_NEUMANN;
    cm.fa = (K & 1) ? v0 : v1; cm.fb = nullptr; cm.fd = vin; cm.starts = starts;
    cm.alpha = alpha; cm.shift = hvp_shift;
    cm.x_mode = 1; cm.rzx_acc = w.rzx; cm.first = 0;
    cm.ws = &w; cm.hoist = &hplan; cm.proj = 1; cm.stop_after_head = 1; cm.nk = K;
    if (int rc = run_chain(m, dir, cm, st)) return rc;
  }
  return BHG_OK;
}

int bhg_mlp_cg_mixed_coeff(const bhg_mlp* m, const int64_t* labels, float* coeff, float cg_alpha, void* fws,
                           size_t fws_bytes, void* stream) {
  if (int rc = check_mlp(m)) return rc;
  BHG_REQUIRE(labels && coeff && fws, "NULL argument");
  BHG_REQUIRE(fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  // x_final = -cg_alpha * sum_k alpha_k p_k  (cg.py:56 and the negation of 59/68)  =>  Rz(x_final) = -cg_alpha * RzX
  hipLaunchKernelGGL(k_coeff_from_rzx, dim3((m->Bp + kThreads - 1) / kThreads), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     (const double*)w.rzx, m->prob, labels, coeff, m->Bp, m->dims[m->L], m->B, -cg_alpha, 0.f);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- once-per-step passes (replace ~80 ATen dispatches around the K loop with 3 native calls) ----------------
static int check_head_problem(const bhg_mlp* m) {
  BHG_REQUIRE(m, "NULL descriptor");
  BHG_REQUIRE(m->L >= 1 && m->L <= BHG_MLP_MAX_LAYERS, "unsupported layer count");
  BHG_REQUIRE(m->Bp > 0 && m->Bp % kTM == 0 && m->B >= 1 && m->B <= m->Bp, "Bp must be a multiple of 128 rows >= B");
  BHG_REQUIRE(m->partial && m->partial_floats >= bhg_mlp_partial_floats(m), "split-K scratch too small");
  return BHG_OK;
}
// narrow head (<= 256 classes, feature width % 4 == 0): the latency-optimised head kernels; anything wider: the output layer as one more
// split-K product + a row kernel (round 6: the ATen fallback of rounds 1-5 is gone from the product)
static bool narrow_head(const bhg_mlp* m) { return m->dims[m->L] <= kSmallC && (m->dims[m->L - 1] & 3) == 0; }

int bhg_mlp_supports_native_prepare(const bhg_mlp* m) {
  return m && m->L >= 1 && m->L <= BHG_MLP_MAX_LAYERS && m->Bp > 0 && m->Bp % kTM == 0;
}

}  // extern "C"
namespace bhg {
namespace {
// One workgroup per sample row, any class count: z (logits, in place) -> softmax; ce[b] = -log softmax(z)[y_b].  Rows >= B: zeros.
__global__ __launch_bounds__(kThreads) void k_softmax_ce_rows(float* __restrict__ z, const int64_t* __restrict__ labels,
                                                              float* __restrict__ ce, int C, int B) {
  __shared__ double red[kWaves];
  __shared__ float redf[kWaves];
  const int b = blockIdx.x, t = threadIdx.x;
  float* row = z + (int64_t)b * C;
  if (b >= B) {
    for (int c = t; c < C; c += kThreads) row[c] = 0.f;
    if (t == 0) ce[b] = 0.f;
    return;
  }
  float mx = -INFINITY;
  for (int c = t; c < C; c += kThreads) mx = fmaxf(mx, row[c]);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) redf[t >> 6] = mx;
  __syncthreads();
  mx = redf[0];
  for (int w = 1; w < kWaves; ++w) mx = fmaxf(mx, redf[w]);
  double s = 0.0;
  for (int c = t; c < C; c += kThreads) s += (double)expf(row[c] - mx);
  s = block_sum(s, red);
  const float lse = mx + logf((float)s);
  const float zy = row[(int)labels[b]];
  __syncthreads();   // every thread has read z[y] before the row is overwritten
  for (int c = t; c < C; c += kThreads) row[c] = expf(row[c] - lse);
  if (t == 0) ce[b] = lse - zy;
}
// coeff[b] = sum_c (prob[b][c] - [c == y_b]) * rz[b][c] / B   (the mixed-derivative coefficient; rows >= B: 0)
__global__ __launch_bounds__(kThreads) void k_coeff_rows(const float* __restrict__ prob, const float* __restrict__ rz,
                                                         const int64_t* __restrict__ labels, float* __restrict__ coeff, int C, int B) {
  __shared__ double red[kWaves];
  const int b = blockIdx.x, t = threadIdx.x;
  if (b >= B) { if (t == 0) coeff[b] = 0.f; return; }
  const int y = (int)labels[b];
  double s = 0.0;
  for (int c = t; c < C; c += kThreads) s += (double)((prob[(int64_t)b * C + c] - (c == y ? 1.f : 0.f)) * rz[(int64_t)b * C + c]);
  s = block_sum(s, red);
  if (t == 0) coeff[b] = (float)(s / (double)B);
}
}  // namespace
}  // namespace bhg
using namespace bhg;
extern "C" {

// Forward pass: h[l+1] = relu(h[l] W_l^T + b_l), mask[l]; prob = softmax(z); ce[b] = -log prob[b][y_b].
// h[0] (the padded input batch) must be filled by the caller; h[1..], mask[], prob and ce are written.
int bhg_mlp_forward(const bhg_mlp* m, const void* const* bias, const int64_t* labels, float* ce, void* stream) {
  if (int rc = check_head_problem(m)) return rc;
  BHG_REQUIRE(bias && labels && ce, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = m->L, Bp = m->Bp, B = m->B;
  for (int l = 0; l < L; ++l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    const float* b = static_cast<const float*>(bias[l]);
    if (l == L - 1 && narrow_head(m)) {
      launch_head_forward(st, Bp, nullptr, m->h[l], m->W[l] /* unused: no Rh */, m->W[l], b, nullptr, nullptr,
                          const_cast<float*>(m->prob), K, N, B, HEAD_LOGITS, labels, ce, nullptr, nullptr, nullptr);
      break;
    }
    GemmArgs a{};
    a.pr[0] = {m->h[l], m->W[l], K, K};
    a.pairs = 1;
    a.M = Bp; a.N = N; a.K = K;
    const int tn = skinny_tile_n();
    a.splits = pick_splits((N + tn - 1) / tn, K, 1);
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_KC>(a, tn, st);
    if (l == L - 1) {   // a wide output layer: logits = product + bias into prob, then the row kernel turns them into softmax / CE
      launch_reduce_mask(st, m->partial, a.splits, Bp * N, b, nullptr, const_cast<float*>(m->prob), Bp, N, B);
      hipLaunchKernelGGL(k_softmax_ce_rows, dim3(Bp), dim3(kThreads), 0, st, const_cast<float*>(m->prob), labels, ce, N, B);
      break;
    }
    launch_reduce_mask(st, m->partial, a.splits, Bp * N, b, nullptr, const_cast<float*>(m->h[l + 1]), Bp, N, B,
                       const_cast<float*>(m->mask[l]));
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// Backward pass for the deltas (needs sd = sample weight / B from the caller):
// delta[L-1] = sd * (prob - onehot(y)); delta[l-1] = mask[l-1] * (delta[l] W_l).
int bhg_mlp_backward(const bhg_mlp* m, const int64_t* labels, void* stream) {
  if (int rc = check_head_problem(m)) return rc;
  BHG_REQUIRE(labels, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = m->L, Bp = m->Bp, B = m->B;
  const int C = m->dims[L];
  hipLaunchKernelGGL(k_delta_top, dim3((Bp * C + 255) / 256), dim3(256), 0, st, m->prob, m->sd, labels,
                     const_cast<float*>(m->delta[L - 1]), Bp, C, B);
  for (int l = L - 1; l >= 1; --l) {
    const int K = m->dims[l + 1], N = m->dims[l];
    if (l == L - 1 && narrow_head(m)) {
      const int blocks = (Bp * (N / 4) + 255) / 256;
      hipLaunchKernelGGL(k_head_backward, dim3(blocks), dim3(256), 0, st, (const float*)nullptr, m->delta[l], m->W[l],
                         (const float*)nullptr, m->mask[l - 1], const_cast<float*>(m->delta[l - 1]), N, K, B, Bp);
      continue;
    }
    GemmArgs a{};
    a.pr[0] = {m->delta[l], m->W[l], K, N};
    a.pairs = 1;
    a.M = Bp; a.N = N; a.K = K;
    const int tn = skinny_tile_n();
    a.splits = pick_splits((N + tn - 1) / tn, K, 1);
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_RC>(a, tn, st);
    launch_reduce_mask(st, m->partial, a.splits, Bp * N, nullptr, m->mask[l - 1], const_cast<float*>(m->delta[l - 1]), Bp,
                       N, B);
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- round 5: the once-per-step passes on the chain's own packed form -------------------------------------------------------------
// bhg_mlp_forward / bhg_mlp_backward run every hidden layer as a split-K GEMM + a reduce launch (5 pairs, 100 us of a 1.4 ms step at
// cfg 2).  The solvers' chain needs the weights packed anyway (k_pack, once per solve): packed FIRST, the hidden layers behind the
// first run as ONE k_wskpf / k_wskpc launch each, and their epilogues leave h_l / delta_l packed for the Gram products and the chain —
// so k_pack no longer reads them.  The first layer keeps its split-K form (its weight is not an operand of the chain: packing it
// would cost more than it saves); its reduce launch stores the packed copy of h_1.
int bhg_mlp_supports_packed_prepare(const bhg_mlp* m) {
  if (!bhg_mlp_supports_native_prepare(m) || !bhg_mlp_supports_fused_solve(m) || dbg(DBG_packed_prepare, 1) == 0) return 0;
  HoistPlan hp;
  hoist_plan(m, &hp);
  return hp.ok && dbg(DBG_packed_chain, 1) != 0 ? 1 : 0;
}

int bhg_mlp_forward_packed(const bhg_mlp* m, const void* const* bias, const int64_t* labels, float* ce, void* fws, size_t fws_bytes,
                           void* stream) {
  if (int rc = check_head_problem(m)) return rc;
  BHG_REQUIRE(narrow_head(m), "the packed once-per-step passes need a narrow classifier head (<= 256 classes, feature width % 4 == 0)");
  BHG_REQUIRE(bias && labels && ce && fws, "NULL argument");
  BHG_REQUIRE(bhg_mlp_supports_packed_prepare(m), "this network does not take the packed form (bhg_mlp_supports_packed_prepare)");
  BHG_REQUIRE(fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = m->L, Bp = m->Bp, B = m->B;
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  if (int rc = pack_operands(m, w, st, true)) return rc;
  {   // first layer: split-K GEMM + reduce (bias, ReLU mask, h_1 row-major and packed)
    const int K = m->dims[0], N = m->dims[1];
    GemmArgs a{};
    a.pr[0] = {m->h[0], m->W[0], K, K};
    a.pairs = 1;
    a.M = Bp; a.N = N; a.K = K;
    const int tn = skinny_tile_n();
    a.splits = pick_splits((N + tn - 1) / tn, K, 1);
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_KC>(a, tn, st);
    launch_reduce_mask(st, m->partial, a.splits, Bp * N, static_cast<const float*>(bias[0]), nullptr, const_cast<float*>(m->h[1]), Bp, N, B,
                       const_cast<float*>(m->mask[0]), w.hpk[1]);
  }
  for (int l = 1; l + 1 < L; ++l) {   // hidden layers behind the first: one launch on packed operands
    WskpfArgs f{};
    WskpProb& q = f.a;
    q.Ap = w.hpk[l]; q.Bq = w.Wf[l]; q.RA = Bp; q.RB = m->dims[l + 1]; q.K = m->dims[l]; q.B = B; q.nsplit = 1;
    q.bias = static_cast<const float*>(bias[l]);
    q.out = const_cast<float*>(m->h[l + 1]);
    q.outp = l + 2 < L ? w.hpk[l + 1] : nullptr;   // (the head reads h_{L-1} row-major)
    f.relu_mask = const_cast<float*>(m->mask[l]);
    wskp_tiling(&q, false);
    const int blk = q.nfull * (q.RB / 32) + q.nstrip;
    hipLaunchKernelGGL(k_wskpf<2>, dim3(blk), dim3(64 * kWskpWaves), 0, st, f);
  }
  launch_head_forward(st, Bp, nullptr, m->h[L - 1], m->W[L - 1] /* unused: no Rh */, m->W[L - 1], static_cast<const float*>(bias[L - 1]),
                      nullptr, nullptr, const_cast<float*>(m->prob), m->dims[L - 1], m->dims[L], B, HEAD_LOGITS, labels, ce, nullptr,
                      nullptr, nullptr);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

int bhg_mlp_backward_packed(const bhg_mlp* m, const int64_t* labels, void* fws, size_t fws_bytes, void* stream) {
  if (int rc = check_head_problem(m)) return rc;
  BHG_REQUIRE(narrow_head(m), "the packed once-per-step passes need a narrow classifier head (<= 256 classes, feature width % 4 == 0)");
  BHG_REQUIRE(labels && fws, "NULL argument");
  BHG_REQUIRE(bhg_mlp_supports_packed_prepare(m), "this network does not take the packed form (bhg_mlp_supports_packed_prepare)");
  BHG_REQUIRE(fws_bytes >= bhg_mlp_fused_ws_bytes(m), "fused workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = m->L, Bp = m->Bp, B = m->B;
  const int C = m->dims[L];
  FusedWs w;
  carve_fused_ws(m, fws, &w);
  hipLaunchKernelGGL(k_delta_top, dim3((Bp * C + 255) / 256), dim3(256), 0, st, m->prob, m->sd, labels,
                     const_cast<float*>(m->delta[L - 1]), Bp, C, B);
  {   // through the head: delta_{L-2}, row-major and packed
    const int l = L - 1, K = m->dims[l + 1], N = m->dims[l];
    const int blocks = (Bp * (N / 4) + 255) / 256;
    hipLaunchKernelGGL(k_head_backward, dim3(blocks), dim3(256), 0, st, (const float*)nullptr, m->delta[l], m->W[l],
                       (const float*)nullptr, m->mask[l - 1], const_cast<float*>(m->delta[l - 1]), N, K, B, Bp, w.dpk[l - 1]);
  }
  for (int l = L - 2; l >= 1; --l) {   // delta_{l-1} = mask_{l-1} * (delta_l W_l): one launch on packed operands
    WskpBuilder wb;
    WskpProb q{};
    q.Ap = w.dpk[l]; q.Bq = w.Wb[l]; q.RA = Bp; q.RB = m->dims[l]; q.K = m->dims[l + 1]; q.B = B; q.nsplit = 1;
    q.mask = m->mask[l - 1]; q.out = const_cast<float*>(m->delta[l - 1]);
    q.outp = l >= 2 ? w.dpk[l - 1] : nullptr;
    wb.add(q);
    wb.launch(st);
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// The step's batch into the padded buffers the passes read: h_0[:B] = x (fp32 [B][d_0]), labels[:B] = y — ONE launch instead of two
// device-to-device copies (8.5 + 1.7 us through the copy engine's blit path for 1.2 MB + 800 B at cfg 2; round 5).
__global__ __launch_bounds__(256) void k_stage_batch(const float* __restrict__ x, const int64_t* __restrict__ y, float* __restrict__ h0,
                                                     int64_t* __restrict__ labels, int64_t n4, int B) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    reinterpret_cast<float4*>(h0)[i] = reinterpret_cast<const float4*>(x)[i];
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b < B; b += 256) labels[b] = y[b];
}
int bhg_mlp_stage_batch(const bhg_mlp* m, const float* x, const int64_t* y, int64_t* labels, void* stream) {
  if (int rc = check_mlp(m)) return rc;
  BHG_REQUIRE(x && y && labels && m->h[0], "NULL argument");
  BHG_REQUIRE((m->dims[0] & 3) == 0 && ((uintptr_t)x & 15) == 0, "the input batch must be 16-byte aligned with a width that is a multiple of 4");
  const int64_t n4 = (int64_t)m->B * m->dims[0] / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_stage_batch, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, const_cast<float*>(m->h[0]), labels, n4, m->B);
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// Mixed-derivative coefficient: coeff[b] = (prob[b] - onehot(y_b)) . Rz_b(direction) / B — one R-forward.
int bhg_mlp_mixed_coeff(const bhg_mlp* m, const void* const* dir, const int64_t* labels, float* coeff, void* stream) {
  if (int rc = check_head_problem(m)) return rc;
  BHG_REQUIRE(dir && labels && coeff, "NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = m->L, Bp = m->Bp, B = m->B;
  for (int l = 0; l < L; ++l) {
    const int K = m->dims[l], N = m->dims[l + 1];
    const float* V = static_cast<const float*>(dir[2 * l]);
    const float* c = static_cast<const float*>(dir[2 * l + 1]);
    if (l == L - 1 && narrow_head(m)) {
      launch_head_forward(st, Bp, l > 0 ? (const float*)m->Rh[l - 1] : nullptr, m->h[l], m->W[l], V, c, m->prob, nullptr,
                          nullptr, K, N, B, HEAD_COEFF, labels, coeff, nullptr, nullptr, nullptr);
      break;
    }
    GemmArgs a{};
    a.pr[0] = {m->h[l], V, K, K};
    a.pairs = 1;
    if (l > 0) { a.pr[1] = {m->Rh[l - 1], m->W[l], K, K}; a.pairs = 2; }
    a.M = Bp; a.N = N; a.K = K;
    const int tn = skinny_tile_n();
    a.splits = pick_splits((N + tn - 1) / tn, K, a.pairs);
    a.out = m->partial; a.ldo = N; a.out_rows = Bp;
    launch_gemm<LAYOUT_KC, LAYOUT_KC>(a, tn, st);
    if (l == L - 1) {   // a wide output layer: Rz into Rd_L's buffer (free here), then the row kernel
      launch_reduce_mask(st, m->partial, a.splits, Bp * N, c, nullptr, m->Rd[l], Bp, N, B);
      hipLaunchKernelGGL(k_coeff_rows, dim3(Bp), dim3(kThreads), 0, st, m->prob, (const float*)m->Rd[l], labels, coeff, N, B);
      break;
    }
    launch_reduce_mask(st, m->partial, a.splits, Bp * N, c, m->mask[l], m->Rh[l], Bp, N, B);
  }
  BHG_HIP_CHECK(hipGetLastError());
  return BHG_OK;
}

// ---- the plan, described without running it (round 6; host only: no launch, no device access) -----------------------------------------
// Which form bhg_mlp_cg_solve (algo 0) / bhg_mlp_neumann_solve (algo 1) takes for this descriptor is decided by host logic alone —
// hoist_plan (shapes, cost model), cg_ctx_init (projection level, the linear first product, the head launch with the recurrences) —
// before the first launch.  This entry point evaluates exactly that logic on the shapes and prints the decision, so that the map
// "shape -> form" has a unit test of its own that runs on a CPU-only box (tests/test_plan_selection.py); the GPU suite ties the
// description to the launch counters (bhg_mlp_hoist_launches / _proj_iterations / _lin_launches) on the same shapes.
int bhg_mlp_plan_describe(const bhg_mlp* m, int algo, int keep_solution, char* buf, size_t buf_bytes) {
  if (int rc = check_mlp(m)) return rc;
  BHG_REQUIRE(buf && buf_bytes >= 64, "output buffer too small");
  BHG_REQUIRE(algo == 0 || algo == 1, "algo: 0 = cg, 1 = neumann");
  const int L = m->L;
  const bool fused = bhg_mlp_supports_fused_solve(m) != 0;
  HoistPlan hp;
  hoist_plan(m, &hp);
  const char* form = "unfused";
  int hoist = 0, proj_level = 0, lin = 0, lin_head = 0, upd_first = 0;
  const char* closing = "k_outer_all";
  if (fused) {
    // the flat layout of [W_1, b_1, ...] (what the callers pass as `starts`), fake state pointers: nothing is dereferenced
    int64_t numel[2 * BHG_MLP_MAX_LAYERS], starts[2 * BHG_MLP_MAX_LAYERS];
    for (int l = 0; l < L; ++l) { numel[2 * l] = (int64_t)m->dims[l] * m->dims[l + 1]; numel[2 * l + 1] = m->dims[l + 1]; }
    const int64_t nch = bhg_layout_num_chunks(numel, 2 * L);
    std::vector<bhg_chunk> chunks((size_t)(nch > 0 ? nch : 1));
    if (int rc = bhg_layout_build(numel, 2 * L, starts, chunks.data())) return rc;
    float* const fake = reinterpret_cast<float*>((uintptr_t)1 << 30);
    if (algo == 0) {
      static CgCtx c;   // (large: off the stack)
      cg_ctx_init(&c, m, keep_solution ? fake : nullptr, fake, fake, starts, nullptr, (int)nch, 2, 1.f, 0.f, fake, fake, false);
      hoist = c.hoist ? 1 : 0; proj_level = c.proj_level; lin = c.lin ? 1 : 0; lin_head = c.lin_head ? 1 : 0;
      upd_first = (c.lin && L > 4) ? 1 : 0;
      form = !c.lazy ? "classic-eager" : (!c.hoist ? "classic" : (c.proj_level == 0 ? "hoisted" : (c.proj_level == 1 ? "projected-keep-state" :
             (c.lin ? (c.lin_head ? "six-launch (k_wskpl .. k_headu .. k_graw)" : "six-launch-class (k_wskpl first, recurrences beside the chain)") :
              "fully-projected (k_pstep launch)"))));
      if (c.proj_level >= 1) closing = graw_batch_ok(m->Bp) ? (m->Bp == 128 ? "k_graw" : "k_grawk") : "k_hoist+k_proj_update";
    } else {
      const bool want_proj = !keep_solution && proj_mode() != 0 && hoist_mode() != 0;
      const bool proj = want_proj && hp.ok && hp.proj_ok && use_head(m);
      hoist = (hp.ok && (hoist_mode() == 2 || want_proj)) ? 1 : 0;
      proj_level = proj ? 1 : 0;
      form = proj ? "projected-neumann (update inside k_graw)" : (hoist ? "hoisted" : "classic");
      if (proj) closing = graw_batch_ok(m->Bp) ? (m->Bp == 128 ? "k_graw" : "k_grawk") : "k_hoist+k_proj_update";
    }
  }
  size_t gram = 0;
  if (hp.ok && hp.proj_ok)
    for (int l = 0; l + 1 < L; ++l) gram += (size_t)m->Bp * m->Bp * (l >= 1 ? (2 + 2 * kGramSplitMax + 6) : 2);
  const int n = snprintf(buf, buf_bytes,
                         "algo=%s keep_solution=%d fused=%d form=\"%s\" hoist=%d proj_level=%d lin=%d lin_head=%d upd_first=%d closing=%s "
                         "plan_ok=%d proj_ok=%d lin_ok=%d hoist_products=%d hoist_wgs=%d gram_floats=%zu fused_ws_bytes=%zu narrow_head=%d packed_prepare=%d "
                         "global_form=%s fx_slab_bytes=%zu fx_ws_bytes_world8=%zu",
                         algo == 0 ? "cg" : "neumann", keep_solution ? 1 : 0, fused ? 1 : 0, form, hoist, proj_level, lin, lin_head, upd_first, closing,
                         hp.ok ? 1 : 0, hp.proj_ok ? 1 : 0, hp.lin_ok ? 1 : 0, hp.n, hp.ok ? hp.blk0[hp.n] : 0, gram,
                         fused ? bhg_mlp_fused_ws_bytes(m) : (size_t)0, narrow_head(m) ? 1 : 0, bhg_mlp_supports_packed_prepare(m),
                         // Config(type="cg_global"): the form of the global-batch solve (round 6: factor exchange whenever the projected plan is
                         // taken and the caller does not ask for x; else the one-pass form; without a fused solver the sharded form)
                         (algo == 0 && !keep_solution && bhg_mlp_fx_supported(m, 1)) ? "factor-exchange" : (fused ? "one-pass" : "sharded"),
                         sizeof(float) * bhg_mlp_fx_slab_floats(m), bhg_mlp_fx_ws_bytes(m, 8));
  BHG_REQUIRE(n > 0 && (size_t)n < buf_bytes, "output buffer too small");
  return BHG_OK;
}

#ifdef BHG_STAMPS
static unsigned long long* g_stamps_dev = nullptr;
int bhg_debug_stamps_enable(int on) {
  const size_t bytes = sizeof(unsigned long long) * bhg::kStampKernels * bhg::kStampBlocks * bhg::kStampSlots;
  if (on && !g_stamps_dev) {
    BHG_HIP_CHECK(hipMalloc(&g_stamps_dev, bytes));
  }
  if (g_stamps_dev) BHG_HIP_CHECK(hipMemset(g_stamps_dev, 0, bytes));
  unsigned long long* p = on ? g_stamps_dev : nullptr;
  BHG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(bhg::d_stamps), &p, sizeof(p)));
  return BHG_OK;
}
int bhg_debug_stamps_read(unsigned long long* host, size_t count) {
  BHG_REQUIRE(g_stamps_dev && host, "stamps are not enabled");
  const size_t total = (size_t)bhg::kStampKernels * bhg::kStampBlocks * bhg::kStampSlots;
  BHG_HIP_CHECK(hipDeviceSynchronize());
  BHG_HIP_CHECK(hipMemcpy(host, g_stamps_dev, sizeof(unsigned long long) * (count < total ? count : total), hipMemcpyDeviceToHost));
  return BHG_OK;
}
#endif

}  // extern "C"

// ---- code placement of the K-loop kernels (round 6) ------------------------------------------------------------------------------------------
// The device code of this translation unit is laid out as: every NON-template kernel in definition order, then the template kernels in the
// order of their first use — among them the four kernels of a CG iteration, the largest of which (k_wskpl: 62 KB of instructions, about the
// size of the instruction cache two CUs share) is sensitive to WHERE it starts: the same instruction stream 512 bytes further on ran 0.43 us
// slower per launch (same-box A/B, profiles/r06_code_placement_of_the_k_loop_kernels.txt: 56.8-57.0 vs 57.6-57.9 us per iteration, after an
// unrelated kernel ahead of it had grown by 600 bytes).  So the template kernels are ANCHORED: this kernel is the last non-template one, it
// starts on a 16 KB boundary whatever precedes it, and its size — BHG_LAYOUT_PAD units of 256 bytes of s_nop — was chosen by two sweeps (the
// sixteen residues modulo 4 KB, then the better ones modulo 16 KB: 56.4 us per iteration at the best, 57.7 at the worst; same file).  It is
// never launched.  tests/test_code_placement.py pins the four kernels' offsets from it.
#ifndef BHG_LAYOUT_PAD
#define BHG_LAYOUT_PAD 8
#endif
namespace bhg {
namespace {
#define BHG_STR2(x) #x
#define BHG_STR(x) BHG_STR2(x)
__global__ __attribute__((aligned(16384))) void k_layout_anchor() {
#if BHG_LAYOUT_PAD > 0
  asm volatile(".rept " BHG_STR(BHG_LAYOUT_PAD) " * 64 - 8\n s_nop 0\n .endr");
#endif
}
}  // namespace
}  // namespace bhg
