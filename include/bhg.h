/*
 * bhg.h — C ABI of the MI355X-native hypergradient backend (libbhg.so).
 *
 * This is the drop-in boundary for the implicit-differentiation hot path of
 * leopard-ai/betty (`betty/hypergradient/{cg,neumann,darts,sama}.py`).  The reference
 * has no FFI of its own (it is pure Python on torch); the entry points below are
 * what a binding for this path has to call: each one replaces a run of per-tensor
 * ATen launches in the reference and cites the lines it replaces.
 *
 * Conventions
 *   - Every pointer named *_dev / x / r / p / v / ws is a DEVICE pointer owned by
 *     the caller (a PyTorch-ROCm tensor).  The library never allocates or frees
 *     caller memory; all scratch lives in the caller-provided workspace `ws`
 *     (size from bhg_workspace_bytes()).
 *   - "tensor tables" (`const void* const* ptrs`, `T` entries) are HOST arrays of
 *     device pointers to T separately allocated, contiguous fp32 tensors (what
 *     torch.autograd.grad hands back each iteration).  They are consumed before
 *     the call returns (copied into kernel arguments / the workspace), so the
 *     caller may reuse the host array immediately.
 *   - A `bhg_layout` describes how those T tensors map onto one flat fp32 vector:
 *     tensor t occupies flat[start_t, start_t + numel_t) with start_t a multiple of
 *     64 elements (256 B).  Padding elements are zero and are never written.
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the null stream).  No call synchronises the host.
 *   - Return value: 0 = ok, negative = error (message via bhg_last_error()).
 *     No C++ exception crosses this boundary.
 *   - All arithmetic is fp32 with fp64 accumulation of dot products.
 */
#ifndef BHG_H_
#define BHG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BHG_VERSION 1
#define BHG_CHUNK_ELEMS 4096 /* elements of one tensor handled by one work item   */
#define BHG_FLAT_ALIGN 64    /* flat start of every tensor is a multiple of this  */

/* error codes */
#define BHG_OK 0
#define BHG_ERR_ARG (-1)     /* bad argument (NULL pointer, negative size, ...)   */
#define BHG_ERR_HIP (-2)     /* a HIP runtime call failed                         */
#define BHG_ERR_WS (-3)      /* workspace too small                               */
#define BHG_ERR_CAPACITY (-4)/* requested kernel variant cannot hold this N       */

/* One unit of multi-tensor work: `len` (<= BHG_CHUNK_ELEMS) elements of tensor
 * `tensor`, starting `src_off` elements into that tensor and `flat_off` elements
 * into the flat vector.  Built once per parameter list by bhg_layout_build(). */
typedef struct bhg_chunk {
  int64_t flat_off;
  int64_t src_off;
  int32_t tensor;
  int32_t len;
} bhg_chunk;

int bhg_version(void);
const char* bhg_last_error(void);

/* ---- layout helpers (host only, no GPU needed) -------------------------------
 * Replace the implicit layout of betty/utils.py:117-118 (`to_vec` = cat of
 * reshaped tensors): same element order, plus 256-B alignment padding. */
int64_t bhg_layout_flat_size(const int64_t* numel, int T);
int64_t bhg_layout_num_chunks(const int64_t* numel, int T);
/* Fills starts[T] (flat start of every tensor) and chunks[num_chunks] (host memory;
 * the caller uploads `chunks` to the device once). */
int bhg_layout_build(const int64_t* numel, int T, int64_t* starts, bhg_chunk* chunks);
/* Bytes of device workspace any call below may need for a T-tensor layout. */
size_t bhg_workspace_bytes(int T);

/* ---- multi-tensor <-> flat ----------------------------------------------------
 * bhg_flatten: flat[...] = scale * tensors   (betty/utils.py:117-118 to_vec)      */
int bhg_flatten(const void* const* src, int T, const bhg_chunk* chunks_dev, int n_chunks,
                float* flat, float scale, void* ws, void* stream);
/* bhg_scatter: tensors = scale * flat[...]   (inverse of to_vec; used to hand
 * results back as a list aligned with the parameters)                             */
int bhg_scatter(const float* flat, void* const* dst, int T, const bhg_chunk* chunks_dev,
                int n_chunks, float scale, void* ws, void* stream);

/* ---- Neumann series: betty/hypergradient/neumann.py:59-66 ---------------------
 * init:  v = p = vector                                    (neumann.py:60)
 * step:  v <- v - alpha*Hv ; p <- p + v                    (neumann.py:62-64)
 *        on the LAST step pass out_scale = -alpha to fold `alpha * p` (66) and the
 *        negation of neumann.py:45/54 into the same pass; otherwise out_scale = 0.
 *        hvp_shift: the operator applied is (HVP + hvp_shift * I); lets a producer leave a diagonal
 *        part of the Hessian (a ridge / proximal term) out of its output (0 for plain autograd HVPs). */
int bhg_neumann_init(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks,
                     float* v, float* p, void* ws, void* stream);
int bhg_neumann_step(const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks,
                     float* v, float* p, float alpha, float out_scale, float hvp_shift, void* ws,
                     void* stream);

/* ---- Conjugate gradient: betty/hypergradient/cg.py:34-56 ----------------------
 * init:  x = 0 ; r = p = vector ; rr = r.r                 (cg.py:34-36,45)
 * step (iteration `iter` = 0,1,..):                        (cg.py:39-55)
 *        den = (cg_alpha*Hp).p ; a = rr/den ; x += a*p ; r -= a*Hp (UN-scaled Hp:
 *        reference quirk) ; b = r'.r'/rr ; p = r' + b*p ; rr = r'.r'
 *        on the LAST step pass out_scale = -cg_alpha to fold cg.py:56 and the
 *        negation of cg.py:59/68 into x; otherwise out_scale = 0.
 * variant: BHG_CG_AUTO picks the register-resident single-pass kernel when the
 *        vector fits on chip and the 3-kernel streaming form otherwise.
 * After the call ws holds the iteration's scalars (see bhg_cg_read_scalars).      */
#define BHG_CG_AUTO 0
#define BHG_CG_STREAM 1     /* 3 streaming kernels, 40*N bytes per iteration         */
#define BHG_CG_RESIDENT 2   /* 1 persistent kernel: 28*N bytes while the vector fits on chip (registers only up to
                               11 chunks per CU, 15 with 9 direction slices per CU parked in LDS); beyond that a
                               hybrid instance keeps 14 chunks per CU resident and streams the rest in the same
                               launch (40 bytes per streamed element), up to bhg_cg_resident_capacity_chunks() */
int bhg_cg_init(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks,
                float* x, float* r, float* p, void* ws, void* stream);
/* The same with a per-tensor write mask (round 5): bit t of keep_mask clear (t < 64) -> tensor t's slices of r and p are left
 * untouched; its share of r.r is still taken.  For a solver that reads those slices of the right-hand side from the caller's own
 * tensors and never reads the direction's (bhg_mlp_cg_solve_rhs with the mask bhg_mlp_cg_state_mask returns): at the metric
 * workload 80 MB of writes per hypergradient step that nothing ever read.  keep_mask = ~0: bhg_cg_init.                        */
int bhg_cg_init_masked(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks,
                       float* x, float* r, float* p, unsigned long long keep_mask, void* ws, void* stream);
int bhg_cg_step(const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks,
                float* x, float* r, float* p, float cg_alpha, int iter, float out_scale,
                float hvp_shift /* operator = HVP + hvp_shift*I, see bhg_neumann_step */, int variant,
                void* ws, void* stream);
/* Largest number of chunks BHG_CG_RESIDENT accepts on the current device = twice the hybrid instance's resident
 * part (at least half of the vector stays on chip); 0 when no device is visible. */
int bhg_cg_resident_capacity_chunks(void);
/* 1 when a one-time census found all workgroups of the resident kernel co-resident on the current
 * device (needed by its grid barrier), else 0; BHG_CG_AUTO falls back to BHG_CG_STREAM when 0.
 * Synchronises the device the first time it is called. */
int bhg_cg_resident_ok(void);
/* The predicate BHG_CG_AUTO uses for an n_chunks-chunk vector: capacity AND the census AND — beyond the register-only
 * size (11 chunks per CU) — a one-time probe that the device grants the LDS-assisted / hybrid instances their 144 KiB of
 * dynamic LDS per workgroup.  0 => AUTO takes the streaming kernels. */
int bhg_cg_resident_usable(int n_chunks);
/* Device address (inside ws) of 8 doubles: {rr_old, pHp, alpha, rr_new, beta, 0,0,0}
 * written by the most recent bhg_cg_step on that workspace. */
const double* bhg_cg_scalars_dev(const void* ws);
/* Device address (inside ws) of one u32 that the resident kernel sets to 1 when a grid barrier gave up
 * polling since the last bhg_cg_init on that workspace (the result is NaN-poisoned as well). */
const unsigned* bhg_cg_timeout_flag_dev(const void* ws);

/* ---- phased CG iteration for SHARDED state vectors (global-HVP mode; not in the reference: SURVEY.md §8(e)(2)) -------
 * The three kernels of BHG_CG_STREAM, one call per phase (0: dot, 1: residual, 2: direction), so the caller can
 * all-reduce(SUM) the per-block partial sums over its process group between the phases; every rank's consumer kernel
 * then sums the same all-reduced partials = the global dot products of cg.py:45-46,51.  All ranks must hold equally
 * sized shards.  bhg_cg_partials_dev(ws, which, iter): device address of bhg_cg_partials_count() doubles —
 * which = 0: after phase 0; which = 1: after phase 1; which = 2: after bhg_cg_init (r.r).                             */
int bhg_cg_phase(int phase, const void* const* hvp, int T, const bhg_chunk* chunks_dev, int n_chunks, float* x,
                 float* r, float* p, float cg_alpha, int iter, float out_scale, float hvp_shift, void* ws, void* stream);
double* bhg_cg_partials_dev(void* ws, int which, int iter);
int bhg_cg_partials_count(void);

/* ---- flat helpers ------------------------------------------------------------- */
/* flat <- scale * flat  (cg.py:56 / neumann.py:66 when iterations == 0)           */
int bhg_scale_flat(float* flat, int64_t n, float scale, void* stream);

/* ---- DARTS finite difference: betty/hypergradient/darts.py:29-38,49-50,62-63 --
 * bhg_darts_eps: eps = R / (||vec||_2 + 1e-15) computed entirely on device
 *        (darts.py:29-35 without the host .item()).  out_dev[0] = sum of squares,
 *        out_dev[1] = eps (both double); eps_f32_dev receives (float)eps.
 * bhg_axpy_multi: dst_t += (mul * *coef_dev) * src_t for all t (coef_dev may be
 *        NULL => coefficient = mul): the three in-place weight perturbations.     */
int bhg_darts_eps(const void* const* vec, int T, const bhg_chunk* chunks_dev, int n_chunks,
                  double R, double* out_dev, float* eps_f32_dev, void* ws, void* stream);
int bhg_axpy_multi(void* const* dst, const void* const* src, int T, const bhg_chunk* chunks_dev,
                   int n_chunks, const float* coef_dev, float mul, void* ws, void* stream);

/* ---- SAMA (betty/hypergradient/sama.py:25, utils.py:37-63): Adam-preconditioned direction ----------
 * out_flat = vec * scale * lr with scale built from the optimizer state (last_grad, exp_avg,
 * exp_avg_sq): one fused pass over the four tensor lists (20*N bytes) instead of ~14 ATen
 * launches per tensor.  All tensors of one call share (beta1, beta2, eps, lr) — one call per
 * optimizer param group.  The finite-difference part of SAMA is then bhg_darts_eps / bhg_axpy_multi
 * on the views of out_flat, exactly as for `darts`.                                                 */
int bhg_sama_adam_precondition(const void* const* vec, const void* const* last_grad,
                               const void* const* exp_avg, const void* const* exp_avg_sq, int T,
                               const bhg_chunk* chunks_dev, int n_chunks, float* out_flat,
                               double beta1, double beta2, double eps, double lr, void* ws, void* stream);

/* ---- debug / A-B switches (measurement and tests only; not part of the drop-in surface) ----
 * libbhg reads NO environment variable: its behaviour depends on its arguments alone.  The arms the
 * test-suite compares against each other and the same-box A/B measurements of profiles/ are selected
 * through a table of named ints, all UNSET by default (unset = the shipped behaviour).  Keys are the
 * lower-case names listed in betty_amd/csrc/bhg_common.hpp (BHG_DBG_KEYS), e.g. "mlp_proj",
 * "mlp_hoist", "packed_chain".  Process-global, not thread-safe against concurrent solves.        */
int bhg_debug_set(const char* key, int value);   /* BHG_ERR_ARG for an unknown key */
int bhg_debug_unset(const char* key);
void bhg_debug_reset(void);                       /* every key back to unset        */
int bhg_debug_key_count(void);
const char* bhg_debug_key_name(int i);            /* NULL past the end              */
/* Round 5: the table lives in the MEASUREMENT build only (libbhg_ab.so, `make -C betty_amd/csrc ab`, -DBHG_AB; same sources, same
 * ABI).  In the product libbhg.so every key is compiled to its default — the arms are not in the code object — bhg_debug_set()
 * returns BHG_ERR_ARG, bhg_debug_key_count() is 0 and bhg_is_ab_build() is 0.                                                      */
int bhg_is_ab_build(void);

/* ---- optional kernel timing (measurement only) ----------------------------------
 * When enabled, every bhg_cg_step / bhg_neumann_step launch group carries start/stop
 * HIP events attached to its first/last kernel on the launch stream (kernel begin ->
 * kernel end, the interval rocprofv3 --kernel-trace reports).  bhg_timing_read waits for
 * the recorded spans of `kind` and returns their summed duration and count.           */
#define BHG_TIMING_CG_STEP 0
#define BHG_TIMING_NEUMANN_STEP 1
#define BHG_TIMING_MLP_HVP 2      /* whole bhg_mlp_hvp call (or one fused-solver iteration's HVP chain incl. its
                                     fused recurrence epilogues): event before the first / after the last launch */
#define BHG_TIMING_MLP_CG_ITER 3  /* one whole iteration of bhg_mlp_cg_solve: HVP chain + direction update      */
int bhg_timing_enable(int on);
int bhg_timing_read(int kind, double* total_ms, int* launches);

/* ---- analytic Hessian-vector products (structured inner problems) --------------
 * Logistic regression with per-weight L2 (the inner problem of
 * examples/logistic_regression_hpo/logistic_regression_implicit.py:80-91):
 *   Hp = X^T( s .* (X p) ) + lam .* p,  s_i = sigma_i (1 - sigma_i) / n.
 * bhg_logreg_prepare computes s from (X, w) once per hypergradient step;
 * bhg_logreg_hvp applies it.  X is row-major [n, d] fp32.                          */
int bhg_logreg_prepare(const float* X, const float* w, float* s, int n, int d, void* stream);
/* fp32 elements of device scratch bhg_logreg_hvp needs in `tmp` */
size_t bhg_logreg_tmp_floats(int n, int d);
int bhg_logreg_hvp(const float* X, const float* s, const float* lam, const float* p, float* out,
                   float* tmp, int n, int d, void* stream);

/* ReLU-MLP with per-sample-weighted cross-entropy (+ ridge) — SURVEY Appendix A.3; the inner
 * problem of examples/learning_to_reweight/main.py:117-127 (BASELINE cfg 2 / the metric's 10 M
 * parameter problem).  The caller computes the direction-independent quantities once per
 * hypergradient step and describes them here; bhg_mlp_hvp then evaluates
 *   H(W_l) = Rd_l^T h_{l-1} + delta_l^T Rh_{l-1} + ridge2 * V_l ,  H(b_l) = colsum(Rd_l) + ridge2 * c_l
 * on the fp32 matrix cores.  Layer l (0-based) maps dims[l] -> dims[l+1]; all [Bp, d] buffers are
 * row-major with Bp rows (a multiple of 128, >= B); rows >= B must hold finite values and are
 * ignored / written as zero.                                                   */
#define BHG_MLP_MAX_LAYERS 32
typedef struct bhg_mlp {
  int32_t L, B, Bp;
  int32_t dims[BHG_MLP_MAX_LAYERS + 1];
  const float* W[BHG_MLP_MAX_LAYERS];     /* [dims[l+1], dims[l]] (torch nn.Linear.weight)          */
  const float* h[BHG_MLP_MAX_LAYERS];     /* input of layer l:      [Bp, dims[l]]   (h[0] = x)      */
  const float* mask[BHG_MLP_MAX_LAYERS];  /* ReLU mask of layer l:  [Bp, dims[l+1]], l < L-1 (1/0)  */
  const float* delta[BHG_MLP_MAX_LAYERS]; /* dL/d(pre-activation l): [Bp, dims[l+1]]                */
  const float* prob;                      /* softmax output [Bp, dims[L]]                           */
  const float* sd;                        /* per-sample weight / B  [Bp]                            */
  float* Rh[BHG_MLP_MAX_LAYERS];          /* scratch [Bp, dims[l+1]], l < L-1                       */
  float* Rd[BHG_MLP_MAX_LAYERS];          /* scratch [Bp, dims[l+1]]                                */
  float* partial;                         /* split-K scratch, >= bhg_mlp_partial_floats() floats    */
  size_t partial_floats;
  float ridge2;                           /* 2 * ridge                                              */
  int32_t prepacked;                      /* round 5: 1 = bhg_mlp_forward_packed + bhg_mlp_backward_packed filled the packed operands
                                             inside the fused workspace for THESE weights and THIS batch (the solvers then skip their
                                             packing launch); 0 otherwise                              */
} bhg_mlp;
size_t bhg_mlp_partial_floats(const bhg_mlp* m);
/* dir / out: host arrays of 2L device pointers [V_0, c_0, V_1, c_1, ...] / [H(W_0), H(b_0), ...]
 * (contiguous fp32, 16-byte aligned; out tensors are fully overwritten).                         */
int bhg_mlp_hvp(const bhg_mlp* m, const void* const* dir, void* const* out, void* stream);
/* The same with the caller's choice of the skinny GEMMs' form (values of BHG_MLP_WSK: 0 split-K launches + reduce, 1 / 2 /
 * 3 in-workgroup split-K everywhere / for short reductions / plus the LDS-staged form for the long R-backward).  The
 * un-fused Neumann loop passes 3 — what the fused Neumann solver uses — so the two arms stay bitwise comparable.     */
int bhg_mlp_hvp_mode(const bhg_mlp* m, const void* const* dir, void* const* out, int gemm_mode, void* stream);

/* Once-per-step passes on the same descriptor (any head since round 6: up to 256 classes with dims[L-1] % 4 == 0 on the head kernels,
 * wider ones as one more split-K product + a row kernel; bhg_mlp_supports_native_prepare tells).  They write the
 * direction-independent buffers the HVP reads:
 *   bhg_mlp_forward   h[1..L-1], mask[], prob, and ce[b] = -log softmax(z_b)[y_b]   (h[0] = padded input, bias = L ptrs)
 *   bhg_mlp_backward  delta[]  from sd (= per-sample weight / B, written by the caller between the two calls)
 *   bhg_mlp_mixed_coeff  coeff[b] = (prob_b - onehot(y_b)) . Rz_b(dir) / B  (the mixed second derivative's
 *                     coefficient w.r.t. the sample weights; cg.py:58-68 for this structure)            */
int bhg_mlp_supports_native_prepare(const bhg_mlp* m);
int bhg_mlp_forward(const bhg_mlp* m, const void* const* bias, const int64_t* labels, float* ce, void* stream);
int bhg_mlp_backward(const bhg_mlp* m, const int64_t* labels, void* stream);
int bhg_mlp_mixed_coeff(const bhg_mlp* m, const void* const* dir, const int64_t* labels, float* coeff, void* stream);

/* ---- fused solvers ("one pass"): K iterations of HVP + recurrence for the MLP structure above with NO N-sized
 * H*direction vector.  The kernels that produce the weight-shaped HVP outputs apply the recurrence to the matching
 * slices of the flat state vectors while the tile is on chip (r <- r - a*Hp, x <- x + a*p and the partial r'.r' for
 * CG; v' <- v - a*Hv, p <- p + v' for Neumann).  For CG the step length a = rr / (cg_alpha * p.Hp) is known BEFORE
 * those kernels run, from batch-sized factors of the R-chain:
 *     p.Hp = sum_b Rz_b.Rd_L,b + 2 sum_{l>=1} <delta_l V_l, Rh_{l-1}> + hvp_shift * p.p .
 * Replaces cg.py:38-56 / neumann.py:61-66 (the whole K loop) for this structure.
 *   starts[2L]: element offsets of [W_0, b_0, W_1, b_1, ...] inside the flat vectors (bhg_layout_build's `starts`);
 *   the direction handed to the HVP chain is the matching slice of p (CG) / v (Neumann).
 *   fws: device scratch of bhg_mlp_fused_ws_bytes(m) bytes.
 * bhg_mlp_cg_solve: call bhg_cg_init(vec, ..., x, r, p, ws, stream) first (x = 0, r = p = vec, r.r partials in ws);
 *   on return x = -cg_alpha * x_K (cg.py:56 and the negation of cg.py:59/68 folded into the last iteration).
 *   x may be NULL in both calls: the N-sized solution is then neither zeroed, read nor written — for this structure the
 *   hypergradient only needs Rz(x), which the solver accumulates from batch-sized factors (bhg_mlp_cg_mixed_coeff).
 * bhg_mlp_neumann_solve: call bhg_neumann_init(vec, ..., v0, p, ...) first; v0 / v1 ping-pong as the direction (the
 *   R-backward GEMMs of an HVP still read v while its epilogues write v'); on return p = -alpha * p_K.              */
/* Round 3 — how bhg_mlp_cg_solve runs by default (>= 3 layers, every hidden width and the input width % 32 == 0; otherwise the
 * round-2 "classic chain" with the recurrence fused into the weight-shaped outputs):
 *   - the products of the batch with the direction (h_l V_l^T, delta_l V_l) are HOISTED out of the R-chain (BHG_MLP_HOIST) and
 *   - PROJECTED (BHG_MLP_PROJ): since H p's weight-shaped outputs are outer products of batch-sized factors, those products of
 *     the residual, and — without a solution vector (x == NULL) — also r.r, p.p, p.Hp, obey batch-sized recurrences through
 *     B x B Gram matrices.  After the first iteration no N-sized state vector is read or written: r and p then only carry the
 *     biases' and the head weight's slices; on return they do NOT hold the final residual / direction of the big slices.
 *   With x != NULL the N-sized r, p, x are maintained as cg.py:49-53 has them (projection level 1).
 * Same results to fp32 rounding (tests/test_cfg2_goldens.py holds every arm to the reference's CPU run at full size).        */
int bhg_mlp_supports_fused_solve(const bhg_mlp* m);
size_t bhg_mlp_fused_ws_bytes(const bhg_mlp* m);
int bhg_mlp_cg_solve(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts,
                     const bhg_chunk* chunks_dev, int n_chunks, int K, float cg_alpha, float hvp_shift, void* ws,
                     void* fws, size_t fws_bytes, void* stream);
int bhg_mlp_neumann_solve(const bhg_mlp* m, float* v0, float* v1, float* p, const int64_t* starts, int K,
                          float alpha, float hvp_shift, void* fws, size_t fws_bytes, int* projected_out, void* stream);
/* Global-batch CG (extension; not in the reference, whose DDP mode replicates the whole solve,
 * betty/problems/problem.py:253-262): ONE inner problem whose batch is spread over `world` ranks, one process per GPU; `m`
 * describes THIS rank's share of the batch.  The oracle is cg.py:8-70 run in one process on the concatenated batch.
 * The flat x, r, p are REPLICATED (bit-identical on every rank).  Because every rank holds the same r, p and step length,
 *     r - alpha (mean_g H_g) p = mean_g (r - alpha H_g p),
 * so bhg_mlp_cg_solve's one-pass iteration runs on each rank's batch unchanged, cut at the two points where the ranks talk:
 *   phase BHG_CG_GLOBAL_CHAIN   (k > 0: beta and the direction's small slices;) the R-chain; php[0] = this rank's p.H_data p
 *       -> caller: all-reduce(SUM) of php[0]                                              (8 bytes)
 *   phase BHG_CG_GLOBAL_UPDATE  alpha = r.r / (cg_alpha (php[0] / world + shift p.p)); the weight-shaped outputs with
 *                               r <- r - alpha H_g p, x += alpha p in their epilogues      (x == NULL: as bhg_mlp_cg_solve)
 *       -> caller, unless k == K - 1: all-reduce(MEAN) of r                               (4*N bytes)
 *   phase BHG_CG_GLOBAL_DOTS    r'.r', r'.p, p.p over the exchanged residual (one 8*N-byte pass) for the next beta
 * Two collectives per iteration, one of them 8 bytes; the last iteration has the 8-byte one only.  Before k = 0:
 * bhg_cg_init on the MEAN over the ranks of the local right-hand sides.  All on one stream, the collectives ordered with it.
 * php: one double of device memory owned by the caller.  world == 1 needs no collective.                                */
#define BHG_CG_GLOBAL_CHAIN 0
#define BHG_CG_GLOBAL_UPDATE 1
#define BHG_CG_GLOBAL_DOTS 2
int bhg_mlp_cg_global_phase(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts, const bhg_chunk* chunks_dev,
                            int n_chunks, int k, int K, int phase, int world, double* php, float cg_alpha, float hvp_shift,
                            void* ws, void* fws, size_t fws_bytes, void* stream);
/* Global-batch CG, FACTOR-EXCHANGE form (round 6; extension like bhg_mlp_cg_global_phase, same oracle: cg.py:8-70 in one process on the
 * concatenated batch; north_star: "DDP hypergradient all-reduce partitioned across the 8 GPUs ... overlapped with the next CG matvec").
 * The fully projected solver on SAMPLE-PARTITIONED data: rank g keeps the products of the Krylov vectors with ITS batch
 * (Gf_l = h_l U_l^T, Gb_l = delta_l U_l: batch-sized), and because every weight-shaped output of H_g' v is an outer product of
 * batch-sized factors of rank g', their recurrences need the other ranks' FACTORS, not their N-sized outputs:
 *     Gf_l^(g)(H v) = 1/G [S_l | T_l] [Rd_l ; delta_l]       Gb_l^(g)(H v) = 1/G [E_l | D_l] [h_l ; Rh_{l-1}]
 * with rectangular Gram blocks [Bp x world*Bp] (this rank's rows against every rank's samples).  Nothing N-sized is exchanged and,
 * after the projections of the right-hand side in iteration 0, nothing N-sized is read.  The caller drives, all on ONE stream:
 *     phase BEGIN (k = 0);                 all-gather const_all   [world][bhg_mlp_fx_const_floats]   h_l, delta_l: once per solve
 *     for k in 0 .. K-1:
 *         phase CHAIN (k);                 all-gather slab_all    [world][bhg_mlp_fx_slab_floats]    Rd_l, Rh_l of this iteration
 *         phase GRAM (k);                  all-gather scal_all    [world][bhg_mlp_fx_scal_doubles]   fp64 partials of r.raw, p.raw, raw.raw
 *     phase END (k = K-1)
 * Every phase writes THIS rank's slot (index `rank`) of the buffer gathered after it; the gathered partials are summed by every rank
 * in rank order, so step lengths are bit-identical on all ranks whatever the collective's own reduction order.  rhs: 2L device
 * pointers [W_0-shaped, b_0-shaped, ...] of the right-hand side, REPLICATED (the mean over the ranks of the local vectors), read in
 * CHAIN (0) only.  All ranks must hold the same B and Bp.  On return Rz(x) sits in `fws` where bhg_mlp_cg_mixed_coeff reads it.
 * world == 1 needs no collective and is the fully projected solver with its Gram products in launches of their own.
 * fws: bhg_mlp_fused_ws_bytes(m);  xws: bhg_mlp_fx_ws_bytes(m, world) (zero-filled by the caller once).                              */
#define BHG_CG_FX_BEGIN 0
#define BHG_CG_FX_CHAIN 1
#define BHG_CG_FX_GRAM 2
#define BHG_CG_FX_END 3
int bhg_mlp_fx_supported(const bhg_mlp* m, int world);
size_t bhg_mlp_fx_ws_bytes(const bhg_mlp* m, int world);
size_t bhg_mlp_fx_const_floats(const bhg_mlp* m);
size_t bhg_mlp_fx_slab_floats(const bhg_mlp* m);
size_t bhg_mlp_fx_scal_doubles(const bhg_mlp* m);
int bhg_mlp_cg_fx_phase(const bhg_mlp* m, const void* const* rhs, int k, int K, int phase, int world, int rank, float* const_all,
                        float* slab_all, double* scal_all, float cg_alpha, float hvp_shift, void* fws, size_t fws_bytes, void* xws,
                        size_t xws_bytes, void* stream);
/* The same form for the Neumann series (neumann.py:59-66 on the global batch: v' = v - alpha (H v), p = alpha sum_{k <= K} v_k) — the series
 * has NO scalars, so the ranks exchange the factor slab ONLY: one all-gather per iteration, no reduction of any kind.  Phases as above with
 *     BEGIN (k = 0) + gather of the constants;   for k in 0 .. K-1:  CHAIN (k), gather of the factor slab, GRAM (k);
 *     CHAIN (K)  — the closing half pass (recurrences, forward chain and head: Rz(v_K));   END (k = K).
 * scal_all is written (the G(raw) tiles emit their partials all the same) but never gathered nor read.  Afterwards
 * bhg_mlp_neumann_mixed_coeff(m, NULL, labels, coeff, alpha, K, 1, fws, ...) gives the mixed coefficient of -alpha sum_k v_k.            */
int bhg_mlp_neumann_fx_phase(const bhg_mlp* m, const void* const* rhs, int k, int K, int phase, int world, int rank, float* const_all,
                             float* slab_all, double* scal_all, float alpha, float hvp_shift, void* fws, size_t fws_bytes, void* xws,
                             size_t xws_bytes, void* stream);
/* bhg_mlp_cg_solve with the right-hand side's tensors named (round 5).  rhs: 2L device pointers [W_0-shaped, b_0-shaped, ...] — the
 * tensors bhg_cg_init(_masked) was given — or NULL (= bhg_mlp_cg_solve).  The fully projected solver reads the N-sized residual
 * exactly once, in iteration 0, where r = the right-hand side: with rhs it reads the MFMA layers' slices THERE, so the caller may
 * skip writing them (and the direction's, which it never reads) in bhg_cg_init_masked.
 * bhg_mlp_cg_state_mask: bit t set -> the solve reads / writes tensor t's slice of r, p (t = 2l: W_l, 2l + 1: b_l); all ones unless the
 * solve will run fully projected (has_x = 0, the plan's cost model, no A/B key against it) — then only the biases and the head
 * weight.  Evaluate it right before bhg_cg_init_masked; a non-trivial mask obliges the caller to pass rhs.                         */
unsigned long long bhg_mlp_cg_state_mask(const bhg_mlp* m, int has_x);
int bhg_mlp_cg_solve_rhs(const bhg_mlp* m, float* x, float* r, float* p, const int64_t* starts, const bhg_chunk* chunks_dev,
                         int n_chunks, int K, float cg_alpha, float hvp_shift, void* ws, void* fws, size_t fws_bytes,
                         const void* const* rhs, void* stream);
/* Mixed-derivative coefficient (see bhg_mlp_mixed_coeff) of the solution of the LAST bhg_mlp_cg_solve on `fws`,
 * without another R-forward pass: x is a linear combination of the CG directions, and the solver accumulated
 * Rz(x) = sum_k alpha_k Rz(p_k) from the Rz every iteration's head kernel computes anyway.                          */
int bhg_mlp_cg_mixed_coeff(const bhg_mlp* m, const int64_t* labels, float* coeff, float cg_alpha, void* fws,
                           size_t fws_bytes, void* stream);
/* Test / measurement hook: how many of the chain's skinny GEMMs this process launched in the in-workgroup split-K form
 * (k_gemm_wsk: a final 32 x 32 tile per workgroup, no partial slabs, no reduce launch; env BHG_MLP_WSK selects where it
 * is used: 0 nowhere, 1 wherever the shape allows, 2 short reductions only — the fused CG solver's default —, 3 the
 * same plus the LDS-staged form for the long R-backward reduction — the Neumann solver's default).              */
int64_t bhg_mlp_wsk_launches(void);
/* Test / measurement hook: how many iterations of bhg_mlp_cg_solve this process ran in the HOISTED form of the chain (env
 * BHG_MLP_HOIST, default 1): every product that depends on the direction alone — h_l V_l^T and delta_l V_l, half of the
 * R-chain's matrix work — in ONE grouped split-K launch on the residual, G(p_k) = G(r_k) + beta G(p_{k-1}) by linearity of the
 * products in the lazy direction p_k = r_k + beta p_{k-1} (cg.py:53); the chain keeps the products with the constant
 * weights, in the in-workgroup split-K form with G as addend.  BHG_MLP_HOIST=0: the classic chain (A/B arm); =2: the
 * Neumann solver takes the hoisted form as well (G(v) directly — measured: no gain there, so not its default).       */
int64_t bhg_mlp_hoist_launches(void);
/* Test / measurement hook: iterations of bhg_mlp_cg_solve that formed their direction products by PROJECTION (env
 * BHG_MLP_PROJ, default 1; needs the hoisted form): G(r_{k+1}) = G(r_k) - alpha_k (G(raw_k) + shift G(p_k)) with
 * G(raw_k) from B x B Gram matrices times batch-sized arrays (raw_k = the weight-shaped outputs of H p_k are outer products of
 * batch-sized factors), G(p_{k+1}) = G(r_{k+1}) + beta_k G(p_k) — no N-sized operand is read after the first iteration. */
int64_t bhg_mlp_proj_iterations(void);
/* Test / measurement hook: launches of k_wskpl — iterations of the fully projected CG solver in its SIX-launch form (the chain's first
 * product by linearity on Rh_0(r'), beta published inside the launch; nets of >= 4 layers, padded batch 128).                      */
int64_t bhg_mlp_lin_launches(void);
/* bhg_mlp_neumann_solve (and bhg_neumann_init) accept p == NULL: the N-sized accumulator of neumann.py:64 is then never
 * written; the head kernel sums Rz(v_k), k < K, into `fws` instead, and this call turns that sum plus one R-forward pass
 * in direction v_K (`v_last`: the 2L slices of the direction buffer that holds v_K — v0 for even K, v1 for odd K) into the
 * mixed-derivative coefficient of p_final = -alpha * sum_{k=0..K} v_k  (the quantity bhg_mlp_mixed_coeff returns for a
 * materialised p).                                                                                                      */
/* Round 3: without an accumulator vector bhg_mlp_neumann_solve runs in PROJECTED form by default (BHG_MLP_PROJ != 0, >= 3 layers,
 * widths % 32 == 0): G(v') = G(v) - alpha (G(raw) + shift G(v)) on batch-sized arrays, nothing N-sized after the first iteration
 * (v0 / v1 then only carry the biases' and the head weight's slices), and a closing half pass adds Rz(v_K) to the sum itself —
 * bhg_mlp_neumann_solve reports which form ran through `projected_out` (host int, may be NULL); the caller hands that value to
 * bhg_mlp_neumann_mixed_coeff as `projected` — with 1 it ignores `v_last`.  (Round 3 kept this in a process-global map keyed by
 * the workspace address; the library holds no per-workspace state any more.) */
int bhg_mlp_neumann_mixed_coeff(const bhg_mlp* m, const void* const* v_last, const int64_t* labels, float* coeff,
                                float alpha, int K, int projected, void* fws, size_t fws_bytes, void* stream);

/* ---- the once-per-step passes on packed operands (round 5) -----------------------------------------------------------------------
 * bhg_mlp_forward / bhg_mlp_backward with the hidden layers behind the first as ONE launch each on the chain's packed form
 * (csrc/mlp/wskp.inc) instead of a split-K GEMM + reduce pair.  They need the fused workspace: the weights are packed into it first
 * (the launch the solvers would otherwise do), and the products' epilogues leave h_l / delta_l packed there as well.  After BOTH calls
 * the caller sets m->prepacked = 1 for the solves of this step (same weights, same batch, same fws).  Same results as the split-K
 * pair up to the summation order of the dot products.  bhg_mlp_supports_packed_prepare: 1 when the network takes this form (the
 * hoisted plan applies: >= 3 layers, hidden widths % 32 == 0, narrow head).                                                        */
/* The step's batch into the padded buffers: m->h[0][:B] = x (fp32 [B][dims[0]], 16-byte aligned, dims[0] % 4 == 0), labels[:B] = y.
 * One launch in place of two device-to-device copies.  Rows >= B of h[0] are the caller's to keep zero.                                */
int bhg_mlp_stage_batch(const bhg_mlp* m, const float* x, const int64_t* y, int64_t* labels, void* stream);
int bhg_mlp_supports_packed_prepare(const bhg_mlp* m);
int bhg_mlp_forward_packed(const bhg_mlp* m, const void* const* bias, const int64_t* labels, float* ce, void* fws, size_t fws_bytes,
                           void* stream);
int bhg_mlp_backward_packed(const bhg_mlp* m, const int64_t* labels, void* fws, size_t fws_bytes, void* stream);

/* Device address of the fused solvers' time-out word inside `fws` (4 bytes, zero in a freshly zeroed workspace): workgroups of the
 * fully projected CG solver's first chain launch receive beta from ANOTHER workgroup of the same launch (published in 64 replicated
 * granules, csrc/mlp/wskp.inc poll_beta).  That wait is bounded: a poller that gives up (~1 s) sets this word to 1 and continues with
 * NaN, so the solve ends with a poisoned result instead of hanging the GPU — the same contract as the resident CG kernel's grid
 * barrier (bhg_cg_timeout_flag_dev).  The caller reads and clears it (HipBackend.check_health).  NULL on a bad descriptor.          */
void* bhg_mlp_timeout_flag_dev(const bhg_mlp* m, void* fws);

/* ---- closed-form meta-weight-net (round 5; csrc/bhg_mwn.hip) --------------------------------------------------------------------
 * The UPPER problem of data reweighting (examples/learning_to_reweight/model.py:98-111 `MLP(hidden_size, num_layers = 1)`, called at
 * main.py:123-125):  s_i = sigmoid(w2 . relu(w1 * ce_i + b1) + b2),  ce: [B] detached per-sample losses, w1, b1, w2: [H], b2: [1].
 * On the reference's path autograd evaluates it when the inner loss is rebuilt (cg.py:27-32 / neumann.py:31-36) and differentiates it in
 * the final mixed VJP (cg.py:58-68 / neumann.py:44-54): ~15 ATen launches for 301 parameters.  Here one launch each way; all pointers
 * are device pointers, fp32, the launches are asynchronous on `stream`.
 *   bhg_mwn_forward   s[i] (may be NULL) and sd[i] = s[i] / B (may be NULL; what bhg_mlp_backward reads through bhg_mlp.sd)
 *   bhg_mwn_backward  gw1, gb1, gw2 [H], gb2 [1]  =  scale * d( sum_i coeff[i] * s_i ) / d(w1, b1, w2, b2)    (overwritten, not added)
 * Deterministic (one workgroup, fixed summation order).  1 <= H <= bhg_mwn_max_hidden().                                          */
int bhg_mwn_max_hidden(void);
int bhg_mwn_forward(const float* ce, int B, const float* w1, const float* b1, const float* w2, const float* b2, int H, float* s,
                    float* sd, void* stream);
int bhg_mwn_backward(const float* ce, const float* coeff, int B, const float* w1, const float* b1, const float* w2, const float* b2,
                     int H, float scale, float* gw1, float* gb1, float* gw2, float* gb2, void* stream);

/* Host only (no launch, no device access): the form bhg_mlp_cg_solve (algo 0) / bhg_mlp_neumann_solve (algo 1) will take for this
 * descriptor, with (keep_solution != 0) or without a materialised solution / accumulator vector — the decision of hoist_plan and of the
 * solvers' set-up code, printed as `key=value` pairs into `buf` (form, hoist, proj_level, lin, lin_head, closing launch, workspace
 * sizes).  Only L, B, Bp and dims of the descriptor are read.  For tests of the shape -> form map and for diagnostics; the reference has
 * no counterpart (its cg / neumann have one form: betty/hypergradient/cg.py:8-70, neumann.py:8-66).                                  */
int bhg_mlp_plan_describe(const bhg_mlp* m, int algo, int keep_solution, char* buf, size_t buf_bytes);

/* ---- networks whose widths are not multiples of 32 (round 6) ------------------------------------------------------------------------
 * The fused solvers' chain works on 32-wide tiles (bhg_mlp_supports_fused_solve).  The reference's cg / neumann are shape-agnostic
 * (betty/hypergradient/cg.py:8-70), so the host side (betty_amd/hypergradient/_mlp_hip.py: PaddedHipMLPState) keeps a ZERO-PADDED TWIN
 * of such a network — input and hidden widths rounded up to 32; a padded unit has zero weights and bias, so its pre-activation is
 * exactly 0, its ReLU mask 0, and every quantity of the solve is unchanged — and runs the same kernels on it.  This entry point is the
 * strided block copy that maintains the twin and un-pads results: dst[r * ldd + c] = src[r * lds + c], r < rows, c < cols (fp32 device
 * pointers, leading dimensions in elements; asynchronous on `stream`).                                                              */
int bhg_copy2d(float* dst, int64_t ldd, const float* src, int64_t lds, int64_t rows, int64_t cols, void* stream);

/* ---- batch normalisation inside an opaque Hessian-vector product (round 6; csrc/bhg_bn.hip) -------------------------------------------
 * The reference takes H p as the double backward torch.autograd.grad(in_grad, params, grad_outputs=p) (betty/hypergradient/cg.py:39-41,
 * neumann.py:62).  For an inner network with training-mode batch norm (BASELINE cfg 3: examples/implicit_maml/models.py:278-483) ATen
 * differentiates batch norm's backward through ~340 element-wise / reduction launches per layer and product.  This entry point is that
 * derivative in two launches: the vector-Jacobian product of
 *     F : (x, gy, gamma) -> (gx, ggamma, gbeta)      (batch_norm_backward with batch statistics, biased variance)
 * i.e. for cotangents a (of gx, shaped like x; may be NULL = zero), b (of ggamma, [C]; may be NULL), c (of gbeta, [C]; may be NULL):
 *     dx, dgy (shaped like x, overwritten), dgamma ([C], overwritten; may be NULL)  =  d( <a, gx> + <b, ggamma> + <c, gbeta> ) / d(x, gy, gamma)
 * with mean / invstd ([C], what the forward saved) treated as the functions of x they are.  x, gy, a, dx, dgy: fp32, NCHW contiguous,
 * N x C x HW elements, 16-byte aligned when HW % 4 == 0; gamma may be NULL (affine = False).  `ws`: bhg_bn_ws_bytes(C) bytes of scratch.
 * Deterministic (two-stage sums in fixed order, fp64 accumulation), asynchronous on `stream`; 32 algorithmic bytes per element.          */
size_t bhg_bn_ws_bytes(int C);
int bhg_bn_backward_vjp(const float* x, const float* gy, const float* a, const float* gamma, const float* mean, const float* invstd,
                        const float* b, const float* c, int N, int C, int HW, float* dx, float* dgy, float* dgamma, void* ws,
                        size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BHG_H_ */
