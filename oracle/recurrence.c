/*
 * recurrence.c — CPU oracle for the flat-vector recurrences.  TEST INFRASTRUCTURE, NOT PRODUCT:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain scalar C restatement of the vector arithmetic of leopard-ai/betty v0.2.1:
 *   betty/hypergradient/cg.py:42-56, neumann.py:62-66, darts.py:29-38,49-50,62-63,
 *   betty/utils.py:117-118.
 * Element-wise operations reproduce the reference's rounding sequence exactly (a*b rounded to
 * fp32, then +/-; compile with -ffp-contract=off).  Dot products accumulate in double — the one
 * deliberate difference to the reference's fp32 `torch.dot` (same choice as the HIP kernels, so the
 * kernels can be compared with this file to the last ulp while both stay within ~1e-7 of torch).
 * Pinned by tests/test_oracle.py against the goldens generated from the real reference.
 */
#include <math.h>
#include <stdint.h>

/* cg.py:42,44,46: den = dot(cg_alpha * Hp, p) */
double orc_dot_scaled(const float* hp, const float* p, int64_t n, float cg_alpha) {
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const float s = cg_alpha * hp[i];
    acc += (double)s * (double)p[i];
  }
  return acc;
}

/* cg.py:45 / 51-52: dot(r, r); also darts.py:30 (squared norm) */
double orc_sqnorm(const float* v, int64_t n) {
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) acc += (double)v[i] * (double)v[i];
  return acc;
}

/* cg.py:34-36 */
void orc_cg_init(const float* v, float* x, float* r, float* p, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    x[i] = 0.0f;
    r[i] = v[i];
    p[i] = v[i];
  }
}

/* cg.py:50: r <- r - a*Hp ; returns the partial r'.r' of cg.py:51-52 */
double orc_cg_resid(const float* hp, float* r, int64_t n, float a) {
  double acc = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const float t = a * hp[i];
    const float nr = r[i] - t;
    r[i] = nr;
    acc += (double)nr * (double)nr;
  }
  return acc;
}

/* cg.py:49,53 (+56 and the negation of 59/68 when out_scale != 0) */
void orc_cg_dir(float* x, const float* r, float* p, int64_t n, float a, float b, float out_scale) {
  for (int64_t i = 0; i < n; ++i) {
    const float ap = a * p[i];
    float nx = x[i] + ap;
    if (out_scale != 0.0f) nx = out_scale * nx;
    const float bp = b * p[i];
    x[i] = nx;
    p[i] = r[i] + bp;
  }
}

/* neumann.py:63-64 (+66 and negation when out_scale != 0) */
void orc_neumann_step(const float* hv, float* v, float* p, int64_t n, float alpha, float out_scale) {
  for (int64_t i = 0; i < n; ++i) {
    const float t = alpha * hv[i];
    const float nv = v[i] - t;
    float np = nv + p[i];
    if (out_scale != 0.0f) np = out_scale * np;
    v[i] = nv;
    p[i] = np;
  }
}

/* darts.py:29-35: eps = R / (float32 norm + 1e-15), division in double */
double orc_darts_eps(double sumsq, double R) {
  float nf = (float)sqrt(sumsq);
  nf = nf + 1e-15f;
  return R / (double)nf;
}

/* darts.py:37-38,49-50,62-63: dst += a * src */
void orc_axpy(float* dst, const float* src, int64_t n, float a) {
  for (int64_t i = 0; i < n; ++i) {
    const float t = a * src[i];
    dst[i] = dst[i] + t;
  }
}

/* utils.py:117-118 / cg.py:56: dst = s * src */
void orc_scale_copy(float* dst, const float* src, int64_t n, float s) {
  for (int64_t i = 0; i < n; ++i) dst[i] = s * src[i];
}
