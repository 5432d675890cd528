// chain_probe.hip — the REAL k_wskp (betty_amd/csrc/mlp/wskp.inc, compiled with per-workgroup stamps) in the launch groupings of
// one projected iteration at cfg 2: [fwd W_1 (+T_1)] [fwd W_2 split (+T_2)] [bwd W_2 (+E_2)] [bwd W_1 (+E_1)], an L2-dirtying
// kernel between the launches.  Prints launch span and, per problem, start / prologue / K loop / epilogue of its workgroups.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DBHG_WSKP_STAMPS -I include -I betty_amd/csrc -o build_probes/chain_probe scripts/probes/chain_probe.hip
#include <algorithm>
#include <vector>
#include <stdlib.h>
#include "bhg_common.hpp"
namespace bhg {
int g_dbg[DBG_COUNT];
void set_error(const char*, ...) {}
namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kWskPad = 33;
#include "mlp/wskp.inc"
}
}
using namespace bhg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void k_touch(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}
static bool g_ragged = true;
struct Builder {
  WskpArgs g{}; int blk = 0;
  void add(const WskpProb& q_in) {
    WskpProb q = q_in;
    q.nfull = q.RA / 32; q.nstrip = 0;
    if (g_ragged && !q.raw && q.nsplit == 1 && q.RB % 64 == 0) {
      int nf = q.B / 32, rem = q.B % 32;
      if (rem > 16) { ++nf; rem = 0; }
      if (nf > 0) { q.nfull = nf; q.nstrip = rem > 0 ? q.RB / 64 : 0; }
    }
    g.p[g.n] = q; g.blk0[g.n++] = blk; blk += (((q.nfull * (q.RB / 32) + q.nstrip) * q.nsplit) + 7) & ~7;
  }
};
template <int D>
void run(const char* name, Builder b, float* junk, int junk_n, unsigned long long* stamps_dev) {
  b.g.blk0[b.g.n] = b.blk; b.g.stamps = nullptr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 100;
  auto launch = [&](WskpArgs& g) {
    bool riders = g.n <= 3;
    for (int i = 1; i < g.n && riders; ++i) riders = g.p[i].raw && g.p[i].nsplit == 1 && g.p[i].RB == g.p[0].RA;
    if (riders) {
      WskpcArgs c{}; c.a = g.p[0]; c.na = g.n >= 2 ? g.blk0[1] : b.blk; c.stamps = g.stamps;
      if (g.n >= 2) { c.r0 = {g.p[1].Ap, g.p[1].Bq, g.p[1].out, g.p[1].outp, g.p[1].K, 0}; c.nr0 = (g.n == 3 ? g.blk0[2] : b.blk) - g.blk0[1]; }
      if (g.n == 3) c.r1 = {g.p[2].Ap, g.p[2].Bq, g.p[2].out, g.p[2].outp, g.p[2].K, 0};
      hipLaunchKernelGGL(k_wskpc<D>, dim3(b.blk), dim3(64 * kWskpWaves), 0, 0, c);
    } else hipLaunchKernelGGL(k_wskp<D>, dim3(b.blk), dim3(64 * kWskpWaves), 0, 0, g);
  };
  for (int i = 0; i < 3; ++i) launch(b.g);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n); launch(b.g); }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms0 = 0; CK(hipEventElapsedTime(&ms0, e0, e1));
  WskpArgs gs = b.g; gs.stamps = stamps_dev;
  CK(hipMemset(stamps_dev, 0, sizeof(unsigned long long) * 4 * b.blk));
  hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
  launch(gs);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(4 * b.blk);
  CK(hipMemcpy(st.data(), stamps_dev, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int k = 0; k < b.blk; ++k) if (st[4 * k + 3]) { tmin = std::min(tmin, st[4 * k]); tmax = std::max(tmax, st[4 * k + 3]); }
  printf("%-30s D=%d grid %4d  launch %6.2f us (with touch, touch alone %5.2f)  span %5.2f us\n", name, D, b.blk, 1e3 * (ms - ms0) / reps, 1e3 * ms0 / reps, (tmax - tmin) * 0.01);
  for (int i = 0; i < b.g.n; ++i) {
    std::vector<double> start, pro, loop, epi, end;
    for (int k = b.g.blk0[i]; k < b.g.blk0[i + 1]; ++k) if (st[4 * k + 3]) {
      start.push_back((st[4 * k] - tmin) * 0.01); pro.push_back((st[4 * k + 1] - st[4 * k]) * 0.01);
      loop.push_back((st[4 * k + 2] - st[4 * k + 1]) * 0.01); epi.push_back((st[4 * k + 3] - st[4 * k + 2]) * 0.01); end.push_back((st[4 * k + 3] - tmin) * 0.01);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
    auto mn = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
    printf("    problem %d (%3d x %4d, K %4d, split %d): %3zu wgs  start med/max %5.2f/%5.2f  pro med %4.2f  loop min/med/max %5.2f/%5.2f/%5.2f  epi med/max %4.2f/%4.2f  end med/max %5.2f/%5.2f\n",
           i, b.g.p[i].RA, b.g.p[i].RB, b.g.p[i].K, b.g.p[i].nsplit, start.size(), med(start), mx(start), med(pro), mn(loop), med(loop), mx(loop), med(epi), mx(epi), med(end), mx(end));
  }
}
int main() {
  for (int i = 0; i < DBG_COUNT; ++i) g_dbg[i] = kDbgUnset;
  const int Bp = 128, B = 100, d1 = 2048, d2 = 1536, d3 = 384;
  auto dalloc = [](size_t n) { float* p; CK(hipMalloc(&p, sizeof(float) * n)); CK(hipMemset(p, 0, sizeof(float) * n)); return p; };
  float *Rh0p = dalloc(Bp * d1), *Rh1p = dalloc(Bp * d2), *Rd2p = dalloc(Bp * d3), *Rd1p = dalloc(Bp * d2);
  float *h1p = dalloc(Bp * d1), *h2p = dalloc(Bp * d2), *dl1p = dalloc(Bp * d2), *dl2p = dalloc(Bp * d3);
  float *W1f = dalloc((size_t)d1 * d2), *W1b = dalloc((size_t)d1 * d2), *W2f = dalloc((size_t)d2 * d3), *W2b = dalloc((size_t)d2 * d3);
  float *Rh1 = dalloc(Bp * d2), *Rd1 = dalloc(Bp * d2), *Rd0 = dalloc(Bp * d1), *mask = dalloc(Bp * d1), *add = dalloc(Bp * d1), *bias = dalloc(d1);
  float *partial = dalloc((size_t)16 * Bp * d3), *slabs = dalloc((size_t)8 * Bp * Bp * 4);
  double* pt2; CK(hipMalloc(&pt2, sizeof(double) * 4096));
  const int junk_n = 8 << 20; float* junk = dalloc(junk_n);
  unsigned long long* stamps; CK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 4096));
  auto chain = [&](const float* A, const float* W, int N, int K, float* out, float* outp, bool t2) {
    WskpProb q{}; q.Ap = A; q.Bq = W; q.RA = Bp; q.RB = N; q.K = K; q.B = B; q.nsplit = 1; q.mask = mask; q.addend = add; q.bias = t2 ? nullptr : bias;
    q.out = out; q.outp = outp; if (t2) { q.rh = mask; q.partT2 = pt2; } return q; };
  auto gram = [&](const float* A, const float* Bm, int K, int ns, int slot) {
    WskpProb q{}; q.Ap = A; q.Bq = Bm; q.RA = Bp; q.RB = Bp; q.K = K; q.B = B; q.nsplit = ns; q.raw = 1; q.out = slabs + (size_t)slot * 8 * Bp * Bp; return q; };
  auto pre_head = [&](int ns) { WskpProb q{}; q.Ap = Rh1p; q.Bq = W2f; q.RA = Bp; q.RB = d3; q.K = d2; q.B = B; q.nsplit = ns; q.raw = 1; q.out = partial; return q; };
#define BOTH(name, expr) { Builder b; expr; run<3>(name, b, junk, junk_n, stamps); } { Builder b; expr; run<2>(name, b, junk, junk_n, stamps); }
  g_ragged = false;
  BOTH("fwd W1 alone, 4 x 48 tiles", b.add(chain(Rh0p, W1f, d2, d1, Rh1, Rh1p, false)));
  BOTH("bwd W1 alone, 4 x 64 tiles", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)));
  g_ragged = true;
  BOTH("fwd W1 alone", b.add(chain(Rh0p, W1f, d2, d1, Rh1, Rh1p, false)));
  BOTH("fwd W1 + T1(4)", b.add(chain(Rh0p, W1f, d2, d1, Rh1, Rh1p, false)); b.add(gram(h1p, Rh0p, d1, 4, 0)));
  BOTH("fwd W2 split5 alone", b.add(pre_head(5)));
  BOTH("fwd W2 split5 + T2(3)", b.add(pre_head(5)); b.add(gram(h2p, Rh1p, d2, 3, 1)));
  BOTH("fwd W2 split4 + T2(3)", b.add(pre_head(4)); b.add(gram(h2p, Rh1p, d2, 3, 1)));
  BOTH("fwd W2 split5 + T2(2)", b.add(pre_head(5)); b.add(gram(h2p, Rh1p, d2, 2, 1)));
  BOTH("fwd W1 + T1(1)", b.add(chain(Rh0p, W1f, d2, d1, Rh1, Rh1p, false)); b.add(gram(h1p, Rh0p, d1, 1, 0)));
  BOTH("fwd W1 + T1(2)", b.add(chain(Rh0p, W1f, d2, d1, Rh1, Rh1p, false)); b.add(gram(h1p, Rh0p, d1, 2, 0)));
  BOTH("bwd W2 alone", b.add(chain(Rd2p, W2b, d2, d3, Rd1, Rd1p, true)));
  BOTH("bwd W2 + E2(1)", b.add(chain(Rd2p, W2b, d2, d3, Rd1, Rd1p, true)); b.add(gram(dl2p, Rd2p, d3, 1, 2)));
  BOTH("bwd W1 alone", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)));
  BOTH("bwd W1 + E1(3)", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)); b.add(gram(dl1p, Rd1p, d2, 3, 3)));
  BOTH("bwd W1 + E1(2)", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)); b.add(gram(dl1p, Rd1p, d2, 2, 3)));
  BOTH("bwd W1 + E1(1)", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)); b.add(gram(dl1p, Rd1p, d2, 1, 3)));
  BOTH("bwd W1 + E1(1) + T2(1)", b.add(chain(Rd1p, W1b, d1, d2, Rd0, nullptr, true)); b.add(gram(dl1p, Rd1p, d2, 1, 3)); b.add(gram(h2p, Rh1p, d2, 1, 1)));
  BOTH("E1(3) alone", b.add(gram(dl1p, Rd1p, d2, 3, 3)));
  BOTH("all four grams", b.add(gram(h1p, Rh0p, d1, 4, 0)); b.add(gram(h2p, Rh1p, d2, 3, 1)); b.add(gram(dl2p, Rd2p, d3, 1, 2)); b.add(gram(dl1p, Rd1p, d2, 3, 3)));
  return 0;
}
