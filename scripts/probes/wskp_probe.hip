// wskp_probe.hip — standalone probe of the packed skinny GEMM (round 4): what bounds a 32 x 32 x K tile per workgroup?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wskp_probe scripts/probes/wskp_probe.hip ; run on the GPU box.
// Variants: operand layout (k-block major / tile major), register stages, loads only / MFMAs only, waves per workgroup;
// per-workgroup timestamps (s_memrealtime, 100 MHz) give launch ramp, K loop and epilogue separately.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

using f32x4 = __attribute__((ext_vector_type(4))) float;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { LAY_KB = 0, LAY_TILE = 1 };
enum { MODE_FULL = 0, MODE_LOADS = 1, MODE_MFMA = 2 };

struct Args {
  const float* Ap; const float* Bq; float* out; unsigned long long* stamps;
  int RA, RB, K;
};

// layout KB:   off(r, kb) = (kb * R + r) * 16            stride between k-blocks = 16 * R floats
// layout TILE: off(r, kb) = (((r >> 5) * (K / 16) + kb) * 32 + (r & 31)) * 16   stride between k-blocks = 512 floats
template <int D, int LAY, int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_probe(Args q) {
  __shared__ float sP[WAVES][32][33];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lk = lane >> 4;
  unsigned long long t0 = 0, t1 = 0, t2 = 0;
  if (q.stamps && threadIdx.x == 0) t0 = wall_clock64();
  const int ntm = q.RA / 32, ntn = q.RB / 32;
  const int tile = blockIdx.x;
  int tm, tn;
  if ((ntn & 7) == 0) { const int xcd = tile & 7, j = tile >> 3; tm = j % ntm; tn = (j / ntm) * 8 + xcd; }
  else { tm = tile % ntm; tn = tile / ntm; }
  const int m0 = tm * 32, n0 = tn * 32;
  const int nct = q.K / 32;
  const int c0 = (wave * nct) / WAVES, c1 = ((wave + 1) * nct) / WAVES;
  const int nch = c1 - c0;
  f32x4 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
  int64_t sA, sB; const float* gA; const float* gB;
  if (LAY == LAY_KB) {
    sA = 16LL * q.RA; sB = 16LL * q.RB;
    gA = q.Ap + (int64_t)(2 * c0) * sA + (m0 + li) * 16 + 4 * lk;
    gB = q.Bq + (int64_t)(2 * c0) * sB + (n0 + li) * 16 + 4 * lk;
  } else {
    sA = 512; sB = 512;
    gA = q.Ap + ((int64_t)tm * (q.K / 16) + 2 * c0) * 512 + li * 16 + 4 * lk;
    gB = q.Bq + ((int64_t)tn * (q.K / 16) + 2 * c0) * 512 + li * 16 + 4 * lk;
  }
  f32x4 sa[D][2][2], sb[D][2][2];
  auto load = [&](const int d, int ch) {
    ch = min(ch, nch - 1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* pa = gA + (int64_t)(2 * ch + h) * sA;
      const float* pb = gB + (int64_t)(2 * ch + h) * sB;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        if (MODE != MODE_MFMA) {
          sa[d][rb][h] = *reinterpret_cast<const f32x4*>(pa + rb * 256);
          sb[d][rb][h] = *reinterpret_cast<const f32x4*>(pb + rb * 256);
        }
      }
    }
  };
  auto compute = [&](const int d) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (MODE == MODE_LOADS) {   // keep the loads alive without the matrix pipe
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) { acc[rb][h] += sa[d][rb][h]; acc[rb][h] += sb[d][rb][h]; }
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float a0 = sa[d][0][h][c], a1 = sa[d][1][h][c];
          const float b0 = sb[d][0][h][c], b1 = sb[d][1][h][c];
          acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
        }
      }
    }
  };
  if (MODE == MODE_MFMA) {
#pragma unroll
    for (int d = 0; d < D; ++d)
      for (int rb = 0; rb < 2; ++rb) for (int h = 0; h < 2; ++h) for (int c = 0; c < 4; ++c) { sa[d][rb][h][c] = 1.f + lane; sb[d][rb][h][c] = 0.5f; }
  }
  if (q.stamps && threadIdx.x == 0) t1 = wall_clock64();
  if (nch > 0) {
#pragma unroll
    for (int d = 0; d < D; ++d) { load(d, d); __builtin_amdgcn_sched_barrier(0); }
    int s = 0;
    for (; s + 2 * D <= nch; s += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        compute(d); __builtin_amdgcn_sched_barrier(0);
        load(d, s + D + d); __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) { if (s + d < nch) compute(d); if (s + D + d < nch) load(d, s + D + d); }
    s += D;
#pragma unroll
    for (int d = 0; d < D; ++d) if (s + d < nch) compute(d);
  }
  if (q.stamps && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0)"); t2 = wall_clock64(); }
  for (int rb = 0; rb < 2; ++rb) for (int cb = 0; cb < 2; ++cb) for (int r = 0; r < 4; ++r) sP[wave][16 * rb + 4 * lk + r][16 * cb + li] = acc[rb][cb][r];
  __syncthreads();
  for (int e = threadIdx.x; e < 1024; e += 64 * WAVES) {
    const int row = e >> 5, col = e & 31;
    float v = 0.f;
    for (int w = 0; w < WAVES; ++w) v += sP[w][row][col];
    q.out[(int64_t)(m0 + row) * q.RB + n0 + col] = v;
  }
  if (q.stamps && threadIdx.x == 0) {
    const unsigned long long t3 = wall_clock64();
    unsigned long long* s = q.stamps + 4 * blockIdx.x;
    s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3;
  }
}

// a cheap kernel that dirties the L2s between the products, like the rest of an iteration does
__global__ void k_touch(float* p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.f;
}

template <int D, int LAY, int MODE, int WAVES>
void run(const char* name, Args fw, Args bw, float* junk, int junk_n, unsigned long long* stamps_dev) {
  const int gf = (fw.RA / 32) * (fw.RB / 32), gb = (bw.RA / 32) * (bw.RB / 32);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 100;
  for (int i = 0; i < 5; ++i) {
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gf), dim3(64 * WAVES), 0, 0, fw);
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gb), dim3(64 * WAVES), 0, 0, bw);
  }
  CK(hipDeviceSynchronize());
  // (1) pair time, back to back
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gf), dim3(64 * WAVES), 0, 0, fw);
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gb), dim3(64 * WAVES), 0, 0, bw);
  }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  // (2) with an L2-dirtying kernel between them
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gf), dim3(64 * WAVES), 0, 0, fw);
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
    hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gb), dim3(64 * WAVES), 0, 0, bw);
  }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms2 = 0; CK(hipEventElapsedTime(&ms2, e0, e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
    hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
  }
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms3 = 0; CK(hipEventElapsedTime(&ms3, e0, e1));
  // (3) stamps of one forward launch
  Args fs = fw; fs.stamps = stamps_dev;
  hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, 0, junk, junk_n);
  hipLaunchKernelGGL((k_probe<D, LAY, MODE, WAVES>), dim3(gf), dim3(64 * WAVES), 0, 0, fs);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(4 * gf);
  CK(hipMemcpy(st.data(), stamps_dev, sizeof(unsigned long long) * 4 * gf, hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int b = 0; b < gf; ++b) { tmin = std::min(tmin, st[4 * b]); tmax = std::max(tmax, st[4 * b + 3]); }
  std::vector<double> start, pro, loop, epi;
  for (int b = 0; b < gf; ++b) {
    start.push_back((st[4 * b] - tmin) * 0.01); pro.push_back((st[4 * b + 1] - st[4 * b]) * 0.01);
    loop.push_back((st[4 * b + 2] - st[4 * b + 1]) * 0.01); epi.push_back((st[4 * b + 3] - st[4 * b + 2]) * 0.01);
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
  auto mn = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
  printf("%-34s pair %6.2f us | with touch %6.2f us (touch alone %5.2f) | fwd stamps: span %5.2f start med/max %4.2f/%4.2f  pro %4.2f  loop min/med/max %5.2f/%5.2f/%5.2f  epi med/max %4.2f/%4.2f\n",
         name, 1e3 * ms / reps, 1e3 * (ms2 - ms3) / reps, 1e3 * ms3 / reps / 2, (tmax - tmin) * 0.01, med(start), mx(start), med(pro),
         mn(loop), med(loop), mx(loop), med(epi), mx(epi));
}

int main() {
  const int Bp = 128, d1 = 2048, d2 = 1536;
  // forward: out [128 x 1536], K = 2048;  backward: out [128 x 2048], K = 1536
  float *A1, *W1f, *A2, *W1b, *o1, *o2, *junk; unsigned long long* stamps;
  const int junk_n = 8 << 20;
  CK(hipMalloc(&A1, sizeof(float) * Bp * d1)); CK(hipMalloc(&W1f, sizeof(float) * d1 * d2));
  CK(hipMalloc(&A2, sizeof(float) * Bp * d2)); CK(hipMalloc(&W1b, sizeof(float) * d1 * d2));
  CK(hipMalloc(&o1, sizeof(float) * Bp * d2)); CK(hipMalloc(&o2, sizeof(float) * Bp * d1));
  CK(hipMalloc(&junk, sizeof(float) * junk_n)); CK(hipMalloc(&stamps, sizeof(unsigned long long) * 4 * 4096));
  std::vector<float> h(d1 * d2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(W1f, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(W1b, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(A1, h.data(), sizeof(float) * Bp * d1, hipMemcpyHostToDevice));
  CK(hipMemcpy(A2, h.data(), sizeof(float) * Bp * d2, hipMemcpyHostToDevice));
  CK(hipMemset(junk, 0, sizeof(float) * junk_n));
  Args fw{A1, W1f, o1, nullptr, Bp, d2, d1}, bw{A2, W1b, o2, nullptr, Bp, d1, d2};
#define RUN(D, LAY, MODE, W) run<D, LAY, MODE, W>(#D " " #LAY " " #MODE " waves" #W, fw, bw, junk, junk_n, stamps)
  RUN(3, LAY_KB, MODE_FULL, 8);
  RUN(2, LAY_KB, MODE_FULL, 8);
  RUN(4, LAY_KB, MODE_FULL, 8);
  RUN(3, LAY_KB, MODE_LOADS, 8);
  RUN(3, LAY_KB, MODE_MFMA, 8);
  RUN(3, LAY_TILE, MODE_FULL, 8);
  RUN(2, LAY_TILE, MODE_FULL, 8);
  RUN(4, LAY_TILE, MODE_FULL, 8);
  RUN(3, LAY_TILE, MODE_LOADS, 8);
  RUN(3, LAY_KB, MODE_FULL, 4);
  RUN(3, LAY_TILE, MODE_FULL, 4);
  RUN(4, LAY_TILE, MODE_FULL, 4);
  RUN(3, LAY_TILE, MODE_LOADS, 4);
  RUN(3, LAY_TILE, MODE_MFMA, 4);
  RUN(2, LAY_TILE, MODE_FULL, 16);
  return 0;
}
