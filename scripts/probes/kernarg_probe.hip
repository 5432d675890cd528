// kernarg_probe.hip — round 6: what does the kernel-argument fetch at the top of a kernel cost in a chain of DEPENDENT launches, and does
// gfx950's kernarg PRELOAD (SPI loads the first dwords of the argument segment into SGPRs before the wave starts; hipcc: -mllvm
// -amdgpu-kernarg-preload-count=N, scalar / pointer arguments only — a struct passed by value is not preloaded) take it away?
//   k_struct : one struct by value (how the K-loop kernels of libbhg take their arguments)
//   k_scalar : the same fields as leading scalar arguments (preloaded when built with the flag)
// Each launch: 256 workgroups x 256 threads, every thread loads 16 bytes through the argument's pointer, adds, stores — the first global
// load cannot issue before the pointer is known.  2000 dependent launches ping-ponging two buffers; microseconds per launch.
// Build twice: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=16] -o build_probes/kernarg_probe[_preload] scripts/probes/kernarg_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Args { const float* in; float* out; const float* bias; const float* mask; int n; int stride; float scale; float shift; long long pad[20]; };
__global__ __launch_bounds__(256) void k_struct(Args a) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i < a.n) {
    float4 v = *reinterpret_cast<const float4*>(a.in + i);
    const float4 b = *reinterpret_cast<const float4*>(a.bias + (i % a.stride));
    v.x = v.x * a.scale + b.x + a.shift; v.y = v.y * a.scale + b.y; v.z = v.z * a.scale + b.z; v.w = v.w * a.scale + b.w;
    *reinterpret_cast<float4*>(a.out + i) = v;
  }
}
__global__ __launch_bounds__(256) void k_scalar(const float* in, float* out, const float* bias, const float* mask, int n, int stride, float scale, float shift, Args rest) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i < n) {
    float4 v = *reinterpret_cast<const float4*>(in + i);
    const float4 b = *reinterpret_cast<const float4*>(bias + (i % stride));
    v.x = v.x * scale + b.x + shift; v.y = v.y * scale + b.y; v.z = v.z * scale + b.z; v.w = v.w * scale + b.w;
    *reinterpret_cast<float4*>(out + i) = v;
  }
}
int main() {
  const int n = 256 * 256 * 4;
  float *a, *b, *bias; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&bias, 4096 * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(bias, 0, 4096 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 2000;
  for (int which = 0; which < 2; ++which) {
    float best = 1e30f;
    for (int trial = 0; trial < 5; ++trial) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) {
        Args q{}; q.in = (i & 1) ? b : a; q.out = (i & 1) ? a : b; q.bias = bias; q.mask = bias; q.n = n; q.stride = 4096; q.scale = 0.999f; q.shift = 1e-3f;
        if (which == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, 0, q);
        else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, 0, q.in, q.out, q.bias, q.mask, q.n, q.stride, q.scale, q.shift, q);
      }
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (1e3f * ms / reps < best) best = 1e3f * ms / reps;
    }
    printf("%s: %.3f us per dependent launch\n", which == 0 ? "struct by value      " : "leading scalar args  ", best);
  }
  return 0;
}
