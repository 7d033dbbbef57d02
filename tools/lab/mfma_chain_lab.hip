// Lab: does a v_mfma_f32_32x32x2_f32 that accumulates into the previous MFMA's result issue back to back, or does it
// wait?  One wave per SIMD runs N MFMAs either on ONE accumulator (fully dependent chain) or round-robin on 2 / 4
// accumulators.  Build: hipcc -O3 --offload-arch=gfx950 tools/lab/mfma_chain_lab.hip -o tools/lab/mfma_chain_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(64) void chain_kernel(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 1024 * 64 * 4);
  hipMalloc(&cyc, 1024 * 8);
  const int iters = 1000;   // 16 MFMAs per iteration
  unsigned long long h[4];
#define RUN(N)                                                                     \
  hipLaunchKernelGGL(chain_kernel<N>, dim3(1024), dim3(64), 0, 0, out, cyc, iters); \
  hipDeviceSynchronize();                                                          \
  hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);                                    \
  printf("%d accumulator(s): %.1f cycles per MFMA (1 wave per SIMD)\n", N, (double)h[0] / (16.0 * iters));
  RUN(1) RUN(2) RUN(4)
  return 0;
}
