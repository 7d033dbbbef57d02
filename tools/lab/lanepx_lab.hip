// Lab: "lane = pixel" arithmetic for the 24-channel BlazeBlocks -- is a 1x1 conv as v_pk_fma_f32 with the weights in SGPR pairs
// (hipcc emits `v_pk_fma_f32 v[a:a+1], v[x:x+1], s[w:w+1], v[a:a+1] op_sel_hi:[0,1,0]` for acc += w_uniform * {x, x}) as fast as its
// instruction count says (288 packed FMAs per 64 pixels = 1152 issue cycles, against 24 fp32 MFMAs of 64 cycles = 1536 for the same
// pixels with the 24 -> 32 channel padding of the MFMA tile), and what do the wave-wide DPP shifts of the depthwise form cost?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/lanepx_lab.hip -o tools/lab/lanepx_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: 1x1 only; 1: + depthwise partial sums (9 taps x 12 pairs, own pixel) + 48 wave shifts; 2: shifts only
template <int MODE>
__global__ __launch_bounds__(256, 2) void lane_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ wd,
                                                      float* __restrict__ y, int n, int iters) {
  const int px = blockIdx.x * 256 + threadIdx.x;
  if (px >= n) return;
  f32x2 xv[12];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const f32x4 t = *(const f32x4*)(x + (long)px * 24 + 4 * i);
    xv[2 * i] = f32x2{t[0], t[1]};
    xv[2 * i + 1] = f32x2{t[2], t[3]};
  }
  const f32x2* wp = (const f32x2*)w;    // [ci][12 pairs]
  const f32x2* wt = (const f32x2*)wd;   // [9][12 pairs]
  for (int it = 0; it < iters; ++it) {
    f32x2 d[12];
    if (MODE >= 1) {
      // three rows of "own pixel" values (here: the same registers), left / centre / right partial sums per channel pair
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        f32x2 L = xv[j] * wt[0 * 12 + j], C = xv[j] * wt[1 * 12 + j], R = xv[j] * wt[2 * 12 + j];
        if (MODE == 1) {
          L += xv[(j + 1) % 12] * wt[3 * 12 + j]; C += xv[(j + 1) % 12] * wt[4 * 12 + j]; R += xv[(j + 1) % 12] * wt[5 * 12 + j];
          L += xv[(j + 2) % 12] * wt[6 * 12 + j]; C += xv[(j + 2) % 12] * wt[7 * 12 + j]; R += xv[(j + 2) % 12] * wt[8 * 12 + j];
        }
        float l0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, L[0]), 0x138, 0xf, 0xf, true));
        float l1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, L[1]), 0x138, 0xf, 0xf, true));
        float r0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, R[0]), 0x130, 0xf, 0xf, true));
        float r1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, R[1]), 0x130, 0xf, 0xf, true));
        d[j] = f32x2{C[0] + l0 + r0, C[1] + l1 + r1};
      }
    } else {
#pragma unroll
      for (int j = 0; j < 12; ++j) d[j] = xv[j];
    }
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < 12; ++j) xv[j] = d[j];
      continue;
    }
    f32x2 acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = xv[j];                      // shortcut
#pragma unroll
    for (int ci = 0; ci < 24; ++ci) {
      const float xs = d[ci >> 1][ci & 1];
      const f32x2 xx = {xs, xs};
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] += wp[ci * 12 + j] * xx;
    }
#pragma unroll
    for (int j = 0; j < 12; ++j) xv[j] = f32x2{__builtin_fmaxf(acc[j][0], 0.f), __builtin_fmaxf(acc[j][1], 0.f)};
  }
#pragma unroll
  for (int j = 0; j < 6; ++j) *(f32x4*)(y + (long)px * 24 + 4 * j) = f32x4{xv[2 * j][0], xv[2 * j][1], xv[2 * j + 1][0], xv[2 * j + 1][1]};
}

template <int MODE>
static void run(const char* name, float* x, float* w, float* wd, float* y, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float t[3];
  const int its[3] = {1, 9, 17};
  for (int k = 0; k < 3; ++k) {
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(lane_kernel<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, x, w, wd, y, n, its[k]);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(lane_kernel<MODE>, dim3((n + 255) / 256), dim3(256), 0, 0, x, w, wd, y, n, its[k]);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[k], e0, e1);
    t[k] *= 200.f;   // us per launch
  }
  const double per = (t[2] - t[1]) / 8.0;                     // us per iteration over all pixels
  const double waves = n / 64.0, cyc = per * 1e-6 * 1.9e9 * 1024.0 / waves;   // SIMD cycles per 64-pixel block at ~1.9 GHz
  printf("%-28s 1 / 9 / 17 iterations: %7.1f %7.1f %7.1f us   -> %6.1f us per iteration = ~%5.0f SIMD cycles per 64 pixels (%s)\n", name,
         t[0], t[1], t[2], per, cyc, hipGetErrorString(hipGetLastError()));
}

int main() {
  const int n = 256 * 128 * 128;
  float *x, *y, *w, *wd;
  hipMalloc(&x, (size_t)n * 96); hipMalloc(&y, (size_t)n * 96); hipMalloc(&w, 576 * 4); hipMalloc(&wd, 216 * 4);
  std::vector<float> hx((size_t)n * 24), hw(576), hd(216);
  unsigned s = 1u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = 0.1f * rnd();
  for (auto& v : hd) v = 0.2f * rnd();
  hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), 576 * 4, hipMemcpyHostToDevice);
  hipMemcpy(wd, hd.data(), 216 * 4, hipMemcpyHostToDevice);
  run<0>("1x1 (288 pk_fma)", x, w, wd, y, n);
  run<2>("shifts only (36 pk + 48 dpp)", x, w, wd, y, n);
  run<1>("dw (108 pk + 48 dpp) + 1x1", x, w, wd, y, n);
  return 0;
}
