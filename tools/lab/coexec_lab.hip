// Lab: do fp32 MFMAs (v_mfma_f32_16x16x4_f32) of one wave overlap with the VALU / LDS work of the other wave on the
// same SIMD?  512-thread workgroups, one per CU: waves 0-3 run an MFMA loop, waves 4-7 a loop of one instruction kind;
// each is timed (s_memtime) alone and together.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/coexec_lab.hip -o tools/lab/coexec_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int kind>
__global__ __launch_bounds__(512, 1) void k(int do_mfma, int iters, unsigned long long* out, float* sink) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  unsigned long long t0, t1;
  float res = 0.f;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (wave < 4) {
    if (do_mfma) {
      f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      float x = lane * 0.01f, y = lane * 0.02f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
      }
      res = a0[0] + a1[1] + a2[2] + a3[3];
    }
  } else if (kind > 0) {
    f32x2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const f32x2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    float s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    const float* lp = lds + lane * 2;
    for (int i = 0; i < iters; ++i) {
      if (kind == 1) {   // 16 packed fp32 FMAs
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(b), "v"(c));
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(b), "v"(c));
        }
      } else if (kind == 2) {   // 16 scalar fp32 FMAs
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s0) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s1) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s2) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s3) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s4) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s5) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s6) : "v"(b[0]), "v"(c[0]));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s7) : "v"(b[0]), "v"(c[0]));
        }
      } else if (kind == 3) {   // 16 integer adds
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i1) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i2) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i3) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i4) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i5) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i6) : "v"(lane));
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(i7) : "v"(lane));
        }
      } else if (kind == 4) {   // 16 ds_read_b64, waited for in groups of 8
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x2 r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = *(const volatile f32x2*)(lp + ((i * 16 + u * 8 + j) & 63) * 128);
#pragma unroll
          for (int j = 0; j < 8; ++j) p0 += r[j];
        }
      } else if (kind == 5) {   // 16 v_cndmask / compare pairs (PReLU-like): v_cmp + v_cndmask
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(s0) : "v"(s1), "v"(c[0]) : "vcc");
        }
      }
    }
    res = p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[0] + p6[0] + p7[0] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7 + i0 + i1 + i2 +
          i3 + i4 + i5 + i6 + i7;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (res == 123.456f) sink[tid] = res;
}

int main() {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 256 * 8 * 8);
  hipMalloc(&sink, 4096);
  const int iters = 2000;   // 32000 MFMAs / 32000 VALU instructions per wave
  const char* names[] = {"none", "v_pk_fma_f32", "v_fma_f32", "v_add_u32", "ds_read_b64", "v_cmp+v_cndmask"};
  std::vector<unsigned long long> h(256 * 8);
  auto run = [&](int do_mfma, int kind) {
    for (int rep = 0; rep < 2; ++rep) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
        default: hipLaunchKernelGGL(k<5>, dim3(256), dim3(512), 0, 0, do_mfma, iters, out, sink); break;
      }
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), out, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    printf("  mfma=%d other=%-16s: MFMA waves %8.0f cycles (%.1f / MFMA), other waves %8.0f cycles (%.2f / instr)\n", do_mfma,
           names[kind], m / 1024, m / 1024 / (iters * 16.0), v / 1024, v / 1024 / (iters * 16.0));
  };
  run(1, 0);
  for (int kind = 1; kind <= 5; ++kind) {
    run(0, kind);
    run(1, kind);
  }
  return 0;
}
