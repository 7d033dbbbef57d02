// Lab: issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD (1 / 2 / 4 accumulator chains, operands random or
// zero), from two waves per SIMD, and how much VALU / LDS work of a second wave fits beside it (the bf16 sibling of
// coexec_lab.hip, whose fp32 MFMAs turned out to BE vector-ALU work).  s_memtime cycles and wall clock side by side.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/lab/coexec_bf16_lab.hip -o tools/lab/coexec_bf16_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BF(x) __builtin_bit_cast(bf16x8, (x))

// chains: accumulators used round-robin by the MFMA waves; mfma_waves: 4 = waves 0-3 only, 8 = all waves run MFMAs
template <int kind, int chains>
__global__ __launch_bounds__(512, 1) void k(int mfma_waves, int iters, unsigned seed, unsigned long long* out, float* sink) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 8192; i += 512) lds[i] = i * 0.001f;
  __syncthreads();
  unsigned long long t0, t1;
  float res = 0.f;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (wave < mfma_waves) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    unsigned r = seed ? seed * (lane + 1) * 2654435761u : 0u;
    u32x4 x, y;
    for (int j = 0; j < 4; ++j) {
      r = r * 1664525u + 1013904223u;
      x[j] = seed ? ((r & 0x7fff7fffu) | 0x3c003c00u) & 0x3fff3fffu : 0u;   // bf16 pairs around 0.01 .. 2
      r = r * 1664525u + 1013904223u;
      y[j] = seed ? ((r & 0x7fff7fffu) | 0x3c003c00u) & 0xbfffbfffu : 0u;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(x), BF(y), a0, 0, 0, 0);
        if (chains == 1) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(y), BF(x), a0, 0, 0, 0);
        else a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(y), BF(x), a1, 0, 0, 0);
        if (chains == 4) a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(x), BF(y), a2, 0, 0, 0);
        else if (chains == 2) a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(x), BF(y), a0, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(x), BF(y), a0, 0, 0, 0);
        if (chains == 4) a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(y), BF(x), a3, 0, 0, 0);
        else if (chains == 2) a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(y), BF(x), a1, 0, 0, 0);
        else a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF(y), BF(x), a0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    res = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (kind > 0) {
    f32x2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const f32x2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    float s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    int i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3, i4 = 4, i5 = 5, i6 = 6, i7 = 7;
    const float* lp = lds + lane * 2;
    for (int i = 0; i < iters; ++i) {
      if (kind == 1) {
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
      } else if (kind == 2) {
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
      } else if (kind == 3) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(i0) : "v"(lane));
          asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(i1) : "v"(lane), "v"(i7));
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(i2) : "v"(lane));
          asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(i3) : "v"(lane), "v"(i7));
          asm volatile("v_and_b32 %0, %0, %1" : "+v"(i4) : "v"(lane));
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s5) : "v"(c[0]));
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s6) : "v"(c[0]));
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s7) : "v"(c[0]));
        }
      } else if (kind == 4) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f32x2 r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = *(const volatile f32x2*)(lp + ((i * 16 + u * 8 + j) & 63) * 128);
#pragma unroll
          for (int j = 0; j < 8; ++j) p0 += r[j];
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

// Same experiment with v_mfma_f32_32x32x16_bf16 (twice the FLOPs per instruction: half as many issue slots for the same
// work): does the partner wave's VALU get more of the SIMD's issue port beside it?
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int kind>
__global__ __launch_bounds__(512, 1) void k32(int mfma_waves, int iters, unsigned seed, unsigned long long* out, float* sink) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned long long t0, t1;
  float res = 0.f;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (wave < mfma_waves) {
    f32x16 a0, a1;
    for (int j = 0; j < 16; ++j) a0[j] = 0.f, a1[j] = 0.f;
    unsigned r = seed * (lane + 1) * 2654435761u;
    u32x4 x, y;
    for (int j = 0; j < 4; ++j) {
      r = r * 1664525u + 1013904223u;
      x[j] = ((r & 0x7fff7fffu) | 0x3c003c00u) & 0x3fff3fffu;
      r = r * 1664525u + 1013904223u;
      y[j] = ((r & 0x7fff7fffu) | 0x3c003c00u) & 0xbfffbfffu;
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {      // 8 instructions = the FLOPs of 16 of the 16x16x32 form
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(x), BF(y), a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF(y), BF(x), a1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    res = a0[0] + a1[1];
  } else if (kind > 0) {
    f32x2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;
    const f32x2 b = {0.999f, 1.001f}, c = {0.001f, -0.001f};
    float s0 = 1, s1 = 2, s2 = 3, s3 = 4, s4 = 5, s5 = 6, s6 = 7, s7 = 8;
    for (int i = 0; i < iters; ++i) {
      if (kind == 1) {
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
      } else {
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
      }
    }
    res = p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[0] + p6[0] + p7[0] + s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7;
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  if (res == 123.456f) sink[tid] = res;
}

int main() {
  unsigned long long* out;
  float* sink;
  (void)hipMalloc(&out, 256 * 8 * 8);
  (void)hipMalloc(&sink, 4096);
  const int iters = 4000;   // 64000 MFMAs / VALU instructions per wave
  const char* names[] = {"none", "v_pk_fma_f32", "v_fma_f32", "and/perm/sub mix", "ds_read_b64"};
  std::vector<unsigned long long> h(256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](int chains, int mfma_waves, int kind, unsigned seed) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
#define L(KD, CH) hipLaunchKernelGGL((k<KD, CH>), dim3(256), dim3(512), 0, 0, mfma_waves, iters, seed, out, sink)
      if (chains == 1) { if (kind == 0) L(0, 1); else if (kind == 1) L(1, 1); else if (kind == 2) L(2, 1); else if (kind == 3) L(3, 1); else L(4, 1); }
      else if (chains == 2) { if (kind == 0) L(0, 2); else if (kind == 1) L(1, 2); else if (kind == 2) L(2, 2); else if (kind == 3) L(3, 2); else L(4, 2); }
      else { if (kind == 0) L(0, 4); else if (kind == 1) L(1, 4); else if (kind == 2) L(2, 4); else if (kind == 3) L(3, 4); else L(4, 4); }
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h.data(), out, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    printf("  chains=%d mfma_waves=%d data=%-6s other=%-16s: waves 0-3 %8.0f cyc (%.1f / instr), waves 4-7 %8.0f cyc (%.2f / instr), wall %.1f us\n",
           chains, mfma_waves, seed ? "random" : "zero", names[kind], m / 1024, m / 1024 / (iters * 16.0), v / 1024,
           v / 1024 / (iters * 16.0), ms * 1e3);
  };
  for (unsigned seed : {0u, 7u})
    for (int chains : {1, 2, 4}) run(chains, 4, 0, seed);
  run(4, 8, 0, 7u);
  run(2, 8, 0, 7u);
  for (int kind = 1; kind <= 4; ++kind) {
    run(4, 0, kind, 7u);
    run(4, 4, kind, 7u);
  }
  printf("-- v_mfma_f32_32x32x16_bf16 (8 per iteration = the FLOPs of 16 of the 16x16x32 form)\n");
  auto run32 = [&](int mfma_waves, int kind) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL((k32<0>), dim3(256), dim3(512), 0, 0, mfma_waves, iters, 7u, out, sink);
      else if (kind == 1) hipLaunchKernelGGL((k32<1>), dim3(256), dim3(512), 0, 0, mfma_waves, iters, 7u, out, sink);
      else hipLaunchKernelGGL((k32<2>), dim3(256), dim3(512), 0, 0, mfma_waves, iters, 7u, out, sink);
      (void)hipEventRecord(e1);
      (void)hipDeviceSynchronize();
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(h.data(), out, 256 * 8 * 8, hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w];
    printf("  mfma_waves=%d other=%-16s: waves 0-3 %8.0f cyc (%.1f / mfma), waves 4-7 %8.0f cyc (%.2f / instr), wall %.1f us\n",
           mfma_waves, names[kind], m / 1024, m / 1024 / (iters * 8.0), v / 1024, v / 1024 / (iters * 16.0), ms * 1e3);
  };
  run32(4, 0);
  run32(8, 0);
  for (int kind = 1; kind <= 2; ++kind) {
    run32(0, kind);
    run32(4, kind);
  }
  return 0;
}
