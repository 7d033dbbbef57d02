// Lab: what a plain read-only / write-only / copy kernel reaches on this part (HBM-resident 805 MB buffers), as the
// yardstick for the read-heavy fused kernels.  Build:  hipcc -O3 --offload-arch=gfx950 tools/lab/read_bw_lab.hip -o tools/lab/read_bw_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f32x4* __restrict__ in, float* out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  for (; i < n4; i += stride) acc += in[i];
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[blockIdx.x] = acc[0];
}

template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) out[i + u * stride] = v[u];
  }
  for (; i < n4; i += stride) out[i] = in[i];
}

template <int U>
__global__ __launch_bounds__(256) void copy_nt_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(&in[i + u * stride]);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
  }
}

template <int U>
__global__ __launch_bounds__(256) void copy_nts_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
  }
}

__global__ __launch_bounds__(256) void write_nt_kernel(f32x4* out, long n4) {
  const long stride = (long)gridDim.x * 256;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(v, &out[i]);
}

__global__ __launch_bounds__(256) void write_kernel(f32x4* out, long n4) {
  const long stride = (long)gridDim.x * 256;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) out[i] = v;
}

int main() {
  const long bytes = 805306368L, n4 = bytes / 16;   // 256 x 128 x 128 x 24 x 4 x 2
  f32x4 *a, *b;
  float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 1 << 20);
  hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn, double moved) {
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.1f us  %5.2f TB/s\n", name, ms * 100, moved / (ms * 1e-4) / 1e12);
  };
  for (int grid : {1024, 2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, 64, "read  U=4 grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(read_kernel<4>, dim3(grid), dim3(256), 0, 0, a, o, n4); }, (double)bytes);
    snprintf(nm, 64, "read  U=8 grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(read_kernel<8>, dim3(grid), dim3(256), 0, 0, a, o, n4); }, (double)bytes);
  }
  for (int grid : {2048, 8192}) {
    char nm[64];
    snprintf(nm, 64, "write grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, b, n4); }, (double)bytes);
    snprintf(nm, 64, "copy  U=4 grid %d (half + half)", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n4 / 2); }, (double)bytes);
    snprintf(nm, 64, "write nt grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(write_nt_kernel, dim3(grid), dim3(256), 0, 0, b, n4); }, (double)bytes);
    snprintf(nm, 64, "copy nt ld+st U=4 grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(copy_nt_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n4 / 2); }, (double)bytes);
    snprintf(nm, 64, "copy nt st only U=4 grid %d", grid);
    timeit(nm, [&] { hipLaunchKernelGGL(copy_nts_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n4 / 2); }, (double)bytes);
  }
  return 0;
}
