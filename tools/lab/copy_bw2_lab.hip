// Lab (round 3): can a plain copy reach the 6.29 TB/s MI355X_MICROARCH.md:36 quotes for a float4 copy?  read_bw_lab.hip
// (round 2: grid-stride loops) topped out at 4.6-5.1 TB/s.  Variants here: what each wave touches per iteration
// (grid-stride vs a contiguous run), bytes in flight per lane, workgroup size, buffer size, and the runtime's own
// hipMemcpyDtoD as an outside reference.  TB/s counts bytes read + bytes written (a copy of N bytes moves 2N).
// Build: hipcc -O3 --offload-arch=gfx950 tools/lab/copy_bw2_lab.hip -o tools/lab/copy_bw2_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave copies contiguous runs of U KiB: lane l moves 16 B at run + u*1 KiB + 16 l
template <int U, int T>
__global__ __launch_bounds__(T) void copy_run_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * (T / 64) + (threadIdx.x >> 6), nwaves = (long)gridDim.x * (T / 64);
  for (long r = wave * (64L * U); r + 64L * U <= n4; r += nwaves * (64L * U)) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[r + u * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) out[r + u * 64 + lane] = v[u];
  }
}

// every WORKGROUP owns one contiguous slab of the buffer (slab = n4 / gridDim.x), waves interleaved inside it
template <int U, int T>
__global__ __launch_bounds__(T) void copy_slab_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
  const long slab = n4 / gridDim.x, base = (long)blockIdx.x * slab;
  long i = threadIdx.x;
  for (; i + (U - 1) * T < slab; i += (long)U * T) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + i + u * T];
#pragma unroll
    for (int u = 0; u < U; ++u) out[base + i + u * T] = v[u];
  }
  for (; i < slab; i += T) out[base + i] = in[base + i];   // tail of the slab (the first run of this lab skipped it: its
                                                           // 7 TB/s lines for tiny slabs were short copies)
}

template <int U>
__global__ __launch_bounds__(256) void copy_stride_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n4) {
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

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (long mb : {403L, 1610L}) {
    const long bytes = mb * 1000000L / 16 * 16, n4 = bytes / 16;
    f32x4 *a, *b;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes);
    hipMemset(b, 0, bytes);
    auto timeit = [&](const char* name, auto fn) {
      for (int i = 0; i < 3; ++i) fn();
      hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) fn();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("  %-44s %8.1f us  %5.2f TB/s\n", name, ms * 100, 2.0 * bytes / (ms * 1e-4) / 1e12);
    };
    printf("copy of %ld MB (read + write = %ld MB):\n", mb, 2 * mb);
    timeit("hipMemcpyDtoDAsync", [&] { hipMemcpyDtoDAsync(b, a, bytes, 0); });
    char nm[96];
    for (int grid : {512, 1024, 2048, 4096, 16384}) {
      snprintf(nm, 96, "grid-stride U=4 T=256 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL(copy_stride_kernel<4>, dim3(grid), dim3(256), 0, 0, a, b, n4); });
      snprintf(nm, 96, "wave runs  U=4 (4 KiB) T=256 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL((copy_run_kernel<4, 256>), dim3(grid), dim3(256), 0, 0, a, b, n4); });
      snprintf(nm, 96, "wave runs  U=8 (8 KiB) T=256 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL((copy_run_kernel<8, 256>), dim3(grid), dim3(256), 0, 0, a, b, n4); });
      snprintf(nm, 96, "wave runs  U=16 (16 KiB) T=512 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL((copy_run_kernel<16, 512>), dim3(grid), dim3(512), 0, 0, a, b, n4); });
      snprintf(nm, 96, "slab per WG U=8 T=1024 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL((copy_slab_kernel<8, 1024>), dim3(grid), dim3(1024), 0, 0, a, b, n4); });
      snprintf(nm, 96, "slab per WG U=4 T=256 grid %d", grid);
      timeit(nm, [&] { hipLaunchKernelGGL((copy_slab_kernel<4, 256>), dim3(grid), dim3(256), 0, 0, a, b, n4); });
    }
    hipFree(a);
    hipFree(b);
  }
  return 0;
}
