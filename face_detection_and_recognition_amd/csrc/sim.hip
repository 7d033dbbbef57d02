// sim.hip — similarity filter kernels (gfx950).
//
//  * cosine filter (cosine formula of extract_and_label_faces_from_dataset.py:106):
//      best[g] = max_j <G[g], R[j]> / (|G[g]| |R[j]|),  arg, keep = best >= tau.
//    S = G R^T is a plain fp32 GEMM (K = D), computed tile by tile on v_mfma_f32_32x32x2_f32 and reduced
//    in the epilogue (row max over the tile with wavefront shuffles, then one 64-bit atomicMax per row and
//    tile) so the M x Nr score matrix is never written to HBM.
//  * L2-to-class-mean filter, the reference's actual semantics
//    (similar_face_filtering/filter_faces_using_reference.py:71-100,183-197).
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, KC = 32, LDA = KC + 4;

__device__ __forceinline__ unsigned int f2ord(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
  return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void cosine_tile_kernel(const float* __restrict__ G, const float* __restrict__ ginv,
                                                          long M, const float* __restrict__ R,
                                                          const float* __restrict__ rinv, int Nr, int D,
                                                          unsigned long long* __restrict__ packed) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDA];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const int c4 = tid & 7, r0 = tid >> 3;

  f32x4 areg[4], breg[4];
  auto load_chunk = [&](int kbase) {
    const int k = kbase + c4 * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = m0 + r0 + 32 * i;
      const int n = n0 + r0 + 32 * i;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      areg[i] = (m < M && k < D) ? *(const f32x4*)(G + m * D + k) : z;
      breg[i] = (n < Nr && k < D) ? *(const f32x4*)(R + (long)n * D + k) : z;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(f32x4*)&As[(r0 + 32 * i) * LDA + c4 * 4] = areg[i];
      *(f32x4*)&Bs[(r0 + 32 * i) * LDA + c4 * 4] = breg[i];
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nchunks = (D + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
    const float* arow = &As[(wave * 32 + lr) * LDA + 4 * h];
#pragma unroll
    for (int kq = 0; kq < KC / 8; ++kq) {
      const f32x4 a = *(const f32x4*)(arow + kq * 8);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const f32x4 b = *(const f32x4*)&Bs[(nb * 32 + lr) * LDA + kq * 8 + 4 * h];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  float rn[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + nb * 32 + lr;
    rn[nb] = n < Nr ? rinv[n] : 0.f;
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const long m = m0 + wave * 32 + row;
    const float gi = m < M ? ginv[m] : 0.f;
    float best = -__builtin_huge_valf();
    int bidx = 0x7FFFFFFF;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int n = n0 + nb * 32 + lr;
      if (n < Nr) {
        const float s = acc[nb][reg] * gi * rn[nb];
        if (s > best) {  // nb ascending => smaller index wins ties
          best = s;
          bidx = n;
        }
      }
    }
    // max over the 32 lanes that hold this row (xor offsets < 32 stay inside the half-wave)
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bidx, off);
      if (ob > best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
      }
    }
    if (lr == 0 && m < M && bidx != 0x7FFFFFFF) {
      const unsigned long long key = ((unsigned long long)f2ord(best) << 32) | (unsigned int)(0xFFFFFFFFu - (unsigned int)bidx);
      atomicMax(&packed[m], key);
    }
  }
}

__global__ __launch_bounds__(256) void cosine_finalize_kernel(const unsigned long long* __restrict__ packed, long M,
                                                              float tau, float* __restrict__ best,
                                                              int* __restrict__ arg, unsigned char* __restrict__ keep) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const unsigned long long key = packed[m];
  const float s = ord2f((unsigned int)(key >> 32));
  const int j = (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
  best[m] = s;
  arg[m] = j;
  keep[m] = s >= tau ? 1 : 0;
}

__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, long M, int D,
                                                           float* __restrict__ inv) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += p[i] * p[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) inv[row] = 1.0f / sqrtf(s);
}

// mean over R rows (sequential row order, like np.mean(axis=0) on a C-contiguous array), then
// thres = max_i ||mean - f_i||  (filter_faces_using_reference.py:86-99).  One workgroup.
__global__ __launch_bounds__(256) void l2_mean_thres_kernel(const float* __restrict__ ref, int R, int D,
                                                            float* __restrict__ mean, float* __restrict__ thres) {
  __shared__ float wmax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int d = tid; d < D; d += 256) {
    float s = 0.f;
    for (int i = 0; i < R; ++i) s += ref[(long)i * D + d];
    mean[d] = s / (float)R;
  }
  __threadfence_block();
  __syncthreads();
  float mx = 0.f;
  for (int i = wave; i < R; i += 4) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float t = mean[d] - ref[(long)i * D + d];
      s += t * t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    mx = fmaxf(mx, sqrtf(s));
  }
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  if (tid == 0) thres[0] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
}

// dist = ||e - mean||, keep = dist <= thres (filter_faces_using_reference.py:189).  One wave per row.
__global__ __launch_bounds__(256) void l2_filter_kernel(const float* __restrict__ E, long M, int D,
                                                        const float* __restrict__ mean, const float* __restrict__ thres,
                                                        float* __restrict__ dist, unsigned char* __restrict__ keep) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* p = E + row * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float t = p[d] - mean[d];
    s += t * t;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0) {
    const float dd = sqrtf(s);
    dist[row] = dd;
    keep[row] = dd <= thres[0] ? 1 : 0;
  }
}

}  // namespace

extern "C" {

int fp_row_inv_norm(const float* x, int64_t M, int D, float* inv_norm, void* stream) {
  if (!x || !inv_norm || M < 0 || D <= 0) return FP_ERR_INVALID_ARG;
  if (M == 0) return FP_OK;
  hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)fp_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, x,
                     (long)M, D, inv_norm);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_cosine_filter(const float* G, const float* ginv, int64_t M, const float* R, const float* rinv, int Nr, int D,
                     float tau, float* best, int32_t* arg, uint8_t* keep, uint64_t* packed, void* stream) {
  if (!G || !ginv || !R || !rinv || !best || !arg || !keep || !packed) return FP_ERR_INVALID_ARG;
  if (M < 0 || Nr <= 0 || D <= 0) return FP_ERR_INVALID_ARG;
  if (D % 4 || ((uintptr_t)G) % 16 || ((uintptr_t)R) % 16) return FP_ERR_ALIGNMENT;
  if (M == 0) return FP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(packed, 0, (size_t)M * sizeof(uint64_t), s) != hipSuccess) {
    fp_set_hip_error(hipGetLastError());
    return FP_ERR_LAUNCH;
  }
  dim3 grid((unsigned)fp_ceil_div(M, BM), (unsigned)fp_ceil_div(Nr, BN));
  hipLaunchKernelGGL(cosine_tile_kernel, grid, dim3(256), 0, s, G, ginv, (long)M, R, rinv, Nr, D,
                     (unsigned long long*)packed);
  FP_CHECK_LAUNCH();
  hipLaunchKernelGGL(cosine_finalize_kernel, dim3((unsigned)fp_ceil_div(M, 256)), dim3(256), 0, s,
                     (const unsigned long long*)packed, (long)M, tau, best, arg, keep);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_l2_mean_thres(const float* ref, int R, int D, float* out_mean, float* out_thres, void* stream) {
  if (!ref || !out_mean || !out_thres || R <= 0 || D <= 0) return FP_ERR_INVALID_ARG;
  hipLaunchKernelGGL(l2_mean_thres_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ref, R, D, out_mean, out_thres);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_l2_filter(const float* E, int64_t M, int D, const float* mean, const float* thres, float* dist, uint8_t* keep,
                 void* stream) {
  if (!E || !mean || !thres || !dist || !keep || M < 0 || D <= 0) return FP_ERR_INVALID_ARG;
  if (M == 0) return FP_OK;
  hipLaunchKernelGGL(l2_filter_kernel, dim3((unsigned)fp_ceil_div(M, 4)), dim3(256), 0, (hipStream_t)stream, E, (long)M,
                     D, mean, thres, dist, keep);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // extern "C"
