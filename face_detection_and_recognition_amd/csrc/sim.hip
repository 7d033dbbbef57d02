// sim.hip — similarity filter kernels (gfx950).
//
//  * cosine filter (cosine formula of extract_and_label_faces_from_dataset.py:106):
//      best[g] = max_j <G[g], R[j]> / (|G[g]| |R[j]|),  arg, keep = best >= tau.
//    S = G R^T is a plain fp32 GEMM (K = D), computed tile by tile on v_mfma_f32_32x32x2_f32 and reduced
//    in the epilogue (row max over the tile with wavefront shuffles, then one 64-bit atomicMax per row and
//    tile) so the M x Nr score matrix is never written to HBM.
//  * L2-to-class-mean filter, the reference's actual semantics
//    (similar_face_filtering/filter_faces_using_reference.py:71-100,183-197).
#include "split.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

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


// ---------------------------------------------------------------------------------------------------------------------
// The same filter with S = G R^T on the bf16 matrix cores (fp32-equivalent split arithmetic, split.h): the fp32 kernel above
// reaches 117 TFLOP/s = 74 % of the fp32 MFMA peak; six bf16 products per fp32 product have a peak of 417.
//   * R is split ONCE into three bf16 planes [D / 32][3][Npad][32] (split3_rows_kernel; the reference set of a filter run is
//     fixed) -- the B operand, streamed slab by slab through LDS by LDS-DMA, double-buffered, one barrier per slab;
//   * a workgroup = 4 waves x (MT x 16) gallery rows x 128 reference columns; a wave's A slab goes global -> registers -> split
//     (as pwx6_kernel); column chunks of one row tile are adjacent in the grid, so the gallery rows are re-read from L2;
//   * swapped operands: a lane holds 4 consecutive COLUMNS of one gallery row per accumulator -> the row max over the lane's
//     32 columns is plain VALU, then two shuffles across the four k-groups, one 64-bit atomicMax per row and workgroup.
__global__ __launch_bounds__(256) void split3_rows_kernel(const float* __restrict__ R, int Nr, int D, int Npad,
                                                          unsigned short* __restrict__ out) {
  // one thread per (row n < Npad, 8 consecutive k): out[((k / 32) * 3 + pl) * Npad + n][k % 32]
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int k8 = D / 8;
  if (i >= (long)Npad * k8) return;
  const int n = (int)(i / k8), k = (int)(i - (long)n * k8) * 8;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
  if (n < Nr) {
    a = *(const f32x4*)(R + (long)n * D + k);
    b = *(const f32x4*)(R + (long)n * D + k + 4);
  }
  const fp_frag3 f = fp_split8(a, b);
  unsigned short* o = out + ((long)((k / 32) * 3) * Npad + n) * 32 + (k % 32);
  *(u32x4*)o = f.h;
  *(u32x4*)(o + (long)Npad * 32) = f.m;
  *(u32x4*)(o + 2L * Npad * 32) = f.l;
}

template <int MT>
__global__ __launch_bounds__(256, MT == 4 ? 2 : 3) void cosine_x6_kernel(const float* __restrict__ G, const float* __restrict__ ginv,
                                                                          long M, const unsigned short* __restrict__ R3,
                                                                          const float* __restrict__ rinv, int Nr, int Npad, int D,
                                                                          unsigned long long* __restrict__ packed) {
  constexpr int NT16 = 8, NC = 128, SLAB = 3 * NC * 32, XBM = 4 * MT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Bl = (unsigned short*)smem_raw;        // [2][3][NC][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nchunk = Npad / NC;
  const unsigned vb = fp_xcd_block();                    // the reference chunks of one gallery row tile on ONE XCD (gallery rows from its L2)
  const int chunk = vb % nchunk;
  const long row0 = (long)(vb / nchunk) * XBM + wave * (MT * 16);
  const int c0 = chunk * NC;
  const int KS = D / 32;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int ks) {
    unsigned char* dst = (unsigned char*)(Bl + (ks & 1) * SLAB);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned char* src = (const unsigned char*)(R3 + ((long)(ks * 3 + pl) * Npad + c0) * 32) + lane * 16;
#pragma unroll
      for (int j = 0; j < NT16 / 4; ++j) {
        const int c = j * 4 + wave;
        __builtin_amdgcn_global_load_lds((gbl_ptr)(src + c * 1024), (lds_ptr)(dst + (pl * NC * 32 + c * 512) * 2), 16, 0, 0);
      }
    }
  };
  const float* arow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    long r = row0 + 16 * t + l15;
    r = r < M ? r : M - 1;
    arow[t] = G + r * D + 8 * q;
  }
  f32x4 araw[MT][2];
  auto load_a = [&](int ks) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      araw[t][0] = *(const f32x4*)(arow[t] + 32 * ks);
      araw[t][1] = *(const f32x4*)(arow[t] + 32 * ks + 4);
    }
  };
  f32x4 acc[MT][NT16];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT16; ++n) acc[t][n] = z;

  stage(0);
  load_a(0);
  for (int ks = 0; ks < KS; ++ks) {
    fp_frag3 af[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) af[t] = fp_split8(araw[t][0], araw[t][1]);
    __syncthreads();
    if (ks + 1 < KS) {
      stage(ks + 1);
      load_a(ks + 1);
    }
    const unsigned short* Bc = Bl + (ks & 1) * SLAB + (l15 * 32 + 8 * q);
    fp_frag3 bf[2];
    auto ldb = [&](int n, fp_frag3& b) {
      b.h = *(const u32x4*)(Bc + n * 512);
      b.m = *(const u32x4*)(Bc + NC * 32 + n * 512);
      b.l = *(const u32x4*)(Bc + 2 * NC * 32 + n * 512);
    };
    ldb(0, bf[0]);
#pragma unroll
    for (int n = 0; n < NT16; ++n) {
      if (n + 1 < NT16) ldb(n + 1, bf[(n + 1) & 1]);
      const fp_frag3& b = bf[n & 1];
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][n] = fp_mfma_x6(b.h, b.m, b.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
    }
  }

  // epilogue: lane = gallery row 16 t + l15, reference columns c0 + 16 n + 4 q + i
  f32x4 rn[NT16];
#pragma unroll
  for (int n = 0; n < NT16; ++n) {
    const int col = c0 + 16 * n + 4 * q;
    if (col + 3 < Nr) {
      rn[n] = *(const f32x4*)(rinv + col);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) rn[n][i] = col + i < Nr ? rinv[col + i] : 0.f;
    }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const long m = row0 + 16 * t + l15;
    const float gi = m < M ? ginv[m] : 0.f;
    float best = -__builtin_huge_valf();
    int bidx = 0x7FFFFFFF;
#pragma unroll
    for (int n = 0; n < NT16; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = c0 + 16 * n + 4 * q + i;
        if (col < Nr) {
          const float sv = acc[t][n][i] * gi * rn[n][i];
          if (sv > best || (sv == best && col < bidx)) {
            best = sv;
            bidx = col;
          }
        }
      }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {                  // the four k-group lanes of this row
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bidx, off);
      if (ob > best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
      }
    }
    if (q == 0 && m < M && bidx != 0x7FFFFFFF) {
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


size_t fp_split3_bytes(int Nr, int D) { return (size_t)(D / 32) * 3 * (size_t)fp_round_up(Nr, 128) * 32 * 2; }

int fp_split3_rows(const float* R, int Nr, int D, void* out, void* stream) {
  if (!R || !out || Nr <= 0 || D <= 0) return FP_ERR_INVALID_ARG;
  if (D % 32 || ((uintptr_t)R) % 16 || ((uintptr_t)out) % 16) return FP_ERR_ALIGNMENT;
  const int Npad = (int)fp_round_up(Nr, 128);
  const long items = (long)Npad * (D / 8);
  hipLaunchKernelGGL(split3_rows_kernel, dim3((unsigned)fp_ceil_div(items, 256)), dim3(256), 0, (hipStream_t)stream, R, Nr, D, Npad,
                     (unsigned short*)out);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_cosine_filter_x6(const float* G, const float* ginv, int64_t M, const void* R3, const float* rinv, int Nr, int D,
                        float tau, float* best, int32_t* arg, uint8_t* keep, uint64_t* packed, void* stream) {
  if (!G || !ginv || !R3 || !rinv || !best || !arg || !keep || !packed) return FP_ERR_INVALID_ARG;
  if (M < 0 || Nr <= 0 || D <= 0) return FP_ERR_INVALID_ARG;
  if (D % 32 || ((uintptr_t)G) % 16 || ((uintptr_t)R3) % 16 || ((uintptr_t)rinv) % 16) return FP_ERR_ALIGNMENT;
  if (M == 0) return FP_OK;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(packed, 0, (size_t)M * sizeof(uint64_t), s) != hipSuccess) {
    fp_set_hip_error(hipGetLastError());
    return FP_ERR_LAUNCH;
  }
  const int Npad = (int)fp_round_up(Nr, 128);
  const long nchunk = Npad / 128;
  constexpr int lds = 2 * 3 * 128 * 32 * 2;
  // 256-row tiles (two workgroups per CU) when they fill several rounds of workgroups, else 128-row tiles (three per CU)
  const long big = (M + 255) / 256 * nchunk;
  if (big >= 2048) {
    if (big >= (1L << 31)) return FP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((cosine_x6_kernel<4>), dim3((unsigned)big), dim3(256), lds, s, G, ginv, (long)M, (const unsigned short*)R3, rinv,
                       Nr, Npad, D, (unsigned long long*)packed);
  } else {
    const long tiles = (M + 127) / 128 * nchunk;
    hipLaunchKernelGGL((cosine_x6_kernel<2>), dim3((unsigned)tiles), dim3(256), lds, s, G, ginv, (long)M, (const unsigned short*)R3, rinv,
                       Nr, Npad, D, (unsigned long long*)packed);
  }
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
