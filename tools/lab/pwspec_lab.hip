// Lab harness: the wave-specialised pointwise kernel with s_memtime stamps (cdna_hip_programming.md "In-kernel
// stamps").  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 pwspec_lab.hip -o pwspec_lab ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { FP_ACT_NONE = 0, FP_ACT_RELU = 1, FP_ACT_PRELU = 2, FP_ACT_SILU = 3 };
enum { FP_RES_NONE = 0, FP_RES_ADD_BEFORE_ACT = 1, FP_RES_ADD_AFTER_ACT = 2, FP_RES_POOL2_BEFORE_ACT = 3 };
#define NSTEP 16
// stamps[(block*8 + wave)*NSTEP*3 + step*3 + k] for block 0..7, first NSTEP steps, lane 0 only
#define STAMP(k)                                                                                     \
  do {                                                                                               \
    if (stamps && blockIdx.x < 8 && t < NSTEP && (threadIdx.x & 63) == 0) {                          \
      unsigned long long tt_;                                                                        \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_) :: "memory");                  \
      stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * NSTEP + t) * 3 + (k)] = tt_;                   \
    }                                                                                                \
  } while (0)
namespace {

struct PwsArgs {
  const float* in;
  float* out;
  const float* res;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int K, Kpad, Cout, Npad, in_ld, out_ld, res_ld, res_C4, act, res_mode, ntiles_n;
  long M, ntiles_m;
};

constexpr int BMS = 64;

__device__ __forceinline__ float pws_act(float v, int act, float slope) {
  switch (act) {
    case FP_ACT_RELU: return v > 0.f ? v : 0.f;
    case FP_ACT_PRELU: return v > 0.f ? v : v * slope;
    case FP_ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}

// NBW = 32-column accumulators per MFMA wave; the block's N tile is BN = NBW*64 columns.
template <int NBW>
__global__ __launch_bounds__(512, 2) void pwspec_kernel(PwsArgs p, unsigned long long* stamps) {
  constexpr int BN = NBW * 64;
  constexpr int LDO = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDA = p.Kpad + 4;
  float* Bs = smem;                               // [Kpad/4][BN][4]
  float* As = Bs + p.Kpad * BN;                   // [2][BMS][LDA]
  float* Os = As + 2 * BMS * LDA;                 // [2][BMS][LDO]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mfma_wave = wave < 4;
  const int ny = p.ntiles_n;
  const int n0 = (int)(blockIdx.x % ny) * BN;
  const long mt0 = blockIdx.x / ny, mstride = gridDim.x / ny;
  const long T = mt0 < p.ntiles_m ? (p.ntiles_m - mt0 + mstride - 1) / mstride : 0;   // tiles of this block
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  // resident weights of this N tile
  for (int i = tid; i < (p.Kpad >> 2) * BN; i += 512) {
    const int q = i / BN, col = i - q * BN;
    const int n = n0 + col;
    f32x4 v = z4;
    if (n < p.Npad) v = *(const f32x4*)(p.w + ((long)q * p.Npad + n) * 4);
    *(f32x4*)&Bs[i * 4] = v;
  }

  const int K4 = p.Kpad >> 2;
  // memory roles: waves 4-5 load panels (their vmcnt queue holds only loads, so waiting for a panel never waits for
  // stores), waves 6-7 run the epilogue (their queue holds the stores and the residual loads)
  const bool loader_wave = wave == 4 || wave == 5;
  const int mtid = loader_wave ? tid - 256 : tid - 384;  // index inside the 128-thread role group
  f32x4 areg[16];
  auto load_panel = [&](long t) {   // memory waves: panel of tile index t (this block's t-th tile) -> registers
    const long m0 = (mt0 + t * mstride) * BMS;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int f = mtid + 128 * j;
      f32x4 v = z4;
      if (f < BMS * K4) {
        const int row = f / K4, k4 = f - row * K4;
        const long m = m0 + row;
        if (m < p.M && k4 * 4 < p.K) v = *(const f32x4*)(p.in + m * p.in_ld + k4 * 4);
      }
      areg[j] = v;
    }
  };
  auto store_panel = [&](int buf) {
    float* A = As + buf * BMS * LDA;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int f = mtid + 128 * j;
      if (f < BMS * K4) {
        const int row = f / K4, k4 = f - row * K4;
        *(f32x4*)&A[row * LDA + k4 * 4] = areg[j];
      }
    }
  };

  // prologue: panel 0 into A[0], panel 1 in flight
  if (loader_wave && T > 0) {
    load_panel(0);
    store_panel(0);
    if (T > 1) load_panel(1);
  }
  // per-column epilogue constants of the MFMA waves
  const int lr = lane & 31, h = lane >> 5;
  const int wr = (wave & 1) * 32, wc = ((wave >> 1) & 1) * (BN / 2);
  float sc[NBW], bi[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) {
    const int n = n0 + wc + nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = p.scale ? p.scale[nn] : 1.f;
    bi[nb] = p.bias ? p.bias[nn] : 0.f;
  }
  // epilogue waves: a lane's column group is the same for all of its rows (128 % (BN/4) == 0)
  const int ec4 = mtid % (BN / 4), erow0 = mtid / (BN / 4);
  const int en = n0 + ec4 * 4;
  f32x4 esl = z4;
  if (!mfma_wave && !loader_wave && p.act == FP_ACT_PRELU && en < p.Cout) esl = *(const f32x4*)(p.slope + en);
  if (p.act == FP_ACT_NONE) esl = f32x4{1.f, 1.f, 1.f, 1.f};
  const float nfloor = p.act == FP_ACT_RELU ? 0.f : -__builtin_inff();
  const bool silu = p.act == FP_ACT_SILU, after = p.res_mode == FP_RES_ADD_AFTER_ACT;
  __syncthreads();

  for (long t = 0; t <= T; ++t) {   // T compute steps + 1 drain step for the last epilogue
    STAMP(0);
    if (mfma_wave) {
      if (t < T) {
        const float* A = As + (t & 1) * BMS * LDA;
        float* O = Os + (t & 1) * BMS * LDO;
        f32x16 acc[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
        const float* arow = &A[(wr + lr) * LDA + 4 * h];
        for (int kq = 0; kq < (p.Kpad >> 3); ++kq) {
          const f32x4 a = *(const f32x4*)(arow + kq * 8);
#pragma unroll
          for (int nb = 0; nb < NBW; ++nb) {
            const f32x4 b = *(const f32x4*)&Bs[((kq * 2 + h) * BN + wc + nb * 32 + lr) * 4];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], b[tt], acc[nb], 0, 0, 0);
          }
        }
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wr + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            O[row * LDO + wc + nb * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
          }
      }
    } else if (loader_wave) {
      if (t + 1 < T) store_panel((int)((t + 1) & 1));   // loaded one step ago
      if (t + 2 < T) load_panel(t + 2);
    } else if (t >= 1) {                                // epilogue of tile t-1
      const float* O = Os + ((t - 1) & 1) * BMS * LDO;
      const long m0 = (mt0 + (t - 1) * mstride) * BMS;
      constexpr int ROWS_PER_IT = 128 / (BN / 4);       // rows covered by the 128 lanes per iteration
      constexpr int NIT = BMS / ROWS_PER_IT;
      // all loads of the epilogue (residual rows) are issued BEFORE the first store: vmcnt counts loads and stores
      // in one in-order queue, so a load inside the store loop makes every iteration wait for all earlier stores
      f32x4 rr[NIT];
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const long m = m0 + erow0 + j * ROWS_PER_IT;
        rr[j] = z4;
        if (p.res_mode != FP_RES_NONE && m < p.M && en < p.res_C4) rr[j] = *(const f32x4*)(p.res + m * p.res_ld + en);
      }
#pragma unroll
      for (int j = 0; j < NIT; ++j) {
        const int row = erow0 + j * ROWS_PER_IT;
        const long m = m0 + row;
        if (m < p.M && en < p.Cout) {
          const f32x4 v = *(const f32x4*)&O[row * LDO + ec4 * 4];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
          {
            // branch-free none/relu/prelu: x>0 ? x : max(x*slope, floor); SiLU stays a (uniform) branch
            const float pre = after ? 0.f : rr[j][e], post = after ? rr[j][e] : 0.f;
            const float x = v[e] + pre;
            float y = x > 0.f ? x : fmaxf(x * esl[e], nfloor);
            if (silu) y = x / (1.0f + expf(-x));
            o[e] = y + post;
          }
          *(f32x4*)(p.out + m * p.out_ld + en) = o;
        }
      }
    }
    STAMP(1);
    // Step barrier.  Only LDS traffic has to be complete here (A panel / output tile hand-over); a plain
    // __syncthreads() would also drain vmcnt, i.e. wait for the prefetch loads and the epilogue's global stores.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(2);
  }
}

}  // namespace
int main(int argc, char** argv) {
  const int K = 64, N = 128;
  const long M = 1088L * 56 * 56;
  const int NBW = 2, BN = 128, Kpad = 64, Npad = 128;
  float *in, *out, *w, *scale, *bias, *slope;
  hipMalloc(&in, M * K * 4); hipMalloc(&out, M * N * 4); hipMalloc(&w, Kpad * Npad * 4);
  hipMalloc(&scale, N * 4); hipMalloc(&bias, N * 4); hipMalloc(&slope, N * 4);
  hipMemset(in, 0x3c, M * K * 4); hipMemset(w, 0x3c, Kpad * Npad * 4);
  hipMemset(scale, 0x3c, N * 4); hipMemset(bias, 0, N * 4); hipMemset(slope, 0x3c, N * 4);
  unsigned long long* stamps;
  hipMalloc(&stamps, 8 * 8 * NSTEP * 3 * 8); hipMemset(stamps, 0, 8 * 8 * NSTEP * 3 * 8);
  PwsArgs a;
  a.in = in; a.out = out; a.res = nullptr; a.w = w; a.scale = scale; a.bias = bias; a.slope = slope;
  a.K = K; a.Kpad = Kpad; a.Cout = N; a.Npad = Npad; a.in_ld = K; a.out_ld = N; a.res_ld = 0; a.res_C4 = 0;
  a.act = FP_ACT_PRELU; a.res_mode = FP_RES_NONE; a.ntiles_n = 1; a.M = M; a.ntiles_m = (M + 63) / 64;
  const size_t lds = 4 * ((size_t)Kpad * BN + 2 * 64 * (Kpad + 4) + 2 * 64 * (BN + 4));
  hipFuncSetAttribute((const void*)pwspec_kernel<NBW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((pwspec_kernel<NBW>), dim3(256), dim3(512), lds, 0, a, (unsigned long long*)nullptr);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((pwspec_kernel<NBW>), dim3(256), dim3(512), lds, 0, a, (unsigned long long*)nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("pwspec K=%d N=%d M=%ld : %.1f us per launch (lds %zu B) err=%s\n", K, N, M, ms * 100, lds, hipGetErrorString(hipGetLastError()));
  hipLaunchKernelGGL((pwspec_kernel<NBW>), dim3(256), dim3(512), lds, 0, a, stamps);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(8 * 8 * NSTEP * 3);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  for (int b = 0; b < 2; ++b)
    for (int wv = 0; wv < 8; ++wv) {
      printf("blk %d wave %d (%s):", b, wv, wv < 4 ? "mfma" : (wv < 6 ? "load" : "epi"));
      for (int t = 1; t < 10; ++t) {
        const unsigned long long* s = &h[((b * 8 + wv) * NSTEP + t) * 3];
        printf("  [work %llu wait %llu]", s[1] - s[0], s[2] - s[1]);
      }
      printf("\n");
    }
  return 0;
}
