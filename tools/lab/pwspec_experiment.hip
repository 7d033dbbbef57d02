// pwspec.hip — wave-specialised persistent pointwise (1x1, stride 1) convolution for K <= 128 (gfx950).
//
// Why: the tile-at-a-time kernel (conv.hip) on the Mobile-FaceNet expand / project convs measured 45 % MFMA-pipe
// and 40 % HBM utilisation with every wave ~40 % of its life in s_waitcnt (profiles/r01_sq_counters.md): each wave
// serialises "wait for the panel -> MFMA -> epilogue -> stores", and 3 waves per SIMD cannot cover that.
// Here the two kinds of work live in different waves of one 512-thread workgroup:
//   waves 0-3  (one per SIMD)  MFMA waves: ds_read fragments -> v_mfma_f32_32x32x2_f32 -> acc*scale+bias -> LDS.
//                              They never touch global memory, so they never wait for HBM.
//   waves 4-7  (one per SIMD)  memory waves: global -> registers -> LDS for the A panel two tiles ahead, and the
//                              epilogue of the previous tile (LDS -> +residual, activation -> 16-B global stores).
// LDS: packed weights of the block's N tile (resident), A panels x2, output tiles x2.  One s_barrier per 64-row
// tile separates the steps of the 2-deep pipeline:
//   step t:  MFMA waves   tile t:   A[t%2] x B -> O[t%2]
//            memory waves store panel t+1 into A[(t+1)%2]; issue loads of panel t+2; epilogue of tile t-1 from O[(t-1)%2]
// Per 64x128 tile at K = 64 an MFMA wave issues 64 MFMAs (4096 cycles) while the memory waves move 48 KiB
// (11.7 B/clk/CU ~ the CU's share of HBM): both pipes are meant to run near their limits.
#include "common.h"

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
__global__ __launch_bounds__(512, 2) void pwspec_kernel(PwsArgs p) {
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
  __syncthreads();

  for (long t = 0; t <= T; ++t) {   // T compute steps + 1 drain step for the last epilogue
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
#pragma unroll
      for (int j = 0; j < BMS / ROWS_PER_IT; ++j) {
        const int row = erow0 + j * ROWS_PER_IT;
        const long m = m0 + row;
        if (m < p.M && en < p.Cout) {
          const f32x4 v = *(const f32x4*)&O[row * LDO + ec4 * 4];
          f32x4 r = z4;
          if (p.res_mode != FP_RES_NONE && en < p.res_C4) r = *(const f32x4*)(p.res + m * p.res_ld + en);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (p.res_mode == FP_RES_ADD_AFTER_ACT) ? pws_act(v[e], p.act, esl[e]) + r[e]
                                                        : pws_act(v[e] + r[e], p.act, esl[e]);
          *(f32x4*)(p.out + m * p.out_ld + en) = o;
        }
      }
    }
    // Step barrier.  Only LDS traffic has to be complete here (A panel / output tile hand-over); a plain
    // __syncthreads() would also drain vmcnt, i.e. wait for the prefetch loads and the epilogue's global stores.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
}

}  // namespace

// N-tile width (in 64-column units) of the wave-specialised kernel for an op, 0 = not eligible.
int fp_pwspec_nbw(const fp_op& op) {
  if (op.kind != FP_OP_CONV) return 0;
  if (op.KH != 1 || op.KW != 1 || op.stride != 1 || op.pad_t || op.pad_l) return 0;
  if (op.OH != op.H || op.OW != op.W || op.out_cmul != 1) return 0;
  const long HW = (long)op.H * op.W;
  if (op.in_ns != HW * op.in_ld || op.out_ns != HW * op.out_ld) return 0;
  if (op.Cin % 4 || op.in_ld % 4 || op.in_off % 4 || op.Cout % 4 || op.out_ld % 4 || op.out_off % 4) return 0;
  if (op.res_mode == FP_RES_POOL2_BEFORE_ACT) return 0;
  if (op.res_mode != FP_RES_NONE &&
      (op.res_ns != HW * op.res_ld || op.res_ld % 4 || op.res_off % 4 || fp_round_up(op.res_C, 4) > op.res_ld))
    return 0;
  if ((op.scale_off >= 0 && op.scale_off % 4) || (op.bias_off >= 0 && op.bias_off % 4) ||
      (op.slope_off >= 0 && op.slope_off % 4))
    return 0;
  const int Kpad = (int)fp_round_up(op.Cin, 8), Npad = (int)fp_round_up(op.Cout, 32);
  if (Kpad > 128 || Kpad < 32 || Npad % 64) return 0;
  if ((long)op.N * HW < 65536) return 0;          // needs enough 64-row tiles to fill 256 persistent workgroups
  int nbw = (Npad % 128 == 0) ? 2 : 1;
  auto lds = [&](int nb) { return 4 * ((size_t)Kpad * nb * 64 + 2 * (size_t)BMS * (Kpad + 4) + 2 * (size_t)BMS * (nb * 64 + 4)); };
  if (lds(nbw) > 160 * 1024) nbw = 1;
  if (lds(nbw) > 160 * 1024) return 0;
  if (Npad / (nbw * 64) > 2) return 0;            // more than two N tiles would re-read the A panel too often
  return nbw;
}

int fp_launch_pwspec(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  const int nbw = fp_pwspec_nbw(op);
  if (!nbw) return FP_ERR_UNSUPPORTED;
  PwsArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  if (op.act == FP_ACT_PRELU && !a.slope) return FP_ERR_INVALID_ARG;
  a.K = op.Cin;
  a.Kpad = (int)fp_round_up(op.Cin, 8);
  a.Cout = op.Cout;
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.res_C4 = (int)fp_round_up(op.res_C, 4);
  a.act = op.act; a.res_mode = op.res_mode;
  a.M = (long)op.N * op.H * op.W;
  a.ntiles_m = (a.M + BMS - 1) / BMS;
  a.ntiles_n = a.Npad / (nbw * 64);
  const size_t lds = 4 * ((size_t)a.Kpad * nbw * 64 + 2 * (size_t)BMS * (a.Kpad + 4) + 2 * (size_t)BMS * (nbw * 64 + 4));
  const int per_cu = (2 * lds <= 160 * 1024) ? 2 : 1;
  long g = 256L * per_cu / a.ntiles_n * a.ntiles_n;
  if (g < a.ntiles_n) g = a.ntiles_n;
  const long maxb = a.ntiles_m * a.ntiles_n;
  if (g > maxb) g = maxb;
  dim3 grid((unsigned)g), block(512);
  if (nbw == 2) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pwspec_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pwspec_kernel<2>), grid, block, lds, s, a);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)pwspec_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((pwspec_kernel<1>), grid, block, lds, s, a);
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}
