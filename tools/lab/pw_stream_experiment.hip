// pw.hip — streaming pointwise (1x1, stride 1) convolution for small K (gfx950).
//
// The 1x1 convs of Mobile-FaceNet (mobile_facenet.py:70-75: 64->128, 128->64, 128->256 ...), of the BlazeFace
// blocks that are too wide to fuse, and of YOLOv5-face (common.py Conv k=1) are skinny GEMMs: M = N*H*W rows is
// huge, K = Cin <= 128, so there are only 1-2 K-chunks per tile and the tile-at-a-time kernel of conv.hip spends
// most of its time in prologue / epilogue / barriers (measured 2.3 TB/s, 32 % MFMA utilisation at K = 64).
// Here:
//   * the packed weights of the block's N tile stay in LDS for the whole kernel (K*BN*4 <= 32 KiB);
//   * every WAVE owns 32-row tiles (NHWC dense => a tile's A panel is one contiguous byte range, loaded with
//     16-B fully coalesced accesses), walks them with a grid stride, and has a private LDS region for the A panel
//     and for the transposed epilogue: no __syncthreads after the weight staging, waves drift apart and overlap
//     each other's memory and MFMA phases;
//   * the next panel's loads are issued before the current panel's MFMAs (register prefetch across tiles);
//   * epilogue = conv.hip's vector epilogue: acc*scale+bias through LDS, then 16-B residual loads / stores.
// v_mfma_f32_32x32x2_f32 as in conv.hip (same fragment scheme, same packed weight layout, same results).
#include "common.h"

namespace {

struct PwArgs {
  const float* in;
  float* out;
  const float* res;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int K, Kpad, Cout, Npad, in_ld, out_ld, res_ld, res_C4, act, res_mode;
  long M, ntiles;
  int priv_floats;
};

constexpr int KCH = 64;  // K chunk held in the private A panel

__device__ __forceinline__ float pw_act(float v, int act, float slope) {
  switch (act) {
    case FP_ACT_RELU: return v > 0.f ? v : 0.f;
    case FP_ACT_PRELU: return v > 0.f ? v : v * slope;
    case FP_ACT_SILU: return v / (1.0f + expf(-v));
    default: return v;
  }
}

template <int NB>
__global__ __launch_bounds__(256, 2) void pw_stream_kernel(PwArgs p) {
  constexpr int BN = NB * 32;
  constexpr int PW = (NB % 2 == 0) ? 2 : 1;
  constexpr int LDO = PW * 32 + 4;
  constexpr int F4_PER_ROW = PW * 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * BN;
  float* Bs = smem;                                   // [Kpad/4][BN][4]
  float* Ap = smem + p.Kpad * BN + wave * p.priv_floats;  // wave-private: A panel [32][kc+4] / output staging

  // weights of this N tile -> LDS, once
  {
    const int nq = p.Kpad >> 2;
    for (int i = tid; i < nq * BN; i += 256) {
      const int q = i / BN, col = i - q * BN;
      const int n = n0 + col;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < p.Npad) v = *(const f32x4*)(p.w + ((long)q * p.Npad + n) * 4);
      *(f32x4*)&Bs[i * 4] = v;
    }
  }
  __syncthreads();

  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = p.scale ? p.scale[nn] : 1.f;
    bi[nb] = p.bias ? p.bias[nn] : 0.f;
  }
  const int nchunks = (p.Kpad + KCH - 1) / KCH;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 areg[8];

  // loads of chunk ch of tile t into areg (rows beyond M and columns beyond K read as zero)
  auto load_panel = [&](long t, int ch) {
    const int kbase = ch * KCH;
    const int kc = min(KCH, p.Kpad - kbase);
    const int kc4 = kc >> 2;
    const long m0 = t * 32;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      f32x4 v = z4;
      if (f < 32 * kc4) {
        const int r = f / kc4, k4 = f - r * kc4;
        const long m = m0 + r;
        const int k = kbase + k4 * 4;
        if (m < p.M && k < p.K) v = *(const f32x4*)(p.in + m * p.in_ld + k);
      }
      areg[j] = v;
    }
  };
  auto store_panel = [&](int ch) {
    const int kc = min(KCH, p.Kpad - ch * KCH);
    const int kc4 = kc >> 2, lda = kc + 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = lane + 64 * j;
      if (f < 32 * kc4) {
        const int r = f / kc4, k4 = f - r * kc4;
        *(f32x4*)&Ap[r * lda + k4 * 4] = areg[j];
      }
    }
  };

  const long tstride = (long)gridDim.x * 4;
  long t = (long)blockIdx.x * 4 + wave;
  if (t < p.ntiles) load_panel(t, 0);
  for (; t < p.ntiles; t += tstride) {
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

    for (int ch = 0; ch < nchunks; ++ch) {
      store_panel(ch);
      __builtin_amdgcn_wave_barrier();
      if (ch + 1 < nchunks) load_panel(t, ch + 1);
      else if (t + tstride < p.ntiles) load_panel(t + tstride, 0);
      const int kc = min(KCH, p.Kpad - ch * KCH);
      const int lda = kc + 4;
      const float* arow = &Ap[lr * lda + 4 * h];
      const int q0 = (ch * KCH) >> 2;
      for (int kq = 0; kq < (kc >> 3); ++kq) {
        const f32x4 a = *(const f32x4*)(arow + kq * 8);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const f32x4 b = *(const f32x4*)&Bs[((q0 + kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tt], b[tt], acc[nb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    // epilogue: PW 32-column blocks per pass through the private region, then 16-B accesses
    const long m0 = t * 32;
#pragma unroll
    for (int pass = 0; pass < NB / PW; ++pass) {
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        const int nb = pass * PW + q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          Ap[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
        }
      }
      __builtin_amdgcn_wave_barrier();
      const int ncol0 = n0 + pass * PW * 32;
#pragma unroll
      for (int j = 0; j < (32 * F4_PER_ROW) / 64; ++j) {
        const int f = lane + 64 * j;
        const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
        const long m = m0 + row;
        const int n = ncol0 + c4 * 4;
        if (m < p.M && n < p.Cout) {
          const f32x4 v = *(const f32x4*)&Ap[row * LDO + c4 * 4];
          f32x4 r = z4;
          if (p.res_mode != FP_RES_NONE && n < p.res_C4) r = *(const f32x4*)(p.res + m * p.res_ld + n);
          f32x4 sl = z4;
          if (p.act == FP_ACT_PRELU) sl = *(const f32x4*)(p.slope + n);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            o[e] = (p.res_mode == FP_RES_ADD_AFTER_ACT) ? pw_act(v[e], p.act, sl[e]) + r[e]
                                                        : pw_act(v[e] + r[e], p.act, sl[e]);
          *(f32x4*)(p.out + m * p.out_ld + n) = o;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

}  // namespace

// Eligibility + N-tile width of the streaming kernel for an op (0 = not eligible: conv_igemm handles it).
int fp_pw_stream_nb(const fp_op& op) {
  if (op.kind != FP_OP_CONV) return 0;
  if (op.KH != 1 || op.KW != 1 || op.stride != 1 || op.pad_t || op.pad_l) return 0;
  if (op.OH != op.H || op.OW != op.W || op.out_cmul != 1) return 0;
  const long HW = (long)op.H * op.W;
  if (op.in_ns != HW * op.in_ld || op.out_ns != HW * op.out_ld) return 0;   // dense row addressing
  if (op.Cin % 4 || op.in_ld % 4 || op.in_off % 4 || op.Cout % 4 || op.out_ld % 4 || op.out_off % 4) return 0;
  if (op.res_mode == FP_RES_POOL2_BEFORE_ACT) return 0;
  if (op.res_mode != FP_RES_NONE &&
      (op.res_ns != HW * op.res_ld || op.res_ld % 4 || op.res_off % 4 || fp_round_up(op.res_C, 4) > op.res_ld))
    return 0;
  if ((op.scale_off >= 0 && op.scale_off % 4) || (op.bias_off >= 0 && op.bias_off % 4) ||
      (op.slope_off >= 0 && op.slope_off % 4))
    return 0;
  const int Kpad = (int)fp_round_up(op.Cin, 8), Npad = (int)fp_round_up(op.Cout, 32);
  if (Kpad > 128) return 0;
  int NB = 32768 / (4 * Kpad) / 32;   // the weights of one N tile must fit 32 KiB of LDS
  if (NB > 4) NB = 4;
  if (NB > Npad / 32) NB = Npad / 32;
  if (NB < 1) return 0;
  if ((long)op.N * HW < 4096) return 0;   // too few tiles to fill the chip: the tile kernel is fine there
  return NB;
}

int fp_launch_pw_stream(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  const int NB = fp_pw_stream_nb(op);
  if (NB == 0) return FP_ERR_UNSUPPORTED;
  const long HW = (long)op.H * op.W;
  const int K = op.Cin, Kpad = (int)fp_round_up(K, 8), Npad = (int)fp_round_up(op.Cout, 32);
  const long M = (long)op.N * HW;
  PwArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  if (op.act == FP_ACT_PRELU && !a.slope) return FP_ERR_INVALID_ARG;
  a.K = K; a.Kpad = Kpad; a.Cout = op.Cout; a.Npad = Npad;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld;
  a.res_C4 = (int)fp_round_up(op.res_C, 4);
  a.act = op.act; a.res_mode = op.res_mode;
  a.M = M;
  a.ntiles = (M + 31) / 32;
  const int kc = Kpad < KCH ? Kpad : KCH;
  const int PW = (NB % 2 == 0) ? 2 : 1;
  const int priv_a = 32 * (kc + 4), priv_o = 32 * (PW * 32 + 4);
  a.priv_floats = priv_a > priv_o ? priv_a : priv_o;
  const size_t lds = 4 * ((size_t)Kpad * NB * 32 + 4 * (size_t)a.priv_floats);
  if (lds > 64 * 1024 + 8 * 1024) return FP_ERR_UNSUPPORTED;
  const int ny = fp_ceil_div(Npad, NB * 32);
  // persistent waves: ~2 resident blocks per CU per N tile column, each wave strides over the 32-row tiles
  long gx = (a.ntiles + 3) / 4;
  const long cap = (256L * 2 + ny - 1) / ny;
  if (gx > cap) gx = cap;
  dim3 grid((unsigned)gx, (unsigned)ny), block(256);
#define FP_PW_CASE(NBV)                                                                                          \
  case NBV:                                                                                                      \
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)pw_stream_kernel<NBV>,                           \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);        \
    hipLaunchKernelGGL((pw_stream_kernel<NBV>), grid, block, lds, s, a);                                         \
    break;
  switch (NB) {
    FP_PW_CASE(1)
    FP_PW_CASE(2)
    FP_PW_CASE(3)
    FP_PW_CASE(4)
    default: return FP_ERR_UNSUPPORTED;
  }
#undef FP_PW_CASE
  FP_CHECK_LAUNCH();
  return FP_OK;
}
