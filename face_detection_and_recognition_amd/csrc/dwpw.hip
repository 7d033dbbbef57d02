// dwpw.hip — fused depthwise 3x3 (+BN affine, +PReLU) -> pointwise 1x1 (+BN affine) [+ residual]  (gfx950).
//
// Mobile-FaceNet's Depth_Wise block (fde/modules/mobile_facenet/mobile_facenet.py:67-88) is
//     conv (1x1 expand, BN, PReLU) -> conv_dw (3x3 depthwise stride s, BN, PReLU) -> project (1x1, BN) [+ x].
// The expanded tensor (G = 2..4x the block width) is the fat one.  This kernel fuses conv_dw + project: the
// depthwise result goes to LDS instead of HBM and is consumed there by the MFMA projection, so per block-layer the
// G-channel tensor is read ONCE (by the depthwise taps) and never written back; the op-granular model
// counts dw in+out and project in+out.
//
// One workgroup = 128 consecutive output pixels.  The expanded channels are processed in chunks of 64:
//   phase 1  all lanes: depthwise for (P-pixel group, 4-channel group) items with branch-free clamped 16-B loads and
//            a sliding window (P = 4 when OW % 4 == 0, else 2 or 1), then x*s+b and PReLU, into At[128][64+4];
//   phase 2  4 waves x 32 rows: v_mfma_f32_32x32x2_f32 against the chunk of packed projection weights in LDS,
//            accumulating over the chunks in registers;
// then the conv.hip vector epilogue (acc*s+b through LDS, 16-B residual loads and stores).
#include "common.h"

namespace {

struct DwPwArgs {
  const float* in;
  float* out;
  const float* res;
  const float* dwp;  // [9*G dw weights][G scale][G bias][G slope]
  const float* pwp;  // [Kpad*Npad packed 1x1 weights][Cout scale][Cout bias]
  int N, H, W, OH, OW, G, Cout, stride;
  int in_ld, out_ld, res_ld;
  long in_ns;
  int Npad, OHW, has_res, has_slope;
  long M;
  int ntiles;
};

constexpr int TM = 128;
constexpr int KCH = 64;
constexpr int LDT = KCH + 4;

template <int NB, int P, int S>
__global__ __launch_bounds__(256, 2) void dwpw_kernel(DwPwArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WIN = (P - 1) * S + 3;
  constexpr int PW = (NB % 2 == 0) ? 2 : 1;
  constexpr int LDO = PW * 32 + 4;
  constexpr int F4_PER_ROW = PW * 8;
  // At [TM][LDT] | Bs [KCH/4][BN][4] | Ws [12][KCH]; the epilogue staging tile [TM][LDO] reuses At(+Bs)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;
  float* Bs = smem + TM * LDT;
  float* Ws = Bs + KCH * BN;
  static_assert(TM * LDO <= TM * LDT + KCH * BN, "epilogue staging must fit");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  int tile;
  {
    const int b = blockIdx.x, q = p.ntiles / 8, r = p.ntiles % 8, xcd = b & 7, k = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long m0 = (long)tile * TM;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nchunks = p.G / KCH;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int g0 = ch * KCH;
    // stage this chunk's projection weights and depthwise parameters
    for (int i = tid; i < (KCH / 4) * BN; i += 256) {
      const int q = i / BN, col = i - q * BN;
      f32x4 v = z;
      if (col < p.Npad) v = *(const f32x4*)(p.pwp + ((long)(g0 / 4 + q) * p.Npad + col) * 4);
      *(f32x4*)&Bs[i * 4] = v;
    }
    for (int i = tid; i < 12 * (KCH / 4); i += 256) {
      const int row = i / (KCH / 4), c4 = i - row * (KCH / 4);  // rows 0..8 taps, 9 scale, 10 bias, 11 slope
      f32x4 v = z;
      if (row < 11 || p.has_slope) v = *(const f32x4*)(p.dwp + (long)row * p.G + g0 + c4 * 4);
      *(f32x4*)&Ws[i * 4] = v;
    }
    __syncthreads();

    // phase 1: depthwise + affine + PReLU for this channel chunk
    for (int it = tid; it < (TM / P) * (KCH / 4); it += 256) {
      const int g = it >> 4, c4 = it & 15;   // KCH/4 == 16 channel groups per pixel group
      const int r = g * P;
      long m = m0 + r;
      m = m < p.M ? m : p.M - P;  // tail groups recompute the last pixels; they are never stored
      const unsigned mm = (unsigned)m;
      const unsigned img = mm / (unsigned)p.OHW;
      const unsigned rem = mm - img * (unsigned)p.OHW;
      const int oy = (int)(rem / (unsigned)p.OW), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
      const int c = c4 * 4;
      const float* ib = p.in + (long)img * p.in_ns + g0 + c;
      f32x4 a[P];
#pragma unroll
      for (int q = 0; q < P; ++q) a[q] = z;
      const int iy0 = oy * S - 1, ix0 = ox * S - 1;
      // all 3 x WIN window loads are issued before the first FMA (one memory round trip per item, not three)
      f32x4 x[3][WIN];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = iy0 + ky;
        const bool vy = (unsigned)iy < (unsigned)p.H;
        const float* rowp = ib + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
          const int ix = ix0 + j;
          const bool v = vy && ((unsigned)ix < (unsigned)p.W);
          const f32x4 t = *(const f32x4*)(rowp + (long)min(max(ix, 0), p.W - 1) * p.in_ld);
          x[ky][j] = v ? t : z;
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * KCH + c];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * KCH + c];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * KCH + c];
#pragma unroll
        for (int q = 0; q < P; ++q) a[q] += x[ky][q * S] * w0 + x[ky][q * S + 1] * w1 + x[ky][q * S + 2] * w2;
      }
      const f32x4 sc = *(const f32x4*)&Ws[9 * KCH + c];
      const f32x4 bi = *(const f32x4*)&Ws[10 * KCH + c];
      const f32x4 sl = *(const f32x4*)&Ws[11 * KCH + c];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        f32x4 v = a[q] * sc + bi;
        if (p.has_slope) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
        }
        *(f32x4*)&At[(r + q) * LDT + c] = v;
      }
    }
    __syncthreads();

    // phase 2: projection MFMAs for this chunk
    const float* arow = &At[(wave * 32 + lr) * LDT + 4 * h];
#pragma unroll
    for (int kq = 0; kq < KCH / 8; ++kq) {
      const f32x4 av = *(const f32x4*)(arow + kq * 8);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 bv = *(const f32x4*)&Bs[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();  // At / Bs / Ws are rewritten by the next chunk (or by the epilogue)
  }

  // epilogue: acc*scale+bias through LDS, then 16-B residual loads and stores (rows are dense: out + m*out_ld)
  const float* pscale = p.pwp + (long)(p.G) * p.Npad;   // Kpad == G (multiple of 64)
  const float* pbias = pscale + ((p.Cout + 3) & ~3);
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = pscale[nn];
    bi[nb] = pbias[nn];
  }
#pragma unroll
  for (int pass = 0; pass < NB / PW; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const int nb = pass * PW + q;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        smem[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
      }
    }
    __syncthreads();
    const int ncol0 = pass * PW * 32;
    for (int f = tid; f < TM * F4_PER_ROW; f += 256) {
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      const long m = m0 + row;
      const int n = ncol0 + c4 * 4;
      if (m >= p.M || n >= p.Cout) continue;
      f32x4 v = *(const f32x4*)&smem[row * LDO + c4 * 4];
      if (p.has_res) v += *(const f32x4*)(p.res + m * p.res_ld + n);
      *(f32x4*)(p.out + m * p.out_ld + n) = v;
    }
  }
}

}  // namespace

int fp_launch_dwpw(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  // op: KH = KW = 3, pad 1, stride 1|2, Cin = G (multiple of 64), Cout multiple of 4 and <= 128;
  // w_off -> depthwise block, slope_off -> pointwise block; act = FP_ACT_PRELU when the depthwise has a PReLU;
  // res_mode = FP_RES_ADD_AFTER_ACT for the residual variant.
  if (op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1 || (op.stride != 1 && op.stride != 2))
    return FP_ERR_UNSUPPORTED;
  if (op.Cin % KCH || op.Cout % 4 || op.Cout > 128 || op.Cout <= 0) return FP_ERR_UNSUPPORTED;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.w_off % 4 || op.slope_off % 4)
    return FP_ERR_ALIGNMENT;
  const long OHW = (long)op.OH * op.OW;
  if (op.out_cmul != 1 || op.out_ns != OHW * op.out_ld) return FP_ERR_UNSUPPORTED;
  if (op.OH != (op.H + 2 - 3) / op.stride + 1 || op.OW != (op.W + 2 - 3) / op.stride + 1) return FP_ERR_INVALID_ARG;
  const bool has_res = op.res_mode != FP_RES_NONE;
  if (has_res && (op.res_mode != FP_RES_ADD_AFTER_ACT || op.res_ns != OHW * op.res_ld || op.res_ld % 4 || op.res_off % 4 ||
                  op.res_C < op.Cout))
    return FP_ERR_UNSUPPORTED;
  DwPwArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = has_res ? arena + op.res_off : nullptr;
  a.dwp = weights + op.w_off;
  a.pwp = weights + op.slope_off;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.G = op.Cin; a.Cout = op.Cout; a.stride = op.stride;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.in_ns = op.in_ns;
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.OHW = (int)OHW;
  a.has_res = has_res ? 1 : 0;
  a.has_slope = op.act == FP_ACT_PRELU ? 1 : 0;
  a.M = (long)op.N * OHW;
  if (a.M >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  a.ntiles = fp_ceil_div(a.M, TM);
  const int NB = a.Npad / 32;
  const int P = (op.OW % 4 == 0) ? 4 : (op.OW % 2 == 0) ? 2 : 1;
  dim3 grid((unsigned)a.ntiles), block(256);
  const size_t lds = 4 * ((size_t)TM * LDT + (size_t)KCH * NB * 32 + 12 * KCH);
#define FP_DWPW_LAUNCH(NBV, PV, SV)                                                                        \
  do {                                                                                                     \
    if (lds > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute((const void*)dwpw_kernel<NBV, PV, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                 \
    hipLaunchKernelGGL((dwpw_kernel<NBV, PV, SV>), grid, block, lds, s, a);                                \
  } while (0)
#define FP_DWPW_P(NBV, SV)                                 \
  if (P == 4) FP_DWPW_LAUNCH(NBV, 4, SV);                  \
  else if (P == 2) FP_DWPW_LAUNCH(NBV, 2, SV);             \
  else FP_DWPW_LAUNCH(NBV, 1, SV);
#define FP_DWPW_S(NBV)                                     \
  if (op.stride == 1) { FP_DWPW_P(NBV, 1) } else { FP_DWPW_P(NBV, 2) }
  switch (NB) {
    case 1: FP_DWPW_S(1) break;
    case 2: FP_DWPW_S(2) break;
    case 3: FP_DWPW_S(3) break;
    case 4: FP_DWPW_S(4) break;
    default: return FP_ERR_UNSUPPORTED;
  }
#undef FP_DWPW_S
#undef FP_DWPW_P
#undef FP_DWPW_LAUNCH
  FP_CHECK_LAUNCH();
  return FP_OK;
}
