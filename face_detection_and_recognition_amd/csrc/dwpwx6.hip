// dwpwx6.hip — depthwise 3x3 (stride 1 / 2, + BN [+ PReLU]) -> 1x1 (+ BN [+ SiLU]) [-> ShuffleV2 cat + channel_shuffle] with
// the 1x1 on the bf16 matrix cores (fp32-equivalent split arithmetic, split.h) — gfx950.
//
// The op of YOLOv5n-face's ShuffleV2 blocks (y5/models/common.py:127-176: branch1 = dw stride 2 -> 1x1, branch2 tail =
// dw -> 1x1 -> cat + channel_shuffle).  dwpw_kernel (dwpw.hip) keeps the depthwise result in LDS, but its 1x1 is an fp32
// MFMA that shares the vector ALU with the depthwise FMAs (25-50 TFLOP/s, 278 us for 128 -> 128 on 40x40 x 256 images
// against 155 us of HBM time).  Same streaming structure here -- a tile is 128 consecutive output pixels, the depthwise
// windows come straight from global memory (their nine-fold reuse is L1 / L2's business) -- with
//   * 32-channel chunks: the depthwise values of a chunk are split into three bf16 planes as they are produced and go to an
//     LDS tile [3][128 pixels][32 + 8] (bf16), the chunk's weight slab [3][N][32] arrives by LDS-DMA while they are computed;
//   * wave w owns pixels 32 w .. 32 w + 31 of the tile (two 16-pixel MFMA tiles) for all N output channels:
//     accumulators [2][N / 16][4], operands swapped (D^T = W^T A^T) so that a lane ends up with 4 consecutive channels of one
//     pixel: the epilogue (BN, SiLU, the ShuffleV2 interleave with the other branch) is 16-byte loads / stores straight from
//     the accumulators -- no staging tile, no barrier;
//   * two workgroup barriers per chunk; the depthwise parameters of all G channels sit in LDS for the whole tile.
// (A first version that staged the INPUT rows of a band in LDS by DMA is tools/lab/dwpwx6_experiment.hip: slower.)
#include <string.h>

#include "split.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

struct DwPwX6Args {
  const float* in;
  float* out;
  const float* res;
  const float* dwp;            // [9][G] taps, [G] BN scale, [G] BN bias, [G] PReLU slope (when act = PReLU)
  const unsigned short* w;     // [G / 32][3][N][32] bf16
  const float* scale2;         // [N] BN scale, then [N] BN bias
  int H, W, OH, OW, OHW, G, N;
  int in_ld, out_ld, res_ld, res_C, has_slope, act2, res_mode;
  long in_ns, M;
};

constexpr int TM = 128;        // output pixels of a tile
constexpr int KC = 32;         // channels of a chunk
constexpr int LDA = KC + 8;    // bf16 elements per pixel row of a plane (80 bytes: conflict-free 16-byte fragment reads)
constexpr int APL = TM * LDA;  // bf16 elements of a plane

// NT = N / 16 (4 or 8), P = output pixels per depthwise item (4: OW % 4 == 0), S = stride
template <int NT, int P, int S>
__global__ __launch_bounds__(256, 2) void dwpwx6_kernel(DwPwX6Args p) {
  constexpr int N = NT * 16, WIN = (P - 1) * S + 3;
  constexpr int BPL = N * 32;                                          // bf16 elements of a weight plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* At = (unsigned short*)smem_raw;                      // [3][TM][LDA]
  unsigned short* Bs = At + 3 * APL;                                   // [3][N][32]
  float* Ws = (float*)(Bs + 3 * BPL);                                  // [12][G]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const long m0 = (long)blockIdx.x * TM;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const int G = p.G;

  // depthwise parameters of all channels -> LDS (rows 0..8 taps, 9 scale, 10 bias, 11 slope: zeros without a PReLU)
  for (int i = tid; i < 12 * G / 4; i += 256) {
    const int row = (i * 4) / G;
    *(f32x4*)&Ws[i * 4] = (row < 11 || p.has_slope) ? *(const f32x4*)(p.dwp + i * 4) : z;
  }

  // this thread's depthwise item: pixel group g (P pixels), channel quad c4 of the chunk
  const int g = tid >> 3, c4 = tid & 7;
  const int r = g * P;
  long m = m0 + r;
  m = m < p.M ? m : p.M - P;                             // tail groups recompute the last pixels; they are never stored
  const unsigned img = (unsigned)m / (unsigned)p.OHW;
  const unsigned rem = (unsigned)m - img * (unsigned)p.OHW;
  const int oy = (int)(rem / (unsigned)p.OW), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
  const int iy0 = oy * S - 1, ix0 = ox * S - 1;
  const float* ib = p.in + (long)img * p.in_ns + 4 * c4;

  f32x4 acc[2][NT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[t][n] = z;

  // the 3 x WIN window of this thread's item, one chunk of channels.  Stride 1 (PF): the window of chunk ch + 1 is requested
  // right after the barrier that ends chunk ch's depthwise phase and lands under its MFMAs -- loaded at the top of the
  // depthwise phase (the stride-2 form, whose 3 x 9 window leaves no registers for it) every chunk starts with an exposed
  // round trip to L2
  constexpr bool PF = S == 1;
  f32x4 x[3][WIN];
  auto load_x = [&](int ch) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = iy0 + ky;
      const float* rowp = ib + KC * ch + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
#pragma unroll
      for (int j = 0; j < WIN; ++j) {
        const int ix = ix0 + j;
        x[ky][j] = *(const f32x4*)(rowp + (long)min(max(ix, 0), p.W - 1) * p.in_ld);   // clamped; masked where it is used
      }
    }
  };
  unsigned inside = 0;                                   // bit ky * WIN + j: window element (ky, j) lies inside the image
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int j = 0; j < WIN; ++j)
      inside |= (((unsigned)(iy0 + ky) < (unsigned)p.H) && ((unsigned)(ix0 + j) < (unsigned)p.W)) ? 1u << (ky * WIN + j) : 0u;

  const int nchunks = G / KC;
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();                                     // parameters staged / the previous chunk's MFMAs are done with At and Bs
    // this chunk's weight slab: three planes of N x 64 bytes by LDS-DMA (lands under the depthwise phase)
    {
      const unsigned char* src = (const unsigned char*)(p.w + (long)ch * 3 * BPL) + lane * 16;
#pragma unroll
      for (int j = 0; j < (3 * NT + 3) / 4; ++j) {
        const int c = j * 4 + wave;                      // 1-KiB piece = 16 columns of one plane
        if (c < 3 * NT) __builtin_amdgcn_global_load_lds((gbl_ptr)(src + c * 1024), (lds_ptr)((unsigned char*)Bs + c * 1024), 16, 0, 0);
      }
    }
    // ---- depthwise + BN [+ PReLU] of channels 32 ch + 4 c4 .. + 3 for P pixels, split, -> At ----
    {
      const int c = KC * ch + 4 * c4;
      if (!PF || ch == 0) load_x(ch);
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int j = 0; j < WIN; ++j) x[ky][j] = (inside >> (ky * WIN + j)) & 1u ? x[ky][j] : z;
      f32x4 a[P];
#pragma unroll
      for (int k = 0; k < P; ++k) a[k] = z;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * G + c];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * G + c];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * G + c];
#pragma unroll
        for (int k = 0; k < P; ++k) {
          a[k] += x[ky][k * S] * w0;
          a[k] += x[ky][k * S + 1] * w1;
          a[k] += x[ky][k * S + 2] * w2;
        }
      }
      const f32x4 sc = *(const f32x4*)&Ws[9 * G + c], bi = *(const f32x4*)&Ws[10 * G + c], sl = *(const f32x4*)&Ws[11 * G + c];
#pragma unroll
      for (int k = 0; k < P; ++k) {
        f32x4 v = a[k] * sc + bi;
        if (p.has_slope) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
        }
        unsigned h0, m0_, l0, h1, m1_, l1;
        fp_split_pair(v[0], v[1], h0, m0_, l0);
        fp_split_pair(v[2], v[3], h1, m1_, l1);
        unsigned short* dst = At + (r + k) * LDA + 4 * c4;
        *(u32x2*)dst = u32x2{h0, h1};
        *(u32x2*)(dst + APL) = u32x2{m0_, m1_};
        *(u32x2*)(dst + 2 * APL) = u32x2{l0, l1};
      }
    }
    __syncthreads();                                     // At complete, the weight slab landed (the barrier drains the DMA)
    if (PF && ch + 1 < nchunks) load_x(ch + 1);
    // ---- 1x1: W^T (LDS) x A^T (LDS) for this wave's two pixel tiles ----
    {
      fp_frag3 af[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned short* src = At + (32 * wave + 16 * t + l15) * LDA + 8 * q;
        af[t].h = *(const u32x4*)src;
        af[t].m = *(const u32x4*)(src + APL);
        af[t].l = *(const u32x4*)(src + 2 * APL);
      }
      const unsigned short* Bc = Bs + (l15 * 32 + 8 * q);
      fp_frag3 bf[2];
      auto ldb = [&](int n, fp_frag3& b) {
        b.h = *(const u32x4*)(Bc + n * 512);
        b.m = *(const u32x4*)(Bc + BPL + n * 512);
        b.l = *(const u32x4*)(Bc + 2 * BPL + n * 512);
      };
      ldb(0, bf[0]);
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        if (n + 1 < NT) ldb(n + 1, bf[(n + 1) & 1]);
        const fp_frag3& b = bf[n & 1];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][n] = fp_mfma_x6(b.h, b.m, b.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
      }
    }
  }

  // ---- epilogue: pixel m0 + 32 wave + 16 t + l15, channels 16 n + 4 q .. + 3 ----
  const bool shuffle = p.res_mode == FP_RES_SHUFFLE2;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int chn = 16 * n + 4 * q;
    const f32x4 sc = *(const f32x4*)(p.scale2 + chn), bi = *(const f32x4*)(p.scale2 + N + chn);
    f32x4 rv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      long mo = m0 + 32 * wave + 16 * t + l15;
      mo = mo < p.M ? mo : p.M - 1;
      rv[t] = z;
      if (p.res_mode != FP_RES_NONE && chn < p.res_C) rv[t] = *(const f32x4*)(p.res + mo * p.res_ld + chn);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long mo = m0 + 32 * wave + 16 * t + l15;
      f32x4 v = acc[t][n] * sc + bi;
      if (p.act2 == FP_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fp_silu(v[e]);
      }
      if (mo < p.M) {
        if (shuffle) {                                   // out[2 c] = res[c], out[2 c + 1] = y[c]: two 16-byte pieces
          float* o = p.out + mo * p.out_ld + 2 * chn;
          *(f32x4*)o = f32x4{rv[t][0], v[0], rv[t][1], v[1]};
          *(f32x4*)(o + 4) = f32x4{rv[t][2], v[2], rv[t][3], v[3]};
        } else {
          if (p.res_mode == FP_RES_ADD_AFTER_ACT) v += rv[t];
          *(f32x4*)(p.out + mo * p.out_ld + chn) = v;
        }
      }
    }
  }
}

template <int NT, int S>
int launch(const DwPwX6Args& a, hipStream_t s) {
  const int lds = 3 * APL * 2 + 3 * NT * 16 * 32 * 2 + 12 * a.G * 4;
  const long tiles = (a.M + TM - 1) / TM;
  if (tiles >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  const hipError_t ae = hipFuncSetAttribute((const void*)dwpwx6_kernel<NT, 4, S>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((dwpwx6_kernel<NT, 4, S>), dim3((unsigned)tiles), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// FP_OP_DWPW with FP_OPF_SPLIT3: G a multiple of 32 (<= 256), Cout 64 or 128, OW a multiple of 4, dense output rows.
bool fp_dwpwx6_eligible(const fp_op& op) {
  if (op.kind != FP_OP_DWPW || !(op.flags & FP_OPF_SPLIT3) || (op.flags & ~FP_OPF_SPLIT3)) return false;
  if (op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1 || (op.stride != 1 && op.stride != 2)) return false;
  if (op.OH != (op.H + 2 - 3) / op.stride + 1 || op.OW != (op.W + 2 - 3) / op.stride + 1 || op.OW % 4) return false;
  if (op.Cin % 32 || op.Cin > 256 || (op.Cout != 64 && op.Cout != 128) || op.out_cmul != 1) return false;
  const long OHW = (long)op.OH * op.OW;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns != OHW * op.out_ld) return false;
  if (op.in_ns < (long)op.H * op.W * op.in_ld || op.w_off % 4 || op.slope_off % 4 || op.bias_off >= 0) return false;
  if (op.act != FP_ACT_NONE && op.act != FP_ACT_PRELU) return false;
  if (op.act2 != FP_ACT_NONE && op.act2 != FP_ACT_SILU) return false;
  if (op.res_mode != FP_RES_NONE && op.res_mode != FP_RES_SHUFFLE2 && op.res_mode != FP_RES_ADD_AFTER_ACT) return false;
  if (op.res_mode != FP_RES_NONE && (op.res_ld % 4 || op.res_off % 4 || op.res_ns != OHW * op.res_ld || op.res_C % 4)) return false;
  if (op.res_mode == FP_RES_SHUFFLE2 && (op.res_C < op.Cout || op.out_ld < 2 * op.Cout)) return false;
  if ((long)op.N * OHW >= (1L << 31) || (long)op.N * OHW < 4) return false;
  return true;
}

long fp_dwpwx6_w_floats(const fp_op& op) { return (long)op.Cin * op.Cout * 3 / 2 + 2L * op.Cout; }

int fp_launch_dwpwx6(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_dwpwx6_eligible(op)) return FP_ERR_UNSUPPORTED;
  DwPwX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.dwp = weights + op.w_off;
  a.w = (const unsigned short*)(weights + op.slope_off);
  a.scale2 = weights + op.slope_off + (long)op.Cin * op.Cout * 3 / 2;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.OHW = op.OH * op.OW; a.G = op.Cin; a.N = op.Cout;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.res_C = op.res_C;
  a.has_slope = op.act == FP_ACT_PRELU ? 1 : 0;
  a.act2 = op.act2; a.res_mode = op.res_mode;
  a.in_ns = op.in_ns;
  a.M = (long)op.N * a.OHW;
  if (op.Cout == 64) return op.stride == 1 ? launch<4, 1>(a, s) : launch<4, 2>(a, s);
  return op.stride == 1 ? launch<8, 1>(a, s) : launch<8, 2>(a, s);
}
