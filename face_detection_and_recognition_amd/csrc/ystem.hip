// ystem.hip — head of YOLOv5-face's StemBlock in one kernel (gfx950).
//
// StemBlock.forward (fde/modules/yolov5_face/pytorch/models/common.py:58-73):
//     s1 = stem_1(x)            Conv 3x3 stride 2 pad 1, 3 -> c, (BN), SiLU         640^2 -> 320^2
//     a  = stem_2a(s1)          Conv 1x1, c -> c/2, (BN), SiLU
//     p  = maxpool2x2(s1)       MaxPool2d(2, 2, ceil_mode=True)                      320^2 -> 160^2
//     out = stem_3(cat(stem_2b(a), p))
// s1 is the largest activation of the network (13 MB per 640^2 image at c = 32) and op-granular execution moved it
// through HBM three times (written by stem_1, read by stem_2a and by the pool): 10 of the block's 19 GB at batch 256
// (profiles/r01: stem_1 1.82 ms + stem_2a 1.23 ms + maxpool 0.78 ms).  Here s1 lives only in LDS:
//   tile    = 8 rows x 32 columns of s1 (256 pixels) of one image; persistent workgroups, XCD-aware tile order
//   phase 0 = the 17 x 65 input pixels (4 floats each) the tile's taps touch -> LDS, coalesced, zero borders
//   phase 1 = stem_1 as in stem.hip: a tap of 4 channels IS one pixel, so MFMA A fragments are ds_read_b128 straight
//             out of the image (no im2col); 32x32x2 f32 MFMA, one 32-pixel row per m tile, two rows per wave with the
//             MFMAs issued alternately (dependent f32 MFMAs issue at half rate); epilogue acc*scale+bias, SiLU -> S1
//   phase 2 = stem_2a: 1x1 conv on S1 with 16x16x4 f32 MFMA (c/2 <= 16 output channels per n tile: a 32-wide tile
//             would be half padding), epilogue + SiLU -> LDS staging -> 16-byte coalesced stores of `a`
//           + the 2x2 max pool of S1 -> 16-byte stores straight into the second half of stem_3's concat buffer
// stem_2b and stem_3 stay FP_OP_CONVs.  Same products in the same k order as conv.hip for stem_1 (k = tap*4 + c);
// stem_2a sums k in the order 16j + 4g + e of its fragment layout (fp32 reassociation only).
#include "common.h"
#include "letterbox.h"

namespace {

constexpr int TR = 8, TC = 32;          // tile of s1: rows x columns
constexpr int IR = 2 * TR + 1;          // 17 input rows
constexpr int IC = 2 * TC + 1;          // 65 input columns (float4 pixels)
constexpr int S1LD = 36;                // floats per S1 pixel (32 channels + 4: conflict-free 16-byte reads)
constexpr int K1PAD = 40;               // 9 taps x 4 channels, padded to a multiple of 8

struct YStemArgs {
  const float* in;
  float* a_out;
  float* p_out;
  const float* w1;      // packed as FP_OP_CONV: [K1PAD/4][32][4]
  const float* sc1;     // [32] or null
  const float* bi1;     // [32]
  const float* w2;      // [2][4][NB2*16][4]: element e of (j, g, n) = W2a[n][16j + 4g + e]
  const float* sc2;     // [NB2*16] or null
  const float* bi2;     // [NB2*16]
  int H, W, H1, W1, W2o, C1, C2, a_ld, p_ld, tiles_x, tiles_per_img, ntiles;
  long in_ns, a_ns, p_ns;
  int c4;               // the pixel's fourth channel carries weights (0: 3-channel image, facepath.h FP_OPF_IN_C3)
  // U8 input (the letterbox fused into the staging, letterbox.h): frames [N][fh][fw][3] u8, tap tables of the H x W canvas
  const uint8_t* frames;
  const fp_lb_tap* xtab;
  const fp_lb_tap* ytab;
  const float* lut;
  long frame_bytes, row_bytes;
  int frame_h, frame_w;
};

template <int NB2, bool U8>
__global__ __launch_bounds__(256, 2) void ystem_kernel(YStemArgs p) {
  constexpr int C2S = NB2 * 16;         // staged channels per `a` pixel
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Img = smem;                              // [IR][IC][4]; reused as the `a` staging tile [256][C2S]
  float* S1 = Img + (IR * IC * 4 > 256 * C2S ? IR * IC * 4 : 256 * C2S);   // [256][S1LD]
  float* W1s = S1 + 256 * S1LD;                   // [K1PAD/4][32][4]
  float* W2s = W1s + K1PAD * 32;                  // [2][4][C2S][4]
  float* LutS = W2s + 32 * C2S;                   // [256] normalisation LUT (U8 input only)
  fp_lb_tap* TabS = (fp_lb_tap*)(LutS + 256);     // [W + H + 1] tap tables (U8 input only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < K1PAD * 32 / 4; i += 256) *(f32x4*)&W1s[i * 4] = *(const f32x4*)(p.w1 + (long)i * 4);
  for (int i = tid; i < 8 * C2S; i += 256) *(f32x4*)&W2s[i * 4] = *(const f32x4*)(p.w2 + (long)i * 4);
  int pad_value = 0, swap_rb = 0;
  if (U8) {
    if (!fp_lb_geometry_ok(p.xtab, p.W, p.H, p.frame_h, p.frame_w)) return;   // tables of another geometry (uniform)
    LutS[tid] = p.lut[tid];
    for (int i = tid; i < p.W + p.H; i += 256) TabS[i] = p.xtab[i];
    const fp_lb_tap tr = p.xtab[p.W + p.H];    // trailer entry of the tables: { pad colour, swap R/B }
    pad_value = tr.a;
    swap_rb = tr.b;
    __syncthreads();
  }
  const float sc1 = p.sc1 ? p.sc1[lr] : 1.f, bi1 = p.bi1[lr];
  float sc2[NB2], bi2[NB2];
#pragma unroll
  for (int nb = 0; nb < NB2; ++nb) {
    sc2[nb] = p.sc2 ? p.sc2[nb * 16 + (lane & 15)] : 1.f;
    bi2[nb] = p.bi2[nb * 16 + (lane & 15)];
  }

  const int G = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }

  // staging of a tile's input pixels: issue (registers) and write (LDS) are split so that the NEXT tile's loads are
  // in flight during phase 2 and the stores of the current tile
  constexpr int NSLOT = (IR * IC + 255) / 256;
  f32x4 stg[U8 ? 1 : NSLOT];
  fp_lb_raw raw[U8 ? NSLOT : 1];
  fp_lb_tap xt[U8 ? NSLOT : 1], yt[U8 ? NSLOT : 1];
  unsigned stg_ok = 0;
  auto issue_stage = [&](int tile) {
    const int img = tile / p.tiles_per_img, tin = tile - img * p.tiles_per_img;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
    const int iy0 = 2 * ty * TR - 1, ix0 = 2 * tx * TC - 1;
    stg_ok = 0;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int i = tid + 256 * j;
      const int r = i / IC, c = i - r * IC;
      const int iy = iy0 + r, ix = ix0 + c;
      if (i < IR * IC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) stg_ok |= 1u << j;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      if (U8) {   // the canvas pixel is resampled from the u8 frame: its two 8-byte tap windows are what is staged
        xt[j] = TabS[cx];
        yt[j] = TabS[p.W + cy];
        raw[j] = fp_lb_issue(p.frames + (long)img * p.frame_bytes, p.row_bytes, xt[j], yt[j]);
      } else {
        stg[j] = *(const f32x4*)(p.in + (long)img * p.in_ns + ((long)cy * p.W + cx) * 4);
      }
    }
  };

  if (pos < p.ntiles) issue_stage(pos);
  for (int tile = pos; tile < p.ntiles; tile += G) {
    const int img = tile / p.tiles_per_img, tin = tile - img * p.tiles_per_img;
    const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
    const int y1_0 = ty * TR, x1_0 = tx * TC;       // s1 coordinates of the tile's first pixel

    // ---- phase 0: staged input pixels -> LDS (zero padding applied at the write) ----
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
      const int i = tid + 256 * j;
      f32x4 v;
      if (U8) v = fp_lb_finish(raw[j], xt[j], yt[j], LutS, pad_value, swap_rb);
      else v = stg[j];
      if (i < IR * IC) *(f32x4*)&Img[i * 4] = ((stg_ok >> j) & 1u) ? v : z4;
    }
    __syncthreads();

    // ---- phase 1: stem_1 (3x3 stride 2) on the LDS image; wave w owns s1 rows 2w and 2w + 1 of the tile ----
    {
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
      const float* base0 = Img + ((2 * (2 * wave)) * IC + 2 * lr) * 4;
      const float* base1 = base0 + 2 * IC * 4;
#pragma unroll
      for (int kq = 0; kq < K1PAD / 8; ++kq) {
        int t = 2 * kq + h;                 // taps past the ninth meet zero weights: any finite pixel will do
        t = t < 9 ? t : 0;
        const int ky = t / 3, kx = t - ky * 3;
        const f32x4 a0 = *(const f32x4*)(base0 + (ky * IC + kx) * 4);
        const f32x4 a1 = *(const f32x4*)(base1 + (ky * IC + kx) * 4);
        const f32x4 bv = *(const f32x4*)&W1s[((kq * 2 + h) * 32 + lr) * 4];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bv[e], acc0, 0, 0, 0);
          FP_MFMA_ORDER();
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bv[e], acc1, 0, 0, 0);
          FP_MFMA_ORDER();
        }
        if (p.c4) {   // the pad channel of a 3-channel image meets zero weights: skipped (facepath.h FP_OPF_IN_C3)
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[3], bv[3], acc0, 0, 0, 0);
          FP_MFMA_ORDER();
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[3], bv[3], acc1, 0, 0, 0);
          FP_MFMA_ORDER();
        }
      }
      // C/D map: column (channel) = lane & 31, row (pixel x) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
      float* s0 = S1 + (2 * wave) * TC * S1LD + lr;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int x = (reg & 3) + 8 * (reg >> 2) + 4 * h;
        s0[x * S1LD] = fp_silu(acc0[reg] * sc1 + bi1);
        s0[(TC + x) * S1LD] = fp_silu(acc1[reg] * sc1 + bi1);
      }
    }
    __syncthreads();   // S1 complete; Img is free
    if (tile + G < p.ntiles) issue_stage(tile + G);

    // ---- phase 2: stem_2a (1x1, 16x16x4 MFMA) on S1 -> staging;  wave w owns pixels 64w .. 64w + 63 ----
    {
      const int r16 = lane & 15, g = lane >> 4;
      f32x4 acc[4][NB2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) acc[mt][nb] = z4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 a[4], b[NB2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a[mt] = *(const f32x4*)&S1[(wave * 64 + mt * 16 + r16) * S1LD + 16 * j + 4 * g];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) b[nb] = *(const f32x4*)&W2s[((j * 4 + g) * C2S + nb * 16 + r16) * 4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nb = 0; nb < NB2; ++nb)
              acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][e], b[nb][e], acc[mt][nb], 0, 0, 0);
      }
      // C/D map (16x16): column = lane & 15, row = 4 * (lane >> 4) + reg
      float* At = Img;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            At[(wave * 64 + mt * 16 + 4 * g + reg) * C2S + nb * 16 + r16] = fp_silu(acc[mt][nb][reg] * sc2[nb] + bi2[nb]);
    }
    // ---- max pool 2x2 of S1 -> concat half (needs only S1: issued before the barrier that publishes `a`) ----
    {
      const int c4n = p.C1 >> 2;                        // float4s per pooled pixel
      const int npool = (TR / 2) * (TC / 2) * c4n;
      float* pb = p.p_out + (long)img * p.p_ns;
      for (int i = tid; i < npool; i += 256) {
        const int c4 = i % c4n, q = i / c4n;
        const int px = q % (TC / 2), py = q / (TC / 2);
        const int oy = y1_0 / 2 + py, ox = x1_0 / 2 + px;
        const float* s = S1 + ((2 * py) * TC + 2 * px) * S1LD + c4 * 4;
        const f32x4 v00 = *(const f32x4*)s, v01 = *(const f32x4*)(s + S1LD);
        const f32x4 v10 = *(const f32x4*)(s + TC * S1LD), v11 = *(const f32x4*)(s + (TC + 1) * S1LD);
        f32x4 m;
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
        if (2 * oy < p.H1 && 2 * ox < p.W1) *(f32x4*)(pb + ((long)oy * p.W2o + ox) * p.p_ld + c4 * 4) = m;
      }
    }
    __syncthreads();   // `a` staging complete
    {
      const int c4n = p.C2 >> 2;
      const int nst = TR * TC * c4n;
      float* ab = p.a_out + (long)img * p.a_ns;
      for (int i = tid; i < nst; i += 256) {
        const int c4 = i % c4n, q = i / c4n;
        const int x = q % TC, y = q / TC;
        const int y1 = y1_0 + y, x1 = x1_0 + x;
        const f32x4 v = *(const f32x4*)&Img[q * C2S + c4 * 4];
        if (y1 < p.H1 && x1 < p.W1) *(f32x4*)(ab + ((long)y1 * p.W1 + x1) * p.a_ld + c4 * 4) = v;
      }
    }
    __syncthreads();   // staging and S1 free for the next tile
  }
}

size_t ystem_lds_bytes(int nb2, int table_entries) {
  const size_t img = (size_t)IR * IC * 4, st = (size_t)256 * nb2 * 16;
  return 4 * ((img > st ? img : st) + (size_t)256 * S1LD + (size_t)K1PAD * 32 + (size_t)8 * nb2 * 16 * 4 + 256) +
         8 * (size_t)table_entries;
}

}  // namespace

// FP_OP_YSTEM fields (include/facepath.h): in = the 4-float-pixel image; out = stem_2a's output view (Cout = its
// physical channels); res_* = the pooled destination view (res_C = stem_1's physical channels c, res_H/res_W = the
// pooled size); w_off/scale_off/bias_off = stem_1 (packed as FP_OP_CONV, [32] scale / bias); slope_off = stem_2a blob.
int fp_ystem_nb2(const fp_op& op) { return (op.Cout + 15) / 16; }

static int ystem_fill(const fp_op& op, const float* weights, float* arena, YStemArgs& a) {
  if (op.KH != 3 || op.KW != 3 || op.stride != 2 || op.pad_t != 1 || op.pad_l != 1) return FP_ERR_UNSUPPORTED;
  if (op.H % 4 || op.W % 4 || op.OH != op.H / 2 || op.OW != op.W / 2) return FP_ERR_UNSUPPORTED;
  if (op.res_C <= 0 || op.res_C > 32 || op.res_C % 4 || op.Cout <= 0 || op.Cout > 32 || op.Cout % 4) return FP_ERR_UNSUPPORTED;
  if (op.out_off % 4 || op.res_off % 4 || op.out_ld % 4 || op.res_ld % 4 || op.out_ns % 4 || op.res_ns % 4 || op.out_cmul != 1)
    return FP_ERR_ALIGNMENT;
  const int nb2 = fp_ystem_nb2(op);
  a.in = nullptr;
  a.frames = nullptr;
  a.a_out = arena + op.out_off;
  a.p_out = arena + op.res_off;
  a.w1 = weights + op.w_off;
  a.sc1 = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bi1 = weights + op.bias_off;
  const float* blob = weights + op.slope_off;
  a.w2 = blob;
  a.sc2 = blob + 8 * nb2 * 16 * 4;          // [nb2*16] scale then [nb2*16] bias (scale = 1 when the BN is folded)
  a.bi2 = a.sc2 + nb2 * 16;
  a.H = op.H; a.W = op.W; a.H1 = op.OH; a.W1 = op.OW; a.W2o = op.OW / 2;
  a.C1 = op.res_C; a.C2 = op.Cout; a.a_ld = op.out_ld; a.p_ld = op.res_ld;
  a.in_ns = op.in_ns; a.a_ns = op.out_ns; a.p_ns = op.res_ns;
  a.tiles_x = fp_ceil_div(op.OW, TC);
  a.tiles_per_img = a.tiles_x * fp_ceil_div(op.OH, TR);
  a.ntiles = op.N * a.tiles_per_img;
  a.xtab = a.ytab = nullptr;
  a.lut = nullptr;
  a.frame_bytes = a.row_bytes = 0;
  a.frame_h = a.frame_w = 0;
  return FP_OK;
}

template <bool U8>
static int ystem_launch(const fp_op& op, const YStemArgs& a, hipStream_t s) {
  const int nb2 = fp_ystem_nb2(op);
  const size_t lds = ystem_lds_bytes(nb2, U8 ? op.H + op.W : 0);
  int grid = 512;                             // two workgroups per CU (LDS ~63 KiB each), persistent
  if (grid > a.ntiles) grid = a.ntiles;
  hipError_t ae;
  if (nb2 == 1) {
    ae = hipFuncSetAttribute((const void*)ystem_kernel<1, U8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ae == hipSuccess) hipLaunchKernelGGL((ystem_kernel<1, U8>), dim3(grid), dim3(256), lds, s, a);
  } else {
    ae = hipFuncSetAttribute((const void*)ystem_kernel<2, U8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ae == hipSuccess) hipLaunchKernelGGL((ystem_kernel<2, U8>), dim3(grid), dim3(256), lds, s, a);
  }
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_ystem(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (op.Cin != 4 || op.in_ld != 4 || op.in_off % 4 || op.in_ns % 4) return FP_ERR_UNSUPPORTED;
  if (op.res_H != op.OH / 2 || op.res_W != op.OW / 2) return FP_ERR_UNSUPPORTED;
  YStemArgs a;
  const int rc = ystem_fill(op, weights, arena, a);
  if (rc != FP_OK) return rc;
  a.in = arena + op.in_off;
  a.c4 = (op.flags & FP_OPF_IN_C3) ? 0 : 1;
  return ystem_launch<false>(op, a, s);
}

// FP_OP_YSTEM_U8: the same op reading the u8 frames through the letterbox tap tables (include/facepath.h).
int fp_launch_ystem_u8(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, int n_ext, hipStream_t s) {
  const long e = op.in_off;
  if (e < 0 || e + 2 >= n_ext || !ext) return FP_ERR_INVALID_ARG;
  const int fh = op.res_H, fw = op.res_W;
  if (op.Cin != 3 || fh <= 0 || fw < 3) return FP_ERR_INVALID_ARG;
  if (ext[e].bytes < (size_t)op.N * fh * fw * 3 || ext[e + 1].bytes < (size_t)(op.H + op.W + 2) * 8 || op.H + op.W > 2048 ||
      ext[e + 2].bytes < 256 * sizeof(float) || !ext[e].ptr || !ext[e + 1].ptr || !ext[e + 2].ptr)
    return FP_ERR_BOUNDS;
  YStemArgs a;
  const int rc = ystem_fill(op, weights, arena, a);
  if (rc != FP_OK) return rc;
  a.frames = (const uint8_t*)ext[e].ptr;
  a.xtab = (const fp_lb_tap*)ext[e + 1].ptr;
  a.ytab = a.xtab + op.W;
  a.lut = (const float*)ext[e + 2].ptr;
  a.row_bytes = (long)fw * 3;
  a.frame_bytes = (long)fh * fw * 3;
  a.frame_h = fh;
  a.frame_w = fw;
  a.c4 = 0;
  return ystem_launch<true>(op, a, s);
}
