// stem.hip — first conv of the networks: KxK stride-2 conv on a 3(+1 pad)-channel image (gfx950).
//
// BlazeFace's stem (fde/modules/blazeface/blazeface.py:176-180: F.pad(1,2,1,2) + Conv2d(3, 24, 5, stride 2) + ReLU) and
// Mobile-FaceNet's conv1 (fde/modules/mobile_facenet/mobile_facenet.py:116: Conv_block(3, 64, 3x3, stride 2, pad 1),
// BN, PReLU) read a 4-float pixel (RGB + one zero channel).  As an implicit GEMM their A operand is pure gather: with
// conv.hip every lane fetched its row's taps one 16-B pixel at a time (25 scattered loads per output pixel for the
// 5x5), 1.4-2.2 TB/s algorithmic (profiles/r01).  A tap of 4 channels IS one pixel, and an MFMA A fragment of 4
// consecutive k is one tap -- so here the input rows a tile needs are staged ONCE into LDS with coalesced loads
// (zero-filled borders) and the fragments are ds_read_b128 straight out of that image:  no im2col, no gathers.
//   tile      = 128 consecutive output pixels of ONE image (raster order; wave w owns pixels 32w..32w+31)
//   LDS       = image rows [nrows][Wp] float4 | packed weights [Kpad/4][Npad][4] | output tile [128][Cout]
//   pipeline  = persistent workgroups; the next tile's image rows are loaded into registers (<= 6 float4 per lane)
//               before the current tile's MFMAs and written to LDS after its epilogue
//   epilogue  = acc*scale+bias, branch-free none/ReLU/PReLU, through LDS, then the tile's rows leave as ONE
//               contiguous run (dense NHWC output: 128 x Cout floats)
// k order and the packed weights are conv.hip's (k = tap*4 + c, zero rows up to Kpad): same products in the same
// order, identical results.
#include "letterbox.h"
#include "split.h"

namespace {

struct StemArgs {
  const float* in;
  float* out;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int H, W, OH, OW, OHW, Cout, Npad, Kpad, act, pad_t, pad_l, Wp, rows_max, tiles_per_img, ntiles;
  long in_ns, out_ns;
  int out_rowpad;            // output in the row-padded layout (facepath.h)
  int c4;                    // the pixel's fourth channel carries weights (0: 3-channel image padded to 16 bytes)
  fp_divisor q4_div, ow_div;
  // U8 input: the H x W image is a letterbox canvas resampled from u8 frames while it is staged (letterbox.h)
  const uint8_t* frames;
  const fp_lb_tap* tabs;     // [W] column taps, [H] row taps, trailer {pad colour, swap R/B}
  const float* lut;
  long frame_bytes, row_bytes;
  int frame_h, frame_w;
};

constexpr int TMS = 128;
constexpr int PF = 6;   // staged float4s per lane and tile (rows_max * Wp <= PF * 256, checked on the host)

template <int KS, int NB, bool U8>
__global__ __launch_bounds__(256, 3) void stem_conv_kernel(StemArgs p) {
  constexpr int BN = NB * 32;
  constexpr int NTAP = KS * KS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Img = smem;                                   // [rows_max][Wp][4]
  float* Bs = Img + p.rows_max * p.Wp * 4;             // [Kpad/4][BN][4]
  float* Ot = Bs + p.Kpad * BN;                        // [128][Cout]
  float* LutS = Ot + TMS * p.Cout;                     // U8: [256] normalisation LUT
  fp_lb_tap* TabS = (fp_lb_tap*)(LutS + 256);          // U8: [W + H] tap tables
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  int pad_value = 0, swap_rb = 0;
  if (U8) {
    if (!fp_lb_geometry_ok(p.tabs, p.W, p.H, p.frame_h, p.frame_w)) return;   // tables of another geometry (uniform)
    LutS[tid] = p.lut[tid];
    for (int i = tid; i < p.W + p.H; i += 256) TabS[i] = p.tabs[i];
    const fp_lb_tap tr = p.tabs[p.W + p.H];
    pad_value = tr.a;
    swap_rb = tr.b;
    __syncthreads();
  }
  // weights -> LDS once (the packed blob is [Kpad/4][Npad][4] with Npad == BN)
  for (int i = tid; i < (p.Kpad >> 2) * BN; i += 256) *(f32x4*)&Bs[i * 4] = *(const f32x4*)(p.w + (long)i * 4);
  float sc[NB], bi[NB], sl[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = p.scale ? p.scale[nn] : 1.f;
    bi[nb] = p.bias ? p.bias[nn] : 0.f;
    sl[nb] = p.act == FP_ACT_PRELU ? p.slope[nn] : (p.act == FP_ACT_RELU ? 0.f : 1.f);
  }

  const int G = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }

  // staging of a tile's image rows: lane -> float4 slots tid + 256*j of the [nrows][Wp] image
  f32x4 pre[U8 ? 1 : PF];
  fp_lb_raw raw[U8 ? PF : 1];
  fp_lb_tap xt[U8 ? PF : 1], yt[U8 ? PF : 1];
  unsigned premask = 0;
  int srow[PF], sxp[PF];   // slot -> (row, column) of the staged image: the same for every tile
#pragma unroll
  for (int j = 0; j < PF; ++j) {
    const int i = tid + 256 * j;
    srow[j] = i / p.Wp;
    sxp[j] = i - srow[j] * p.Wp;
  }
  auto issue_stage = [&](int tile) {
    const int img = tile / p.tiles_per_img, tin = tile - img * p.tiles_per_img;
    const int m_lo = tin * TMS, m_hi = min(m_lo + TMS, p.OHW) - 1;
    const int oy_lo = m_lo / p.OW, oy_hi = m_hi / p.OW;
    const int iy_lo = oy_lo * 2 - p.pad_t, nslots = ((oy_hi - oy_lo) * 2 + KS) * p.Wp;
    const float* ib = U8 ? nullptr : p.in + (long)img * p.in_ns;
    premask = 0;
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = tid + 256 * j;
      const int iy = iy_lo + srow[j], ix = sxp[j] - p.pad_l;
      const bool ok = i < nslots && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      if (U8) {
        xt[j] = TabS[cx];
        yt[j] = TabS[p.W + cy];
        raw[j] = fp_lb_issue(p.frames + (long)img * p.frame_bytes, p.row_bytes, xt[j], yt[j]);
      } else {
        pre[j] = *(const f32x4*)(ib + ((long)cy * p.W + cx) * 4);
      }
      if (ok) premask |= 1u << j;
    }
  };

  auto write_stage = [&]() {   // staged registers -> LDS image; zero padding applied here, when they are consumed
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int i = tid + 256 * j;
      f32x4 v;
      if (U8) v = fp_lb_finish(raw[j], xt[j], yt[j], LutS, pad_value, swap_rb);
      else v = pre[j];
      if (i < p.rows_max * p.Wp) *(f32x4*)&Img[i * 4] = ((premask >> j) & 1u) ? v : z4;
    }
  };

  int tile = pos;
  if (tile < p.ntiles) {
    issue_stage(tile);
    write_stage();
  }
  __syncthreads();   // weights and the first image staged
  for (int k = 0; tile < p.ntiles; ++k) {
    const int next = (k + 1) * G + pos;
    if (next < p.ntiles) issue_stage(next);   // in flight during the MFMAs and the epilogue

    const int img = tile / p.tiles_per_img, tin = tile - img * p.tiles_per_img;
    const int m_lo = tin * TMS;
    const int nvalid = min(TMS, p.OHW - m_lo);
    // this lane's output pixel (rows past the image end recompute the last pixel; they are not stored)
    const int m = min(m_lo + wave * 32 + lr, p.OHW - 1);
    const int oy = m / p.OW, ox = m - oy * p.OW;
    const float* base = Img + (((oy - m_lo / p.OW) * 2) * p.Wp + ox * 2) * 4;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    for (int kq = 0; kq < (p.Kpad >> 3); ++kq) {
      // fragment of 4 consecutive k = tap 2*kq + h (taps past KS*KS meet zero weights: any finite pixel will do)
      int t = 2 * kq + h;
      t = t < NTAP ? t : 0;
      const int ky = t / KS, kx = t - ky * KS;
      const f32x4 a = *(const f32x4*)(base + (ky * p.Wp + kx) * 4);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f32x4 bv = *(const f32x4*)&Bs[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
        for (int e = 0; e < 3; ++e) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[e], acc[nb], 0, 0, 0);
        // the pad channel of a 3-channel image meets zero weights: its MFMA (a quarter of them) is skipped
        if (p.c4) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], bv[3], acc[nb], 0, 0, 0);
      }
    }
    // epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 32 + lr;
      if (n < p.Cout) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          const float x = acc[nb][reg] * sc[nb] + bi[nb];
          Ot[row * p.Cout + n] = x > 0.f ? x : __builtin_fmaf(x, sl[nb], 0.0f);
        }
      }
    }
    __syncthreads();   // Ot complete; every wave is done reading Img
    // next image -> LDS BEFORE this tile's stores are issued: the wait for its loads (issued a whole MFMA phase ago)
    // would otherwise also wait for those stores (vmcnt counts both)
    if (next < p.ntiles) write_stage();
    {
      const int n4 = nvalid * p.Cout / 4;
      if (!p.out_rowpad) {
        float* ob = p.out + ((long)img * p.OHW + m_lo) * p.Cout;
        for (int i = tid; i < n4; i += 256) *(f32x4*)(ob + (long)i * 4) = *(const f32x4*)&Ot[i * 4];
      } else {   // row-padded output (facepath.h FP_OPF_OUT_ROWPAD): pixel (y, x) at (y*(OW + 1) + x)*Cout
        float* ob = p.out + (long)img * p.out_ns;
        const unsigned q4 = (unsigned)p.Cout >> 2;
        for (int i = tid; i < n4; i += 256) {
          const unsigned px = fp_fastdiv((unsigned)i, p.q4_div), cq = (unsigned)i - px * q4;
          const unsigned m = (unsigned)m_lo + px, oy = fp_fastdiv(m, p.ow_div);
          *(f32x4*)(ob + (long)(m + oy) * p.Cout + cq * 4) = *(const f32x4*)&Ot[i * 4];
        }
      }
    }
    __syncthreads();   // next image visible; Ot free
    tile = next;
  }
}

size_t stem_lds_bytes(int rows_max, int Wp, int Kpad, int Npad, int Cout, int table_entries = 0) {
  return 4 * ((size_t)rows_max * Wp * 4 + (size_t)Kpad * Npad + (size_t)TMS * Cout) +
         (table_entries ? 1024 + 8 * (size_t)table_entries : 0);
}

}  // namespace

// geometry shared by the eligibility test and the launcher
static void stem_geometry(const fp_op& op, int* Wp, int* rows_max) {
  const int pad_r = (op.OW - 1) * 2 - op.pad_l + op.KW - op.W;   // columns read right of the image
  *Wp = op.pad_l + op.W + (pad_r > 0 ? pad_r : 0);
  const int OHW = op.OH * op.OW;
  int rows_touched = 1;                                          // output rows a tile (128-pixel run) can touch
  for (int m_lo = 0; m_lo < OHW; m_lo += TMS) {
    const int m_hi = (m_lo + TMS < OHW ? m_lo + TMS : OHW) - 1;
    const int r = m_hi / op.OW - m_lo / op.OW + 1;
    if (r > rows_touched) rows_touched = r;
  }
  *rows_max = (rows_touched - 1) * 2 + op.KH;
}

// shape conditions shared by the fp32-image and the u8-frame forms
static bool stem_shape_ok(const fp_op& op, int table_entries) {
  if (op.KH != op.KW || (op.KH != 3 && op.KH != 5) || op.stride != 2) return false;
  if (op.out_cmul != 1 || op.out_ld != op.Cout) return false;
  if (!(op.flags & FP_OPF_OUT_ROWPAD) && op.out_ns != (int64_t)op.OH * op.OW * op.Cout) return false;
  if ((op.flags & FP_OPF_OUT_ROWPAD) && (op.Cout < 8 || op.OW < 2)) return false;   // fp_make_divisor needs d >= 2
  if (op.Cout % 4 || op.Cout <= 0 || op.Cout > 64 || op.out_off % 4 || op.w_off % 4) return false;
  if (op.res_mode != FP_RES_NONE) return false;
  if (op.act != FP_ACT_NONE && op.act != FP_ACT_RELU && op.act != FP_ACT_PRELU) return false;
  if (op.act == FP_ACT_PRELU && op.slope_off < 0) return false;
  if (op.pad_t < 0 || op.pad_l < 0 || op.pad_t >= op.KH || op.pad_l >= op.KW) return false;
  if (op.OH <= 0 || op.OW <= 0 || (op.OH - 1) * 2 - op.pad_t >= op.H || (op.OW - 1) * 2 - op.pad_l >= op.W) return false;
  int Wp, rows_max;
  stem_geometry(op, &Wp, &rows_max);
  if ((long)rows_max * Wp > PF * 256) return false;
  const int Kpad = (int)fp_round_up(op.KH * op.KW * 4, 8), Npad = (int)fp_round_up(op.Cout, 32);
  return stem_lds_bytes(rows_max, Wp, Kpad, Npad, op.Cout, table_entries) <= (table_entries ? 80 : 64) * 1024;
}

// KxK (3 or 5) stride-2 conv on a dense 4-float-pixel image into a dense NHWC tensor, Cout <= 64, no residual,
// activation none / ReLU / PReLU.  Everything else stays with conv_igemm_kernel.
bool fp_stem_eligible(const fp_op& op) {
  if (op.kind != FP_OP_CONV) return false;
  if (op.Cin != 4 || op.in_ld != 4 || op.in_ns != (int64_t)op.H * op.W * 4 || op.in_off % 4) return false;
  // small batches: one tile per workgroup (conv_igemm_kernel) is fine -- unless the output is row-padded, which only
  // this kernel writes
  if (!(op.flags & FP_OPF_OUT_ROWPAD) && (long)op.N * op.OH * op.OW < 1024L * TMS) return false;
  return stem_shape_ok(op, 0);
}

static void stem_fill(const fp_op& op, const float* weights, float* arena, StemArgs& a) {
  a.in = nullptr;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.OHW = op.OH * op.OW; a.Cout = op.Cout;
  a.Kpad = (int)fp_round_up(op.KH * op.KW * 4, 8);
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.act = op.act; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
  stem_geometry(op, &a.Wp, &a.rows_max);
  a.tiles_per_img = fp_ceil_div(a.OHW, TMS);
  a.ntiles = op.N * a.tiles_per_img;
  a.in_ns = op.in_ns;
  a.out_rowpad = (op.flags & FP_OPF_OUT_ROWPAD) != 0;
  a.c4 = (op.kind == FP_OP_STEM_U8 || (op.flags & FP_OPF_IN_C3)) ? 0 : 1;
  a.out_ns = op.out_ns;
  a.q4_div = fp_make_divisor((unsigned)(op.Cout >= 8 ? op.Cout / 4 : 2));
  a.ow_div = fp_make_divisor((unsigned)(op.OW >= 2 ? op.OW : 2));
  a.frames = nullptr; a.tabs = nullptr; a.lut = nullptr;
  a.frame_bytes = a.row_bytes = 0;
  a.frame_h = a.frame_w = 0;
}

template <bool U8>
static int stem_launch(const fp_op& op, const StemArgs& a, hipStream_t s) {
  const size_t lds = stem_lds_bytes(a.rows_max, a.Wp, a.Kpad, a.Npad, op.Cout, U8 ? op.H + op.W : 0);
  int per_cu = (int)(160 * 1024 / lds);   // resident workgroups per CU by LDS (registers allow 3)
  const int cap = 3;
  per_cu = per_cu > cap ? cap : (per_cu < 1 ? 1 : per_cu);
  int grid = 256 * per_cu;
  if (grid > a.ntiles) grid = a.ntiles;
  const int NB = a.Npad / 32;
  hipError_t ae = hipSuccess;
#define FP_STEM_CASE(KSV, NBV)                                                                                       \
  {                                                                                                                  \
    if (lds > 64 * 1024)                                                                                             \
      ae = hipFuncSetAttribute((const void*)stem_conv_kernel<KSV, NBV, U8>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)lds);                                                                            \
    if (ae == hipSuccess) hipLaunchKernelGGL((stem_conv_kernel<KSV, NBV, U8>), dim3(grid), dim3(256), lds, s, a);    \
  }
  if (op.KH == 3 && NB == 1) FP_STEM_CASE(3, 1)
  else if (op.KH == 3 && NB == 2) FP_STEM_CASE(3, 2)
  else if (op.KH == 5 && NB == 1) FP_STEM_CASE(5, 1)
  else if (op.KH == 5 && NB == 2) FP_STEM_CASE(5, 2)
  else return FP_ERR_UNSUPPORTED;
#undef FP_STEM_CASE
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_stem(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  StemArgs a;
  stem_fill(op, weights, arena, a);
  a.in = arena + op.in_off;
  return stem_launch<false>(op, a, s);
}

bool fp_stem_u8_band_eligible(const fp_op& op);
int fp_launch_stem_u8_band(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, hipStream_t s);

// FP_OP_STEM_U8 (include/facepath.h): the same conv with the H x W input resampled from u8 frames while it is staged.
bool fp_stem_u8_shape_ok(const fp_op& op) { return op.H + op.W <= 2048 && stem_shape_ok(op, op.H + op.W); }

int fp_launch_stem_u8(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, int n_ext, hipStream_t s) {
  const long e = op.in_off;
  if (e < 0 || e + 2 >= n_ext || !ext) return FP_ERR_INVALID_ARG;
  const int fh = op.res_H, fw = op.res_W;
  if (op.Cin != 3 || fh <= 0 || fw < 3 || !fp_stem_u8_shape_ok(op)) return FP_ERR_UNSUPPORTED;
  if (ext[e].bytes < (size_t)op.N * fh * fw * 3 || ext[e + 1].bytes < (size_t)(op.H + op.W + 2) * 8 ||
      ext[e + 2].bytes < 256 * sizeof(float) || !ext[e].ptr || !ext[e + 1].ptr || !ext[e + 2].ptr)
    return FP_ERR_BOUNDS;
  if (fp_stem_u8_band_eligible(op)) return fp_launch_stem_u8_band(op, weights, arena, ext, s);
  StemArgs a;
  stem_fill(op, weights, arena, a);
  a.frames = (const uint8_t*)ext[e].ptr;
  a.tabs = (const fp_lb_tap*)ext[e + 1].ptr;
  a.lut = (const float*)ext[e + 2].ptr;
  a.row_bytes = (long)fw * 3;
  a.frame_bytes = (long)fh * fw * 3;
  a.frame_h = fh;
  a.frame_w = fw;
  return stem_launch<true>(op, a, s);
}

// ---------------------------------------------------------------------------------------------------------------
// BlazeFace's stem on u8 frames, band form (FP_OP_STEM_U8 with KxK = 5x5, OW = 128, Cout = 24, bias + ReLU): the
// generic kernel above stages the 5 canvas rows of every output row afresh -- with the letterbox fused in, every canvas
// pixel is RESAMPLED 2.5 times (~70 VALU instructions each, and fp32 MFMAs do not hide VALU work: tools/lab/coexec_lab).
// Here a workgroup owns a band of R consecutive output rows of one image and keeps the canvas rows in an 8-row LDS ring:
// output row oy reads ring rows 2oy-1 .. 2oy+3 while the two rows the NEXT output row adds (2oy+4, 2oy+5) are resampled
// into the ring -- one resample per canvas pixel (+ 3 rows per band), one workgroup barrier per output row.
//   * wave w owns output pixels 32w .. 32w+31 of the row; the 5x5 weights of its MFMAs live in registers;
//   * D^T = W^T x A^T (operands swapped), so a lane ends up with 16-byte channel pieces of ITS pixel: bias + ReLU on
//     float4s, three ds_write_b128 into the wave's private output tile, three coalesced 16-byte global stores;
//   * same k order (tap pairs 2kq + h, three channels per tap) and the same resampling code as the generic kernel and
//     the stand-alone letterbox: bit-identical results.
#ifndef FP_STEM_ACC2
#define FP_STEM_ACC2 0
#endif
namespace {

struct StemBandArgs {
  float* out;
  const float* w;        // packed [Kpad/4][32][4], Kpad = 104
  const float* bias;     // [24]
  const uint8_t* frames;
  const fp_lb_tap* tabs; // [256] column taps, [256] row taps, trailer, geometry
  const float* lut;
  long frame_bytes, row_bytes, out_ns;
  int frame_h, frame_w, out_rp;   // out_rp: output row pitch in floats ((OW + 1) * 24 row-padded, OW * 24 dense)
  int R, bands;                   // rows per band, bands per image
};

constexpr int SB_W = 256, SB_OW = 128, SB_C = 24, SB_WP = 260, SB_KQ = 13;   // ring row: columns -1 .. 258 (zero borders)

__global__ __launch_bounds__(256, 3) void stem5_u8_band_kernel(StemBandArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ring = smem;                                   // [8][SB_WP][4]
  float* Ot = Ring + 8 * SB_WP * 4;                     // 4 wave tiles [32][24]
  float* LutS = Ot + 4 * 32 * SB_C;                     // [256]
  fp_lb_tap* TabS = (fp_lb_tap*)(LutS + 256);           // [512]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  if (!fp_lb_geometry_ok(p.tabs, SB_W, SB_W, p.frame_h, p.frame_w)) return;   // tables of another geometry (uniform)
  LutS[tid] = p.lut[tid];
  for (int i = tid; i < 2 * SB_W; i += 256) TabS[i] = p.tabs[i];
  const fp_lb_tap tr = p.tabs[2 * SB_W];
  const int pad_value = tr.a, swap_rb = tr.b;
  for (int i = tid; i < 8 * SB_WP; i += 256) *(f32x4*)&Ring[i * 4] = z4;       // borders (columns -1, 256 .. 258) stay zero

  // weights of this lane's MFMAs: k-quad 2kq + h (= tap 2kq + h), output channel lr
  f32x4 wv[SB_KQ];
#pragma unroll
  for (int kq = 0; kq < SB_KQ; ++kq) wv[kq] = *(const f32x4*)(p.w + ((kq * 2 + h) * 32 + lr) * 4);
  f32x4 bias4[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) bias4[j] = *(const f32x4*)(p.bias + 8 * j + 4 * h);

  const int img = blockIdx.x / p.bands, band = blockIdx.x - img * p.bands;
  const int oy0 = band * p.R;
  const uint8_t* frame = p.frames + (long)img * p.frame_bytes;
  float* outi = p.out + (long)img * p.out_ns;
  __syncthreads();

  // canvas row iy (column tid) -> ring: resampled inside the canvas, zero outside it (the conv's F.pad(1, 2, 1, 2))
  fp_lb_raw raw[2];
  fp_lb_tap yt[2];
  const fp_lb_tap xt = TabS[tid];
  auto issue_rows = [&](int iy0) {          // rows iy0, iy0 + 1
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = min(max(iy0 + j, 0), SB_W - 1);
      yt[j] = TabS[SB_W + iy];
      raw[j] = fp_lb_issue(frame, p.row_bytes, xt, yt[j]);
    }
  };
  auto finish_rows = [&](int iy0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = iy0 + j;
      const f32x4 v = fp_lb_finish(raw[j], xt, yt[j], LutS, pad_value, swap_rb);
      *(f32x4*)&Ring[((iy & 7) * SB_WP + 1 + tid) * 4] = (unsigned)iy < (unsigned)SB_W ? v : z4;
    }
  };
  // band prologue: rows 2*oy0 - 1 .. 2*oy0 + 3 (the pair starting at 2*oy0 + 4 belongs to the first step)
  issue_rows(2 * oy0 - 1);
  finish_rows(2 * oy0 - 1);
  issue_rows(2 * oy0 + 1);
  finish_rows(2 * oy0 + 1);
  {
    const int iy = 2 * oy0 + 3, iyc = min(iy, SB_W - 1);
    yt[0] = TabS[SB_W + iyc];
    raw[0] = fp_lb_issue(frame, p.row_bytes, xt, yt[0]);
    const f32x4 v = fp_lb_finish(raw[0], xt, yt[0], LutS, pad_value, swap_rb);
    *(f32x4*)&Ring[((iy & 7) * SB_WP + 1 + tid) * 4] = iy < SB_W ? v : z4;
  }
  __syncthreads();

  float* Ow = Ot + wave * (32 * SB_C);
  const int px_off = ((wave * 32 + lr) * 2) * 4;        // this lane's output pixel: canvas column 2*ox - 1 -> ring column 2*ox
  for (int s = 0; s < p.R; ++s) {
    const int oy = oy0 + s;
    const int r0 = 2 * oy - 1;                           // first of the 5 canvas rows of this output row
    const bool more = s + 1 < p.R;
    if (more) issue_rows(r0 + 5);                        // rows 2oy + 4, 2oy + 5: in flight during the MFMAs
    int rb[5];                                           // float offsets of ring rows r0 .. r0 + 4 (wave-uniform)
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) rb[ky] = ((r0 + ky) & 7) * (SB_WP * 4);
    f32x16 acc, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f, acc1[i] = 0.f;
#pragma unroll
    for (int kq = 0; kq < SB_KQ; ++kq) {
      // tap 2kq + h (a tap past the 25th meets zero weights: any finite pixel will do -> tap 0)
      const int t0 = 2 * kq < 25 ? 2 * kq : 0, t1 = 2 * kq + 1 < 25 ? 2 * kq + 1 : 0;
      const int o0 = rb[t0 / 5] + (t0 % 5) * 4, o1 = rb[t1 / 5] + (t1 % 5) * 4;
      const f32x4 a = *(const f32x4*)&Ring[(h ? o1 : o0) + px_off];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
#if FP_STEM_ACC2
        // two accumulators, strictly alternating (a 32x32x2 MFMA that accumulates into the previous one's result issues at
        // half rate, common.h): MFMA i of the k sequence goes to accumulator i & 1, the two are summed once at the end
        if ((3 * kq + e) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kq][e], a[e], acc1, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kq][e], a[e], acc, 0, 0, 0);
        FP_MFMA_ORDER();
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[kq][e], a[e], acc, 0, 0, 0);
#endif
      }
    }
#if FP_STEM_ACC2
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += acc1[i];
#endif
    if (more) finish_rows(r0 + 5);                       // other rows than the ones any wave is reading in this step
    // bias + ReLU on this lane's pixel, channels 8j + 4h .. + 3 -> the wave's tile -> 3 x 16-byte coalesced stores
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float x = acc[4 * j + e] + bias4[j][e];
        v[e] = x > 0.f ? x : 0.f;
      }
      *(f32x4*)&Ow[lr * SB_C + 8 * j + 4 * h] = v;
    }
    float* orow = outi + (long)oy * p.out_rp + wave * (32 * SB_C);
#pragma unroll
    for (int j = 0; j < 3; ++j) *(f32x4*)(orow + (lane + 64 * j) * 4) = *(const f32x4*)&Ow[(lane + 64 * j) * 4];
    __syncthreads();                                     // the next step reads the rows just written
  }
}

}  // namespace


// ---------------------------------------------------------------------------------------------------------------
// The band form with the conv on the bf16 matrix cores (FP_OP_STEM_U8 + FP_OPF_SPLIT3; fp32-equivalent split arithmetic,
// split.h).  The fp32 band kernel above is bound by its 39 fp32 MFMAs per 32 pixels: an fp32 MFMA is a vector-ALU
// instruction of 64 cycles (FINDINGS 18), 168 of the kernel's 223 us.  Here
//   * every canvas pixel is split ONCE, where it is resampled, and the ring holds three bf16 planes of dense RGB rows
//     ([plane][8 rows][260 columns x 3 channels]): 14 VALU instructions per canvas pixel instead of 36 per gathered fragment;
//   * K = 75 is ordered (ky, kx, c) with every ky padded to 16 (zero weights behind the 15): three 32-k slabs, slab s =
//     canvas rows 2s and 2s + 1 of the window.  With stride 2 the 16 k of one ky are CONTIGUOUS in a ring row (offset 6 ox
//     + 3 kx + c), so a lane's fragment (pixel l15 of a 16-pixel tile, 8 k) is 16 bytes at a 4-byte boundary: four
//     ds_read_b32 per plane, no im2col;
//   * D^T = W^T x A^T as everywhere: the weights (18 fragments of three planes, 72 registers) are the A operand, a lane ends
//     up with four consecutive channels of ITS pixel and stores them straight to memory (16-byte pieces, 64 + 32 contiguous
//     bytes per pixel: blazepair's direct store, FINDINGS 37);
//   * 72 MFMAs of 16 cycles per wave and output row on the matrix pipe beside the resampling, against 39 of 64 on the
//     vector ALU in front of it.
// Products and sum order differ from the fp32 kernels: results agree to fp32 rounding, not bit for bit (tests: 1e-5 of the
// output scale against the fp32 band kernel, and the reference goldens through the whole detector).
namespace {

constexpr int SX_ROWE = SB_WP * 3;            // bf16 elements per ring row (260 columns x 3 channels)
constexpr int SX_PL = 8 * SX_ROWE;            // per plane
static_assert((SX_ROWE * 2) % 4 == 0, "ring rows start at 4-byte boundaries");

__global__ __launch_bounds__(256, 3) void stem5_u8_x6_kernel(StemBandArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x6[];
  unsigned short* Ring = (unsigned short*)smem_x6;                 // [3][8][SX_ROWE] bf16
  float* LutS = (float*)(Ring + 3 * SX_PL);                        // [256]
  fp_lb_tap* TabS = (fp_lb_tap*)(LutS + 256);                      // [512]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  if (!fp_lb_geometry_ok(p.tabs, SB_W, SB_W, p.frame_h, p.frame_w)) return;   // tables of another geometry (uniform)
  LutS[tid] = p.lut[tid];
  for (int i = tid; i < 2 * SB_W; i += 256) TabS[i] = p.tabs[i];
  const fp_lb_tap tr = p.tabs[2 * SB_W];
  const int pad_value = tr.a, swap_rb = tr.b;
  for (int i = tid; i < 3 * SX_PL / 2; i += 256) ((unsigned*)Ring)[i] = 0u;   // borders (columns -1, 256 .. 258) stay zero

  // weights: slab s, channel tile nt (16 nt + l15), plane, k = 8 q .. + 7 of the slab.  Blob: [3 s][2 nt][3 planes][16][32] bf16
  fp_frag3 w3[3][2];
  {
    const unsigned short* wp = (const unsigned short*)p.w;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const unsigned short* b = wp + (((sl * 2 + nt) * 3) * 16 + l15) * 32 + 8 * q;
        w3[sl][nt].h = *(const u32x4*)b;
        w3[sl][nt].m = *(const u32x4*)(b + 16 * 32);
        w3[sl][nt].l = *(const u32x4*)(b + 2 * 16 * 32);
      }
  }
  f32x4 bias4[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bias4[nt] = (nt == 0 || q < 2) ? *(const f32x4*)(p.bias + 16 * nt + 4 * q) : z4;

  const int img = blockIdx.x / p.bands, band = blockIdx.x - img * p.bands;
  const int oy0 = band * p.R;
  const uint8_t* frame = p.frames + (long)img * p.frame_bytes;
  float* outi = p.out + (long)img * p.out_ns;
  __syncthreads();

  // canvas row iy (column tid) -> ring: resampled inside the canvas (zero outside it: the conv's F.pad(1, 2, 1, 2)), split
  // into its three bf16 pieces, 3 x 3 two-byte stores
  const fp_lb_tap xt = TabS[tid];
  auto put = [&](int iy, const f32x4 v) {
    unsigned h01, m01, l01, h2, m2, l2;
    fp_split_pair(v[0], v[1], h01, m01, l01);
    fp_split_one(v[2], h2, m2, l2);
    unsigned short* d = Ring + (iy & 7) * SX_ROWE + (1 + tid) * 3;
    d[0] = (unsigned short)h01, d[1] = (unsigned short)(h01 >> 16), d[2] = (unsigned short)h2;
    d[SX_PL] = (unsigned short)m01, d[SX_PL + 1] = (unsigned short)(m01 >> 16), d[SX_PL + 2] = (unsigned short)m2;
    d[2 * SX_PL] = (unsigned short)l01, d[2 * SX_PL + 1] = (unsigned short)(l01 >> 16), d[2 * SX_PL + 2] = (unsigned short)l2;
  };
  auto issue_rows = [&](int iy0, fp_lb_raw (&raw)[2]) {          // rows iy0, iy0 + 1
#pragma unroll
    for (int j = 0; j < 2; ++j) raw[j] = fp_lb_issue(frame, p.row_bytes, xt, TabS[SB_W + min(max(iy0 + j, 0), SB_W - 1)]);
  };
  auto finish_rows = [&](int iy0, const fp_lb_raw (&raw)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int iy = iy0 + j;
      const f32x4 v = fp_lb_finish(raw[j], xt, TabS[SB_W + min(max(iy, 0), SB_W - 1)], LutS, pad_value, swap_rb);
      put(iy, (unsigned)iy < (unsigned)SB_W ? v : z4);
    }
  };
  fp_lb_raw rawa[2];
  // band prologue: rows 2*oy0 - 1 .. 2*oy0 + 3 (the pair starting at 2*oy0 + 4 belongs to the first step)
  issue_rows(2 * oy0 - 1, rawa);
  finish_rows(2 * oy0 - 1, rawa);
  issue_rows(2 * oy0 + 1, rawa);
  finish_rows(2 * oy0 + 1, rawa);
  {
    const int iy = 2 * oy0 + 3, iyc = min(iy, SB_W - 1);
    const fp_lb_tap yt = TabS[SB_W + iyc];
    const fp_lb_raw r = fp_lb_issue(frame, p.row_bytes, xt, yt);
    const f32x4 v = fp_lb_finish(r, xt, yt, LutS, pad_value, swap_rb);
    put(iy, iy < SB_W ? v : z4);
  }
  __syncthreads();

  // this lane's fragments: output pixel ox = 32 wave + 16 t + l15 -> ring offset 6 ox (+ 8 for the second half of a ky's 16 k)
  const int frag_off = 6 * (32 * wave + l15) + 8 * (q & 1);
  for (int s = 0; s < p.R; ++s) {
    const int oy = oy0 + s;
    const int r0 = 2 * oy - 1;                           // first of the 5 canvas rows of this output row
    const bool more = s + 1 < p.R;
    if (more) issue_rows(r0 + 5, rawa);                  // rows 2oy + 4, 2oy + 5: in flight during the MFMAs (requesting them a
                                                         // step earlier changes nothing: 196-201 against 190 us, same step time)
    f32x4 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = z4;
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      // slab sl: canvas rows r0 + 2 sl (k 0 .. 15) and r0 + 2 sl + 1 (k 16 .. 31); the sixth row does not exist (zero weights:
      // any finite data will do -> row 4 again)
      const int ky = min(2 * sl + (q >> 1), 4);
      const unsigned short* rowp = Ring + ((r0 + ky) & 7) * SX_ROWE + frag_off;
      fp_frag3 pf[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned* a = (const unsigned*)(rowp + 96 * t);      // 16 pixels further: 6 x 16 elements
        const unsigned* b = (const unsigned*)(rowp + 96 * t + SX_PL);
        const unsigned* c = (const unsigned*)(rowp + 96 * t + 2 * SX_PL);
        pf[t].h = u32x4{a[0], a[1], a[2], a[3]};
        pf[t].m = u32x4{b[0], b[1], b[2], b[3]};
        pf[t].l = u32x4{c[0], c[1], c[2], c[3]};
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) fp_mfma_x6_2a(w3[sl][0], w3[sl][1], pf[t].h, pf[t].m, pf[t].l, acc[t][0], acc[t][1]);
    }
    if (more) finish_rows(r0 + 5, rawa);                 // other rows than the ones any wave is reading in this step
    // bias + ReLU, lane (l15, q): pixel 32 wave + 16 t + l15, channels 16 nt + 4 q .. + 3 (nt = 1: q < 2)
    float* orow = outi + (long)oy * p.out_rp + (32 * wave + l15) * SB_C + 4 * q;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = acc[t][nt][e] + bias4[nt][e];
          v[e] = x > 0.f ? x : 0.f;
        }
        if (nt == 0 || q < 2) *(f32x4*)(orow + 16 * t * SB_C + 16 * nt) = v;
      }
    __syncthreads();                                     // the next step reads the rows just written
  }
}

}  // namespace

bool fp_stem_u8_band_eligible(const fp_op& op) {
  return op.kind == FP_OP_STEM_U8 && op.KH == 5 && op.KW == 5 && op.stride == 2 && op.pad_t == 1 && op.pad_l == 1 &&
         op.H == SB_W && op.W == SB_W && op.OH == SB_OW && op.OW == SB_OW && op.Cout == SB_C && op.out_ld == SB_C &&
         op.scale_off < 0 && op.bias_off >= 0 && op.act == FP_ACT_RELU && op.res_mode == FP_RES_NONE && op.N >= 16;
}

int fp_launch_stem_u8_band(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, hipStream_t s) {
  const long e = op.in_off;
  StemBandArgs a;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.bias = weights + op.bias_off;
  a.frames = (const uint8_t*)ext[e].ptr;
  a.tabs = (const fp_lb_tap*)ext[e + 1].ptr;
  a.lut = (const float*)ext[e + 2].ptr;
  a.frame_h = op.res_H;
  a.frame_w = op.res_W;
  a.row_bytes = (long)op.res_W * 3;
  a.frame_bytes = (long)op.res_H * op.res_W * 3;
  a.out_ns = op.out_ns;
  a.out_rp = ((op.flags & FP_OPF_OUT_ROWPAD) ? SB_OW + 1 : SB_OW) * SB_C;
#ifndef FP_STEM_BAND_ROWS
#define FP_STEM_BAND_ROWS 16
#endif
  a.R = FP_STEM_BAND_ROWS;
  a.bands = SB_OW / a.R;
  if (op.flags & FP_OPF_SPLIT3) {
    const size_t lds6 = (size_t)3 * SX_PL * 2 + 256 * 4 + 8 * (size_t)(2 * SB_W);
    hipLaunchKernelGGL(stem5_u8_x6_kernel, dim3(op.N * a.bands), dim3(256), lds6, s, a);
    FP_CHECK_LAUNCH();
    return FP_OK;
  }
  const size_t lds = 4 * ((size_t)8 * SB_WP * 4 + 4 * 32 * SB_C + 256) + 8 * (size_t)(2 * SB_W);
  hipLaunchKernelGGL(stem5_u8_band_kernel, dim3(op.N * a.bands), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
