// stemdw.hip — Mobile-FaceNet's first two layers in one kernel (gfx950):
//     conv1    = Conv_block(3, 64, 3x3, stride 2, pad 1): conv -> BN -> PReLU      (mobile_facenet.py:107, :141)
//     conv2_dw = Conv_block(64, 64, 3x3, stride 1, pad 1, groups 64): depthwise conv -> BN -> PReLU  (:108, :142)
// FP_OP_CONV + FP_OPF_OUT_DW (include/facepath.h).  conv1's 56 x 56 x 64 output (803 KB per crop) used to be written by the
// stem kernel (stem.hip, at the rate this part writes memory: 123 us per 512 crops) and read back by the depthwise conv in
// the prologue of conv_23's kernel (dwblockx6.hip FP_OPF_IN_DW: an LDS image of conv1's rows, nine b128 reads per element,
// +143 us on that kernel).  Here conv1's rows never leave the CU: the depthwise conv runs on them in LDS and its output --
// the same bytes the stem alone wrote -- is what goes to memory; conv_23 then runs in its plain form.
//   tile     = RB = 4 output rows x 56 columns x 64 channels of one crop, one 256-thread workgroup, two per CU (<= 78 KiB)
//   stage    = the 13 input rows (4-float pixels, zero borders) of the tile's 6 conv1 rows (halo rows recomputed: 6 for 4);
//              persistent workgroups: the NEXT tile's rows are loaded into registers while the current tile computes (a tile's
//              23 KB arriving as six dependent round trips per thread was the first form's long pole: 296 us), conv1's
//              weights live in registers as MFMA fragments, all other parameters in LDS
//   conv1    = stem.hip's arithmetic: a tap of 4 channels is one LDS pixel = one MFMA fragment quad,
//              v_mfma_f32_32x32x2_f32, k = tap * 4 + c, the pad channel's MFMA skipped -- with the two operands SWAPPED
//              (D^T = W^T A^T) and the k steps dealt alternately to two accumulators (a chained 32x32x2 MFMA issues at half rate) so that a lane holds four
//              consecutive channels of ONE pixel: BN + PReLU on float4s and four 16-byte LDS writes per 32-pixel tile instead of
//              sixteen scalar ones with per-element index arithmetic (296 -> 1xx us at 512 crops);
//              two passes of 32 output channels (the 6 x 58 conv1 image, 36 floats per pixel so that 16-byte accesses of
//              consecutive pixels fall into different banks, is 50 KiB; 64 channels would not leave room for a second workgroup)
//   dw       = lane = (4 channels, column), marching down the 4 rows with a 3-row window in registers; fma chain over the
//              taps in dwblockx6.hip's order, BN as acc * s + b, PReLU as v + (slope - 1) * min(v, 0)
//   output   = 16-byte stores, 128 contiguous bytes per pixel and pass
#include "split.h"

namespace {

struct StemDwArgs {
  const float* in;     // [N][112][112][4]
  float* out;          // [N][56][56][64]
  const float* w;      // conv1 packed [Kpad/4][64][4] (pack_conv_weight: k-quad = tap); X6: three bf16 planes [4 nt][3][16 ch][32 k]
  const float* scale;  // [64] conv1 BN scale, bias, PReLU slope
  const float* bias;
  const float* slope;
  const float* dw;     // [12][64]: nine depthwise taps, BN scale, BN bias, PReLU slope
  long in_ns, out_ns;
  int N;
};

constexpr int SD_H = 112, SD_OH = 56, SD_C = 64, SD_RB = 4, SD_NBAND = SD_OH / SD_RB;
constexpr int SD_IR = 2 * (SD_RB + 2) + 1;           // 13 input rows
constexpr int SD_IW = SD_H + 2;                       // 114 staged pixels per row: column -1 .. 112
constexpr int SD_CR = SD_RB + 2, SD_CW = SD_OH + 2;   // conv1 image: 6 rows x 58 columns (zero border columns)
constexpr int SD_LDC = 36;                            // floats per conv1-image pixel (32 channels + 4: an odd number of 16-byte units)
constexpr int SD_IMG = SD_IR * SD_IW * 4, SD_C1 = SD_CR * SD_CW * SD_LDC, SD_PAR = 3 * SD_C, SD_DWP = 12 * SD_C;
constexpr int SD_LDS = 4 * (SD_IMG + SD_C1 + SD_PAR + SD_DWP);
constexpr int SD_PF = (SD_IR * SD_IW + 255) / 256;    // staged float4s per thread and tile (6)
static_assert(2 * SD_LDS <= 160 * 1024, "two workgroups per CU");

// X6 = true (FP_OPF_SPLIT3): conv1 on the bf16 matrix cores with fp32-equivalent arithmetic (split.h): K = 27 = (tap, channel)
// flattened into ONE 32-k slab (zero weights behind it), 16-pixel tiles, the pixel fragments gathered value by value from the
// staged rows and split per tile and pass, weights as three bf16 planes in registers:
// 12 MFMAs of 16 cycles per 16 pixels and pass on the matrix pipe instead of 15 of 64 on the vector ALU.
template <bool X6>
__global__ __launch_bounds__(256, 2) void stemdw_kernel(StemDwArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Img = smem;              // [13][114][4]
  float* C1 = Img + SD_IMG;       // [6][58][36]
  float* Par = C1 + SD_C1;        // [3][64]: conv1 BN scale, bias, PReLU slope
  float* Dwp = Par + SD_PAR;      // [12][64]: depthwise taps, BN scale, BN bias, PReLU slope
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  const int ntiles = p.N * SD_NBAND, G = gridDim.x;

  // ---- once per workgroup: parameters -> LDS, conv1 weights -> registers, zero border columns of the conv1 image ----
  if (tid < SD_PAR) Par[tid] = tid < SD_C ? p.scale[tid] : tid < 2 * SD_C ? p.bias[tid - SD_C] : p.slope[tid - 2 * SD_C];
  for (int i = tid; i < SD_DWP / 4; i += 256) *(f32x4*)&Dwp[i * 4] = *(const f32x4*)(p.dw + i * 4);
  if (tid < SD_CR * 2 * 8) {      // columns -1 and 56 of every conv1 row: 32 channels = 8 float4 each
    const int r = tid / 16, e = tid - r * 16, side = e >> 3, q = e & 7;
    *(f32x4*)&C1[((r * SD_CW) + (side ? SD_CW - 1 : 0)) * SD_LDC + q * 4] = z4;
  }
  // A operand of the swapped MFMA = conv1's weights: fragment (k-quad = tap 2 kq + h, row = channel 32 pass + lr); the tenth
  // tap does not exist: zeros.  Packed blob: [k-quad = tap][64][4]
  f32x4 wf[X6 ? 1 : 2][X6 ? 1 : 5];
  fp_frag3 w3[X6 ? 4 : 1];            // X6: channel tile nt = 16 nt + l15, k = 8 q .. + 7
  const int l15 = lane & 15, q = lane >> 4;
  int koff[8];                        // X6: float offset of k = 8 q + i = (tap, channel) inside a pixel's 3 x 3 window
  if (X6) {
    const unsigned short* wp = (const unsigned short*)p.w;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      w3[nt].h = *(const u32x4*)(wp + ((nt * 3 + 0) * 16 + l15) * 32 + 8 * q);
      w3[nt].m = *(const u32x4*)(wp + ((nt * 3 + 1) * 16 + l15) * 32 + 8 * q);
      w3[nt].l = *(const u32x4*)(wp + ((nt * 3 + 2) * 16 + l15) * 32 + 8 * q);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 8 * q + i, tap = k < 27 ? k / 3 : 0, c = k < 27 ? k - tap * 3 : 0;
      koff[i] = ((tap / 3) * SD_IW + tap % 3) * 4 + c;
    }
  } else {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int kq = 0; kq < 5; ++kq) {
        const int t = 2 * kq + h;
        const f32x4 v = *(const f32x4*)(p.w + ((long)(t < 9 ? t : 0) * SD_C + pass * 32 + lr) * 4);
        wf[pass][kq] = t < 9 ? v : z4;
      }
  }

  // staging of a tile's input rows 2 r0 - 3 .. 2 r0 + 9, columns -1 .. 112: thread -> float4 slots tid + 256 j
  int srow[SD_PF], scol[SD_PF];
#pragma unroll
  for (int j = 0; j < SD_PF; ++j) {
    const int i = tid + 256 * j;
    srow[j] = i / SD_IW;
    scol[j] = i - srow[j] * SD_IW - 1;
  }
  f32x4 pre[SD_PF];
  unsigned premask = 0;
  auto issue_stage = [&](int tile) {      // loads only: the zero padding is applied when the registers are written to LDS
    const int img = tile / SD_NBAND, r0 = (tile - img * SD_NBAND) * SD_RB;
    const float* ib = p.in + (long)img * p.in_ns;
    premask = 0;
#pragma unroll
    for (int j = 0; j < SD_PF; ++j) {
      const int iy = 2 * r0 - 3 + srow[j], ix = scol[j];
      const bool ok = tid + 256 * j < SD_IR * SD_IW && (unsigned)iy < (unsigned)SD_H && (unsigned)ix < (unsigned)SD_H;
      pre[j] = *(const f32x4*)(ib + ((long)min(max(iy, 0), SD_H - 1) * SD_H + min(max(ix, 0), SD_H - 1)) * 4);
      if (ok) premask |= 1u << j;
    }
  };
  auto write_stage = [&]() {
#pragma unroll
    for (int j = 0; j < SD_PF; ++j)
      if (tid + 256 * j < SD_IR * SD_IW) *(f32x4*)&Img[(tid + 256 * j) * 4] = ((premask >> j) & 1u) ? pre[j] : z4;
  };

  // depthwise item of this thread: channels 4 c4 .. + 3 of a pass's 32, columns col0 and col0 + 32
  const int c4 = tid & 7, col0 = tid >> 3;

  int tile = (int)fp_xcd_block();
  if (tile < ntiles) {
    issue_stage(tile);
    write_stage();
  }
  for (; tile < ntiles; tile += G) {
    const int img = tile / SD_NBAND, r0 = (tile - img * SD_NBAND) * SD_RB;       // output rows r0 .. r0 + 3
    float* ob = p.out + (long)img * p.out_ns;
    const bool more = tile + G < ntiles;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();     // pass 0: this tile's input rows are in LDS; both: the previous depthwise reads of C1 are done
      if (pass == 0 && more) issue_stage(tile + G);      // in flight during both conv1 passes
      // ---- conv1 for channels 32 pass .. + 31: 6 x 56 = 336 pixels = 10.5 tiles of 32, wave w owns tiles w, w + 4, w + 8 ----
      // swapped operands: A = weights (row = channel lr of the pass), B = pixels (column = pixel lr of the tile); lane (lr, h)
      // ends up with PIXEL lr and channels (reg & 3) + 8 (reg >> 2) + 4 h: 16-byte pieces of the pixel's channel row
      if (X6) {
        // 336 pixels = 21 tiles of 16; wave w owns tiles w, w + 4, ... (6 / 5 / 5 / 5); lane (l15, q) ends up with PIXEL l15 and
        // channels 4 q .. + 3 of each 16-channel tile
        constexpr int NT16 = SD_CR * SD_OH / 16, NOWN = (NT16 + 3) / 4;
#pragma unroll
        for (int t = 0; t < NOWN; ++t) {
          const int mt = wave + 4 * t;
          if (mt < NT16) {                                           // wave-uniform
            const int m = mt * 16 + l15;
            const int cy = m / SD_OH, cx = m - cy * SD_OH;
            const float* base = Img + ((cy * 2) * SD_IW + cx * 2) * 4;
            f32x4 lo, hi;
#pragma unroll
            for (int i = 0; i < 4; ++i) lo[i] = base[koff[i]], hi[i] = base[koff[4 + i]];
            const fp_frag3 pf = fp_split8(lo, hi);    // (gathered and split again in the second pass: 8 LDS reads + 36 VALU per tile
                                                      // are cheaper than 72 registers held across the depthwise phase)
            f32x4 acc0 = z4, acc1 = z4;
            fp_mfma_x6_2a(w3[2 * pass], w3[2 * pass + 1], pf.h, pf.m, pf.l, acc0, acc1);
            const bool inside = (unsigned)(r0 - 1 + cy) < (unsigned)SD_OH;
            float* dst = C1 + (cy * SD_CW + cx + 1) * SD_LDC + 4 * q;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const int ch = pass * 32 + 16 * nt + 4 * q;
              const f32x4 sc = *(const f32x4*)&Par[ch], bi = *(const f32x4*)&Par[SD_C + ch], sl = *(const f32x4*)&Par[2 * SD_C + ch];
              const f32x4 av = nt ? acc1 : acc0;
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = av[e] * sc[e] + bi[e];
                v[e] = x > 0.f ? x : __builtin_fmaf(x, sl[e], 0.0f);
              }
              *(f32x4*)(dst + 16 * nt) = inside ? v : z4;
            }
          }
        }
      } else {
  #pragma unroll 1
        for (int mt = wave; mt * 32 < SD_CR * SD_OH; mt += 4) {
          const int m = min(mt * 32 + lr, SD_CR * SD_OH - 1);
          const int cy = m / SD_OH, cx = m - cy * SD_OH;             // conv1 pixel (row r0 - 1 + cy, column cx)
          const float* base = Img + ((cy * 2) * SD_IW + cx * 2) * 4;   // its window: staged rows 2 cy .. + 2, columns 2 cx .. + 2
          // two accumulators, strictly alternating: a 32x32x2 MFMA that accumulates into the previous MFMA's result issues at
          // half rate (common.h FP_MFMA_ORDER); even MFMAs of the k sequence go to acc0, odd ones to acc1, summed at the end
          f32x16 acc0, acc1;
  #pragma unroll
          for (int r = 0; r < 16; ++r) acc0[r] = 0.f, acc1[r] = 0.f;
  #pragma unroll
          for (int kq = 0; kq < 5; ++kq) {
            // fragment of 4 consecutive k = tap 2 kq + h (the tenth: zero weights, any staged pixel)
            const int t = 2 * kq + h;
            const int tt = t < 9 ? t : 0;
            const int ky = tt / 3, kx = tt - ky * 3;
            const f32x4 a = *(const f32x4*)(base + (ky * SD_IW + kx) * 4);
  #pragma unroll
            for (int e = 0; e < 3; ++e) {
              if ((3 * kq + e) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[pass][kq][e], a[e], acc1, 0, 0, 0);
              else acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[pass][kq][e], a[e], acc0, 0, 0, 0);
              FP_MFMA_ORDER();
            }
          }
          if (mt * 32 + lr < SD_CR * SD_OH) {
            // conv1 rows outside the image are the depthwise conv's zero padding
            const bool inside = (unsigned)(r0 - 1 + cy) < (unsigned)SD_OH;
            float* dst = C1 + (cy * SD_CW + cx + 1) * SD_LDC + 4 * h;
  #pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int ch = pass * 32 + 8 * j + 4 * h;
              const f32x4 sc = *(const f32x4*)&Par[ch], bi = *(const f32x4*)&Par[SD_C + ch], sl = *(const f32x4*)&Par[2 * SD_C + ch];
              f32x4 v;
  #pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x = (acc0[4 * j + e] + acc1[4 * j + e]) * sc[e] + bi[e];
                v[e] = x > 0.f ? x : __builtin_fmaf(x, sl[e], 0.0f);
              }
              *(f32x4*)(dst + 8 * j) = inside ? v : z4;
            }
          }
        }
      }
      __syncthreads();     // conv1 image of this pass complete; (pass 1) every wave is done reading the input rows
      if (pass == 1 && more) write_stage();              // the next tile's rows (loaded two conv1 passes ago) -> LDS

      // ---- depthwise 3x3 + BN + PReLU on the 32 channels, 4 rows x 56 columns ----
      {
        const float* dp = Dwp + pass * 32 + 4 * c4;
        f32x4 tap[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) tap[t] = *(const f32x4*)(dp + t * SD_C);
        const f32x4 one = {1.f, 1.f, 1.f, 1.f};
        const f32x4 dsc = *(const f32x4*)(dp + 9 * SD_C), dbi = *(const f32x4*)(dp + 10 * SD_C);
        const f32x4 dsl = *(const f32x4*)(dp + 11 * SD_C) - one;
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const int col = col0 + 32 * cp;
          if (col < SD_OH) {
            // conv1 pixel (r0 - 1 + yy, col + dx - 1) sits at C1[(yy * 58 + col + dx) * 36]
            const float* base = C1 + col * SD_LDC + 4 * c4;
            f32x4 w0[3], w1[3], w2[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              w0[dx] = *(const f32x4*)(base + dx * SD_LDC);
              w1[dx] = *(const f32x4*)(base + (SD_CW + dx) * SD_LDC);
            }
#pragma unroll
            for (int r = 0; r < SD_RB; ++r) {
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) w2[dx] = *(const f32x4*)(base + ((r + 2) * SD_CW + dx) * SD_LDC);
              f32x4 a = w0[0] * tap[0];
              a += w0[1] * tap[1];
              a += w0[2] * tap[2];
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) a += w1[dx] * tap[3 + dx];
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) a += w2[dx] * tap[6 + dx];
              f32x4 v = a * dsc + dbi, ng;
#pragma unroll
              for (int i = 0; i < 4; ++i) ng[i] = __builtin_fminf(v[i], 0.f);
              v = ng * dsl + v;
              *(f32x4*)(ob + ((long)(r0 + r) * SD_OH + col) * SD_C + pass * 32 + 4 * c4) = v;
#pragma unroll
              for (int dx = 0; dx < 3; ++dx) {
                w0[dx] = w1[dx];
                w1[dx] = w2[dx];
              }
            }
          }
        }
      }
    }
  }
}

}  // namespace

// Mobile-FaceNet's conv1 with conv2_dw behind it: FP_OP_CONV + FP_OPF_OUT_DW on a dense 112 x 112 four-float-pixel image.
bool fp_stemdw_supported(const fp_op& op) {
  if (op.kind != FP_OP_CONV || !(op.flags & FP_OPF_OUT_DW) || !(op.flags & FP_OPF_IN_C3)) return false;
  if (op.flags & ~(FP_OPF_OUT_DW | FP_OPF_IN_C3 | FP_OPF_SPLIT3)) return false;
  if (op.H != SD_H || op.W != SD_H || op.OH != SD_OH || op.OW != SD_OH || op.Cin != 4 || op.Cout != SD_C) return false;
  if (op.KH != 3 || op.KW != 3 || op.stride != 2 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.in_ld != 4 || op.out_ld != SD_C || op.out_cmul != 1) return false;
  if (op.act != FP_ACT_PRELU || op.res_mode != FP_RES_NONE) return false;
  if (op.scale_off < 0 || op.bias_off < 0 || op.slope_off < 0) return false;
  if (op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4 || op.w_off % 4 || op.slope_off % 4) return false;
  return op.in_ns >= (long)SD_H * SD_H * 4 && op.out_ns >= (long)SD_OH * SD_OH * SD_C;
}

// floats behind w_off: the packed fp32 weights (40 x 64), or with FP_OPF_SPLIT3 three bf16 planes [4][3][16][32]
long fp_stemdw_w_floats(const fp_op& op) { return (op.flags & FP_OPF_SPLIT3) ? 4 * 3 * 16 * 32 / 2 : 40 * 64; }

int fp_launch_stemdw(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_stemdw_supported(op)) return FP_ERR_UNSUPPORTED;
  StemDwArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.scale = weights + op.scale_off;
  a.bias = weights + op.bias_off;
  a.slope = weights + op.slope_off;
  a.dw = weights + op.slope_off + SD_C;      // the depthwise block follows conv1's slopes (facepath.h FP_OPF_OUT_DW)
  a.in_ns = op.in_ns;
  a.out_ns = op.out_ns;
  a.N = op.N;
  const bool x6 = (op.flags & FP_OPF_SPLIT3) != 0;
  const hipError_t ae = hipFuncSetAttribute(x6 ? (const void*)stemdw_kernel<true> : (const void*)stemdw_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SD_LDS);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int ntiles = op.N * SD_NBAND, grid = ntiles < 512 ? ntiles : 512;      // persistent: two workgroups per CU
  if (x6) hipLaunchKernelGGL(stemdw_kernel<true>, dim3(grid), dim3(256), SD_LDS, s, a);
  else hipLaunchKernelGGL(stemdw_kernel<false>, dim3(grid), dim3(256), SD_LDS, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
