// blazepairs2.hip -- a stride-1 24 -> 24 BlazeBlock and the STRIDE-2 BlazeBlock behind it in one kernel (gfx950).
//
//   y1 = ReLU( pw1( dw1(x) ) + x )                                             (fde/modules/blazeface/blazeface.py:12-47)
//   y2 = ReLU( pw2( dw2_s2( pad(y1, (0, 2, 0, 2)) ) ) + cpad( maxpool2x2(y1) ) )        (:34-47, stride == 2; C2 = 24 or 48)
//
// Every 24-channel stage of BlazeFace-back ends with a single stride-1 block (seven blocks = three FP_OP_BLAZEPAIRs + one)
// followed by the stride-2 block that halves the map (blazeface.py:122-152).  As two launches y1 is written (403 MB at
// 128 x 128, batch 256) and read back; here it lives in the LDS ring of blazepair.hip and only the quarter-size y2 leaves
// the CU: x in + y2 out = 503 MB instead of 1.3 GB at 128 x 128.
//   * block 1 is blazepair_kernel's block 1, line for line: a workgroup owns a band of R2 output rows of y2 = 2 R2 + 1 rows
//     of y1 (the last one is the stride-2 window's third row: the next band's first, or the zero row below the image), its
//     four waves are the four 32-pixel strips of a 128-wide y1 row (64-wide maps: two bands per workgroup, two strips each),
//     one y1 row per step into a ring of four rows;
//   * every second step, after the barrier, HALF the waves (an output row has half the pixels) make output row yo from
//     ring rows 2 yo .. 2 yo + 2: depthwise stride 2 (window of 9 columns x 3 rows per four output pixels, the two pad
//     columns / the pad row are the ring's zero borders), 12 (C2 = 48: 24) fp32 MFMAs, shortcut = max over the 2 x 2 ring
//     pixels for channels < 24 and 0 above, bias, ReLU, 32 x C2 tile -> coalesced 16-byte stores; the other waves run
//     ahead into the next step's block 1.
// Arithmetic per block identical to blazeblock_wp_kernel / blazeblock_persist_kernel<2, ...> (same tap order, same k order).
#include "common.h"

namespace {

struct BlazePairS2Args {
  const float* in;    // pixel (0, 0) of image 0, row-padded
  float* out;
  const float* wd;    // [2][9][C]
  const float* bd;    // [2][C]
  const float* wp;    // block 1: packed [C/4][32][4]; block 2 behind it: packed [C/4][Npad2][4], Npad2 = 32 (C2 = 24) / 64 (C2 = 48)
  const float* bp;    // [C] then [C2]
  int H, R, bands;    // R = output rows of y2 per band, bands per image ((H / 2) / R)
  int nbands;         // N * bands
  int in_rp, out_rp;  // row pitch, floats
  long in_ns, out_ns;
  fp_divisor bands_div;
};

template <int W, int C2>
__global__ __launch_bounds__(256, 2) void blazepair_s2_kernel(BlazePairS2Args p) {
  constexpr int C = 24, LDT = C + 4, C4 = C / 4, KG = C / 8, NS = W / 32, NSUB = 4 / NS, NS2 = NS / 2;
  constexpr int NB2 = C2 > 32 ? 2 : 1;                     // 32-column halves of block 2's 1x1
  constexpr int RROW = (W + 2) * C;                        // floats per ring row: pixels -1 .. W
  constexpr int RING = 4 * RROW;
  constexpr int PWF = KG * 2 * 32 * 4;                     // packed 1x1 weights of block 1 (and of one 32-column half of block 2)
  constexpr int OTF = C2 > C ? 32 * C2 : 0;                // separate output tile of a block-2 wave (C2 = 24: over its A tile)
  static_assert((1 + NB2) * PWF <= 4 * 32 * LDT + NSUB * NS2 * OTF, "weight staging fits the wave regions");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                        // [2][10][C] depthwise taps + bias
  float* Bp = Ws + 2 * 10 * C;                             // [32] block 1, [64] block 2
  float* Rg = Bp + 96;                                     // [NSUB][4][RROW]
  float* Av = Rg + NSUB * RING;                            // 4 wave regions [32][LDT], then the output tiles (first: weight staging)
  // (the region behind the A tiles, NSUB * NS2 * OTF floats, is weight staging only: y2 goes from the epilogue registers to
  // global memory)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < 2 * 10 * C / 4; i += 256) {
    const int b = i / (10 * C / 4), k = i - b * (10 * C / 4);
    *(f32x4*)&Ws[i * 4] = (k * 4 < 9 * C) ? *(const f32x4*)(p.wd + b * 9 * C + k * 4) : *(const f32x4*)(p.bd + b * C + (k * 4 - 9 * C));
  }
  if (tid < 96) Bp[tid] = tid < 32 ? (tid < C ? p.bp[tid] : 0.f) : (tid - 32 < C2 ? p.bp[C + tid - 32] : 0.f);
  for (int i = tid; i < (1 + NB2) * PWF / 4; i += 256) *(f32x4*)&Av[i * 4] = *(const f32x4*)(p.wp + i * 4);
  for (int i = tid; i < NSUB * RING / 4; i += 256) *(f32x4*)&Rg[i * 4] = z;     // pads (and everything else) zero
  __syncthreads();
  f32x4 bf1[KG], bf2[NB2][KG];                             // B fragments: k-quad 2*kq + h, column lr (+ 32 nb)
#pragma unroll
  for (int kq = 0; kq < KG; ++kq) {
    bf1[kq] = *(const f32x4*)&Av[((kq * 2 + h) * 32 + lr) * 4];
#pragma unroll
    for (int nb = 0; nb < NB2; ++nb) bf2[nb][kq] = *(const f32x4*)&Av[PWF + ((kq * 2 + h) * (32 * NB2) + 32 * nb + lr) * 4];
  }
  __syncthreads();                                         // staging area becomes the wave regions

  const int sub = wv / NS, strip = wv - sub * NS;
  float* At = Av + wv * (32 * LDT);                        // A tile [32][LDT]
  float* ring = Rg + sub * RING;
  const int x0 = strip * 32;
  const bool b2wave = strip < NS2;                         // this wave makes output pixels 32 strip .. + 31 of a y2 row

  // depthwise item of this lane: pixels 4g .. 4g+3 of the strip (block 2: of the output tile), channels 4c4 .. 4c4+3
  // (lanes >= 48 repeat item 0 and write nothing)
  const bool dw_lane = lane < 8 * C4;
  const int la = dw_lane ? lane : 0;
  const int g = la / C4, c4 = la - g * C4;
  const unsigned voff_in = (unsigned)((4 * g * C + 4 * c4) * 4);
  const float* wl1 = &Ws[4 * c4];
  const float* wl2 = &Ws[10 * C + 4 * c4];
  const f32x4 pbias1 = *(const f32x4*)&Bp[4 * c4];         // block 1's 1x1 bias rides its shortcut
  const int rg_dw = (x0 + 4 * g) * C + 4 * c4;             // ring column x0 + 4g - 1 of this lane's channels
  const int rg_ep = (x0 + 1 + lr) * C + 4 * h;             // block 1's epilogue: ring pixel x0 + lr, channels 4h (+ 8j)
  const int rg_dw2 = (64 * strip + 8 * g + 1) * C + 4 * c4;   // block 2: ring pixel 2 (32 strip + 4g) of this lane's channels
  const int rg_ep2 = (64 * strip + 2 * lr + 1) * C + 4 * h;   // block 2's shortcut: ring pixel 2 (32 strip + lr)

  // this wave's band: (image, band) -> first output row yo0 of y2; y1 rows 2 yo0 .. 2 yo0 + 2 R
  const int bi = min((int)blockIdx.x * NSUB + sub, p.nbands - 1);
  const bool live = (int)blockIdx.x * NSUB + sub < p.nbands;
  const unsigned img = __builtin_amdgcn_readfirstlane(fp_fastdiv((unsigned)bi, p.bands_div));
  const int yo0 = __builtin_amdgcn_readfirstlane((bi - (int)img * p.bands) * p.R);
  const int ya = 2 * yo0;                                  // first y1 row of the band
  const long in_rb = (long)p.in_rp * 4, out_rb = (long)p.out_rp * 4;
  const char* inb = (const char*)p.in + fp_uniform(((long)img * p.in_ns + (long)(x0 - 1) * C) * 4);           // (row 0, column x0 - 1)
  char* outb = (char*)p.out + fp_uniform(((long)img * p.out_ns + (long)(32 * (b2wave ? strip : 0)) * C2) * 4);   // (row 0, column 32 strip)

  // x window: ring of three rows in registers; at step i (row y = ya + i) rows y-1, y, y+1 sit in slots i%3, (i+1)%3, (i+2)%3
  f32x4 x[3][6];
  const int nsteps = 2 * p.R + 1;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const char* rowp = inb + fp_uniform((long)(ya - 1 + ky) * in_rb);
#pragma unroll
    for (int j = 0; j < 6; ++j) x[ky][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
  }

  for (int ib = 0; ib < nsteps; ib += 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = ib + r;
      if (i < nsteps) {
        const int s0 = r, s1 = (r + 1) % 3, s2 = (r + 2) % 3;   // static ring slots of rows y-1, y, y+1
        const int y = ya + i;
        float* ry = ring + ((y + 1) & 3) * RROW;                // ring row of y1 row y
        if (y < p.H) {
          // ---- block 1: depthwise -> A tile; shortcut (+ 1x1 bias) -> the ring slot y1 row y will occupy ----
          {
            const f32x4 dbias = *(const f32x4*)(wl1 + 9 * C);
            f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const int sl = ky == 0 ? s0 : ky == 1 ? s1 : s2;
              const f32x4 w0 = *(const f32x4*)(wl1 + (ky * 3 + 0) * C);
              const f32x4 w1 = *(const f32x4*)(wl1 + (ky * 3 + 1) * C);
              const f32x4 w2 = *(const f32x4*)(wl1 + (ky * 3 + 2) * C);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[q] += x[sl][q] * w0;
                acc[q] += x[sl][q + 1] * w1;
                acc[q] += x[sl][q + 2] * w2;
              }
            }
            if (dw_lane) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                *(f32x4*)&At[(4 * g + q) * LDT + 4 * c4] = acc[q];
                *(f32x4*)&ry[rg_dw + (q + 1) * C] = x[s1][q + 1] + pbias1;
              }
            }
          }
          // row y + 2 replaces row y - 1 in the register ring (if the next step computes a row)
          if (y + 1 < p.H && i + 1 < nsteps) {
            const char* rowp = inb + fp_uniform((long)(y + 2) * in_rb);
#pragma unroll
            for (int j = 0; j < 6; ++j) x[s0][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
          }
          f32x16 m0, m1;
#pragma unroll
          for (int k = 0; k < 16; ++k) m0[k] = 0.f, m1[k] = 0.f;
          const float* arow = &At[lr * LDT + 4 * h];
          // D^T = W^T x A^T (operands swapped): lane (lr, h) ends up with PIXEL lr and channels (k & 3) + 8*(k >> 2) + 4h
#pragma unroll
          for (int kq = 0; kq < KG; ++kq) {
            const f32x4 a = *(const f32x4*)(arow + kq * 8);
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1[kq][0], a[0], m0, 0, 0, 0);
            FP_MFMA_ORDER();
            m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1[kq][1], a[1], m1, 0, 0, 0);
            FP_MFMA_ORDER();
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1[kq][2], a[2], m0, 0, 0, 0);
            FP_MFMA_ORDER();
            m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf1[kq][3], a[3], m1, 0, 0, 0);
            FP_MFMA_ORDER();
          }
          // y1 = ReLU(1x1 + shortcut), in place in the ring: pixel x0 + lr, channels 8j + 4h .. + 3
          {
            float* rpx = ry + rg_ep;
#pragma unroll
            for (int j = 0; j < C / 8; ++j) {
              const f32x4 sv = *(const f32x4*)(rpx + 8 * j);
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (m0[4 * j + e] + m1[4 * j + e]) + sv[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
              *(f32x4*)(rpx + 8 * j) = v;
            }
          }
        } else {
          // y1 row H: the zero row of block 2's F.pad(.., (0, 2, 0, 2))
#pragma unroll
          for (int j = 0; j < 3; ++j) *(f32x4*)&ry[(x0 + 1) * C + (lane + 64 * j) * 4] = z;
        }
        __syncthreads();
        if (i >= 2 && !(i & 1) && b2wave) {
          // ---- block 2 (stride 2): output row yo from ring rows 2 yo, 2 yo + 1, 2 yo + 2 = y - 2, y - 1, y ----
          const int yo = yo0 + (i - 2) / 2;
          {
            const f32x4 dbias = *(const f32x4*)(wl2 + 9 * C);
            f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const float* rr = ring + ((y - 1 + ky) & 3) * RROW + rg_dw2;      // ring row of y1 row y - 2 + ky
              f32x4 xv[9];
#pragma unroll
              for (int j = 0; j < 9; ++j) xv[j] = *(const f32x4*)(rr + j * C);
              const f32x4 w0 = *(const f32x4*)(wl2 + (ky * 3 + 0) * C);
              const f32x4 w1 = *(const f32x4*)(wl2 + (ky * 3 + 1) * C);
              const f32x4 w2 = *(const f32x4*)(wl2 + (ky * 3 + 2) * C);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[q] += xv[2 * q] * w0;
                acc[q] += xv[2 * q + 1] * w1;
                acc[q] += xv[2 * q + 2] * w2;
              }
            }
            if (dw_lane) {
#pragma unroll
              for (int q = 0; q < 4; ++q) *(f32x4*)&At[(4 * g + q) * LDT + 4 * c4] = acc[q];
            }
          }
          const float* arow = &At[lr * LDT + 4 * h];
          const float* s00 = ring + ((y - 1) & 3) * RROW + rg_ep2;            // y1 row 2 yo, pixel 2 X
          const float* s10 = ring + (y & 3) * RROW + rg_ep2;                  // y1 row 2 yo + 1
          char* orow_g = outb + fp_uniform((long)yo * out_rb);
#pragma unroll
          for (int nb = 0; nb < NB2; ++nb) {
            f32x16 m0, m1;
#pragma unroll
            for (int k = 0; k < 16; ++k) m0[k] = 0.f, m1[k] = 0.f;
#pragma unroll
            for (int kq = 0; kq < KG; ++kq) {
              const f32x4 a = *(const f32x4*)(arow + kq * 8);
              m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[nb][kq][0], a[0], m0, 0, 0, 0);
              FP_MFMA_ORDER();
              m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[nb][kq][1], a[1], m1, 0, 0, 0);
              FP_MFMA_ORDER();
              m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[nb][kq][2], a[2], m0, 0, 0, 0);
              FP_MFMA_ORDER();
              m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[nb][kq][3], a[3], m1, 0, 0, 0);
              FP_MFMA_ORDER();
            }
            // y2 = ReLU(1x1 + bias + shortcut): channels 32 nb + 8j + 4h .. + 3 of output pixel 32 strip + lr; the shortcut is
            // the 2 x 2 max of y1 for channels < 24 and 0 above (blazeface.py:38-45)
            const int nj = (C2 - 32 * nb < 32 ? C2 - 32 * nb : 32) / 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j >= nj) break;
              const int oc = 32 * nb + 8 * j;               // (+ 4h: a lane's four channels are on one side of 24)
              f32x4 sv = *(const f32x4*)&Bp[32 + oc + 4 * h];
              if (oc < C) {
                const f32x4 a0 = *(const f32x4*)(s00 + 8 * j), a1 = *(const f32x4*)(s00 + C + 8 * j);
                const f32x4 b0 = *(const f32x4*)(s10 + 8 * j), b1 = *(const f32x4*)(s10 + C + 8 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) sv[e] += fmaxf(fmaxf(a0[e], a1[e]), fmaxf(b0[e], b1[e]));
              }
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (m0[4 * j + e] + m1[4 * j + e]) + sv[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
              // straight to global memory: a lane pair (h = 0, 1) writes 32 contiguous bytes of its pixel (as blazepair.hip)
              if (live) *(f32x4*)(orow_g + (unsigned)((lr * C2 + oc + 4 * h) * 4)) = v;
            }
          }
        }
      }
    }
  }
}

template <int W, int C2>
int launch_pair_s2(const BlazePairS2Args& a, hipStream_t s) {
  constexpr int C = 24, NSUB = 4 / (W / 32), NS2 = W / 64;
  const size_t lds = 4 * ((size_t)2 * 10 * C + 96 + (size_t)NSUB * 4 * (W + 2) * C + 4 * (size_t)32 * (C + 4) +
                          (C2 > C ? (size_t)NSUB * NS2 * 32 * C2 : 0));
  const hipError_t ae = hipFuncSetAttribute((const void*)blazepair_s2_kernel<W, C2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((blazepair_s2_kernel<W, C2>), dim3(fp_ceil_div(a.nbands, NSUB)), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// Output rows of y2 per band: the largest divisor of OH that is <= 32 and leaves at least 512 workgroups (two per CU), never
// below 4 (a band computes 2 R + 1 rows of y1 for its R rows of y2).
int fp_blazepair_s2_band_rows(const fp_op& op) {
  const int nsub = 4 / (op.W / 32);
  int best = 0;
  for (int r = 4; r <= 32 && 2 * r <= op.OH; r += 4) {    // at least two bands per image
    if (op.OH % r) continue;
    if (best == 0 || (long)op.N * (op.OH / r) / nsub >= 512) best = r;
  }
  return best;
}

// A stride-1 24 -> 24 block and the stride-2 24 -> 24 / 48 block behind it on a row-padded 128- or 64-pixel-wide map
// (include/facepath.h, BLAZEPAIR with stride = 2).
bool fp_blazepair_s2_supported(const fp_op& op) {
  if (op.kind != FP_OP_BLAZEPAIR || !(op.flags & FP_OPF_IN_ROWPAD) || (op.flags & ~(FP_OPF_IN_ROWPAD | FP_OPF_OUT_ROWPAD))) return false;
  if (op.stride != 2 || op.KH != 3 || op.KW != 3 || op.pad_t != 0 || op.pad_l != 0) return false;
  if (op.Cin != 24 || (op.Cout != 24 && op.Cout != 48) || op.in_ld != 24 || op.out_ld != op.Cout || op.out_cmul != 1) return false;
  if (op.H % 2 || op.W % 2 || op.OH != op.H / 2 || op.OW != op.W / 2 || (op.W != 128 && op.W != 64) || op.H < 16) return false;
  if (op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4) return false;
  if (op.w_off % 4 || op.scale_off % 4 || op.slope_off % 4 || op.bias_off % 4) return false;
  if (op.res_mode != FP_RES_POOL2_BEFORE_ACT || op.act != FP_ACT_RELU) return false;
  return fp_blazepair_s2_band_rows(op) > 0;
}

int fp_launch_blazepair_s2(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_blazepair_s2_supported(op)) return FP_ERR_UNSUPPORTED;
  BlazePairS2Args a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.wd = weights + op.w_off;
  a.bd = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.bp = weights + op.bias_off;
  a.H = op.H;
  a.R = fp_blazepair_s2_band_rows(op);
  a.bands = op.OH / a.R;
  a.nbands = op.N * a.bands;
  a.in_rp = (op.W + 1) * 24;
  a.out_rp = (op.OW + ((op.flags & FP_OPF_OUT_ROWPAD) ? 1 : 0)) * op.Cout;
  a.in_ns = op.in_ns;
  a.out_ns = op.out_ns;
  a.bands_div = fp_make_divisor((unsigned)a.bands);      // >= 2 by fp_blazepair_s2_band_rows
  if (op.Cout == 24) return op.W == 128 ? launch_pair_s2<128, 24>(a, s) : launch_pair_s2<64, 24>(a, s);
  return op.W == 128 ? launch_pair_s2<128, 48>(a, s) : launch_pair_s2<64, 48>(a, s);
}
