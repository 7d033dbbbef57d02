// blazepair.hip -- TWO consecutive stride-1 24 -> 24 BlazeBlocks in one kernel, row-padded activations (gfx950).
//
//   y1 = ReLU( pw1( dw1(x)  ) + x  )
//   y2 = ReLU( pw2( dw2(y1) ) + y1 )          (fde/modules/blazeface/blazeface.py:12-47, twice; :122-152 chains 7 of them per map)
//
// One BlazeBlock at a time (blazewp.hip) moves x in and y out per block: the 14 narrow blocks of the back model are 55 %
// of its bytes and run at the rate this part copies memory.  Here y1 never leaves the CU:
//   * a WORKGROUP owns a band of R output rows of one image (128-wide maps: its four waves are the four 32-pixel strips
//     of a row; 64-wide maps: two bands, two strips each) and marches down it one row per step;
//   * step for row y:  block 1 makes y1 row y exactly as blazeblock_wp_kernel does (3-row window of x as a ring in
//     registers, depthwise -> wave-private A tile -> 12 MFMAs -> + shortcut -> ReLU) but writes it into an LDS ring of
//     four y1 rows (row-padded like the tensors in memory: a zero pixel left and right, zero rows above / below the
//     image) -- ONE workgroup barrier -- then block 2 makes output row y - 1 from ring rows y - 2 .. y (its depthwise
//     reads the neighbour strips' columns, which is why the ring is shared), shortcut = ring row y - 1;
//   * the shortcut + bias of block 1 are parked in the ring slot the row will occupy, and the epilogue updates them in
//     place, so there is no shortcut tile; block 2's output tile reuses the A tile: 66 KiB of LDS per workgroup, two
//     workgroups per CU;
//   * a band needs y1 rows y0 - 1 .. y0 + R: R + 2 block-1 rows for R output rows (R = 64: 3 % recomputed).
// Bytes per pair: x once (+ the band halos) and y2 once -- half of what two launches move; the instruction count per
// pixel is the same as two blazeblock_wp launches (fp32 MFMAs and VALU share the SIMD's ALU: tools/lab/coexec_lab.hip).
#include "common.h"

#ifndef FP_PAIR_ABLATE
#define FP_PAIR_ABLATE 0   // lab only (FINDINGS.md finding 31): 1 no per-row barrier, 2 no MFMAs, 4 no depthwise FMAs, 8 no row loads / stores,
                           // 16 block 2 reads one ring row instead of three, 32 no LDS round trip of the output row, 64 depthwise taps not
                           // re-read from LDS -- wrong results, timing only
#endif
#ifndef FP_PAIR_DIRECT_STORE
#define FP_PAIR_DIRECT_STORE 1   // 1: y2 goes from the epilogue registers straight to global memory (32-byte pieces: a lane pair = 8 channels of
                                 // one pixel; L2 merges the pieces of a line) instead of through the A tile: 6 LDS accesses per row less, 258 -> 250 us
#endif

namespace {

struct BlazePairArgs {
  const float* in;    // pixel (0, 0) of image 0, row-padded
  float* out;
  const float* wd;    // [2][9][C]
  const float* bd;    // [2][C]
  const float* wp;    // [2] packed [C/4][32][4]
  const float* bp;    // [2][C]
  int H, R, bands;    // bands per image (H / R)
  int nbands;         // N * bands
  int in_rp, out_rp;  // row pitch, floats
  long in_ns, out_ns;
  fp_divisor bands_div;
};

template <int W>
__global__ __launch_bounds__(256, 2) void blazepair_kernel(BlazePairArgs p) {
  constexpr int C = 24, LDT = C + 4, C4 = C / 4, KG = C / 8, NS = W / 32, NSUB = 4 / NS;
  constexpr int RROW = (W + 2) * C;                        // floats per ring row: pixels -1 .. W
  constexpr int RING = 4 * RROW;
  constexpr int PWF = KG * 2 * 32 * 4;                     // packed 1x1 weights of one block
  static_assert(2 * PWF <= 4 * 32 * LDT, "weight staging fits the A tiles");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                        // [2][10][C] depthwise taps + bias
  float* Bp = Ws + 2 * 10 * C;                             // [2][32]
  float* Rg = Bp + 64;                                     // [NSUB][4][RROW]
  float* Av = Rg + NSUB * RING;                            // 4 wave regions [32][LDT] (first: weight staging)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < 2 * 10 * C / 4; i += 256) {
    const int b = i / (10 * C / 4), k = i - b * (10 * C / 4);
    *(f32x4*)&Ws[i * 4] = (k * 4 < 9 * C) ? *(const f32x4*)(p.wd + b * 9 * C + k * 4) : *(const f32x4*)(p.bd + b * C + (k * 4 - 9 * C));
  }
  if (tid < 64) Bp[tid] = (tid & 31) < C ? p.bp[(tid >> 5) * C + (tid & 31)] : 0.f;
  for (int i = tid; i < 2 * PWF / 4; i += 256) *(f32x4*)&Av[i * 4] = *(const f32x4*)(p.wp + i * 4);
  for (int i = tid; i < NSUB * RING / 4; i += 256) *(f32x4*)&Rg[i * 4] = z;     // pads (and everything else) zero
  __syncthreads();
  f32x4 bf1[KG], bf2[KG];                                  // B fragments of both blocks: k-quad 2*kq + h, column lr
#pragma unroll
  for (int kq = 0; kq < KG; ++kq) {
    bf1[kq] = *(const f32x4*)&Av[((kq * 2 + h) * 32 + lr) * 4];
    bf2[kq] = *(const f32x4*)&Av[PWF + ((kq * 2 + h) * 32 + lr) * 4];
  }
  __syncthreads();                                         // staging area becomes the wave regions

  const int sub = wv / NS, strip = wv - sub * NS;
  float* At = Av + wv * (32 * LDT);                        // A tile [32][LDT]; block 2's output tile [32][C] afterwards
  float* ring = Rg + sub * RING;
  const int x0 = strip * 32;

  // depthwise item of this lane: pixels 4g .. 4g+3 of the strip, channels 4c4 .. 4c4+3 (lanes >= 48 repeat item 0 and
  // write nothing)
  const bool dw_lane = lane < 8 * C4;
  const int la = dw_lane ? lane : 0;
  const int g = la / C4, c4 = la - g * C4;
  const unsigned voff_in = (unsigned)((4 * g * C + 4 * c4) * 4);
  const unsigned voff_out = (unsigned)lane * 16u;
  const float* wl1 = &Ws[4 * c4];
  const float* wl2 = &Ws[10 * C + 4 * c4];
  const f32x4 pbias1 = *(const f32x4*)&Bp[4 * c4];         // block 1's 1x1 bias rides its shortcut
  const int rg_dw = (x0 + 4 * g) * C + 4 * c4;             // ring column x0 + 4g - 1 of this lane's channels
  const int rg_ep = (x0 + 1 + lr) * C + 4 * h;             // epilogue: ring pixel x0 + lr, channels 4h (+ 8j)

  // this wave's band: (image, band) -> first output row y0; the two halves of a 64-wide workgroup take bands 2b, 2b + 1
  const int bi = min((int)blockIdx.x * NSUB + sub, p.nbands - 1);
  const bool live = (int)blockIdx.x * NSUB + sub < p.nbands;
  const unsigned img = __builtin_amdgcn_readfirstlane(fp_fastdiv((unsigned)bi, p.bands_div));
  const int y0 = __builtin_amdgcn_readfirstlane((bi - (int)img * p.bands) * p.R);   // wave-uniform: the row tests below are scalar branches
  const long in_rb = (long)p.in_rp * 4, out_rb = (long)p.out_rp * 4;
  const char* inb = (const char*)p.in + fp_uniform(((long)img * p.in_ns + (long)(x0 - 1) * C) * 4);    // (row 0, column x0 - 1)
  char* outb = (char*)p.out + fp_uniform(((long)img * p.out_ns + (long)x0 * C) * 4);                   // (row 0, column x0)

  // x window: ring of three rows in registers; at step i (row y = y0 - 1 + i) rows y-1, y, y+1 sit in slots i%3, (i+1)%3, (i+2)%3
  f32x4 x[3][6];
  const int nsteps = p.R + 2;
  {
    const int ifirst = y0 == 0 ? 1 : 0;                    // the top band's first step only zero-fills ring row -1
    const int yf = y0 - 1 + ifirst;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const char* rowp = inb + fp_uniform((long)(yf - 1 + ky) * in_rb);
      if (ifirst == 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) x[ky][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
      } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) x[(ky + 1) % 3][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
      }
    }
  }

  for (int ib = 0; ib < nsteps; ib += 3) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int i = ib + r;
      if (i < nsteps) {
        const int s0 = r, s1 = (r + 1) % 3, s2 = (r + 2) % 3;   // static ring slots of rows y-1, y, y+1
        const int y = y0 - 1 + i;
        float* ry = ring + ((y + 1) & 3) * RROW;                // ring row of y1 row y
        if ((unsigned)y < (unsigned)p.H) {
          // ---- block 1: depthwise -> A tile; shortcut (+ 1x1 bias) -> the ring slot y1 row y will occupy ----
          {
            const f32x4 dbias = *(const f32x4*)(wl1 + 9 * C);
            f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
            for (int ky = 0; ky < ((FP_PAIR_ABLATE & 4) ? 0 : 3); ++ky) {
              const int sl = ky == 0 ? s0 : ky == 1 ? s1 : s2;
              const f32x4 w0 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl1 + (ky * 3 + 0) * C);
              const f32x4 w1 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl1 + (ky * 3 + 1) * C);
              const f32x4 w2 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl1 + (ky * 3 + 2) * C);
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
          if (!(FP_PAIR_ABLATE & 8) && y + 1 < p.H && i + 1 < nsteps) {
            const char* rowp = inb + fp_uniform((long)(y + 2) * in_rb);
#pragma unroll
            for (int j = 0; j < 6; ++j) x[s0][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
          }
          f32x16 m0, m1;
#pragma unroll
          for (int k = 0; k < 16; ++k) m0[k] = 0.f, m1[k] = 0.f;
          const float* arow = &At[lr * LDT + 4 * h];
          // D^T = W^T x A^T (the two operands swapped): lane (lr, h) ends up with PIXEL lr and channels
          // (k & 3) + 8*(k >> 2) + 4h -- four consecutive channels per register quad, i.e. 16-byte pieces of a row-major
          // pixel, so the epilogue is 3 x (ds_read_b128, packed adds, ds_write_b128) instead of 16 + 16 scalar LDS accesses
#pragma unroll
          for (int kq = 0; kq < ((FP_PAIR_ABLATE & 2) ? 0 : KG); ++kq) {
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
          // y1 rows -1 and H are block 2's zero padding
#pragma unroll
          for (int j = 0; j < 3; ++j) *(f32x4*)&ry[(x0 + 1) * C + (lane + 64 * j) * 4] = z;
        }
#if !(FP_PAIR_ABLATE & 1)
        __syncthreads();
#endif
        if (i >= 2) {
          // ---- block 2: output row yo = y - 1 from ring rows yo-1, yo, yo+1 ----
          const int yo = y - 1;
          {
            const f32x4 dbias = *(const f32x4*)(wl2 + 9 * C);
            f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
            for (int ky = 0; ky < ((FP_PAIR_ABLATE & 4) ? 0 : 3); ++ky) {
              const float* rr = ring + ((yo + ((FP_PAIR_ABLATE & 16) ? 1 : ky)) & 3) * RROW + rg_dw;      // ring row of y1 row yo - 1 + ky
              f32x4 xv[6];
#pragma unroll
              for (int j = 0; j < 6; ++j) xv[j] = *(const f32x4*)(rr + j * C);
              const f32x4 w0 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl2 + (ky * 3 + 0) * C);
              const f32x4 w1 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl2 + (ky * 3 + 1) * C);
              const f32x4 w2 = (FP_PAIR_ABLATE & 64) ? dbias : *(const f32x4*)(wl2 + (ky * 3 + 2) * C);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                acc[q] += xv[q] * w0;
                acc[q] += xv[q + 1] * w1;
                acc[q] += xv[q + 2] * w2;
              }
            }
            if (dw_lane) {
#pragma unroll
              for (int q = 0; q < 4; ++q) *(f32x4*)&At[(4 * g + q) * LDT + 4 * c4] = acc[q];
            }
          }
          f32x16 m0, m1;
#pragma unroll
          for (int k = 0; k < 16; ++k) m0[k] = 0.f, m1[k] = 0.f;
          const float* arow = &At[lr * LDT + 4 * h];
#pragma unroll
          for (int kq = 0; kq < ((FP_PAIR_ABLATE & 2) ? 0 : KG); ++kq) {
            const f32x4 a = *(const f32x4*)(arow + kq * 8);
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[kq][0], a[0], m0, 0, 0, 0);
            FP_MFMA_ORDER();
            m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[kq][1], a[1], m1, 0, 0, 0);
            FP_MFMA_ORDER();
            m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[kq][2], a[2], m0, 0, 0, 0);
            FP_MFMA_ORDER();
            m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bf2[kq][3], a[3], m1, 0, 0, 0);
            FP_MFMA_ORDER();
          }
          // y2 = ReLU(1x1 + bias + y1 row yo) -> output tile [32][C] over the A tile (this wave's MFMAs have consumed it)
          {
            const float* spx = ring + ((yo + 1) & 3) * RROW + rg_ep;
            float* opx = &At[lr * C + 4 * h];
            char* orow_g = outb + fp_uniform((long)yo * out_rb);
#pragma unroll
            for (int j = 0; j < C / 8; ++j) {
              const f32x4 sv = *(const f32x4*)(spx + 8 * j) + *(const f32x4*)&Bp[32 + 8 * j + 4 * h];
              f32x4 v;
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (m0[4 * j + e] + m1[4 * j + e]) + sv[e];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
              if (FP_PAIR_DIRECT_STORE | (FP_PAIR_ABLATE & 32)) {
                // lane (lr, h): pixel x0 + lr, channels 8j + 4h .. + 3 -- with its partner lane (h ^ 1) a 32-byte piece
                if (live) *(f32x4*)(orow_g + (unsigned)((lr * C + 4 * h + 8 * j) * 4)) = v;
              } else {
                *(f32x4*)(opx + 8 * j) = v;
              }
            }
          }
          if (!FP_PAIR_DIRECT_STORE && !(FP_PAIR_ABLATE & 32) && live && !((FP_PAIR_ABLATE & 8) && yo > y0)) {
            char* orow_g = outb + fp_uniform((long)yo * out_rb);
#pragma unroll
            for (int j = 0; j < 3; ++j) *(f32x4*)(orow_g + voff_out + j * 1024) = *(const f32x4*)&At[(lane + 64 * j) * 4];
          }
        }
      }
    }
  }
}

template <int W>
int launch_pair(const BlazePairArgs& a, hipStream_t s) {
  constexpr int C = 24, NSUB = 4 / (W / 32);
  size_t lds = 4 * ((size_t)2 * 10 * C + 64 + (size_t)NSUB * 4 * (W + 2) * C + 4 * (size_t)32 * (C + 4));
  if ((size_t)fp_get_knobs().pair_lds_min > lds) lds = (size_t)fp_get_knobs().pair_lds_min;      // lab knob, 0 in the product
  const hipError_t ae = hipFuncSetAttribute((const void*)blazepair_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((blazepair_kernel<W>), dim3(fp_ceil_div(a.nbands, NSUB)), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// Rows per band: the largest divisor of H that is <= 64 and leaves at least 512 workgroups (two per CU), never below 8
// (batch 256 at 128 x 128: 64-row bands = 512 workgroups, 3 % of block 1 recomputed: 255 us against 269 with 32-row bands).
#ifndef FP_PAIR_MAX_ROWS
#define FP_PAIR_MAX_ROWS 64
#endif
int fp_blazepair_band_rows(const fp_op& op) {
  const int nsub = 4 / (op.W / 32);
  int best = 0;
  for (int r = 8; r <= FP_PAIR_MAX_ROWS && 2 * r <= op.H; r += 4) {      // at least two bands per image (the kernel's band index math)
    if (op.H % r) continue;
    if (best == 0 || (long)op.N * (op.H / r) / nsub >= 512) best = r;
  }
  return best;
}

// Two stride-1 24 -> 24 blocks on a row-padded 128- or 64-pixel-wide map (include/facepath.h, BLAZEPAIR).
bool fp_blazepair_supported(const fp_op& op) {
  if (op.kind != FP_OP_BLAZEPAIR || !(op.flags & FP_OPF_IN_ROWPAD)) return false;
  if (op.stride != 1 || op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.Cin != 24 || op.Cout != 24 || op.in_ld != 24 || op.out_ld != 24 || op.out_cmul != 1) return false;
  if (op.OH != op.H || op.OW != op.W || (op.W != 128 && op.W != 64) || op.H % 8 || op.H < 8) return false;
  if (op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4) return false;
  if (op.w_off % 4 || op.scale_off % 4 || op.slope_off % 4 || op.bias_off % 4) return false;
  if (op.res_mode != FP_RES_ADD_BEFORE_ACT || op.act != FP_ACT_RELU) return false;
  const int r = fp_blazepair_band_rows(op);
  return r > 0 && op.H / r >= 2;
}

int fp_launch_blazepair(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_blazepair_supported(op)) return FP_ERR_UNSUPPORTED;
  constexpr int C = 24;
  BlazePairArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.wd = weights + op.w_off;
  a.bd = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.bp = weights + op.bias_off;
  a.H = op.H;
  a.R = fp_blazepair_band_rows(op);
  a.bands = op.H / a.R;
  a.nbands = op.N * a.bands;
  a.in_rp = (op.W + 1) * C;
  a.out_rp = (op.OW + ((op.flags & FP_OPF_OUT_ROWPAD) ? 1 : 0)) * C;
  a.in_ns = op.in_ns;
  a.out_ns = op.out_ns;
  a.bands_div = fp_make_divisor((unsigned)a.bands);
  return op.W == 128 ? launch_pair<128>(a, s) : launch_pair<64>(a, s);
}
