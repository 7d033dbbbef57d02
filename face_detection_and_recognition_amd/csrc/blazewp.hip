// blazewp.hip -- wave-private fused BlazeBlock kernels on ROW-PADDED activations (include/facepath.h FP_OPF_*) for gfx950.
//
//   y = ReLU( pw1x1( dw3x3(x) ) + x )      stride 1, C -> C        (fde/modules/blazeface/blazeface.py:12-47)
//
// blaze.hip keeps the workgroup kernels (dense input: stride 2, other widths, small batches).  The two kernels here are
// what the BlazeFace planner uses wherever it can keep a tensor in the row-padded layout: blazeblock_wp_kernel for the
// 24 -> 24 blocks on 128 x 128 / 64 x 64 maps (55 % of the back model's bytes) and blazeblock_wps_kernel for the
// 48 -> 48 / 96 -> 96 blocks on 32 x 32 / 16 x 16 maps.  Both give a WAVE a tile of 32 pixels end to end (window ->
// depthwise -> LDS -> MFMA 1x1 -> shortcut -> ReLU -> store) with no workgroup barrier in the loop; FINDINGS.md findings
// 14-15 record what that bought and why.
#include "common.h"

namespace {
// ---------------------------------------------------------------------------------------------------------------
// Wave-private BlazeBlock, stride 1, C -> C channels, on a ROW-PADDED input (facepath.h FP_OPF_IN_ROWPAD: every image
// row is followed by one zero pixel, the image by a zero row above and below; (y, -1) is the pad pixel of row y - 1).
//
// The persistent workgroup kernel (blaze.hip) synchronises four waves twice per tile and spends ~390 VALU instructions per wave and
// tile, most of them on what the padded layout makes unnecessary: clamps, validity masks and selects for the zero
// padding, per-lane index decode.  With no memory traffic at all its skeleton still takes 140 of its 207 us on the
// 128 x 128 blocks (lab toggles, FINDINGS.md finding 14).  Here
//   * a WAVE owns a tile of 32 consecutive pixels of one image row and runs the whole chain on it with its own LDS
//     region: window -> depthwise (48 of 64 lanes = 8 pixel groups x 6 channel groups) -> A tile -> 12 MFMAs against
//     the 1x1 weights held in registers -> shortcut + ReLU -> output tile -> 3 x 16-byte stores per lane.  No
//     __syncthreads() in the loop: LDS executes one wave's instructions in order, and the 12 waves of a CU interleave
//     freely instead of meeting at barriers;
//   * the tile index is wave-uniform, so (image, row, column) and the row base pointers are SALU work; a load or store
//     is `global_* v, v_lane_offset, s[base] offset:imm` with a lane offset computed once per kernel.  The zero
//     padding is in memory: no clamps, no masks, no selects;
//   * the next tile's window is requested right after the depthwise phase has consumed the current one; the previous
//     tile's output leaves LDS just before that request (older than the prefetch in the in-order vmcnt queue, so the
//     wait for the window does not drain stores that were only just issued).
// The output is written row-padded or dense (out_rp / out_ns), so a chain of blocks keeps the layout and the last one
// hands a dense tensor to the next kernel.
struct BlazeWpArgs {
  const float* in;    // pixel (0, 0) of image 0
  float* out;
  const float* wd;    // [9][C]
  const float* bd;    // [C]
  const float* wp;    // packed [C/4][32][4]
  const float* bp;    // [C]
  int OH, OW, strips, bands, ntiles;   // tile = (image, band of R rows, strip of 32 columns)
  int in_rp, out_rp;  // row pitch, floats
  long in_ns, out_ns;
  fp_divisor strips_div, bands_div;
};

// R = rows a wave marches down per tile.  The window is a ring of three input rows in registers: a new output row costs
// ONE new row of 6 loads instead of 18 (R = 4: 9 loads per output row on average).  The window loads, not HBM, were
// the slow part of the first form of this kernel: with the compute stripped it took 150 us to read a 403 MB tensor that
// a linear read gets through in 107 us -- 4.5 16-byte requests per output (pixel, channel-quad) in 96-byte pieces keep
// the CU's address/L1 path busy, and 12 waves x 10 KB of window do not fit the 32 KB L1 (FINDINGS.md finding 14).
template <int C, int R>
__global__ __launch_bounds__(256, 3) void blazeblock_wp_kernel(BlazeWpArgs p) {
  static_assert(C % 8 == 0 && C <= 32, "one 32-column n tile, K a multiple of 8");
  constexpr int LDT = C + 4, C4 = C / 4, NIT = 8 * C4, KG = C / 8;
  constexpr int WAVE_FLOATS = 2 * 32 * LDT + 32 * C;      // A tile, shortcut tile, output tile
  constexpr int NST = (32 * C / 4) / 64;                   // 16-byte stores per lane and row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                        // [10][C] depthwise taps + bias
  float* Bp = Ws + 10 * C;                                 // [32]
  float* Wv = Bp + 32;                                     // 4 wave regions (first used to stage the 1x1 weights)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;

  for (int i = tid; i < (10 * C) / 4; i += 256)
    *(f32x4*)&Ws[i * 4] = (i * 4 < 9 * C) ? *(const f32x4*)(p.wd + i * 4) : *(const f32x4*)(p.bd + (i * 4 - 9 * C));
  if (tid < 32) Bp[tid] = tid < C ? p.bp[tid] : 0.f;
  for (int i = tid; i < KG * 2 * 32; i += 256) *(f32x4*)&Wv[i * 4] = *(const f32x4*)(p.wp + i * 4);
  __syncthreads();
  f32x4 bfrag[KG];                                         // this lane's B fragments: k-quad 2*kq + h, column lr
#pragma unroll
  for (int kq = 0; kq < KG; ++kq) bfrag[kq] = *(const f32x4*)&Wv[((kq * 2 + h) * 32 + lr) * 4];
  __syncthreads();                                         // staging area becomes the wave regions

  float* At = Wv + wv * WAVE_FLOATS;                       // [32][LDT]
  float* St = At + 32 * LDT;                               // [32][LDT]
  float* Ot = St + 32 * LDT;                               // [32][C]

  // depthwise item of this lane: pixels 4g .. 4g+3 of the strip, channels 4c4 .. 4c4+3 (lanes >= NIT repeat item 0
  // and write nothing)
  const bool dw_lane = lane < NIT;
  const int la = dw_lane ? lane : 0;
  const int g = la / C4, c4 = la - g * C4;
  const unsigned voff_in = (unsigned)((4 * g * C + 4 * c4) * 4);
  const unsigned voff_out = (unsigned)lane * 16u;
  const float* wl = &Ws[4 * c4];                           // this lane's taps [k][4] at wl + k*C, bias at k = 9
  const f32x4 pbias = *(const f32x4*)&Bp[4 * c4];          // pointwise bias of this lane's channels: rides the shortcut

  // XCD-aware order: block b runs on XCD b % 8; each XCD gets a contiguous range of every round's tiles, so the bands
  // above and below a tile (its halo rows) are fetched into the same L2
  const int G = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }

  // byte offsets from p.in / p.out, all scalar: window origin (y0 - 1, x0 - 1) and first output pixel of a tile
  const char* inb = (const char*)p.in;
  char* outb = (char*)p.out;
  // tile index = (image * strips + strip) * bands + band: a wave's consecutive tiles walk DOWN a strip, so the two halo
  // rows a band shares with the one above were read by this very wave a few microseconds ago (L1 / L2 hits instead of
  // the 34 % extra HBM fetch that interleaved tiles cost: profiles/r02_bench_pmc_traffic.json history)
  auto locate = [&](int t, long& ip, long& op) {
    const unsigned q = fp_fastdiv((unsigned)t, p.bands_div), y0 = ((unsigned)t - q * (unsigned)p.bands) * R;
    const unsigned img = fp_fastdiv(q, p.strips_div), sx = q - img * (unsigned)p.strips;
    ip = fp_uniform(((long)img * p.in_ns + ((long)y0 - 1) * p.in_rp + ((long)sx * 32 - 1) * C) * 4);
    op = fp_uniform(((long)img * p.out_ns + (long)y0 * p.out_rp + (long)sx * 32 * C) * 4);
  };
  f32x4 x[3][6];                                           // ring of three input rows
  const long in_rb = (long)p.in_rp * 4, out_rb = (long)p.out_rp * 4;

  // this wave's contiguous run [t, t_end) of tiles: equal shares, the first (ntiles mod waves) waves take one more
  const int wi = pos * 4 + wv, nw = G * 4;
  const int share = p.ntiles / nw, extra = p.ntiles - share * nw;
  int t = wi * share + min(wi, extra);
  const int t_end = t + share + (wi < extra ? 1 : 0);
  long in_px = 0;               // byte offset of the current tile's window origin
  long out_px = 0;              // ... of its first output pixel
  long st_px = 0;               // ... of the row waiting in Ot
  if (t < t_end) {
    locate(t, in_px, out_px);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const char* rowp = inb + fp_uniform(in_px + ky * in_rb);
#pragma unroll
      for (int j = 0; j < 6; ++j) x[ky][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
    }
  }
  bool have_prev = false;
  while (t < t_end) {
    const int tn = t + 1;
    long in_nx = 0, out_nx = 0;
    if (tn < t_end) locate(tn, in_nx, out_nx);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int s0 = r % 3, s1 = (r + 1) % 3, s2 = (r + 2) % 3;   // ring slots of rows y - 1, y, y + 1 (static: unrolled)
      // ---- depthwise + shortcut -> A, S (this wave's region) ----
      {
        const f32x4 dbias = *(const f32x4*)(wl + 9 * C);
        f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int sl = ky == 0 ? s0 : ky == 1 ? s1 : s2;
          const f32x4 w0 = *(const f32x4*)(wl + (ky * 3 + 0) * C);
          const f32x4 w1 = *(const f32x4*)(wl + (ky * 3 + 1) * C);
          const f32x4 w2 = *(const f32x4*)(wl + (ky * 3 + 2) * C);
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // three statements: each contracts to one packed FMA on the accumulator
            acc[q] += x[sl][q] * w0;
            acc[q] += x[sl][q + 1] * w1;
            acc[q] += x[sl][q + 2] * w2;
          }
        }
        if (dw_lane) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            *(f32x4*)&At[(4 * g + q) * LDT + 4 * c4] = acc[q];
            *(f32x4*)&St[(4 * g + q) * LDT + 4 * c4] = x[s1][q + 1] + pbias;   // shortcut = the centre tap (+ 1x1 bias)
          }
        }
      }
      // ---- previous row's output: LDS -> HBM, then the next window row(s) ----
      if (have_prev) {
#pragma unroll
        for (int j = 0; j < NST; ++j)
          *(f32x4*)(outb + st_px + voff_out + j * 1024) = *(const f32x4*)&Ot[(lane + 64 * j) * 4];
      }
      st_px = fp_uniform(out_px + r * out_rb);
      if (r + 1 < R) {                  // row y + 2 replaces row y - 1 in the ring
        const char* rowp = inb + fp_uniform(in_px + (r + 3) * in_rb);
#pragma unroll
        for (int j = 0; j < 6; ++j) x[s0][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
      } else if (tn < t_end) {          // last row of the band: the next tile's first three rows
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const char* rowp = inb + fp_uniform(in_nx + ky * in_rb);
#pragma unroll
          for (int j = 0; j < 6; ++j) x[ky][j] = *(const f32x4*)(rowp + voff_in + j * C * 4);
        }
      }
      // ---- 1x1 on the MFMA pipe ----
      f32x16 m0, m1;
#pragma unroll
      for (int i = 0; i < 16; ++i) m0[i] = 0.f, m1[i] = 0.f;
      const float* arow = &At[lr * LDT + 4 * h];
#pragma unroll
      for (int kq = 0; kq < KG; ++kq) {
        const f32x4 a = *(const f32x4*)(arow + kq * 8);
        // operands swapped: D^T = W^T x A^T, so lane (lr, h) holds PIXEL lr and channels (k & 3) + 8*(k >> 2) + 4h --
        // 16-byte pieces of a row-major pixel: the epilogue is 3 x (b128 read, adds, b128 write) instead of 16 + 16
        // scalar LDS accesses (csrc/blazepair.hip has the same form)
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bfrag[kq][0], a[0], m0, 0, 0, 0);
        FP_MFMA_ORDER();
        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bfrag[kq][1], a[1], m1, 0, 0, 0);
        FP_MFMA_ORDER();
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(bfrag[kq][2], a[2], m0, 0, 0, 0);
        FP_MFMA_ORDER();
        m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(bfrag[kq][3], a[3], m1, 0, 0, 0);
        FP_MFMA_ORDER();
      }
      // ---- shortcut + ReLU -> output tile: pixel lr, channels 8j + 4h .. + 3 ----
      {
        const float* spx = &St[lr * LDT + 4 * h];
        float* opx = &Ot[lr * C + 4 * h];
#pragma unroll
        for (int j = 0; j < C / 8; ++j) {
          const f32x4 sv = *(const f32x4*)(spx + 8 * j);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (m0[4 * j + e] + m1[4 * j + e]) + sv[e];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          *(f32x4*)(opx + 8 * j) = v;
        }
      }
      have_prev = true;
    }
    t = tn;
    in_px = in_nx;
    out_px = out_nx;
  }
  if (have_prev) {
#pragma unroll
    for (int j = 0; j < NST; ++j) *(f32x4*)(outb + st_px + voff_out + j * 1024) = *(const f32x4*)&Ot[(lane + 64 * j) * 4];
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Wave-private BlazeBlock for the WIDE blocks on SMALL maps (48 -> 48 on 32 x 32, 96 -> 96 on 16 x 16: the second half
// of BlazeFace's backbone, blazeface.py:148-160), row-padded input.  These launches move 50-100 MB at batch 256 -- the
// tensors live in L2 / MALL -- and were latency bound: the per-tile kernel stages 12 KB of weights per 128-pixel tile
// and starts every tile cold (49 us per 48-channel block), the 96-channel blocks ran as two kernels (depthwise, then
// the 1x1 GEMM: 16 + 28 us) because the fused tile does not fit LDS.
// Same scheme as blazeblock_wp_kernel with the channels in PASSES of 24 (48 lanes = 8 pixel groups x 6 channel groups
// per pass): a wave owns a tile of 32 pixels (one row of a 32-wide map, two rows of a 16-wide one) end to end, no
// workgroup barrier in the loop; the 1x1 weights of the whole block stay in LDS (shared by the four waves); the output
// tile overwrites the A tile in LDS once the MFMAs have consumed it; the shortcut is re-read from the input (an L2 hit,
// requested before the MFMAs) when the tile is stored, so no shortcut tile is kept.
// Measured (batch 256): 48 -> 48 blocks 52 -> 41 us, 96 -> 96 blocks 44 (two kernels) -> 33 us.  A tile costs a wave
// ~5 / ~8.5 us (C = 48 / 96) of mostly serial latency -- window round trip, depthwise, an MFMA phase with one or two
// waves per SIMD -- so the grid gives every wave as few tiles as the resident workgroup count allows.
struct BlazeWpsArgs {
  const float* in;    // pixel (0, 0) of image 0, row-padded
  float* out;
  const float* wd;    // [9][C]
  const float* bd;    // [C]
  const float* wp;    // packed [C/4][Npad][4]
  const float* bp;    // [C]
  int W, wshift;      // map width 16 or 32 (1 << wshift)
  int tiles_per_img, ntiles, per_wave;
  int in_rp, out_rp;  // row pitch, floats
  long in_ns, out_ns;
  fp_divisor tpi_div;
};

template <int C>
__global__ __launch_bounds__(256, C <= 48 ? 2 : 1) void blazeblock_wps_kernel(BlazeWpsArgs p) {
  static_assert(C % 24 == 0 && C % 8 == 0, "channel passes of 24, K a multiple of 8");
  constexpr int NB = (C + 31) / 32, NPAD = NB * 32, KG = C / 8, LDT = C + 4, PASSES = C / 24, C4T = C / 4;
  constexpr int NST = (32 * C4T) / 64;                      // 16-byte stores per lane and tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;                                         // [10][C] depthwise taps + bias
  float* Bp = Ws + 10 * C;                                  // [NPAD] pointwise bias
  float* Bs = Bp + NPAD;                                    // [C/4][NPAD][4] pointwise weights
  float* Wv = Bs + C * NPAD;                                // 4 wave regions [32][LDT]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;

  for (int i = tid; i < (10 * C) / 4; i += 256)
    *(f32x4*)&Ws[i * 4] = (i * 4 < 9 * C) ? *(const f32x4*)(p.wd + i * 4) : *(const f32x4*)(p.bd + (i * 4 - 9 * C));
  if (tid < NPAD) Bp[tid] = tid < C ? p.bp[tid] : 0.f;
  for (int i = tid; i < C4T * NPAD; i += 256) *(f32x4*)&Bs[i * 4] = *(const f32x4*)(p.wp + (long)i * 4);
  __syncthreads();                                          // the only workgroup barrier

  float* At = Wv + wv * (32 * LDT);                         // A tile [32][LDT]; later the output tile [32][C]
  float bias_n[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) bias_n[nb] = Bp[nb * 32 + lr];

  // depthwise item of this lane in every pass: pixels 4g .. 4g+3 of the tile (one image row: W % 4 == 0), channels
  // 24*pass + 4c4 .. +3; lanes >= 48 repeat item 0 and write nothing
  const bool dw_lane = lane < 48;
  const int la = dw_lane ? lane : 0;
  const int g = la / 6, c4 = la - g * 6;
  const int gr = (4 * g) >> p.wshift, gx = (4 * g) & (p.W - 1);
  const unsigned voff_dw = (unsigned)(((gr * (p.W + 1) + gx) * C + 4 * c4) * 4);
  const char* inb = (const char*)p.in;
  char* outb = (char*)p.out;
  const long in_rb = (long)p.in_rp * 4;

  // XCD-aware order of the waves' runs (neighbouring tiles share halo rows: same L2)
  const int G = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  // byte offsets (scalar) of pixel (y0, 0) of a tile in the input and the output
  auto locate = [&](int t, long& ic, long& oo) {
    const unsigned img = fp_fastdiv((unsigned)t, p.tpi_div), k = (unsigned)t - img * (unsigned)p.tiles_per_img;
    const unsigned y0 = (k * 32u) >> p.wshift;
    ic = fp_uniform(((long)img * p.in_ns + (long)y0 * p.in_rp) * 4);
    oo = fp_uniform(((long)img * p.out_ns + (long)y0 * p.out_rp) * 4);
  };
  // The windows of ALL channel passes of a tile are requested in one batch (one memory round trip per tile instead of
  // one per pass: these launches are latency bound), the next tile's right after this tile's depthwise phase, so they
  // arrive under its MFMAs.  The register budget (72 VGPRs per pass) is why only 2 (C = 48) / 1 (C = 96) workgroups
  // share a CU.
  f32x4 x[PASSES][3][6];
  auto issue_windows = [&](long ic) {
    const long win = ic - in_rb - C * 4;                    // pixel (y0 - 1, -1)
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const char* rowp = inb + fp_uniform(win + ky * in_rb + ps * 96);
#pragma unroll
        for (int j = 0; j < 6; ++j) x[ps][ky][j] = *(const f32x4*)(rowp + voff_dw + j * C * 4);
      }
  };

  int t = (pos * 4 + wv) * p.per_wave;
  const int t_end = min(t + p.per_wave, p.ntiles);
  long in_ctr = 0, out_o = 0;
  if (t < t_end) {
    locate(t, in_ctr, out_o);
    issue_windows(in_ctr);
  }
  for (; t < t_end; ++t) {
    // ---- depthwise, 24 channels per pass -> A tile ----
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const float* wl = &Ws[ps * 24 + 4 * c4];
      const f32x4 dbias = *(const f32x4*)(wl + 9 * C);
      f32x4 acc[4] = {dbias, dbias, dbias, dbias};
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)(wl + (ky * 3 + 0) * C);
        const f32x4 w1 = *(const f32x4*)(wl + (ky * 3 + 1) * C);
        const f32x4 w2 = *(const f32x4*)(wl + (ky * 3 + 2) * C);
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // three statements: each contracts to one packed FMA on the accumulator
          acc[q] += x[ps][ky][q] * w0;
          acc[q] += x[ps][ky][q + 1] * w1;
          acc[q] += x[ps][ky][q + 2] * w2;
        }
      }
      if (dw_lane) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4*)&At[(4 * g + q) * LDT + ps * 24 + 4 * c4] = acc[q];
      }
    }
    // ---- shortcut values of the float4s this lane will store, then the next tile's windows: both in flight during
    // the MFMAs ----
    f32x4 xc[NST];
    unsigned ooff[NST];
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const unsigned i = (unsigned)lane + 64u * j;
      const unsigned px = i / C4T, cq = i - px * C4T;
      const unsigned r = px >> p.wshift, xx = px & (unsigned)(p.W - 1);
      xc[j] = *(const f32x4*)(inb + in_ctr + ((r * (unsigned)p.in_rp + xx * C + cq * 4) * 4u));
      ooff[j] = (r * (unsigned)p.out_rp + xx * C + cq * 4) * 4u;
    }
    const long out_cur = out_o;
    if (t + 1 < t_end) {
      locate(t + 1, in_ctr, out_o);
      issue_windows(in_ctr);
    }
    // ---- 1x1 on the MFMA pipe ----
    f32x16 macc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) macc[nb][r] = 0.f;
    const float* arow = &At[lr * LDT + 4 * h];
#pragma unroll
    for (int kq = 0; kq < KG; ++kq) {
      const f32x4 a = *(const f32x4*)(arow + kq * 8);
      f32x4 b[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)&Bs[((kq * 2 + h) * NPAD + nb * 32 + lr) * 4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          macc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[nb][e], macc[nb], 0, 0, 0);
          FP_MFMA_ORDER();
        }
    }
    // ---- + bias -> output tile [32][C] over the A tile (this wave's MFMAs have consumed it) ----
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 32 + lr;
      if (n < C) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          At[((reg & 3) + 8 * (reg >> 2) + 4 * h) * C + n] = macc[nb][reg] + bias_n[nb];
      }
    }
    // ---- + shortcut, ReLU, 16-byte stores ----
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const f32x4 v = *(const f32x4*)&At[(lane + 64 * j) * 4] + xc[j];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : 0.f;
      *(f32x4*)(outb + out_cur + ooff[j]) = o;
    }
  }
}

}  // namespace

// Wave-private kernel: stride 1, 24 -> 24 with the full shortcut, rows that split into 32-pixel tiles, row-padded input.
bool fp_blazeblock_wp_eligible(const fp_op& op) {
  return op.kind == FP_OP_BLAZEBLOCK && (op.flags & FP_OPF_IN_ROWPAD) && op.stride == 1 && op.KH == 3 && op.KW == 3 &&
         op.Cin == 24 && op.Cout == 24 && op.res_C == 24 && op.in_ld == 24 && op.out_ld == 24 && op.out_cmul == 1 &&
         op.OH == op.H && op.OW == op.W && op.OW % 32 == 0 && op.OW >= 64 && op.OH % 4 == 0 && op.OH >= 8 &&
         op.in_off % 4 == 0 && op.out_off % 4 == 0 &&
         op.in_ns % 4 == 0 && op.out_ns % 4 == 0 && (long)op.N * (op.OH / 4) * (op.OW / 32) < (1L << 31);
}

// Wide blocks on small maps: stride 1, C -> C with C = 48 or 96, 16- or 32-pixel-wide maps, row-padded input.
bool fp_blazeblock_wps_eligible(const fp_op& op) {
  return op.kind == FP_OP_BLAZEBLOCK && (op.flags & FP_OPF_IN_ROWPAD) && op.stride == 1 && op.KH == 3 && op.KW == 3 &&
         (op.Cin == 48 || op.Cin == 96) && op.Cout == op.Cin && op.res_C == op.Cin && op.in_ld == op.Cin &&
         op.out_ld == op.Cin && op.out_cmul == 1 && op.OH == op.H && op.OW == op.W && (op.W == 16 || op.W == 32) &&
         ((long)op.H * op.W) % 32 == 0 && (long)op.H * op.W >= 64 && op.in_off % 4 == 0 && op.out_off % 4 == 0 &&
         op.in_ns % 4 == 0 && op.out_ns % 4 == 0 && (long)op.N * op.H * op.W / 32 < (1L << 31);
}

template <int C>
static int launch_blazeblock_wps(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  constexpr int NPAD = (C + 31) / 32 * 32;
  BlazeWpsArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.wd = weights + op.w_off;
  a.bd = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.bp = weights + op.bias_off;
  a.W = op.W;
  a.wshift = op.W == 32 ? 5 : 4;
  a.tiles_per_img = op.H * op.W / 32;       // >= 2
  a.ntiles = op.N * a.tiles_per_img;
  a.in_rp = (op.W + 1) * C;
  a.out_rp = (op.OW + ((op.flags & FP_OPF_OUT_ROWPAD) ? 1 : 0)) * C;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.tpi_div = fp_make_divisor((unsigned)a.tiles_per_img);
  const size_t lds = 4 * ((size_t)10 * C + NPAD + (size_t)C * NPAD + 4 * (size_t)32 * (C + 4));
  // persistent grid: 2 (C = 48) / 1 (C = 96: 92 KB of LDS) workgroups per CU, every wave a contiguous run of tiles
  const int max_wg = 256 * (C <= 48 ? 2 : 1);
  a.per_wave = fp_ceil_div(a.ntiles, max_wg * 4);
  const int G = fp_ceil_div(a.ntiles, 4 * a.per_wave);
  hipError_t ae = hipFuncSetAttribute((const void*)blazeblock_wps_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((blazeblock_wps_kernel<C>), dim3(G), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

static int launch_blazeblock_wp(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  // R = 8 (7.5 instead of 9 window loads per row) measured the same within noise on 128 x 128 maps and 6 % faster on
  // 64 x 64 ones, against twice the code: R = 4 everywhere
  constexpr int C = 24, R = 4;
  BlazeWpArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.wd = weights + op.w_off;
  a.bd = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.bp = weights + op.bias_off;
  a.OH = op.OH; a.OW = op.OW;
  a.strips = op.OW / 32;            // >= 2 (eligibility: OW >= 64)
  a.bands = op.OH / R;              // >= 2 (eligibility: OH % 4 == 0, OH >= 8)
  a.ntiles = op.N * a.bands * a.strips;
  a.in_rp = (op.W + 1) * C;
  a.out_rp = (op.OW + ((op.flags & FP_OPF_OUT_ROWPAD) ? 1 : 0)) * C;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.strips_div = fp_make_divisor((unsigned)a.strips);
  a.bands_div = fp_make_divisor((unsigned)a.bands);
  const size_t lds = 4 * ((size_t)10 * C + 32 + 4 * (size_t)(2 * 32 * (C + 4) + 32 * C));
  // persistent grid: 3 workgroups per CU; every wave owns a contiguous run of tiles (+-1), at least two when there are few
  int G = fp_ceil_div(a.ntiles, 8);
  if (G > 256 * 3) G = 256 * 3;
  hipLaunchKernelGGL((blazeblock_wp_kernel<C, R>), dim3(G), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// Row-padded input: dispatch to the kernel that takes the shape (FP_ERR_UNSUPPORTED otherwise; fp_plan_validate checks
// the same predicates on the host).
int fp_launch_blazeblock_rowpad(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (fp_blazeblock_wp_eligible(op)) return launch_blazeblock_wp(op, weights, arena, s);
  if (fp_blazeblock_wps_eligible(op))
    return op.Cin == 48 ? launch_blazeblock_wps<48>(op, weights, arena, s) : launch_blazeblock_wps<96>(op, weights, arena, s);
  return FP_ERR_UNSUPPORTED;
}
