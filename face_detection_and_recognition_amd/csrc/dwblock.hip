// dwblock.hip — a WHOLE Mobile-FaceNet Depth_Wise block in one kernel (gfx950).
//
// Depth_Wise.forward (fde/modules/mobile_facenet/mobile_facenet.py:77-88) is
//     conv (1x1 expand C -> G, BN, PReLU) -> conv_dw (3x3 depthwise, BN, PReLU) -> project (1x1 G -> C, BN) [+ x].
// Round 2 ran it as two launches (pws.hip: expand, dwpw.hip: depthwise + project) with the G-channel tensor written
// to HBM by the first and read back by the second.  Here it never leaves the CU.
//
//   tile  = one 512-thread workgroup = 196 (147) output pixels: a whole 14x14 image, 7 rows of a 28x28 image (+ one
//           halo row on either side whose expand values are recomputed), or three 7x7 images.
//   x     = the tile's input pixels live in REGISTERS for the whole tile, as the MFMA A fragments of the expand GEMM:
//           wave w owns the 16-pixel row tiles w and w + 8 (100 KiB of x at 14x14 = 49 registers per lane over 8 waves;
//           the first version re-fetched them from L2 every round and spent 24 k of its 41 k cycles per round there).
//   round = 32 expanded channels.  Step s of the software pipeline does, with ONE workgroup barrier at its end,
//             E(s+1)  x (registers) x expand weights (LDS, staged by LDS-DMA) -> BN + PReLU -> E-image[(s+1)&1]
//                     (row-padded: a zero pixel after every row, zero rows around / between images: no bounds checks)
//             D(s)    3x3 depthwise + BN + PReLU, E-image[s&1] -> D-tile[s&1]  (a lane marches down a 7-row strip with
//                     the 3x3 window in registers: 3 LDS reads per output instead of 9)
//             P(s-1)  D-tile[(s-1)&1] x project weights (registers) -> the output tile, which stays in registers
//                     (wave w owns 16 output channels of every pixel: no imbalance, no reduction)
//           Waves 0-3 run E, P, D and waves 4-7 run D, E, P: the two waves of a SIMD are in different pipes most of
//           the time (VALU under the partner's MFMAs) although every wave runs the same three phases.
//   v_mfma_f32_16x16x4_f32: 196 pixels are 12.25 tiles of 16 rows (13: 6 % padding) but 6.1 of 32 (7: 14 %).
//
// Traffic per tile: x once, y once, weights from L2.  The expanded tensor (2 G/C times the size of x) is neither written
// nor read: SURVEY 8(d)'s op-granular model counts it four times.
#include <string.h>

#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Nothing may be scheduled across this point: keeps the loads / LDS reads of one unrolled item with that item.
#define FP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct DwBlockArgs {
  const float* in;
  float* out;
  const float* we;    // expand weights packed [C/4][G][4]
  const float* par;   // [15][G]: expand scale, bias, slope; 9 depthwise taps; depthwise scale, bias, slope
  const float* wp;    // project weights packed [G/4][C][4], then [C] scale, [C] bias
  int N, has_res;   // dense NHWC in / out: pixel stride C, image stride H*W*C; G = 2*C (every residual block)
#ifdef FP_DWB_STAMPS
  unsigned long long* stamps;   // lab builds only (tools/lab/dwblock_lab.hip): s_memtime per phase of every step
#endif
};

// In-kernel phase stamps, compiled in by the lab harness only: [block < 8][wave < 8][step < 8][5]
#ifdef FP_DWB_STAMPS
#define DWB_STAMP(k)                                                                                        \
  do {                                                                                                      \
    if (p.stamps && blockIdx.x < 8 && s < 8 && (threadIdx.x & 63) == 0) {                                   \
      unsigned long long tt_;                                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                           \
      p.stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 8 + s) * 5 + (k)] = tt_;                            \
    }                                                                                                       \
  } while (0)
#else
#define DWB_STAMP(k) do { } while (0)
#endif

template <int C, int HW, int RB, int NIMG>
struct DwbCfg {
  static_assert(NIMG == 1 || RB == HW, "several images per tile: whole images only");
  static_assert(HW % RB == 0 && RB % 7 == 0 && (C == 64 || C == 128), "");
  static constexpr int G = 2 * C;                      // expanded channels (Depth_Wise `groups`)
  static constexpr int HALO = RB < HW ? 1 : 0;
  static constexpr int NBAND = HW / RB;
  static constexpr int IPX = HW * HW;                  // pixels of an image
  static constexpr int OPX = NIMG * RB * HW;           // output pixels of a tile
  static constexpr int ER = RB + 2 * HALO;             // expanded rows of a tile (one image)
  static constexpr int EPX = NIMG * ER * HW;           // expanded pixels of a tile
  static constexpr int MTE = (EPX + 15) / 16;
  static constexpr int MTP = (OPX + 15) / 16;
  static constexpr int NOWN = (MTE + 7) / 8;           // 16-pixel row tiles of x a wave owns (w, w + 8)
  static constexpr int ROWP = HW + 1;                  // slots per E-image row (one zero pad pixel)
  static constexpr int VR = NIMG > 1 ? NIMG * (HW + 1) - 1 : ER;   // rows of the E-image (zero rows between images)
  static constexpr int NSLOT = (VR + 2) * ROWP + 1;    // + zero row above / below, + the leading pad pixel
  static constexpr int KCH = 32;                       // expanded channels per round
  static constexpr int LDD = KCH + 4;                  // D-tile row stride
  static constexpr int EB = (NSLOT + 1) * KCH;         // floats; slot NSLOT swallows the rows past EPX
  static constexpr int DB = MTP * 16 * LDD;
  static constexpr bool WE_LDS = C == 128;             // expand weights of a round through LDS (C = 64: registers)
  static constexpr int WL = WE_LDS ? C * KCH : 0;
  static constexpr int PL = 15 * KCH;
  static constexpr int LDS_FLOATS = 2 * EB + 2 * DB + 2 * WL + 3 * PL;
  static constexpr int KQ = C / 16;                    // float4 A fragments per row tile
  static constexpr int NSTRIP = OPX / 7;               // depthwise strips (7 output rows of one column) per channel pair
  static_assert(MTE <= 16 && NSTRIP <= 32 && OPX % 7 == 0, "");
  // project: C = 128: wave w owns output columns 16w.. of every row tile; C = 64: columns 16(w & 3).., row tiles
  // NPM(w >> 2) .. NPM(w >> 2) + NPM - 1
  static constexpr int NPM = C == 128 ? MTP : (MTP + 1) / 2;
};

template <int C, int HW, int RB, int NIMG>
__global__ __launch_bounds__(512, 1) void dwblock_kernel(DwBlockArgs p) {
  using K = DwbCfg<C, HW, RB, NIMG>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Eb = smem;                  // [2][EB]
  float* Db = Eb + 2 * K::EB;        // [2][DB]
  float* Wl = Db + 2 * K::DB;        // [2][WL]
  float* Pl = Wl + 2 * K::WL;        // [3][PL]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int img0 = (tile / K::NBAND) * NIMG;
  const int r0 = (tile % K::NBAND) * RB;
  constexpr int G = K::G, R = G / K::KCH;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // wave-uniform bases (SGPR pairs) + 32-bit lane offsets
  const float* xin = p.in + (long)img0 * (K::IPX * C);
  float* yout = p.out + (long)img0 * (K::IPX * C);
  const int nimg = min(NIMG, p.N - img0);              // images of this tile that exist

  // E-pixel e of the tile -> image of the tile, row / column inside the expanded rows of that image
  auto e_decode = [&](int e, int& ii, int& er, int& ec) {
    ii = NIMG > 1 ? e / (K::ER * HW) : 0;
    const int rem = e - ii * (K::ER * HW);
    er = rem / HW;
    ec = rem - er * HW;
  };

  // lane ids as the step loop sees them: re-materialised (opaque to the optimiser) at the top of every step, so that
  // per-item slot / offset arithmetic is recomputed there instead of being hoisted out of the loop and kept live
  int lv = l15, qv = q, tv = tid;

  // ---- staging of a round's weights: LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, per-lane
  // source address), no staging registers.  Wl[c & 1] is [C/4][32][4] floats = C/8 pieces of 1 KiB (two k4 rows
  // each), Pl[c % 3] is [15][32] floats = 1920 B (waves 0 and 1; lanes past the end are masked off by EXEC).
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  auto stage = [&](int c) {
    if (K::WE_LDS) {
#pragma unroll
      for (int jj = 0; jj < C / 8 / 8; ++jj) {
        const int piece = jj * 8 + wave;                       // k4 rows 2*piece, 2*piece + 1
        const float* src = p.we + K::KCH * 4 * c + (((2 * piece + ((tv >> 5) & 1)) * G + (tv & 31)) * 4);
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Wl + (c & 1) * K::WL + piece * 256), 16, 0, 0);
      }
    }
    if (wave < 2) {
      const int t = tv;   // waves 0 and 1: t = tid
      if (t < 15 * 8) {
        const float* src = p.par + K::KCH * c + ((t >> 3) * G + 4 * (t & 7));
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Pl + (c % 3) * K::PL + wave * 256), 16, 0, 0);
      }
    }
  };
  // expand weights of a round in registers (C = 64): both 16-column tiles, k = 16j + 4q + i
  f32x4 breg[2][K::KQ];
  auto load_breg = [&](int c) {
    if (!K::WE_LDS) {
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int j = 0; j < K::KQ; ++j)
          breg[n][j] = *(const f32x4*)(p.we + K::KCH * 4 * c + ((qv * G + lv) * 4 + (4 * j * G + n * 16) * 4));
    }
  };
  // project weights of a round: rows k = 32c + 16jj + 4q + i, this wave's 16 columns
  const int pcol = C == 128 ? wave : (wave & 3);
  const int pm0 = C == 128 ? 0 : (wave >> 2) * K::NPM;
  f32x4 pbw[2];
  auto load_pbw = [&](int c) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      pbw[jj] = *(const f32x4*)(p.wp + 8 * C * 4 * c + ((qv * C + pcol * 16 + lv) * 4 + 4 * jj * C * 4));
  };

  // ---- prologue: x -> registers, zero both E-images (the pads stay zero for the whole tile), stage rounds 0 and 1 ----
  f32x4 afr[K::NOWN][K::KQ];
#pragma unroll
  for (int t = 0; t < K::NOWN; ++t) {
    const int m = wave + 8 * t;
    const int e = min(16 * min(m, K::MTE - 1) + l15, K::EPX - 1);
    int ii, er, ec;
    e_decode(e, ii, er, ec);
    const int gr = min(max(r0 - K::HALO + er, 0), HW - 1);
    const int aoff = (min(ii, nimg - 1) * K::IPX + gr * HW + ec) * C + 4 * q;
#pragma unroll
    for (int j = 0; j < K::KQ; ++j) afr[t][j] = *(const f32x4*)(xin + (aoff + 16 * j));
  }
  stage(0);
  if (R > 1) stage(1);
  load_breg(0);
  for (int i = tid; i < 2 * K::EB / 4; i += 512) *(f32x4*)&Eb[i * 4] = z;

  // output tile of this wave
  f32x4 pacc[K::NPM];
#pragma unroll
  for (int m = 0; m < K::NPM; ++m) pacc[m] = z;

  // E-image slot (in floats, + column) of the 4 accumulator rows of each owned row tile, and their validity as a 0/1
  // factor: computed once per tile (fp32 MFMAs and VALU instructions share the SIMD's vector ALU -- a wave's VALU work is
  // NOT hidden under its partner's MFMAs (tools/lab/coexec_lab.hip: one VALU instruction per 32-cycle MFMA) -- so every
  // instruction of the per-round epilogues counts)
  int eslot[K::NOWN][4];
  f32x2 evalid[K::NOWN][2];
#pragma unroll
  for (int t = 0; t < K::NOWN; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int e = 16 * (wave + 8 * t) + 4 * q + r;
      int ii, er, ec;
      e_decode(min(e, K::EPX - 1), ii, er, ec);
      const int vrow = NIMG > 1 ? ii * (HW + 1) + er : er;
      const int slot = e < K::EPX ? (vrow + 1) * K::ROWP + ec + 1 : K::NSLOT;
      eslot[t][r] = slot * K::KCH + l15;
      evalid[t][r >> 1][r & 1] = (K::HALO == 0 || (unsigned)(r0 - 1 + er) < (unsigned)HW) ? 1.f : 0.f;
    }

  // E(c): x (registers) x expand weights -> BN + PReLU -> E-image[c & 1]
  auto expand = [&](int c) {
    float* Ec = Eb + (c & 1) * K::EB;
    const float* Wc = Wl + (c & 1) * K::WL;
    const float* Pc = Pl + (c % 3) * K::PL;
    f32x2 es[2], eb[2], em[2];   // BN scale, BN bias, PReLU slope - 1 (both halves equal: packed math on row pairs)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const float a = Pc[n * 16 + lv], b = Pc[K::KCH + n * 16 + lv], sl = Pc[2 * K::KCH + n * 16 + lv] - 1.f;
      es[n] = f32x2{a, a};
      eb[n] = f32x2{b, b};
      em[n] = f32x2{sl, sl};
    }
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const int m = wave + 8 * t;
      if (m < K::MTE) {
        f32x4 acc[2] = {z, z};
#pragma unroll
        for (int j = 0; j < K::KQ; ++j) {
          f32x4 b[2];
#pragma unroll
          for (int n = 0; n < 2; ++n)
            b[n] = K::WE_LDS ? *(const f32x4*)&Wc[((4 * j + qv) * K::KCH + n * 16 + lv) * 4] : breg[n][j];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t][j][i], b[n][i], acc[n], 0, 0, 0);
              FP_MFMA_ORDER();
            }
        }
        // v = acc*s + b; PReLU(v) = v + (slope - 1)*min(v, 0); rows outside the image are zeros (the depthwise pads
        // the EXPANDED tensor); rows past EPX go to slot NSLOT
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 v = f32x2{acc[n][2 * h], acc[n][2 * h + 1]} * es[n] + eb[n];
            const f32x2 neg = {__builtin_fminf(v[0], 0.f), __builtin_fminf(v[1], 0.f)};
            v = neg * em[n] + v;
            if (K::HALO) v *= evalid[t][h];
            Ec[eslot[t][2 * h] + n * 16] = v[0];
            Ec[eslot[t][2 * h + 1] + n * 16] = v[1];
          }
        FP_SCHED_FENCE();
      }
    }
  };

  // D(c): 3x3 depthwise + BN + PReLU, E-image[c & 1] -> D-tile[c & 1].  Lane = (channel pair c2, strip): a strip is 7
  // consecutive output rows of one column; the window slides down with 3 new LDS reads per output.
  auto depthwise = [&](int c) {
    const float* Ec = Eb + (c & 1) * K::EB;
    float* Dc = Db + (c & 1) * K::DB;
    const float* Pc = Pl + (c % 3) * K::PL;
    const int c2 = tv & 15, strip = tv >> 4;
    if (strip < K::NSTRIP) {
      int vrow0, col, o0;
      if (K::HALO) {                 // band of a larger image: strip = column, E row 0 is the halo row above
        col = strip; vrow0 = 1; o0 = col;
      } else if (NIMG > 1) {         // several small images: strip = (image, column)
        const int im = strip / HW;
        col = strip - im * HW; vrow0 = im * (HW + 1); o0 = im * K::IPX + col;
      } else {                       // one image: strip = (7-row part, column)
        const int part = strip / HW;
        col = strip - part * HW; vrow0 = 7 * part; o0 = vrow0 * HW + col;
      }
      f32x2 tap[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) tap[t] = *(const f32x2*)&Pc[(3 + t) * K::KCH + 2 * c2];
      const f32x2 dsc = *(const f32x2*)&Pc[12 * K::KCH + 2 * c2];
      const f32x2 dbi = *(const f32x2*)&Pc[13 * K::KCH + 2 * c2];
      const f32x2 dsl = *(const f32x2*)&Pc[14 * K::KCH + 2 * c2] - f32x2{1.f, 1.f};   // PReLU slope - 1
      // window rows: w0 = row r-1, w1 = row r, w2 = row r+1 (columns col-1 .. col+1)
      const float* base = &Ec[((vrow0 + 1) * K::ROWP + col + 1) * K::KCH + 2 * c2];   // E pixel (vrow0, col)
      f32x2 w0[3], w1[3], w2[3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        w0[dx] = *(const f32x2*)(base + (-K::ROWP + dx - 1) * K::KCH);
        w1[dx] = *(const f32x2*)(base + (dx - 1) * K::KCH);
      }
#pragma unroll
      for (int r = 0; r < 7; ++r) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) w2[dx] = *(const f32x2*)(base + ((r + 1) * K::ROWP + dx - 1) * K::KCH);
        f32x2 sacc = {0.f, 0.f};
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) sacc += w0[dx] * tap[dx];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) sacc += w1[dx] * tap[3 + dx];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) sacc += w2[dx] * tap[6 + dx];
        f32x2 v = sacc * dsc + dbi;
        const f32x2 neg = {__builtin_fminf(v[0], 0.f), __builtin_fminf(v[1], 0.f)};
        v = neg * dsl + v;                                     // PReLU(v) = v + (slope - 1)*min(v, 0)
        *(f32x2*)&Dc[(o0 + r * HW) * K::LDD + 2 * c2] = v;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          w0[dx] = w1[dx];
          w1[dx] = w2[dx];
        }
      }
    }
  };

  // P(c): D-tile[c & 1] x pbw, two 16-row tiles at a time so that consecutive MFMAs never share an accumulator
  auto project = [&](int c) {
    const float* Dc = Db + (c & 1) * K::DB;
#pragma unroll
    for (int m = 0; m < K::NPM; m += 2) {
      f32x4 af[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          af[mm][jj] = *(const f32x4*)&Dc[(16 * min(pm0 + m + mm, K::MTP - 1) + lv) * K::LDD + 16 * jj + 4 * qv];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm) {
            if (m + mm < K::NPM) {
              pacc[m + mm] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mm][jj][i], pbw[jj][i], pacc[m + mm], 0, 0, 0);
              FP_MFMA_ORDER();
            }
          }
      FP_SCHED_FENCE();   // hipcc otherwise hoists the LDS reads of every unrolled pair to the top (and spills)
    }
  };

  __syncthreads();          // E-images zeroed, Wl[0..1] / Pl[0..1] landed (the barrier drains the LDS-DMA)
  expand(0);
  load_breg(1 < R ? 1 : 0);
  load_pbw(0);
  __syncthreads();

  const bool early = wave < 4;   // waves 0-3: E, P, D; waves 4-7: D, E, P
  for (int s = 0; s < R; ++s) {
    asm volatile("" : "+v"(lv), "+v"(qv), "+v"(tv));
    DWB_STAMP(0);
    if (s + 2 < R) stage(s + 2);   // Wl[s & 1]: E(s) finished with it in the previous step; Pl[(s + 2) % 3]: free
    if (!early) depthwise(s);
    DWB_STAMP(1);
    if (s + 1 < R) {
      expand(s + 1);
      if (s + 2 < R) load_breg(s + 2);
    }
    DWB_STAMP(2);
    if (s > 0) {
      project(s - 1);
      load_pbw(s);
    }
    DWB_STAMP(3);
    if (early) depthwise(s);
    DWB_STAMP(4);
    __syncthreads();
  }
  project(R - 1);

  // ---- epilogue: BN affine (+ x), straight from the accumulators: rows 4q + r of 16-row tile m, column l15 ----
  const float ps = (p.wp + G * C)[pcol * 16 + l15];
  const float pb = (p.wp + G * C + C)[pcol * 16 + l15];
  f32x4 rv[K::NPM];
  int off[K::NPM][4];   // relative to image img0, -1 = not stored
#pragma unroll
  for (int mi = 0; mi < K::NPM; ++mi) {   // every residual load first, then the stores (vmcnt counts both, in order)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = 16 * (pm0 + mi) + 4 * q + r;
      const int oc = min(o, K::OPX - 1);
      const int ii = NIMG > 1 ? oc / K::IPX : 0;
      const int rem = oc - ii * K::IPX;
      const bool ok = pm0 + mi < K::MTP && o < K::OPX && ii < nimg;
      const int po = (min(ii, nimg - 1) * K::IPX + r0 * HW + rem) * C + pcol * 16 + l15;
      off[mi][r] = ok ? po : -1;
      rv[mi][r] = p.has_res ? xin[po] : 0.f;
    }
  }
#pragma unroll
  for (int mi = 0; mi < K::NPM; ++mi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = pacc[mi][r] * ps + pb + rv[mi][r];
      if (off[mi][r] >= 0) yout[off[mi][r]] = v;
    }
}

template <int C, int HW, int RB, int NIMG>
int launch_variant(const DwBlockArgs& a, hipStream_t s) {
  using K = DwbCfg<C, HW, RB, NIMG>;
  static_assert(K::LDS_FLOATS * 4 <= 160 * 1024, "one workgroup per CU");
  constexpr int lds = K::LDS_FLOATS * 4;
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_kernel<C, HW, RB, NIMG>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int tiles = fp_ceil_div(a.N, NIMG) * K::NBAND;
  hipLaunchKernelGGL((dwblock_kernel<C, HW, RB, NIMG>), dim3(tiles), dim3(512), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// Shapes the kernel is instantiated for (include/facepath.h, DWBLOCK).
bool fp_dwblock_supported(const fp_op& op) {
  if (op.kind != FP_OP_DWBLOCK) return false;
  if (op.stride != 1 || op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.OH != op.H || op.OW != op.W || op.H != op.W || op.Cout != op.Cin || op.out_cmul != 1) return false;
  const bool shape = (op.Cin == 128 && (op.H == 14 || op.H == 7)) || (op.Cin == 64 && op.H == 28);
  if (!shape || op.Cmid != 2 * op.Cin) return false;
  // dense NHWC on both sides: pixel stride C, image stride H*W*C (compile-time strides in the kernel)
  const long ns = (long)op.H * op.W * op.Cin;
  if (op.in_ld != op.Cin || op.out_ld != op.Cout || op.in_ns != ns || op.out_ns != ns || op.in_off % 4 || op.out_off % 4) return false;
  if (op.w_off % 4 || op.scale_off % 4 || op.slope_off % 4) return false;
  if (op.res_mode != FP_RES_NONE && op.res_mode != FP_RES_ADD_AFTER_ACT) return false;
  if (op.res_mode == FP_RES_ADD_AFTER_ACT &&
      (op.res_off != op.in_off || op.res_ns != op.in_ns || op.res_ld != op.in_ld || op.res_C != op.Cin)) return false;
  if (op.flags || op.act2) return false;
  return true;
}

int fp_launch_dwblock(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_dwblock_supported(op)) return FP_ERR_UNSUPPORTED;
  DwBlockArgs a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.we = weights + op.w_off;
  a.par = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.N = op.N;
  a.has_res = op.res_mode == FP_RES_ADD_AFTER_ACT;
  if (op.Cin == 128 && op.H == 14) return launch_variant<128, 14, 14, 1>(a, s);
  if (op.Cin == 128 && op.H == 7) return launch_variant<128, 7, 7, 3>(a, s);
  return launch_variant<64, 28, 7, 1>(a, s);
}
