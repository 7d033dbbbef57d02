// dwblockx6.hip — a whole stride-1 Mobile-FaceNet Depth_Wise block on the bf16 matrix cores with fp32-equivalent
// arithmetic (split.h: exact three-way operand split, six products, fp32 accumulation) — gfx950.
//
// Depth_Wise.forward (fde/modules/mobile_facenet/mobile_facenet.py:77-88):
//     conv (1x1 expand C -> G, BN, PReLU) -> conv_dw (3x3 depthwise, BN, PReLU) -> project (1x1 G -> C, BN) [+ x].
// dwblock.hip (fp32 MFMA) showed that the fused block is bound by vector-ALU issue: the fp32 MFMA IS a vector-ALU
// instruction (FINDINGS.md finding 18).  Here both 1x1 convs run on v_mfma_f32_16x16x32_bf16, the depthwise conv and the
// operand split run on the VALU beside them, and the expanded tensor still never leaves the CU.
//
//   tile      = 7 output rows of one image (14x14: two bands, 28x28: four) = one 256-thread workgroup; two workgroups
//               per CU (<= 76 KiB of LDS each), so one's VALU phase runs under the other's MFMA phase without any
//               software pipelining inside a workgroup.
//   x         = the band's input rows (+ the halo row above / below, whose expand values are recomputed: 8 rows for 7
//               at 14x14, 9 at 28x28) are loaded ONCE, split into three bf16 planes and kept in registers as the
//               activation fragments of the expand GEMM: wave w owns the 16-pixel tiles w, w + 4, ...
//   round     = 32 expanded channels:
//       E   W_e^T (LDS, three bf16 planes, staged by LDS-DMA) x x^T (registers) -> BN + PReLU -> E-image (fp32,
//           row-padded: zero pixel after every row, zero rows outside the image).  The operands are swapped
//           (D^T = W^T X^T), so a lane holds four consecutive channels of ONE pixel: 16-byte LDS writes.
//       D   3x3 depthwise + BN + PReLU on the VALU (a lane = channel pair x column, marching down the rows with the
//           window in registers), result split into bf16 planes -> D-tile [3][pixels][32]
//       P   W_p^T (registers, prefetched during D) x D^T -> output tile in registers (wave w owns C / 4 output channels)
//     two workgroup barriers per round (three more at 28x28, where D / P run in two row chunks to keep LDS <= 80 KiB).
//   epilogue  = BN affine + x, 16-byte loads / stores straight from the accumulators.
//
// Traffic per tile: x once (+ halo rows), y once, weights from L2 (the split weights are 1.5x the fp32 ones).
// MFMA work per 14x14 block: 2 bands x 8 rounds x (7 x 2 x 4 + 7 x 8) tile-slabs x 6 = 10.7 k MFMAs of 16 cycles per
// image against 25 k of 32 cycles for dwblock.hip.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "split.h"

namespace {

// Workgroup -> tile: XCD-aware (common.h fp_xcd_block): the bands of one image -- which share their halo rows -- run on the
// same XCD next to each other in time, and the second reader finds the rows in that L2 (Mobile-FaceNet at 528 crops:
// 2.37-2.40 -> 2.35 ms).  FP_X6_XCD = 0 (lab): tile = blockIdx.
#ifndef FP_X6_XCD
#define FP_X6_XCD 1
#endif
// Lab knob: wave priority during the depthwise (VALU) phase.  The partner workgroup's MFMA stream takes the SIMD's issue port
// for half of every MFMA; a D-phase wave with a higher priority gets the port whenever it is ready.
#ifndef FP_X6_DPRIO
#define FP_X6_DPRIO 0
#endif
// Lab knob: the depthwise taps of dwblock_x6_kernel as plain instead of packed FMAs.
#ifndef FP_X6_DSCALAR
#define FP_X6_DSCALAR 0
#endif
#define X6_DPRIO_ON()  do { if (FP_X6_DPRIO) __builtin_amdgcn_s_setprio(FP_X6_DPRIO); } while (0)
#define X6_DPRIO_OFF() do { if (FP_X6_DPRIO) __builtin_amdgcn_s_setprio(0); } while (0)
__device__ __forceinline__ int x6_tile_of_block() { return FP_X6_XCD ? (int)fp_xcd_block() : (int)blockIdx.x; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

struct DwbX6Args {
  const float* in;
  float* out;
  const unsigned short* we;   // expand weights, split: [R][3][C/32][32 g][32 k] bf16
  const float* par;           // [15][G]: expand scale, bias, slope; 9 depthwise taps; depthwise scale, bias, slope
  const unsigned short* wp;   // project weights, split: [R][3][C co][32 g] bf16
  const float* paff;          // [C] project BN scale, [C] bias
  const float* dwin;          // FP_OPF_IN_DW: [12][Cin] of the depthwise Conv_block in front: 9 taps, BN scale, BN bias, PReLU slope
  int N, has_res;
#ifdef FP_X6_STAMPS
  unsigned long long* stamps;   // lab builds only (tools/lab/x6_lab.hip): s_memtime per phase, [block < 4][wave][round][8]
#endif
};

#ifdef FP_X6_STAMPS
#define X6_STAMP(k)                                                                                   \
  do {                                                                                                \
    if (p.stamps && blockIdx.x < 4 && (threadIdx.x & 63) == 0) {                                      \
      unsigned long long tt_;                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                     \
      p.stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 9 + s) * 8 + (k)] = tt_;                      \
    }                                                                                                 \
  } while (0)
#else
#define X6_STAMP(k) do { } while (0)
#endif

template <int C, int HW>
struct X6Cfg {
  static_assert((C == 128 && HW == 14) || (C == 64 && HW == 28), "");
  static constexpr int RB = 7;                           // output rows of a tile
  static constexpr int G = 2 * C, KCH = 32, R = G / KCH, KS = C / 32;
  static constexpr int NBAND = HW / RB;
  static constexpr int ERMAX = NBAND == 2 ? RB + 1 : RB + 2;   // computed expand rows of a band, at most
  static constexpr int EPX = ERMAX * HW;
  static constexpr int MTE = (EPX + 15) / 16;
  static constexpr int NOWN = (MTE + 3) / 4;             // 16-pixel tiles of x a wave owns (w, w + 4, ...)
  static constexpr int ROWP = HW + 1;                    // slots per E-image row (one zero pad pixel)
  static constexpr int NSLOT = (RB + 2) * ROWP + 1;      // + the leading pad pixel; slot NSLOT swallows junk rows
  static constexpr int LDE = 36;                         // floats per slot (odd number of 16-byte units)
  static constexpr int EB = (NSLOT + 1) * LDE;           // floats
  static constexpr int NCHUNK = HW == 28 ? 2 : 1;        // D / P row chunks
  static constexpr int crow0(int c) { return c == 0 ? 0 : 4; }
  static constexpr int crows(int c) { return NCHUNK == 1 ? RB : (c == 0 ? 4 : 3); }
  static constexpr int mtc(int c) { return (crows(c) * HW + 15) / 16; }
  static constexpr int tbase(int c) { return c == 0 ? 0 : mtc(0); }
  static constexpr int MTP = NCHUNK == 1 ? mtc(0) : mtc(0) + mtc(1);
  static constexpr int DPL = mtc(0) * 16 * 32;           // bf16 elements per plane of the D tile (largest chunk)
  static constexpr int WL = 3 * KS * 32 * 32;            // bf16 elements of a round's expand weights
  static constexpr int PL = 15 * KCH;                    // floats of a round's parameters
  static constexpr int NCT = C / 64;                     // 16-channel output tiles per wave
  static constexpr int NPASS = (HW + 15) / 16;           // depthwise passes over the columns
  static constexpr int LDS_BYTES = EB * 4 + 3 * DPL * 2 + WL * 2 + 2 * PL * 4;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
  static_assert(MTE <= 4 * NOWN && NOWN % 2 == 0 && (WL * 2) % 4096 == 0, "");
};

template <int C, int HW>
__global__ __launch_bounds__(256, 2) void dwblock_x6_kernel(DwbX6Args p) {
  using K = X6Cfg<C, HW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;                                        // [NSLOT + 1][LDE]
  unsigned short* Dl = (unsigned short*)(smem_raw + K::EB * 4);        // [3][DPL]
  unsigned short* Wl = Dl + 3 * K::DPL;                                // [3][KS][32][32]
  float* Pl = (float*)(Wl + K::WL);                                    // [2][15][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = x6_tile_of_block();
  const int img = tile / K::NBAND;
  const int r0 = (tile % K::NBAND) * K::RB;
  const int elo = r0 > 0 ? r0 - 1 : 0;                                 // first / last computed expand row
  const int ehi = r0 + K::RB < HW ? r0 + K::RB : HW - 1;
  const int epx = (ehi - elo + 1) * HW;                                // computed expand pixels of this band
  const int vr0 = r0 > 0 ? 0 : 1;                                      // E-image row of expand row elo
  constexpr int G = K::G, R = K::R, KS = K::KS;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const float* xin = p.in + (long)img * (HW * HW * C);
  float* yout = p.out + (long)img * (HW * HW * C);

  // ---- staging: a round's expand weights (LDS-DMA, 1 KiB per wave and instruction) and parameters ----
  auto stage = [&](int s) {
    int ln = lane;
    asm volatile("" : "+v"(ln));   // rebuilt per round: hoisted out of the round loop the address is one more live register pair, and a spill
    const unsigned char* src = (const unsigned char*)p.we + (long)s * (K::WL * 2) + ln * 16;
#pragma unroll
    for (int j = 0; j < K::WL * 2 / 4096; ++j) {
      const int chunk = j * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(src + chunk * 1024), (lds_ptr)((unsigned char*)Wl + chunk * 1024), 16, 0, 0);
    }
    if (wave < 2 && tid < 15 * 8) {
      const float* ps = p.par + K::KCH * s + ((tid >> 3) * G + 4 * (tid & 7));
      __builtin_amdgcn_global_load_lds((gbl_ptr)ps, (lds_ptr)(Pl + (s & 1) * K::PL + wave * 256), 16, 0, 0);
    }
  };

  // ---- prologue ----
  stage(0);
  for (int i = tid; i < K::EB / 4; i += 256) *(f32x4*)&El[i * 4] = z;
  // x -> registers as split fragments: lane = (pixel l15 of the tile, k group q): k = 32 ks + 8 q .. + 7
  fp_frag3 xf[K::NOWN][KS];
  int eoff[K::NOWN];   // float offset of this lane's pixel of each owned tile in the E-image (+ 4 q)
#pragma unroll
  for (int t = 0; t < K::NOWN; ++t) {
    const int m = wave + 4 * t;
    const int e = 16 * m + l15;
    const int ec_ = min(e, epx - 1);
    const float* src = xin + (elo * HW + ec_) * C + 8 * q;
    f32x4 lo[KS], hi[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      lo[ks] = *(const f32x4*)(src + 32 * ks);
      hi[ks] = *(const f32x4*)(src + 32 * ks + 4);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[t][ks] = fp_split8(lo[ks], hi[ks]);
    const int er = ec_ / HW, ecol = ec_ - er * HW;
    const int slot = e < epx ? (vr0 + er) * K::ROWP + ecol + 1 : K::NSLOT;
    eoff[t] = slot * K::LDE + 4 * q;
  }
  // output tile of this wave: [pixel tile][channel tile], lane = (pixel l15, channels 16 ct + 4 q .. + 3)
  f32x4 pacc[K::MTP][K::NCT];
#pragma unroll
  for (int t = 0; t < K::MTP; ++t)
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) pacc[t][j] = z;

  // project weights of a round: this wave's channel tiles, k = 32 s + 8 q .. + 7
  fp_frag3 pbw[K::NCT];
  auto load_pbw = [&](int s) {
#pragma unroll
    for (int j = 0; j < K::NCT; ++j)
    {
      const unsigned short* src = p.wp + ((long)(s * 3 * C + 16 * (wave * K::NCT + j) + l15) * 32 + 8 * q);
      pbw[j].h = *(const u32x4*)src;
      pbw[j].m = *(const u32x4*)(src + C * 32);
      pbw[j].l = *(const u32x4*)(src + 2 * C * 32);
    }
  };

  // E(s): expand round s -> E-image.  Straight-line: a wave whose last owned tile does not exist (m >= MTE) computes it
  // on clamped pixels and drops it into the junk slot -- a wave-uniform branch around every MFMA group keeps hipcc from
  // moving the next group's LDS reads above this group's MFMAs.  The weight fragments of group g + 1 are requested
  // before the MFMAs of group g.
  auto expand = [&](int s) {
    const float* Pc = Pl + (s & 1) * K::PL;
    f32x4 acc[K::NOWN][2];
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) acc[t][0] = acc[t][1] = z;
    u32x4 wf[2][3];
    auto ldw = [&](int g, u32x4* w) {
      const int ks = g >> 1, nt = g & 1;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) w[pl] = *(const u32x4*)(Wl + (((pl * KS + ks) * 32 + 16 * nt + l15) * 32 + 8 * q));
    };
    ldw(0, wf[0]);
#pragma unroll
    for (int g = 0; g < 2 * KS; ++g) {
      if (g + 1 < 2 * KS) ldw(g + 1, wf[(g + 1) & 1]);
      const int ks = g >> 1, nt = g & 1;
#pragma unroll
      for (int t = 0; t < K::NOWN; t += 2)
        fp_mfma_x6_2b(wf[g & 1][0], wf[g & 1][1], wf[g & 1][2], xf[t][ks], xf[t + 1][ks], acc[t][nt], acc[t + 1][nt]);
    }
    // v = acc*s + b; PReLU(v) = v + (slope - 1)*min(v, 0); channels 16 nt + 4 q + i of pixel l15
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 es = *(const f32x4*)&Pc[16 * nt + 4 * q];
      const f32x4 eb = *(const f32x4*)&Pc[K::KCH + 16 * nt + 4 * q];
      const f32x4 em = *(const f32x4*)&Pc[2 * K::KCH + 16 * nt + 4 * q] - f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t) {
        f32x4 v = acc[t][nt] * es + eb;
        f32x4 neg;
#pragma unroll
        for (int i = 0; i < 4; ++i) neg[i] = __builtin_fminf(v[i], 0.f);
        v = neg * em + v;
        *(f32x4*)&El[eoff[t] + 16 * nt] = v;
      }
    }
  };

  // D(s, c): 3x3 depthwise + BN + PReLU of chunk c's rows, E-image -> D-tile (three bf16 planes).
  // lane = (channel pair c2, column strip): the window slides down the rows, TWO output rows per step (two independent FMA
  // chains), and the two E-image rows of the next step are requested between this step's FMAs and its BN / PReLU / split /
  // stores, into the registers of the rows that just left the window.  hipcc keeps an LDS read behind every earlier LDS
  // write (E-image and D-tile may alias for all it knows): with the reads at the top of each row the seven rows were seven
  // serial round trips read -> 9 dependent FMAs -> split -> write (3000 cycles alone, 5300 beside the partner's MFMAs; FINDINGS 47)
  auto depthwise = [&](int s, auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int NR = K::crows(c);
    const float* Pc = Pl + (s & 1) * K::PL;
    const int c2 = tid & 15, strip = tid >> 4;
    f32x2 tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap[t] = *(const f32x2*)&Pc[(3 + t) * K::KCH + 2 * c2];
    const f32x2 dsc = *(const f32x2*)&Pc[12 * K::KCH + 2 * c2];
    const f32x2 dbi = *(const f32x2*)&Pc[13 * K::KCH + 2 * c2];
    const f32x2 dsl = *(const f32x2*)&Pc[14 * K::KCH + 2 * c2] - f32x2{1.f, 1.f};
#pragma unroll
    for (int ps = 0; ps < K::NPASS; ++ps) {
      const int col = strip + 16 * ps;
      if (col < HW) {
        // E pixel (vr, col + dx - 1) is slot vr*ROWP + col + dx; output row r of the band reads vr = r, r + 1, r + 2
        const float* base = &El[(K::crow0(c) * K::ROWP + col) * K::LDE + 2 * c2];
        unsigned* dst = (unsigned*)Dl + (col * 32 + 2 * c2) / 2;
        auto ldrow = [&](int vr, f32x2* w) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) w[dx] = *(const f32x2*)(base + (vr * K::ROWP + dx) * K::LDE);
        };
        auto taps = [&](const f32x2* w0, const f32x2* w1, const f32x2* w2) {
#if FP_X6_DSCALAR
          // lab: plain v_fma_f32 -- beside an MFMA stream a packed FMA retires every 22.5 cycles, a plain one every 8.7 (finding 24)
          float a0, a1;
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a0) : "v"(w0[0][0]), "v"(tap[0][0]));
          asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a1) : "v"(w0[0][1]), "v"(tap[0][1]));
#define X6_SFMA(W, T)                                                                  \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"((W)[0]), "v"((T)[0]));         \
  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"((W)[1]), "v"((T)[1]));
          X6_SFMA(w0[1], tap[1]) X6_SFMA(w0[2], tap[2])
          X6_SFMA(w1[0], tap[3]) X6_SFMA(w1[1], tap[4]) X6_SFMA(w1[2], tap[5])
          X6_SFMA(w2[0], tap[6]) X6_SFMA(w2[1], tap[7]) X6_SFMA(w2[2], tap[8])
#undef X6_SFMA
          return f32x2{a0, a1};
#else
          f32x2 sacc = w0[0] * tap[0];
          sacc += w0[1] * tap[1];
          sacc += w0[2] * tap[2];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w1[dx] * tap[3 + dx];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w2[dx] * tap[6 + dx];
          return sacc;
#endif
        };
        auto finish = [&](f32x2 sacc, int r) {
          f32x2 v = sacc * dsc + dbi;
          const f32x2 neg = {__builtin_fminf(v[0], 0.f), __builtin_fminf(v[1], 0.f)};
          v = neg * dsl + v;
          unsigned h, m, l;
          fp_split_pair(v[0], v[1], h, m, l);
          dst[(r * HW * 32) / 2] = h;
          dst[(K::DPL + r * HW * 32) / 2] = m;
          dst[(2 * K::DPL + r * HW * 32) / 2] = l;
        };
        f32x2 wa[3], wb[3], c0[3], c1[3];
        ldrow(0, wa);
        ldrow(1, wb);
        ldrow(2, c0);
        if (NR > 1) ldrow(3, c1);
#pragma unroll
        for (int r = 0; r < NR; r += 2) {
          const f32x2 s0 = taps(wa, wb, c0);
          f32x2 s1 = s0;
          if (r + 1 < NR) s1 = taps(wb, c0, c1);
          f32x2 na[3], nb[3];
          if (r + 2 < NR) {
            ldrow(r + 4, na);
            if (r + 3 < NR) ldrow(r + 5, nb);
          }
          finish(s0, r);
          if (r + 1 < NR) finish(s1, r + 1);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            wa[dx] = c0[dx];
            wb[dx] = c1[dx];
            c0[dx] = na[dx];
            c1[dx] = nb[dx];
          }
        }
      }
    }
  };

  // P(s, c): W_p^T (registers) x D^T -> pacc; two accumulators per step (two channel tiles, or two pixel tiles where a
  // wave owns one channel tile); the D fragments of the next step are requested before this step's MFMAs
  auto project = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int TS = K::NCT == 2 ? 1 : 2;                 // pixel tiles per step
    constexpr int NSTEP = (K::mtc(c) + TS - 1) / TS;
    fp_frag3 df[2][TS];
    auto ldd = [&](int st, fp_frag3* d) {
#pragma unroll
      for (int i = 0; i < TS; ++i) {
        const int t = st * TS + i < K::mtc(c) ? st * TS + i : K::mtc(c) - 1;
        const unsigned short* src = Dl + ((16 * t + l15) * 32 + 8 * q);
        d[i].h = *(const u32x4*)src;
        d[i].m = *(const u32x4*)(src + K::DPL);
        d[i].l = *(const u32x4*)(src + 2 * K::DPL);
      }
    };
    ldd(0, df[0]);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      if (st + 1 < NSTEP) ldd(st + 1, df[(st + 1) & 1]);
      const fp_frag3* d = df[st & 1];
      if (K::NCT == 2) {
        fp_mfma_x6_2a(pbw[0], pbw[K::NCT - 1], d[0].h, d[0].m, d[0].l, pacc[K::tbase(c) + st][0], pacc[K::tbase(c) + st][K::NCT - 1]);
      } else if (st * TS + 1 < K::mtc(c)) {
        fp_mfma_x6_2(pbw[0], d[0], pbw[0], d[TS - 1], pacc[K::tbase(c) + st * TS][0], pacc[K::tbase(c) + st * TS + TS - 1][0]);
      } else {
        pacc[K::tbase(c) + st * TS][0] = fp_mfma_x6(pbw[0].h, pbw[0].m, pbw[0].l, d[0].h, d[0].m, d[0].l, pacc[K::tbase(c) + st * TS][0]);
      }
    }
  };

  // workgroup barrier that waits for this wave's LDS traffic only: an LDS-DMA (vmcnt) may stay in flight across it
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  __syncthreads();   // E-image zeroed, round 0 staged (the barrier drains the LDS-DMA)

  for (int s = 0; s < R; ++s) {
    X6_STAMP(0);
    expand(s);
    X6_STAMP(1);
    __syncthreads();                       // E-image complete; every wave is done with this round's expand weights
    X6_STAMP(2);
    load_pbw(s);                           // first: vmcnt is in order, P waits for these and not for the DMA behind them
    if (s + 1 < R) stage(s + 1);
    X6_DPRIO_ON();
    depthwise(s, std::integral_constant<int, 0>());
    X6_DPRIO_OFF();
    X6_STAMP(3);
    lds_barrier();                         // D-tile complete
    X6_STAMP(4);
    project(std::integral_constant<int, 0>());
    X6_STAMP(5);
    if (K::NCHUNK > 1) {
      lds_barrier();                       // chunk 0's D-tile consumed
      X6_DPRIO_ON();
      depthwise(s, std::integral_constant<int, K::NCHUNK - 1>());
      X6_DPRIO_OFF();
      lds_barrier();
      project(std::integral_constant<int, K::NCHUNK - 1>());
    }
    if (s + 1 < R) __syncthreads();        // next round's weights and parameters landed (the DMA had D + P to do so)
  }
  {
    [[maybe_unused]] const int s = R;
    X6_STAMP(0);
  }

  // ---- epilogue: y = acc*s + b (+ x), lane = pixel l15 of the tile, channels 16 ct + 4 q .. + 3 ----
  // every shortcut value is requested before the first one is used: one round trip to L2 / HBM for the tile instead of one
  // per (chunk, channel tile)
  f32x4 rv[K::NCHUNK][K::NCT][K::mtc(0)];
#pragma unroll
  for (int c = 0; c < K::NCHUNK; ++c)
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) {
      const int ch = 16 * (wave * K::NCT + j) + 4 * q;
#pragma unroll
      for (int t = 0; t < K::mtc(c); ++t) {
        const int o = min(16 * t + l15, K::crows(c) * HW - 1);
        const int off = ((r0 + K::crow0(c)) * HW + o) * C + ch;
        rv[c][j][t] = p.has_res ? *(const f32x4*)(xin + off) : z;
      }
    }
#pragma unroll
  for (int c = 0; c < K::NCHUNK; ++c) {
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) {
      const int ch = 16 * (wave * K::NCT + j) + 4 * q;
      const f32x4 ps = *(const f32x4*)(p.paff + ch);
      const f32x4 pb = *(const f32x4*)(p.paff + C + ch);
#pragma unroll
      for (int t = 0; t < K::mtc(c); ++t) {
        const int o = 16 * t + l15;
        const int off = ((r0 + K::crow0(c)) * HW + o) * C + ch;
        const f32x4 v = pacc[K::tbase(c) + t][j] * ps + pb + rv[c][j][t];
        if (o < K::crows(c) * HW) *(f32x4*)(yout + off) = v;
      }
    }
  }
}

template <int C, int HW>
int launch_x6(const DwbX6Args& a, hipStream_t s) {
  using K = X6Cfg<C, HW>;
  const int lds_bytes = fp_get_knobs().x6_lds_min > K::LDS_BYTES ? fp_get_knobs().x6_lds_min : K::LDS_BYTES;   // lab knob, 0 in the product
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_x6_kernel<C, HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            lds_bytes);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((dwblock_x6_kernel<C, HW>), dim3(a.N * K::NBAND), dim3(256), lds_bytes, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// 7 x 7 output tiles for the 128-channel blocks (14x14 maps: four tiles per image, 7x7 maps: one).  Same rounds and the
// same arithmetic as the band kernel above; the smaller tile is about occupancy and granularity:
//   * 51 KiB of LDS and <= 168 VGPRs: THREE workgroups per CU (the band form: two, 250 VGPRs), i.e. three waves per
//     SIMD whose MFMA / VALU / waiting phases fill each other's gaps (the stamps of the band form showed each phase
//     latency-bound on its own: E 2400, D 3000 -> 5300 beside the partner, P 1850 cycles per round);
//   * 2112 tiles on 768 slots at 528 crops instead of 1056 on 512 (three rounds of workgroups for 2.06 rounds of work);
//   * the halo costs nothing at 14x14: the 8 x 8 expand pixels of a tile are exactly four 16-pixel MFMA tiles (one per
//     wave) where the band form has seven for four waves; the project GEMM pays 64 rows for 49 pixels.
//   E-image: explicit 9 x 9 slot grid (tile + one pixel around), border slots outside the image stay zero.
template <int HW>
struct X6QCfg {
  static_assert(HW == 14 || HW == 7, "");
  static constexpr int C = 128, G = 256, KCH = 32, R = G / KCH, KS = C / 32;
  static constexpr int TPR = HW / 7, TPI = TPR * TPR;   // tiles per row / per image
  static constexpr int NC = HW == 14 ? 8 : 7;           // computed expand rows = columns of a tile
  static constexpr int EPX = NC * NC;                   // 64 / 49 expand pixels: four 16-pixel tiles, one per wave
  static constexpr int NSLOT = 81, LDE = 36, EB = (NSLOT + 1) * LDE;
  static constexpr int MTP = 4;                         // 49 output pixels in four 16-pixel tiles
  static constexpr int DPL = MTP * 16 * 32;
  static constexpr int WL = 3 * KS * 32 * 32, PL = 15 * KCH;
  static constexpr int LDS_BYTES = EB * 4 + 3 * DPL * 2 + WL * 2 + 2 * PL * 4;
  static_assert(LDS_BYTES * 3 <= 160 * 1024, "three workgroups per CU");
};

template <int HW>
__global__ __launch_bounds__(256, 3) void dwblock_x6q_kernel(DwbX6Args p) {
  using K = X6QCfg<HW>;
  constexpr int C = K::C, G = K::G, R = K::R, KS = K::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;
  unsigned short* Dl = (unsigned short*)(smem_raw + K::EB * 4);
  unsigned short* Wl = Dl + 3 * K::DPL;
  float* Pl = (float*)(Wl + K::WL);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = x6_tile_of_block();
  const int img = tile / K::TPI, tq = tile % K::TPI;
  const int r0 = (tq / K::TPR) * 7, c0 = (tq % K::TPR) * 7;
  const int er0 = r0 > 0 ? r0 - 1 : 0, ec0 = c0 > 0 ? c0 - 1 : 0;     // first computed expand row / column
  const int vr0 = r0 > 0 ? 0 : 1, vc0 = c0 > 0 ? 0 : 1;               // its place in the 9 x 9 grid
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const float* xin = p.in + (long)img * (HW * HW * C);
  float* yout = p.out + (long)img * (HW * HW * C);

  auto stage = [&](int s) {
    const unsigned char* src = (const unsigned char*)p.we + (long)s * (K::WL * 2) + lane * 16;
#pragma unroll
    for (int j = 0; j < K::WL * 2 / 4096; ++j) {
      const int chunk = j * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(src + chunk * 1024), (lds_ptr)((unsigned char*)Wl + chunk * 1024), 16, 0, 0);
    }
    if (wave < 2 && tid < 15 * 8) {
      const float* ps = p.par + K::KCH * s + ((tid >> 3) * G + 4 * (tid & 7));
      __builtin_amdgcn_global_load_lds((gbl_ptr)ps, (lds_ptr)(Pl + (s & 1) * K::PL + wave * 256), 16, 0, 0);
    }
  };

  // ---- prologue ----
  stage(0);
  for (int i = tid; i < K::EB / 4; i += 256) *(f32x4*)&El[i * 4] = z;
  // this wave's 16 expand pixels, split: lane = (pixel l15, k group q)
  fp_frag3 xf[KS];
  int eoff;
  {
    const int e = 16 * wave + l15;
    const int ecl = min(e, K::EPX - 1);
    const int er = ecl / K::NC, ec = ecl - er * K::NC;
    const float* src = xin + ((er0 + er) * HW + ec0 + ec) * C + 8 * q;
    f32x4 lo[KS], hi[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      lo[ks] = *(const f32x4*)(src + 32 * ks);
      hi[ks] = *(const f32x4*)(src + 32 * ks + 4);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = fp_split8(lo[ks], hi[ks]);
    const int slot = e < K::EPX ? (vr0 + er) * 9 + vc0 + ec : K::NSLOT;
    eoff = slot * K::LDE + 4 * q;
  }
  f32x4 pacc[K::MTP][2];
#pragma unroll
  for (int t = 0; t < K::MTP; ++t) pacc[t][0] = pacc[t][1] = z;

  fp_frag3 pbw[2];
  auto load_pbw = [&](int s) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const unsigned short* src = p.wp + ((long)(s * 3 * C + 16 * (wave * 2 + j) + l15) * 32 + 8 * q);
      pbw[j].h = *(const u32x4*)src;
      pbw[j].m = *(const u32x4*)(src + C * 32);
      pbw[j].l = *(const u32x4*)(src + 2 * C * 32);
    }
  };

  // E(s): both 16-channel tiles of the round for this wave's pixel tile (two accumulators)
  auto expand = [&](int s) {
    const float* Pc = Pl + (s & 1) * K::PL;
    f32x4 acc0 = z, acc1 = z;
    // steps g = (ks, nt): the weight fragment of step g + 1 is requested before the six MFMAs of step g (a chain on one
    // accumulator issues at the full rate: tools/lab/coexec_bf16_lab.hip)
    fp_frag3 wf[2];
    auto ldw = [&](int g, fp_frag3& w) {
      const unsigned short* src = Wl + (((g >> 1) * 32 + 16 * (g & 1) + l15) * 32 + 8 * q);
      w.h = *(const u32x4*)src;
      w.m = *(const u32x4*)(src + KS * 1024);
      w.l = *(const u32x4*)(src + 2 * KS * 1024);
    };
    ldw(0, wf[0]);
#pragma unroll
    for (int g = 0; g < 2 * KS; ++g) {
      if (g + 1 < 2 * KS) ldw(g + 1, wf[(g + 1) & 1]);
      const fp_frag3& w = wf[g & 1];
      const fp_frag3& xx = xf[g >> 1];
      if (g & 1) acc1 = fp_mfma_x6(w.h, w.m, w.l, xx.h, xx.m, xx.l, acc1);
      else acc0 = fp_mfma_x6(w.h, w.m, w.l, xx.h, xx.m, xx.l, acc0);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 es = *(const f32x4*)&Pc[16 * nt + 4 * q];
      const f32x4 eb = *(const f32x4*)&Pc[K::KCH + 16 * nt + 4 * q];
      const f32x4 em = *(const f32x4*)&Pc[2 * K::KCH + 16 * nt + 4 * q] - f32x4{1.f, 1.f, 1.f, 1.f};
      f32x4 v = (nt ? acc1 : acc0) * es + eb;
      f32x4 neg;
#pragma unroll
      for (int i = 0; i < 4; ++i) neg[i] = __builtin_fminf(v[i], 0.f);
      v = neg * em + v;
      *(f32x4*)&El[eoff + 16 * nt] = v;
    }
  };

  // D(s): lane = (channel, column): ONE channel of one of the 7 columns, all 7 rows.  Rolling form: input row i feeds the
  // bottom taps of output i - 2, the middle taps of i - 1 and the top taps of i -- three accumulators instead of a 3 x 3
  // window (same order of additions per output).  Scalar fp32 math on purpose: beside the partner waves' MFMAs a
  // v_pk_fma_f32 costs 22 cycles of issue against 2 x 8.7 for two v_fma_f32 (tools/lab/coexec_bf16_lab.hip), and one
  // channel per lane needs a third of the registers of the channel-pair form (three workgroups per CU need <= 168).
  // The three bf16 pieces are the upper halves of h, m, l: stored with ds_write_b16_d16_hi, no packing.
  auto depthwise = [&](int s) {
    const float* Pc = Pl + (s & 1) * K::PL;
    const int ch = tid & 31, col = tid >> 5;
    if (col < 7) {
      float tap[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) tap[t] = Pc[(3 + t) * K::KCH + ch];
      const float dsc = Pc[12 * K::KCH + ch], dbi = Pc[13 * K::KCH + ch], dsl = Pc[14 * K::KCH + ch] - 1.f;
      // grid slot (vr, vc) = vr*9 + vc; output (row, col) reads vr = row .. row + 2, vc = col .. col + 2
      const float* base = &El[col * K::LDE + ch];
      unsigned short* dst = Dl + (col * 32 + ch);
      float a0 = 0.f, a1 = 0.f, a2;   // a0: output i - 2 (complete after this row), a1: output i - 1, a2: output i
      // row i + 1 is requested before row i's output is stored: hipcc keeps an LDS read behind every earlier LDS write (FINDINGS 47)
      float n0 = base[0], n1 = base[K::LDE], n2 = base[2 * K::LDE];
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float x0 = n0, x1 = n1, x2 = n2;
        if (i + 1 < 9) n0 = base[((i + 1) * 9) * K::LDE], n1 = base[((i + 1) * 9 + 1) * K::LDE], n2 = base[((i + 1) * 9 + 2) * K::LDE];
        if (i >= 2) {
          a0 = __builtin_fmaf(x0, tap[6], a0);
          a0 = __builtin_fmaf(x1, tap[7], a0);
          a0 = __builtin_fmaf(x2, tap[8], a0);
          float v = __builtin_fmaf(a0, dsc, dbi);
          v = __builtin_fmaf(__builtin_fminf(v, 0.f), dsl, v);
          unsigned hu, mu, lu;
          fp_split_one(v, hu, mu, lu);
          unsigned short* d = dst + (i - 2) * 7 * 32;
          d[0] = (unsigned short)hu;
          d[K::DPL] = (unsigned short)mu;
          d[2 * K::DPL] = (unsigned short)lu;
        }
        if (i >= 1 && i < 8) {
          a1 = __builtin_fmaf(x0, tap[3], a1);
          a1 = __builtin_fmaf(x1, tap[4], a1);
          a1 = __builtin_fmaf(x2, tap[5], a1);
        }
        if (i < 7) {
          a2 = x0 * tap[0];
          a2 = __builtin_fmaf(x1, tap[1], a2);
          a2 = __builtin_fmaf(x2, tap[2], a2);
        }
        a0 = a1;
        a1 = a2;
      }
    }
  };

  auto project = [&]() {
    fp_frag3 df[2];
    auto ldd = [&](int t, fp_frag3& d) {
      const unsigned short* src = Dl + ((16 * t + l15) * 32 + 8 * q);
      d.h = *(const u32x4*)src;
      d.m = *(const u32x4*)(src + K::DPL);
      d.l = *(const u32x4*)(src + 2 * K::DPL);
    };
    ldd(0, df[0]);
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      if (t + 1 < K::MTP) ldd(t + 1, df[(t + 1) & 1]);
      const fp_frag3& d = df[t & 1];
      pacc[t][0] = fp_mfma_x6(pbw[0].h, pbw[0].m, pbw[0].l, d.h, d.m, d.l, pacc[t][0]);
      pacc[t][1] = fp_mfma_x6(pbw[1].h, pbw[1].m, pbw[1].l, d.h, d.m, d.l, pacc[t][1]);
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  __syncthreads();
  for (int s = 0; s < R; ++s) {
    X6_STAMP(0);
    expand(s);
    X6_STAMP(1);
    __syncthreads();
    X6_STAMP(2);
    load_pbw(s);
    if (s + 1 < R) stage(s + 1);
    X6_DPRIO_ON();
    depthwise(s);
    X6_DPRIO_OFF();
    X6_STAMP(3);
    lds_barrier();
    X6_STAMP(4);
    project();
    X6_STAMP(5);
    if (s + 1 < R) __syncthreads();
  }
  {
    [[maybe_unused]] const int s = R;
    X6_STAMP(0);
  }

  // ---- epilogue: pixel o = 16 t + l15 of the tile (row o / 7, column o % 7), channels 16 (2 wave + j) + 4 q .. + 3 ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ch = 16 * (wave * 2 + j) + 4 * q;
    const f32x4 ps = *(const f32x4*)(p.paff + ch);
    const f32x4 pb = *(const f32x4*)(p.paff + C + ch);
    f32x4 rv[K::MTP];
    int off[K::MTP];
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      const int o = min(16 * t + l15, 48);
      const int orow = o / 7, ocol = o - orow * 7;
      off[t] = ((r0 + orow) * HW + c0 + ocol) * C + ch;
      rv[t] = p.has_res ? *(const f32x4*)(xin + off[t]) : z;
    }
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      const f32x4 v = pacc[t][j] * ps + pb + rv[t];
      if (16 * t + l15 < 49) *(f32x4*)(yout + off[t]) = v;
    }
  }
}

template <int HW>
int launch_x6q(const DwbX6Args& a, hipStream_t s) {
  using K = X6QCfg<HW>;
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_x6q_kernel<HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            K::LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((dwblock_x6q_kernel<HW>), dim3(a.N * K::TPI), dim3(256), K::LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised form of the band kernel: one 512-thread workgroup per CU, waves 0-3 are MATRIX waves (E and P: nothing
// but MFMAs, fragment reads and the expand epilogue), waves 4-7 are VECTOR waves (the depthwise phase D, the operand
// split, all staging).  The stamps of the symmetric form (tools/lab/x6_lab.hip) showed why: every phase of a round is
// latency-bound on its own (E 2400, D 3000, P 1850 cycles for 3240 cycles of matrix-core work), and two symmetric
// workgroups per CU only interleave by accident.  Here a SIMD always holds one wave of each kind, and the software
// pipeline of dwblock.hip puts independent work into every step (ONE workgroup barrier per step):
//     step k:   matrix waves  E(k + 1) -> E-image[(k + 1) & 1],   P(k - 1) <- D-tile[(k - 1) & 1]
//               vector waves  stage round k + 2 (LDS-DMA),        D(k): E-image[k & 1] -> D-tile[k & 1]
// An MFMA holds the SIMD's issue port for 8 of its ~18 cycles: the vector wave gets about two VALU slots per MFMA
// (tools/lab/coexec_bf16_lab.hip), which is what D needs (~1.6 per MFMA of its SIMD).
#ifdef FP_X6_STAMPS
#define X6S_STAMP(k)                                                                                  \
  do {                                                                                                \
    if (p.stamps && blockIdx.x < 2 && (threadIdx.x & 63) == 0) {                                      \
      unsigned long long tt_;                                                                         \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                     \
      p.stamps[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 9 + s) * 4 + (k)] = tt_;                      \
    }                                                                                                 \
  } while (0)
#else
#define X6S_STAMP(k) do { } while (0)
#endif

template <int C, int HW>
struct X6SCfg : X6Cfg<C, HW> {
  using B = X6Cfg<C, HW>;
  static_assert(B::NCHUNK == 1, "whole-band D / P");
  static constexpr int LDS_BYTES = 2 * B::EB * 4 + 2 * 3 * B::DPL * 2 + 2 * B::WL * 2 + 3 * B::PL * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

template <int C, int HW>
__global__ __launch_bounds__(512, 1) void dwblock_x6s_kernel(DwbX6Args p) {
  using K = X6SCfg<C, HW>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;                                            // [2][EB]
  unsigned short* Dl = (unsigned short*)(smem_raw + 2 * K::EB * 4);        // [2][3][DPL]
  unsigned short* Wl = Dl + 2 * 3 * K::DPL;                                // [2][WL]
  float* Pl = (float*)(Wl + 2 * K::WL);                                    // [3][PL]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool matrix = wave < 4;
  const int mw = wave & 3;                                                 // index inside the role
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = x6_tile_of_block();
  const int img = tile / K::NBAND;
  const int r0 = (tile % K::NBAND) * K::RB;
  const int elo = r0 > 0 ? r0 - 1 : 0;
  const int ehi = r0 + K::RB < HW ? r0 + K::RB : HW - 1;
  const int epx = (ehi - elo + 1) * HW;
  const int vr0 = r0 > 0 ? 0 : 1;
  constexpr int G = K::G, R = K::R, KS = K::KS;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const float* xin = p.in + (long)img * (HW * HW * C);
  float* yout = p.out + (long)img * (HW * HW * C);

  // staging by the four vector waves (mw = 0 .. 3): round s -> Wl[s & 1], Pl[s % 3]
  auto stage = [&](int s) {
    const unsigned char* src = (const unsigned char*)p.we + (long)s * (K::WL * 2) + lane * 16;
    unsigned char* dst = (unsigned char*)(Wl + (s & 1) * K::WL);
#pragma unroll
    for (int j = 0; j < K::WL * 2 / 4096; ++j) {
      const int chunk = j * 4 + mw;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(src + chunk * 1024), (lds_ptr)(dst + chunk * 1024), 16, 0, 0);
    }
    const int t = tid - 256;
    if (mw < 2 && t < 15 * 8) {
      const float* ps = p.par + K::KCH * s + ((t >> 3) * G + 4 * (t & 7));
      __builtin_amdgcn_global_load_lds((gbl_ptr)ps, (lds_ptr)(Pl + (s % 3) * K::PL + mw * 256), 16, 0, 0);
    }
  };

  // ---- prologue ----
  if (!matrix) {
    stage(0);
    if (R > 1) stage(1);
  }
  for (int i = tid; i < 2 * K::EB / 4; i += 512) *(f32x4*)&El[i * 4] = z;
  fp_frag3 xf[K::NOWN][KS];
  int eoff[K::NOWN];
  f32x4 pacc[K::MTP][K::NCT];
  fp_frag3 pbw[K::NCT];
  if (matrix) {
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const int m = mw + 4 * t;
      const int e = 16 * m + l15;
      const int ec_ = min(e, epx - 1);
      const float* src = xin + (elo * HW + ec_) * C + 8 * q;
      f32x4 lo[KS], hi[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        lo[ks] = *(const f32x4*)(src + 32 * ks);
        hi[ks] = *(const f32x4*)(src + 32 * ks + 4);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[t][ks] = fp_split8(lo[ks], hi[ks]);
      const int er = ec_ / HW, ecol = ec_ - er * HW;
      const int slot = e < epx ? (vr0 + er) * K::ROWP + ecol + 1 : K::NSLOT;
      eoff[t] = slot * K::LDE + 4 * q;
    }
#pragma unroll
    for (int t = 0; t < K::MTP; ++t)
#pragma unroll
      for (int j = 0; j < K::NCT; ++j) pacc[t][j] = z;
  }

  auto load_pbw = [&](int s) {
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) {
      const unsigned short* src = p.wp + ((long)(s * 3 * C + 16 * (mw * K::NCT + j) + l15) * 32 + 8 * q);
      pbw[j].h = *(const u32x4*)src;
      pbw[j].m = *(const u32x4*)(src + C * 32);
      pbw[j].l = *(const u32x4*)(src + 2 * C * 32);
    }
  };

  // E(s) (matrix waves): expand round s -> E-image[s & 1]
  auto expand = [&](int s) {
    const float* Pc = Pl + (s % 3) * K::PL;
    const unsigned short* Wc = Wl + (s & 1) * K::WL;
    float* Ec = El + (s & 1) * K::EB;
    f32x4 acc[K::NOWN][2];
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) acc[t][0] = acc[t][1] = z;
    fp_frag3 wf[2];
    auto ldw = [&](int g, fp_frag3& w) {
      const unsigned short* src = Wc + (((g >> 1) * 32 + 16 * (g & 1) + l15) * 32 + 8 * q);
      w.h = *(const u32x4*)src;
      w.m = *(const u32x4*)(src + KS * 1024);
      w.l = *(const u32x4*)(src + 2 * KS * 1024);
    };
    ldw(0, wf[0]);
#pragma unroll
    for (int g = 0; g < 2 * KS; ++g) {
      if (g + 1 < 2 * KS) ldw(g + 1, wf[(g + 1) & 1]);
      const fp_frag3& w = wf[g & 1];
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t)
        acc[t][g & 1] = fp_mfma_x6(w.h, w.m, w.l, xf[t][g >> 1].h, xf[t][g >> 1].m, xf[t][g >> 1].l, acc[t][g & 1]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 es = *(const f32x4*)&Pc[16 * nt + 4 * q];
      const f32x4 eb = *(const f32x4*)&Pc[K::KCH + 16 * nt + 4 * q];
      const f32x4 em = *(const f32x4*)&Pc[2 * K::KCH + 16 * nt + 4 * q] - f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t) {
        f32x4 v = acc[t][nt] * es + eb;
        f32x4 neg;
#pragma unroll
        for (int i = 0; i < 4; ++i) neg[i] = __builtin_fminf(v[i], 0.f);
        v = neg * em + v;
        *(f32x4*)&Ec[eoff[t] + 16 * nt] = v;
      }
    }
  };

  // D(s) (vector waves): lane = (channel pair c2, column strip), all 7 rows
  auto depthwise = [&](int s) {
    const float* Pc = Pl + (s % 3) * K::PL;
    const float* Ec = El + (s & 1) * K::EB;
    unsigned short* Dc = Dl + (s & 1) * 3 * K::DPL;
    const int vt = tid - 256;
    const int c2 = vt & 15, strip = vt >> 4;
    f32x2 tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap[t] = *(const f32x2*)&Pc[(3 + t) * K::KCH + 2 * c2];
    const f32x2 dsc = *(const f32x2*)&Pc[12 * K::KCH + 2 * c2];
    const f32x2 dbi = *(const f32x2*)&Pc[13 * K::KCH + 2 * c2];
    const f32x2 dsl = *(const f32x2*)&Pc[14 * K::KCH + 2 * c2] - f32x2{1.f, 1.f};
#pragma unroll
    for (int ps = 0; ps < K::NPASS; ++ps) {
      const int col = strip + 16 * ps;
      if (col < HW) {
        const float* base = &Ec[col * K::LDE + 2 * c2];
        unsigned* dst = (unsigned*)Dc + (col * 32 + 2 * c2) / 2;
        f32x2 w0[3], w1[3], w2[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          w0[dx] = *(const f32x2*)(base + dx * K::LDE);
          w1[dx] = *(const f32x2*)(base + (K::ROWP + dx) * K::LDE);
        }
#pragma unroll
        for (int r = 0; r < K::RB; ++r) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) w2[dx] = *(const f32x2*)(base + ((r + 2) * K::ROWP + dx) * K::LDE);
          f32x2 sacc = w0[0] * tap[0];
          sacc += w0[1] * tap[1];
          sacc += w0[2] * tap[2];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w1[dx] * tap[3 + dx];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w2[dx] * tap[6 + dx];
          f32x2 v = sacc * dsc + dbi;
          const f32x2 neg = {__builtin_fminf(v[0], 0.f), __builtin_fminf(v[1], 0.f)};
          v = neg * dsl + v;
          unsigned h, m, l;
          fp_split_pair(v[0], v[1], h, m, l);
          dst[(r * HW * 32) / 2] = h;
          dst[(K::DPL + r * HW * 32) / 2] = m;
          dst[(2 * K::DPL + r * HW * 32) / 2] = l;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            w0[dx] = w1[dx];
            w1[dx] = w2[dx];
          }
        }
      }
    }
  };

  // P(s) (matrix waves): W_p^T (registers) x D^T -> pacc
  auto project = [&](int s) {
    const unsigned short* Dc = Dl + (s & 1) * 3 * K::DPL;
    fp_frag3 df[2];
    auto ldd = [&](int t, fp_frag3& d) {
      const unsigned short* src = Dc + ((16 * t + l15) * 32 + 8 * q);
      d.h = *(const u32x4*)src;
      d.m = *(const u32x4*)(src + K::DPL);
      d.l = *(const u32x4*)(src + 2 * K::DPL);
    };
    ldd(0, df[0]);
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      if (t + 1 < K::MTP) ldd(t + 1, df[(t + 1) & 1]);
      const fp_frag3& d = df[t & 1];
#pragma unroll
      for (int j = 0; j < K::NCT; ++j) pacc[t][j] = fp_mfma_x6(pbw[j].h, pbw[j].m, pbw[j].l, d.h, d.m, d.l, pacc[t][j]);
    }
  };

  __syncthreads();              // E-images zeroed, rounds 0 and 1 staged
  if (matrix) {
    expand(0);
    load_pbw(0);
  }
  __syncthreads();
  for (int k = 0; k < R; ++k) {
    [[maybe_unused]] const int s = k;
    X6S_STAMP(0);
    if (matrix) {
      if (k + 1 < R) expand(k + 1);
      X6S_STAMP(1);
      if (k > 0) {
        project(k - 1);
        load_pbw(k);            // consumed by P(k) in the next step
      }
    } else {
      if (k + 2 < R) stage(k + 2);   // Wl[k & 1]: E(k) read it in the previous step; Pl[(k + 2) % 3]: last read by D(k - 1)
      X6S_STAMP(1);
      depthwise(k);
    }
    X6S_STAMP(2);
    __syncthreads();
    X6S_STAMP(3);
  }
  if (!matrix) return;
  project(R - 1);

  // ---- epilogue (matrix waves) ----
#pragma unroll
  for (int j = 0; j < K::NCT; ++j) {
    const int ch = 16 * (mw * K::NCT + j) + 4 * q;
    const f32x4 ps = *(const f32x4*)(p.paff + ch);
    const f32x4 pb = *(const f32x4*)(p.paff + C + ch);
    f32x4 rv[K::MTP];
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      const int o = min(16 * t + l15, K::RB * HW - 1);
      rv[t] = p.has_res ? *(const f32x4*)(xin + ((r0 * HW + o) * C + ch)) : z;
    }
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      const int o = 16 * t + l15;
      const f32x4 v = pacc[t][j] * ps + pb + rv[t];
      if (o < K::RB * HW) *(f32x4*)(yout + ((r0 * HW + o) * C + ch)) = v;
    }
  }
}

template <int C, int HW>
int launch_x6s(const DwbX6Args& a, hipStream_t s) {
  using K = X6SCfg<C, HW>;
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_x6s_kernel<C, HW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            K::LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((dwblock_x6s_kernel<C, HW>), dim3(a.N * K::NBAND), dim3(512), K::LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Stride-2 Depth_Wise blocks (conv_34: 64 -> 128 on 28x28 -> 14x14, conv_45: 128 -> 128 on 14x14 -> 7x7;
// mobile_facenet.py:118,123: expand to `groups` channels, depthwise 3x3 stride 2, project, no shortcut) in the band
// form: a tile is 4 output rows = 9 input rows (2r - 1 .. 2r + 7), whose expand values fill the same 9-row E-image as
// the stride-1 bands; the depthwise phase reads rows 2r .. 2r + 2 and columns 2c - 1 .. 2c + 1 of it.  Round 2 ran
// these blocks as pws_kernel (expand, the G-channel tensor to HBM and back) + dwpw_kernel: 317 and 251 us at 528 crops.
template <int CI, int G_, int CO, int HW>
struct X6DCfg {
  static_assert((CI == 64 && G_ == 256 && CO == 128 && HW == 28) || (CI == 128 && G_ == 512 && CO == 128 && HW == 14) ||
                (CI == 64 && G_ == 128 && CO == 64 && HW == 56), "");
  static constexpr int RBO = HW == 56 ? 2 : 4;           // output rows of a tile (56-wide input: 5 x 56 expand pixels)
  static constexpr int G = G_, KCH = 32, R = G / KCH, KS = CI / 32;
  static constexpr int HO = HW / 2, WO = HW / 2;
  static constexpr int NBAND = (HO + RBO - 1) / RBO;
  static constexpr int ER = 2 * RBO + 1;                 // expand rows of a tile
  static constexpr int EPX = ER * HW;
  static constexpr int MTE = (EPX + 15) / 16;
  static constexpr int NOWN = (MTE + 3) / 4;
  static constexpr int ROWP = HW + 1;
  static constexpr int NSLOT = ER * ROWP + 1;
  static constexpr int LDE = 36;
  static constexpr int EB = (NSLOT + 1) * LDE;
  static constexpr int OPX = RBO * WO;                   // output pixels of a tile
  static constexpr int MTP = (OPX + 15) / 16;
  static constexpr int DPL = MTP * 16 * 32;
  static constexpr int WL = 3 * KS * 32 * 32;
  static constexpr int PL = 15 * KCH;
  static constexpr int NCT = CO / 64;
  static constexpr int NPART = WO <= 8 ? 2 : 1;          // depthwise strips: (row part, column)
  static constexpr int RP = RBO / NPART;                 // output rows per strip
  static constexpr int NPASS = (NPART * WO + 15) / 16;   // passes over the strips (16 per pass)
  static constexpr int LDS_BYTES = EB * 4 + 3 * DPL * 2 + WL * 2 + 2 * PL * 4;
  static_assert(LDS_BYTES <= 80 * 1024 && MTE <= 4 * NOWN && (WL * 2) % 4096 == 0, "");
};

// DWIN: the block's input first passes through a depthwise 3x3 stride-1 Conv_block (+ BN + PReLU) -- Mobile-FaceNet's conv2_dw
// in front of conv_23 (mobile_facenet.py:107-108,141-143).  It is computed in the prologue, one 32-channel slab at a time:
// the band's rows of the tensor BEFORE conv2_dw (+ one halo row and column all around, zeros outside the image) go into an
// LDS image by LDS-DMA, every lane forms the depthwise output of ITS fragment elements (pixel, 8 channels) from it and
// splits it into the expand GEMM's operand -- the conv2_dw output (424 MB at 528 crops) is neither written nor read.
template <int CI, int G_, int CO, int HW, bool DWIN>
__global__ __launch_bounds__(256, 2) void dwblock_x6d_kernel(DwbX6Args p) {
  using K = X6DCfg<CI, G_, CO, HW>;
  constexpr int G = K::G, R = K::R, KS = K::KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;
  unsigned short* Dl = (unsigned short*)(smem_raw + K::EB * 4);
  unsigned short* Wl = Dl + 3 * K::DPL;
  float* Pl = (float*)(Wl + K::WL);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = x6_tile_of_block();
  const int img = tile / K::NBAND;
  const int ro0 = (tile % K::NBAND) * K::RBO;                          // first output row
  const int nro = min(K::RBO, K::HO - ro0);                            // output rows of this band
  const int first = 2 * ro0 - 1;                                       // input row of E-image row 0
  const int elo = first > 0 ? first : 0;
  const int ehi = min(2 * ro0 + 2 * nro - 1, HW - 1);
  const int epx = (ehi - elo + 1) * HW;
  const int vr0 = elo - first;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const float* xin = p.in + (long)img * (HW * HW * CI);
  float* yout = p.out + (long)img * (K::HO * K::WO * CO);

  auto stage = [&](int s) {
    const unsigned char* src = (const unsigned char*)p.we + (long)s * (K::WL * 2) + lane * 16;
#pragma unroll
    for (int j = 0; j < K::WL * 2 / 4096; ++j) {
      const int chunk = j * 4 + wave;
      __builtin_amdgcn_global_load_lds((gbl_ptr)(src + chunk * 1024), (lds_ptr)((unsigned char*)Wl + chunk * 1024), 16, 0, 0);
    }
    if (wave < 2 && tid < 15 * 8) {
      const float* ps = p.par + K::KCH * s + ((tid >> 3) * G + 4 * (tid & 7));
      __builtin_amdgcn_global_load_lds((gbl_ptr)ps, (lds_ptr)(Pl + (s & 1) * K::PL + wave * 256), 16, 0, 0);
    }
  };

  // ---- prologue ----
  stage(0);
  fp_frag3 xf[K::NOWN][KS];
  int eoff[K::NOWN];
  if (DWIN) {
    // S-image: rows elo - 1 .. ehi + 1 of the input (ER + 2 rows), row-padded like the E-image, 32 channels per slot; it
    // overlays the E-image and D-tile regions (used only after this prologue)
    constexpr int SROWS = K::ER + 2, SSLOT = SROWS * K::ROWP + 1;
    static_assert(!DWIN || SSLOT * 32 * 4 <= K::EB * 4 + 3 * K::DPL * 2, "S-image overlays the E-image + D-tile");
    float* Sl = El;
    const int nrow = ehi - elo + 1;
    // zeroed once: the DMA below only ever writes pixels inside the image, pads and outside rows stay zero for both slabs
    for (int i = tid; i < (SSLOT * 32 + 3) / 4; i += 256) *(f32x4*)&Sl[i * 4] = z;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      __syncthreads();                                   // zero fill done / every lane is done reading the previous slab
      // valid pixels: row vr of the S-image = input row elo - 1 + vr; 8 pixels (8 lanes x 16 B each) per wave instruction.
      // The eight 16-byte units of a slot are XOR-swizzled by the slot index (the LDS-DMA writes lane * 16 contiguously, but
      // WHICH global unit a lane fetches is free): 16 consecutive slots x one unit would otherwise sit in 2 bank groups.
      for (int it = wave; it < (nrow + 2) * (HW / 8); it += 4) {
        const int vr = it / (HW / 8), c8 = it - vr * (HW / 8);
        const int row = elo - 1 + vr;
        if (row >= 0 && row < HW) {                       // wave-uniform
          const int slot0 = vr * K::ROWP + 8 * c8 + 1;
          const int unit = (lane & 7) ^ ((slot0 + (lane >> 3)) & 7);
          const float* src = xin + ((row * HW + 8 * c8 + (lane >> 3)) * CI + 32 * ks + 4 * unit);
          __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Sl + slot0 * 32), 16, 0, 0);
        }
      }
      __syncthreads();                                   // the DMA landed
      // taps outermost: one tap's weights (8 registers) at a time, one accumulator pair per owned tile -- holding all nine
      // taps' weights spilled 28 registers
      const float* dp = p.dwin + 32 * ks + 8 * q;        // this lane's depthwise parameters: channels 32 ks + 8 q .. + 7
      f32x4 a0[K::NOWN], a1[K::NOWN];
      int s0[K::NOWN];
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t) {
        const int ec_ = min(16 * (wave + 4 * t) + l15, epx - 1);
        const int er = ec_ / HW;
        s0[t] = er * K::ROWP + (ec_ - er * HW);          // input pixel (elo + er + dy - 1, ecol + dx - 1) = S slot s0 + dy*ROWP + dx
      }
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const f32x4 w0 = *(const f32x4*)(dp + t9 * CI), w1 = *(const f32x4*)(dp + t9 * CI + 4);
#pragma unroll
        for (int t = 0; t < K::NOWN; ++t) {
          const int sl = s0[t] + (t9 / 3) * K::ROWP + (t9 % 3);
          const float* sp = Sl + sl * 32;
          const f32x4 v0 = *(const f32x4*)(sp + 4 * ((2 * q) ^ (sl & 7))), v1 = *(const f32x4*)(sp + 4 * ((2 * q + 1) ^ (sl & 7)));
          if (t9 == 0) {
            a0[t] = v0 * w0;
            a1[t] = v1 * w1;
          } else {
            a0[t] += v0 * w0;
            a1[t] += v1 * w1;
          }
        }
      }
      {
        const f32x4 one = {1.f, 1.f, 1.f, 1.f};
        const f32x4 sc0 = *(const f32x4*)(dp + 9 * CI), sc1 = *(const f32x4*)(dp + 9 * CI + 4);
        const f32x4 bi0 = *(const f32x4*)(dp + 10 * CI), bi1 = *(const f32x4*)(dp + 10 * CI + 4);
        const f32x4 sl0 = *(const f32x4*)(dp + 11 * CI) - one, sl1 = *(const f32x4*)(dp + 11 * CI + 4) - one;
#pragma unroll
        for (int t = 0; t < K::NOWN; ++t) {
          f32x4 v0 = a0[t] * sc0 + bi0, v1 = a1[t] * sc1 + bi1, n0, n1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            n0[i] = __builtin_fminf(v0[i], 0.f);
            n1[i] = __builtin_fminf(v1[i], 0.f);
          }
          xf[t][ks] = fp_split8(n0 * sl0 + v0, n1 * sl1 + v1);
        }
      }
    }
    __syncthreads();                                     // before the E-image region is zeroed
  }
  for (int i = tid; i < K::EB / 4; i += 256) *(f32x4*)&El[i * 4] = z;
#pragma unroll
  for (int t = 0; t < K::NOWN; ++t) {
    const int e = 16 * (wave + 4 * t) + l15;
    const int ec_ = min(e, epx - 1);
    if (!DWIN) {
      const float* src = xin + (elo * HW + ec_) * CI + 8 * q;
      f32x4 lo[KS], hi[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        lo[ks] = *(const f32x4*)(src + 32 * ks);
        hi[ks] = *(const f32x4*)(src + 32 * ks + 4);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[t][ks] = fp_split8(lo[ks], hi[ks]);
    }
    const int er = ec_ / HW, ecol = ec_ - er * HW;
    const int slot = e < epx ? (vr0 + er) * K::ROWP + ecol + 1 : K::NSLOT;
    eoff[t] = slot * K::LDE + 4 * q;
  }
  f32x4 pacc[K::MTP][K::NCT];
#pragma unroll
  for (int t = 0; t < K::MTP; ++t)
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) pacc[t][j] = z;

  fp_frag3 pbw[K::NCT];
  auto load_pbw = [&](int s) {
#pragma unroll
    for (int j = 0; j < K::NCT; ++j) {
      const unsigned short* src = p.wp + ((long)(s * 3 * CO + 16 * (wave * K::NCT + j) + l15) * 32 + 8 * q);
      pbw[j].h = *(const u32x4*)src;
      pbw[j].m = *(const u32x4*)(src + CO * 32);
      pbw[j].l = *(const u32x4*)(src + 2 * CO * 32);
    }
  };

  auto expand = [&](int s) {
    const float* Pc = Pl + (s & 1) * K::PL;
    f32x4 acc[K::NOWN][2];
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) acc[t][0] = acc[t][1] = z;
    fp_frag3 wf[2];
    auto ldw = [&](int g, fp_frag3& w) {
      const unsigned short* src = Wl + (((g >> 1) * 32 + 16 * (g & 1) + l15) * 32 + 8 * q);
      w.h = *(const u32x4*)src;
      w.m = *(const u32x4*)(src + KS * 1024);
      w.l = *(const u32x4*)(src + 2 * KS * 1024);
    };
    ldw(0, wf[0]);
#pragma unroll
    for (int g = 0; g < 2 * KS; ++g) {
      if (g + 1 < 2 * KS) ldw(g + 1, wf[(g + 1) & 1]);
      const fp_frag3& w = wf[g & 1];
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t)
        acc[t][g & 1] = fp_mfma_x6(w.h, w.m, w.l, xf[t][g >> 1].h, xf[t][g >> 1].m, xf[t][g >> 1].l, acc[t][g & 1]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const f32x4 es = *(const f32x4*)&Pc[16 * nt + 4 * q];
      const f32x4 eb = *(const f32x4*)&Pc[K::KCH + 16 * nt + 4 * q];
      const f32x4 em = *(const f32x4*)&Pc[2 * K::KCH + 16 * nt + 4 * q] - f32x4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
      for (int t = 0; t < K::NOWN; ++t) {
        f32x4 v = acc[t][nt] * es + eb;
        f32x4 neg;
#pragma unroll
        for (int i = 0; i < 4; ++i) neg[i] = __builtin_fminf(v[i], 0.f);
        v = neg * em + v;
        *(f32x4*)&El[eoff[t] + 16 * nt] = v;
      }
    }
  };

  // D(s): lane = (channel pair c2, strip); strip = (row part, output column); output (r, c) reads E-image rows
  // 2r .. 2r + 2 and input columns 2c - 1 .. 2c + 1 = slots vr*ROWP + 2c + dx.  Consecutive outputs of a strip share a row.
  auto depthwise = [&](int s) {
    const float* Pc = Pl + (s & 1) * K::PL;
    const int c2 = tid & 15;
    f32x2 tap[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tap[t] = *(const f32x2*)&Pc[(3 + t) * K::KCH + 2 * c2];
    const f32x2 dsc = *(const f32x2*)&Pc[12 * K::KCH + 2 * c2];
    const f32x2 dbi = *(const f32x2*)&Pc[13 * K::KCH + 2 * c2];
    const f32x2 dsl = *(const f32x2*)&Pc[14 * K::KCH + 2 * c2] - f32x2{1.f, 1.f};
#pragma unroll
    for (int ps = 0; ps < K::NPASS; ++ps) {
      const int strip = (tid >> 4) + 16 * ps;
      if (strip < K::NPART * K::WO) {
        const int part = K::NPART > 1 && strip >= K::WO ? 1 : 0, col = strip - part * K::WO;
        const float* base = &El[((2 * K::RP * part) * K::ROWP + 2 * col) * K::LDE + 2 * c2];
        unsigned* dst = (unsigned*)Dl + ((K::RP * part * K::WO + col) * 32 + 2 * c2) / 2;
        // the two E-image rows of the next output row are requested between this row's FMAs and its BN / PReLU / split /
        // stores (hipcc keeps an LDS read behind every earlier LDS write; see dwblock_x6_kernel's D phase)
        auto ldrow = [&](int vr, f32x2* w) {
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) w[dx] = *(const f32x2*)(base + (vr * K::ROWP + dx) * K::LDE);
        };
        f32x2 w0[3], w1[3], w2[3];
        ldrow(0, w0);
        ldrow(1, w1);
        ldrow(2, w2);
#pragma unroll
        for (int r = 0; r < K::RP; ++r) {
          f32x2 sacc = w0[0] * tap[0];
          sacc += w0[1] * tap[1];
          sacc += w0[2] * tap[2];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w1[dx] * tap[3 + dx];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) sacc += w2[dx] * tap[6 + dx];
          f32x2 n1[3], n2[3];
          if (r + 1 < K::RP) {
            ldrow(2 * r + 3, n1);
            ldrow(2 * r + 4, n2);
          }
          f32x2 v = sacc * dsc + dbi;
          const f32x2 neg = {__builtin_fminf(v[0], 0.f), __builtin_fminf(v[1], 0.f)};
          v = neg * dsl + v;
          unsigned h, m, l;
          fp_split_pair(v[0], v[1], h, m, l);
          dst[(r * K::WO * 32) / 2] = h;
          dst[(K::DPL + r * K::WO * 32) / 2] = m;
          dst[(2 * K::DPL + r * K::WO * 32) / 2] = l;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            w0[dx] = w2[dx];
            w1[dx] = n1[dx];
            w2[dx] = n2[dx];
          }
        }
      }
    }
  };

  auto project = [&]() {
    fp_frag3 df[2];
    auto ldd = [&](int t, fp_frag3& d) {
      const unsigned short* src = Dl + ((16 * t + l15) * 32 + 8 * q);
      d.h = *(const u32x4*)src;
      d.m = *(const u32x4*)(src + K::DPL);
      d.l = *(const u32x4*)(src + 2 * K::DPL);
    };
    ldd(0, df[0]);
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      if (t + 1 < K::MTP) ldd(t + 1, df[(t + 1) & 1]);
      const fp_frag3& d = df[t & 1];
#pragma unroll
      for (int j = 0; j < K::NCT; ++j) pacc[t][j] = fp_mfma_x6(pbw[j].h, pbw[j].m, pbw[j].l, d.h, d.m, d.l, pacc[t][j]);
    }
  };
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

  __syncthreads();
  for (int s = 0; s < R; ++s) {
    expand(s);
    __syncthreads();
    load_pbw(s);
    if (s + 1 < R) stage(s + 1);
    X6_DPRIO_ON();
    depthwise(s);
    X6_DPRIO_OFF();
    lds_barrier();
    project();
    if (s + 1 < R) __syncthreads();
  }

  // ---- epilogue: output pixel o = 16 t + l15 of the band (row o / WO, column o % WO): y = acc*s + b ----
#pragma unroll
  for (int j = 0; j < K::NCT; ++j) {
    const int ch = 16 * (wave * K::NCT + j) + 4 * q;
    const f32x4 ps = *(const f32x4*)(p.paff + ch);
    const f32x4 pb = *(const f32x4*)(p.paff + CO + ch);
#pragma unroll
    for (int t = 0; t < K::MTP; ++t) {
      const int o = 16 * t + l15;
      const f32x4 v = pacc[t][j] * ps + pb;
      if (o < nro * K::WO) *(f32x4*)(yout + ((ro0 * K::WO + o) * CO + ch)) = v;
    }
  }
}

template <int CI, int G_, int CO, int HW, bool DWIN = false>
int launch_x6d(const DwbX6Args& a, hipStream_t s) {
  using K = X6DCfg<CI, G_, CO, HW>;
  const int lds_bytes = fp_get_knobs().x6_lds_min > K::LDS_BYTES ? fp_get_knobs().x6_lds_min : K::LDS_BYTES;   // lab knob, 0 in the product
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_x6d_kernel<CI, G_, CO, HW, DWIN>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL((dwblock_x6d_kernel<CI, G_, CO, HW, DWIN>), dim3(a.N * K::NBAND), dim3(256), lds_bytes, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// Shapes the split kernel is instantiated for (include/facepath.h, DWBLOCK with FP_OPF_SPLIT3).
static bool x6_stride2_shape(const fp_op& op) {
  return (op.Cin == 64 && op.Cmid == 256 && op.Cout == 128 && op.H == 28) ||
         (op.Cin == 128 && op.Cmid == 512 && op.Cout == 128 && op.H == 14) ||
         (op.Cin == 64 && op.Cmid == 128 && op.Cout == 64 && op.H == 56);
}

bool fp_dwblock_x6_supported(const fp_op& op) {
  if (op.kind != FP_OP_DWBLOCK || !(op.flags & FP_OPF_SPLIT3)) return false;
  if ((op.stride != 1 && op.stride != 2) || op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.H != op.W || op.OH != op.H / op.stride || op.OW != op.W / op.stride || op.out_cmul != 1) return false;
  if (op.stride == 1) {
    const bool shape = (op.Cin == 128 && (op.H == 14 || op.H == 7)) || (op.Cin == 64 && op.H == 28);
    if (!shape || op.Cout != op.Cin || op.Cmid != 2 * op.Cin) return false;
  } else if (!x6_stride2_shape(op) || op.res_mode != FP_RES_NONE) {
    return false;
  }
  const long ins = (long)op.H * op.W * op.Cin, ons = (long)op.OH * op.OW * op.Cout;
  if (op.in_ld != op.Cin || op.out_ld != op.Cout || op.in_ns != ins || op.out_ns != ons || op.in_off % 4 || op.out_off % 4) return false;
  if (op.w_off % 4 || op.scale_off % 4 || op.slope_off % 4) return false;
  if (op.res_mode != FP_RES_NONE && op.res_mode != FP_RES_ADD_AFTER_ACT) return false;
  if (op.res_mode == FP_RES_ADD_AFTER_ACT &&
      (op.res_off != op.in_off || op.res_ns != op.in_ns || op.res_ld != op.in_ld || op.res_C != op.Cin)) return false;
  if ((op.flags & ~(FP_OPF_SPLIT3 | FP_OPF_IN_DW)) || op.act2) return false;
  if ((op.flags & FP_OPF_IN_DW) && !(op.stride == 2 && op.H == 56 && op.bias_off >= 0 && op.bias_off % 4 == 0)) return false;
  if ((long)op.N * (op.H / 7) * (op.H / 7) > 0x7fffffffL) return false;
  return true;
}

// floats of the weight blob behind w_off / slope_off for a split DWBLOCK (capi.cpp bounds checks)
long fp_dwblock_x6_we_floats(const fp_op& op) { return (long)op.Cmid * op.Cin * 3 / 2; }
long fp_dwblock_x6_wp_floats(const fp_op& op) { return (long)op.Cmid * op.Cout * 3 / 2 + 2L * op.Cout; }

int fp_launch_dwblock_x6(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_dwblock_x6_supported(op)) return FP_ERR_UNSUPPORTED;
  DwbX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.we = (const unsigned short*)(weights + op.w_off);
  a.par = weights + op.scale_off;
  a.wp = (const unsigned short*)(weights + op.slope_off);
  a.paff = weights + op.slope_off + (long)op.Cmid * op.Cout * 3 / 2;
  a.dwin = (op.flags & FP_OPF_IN_DW) ? weights + op.bias_off : nullptr;
  a.N = op.N;
  a.has_res = op.res_mode == FP_RES_ADD_AFTER_ACT;
  // lab knob: 14x14 as 7x7 tiles (three workgroups per CU; measured slower than the bands: 146 against 126 us at 528 crops)
  const int quarter14 = fp_get_knobs().x6_quarter14;
  if (op.stride == 2) {
    if (op.H == 56) return (op.flags & FP_OPF_IN_DW) ? launch_x6d<64, 128, 64, 56, true>(a, s) : launch_x6d<64, 128, 64, 56>(a, s);
    return op.Cin == 64 ? launch_x6d<64, 256, 128, 28>(a, s) : launch_x6d<128, 512, 128, 14>(a, s);
  }
  if (op.Cin == 128 && op.H == 7) return launch_x6q<7>(a, s);
  // lab knob: the wave-specialised form (matrix waves / vector waves; measured 141 against 126 us at 528 crops: its D phase
  // runs 4500 cycles beside the matrix waves' MFMAs, tools/lab/x6_lab.hip)
  const int spec14 = fp_get_knobs().x6_spec14;
  if (op.Cin == 128) return quarter14 ? launch_x6q<14>(a, s) : spec14 ? launch_x6s<128, 14>(a, s) : launch_x6<128, 14>(a, s);
  return launch_x6<64, 28>(a, s);
}
