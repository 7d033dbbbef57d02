// shufdown.hip — a WHOLE stride-2 ShuffleV2Block of YOLOv5n-face in one kernel, bf16x6 split MFMAs (split.h) — gfx950.
//
// ShuffleV2Block.forward with stride 2 (fde/modules/yolov5_face/pytorch/models/common.py:127-176):
//     branch1(x) = dw3x3 s2 + BN -> 1x1 + BN + SiLU
//     branch2(x) = 1x1 + BN + SiLU -> dw3x3 s2 + BN -> 1x1 + BN + SiLU
//     out        = channel_shuffle(cat(branch1, branch2), 2)            out[2c] = branch1[c], out[2c + 1] = branch2[c]
// Op by op (round 3: dwconv + conv for branch 1, conv + FP_OP_DWPW for branch 2) the first unit of the network -- 32 channels
// at 160 x 160 -> 128 at 80 x 80, 256 images -- moves 6.7 GB: x is read twice, branch 2's 1x1 output (64 channels at FULL
// resolution, 1.68 GB) is written and read back, branch 1 makes a round trip before the shuffle: 1.4-1.5 ms of a 16.4 ms
// forward.  Here a workgroup owns a 4 x 16 tile of OUTPUT pixels (persistent: 512 workgroups walk contiguous runs of tiles) and
// nothing but x (once, + the tile's halo) and the shuffled output touch HBM:
//   x      the 9 x 33 input pixels under the tile are loaded once (wave w owns the 16-pixel groups w, w + 4, ...), written as
//          fp32 into the E-image (its 32 channels are exactly x's: branch 1's depthwise input) and split into three bf16 planes
//          kept in registers as the B fragments of branch 2's first 1x1;
//   b1     branch 1: the depthwise phase below on the x copy -> D tile -> 1x1 on the matrix cores (wave w = output row w);
//   round  = 32 channels of branch 2's first 1x1:
//      E   W1^T (fragments prefetched from L2 a phase ahead) x x^T -> BN + SiLU -> E-image in LDS (fp32, zero outside the picture;
//          even and odd columns in separate runs of a row, so that the stride-2 reads below touch consecutive slots: conflict-free)
//      D   depthwise 3x3 stride 2 + BN on the VALU (16 lanes = 16 consecutive output columns of one channel quad) -> split ->
//          D tile [3][64 px][32 ch]
//      P   W2^T x D^T accumulated over the rounds
//   out    BN + SiLU of both branches, interleaved: a lane holds four consecutive channels of one pixel of EACH branch = two
//          16-byte stores.
// The fp32 parameters (taps, BN affines: 5.8 KB) sit in LDS for the whole launch; the operands of every MFMA are swapped
// (D^T = W^T A^T, FINDINGS 21), so all LDS / global epilogue accesses are 16 bytes.  64 KB of LDS, 220 VGPRs: two workgroups per CU.
// 256 images at 160 x 160: 1.40-1.50 ms (four ops) -> 0.75 ms; the kernel is bound by vector-ALU issue (per tile and wave ~900
// plain VALU instructions, 230 transcendental ones for 113 SiLUs, 160 packed FMAs, 190 MFMAs), not by memory: FINDINGS 44.
#include <string.h>

#include "split.h"

namespace {

struct ShufDownArgs {
  const float* in;
  float* out;
  const float* w;              // parameter blob (SDCfg offsets)
  int H, W, OH, OW, in_ld, out_ld, tiles_x, tiles_per_img, ntiles;
  long in_ns, out_ns;
};

template <int KS, int CB, bool LDSW = false>
struct SDCfg {
  static constexpr int CIN = 32 * KS, R = CB / 32, NCT = CB / 16;
  static constexpr int TH = 4, TW = 16;                      // output pixels of a tile
  static constexpr int ER = 2 * TH + 1, EC = 2 * TW + 1;     // input pixels under it
  static constexpr int NSLOT = ER * EC;                      // 297
  // LDSW: eight waves, ONE workgroup per CU, all three weight matrices in LDS for the whole launch (rows of 32 bf16 at an 80-byte pitch:
  // conflict-free fragment reads) instead of 48 KiB-per-wave-and-tile of fragment loads through the CU's texture path
  static constexpr int NW = LDSW ? 8 : 4;                    // waves of a workgroup: NW / TH per output row of the tile (1: all channels; 2: half each)
  static constexpr int WPT = NW / TH, DWI = 2 * TH / NW;     // waves per pixel tile; depthwise items per thread
  static constexpr int MTE = (NSLOT + 15) / 16, NOWN = (MTE + NW - 1) / NW;
  static constexpr int LDE = 36;                             // floats per E slot (odd number of 16-byte units)
  static constexpr int EB = (NSLOT + 1) * LDE;               // + one slot that swallows the fragments' padding pixels
  static constexpr int LDA = 40;                             // bf16 per D-tile pixel row (80 bytes: conflict-free fragment reads)
  static constexpr int DPL = TH * TW * LDA;
  static constexpr int PL = 11 * CIN + 2 * CB + 2 * CB + 11 * CB + 2 * CB;   // fp32 parameter rows kept in LDS
  static constexpr int WROWS = LDSW ? KS * 3 * CB + R * 3 * KS * 32 + R * 3 * CB : 0, WPITCH = 40;   // weight rows kept in LDS
  static constexpr int LDS_BYTES = EB * 4 + 3 * DPL * 2 + PL * 4 + WROWS * WPITCH * 2;
  // parameter blob, in floats (two bf16 per float in the weight planes)
  static constexpr long O_B1DW = 0;                                    // [9][CIN] taps, [CIN] BN scale, [CIN] BN bias
  static constexpr long O_B1PW = O_B1DW + 11 * CIN;                    // [KS][3 planes][CB][32 k] bf16
  static constexpr long O_B1AFF = O_B1PW + (long)KS * 3 * CB * 16;     // [CB] scale, [CB] bias
  static constexpr long O_W1 = O_B1AFF + 2 * CB;                       // [R][3][KS][32 g][32 k] bf16
  static constexpr long O_AFF1 = O_W1 + (long)R * 3 * KS * 512;        // [CB] scale, [CB] bias
  static constexpr long O_DW2 = O_AFF1 + 2 * CB;                       // [9][CB] taps, [CB] scale, [CB] bias
  static constexpr long O_W2 = O_DW2 + 11 * CB;                        // [R][3][CB co][32 g] bf16
  static constexpr long O_AFF2 = O_W2 + (long)R * 3 * CB * 16;         // [CB] scale, [CB] bias
  static constexpr long TOTAL = O_AFF2 + 2 * CB;
  static_assert(LDS_BYTES <= (LDSW ? 160 : 80) * 1024 && CB % 64 == 0 && NOWN * NW >= MTE && TH * TW * 8 == NW * 64 * DWI && WPT * TH == NW, "two workgroups per CU");
};

__device__ __forceinline__ u32x4 ldg16(const unsigned short* p) { return *(const u32x4*)p; }

template <int KS, int CB, bool LDSW>
__global__ __launch_bounds__((LDSW ? 512 : 256), (LDSW ? 1 : 2)) void shufdown_x6_kernel(ShufDownArgs p) {
  using K = SDCfg<KS, CB, LDSW>;
  static_assert(KS == 1, "the E-image's 32 channels double as the tile's copy of x");
  constexpr int CIN = K::CIN, R = K::R, NCT = K::NCT, LDE = K::LDE, LDA = K::LDA, DPL = K::DPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;                                        // [NSLOT + 1][LDE]
  unsigned short* Dl = (unsigned short*)(smem_raw + K::EB * 4);        // [3][TH * TW][LDA]
  float* Pl = (float*)(Dl + 3 * DPL);                                  // the fp32 parameters (K::PL floats), for the whole launch
  // LDS copies of the fp32 parameter rows
  constexpr int P_B1DW = 0, P_B1AFF = P_B1DW + 11 * CIN, P_AFF1 = P_B1AFF + 2 * CB, P_DW2 = P_AFF1 + 2 * CB, P_AFF2 = P_DW2 + 11 * CB;
  static_assert(P_AFF2 + 2 * CB == K::PL, "");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const unsigned short* wb = (const unsigned short*)p.w;
  unsigned short* Wl = (unsigned short*)(Pl + K::PL);                  // LDSW: [WROWS][WPITCH] bf16: b1's 1x1, then W1, then W2
  constexpr int WR_B1 = 0, WR_W1 = KS * 3 * CB, WR_W2 = WR_W1 + R * 3 * KS * 32;
  if (LDSW) {
    for (int i = tid; i < K::WROWS * 4; i += 64 * K::NW) {            // a row = 32 bf16 = four 16-byte pieces
      const int row = i >> 2, piece = i & 3;
      const unsigned short* src = row < WR_W1 ? wb + 2 * K::O_B1PW + row * 32
                                : row < WR_W2 ? wb + 2 * K::O_W1 + (row - WR_W1) * 32 : wb + 2 * K::O_W2 + (row - WR_W2) * 32;
      *(u32x4*)(Wl + row * K::WPITCH + 8 * piece) = ldg16(src + 8 * piece);
    }
  }

  for (int i = tid; i < K::PL / 4; i += 64 * K::NW) {
    const int f = 4 * i;
    const long src = f < P_B1AFF ? K::O_B1DW + f : f < P_AFF1 ? K::O_B1AFF + (f - P_B1AFF) : f < P_DW2 ? K::O_AFF1 + (f - P_AFF1)
                   : f < P_AFF2 ? K::O_DW2 + (f - P_DW2) : K::O_AFF2 + (f - P_AFF2);
    *(f32x4*)&Pl[f] = *(const f32x4*)(p.w + src);
  }
  if (LDSW) __syncthreads();                             // the weight rows are read before the tile loop's first barrier

  // this thread's depthwise items: 16 consecutive output columns of one (row, channel quad) per 16 lanes
  const int txl = tid & 15, grp = tid >> 4;
  // depthwise 3x3 stride 2 + BN over the 32 channels the E-image holds -> split -> D tile.  par: [9][C] taps, [C] scale, [C] bias
  // (LDS), c0 = first channel of the 32 within the rows of width C.
  auto dw_phase = [&](const float* par, int C, int c0) {
#pragma unroll
    for (int j = 0; j < K::DWI; ++j) {
      const int g2 = grp + 4 * K::NW * j, tyl = g2 >> 3, cq = g2 & 7;    // TH x 16 pixels x 8 channel quads over 64 NW threads
      const float* dwp = par + c0 + 4 * cq;
      f32x4 a = z;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* er = El + ((2 * tyl + ky) * K::EC + txl) * LDE + 4 * cq;
        // columns 2 txl, 2 txl + 1, 2 txl + 2 of the row = even slot txl, odd slot 17 + txl, even slot txl + 1
        a += *(const f32x4*)er * *(const f32x4*)(dwp + (ky * 3 + 0) * C);
        a += *(const f32x4*)(er + 17 * LDE) * *(const f32x4*)(dwp + (ky * 3 + 1) * C);
        a += *(const f32x4*)(er + LDE) * *(const f32x4*)(dwp + (ky * 3 + 2) * C);
      }
      const f32x4 v = a * *(const f32x4*)(dwp + 9 * C) + *(const f32x4*)(dwp + 10 * C);
      unsigned h0, m0, l0, h1, m1, l1;
      fp_split_pair(v[0], v[1], h0, m0, l0);
      fp_split_pair(v[2], v[3], h1, m1, l1);
      unsigned short* dst = Dl + (16 * tyl + txl) * LDA + 4 * cq;
      *(u32x2*)dst = u32x2{h0, h1};
      *(u32x2*)(dst + DPL) = u32x2{m0, m1};
      *(u32x2*)(dst + 2 * DPL) = u32x2{l0, l1};
    }
  };
  // (wofs = l15 * 32 + 8 q, re-materialised per tile: the weight addresses are invariant across the tile loop, and hipcc would hoist
  // all 240 registers of fragments out of it and spill them -- FINDINGS 20)
  int wofs = l15 * 32 + 8 * q;
  constexpr int NCW = NCT / K::WPT;                       // channel tiles of a wave
  const int ptile = wave % K::TH, ct0 = (wave / K::TH) * NCW;
  const int lofs = l15 * K::WPITCH + 8 * q;               // LDSW: the same fragment in the LDS copy (80-byte rows)
  auto load_w = [&](const unsigned short* slab, int lrow, fp_frag3 (&w)[NCW]) {     // slab: [3][CB][32] in the blob; lrow: its first LDS row
    if (LDSW) {
      const unsigned short* wp = Wl + (lrow + 16 * ct0) * K::WPITCH + lofs;
#pragma unroll
      for (int ct = 0; ct < NCW; ++ct) {
        w[ct].h = *(const u32x4*)(wp + (16 * ct) * K::WPITCH), w[ct].m = *(const u32x4*)(wp + (CB + 16 * ct) * K::WPITCH);
        w[ct].l = *(const u32x4*)(wp + (2 * CB + 16 * ct) * K::WPITCH);
      }
      return;
    }
    const unsigned short* wp = slab + wofs + (16 * ct0) * 32;
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct)
      w[ct].h = ldg16(wp + (16 * ct) * 32), w[ct].m = ldg16(wp + (CB + 16 * ct) * 32), w[ct].l = ldg16(wp + (2 * CB + 16 * ct) * 32);
  };
  // acc[ct] += W[ct0 + ct]^T x (this wave's 16 pixels of the D tile)^T
  auto pw_phase = [&](const fp_frag3 (&w)[NCW], f32x4 (&acc)[NCW]) {
    const unsigned short* src = Dl + (16 * ptile + l15) * LDA + 8 * q;
    const u32x4 dh = *(const u32x4*)src, dm = *(const u32x4*)(src + DPL), dl = *(const u32x4*)(src + 2 * DPL);
#pragma unroll
    for (int ct = 0; ct < NCW; ct += 2) fp_mfma_x6_2a(w[ct], w[ct + 1], dh, dm, dl, acc[ct], acc[ct + 1]);
  };
  // the two 16-channel tiles of round r of the first 1x1: rows 16 gt + l15 of [3][32 g][32 k]
  auto load_w1 = [&](int r, fp_frag3& wa, fp_frag3& wc) {
    if (LDSW) {
      const unsigned short* b = Wl + (WR_W1 + r * 3 * 32) * K::WPITCH + lofs;
      wa.h = *(const u32x4*)b, wa.m = *(const u32x4*)(b + 32 * K::WPITCH), wa.l = *(const u32x4*)(b + 64 * K::WPITCH);
      wc.h = *(const u32x4*)(b + 16 * K::WPITCH), wc.m = *(const u32x4*)(b + 48 * K::WPITCH), wc.l = *(const u32x4*)(b + 80 * K::WPITCH);
      return;
    }
    const unsigned short* b = wb + 2 * K::O_W1 + (long)r * 3 * 32 * 32 + wofs;
    wa.h = ldg16(b), wa.m = ldg16(b + 1024), wa.l = ldg16(b + 2048);
    wc.h = ldg16(b + 512), wc.m = ldg16(b + 1024 + 512), wc.l = ldg16(b + 2048 + 512);
  };

  // persistent: workgroup b takes a contiguous run of tiles (neighbours in time share their halo in L1 / L2)
  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(t_begin + per, p.ntiles);
  for (int tile = t_begin; tile < t_end; ++tile) {
    asm volatile("" : "+v"(wofs));
    const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
    const int ty0 = (tt / p.tiles_x) * K::TH, tx0 = (tt % p.tiles_x) * K::TW;
    const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;                     // input pixel of E slot (0, 0)
    const float* xin = p.in + (long)img * p.in_ns;

    // ---- x: the pixels under the tile, 16 per group, wave w owns groups w, w + 4, ...: fp32 copy -> E-image (branch 1's
    //      depthwise input), split -> B fragments of branch 2's first 1x1 ----
    fp_frag3 xf[K::NOWN];
    int eoff[K::NOWN];
    bool pin[K::NOWN];
    f32x4 xa[K::NOWN], xb[K::NOWN];
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const int slot = 16 * (wave + K::NW * t) + l15;
      const int r_ = slot / K::EC, c_ = slot - r_ * K::EC;
      const int iy = iy0 + r_, ix = ix0 + c_;
      const bool real = slot < K::NSLOT;
      pin[t] = real && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      // E slot: even columns first (17 of them), then the odd ones -- a stride-2 walk over the columns reads consecutive slots
      const int es = r_ * K::EC + ((c_ & 1) ? 17 + (c_ >> 1) : (c_ >> 1));
      eoff[t] = (real ? es : K::NSLOT) * LDE;
      const float* px = xin + ((long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix, 0), p.W - 1)) * p.in_ld + 8 * q;
      xa[t] = *(const f32x4*)px;
      xb[t] = *(const f32x4*)(px + 4);
    }
    fp_frag3 wb1[NCW];
    load_w(wb + 2 * K::O_B1PW, WR_B1, wb1);                       // branch 1's 1x1: lands under the depthwise phase
    __syncthreads();                                       // the previous tile is done with the E-image and the D tile (first tile: the parameters are in LDS)
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const f32x4 a = pin[t] ? xa[t] : z, b = pin[t] ? xb[t] : z;
      *(f32x4*)&El[eoff[t] + 8 * q] = a;
      *(f32x4*)&El[eoff[t] + 8 * q + 4] = b;
      xf[t] = fp_split8(a, b);
    }
    __syncthreads();

    // ---- branch 1: dw3x3 s2 + BN -> D tile -> 1x1 ----
    f32x4 acc1[NCW], acc2[NCW];
#pragma unroll
    for (int ct = 0; ct < NCW; ++ct) acc1[ct] = z, acc2[ct] = z;
    dw_phase(Pl + P_B1DW, CIN, 0);
    fp_frag3 wa, wc;
    load_w1(0, wa, wc);
    __syncthreads();                                       // D tile complete, the E-image is free
    pw_phase(wb1, acc1);

    // ---- branch 2 ----
#pragma unroll
    for (int r = 0; r < R; ++r) {
      // E: 32 channels of the first 1x1 for every pixel under the tile -> BN + SiLU -> E-image (zero outside the picture)
      {
        const float* aff = Pl + P_AFF1 + 32 * r + 4 * q;
        const f32x4 sc0 = *(const f32x4*)aff, sc1 = *(const f32x4*)(aff + 16);
        const f32x4 bi0 = *(const f32x4*)(aff + CB), bi1 = *(const f32x4*)(aff + CB + 16);
#pragma unroll
        for (int t = 0; t < K::NOWN; ++t) {
          f32x4 e0 = z, e1 = z;
          fp_mfma_x6_2a(wa, wc, xf[t].h, xf[t].m, xf[t].l, e0, e1);
          f32x4 v0 = e0 * sc0 + bi0, v1 = e1 * sc1 + bi1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = pin[t] ? fp_silu(v0[e]) : 0.f;
            v1[e] = pin[t] ? fp_silu(v1[e]) : 0.f;
          }
          *(f32x4*)&El[eoff[t] + 4 * q] = v0;
          *(f32x4*)&El[eoff[t] + 16 + 4 * q] = v1;
        }
      }
      fp_frag3 w2[NCW];
      load_w(wb + 2 * K::O_W2 + (long)r * 3 * CB * 32, WR_W2 + r * 3 * CB, w2);          // lands under the depthwise phase
      __syncthreads();                                     // E-image complete (and the previous 1x1 is done with the D tile)
      dw_phase(Pl + P_DW2, CB, 32 * r);
      if (r + 1 < R) load_w1(r + 1, wa, wc);
      __syncthreads();                                     // D tile complete; every wave is done reading the E-image
      pw_phase(w2, acc2);
    }

    // ---- epilogue: pixel (ty0 + ptile, tx0 + l15), channels 16 (ct0 + ct) + 4 q .. + 3 of both branches, interleaved ----
    const int oy = ty0 + ptile, oxe = tx0 + l15;
    if (oy < p.OH && oxe < p.OW) {
      float* o = p.out + (long)img * p.out_ns + ((long)oy * p.OW + oxe) * p.out_ld;
#pragma unroll
      for (int ct = 0; ct < NCW; ++ct) {
        const int c = 16 * (ct0 + ct) + 4 * q;
        f32x4 b1 = acc1[ct] * *(const f32x4*)&Pl[P_B1AFF + c] + *(const f32x4*)&Pl[P_B1AFF + CB + c];
        f32x4 b2 = acc2[ct] * *(const f32x4*)&Pl[P_AFF2 + c] + *(const f32x4*)&Pl[P_AFF2 + CB + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) b1[e] = fp_silu(b1[e]), b2[e] = fp_silu(b2[e]);
        *(f32x4*)(o + 2 * c) = f32x4{b1[0], b2[0], b1[1], b2[1]};
        *(f32x4*)(o + 2 * c + 4) = f32x4{b1[2], b2[2], b1[3], b2[3]};
      }
    }
  }
}

template <int KS, int CB, bool LDSW>
int launch(const ShufDownArgs& a, hipStream_t s) {
  using K = SDCfg<KS, CB, LDSW>;
  const hipError_t ae = hipFuncSetAttribute((const void*)shufdown_x6_kernel<KS, CB, LDSW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            K::LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int slots = LDSW ? 256 : 512;                     // persistent: one / two workgroups per CU, contiguous runs of tiles
  const int grid = a.ntiles < slots ? a.ntiles : slots;
  hipLaunchKernelGGL((shufdown_x6_kernel<KS, CB, LDSW>), dim3((unsigned)grid), dim3(64 * K::NW), K::LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The stride-1 ShuffleV2Block (common.py:127-176 with stride 1): x1, x2 = x.chunk(2); out = shuffle(cat(x1, branch2(x2))).
// Same three phases per round as above on an 8 x 16 tile of output pixels: the 10 x 18 pixels under it carry x2 (CB channels =
// KS = CB / 32 slabs of B fragments in registers), E-image rows hold consecutive columns (stride 1: conflict-free as they are),
// the D tile is 128 pixels (a wave owns output rows w and w + 4 for all CB channels), and the epilogue interleaves the second
// 1x1's output with x1 read straight from global memory: out[2c] = x1[c], out[2c + 1] = branch2[c].  There is no branch 1.
// Round 3 ran this as pwx6_kernel (first 1x1; its output to HBM and back) + FP_OP_DWPW: 0.60 ms per unit at 80 x 80 x 256 images.
template <int CB>
struct SUCfg {
  static constexpr int KS = CB / 32, R = CB / 32, NCT = CB / 16;
  static constexpr int TH = 8, TW = 16, NW = 4;
  static constexpr int ER = TH + 2, EC = TW + 2;
  static constexpr int NSLOT = ER * EC;                      // 180
  static constexpr int MTE = (NSLOT + 15) / 16, NOWN = (MTE + NW - 1) / NW;
  static constexpr int LDE = 36, EB = (NSLOT + 1) * LDE;
  static constexpr int LDA = 40, DPL = TH * TW * LDA;
  static constexpr int PL = 2 * CB + 11 * CB + 2 * CB;       // fp32 parameter rows kept in LDS
  static constexpr int LDS_BYTES = EB * 4 + 3 * DPL * 2 + PL * 4;
  static constexpr int DWI = TH * TW * 8 / (64 * NW);        // depthwise items per thread
  static constexpr int NPT = TH / NW;                        // pixel tiles (output rows) of a wave
  static constexpr long O_W1 = 0;                            // [R][3][KS][32 g][32 k] bf16
  static constexpr long O_AFF1 = O_W1 + (long)R * 3 * KS * 512;
  static constexpr long O_DW2 = O_AFF1 + 2 * CB;             // [9][CB] taps, [CB] scale, [CB] bias
  static constexpr long O_W2 = O_DW2 + 11 * CB;              // [R][3][CB co][32 g] bf16
  static constexpr long O_AFF2 = O_W2 + (long)R * 3 * CB * 16;
  static constexpr long TOTAL = O_AFF2 + 2 * CB;
  static_assert(LDS_BYTES <= 80 * 1024 && NOWN * NW >= MTE && NPT * NW == TH && NW == 4, "two workgroups per CU");
};

template <int CB>
__global__ __launch_bounds__(256, 2) void shufunit_x6_kernel(ShufDownArgs p) {
  using K = SUCfg<CB>;
  constexpr int KS = K::KS, R = K::R, NCT = K::NCT, LDE = K::LDE, LDA = K::LDA, DPL = K::DPL, NPT = K::NPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* El = (float*)smem_raw;                                        // [NSLOT + 1][LDE]
  unsigned short* Dl = (unsigned short*)(smem_raw + K::EB * 4);        // [3][TH * TW][LDA]
  float* Pl = (float*)(Dl + 3 * DPL);
  constexpr int P_AFF1 = 0, P_DW2 = 2 * CB, P_AFF2 = P_DW2 + 11 * CB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const unsigned short* wb = (const unsigned short*)p.w;

  for (int i = tid; i < K::PL / 4; i += 256) {
    const int f = 4 * i;
    const long src = f < P_DW2 ? K::O_AFF1 + f : f < P_AFF2 ? K::O_DW2 + (f - P_DW2) : K::O_AFF2 + (f - P_AFF2);
    *(f32x4*)&Pl[f] = *(const f32x4*)(p.w + src);
  }
  __syncthreads();
  // (barriers inside a tile: E -> [sync] -> D -> [sync] -> P.  The E phase of the next round / tile may overwrite the E-image at once --
  // every wave left the depthwise phase before the second barrier -- and the next depthwise phase writes the D tile only behind the
  // next first barrier, which every wave reaches after its P phase.)

  const int txl = tid & 15, grp = tid >> 4;
  int wofs = l15 * 32 + 8 * q;
  static_assert(R == 2, "the round loop is written out: what is prefetched where differs between the rounds");
  // raw x2 of a tile's groups (8 channels per slab and lane); out-of-picture pixels read a clamped address and are zeroed at the split
  auto load_x = [&](int tile, f32x4 (&xa)[K::NOWN][KS], f32x4 (&xb)[K::NOWN][KS]) {
    const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
    const int ty0 = (tt / p.tiles_x) * K::TH, tx0 = (tt % p.tiles_x) * K::TW;
    const float* xin = p.in + (long)img * p.in_ns;
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const int slot = 16 * (wave + K::NW * t) + l15;
      const int r_ = slot / K::EC, c_ = slot - r_ * K::EC;
      const int iy = ty0 - 1 + r_, ix = tx0 - 1 + c_;
      const float* px = xin + ((long)min(max(iy, 0), p.H - 1) * p.W + min(max(ix, 0), p.W - 1)) * p.in_ld + CB + 8 * q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xa[t][ks] = *(const f32x4*)(px + 32 * ks), xb[t][ks] = *(const f32x4*)(px + 32 * ks + 4);
    }
  };
  auto load_w1 = [&](int r, fp_frag3 (&wa)[KS], fp_frag3 (&wc)[KS]) {     // channel tiles 0 / 1 of round r: rows 16 gt + l15 of [3][KS][32 g][32 k]
    const unsigned short* b = wb + 2 * K::O_W1 + (long)r * 3 * KS * 1024 + wofs;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wa[ks].h = ldg16(b + ks * 1024), wa[ks].m = ldg16(b + (KS + ks) * 1024), wa[ks].l = ldg16(b + (2 * KS + ks) * 1024);
      wc[ks].h = ldg16(b + ks * 1024 + 512), wc[ks].m = ldg16(b + (KS + ks) * 1024 + 512), wc[ks].l = ldg16(b + (2 * KS + ks) * 1024 + 512);
    }
  };
  auto load_w2 = [&](int r, fp_frag3 (&w2)[NCT]) {
    const unsigned short* wp = wb + 2 * K::O_W2 + (long)r * 3 * CB * 32 + wofs;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
      w2[ct].h = ldg16(wp + (16 * ct) * 32), w2[ct].m = ldg16(wp + (CB + 16 * ct) * 32), w2[ct].l = ldg16(wp + (2 * CB + 16 * ct) * 32);
  };
  // depthwise 3x3 stride 1 + BN of round r's 32 channels -> split -> D tile
  auto dw_phase = [&](int r) {
#pragma unroll 1
    for (int j = 0; j < K::DWI; ++j) {
      const int g2 = grp + 16 * j, tyl = g2 >> 3, cq = g2 & 7;
      const float* dwp = Pl + P_DW2 + 32 * r + 4 * cq;
      f32x4 a = z;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const float* er = El + ((tyl + ky) * K::EC + txl) * LDE + 4 * cq;
        a += *(const f32x4*)er * *(const f32x4*)(dwp + (ky * 3 + 0) * CB);
        a += *(const f32x4*)(er + LDE) * *(const f32x4*)(dwp + (ky * 3 + 1) * CB);
        a += *(const f32x4*)(er + 2 * LDE) * *(const f32x4*)(dwp + (ky * 3 + 2) * CB);
      }
      const f32x4 v = a * *(const f32x4*)(dwp + 9 * CB) + *(const f32x4*)(dwp + 10 * CB);
      unsigned h0, m0, l0, h1, m1, l1;
      fp_split_pair(v[0], v[1], h0, m0, l0);
      fp_split_pair(v[2], v[3], h1, m1, l1);
      unsigned short* dst = Dl + (16 * tyl + txl) * LDA + 4 * cq;
      *(u32x2*)dst = u32x2{h0, h1};
      *(u32x2*)(dst + DPL) = u32x2{m0, m1};
      *(u32x2*)(dst + 2 * DPL) = u32x2{l0, l1};
    }
  };

  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(t_begin + per, p.ntiles);
  f32x4 xa[K::NOWN][KS], xb[K::NOWN][KS];                  // the NEXT tile's x2, fetched while this tile finishes
  if (t_begin < t_end) load_x(t_begin, xa, xb);
  for (int tile = t_begin; tile < t_end; ++tile) {
    asm volatile("" : "+v"(wofs));                         // (keeps the weight fragment loads inside the tile loop: FINDINGS 20)
    const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
    const int ty0 = (tt / p.tiles_x) * K::TH, tx0 = (tt % p.tiles_x) * K::TW;
    const float* xin = p.in + (long)img * p.in_ns;         // x1 = channels 0 .. CB - 1, x2 = CB .. 2 CB - 1
    fp_frag3 wa[KS], wc[KS];
    load_w1(0, wa, wc);

    // ---- x2 under the tile: 16 pixels per group, wave w owns groups w, w + 4, ... -> B fragments of the first 1x1 ----
    fp_frag3 xf[K::NOWN][KS];
    int eoff[K::NOWN];
    bool pin[K::NOWN];
#pragma unroll
    for (int t = 0; t < K::NOWN; ++t) {
      const int slot = 16 * (wave + K::NW * t) + l15;
      const int r_ = slot / K::EC, c_ = slot - r_ * K::EC;
      const int iy = ty0 - 1 + r_, ix = tx0 - 1 + c_;
      const bool real = slot < K::NSLOT;
      pin[t] = real && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      eoff[t] = (real ? slot : K::NSLOT) * LDE;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[t][ks] = fp_split8(pin[t] ? xa[t][ks] : z, pin[t] ? xb[t][ks] : z);
    }
    f32x4 acc[NPT][NCT];
#pragma unroll
    for (int i = 0; i < NPT; ++i)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) acc[i][ct] = z;
    fp_frag3 w2[NCT];
    f32x4 x1v[NPT][NCT];                                   // the passthrough half of this wave's output pixels (epilogue)

#pragma unroll
    for (int r = 0; r < R; ++r) {
      // E: 32 channels of the first 1x1 for every pixel under the tile -> BN + SiLU -> E-image (zero outside the picture)
      {
        const float* aff = Pl + P_AFF1 + 32 * r + 4 * q;
        const f32x4 sc0 = *(const f32x4*)aff, sc1 = *(const f32x4*)(aff + 16);
        const f32x4 bi0 = *(const f32x4*)(aff + CB), bi1 = *(const f32x4*)(aff + CB + 16);
#pragma unroll
        for (int t = 0; t < K::NOWN; ++t) {
          f32x4 e0 = z, e1 = z;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) fp_mfma_x6_2a(wa[ks], wc[ks], xf[t][ks].h, xf[t][ks].m, xf[t][ks].l, e0, e1);
          f32x4 v0 = e0 * sc0 + bi0, v1 = e1 * sc1 + bi1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v0[e] = pin[t] ? fp_silu(v0[e]) : 0.f;
            v1[e] = pin[t] ? fp_silu(v1[e]) : 0.f;
          }
          *(f32x4*)&El[eoff[t] + 4 * q] = v0;
          *(f32x4*)&El[eoff[t] + 16 + 4 * q] = v1;
        }
      }
      load_w2(r, w2);                                      // lands under the depthwise phase
      if (r == R - 1) {
        // the x fragments are dead: the registers take what the epilogue and the next tile need
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          const int oy = min(ty0 + wave + K::NW * i, p.OH - 1), ox = min(tx0 + l15, p.OW - 1);
          const float* x1 = xin + ((long)oy * p.OW + ox) * p.in_ld + 4 * q;
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct) x1v[i][ct] = *(const f32x4*)(x1 + 16 * ct);
        }
        if (tile + 1 < t_end) load_x(tile + 1, xa, xb);
      }
      __syncthreads();                                     // E-image complete (and the previous 1x1 is done with the D tile)
      dw_phase(r);
      __syncthreads();                                     // D tile complete
      // P: second 1x1 for this wave's output rows (w, w + 4), all CB channels
      {
        fp_frag3 df[NPT];
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
          const unsigned short* src = Dl + (16 * (wave + K::NW * i) + l15) * LDA + 8 * q;
          df[i].h = *(const u32x4*)src, df[i].m = *(const u32x4*)(src + DPL), df[i].l = *(const u32x4*)(src + 2 * DPL);
        }
        static_assert(NPT == 2, "");
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) fp_mfma_x6_2b(w2[ct].h, w2[ct].m, w2[ct].l, df[0], df[1], acc[0][ct], acc[1][ct]);
      }
      if (r + 1 < R) load_w1(r + 1, wa, wc);               // (before the P phase it would cost 48 registers the phase does not have)
    }

    // ---- epilogue: pixels (ty0 + w + 4 i, tx0 + l15): out[2c] = x1[c], out[2c + 1] = SiLU(BN(second 1x1))[c] ----
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      const int oy = ty0 + wave + K::NW * i, ox = tx0 + l15;
      if (oy < p.OH && ox < p.OW) {
        float* o = p.out + (long)img * p.out_ns + ((long)oy * p.OW + ox) * p.out_ld;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const int c = 16 * ct + 4 * q;
          const f32x4 a = x1v[i][ct];
          f32x4 b = acc[i][ct] * *(const f32x4*)&Pl[P_AFF2 + c] + *(const f32x4*)&Pl[P_AFF2 + CB + c];
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] = fp_silu(b[e]);
          *(f32x4*)(o + 2 * c) = f32x4{a[0], b[0], a[1], b[1]};
          *(f32x4*)(o + 2 * c + 4) = f32x4{a[2], b[2], a[3], b[3]};
        }
      }
    }
  }
}

template <int CB>
int launch_unit(const ShufDownArgs& a, hipStream_t s) {
  using K = SUCfg<CB>;
  const hipError_t ae = hipFuncSetAttribute((const void*)shufunit_x6_kernel<CB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            K::LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int grid = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL((shufunit_x6_kernel<CB>), dim3((unsigned)grid), dim3(256), K::LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // namespace

// FP_OP_SHUFDOWN: Cin = 32, Cmid = 64 (the branch width), Cout = 128, 3x3 stride 2 pad 1, even H and W, dense pixels.
bool fp_shufdown_supported(const fp_op& op) {
  if (op.kind != FP_OP_SHUFDOWN || op.flags != FP_OPF_SPLIT3) return false;
  if (op.Cin != 32 || op.Cmid != 64 || op.Cout != 128) return false;
  if (op.KH != 3 || op.KW != 3 || op.stride != 2 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.H % 2 || op.W % 2 || op.OH != op.H / 2 || op.OW != op.W / 2 || op.out_cmul != 1) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns % 4 || op.w_off % 4) return false;
  if (op.in_ld < op.Cin || op.out_ld < op.Cout || op.in_ns < (long)op.H * op.W * op.in_ld || op.out_ns < (long)op.OH * op.OW * op.out_ld) return false;
  if (op.act != FP_ACT_SILU || op.act2 != FP_ACT_SILU || op.res_mode != FP_RES_NONE) return false;
  const long tiles = (long)op.N * ((op.OH + SDCfg<1, 64>::TH - 1) / SDCfg<1, 64>::TH) * ((op.OW + 15) / 16);
  return tiles > 0 && tiles < (1L << 31);
}

long fp_shufdown_w_floats(const fp_op& op) { return op.Cin == 32 && op.Cmid == 64 ? SDCfg<1, 64>::TOTAL : 0; }

int fp_launch_shufdown(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_shufdown_supported(op)) return FP_ERR_UNSUPPORTED;
  ShufDownArgs a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.tiles_x = (op.OW + 15) / 16;
  a.tiles_per_img = a.tiles_x * ((op.OH + SDCfg<1, 64>::TH - 1) / SDCfg<1, 64>::TH);
  a.ntiles = op.N * a.tiles_per_img;
  return fp_get_knobs().shuf_ldsw ? launch<1, 64, true>(a, s) : launch<1, 64, false>(a, s);
}

// FP_OP_SHUFUNIT: the stride-1 block, Cin = Cout = 128 (x and out dense 128-channel views), Cmid = 64, 3x3 stride 1 pad 1.
bool fp_shufunit_supported(const fp_op& op) {
  if (op.kind != FP_OP_SHUFUNIT || op.flags != FP_OPF_SPLIT3) return false;
  if (op.Cin != 128 || op.Cmid != 64 || op.Cout != 128) return false;
  if (op.KH != 3 || op.KW != 3 || op.stride != 1 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.OH != op.H || op.OW != op.W || op.out_cmul != 1) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns % 4 || op.w_off % 4) return false;
  if (op.in_ld < op.Cin || op.out_ld < op.Cout || op.in_ns < (long)op.H * op.W * op.in_ld || op.out_ns < (long)op.OH * op.OW * op.out_ld) return false;
  if (op.act != FP_ACT_SILU || op.act2 != FP_ACT_SILU || op.res_mode != FP_RES_NONE) return false;
  // in place is not possible: a tile reads its neighbours' pixels (halo, x1) after they may have been written
  // (plans lay the arena out image-major: every image has the same stride and the views are regions inside it)
  const long in_img = ((long)op.H * op.W - 1) * op.in_ld + op.Cin, out_img = ((long)op.OH * op.OW - 1) * op.out_ld + op.Cout;   // exact extents of the views
  if (op.in_ns == op.out_ns) {
    if (op.in_off < op.out_off + out_img && op.out_off < op.in_off + in_img) return false;
  } else {
    const long in_hi = op.in_off + (long)(op.N - 1) * op.in_ns + in_img, out_hi = op.out_off + (long)(op.N - 1) * op.out_ns + out_img;
    if (op.in_off < out_hi && op.out_off < in_hi) return false;
  }
  const long tiles = (long)op.N * ((op.OH + 7) / 8) * ((op.OW + 15) / 16);
  return tiles > 0 && tiles < (1L << 31);
}

long fp_shufunit_w_floats(const fp_op& op) { return op.Cmid == 64 ? SUCfg<64>::TOTAL : 0; }

int fp_launch_shufunit(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_shufunit_supported(op)) return FP_ERR_UNSUPPORTED;
  ShufDownArgs a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.tiles_x = (op.OW + 15) / 16;
  a.tiles_per_img = a.tiles_x * ((op.OH + 7) / 8);
  a.ntiles = op.N * a.tiles_per_img;
  return launch_unit<64>(a, s);
}
