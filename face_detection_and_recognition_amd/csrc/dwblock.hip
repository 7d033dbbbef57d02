// dwblock.hip — a WHOLE Mobile-FaceNet Depth_Wise block in one kernel (gfx950).
//
// Depth_Wise.forward (fde/modules/mobile_facenet/mobile_facenet.py:77-88) is
//     conv (1x1 expand C -> G, BN, PReLU) -> conv_dw (3x3 depthwise, BN, PReLU) -> project (1x1 G -> C, BN) [+ x].
// Round 2 ran it as two launches (pws.hip: expand, dwpw.hip: depthwise + project) with the G-channel tensor written
// to HBM by the first and read back by the second.  Here it never leaves the CU:
//
//   tile   = 196 output pixels: a whole 14x14 image, 7 rows of a 28x28 image (+ one halo row on either side whose
//            expand values are recomputed), or two 7x7 images;
//   round  = 32 expanded channels; per round, two phases separated by workgroup barriers
//     phase 1 (matrix pipe)  P(c-1): D-tile (LDS) x project weights (registers) accumulated into the output tile that
//                            the four waves keep in registers (wave w owns output channels [w*C/4, (w+1)*C/4));
//                            E(c):   x (straight from L2 into MFMA A fragments) x expand weights (LDS / registers)
//                            -> BN + PReLU -> E-image in LDS (row-padded: one zero pixel after every row, zero rows
//                            between / around images, so the depthwise taps need no bounds checks);
//     phase 2 (VALU)         D(c):   3x3 depthwise + BN + PReLU from the E-image -> D-tile [208][32+4] in LDS;
//                            the next round's weights are staged meanwhile.
//   Two workgroups per CU (<= 80 KiB LDS, <= 256 VGPRs each): while one is in its VALU phase or waits at a barrier
//   the other one's MFMAs fill the SIMD's matrix pipe — the phases of ONE workgroup never overlap, and do not need to.
//   v_mfma_f32_16x16x4_f32: 196 pixels are 12.25 tiles of 16 rows (13, 6 % padding) but 6.1 of 32 (7, 14 %).
//
// Traffic per tile: x once from HBM (re-read per round from L2), y once; weights from L2.  The expanded tensor
// (2 x G/C times the size of x) is neither written nor read: SURVEY 8(d)'s op-granular model counts it four times.
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
  unsigned long long* stamps;   // lab builds only (tools/lab/dwblock_lab.hip): s_memtime per phase of every round
#endif
};

// In-kernel phase stamps, compiled in by the lab harness only: [block < 8][wave][round < 8][6]
#ifdef FP_DWB_STAMPS
#define DWB_STAMP(k)                                                                                        \
  do {                                                                                                      \
    if (p.stamps && blockIdx.x < 8 && c < 8 && (threadIdx.x & 63) == 0) {                                   \
      unsigned long long tt_;                                                                               \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                           \
      p.stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + c) * 6 + (k)] = tt_;                            \
    }                                                                                                       \
  } while (0)
#else
#define DWB_STAMP(k) do { } while (0)
#endif

template <int C, int HW, int RB, int NIMG>
struct DwbCfg {
  static_assert(NIMG == 1 || RB == HW, "several images per tile: whole images only");
  static_assert(HW % RB == 0 && (C == 64 || C == 128), "");
  static constexpr int G = 2 * C;                      // expanded channels (Depth_Wise `groups`)
  static constexpr int HALO = RB < HW ? 1 : 0;
  static constexpr int NBAND = HW / RB;
  static constexpr int IPX = HW * HW;                  // pixels of an image
  static constexpr int OPX = NIMG * RB * HW;           // output pixels of a tile
  static constexpr int ER = RB + 2 * HALO;             // expanded rows of a tile (one image)
  static constexpr int EPX = NIMG * ER * HW;           // expanded pixels of a tile
  static constexpr int MTE = (EPX + 15) / 16;
  static constexpr int MTP = (OPX + 15) / 16;
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
  static constexpr int LDS_FLOATS = EB + DB + WL + 2 * PL;
  static constexpr int NPW = C / 64;                   // 16-column tiles of the project output per wave
  static constexpr int NT = C == 128 ? 1 : 2;          // 16-column tiles of the expand chunk per E unit
  static constexpr int MSTR = NT == 1 ? 2 : 4;         // E units of a wave: m = m0 + MSTR*i
  static constexpr int NU = (MTE + MSTR - 1) / MSTR;
  static constexpr int KQ = C / 16;                    // float4 A fragments per unit
  static constexpr int DIT = MTP;                      // depthwise iterations of a lane (16 pixels x 16 channel pairs each)
};

template <int C, int HW, int RB, int NIMG>
__global__ __launch_bounds__(256, 2) void dwblock_kernel(DwBlockArgs p) {
  using K = DwbCfg<C, HW, RB, NIMG>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Eb = smem;
  float* Db = Eb + K::EB;
  float* Wl = Db + K::DB;
  float* Pl = Wl + K::WL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int img0 = (tile / K::NBAND) * NIMG;
  const int r0 = (tile % K::NBAND) * RB;
  constexpr int G = K::G, R = G / K::KCH;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  // wave-uniform bases (SGPR pairs) + 32-bit lane offsets: 64-bit per-lane addresses for ~40 load sites, hoisted out of
  // the round loop by LICM, cost more registers than the accumulators
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

  // lane ids as the round loop sees them: re-materialised (opaque to the optimiser) at the top of every round, so that
  // the per-unit slot / offset arithmetic (~45 values) is recomputed there instead of being hoisted out of the loop
  // and kept -- or spilled -- across it
  int lv = l15, qv = q, tv = tid;

  // ---- staging of a round's weights: LDS-DMA (global_load_lds_dwordx4: wave-uniform LDS base + lane*16, per-lane
  // source address), no staging registers.  Wl is [C/4][32][4] floats = C/8 pieces of 1 KiB (two k4 rows each), Pl
  // [15][32] floats = 1920 B (waves 0 and 1; lanes past the end are masked off by EXEC).
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* gbl_ptr;
  auto stage = [&](int c) {
    if (K::WE_LDS) {
#pragma unroll
      for (int jj = 0; jj < C / 8 / 4; ++jj) {
        const int piece = jj * 4 + wave;                       // k4 rows 2*piece, 2*piece + 1
        const float* src = p.we + K::KCH * 4 * c + (((2 * piece + ((tv >> 5) & 1)) * G + (tv & 31)) * 4);
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Wl + piece * 256), 16, 0, 0);
      }
    }
    if (wave < 2) {
      const int t = tv;   // waves 0 and 1: t = tid
      if (t < 15 * 8) {
        const float* src = p.par + K::KCH * c + ((t >> 3) * G + 4 * (t & 7));
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Pl + (c & 1) * K::PL + wave * 256), 16, 0, 0);
      }
    }
  };
  // expand weights of a round in registers (C = 64)
  f32x4 breg[K::NT][K::KQ];
  auto load_breg = [&](int c) {
    if (!K::WE_LDS) {
#pragma unroll
      for (int n = 0; n < K::NT; ++n)
#pragma unroll
        for (int j = 0; j < K::KQ; ++j)
          breg[n][j] = *(const f32x4*)(p.we + K::KCH * 4 * c + ((q * G + l15) * 4 + (4 * j * G + n * 16) * 4));
    }
  };
  // project weights of a round: rows k = 32c + 16jj + 4q + i, this wave's columns
  f32x4 pbw[K::NPW][2];
  auto load_pbw = [&](int c) {
#pragma unroll
    for (int n = 0; n < K::NPW; ++n)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
        pbw[n][jj] = *(const f32x4*)(p.wp + 8 * C * 4 * c + ((q * C + wave * K::NPW * 16 + l15) * 4 + (4 * jj * C + n * 16) * 4));
  };

  // ---- prologue: zero the E-image (pads stay zero for the whole tile), stage round 0 ----
  stage(0);
  load_breg(0);
  for (int i = tid; i < K::EB / 4; i += 256) *(f32x4*)&Eb[i * 4] = z;

  // A fragments of an E unit: rows = 16 E-pixels, k = 16j + 4q + i
  const int em0 = K::NT == 1 ? (wave >> 1) : wave;
  const int en0 = K::NT == 1 ? (wave & 1) : 0;
  f32x4 afr[2][K::KQ];
  auto load_a = [&](int m, int buf) {
    const int e = min(16 * m + lv, K::EPX - 1);
    int ii, er, ec;
    e_decode(e, ii, er, ec);
    const int gr = min(max(r0 - K::HALO + er, 0), HW - 1);
    const int aoff = (min(ii, nimg - 1) * K::IPX + gr * HW + ec) * C + 4 * qv;
#pragma unroll
    for (int j = 0; j < K::KQ; ++j) afr[buf][j] = *(const f32x4*)(xin + (aoff + 16 * j));
  };

  // output tile of this wave: [MTP][NPW] 16x16 accumulators
  f32x4 pacc[K::MTP][K::NPW];
#pragma unroll
  for (int m = 0; m < K::MTP; ++m)
#pragma unroll
    for (int n = 0; n < K::NPW; ++n) pacc[m][n] = z;

  // P: D-tile (LDS) x pbw, two 16-row tiles at a time so that consecutive MFMAs never share an accumulator
  auto project = [&]() {
#pragma unroll
    for (int m = 0; m < K::MTP; m += 2) {
      f32x4 af[2][2];
#pragma unroll
      for (int mm = 0; mm < 2; ++mm)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          af[mm][jj] = *(const f32x4*)&Db[(16 * min(m + mm, K::MTP - 1) + l15) * K::LDD + 16 * jj + 4 * q];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int n = 0; n < K::NPW; ++n) {
              if (m + mm < K::MTP) {
                pacc[m + mm][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mm][jj][i], pbw[n][jj][i], pacc[m + mm][n], 0, 0, 0);
                FP_MFMA_ORDER();
              }
            }
      FP_SCHED_FENCE();   // hipcc otherwise hoists the LDS reads of every unrolled pair to the top (and spills)
    }
  };

  __syncthreads();

  for (int c = 0; c < R; ++c) {
    const float* Pc = Pl + (c & 1) * K::PL;
    asm volatile("" : "+v"(lv), "+v"(qv), "+v"(tv));
    DWB_STAMP(0);
    // ================= phase 1: matrix pipe =================
    if (em0 < K::MTE) load_a(em0, 0);
    FP_SCHED_FENCE();
    if (c > 0) project();
    DWB_STAMP(1);
    {
      float es[K::NT], eb[K::NT], esl[K::NT];
#pragma unroll
      for (int n = 0; n < K::NT; ++n) {
        const int ch = (en0 + n) * 16 + l15;
        es[n] = Pc[ch];
        eb[n] = Pc[K::KCH + ch];
        esl[n] = Pc[2 * K::KCH + ch];
      }
#pragma unroll
      for (int u = 0; u < K::NU; ++u) {
        const int m = em0 + K::MSTR * u;
        if (m < K::MTE) {
          if (u + 1 < K::NU && m + K::MSTR < K::MTE) load_a(m + K::MSTR, (u + 1) & 1);
          f32x4 acc[2] = {z, z};
#pragma unroll
          for (int j = 0; j < K::KQ; ++j) {
            if (K::NT == 1) {
              const f32x4 b = *(const f32x4*)&Wl[((4 * j + q) * K::KCH + en0 * 16 + l15) * 4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                acc[j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[u & 1][j][i], b[i], acc[j & 1], 0, 0, 0);
                FP_MFMA_ORDER();
              }
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                  acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[u & 1][j][i], breg[n % K::NT][j][i], acc[n], 0, 0, 0);
                  FP_MFMA_ORDER();
                }
            }
          }
          if (K::NT == 1) acc[0] += acc[1];
          // BN + PReLU -> E-image; rows outside the image are zeros (the depthwise pads the EXPANDED tensor)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int e = 16 * m + 4 * qv + r;
            int ii, er, ec;
            e_decode(min(e, K::EPX - 1), ii, er, ec);
            const int vrow = NIMG > 1 ? ii * (HW + 1) + er : er;
            int slot = (vrow + 1) * K::ROWP + ec + 1;
            slot = e < K::EPX ? slot : K::NSLOT;
            const bool valid = K::HALO == 0 || (unsigned)(r0 - 1 + er) < (unsigned)HW;
#pragma unroll
            for (int n = 0; n < K::NT; ++n) {
              float v = acc[n][r] * es[n] + eb[n];
              v = v > 0.f ? v : v * esl[n];
              Eb[slot * K::KCH + (en0 + n) * 16 + lv] = valid ? v : 0.f;
            }
          }
        }
        FP_SCHED_FENCE();
      }
    }
    DWB_STAMP(2);
    __syncthreads();
    DWB_STAMP(3);

    // ================= phase 2: VALU =================
    const bool more = c + 1 < R;
    if (more) {
      stage(c + 1);   // Wl: E(c) has finished with it (barrier above); Pl: the other buffer; drained by the barrier below
      load_breg(c + 1);
    }
    load_pbw(c);
    {
      // lane = 2 channels (c2) of pixel it*16 + pxo: with 4 channels per lane the taps alone are 48 registers, and this
      // phase also carries the whole output tile (104) and the next round's project weights
      const int c2 = tv & 15, pxo = tv >> 4;
      f32x2 tap[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) tap[t] = *(const f32x2*)&Pc[(3 + t) * K::KCH + 2 * c2];
      const f32x2 dsc = *(const f32x2*)&Pc[12 * K::KCH + 2 * c2];
      const f32x2 dbi = *(const f32x2*)&Pc[13 * K::KCH + 2 * c2];
      const f32x2 dsl = *(const f32x2*)&Pc[14 * K::KCH + 2 * c2];
#pragma unroll
      for (int it = 0; it < K::DIT; ++it) {
        const int o = it * 16 + pxo;
        const int oc = min(o, K::OPX - 1);
        const int ii = NIMG > 1 ? oc / K::IPX : 0;
        const int rem = oc - ii * K::IPX;
        const int orow = rem / HW, ocol = rem - orow * HW;
        const int vrow = (NIMG > 1 ? ii * (HW + 1) : 0) + orow + K::HALO;
        const float* ctr = &Eb[((vrow + 1) * K::ROWP + ocol + 1) * K::KCH + 2 * c2];
        f32x2 s = {0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            s += *(const f32x2*)(ctr + ((dy - 1) * K::ROWP + (dx - 1)) * K::KCH) * tap[dy * 3 + dx];
        f32x2 v = s * dsc + dbi;
#pragma unroll
        for (int e = 0; e < 2; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * dsl[e];
        *(f32x2*)&Db[o * K::LDD + 2 * c2] = v;
        FP_SCHED_FENCE();
      }
    }
    DWB_STAMP(4);
    __syncthreads();
    DWB_STAMP(5);
  }
  project();

  // ---- epilogue: BN affine (+ x), straight from the accumulators: rows 4q + r of 16-row tile m, column l15 ----
  const float* pscale = p.wp + G * C;
  const float* pbias = pscale + C;
  float ps[K::NPW], pb[K::NPW];
#pragma unroll
  for (int n = 0; n < K::NPW; ++n) {
    ps[n] = pscale[(wave * K::NPW + n) * 16 + l15];
    pb[n] = pbias[(wave * K::NPW + n) * 16 + l15];
  }
  constexpr int HALF = (K::MTP + 1) / 2;
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    // every residual load of this half first, then its stores (vmcnt is one in-order counter for loads and stores)
    f32x4 rv[HALF][K::NPW];
    int off[HALF][4];   // relative to image img0, -1 = not stored
#pragma unroll
    for (int mi = 0; mi < HALF; ++mi) {
      const int m = hh * HALF + mi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * m + 4 * q + r;
        const int oc = min(o, K::OPX - 1);
        const int ii = NIMG > 1 ? oc / K::IPX : 0;
        const int rem = oc - ii * K::IPX;
        const bool ok = m < K::MTP && o < K::OPX && ii < nimg;
        const int po = (min(ii, nimg - 1) * K::IPX + r0 * HW + rem) * C + wave * K::NPW * 16 + l15;
        off[mi][r] = ok ? po : -1;
        if (p.has_res) {
#pragma unroll
          for (int n = 0; n < K::NPW; ++n) rv[mi][n][r] = m < K::MTP ? xin[po + n * 16] : 0.f;
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < HALF; ++mi) {
      const int m = hh * HALF + mi;
      if (m < K::MTP) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int n = 0; n < K::NPW; ++n) {
            float v = pacc[m][n][r] * ps[n] + pb[n];
            if (p.has_res) v += rv[mi][n][r];
            if (off[mi][r] >= 0) yout[off[mi][r] + n * 16] = v;
          }
      }
    }
  }
}

template <int C, int HW, int RB, int NIMG>
int launch_variant(const DwBlockArgs& a, hipStream_t s) {
  using K = DwbCfg<C, HW, RB, NIMG>;
  static_assert(K::LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
  constexpr int lds = K::LDS_FLOATS * 4;
  const hipError_t ae = hipFuncSetAttribute((const void*)dwblock_kernel<C, HW, RB, NIMG>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int tiles = fp_ceil_div(a.N, NIMG) * K::NBAND;
  hipLaunchKernelGGL((dwblock_kernel<C, HW, RB, NIMG>), dim3(tiles), dim3(256), lds, s, a);
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
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.we = weights + op.w_off;
  a.par = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.N = op.N;
  a.has_res = op.res_mode == FP_RES_ADD_AFTER_ACT;
  if (op.Cin == 128 && op.H == 14) return launch_variant<128, 14, 14, 1>(a, s);
  if (op.Cin == 128 && op.H == 7) return launch_variant<128, 7, 7, 2>(a, s);
  return launch_variant<64, 28, 7, 1>(a, s);
}
