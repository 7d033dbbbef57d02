// conv.hip — dense convolution as an fp32-MFMA implicit GEMM with a fused epilogue (gfx950).
//
// Replaces, for the hot path, every nn.Conv2d that is not depthwise:
//   BlazeFace stem 5x5 s2 + 1x1 convs + heads (fde/modules/blazeface/blazeface.py:29-33,118-120,155-159)
//   Mobile-FaceNet conv1 / 1x1 expand / 1x1 project / Linear (fde/modules/mobile_facenet/mobile_facenet.py:39-64,135)
//   YOLOv5-face Conv k in {1,3}, s in {1,2} (+folded or live BN, SiLU) (y5/models/common.py:39-55,127-176)
//
// GEMM view: M = N*OH*OW output pixels, Kdim = KH*KW*Cin, Ncol = Cout.  Activations are NHWC, so for a
// 1x1 conv the A panel of a block is one contiguous byte range.  A block owns 128 rows x (NB*32) columns;
// each of its 4 waves owns 32 rows and NB 32x32 accumulators (v_mfma_f32_32x32x2_f32: exact fp32 fma chain).
// Per 32-deep K chunk the block stages A (gathered im2col rows, zero padded) and the packed weights into
// LDS with 16-byte accesses; the next chunk's global loads are issued before the current chunk's MFMAs and
// written to LDS after them (register prefetch).  Fragments are read with one ds_read_b128 per 4 MFMA
// k-steps: lane (r = lane&31, h = lane>>5) holds A[r][8g+4h .. 8g+4h+3] and B[8g+4h .. +3][r], i.e. k-step t
// multiplies k = 8g+t (h=0) and 8g+4+t (h=1) — a permutation of the k order, which a sum does not care about.
// LDS row stride for A is 36 floats (144 B): 16 consecutive rows hit 16 distinct 16-B slots -> conflict free.
#include <type_traits>

#include "common.h"

namespace {

struct ConvArgs {
  const float* in;
  float* out;
  const float* res;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int N, H, W, OH, OW, Cin, Cout, KH, KW, stride, pad_t, pad_l;
  int in_ld, out_ld, res_ld, out_cmul, res_C, res_H, res_W;
  long in_ns, out_ns, res_ns;
  int act, res_mode;
  int K, Kpad, Npad, OHW;
  long M;
  int vec_epi, res_C4, ntiles_n;
};

constexpr int BM = 128;
constexpr int KC = 32;
constexpr int LDA = KC + 4;

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case FP_ACT_RELU: return v > 0.f ? v : 0.f;
    case FP_ACT_PRELU: return v > 0.f ? v : v * slope;
    case FP_ACT_SILU: return fp_silu(v);
    default: return v;
  }
}

// PWD = pointwise dense fast path (1x1, stride 1, no padding, dense NHWC in/out/residual, vector epilogue):
// row m lives at base + m*ld, so staging and epilogue need no (img, y, x) decode, no divisions and no
// per-tap bounds checks.  Measured motivation: the generic path issued ~1900 VALU instructions per wave for
// 128 MFMAs on the K=64 -> N=128 Mobile-FaceNet conv (rocprofv3 SQ_INSTS_VALU / SQ_INSTS_MFMA).
template <int NB, bool VEC, bool PWD>
__global__ __launch_bounds__(256, PWD ? (NB == 4 ? 3 : 4) : 1) void conv_igemm_kernel(ConvArgs p) {
  constexpr int BN = NB * 32;
  // one LDS array: A tile, then B tile; the vector epilogue reuses it as the output staging tile
  __shared__ __attribute__((aligned(16))) float smem[BM * LDA + (KC / 4) * BN * 4];
  float* As = smem;
  float* Bs = smem + BM * LDA;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  // 1-D grid, N tile fastest: the blocks that share an A panel are dispatched back to back, so the panel's second
  // read is an L2 / Infinity-Cache hit rather than an HBM re-fetch.
  const int ny = p.ntiles_n;
  const long m0 = (long)(blockIdx.x / ny) * BM;
  const int n0 = (int)(blockIdx.x % ny) * BN;

  // A staging: thread -> k-quad column c4 (0..7) and rows r0 + 32*i
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  long pixbase[4];  // img*in_ns, folded with validity  (PWD: element offset of the row's c4 column)
  int iy0[4], ix0[4];
  bool rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + r0 + 32 * i;
    rv[i] = m < p.M;
    if (PWD) {
      pixbase[i] = (rv[i] ? m : 0) * p.in_ld;
      iy0[i] = ix0[i] = 0;
    } else {
      const unsigned mm = rv[i] ? (unsigned)m : 0u;
      const unsigned img = mm / (unsigned)p.OHW;
      const unsigned rem = mm - img * (unsigned)p.OHW;
      const int oy = (int)(rem / (unsigned)p.OW), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
      pixbase[i] = (long)img * p.in_ns;
      iy0[i] = oy * p.stride - p.pad_t;
      ix0[i] = ox * p.stride - p.pad_l;
    }
  }

  f32x4 areg[4];
  f32x4 breg[NB];
  unsigned amask = 0xfu;   // VEC gather: which of areg[] are real taps (the others are zero padding / tails)
  unsigned bmask = 0;      // which of breg[] lie inside the packed weight block
  const int KHW = p.KH * p.KW;

  auto load_chunk = [&](int kbase) {
    if (PWD) {
      const int kc = min(kbase + c4 * 4, p.K - 4);   // unconditional loads, masked when consumed (see below)
      const bool kv = kbase + c4 * 4 < p.K;
      amask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        areg[i] = *(const f32x4*)(p.in + pixbase[i] + kc);
        if (kv && rv[i]) amask |= 1u << i;
      }
    } else if (VEC) {
      // Unconditional loads from clamped addresses; the padding / tail mask is applied in store_chunk, when the
      // registers are consumed.  (A load behind `if (ok)` is waited for before the next one is issued, and a
      // select right here would pull that wait in front of the MFMAs the loads are meant to hide under.)
      const int k4 = min(kbase + c4 * 4, p.K - 4);
      const bool kv = kbase + c4 * 4 < p.K;
      int tap = 0, c = k4;
      if (KHW > 1) {
        tap = k4 / p.Cin;
        c = k4 - tap * p.Cin;
      }
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      amask = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int iy = iy0[i] + ky, ix = ix0[i] + kx;
        const bool ok = kv && rv[i] && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        areg[i] = *(const f32x4*)(p.in + pixbase[i] + ((long)cy * p.W + cx) * p.in_ld + c);
        if (ok) amask |= 1u << i;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = kbase + c4 * 4 + e;
          if (k < p.K && rv[i]) {
            const int tap = k / p.Cin, c = k - tap * p.Cin;
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
              v[e] = p.in[pixbase[i] + ((long)iy * p.W + ix) * p.in_ld + c];
          }
        }
        areg[i] = v;
      }
    }
    const int q0 = kbase >> 2;
    bmask = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int idx = tid + 256 * j;
      const int q = idx / BN, col = idx - q * BN;
      const int gq = q0 + q, n = n0 + col;
      breg[j] = *(const f32x4*)(p.w + ((long)min(gq, (p.Kpad >> 2) - 1) * p.Npad + min(n, p.Npad - 1)) * 4);
      if (gq < (p.Kpad >> 2) && n < p.Npad) bmask |= 1u << j;
    }
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = areg[i];
      if (VEC) {
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        v = ((amask >> i) & 1u) ? v : z4;
      }
      *(f32x4*)&As[(r0 + 32 * i) * LDA + c4 * 4] = v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
      *(f32x4*)&Bs[(tid + 256 * j) * 4] = ((bmask >> j) & 1u) ? breg[j] : z4;
    }
  };

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nchunks = (p.Kpad + KC - 1) / KC;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * KC);
    const int kleft = p.Kpad - ch * KC;
    const int kqmax = kleft >= KC ? KC / 8 : kleft / 8;
    const float* arow = &As[(wave * 32 + lr) * LDA + 4 * h];
#pragma unroll
    for (int kq = 0; kq < KC / 8; ++kq) {
      if (kq < kqmax) {
        const f32x4 a = *(const f32x4*)(arow + kq * 8);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const f32x4 b = *(const f32x4*)&Bs[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
          for (int t = 0; t < 4; ++t)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[nb], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // Epilogue.  C/D map of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  if (PWD || p.vec_epi) {
    // Vector epilogue: acc*scale+bias goes through LDS (PW 32-column blocks per pass) so that every lane then
    // owns 4 consecutive channels of one pixel: residual loads, activation and the output stores are 16-byte,
    // fully coalesced accesses (256 B per row for PW = 2) instead of 4-byte stores in 128-B segments.
    // Passes: every pass stages WHOLE output rows (all NB*32 columns) of a subset of the tile's rows, so each row
    // leaves the CU as one contiguous run (512 B for NB = 4).  Writing a row as two 256-B halves in two passes
    // measured ~3.3 TB/s on this chip (the same as torch.cat with that pattern) against 5-6 TB/s for whole rows.
    // NB >= 2: 2 passes of 64 rows (waves 2p, 2p+1 stage their accumulators); NB = 1: one pass of 128 rows.
    // The staging tile [rows][NB*32 + 4] fits the A+B LDS area in every case.
    constexpr int NPASS = (NB >= 2) ? 2 : 1;
    constexpr int ROWS = BM / NPASS;
    constexpr int WAVES_PER_PASS = 4 / NPASS;
    constexpr int LDO = NB * 32 + 4;
    constexpr int F4_PER_ROW = NB * 8;
    static_assert(ROWS * LDO <= BM * LDA + (KC / 4) * BN * 4, "epilogue staging must fit in the A+B tiles");
    float sc[NB], bi[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + nb * 32 + lr;
      const int nn = n < p.Cout ? n : 0;
      sc[nb] = p.scale ? p.scale[nn] : 1.f;
      bi[nb] = p.bias ? p.bias[nn] : 0.f;
    }
    const unsigned uOHW = (unsigned)p.OHW;
    const int act = p.act, res_mode = p.res_mode;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool silu = act == FP_ACT_SILU, has_res = res_mode != FP_RES_NONE, after = res_mode == FP_RES_ADD_AFTER_ACT;
    const bool shuf = res_mode == FP_RES_SHUFFLE2;   // out[2n] = res[n], out[2n+1] = act(conv)[n]
    // Store loop of one pass.  Two rules shape it (both measured, tools/lab/README.md):
    //  * vmcnt counts loads AND stores of a wave in one queue, so a global load inside the store loop (slope,
    //    residual) makes every iteration wait for all earlier stores to be acknowledged (~1 us each under load).
    //    All loads of a batch of BATCH float4s are therefore issued before its first store.
    //  * the activation is branch-free: x > 0 ? x : x*s + 0 with s = 1 (none), 0 (relu), slope (prelu); a switch
    //    on the runtime act code costs ~8 scalar branches per element.  SiLU and "has residual" select one of four
    //    straight-line copies of the loop instead.
    constexpr int NIT = ROWS * F4_PER_ROW / 256;
    static_assert(ROWS * F4_PER_ROW % 256 == 0, "whole iterations");
    constexpr int BATCH = NIT % 4 == 0 ? 4 : (NIT % 3 == 0 ? 3 : (NIT % 2 == 0 ? 2 : 1));
    static_assert(NIT % BATCH == 0, "batches tile the pass");
    constexpr bool C4_FIXED = 256 % F4_PER_ROW == 0;   // a lane keeps its channel group for all of its rows
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f};
    auto neg_slope = [&](int n) -> f32x4 {
      if (act == FP_ACT_PRELU) return n < p.Cout ? *(const f32x4*)(p.slope + n) : z4;
      return act == FP_ACT_RELU ? z4 : one4;
    };
    f32x4 sl_fixed = z4;
    if (C4_FIXED) sl_fixed = neg_slope(n0 + (tid % F4_PER_ROW) * 4);
    auto store_rows = [&](long mp, auto silu_c, auto res_c) {
      constexpr bool SILU = decltype(silu_c)::value, RES = decltype(res_c)::value;
      unsigned img_b = 0, pix_b = 0;
      if (!PWD) {
        img_b = (unsigned)mp / uOHW;
        pix_b = (unsigned)mp - img_b * uOHW;
      }
#pragma unroll
      for (int jb = 0; jb < NIT; jb += BATCH) {
        f32x4 rr[BATCH], sl[BATCH];
        long ooff[BATCH];
        bool ok[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {   // addresses + every load of the batch
          const int f = tid + 256 * (jb + j);
          const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
          const long m = mp + row;
          const int n = n0 + c4 * 4;
          ok[j] = m < p.M && n < p.Cout;
          sl[j] = C4_FIXED ? sl_fixed : neg_slope(n);
          rr[j] = z4;
          if (PWD) {
            ooff[j] = m * p.out_ld + (shuf ? 2 * n : n);
            if (RES && ok[j] && n < p.res_C4) rr[j] = *(const f32x4*)(p.res + m * p.res_ld + n);
          } else {
            unsigned img = img_b, pix = pix_b + (unsigned)row;
            if (uOHW >= (unsigned)BM) {
              if (pix >= uOHW) { pix -= uOHW; ++img; }
            } else {
              const unsigned qd = pix / uOHW;
              img += qd;
              pix -= qd * uOHW;
            }
            ooff[j] = (long)img * p.out_ns + (long)pix * p.out_ld + (shuf ? 2 * n : n);
            if (RES && ok[j] && n < p.res_C4) {
              if (res_mode == FP_RES_POOL2_BEFORE_ACT) {
                const unsigned oy = pix / (unsigned)p.OW, ox = pix - oy * (unsigned)p.OW;
                const float* r0 = p.res + (long)img * p.res_ns + ((long)(2 * oy) * p.res_W + 2 * ox) * p.res_ld + n;
                const float* r1 = r0 + (long)p.res_W * p.res_ld;
                const f32x4 a0 = *(const f32x4*)r0, a1 = *(const f32x4*)(r0 + p.res_ld);
                const f32x4 b0 = *(const f32x4*)r1, b1 = *(const f32x4*)(r1 + p.res_ld);
#pragma unroll
                for (int e = 0; e < 4; ++e) rr[j][e] = fmaxf(fmaxf(a0[e], a1[e]), fmaxf(b0[e], b1[e]));
              } else {
                rr[j] = *(const f32x4*)(p.res + (long)img * p.res_ns + (long)pix * p.res_ld + n);
              }
            }
          }
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {   // LDS -> activation -> 16-byte stores, nothing to wait for in between
          const int f = tid + 256 * (jb + j);
          const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
          if (!ok[j]) continue;
          const f32x4 v = *(const f32x4*)&smem[row * LDO + c4 * 4];
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pre = (RES && !after && !shuf) ? rr[j][e] : 0.f;
            const float x = RES ? v[e] + pre : v[e];
            float y;
            if (SILU) y = fp_silu(x);
            else y = x > 0.f ? x : __builtin_fmaf(x, sl[j][e], 0.0f);
            o[e] = (RES && after) ? y + rr[j][e] : y;
          }
          if (RES && shuf) {   // channel_shuffle(cat(res, y), 2) written as two 16-byte pieces
            const f32x4 o0 = {rr[j][0], o[0], rr[j][1], o[1]}, o1 = {rr[j][2], o[2], rr[j][3], o[3]};
            *(f32x4*)(p.out + ooff[j]) = o0;
            *(f32x4*)(p.out + ooff[j] + 4) = o1;
          } else {
            *(f32x4*)(p.out + ooff[j]) = o;
          }
        }
      }
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      if (pass) __syncthreads();  // the previous pass has been read out
      if (wave / WAVES_PER_PASS == pass) {
        const int wrow = (wave % WAVES_PER_PASS) * 32;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wrow + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            smem[row * LDO + nb * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
          }
        }
      }
      __syncthreads();
      const long mp = m0 + pass * ROWS;
      if (silu) {
        if (has_res) store_rows(mp, std::true_type{}, std::true_type{});
        else store_rows(mp, std::true_type{}, std::false_type{});
      } else {
        if (has_res) store_rows(mp, std::false_type{}, std::true_type{});
        else store_rows(mp, std::false_type{}, std::false_type{});
      }
    }
    return;
  }
  // Scalar epilogue (odd channel counts, interleaved outputs: heads, shuffle writes).
  // Per-column parameters are lane constants: fetch them once, before the store loop.
  float sc[NB], bi[NB], sl[NB];
  bool nv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + lr;
    nv[nb] = n < p.Cout;
    const int nn = nv[nb] ? n : 0;
    sc[nb] = p.scale ? p.scale[nn] : 1.f;
    bi[nb] = p.bias ? p.bias[nn] : 0.f;
    sl[nb] = (p.act == FP_ACT_PRELU) ? p.slope[nn] : 0.f;
  }
  // (img, pix) of this wave's first row with one 32-bit division; the 16 rows follow by carry.
  const unsigned mw = (unsigned)(m0 + wave * 32);
  const unsigned uOHW = (unsigned)p.OHW;
  const unsigned img_w = mw / uOHW;
  const unsigned pix_w = mw - img_w * uOHW;
  const int act = p.act, res_mode = p.res_mode;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
    const long m = m0 + wave * 32 + row;
    if (m >= p.M) continue;
    unsigned img = img_w, pix = pix_w + (unsigned)row;
    if (uOHW >= 32u) {
      if (pix >= uOHW) { pix -= uOHW; ++img; }
    } else {
      const unsigned q = pix / uOHW;
      img += q;
      pix -= q * uOHW;
    }
    float* orow = p.out + (long)img * p.out_ns + (long)pix * p.out_ld;
    const float* rrow = nullptr;
    const float* rrow2 = nullptr;
    if (res_mode == FP_RES_ADD_BEFORE_ACT || res_mode == FP_RES_ADD_AFTER_ACT) {
      rrow = p.res + (long)img * p.res_ns + (long)pix * p.res_ld;
    } else if (res_mode == FP_RES_POOL2_BEFORE_ACT) {
      const unsigned oy = pix / (unsigned)p.OW, ox = pix - oy * (unsigned)p.OW;
      rrow = p.res + (long)img * p.res_ns + ((long)(2 * oy) * p.res_W + 2 * ox) * p.res_ld;
      rrow2 = rrow + (long)p.res_W * p.res_ld;
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!nv[nb]) continue;
      const int n = n0 + nb * 32 + lr;
      float v = acc[nb][reg] * sc[nb] + bi[nb];
      float r = 0.f;
      if (rrow && n < p.res_C) {
        if (rrow2) {
          const float a0 = rrow[n], a1 = rrow[p.res_ld + n];
          const float b0 = rrow2[n], b1 = rrow2[p.res_ld + n];
          r = fmaxf(fmaxf(a0, a1), fmaxf(b0, b1));
        } else {
          r = rrow[n];
        }
      }
      if (res_mode == FP_RES_ADD_AFTER_ACT)
        v = apply_act(v, act, sl[nb]) + r;
      else
        v = apply_act(v + r, act, sl[nb]);
      orow[(long)n * p.out_cmul] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Depthwise KSxKS convolution, one thread = one output pixel x 4 channels (16-B accesses, NHWC).
// BlazeBlock convs[0] (blazeface.py:26-29), Mobile-FaceNet conv_dw / conv_6_dw (mobile_facenet.py:72-73,130),
// ShuffleV2Block.depthwise_conv (y5/models/common.py:166-167).  HBM-bound: neighbours re-read through L1/L2.
struct DwArgs {
  const float* in;
  float* out;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int N, H, W, OH, OW, C, stride, pad_t, pad_l;
  int in_ld, out_ld;
  long in_ns, out_ns;
  int act, OHW, C4;
  long total;  // M * C4
};

template <int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(DwArgs p) {
  // XCD-aware block order (blocks b, b+8, ... share an XCD): each XCD walks a contiguous range of the flattened
  // (pixel, channel-group) space, so the rows a 3x3 window shares with its vertical neighbours are hits in that
  // XCD's L2 instead of HBM re-fetches by another XCD (measured: FETCH_SIZE 2x the tensor without this).
  const int nblk = gridDim.x, q8 = nblk / 8, r8 = nblk % 8, xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
  const long idx = (long)vb * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const unsigned uidx = (unsigned)idx;  // total < 2^31 (checked by the launcher)
  const unsigned m = uidx / (unsigned)p.C4;
  const int c4 = (int)(uidx - m * (unsigned)p.C4);
  const unsigned img = m / (unsigned)p.OHW;
  const int pix = (int)(m - img * (unsigned)p.OHW);
  const int oy = pix / p.OW, ox = pix - oy * p.OW;
  const int c = c4 * 4;
  const float* ibase = p.in + (long)img * p.in_ns + c;
  const float* wbase = p.w + c;
  // Branch-free taps: every load is issued (address clamped into the map) and masked afterwards, so the
  // KS*KS activation loads and the weight loads go out back to back instead of one wait per tap.
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
#pragma unroll
  for (int ky = 0; ky < KS; ++ky) {
    const int iy = iy0 + ky;
    const bool vy = (unsigned)iy < (unsigned)p.H;
    const int iyc = min(max(iy, 0), p.H - 1);
    const float* rowp = ibase + (long)iyc * p.W * p.in_ld;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
      const int ix = ix0 + kx;
      const bool v = vy && ((unsigned)ix < (unsigned)p.W);
      const int ixc = min(max(ix, 0), p.W - 1);
      f32x4 x = *(const f32x4*)(rowp + (long)ixc * p.in_ld);
      const f32x4 wv = *(const f32x4*)(wbase + (ky * KS + kx) * p.C);
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      x = v ? x : z;
      acc += x * wv;
    }
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f}, sl = {0.f, 0.f, 0.f, 0.f};
  if (p.scale) sc = *(const f32x4*)(p.scale + c);
  if (p.bias) bi = *(const f32x4*)(p.bias + c);
  if (p.act == FP_ACT_PRELU) sl = *(const f32x4*)(p.slope + c);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = apply_act(acc[e] * sc[e] + bi[e], p.act, sl[e]);
  *(f32x4*)(p.out + (long)img * p.out_ns + (long)pix * p.out_ld + c) = o;
}

// Depthwise 3x3, 4 adjacent output columns per lane.  The generic kernel above issues 18 vector loads per output
// float4 (9 taps + 9 weights) and is bound by the L1/TA rate, not by HBM.  Here a lane owns (row, 4 output
// columns, 4 channels): the 3 x (3*S+3) input window is loaded once (sliding-window reuse between the 4 outputs)
// and the 9 weight vectors stay in registers: 6.75 (S=1) / 9 (S=2) loads per output.  Lanes run over the channel
// groups first, so every load is still a contiguous 16*C4-byte run per pixel.
template <int S>
__global__ __launch_bounds__(256) void dwconv3_row_kernel(DwArgs p, int OWG, long total_items) {
  constexpr int P = 4;
  constexpr int WIN = (P - 1) * S + 3;
  const int nblk = gridDim.x, q8 = nblk / 8, r8 = nblk % 8, xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int vb = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + kk;
  const long idx = (long)vb * 256 + threadIdx.x;
  if (idx >= total_items) return;
  unsigned u = (unsigned)idx;
  const int c4 = (int)(u % (unsigned)p.C4); u /= (unsigned)p.C4;
  const int xg = (int)(u % (unsigned)OWG); u /= (unsigned)OWG;
  const int oy = (int)(u % (unsigned)p.OH);
  const unsigned img = u / (unsigned)p.OH;
  const int c = c4 * 4;
  const int ox0 = xg * P;
  const float* ibase = p.in + (long)img * p.in_ns + c;
  f32x4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = *(const f32x4*)(p.w + t * p.C + c);
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[P] = {z, z, z, z};
  const int iy0 = oy * S - p.pad_t, ix0 = ox0 * S - p.pad_l;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = iy0 + ky;
    const bool vy = (unsigned)iy < (unsigned)p.H;
    const float* rowp = ibase + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
    f32x4 x[WIN];
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int ix = ix0 + j;
      const bool v = vy && ((unsigned)ix < (unsigned)p.W);
      const f32x4 t = *(const f32x4*)(rowp + (long)min(max(ix, 0), p.W - 1) * p.in_ld);
      x[j] = v ? t : z;
    }
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[q] += x[q * S + kx] * w[ky * 3 + kx];
  }
  f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = z, sl = z;
  if (p.scale) sc = *(const f32x4*)(p.scale + c);
  if (p.bias) bi = *(const f32x4*)(p.bias + c);
  if (p.act == FP_ACT_PRELU) sl = *(const f32x4*)(p.slope + c);
  float* obase = p.out + (long)img * p.out_ns + ((long)oy * p.OW + ox0) * p.out_ld + c;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    if (ox0 + q < p.OW) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = apply_act(acc[q][e] * sc[e] + bi[e], p.act, sl[e]);
      *(f32x4*)(obase + (long)q * p.out_ld) = o;
    }
  }
}

// Max pooling (nn.MaxPool2d: y5/models/common.py:64,186-187; padding behaves as -inf).
struct PoolArgs {
  const float* in;
  float* out;
  int N, H, W, OH, OW, C, K, stride, pad_t, pad_l, in_ld, out_ld;
  long in_ns, out_ns;
  int OHW, C4;
  long total;
};

// One lane = one output pixel x 4 channels.  All K*K taps are requested with clamped addresses before the first max
// (a load behind an `if (inside)` branch is waited for before the next one is issued: K*K serial round trips), and
// the row decode is 32-bit (total < 2^31 is checked on the host).
template <int K>
__global__ __launch_bounds__(256) void maxpool_kernel(PoolArgs p) {
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;
  if (idx >= (unsigned)p.total) return;
  const unsigned m = idx / (unsigned)p.C4, c4 = idx - m * (unsigned)p.C4;
  const unsigned img = m / (unsigned)p.OHW, pix = m - img * (unsigned)p.OHW;
  const int oy = (int)(pix / (unsigned)p.OW), ox = (int)(pix - (unsigned)oy * (unsigned)p.OW);
  const float* ibase = p.in + (long)img * p.in_ns + c4 * 4;
  const float ninf = -__builtin_huge_valf();
  const f32x4 ninf4 = {ninf, ninf, ninf, ninf};
  const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
  f32x4 x[K][K];
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int iy = iy0 + ky;
    const float* rowp = ibase + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) x[ky][kx] = *(const f32x4*)(rowp + (long)min(max(ix0 + kx, 0), p.W - 1) * p.in_ld);
  }
  f32x4 acc = ninf4;
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const bool vy = (unsigned)(iy0 + ky) < (unsigned)p.H;
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const bool v = vy && (unsigned)(ix0 + kx) < (unsigned)p.W;
      const f32x4 t = v ? x[ky][kx] : ninf4;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], t[e]);
    }
  }
  *(f32x4*)(p.out + (long)img * p.out_ns + (long)pix * p.out_ld + c4 * 4) = acc;
}

// Any other window size: the plain loop.
__global__ __launch_bounds__(256) void maxpool_generic_kernel(PoolArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int c4 = (int)(idx % p.C4);
  const long m = idx / p.C4;
  const int img = (int)(m / p.OHW);
  const int pix = (int)(m - (long)img * p.OHW);
  const int oy = pix / p.OW, ox = pix - oy * p.OW;
  const float* ibase = p.in + (long)img * p.in_ns + c4 * 4;
  const float ninf = -__builtin_huge_valf();
  f32x4 acc = {ninf, ninf, ninf, ninf};
  for (int ky = 0; ky < p.K; ++ky) {
    const int iy = oy * p.stride - p.pad_t + ky;
    if (iy < 0 || iy >= p.H) continue;
    for (int kx = 0; kx < p.K; ++kx) {
      const int ix = ox * p.stride - p.pad_l + kx;
      if (ix < 0 || ix >= p.W) continue;
      const f32x4 x = *(const f32x4*)(ibase + ((long)iy * p.W + ix) * p.in_ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], x[e]);
    }
  }
  *(f32x4*)(p.out + (long)img * p.out_ns + (long)pix * p.out_ld + c4 * 4) = acc;
}

// nn.Upsample(scale_factor=2, mode='nearest') (y5/models/yolov5n.yaml:26,31) writing into a concat slice.
__global__ __launch_bounds__(256) void upsample2x_kernel(PoolArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int c4 = (int)(idx % p.C4);
  const long m = idx / p.C4;
  const int img = (int)(m / p.OHW);
  const int pix = (int)(m - (long)img * p.OHW);
  const int oy = pix / p.OW, ox = pix - oy * p.OW;
  const f32x4 x = *(const f32x4*)(p.in + (long)img * p.in_ns + ((long)(oy >> 1) * p.W + (ox >> 1)) * p.in_ld + c4 * 4);
  *(f32x4*)(p.out + (long)img * p.out_ns + (long)pix * p.out_ld + c4 * 4) = x;
}

// Channel-slice copy with an output channel multiplier: torch.cat / chunk / channel_shuffle
// (y5/models/common.py:21-31,169-176,241-242).  Scalar because out_cmul = 2 interleaves.
struct CopyArgs {
  const float* in;
  float* out;
  int C, in_ld, out_ld, out_cmul, HW;
  long in_ns, out_ns, total;
  int W, out_rowpad;   // copy4_kernel: the output is row-padded (facepath.h FP_OPF_OUT_ROWPAD): one more pixel per row
};

__global__ __launch_bounds__(256) void copy_kernel(CopyArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int c = (int)(idx % p.C);
  const long m = idx / p.C;
  const int img = (int)(m / p.HW);
  const int pix = (int)(m - (long)img * p.HW);
  p.out[(long)img * p.out_ns + (long)pix * p.out_ld + (long)c * p.out_cmul] =
      p.in[(long)img * p.in_ns + (long)pix * p.in_ld + c];
}

// Same copy with 16-byte accesses (dense channel slices: torch.cat, common.py:241-242).
__global__ __launch_bounds__(256) void copy4_kernel(CopyArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= p.total) return;
  const int C4 = p.C >> 2;
  const int c4 = (int)(idx % C4);
  const long m = idx / C4;
  const int img = (int)(m / p.HW);
  const int pix = (int)(m - (long)img * p.HW);
  const int opix = p.out_rowpad ? pix + pix / p.W : pix;
  *(f32x4*)(p.out + (long)img * p.out_ns + (long)opix * p.out_ld + c4 * 4) =
      *(const f32x4*)(p.in + (long)img * p.in_ns + (long)pix * p.in_ld + c4 * 4);
}

// l2_norm (mobile_facenet.py:30-33): one wave per row, x / sqrt(sum x^2), no epsilon.
__global__ __launch_bounds__(256) void l2norm_kernel(const float* in, float* out, long M, int D, long in_ld, long out_ld) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int lane = threadIdx.x & 63;
  const float* x = in + row * in_ld;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += x[i] * x[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float nrm = sqrtf(s);
  for (int i = lane; i < D; i += 64) out[row * out_ld + i] = x[i] / nrm;
}

}  // namespace

static bool conv_vec_epilogue(const fp_op& op) {
  bool ve = op.out_cmul == 1 && op.Cout % 4 == 0 && op.out_ld % 4 == 0 && op.out_off % 4 == 0 && op.out_ns % 4 == 0 &&
            (op.scale_off < 0 || op.scale_off % 4 == 0) && (op.bias_off < 0 || op.bias_off % 4 == 0) &&
            (op.slope_off < 0 || op.slope_off % 4 == 0);
  if (op.res_mode != FP_RES_NONE)
    ve = ve && op.res_ld % 4 == 0 && op.res_off % 4 == 0 && op.res_ns % 4 == 0 && fp_round_up(op.res_C, 4) <= op.res_ld;
  return ve;
}

// K bound below which pointwise-dense ops take 64-wide N tiles (NB = 2, occupancy 4).  Measured equal in time to
// NB = 4 for 64 -> 128 but with the A panel fetched twice (rocprofv3 FETCH_SIZE 2x): disabled.
#ifndef FP_PWD_NB2_MAX_K
#define FP_PWD_NB2_MAX_K 0
#endif
void fp_conv_variant(const fp_op& op, int* nb, int* vec, int* pwd) {
  *vec = ((op.Cin % 4 == 0) && (op.in_ld % 4 == 0) && (op.in_off % 4 == 0) && (op.in_ns % 4 == 0)) ? 1 : 0;
  const long HWl = (long)op.H * op.W;
  bool pw = *vec && conv_vec_epilogue(op) && op.KH == 1 && op.KW == 1 && op.stride == 1 && op.pad_t == 0 &&
            op.pad_l == 0 && op.OH == op.H && op.OW == op.W && op.in_ns == HWl * op.in_ld &&
            op.out_ns == HWl * op.out_ld && op.res_mode != FP_RES_POOL2_BEFORE_ACT;
  if (op.res_mode != FP_RES_NONE) pw = pw && op.res_ns == HWl * op.res_ld;
  *pwd = pw ? 1 : 0;
  const int nblk32 = (int)fp_round_up(op.Cout, 32) / 32;
  if (pw && op.Cin <= FP_PWD_NB2_MAX_K && nblk32 % 2 == 0) { *nb = 2; return; }
  if (nblk32 % 4 == 0) *nb = 4;
  else if (nblk32 % 3 == 0) *nb = 3;
  else if (nblk32 % 2 == 0) *nb = 2;
  else if (nblk32 == 1) *nb = 1;
  else *nb = 4;  // partial last tile (guarded in-kernel)
}

int fp_launch_conv(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (op.flags & FP_OPF_OUT_DW) return fp_launch_stemdw(op, weights, arena, s);
  if (op.flags & FP_OPF_SPLIT3)
    return fp_pwx6_eligible(op) ? fp_launch_pwx6(op, weights, arena, s) : fp_launch_convx6(op, weights, arena, s);
  if (fp_pws_eligible(op)) return fp_launch_pws(op, weights, arena, s);
  if (fp_stem_eligible(op)) return fp_launch_stem(op, weights, arena, s);
  if (fp_conv3_eligible(op)) return fp_launch_conv3(op, weights, arena, s);
  ConvArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
  a.Cin = op.Cin; a.Cout = op.Cout; a.KH = op.KH; a.KW = op.KW; a.stride = op.stride;
  a.pad_t = op.pad_t; a.pad_l = op.pad_l;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.out_cmul = op.out_cmul;
  a.res_C = op.res_C; a.res_H = op.res_H; a.res_W = op.res_W;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns; a.res_ns = op.res_ns;
  a.act = op.act; a.res_mode = op.res_mode;
  a.K = op.KH * op.KW * op.Cin;
  a.Kpad = (int)fp_round_up(a.K, 8);
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.OHW = op.OH * op.OW;
  a.M = (long)op.N * a.OHW;
  if (a.act == FP_ACT_PRELU && !a.slope) return FP_ERR_INVALID_ARG;
  if (a.M >= (1L << 31)) return FP_ERR_UNSUPPORTED;  // 32-bit row decode in the kernel
  a.res_C4 = (int)fp_round_up(op.res_C, 4);
  const bool ve = conv_vec_epilogue(op);
  a.vec_epi = ve ? 1 : 0;
  if (op.res_mode == FP_RES_SHUFFLE2 && (!ve || op.res_C < op.Cout)) return FP_ERR_UNSUPPORTED;
  int NB, vec_i, pwd_i;
  fp_conv_variant(op, &NB, &vec_i, &pwd_i);
  const bool vec = vec_i != 0, pwd = pwd_i != 0;
  a.ntiles_n = fp_ceil_div(a.Npad, NB * 32);
  const long nblocks = (long)fp_ceil_div(a.M, BM) * a.ntiles_n;
  if (nblocks >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  dim3 grid((unsigned)nblocks);
  dim3 block(256);
#define FP_CONV_CASE(NBV)                                                                      \
  case NBV:                                                                                    \
    if (pwd) hipLaunchKernelGGL((conv_igemm_kernel<NBV, true, true>), grid, block, 0, s, a);   \
    else if (vec) hipLaunchKernelGGL((conv_igemm_kernel<NBV, true, false>), grid, block, 0, s, a);  \
    else hipLaunchKernelGGL((conv_igemm_kernel<NBV, false, false>), grid, block, 0, s, a);     \
    break;
  switch (NB) {
    FP_CONV_CASE(1)
    FP_CONV_CASE(2)
    FP_CONV_CASE(3)
    FP_CONV_CASE(4)
  }
#undef FP_CONV_CASE
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_dwconv(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (op.Cin % 4 || op.in_ld % 4 || op.out_ld % 4 || op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4 ||
      op.out_cmul != 1)
    return FP_ERR_ALIGNMENT;
  if (op.KH != op.KW) return FP_ERR_UNSUPPORTED;
  DwArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.C = op.Cin;
  a.stride = op.stride; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.act = op.act; a.OHW = op.OH * op.OW; a.C4 = op.Cin / 4;
  a.total = (long)op.N * a.OHW * a.C4;
  if (a.act == FP_ACT_PRELU && !a.slope) return FP_ERR_INVALID_ARG;
  if (a.total >= (1L << 31)) return FP_ERR_UNSUPPORTED;  // 32-bit item decode in the kernel
  dim3 grid((unsigned)fp_ceil_div(a.total, 256)), block(256);
  if (op.KH == 3 && (op.stride == 1 || op.stride == 2)) {
    const int OWG = (op.OW + 3) / 4;
    const long items = (long)op.N * op.OH * OWG * a.C4;
    dim3 g2((unsigned)fp_ceil_div(items, 256));
    if (op.stride == 1) hipLaunchKernelGGL((dwconv3_row_kernel<1>), g2, block, 0, s, a, OWG, items);
    else hipLaunchKernelGGL((dwconv3_row_kernel<2>), g2, block, 0, s, a, OWG, items);
    FP_CHECK_LAUNCH();
    return FP_OK;
  }
  switch (op.KH) {
    case 3: hipLaunchKernelGGL((dwconv_kernel<3>), grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL((dwconv_kernel<5>), grid, block, 0, s, a); break;
    case 7: hipLaunchKernelGGL((dwconv_kernel<7>), grid, block, 0, s, a); break;
    default: return FP_ERR_UNSUPPORTED;
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}

static int fill_pool_args(const fp_op& op, float* arena, PoolArgs& a) {
  if (op.Cin % 4 || op.in_ld % 4 || op.out_ld % 4 || op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4 ||
      op.out_cmul != 1)
    return FP_ERR_ALIGNMENT;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.C = op.Cin;
  a.K = op.KH; a.stride = op.stride; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.OHW = op.OH * op.OW; a.C4 = op.Cin / 4;
  a.total = (long)op.N * a.OHW * a.C4;
  return FP_OK;
}

int fp_launch_maxpool(const fp_op& op, float* arena, hipStream_t s) {
  PoolArgs a;
  int rc = fill_pool_args(op, arena, a);
  if (rc) return rc;
  const dim3 grid((unsigned)fp_ceil_div(a.total, 256)), block(256);
  const bool small = a.total < (1L << 31);
  if (small && a.K == 2) hipLaunchKernelGGL(maxpool_kernel<2>, grid, block, 0, s, a);
  else if (small && a.K == 3) hipLaunchKernelGGL(maxpool_kernel<3>, grid, block, 0, s, a);
  else if (small && a.K == 5) hipLaunchKernelGGL(maxpool_kernel<5>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(maxpool_generic_kernel, grid, block, 0, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_upsample2x(const fp_op& op, float* arena, hipStream_t s) {
  PoolArgs a;
  int rc = fill_pool_args(op, arena, a);
  if (rc) return rc;
  if (op.OH != 2 * op.H || op.OW != 2 * op.W) return FP_ERR_INVALID_ARG;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)fp_ceil_div(a.total, 256)), dim3(256), 0, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_copy(const fp_op& op, float* arena, hipStream_t s) {
  CopyArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.C = op.Cin; a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.out_cmul = op.out_cmul;
  a.HW = op.H * op.W; a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.W = op.W; a.out_rowpad = (op.flags & FP_OPF_OUT_ROWPAD) != 0;
  const bool v4 = op.out_cmul == 1 && op.Cin % 4 == 0 && op.in_ld % 4 == 0 && op.out_ld % 4 == 0 && op.in_off % 4 == 0 &&
                  op.out_off % 4 == 0 && op.in_ns % 4 == 0 && op.out_ns % 4 == 0;
  if (v4) {
    a.total = (long)op.N * a.HW * (a.C / 4);
    hipLaunchKernelGGL(copy4_kernel, dim3((unsigned)fp_ceil_div(a.total, 256)), dim3(256), 0, s, a);
    FP_CHECK_LAUNCH();
    return FP_OK;
  }
  if (a.out_rowpad) return FP_ERR_UNSUPPORTED;   // the scalar form has no row-padded output
  a.total = (long)op.N * a.HW * a.C;
  hipLaunchKernelGGL(copy_kernel, dim3((unsigned)fp_ceil_div(a.total, 256)), dim3(256), 0, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_launch_l2norm(const fp_op& op, float* arena, hipStream_t s) {
  const long M = (long)op.N * op.H * op.W;
  hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)fp_ceil_div(M, 4)), dim3(256), 0, s, arena + op.in_off,
                     arena + op.out_off, M, op.Cin, (long)op.in_ld, (long)op.out_ld);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
