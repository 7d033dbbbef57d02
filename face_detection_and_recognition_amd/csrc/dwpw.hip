// dwpw.hip — fused depthwise 3x3 (+BN affine, +PReLU) -> pointwise 1x1 (+BN affine) [+ residual]  (gfx950).
//
// Mobile-FaceNet's Depth_Wise block (fde/modules/mobile_facenet/mobile_facenet.py:67-88) is
//     conv (1x1 expand, BN, PReLU) -> conv_dw (3x3 depthwise stride s, BN, PReLU) -> project (1x1, BN) [+ x].
// The expanded tensor (G = 2..4x the block width) is the fat one.  This kernel fuses conv_dw + project: the
// depthwise result goes to LDS instead of HBM and is consumed there by the MFMA projection, so per block-layer the
// G-channel tensor is read ONCE (by the depthwise taps) and never written back; the op-granular model
// counts dw in+out and project in+out.
//
// One workgroup = 128 consecutive output pixels.  The expanded channels are processed in chunks of 64:
//   phase 1  all lanes: depthwise for (P-pixel group, 4-channel group) items with branch-free clamped 16-B loads and
//            a sliding window (P = 4 when OW % 4 == 0, else 2 or 1), then x*s+b and PReLU, into At[128][64+4];
//   phase 2  4 waves x 32 rows: v_mfma_f32_32x32x2_f32 against the chunk of packed projection weights in LDS,
//            accumulating over the chunks in registers;
// then the conv.hip vector epilogue (acc*s+b through LDS, 16-B residual loads and stores).
#include <type_traits>

#include "common.h"

namespace {

struct DwPwArgs {
  const float* in;
  float* out;
  const float* res;
  const float* dwp;  // [9*G dw weights][G scale][G bias][G slope]
  const float* pwp;  // [Kpad*Npad packed 1x1 weights][Cout scale][Cout bias]
  const float* oslope;  // optional [Cout4] PReLU slopes applied to the projection output (before the residual add)
  int N, H, W, OH, OW, G, Cout, stride;
  int in_ld, out_ld, res_ld;
  long in_ns;
  int Npad, OHW, has_res, has_slope;
  int out_silu, shuffle;   // per-tile kernel only: SiLU on the 1x1 output; ShuffleV2 tail (out[2n] = res[n], out[2n+1] = y[n])
  long M;
  int ntiles;
#ifdef FP_DWPW_STAMPS
  unsigned long long* stamps;   // lab builds only (tools/lab/dwpw_lab.hip): s_memtime per phase of the first units
#endif
};

// In-kernel phase stamps of the persistent kernel, compiled in by the lab harness only.
#ifdef FP_DWPW_STAMPS
#define FP_DWPW_NUNIT 24
#define FP_DWPW_NSTAMP 6
#define DWPW_STAMP(k)                                                                                  \
  do {                                                                                                 \
    if (p.stamps && blockIdx.x < 4 && unit < FP_DWPW_NUNIT && (threadIdx.x & 63) == 0) {               \
      unsigned long long tt_;                                                                          \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                      \
      p.stamps[((blockIdx.x * 4 + (threadIdx.x >> 6)) * FP_DWPW_NUNIT + unit) * FP_DWPW_NSTAMP + (k)] = tt_; \
    }                                                                                                  \
  } while (0)
#else
#define DWPW_STAMP(k) do { } while (0)
#endif

constexpr int TM = 128;
constexpr int KCH = 64;
constexpr int LDT = KCH + 4;

template <int NB, int P, int S>
__global__ __launch_bounds__(256, 2) void dwpw_kernel(DwPwArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WIN = (P - 1) * S + 3;
  constexpr int PW = (NB % 2 == 0) ? 2 : 1;
  constexpr int LDO = PW * 32 + 4;
  constexpr int F4_PER_ROW = PW * 8;
  // At [TM][LDT] | Bs [KCH/4][BN][4] | Ws [12][KCH]; the epilogue staging tile [TM][LDO] reuses At(+Bs)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;
  float* Bs = smem + TM * LDT;
  float* Ws = Bs + KCH * BN;
  static_assert(TM * LDO <= TM * LDT + KCH * BN, "epilogue staging must fit");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  int tile;
  {
    const int b = blockIdx.x, q = p.ntiles / 8, r = p.ntiles % 8, xcd = b & 7, k = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long m0 = (long)tile * TM;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  const int nchunks = p.G / KCH;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int g0 = ch * KCH;
    // stage this chunk's projection weights and depthwise parameters
    {
      // all staging loads first (unconditional, clamped), then the LDS writes: a load -> wait -> ds_write loop costs
      // one L2 round trip per 256 float4s (NB*2 + 1 of them per chunk)
      constexpr int NWB = (KCH / 4) * BN / 256;   // = NB * 2
      f32x4 wv[NWB], dv;
#pragma unroll
      for (int j = 0; j < NWB; ++j) {
        const int i = tid + 256 * j;
        const int q = i / BN, col = i - q * BN;
        wv[j] = *(const f32x4*)(p.pwp + ((long)(g0 / 4 + q) * p.Npad + min(col, p.Npad - 1)) * 4);
      }
      const int di = min(tid, 12 * (KCH / 4) - 1);
      const int drow = di / (KCH / 4), dc4 = di - drow * (KCH / 4);  // rows 0..8 taps, 9 scale, 10 bias, 11 slope
      const bool dslope = drow == 11 && !p.has_slope;                // no PReLU: the slope row is not in the blob
      dv = *(const f32x4*)(p.dwp + (long)(dslope ? 10 : drow) * p.G + g0 + dc4 * 4);
#pragma unroll
      for (int j = 0; j < NWB; ++j) {
        const int i = tid + 256 * j;
        const int col = i % BN;
        *(f32x4*)&Bs[i * 4] = col < p.Npad ? wv[j] : z;
      }
      if (tid < 12 * (KCH / 4)) *(f32x4*)&Ws[tid * 4] = dslope ? z : dv;
    }
    __syncthreads();

    // phase 1: depthwise + affine + PReLU for this channel chunk
    for (int it = tid; it < (TM / P) * (KCH / 4); it += 256) {
      const int g = it >> 4, c4 = it & 15;   // KCH/4 == 16 channel groups per pixel group
      const int r = g * P;
      long m = m0 + r;
      m = m < p.M ? m : p.M - P;  // tail groups recompute the last pixels; they are never stored
      const unsigned mm = (unsigned)m;
      const unsigned img = mm / (unsigned)p.OHW;
      const unsigned rem = mm - img * (unsigned)p.OHW;
      const int oy = (int)(rem / (unsigned)p.OW), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
      const int c = c4 * 4;
      const float* ib = p.in + (long)img * p.in_ns + g0 + c;
      f32x4 a[P];
#pragma unroll
      for (int q = 0; q < P; ++q) a[q] = z;
      const int iy0 = oy * S - 1, ix0 = ox * S - 1;
      // all 3 x WIN window loads are issued before the first FMA (one memory round trip per item, not three)
      f32x4 x[3][WIN];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = iy0 + ky;
        const bool vy = (unsigned)iy < (unsigned)p.H;
        const float* rowp = ib + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
          const int ix = ix0 + j;
          const bool v = vy && ((unsigned)ix < (unsigned)p.W);
          const f32x4 t = *(const f32x4*)(rowp + (long)min(max(ix, 0), p.W - 1) * p.in_ld);
          x[ky][j] = v ? t : z;
        }
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * KCH + c];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * KCH + c];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * KCH + c];
#pragma unroll
        for (int q = 0; q < P; ++q) {   // three statements: each contracts to one packed FMA on the accumulator
          a[q] += x[ky][q * S] * w0;
          a[q] += x[ky][q * S + 1] * w1;
          a[q] += x[ky][q * S + 2] * w2;
        }
      }
      const f32x4 sc = *(const f32x4*)&Ws[9 * KCH + c];
      const f32x4 bi = *(const f32x4*)&Ws[10 * KCH + c];
      const f32x4 sl = *(const f32x4*)&Ws[11 * KCH + c];
#pragma unroll
      for (int q = 0; q < P; ++q) {
        f32x4 v = a[q] * sc + bi;
        if (p.has_slope) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * sl[e];
        }
        *(f32x4*)&At[(r + q) * LDT + c] = v;
      }
    }
    __syncthreads();

    // phase 2: projection MFMAs for this chunk
    const float* arow = &At[(wave * 32 + lr) * LDT + 4 * h];
#pragma unroll
    for (int kq = 0; kq < KCH / 8; ++kq) {
      const f32x4 av = *(const f32x4*)(arow + kq * 8);
      f32x4 bv[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const f32x4*)&Bs[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
      for (int t = 0; t < 4; ++t)   // round-robin over the accumulators (common.h FP_MFMA_ORDER)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nb][t], acc[nb], 0, 0, 0);
          FP_MFMA_ORDER();
        }
    }
    __syncthreads();  // At / Bs / Ws are rewritten by the next chunk (or by the epilogue)
  }

  // epilogue: acc*scale+bias through LDS, then 16-B residual loads and stores (rows are dense: out + m*out_ld)
  const float* pscale = p.pwp + (long)(p.G) * p.Npad;   // Kpad == G (multiple of 64)
  const float* pbias = pscale + ((p.Cout + 3) & ~3);
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = pscale[nn];
    bi[nb] = pbias[nn];
  }
#pragma unroll
  for (int pass = 0; pass < NB / PW; ++pass) {
    if (pass) __syncthreads();
#pragma unroll
    for (int q = 0; q < PW; ++q) {
      const int nb = pass * PW + q;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        smem[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
      }
    }
    __syncthreads();
    const int ncol0 = pass * PW * 32;
    constexpr int NEP = TM * F4_PER_ROW / 256;   // float4s per lane and pass
    f32x4 rr[NEP];
    bool ok[NEP];
#pragma unroll
    for (int j = 0; j < NEP; ++j) {              // every residual load of the pass first: vmcnt counts loads and stores
      const int f = tid + 256 * j;               // together, a load inside the store loop waits for the stores before it
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      const long m = m0 + row;
      const int n = ncol0 + c4 * 4;
      ok[j] = m < p.M && n < p.Cout;
      rr[j] = z;
      if (ok[j] && (p.has_res || p.shuffle)) rr[j] = *(const f32x4*)(p.res + m * p.res_ld + n);
    }
#pragma unroll
    for (int j = 0; j < NEP; ++j) {
      const int f = tid + 256 * j;
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      const long m = m0 + row;
      const int n = ncol0 + c4 * 4;
      if (!ok[j]) continue;
      f32x4 v = *(const f32x4*)&smem[row * LDO + c4 * 4];
      if (p.oslope) {
        const f32x4 os = *(const f32x4*)(p.oslope + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * os[e];
      }
      if (p.out_silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fp_silu(v[e]);
      }
      if (p.shuffle) {   // cat(res, y) + channel_shuffle(2) as two 16-byte pieces (y5/models/common.py:21-31,169-176)
        const f32x4 o0 = {rr[j][0], v[0], rr[j][1], v[1]}, o1 = {rr[j][2], v[2], rr[j][3], v[3]};
        *(f32x4*)(p.out + m * p.out_ld + 2 * n) = o0;
        *(f32x4*)(p.out + m * p.out_ld + 2 * n + 4) = o1;
        continue;
      }
      if (p.has_res) v += rr[j];
      *(f32x4*)(p.out + m * p.out_ld + n) = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Persistent, software-pipelined variant (even OH/OW, Cout = 64 or 128: every 56x56 .. 14x14 block of
// Mobile-FaceNet).  The per-tile kernel above starts each 64-channel chunk cold: the projection weights are staged
// by a load -> wait -> ds_write loop (one L2 round trip per 256 float4s), then each depthwise item waits for its own
// window, and nothing overlaps the MFMAs (rocprofv3 r01: 2.7-4.2 TB/s algorithmic, MFMA pipe < 30 % busy).
// Here a workgroup keeps the depthwise parameters of all G channels in LDS, walks tiles t = k*grid + pos, and
// treats (tile, 32-channel chunk) as the unit of a two-stage pipeline: the window and the projection-weight chunk
// of unit u+1 are loaded into registers right after unit u's depthwise values are in LDS, so their latency hides
// under unit u's MFMAs (and, at a tile boundary, under the epilogue and its stores).
// A tile is 32 patches of 2x2 output pixels (row r of the tile = patch r>>2, pixel (r>>1)&1, r&1): one lane = one
// (patch, 4-channel group) item whose (S+3)^2 window feeds all four outputs -- 16 loads for 4 outputs at stride 1
// whatever OW is (a 1x4 strip needs OW % 4 == 0 and 18 loads), and exactly one item per lane and unit.
constexpr int PKC = 32;
constexpr int PLDT = PKC + 4;

template <int NB, int S>
__global__ __launch_bounds__(256, 2) void dwpw_persist_kernel(DwPwArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WR = S + 3;   // window rows = window columns
  constexpr int PW = 2;
  constexpr int LDO = PW * 32 + 4;
  constexpr int F4_PER_ROW = PW * 8;
  constexpr int AB = TM * PLDT + PKC * BN, ST = TM * LDO;
  constexpr int WS0 = AB > ST ? AB : ST;
  static_assert(NB % 2 == 0, "instantiated for Cout 64 / 128");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* At = smem;                 // [TM][PLDT]           depthwise output of the current unit
  float* Bs = smem + TM * PLDT;     // [PKC/4][BN][4]       projection weights of the current unit
  float* Ws = smem + WS0;           // [12][G]              taps 0..8, scale, bias, slope (ones without PReLU)
  int* Mrow = (int*)(Ws + 12 * p.G);  // [TM]               output pixel index of each tile row (-1: past the end)
  float* Osl = (float*)(Mrow + TM);   // [BN]               output PReLU slopes (ones without one)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const int Gc = p.G;

  for (int i = tid; i < 3 * Gc; i += 256) {   // 12*G/4 float4s
    const int row = (i * 4) / Gc;
    f32x4 v = {1.f, 1.f, 1.f, 1.f};
    if (row < 11 || p.has_slope) v = *(const f32x4*)(p.dwp + (long)i * 4);
    *(f32x4*)&Ws[i * 4] = v;
  }

  if (tid < BN) Osl[tid] = (p.oslope && tid < p.Cout) ? p.oslope[tid] : 1.f;

  // XCD-aware position of this block inside a window of gridDim.x tiles (bijective for any grid size)
  const int NBLK = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = NBLK / 8, rr = NBLK % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }
  const int nch = Gc / PKC;
  const int c = (tid & 7) * 4;      // this lane's 4 channels inside a chunk
  const int g = tid >> 3;           // this lane's patch inside the tile
  const unsigned PWc = (unsigned)p.OW >> 1, PPI = ((unsigned)p.OH >> 1) * PWc;   // patches per row / per image
  const unsigned npatch = (unsigned)p.N * PPI;

  // prefetch side: window of this lane's patch in tile `ptile`, decoded once per tile (the chunks of a tile share it):
  // clamped row / column offsets and the in-image mask, so that a unit's loads cost one 64-bit add each
  // Round k hands tile k*NBLK + pos to this block (XCD-contiguous); the last, partial round goes to the blocks with
  // the lowest raw indices instead, which the dispatcher spreads over all XCDs and distinct CUs -- with pos, whole XCDs
  // would take the extra tile on both of their resident workgroups (8 tiles per CU against 6 on the 14x14 layers).
  const long full_rounds = p.ntiles / NBLK;
  auto tile_of = [&](long k) { return k * NBLK + (k < full_rounds ? pos : (int)blockIdx.x); };
  long pk = 0, ck = 0;
  long ptile = tile_of(0);
  int pch = 0;
  long rowoff[WR];
  int coloff[WR];
  unsigned xmask = 0;                // bit wy*WR+wx: window position inside the image (else zero padding)
  auto decode = [&](long tile) {
    unsigned pi = (unsigned)tile * 32u + (unsigned)g;
    pi = pi < npatch ? pi : npatch - 1;   // tail patches recompute the last one; they are never stored
    const unsigned img = pi / PPI, rem = pi - img * PPI;
    const unsigned py = rem / PWc, px = rem - py * PWc;
    const int iy0 = (int)(2 * py) * S - 1, ix0 = (int)(2 * px) * S - 1;
    unsigned rowv = 0, colv = 0;
#pragma unroll
    for (int w = 0; w < WR; ++w) {
      const int iy = iy0 + w, ix = ix0 + w;
      rowoff[w] = (long)img * p.in_ns + c + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
      coloff[w] = min(max(ix, 0), p.W - 1) * p.in_ld;
      if ((unsigned)iy < (unsigned)p.H) rowv |= 1u << w;
      if ((unsigned)ix < (unsigned)p.W) colv |= 1u << w;
    }
    xmask = 0;
#pragma unroll
    for (int w = 0; w < WR; ++w)
      if ((rowv >> w) & 1u) xmask |= colv << (w * WR);
  };
  f32x4 x[WR][WR];
  f32x4 wreg[NB];
  auto issue_loads = [&](int ch) {   // every load of the unit back to back: one memory round trip
    const int g0 = ch * PKC;
    const float* ib = p.in + g0;
    // raw (clamped-address) values; the padding mask is applied when the window is consumed, one unit later -- a
    // select right here would make the wave wait for the loads before it starts the MFMAs they should hide under
#pragma unroll
    for (int wy = 0; wy < WR; ++wy)
#pragma unroll
      for (int wx = 0; wx < WR; ++wx) x[wy][wx] = *(const f32x4*)(ib + rowoff[wy] + coloff[wx]);
    const float* wb = p.pwp + (long)g0 * BN;   // Npad == BN: the chunk's [PKC/4][BN][4] block is contiguous
#pragma unroll
    for (int j = 0; j < NB; ++j) wreg[j] = *(const f32x4*)(wb + (long)(tid + 256 * j) * 4);
  };

  if (ptile < p.ntiles) {
    decode(ptile);
    issue_loads(0);
  }

  // per-column projection affine (lane constants for the whole kernel)
  const float* pscale = p.pwp + (long)Gc * p.Npad;
  const float* pbias = pscale + ((p.Cout + 3) & ~3);
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = pscale[nn];
    bi[nb] = pbias[nn];
  }
  __syncthreads();   // Ws staged

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  long tile = tile_of(0);
  int ch = 0;
  int unit = -1;
  (void)unit;
  while (tile < p.ntiles) {
    ++unit;
    DWPW_STAMP(0);
    if (ch == 0 && tid < TM) {   // output pixel of every tile row, for the epilogue (visible after this unit's barrier)
      const unsigned pi = (unsigned)tile * 32u + ((unsigned)tid >> 2);
      int m = -1;
      if (pi < npatch) {
        const unsigned img = pi / PPI, rem = pi - img * PPI;
        const unsigned py = rem / PWc, px = rem - py * PWc;
        m = (int)(img * (unsigned)p.OHW + (2 * py + ((tid >> 1) & 1)) * (unsigned)p.OW + 2 * px + (tid & 1));
      }
      Mrow[tid] = m;
    }
    // phase 1: depthwise + affine + PReLU of this unit from the prefetched window -> At; weight chunk -> Bs
    {
      const int gc = ch * PKC + c;
      f32x4 a[4] = {z, z, z, z};
      // zero padding: only the window's border can fall outside the image (S = 2, even H/W: only its top / left)
#pragma unroll
      for (int wy = 0; wy < WR; ++wy)
#pragma unroll
        for (int wx = 0; wx < WR; ++wx) {
          const bool edge = wy == 0 || wx == 0 || (S == 1 && (wy == WR - 1 || wx == WR - 1));
          if (edge) x[wy][wx] = ((xmask >> (wy * WR + wx)) & 1u) ? x[wy][wx] : z;
        }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * Gc + gc];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * Gc + gc];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * Gc + gc];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int wy = (q >> 1) * S + ky, wx = (q & 1) * S;
          a[q] += x[wy][wx] * w0;   // three statements: each contracts to one packed FMA on the accumulator
          a[q] += x[wy][wx + 1] * w1;
          a[q] += x[wy][wx + 2] * w2;
        }
      }
      const f32x4 dsc = *(const f32x4*)&Ws[9 * Gc + gc];
      const f32x4 dbi = *(const f32x4*)&Ws[10 * Gc + gc];
      const f32x4 dsl = *(const f32x4*)&Ws[11 * Gc + gc];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = a[q] * dsc + dbi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * dsl[e];
        *(f32x4*)&At[(g * 4 + q) * PLDT + c] = v;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) *(f32x4*)&Bs[(tid + 256 * j) * 4] = wreg[j];
    }
    DWPW_STAMP(1);
    __syncthreads();
    DWPW_STAMP(2);

    // prefetch the next unit (same tile, next chunk -- or the first chunk of this block's next tile)
    if (++pch == nch) {
      pch = 0;
      ptile = tile_of(++pk);
      if (ptile < p.ntiles) decode(ptile);
    }
    if (ptile < p.ntiles) issue_loads(pch);
#ifdef FP_DWPW_LAB_WAIT_EARLY
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // lab: raw latency of the unit's loads, nothing overlapped
#endif
    DWPW_STAMP(3);

    // phase 2: projection MFMAs of this unit
    {
      const float* arow = &At[(wave * 32 + lr) * PLDT + 4 * h];
#pragma unroll
      for (int kq = 0; kq < PKC / 8; ++kq) {
        const f32x4 av = *(const f32x4*)(arow + kq * 8);
        f32x4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const f32x4*)&Bs[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
        // round-robin over the accumulators: a 32x32x2 f32 MFMA that accumulates into the previous one's result issues
        // at half rate (common.h FP_MFMA_ORDER)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nb][t], acc[nb], 0, 0, 0);
            FP_MFMA_ORDER();
          }
      }
    }
    DWPW_STAMP(4);
#ifdef FP_DWPW_STAMPS
    if (p.stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // lab: expose the prefetch's residual latency
#endif
    __syncthreads();   // At / Bs are rewritten by the next unit (or by the epilogue staging)
    DWPW_STAMP(5);
    if (++ch < nch) continue;

    // epilogue of the tile: acc*scale+bias through LDS (64 columns per pass), 16-B residual loads and stores
    const bool full_tile = ((unsigned)tile + 1u) * 32u <= npatch && p.Cout == BN;   // uniform
    if (full_tile) {
      // Straight-line epilogue.  vmcnt is ONE in-order counter for a wave's loads and stores, so the residual loads
      // of the second pass are issued BEFORE the first pass's stores (waiting for them then leaves the 8 stores in
      // flight: a counted vmcnt(8) instead of a drain), and no load ever follows a store it has to wait behind.
      constexpr int PASSES = NB / PW;
      constexpr int EB = NB == 4 ? 4 : 8;          // float4s per lane and group (register budget: 2 x EB float4s live)
      constexpr int SUBS = 8 / EB, NG = PASSES * SUBS;
      auto fast = [&](auto res_c, auto oact_c) {
        constexpr bool RES = decltype(res_c)::value, OACT = decltype(oact_c)::value;
        // wave-private: a wave stages its own 32 rows and reads the same rows back as 16-B pieces (LDS is in order
        // within a wave), so the passes need no workgroup barrier and the four waves drift apart
        const int row0 = wave * 32 + (lane >> 4), cc = (lane & 15) * 4;   // rows row0 + 4*j, 16 float4s per 64-col row
        f32x4 rr[EB], v[EB];
        auto load_res = [&](int grp) {   // group = (pass, sub): rows row0 + 4*(sub*EB + j), columns pass*64 + cc
          const int pass = grp / SUBS, sub = grp % SUBS;
#pragma unroll
          for (int j = 0; j < EB; ++j)
            rr[j] = *(const f32x4*)(p.res + (long)Mrow[row0 + 4 * (sub * EB + j)] * p.res_ld + pass * PW * 32 + cc);
        };
        if (RES) load_res(0);
#pragma unroll
        for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
          for (int q = 0; q < PW; ++q) {
            const int nb = pass * PW + q;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
              smem[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
              acc[nb][reg] = 0.f;
            }
          }
          f32x4 osl = {1.f, 1.f, 1.f, 1.f};
          if (OACT) osl = *(const f32x4*)&Osl[pass * PW * 32 + cc];
#pragma unroll
          for (int sub = 0; sub < SUBS; ++sub) {
            const int grp = pass * SUBS + sub;
#pragma unroll
            for (int j = 0; j < EB; ++j) {
              v[j] = *(const f32x4*)&smem[(row0 + 4 * (sub * EB + j)) * LDO + cc];
              if (OACT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * osl[e];
              }
              if (RES) v[j] += rr[j];
            }
            if (RES && grp + 1 < NG) load_res(grp + 1);   // before this group's stores (see above)
#pragma unroll
            for (int j = 0; j < EB; ++j)
              *(f32x4*)(p.out + (long)Mrow[row0 + 4 * (sub * EB + j)] * p.out_ld + pass * PW * 32 + cc) = v[j];
          }
        }
      };
      if (p.oslope) fast(std::false_type{}, std::true_type{});   // the planner never combines the two
      else if (p.has_res) fast(std::true_type{}, std::false_type{});
      else fast(std::false_type{}, std::false_type{});
      __syncthreads();   // staging read out; At / Bs free for the next tile's first unit
      ch = 0;
      tile = tile_of(++ck);
      continue;
    }
    // last (partial) tile / Cout below the padded width: bounds-checked read-out
#pragma unroll
    for (int pass = 0; pass < NB / PW; ++pass) {
      if (pass) __syncthreads();
#pragma unroll
      for (int q = 0; q < PW; ++q) {
        const int nb = pass * PW + q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
          smem[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
          acc[nb][reg] = 0.f;
        }
      }
      __syncthreads();
      const int ncol0 = pass * PW * 32;
      for (int f = tid; f < TM * F4_PER_ROW; f += 256) {
        const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
        const int n = ncol0 + c4 * 4;
        const int m = Mrow[row];
        if (m < 0 || n >= p.Cout) continue;
        f32x4 v = *(const f32x4*)&smem[row * LDO + c4 * 4];
        if (p.oslope) {
          const f32x4 os = *(const f32x4*)&Osl[n];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * os[e];
        }
        if (p.has_res) v += *(const f32x4*)(p.res + (long)m * p.res_ld + n);
        *(f32x4*)(p.out + (long)m * p.out_ld + n) = v;
      }
    }
    __syncthreads();   // staging read out; At / Bs free for the next tile's first unit
    ch = 0;
    tile = tile_of(++ck);
  }
}


// ---------------------------------------------------------------------------------------------
// Wave-private form of the fused depthwise + projection op, for the layers whose WHOLE projection matrix fits LDS
// (G * Npad * 4 <= 64 KB: every Mobile-FaceNet block on the 56 x 56 and 28 x 28 maps).
// dwpw_persist_kernel synchronises its four waves twice per (tile, chunk) unit, which keeps them -- and, in practice,
// the two workgroups of a CU -- in lock step: they run their MFMA phases together (8.6-9.3 k cycles instead of 4.4 k,
// tools/lab/dwpw_lab) and their VALU phases together (matrix pipe idle): 52 % pipe utilisation.  pws_kernel, whose
// waves own their tiles end to end and never meet at a barrier, reaches 87 %.  Same idea here: a WAVE owns a tile of 8
// patches (32 output pixels) for all chunks -- window (one (patch, 4-channel) item per lane, prefetched a unit ahead)
// -> depthwise + affine + PReLU -> its private A tile in LDS -> MFMAs against the projection weights resident in LDS
// -> (after the last chunk) epilogue staged in the same private rows -> 16-byte stores.  The only barrier is the one
// after the weights are staged; the 8 waves of a CU drift apart and fill each other's gaps.
template <int NB, int S, int NW>
__global__ __launch_bounds__(NW * 64, NW == 6 ? 3 : 2) void dwpw_wp_kernel(DwPwArgs p) {
  constexpr int BN = NB * 32;
  constexpr int WR = S + 3;   // window rows = window columns
  constexpr int WV = 32 * PLDT + 32;    // floats per wave region: A tile [32][PLDT], then Mrow[32]
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Gc = p.G;
  float* Ws = smem;                     // [12][G]     taps 0..8, scale, bias, slope (ones without PReLU)
  float* Osl = Ws + 12 * Gc;            // [BN]        output PReLU slopes (ones without one)
  float* W2s = Osl + BN;                // [G/4][BN][4] packed projection weights, all chunks
  float* Wv = W2s + Gc * BN;            // NW wave regions
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, h = lane >> 5;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  for (int i = tid; i < 3 * Gc; i += NW * 64) {   // 12*G/4 float4s
    const int row = (i * 4) / Gc;
    f32x4 v = {1.f, 1.f, 1.f, 1.f};
    if (row < 11 || p.has_slope) v = *(const f32x4*)(p.dwp + (long)i * 4);
    *(f32x4*)&Ws[i * 4] = v;
  }
  if (tid < BN) Osl[tid] = (p.oslope && tid < p.Cout) ? p.oslope[tid] : 1.f;
  for (int i = tid; i < (Gc * BN) / 4; i += NW * 64) *(f32x4*)&W2s[i * 4] = *(const f32x4*)(p.pwp + (long)i * 4);

  float* At = Wv + wave * WV;           // [32][PLDT]  this wave's depthwise output; epilogue staging
  int* Mrow = (int*)(At + 32 * PLDT);   // [32]        output pixel index of each tile row (-1: past the end)

  // wave-level virtual blocks: wave w of workgroup b is virtual block 4*pos(b) + w of NV = 4 * gridDim.x; a workgroup's
  // four waves take four consecutive 8-patch tiles (their windows overlap: L1)
  const int NV = gridDim.x * NW;
  int vb;
  {
    const int G = gridDim.x, b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    vb = ((xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k) * NW + wave;
  }
  const int nch = Gc / PKC;
  const int c = (lane & 7) * 4;     // this lane's 4 channels inside a chunk
  const int g = lane >> 3;          // this lane's patch inside the tile (0..7)
  const unsigned PWc = (unsigned)p.OW >> 1, PPI = ((unsigned)p.OH >> 1) * PWc;   // patches per row / per image
  const unsigned npatch = (unsigned)p.N * PPI;
  const long ntw = ((long)npatch + 7) / 8;   // 8-patch tiles

  long pk = 0;
  long ptile = vb;
  int pch = 0;
  unsigned rowoff[WR];   // byte offsets from p.in (the launcher checks the tensor is < 4 GiB)
  unsigned coloff[WR];
  unsigned xmask = 0;                // bit wy*WR+wx: window position inside the image (else zero padding)
  auto decode = [&](long tile) {
    unsigned pi = (unsigned)tile * 8u + (unsigned)g;
    pi = pi < npatch ? pi : npatch - 1;   // tail patches recompute the last one; they are never stored
    const unsigned img = pi / PPI, rem = pi - img * PPI;
    const unsigned py = rem / PWc, px = rem - py * PWc;
    const int iy0 = (int)(2 * py) * S - 1, ix0 = (int)(2 * px) * S - 1;
    unsigned rowv = 0, colv = 0;
#pragma unroll
    for (int w = 0; w < WR; ++w) {
      const int iy = iy0 + w, ix = ix0 + w;
      rowoff[w] = (img * (unsigned)p.in_ns + (unsigned)c + (unsigned)min(max(iy, 0), p.H - 1) * (unsigned)(p.W * p.in_ld)) * 4u;
      coloff[w] = (unsigned)(min(max(ix, 0), p.W - 1) * p.in_ld) * 4u;
      if ((unsigned)iy < (unsigned)p.H) rowv |= 1u << w;
      if ((unsigned)ix < (unsigned)p.W) colv |= 1u << w;
    }
    xmask = 0;
#pragma unroll
    for (int w = 0; w < WR; ++w)
      if ((rowv >> w) & 1u) xmask |= colv << (w * WR);
  };
  f32x4 x[WR][WR];
  auto issue_loads = [&](int ch) {   // raw (clamped-address) window; the padding mask is applied when it is consumed
    const char* ib = (const char*)(p.in + ch * PKC);
#pragma unroll
    for (int wy = 0; wy < WR; ++wy)
#pragma unroll
      for (int wx = 0; wx < WR; ++wx) x[wy][wx] = *(const f32x4*)(ib + (rowoff[wy] + coloff[wx]));
  };
  if (ptile < ntw) {
    decode(ptile);
    issue_loads(0);
  }

  // per-column projection affine (lane constants for the whole kernel)
  const float* pscale = p.pwp + (long)Gc * p.Npad;
  const float* pbias = pscale + ((p.Cout + 3) & ~3);
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = pscale[nn];
    bi[nb] = pbias[nn];
  }
  __syncthreads();   // Ws / Osl / W2s staged: the only workgroup barrier

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  long tile = vb;
  int ch = 0;
  while (tile < ntw) {
    if (ch == 0 && lane < 32) {   // output pixel of every tile row, for the epilogue
      const unsigned pi = (unsigned)tile * 8u + ((unsigned)lane >> 2);
      int m = -1;
      if (pi < npatch) {
        const unsigned img = pi / PPI, rem = pi - img * PPI;
        const unsigned py = rem / PWc, px = rem - py * PWc;
        m = (int)(img * (unsigned)p.OHW + (2 * py + ((lane >> 1) & 1)) * (unsigned)p.OW + 2 * px + (lane & 1));
      }
      Mrow[lane] = m;
    }
    // ---- depthwise + affine + PReLU of this unit from the prefetched window -> A tile ----
    {
      const int gc = ch * PKC + c;
      f32x4 a[4] = {z, z, z, z};
#pragma unroll
      for (int wy = 0; wy < WR; ++wy)
#pragma unroll
        for (int wx = 0; wx < WR; ++wx) {
          const bool edge = wy == 0 || wx == 0 || (S == 1 && (wy == WR - 1 || wx == WR - 1));
          if (edge) x[wy][wx] = ((xmask >> (wy * WR + wx)) & 1u) ? x[wy][wx] : z;
        }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * Gc + gc];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * Gc + gc];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * Gc + gc];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int wy = (q >> 1) * S + ky, wx = (q & 1) * S;
          a[q] += x[wy][wx] * w0;   // three statements: each contracts to one packed FMA on the accumulator
          a[q] += x[wy][wx + 1] * w1;
          a[q] += x[wy][wx + 2] * w2;
        }
      }
      const f32x4 dsc = *(const f32x4*)&Ws[9 * Gc + gc];
      const f32x4 dbi = *(const f32x4*)&Ws[10 * Gc + gc];
      const f32x4 dsl = *(const f32x4*)&Ws[11 * Gc + gc];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = a[q] * dsc + dbi;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * dsl[e];
        *(f32x4*)&At[(g * 4 + q) * PLDT + c] = v;
      }
    }
    // ---- next unit's window (same tile, next chunk -- or the first chunk of this wave's next tile) ----
    if (++pch == nch) {
      pch = 0;
      ptile = ++pk * NV + vb;
      if (ptile < ntw) decode(ptile);
    }
    if (ptile < ntw) issue_loads(pch);
    // ---- projection MFMAs of this unit ----
    {
      const float* arow = &At[lr * PLDT + 4 * h];
      const float* bch = &W2s[(long)ch * PKC * BN];
#pragma unroll
      for (int kq = 0; kq < PKC / 8; ++kq) {
        const f32x4 av = *(const f32x4*)(arow + kq * 8);
        f32x4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const f32x4*)&bch[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[nb][t], acc[nb], 0, 0, 0);
            FP_MFMA_ORDER();
          }
      }
    }
    if (++ch < nch) continue;
    ch = 0;
    // ---- epilogue of the tile: acc*scale+bias staged in this wave's own A rows (32 columns per pass), output PReLU or
    // residual, 16-byte stores ----
    {
      const bool full_tile = ((unsigned)tile + 1u) * 8u <= npatch && p.Cout == BN;   // uniform
      auto epi = [&](auto res_c, auto oact_c, auto full_c) {
        constexpr bool RES = decltype(res_c)::value, OACT = decltype(oact_c)::value, FULL = decltype(full_c)::value;
        const int rl0 = lane >> 3, cc = (lane & 7) * 4;   // local rows rl0 + 8*j, 8 float4s per 32-column row
        f32x4 rr[4], v[4];
        auto load_res = [&](int nb) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = Mrow[rl0 + 8 * j], n = nb * 32 + cc;
            const bool ok = FULL || (m >= 0 && n < p.Cout);
            rr[j] = ok ? *(const f32x4*)(p.res + (long)m * p.res_ld + n) : z;
          }
        };
        if (RES) load_res(0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            At[row * PLDT + lr] = acc[nb][reg] * sc[nb] + bi[nb];
            acc[nb][reg] = 0.f;
          }
          f32x4 osl = {1.f, 1.f, 1.f, 1.f};
          if (OACT) osl = *(const f32x4*)&Osl[nb * 32 + cc];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = *(const f32x4*)&At[(rl0 + 8 * j) * PLDT + cc];
            if (OACT) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : v[j][e] * osl[e];
            }
            if (RES) v[j] += rr[j];
          }
          if (RES && nb + 1 < NB) load_res(nb + 1);   // before this pass's stores: vmcnt is one in-order queue
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int m = Mrow[rl0 + 8 * j], n = nb * 32 + cc;
            if (FULL || (m >= 0 && n < p.Cout)) *(f32x4*)(p.out + (long)m * p.out_ld + n) = v[j];
          }
        }
      };
      if (full_tile) {
        if (p.oslope) epi(std::false_type{}, std::true_type{}, std::true_type{});   // the planner never combines the two
        else if (p.has_res) epi(std::true_type{}, std::false_type{}, std::true_type{});
        else epi(std::false_type{}, std::false_type{}, std::true_type{});
      } else {
        if (p.oslope) epi(std::false_type{}, std::true_type{}, std::false_type{});
        else if (p.has_res) epi(std::true_type{}, std::false_type{}, std::false_type{});
        else epi(std::false_type{}, std::false_type{}, std::false_type{});
      }
    }
    tile += NV;
  }
}

}  // namespace

static size_t dwpw_persist_lds(int NB, int G) {
  const size_t ab = (size_t)TM * PLDT + (size_t)PKC * NB * 32, st = (size_t)TM * (2 * 32 + 4);
  return 4 * ((ab > st ? ab : st) + (size_t)12 * G + TM + (size_t)NB * 32);
}

// Persistent pipelined kernel over tiles of 32 2x2 patches, 2 resident workgroups per CU.  Cout = 128 at stride 2
// (a 5x5 window + 4 accumulators per lane) spills and measured slower than the per-tile kernel: left to that one.
bool fp_dwpw_persistent(const fp_op& op) {
  if (op.act2 != FP_ACT_NONE || op.res_mode == FP_RES_SHUFFLE2) return false;   // the per-tile kernel owns those epilogues
  const int NB = (int)fp_round_up(op.Cout, 32) / 32;
  if (!(NB == 2 || (NB == 4 && op.stride == 1))) return false;
  if (op.OH % 2 || op.OW % 2 || (op.stride == 2 && (op.H % 2 || op.W % 2))) return false;
#ifndef FP_DWPW_PERSIST_MIN_TILES
#define FP_DWPW_PERSIST_MIN_TILES 512L   // one tile per persistent workgroup: 108 vs 119 us on the 14x14 blocks at 528 crops
#endif
  if ((long)op.N * op.OH * op.OW < FP_DWPW_PERSIST_MIN_TILES * TM) return false;
  return dwpw_persist_lds(NB, op.Cin) <= 80 * 1024;
}

// The persistent shapes whose whole packed projection matrix (G x Npad floats) fits 64 KB of LDS take the wave-private
// kernel: every Mobile-FaceNet block on the 56 x 56 and 28 x 28 maps (the 14 x 14 blocks have 128 KB of it).
bool fp_dwpw_wave_private(const fp_op& op) {
  return fp_dwpw_persistent(op) && (size_t)op.Cin * fp_round_up(op.Cout, 32) * 4 <= 64 * 1024 &&
         (unsigned long long)op.N * (unsigned long long)op.in_ns * 4ull < (1ull << 32);   // 32-bit byte offsets
}

#ifdef FP_DWPW_STAMPS
static unsigned long long* g_dwpw_stamps = nullptr;
#endif

int fp_launch_dwpw(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  // op: KH = KW = 3, pad 1, stride 1|2, Cin = G (multiple of 64), Cout multiple of 4 and <= 128;
  // w_off -> depthwise block, slope_off -> pointwise block; act = FP_ACT_PRELU when the depthwise has a PReLU;
  // res_mode = FP_RES_ADD_AFTER_ACT for the residual variant.
  if (op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1 || (op.stride != 1 && op.stride != 2))
    return FP_ERR_UNSUPPORTED;
  if (op.Cin % KCH || op.Cout % 4 || op.Cout > 128 || op.Cout <= 0) return FP_ERR_UNSUPPORTED;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.w_off % 4 || op.slope_off % 4)
    return FP_ERR_ALIGNMENT;
  const long OHW = (long)op.OH * op.OW;
  if (op.out_cmul != 1 || op.out_ns != OHW * op.out_ld) return FP_ERR_UNSUPPORTED;
  if (op.OH != (op.H + 2 - 3) / op.stride + 1 || op.OW != (op.W + 2 - 3) / op.stride + 1) return FP_ERR_INVALID_ARG;
  const bool has_res = op.res_mode != FP_RES_NONE, shuffle = op.res_mode == FP_RES_SHUFFLE2;
  if (has_res && ((op.res_mode != FP_RES_ADD_AFTER_ACT && !shuffle) || op.res_ns != OHW * op.res_ld || op.res_ld % 4 ||
                  op.res_off % 4 || op.res_C < op.Cout))
    return FP_ERR_UNSUPPORTED;
  if (shuffle && op.out_ld < 2 * op.Cout) return FP_ERR_INVALID_ARG;
  if (op.act2 != FP_ACT_NONE && op.act2 != FP_ACT_SILU) return FP_ERR_UNSUPPORTED;
  DwPwArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = has_res ? arena + op.res_off : nullptr;
  a.dwp = weights + op.w_off;
  a.pwp = weights + op.slope_off;
  a.oslope = op.bias_off >= 0 ? weights + op.bias_off : nullptr;   // output PReLU (never together with a residual)
  if (a.oslope && (has_res || op.bias_off % 4)) return FP_ERR_UNSUPPORTED;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.G = op.Cin; a.Cout = op.Cout; a.stride = op.stride;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.in_ns = op.in_ns;
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.OHW = (int)OHW;
  a.has_res = (has_res && !shuffle) ? 1 : 0;
  a.shuffle = shuffle ? 1 : 0;
  a.out_silu = op.act2 == FP_ACT_SILU ? 1 : 0;
  a.has_slope = op.act == FP_ACT_PRELU ? 1 : 0;
  a.M = (long)op.N * OHW;
  if (a.M >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  a.ntiles = fp_ceil_div(a.M, TM);
#ifdef FP_DWPW_STAMPS
  a.stamps = g_dwpw_stamps;
#endif
  const int NB = a.Npad / 32;
  const int P = (op.OW % 4 == 0) ? 4 : (op.OW % 2 == 0) ? 2 : 1;
  if (fp_dwpw_persistent(op)) {
    const size_t plds = dwpw_persist_lds(NB, a.G);
    {
      DwPwArgs b = a;
      b.ntiles = (int)fp_ceil_div((long)op.N * (op.OH / 2) * (op.OW / 2), 32);
      int nblk = 512;
      if (nblk > b.ntiles) nblk = b.ntiles;
#define FP_DWPW_PERSIST(NBV, SV)                                                                                \
  do {                                                                                                          \
    if (plds > 64 * 1024)                                                                                       \
      (void)hipFuncSetAttribute((const void*)dwpw_persist_kernel<NBV, SV>,                                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);                         \
    hipLaunchKernelGGL((dwpw_persist_kernel<NBV, SV>), dim3(nblk), dim3(256), plds, s, b);                      \
  } while (0)
      if (fp_dwpw_wave_private(op)) {
        // whole projection matrix resident in LDS: the wave-private kernel, 4 waves per workgroup, 2 workgroups per CU
        // (6 waves x 2 -- three per SIMD -- needs <= 168 VGPRs: the NB = 2 form spills 92 bytes there and runs 2x slower)
        constexpr int NWv = 4;
        const size_t wlds = 4 * ((size_t)12 * a.G + (size_t)NB * 32 + (size_t)a.G * NB * 32 + NWv * (size_t)(32 * PLDT + 32));
        const long ntw = ((long)op.N * (op.OH / 2) * (op.OW / 2) + 7) / 8;
        int nwg = 512;
        if ((long)nwg * NWv > ntw) nwg = (int)((ntw + NWv - 1) / NWv);
#define FP_DWPW_WPK(NBV, SV)                                                                                     \
  do {                                                                                                          \
    (void)hipFuncSetAttribute((const void*)dwpw_wp_kernel<NBV, SV, NWv>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)wlds);                                                                       \
    hipLaunchKernelGGL((dwpw_wp_kernel<NBV, SV, NWv>), dim3(nwg), dim3(NWv * 64), wlds, s, b);                   \
  } while (0)
        if (NB == 2) { if (op.stride == 1) FP_DWPW_WPK(2, 1); else FP_DWPW_WPK(2, 2); }
        else FP_DWPW_WPK(4, 1);
#undef FP_DWPW_WPK
        FP_CHECK_LAUNCH();
        return FP_OK;
      }
      if (NB == 2) { if (op.stride == 1) FP_DWPW_PERSIST(2, 1); else FP_DWPW_PERSIST(2, 2); }
      else FP_DWPW_PERSIST(4, 1);
#undef FP_DWPW_PERSIST
      FP_CHECK_LAUNCH();
      return FP_OK;
    }
  }
  dim3 grid((unsigned)a.ntiles), block(256);
  const size_t lds = 4 * ((size_t)TM * LDT + (size_t)KCH * NB * 32 + 12 * KCH);
#define FP_DWPW_LAUNCH(NBV, PV, SV)                                                                        \
  do {                                                                                                     \
    if (lds > 64 * 1024)                                                                                   \
      (void)hipFuncSetAttribute((const void*)dwpw_kernel<NBV, PV, SV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                 \
    hipLaunchKernelGGL((dwpw_kernel<NBV, PV, SV>), grid, block, lds, s, a);                                \
  } while (0)
#define FP_DWPW_P(NBV, SV)                                 \
  if (P == 4) FP_DWPW_LAUNCH(NBV, 4, SV);                  \
  else if (P == 2) FP_DWPW_LAUNCH(NBV, 2, SV);             \
  else FP_DWPW_LAUNCH(NBV, 1, SV);
#define FP_DWPW_S(NBV)                                     \
  if (op.stride == 1) { FP_DWPW_P(NBV, 1) } else { FP_DWPW_P(NBV, 2) }
  switch (NB) {
    case 1: FP_DWPW_S(1) break;
    case 2: FP_DWPW_S(2) break;
    case 3: FP_DWPW_S(3) break;
    case 4: FP_DWPW_S(4) break;
    default: return FP_ERR_UNSUPPORTED;
  }
#undef FP_DWPW_S
#undef FP_DWPW_P
#undef FP_DWPW_LAUNCH
  FP_CHECK_LAUNCH();
  return FP_OK;
}
