// blaze.hip — fused BlazeBlock (fde/modules/blazeface/blazeface.py:12-47) for gfx950.
//
//   y = ReLU( pw1x1( dw3x3_stride_s(h) ) + shortcut ),   h = x (s=1, pad 1) | pad(x,(0,2,0,2)) (s=2, pad 0)
//   shortcut = x (s=1) | maxpool2x2(x) (s=2), zero-padded on channels when Cout > Cin.
//
// The unfused pair (DWCONV + CONV) writes the depthwise result to HBM and reads it back, and reads x a second
// time for the shortcut: 4 tensor passes.  Here one workgroup owns 128 consecutive output pixels (NHWC, so its
// output is ONE contiguous byte range) and keeps everything in between on chip:
//   phase 1  all 256 lanes: depthwise 3x3 for (pixel, 4-channel group) items straight from global memory with
//            16-B loads (the 9 taps of neighbouring pixels hit L1/L2; the block->tile map below keeps vertically
//            adjacent tiles on one XCD so the halo rows are L2 hits, not HBM re-reads).  The result goes to an LDS
//            tile A[128][K+4]; the shortcut value is a by-product (s=1: the centre tap; s=2: max of taps
//            (0..1,0..1), which ARE the 2x2 pool window) and goes to a second LDS tile.
//   phase 2  4 waves x 32 rows: v_mfma_f32_32x32x2_f32 over K = Cin against the packed 1x1 weights in LDS
//            (same fragment scheme as conv.hip: one ds_read_b128 per 4 k-steps, row stride K+4 floats -> an odd
//            number of 16-B slots -> conflict free).
//   phase 3  epilogue in registers (bias + shortcut from LDS + ReLU), transposed through LDS so the tile leaves
//            the CU as fully coalesced 16-B stores.
// HBM traffic per block-layer: x once + y once (op-granular model counts 4 passes).
#include "common.h"

namespace {

struct BlazeArgs {
  const float* in;
  float* out;
  const float* wd;   // [9][Cin]
  const float* bd;   // [Cin]
  const float* wp;   // packed [Kpad/4][Npad][4]
  const float* bp;   // [Cout]
  int N, H, W, OH, OW, Cin, Cout, stride;
  int in_ld, out_ld;
  long in_ns, out_ns;
  int Kpad, Npad, OHW, C4, res_C;
  long M;
  int ntiles;
  fp_divisor ohw_div, ow_div;   // persistent kernel: pixel index -> (image, y, x)
  int out_rowpad;               // persistent kernel: output in the row-padded layout
};

constexpr int TM = 128;
constexpr size_t FP_BLAZEBLOCK_MAX_LDS = 80 * 1024;  // two workgroups per CU (160 KiB LDS)

template <int NB>
__global__ __launch_bounds__(256) void blazeblock_kernel(BlazeArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDT = p.Kpad + 4;          // A tile row stride
  const int LDS_ = p.Cin + 4;          // shortcut tile row stride
  float* At = smem;                                  // [TM][LDT]   (later reused as the output tile [TM][Cout])
  const int a_floats = TM * (LDT > p.Cout ? LDT : p.Cout);
  float* St = At + a_floats;                         // [TM][LDS_]
  float* Bs = St + TM * LDS_;                        // [Kpad/4][Npad][4]
  float* Ws = Bs + p.Kpad * p.Npad;                  // depthwise weights [9][Cin] + bias [Cin]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;

  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so give each XCD a
  // contiguous range of tiles: vertically adjacent rows (the 3x3 halo) then meet in the same L2.  Bijective
  // for any ntiles (cdna_hip_programming.md T1).
  int tile;
  {
    const int b = blockIdx.x, q = p.ntiles / 8, r = p.ntiles % 8, xcd = b & 7, k = b >> 3;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
  }
  const long m0 = (long)tile * TM;

  // stage the pointwise weights
  const int nB4 = (p.Kpad >> 2) * p.Npad;
  for (int i = tid; i < nB4; i += 256) *(f32x4*)&Bs[i * 4] = *(const f32x4*)(p.wp + (long)i * 4);

  // depthwise weights [9][Cin] + bias [Cin] -> LDS (read back with ds_read_b128, no global latency per tap)
  const int nW4 = (10 * p.Cin) >> 2;
  for (int i = tid; i < nW4; i += 256)
    *(f32x4*)&Ws[i * 4] = (i * 4 < 9 * p.Cin) ? *(const f32x4*)(p.wd + (long)i * 4)
                                              : *(const f32x4*)(p.bd + ((long)i * 4 - 9 * p.Cin));
  __syncthreads();

  // phase 1: depthwise + shortcut.  One lane = 4 consecutive output pixels (same image row: OW % 4 == 0) x 4
  // channels: the 3 x (3*s+3) input window is loaded once per row with branch-free, clamped 16-B loads and slides
  // over the 4 outputs (18 loads per 4 outputs at s=1 instead of 36; 27 instead of 36 at s=2).
  const int KC4 = p.Kpad >> 2;  // A columns (in float4) incl. zero padding
  for (int it = tid; it < (TM / 4) * KC4; it += 256) {
    const int g = it / KC4, c4 = it - g * KC4;
    const int r = g * 4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {z, z, z, z}, sc[4] = {z, z, z, z};
    if (c4 < p.C4) {
      long m = m0 + r;
      m = m < p.M ? m : p.M - 4;  // tail groups recompute the last pixels; they are never stored
      const unsigned mm = (unsigned)m;
      const unsigned img = mm / (unsigned)p.OHW;
      const unsigned rem = mm - img * (unsigned)p.OHW;
      const int oy = (int)(rem / (unsigned)p.OW), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
      const int c = c4 * 4;
      const float* ib = p.in + (long)img * p.in_ns + c;
      const f32x4 bias = *(const f32x4*)&Ws[9 * p.Cin + c];
      if (p.stride == 1) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy - 1 + ky;
          const bool vy = (unsigned)iy < (unsigned)p.H;
          const float* rowp = ib + (long)min(max(iy, 0), p.H - 1) * p.W * p.in_ld;
          f32x4 x[6];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            const int ix = ox - 1 + j;
            const bool v = vy && ((unsigned)ix < (unsigned)p.W);
            const f32x4 t = *(const f32x4*)(rowp + (long)min(max(ix, 0), p.W - 1) * p.in_ld);
            x[j] = v ? t : z;
          }
          const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * p.Cin + c];
          const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * p.Cin + c];
          const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * p.Cin + c];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[q] += x[q] * w0;
            acc[q] += x[q + 1] * w1;
            acc[q] += x[q + 2] * w2;
            if (ky == 1) sc[q] = x[q + 1];  // centre tap = x itself
          }
        }
      } else {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = 2 * oy + ky;
          const bool vy = iy < p.H;
          const float* rowp = ib + (long)min(iy, p.H - 1) * p.W * p.in_ld;
          f32x4 x[9];
#pragma unroll
          for (int j = 0; j < 9; ++j) {
            const int ix = 2 * ox + j;
            const bool v = vy && ix < p.W;
            const f32x4 t = *(const f32x4*)(rowp + (long)min(ix, p.W - 1) * p.in_ld);
            x[j] = v ? t : z;
          }
          const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * p.Cin + c];
          const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * p.Cin + c];
          const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * p.Cin + c];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[q] += x[2 * q] * w0;
            acc[q] += x[2 * q + 1] * w1;
            acc[q] += x[2 * q + 2] * w2;
            if (ky < 2) {  // rows 0,1 x cols 2q,2q+1 are the 2x2 max-pool window (always inside the map)
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float mx = fmaxf(x[2 * q][e], x[2 * q + 1][e]);
                sc[q][e] = ky == 0 ? mx : fmaxf(sc[q][e], mx);
              }
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] += bias;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *(f32x4*)&At[(r + q) * LDT + c4 * 4] = acc[q];
      if (c4 < p.C4) *(f32x4*)&St[(r + q) * LDS_ + c4 * 4] = sc[q];
    }
  }
  __syncthreads();

  // phase 2: 1x1 conv on the MFMA pipe
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  const float* arow = &At[(wave * 32 + lr) * LDT + 4 * h];
  const int ngroups = p.Kpad >> 3;
  for (int kq = 0; kq < ngroups; ++kq) {
    const f32x4 a = *(const f32x4*)(arow + kq * 8);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const f32x4 b = *(const f32x4*)&Bs[((kq * 2 + h) * p.Npad + nb * 32 + lr) * 4];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc[nb], 0, 0, 0);
    }
  }
  __syncthreads();  // every wave is done reading At before it becomes the output tile

  // phase 3: epilogue -> LDS output tile [TM][Cout] -> coalesced 16-B stores
  float* Ot = At;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 32 + lr;
    if (n < p.Cout) {
      const float bias = p.bp[n];
      const bool has_sc = n < p.res_C;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = wave * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        float v = acc[nb][reg] + bias;
        if (has_sc) v += St[row * LDS_ + n];
        Ot[row * p.Cout + n] = v > 0.f ? v : 0.f;
      }
    }
  }
  __syncthreads();
  {
    // rows m0 .. m0+TM-1 are consecutive pixels of a dense NHWC tensor (out_ld == Cout, out_ns == OHW*Cout)
    const long rows_left = p.M - m0;
    const int nrows = rows_left < TM ? (int)rows_left : TM;
    const int n4 = nrows * p.Cout / 4;
    if (!p.out_rowpad) {
      float* obase = p.out + m0 * p.Cout;
      for (int i = tid; i < n4; i += 256) *(f32x4*)(obase + (long)i * 4) = *(const f32x4*)&Ot[i * 4];
    } else {   // row-padded output (facepath.h FP_OPF_OUT_ROWPAD)
      const int q4 = p.Cout >> 2;
      for (int i = tid; i < n4; i += 256) {
        const unsigned px = (unsigned)i / (unsigned)q4, cq = (unsigned)i - px * (unsigned)q4;
        const unsigned m = (unsigned)m0 + px;
        const unsigned img = fp_fastdiv(m, p.ohw_div), rem = m - img * (unsigned)p.OHW;
        const unsigned oy = fp_fastdiv(rem, p.ow_div);
        *(f32x4*)(p.out + (long)img * p.out_ns + ((long)rem + oy) * p.Cout + cq * 4) = *(const f32x4*)&Ot[i * 4];
      }
    }
  }
}

// Persistent variant for the narrow blocks (Kpad <= 32, Cout <= 32: the 24-channel BlazeBlocks that carry ~60 % of
// BlazeFace's bytes).  The per-tile kernel above starts every tile cold: weight staging, then the depthwise window
// loads, each a full memory round trip with nothing else to do (rocprofv3: waves 50 % in s_waitcnt, 2.8 TB/s).
// Here a workgroup stages the weights ONCE and walks tiles t = k*G + swz(block); the 3 x (3S+3) window of the NEXT
// tile is loaded into registers right after the current tile's depthwise values are in LDS, so its HBM latency
// hides under the MFMAs, the epilogue and the stores.  One lane = one (4-pixel group, 4-channel group) item.
// CIN / COUT != 0 fix the widths at compile time (the 24 -> 24 blocks of the back model): every LDS stride becomes an
// immediate and the k loop unrolls.
//
// Nothing in this kernel is saturated (rocprofv3 SQ counters: VALU ~36 % of a SIMD, MFMA pipe 21 %, HBM 3.5-3.9
// TB/s): it is bound by the serial chain of one tile inside a workgroup, of which only 3 overlap on a CU.  Round 2
// shortened that chain (ISA of the first version in FINDINGS.md finding 12), 231 -> 208 us on the 128 x 128 blocks:
//   * the window registers hold RAW loads; the zero padding is applied from a bit mask when the window is consumed,
//     one tile later.  A select next to the load made every wave wait for its 18 loads inside the "prefetch";
//   * the shortcut values are read from LDS in one batch before the epilogue arithmetic (an `if (has_sc) v += St[..]`
//     per accumulator register had compiled to 16 exec-masked LDS round trips, each waited for on its own);
//   * two MFMA accumulators (even / odd k steps): a 32x32x2 f32 MFMA that accumulates into the previous one's result
//     issues at half rate;
//   * pixel index -> (image, y, x) by multiply-high with host-computed reciprocals (fp_fastdiv) instead of two
//     16-instruction integer divisions per tile; the pointwise bias comes from LDS (a global load first used inside
//     the loop put a vmcnt(0) -- a wait for the prefetch -- into the epilogue);
//   * the three waves that do the depthwise phase (192 of 256 lanes at 24 channels) rotate with the block index, so the
//     idle role does not land on the same SIMD in every co-resident workgroup.
// Measured and rejected: storing a tile after the NEXT tile's depthwise phase (so the stores are older than the prefetch
// in the in-order vmcnt queue) together with the shortcut as identity rows of the GEMM (no shortcut tile, two barriers):
// 228 us -- the doubled MFMA chain and LDS A traffic cost more than the store drain they avoid.
template <int S, int CIN, int COUT>
__global__ __launch_bounds__(256, S == 1 ? 3 : 2) void blazeblock_persist_kernel(BlazeArgs p) {
  constexpr int WIN = 3 * S + 3;
  const int Cin = CIN ? CIN : p.Cin, Cout = COUT ? COUT : p.Cout;
  const int Kpad = CIN ? (CIN + 7) / 8 * 8 : p.Kpad;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LDT = Kpad + 4, LDS_ = Cin + 4;
  float* At = smem;                     // [TM][LDT]
  float* St = At + TM * LDT;            // [TM][LDS_]
  float* Ot = St + TM * LDS_;           // [TM][Cout]   (separate from At: one barrier less per tile)
  float* Bs = Ot + TM * Cout;           // [Kpad/4][32][4]
  float* Ws = Bs + Kpad * 32;           // [10][Cin]
  float* Bp = Ws + 10 * Cin;            // [32] pointwise bias
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;

  if (tid < 32) Bp[tid] = tid < Cout ? p.bp[tid] : 0.f;
  for (int i = tid; i < (Kpad >> 2) * 32; i += 256) *(f32x4*)&Bs[i * 4] = *(const f32x4*)(p.wp + (long)i * 4);
  for (int i = tid; i < (10 * Cin) >> 2; i += 256)
    *(f32x4*)&Ws[i * 4] = (i * 4 < 9 * Cin) ? *(const f32x4*)(p.wd + (long)i * 4)
                                            : *(const f32x4*)(p.bd + ((long)i * 4 - 9 * Cin));

  // depthwise items: (4-pixel group g, 4-channel group c4); the wave -> item-range map rotates with the block index
  const int KC4 = Kpad >> 2;
  const int dtid = (((wave - (int)(blockIdx.x >> 3) - (int)(blockIdx.x >> 8)) & 3) << 6) + lane;
  const int g = dtid / KC4, c4 = dtid - g * KC4;
  const bool in_tile = dtid < (TM / 4) * KC4;
  const bool active = in_tile && c4 < (Cin >> 2);
  const int r = g * 4, c = c4 * 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  // XCD-aware position of this block inside a window of G tiles (bijective for any G)
  const int G = gridDim.x;
  int pos;
  {
    const int b = blockIdx.x, q = G / 8, rr = G % 8, xcd = b & 7, k = b >> 3;
    pos = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
  }

  // Window loads use 32-bit BYTE offsets from the (wave-uniform, SGPR) tensor base: one address register and one
  // v_add_u32 per load instead of a 64-bit multiply-add each (the launcher checks the tensor is < 4 GiB).
  const char* inb = (const char*)p.in;
  const unsigned row_b = (unsigned)(p.W * p.in_ld) * 4u, px_b = (unsigned)p.in_ld * 4u;
  f32x4 x[3][WIN];
  unsigned vmask = 0;                   // bits 0..2: row ky inside the map, bits 3..3+WIN-1: column j inside the map
  auto issue_loads = [&](int tile) {
    const unsigned mm = min((unsigned)tile * TM + (unsigned)r, (unsigned)p.M - 4u);
    const unsigned img = fp_fastdiv(mm, p.ohw_div);
    const unsigned rem = mm - img * (unsigned)p.OHW;
    const int oy = (int)fp_fastdiv(rem, p.ow_div), ox = (int)(rem - (unsigned)oy * (unsigned)p.OW);
    const unsigned ib = (img * (unsigned)p.in_ns + (unsigned)c) * 4u;
    const int iy0 = S == 1 ? oy - 1 : 2 * oy, ix0 = S == 1 ? ox - 1 : 2 * ox;
    unsigned colo[WIN];
    unsigned vm = 0;
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
      const int ix = ix0 + j;
      colo[j] = (unsigned)min(max(ix, 0), p.W - 1) * px_b;
      if ((unsigned)ix < (unsigned)p.W) vm |= 8u << j;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = iy0 + ky;
      if ((unsigned)iy < (unsigned)p.H) vm |= 1u << ky;
      const unsigned rowo = ib + (unsigned)min(max(iy, 0), p.H - 1) * row_b;
#pragma unroll
      for (int j = 0; j < WIN; ++j) x[ky][j] = *(const f32x4*)(inb + (rowo + colo[j]));
    }
    vmask = vm;
  };
  auto store_tile = [&](int tile) {     // Ot -> the tile's contiguous byte range of the dense NHWC output
    const long m0 = (long)tile * TM;
    const long rows_left = p.M - m0;
    const int n4 = (rows_left < TM ? (int)rows_left : TM) * Cout / 4;
    if (!p.out_rowpad) {
      float* obase = p.out + m0 * Cout;
      for (int i = tid; i < n4; i += 256) *(f32x4*)(obase + (long)i * 4) = *(const f32x4*)&Ot[i * 4];
    } else {                            // row-padded output (facepath.h FP_OPF_OUT_ROWPAD): one more pixel per row
      const int q4 = Cout >> 2;
      for (int i = tid; i < n4; i += 256) {
        const unsigned px = (unsigned)i / (unsigned)q4, cq = (unsigned)i - px * (unsigned)q4;
        const unsigned m = (unsigned)m0 + px;
        const unsigned img = fp_fastdiv(m, p.ohw_div), rem = m - img * (unsigned)p.OHW;
        const unsigned oy = fp_fastdiv(rem, p.ow_div);
        *(f32x4*)(p.out + (long)img * p.out_ns + ((long)rem + oy) * Cout + cq * 4) = *(const f32x4*)&Ot[i * 4];
      }
    }
  };

  int tile = pos;
  if (tile < p.ntiles && active) issue_loads(tile);
  __syncthreads();  // weights staged
  const float bias_n = Bp[lr];
  const bool has_sc = lr < p.res_C;
  const int lrs = min(lr, p.res_C - 1);
  while (tile < p.ntiles) {
    // depthwise + shortcut from the prefetched window
    if (active) {
      const f32x4 bias = *(const f32x4*)&Ws[9 * Cin + c];
      f32x4 acc[4] = {bias, bias, bias, bias}, sc[4];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const f32x4 w0 = *(const f32x4*)&Ws[(ky * 3 + 0) * Cin + c];
        const f32x4 w1 = *(const f32x4*)&Ws[(ky * 3 + 1) * Cin + c];
        const f32x4 w2 = *(const f32x4*)&Ws[(ky * 3 + 2) * Cin + c];
        const bool vy = (vmask >> ky) & 1u;
#pragma unroll
        for (int j = 0; j < WIN; ++j) {
          // only the first and the last window column (stride 1) / the last one (stride 2) and the first / last row
          // can lie outside the map; the shortcut taps (centre, 2x2 pool window) never do
          const bool edge_col = S == 1 ? (j == 0 || j == WIN - 1) : (j == WIN - 1);
          const bool edge_row = S == 1 ? ky != 1 : ky == 2;
          if (edge_col && edge_row) x[ky][j] = (vy && ((vmask >> (3 + j)) & 1u)) ? x[ky][j] : z;
          else if (edge_col) x[ky][j] = ((vmask >> (3 + j)) & 1u) ? x[ky][j] : z;
          else if (edge_row) x[ky][j] = vy ? x[ky][j] : z;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // three statements: each contracts to one packed FMA on the accumulator
          acc[q] += x[ky][q * S] * w0;
          acc[q] += x[ky][q * S + 1] * w1;
          acc[q] += x[ky][q * S + 2] * w2;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (S == 1) {
          sc[q] = x[1][q + 1];  // centre tap = x itself
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)  // rows 0,1 x cols 2q,2q+1 = the 2x2 max-pool window
            sc[q][e] = fmaxf(fmaxf(x[0][2 * q][e], x[0][2 * q + 1][e]), fmaxf(x[1][2 * q][e], x[1][2 * q + 1][e]));
        }
        *(f32x4*)&At[(r + q) * LDT + c] = acc[q];
        *(f32x4*)&St[(r + q) * LDS_ + c] = sc[q];
      }
    } else if (in_tile) {
#pragma unroll
      for (int q = 0; q < 4; ++q) *(f32x4*)&At[(r + q) * LDT + c] = z;  // zero padding columns of the K dimension
    }
    __syncthreads();
    const int next = tile + G;
    if (next < p.ntiles && active) issue_loads(next);  // in flight during MFMA, epilogue and the stores

    f32x16 m0, m1;
#pragma unroll
    for (int i = 0; i < 16; ++i) m0[i] = bias_n, m1[i] = 0.f;   // pointwise bias as the accumulator's start value
    const float* arow = &At[(wave * 32 + lr) * LDT + 4 * h];
    for (int kq = 0; kq < (Kpad >> 3); ++kq) {
      const f32x4 a = *(const f32x4*)(arow + kq * 8);
      const f32x4 b = *(const f32x4*)&Bs[((kq * 2 + h) * 32 + lr) * 4];
      m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], m0, 0, 0, 0);
      FP_MFMA_ORDER();
      m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], m1, 0, 0, 0);
      FP_MFMA_ORDER();
      m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], m0, 0, 0, 0);
      FP_MFMA_ORDER();
      m1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], m1, 0, 0, 0);
      FP_MFMA_ORDER();
    }
    {
      // shortcut values of this lane's 16 (row, column lr) outputs: one batch of unconditional LDS reads from a
      // clamped column, selected below; then the arithmetic
      float sv[16];
      const float* srow = &St[(wave * 32 + 4 * h) * LDS_ + lrs];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) sv[reg] = srow[((reg & 3) + 8 * (reg >> 2)) * LDS_];
      if (lr < Cout) {
        float* orow = &Ot[(wave * 32 + 4 * h) * Cout + lr];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float v = (m0[reg] + m1[reg]) + (has_sc ? sv[reg] : 0.f);
          orow[((reg & 3) + 8 * (reg >> 2)) * Cout] = v > 0.f ? v : 0.f;
        }
      }
    }
    __syncthreads();  // Ot complete; At / St free for the next tile
    store_tile(tile);
    tile = next;
  }
}


}  // namespace

size_t fp_blazeblock_lds_bytes(int Cin, int Cout) {
  const int Kpad = (int)fp_round_up(Cin, 8), Npad = (int)fp_round_up(Cout, 32);
  const int LDT = Kpad + 4;
  const size_t a = (size_t)TM * (LDT > Cout ? LDT : Cout);
  return 4 * (a + (size_t)TM * (Cin + 4) + (size_t)Kpad * Npad + (size_t)10 * Cin);
}

bool fp_blazeblock_fixed24(const fp_op& op) { return op.Cin == 24 && op.Cout == 24; }

int fp_launch_blazeblock(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  // op fields: w_off = depthwise weights [9][Cin], scale_off = depthwise bias, slope_off = packed pointwise
  // weights, bias_off = pointwise bias; res_C = channels of the shortcut (= logical Cin).
  if (op.KH != 3 || op.KW != 3 || (op.stride != 1 && op.stride != 2)) return FP_ERR_UNSUPPORTED;
  if (op.flags & FP_OPF_IN_ROWPAD) return fp_launch_blazeblock_rowpad(op, weights, arena, s);   // blazewp.hip
  if (op.Cin % 4 || op.Cout % 4 || op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_off % 4) return FP_ERR_ALIGNMENT;
  const bool out_rowpad = (op.flags & FP_OPF_OUT_ROWPAD) != 0;
  if (op.out_cmul != 1 || op.out_ld != op.Cout || (!out_rowpad && op.out_ns != (int64_t)op.OH * op.OW * op.Cout))
    return FP_ERR_UNSUPPORTED;
  if (op.stride == 1 && (op.OH != op.H || op.OW != op.W)) return FP_ERR_INVALID_ARG;
  if (op.stride == 2 && (op.H % 2 || op.W % 2 || op.OH != op.H / 2 || op.OW != op.W / 2)) return FP_ERR_INVALID_ARG;
  if (op.Cout > 128 || op.res_C > op.Cin || op.OW % 4) return FP_ERR_UNSUPPORTED;
  const size_t lds = fp_blazeblock_lds_bytes(op.Cin, op.Cout);
  if (lds > FP_BLAZEBLOCK_MAX_LDS) return FP_ERR_UNSUPPORTED;  // the planner emits the unfused pair for wider blocks
  BlazeArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.wd = weights + op.w_off;
  a.bd = weights + op.scale_off;
  a.wp = weights + op.slope_off;
  a.bp = weights + op.bias_off;
  a.N = op.N; a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.Cin = op.Cin; a.Cout = op.Cout;
  a.stride = op.stride; a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.in_ns = op.in_ns; a.out_ns = op.out_ns;
  a.Kpad = (int)fp_round_up(op.Cin, 8);
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.OHW = op.OH * op.OW;
  a.C4 = op.Cin / 4;
  a.res_C = op.res_C;
  a.M = (long)op.N * a.OHW;
  if (a.M >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  a.ntiles = fp_ceil_div(a.M, TM);
  a.out_rowpad = out_rowpad;
  a.ohw_div = fp_make_divisor((unsigned)a.OHW);   // OHW >= 4, OW >= 4 (OW % 4 == 0 above)
  a.ow_div = fp_make_divisor((unsigned)op.OW);
  if (a.Kpad <= 32 && a.Npad == 32 && a.ntiles >= 2048 &&
      (unsigned long long)op.N * (unsigned long long)op.in_ns * 4ull < (1ull << 32)) {   // 32-bit byte offsets
    // persistent kernel: 3 (stride 1) / 2 (stride 2) resident workgroups per CU, each striding over the tiles
    const size_t plds = 4 * ((size_t)TM * (a.Kpad + 4) + (size_t)TM * (op.Cin + 4) + (size_t)TM * op.Cout +
                             (size_t)a.Kpad * 32 + (size_t)10 * op.Cin + 32);
    const int per_cu = op.stride == 1 ? 3 : 2;
    int G = 256 * per_cu;
    if (G > a.ntiles) G = a.ntiles;
    if (plds <= 64 * 1024) {
      const bool w24 = fp_blazeblock_fixed24(op);
      if (op.stride == 1) {
        if (w24) hipLaunchKernelGGL((blazeblock_persist_kernel<1, 24, 24>), dim3(G), dim3(256), plds, s, a);
        else hipLaunchKernelGGL((blazeblock_persist_kernel<1, 0, 0>), dim3(G), dim3(256), plds, s, a);
      } else {
        if (w24) hipLaunchKernelGGL((blazeblock_persist_kernel<2, 24, 24>), dim3(G), dim3(256), plds, s, a);
        else hipLaunchKernelGGL((blazeblock_persist_kernel<2, 0, 0>), dim3(G), dim3(256), plds, s, a);
      }
      FP_CHECK_LAUNCH();
      return FP_OK;
    }
  }
  dim3 grid((unsigned)a.ntiles), block(256);
#define FP_BB_CASE(NBV)                                                                                     \
  case NBV:                                                                                                 \
    if (lds > 64 * 1024)  /* opt in to more than the default 64 KiB of dynamic LDS (idempotent) */          \
      (void)hipFuncSetAttribute((const void*)blazeblock_kernel<NBV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                  \
    hipLaunchKernelGGL((blazeblock_kernel<NBV>), grid, block, lds, s, a);                                   \
    break;
  switch (a.Npad / 32) {
    FP_BB_CASE(1)
    FP_BB_CASE(2)
    FP_BB_CASE(3)
    FP_BB_CASE(4)
    default: return FP_ERR_UNSUPPORTED;
  }
#undef FP_BB_CASE
  FP_CHECK_LAUNCH();
  return FP_OK;
}
