// pws.hip — streaming pointwise (1x1) convolution for K = 64 with wave-private tiles (gfx950).
//
// The expand convs of Mobile-FaceNet's Depth_Wise blocks (fde/modules/mobile_facenet/mobile_facenet.py:70-71:
// Conv_block 64 -> 128 / 256, 1x1, BN, PReLU) are skinny GEMMs: M = N*H*W rows is huge, K = 64.  conv.hip's
// tile-at-a-time kernel starts every 128-row tile cold (A panel load, two barriers per K chunk, a two-pass
// epilogue with four more) and reached 51 % MFMA utilisation / 3.8 TB/s at 3 workgroups per CU (profiles/r01).
// Here nothing is shared between waves except the weights:
//   * the packed weights of the block's 128-column N tile (64 x 128 x 4 B = 32 KiB) are staged into LDS once;
//   * every WAVE owns 32-row tiles: dense NHWC makes a tile's A panel one contiguous 8 KiB range, fetched with 8
//     unconditional 16-B loads per lane into registers, written to the wave's private LDS panel [32][64+4], read
//     back as MFMA fragments; the same private region then stages the transposed output;
//   * no __syncthreads() after the weight staging: the 8 waves of a workgroup drift apart, and while one wave of a
//     SIMD is in its epilogue / waiting for memory the other one runs its 128 MFMAs;
//   * the next tile's panel is requested BEFORE the current tile's MFMAs, and a tile's stores come after the
//     request for the next panel, so waiting for that panel never waits for the stores (vmcnt is one in-order
//     counter for loads and stores);
//   * activation is branch-free: x > 0 ? x : x*s + 0 with s = 1 (none), 0 (ReLU), slope (PReLU).
// Eligibility is decided on the host (fp_pws_eligible); everything else keeps using conv_igemm_kernel.
// Numerics: v_mfma_f32_32x32x2_f32 over k = 0..63 in the same order as conv.hip -> identical results.
#include "common.h"

namespace {

struct PwsArgs {
  const float* in;
  float* out;
  const float* w;
  const float* scale;
  const float* bias;
  const float* slope;
  int Npad, in_ld, out_ld, act, ntiles_n;
  long ntiles;   // 32-row tiles: M / 32
};

constexpr int KC = 64;          // K chunk held in the private panel (K = 64: the whole panel; K = 128: two chunks)
constexpr int LDA = KC + 4;     // private A panel row stride (odd number of 16-B slots: conflict-free b128 reads)
constexpr int BN = 128;         // N tile
constexpr int NB = BN / 32;
constexpr int LDO = 64 + 4;     // output staging row stride (64 columns per pass)
constexpr int PRIV = 32 * LDA;  // floats per wave (32*68; the staging tile [32][LDO] has the same size)

template <int KP, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void pws_kernel(PwsArgs p) {
  constexpr int NCH = KP / KC;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Bs = smem;                                  // [KP/4][BN][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* Ap = smem + KP * BN + wave * PRIV;          // wave-private: A panel, then output staging
  const int lr = lane & 31, h = lane >> 5;

  // block -> (N tile, M sequence): the workgroups that share an A panel sit on the same XCD (b & 7), one after
  // the other in dispatch order, so the second read of a panel is an L2 hit
  const int b = blockIdx.x, ny = p.ntiles_n;
  const int nt = (b >> 3) % ny;
  const int mseq = (b & 7) + 8 * ((b >> 3) / ny);
  const int nseq = (gridDim.x / (8 * ny)) * 8;       // number of M sequences (gridDim.x is a multiple of 8*ny)
  const int n0 = nt * BN;

  // weights of this N tile -> LDS, once: all loads first, then the LDS writes
  {
    constexpr int NW4 = KP * BN / 4, PER = (NW4 + WAVES * 64 - 1) / (WAVES * 64);
    f32x4 wv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = min(tid + WAVES * 64 * j, NW4 - 1);
      const int q = i / BN, col = i - q * BN;
      wv[j] = *(const f32x4*)(p.w + ((long)q * p.Npad + n0 + col) * 4);
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
      if (tid + WAVES * 64 * j < NW4) *(f32x4*)&Bs[(tid + WAVES * 64 * j) * 4] = wv[j];
  }
  // per-column affine of the MFMA layout (column = lane & 31 of each 32-column block)
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + nb * 32 + lr;
    sc[nb] = p.scale ? p.scale[n] : 1.f;
    bi[nb] = p.bias ? p.bias[n] : 0.f;
  }
  // negative-side multiplier of the read-out layout (lane -> 4 consecutive columns of each 64-column half)
  const int ec = (lane & 15) * 4, er = lane >> 4;    // read-out: rows er + 4*j, columns ec .. ec+3 of the half
  f32x4 sl[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const f32x4 one4 = {1.f, 1.f, 1.f, 1.f}, z4 = {0.f, 0.f, 0.f, 0.f};
    sl[hh] = p.act == FP_ACT_PRELU ? *(const f32x4*)(p.slope + n0 + hh * 64 + ec) : (p.act == FP_ACT_RELU ? z4 : one4);
  }
  __syncthreads();   // the only workgroup barrier

  const long tstride = (long)nseq * WAVES;
  long t = (long)mseq * WAVES + wave;
  f32x4 areg[NCH * 8];
  // panel of tile t: rows 32t .. 32t+31 are one contiguous range; lane -> row er + 4*j, 16-B column ec of chunk c
  // (16 lanes = 256 contiguous bytes of a row)
  auto load_panel = [&](long tt) {
    const float* src = p.in + (tt * 32 + er) * (long)p.in_ld + ec;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) areg[c * 8 + j] = *(const f32x4*)(src + (long)(4 * j) * p.in_ld + c * KC);
  };
  if (t < p.ntiles) load_panel(t);

  for (; t < p.ntiles; t += tstride) {
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const float* arow = &Ap[lr * LDA + 4 * h];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // chunk c of the A panel -> private LDS (earlier reads of this region are complete: same wave, in order)
#pragma unroll
      for (int j = 0; j < 8; ++j) *(f32x4*)&Ap[(er + 4 * j) * LDA + ec] = areg[c * 8 + j];
      // next tile's panel: in flight during the (last chunk's) MFMAs, the epilogue and its stores
      if (c == NCH - 1 && t + tstride < p.ntiles) load_panel(t + tstride);
#pragma unroll
      for (int kq = 0; kq < KC / 8; ++kq) {
        const f32x4 a = *(const f32x4*)(arow + kq * 8);
        f32x4 bv[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) bv[nb] = *(const f32x4*)&Bs[((c * (KC / 4) + kq * 2 + h) * BN + nb * 32 + lr) * 4];
        // round-robin over the NB accumulators: an MFMA that accumulates into the previous one's result waits for it
        // (back-to-back dependent 32x32x2 issue), four independent chains keep the pipe full
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], bv[nb][e], acc[nb], 0, 0, 0);
            FP_MFMA_ORDER();
          }
      }
    }

    // epilogue, 64 columns per pass: acc*scale+bias -> private LDS (C/D map: col = lane&31, row = (reg&3) +
    // 8*(reg>>2) + 4*(lane>>5)) -> 16-B rows -> activation -> 256-B row pieces to HBM
    float* orow = p.out + (t * 32 + er) * (long)p.out_ld + n0 + ec;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int nb = hh * 2 + q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;
          Ap[row * LDO + q * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
        }
      }
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)&Ap[(er + 4 * j) * LDO + ec];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] = v[j][e] > 0.f ? v[j][e] : __builtin_fmaf(v[j][e], sl[hh][e], 0.0f);
        *(f32x4*)(orow + (long)(4 * j) * p.out_ld + hh * 64) = v[j];
      }
    }
  }
}

}  // namespace

// Host-side eligibility: pointwise, dense NHWC in/out, K = 64 or 128, Cout a multiple of 128, M a multiple of 32,
// no residual, activation none / ReLU / PReLU.  Everything else stays with conv_igemm_kernel.
bool fp_pws_eligible(const fp_op& op) {
  if (op.kind != FP_OP_CONV) return false;
  if (op.KH != 1 || op.KW != 1 || op.stride != 1 || op.pad_t || op.pad_l) return false;
  if (op.OH != op.H || op.OW != op.W || op.out_cmul != 1) return false;
  if ((op.Cin != 64 && op.Cin != 128) || op.Cout % BN || op.Cout <= 0) return false;
  if (8 * (op.Cout / BN) > 256) return false;   // the grid is a multiple of 8 * ntiles_n workgroups, at most 256
  const long HW = (long)op.H * op.W, M = (long)op.N * HW;
  if (op.in_ns != HW * op.in_ld || op.out_ns != HW * op.out_ld) return false;   // row m at base + m*ld
  if (op.in_ld % 4 || op.in_off % 4 || op.out_ld % 4 || op.out_off % 4 || op.w_off % 4) return false;
  if (op.res_mode != FP_RES_NONE) return false;
  if (op.act != FP_ACT_NONE && op.act != FP_ACT_RELU && op.act != FP_ACT_PRELU) return false;
  if (op.act == FP_ACT_PRELU && (op.slope_off < 0 || op.slope_off % 4)) return false;
  if (M % 32 || M < 32L * 12 * 256) return false;   // whole 32-row tiles, and enough of them to fill the chip
  return true;
}

int fp_launch_pws(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  PwsArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.Npad = op.Cout;   // a multiple of 128
  a.in_ld = op.in_ld;
  a.out_ld = op.out_ld;
  a.act = op.act;
  a.ntiles_n = op.Cout / BN;
  a.ntiles = (long)op.N * op.H * op.W / 32;
  // K = 64: weights 32 KiB + 12 x 8.5 KiB private panels = 134 KiB, 12 waves (3 per SIMD, 167 VGPRs);
  // K = 128: weights 64 KiB + 8 x 8.5 KiB = 132 KiB, 8 waves.  One workgroup per CU either way.
  const int waves = op.Cin == 64 ? 12 : 8;
  const size_t lds = 4 * ((size_t)op.Cin * BN + (size_t)waves * PRIV);
  // more than the default 64 KiB of dynamic LDS: the opt-in is per device and idempotent, so it is made on every
  // launch (a process-wide "done" flag would miss a second GPU and is not thread-safe)
  const hipError_t ae = op.Cin == 64
      ? hipFuncSetAttribute((const void*)pws_kernel<64, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
      : hipFuncSetAttribute((const void*)pws_kernel<128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  // one workgroup per CU; the grid is a multiple of 8 * ntiles_n so that every XCD serves every N tile
  int grid = 256;
  const int unit = 8 * a.ntiles_n;
  grid = grid / unit * unit;
  if (op.Cin == 64) hipLaunchKernelGGL((pws_kernel<64, 12>), dim3(grid), dim3(12 * 64), lds, s, a);
  else hipLaunchKernelGGL((pws_kernel<128, 8>), dim3(grid), dim3(8 * 64), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
