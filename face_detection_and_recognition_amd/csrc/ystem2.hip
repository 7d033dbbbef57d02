// ystem2.hip — the tail of YOLOv5-face's StemBlock in one kernel, bf16x6 split MFMAs (split.h) — gfx950.
//
// StemBlock.forward (fde/modules/yolov5_face/pytorch/models/common.py:58-73) behind ystem.hip's head (stem_1, stem_2a, max pool):
//     b   = stem_2b(a)                 Conv 3x3 stride 2 pad 1, c/2 -> c, (BN), SiLU          320^2 -> 160^2 for 640^2 images
//     out = stem_3(cat(b, p))          Conv 1x1, 2c -> c, (BN), SiLU                          p = the pooled stem_1 map
// Op by op (round 3: convx6_kernel<2> with a flattened K = 144, then conv_igemm) b makes a round trip through HBM and the two
// launches run at 3.1 and 5.7 TB/s of a 5 GB total: 0.80 + 0.44 ms per 256 images, more than the head of the block.  Here a
// workgroup owns a 4 x 16 tile of output pixels (persistent, contiguous runs of tiles, the next tile's input prefetched into
// registers under the MFMAs):
//   A    the 9 x 33 pixels of `a` under the tile (16 channels each) are split ONCE into three bf16 planes in LDS -- even and odd
//        columns in separate runs of a row, 48-byte slots: a lane's fragment (one tap's 8 channels of one output pixel) is a
//        conflict-free 16-byte read, no im2col anywhere;
//   2b   K = 9 taps x 16 channels = 4.5 slabs of 32: wave w = output row w of the tile, both 16-channel tiles of c = 32;
//        W^T x A^T (operands swapped: a lane ends up with four consecutive channels of one pixel) -> BN + SiLU -> split ->
//        a wave-private D tile (the layout change between an MFMA's output and the next one's operand);
//   3    [b | p]: slab 0 from the D tile, slab 1 = the pooled pixel's 32 channels straight from global memory, split in
//        registers -> BN + SiLU -> 16-byte stores.
// HBM: a once (+ halo), p once, out once = 3.4 GB for 256 images at 640^2.
#include <string.h>

#include "split.h"

namespace {

struct YStem2Args {
  const float* a;
  const float* pool;
  float* out;
  const float* w;              // parameter blob (YS2 offsets)
  int H, W, OH, OW, a_ld, p_ld, out_ld, tiles_x, tiles_per_img, ntiles;
  long a_ns, p_ns, out_ns;
};

struct YS2 {
  static constexpr int TH = 4, TW = 16, ER = 2 * TH + 1, EC = 2 * TW + 1, NSLOT = ER * EC;   // 9 x 33 = 297 input pixels
  static constexpr int LDS_A = 24;                           // bf16 per slot (16 channels + 8: 48 bytes, odd number of 16-byte units)
  static constexpr int APL = (NSLOT + 1) * LDS_A;            // bf16 per plane (+ a zero slot for the taps that do not exist)
  static constexpr int LDA = 40, DPL = TH * TW * LDA;        // D tile: [3][64 px][32 + 8] bf16
  static constexpr int NSLAB = 5;                            // ceil(9 * 16 / 32)
  static constexpr int NITEM = (NSLOT * 4 + 255) / 256;      // staging items (slot, channel quad) per thread
  static constexpr int W3L = 2 * 3 * 32 * 32;                // bf16 elements of stem_3's weights (kept in LDS)
  static constexpr int LDS_BYTES = 3 * APL * 2 + 3 * DPL * 2 + W3L * 2;
  static constexpr long O_W2B = 0;                           // [5 slabs][3 planes][32 co][32 k] bf16, k = tap * 16 + c
  static constexpr long O_AFF2B = O_W2B + NSLAB * 3 * 32 * 16;   // [32] scale, [32] bias
  static constexpr long O_W3 = O_AFF2B + 64;                 // [2 slabs][3 planes][32 co][32 k] bf16, k: b's channels, then p's
  static constexpr long O_AFF3 = O_W3 + 2 * 3 * 32 * 16;     // [32] scale, [32] bias
  static constexpr long TOTAL = O_AFF3 + 64;
  static_assert(LDS_BYTES <= 80 * 1024, "two workgroups per CU");
};

__device__ __forceinline__ u32x4 ldg16(const unsigned short* p) { return *(const u32x4*)p; }

__global__ __launch_bounds__(256, 2) void ystem2_x6_kernel(YStem2Args p) {
  using K = YS2;
  constexpr int LDA = K::LDA, DPL = K::DPL, APL = K::APL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Al = (unsigned short*)smem_raw;                      // [3][NSLOT + 1][LDS_A]
  unsigned short* Dl = Al + 3 * APL;                                   // [3][64][LDA]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const unsigned short* wb = (const unsigned short*)p.w;

  // the zero slot (taps 9 .. 9: the second half of slab 4) -- written once, never overwritten
  if (tid < 3 * K::LDS_A / 2) ((unsigned*)Al)[(tid / (K::LDS_A / 2)) * (APL / 2) + K::NSLOT * (K::LDS_A / 2) + tid % (K::LDS_A / 2)] = 0u;

  // raw fp32 of a tile's staging items: item = (slot, channel quad); out-of-picture pixels are zeroed when they are split
  auto load_a = [&](int tile, f32x4 (&ar)[K::NITEM]) {
    const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
    const int iy0 = 2 * (tt / p.tiles_x) * K::TH - 1, ix0 = 2 * (tt % p.tiles_x) * K::TW - 1;
    const float* ain = p.a + (long)img * p.a_ns;
#pragma unroll
    for (int j = 0; j < K::NITEM; ++j) {
      const int item = tid + 256 * j, slot = item >> 2, cq = item & 3;
      const int r_ = slot / K::EC, c_ = slot - r_ * K::EC;
      const int iy = min(max(iy0 + r_, 0), p.H - 1), ix = min(max(ix0 + c_, 0), p.W - 1);
      ar[j] = *(const f32x4*)(ain + ((long)iy * p.W + ix) * p.a_ld + 4 * cq);
    }
  };

  const int wofs = l15 * 32 + 8 * q;
  // this lane's A-fragment addresses: slab s covers taps 2 s and 2 s + 1; the lane's tap = 2 s + (q >> 1), channels 8 (q & 1) .. + 7
  // of output pixel (wave, l15): input pixel (2 wave + ky, 2 l15 + kx) -> slot (row, even / odd run of the column)
  int aoff[K::NSLAB];
#pragma unroll
  for (int s = 0; s < K::NSLAB; ++s) {
    const int tap = 2 * s + (q >> 1);
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int slot = (2 * wave + ky) * K::EC + (kx == 1 ? 17 + l15 : l15 + (kx >> 1));
    aoff[s] = (tap < 9 ? slot : K::NSLOT) * K::LDS_A + 8 * (q & 1);
  }
  // stem_3's weights [2 slabs][3 planes][32 co][32 k] sit in LDS for the whole launch (12 KB; the registers go to stem_2b's)
  unsigned short* W3l = Dl + 3 * DPL;
  for (int i = tid; i < K::W3L / 8; i += 256) *(u32x4*)(W3l + 8 * i) = ldg16(wb + 2 * K::O_W3 + 8 * i);
  auto w3frag = [&](int s, int nt) {
    const unsigned short* b = W3l + (s * 3 * 32 + 16 * nt) * 32 + wofs;
    fp_frag3 f;
    f.h = *(const u32x4*)b, f.m = *(const u32x4*)(b + 1024), f.l = *(const u32x4*)(b + 2048);
    return f;
  };
  const f32x4 sc2[2] = {*(const f32x4*)(p.w + K::O_AFF2B + 4 * q), *(const f32x4*)(p.w + K::O_AFF2B + 16 + 4 * q)};
  const f32x4 bi2[2] = {*(const f32x4*)(p.w + K::O_AFF2B + 32 + 4 * q), *(const f32x4*)(p.w + K::O_AFF2B + 48 + 4 * q)};
  const f32x4 sc3[2] = {*(const f32x4*)(p.w + K::O_AFF3 + 4 * q), *(const f32x4*)(p.w + K::O_AFF3 + 16 + 4 * q)};
  const f32x4 bi3[2] = {*(const f32x4*)(p.w + K::O_AFF3 + 32 + 4 * q), *(const f32x4*)(p.w + K::O_AFF3 + 48 + 4 * q)};

  // ... and so do stem_2b's: [5 slabs][2 channel tiles] = 120 registers.  (Fetched from L2 per tile they cost more than the MFMAs:
  // 30 x 1 KiB per wave and tile through the CU's one texture path, 12.6 GB for 256 images against 3.4 GB of HBM traffic.)
  fp_frag3 w2[K::NSLAB][2];
#pragma unroll
  for (int s = 0; s < K::NSLAB; ++s) {
    const unsigned short* b = wb + 2 * K::O_W2B + (long)s * 3 * 1024 + wofs;
    w2[s][0].h = ldg16(b), w2[s][0].m = ldg16(b + 1024), w2[s][0].l = ldg16(b + 2048);
    w2[s][1].h = ldg16(b + 512), w2[s][1].m = ldg16(b + 1024 + 512), w2[s][1].l = ldg16(b + 2048 + 512);
  }

  const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int t_begin = (int)blockIdx.x * per, t_end = min(t_begin + per, p.ntiles);
  f32x4 ar[K::NITEM];
  if (t_begin < t_end) load_a(t_begin, ar);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int img = tile / p.tiles_per_img, tt = tile - img * p.tiles_per_img;
    const int ty0 = (tt / p.tiles_x) * K::TH, tx0 = (tt % p.tiles_x) * K::TW;
    const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;

    __syncthreads();                                       // every wave is done with the previous tile's A planes
    // ---- A: split the tile's input pixels into the three planes ----
#pragma unroll
    for (int j = 0; j < K::NITEM; ++j) {
      const int item = tid + 256 * j, slot = item >> 2, cq = item & 3;
      if (slot < K::NSLOT) {
        const int r_ = slot / K::EC, c_ = slot - r_ * K::EC;
        const bool in = (unsigned)(iy0 + r_) < (unsigned)p.H && (unsigned)(ix0 + c_) < (unsigned)p.W;
        const f32x4 v = in ? ar[j] : z;
        unsigned h0, m0, l0, h1, m1, l1;
        fp_split_pair(v[0], v[1], h0, m0, l0);
        fp_split_pair(v[2], v[3], h1, m1, l1);
        const int es = r_ * K::EC + ((c_ & 1) ? 17 + (c_ >> 1) : (c_ >> 1));
        unsigned short* dst = Al + es * K::LDS_A + 4 * cq;
        *(u32x2*)dst = u32x2{h0, h1};
        *(u32x2*)(dst + APL) = u32x2{m0, m1};
        *(u32x2*)(dst + 2 * APL) = u32x2{l0, l1};
      }
    }
    // the pooled pixel of this lane's output pixel (slab 1 of stem_3's K), and the next tile's input
    const int oy = ty0 + wave, ox = tx0 + l15;
    const long opix = (long)min(oy, p.OH - 1) * p.OW + min(ox, p.OW - 1);
    const float* pp = p.pool + (long)img * p.p_ns + opix * p.p_ld + 8 * q;
    const f32x4 pa = *(const f32x4*)pp, pb = *(const f32x4*)(pp + 4);
    if (tile + 1 < t_end) load_a(tile + 1, ar);
    __syncthreads();                                       // planes complete

    // ---- stem_2b: this wave's output row, both channel tiles ----
    f32x4 acc0 = z, acc1 = z;
#pragma unroll
    for (int s = 0; s < K::NSLAB; ++s) {
      const unsigned short* as = Al + aoff[s];
      const u32x4 ah = *(const u32x4*)as, am = *(const u32x4*)(as + APL), al = *(const u32x4*)(as + 2 * APL);
      fp_mfma_x6_2a(w2[s][0], w2[s][1], ah, am, al, acc0, acc1);
    }
    // BN + SiLU -> split -> this wave's 16 pixels of the D tile: lane (pixel l15, channels 16 nt + 4 q ..) -> [pixel][channel]
    {
      f32x4 v0 = acc0 * sc2[0] + bi2[0], v1 = acc1 * sc2[1] + bi2[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) v0[e] = fp_silu(v0[e]), v1[e] = fp_silu(v1[e]);
      unsigned short* dst = Dl + (16 * wave + l15) * LDA + 4 * q;
      unsigned h0, m0, l0, h1, m1, l1;
      fp_split_pair(v0[0], v0[1], h0, m0, l0);
      fp_split_pair(v0[2], v0[3], h1, m1, l1);
      *(u32x2*)dst = u32x2{h0, h1};
      *(u32x2*)(dst + DPL) = u32x2{m0, m1};
      *(u32x2*)(dst + 2 * DPL) = u32x2{l0, l1};
      fp_split_pair(v1[0], v1[1], h0, m0, l0);
      fp_split_pair(v1[2], v1[3], h1, m1, l1);
      *(u32x2*)(dst + 16) = u32x2{h0, h1};
      *(u32x2*)(dst + 16 + DPL) = u32x2{m0, m1};
      *(u32x2*)(dst + 16 + 2 * DPL) = u32x2{l0, l1};
    }
    // (the D tile rows 16 wave .. + 15 are written and read by this wave only: program order, no barrier)
    // ---- stem_3: K = [b | p] ----
    f32x4 o0 = z, o1 = z;
    {
      const unsigned short* src = Dl + (16 * wave + l15) * LDA + 8 * q;
      const u32x4 dh = *(const u32x4*)src, dm = *(const u32x4*)(src + DPL), dl = *(const u32x4*)(src + 2 * DPL);
      fp_mfma_x6_2a(w3frag(0, 0), w3frag(0, 1), dh, dm, dl, o0, o1);
      const fp_frag3 pf = fp_split8(pa, pb);
      fp_mfma_x6_2a(w3frag(1, 0), w3frag(1, 1), pf.h, pf.m, pf.l, o0, o1);
    }
    if (oy < p.OH && ox < p.OW) {
      float* o = p.out + (long)img * p.out_ns + ((long)oy * p.OW + ox) * p.out_ld + 4 * q;
      f32x4 v0 = o0 * sc3[0] + bi3[0], v1 = o1 * sc3[1] + bi3[1];
#pragma unroll
      for (int e = 0; e < 4; ++e) v0[e] = fp_silu(v0[e]), v1[e] = fp_silu(v1[e]);
      *(f32x4*)o = v0;
      *(f32x4*)(o + 16) = v1;
    }
  }
}

}  // namespace

// FP_OP_YSTEM2: in = a (H x W even, 16 channels), res = the pooled map (OH x OW, 32 channels), out 32 channels, 3x3 stride 2 pad 1.
bool fp_ystem2_supported(const fp_op& op) {
  if (op.kind != FP_OP_YSTEM2 || op.flags != FP_OPF_SPLIT3) return false;
  if (op.Cin != 16 || op.Cout != 32 || op.res_C != 32 || op.Cmid != 0) return false;
  if (op.KH != 3 || op.KW != 3 || op.stride != 2 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.H % 2 || op.W % 2 || op.OH != op.H / 2 || op.OW != op.W / 2 || op.res_H != op.OH || op.res_W != op.OW || op.out_cmul != 1) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns % 4 || op.w_off % 4) return false;
  if (op.res_ld % 4 || op.res_off % 4 || op.res_ns % 4 || op.res_ld < 32) return false;
  if (op.in_ld < 16 || op.out_ld < 32 || op.in_ns < (long)op.H * op.W * op.in_ld || op.out_ns < (long)op.OH * op.OW * op.out_ld ||
      op.res_ns < (long)op.OH * op.OW * op.res_ld) return false;
  if (op.act != FP_ACT_SILU || op.act2 != FP_ACT_SILU || op.res_mode != FP_RES_NONE) return false;
  const long tiles = (long)op.N * ((op.OH + 3) / 4) * ((op.OW + 15) / 16);
  return tiles > 0 && tiles < (1L << 31);
}

long fp_ystem2_w_floats(const fp_op&) { return YS2::TOTAL; }

int fp_launch_ystem2(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_ystem2_supported(op)) return FP_ERR_UNSUPPORTED;
  YStem2Args a;
  memset(&a, 0, sizeof(a));
  a.a = arena + op.in_off;
  a.pool = arena + op.res_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW;
  a.a_ld = op.in_ld; a.p_ld = op.res_ld; a.out_ld = op.out_ld;
  a.a_ns = op.in_ns; a.p_ns = op.res_ns; a.out_ns = op.out_ns;
  a.tiles_x = (op.OW + 15) / 16;
  a.tiles_per_img = a.tiles_x * ((op.OH + 3) / 4);
  a.ntiles = op.N * a.tiles_per_img;
  const hipError_t ae = hipFuncSetAttribute((const void*)ystem2_x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, YS2::LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  const int grid = a.ntiles < 512 ? a.ntiles : 512;       // persistent: two workgroups per CU, contiguous runs of tiles
  hipLaunchKernelGGL(ystem2_x6_kernel, dim3((unsigned)grid), dim3(256), YS2::LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
