// conv3.hip — dense 3x3 convolution (pad 1, stride 1 or 2) with the A operand read from an LDS image (gfx950).
//
// YOLOv5-face's 3x3 `Conv`s (fde/modules/yolov5_face/pytorch/models/common.py:39-55: Bottleneck.cv2 :76-87, the
// stride-2 downsampling convs of the yamls, StemBlock.stem_2b :58-73) as an implicit GEMM gather every input pixel
// nine times, one 16-byte piece per lane: conv_igemm_kernel ran them at 50-97 TFLOP/s while its 1x1 convs with the
// same K reach 105-113 (profiles/r01, r02).  Here a workgroup owns a SPATIAL tile of 8 x 16 output pixels and stages
// the input pixels its nine taps touch once per channel slab:
//   image     = [(8-1)*S+3][(16-1)*S+3] pixels x CK channels (+4 pad floats per pixel) in LDS, CK = 32 (S = 1) / 16
//               (S = 2); stride 2 keeps even and odd columns in separate planes so that the 32 pixels of an MFMA
//               fragment are consecutive in LDS for every tap (conflict-free ds_read_b128)
//   K order   = slab-major: for slab (CK channels) { for tap (9) { CK/2 MFMA k-steps } } -- the packed weights of
//               FP_OP_CONV (k = tap*Cin + c) are used as they are: chunk (tap, slab) is a contiguous run of k-quads
//   pipeline  = the next slab's pixels and the next (tap, slab) weight chunk are loaded into registers while the
//               current chunk's MFMAs run; weights are double-buffered in LDS: one barrier per tap
//   MFMA      = v_mfma_f32_32x32x2_f32, wave w owns output rows 2w, 2w+1 of the tile (32 pixels), NB n tiles of 32;
//               T16 variant: v_mfma_f32_16x16x4_f32 with NB n tiles of 16 columns, for widths whose padding to 32
//               would waste a quarter of the MFMA work (YOLOv5s' 48 -> 48 bottlenecks: 3 tiles of 16 instead of 64
//               columns).  Lane (r16, g) feeds A = pixel r16 of one tile row, k = 16*jj + 4*g + e -- one float4 of the
//               LDS pixel per 4 MFMAs -- and B = k-quad 4*jj + g of the packed weights, which is the FP_OP_CONV layout
//               as it is (the K permutation inside a 16-channel group is the same on both operands)
//   epilogue  = acc*scale+bias through LDS, SiLU / none, optional residual added after the activation (Bottleneck
//               shortcut), 16-byte stores into an arbitrary NHWC view (concat slices)
// Sums run in a different k order than conv_igemm_kernel (slab-major instead of tap-major): fp32 reassociation only.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 16;

struct Conv3Args {
  const float* in;
  float* out;
  const float* res;
  const float* w;
  const float* scale;
  const float* bias;
  int H, W, OH, OW, Cin, Cout, Npad, in_ld, out_ld, res_ld, act, has_res, tiles_x, tiles_per_img;
  long in_ns, out_ns, res_ns;
};

template <int S>
struct Geo {
  static constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
  static constexpr int CK = S == 1 ? 32 : 16;
  static constexpr int P = CK + 4;                         // floats per pixel in LDS
  static constexpr int PW = S == 1 ? IW : (IW + 1) / 2;    // pixels per (row, plane)
  static constexpr int NPIX = S == 1 ? IH * IW : IH * 2 * PW;
  static constexpr int F4_PER_PX = CK / 4;
  static constexpr int NSLOT = (IH * IW * F4_PER_PX + 255) / 256;
  __device__ static __forceinline__ int lds_px(int r, int c) {   // LDS pixel index of image pixel (r, c)
    return S == 1 ? r * IW + c : (r * 2 + (c & 1)) * PW + (c >> 1);
  }
};

template <int NB, int S, bool T16>
__global__ __launch_bounds__(256, 2) void conv3_kernel(Conv3Args p) {
  using G = Geo<S>;
  constexpr int BN = NB * (T16 ? 16 : 32), CK = G::CK, P = G::P;
  constexpr int LDO = BN + 4;
  constexpr int NWS = (CK / 4 * BN + 255) / 256;   // weight-chunk float4 slots per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Img = smem;                               // [NPIX][P]
  float* Bs = Img + G::NPIX * P;                   // [2][CK/4 + 1][BN][4]   (+1 zero quad for odd quad counts)
  constexpr int BQ = CK / 4 + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, h = lane >> 5;
  const int r16 = lane & 15, g = lane >> 4;        // T16 fragment coordinates
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

  const int tile = blockIdx.x, n0 = blockIdx.y * BN;
  const int img = tile / p.tiles_per_img, tin = tile - img * p.tiles_per_img;
  const int ty = tin / p.tiles_x, tx = tin - ty * p.tiles_x;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const float* ib = p.in + (long)img * p.in_ns;

  // ---- staging helpers ----
  f32x4 ireg[G::NSLOT];
  unsigned imask = 0;
  auto load_img = [&](int c0, int ck) {            // pixels of channel slab [c0, c0 + ck) -> registers
    imask = 0;
#pragma unroll
    for (int j = 0; j < G::NSLOT; ++j) {
      const int i = tid + 256 * j;
      const int px = i / G::F4_PER_PX, q = i - px * G::F4_PER_PX;
      const int r = px / G::IW, c = px - r * G::IW;
      const int iy = iy0 + r, ix = ix0 + c;
      const bool ok = px < G::IH * G::IW && q * 4 < ck && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
      ireg[j] = *(const f32x4*)(ib + ((long)cy * p.W + cx) * p.in_ld + c0 + min(q * 4, ck - 4));
      if (ok) imask |= 1u << j;
    }
  };
  auto store_img = [&]() {
#pragma unroll
    for (int j = 0; j < G::NSLOT; ++j) {
      const int i = tid + 256 * j;
      const int px = i / G::F4_PER_PX, q = i - px * G::F4_PER_PX;
      const int r = px / G::IW, c = px - r * G::IW;
      if (px < G::IH * G::IW) *(f32x4*)&Img[G::lds_px(r, c) * P + q * 4] = ((imask >> j) & 1u) ? ireg[j] : z4;
    }
  };
  f32x4 breg[NWS];
  unsigned bmask = 0;
  auto load_w = [&](int kq0, int nq) {             // nq k-quads starting at global quad kq0, columns n0 .. n0 + BN
    bmask = 0;
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
      const int idx = tid + 256 * j;
      const int q = idx / BN, col = idx - q * BN;
      breg[j] = *(const f32x4*)(p.w + ((long)(kq0 + min(q, nq - 1)) * p.Npad + min(n0 + col, p.Npad - 1)) * 4);
      if (q < nq && n0 + col < p.Npad) bmask |= 1u << j;
    }
  };
  auto store_w = [&](int buf) {
    float* b = Bs + buf * BQ * BN * 4;
#pragma unroll
    for (int j = 0; j < NWS; ++j) {
      const int idx = tid + 256 * j;
      if (idx < CK / 4 * BN) *(f32x4*)&b[idx * 4] = ((bmask >> j) & 1u) ? breg[j] : z4;
    }
  };

  // zero quad behind each weight buffer (read by the h = 1 half of the last k-pair when a slab has an odd quad count)
  for (int i = tid; i < 2 * BN; i += 256) *(f32x4*)&Bs[((i / BN) * BQ + CK / 4) * BN * 4 + (i % BN) * 4] = z4;
  // pad floats of every LDS pixel (read as A values beyond a short slab): zero once
  for (int i = tid; i < G::NPIX; i += 256) *(f32x4*)&Img[i * P + CK] = z4;

  f32x16 acc[T16 ? 1 : NB];
#pragma unroll
  for (int nb = 0; nb < (T16 ? 1 : NB); ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  f32x16 acc1;                                     // NB = 1: second partial sum (odd k-steps)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
  f32x4 acc16[2][T16 ? NB : 1];                    // T16: [tile row 2w + m][n tile of 16]
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int nb = 0; nb < (T16 ? NB : 1); ++nb) acc16[m][nb] = z4;

  // this lane's output pixel inside the tile: rows 2w, 2w+1; lr -> (y, x)
  const int py = 2 * wave + (lr >> 4), pxx = lr & 15;

  const int nslab = (p.Cin + CK - 1) / CK;
  load_img(0, min(CK, p.Cin));
  load_w(0, min(CK, p.Cin) / 4);
  int buf = 0;
  for (int s = 0; s < nslab; ++s) {
    const int c0 = s * CK, ck = min(CK, p.Cin - c0), nq = ck >> 2;
    if (s) __syncthreads();                        // every wave is done with the previous slab's image
    store_img();
    if (s + 1 < nslab) load_img(c0 + CK, min(CK, p.Cin - c0 - CK));
    for (int t = 0; t < 9; ++t) {
      store_w(buf);
      __syncthreads();                             // image (t = 0) and weight chunk visible; the other buffer is free
      {                                            // next chunk: (t + 1, s) or (0, s + 1)
        const int tn = t + 1 < 9 ? t + 1 : 0, sn = t + 1 < 9 ? s : s + 1;
        if (sn < nslab) load_w((tn * p.Cin + sn * CK) >> 2, min(CK, p.Cin - sn * CK) >> 2);
      }
      const int ky = t / 3, kx = t - ky * 3;
      const float* bb = Bs + buf * BQ * BN * 4;
      if constexpr (T16) {
        const float* arow0 = Img + G::lds_px((2 * wave) * S + ky, r16 * S + kx) * P + 4 * g;
        const float* arow1 = Img + G::lds_px((2 * wave + 1) * S + ky, r16 * S + kx) * P + 4 * g;
#pragma unroll
        for (int jj = 0; jj < CK / 16; ++jj) {
          if (jj * 4 < nq) {
            const f32x4 a0 = *(const f32x4*)(arow0 + jj * 16), a1 = *(const f32x4*)(arow1 + jj * 16);
            f32x4 b[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)&bb[((jj * 4 + g) * BN + nb * 16 + r16) * 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                acc16[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b[nb][e], acc16[0][nb], 0, 0, 0);
                FP_MFMA_ORDER();
                acc16[1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], b[nb][e], acc16[1][nb], 0, 0, 0);
                FP_MFMA_ORDER();
              }
            }
          }
        }
      } else {
      const float* arow = Img + G::lds_px(py * S + ky, pxx * S + kx) * P + 4 * h;
#pragma unroll
      for (int kq = 0; kq < CK / 8; ++kq) {
        if (kq * 2 < nq) {
          const f32x4 a = *(const f32x4*)(arow + kq * 8);
          f32x4 b[NB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b[nb] = *(const f32x4*)&bb[((kq * 2 + h) * BN + nb * 32 + lr) * 4];
          // MFMAs round-robin over the accumulators: a 32x32x2 f32 MFMA that accumulates into the previous one's
          // result issues at half rate (common.h FP_MFMA_ORDER); NB = 1 alternates two partial sums instead
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (NB == 1) {
              if (e & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[0][e], acc1, 0, 0, 0);
              else acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[0][e], acc[0], 0, 0, 0);
              FP_MFMA_ORDER();
            } else {
#pragma unroll
              for (int nb = 0; nb < NB; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[nb][e], acc[nb], 0, 0, 0);
                FP_MFMA_ORDER();
              }
            }
          }
        }
      }
      }
      buf ^= 1;
    }
  }
  if (NB == 1 && !T16) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] += acc1[r];
  }
  __syncthreads();                                 // all MFMAs done: LDS becomes the output staging tile

  // ---- epilogue: two passes of 64 rows (waves 2p, 2p+1), [64][LDO] staging ----
  float sc[NB], bi[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = T16 ? n0 + nb * 16 + r16 : n0 + nb * 32 + lr;
    const int nn = n < p.Cout ? n : 0;
    sc[nb] = p.scale ? p.scale[nn] : 1.f;
    bi[nb] = p.bias ? p.bias[nn] : 0.f;
  }
  float* Ot = smem;
  constexpr int F4_PER_ROW = BN / 4;
  constexpr int NIT = 64 * F4_PER_ROW / 256;       // BN / 16 float4 per thread and pass
  const bool silu = p.act == FP_ACT_SILU;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass) __syncthreads();
    if ((wave >> 1) == pass) {
      const int wrow = (wave & 1) * 32;
      if constexpr (T16) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)      // 16x16x4 C/D: column = lane & 15, row = 4 * (lane >> 4) + reg
              Ot[(wrow + m * 16 + 4 * g + reg) * LDO + nb * 16 + r16] = acc16[m][nb][reg] * sc[nb] + bi[nb];
      } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int row = wrow + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            Ot[row * LDO + nb * 32 + lr] = acc[nb][reg] * sc[nb] + bi[nb];
          }
      }
    }
    __syncthreads();
    f32x4 rr[NIT];
    long ooff[NIT];
    bool ok[NIT];
#pragma unroll
    for (int j = 0; j < NIT; ++j) {                 // addresses and every residual load of the pass first
      const int f = tid + 256 * j;
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      const int prow = pass * 64 + row;             // pixel index in the tile: wave-major (wave w: rows 2w, 2w+1)
      const int y = oy0 + (prow >> 4), x = ox0 + (prow & 15);
      const int n = n0 + c4 * 4;
      ok[j] = y < p.OH && x < p.OW && n < p.Cout;
      const long pix = (long)y * p.OW + x;
      ooff[j] = (long)img * p.out_ns + pix * p.out_ld + n;
      rr[j] = z4;
      if (p.has_res && ok[j]) rr[j] = *(const f32x4*)(p.res + (long)img * p.res_ns + pix * p.res_ld + n);
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int f = tid + 256 * j;
      const int row = f / F4_PER_ROW, c4 = f - row * F4_PER_ROW;
      if (!ok[j]) continue;
      const f32x4 v = *(const f32x4*)&Ot[row * LDO + c4 * 4];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (silu ? fp_silu(v[e]) : v[e]) + rr[j][e];
      *(f32x4*)(p.out + ooff[j]) = o;
    }
  }
}

template <int NB, int S, bool T16>
size_t conv3_lds_bytes() {
  using G = Geo<S>;
  constexpr int BN = NB * (T16 ? 16 : 32);
  const size_t main_b = 4 * ((size_t)G::NPIX * G::P + 2 * (size_t)(G::CK / 4 + 1) * BN * 4);
  const size_t epi_b = 4 * (size_t)64 * (BN + 4);
  return main_b > epi_b ? main_b : epi_b;
}

}  // namespace

// Dense 3x3, pad 1, stride 1 / 2, 16-byte-aligned NHWC views, act none / SiLU, residual none / add-after-act.
bool fp_conv3_eligible(const fp_op& op) {
  if (op.kind != FP_OP_CONV || op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.stride != 1 && op.stride != 2) return false;
  if (op.OH != (op.H + 2 - 3) / op.stride + 1 || op.OW != (op.W + 2 - 3) / op.stride + 1) return false;
  if (op.Cin % 4 || op.Cin < 8 || op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4) return false;
  if (op.Cout % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns % 4 || op.out_cmul != 1 || op.w_off % 4) return false;
  if (op.act != FP_ACT_NONE && op.act != FP_ACT_SILU) return false;
  if (op.res_mode != FP_RES_NONE && op.res_mode != FP_RES_ADD_AFTER_ACT) return false;
  if (op.res_mode != FP_RES_NONE && (op.res_C < op.Cout || op.res_ld % 4 || op.res_off % 4 || op.res_ns % 4)) return false;
  if ((op.scale_off >= 0 && op.scale_off % 4) || (op.bias_off >= 0 && op.bias_off % 4)) return false;
  // Where it pays (measured against conv_igemm_kernel at batch 256, profiles/r02_yolo_*): stride 1 on maps of 32 x 32
  // and more (1.06-1.37x; 20 x 20 maps leave half of the 8 x 16 tiles empty).  Stride 2 reuses a staged pixel only
  // 2.25 times instead of 9, so it wins only for wide inputs with a single 128-column chunk (128 -> 128: 1.10x).
  if (op.OH < 32 || op.OW < 32) return false;
  if (op.stride == 2 && (op.Cin < 128 || op.Cout > 128)) return false;
  return (long)op.N * fp_ceil_div(op.OH, TH) * fp_ceil_div(op.OW, TW) >= 256;
}

// 16-column n tiles (v_mfma_f32_16x16x4_f32), stride 1, one chunk.  Measured at batch 256 (gpurun_out/r2_yp_m*.log,
// summarised in DESIGN.md section 6): 48 -> 48 @ 80 x 80 1070 -> 636 us and 64 -> 64 @ 80 x 80 1282 -> 1041 us (117
// TFLOP/s) against the 32 x 32 x 2 tiles; 24 -> 24 (2 tiles), 92 -> 92 (6 tiles) and the stride-2 128 -> 128 (8 tiles)
// are 3-20 % slower that way and stay on 32-column tiles.
bool fp_conv3_t16(const fp_op& op) {
  if (op.stride != 1 || op.Cout > 112) return false;
  if (op.Cout > 32 && op.Cout <= 64) return true;                           // 3 or 4 tiles of 16
  return fp_round_up(op.Cout, 16) < fp_round_up(op.Cout, 32);               // a whole idle 16-column tile otherwise
}

int fp_conv3_nb(const fp_op& op) {
  if (fp_conv3_t16(op)) return (int)fp_round_up(op.Cout, 16) / 16;          // 1, 3, 4, 5 or 7 tiles of 16
  const int nblk = (int)fp_round_up(op.Cout, 32) / 32;
  return nblk >= 4 ? 4 : nblk;   // wider outputs run as several 128-column chunks (grid.y)
}

int fp_launch_conv3(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  Conv3Args a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = weights + op.w_off;
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.Cin = op.Cin; a.Cout = op.Cout;
  a.Npad = (int)fp_round_up(op.Cout, 32);
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.act = op.act;
  a.has_res = op.res_mode != FP_RES_NONE;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns; a.res_ns = op.res_ns;
  a.tiles_x = fp_ceil_div(op.OW, TW);
  a.tiles_per_img = a.tiles_x * fp_ceil_div(op.OH, TH);
  const int NB = fp_conv3_nb(op);
  const bool t16 = fp_conv3_t16(op);
  const dim3 grid((unsigned)((long)op.N * a.tiles_per_img), (unsigned)(t16 ? fp_ceil_div((int)fp_round_up(op.Cout, 16), NB * 16) : fp_ceil_div(a.Npad, NB * 32))), block(256);
  hipError_t ae = hipSuccess;
#define FP_CONV3_CASE(NBV, SV, T16V)                                                                               \
  {                                                                                                                \
    const size_t lds = conv3_lds_bytes<NBV, SV, T16V>();                                                           \
    ae = hipFuncSetAttribute((const void*)conv3_kernel<NBV, SV, T16V>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                             (int)lds);                                                                            \
    if (ae == hipSuccess) hipLaunchKernelGGL((conv3_kernel<NBV, SV, T16V>), grid, block, lds, s, a);               \
  }
  if (t16) {
    switch (NB) {
      case 1: FP_CONV3_CASE(1, 1, true) break;
      case 3: FP_CONV3_CASE(3, 1, true) break;
      case 4: FP_CONV3_CASE(4, 1, true) break;
      case 5: FP_CONV3_CASE(5, 1, true) break;
      default: FP_CONV3_CASE(7, 1, true) break;
    }
  } else if (op.stride == 1) {
    switch (NB) {
      case 1: FP_CONV3_CASE(1, 1, false) break;
      case 2: FP_CONV3_CASE(2, 1, false) break;
      case 3: FP_CONV3_CASE(3, 1, false) break;
      default: FP_CONV3_CASE(4, 1, false) break;
    }
  } else {
    switch (NB) {
      case 1: FP_CONV3_CASE(1, 2, false) break;
      case 2: FP_CONV3_CASE(2, 2, false) break;
      case 3: FP_CONV3_CASE(3, 2, false) break;
      default: FP_CONV3_CASE(4, 2, false) break;
    }
  }
#undef FP_CONV3_CASE
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}
