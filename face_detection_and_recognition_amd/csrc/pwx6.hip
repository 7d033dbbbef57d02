// pwx6.hip — pointwise (1x1) convolution on the bf16 matrix cores with fp32-equivalent arithmetic (split.h) — gfx950.
//
// The 1x1 convs with K >= 128 input channels (YOLOv5-face's C3 / ShuffleV2 / PAN convs, y5/models/common.py:35-56,127-176;
// Mobile-FaceNet's conv_6_sep, mobile_facenet.py:131) are bound by the fp32 MFMA in conv_igemm_kernel (~100 TFLOP/s =
// 65 % of the 157 TFLOP/s fp32 matrix peak, which is the fp32 VECTOR rate, FINDINGS.md finding 18).  Here the fp32 operands are
// split exactly into three bf16 pieces each and multiplied as six bf16 MFMAs per product (fp32 accumulation, csrc/split.h):
//   * a workgroup = 4 waves owns 256 consecutive rows (pixels) x one chunk of <= 128 output channels; a wave owns 64 rows
//     (four 16-row MFMA tiles) for ALL of the chunk's columns: accumulators 4 x 8 x 4 registers, A split ONCE per element;
//   * K runs in slabs of 32: a wave's A slab comes straight from global memory into registers (lane = (row, 8 consecutive k):
//     two 16-byte loads), is split there (5.5 VALU per element, beside the MFMAs) -- no LDS for A;
//   * the weight slab [3 planes][chunk columns][32 k] bf16 (pre-split on the host, plan.py) is staged by LDS-DMA, double
//     buffered, one workgroup barrier per slab; every wave reads each 16-column fragment once per slab and uses it for its
//     four row tiles (24 MFMAs per 3 ds_read_b128);
//   * operands swapped (D^T = W^T A^T): a lane ends up with 4 consecutive channels of ONE pixel -- 16-byte epilogue
//     (scale / bias, residual, ReLU / PReLU / SiLU, ShuffleV2's interleaved store) straight from the accumulators.
// Eligibility (fp_pwx6_eligible) mirrors plan.py's PlanBuilder.pwx6_ok: the op carries FP_OPF_SPLIT3 and split weights.
#include <stdlib.h>
#include <string.h>

#include "split.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

struct PwX6Args {
  const float* in;
  float* out;
  const float* res;
  const unsigned short* w;   // [K / 32][3][N][32] bf16
  const float* scale;
  const float* bias;
  const float* slope;
  long M;                    // rows = N * H * W
  int K, N, in_ld, out_ld, res_ld, res_C, act, res_mode;
  // FP_OPF_IN_UP2: input channels [0, 32 * up_slabs) are the nearest-neighbour 2x upsampling of `up` (an H/2 x W/2 map)
  const float* up;
  long up_ns;
  int up_ld, up_slabs, W2, HW;   // W2 = W / 2, HW = H * W
  fp_divisor hw_div, w_div;  // H * W, W
};

constexpr int MT = 4;        // 16-row tiles per wave (convx6_kernel; pwx6_kernel's default)
constexpr int BM = 4 * MT * 16;

// NT16 = 16-column tiles of a column chunk (3: N = 48, 4: N = 64, 8: chunks of 128); MT_ = 16-row tiles per wave: 4 (256-row
// workgroup tiles, two workgroups per CU) or 2 (128-row tiles, three per CU) -- the launcher takes the small form when the
// large one would leave the last round of workgroups mostly empty (409 600 rows = 1600 large tiles on 512 slots = 3.1 rounds)
// UP: the op carries FP_OPF_IN_UP2 -- the first up_slabs K slabs of a row come from the pixel (y / 2, x / 2) of the half-size
// map `up` (nn.Upsample(scale_factor=2, mode="nearest") + Concat of y5/models/yolo.py's head folded into the operand
// addressing: the upsampled tensor is never written, and the small map is read from L2 four times instead of the large one
// from HBM once).
template <int NT16, int MT_, bool UP>
__global__ __launch_bounds__(256, MT_ == 4 ? 2 : 3) void pwx6_kernel(PwX6Args p) {
  constexpr int MT = MT_, BM = 4 * MT_ * 16;
  constexpr int NC = NT16 * 16;
  constexpr int SLAB = 3 * NC * 32;                      // bf16 elements of a weight slab
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Bl = (unsigned short*)smem_raw;        // [2][3][NC][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nchunk = p.N / NC;
  const unsigned vb = fp_xcd_block();                    // the column chunks of one row tile run on ONE XCD, next to each other: A re-read from its L2
  const int chunk = vb % nchunk;
  const long row0 = (long)(vb / nchunk) * BM + wave * (MT * 16);
  const int c0 = chunk * NC;
  const int KS = p.K / 32;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  // weight slab ks -> Bl[ks & 1]: three planes of NC x 64 bytes, 1 KiB per wave and instruction
  auto stage = [&](int ks) {
    unsigned char* dst = (unsigned char*)(Bl + (ks & 1) * SLAB);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned char* src = (const unsigned char*)(p.w + ((long)(ks * 3 + pl) * p.N + c0) * 32) + lane * 16;
#pragma unroll
      for (int j = 0; j < (NT16 + 3) / 4; ++j) {
        const int c = j * 4 + wave;                      // 1-KiB piece = 16 columns
        if (c < NT16)
          __builtin_amdgcn_global_load_lds((gbl_ptr)(src + c * 1024), (lds_ptr)(dst + (pl * NC * 32 + c * 512) * 2), 16, 0, 0);
      }
    }
  };

  // this lane's rows: tile t -> row row0 + 16 t + l15 (clamped), k = 32 ks + 8 q .. + 7
  const float* arow[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    long r = row0 + 16 * t + l15;
    r = r < p.M ? r : p.M - 1;
    arow[t] = p.in + r * p.in_ld + 8 * q;
  }
  const float* urow[UP ? MT : 1];
  if constexpr (UP) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      long r = row0 + 16 * t + l15;
      r = r < p.M ? r : p.M - 1;
      const unsigned img = fp_fastdiv((unsigned)r, p.hw_div);
      const unsigned pix = (unsigned)r - img * (unsigned)p.HW;
      const unsigned y = fp_fastdiv(pix, p.w_div), x = pix - y * (unsigned)(2 * p.W2);
      urow[t] = p.up + (long)img * p.up_ns + (long)((y >> 1) * p.W2 + (x >> 1)) * p.up_ld + 8 * q;
    }
  }
  // A slabs in flight: one ahead (large tiles: no registers for more) or two ahead (small tiles: a slab's MFMAs, 0.85 us, do not
  // cover an HBM round trip).  The slab loop is written out for two buffers so that the register arrays are indexed statically.
  constexpr int AD = MT == 2 ? 2 : 1;
  f32x4 araw[AD][MT][2];
  auto load_a = [&](int ks, int buf) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const float* a = arow[t];
      if constexpr (UP) a = ks < p.up_slabs ? urow[t] : a;
      araw[buf][t][0] = *(const f32x4*)(a + 32 * ks);
      araw[buf][t][1] = *(const f32x4*)(a + 32 * ks + 4);
    }
  };

  f32x4 acc[MT][NT16];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT16; ++n) acc[t][n] = z;

  auto slab = [&](int ks, int buf) {
    fp_frag3 af[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) af[t] = fp_split8(araw[buf][t][0], araw[buf][t][1]);
    __syncthreads();   // slab ks landed (own DMA waited for, then everybody's); every wave is done with slab ks - 1
    if (AD == 1) {
      if (ks + 1 < KS) {
        stage(ks + 1);
        load_a(ks + 1, 0);
      }
    } else {
      if (ks + 1 < KS) stage(ks + 1);
      if (ks + AD < KS) load_a(ks + AD, buf);
    }
    const unsigned short* Bc = Bl + (ks & 1) * SLAB + (l15 * 32 + 8 * q);
    fp_frag3 bf[2];
    auto ldb = [&](int n, fp_frag3& b) {
      b.h = *(const u32x4*)(Bc + n * 512);
      b.m = *(const u32x4*)(Bc + NC * 32 + n * 512);
      b.l = *(const u32x4*)(Bc + 2 * NC * 32 + n * 512);
    };
    ldb(0, bf[0]);
#pragma unroll
    for (int n = 0; n < NT16; ++n) {
      if (n + 1 < NT16) ldb(n + 1, bf[(n + 1) & 1]);
      const fp_frag3& b = bf[n & 1];
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][n] = fp_mfma_x6(b.h, b.m, b.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
    }
  };

  stage(0);
  load_a(0, 0);
  if constexpr (AD == 2) {
    if (KS > 1) load_a(1, 1);
    for (int ks = 0; ks < KS; ks += 2) {
      slab(ks, 0);
      if (ks + 1 < KS) slab(ks + 1, 1);
    }
  } else {
    // (the same slab step written in line: through the lambda above hipcc allocates five registers more and spills them)
    for (int ks = 0; ks < KS; ++ks) {
      fp_frag3 af[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) af[t] = fp_split8(araw[0][t][0], araw[0][t][1]);
      __syncthreads();
      if (ks + 1 < KS) {
        stage(ks + 1);
        load_a(ks + 1, 0);
      }
      const unsigned short* Bc = Bl + (ks & 1) * SLAB + (l15 * 32 + 8 * q);
      fp_frag3 bf[2];
      auto ldb = [&](int n, fp_frag3& b) {
        b.h = *(const u32x4*)(Bc + n * 512);
        b.m = *(const u32x4*)(Bc + NC * 32 + n * 512);
        b.l = *(const u32x4*)(Bc + 2 * NC * 32 + n * 512);
      };
      ldb(0, bf[0]);
#pragma unroll
      for (int n = 0; n < NT16; ++n) {
        if (n + 1 < NT16) ldb(n + 1, bf[(n + 1) & 1]);
        const fp_frag3& b = bf[n & 1];
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t][n] = fp_mfma_x6(b.h, b.m, b.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
      }
    }
  }

  // ---- epilogue: lane = row 16 t + l15, channels c0 + 16 n + 4 q .. + 3 ----
  const bool shuffle = p.res_mode == FP_RES_SHUFFLE2;
#pragma unroll
  for (int n = 0; n < NT16; ++n) {
    const int ch = c0 + 16 * n + 4 * q;
    const f32x4 one = {1.f, 1.f, 1.f, 1.f};
    const f32x4 sc = p.scale ? *(const f32x4*)(p.scale + ch) : one;
    const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + ch) : z;
    const f32x4 sl = p.act == FP_ACT_PRELU ? *(const f32x4*)(p.slope + ch) : z;
    f32x4 rv[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      long r = row0 + 16 * t + l15;
      r = r < p.M ? r : p.M - 1;
      rv[t] = z;
      if (p.res_mode != FP_RES_NONE) {
        // channels beyond res_C add 0 (res_C is a multiple of 4 for every eligible op)
        if (ch < p.res_C) rv[t] = *(const f32x4*)(p.res + r * p.res_ld + ch);
      }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const long r = row0 + 16 * t + l15;
      f32x4 v = acc[t][n] * sc + bi;
      if (p.res_mode == FP_RES_ADD_BEFORE_ACT) v += rv[t];
      if (p.act == FP_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
      } else if (p.act == FP_ACT_PRELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * sl[i];
      } else if (p.act == FP_ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fp_silu(v[i]);
      }
      if (p.res_mode == FP_RES_ADD_AFTER_ACT) v += rv[t];
      if (r < p.M) {
        if (shuffle) {   // out[2 c] = res[c], out[2 c + 1] = act(conv)[c]: two 16-byte pieces
          float* o = p.out + r * p.out_ld + 2 * ch;
          *(f32x4*)o = f32x4{rv[t][0], v[0], rv[t][1], v[1]};
          *(f32x4*)(o + 4) = f32x4{rv[t][2], v[2], rv[t][3], v[3]};
        } else {
          *(f32x4*)(p.out + r * p.out_ld + ch) = v;
        }
      }
    }
  }
}

// rounds of workgroups the large tiles need (512 slots): below 8, and with a last round less than 70 % full -> small tiles
// K at or below which the small tiles are taken regardless of the round count (0 = never): with two or four K slabs a tile is
// one HBM round trip + its MFMAs, and three co-resident workgroups with every slab of A in flight hide more of it than two.
static int pwx6_small_maxk() { return fp_get_knobs().pwx6_small_maxk; }

bool pwx6_small_tiles(long M, int nchunk) {
  const double rounds = (double)((M + BM - 1) / BM * nchunk) / 512.0;
  const double frac = rounds - (double)(long)rounds;
  return rounds < 8.0 && frac > 0.0 && frac < 0.7;
}

template <int NT16>
int launch(const PwX6Args& a, hipStream_t s) {
  constexpr int lds = 2 * 3 * NT16 * 16 * 32 * 2;
  const long nchunk = a.N / (NT16 * 16);
  const long big = (a.M + BM - 1) / BM * nchunk;
  if (2 * big >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  const bool up = a.up != nullptr;
  // (the large-tile form with the second set of row pointers of an FP_OPF_IN_UP2 op needs 257 registers at 128-column chunks)
  if (pwx6_small_tiles(a.M, (int)nchunk) || (up && NT16 == 8) || a.K <= pwx6_small_maxk()) {
    const long tiles = (a.M + BM / 2 - 1) / (BM / 2) * nchunk;
    if (up) hipLaunchKernelGGL((pwx6_kernel<NT16, 2, true>), dim3((unsigned)tiles), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((pwx6_kernel<NT16, 2, false>), dim3((unsigned)tiles), dim3(256), lds, s, a);
  } else {
    if constexpr (NT16 != 8) {
      if (up) {
        hipLaunchKernelGGL((pwx6_kernel<NT16, 4, true>), dim3((unsigned)big), dim3(256), lds, s, a);
        FP_CHECK_LAUNCH();
        return FP_OK;
      }
    }
    hipLaunchKernelGGL((pwx6_kernel<NT16, 4, false>), dim3((unsigned)big), dim3(256), lds, s, a);
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// convx6_kernel: the same scheme for dense 3x3 convs (pad 1, stride 1 / 2: YOLOv5-face's Bottleneck.cv2 and downsampling
// Convs, y5/models/common.py:35-56,76-88) and for pointwise convs whose widths are not multiples of 32 / 16 (the in-tree
// yolov5s widths 92, 184, 360).  Rows are OUTPUT pixels; the K loop runs over (tap, 32-channel slab); a lane's A fragment
// for a slab is 8 consecutive channels of input pixel (oy*s + dy - pad, ox*s + dx - pad) -- two 16-byte loads straight from
// global memory (the nine taps of a pixel hit L1 / L2), zero outside the image or beyond Cin -- split in registers as
// above.  Weight slabs [tap * CS + cs][3][Npad][32] with zero rows / columns in the padding; outputs beyond Cout are not stored.
struct ConvX6Args {
  const float* in;
  float* out;
  const float* res;
  const unsigned short* w;
  const float* scale;
  const float* bias;
  const float* slope;
  long M;
  int H, W, OH, OW, Cin, Cout, Npad, KH, stride, pad;
  int in_ld, out_ld, res_ld, res_C, act, res_mode;
  long in_ns;
  // FP_OPF_IN_UP2 (1x1 only): channels [0, up_C) of input pixel (y, x) come from pixel (y / 2, x / 2) of `up` (up_C % 8 == 0:
  // a lane's eight channels come from one of the two tensors)
  const float* up;
  long up_ns;
  int up_ld, up_C, W2;
  // flat: Cin < 32 (a multiple of 8): K runs over the flattened (tap, channel) index in slabs of 32 -- two taps of a 16-channel
  // input per slab (YOLOv5n-face's stem_2b, 3x3 stride 2 on 16 channels: K = 144 in 5 slabs instead of 9 half-empty ones)
  int flat;
  fp_divisor div_cin;
  fp_divisor div_ohw, div_ow;
};

template <int NT16>
__global__ __launch_bounds__(256, 2) void convx6_kernel(ConvX6Args p) {
  constexpr int NC = NT16 * 16;
  constexpr int SLAB = 3 * NC * 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Bl = (unsigned short*)smem_raw;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int nchunk = p.Npad / NC;
  const unsigned vb = fp_xcd_block();
  const int chunk = vb % nchunk;
  const long row0 = (long)(vb / nchunk) * BM + wave * (MT * 16);
  const int c0 = chunk * NC;
  const int CS = (p.Cin + 31) / 32;
  const int NSL = p.flat ? (p.KH * p.KH * p.Cin + 31) / 32 : p.KH * p.KH * CS;   // slabs: (tap, channel slab), or 32 flattened k
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int sl) {
    unsigned char* dst = (unsigned char*)(Bl + (sl & 1) * SLAB);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const unsigned char* src = (const unsigned char*)(p.w + ((long)(sl * 3 + pl) * p.Npad + c0) * 32) + lane * 16;
#pragma unroll
      for (int j = 0; j < (NT16 + 3) / 4; ++j) {
        const int c = j * 4 + wave;
        if (c < NT16)
          __builtin_amdgcn_global_load_lds((gbl_ptr)(src + c * 1024), (lds_ptr)(dst + (pl * NC * 32 + c * 512) * 2), 16, 0, 0);
      }
    }
  };

  // this lane's output pixels: tile t -> row row0 + 16 t + l15 -> (n, oy, ox); kept as the float offset of input pixel
  // (oy*s - pad, ox*s - pad) of image n (may point outside the image: only dereferenced for valid taps) + iy0 / ix0
  int nimg[MT], iyx[MT];   // image index; (iy0 << 16) | (ix0 & 0xffff) with iy0 / ix0 = oy*s - pad / ox*s - pad (two ints, not
                           // a 64-bit address + two coordinates per tile: the 128-accumulator form has no registers to spare)
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    long r = row0 + 16 * t + l15;
    r = r < p.M ? r : p.M - 1;
    const unsigned n = fp_fastdiv((unsigned)r, p.div_ohw);
    const unsigned rem = (unsigned)r - n * (unsigned)(p.OH * p.OW);
    const unsigned oy = fp_fastdiv(rem, p.div_ow), ox = rem - oy * p.OW;
    nimg[t] = (int)n;
    iyx[t] = (((int)oy * p.stride - p.pad) << 16) | (((int)ox * p.stride - p.pad) & 0xffff);
  }
  f32x4 araw[MT][2];
  auto load_a = [&](int sl) {
    int tap, cs, k0;
    bool k_lo, k_hi;
    if (p.flat) {                      // per-lane tap: this lane's eight k are channels k0 .. k0 + 7 of tap (32 sl + 8 q) / Cin
      const unsigned kf = 32u * sl + 8u * q;
      tap = (int)fp_fastdiv(kf, p.div_cin);
      k0 = (int)kf - tap * p.Cin;
      cs = 0;
      k_lo = k_hi = tap < p.KH * p.KH;
    } else {
      tap = sl / CS;
      cs = sl - tap * CS;
      k0 = 32 * cs + 8 * q;
      k_lo = k0 < p.Cin;
      k_hi = k0 + 4 < p.Cin;
    }
    const int dy = (tap >= p.KH) + (tap >= 2 * p.KH), dx = tap - dy * p.KH;     // KH in {1, 3}
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int iy = (iyx[t] >> 16) + dy, ix = (int)(short)(iyx[t] & 0xffff) + dx;
      const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const float* src = p.in + ((long)nimg[t] * p.in_ns + (long)((iy * p.W + ix) * p.in_ld + k0));
      if (p.up && k0 < p.up_C)           // the folded nn.Upsample: this lane's channels live in the half-size map
        src = p.up + ((long)nimg[t] * p.up_ns + (long)(((iy >> 1) * p.W2 + (ix >> 1)) * p.up_ld + k0));
      araw[t][0] = ok && k_lo ? *(const f32x4*)src : z;
      araw[t][1] = ok && k_hi ? *(const f32x4*)(src + 4) : z;
    }
  };

  f32x4 acc[MT][NT16];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int n = 0; n < NT16; ++n) acc[t][n] = z;

  stage(0);
  load_a(0);
  for (int sl = 0; sl < NSL; ++sl) {
    fp_frag3 af[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) af[t] = fp_split8(araw[t][0], araw[t][1]);
    __syncthreads();
    if (sl + 1 < NSL) {
      stage(sl + 1);
      load_a(sl + 1);
    }
    const unsigned short* Bc = Bl + (sl & 1) * SLAB + (l15 * 32 + 8 * q);
    fp_frag3 bf[2];
    auto ldb = [&](int n, fp_frag3& b) {
      b.h = *(const u32x4*)(Bc + n * 512);
      b.m = *(const u32x4*)(Bc + NC * 32 + n * 512);
      b.l = *(const u32x4*)(Bc + 2 * NC * 32 + n * 512);
    };
    ldb(0, bf[0]);
#pragma unroll
    for (int n = 0; n < NT16; ++n) {
      if (n + 1 < NT16) ldb(n + 1, bf[(n + 1) & 1]);
      const fp_frag3& b = bf[n & 1];
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t][n] = fp_mfma_x6(b.h, b.m, b.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
    }
  }

  const bool shuffle = p.res_mode == FP_RES_SHUFFLE2;
#pragma unroll
  for (int n = 0; n < NT16; ++n) {
    const int ch = c0 + 16 * n + 4 * q;
    if (ch < p.Cout) {
      const f32x4 one = {1.f, 1.f, 1.f, 1.f};
      const f32x4 sc = p.scale ? *(const f32x4*)(p.scale + ch) : one;
      const f32x4 bi = p.bias ? *(const f32x4*)(p.bias + ch) : z;
      const f32x4 sl = p.act == FP_ACT_PRELU ? *(const f32x4*)(p.slope + ch) : z;
      f32x4 rv[MT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        long r = row0 + 16 * t + l15;
        r = r < p.M ? r : p.M - 1;
        rv[t] = z;
        if (p.res_mode != FP_RES_NONE && ch < p.res_C) rv[t] = *(const f32x4*)(p.res + r * p.res_ld + ch);
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const long r = row0 + 16 * t + l15;
        f32x4 v = acc[t][n] * sc + bi;
        if (p.res_mode == FP_RES_ADD_BEFORE_ACT) v += rv[t];
        if (p.act == FP_ACT_RELU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaxf(v[i], 0.f);
        } else if (p.act == FP_ACT_PRELU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * sl[i];
        } else if (p.act == FP_ACT_SILU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fp_silu(v[i]);
        }
        if (p.res_mode == FP_RES_ADD_AFTER_ACT) v += rv[t];
        if (r < p.M) {
          if (shuffle) {
            float* o = p.out + r * p.out_ld + 2 * ch;
            *(f32x4*)o = f32x4{rv[t][0], v[0], rv[t][1], v[1]};
            *(f32x4*)(o + 4) = f32x4{rv[t][2], v[2], rv[t][3], v[3]};
          } else {
            *(f32x4*)(p.out + r * p.out_ld + ch) = v;
          }
        }
      }
    }
  }
}

template <int NT16>
int launch_conv(const ConvX6Args& a, hipStream_t s) {
  constexpr int lds = 2 * 3 * NT16 * 16 * 32 * 2;
  const long tiles = (a.M + BM - 1) / BM * (a.Npad / (NT16 * 16));
  if (tiles >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL((convx6_kernel<NT16>), dim3((unsigned)tiles), dim3(256), lds, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// column tiling of a general Cout: (16-column tiles per chunk, padded width)
void general_tiles(int cout, int* nt16, int* npad) {
  const int nt = (cout + 15) / 16;
  int per;
  if (nt <= 2) per = 2;
  else if (nt <= 3) per = 3;
  else if (nt <= 4) per = 4;
  else if (nt <= 6) per = 6;
  else {   // chunks of 6 or 4 tiles, whichever pads less (8-tile chunks leave convx6_kernel no registers for its addressing)
    const int p6 = (nt + 5) / 6 * 6, p4 = (nt + 3) / 4 * 4;
    per = p6 <= p4 ? 6 : 4;
  }
  *nt16 = per;
  *npad = (nt + per - 1) / per * per * 16;
}

int chunk_tiles(int N) { return N == 48 ? 3 : N == 64 ? 4 : (N % 128 == 0 ? 8 : 0); }

}  // namespace

// A CONV that carries FP_OPF_SPLIT3: pointwise on dense rows, K a multiple of 32 (>= 64), Cout 48 / 64 / a multiple of 128,
// 16-byte aligned views, residual modes none / add before / add after / ShuffleV2 interleave.
bool fp_pwx6_eligible(const fp_op& op) {
  if (op.kind != FP_OP_CONV || !(op.flags & FP_OPF_SPLIT3) || (op.flags & ~(FP_OPF_SPLIT3 | FP_OPF_IN_UP2))) return false;
  if (op.KH != 1 || op.KW != 1 || op.stride != 1 || op.pad_t || op.pad_l || op.OH != op.H || op.OW != op.W) return false;
  if (op.flags & FP_OPF_IN_UP2) {
    // channels [0, res_C) of the input = the res view (H/2 x W/2, dense rows) upsampled 2x; no residual on such an op
    if (op.res_mode != FP_RES_NONE || op.res_C % 32 || op.res_C <= 0 || op.res_C >= op.Cin) return false;
    if (op.H % 2 || op.W % 2 || op.res_H != op.H / 2 || op.res_W != op.W / 2 || op.W < 2 || (long)op.H * op.W < 2) return false;
    if (op.res_ld % 4 || op.res_off % 4 || op.res_ns % 4 || op.res_ld < op.res_C || op.res_ns < (long)op.res_H * op.res_W * op.res_ld)
      return false;
    if ((long)op.N * op.H * op.W >= (1L << 31)) return false;
  }
  if (op.Cin % 32 || op.Cin < 64 || chunk_tiles(op.Cout) == 0 || op.out_cmul != 1) return false;
  const long HW = (long)op.H * op.W;
  if (op.in_ns != HW * op.in_ld || op.out_ns != HW * op.out_ld) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.out_ld % 4 || op.out_off % 4 || op.w_off % 4) return false;
  if ((op.scale_off >= 0 && op.scale_off % 4) || (op.bias_off >= 0 && op.bias_off % 4) || (op.slope_off >= 0 && op.slope_off % 4)) return false;
  if (op.act == FP_ACT_PRELU && op.slope_off < 0) return false;
  if (op.res_mode != FP_RES_NONE) {
    if (op.res_mode == FP_RES_POOL2_BEFORE_ACT) return false;
    if (op.res_ns != HW * op.res_ld || op.res_ld % 4 || op.res_off % 4 || op.res_C % 4) return false;
    if (op.res_mode == FP_RES_SHUFFLE2 && (op.res_C < op.Cout || op.out_ld < 2 * op.Cout)) return false;
  }
  return true;
}

long fp_pwx6_w_floats(const fp_op& op) { return (long)op.Cin * op.Cout * 3 / 2; }

// 16-row tiles per wave the launcher will pick for this op at its current batch (kernel-name reporting)
int fp_pwx6_mt(const fp_op& op) {
  const int nt = chunk_tiles(op.Cout);
  if ((nt == 8 && (op.flags & FP_OPF_IN_UP2)) || op.Cin <= pwx6_small_maxk()) return 2;
  return nt && pwx6_small_tiles((long)op.N * op.H * op.W, op.Cout / (nt * 16)) ? 2 : 4;
}

int fp_launch_pwx6(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_pwx6_eligible(op)) return FP_ERR_UNSUPPORTED;
  PwX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = (const unsigned short*)(weights + op.w_off);
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.M = (long)op.N * op.H * op.W;
  a.K = op.Cin;
  a.N = op.Cout;
  a.in_ld = op.in_ld;
  a.out_ld = op.out_ld;
  a.res_ld = op.res_ld;
  a.res_C = op.res_C;
  a.act = op.act;
  a.res_mode = op.res_mode;
  if (op.flags & FP_OPF_IN_UP2) {
    a.up = arena + op.res_off;
    a.up_ns = op.res_ns;
    a.up_ld = op.res_ld;
    a.up_slabs = op.res_C / 32;
    a.W2 = op.W / 2;
    a.HW = op.H * op.W;
    a.hw_div = fp_make_divisor((unsigned)a.HW);
    a.w_div = fp_make_divisor((unsigned)op.W);
  }
  switch (chunk_tiles(op.Cout)) {
    case 3: return launch<3>(a, s);
    case 4: return launch<4>(a, s);
    default: return launch<8>(a, s);
  }
}

// The general form (convx6_kernel): 3x3 pad 1 stride 1 / 2, or pointwise with widths that pwx6_kernel does not take.
static bool convx6_shape(const fp_op& op) {
  const bool k3 = op.KH == 3 && op.KW == 3 && op.pad_t == 1 && op.pad_l == 1 && (op.stride == 1 || op.stride == 2) &&
                  op.OH == (op.H + 2 - 3) / op.stride + 1 && op.OW == (op.W + 2 - 3) / op.stride + 1;
  const bool k1 = op.KH == 1 && op.KW == 1 && op.stride == 1 && !op.pad_t && !op.pad_l && op.OH == op.H && op.OW == op.W;
  return k3 || k1;
}

bool fp_convx6_eligible(const fp_op& op) {
  if (op.kind != FP_OP_CONV || !(op.flags & FP_OPF_SPLIT3) || (op.flags & ~(FP_OPF_SPLIT3 | FP_OPF_IN_UP2))) return false;
  if (op.flags & FP_OPF_IN_UP2) {
    if (op.KH != 1 || op.res_mode != FP_RES_NONE || op.res_C % 8 || op.res_C <= 0 || op.res_C >= op.Cin) return false;
    if (op.H % 2 || op.W % 2 || op.res_H != op.H / 2 || op.res_W != op.W / 2) return false;
    if (op.res_ld % 4 || op.res_off % 4 || op.res_ns % 4 || op.res_ld < op.res_C || op.res_ns < (long)op.res_H * op.res_W * op.res_ld)
      return false;
  }
  // Cin >= 32, or 8 / 16 / 24 channels under a 3x3 (K flattened over taps and channels)
  const bool flat = op.Cin < 32 && op.KH == 3 && op.Cin % 8 == 0 && op.Cin >= 8 && !(op.flags & FP_OPF_IN_UP2);
  if (!convx6_shape(op) || op.Cin % 4 || (op.Cin < 32 && !flat) || op.Cout % 4 || op.Cout < 32 || op.out_cmul != 1) return false;
  const long OHW = (long)op.OH * op.OW;
  if (op.in_ns < (long)op.H * op.W * op.in_ld || op.out_ns != OHW * op.out_ld) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.w_off % 4) return false;
  if ((op.scale_off >= 0 && op.scale_off % 4) || (op.bias_off >= 0 && op.bias_off % 4) || (op.slope_off >= 0 && op.slope_off % 4)) return false;
  if (op.act == FP_ACT_PRELU && op.slope_off < 0) return false;
  if (op.res_mode != FP_RES_NONE) {
    if (op.res_mode == FP_RES_POOL2_BEFORE_ACT) return false;
    if (op.res_ns != OHW * op.res_ld || op.res_ld % 4 || op.res_off % 4 || op.res_C % 4) return false;
    if (op.res_mode == FP_RES_SHUFFLE2 && (op.res_C < op.Cout || op.out_ld < 2 * op.Cout)) return false;
  }
  if ((long)op.N * OHW >= (1L << 31) || OHW < 2 || op.OW < 2) return false;   // 32-bit row decode by multiply-high (divisors >= 2)
  return true;
}

int fp_convx6_nt16(const fp_op& op) {
  int nt16, npad;
  general_tiles(op.Cout, &nt16, &npad);
  return nt16;
}

long fp_convx6_w_floats(const fp_op& op) {
  int nt16, npad;
  general_tiles(op.Cout, &nt16, &npad);
  const long slabs = op.Cin < 32 ? ((long)op.KH * op.KW * op.Cin + 31) / 32 : (long)op.KH * op.KW * ((op.Cin + 31) / 32);
  return slabs * 3 * npad * 32 / 2;
}

int fp_launch_convx6(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_convx6_eligible(op)) return FP_ERR_UNSUPPORTED;
  ConvX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.w = (const unsigned short*)(weights + op.w_off);
  a.scale = op.scale_off >= 0 ? weights + op.scale_off : nullptr;
  a.bias = op.bias_off >= 0 ? weights + op.bias_off : nullptr;
  a.slope = op.slope_off >= 0 ? weights + op.slope_off : nullptr;
  a.M = (long)op.N * op.OH * op.OW;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.Cin = op.Cin; a.Cout = op.Cout;
  a.KH = op.KH; a.stride = op.stride; a.pad = op.pad_t;
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.res_C = op.res_C; a.act = op.act; a.res_mode = op.res_mode;
  a.in_ns = op.in_ns;
  a.div_ohw = fp_make_divisor((unsigned)(op.OH * op.OW));
  a.div_ow = fp_make_divisor((unsigned)op.OW);
  a.flat = op.Cin < 32;
  a.div_cin = fp_make_divisor((unsigned)(op.Cin >= 2 ? op.Cin : 2));
  if (op.flags & FP_OPF_IN_UP2) {
    a.up = arena + op.res_off;
    a.up_ns = op.res_ns;
    a.up_ld = op.res_ld;
    a.up_C = op.res_C;
    a.W2 = op.W / 2;
  }
  int nt16;
  general_tiles(op.Cout, &nt16, &a.Npad);
  switch (nt16) {
    case 2: return launch_conv<2>(a, s);
    case 3: return launch_conv<3>(a, s);
    case 4: return launch_conv<4>(a, s);
    default: return launch_conv<6>(a, s);
  }
}
