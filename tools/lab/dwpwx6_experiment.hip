// dwpwx6.hip — depthwise 3x3 (stride 1 / 2, + BN [+ PReLU]) -> 1x1 (+ BN [+ SiLU]) [-> ShuffleV2 cat + channel_shuffle]
// with the 1x1 on the bf16 matrix cores (fp32-equivalent split arithmetic, split.h) — gfx950.
//
// YOLOv5n-face's ShuffleV2 blocks (y5/models/common.py:127-176: branch1 = dw s2 -> 1x1, branch2 tail = dw -> 1x1 -> cat +
// channel_shuffle) ran on dwpw_kernel: the depthwise result stays in LDS, but its 1x1 is an fp32 MFMA that shares the
// vector ALU with the depthwise FMAs (25-50 TFLOP/s, 3-5 TB/s of op-granular traffic).  Here:
//   tile   = TR output rows x the whole output width of one image = one 256-thread workgroup (two per CU);
//   slab   = 32 input channels: the (TR - 1)*s + 3 input rows of the slab go into an LDS image by LDS-DMA (one zero pixel
//            after every row, zero rows outside the image; the eight 16-byte units of a pixel XOR-swizzled by the pixel
//            index, see dwblockx6.hip);
//     D    depthwise + BN [+ PReLU] on the VALU, a lane = (channel pair, output pixel), result split into three bf16 planes
//          -> D-tile [3][pixels][32];  the next slab's DMA is issued when the last lane has read this one;
//     P    W^T (this wave's 16 / 32 output channels, registers) x D^T -> accumulators [pixel tile][channel tile];
//   epilogue = BN + SiLU (+ the ShuffleV2 interleave with the other branch: two 16-byte stores) from the accumulators.
#include <string.h>

#include "split.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

struct DwPwX6Args {
  const float* in;
  float* out;
  const float* res;
  const float* dwp;            // [12][G]: 9 taps, BN scale, BN bias, PReLU slope
  const unsigned short* w;     // [G / 32][3][N][32] bf16
  const float* scale2;         // [N] BN scale, then [N] BN bias
  int H, W, OH, OW, G, N, S, TR, rows, mtp;
  int in_ld, out_ld, res_ld, res_C, act, act2, res_mode;
  long in_ns, out_ns, res_ns;
  int sbytes;                  // bytes of the S-image (D-tile behind it)
  fp_divisor div_ow;
  int nband;
};

constexpr int MAXT = 10;       // 16-pixel tiles of a workgroup tile, at most

// NCT = 16-channel output tiles per wave (1: N = 64, 2: N = 128)
template <int NCT>
__global__ __launch_bounds__(256, 2) void dwpwx6_kernel(DwPwX6Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Sl = (float*)smem_raw;
  unsigned short* Dl = (unsigned short*)(smem_raw + p.sbytes);
  const int DPL = p.mtp * 16 * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  const int img = blockIdx.x / p.nband, band = blockIdx.x - img * p.nband;
  const int r0 = band * p.TR;
  const int nro = min(p.TR, p.OH - r0);
  const int npx = nro * p.OW;
  const int ROWP = p.W + 1;
  const int first = r0 * p.S - 1;                        // input row of S-image row 0
  const int KS = p.G / 32;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const float* xin = p.in + (long)img * p.in_ns;

  auto stage = [&](int s) {
    const int w8 = p.W >> 3;
    for (int it = wave; it < p.rows * w8; it += 4) {
      const int vr = it / w8, c8 = it - vr * w8;
      const int row = first + vr;
      if (row >= 0 && row < p.H) {
        const int slot0 = vr * ROWP + 8 * c8 + 1;
        const int unit = (lane & 7) ^ ((slot0 + (lane >> 3)) & 7);
        const float* src = xin + ((long)(row * p.W + 8 * c8 + (lane >> 3)) * p.in_ld + 32 * s + 4 * unit);
        __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(Sl + slot0 * 32), 16, 0, 0);
      }
    }
  };

  for (int i = tid; i < p.sbytes / 16; i += 256) *(f32x4*)&Sl[i * 4] = z;
  f32x4 pacc[MAXT][NCT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int j = 0; j < NCT; ++j) pacc[t][j] = z;
  __syncthreads();
  stage(0);

  for (int s = 0; s < KS; ++s) {
    // this wave's weight fragments of the slab: output channels 16 (wave NCT + j) + l15, k = 32 s + 8 q .. + 7
    fp_frag3 pbw[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j) {
      const unsigned short* src = p.w + ((long)(s * 3 * p.N + 16 * (wave * NCT + j) + l15) * 32 + 8 * q);
      pbw[j].h = *(const u32x4*)src;
      pbw[j].m = *(const u32x4*)(src + p.N * 32);
      pbw[j].l = *(const u32x4*)(src + 2 * p.N * 32);
    }
    const int c2 = tid & 15;
    const float* dp = p.dwp + 32 * s + 2 * c2;
    f32x2 tap[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) tap[t9] = *(const f32x2*)(dp + t9 * p.G);
    const f32x2 dsc = *(const f32x2*)(dp + 9 * p.G), dbi = *(const f32x2*)(dp + 10 * p.G);
    const f32x2 dsl = *(const f32x2*)(dp + 11 * p.G) - f32x2{1.f, 1.f};
    __syncthreads();                                     // the slab's S-image landed; P(s - 1) is done with the D-tile
    // ---- D ----
    for (int px = tid >> 4; px < npx; px += 16) {
      const int r = (int)fp_fastdiv((unsigned)px, p.div_ow), c = px - r * p.OW;
      const int s0 = (r * p.S) * ROWP + c * p.S;         // S slot of input pixel (r*S - 1 + 0, c*S - 1 + 0)
      f32x2 a;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int sl = s0 + (t9 / 3) * ROWP + (t9 % 3);
        const f32x2 v = *(const f32x2*)(Sl + sl * 32 + 4 * ((c2 >> 1) ^ (sl & 7)) + 2 * (c2 & 1));
        if (t9 == 0) a = v * tap[0];
        else a += v * tap[t9];
      }
      a = a * dsc + dbi;
      if (p.act == FP_ACT_PRELU) {
        const f32x2 neg = {__builtin_fminf(a[0], 0.f), __builtin_fminf(a[1], 0.f)};
        a = neg * dsl + a;
      }
      unsigned h, m, l;
      fp_split_pair(a[0], a[1], h, m, l);
      unsigned* dst = (unsigned*)Dl + (px * 32 + 2 * c2) / 2;
      dst[0] = h;
      dst[DPL / 2] = m;
      dst[DPL] = l;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // D-tile complete, everybody is done with the S-image
    if (s + 1 < KS) stage(s + 1);
    // ---- P ----
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < p.mtp) {
        const unsigned short* src = Dl + ((16 * t + l15) * 32 + 8 * q);
        const u32x4 dh = *(const u32x4*)src, dm = *(const u32x4*)(src + DPL), dl = *(const u32x4*)(src + 2 * DPL);
#pragma unroll
        for (int j = 0; j < NCT; ++j) pacc[t][j] = fp_mfma_x6(pbw[j].h, pbw[j].m, pbw[j].l, dh, dm, dl, pacc[t][j]);
      }
    }
  }

  // ---- epilogue: output pixel o = 16 t + l15 of the tile, channels 16 (wave NCT + j) + 4 q .. + 3 ----
  const bool shuffle = p.res_mode == FP_RES_SHUFFLE2;
  float* yout = p.out + (long)img * p.out_ns + (long)r0 * p.OW * p.out_ld;
  const float* rin = p.res ? p.res + (long)img * p.res_ns + (long)r0 * p.OW * p.res_ld : nullptr;
#pragma unroll
  for (int j = 0; j < NCT; ++j) {
    const int ch = 16 * (wave * NCT + j) + 4 * q;
    const f32x4 sc = *(const f32x4*)(p.scale2 + ch), bi = *(const f32x4*)(p.scale2 + p.N + ch);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int o = 16 * t + l15;
      if (t < p.mtp && o < npx) {
        f32x4 v = pacc[t][j] * sc + bi;
        if (p.act2 == FP_ACT_SILU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fp_silu(v[i]);
        }
        f32x4 rv = z;
        if (p.res_mode != FP_RES_NONE && ch < p.res_C) rv = *(const f32x4*)(rin + (long)o * p.res_ld + ch);
        if (shuffle) {
          float* op = yout + (long)o * p.out_ld + 2 * ch;
          *(f32x4*)op = f32x4{rv[0], v[0], rv[1], v[1]};
          *(f32x4*)(op + 4) = f32x4{rv[2], v[2], rv[3], v[3]};
        } else {
          if (p.res_mode == FP_RES_ADD_AFTER_ACT) v += rv;
          *(f32x4*)(yout + (long)o * p.out_ld + ch) = v;
        }
      }
    }
  }
}

// tile height: the largest TR with <= 160 output pixels whose S-image + D-tile fit 78 KiB (two workgroups per CU)
int pick_tr(const fp_op& op, int* rows, int* mtp, int* sbytes) {
  int best = 0;
  for (int tr = 1; tr <= op.OH && tr * op.OW <= 16 * MAXT; ++tr) {
    const int r = (tr - 1) * op.stride + 3;
    const int sb = ((r * (op.W + 1) + 1) * 128 + 15) / 16 * 16;
    const int mt = (tr * op.OW + 15) / 16;
    if (sb + 3 * mt * 16 * 64 > 78 * 1024) break;
    best = tr; *rows = r; *mtp = mt; *sbytes = sb;
  }
  return best;
}

}  // namespace

// FP_OP_DWPW with FP_OPF_SPLIT3: G a multiple of 32, Cout 64 or 128, W a multiple of 8, dense rows, no second PReLU.
bool fp_dwpwx6_eligible(const fp_op& op) {
  if (op.kind != FP_OP_DWPW || !(op.flags & FP_OPF_SPLIT3) || (op.flags & ~FP_OPF_SPLIT3)) return false;
  if (op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1 || (op.stride != 1 && op.stride != 2)) return false;
  if (op.OH != (op.H + 2 - 3) / op.stride + 1 || op.OW != (op.W + 2 - 3) / op.stride + 1 || op.OW < 2) return false;
  if (op.Cin % 32 || (op.Cout != 64 && op.Cout != 128) || op.W % 8 || op.out_cmul != 1) return false;
  if (op.in_ld % 4 || op.in_off % 4 || op.in_ns % 4 || op.out_ld % 4 || op.out_off % 4 || op.out_ns % 4) return false;
  if (op.in_ns < (long)op.H * op.W * op.in_ld || op.w_off % 4 || op.slope_off % 4 || op.bias_off >= 0) return false;
  if (op.act != FP_ACT_NONE && op.act != FP_ACT_PRELU) return false;
  if (op.act2 != FP_ACT_NONE && op.act2 != FP_ACT_SILU) return false;
  if (op.res_mode != FP_RES_NONE && op.res_mode != FP_RES_SHUFFLE2 && op.res_mode != FP_RES_ADD_AFTER_ACT) return false;
  if (op.res_mode != FP_RES_NONE && (op.res_ld % 4 || op.res_off % 4 || op.res_ns % 4 || op.res_C % 4)) return false;
  if (op.res_mode == FP_RES_SHUFFLE2 && (op.res_C < op.Cout || op.out_ld < 2 * op.Cout)) return false;
  int rows, mtp, sbytes;
  return pick_tr(op, &rows, &mtp, &sbytes) > 0;
}

long fp_dwpwx6_w_floats(const fp_op& op) { return (long)op.Cin * op.Cout * 3 / 2 + 2L * op.Cout; }

int fp_launch_dwpwx6(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_dwpwx6_eligible(op)) return FP_ERR_UNSUPPORTED;
  DwPwX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.res = op.res_mode != FP_RES_NONE ? arena + op.res_off : nullptr;
  a.dwp = weights + op.w_off;
  a.w = (const unsigned short*)(weights + op.slope_off);
  a.scale2 = weights + op.slope_off + (long)op.Cin * op.Cout * 3 / 2;
  a.H = op.H; a.W = op.W; a.OH = op.OH; a.OW = op.OW; a.G = op.Cin; a.N = op.Cout; a.S = op.stride;
  a.TR = pick_tr(op, &a.rows, &a.mtp, &a.sbytes);
  a.in_ld = op.in_ld; a.out_ld = op.out_ld; a.res_ld = op.res_ld; a.res_C = op.res_C;
  a.act = op.act; a.act2 = op.act2; a.res_mode = op.res_mode;
  a.in_ns = op.in_ns; a.out_ns = op.out_ns; a.res_ns = op.res_ns;
  a.div_ow = fp_make_divisor((unsigned)op.OW);
  a.nband = (op.OH + a.TR - 1) / a.TR;
  const long tiles = (long)op.N * a.nband;
  if (tiles >= (1L << 31)) return FP_ERR_UNSUPPORTED;
  const int lds = a.sbytes + 3 * a.mtp * 16 * 64;
  hipError_t ae;
  if (op.Cout == 64) {
    ae = hipFuncSetAttribute((const void*)dwpwx6_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (ae == hipSuccess) hipLaunchKernelGGL((dwpwx6_kernel<1>), dim3((unsigned)tiles), dim3(256), lds, s, a);
  } else {
    ae = hipFuncSetAttribute((const void*)dwpwx6_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (ae == hipSuccess) hipLaunchKernelGGL((dwpwx6_kernel<2>), dim3((unsigned)tiles), dim3(256), lds, s, a);
  }
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  FP_CHECK_LAUNCH();
  return FP_OK;
}
