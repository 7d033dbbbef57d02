// blazechain.hip — a RUN of stride-1 96 -> 96 BlazeBlocks on a 16 x 16 map as ONE kernel (gfx950).
//
// BlazeFace-back ends in seven BlazeBlock(96, 96) on the 16 x 16 map (fde/modules/blazeface/blazeface.py:146-152; the block
// itself :12-47: y = ReLU(conv1x1(dw3x3(x)) + x)).  One image of that map is 98 KB of fp32: it fits the 160 KB LDS of a CU.
// As seven launches (blazeblock_wps_kernel<96>) every block reads and writes the tensor in HBM and is a short latency chain
// (31 us each at batch 256 = 1.6 TB/s); here ONE workgroup (8 waves) keeps one image in LDS, runs all blocks on it in
// place and touches HBM twice.
//   * LDS image [18 rows][16 px][100 floats]: rows 0 / 17 are the zero rows above / below the map, the pixel stride 100
//     (not 96) spreads the 16 pixels a 16-byte access touches over all banks;
//   * wave w owns image rows 2w, 2w + 1.  A 16-pixel image row IS a 16-lane DPP row: lane = (column x = lane & 15, channel
//     octet g = lane >> 4), so the depthwise 3x3 needs only the lane's own pixel of rows y-1..y+1 from LDS; the left /
//     right taps are accumulated per lane for the NEIGHBOUR and moved over with row_shr:1 / row_shl:1 (bound_ctrl: lanes at
//     the image border receive 0 = the zero padding of the conv);
//   * (column, 8 consecutive channels) per lane is exactly the operand layout of v_mfma_f32_16x16x32_bf16, so the depthwise
//     output goes from registers through the exact three-way bf16 split (split.h) straight into the 1x1 conv: six bf16
//     MFMAs per product, fp32 accumulation, K = 96 in three steps of 32; the weight slab of a step ([3 planes][96][32]
//     bf16, split on the host) is staged by LDS-DMA one step ahead, double-buffered;
//   * operands swapped (D^T = W^T A^T): a lane ends with 4 consecutive output channels of ONE pixel: bias + shortcut (from
//     the LDS image) + ReLU on 16-byte pieces, written back in place (the last block: to global memory).
// Four workgroup barriers per block: one per weight slab, one between the last depthwise read of the image and its update.
#include <stdlib.h>

#include "split.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

constexpr int C = 96, HW = 16;
constexpr int PS = 100;                          // LDS pixel stride (floats)
constexpr int ROWF = HW * PS;                    // LDS row stride
constexpr int IMG_F = (HW + 2) * ROWF;
constexpr int SLAB_B = 3 * C * 32 * 2;           // bytes of one weight slab: three bf16 planes of [96][32]
constexpr int PARP = 1280;                       // floats of a block's fp32 parameters, padded to 5 KiB: [9][96] taps, [96] dw bias, [96] 1x1 bias
constexpr int BLK_F = PARP + 3 * SLAB_B / 4;     // floats of one block in the weight blob
constexpr size_t LDS_BYTES = (size_t)IMG_F * 4 + 2 * SLAB_B + 2 * PARP * 4;

struct ChainArgs {
  const float* in;
  float* out;
  const float* w;
  long in_ns, out_ns;
  int nblk, N;
};

// column x - 1 / x + 1 of the same image row and channel (0 outside the row)
__device__ __forceinline__ float from_left(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_right(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true));
}

__global__ __launch_bounds__(512, 1) void blazechain96_kernel(ChainArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* img = (float*)smem_raw;
  unsigned char* slab = smem_raw + (size_t)IMG_F * 4;
  float* par = (float*)(slab + 2 * SLAB_B);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nblk = p.nblk, nslab = 3 * nblk;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int q) {        // weight slab q = (block q / 3, k-step q % 3) -> slab buffer q & 1: 18 pieces of 1 KiB
    const unsigned char* src = (const unsigned char*)(p.w + (long)(q / 3) * BLK_F + PARP) + (q % 3) * SLAB_B + lane * 16;
    unsigned char* dst = slab + (q & 1) * SLAB_B;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int piece = j * 8 + wave;
      if (piece < SLAB_B / 1024) __builtin_amdgcn_global_load_lds((gbl_ptr)(src + piece * 1024), (lds_ptr)(dst + piece * 1024), 16, 0, 0);
    }
  };
  auto stage_par = [&](int b) {    // fp32 parameters of block b -> par buffer b & 1: 5 pieces of 1 KiB
    if (wave < PARP * 4 / 1024)
      __builtin_amdgcn_global_load_lds((gbl_ptr)((const unsigned char*)(p.w + (long)b * BLK_F) + wave * 1024 + lane * 16),
                                       (lds_ptr)((unsigned char*)(par + (b & 1) * PARP) + wave * 1024), 16, 0, 0);
  };

  // persistent over images: the launcher may start fewer workgroups than images (FP_CHAIN_GRID / co-scheduled plans: a
  // workgroup owns its CU's LDS, so a capped grid leaves the other CUs to whatever runs beside this kernel)
  for (int img_i = blockIdx.x; img_i < p.N; img_i += gridDim.x) {
  const float* in = p.in + fp_uniform((long)img_i * p.in_ns);
  float* out = p.out + fp_uniform((long)img_i * p.out_ns);
  __syncthreads();      // every wave is done with the previous image (its last epilogue read the LDS image)
  stage(0);
  stage_par(0);
  // the image: 256 px x 24 float4, dense in global memory, pixel stride PS in LDS; zero rows above and below
  {
    f32x4 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = *(const f32x4*)(in + (tid + 512 * k) * 4);
    if (tid < ROWF / 4) {
      *(f32x4*)(img + tid * 4) = z;
      *(f32x4*)(img + (HW + 1) * ROWF + tid * 4) = z;
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const int idx = tid + 512 * k, px = idx / 24, j = idx - px * 24;
      *(f32x4*)(img + ROWF + px * PS + 4 * j) = v[k];
    }
  }

  const int lane_px = c * PS;
  for (int b = 0; b < nblk; ++b) {
    const float* pb = par + (b & 1) * PARP;
    f32x4 acc[2][6];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int n = 0; n < 6; ++n) acc[t][n] = z;

#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int q = 3 * b + s;
      __syncthreads();      // slab q (and, s = 0: the parameters and the image of this block) landed; everybody is done with slab q - 1
      if (q + 1 < nslab) stage(q + 1);
      if (s == 1 && b + 1 < nblk) stage_par(b + 1);

      // depthwise 3x3 + bias for image rows 2 wave, 2 wave + 1, channels 32 s + 8 g .. + 7 of pixel column c
      const float* ip = img + (2 * wave) * ROWF + lane_px + 32 * s + 8 * g;      // LDS row 2 wave = image row 2 wave - 1
      const float* wt = pb + 32 * s + 8 * g;
      f32x4 x[4][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[i][0] = *(const f32x4*)(ip + i * ROWF);
        x[i][1] = *(const f32x4*)(ip + i * ROWF + 4);
      }
      fp_frag3 af[2];
      {
        f32x4 oc[2][2], ol[2][2], orr[2][2];     // centre column taps (+ bias); what the right / left neighbour gets from this pixel
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 bias = *(const f32x4*)(wt + 9 * C + 4 * h);
          oc[0][h] = oc[1][h] = bias;
          ol[0][h] = ol[1][h] = orr[0][h] = orr[1][h] = z;
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 w0 = *(const f32x4*)(wt + (dy * 3 + 0) * C + 4 * h);
            const f32x4 w1 = *(const f32x4*)(wt + (dy * 3 + 1) * C + 4 * h);
            const f32x4 w2 = *(const f32x4*)(wt + (dy * 3 + 2) * C + 4 * h);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const f32x4 v = x[t + dy][h];
              oc[t][h] += w1 * v;
              ol[t][h] += w0 * v;    // tap (dy, x - 1) of the pixel to the RIGHT of this one
              orr[t][h] += w2 * v;   // tap (dy, x + 1) of the pixel to the LEFT
            }
          }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) oc[t][h][e] += from_left(ol[t][h][e]) + from_right(orr[t][h][e]);
          af[t] = fp_split8(oc[t][0], oc[t][1]);
        }
      }

      // 1x1 conv, k-step s: weights (rows = 16 output channels) x activations (columns = the 16 pixels of an image row)
      const unsigned short* Bc = (const unsigned short*)(slab + (q & 1) * SLAB_B) + (c * 32 + 8 * g);
      fp_frag3 bf[2];
      auto ldb = [&](int n, fp_frag3& f) {
        f.h = *(const u32x4*)(Bc + n * 512);
        f.m = *(const u32x4*)(Bc + C * 32 + n * 512);
        f.l = *(const u32x4*)(Bc + 2 * C * 32 + n * 512);
      };
      ldb(0, bf[0]);
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        if (n + 1 < 6) ldb(n + 1, bf[(n + 1) & 1]);
        const fp_frag3& w = bf[n & 1];
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t][n] = fp_mfma_x6(w.h, w.m, w.l, af[t].h, af[t].m, af[t].l, acc[t][n]);
      }
    }

    __syncthreads();        // every wave has read what it needs of this block's input image
    const bool last = b + 1 == nblk;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = 2 * wave + t;
      float* xp = img + (r + 1) * ROWF + lane_px + 4 * g;
      float* op = out + (r * HW + c) * C + 4 * g;
#pragma unroll
      for (int n = 0; n < 6; ++n) {
        f32x4 v = acc[t][n] + *(const f32x4*)(pb + 10 * C + 16 * n + 4 * g) + *(const f32x4*)(xp + 16 * n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        if (last)
          *(f32x4*)(op + 16 * n) = v;
        else
          *(f32x4*)(xp + 16 * n) = v;
      }
    }
  }
  }
}

}  // namespace

// A run of Cmid (1..16) stride-1 96 -> 96 BlazeBlocks on a dense 16 x 16 map (include/facepath.h, BLAZECHAIN).
bool fp_blazechain_supported(const fp_op& op) {
  if (op.kind != FP_OP_BLAZECHAIN || op.flags != FP_OPF_SPLIT3) return false;
  if (op.stride != 1 || op.KH != 3 || op.KW != 3 || op.pad_t != 1 || op.pad_l != 1) return false;
  if (op.Cin != C || op.Cout != C || op.in_ld != C || op.out_ld != C || op.out_cmul != 1) return false;
  if (op.H != HW || op.W != HW || op.OH != HW || op.OW != HW) return false;
  if (op.Cmid < 1 || op.Cmid > 16) return false;
  if (op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4 || op.w_off % 4) return false;
  return op.res_mode == FP_RES_ADD_BEFORE_ACT && op.act == FP_ACT_RELU;
}

int64_t fp_blazechain_w_floats(const fp_op& op) { return (int64_t)op.Cmid * BLK_F; }

int fp_launch_blazechain(const fp_op& op, const float* weights, float* arena, hipStream_t s) {
  if (!fp_blazechain_supported(op)) return FP_ERR_UNSUPPORTED;
  ChainArgs a;
  a.in = arena + op.in_off;
  a.out = arena + op.out_off;
  a.w = weights + op.w_off;
  a.in_ns = op.in_ns;
  a.out_ns = op.out_ns;
  a.nblk = op.Cmid;
  a.N = op.N;
  int grid = op.N;
  const int cap = fp_get_knobs().chain_grid;           // lab: cap the number of workgroups (= CUs this kernel occupies)
  if (cap > 0 && cap < grid) grid = cap;
  const hipError_t ae = hipFuncSetAttribute((const void*)blazechain96_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
  if (ae != hipSuccess) {
    fp_set_hip_error(ae);
    return FP_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(blazechain96_kernel, dim3(grid), dim3(512), LDS_BYTES, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
