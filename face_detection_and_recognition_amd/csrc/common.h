// Internal helpers shared by the HIP translation units of libfacepath.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/facepath.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
// Pins the ISSUE ORDER of MFMAs (everything else may still move across): hipcc regroups v_mfma instructions by
// accumulator, and a 32x32x2 f32 MFMA that accumulates into the previous MFMA's result issues at HALF rate (128 instead
// of 64 cycles, tools/lab/mfma_chain_lab.hip).  Sources write the MFMAs round-robin over >= 2 accumulators and put
// this after each one.  Mask = every class but MFMA / generic-ALU may be scheduled across (LLVM sched_barrier bits).
#define FP_MFMA_ORDER() __builtin_amdgcn_sched_barrier(0x7F6)

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FP_WAVE 64

// SiLU x * sigmoid(x) (nn.SiLU, y5/models/common.py:47) with the hardware exp2 / rcp (v_exp_f32, v_rcp_f32: <= 1 ulp
// each, relative error of the result ~1e-6 x |x|/16): 5 VALU instructions.  The IEEE expf + division form costs 26 and,
// applied to every conv output of YOLOv5-face, was ~3 ms of VALU issue per 256-image forward pass.  x -> -inf gives
// exp2 = +inf, rcp = 0, result -0; x -> +inf gives x.  Decisions that must be bit-exact (decode / NMS, post.hip)
// do not use this.
__device__ __forceinline__ float fp_silu(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896341f * x);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// Lab / test knobs of the launchers, taken from the environment ONCE when the library is loaded (a launch never calls
// getenv); fp_debug_reload_env() (facepath.h) re-reads them.  0 = the product path for every knob.
struct fp_knobs {
  int chain_grid;        // FP_CHAIN_GRID: cap on blazechain96_kernel's workgroups (lab)
  int resize_per_pixel;  // FP_RESIZE_PER_PIXEL: fp_resize_normalize takes the per-pixel kernel (the tabled kernel's reference in tests)
  int x6_quarter14;      // FP_X6_QUARTER14: 14 x 14 Depth_Wise blocks as 7 x 7 tiles (lab)
  int x6_spec14;         // FP_X6_SPEC14: the wave-specialised 14 x 14 form (lab)
  int pwx6_small_maxk;   // FP_PWX6_SMALL_MAXK: K at or below which pwx6 takes its small tiles (lab)
  int pair_lds_min;      // FP_PAIR_LDS_MIN: blazepair kernels request at least this much LDS (lab: > 80 KiB = one workgroup per CU)
  int x6_lds_min;        // FP_X6_LDS_MIN: the same for the dwblock_x6 / x6d kernels (lab)
  int shuf_ldsw;         // FP_SHUF_LDSW: shufdown_x6_kernel in its eight-wave, weights-in-LDS form (lab)
};
const fp_knobs& fp_get_knobs();

// Per-thread record of the last HIP error text (fp_last_hip_error()).
void fp_set_hip_error(hipError_t e);

#define FP_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) {                      \
      fp_set_hip_error(e__);                      \
      return FP_ERR_LAUNCH;                       \
    }                                             \
  } while (0)

// n / d by multiply-high for 0 <= n < 2^31 and d >= 2: with s = ceil(log2 d), k = 31 + s and M = floor(2^k / d) + 1
// (M < 2^32 because d > 2^(s-1)), floor(n * M / 2^k) = floor(n / d): the error M*d - 2^k lies in (0, d], and
// n * d < 2^31 * 2^s = 2^k.  Device side: __umulhi(n, M) >> (s - 1).
struct fp_divisor {
  unsigned mul, shift;
};
static inline fp_divisor fp_make_divisor(unsigned d) {
  unsigned s = 0;
  while ((1ull << s) < d) ++s;
  fp_divisor r;
  r.mul = (unsigned)((1ull << (31 + s)) / d + 1ull);
  r.shift = s - 1;   // d >= 2 -> s >= 1
  return r;
}
__device__ __forceinline__ unsigned fp_fastdiv(unsigned n, fp_divisor d) { return __umulhi(n, d.mul) >> d.shift; }

// Workgroup -> work item for 1-D grids.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx & 7), each with its own
// L2: XCD x gets the contiguous range [x G/8, (x + 1) G/8) of items, so items next to each other -- the column chunks of one
// row tile, the bands of one image -- run on the same XCD close in time and share their re-read operands in that L2.
__device__ __forceinline__ unsigned fp_xcd_block() {
  const unsigned G = gridDim.x, b = blockIdx.x, q = G >> 3, rr = G & 7u, xcd = b & 7u, k = b >> 3;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + k;
}

// A 64-bit offset the caller knows to be wave-uniform, rebuilt from readfirstlane'd halves so that the compiler keeps
// it in SGPRs: base + offset stays a scalar address and `global_load v, v_offset32, s[addr:addr+1] offset:imm` needs no
// 64-bit vector address arithmetic.  (Offsets, not pointers: an integer -> pointer cast loses the global address space
// and turns the accesses into flat_load / flat_store.)
__device__ __forceinline__ long fp_uniform(long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
  return (long)(((unsigned long long)hi << 32) | lo);
}

static inline int fp_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline long fp_round_up(long a, long b) { return (a + b - 1) / b * b; }

// launchers implemented in the .hip files, called by the plan executor (capi.cpp)
int fp_launch_conv(const fp_op& op, const float* weights, float* arena, hipStream_t s);
int fp_launch_dwconv(const fp_op& op, const float* weights, float* arena, hipStream_t s);
int fp_launch_maxpool(const fp_op& op, float* arena, hipStream_t s);
int fp_launch_upsample2x(const fp_op& op, float* arena, hipStream_t s);
int fp_launch_copy(const fp_op& op, float* arena, hipStream_t s);
int fp_launch_l2norm(const fp_op& op, float* arena, hipStream_t s);
int fp_launch_blazeblock(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_blazeblock_wp_eligible(const fp_op& op);   // row-padded input: the wave-private kernel takes it (24 -> 24)
bool fp_blazeblock_wps_eligible(const fp_op& op);  // ... its small-map form for the 48- and 96-channel blocks
int fp_launch_blazeblock_rowpad(const fp_op& op, const float* weights, float* arena, hipStream_t s);   // blazewp.hip
bool fp_blazeblock_fixed24(const fp_op& op);   // persistent BlazeBlock instantiated with compile-time 24 -> 24 widths
int fp_launch_dwpw(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_dwpwx6_eligible(const fp_op& op);   // DWPW with FP_OPF_SPLIT3: depthwise on the VALU, 1x1 on the split MFMA (dwpwx6.hip)
long fp_dwpwx6_w_floats(const fp_op& op);
int fp_launch_dwpwx6(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_dwpw_persistent(const fp_op& op);   // true: dwpw_persist_kernel / dwpw_wp_kernel, false: dwpw_kernel
bool fp_dwpw_wave_private(const fp_op& op); // true: dwpw_wp_kernel (projection weights resident in LDS)
bool fp_blazepair_supported(const fp_op& op);   // two stride-1 24 -> 24 BlazeBlocks in one kernel (blazepair.hip)
int fp_blazepair_band_rows(const fp_op& op);
int fp_launch_blazepair(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_blazepair_s2_supported(const fp_op& op);   // a stride-1 24 -> 24 block + the stride-2 block behind it (blazepairs2.hip)
int fp_blazepair_s2_band_rows(const fp_op& op);
int fp_launch_blazepair_s2(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_blazechain_supported(const fp_op& op);  // a run of stride-1 96 -> 96 BlazeBlocks on a 16 x 16 map in one kernel (blazechain.hip)
int64_t fp_blazechain_w_floats(const fp_op& op);
int fp_launch_blazechain(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_dwblock_supported(const fp_op& op); // whole Depth_Wise block shapes dwblock.hip is instantiated for
int fp_launch_dwblock(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_dwblock_x6_supported(const fp_op& op);   // DWBLOCK with FP_OPF_SPLIT3: bf16x6 split-MFMA kernel (dwblockx6.hip)
long fp_dwblock_x6_we_floats(const fp_op& op);   // floats behind w_off / slope_off of such an op
long fp_dwblock_x6_wp_floats(const fp_op& op);
int fp_launch_dwblock_x6(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_shufdown_supported(const fp_op& op);     // FP_OP_SHUFDOWN: a whole stride-2 ShuffleV2Block (shufdown.hip)
long fp_shufdown_w_floats(const fp_op& op);
int fp_launch_shufdown(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_shufunit_supported(const fp_op& op);     // FP_OP_SHUFUNIT: a whole stride-1 ShuffleV2Block (shufdown.hip)
long fp_shufunit_w_floats(const fp_op& op);
int fp_launch_shufunit(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_ystem2_supported(const fp_op& op);       // FP_OP_YSTEM2: stem_2b + cat + stem_3 of YOLOv5-face's StemBlock (ystem2.hip)
long fp_ystem2_w_floats(const fp_op& op);
int fp_launch_ystem2(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_pwx6_eligible(const fp_op& op);     // CONV with FP_OPF_SPLIT3: pointwise conv on the bf16x6 split-MFMA kernel (pwx6.hip)
long fp_pwx6_w_floats(const fp_op& op);
int fp_pwx6_mt(const fp_op& op);
bool fp_convx6_eligible(const fp_op& op);   // ... the general form: 3x3 pad 1 stride 1 / 2, widths padded to 32 / 16 (convx6_kernel)
long fp_convx6_w_floats(const fp_op& op);
int fp_convx6_nt16(const fp_op& op);
int fp_launch_convx6(const fp_op& op, const float* weights, float* arena, hipStream_t s);
int fp_launch_pwx6(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_pws_eligible(const fp_op& op);      // pointwise K = 64 convs that take the wave-private streaming kernel
int fp_launch_pws(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_stem_eligible(const fp_op& op);     // KxK stride-2 convs on a 4-float-pixel image (network stems)
int fp_launch_stem(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_stemdw_supported(const fp_op& op);   // CONV + FP_OPF_OUT_DW: Mobile-FaceNet's conv1 + conv2_dw in one kernel (stemdw.hip)
long fp_stemdw_w_floats(const fp_op& op);
int fp_launch_stemdw(const fp_op& op, const float* weights, float* arena, hipStream_t s);
bool fp_stem_u8_shape_ok(const fp_op& op);
bool fp_stem_u8_band_eligible(const fp_op& op);   // BlazeFace's 5x5 stem on u8 frames, band kernel (canvas rows in an LDS ring)
int fp_launch_stem_u8(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, int n_ext, hipStream_t s);
int fp_launch_ystem(const fp_op& op, const float* weights, float* arena, hipStream_t s);
int fp_ystem_nb2(const fp_op& op);
int fp_launch_ystem_u8(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, int n_ext, hipStream_t s);
bool fp_conv3_eligible(const fp_op& op);    // dense 3x3 pad-1 convs that take the LDS-image kernel (conv3.hip)
int fp_conv3_nb(const fp_op& op);
bool fp_conv3_t16(const fp_op& op);        // conv3 with 16-column n tiles (16x16x4 MFMA)
int fp_launch_conv3(const fp_op& op, const float* weights, float* arena, hipStream_t s);
void fp_conv_variant(const fp_op& op, int* nb, int* vec, int* pwd);  // conv_igemm template arguments for an op
