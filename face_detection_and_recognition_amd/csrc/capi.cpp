// capi.cpp — plan validation + executor and the small ABI utilities of libfacepath.so.
// Compiled by hipcc as host code; kernels live in the .hip files.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

static thread_local char g_hip_err[256] = "";

void fp_set_hip_error(hipError_t e) {
  const char* s = hipGetErrorString(e);
  strncpy(g_hip_err, s ? s : "unknown", sizeof(g_hip_err) - 1);
  g_hip_err[sizeof(g_hip_err) - 1] = 0;
}

static int env_int(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}
static fp_knobs read_knobs() {
  fp_knobs k;
  k.chain_grid = env_int("FP_CHAIN_GRID");
  k.resize_per_pixel = getenv("FP_RESIZE_PER_PIXEL") != nullptr;
  k.x6_quarter14 = env_int("FP_X6_QUARTER14");
  k.x6_spec14 = env_int("FP_X6_SPEC14");
  k.pwx6_small_maxk = env_int("FP_PWX6_SMALL_MAXK");
  k.pair_lds_min = env_int("FP_PAIR_LDS_MIN");
  k.x6_lds_min = env_int("FP_X6_LDS_MIN");
  k.shuf_ldsw = env_int("FP_SHUF_LDSW");
  return k;
}
static fp_knobs g_knobs = read_knobs();
const fp_knobs& fp_get_knobs() { return g_knobs; }

extern "C" {

int fp_abi_version(void) { return FP_ABI_VERSION; }

void fp_debug_reload_env(void) { g_knobs = read_knobs(); }

const char* fp_last_hip_error(void) { return g_hip_err; }

// host mirror of the device-side fp_fastdiv (common.h): __umulhi(n, mul) >> shift
static unsigned fastdiv_host(unsigned n, fp_divisor d) { return (unsigned)(((unsigned long long)n * d.mul) >> 32) >> d.shift; }

int fp_selftest(void) {
  auto check = [&](unsigned d) -> bool {
    const fp_divisor dv = fp_make_divisor(d);
    const unsigned nmax = 0x7fffffffu, qmax = nmax / d;
    const unsigned qs[] = {0u, 1u, 2u, qmax / 3, qmax / 2, qmax - 1, qmax};
    for (unsigned q : qs) {
      const unsigned long long lo = (unsigned long long)q * d;
      const unsigned long long cand[] = {lo, lo + 1, lo + d - 1, lo + d / 2};
      for (unsigned long long n64 : cand) {
        if (n64 > nmax) continue;
        const unsigned n = (unsigned)n64;
        if (fastdiv_host(n, dv) != n / d) {
          snprintf(g_hip_err, sizeof(g_hip_err), "fp_fastdiv(%u, %u) = %u, expected %u", n, d, fastdiv_host(n, dv), n / d);
          return false;
        }
      }
    }
    if (fastdiv_host(nmax, dv) != nmax / d) {
      snprintf(g_hip_err, sizeof(g_hip_err), "fp_fastdiv(%u, %u) wrong at the top of the range", nmax, d);
      return false;
    }
    return true;
  };
  for (unsigned d = 2; d <= 4096; ++d)
    if (!check(d)) return FP_ERR_INVALID_ARG;
  const unsigned big[] = {4097u, 12321u, 16384u, 65535u, 65536u, 65537u, 1000003u, 16777216u, 123456789u, 1u << 30, (1u << 30) + 1u};
  for (unsigned d : big)
    if (!check(d)) return FP_ERR_INVALID_ARG;
  return FP_OK;
}

const char* fp_strerror(int status) {
  switch (status) {
    case FP_OK: return "ok";
    case FP_ERR_INVALID_ARG: return "invalid argument";
    case FP_ERR_BOUNDS: return "op touches memory outside the arena or weight blob";
    case FP_ERR_UNSUPPORTED: return "unsupported op or parameter combination";
    case FP_ERR_LAUNCH: return "HIP kernel launch failed";
    case FP_ERR_ALIGNMENT: return "channel count / stride / offset not a multiple of 4 floats";
    default: return "unknown status";
  }
}

static bool span_ok(int64_t off, int64_t extent, size_t limit) {
  return off >= 0 && extent >= 0 && (uint64_t)(off + extent) <= (uint64_t)limit;
}

static int validate_op(const fp_op& op, size_t weight_floats, size_t arena_floats) {
  if (op.N <= 0 || op.H <= 0 || op.W <= 0 || op.Cin <= 0) return FP_ERR_INVALID_ARG;
  const bool spatial = op.kind != FP_OP_L2NORM && op.kind != FP_OP_COPY;
  const int OH = spatial ? op.OH : op.H, OW = spatial ? op.OW : op.W;
  if (OH <= 0 || OW <= 0) return FP_ERR_INVALID_ARG;
  const bool ext_in = op.kind == FP_OP_YSTEM_U8 || op.kind == FP_OP_STEM_U8;   // input in an external buffer (checked at launch)
  const int Cout = (op.kind == FP_OP_CONV || op.kind == FP_OP_BLAZEBLOCK || op.kind == FP_OP_DWPW ||
                    op.kind == FP_OP_DWBLOCK || op.kind == FP_OP_BLAZEPAIR || op.kind == FP_OP_BLAZECHAIN || op.kind == FP_OP_YSTEM ||
                    op.kind == FP_OP_SHUFDOWN || op.kind == FP_OP_SHUFUNIT || op.kind == FP_OP_YSTEM2 || ext_in) ? op.Cout : op.Cin;
  if (op.reserved0 != 0 || (op.kind != FP_OP_DWBLOCK && op.kind != FP_OP_BLAZECHAIN && op.kind != FP_OP_SHUFDOWN && op.kind != FP_OP_SHUFUNIT && op.Cmid != 0)) return FP_ERR_INVALID_ARG;
  if (Cout <= 0 || op.out_cmul < 1 || op.in_ld < op.Cin) return FP_ERR_INVALID_ARG;
  // row-padded views (facepath.h FP_OPF_*): which ops take them, and their extent including the pads
  if (op.flags & ~(FP_OPF_IN_ROWPAD | FP_OPF_OUT_ROWPAD | FP_OPF_IN_C3 | FP_OPF_SPLIT3 | FP_OPF_IN_DW | FP_OPF_IN_UP2 | FP_OPF_OUT_DW))
    return FP_ERR_INVALID_ARG;
  if (op.flags & FP_OPF_OUT_DW) {
    // Mobile-FaceNet's conv1 + conv2_dw (facepath.h): the conv's slopes are followed by the depthwise block's [12][Cout]
    if (op.kind != FP_OP_CONV) return FP_ERR_INVALID_ARG;
    if (!fp_stemdw_supported(op)) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.slope_off, 13 * (int64_t)op.Cout, weight_floats)) return FP_ERR_BOUNDS;
    if (!span_ok(op.w_off, fp_stemdw_w_floats(op), weight_floats)) return FP_ERR_BOUNDS;
  }
  if (op.flags & FP_OPF_IN_UP2) {
    // channels [0, res_C) come from the res view at half resolution (facepath.h): only the split-MFMA pointwise kernel reads that
    if (op.kind != FP_OP_CONV || !(op.flags & FP_OPF_SPLIT3)) return FP_ERR_INVALID_ARG;
    if (!fp_pwx6_eligible(op) && !fp_convx6_eligible(op)) return FP_ERR_UNSUPPORTED;
    const int64_t up_ext = (int64_t)(op.N - 1) * op.res_ns + ((int64_t)op.res_H * op.res_W - 1) * op.res_ld + op.res_C;
    if (!span_ok(op.res_off, up_ext, arena_floats)) return FP_ERR_BOUNDS;
  }
  if ((op.flags & FP_OPF_IN_DW) && (op.kind != FP_OP_DWBLOCK || !(op.flags & FP_OPF_SPLIT3))) return FP_ERR_INVALID_ARG;
  if ((op.flags & FP_OPF_SPLIT3) && op.kind != FP_OP_DWBLOCK && op.kind != FP_OP_CONV && op.kind != FP_OP_DWPW && op.kind != FP_OP_BLAZECHAIN &&
      op.kind != FP_OP_SHUFDOWN && op.kind != FP_OP_SHUFUNIT && op.kind != FP_OP_YSTEM2 && op.kind != FP_OP_STEM_U8)
    return FP_ERR_INVALID_ARG;
  if ((op.flags & FP_OPF_SPLIT3) && op.kind == FP_OP_STEM_U8 && !fp_stem_u8_band_eligible(op)) return FP_ERR_UNSUPPORTED;
  if ((op.flags & FP_OPF_SPLIT3) && op.kind == FP_OP_DWPW && !fp_dwpwx6_eligible(op)) return FP_ERR_UNSUPPORTED;
  if ((op.flags & FP_OPF_SPLIT3) && op.kind == FP_OP_CONV && !(op.flags & FP_OPF_OUT_DW) && !fp_pwx6_eligible(op) && !fp_convx6_eligible(op))
    return FP_ERR_UNSUPPORTED;
  if ((op.flags & FP_OPF_IN_C3) && (op.Cin != 4 || (op.kind != FP_OP_CONV && op.kind != FP_OP_YSTEM)))
    return FP_ERR_INVALID_ARG;
  const bool in_rp = (op.flags & FP_OPF_IN_ROWPAD) != 0, out_rp = (op.flags & FP_OPF_OUT_ROWPAD) != 0;
  if (in_rp && !fp_blazeblock_wp_eligible(op) && !fp_blazeblock_wps_eligible(op) && !fp_blazepair_supported(op) &&
      !fp_blazepair_s2_supported(op))
    return FP_ERR_UNSUPPORTED;
  if (out_rp && !(op.kind == FP_OP_BLAZEBLOCK || op.kind == FP_OP_BLAZEPAIR || op.kind == FP_OP_STEM_U8 || op.kind == FP_OP_COPY ||
                  (op.kind == FP_OP_CONV && fp_stem_eligible(op))))
    return FP_ERR_UNSUPPORTED;
  if (out_rp && (op.out_cmul != 1 || op.out_ld != Cout)) return FP_ERR_UNSUPPORTED;
  // a row-padded COPY exists only in its 16-byte form (copy4_kernel); the scalar copy would fail at launch
  if (out_rp && op.kind == FP_OP_COPY &&
      (op.Cin % 4 || op.in_ld % 4 || op.out_ld % 4 || op.in_off % 4 || op.out_off % 4 || op.in_ns % 4 || op.out_ns % 4))
    return FP_ERR_UNSUPPORTED;
  // input extent
  if (in_rp) {
    const int64_t lead = (int64_t)(op.W + 2) * op.in_ld;
    if (op.in_ns < ((int64_t)(op.H + 2) * (op.W + 1) + 1) * op.in_ld) return FP_ERR_INVALID_ARG;
    if (op.in_off < lead || !span_ok(op.in_off - lead, (int64_t)op.N * op.in_ns, arena_floats)) return FP_ERR_BOUNDS;
  } else {
    const int64_t in_ext = (int64_t)(op.N - 1) * op.in_ns + ((int64_t)op.H * op.W - 1) * op.in_ld + op.Cin;
    if (!ext_in && !span_ok(op.in_off, in_ext, arena_floats)) return FP_ERR_BOUNDS;
  }
  if (ext_in && op.in_off < 0) return FP_ERR_INVALID_ARG;
  const int64_t out_ch = ((op.kind == FP_OP_CONV || op.kind == FP_OP_DWPW) && op.res_mode == FP_RES_SHUFFLE2)
                             ? 2 * (int64_t)Cout : Cout;
  if (out_rp) {
    const int64_t lead = (int64_t)(OW + 2) * op.out_ld;
    if (op.out_ns < ((int64_t)(OH + 2) * (OW + 1) + 1) * op.out_ld) return FP_ERR_INVALID_ARG;
    if (op.out_off < lead || !span_ok(op.out_off - lead, (int64_t)op.N * op.out_ns, arena_floats)) return FP_ERR_BOUNDS;
  } else {
    const int64_t out_ext =
        (int64_t)(op.N - 1) * op.out_ns + ((int64_t)OH * OW - 1) * op.out_ld + (out_ch - 1) * op.out_cmul + 1;
    if (!span_ok(op.out_off, out_ext, arena_floats)) return FP_ERR_BOUNDS;
  }
  if (op.in_ns < 0 || op.out_ns < 0) return FP_ERR_INVALID_ARG;

  if (op.kind == FP_OP_CONV || op.kind == FP_OP_DWCONV || op.kind == FP_OP_MAXPOOL || op.kind == FP_OP_BLAZEBLOCK ||
      op.kind == FP_OP_DWPW || op.kind == FP_OP_DWBLOCK || op.kind == FP_OP_BLAZEPAIR || op.kind == FP_OP_BLAZECHAIN ||
      op.kind == FP_OP_YSTEM || op.kind == FP_OP_SHUFDOWN || op.kind == FP_OP_SHUFUNIT || op.kind == FP_OP_YSTEM2 || ext_in) {
    if (op.KH <= 0 || op.KW <= 0 || op.stride <= 0 || op.pad_t < 0 || op.pad_l < 0) return FP_ERR_INVALID_ARG;
    // every output pixel must have at least its first tap row/col addressable without overflow of int math
    if ((int64_t)(OH - 1) * op.stride - op.pad_t >= op.H || (int64_t)(OW - 1) * op.stride - op.pad_l >= op.W)
      return FP_ERR_INVALID_ARG;
  }
  if (op.kind == FP_OP_CONV || op.kind == FP_OP_BLAZEBLOCK || op.kind == FP_OP_STEM_U8) {
    int64_t wext;
    if (op.kind != FP_OP_BLAZEBLOCK) {
      const int64_t K = (int64_t)op.KH * op.KW * (op.kind == FP_OP_STEM_U8 ? 4 : op.Cin);
      wext = ((K + 7) / 8 * 8) * ((op.Cout + 31) / 32 * 32);
      if (op.flags & FP_OPF_SPLIT3)     // three bf16 planes: 1.5 floats per (padded) weight; the u8 stem: [3 slabs][2][3][16][32] bf16
        wext = op.kind == FP_OP_STEM_U8 ? 3 * 2 * 3 * 16 * 32 / 2 : fp_convx6_w_floats(op);
      if (op.flags & FP_OPF_OUT_DW) wext = fp_stemdw_w_floats(op);
    } else {
      // BLAZEBLOCK: dw weights [9][Cin] followed (separately addressed) by the packed 1x1; w_off addresses the
      // dw weights, scale_off the dw bias, slope_off the packed pointwise weights, bias_off the pointwise bias.
      wext = 9 * (int64_t)op.Cin;
      const int64_t pw = ((op.Cin + 7) / 8 * 8) * (int64_t)((op.Cout + 31) / 32 * 32);
      if (!span_ok(op.slope_off, pw, weight_floats)) return FP_ERR_BOUNDS;
      if (!span_ok(op.scale_off, op.Cin, weight_floats)) return FP_ERR_BOUNDS;
      if (!span_ok(op.bias_off, op.Cout, weight_floats)) return FP_ERR_BOUNDS;
    }
    if (!span_ok(op.w_off, wext, weight_floats)) return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_DWCONV) {
    if (!span_ok(op.w_off, (int64_t)op.KH * op.KW * op.Cin, weight_floats)) return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_DWPW) {
    // w_off: [9*G taps][G scale][G bias][G slope]; slope_off: [Kpad*Npad packed 1x1][Cout4 scale][Cout4 bias]
    if (!span_ok(op.w_off, 12 * (int64_t)op.Cin, weight_floats)) return FP_ERR_BOUNDS;
    const int64_t pw = (op.flags & FP_OPF_SPLIT3) ? fp_dwpwx6_w_floats(op) :
        ((op.Cin + 7) / 8 * 8) * (int64_t)((op.Cout + 31) / 32 * 32) + 2 * (int64_t)((op.Cout + 3) / 4 * 4);
    if (!span_ok(op.slope_off, pw, weight_floats)) return FP_ERR_BOUNDS;
    if (op.act != FP_ACT_NONE && op.act != FP_ACT_PRELU) return FP_ERR_INVALID_ARG;
    // bias_off: optional [Cout4] PReLU slopes of the projection output
    if (op.bias_off >= 0 && !span_ok(op.bias_off, (op.Cout + 3) / 4 * 4, weight_floats)) return FP_ERR_BOUNDS;
    if (op.bias_off >= 0 && op.res_mode != FP_RES_NONE) return FP_ERR_UNSUPPORTED;
    if (op.act2 != FP_ACT_NONE && op.act2 != FP_ACT_SILU) return FP_ERR_INVALID_ARG;
  } else if (op.kind == FP_OP_SHUFDOWN) {
    // one parameter block at w_off (facepath.h SHUFDOWN); the shapes the kernel exists for are fp_shufdown_supported's
    if (!fp_shufdown_supported(op)) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.w_off, fp_shufdown_w_floats(op), weight_floats)) return FP_ERR_BOUNDS;
  } else if (op.kind == FP_OP_YSTEM2) {
    // parameter block at w_off, the pooled map in the res view (facepath.h YSTEM2)
    if (!fp_ystem2_supported(op)) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.w_off, fp_ystem2_w_floats(op), weight_floats)) return FP_ERR_BOUNDS;
    const int64_t res_ext = (int64_t)(op.N - 1) * op.res_ns + ((int64_t)op.OH * op.OW - 1) * op.res_ld + op.res_C;
    if (!span_ok(op.res_off, res_ext, arena_floats)) return FP_ERR_BOUNDS;
  } else if (op.kind == FP_OP_SHUFUNIT) {
    if (!fp_shufunit_supported(op)) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.w_off, fp_shufunit_w_floats(op), weight_floats)) return FP_ERR_BOUNDS;
  } else if (op.act2 != FP_ACT_NONE) {
    return FP_ERR_INVALID_ARG;
  }
  if (op.kind == FP_OP_BLAZEPAIR) {
    // both blocks' parameters back to back (facepath.h BLAZEPAIR)
    // stride 2: the second block is the stride-2 block behind a stride-1 block (24 -> 24 or 24 -> 48; blazepairs2.hip)
    if (op.stride == 2 ? !fp_blazepair_s2_supported(op) : !fp_blazepair_supported(op)) return FP_ERR_UNSUPPORTED;
    const int64_t pw2 = op.stride == 2 && op.Cout == 48 ? 2 * 768 : 768, c2 = op.stride == 2 ? op.Cout : 24;
    if (!span_ok(op.w_off, 2 * 9 * 24, weight_floats) || !span_ok(op.scale_off, 2 * 24, weight_floats) ||
        !span_ok(op.slope_off, 768 + pw2, weight_floats) || !span_ok(op.bias_off, 24 + c2, weight_floats))
      return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_BLAZECHAIN) {
    // Cmid blocks back to back at w_off, each [1280 fp32 parameters][3 slabs of split 1x1 weights] (facepath.h BLAZECHAIN)
    if (!fp_blazechain_supported(op)) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.w_off, fp_blazechain_w_floats(op), weight_floats)) return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_DWBLOCK) {
    // w_off: expand packed as CONV (K = Cin, Npad = Cmid); scale_off: [15][Cmid]; slope_off: project packed as CONV
    // (K = Cmid, Npad = Cout) + [Cout] scale + [Cout] bias.  The shapes the kernel exists for are fp_dwblock_supported's.
    // FP_OPF_SPLIT3: both weight matrices as three bf16 planes (1.5 floats per weight), see facepath.h
    const bool x6 = (op.flags & FP_OPF_SPLIT3) != 0;
    if (!(x6 ? fp_dwblock_x6_supported(op) : fp_dwblock_supported(op))) return FP_ERR_UNSUPPORTED;
    if (!span_ok(op.w_off, x6 ? fp_dwblock_x6_we_floats(op) : (int64_t)op.Cin * op.Cmid, weight_floats)) return FP_ERR_BOUNDS;
    if (!span_ok(op.scale_off, 15 * (int64_t)op.Cmid, weight_floats)) return FP_ERR_BOUNDS;
    if ((op.flags & FP_OPF_IN_DW) && !span_ok(op.bias_off, 12 * (int64_t)op.Cin, weight_floats)) return FP_ERR_BOUNDS;
    if (!span_ok(op.slope_off, x6 ? fp_dwblock_x6_wp_floats(op) : (int64_t)op.Cmid * op.Cout + 2 * (int64_t)op.Cout, weight_floats))
      return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_YSTEM || op.kind == FP_OP_YSTEM_U8) {
    if (op.Cin != (ext_in ? 3 : 4) || op.res_C <= 0 || op.res_C > 32 || op.Cout > 32) return FP_ERR_UNSUPPORTED;
    if (op.res_H <= 0 || op.res_W <= 0 || op.res_ld < op.res_C || op.res_ns < 0) return FP_ERR_INVALID_ARG;
    const int nb2 = fp_ystem_nb2(op);
    if (!span_ok(op.w_off, 40 * 32, weight_floats) || !span_ok(op.bias_off, 32, weight_floats)) return FP_ERR_BOUNDS;
    if (op.scale_off >= 0 && !span_ok(op.scale_off, 32, weight_floats)) return FP_ERR_BOUNDS;
    if (!span_ok(op.slope_off, (int64_t)nb2 * 16 * (32 + 2), weight_floats)) return FP_ERR_BOUNDS;
    const int64_t res_ext = (int64_t)(op.N - 1) * op.res_ns + ((int64_t)(op.OH / 2) * (op.OW / 2) - 1) * op.res_ld + op.res_C;
    if (!span_ok(op.res_off, res_ext, arena_floats)) return FP_ERR_BOUNDS;
  }
  if (op.kind == FP_OP_STEM_U8 && (op.Cin != 3 || op.res_H <= 0 || op.res_W < 3 || !fp_stem_u8_shape_ok(op)))
    return FP_ERR_UNSUPPORTED;
  if (op.kind == FP_OP_CONV || op.kind == FP_OP_DWCONV || op.kind == FP_OP_STEM_U8) {
    if (op.scale_off >= 0 && !span_ok(op.scale_off, Cout, weight_floats)) return FP_ERR_BOUNDS;
    if (op.bias_off >= 0 && !span_ok(op.bias_off, Cout, weight_floats)) return FP_ERR_BOUNDS;
    if (op.slope_off >= 0 && !span_ok(op.slope_off, Cout, weight_floats)) return FP_ERR_BOUNDS;
    if (op.act == FP_ACT_PRELU && op.slope_off < 0) return FP_ERR_INVALID_ARG;
    if (op.act < FP_ACT_NONE || op.act > FP_ACT_SILU) return FP_ERR_INVALID_ARG;
  }
  if ((op.kind == FP_OP_CONV || op.kind == FP_OP_BLAZEBLOCK || op.kind == FP_OP_DWPW || op.kind == FP_OP_DWBLOCK) &&
      op.res_mode != FP_RES_NONE) {
    if (op.res_mode < FP_RES_NONE || op.res_mode > FP_RES_SHUFFLE2) return FP_ERR_INVALID_ARG;
    if (op.res_mode == FP_RES_SHUFFLE2 &&
        ((op.kind != FP_OP_CONV && op.kind != FP_OP_DWPW) || op.out_cmul != 1 || op.out_ld < 2 * op.Cout))
      return FP_ERR_INVALID_ARG;
    if (op.res_C <= 0 || op.res_ld < op.res_C || op.res_ns < 0) return FP_ERR_INVALID_ARG;
    int rh = OH, rw = OW;
    if (op.res_mode == FP_RES_POOL2_BEFORE_ACT) {
      rh = op.res_H;
      rw = op.res_W;
      if (rh < 2 * OH || rw < 2 * OW) return FP_ERR_INVALID_ARG;
    }
    const int64_t res_ext = (int64_t)(op.N - 1) * op.res_ns + ((int64_t)rh * rw - 1) * op.res_ld + op.res_C;
    if (!span_ok(op.res_off, res_ext, arena_floats)) return FP_ERR_BOUNDS;
  }
  switch (op.kind) {
    case FP_OP_CONV:
    case FP_OP_DWCONV:
    case FP_OP_MAXPOOL:
    case FP_OP_UPSAMPLE2X:
    case FP_OP_COPY:
    case FP_OP_L2NORM:
    case FP_OP_BLAZEBLOCK:
    case FP_OP_DWPW:
    case FP_OP_DWBLOCK:
    case FP_OP_BLAZEPAIR:
    case FP_OP_BLAZECHAIN:
    case FP_OP_YSTEM:
    case FP_OP_YSTEM_U8:
    case FP_OP_STEM_U8:
    case FP_OP_SHUFDOWN:
    case FP_OP_SHUFUNIT:
    case FP_OP_YSTEM2:
      return FP_OK;
    default:
      return FP_ERR_UNSUPPORTED;
  }
}

// Name of the HIP kernel an op launches (the family rocprofv3's kernel trace shows), for measurement tools.
const char* fp_op_kernel_name(const fp_op* op) {
  static thread_local char buf[64];
  if (!op) return "?";
  switch (op->kind) {
    case FP_OP_CONV: {
      if (op->flags & FP_OPF_OUT_DW) return (op->flags & FP_OPF_SPLIT3) ? "stemdw_kernel<true>" : "stemdw_kernel<false>";
      if (op->flags & FP_OPF_SPLIT3) {
        if (fp_pwx6_eligible(*op))
          snprintf(buf, sizeof(buf), "pwx6_kernel<%d, %d, %s>", op->Cout == 48 ? 3 : op->Cout == 64 ? 4 : 8, fp_pwx6_mt(*op),
                   (op->flags & FP_OPF_IN_UP2) ? "true" : "false");
        else snprintf(buf, sizeof(buf), "convx6_kernel<%d>", fp_convx6_nt16(*op));
        return buf;
      }
      if (fp_pws_eligible(*op)) { snprintf(buf, sizeof(buf), "pws_kernel<%d, %d>", op->Cin, op->Cin == 64 ? 12 : 8); return buf; }
      if (fp_stem_eligible(*op)) {
        snprintf(buf, sizeof(buf), "stem_conv_kernel<%d, %d, false>", op->KH, (int)fp_round_up(op->Cout, 32) / 32);
        return buf;
      }
      if (fp_conv3_eligible(*op)) {
        snprintf(buf, sizeof(buf), "conv3_kernel<%d, %d, %s>", fp_conv3_nb(*op), op->stride, fp_conv3_t16(*op) ? "true" : "false");
        return buf;
      }
      int nb, vec, pwd;
      fp_conv_variant(*op, &nb, &vec, &pwd);
      snprintf(buf, sizeof(buf), "conv_igemm_kernel<%d, %s, %s>", nb, vec ? "true" : "false", pwd ? "true" : "false");
      return buf;
    }
    case FP_OP_DWCONV:
      if (op->KH == 3 && (op->stride == 1 || op->stride == 2)) snprintf(buf, sizeof(buf), "dwconv3_row_kernel<%d>", op->stride);
      else snprintf(buf, sizeof(buf), "dwconv_kernel<%d>", op->KH);
      return buf;
    case FP_OP_MAXPOOL: return "maxpool_kernel";
    case FP_OP_UPSAMPLE2X: return "upsample2x_kernel";
    case FP_OP_COPY:
      return (op->out_cmul == 1 && op->Cin % 4 == 0 && op->in_ld % 4 == 0 && op->out_ld % 4 == 0 && op->in_off % 4 == 0 &&
              op->out_off % 4 == 0 && op->in_ns % 4 == 0 && op->out_ns % 4 == 0) ? "copy4_kernel" : "copy_kernel";
    case FP_OP_L2NORM: return "l2norm_kernel";
    case FP_OP_BLAZEBLOCK:
      if (op->flags & FP_OPF_IN_ROWPAD) {
        if (fp_blazeblock_wp_eligible(*op)) return "blazeblock_wp_kernel<24, 4>";
        snprintf(buf, sizeof(buf), "blazeblock_wps_kernel<%d>", op->Cin);
        return buf;
      }
      if (fp_round_up(op->Cin, 8) <= 32 && fp_round_up(op->Cout, 32) == 32 &&
          fp_ceil_div((long)op->N * op->OH * op->OW, 128) >= 2048)
        snprintf(buf, sizeof(buf), "blazeblock_persist_kernel<%d, %s>", op->stride, fp_blazeblock_fixed24(*op) ? "24, 24" : "0, 0");
      else
        snprintf(buf, sizeof(buf), "blazeblock_kernel<%d>", (int)fp_round_up(op->Cout, 32) / 32);
      return buf;
    case FP_OP_DWPW:
      if (op->flags & FP_OPF_SPLIT3) { snprintf(buf, sizeof(buf), "dwpwx6_kernel<%d, 4, %d>", op->Cout / 16, op->stride); return buf; }
      if (fp_dwpw_persistent(*op)) {
        snprintf(buf, sizeof(buf), fp_dwpw_wave_private(*op) ? "dwpw_wp_kernel<%d, %d, 4>" : "dwpw_persist_kernel<%d, %d>",
                 (int)fp_round_up(op->Cout, 32) / 32, op->stride);
        return buf;
      }
      snprintf(buf, sizeof(buf), "dwpw_kernel<%d, %d, %d>", (int)fp_round_up(op->Cout, 32) / 32,
               (op->OW % 4 == 0) ? 4 : (op->OW % 2 == 0) ? 2 : 1, op->stride);
      return buf;
    case FP_OP_BLAZECHAIN:
      snprintf(buf, sizeof(buf), "blazechain96_kernel");
      return buf;
    case FP_OP_BLAZEPAIR:
      if (op->stride == 2) {
        snprintf(buf, sizeof(buf), "blazepair_s2_kernel<%d, %d>", op->W, op->Cout);
        return buf;
      }
      snprintf(buf, sizeof(buf), "blazepair_kernel<%d>", op->W);
      return buf;
    case FP_OP_DWBLOCK:
      if (op->flags & FP_OPF_SPLIT3) {
        if (op->stride == 2)
          snprintf(buf, sizeof(buf), "dwblock_x6d_kernel<%d, %d, %d, %d, %s>", op->Cin, op->Cmid, op->Cout, op->H,
                   (op->flags & FP_OPF_IN_DW) ? "true" : "false");
        else if (op->Cin == 128 && op->H == 7) snprintf(buf, sizeof(buf), "dwblock_x6q_kernel<%d>", op->H);
        else snprintf(buf, sizeof(buf), "dwblock_x6_kernel<%d, %d>", op->Cin, op->H);
        return buf;
      }
      snprintf(buf, sizeof(buf), "dwblock_kernel<%d, %d, %d, %d>", op->Cin, op->H, op->H == 28 ? 7 : op->H, op->H == 7 ? 3 : 1);
      return buf;
    case FP_OP_SHUFDOWN:
      snprintf(buf, sizeof(buf), "shufdown_x6_kernel<%d, %d, %s>", op->Cin / 32, op->Cmid, fp_get_knobs().shuf_ldsw ? "true" : "false");
      return buf;
    case FP_OP_SHUFUNIT:
      snprintf(buf, sizeof(buf), "shufunit_x6_kernel<%d>", op->Cmid);
      return buf;
    case FP_OP_YSTEM2: return "ystem2_x6_kernel";
    case FP_OP_STEM_U8:
      if (fp_stem_u8_band_eligible(*op)) return (op->flags & FP_OPF_SPLIT3) ? "stem5_u8_x6_kernel" : "stem5_u8_band_kernel";
      snprintf(buf, sizeof(buf), "stem_conv_kernel<%d, %d, true>", op->KH, (int)fp_round_up(op->Cout, 32) / 32);
      return buf;
    case FP_OP_YSTEM:
    case FP_OP_YSTEM_U8:
      snprintf(buf, sizeof(buf), "ystem_kernel<%d, %s>", fp_ystem_nb2(*op), op->kind == FP_OP_YSTEM_U8 ? "true" : "false");
      return buf;
    default: return "?";
  }
}

int fp_plan_validate(const fp_op* ops, int n_ops, size_t weight_floats, size_t arena_floats) {
  if (!ops || n_ops < 0) return FP_ERR_INVALID_ARG;
  for (int i = 0; i < n_ops; ++i) {
    int rc = validate_op(ops[i], weight_floats, arena_floats);
    if (rc != FP_OK) return rc;
  }
  return FP_OK;
}

static int launch_op(const fp_op& op, const float* weights, float* arena, const fp_ext* ext, int n_ext, hipStream_t s) {
  switch (op.kind) {
    case FP_OP_CONV: return fp_launch_conv(op, weights, arena, s);
    case FP_OP_DWCONV: return fp_launch_dwconv(op, weights, arena, s);
    case FP_OP_MAXPOOL: return fp_launch_maxpool(op, arena, s);
    case FP_OP_UPSAMPLE2X: return fp_launch_upsample2x(op, arena, s);
    case FP_OP_COPY: return fp_launch_copy(op, arena, s);
    case FP_OP_L2NORM: return fp_launch_l2norm(op, arena, s);
    case FP_OP_BLAZEBLOCK: return fp_launch_blazeblock(op, weights, arena, s);
    case FP_OP_DWPW:
      return (op.flags & FP_OPF_SPLIT3) ? fp_launch_dwpwx6(op, weights, arena, s) : fp_launch_dwpw(op, weights, arena, s);
    case FP_OP_DWBLOCK:
      return (op.flags & FP_OPF_SPLIT3) ? fp_launch_dwblock_x6(op, weights, arena, s) : fp_launch_dwblock(op, weights, arena, s);
    case FP_OP_BLAZEPAIR: return op.stride == 2 ? fp_launch_blazepair_s2(op, weights, arena, s) : fp_launch_blazepair(op, weights, arena, s);
    case FP_OP_BLAZECHAIN: return fp_launch_blazechain(op, weights, arena, s);
    case FP_OP_SHUFDOWN: return fp_launch_shufdown(op, weights, arena, s);
    case FP_OP_SHUFUNIT: return fp_launch_shufunit(op, weights, arena, s);
    case FP_OP_YSTEM2: return fp_launch_ystem2(op, weights, arena, s);
    case FP_OP_YSTEM: return fp_launch_ystem(op, weights, arena, s);
    case FP_OP_YSTEM_U8: return fp_launch_ystem_u8(op, weights, arena, ext, n_ext, s);
    case FP_OP_STEM_U8: return fp_launch_stem_u8(op, weights, arena, ext, n_ext, s);
    default: return FP_ERR_UNSUPPORTED;
  }
}

int fp_plan_run_ext(const fp_op* ops, int n_ops, const float* weights, size_t weight_floats, float* arena,
                    size_t arena_floats, const fp_ext* ext, int n_ext, void* stream) {
  if (!weights || !arena || n_ext < 0 || (n_ext > 0 && !ext)) return FP_ERR_INVALID_ARG;
  int rc = fp_plan_validate(ops, n_ops, weight_floats, arena_floats);
  if (rc != FP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_ops; ++i) {
    rc = launch_op(ops[i], weights, arena, ext, n_ext, s);
    if (rc != FP_OK) return rc;
  }
  return FP_OK;
}

int fp_plan_run(const fp_op* ops, int n_ops, const float* weights, size_t weight_floats, float* arena,
                size_t arena_floats, void* stream) {
  return fp_plan_run_ext(ops, n_ops, weights, weight_floats, arena, arena_floats, nullptr, 0, stream);
}

struct fp_timer {
  int n;
  hipEvent_t* start;
  hipEvent_t* stop;
  unsigned char* used;
};

int fp_timer_create(int n_ops, void** out) {
  if (!out || n_ops <= 0) return FP_ERR_INVALID_ARG;
  fp_timer* t = new fp_timer;
  t->n = n_ops;
  t->start = new hipEvent_t[n_ops];
  t->stop = new hipEvent_t[n_ops];
  t->used = new unsigned char[n_ops];
  for (int i = 0; i < n_ops; ++i) {
    t->used[i] = 0;
    if (hipEventCreate(&t->start[i]) != hipSuccess || hipEventCreate(&t->stop[i]) != hipSuccess) {
      fp_set_hip_error(hipGetLastError());
      return FP_ERR_LAUNCH;
    }
  }
  *out = t;
  return FP_OK;
}

void fp_timer_destroy(void* timer) {
  fp_timer* t = (fp_timer*)timer;
  if (!t) return;
  for (int i = 0; i < t->n; ++i) {
    (void)hipEventDestroy(t->start[i]);
    (void)hipEventDestroy(t->stop[i]);
  }
  delete[] t->start;
  delete[] t->stop;
  delete[] t->used;
  delete t;
}

int fp_plan_run_timed_ext(const fp_op* ops, int n_ops, const float* weights, size_t weight_floats, float* arena,
                          size_t arena_floats, const fp_ext* ext, int n_ext, void* stream, void* timer,
                          const unsigned char* op_mask) {
  fp_timer* t = (fp_timer*)timer;
  if (!weights || !arena || !t || !op_mask || t->n < n_ops || n_ext < 0 || (n_ext > 0 && !ext)) return FP_ERR_INVALID_ARG;
  int rc = fp_plan_validate(ops, n_ops, weight_floats, arena_floats);
  if (rc != FP_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  for (int i = 0; i < n_ops; ++i) {
    t->used[i] = op_mask[i];
    if (op_mask[i]) (void)hipEventRecord(t->start[i], s);
    rc = launch_op(ops[i], weights, arena, ext, n_ext, s);
    if (rc != FP_OK) return rc;
    if (op_mask[i]) (void)hipEventRecord(t->stop[i], s);
  }
  return FP_OK;
}

int fp_plan_run_timed(const fp_op* ops, int n_ops, const float* weights, size_t weight_floats, float* arena,
                      size_t arena_floats, void* stream, void* timer, const unsigned char* op_mask) {
  return fp_plan_run_timed_ext(ops, n_ops, weights, weight_floats, arena, arena_floats, nullptr, 0, stream, timer, op_mask);
}

int fp_timer_accumulate(void* timer, float* ms_accum, int n_ops) {
  fp_timer* t = (fp_timer*)timer;
  if (!t || !ms_accum || n_ops > t->n) return FP_ERR_INVALID_ARG;
  for (int i = 0; i < n_ops; ++i) {
    if (!t->used[i]) continue;
    if (hipEventSynchronize(t->stop[i]) != hipSuccess) {
      fp_set_hip_error(hipGetLastError());
      return FP_ERR_LAUNCH;
    }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t->start[i], t->stop[i]) != hipSuccess) {
      fp_set_hip_error(hipGetLastError());
      return FP_ERR_LAUNCH;
    }
    ms_accum[i] += ms;
  }
  return FP_OK;
}

}  // extern "C"
