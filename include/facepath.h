/*
 * facepath.h — C ABI of libfacepath.so (MI355X / gfx950 only).
 *
 * The drop-in boundary for the detect -> embed -> similarity-filter hot path of
 * SamSamhuns/face_detection_and_recognition.  The reference is 100 % Python and
 * has no native boundary of its own; every entry point below
 * therefore names the *Python* symbol of the reference whose arithmetic it
 * replaces (paths relative to the reference root, fde = face_detection_and_extraction).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - Every pointer is a DEVICE pointer unless the parameter is documented "host".
 *   - The caller owns every buffer (activations arena, weights, outputs).  The
 *     library never allocates device memory and keeps no mutable global state.
 *   - Stream-ordered: work is enqueued on `stream` (a hipStream_t passed as void*;
 *     NULL = the null stream) and the call returns without synchronising.
 *   - Return value: FP_OK (0) or a negative fp_status; fp_strerror() gives text.
 *   - Activations are NHWC fp32: element (n,y,x,c) of a view lives at
 *     base + n*ns + (y*W + x)*ld + c   (ns, ld in floats).
 */
#ifndef FACEPATH_H
#define FACEPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP_ABI_VERSION 11

typedef enum fp_status {
  FP_OK = 0,
  FP_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, unsupported shape */
  FP_ERR_BOUNDS = -2,        /* an op would touch memory outside the arena / weight blob */
  FP_ERR_UNSUPPORTED = -3,   /* op kind / parameter combination not implemented */
  FP_ERR_LAUNCH = -4,        /* hipGetLastError() != hipSuccess after a launch */
  FP_ERR_ALIGNMENT = -5      /* channel counts / strides / offsets not multiples of 4 floats where required */
} fp_status;

int fp_abi_version(void);
const char* fp_strerror(int status);
/* Last HIP error string seen by this thread's failing call ("" if none). */
const char* fp_last_hip_error(void);
/*
 * Host-only consistency check of the library's integer helpers (no GPU needed): the reciprocal-multiply division the
 * persistent kernels use for pixel -> (image, row, column) decode, fp_make_divisor / fp_fastdiv, against the hardware
 * division for every divisor 2 .. 4096 and selected ones up to 2^30, at the boundaries of each quotient and at 2^31 - 1.
 * Returns FP_OK or FP_ERR_INVALID_ARG (with the failing pair in fp_last_hip_error's buffer).
 */
int fp_selftest(void);
/*
 * Lab / test knobs (FP_CHAIN_GRID, FP_RESIZE_PER_PIXEL, FP_X6_QUARTER14, FP_X6_SPEC14, FP_PWX6_SMALL_MAXK) are read from
 * the environment ONCE, when the library is loaded; no launch calls getenv.  This re-reads them (tests that switch a
 * knob at run time call it after changing the environment).  Not for production use: it is the library's only
 * process-wide mutable state, and it must not be called while another thread is launching.
 */
void fp_debug_reload_env(void);

/* ------------------------------------------------------------------------- */
/* 1. Network plans: one flat array of ops, executed back-to-back on a stream. */
/* ------------------------------------------------------------------------- */

enum fp_op_kind {
  FP_OP_CONV = 1,       /* dense conv KHxKW (1x1 = pointwise GEMM), fp32 MFMA implicit GEMM */
  FP_OP_DWCONV = 2,     /* depthwise conv KHxKW */
  FP_OP_MAXPOOL = 3,    /* max pool, -inf padding, OH/OW given by caller (covers ceil_mode) */
  FP_OP_UPSAMPLE2X = 4, /* nearest-neighbour 2x */
  FP_OP_COPY = 5,       /* channel-slice copy (concat / chunk / shuffle interleave) */
  FP_OP_L2NORM = 6,     /* out[m,:] = in[m,:] / ||in[m,:]||_2 over Cin channels (mobile_facenet.py:30-33) */
  FP_OP_BLAZEBLOCK = 7, /* fused BlazeBlock: dw3x3 -> 1x1 -> (+shortcut) -> ReLU (blazeface.py:12-47) */
  FP_OP_DWPW = 8,       /* fused Depth_Wise tail: dw3x3(+BN,+PReLU) -> 1x1(+BN) [+x] (mobile_facenet.py:72-85) */
  FP_OP_YSTEM = 9,      /* head of YOLOv5-face's StemBlock (y5/models/common.py:58-73): stem_1 (3x3 s2, SiLU) kept in LDS ->
                           stem_2a (1x1, SiLU) -> out, and maxpool2x2(stem_1) -> res view (the concat half stem_3 reads) */
  FP_OP_YSTEM_U8 = 10,  /* FP_OP_YSTEM reading the u8 frames directly: the letterbox (fp_resize_normalize's arithmetic through
                           fp_letterbox_tables) happens while the input tile is staged; needs fp_plan_run_ext */
  FP_OP_STEM_U8 = 11,   /* first conv of a network (KxK in {3,5}, stride 2, Cout <= 64, dense NHWC output; BlazeFace's stem,
                           blazeface.py:118-120,195) reading the u8 frames the same way; needs fp_plan_run_ext */
  FP_OP_BLAZEPAIR = 13, /* TWO consecutive stride-1 24 -> 24 BlazeBlocks (blazeface.py:12-47,122-152) in one kernel: the tensor
                           between them stays in an LDS ring.  Row-padded input, 128- or 64-pixel-wide map; with stride = 2 the second block is
                           the stride-2 block that ends the stage (24 -> 24 / 48, half-size output); see "BLAZEPAIR" */
  FP_OP_BLAZECHAIN = 14, /* a RUN of fp_op.Cmid consecutive stride-1 96 -> 96 BlazeBlocks on the 16 x 16 map (blazeface.py:12-47,
                           146-152) in one kernel: one image per workgroup stays in LDS for the whole run.  See "BLAZECHAIN" */
  FP_OP_DWBLOCK = 12,   /* a WHOLE Depth_Wise block (mobile_facenet.py:67-88) in one kernel: 1x1 expand (+BN, PReLU) ->
                           dw3x3 stride 1 (+BN, PReLU) -> 1x1 project (+BN) [+ x]; the expanded tensor (Cmid channels)
                           lives in LDS only.  Shapes: see "DWBLOCK" below; anything else fails validation */
  FP_OP_YSTEM2 = 17,    /* ABI 10.  the tail of YOLOv5-face's StemBlock (y5/models/common.py:58-73) in one kernel: stem_2b (3x3 stride 2, c/2 -> c,
                           SiLU) -> cat with the pooled stem_1 map -> stem_3 (1x1, 2c -> c, SiLU); stem_2b's output never reaches HBM.  See "YSTEM2" */
  FP_OP_SHUFUNIT = 16,  /* ABI 10.  a WHOLE stride-1 ShuffleV2Block (y5/models/common.py:127-176): x1, x2 = x.chunk(2); branch2(x2) = 1x1 + BN
                           + SiLU -> dw3x3 + BN -> 1x1 + BN + SiLU; out = channel_shuffle(cat(x1, branch2)) in one kernel.  See "SHUFUNIT" */
  FP_OP_SHUFDOWN = 15   /* ABI 10.  a WHOLE stride-2 ShuffleV2Block (y5/models/common.py:127-176) in one kernel: branch1 = dw3x3 s2 + BN
                           -> 1x1 + BN + SiLU, branch2 = 1x1 + BN + SiLU -> dw3x3 s2 + BN -> 1x1 + BN + SiLU, cat + channel_shuffle(2)
                           as the store pattern; nothing but x and the shuffled output touches HBM.  See "SHUFDOWN" */
};

enum fp_act { FP_ACT_NONE = 0, FP_ACT_RELU = 1, FP_ACT_PRELU = 2, FP_ACT_SILU = 3 };

enum fp_res_mode {
  FP_RES_NONE = 0,
  FP_RES_ADD_BEFORE_ACT = 1, /* act(conv + res): BlazeBlock (blazeface.py:47) */
  FP_RES_ADD_AFTER_ACT = 2,  /* res + act(conv): Depth_Wise residual (mobile_facenet.py:84-85), Bottleneck (common.py:87) */
  FP_RES_POOL2_BEFORE_ACT = 3, /* res = maxpool2x2(input map) zero-padded on channels (blazeface.py:38-45) */
  FP_RES_SHUFFLE2 = 4        /* ShuffleV2Block tail, cat + channel_shuffle(2) (y5/models/common.py:169-176,21-31): the op
                                writes 2*Cout channels, out[2n] = res[n] (the other branch / pass-through half),
                                out[2n+1] = act(conv)[n]; out_cmul = 1, 16-byte stores (vector epilogue only) */
};

/*
 * One op.  All *_off fields are offsets in FLOATS: in/out/res into the
 * activation arena, w/scale/bias/slope into the weight blob (-1 = absent).
 * Epilogue of CONV / DWCONV:  v = acc * scale[c] + bias[c]  (scale absent = 1,
 * bias absent = 0), then residual/activation per res_mode/act; PReLU uses slope[c].
 * Output element (n, oy, ox, c) is written at
 *   out_off + n*out_ns + (oy*OW + ox)*out_ld + c*out_cmul        (out_cmul >= 1)
 * Zero padding: taps that fall outside [0,H)x[0,W) read 0 (MAXPOOL: -inf), so
 * asymmetric TFLite-style padding (blazeface.py:195, :38) is pad_t/pad_l plus OH/OW.
 */
typedef struct fp_op {
  int32_t kind;
  int32_t act;
  int32_t res_mode;
  int32_t N, H, W;        /* input batch / height / width */
  int32_t OH, OW;         /* output height / width */
  int32_t Cin, Cout;
  int32_t KH, KW, stride;
  int32_t pad_t, pad_l;
  int32_t in_ld, out_ld, res_ld;   /* per-pixel stride (floats) */
  int32_t out_cmul;                /* output channel multiplier (1 = dense, 2 = shuffle interleave) */
  int32_t res_C;                   /* channels available in the residual; c >= res_C adds 0 */
  int32_t res_H, res_W;            /* spatial dims of the residual map (POOL2: the un-pooled map) */
  int64_t in_ns, out_ns, res_ns;   /* per-image stride (floats) */
  int64_t in_off, out_off, res_off;
  int64_t w_off, scale_off, bias_off, slope_off;
  int32_t act2;                    /* fused ops: activation of the SECOND conv's output (DWPW: FP_ACT_NONE / FP_ACT_SILU) */
  int32_t flags;                   /* FP_OPF_* bits (0 for dense tensors) */
  int32_t Cmid;                    /* ABI 4.  DWBLOCK: channels of the expanded tensor (Depth_Wise `groups`); BLAZECHAIN: number of blocks; 0 elsewhere */
  int32_t reserved0;               /* ABI 4.  must be 0 */
} fp_op;

/*
 * Row-padded activation layout (ABI 3).  A tensor in this layout keeps one ZERO pixel after every image row, one zero
 * row above and one below every image, and one zero pixel in front of the first pad row:
 *     pixel (y, x) of image n  at  off + n*ns + (y*(W + 1) + x)*ld          y in [-1, H], x in [-1, W]
 * (x = -1 is the pad pixel of the row above, so a 3x3 window never needs a bounds check), with
 * ns >= ((H + 2)*(W + 1) + 1)*ld and off - (W + 2)*ld >= 0 the first float of image 0.  Producers write only
 * y in [0, H), x in [0, W); whoever owns the arena zeroes the pads once (plan.py allocates such buffers from a region
 * it never recycles).  The depthwise taps of the BlazeBlocks (blazeface.py:12-47) read it without clamps or masks.
 *   FP_OPF_IN_ROWPAD  : the input view is row-padded.  Accepted (exactly fp_plan_validate's list):
 *                       FP_OP_BLAZEBLOCK stride 1, 24 -> 24, OW % 32 == 0, OW >= 64, OH % 4 == 0, OH >= 8 (blazeblock_wp_kernel);
 *                       FP_OP_BLAZEBLOCK stride 1, 48 -> 48 or 96 -> 96, W = 16 or 32, H*W % 32 == 0, H*W >= 64
 *                       (blazeblock_wps_kernel); FP_OP_BLAZEPAIR (always).  Anything else: FP_ERR_UNSUPPORTED.
 *   FP_OPF_OUT_ROWPAD : the output view is row-padded (out_cmul = 1, out_ld = Cout).  Accepted: FP_OP_BLAZEBLOCK (every
 *                       kernel form), FP_OP_BLAZEPAIR, FP_OP_STEM_U8, FP_OP_CONV on the stem kernel (KxK stride 2 on a
 *                       4-float pixel), FP_OP_COPY with 16-byte-aligned views (Cin, strides and offsets multiples of 4:
 *                       copy4_kernel; the scalar copy cannot write the layout and fails validation).
 */
#define FP_OPF_IN_ROWPAD 1
#define FP_OPF_OUT_ROWPAD 2
/*
 * FP_OPF_IN_C3 : the op reads a 4-float pixel (Cin = in_ld = 4) whose FOURTH channel meets zero weights -- a 3-channel
 *                image padded to 16 bytes, as every network stem here is.  FP_OP_CONV on the stem kernel and
 *                FP_OP_YSTEM skip that channel's MFMAs (a quarter of the stem's matrix work); the result is
 *                unchanged.  The *_U8 ops (Cin = 3) always do.  Only valid with Cin = 4.
 */
#define FP_OPF_IN_C3 4
/*
 * FP_OPF_SPLIT3 (ABI 5) : the op's 1x1-conv weight matrices are packed as THREE bf16 planes (the exact three-way split
 *                of every fp32 weight, w == h + m + l with h = bf16(w), m = bf16(w - h), l = bf16(w - h - m), every conversion
 *                round-to-nearest-even; planes = the bf16 bit patterns of h, m, l) for the "bf16x6" kernels: fp32 operands on
 *                v_mfma_f32_16x16x32_bf16, six products per operand pair, fp32 accumulation; the three dropped products are
 *                < 2^-22.99 of a product and carry either sign (csrc/split.h; gfx950 has no TF32 and its fp32 MFMA runs at
 *                the vector rate).  Accepted (exactly fp_plan_validate's list): FP_OP_DWBLOCK stride 1 with (Cin, H = W) in
 *                {(128, 14), (128, 7), (64, 28)} and stride 2 with (Cin, Cmid, Cout, H) in {(64, 128, 64, 56),
 *                (64, 256, 128, 28), (128, 512, 128, 14)} (layouts under "DWBLOCK"); FP_OP_CONV 1x1 stride 1 or 3x3 pad 1
 *                stride 1 / 2, dense views, Cin and Cout multiples of 4 and >= 32, or 3x3 with Cin 8 / 16 / 24 (K flattened: k = tap * Cin
 *                + channel in slabs of 32) (weights [tap * ceil(Cin / 32) + slab][3
 *                planes][Npad][32] bf16, zero rows / columns in the padding); FP_OP_DWPW 3x3 pad 1 with Cin a multiple of
 *                32 (<= 256) and Cout 64 or 128; FP_OP_BLAZECHAIN (always); FP_OP_CONV with
 *                FP_OPF_OUT_DW (planes [Cout / 16][3][16][32] over k = tap * 3 + channel); FP_OP_STEM_U8 (ABI 11) in BlazeFace's
 *                band shape only (5x5 stride 2 pad 1 on a 256 x 256 canvas, Cout 24, bias + ReLU, N >= 16: stem5_u8_x6_kernel) with
 *                planes [3 slabs][2 channel tiles][3 planes][16][32] over k = 16 (ky - 2 slab) + 3 kx + channel (every ky padded
 *                to 16 k, the sixth ky and channels 24 .. 31 zero).  The semantics of the ops do not change.
 */
#define FP_OPF_SPLIT3 8
/*
 * FP_OPF_IN_DW (ABI 5) : an FP_OP_DWBLOCK whose input first passes through a depthwise 3x3 stride-1 Conv_block (conv + BN +
 *                PReLU, mobile_facenet.py:39-51) -- Mobile-FaceNet's conv2_dw in front of conv_23 (:107-108,141-143).  bias_off
 *                -> [12][Cin]: its nine taps (ky*3 + kx), BN scale, BN bias, PReLU slope.  Only with FP_OPF_SPLIT3 on the
 *                (Cin 64, Cmid 128, Cout 64, 56 x 56, stride 2) block: the depthwise output is formed in the kernel's
 *                prologue from an LDS image of the input rows and never goes to memory.
 */
#define FP_OPF_IN_DW 16
/*
 * FP_OPF_IN_UP2 (ABI 7) : a pointwise (1x1) FP_OP_CONV on the split-MFMA kernels (FP_OPF_SPLIT3) whose input is
 *                cat(upsample2x_nearest(u), v) -- nn.Upsample + Concat in front of a C3 in YOLOv5-face's head
 *                (y5/models/yolo.py:177-198, common.py:235-242): channels [0, res_C) of pixel (y, x) are read from pixel
 *                (y / 2, x / 2) of the res view (res_H = H / 2, res_W = W / 2, res_C a multiple of 8, res_ld, res_ns,
 *                res_off as for a residual; res_mode must be FP_RES_NONE), channels [res_C, Cin) from the in view as
 *                usual (in_off is the address of channel 0 of the concatenated row; its first res_C floats are never
 *                read).  The upsampled tensor does not exist.
 */
#define FP_OPF_IN_UP2 32
/*
 * FP_OPF_OUT_DW (ABI 9) : an FP_OP_CONV followed by a depthwise 3x3 stride-1 pad-1 Conv_block (conv + BN + PReLU,
 *                mobile_facenet.py:39-51) computed in the same kernel -- Mobile-FaceNet's conv1 + conv2_dw (:107-108, :141-142):
 *                the conv's output tile stays in LDS, the depthwise conv runs on it there and the op writes the DEPTHWISE
 *                output (same shape: OH x OW x Cout).  slope_off -> [Cout] PReLU slopes of the conv followed by [12][Cout] of
 *                the depthwise block: its nine taps (ky*3 + kx), BN scale, BN bias, PReLU slope.  Only the stem shape:
 *                3x3 stride 2 pad 1 on a dense 112 x 112 four-float-pixel image (FP_OPF_IN_C3), Cout 64, scale / bias / PReLU
 *                (csrc/stemdw.hip).
 */
#define FP_OPF_OUT_DW 64

/*
 * Weight blob layouts (packed by the host side, see
 * face_detection_and_recognition_amd/plan.py):
 *   CONV   : Wp[Kpad/4][Npad][4]  with k = (ky*KW + kx)*Cin + ci, Kpad = roundup(KH*KW*Cin, 8),
 *            Npad = roundup(Cout, 32); element (k, n) at ((k/4)*Npad + n)*4 + k%4; zero padded.
 *   DWCONV : Wd[KH*KW][C]
 *   scale / bias / slope : [Cout]
 *   DWPW   : w_off     -> [9][G] depthwise taps, [G] scale, [G] bias, [G] PReLU slope   (12*G floats)
 *            slope_off -> packed 1x1 weights as for CONV (K = G), then [roundup(Cout,4)] scale, [roundup(Cout,4)] bias
 *            act = FP_ACT_PRELU if the depthwise has a PReLU; res_mode = FP_RES_ADD_AFTER_ACT adds res after the 1x1.
 *            bias_off  -> optional [roundup(Cout,4)] PReLU slopes applied to the 1x1 output (a depthwise Conv_block
 *            followed by a 1x1 Conv_block, mobile_facenet.py:117-118,70); not combined with a residual.
 *            act2 = FP_ACT_SILU applies SiLU to the 1x1 output; res_mode = FP_RES_SHUFFLE2 then writes the tail of a
 *            ShuffleV2Block branch (y5/models/common.py:127-176: dw3x3 + BN -> 1x1 + BN + SiLU, cat with the other
 *            half, channel_shuffle(2)) exactly as FP_OP_CONV does: out[2n] = res[n], out[2n+1] = y[n].
 *   BLAZEBLOCK : w_off -> [9][Cin] taps, scale_off -> [Cin] depthwise bias, slope_off -> packed 1x1, bias_off -> [Cout]
 *   YSTEM  : in = the 4-float-pixel image (Cin = in_ld = 4), H and W multiples of 4; OH x OW = H/2 x W/2 (stem_1 / stem_2a
 *            map), Cout = stem_2a's physical channels (<= 32); the res_* view receives maxpool2x2(stem_1):
 *            res_C = stem_1's physical channels (<= 32), res_H x res_W = OH/2 x OW/2.
 *            w_off -> stem_1 packed as CONV (K = 36 -> 40, Npad = 32), scale_off (-1 = none) / bias_off -> [32];
 *            slope_off -> stem_2a: [2][4][NB2*16][4] with element e of (j, g, n) = W[n][16j + 4g + e] (NB2 =
 *            ceil(Cout/16), zero padded), then [NB2*16] scale, [NB2*16] bias.  Both convs end in SiLU.
 *   YSTEM_U8 : as YSTEM, but the input is EXTERNAL: in_off = index e into the ext[] array of fp_plan_run_ext with
 *            ext[e] = frames [N][fh][fw][3] u8, ext[e+1] = the tap tables fp_letterbox_tables wrote for an H x W canvas
 *            ((W + H + 2) x 8 bytes, W + H <= 2048), ext[e+2] = 256-float normalisation LUT.  H, W = the canvas (model input) size,
 *            Cin = 3, res_H = fh, res_W = fw (the pooled map is OH/2 x OW/2 as for YSTEM).
 *   STEM_U8 : a CONV (weights packed for Cin = 4: k = tap*4 + c, zero fourth channel; scale / bias / slope / act as
 *            CONV, no residual) whose H x W input is the letterbox canvas of external u8 frames: in_off = e with
 *            ext[e .. e+2] = frames, tap tables, LUT as for YSTEM_U8; Cin = 3, res_H = fh, res_W = fw.
 */

/*
 *   DWBLOCK : in = x [N][H][W][Cin] dense NHWC, out = y [N][H][W][Cout], Cout == Cin, stride 1, pad 1, KH = KW = 3.
 *            (Cin, H = W) in {(128, 14), (128, 7), (64, 28)} (every residual block of Mobile-FaceNet,
 *            mobile_facenet.py:119-122,126-130), Cmid a multiple of 32.  res_mode = FP_RES_ADD_AFTER_ACT adds x itself
 *            (the res_* view must be the in_* view), FP_RES_NONE omits the shortcut.
 *            w_off     -> expand weights packed as CONV (K = Cin, Npad = Cmid)
 *            scale_off -> [15][Cmid]: rows 0..2 expand BN scale / BN bias / PReLU slope, rows 3..11 the depthwise taps
 *                         (ky*3 + kx), rows 12..14 depthwise BN scale / BN bias / PReLU slope
 *            slope_off -> project weights packed as CONV (K = Cmid, Npad = Cout), then [Cout] BN scale, [Cout] BN bias
 *            With FP_OPF_SPLIT3 (bf16 elements, 2 per float of the blob; R = Cmid / 32 rounds of 32 expanded channels):
 *            w_off     -> [R][3 planes][Cin / 32][32 g][32 k]: plane p of expand weight (g = 32 r + g', k = 32 ks + k')
 *            slope_off -> [R][3 planes][Cout][32 g]: plane p of project weight (co, g = 32 r + g'), then (fp32) [Cout]
 *                         BN scale, [Cout] BN bias;  scale_off as above.
 *   BLAZEPAIR : in = x (row-padded, FP_OPF_IN_ROWPAD, 24 channels, W = 128 or 64, H a multiple of 8 with H / band rows
 *            >= 2), out = y2 (dense or FP_OPF_OUT_ROWPAD); act = FP_ACT_RELU, res_mode = FP_RES_ADD_BEFORE_ACT (each
 *            block's shortcut is its own input).  The two blocks' parameters back to back, each as for BLAZEBLOCK:
 *            w_off -> [2][9][24] taps, scale_off -> [2][24] depthwise bias, slope_off -> [2] packed 1x1 (K = 24,
 *            Npad = 32: 768 floats each), bias_off -> [2][24] 1x1 bias.
 *            With stride = 2 (ABI 8; csrc/blazepairs2.hip) the SECOND block is the stride-2 block that ends a stage
 *            (blazeface.py:34-47: F.pad(0, 2, 0, 2), depthwise stride 2, shortcut = 2 x 2 max pool padded on channels): in = x
 *            as above, out = y2 on the H/2 x W/2 map (dense or FP_OPF_OUT_ROWPAD, out_ld = Cout), Cout = 24 or 48, pad_t =
 *            pad_l = 0, res_mode = FP_RES_POOL2_BEFORE_ACT; parameters as above with the second block's 1x1 packed for its
 *            own width (K = 24, Npad = 32 or 64) and bias_off -> [24] then [Cout].
 *   BLAZECHAIN : in = x, out = y (both dense, 96 channels, H = W = 16; in and out may be the same view), Cmid = number of
 *            blocks (1..16); act = FP_ACT_RELU, res_mode = FP_RES_ADD_BEFORE_ACT (each block's shortcut is its own input);
 *            flags = FP_OPF_SPLIT3: the 1x1 convs run as bf16x6 split MFMAs.  w_off -> Cmid blocks back to back, each
 *            [1280 floats: [9][96] depthwise taps (ky*3 + kx), [96] depthwise bias, [96] 1x1 bias, 224 pad] followed by
 *            three slabs (k = 32 s .. 32 s + 31) of [3 planes][96 output channels][32 k] bf16 (13 824 floats).
 *   SHUFDOWN : in = x (H x W even, Cin = 32 channels), out = the block's output (OH x OW = H/2 x W/2, Cout = 128 = 2 * Cmid dense
 *            channels at out_off, out_ld >= 128), Cmid = 64 (the width of a branch), 3x3 stride 2 pad 1, act = act2 = FP_ACT_SILU,
 *            flags = FP_OPF_SPLIT3 (all three 1x1 convs run as bf16x6 split MFMAs).  One parameter block at w_off (floats; a weight
 *            plane holds two bf16 per float, planes in the order h, m, l of plan.py split3_bf16):
 *              [9][32] branch1 depthwise taps (ky*3 + kx), [32] BN scale, [32] BN bias;
 *              [3 planes][64 co][32 k] branch1 1x1;  [64] BN scale, [64] BN bias;
 *              [2 rounds][3 planes][32 g][32 k] branch2 first 1x1 (round r = output channels 32 r .. 32 r + 31);  [64] scale, [64] bias;
 *              [9][64] branch2 depthwise taps, [64] BN scale, [64] BN bias;
 *              [2 rounds][3 planes][64 co][32 g] branch2 second 1x1 (round r = input channels 32 r ..);  [64] scale, [64] bias.
 *            out[2 c] = branch1[c], out[2 c + 1] = branch2[c].
 *   SHUFUNIT : in = x and out = the block's output, both H x W x 128 dense channels (distinct buffers: tiles read their neighbours'
 *            pixels), Cin = Cout = 128, Cmid = 64, 3x3 stride 1 pad 1, act = act2 = FP_ACT_SILU, flags = FP_OPF_SPLIT3.  Parameter
 *            block at w_off: [2 rounds][3 planes][2 slabs][32 g][32 k] first 1x1 of branch2 (on channels 64 .. 127 of x);  [64] scale,
 *            [64] bias;  [9][64] depthwise taps, [64] BN scale, [64] BN bias;  [2 rounds][3 planes][64 co][32 g] second 1x1;  [64] scale,
 *            [64] bias.  out[2 c] = x[c] (c < 64), out[2 c + 1] = branch2[c].
 *   YSTEM2 : in = a (stem_2a's output: H x W even, Cin = 16 channels), res_* = the pooled stem_1 map (res_H x res_W = OH x OW = H/2 x W/2,
 *            res_C = 32 channels), out = stem_3's output (Cout = 32), 3x3 stride 2 pad 1, act = act2 = FP_ACT_SILU, flags = FP_OPF_SPLIT3.
 *            Parameter block at w_off: [5 slabs][3 planes][32 co][32 k] stem_2b with k = (ky*3 + kx)*16 + c (zero for k >= 144);
 *            [32] scale, [32] bias;  [2 slabs][3 planes][32 co][32 k] stem_3 (k: stem_2b's 32 channels, then the pooled map's);  [32] scale,
 *            [32] bias (scale = 1 where the BatchNorm is folded into the conv).
 * ABI history: 1 = round-1 ops; 2 = fp_ext / *_U8 ops (never shipped in a VERDICT-ed tree); 3 = fp_op.flags, row-padded
 * views; 4 = fp_op.Cmid / reserved0, FP_OP_DWBLOCK, FP_OP_BLAZEPAIR; 5 = FP_OPF_SPLIT3; 6 = FP_OP_BLAZECHAIN; 7 = FP_OPF_IN_UP2; 8 = FP_OP_BLAZEPAIR with stride 2; 9 = FP_OPF_OUT_DW, fp_debug_reload_env; 10 = FP_OP_SHUFDOWN, FP_OP_SHUFUNIT, FP_OP_YSTEM2; 11 = FP_OPF_SPLIT3 on FP_OP_STEM_U8.
 */

/* Validates every op against arena_floats / weight_floats, then launches them in order. */
int fp_plan_run(const fp_op* ops /*host*/, int n_ops,
                const float* weights, size_t weight_floats,
                float* arena, size_t arena_floats,
                void* stream);

/*
 * External buffers of a plan: device memory that is not part of the arena (the u8 frames a *_U8 op reads, its tap
 * tables and LUT).  fp_plan_run_ext = fp_plan_run with such buffers; ops that need them fail with FP_ERR_INVALID_ARG
 * under plain fp_plan_run.
 */
typedef struct fp_ext {
  const void* ptr;   /* device pointer */
  size_t bytes;      /* extent the library may read */
} fp_ext;

int fp_plan_run_ext(const fp_op* ops /*host*/, int n_ops,
                    const float* weights, size_t weight_floats,
                    float* arena, size_t arena_floats,
                    const fp_ext* ext /*host*/, int n_ext, void* stream);

/* Validation only (no GPU needed): same checks as fp_plan_run. */
int fp_plan_validate(const fp_op* ops /*host*/, int n_ops, size_t weight_floats, size_t arena_floats);

/*
 * Per-op timing for measurement (bench.py): fp_plan_run with a hipEvent pair recorded ON `stream`
 * around every op whose op_mask[i] != 0.  A timer owns n_ops event pairs (host objects, no device
 * memory); fp_timer_accumulate waits for the events of the last timed run of that timer and adds
 * each op's elapsed milliseconds into ms_accum[i] (host array).  Use one timer per in-flight run.
 */
int fp_timer_create(int n_ops, void** out_timer);
void fp_timer_destroy(void* timer);
int fp_plan_run_timed(const fp_op* ops /*host*/, int n_ops,
                      const float* weights, size_t weight_floats,
                      float* arena, size_t arena_floats,
                      void* stream, void* timer, const unsigned char* op_mask /*host, n_ops*/);
int fp_plan_run_timed_ext(const fp_op* ops /*host*/, int n_ops,
                          const float* weights, size_t weight_floats,
                          float* arena, size_t arena_floats,
                          const fp_ext* ext /*host*/, int n_ext,
                          void* stream, void* timer, const unsigned char* op_mask /*host, n_ops*/);
int fp_timer_accumulate(void* timer, float* ms_accum /*host, n_ops*/, int n_ops);
/* Name of the HIP kernel family an op launches (as rocprofv3's kernel trace shows it); thread-local buffer. */
const char* fp_op_kernel_name(const fp_op* op /*host*/);

/* ------------------------------------------------------------------------- */
/* 2. Image front end                                                          */
/* ------------------------------------------------------------------------- */

/*
 * Bilinear u8 resize of a source rectangle into a destination rectangle of an
 * fp32 NHWC canvas, then per-value LUT normalisation; pixels of the canvas
 * outside the destination rectangle are set to lut[pad_value].
 * Replaces: pad_resize_image (fde/modules/utils/image.py:31-59) + BGR->RGB
 * (fde/modules/blazeface/model.py:75, y5/__init__.py:15) + normalisation
 * (blazeface.py:248-250  x/127.5-1 ; y5/__init__.py:19-20  x/255 ;
 *  mobile_facenet/utils.py:13  (x-127.5)/127.5) and the per-face crop + resize of
 * extract_faces_from_dataset.py:289-303.  Resize arithmetic is OpenCV's
 * INTER_LINEAR u8 fixed-point scheme (11-bit coefficients); cv2 is absent in
 * the build container so this boundary is "parity unpinned" (DESIGN.md).
 *
 * One item = one canvas: item i resizes its source rectangle of frame src_image into the
 * destination rectangle (dx,dy,dw,dh) of canvas i (letterbox: the padded sub-rectangle; face crop:
 * the whole 112x112 canvas).  Items live in DEVICE memory (they may be produced by
 * fp_dets_to_crops); source rectangles are clamped to the frame inside the kernel.
 */
typedef struct fp_resize_item {
  int32_t src_image;   /* index into frames */
  int32_t sx, sy, sw, sh; /* source rectangle (pixels, inside the frame) */
  int32_t dx, dy, dw, dh; /* destination rectangle inside the canvas */
} fp_resize_item;

int fp_resize_normalize(const uint8_t* frames, int n_frames, int frame_h, int frame_w, /* [n,H,W,3] u8 */
                        const fp_resize_item* items /*device*/, int n_items,
                        float* canvas, int canvas_h, int canvas_w, int canvas_c /* >=3, extra channels zeroed */,
                        const float* lut256 /*device, 256 floats*/, int pad_value, int swap_rb,
                        void* stream);

/*
 * Tap tables of ONE resize geometry, for the network stems that read u8 frames themselves (FP_OP_YSTEM_U8,
 * FP_OP_STEM_U8): source rectangle (sx, sy, sw, sh) of a frame_h x frame_w frame -> destination rectangle
 * (dx, dy, dw, dh) of a canvas_h x canvas_w canvas, everything else pad_value.  Same arithmetic as
 * fp_resize_normalize (pad_resize_image, fde/modules/utils/image.py:31-59), evaluated once per column and row:
 * tables = (canvas_w + canvas_h + 2) entries of 2 int32 (layout: csrc/letterbox.h); frame_h, canvas_h <= 65535,
 * frame_w, canvas_w <= 32767.  The last entry records (frame_h, frame_w, canvas_h, canvas_w): a *_U8 op that is handed
 * tables built for another geometry writes NOTHING (a device-side check per workgroup; the host cannot read the device
 * buffer without a sync).  Trust boundary: the library checks the BYTE SIZES of the three external buffers and that
 * geometry record; that the frames buffer really holds N frames of res_H x res_W x 3 bytes is the caller's contract.
 */
int fp_letterbox_tables(int frame_h, int frame_w, int canvas_h, int canvas_w, int sx, int sy, int sw, int sh,
                        int dx, int dy, int dw, int dh, int pad_value, int swap_rb, int32_t* tables /*device*/,
                        void* stream);

/*
 * Detections -> face crop rectangles on device.  fmt 0: BlazeFace rows (ymin,xmin,ymax,xmax,...,score@16)
 * normalised to the model input — column reorder (fde/modules/blazeface/model.py:70) +
 * get_dets_bboxes_confs_lmarks_areas (fde/modules/utils/inference.py:11-58).  fmt 1: YOLOv5-face rows
 * (x1,y1,x2,y2,conf@4) in input pixels — get_bboxes_confs_areas (fde/modules/yolov5_face/onnx/onnx_utils.py:313-340).
 * Then scale_coords/clip/round (fde/modules/utils/image.py:62-99) and the crop arithmetic of
 * fde/face_extraction/extract_faces_from_dataset.py:289-303 (int(), offsets, clamp).  gain/pad are
 * scale_coords' values computed by the caller.  items[max_faces], face_info[max_faces][7] =
 * (frame, x1, y1, x2, y2 in original-frame pixels (rounded), conf, bbox area: fmt 0 the FRACTION of the model input
 * as inference.py:40-46 reports it (filter 100*(area/total) > thr); fmt 1 the PERCENT (100*area)/total as
 * onnx_utils.py:329-332 computes, compares and returns it), n_faces[1] = total found (the caller
 * must check n_faces <= max_faces).  Faces are ordered by (frame, detection).
 */
int fp_dets_to_crops(const float* dets, const int32_t* counts, int B, int max_dets, int row_floats, int fmt,
                     int in_w, int in_h, int orig_w, int orig_h, float det_thres, float area_thres,
                     float gain, float pad_x, float pad_y, int off_tx, int off_ty, int off_bx, int off_by,
                     int dst_w, int dst_h, int max_faces,
                     fp_resize_item* items, float* face_info, int32_t* n_faces, void* stream);

/* ------------------------------------------------------------------------- */
/* 3. BlazeFace post-processing                                                */
/* ------------------------------------------------------------------------- */

/*
 * _tensors_to_detections + _decode_boxes (blazeface.py:321-402): clamp +-clip,
 * sigmoid, score >= min_score_thresh, anchor decode.  Candidates are compacted
 * per image in anchor order into cand[B][num_anchors][17] / cand_count[B].
 */
int fp_blaze_decode(const float* raw_boxes /*[B,A,16]*/, const float* raw_scores /*[B,A]*/,
                    const float* anchors /*[A,4]*/, int B, int A,
                    float x_scale, float y_scale, float w_scale, float h_scale,
                    float score_clip, float min_score_thresh,
                    float* cand /*[B,A,17]*/, int32_t* cand_count /*[B]*/, void* stream);

/*
 * _weighted_non_max_suppression (blazeface.py:404-458) + overlap_similarity
 * (:463-526) for a batch: one workgroup per image.  dets [B][max_in][17] with
 * counts[B]; out [B][max_in][17], out_count[B]; member_of [B][max_in] (optional,
 * may be NULL) receives, per INPUT detection, the index of the output cluster it
 * was blended into (the bit-exact parity object).  Ties in score are ordered by
 * input index (stable).  A box whose self-IoU is not > thr (degenerate, SURVEY F8:
 * the reference never terminates) is emitted alone and removed.
 */
int fp_blaze_weighted_nms(const float* dets, const int32_t* counts, int B, int max_in,
                          float iou_thresh, float* out, int32_t* out_count, int32_t* member_of,
                          void* stream);

/* ------------------------------------------------------------------------- */
/* 4. YOLOv5-face post-processing                                              */
/* ------------------------------------------------------------------------- */

/*
 * Detect.forward inference decode (y5/models/yolo.py:62-108): heads are the raw
 * 1x1-conv outputs in NHWC [B,ny,nx,na*16]; out is [B, sum(na*ny*nx), 16] in the
 * reference's (level, anchor, y, x) order.
 */
int fp_yolo_decode(const float* head, int B, int ny, int nx, int na,
                   float stride, const float* anchors_px /*host [na][2]*/,
                   float* out, int64_t out_image_stride /*floats*/, int64_t out_row_offset /*rows*/,
                   void* stream);

/*
 * non_max_suppression_face (y5/utils/general.py:370-453), nc = 1: obj > conf,
 * conf = obj*cls > conf, xywh->xyxy, torchvision.ops.nms(iou) semantics
 * (greedy, IoU = inter/(a+b-inter), suppress IoU > thr, score-descending, ties by index).
 * pred [B][n_rows][16]; out [B][max_out][16] rows = (x1,y1,x2,y2,conf,lmk[10],cls);
 * out_count[B]; keep_idx [B][max_out] = row index into pred (bit-exact parity object);
 * overflow[B] is set to 1 when an image had more than max_cand candidates
 * (the host raises; the reference has no cap).
 */
int fp_yolo_nms(const float* pred, int B, int n_rows, float conf_thres, float iou_thres,
                int max_cand, int max_out,
                float* out, int32_t* out_count, int32_t* keep_idx, int32_t* overflow,
                void* scratch, size_t scratch_bytes, void* stream);
size_t fp_yolo_nms_scratch_bytes(int B, int max_cand);

/*
 * w_non_max_suppression (fde/modules/yolov5_face/onnx/onnx_utils.py:107-163), num_classes=1:
 * obj >= conf, +1-pixel IoU (:76-104), keep iou < nms_thres.  out rows = (x1,y1,x2,y2,obj,cls_conf,cls).
 */
int fp_yolo_w_nms(const float* pred, int B, int n_rows, float conf_thres, float nms_thres,
                  int max_cand, int max_out,
                  float* out /*[B][max_out][7]*/, int32_t* out_count, int32_t* keep_idx, int32_t* overflow,
                  void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* 5. Similarity filter                                                        */
/* ------------------------------------------------------------------------- */

/* inv_norm[m] = 1/||x[m,:]||  (cosine: extract_and_label_faces_from_dataset.py:106) */
int fp_row_inv_norm(const float* x, int64_t M, int D, float* inv_norm, void* stream);

/*
 * Cosine filter: for each gallery row g: best = max_j <g, r_j> * ginv[g] * rinv[j],
 * arg = the smallest j attaining it, keep = best >= tau.  S is never materialised.
 * packed is a caller-provided [M] uint64 scratch.
 */
int fp_cosine_filter(const float* G, const float* ginv, int64_t M,
                     const float* R, const float* rinv, int Nr, int D, float tau,
                     float* best, int32_t* arg, uint8_t* keep,
                     uint64_t* packed, void* stream);

/*
 * The same filter with S = G R^T on the bf16 matrix cores, fp32-equivalent split arithmetic (ABI 5, csrc/split.h): the
 * reference rows are split ONCE into three bf16 planes (fp_split3_rows: out holds fp_split3_bytes(Nr, D) bytes, layout
 * [D / 32][3][round_up(Nr, 128)][32] bf16, zero rows in the padding; D a multiple of 32), then every gallery batch runs
 * fp_cosine_filter_x6 against them.  Same results within fp32 rounding (best: 1e-6; arg: equal unless two scores tie within
 * it), 2x the rate of the fp32-MFMA kernel at 1 M x 10 k x 512.  rinv must be 16-byte aligned.
 */
size_t fp_split3_bytes(int Nr, int D);
int fp_split3_rows(const float* R, int Nr, int D, void* out, void* stream);
int fp_cosine_filter_x6(const float* G, const float* ginv, int64_t M,
                        const void* R3, const float* rinv, int Nr, int D, float tau,
                        float* best, int32_t* arg, uint8_t* keep,
                        uint64_t* packed, void* stream);

/*
 * get_ref_mean_vec_and_thres_from_imgs (sff/filter_faces_using_reference.py:71-100):
 * mean over the R reference rows, thres = max_i ||mean - f_i||_2.  out_mean [D], out_thres [1].
 */
int fp_l2_mean_thres(const float* ref, int R, int D, float* out_mean, float* out_thres, void* stream);

/* filter loop (sff/filter_faces_using_reference.py:183-197): dist = ||e - mean||_2 ; keep = dist <= thres. */
int fp_l2_filter(const float* E, int64_t M, int D, const float* mean, const float* thres /*device [1]*/,
                 float* dist, uint8_t* keep, void* stream);

/*
 * TF-style preprocess of the FaceNet filter (similar_face_filtering/filter_faces_using_reference.py:60-68):
 * frames [n][H][W][3] u8 RGB -> [0,1] -> tf.image.resize bilinear (half-pixel centres, no antialias) to
 * (out_h, out_w) -> per_image_standardization (x - mean)/max(std, 1/sqrt(out_h*out_w*3)).  out [n][out_h][out_w][3]
 * fp32; stats_scratch = 2*n doubles of device scratch.  The formula is pinned by the reference's own test
 * (sff/tests/base/test_similar_faces_filter.py:19-27); the resize is restated (TF absent offline).
 */
int fp_resize_standardize(const uint8_t* frames, int n, int H, int W, float* out, int out_h, int out_w,
                          double* stats_scratch, void* stream);

/*
 * Face crops from a float CHW image, the tail of the Triton python post-process model
 * (fde/modules/face_detection_trt_server/models/yolov5_face_postprocess/1/model.py:47-49,85-103): image (3,H,W) fp32
 * RGB in [0,1] -> *255, RGB->BGR -> crop [y:yh, x:xw] -> cv2.resize (float INTER_LINEAR) to (out_w, out_h) ->
 * (v - 127.5)/127.5 -> out [n][3][out_h][out_w].  boxes [n][4] int32 (x, y, xw, yh) in DEVICE memory, already clamped
 * to the image with xw > x and yh > y.  cv2 is absent offline: parity unpinned (restated in oracle/).
 */
int fp_crop_resize_f32(const float* image_chw, int H, int W, const int32_t* boxes, int n_boxes, float* out,
                       int out_h, int out_w, void* stream);

/*
 * Face-tracker matching for one frame (fde/face_extraction/extract_and_label_faces_from_dataset.py:101-121,
 * Net.check_if_face_exists + Net.add_face; IoU = fde/modules/utils/image.py:124-143).  The F new faces are
 * processed in order; each is compared with the known faces in insertion order and takes the FIRST one with
 * (dist < normal_thres && iou > 0.1) || dist < harsh_thres, whose stored feature and box it then replaces; without
 * a match it is appended (id = count + 1).  mode 0: dist = ||a - b||_2 (MOBILE_FACENET), mode 1: cosine distance.
 * feats [cap][D] fp32, bboxes [cap][4] int32 (x, y, xw, yh), count [1] int32 are device state updated in place;
 * ids[f] = 1-based face id (0 when the gallery is full), exists[f] = 1 if an existing face was matched.
 */
int fp_tracker_step(float* feats, int* bboxes, int* count, int cap, int D, const float* new_feats,
                    const int* new_bboxes, int F, int mode, float normal_thres, float harsh_thres,
                    int* ids, uint8_t* exists, void* stream);

/* ------------------------------------------------------------------------- */
/* 7. JPEG decode (the step in front of the path)           */
/* ------------------------------------------------------------------------- */

/*
 * Baseline JPEG decode, byte-identical to libjpeg-turbo's default decompressor -- what cv2.imread
 * (fde/modules/utils/inference.py:68-76, fde/face_extraction/extract_faces_from_dataset.py:393-420) and Pillow return.
 * Split as hardware decoders split it: marker parsing and Huffman decoding on the HOST (serial per scan; plain C, no GPU:
 * fp_jpeg_parse, fp_jpeg_entropy_decode -- run one thread per image), everything after it on the DEVICE
 * (fp_jpeg_reconstruct: dequantisation + the "islow" integer inverse DCT of jidctint.c, fancy chroma upsampling of
 * jdsample.c for 4:2:0 / 4:2:2, YCbCr -> RGB of jdcolor.c, csrc/jpeg.hip).  Accepted: SOF0 / SOF1 (sequential) and SOF2
 * (progressive: spectral selection and successive approximation), 8-bit, Huffman, any number of scans, 1 or 3 components, luma
 * sampling 1x1 / 2x1 / 2x2 with 1x1 chroma, restart intervals.  Everything else (arithmetic, lossless, 12-bit, CMYK, other
 * sampling layouts): FP_ERR_UNSUPPORTED, the caller decodes those on the host.
 */
typedef struct fp_jpeg_info {
  int32_t width, height, ncomp, restart_interval;
  int32_t progressive, reserved;      /* 1: SOF2 (several scans build the coefficients up) */
  int32_t hs[3], vs[3];               /* sampling factors */
  int32_t mcux, mcuy;                 /* MCUs per row / column */
  int32_t blocks_w[3], blocks_h[3];   /* 8 x 8 blocks per component, padded to whole MCUs */
  int32_t comp_w[3], comp_h[3];       /* a component's true size: ceil(image * samp / max samp) */
  int64_t coef_off[3];                /* component c: coefs + coef_off[c], [blocks_h][blocks_w][64] int16, natural (row-major) order */
  int64_t n_coefs;                    /* total int16 coefficients */
  uint16_t quant[3][64];              /* quantisation table of each component, natural order */
} fp_jpeg_info;

/* HOST.  Parses the markers up to the scan header. */
int fp_jpeg_parse(const uint8_t* data, size_t n, fp_jpeg_info* info);
/* HOST.  Huffman-decodes the scan into coefs (HOST memory, info->n_coefs int16: quantised, de-zigzagged). */
int fp_jpeg_entropy_decode(const uint8_t* data, size_t n, const fp_jpeg_info* info, int16_t* coefs);
/* Bytes of device workspace fp_jpeg_reconstruct needs (the components' sample planes). */
size_t fp_jpeg_workspace_bytes(const fp_jpeg_info* info);
/*
 * DEVICE.  coefs: the coefficients in DEVICE memory; info: HOST struct (passed by value to the kernels); workspace:
 * device, 8-byte aligned; out: device u8 [height][width][3], RGB (bgr = 0) or BGR (bgr = 1, cv2.imread's order);
 * a one-component image is replicated into the three channels (cv2.imread's default IMREAD_COLOR).
 */
int fp_jpeg_reconstruct(const int16_t* coefs, const fp_jpeg_info* info, uint8_t* workspace, size_t ws_bytes, uint8_t* out,
                        int bgr, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FACEPATH_H */
