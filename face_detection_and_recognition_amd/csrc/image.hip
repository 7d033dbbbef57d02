// image.hip — u8 frame -> normalised fp32 NHWC canvas (letterbox and per-face crop), gfx950.
//
// Replaces pad_resize_image (fde/modules/utils/image.py:31-59: cv2.resize bilinear + copyMakeBorder 125),
// the BGR->RGB flip and the three normalisations (blazeface.py:248-250, y5/__init__.py:15-20,
// mobile_facenet/utils.py:13-16) and the crop->resize of extract_faces_from_dataset.py:289-303.
// The resize follows OpenCV's INTER_LINEAR scheme for 8-bit images: 11-bit fixed-point coefficients
// (saturate_cast<short>(w * 2048), round-half-even), horizontal pass in int, vertical pass
//   ((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2.
// cv2 is absent offline, so this boundary is parity-unpinned against cv2 itself and pinned
// against oracle/image_ref.py, which restates the same arithmetic in numpy.
// resize_normalize_rows_kernel: a workgroup per (item, band of canvas rows), taps tabled once in LDS; the per-pixel
// kernel remains for canvases with other than 4 channels.
#include <stdlib.h>

#include "common.h"
#include "letterbox.h"

namespace {

struct ResizeArgs {
  const uint8_t* frames;
  const fp_resize_item* items;
  float* canvas;
  const float* lut;
  int n_frames, fh, fw, n_items, ch, cw, cc, pad_value, swap_rb;
};

__global__ __launch_bounds__(256) void resize_normalize_kernel(ResizeArgs p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)p.ch * p.cw;
  if (idx >= per * p.n_items) return;
  const int item = (int)(idx / per);
  const int pix = (int)(idx - (long)item * per);
  const int y = pix / p.cw, x = pix - y * p.cw;
  fp_resize_item it = p.items[item];
  // items live in device memory (they can come straight from the NMS kernels), so the source
  // rectangle is clamped here: a bad box can shrink the crop but never read outside the frame.
  it.sx = min(max(it.sx, 0), p.fw - 1);
  it.sy = min(max(it.sy, 0), p.fh - 1);
  it.sw = min(max(it.sw, 1), p.fw - it.sx);
  it.sh = min(max(it.sh, 1), p.fh - it.sy);
  int v[3];
  const bool inside = it.dw > 0 && it.dh > 0 && x >= it.dx && x < it.dx + it.dw && y >= it.dy &&
                      y < it.dy + it.dh && it.src_image >= 0 && it.src_image < p.n_frames;
  if (inside) {
    int sx0, sx1, ax0, ax1, sy0, sy1, by0, by1;
    fp_lb_coef(x - it.dx, (double)it.sw / (double)it.dw, it.sw, sx0, sx1, ax0, ax1);
    fp_lb_coef(y - it.dy, (double)it.sh / (double)it.dh, it.sh, sy0, sy1, by0, by1);
    const uint8_t* f = p.frames + (long)it.src_image * p.fh * p.fw * 3;
    const uint8_t* r0 = f + ((long)(it.sy + sy0) * p.fw + it.sx) * 3;
    const uint8_t* r1 = f + ((long)(it.sy + sy1) * p.fw + it.sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = (int)r0[sx0 * 3 + c] * ax0 + (int)r0[sx1 * 3 + c] * ax1;
      const int h1 = (int)r1[sx0 * 3 + c] * ax0 + (int)r1[sx1 * 3 + c] * ax1;
      v[c] = fp_lb_vpass(h0, h1, by0, by1);
    }
  } else {
    v[0] = v[1] = v[2] = p.pad_value;
  }
  float* o = p.canvas + ((long)item * per + pix) * p.cc;
  const float c0 = p.lut[p.swap_rb ? v[2] : v[0]];
  const float c1 = p.lut[v[1]];
  const float c2 = p.lut[p.swap_rb ? v[0] : v[2]];
  if (p.cc == 4) {
    f32x4 w = {c0, c1, c2, 0.f};
    *(f32x4*)o = w;
  } else {
    o[0] = c0;
    o[1] = c1;
    o[2] = c2;
    for (int c = 3; c < p.cc; ++c) o[c] = 0.f;
  }
}

// The same resize with the taps of an item computed ONCE per workgroup: a workgroup owns `rpb` canvas rows of one item,
// builds the item's column taps and its rows' taps in LDS with the coef() of the per-pixel kernel (the table format of
// letterbox.h: an 8-byte window per horizontal tap pair) and then turns out pixels with two 8-byte loads, four byte
// permutes and 24-bit integer multiplies each -- instead of two fp64 divisions, two coef() evaluations and twelve byte
// loads per pixel.  Bit-identical to resize_normalize_kernel (tests: test_resize_tabled_kernel_is_bit_exact).
struct ResizeRowsArgs {
  ResizeArgs a;
  int rpb, bpi;           // canvas rows per workgroup, workgroups per item
  fp_divisor cw_div;
};

__global__ __launch_bounds__(256) void resize_normalize_rows_kernel(ResizeRowsArgs q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* lut = (float*)smem_raw;                               // [256]
  fp_lb_tap* tabs = (fp_lb_tap*)(smem_raw + 1024);             // [cw] column taps, [rpb] row taps
  const ResizeArgs& p = q.a;
  const int tid = threadIdx.x;
  const int item = blockIdx.x / q.bpi, y0 = (blockIdx.x - item * q.bpi) * q.rpb;
  fp_resize_item it = p.items[item];
  it.sx = min(max(it.sx, 0), p.fw - 1);
  it.sy = min(max(it.sy, 0), p.fh - 1);
  it.sw = min(max(it.sw, 1), p.fw - it.sx);
  it.sh = min(max(it.sh, 1), p.fh - it.sy);
  const bool item_ok = it.dw > 0 && it.dh > 0 && it.src_image >= 0 && it.src_image < p.n_frames;
  lut[tid] = p.lut[tid];
  for (int i = tid; i < p.cw + q.rpb; i += 256) {
    fp_lb_tap t = {0, 0};
    if (i < p.cw) {
      if (item_ok && i >= it.dx && i < it.dx + it.dw) {
        int s0, s1, a0, a1;
        fp_lb_coef(i - it.dx, (double)it.sw / (double)it.dw, it.sw, s0, s1, a0, a1);
        const int off0 = (it.sx + s0) * 3, off1 = (it.sx + s1) * 3;
        const int base = min(off0, p.fw * 3 - 8);              // the 8-byte window never leaves the frame row
        t.a = base;
        t.b = (off0 - base) | ((off1 - base) << 3) | (a0 << 6) | (a1 << 18) | FP_LB_VALID;
      }
    } else {
      const int y = y0 + (i - p.cw);
      if (item_ok && y >= it.dy && y < it.dy + it.dh) {
        int s0, s1, b0, b1;
        fp_lb_coef(y - it.dy, (double)it.sh / (double)it.dh, it.sh, s0, s1, b0, b1);
        t.a = (it.sy + s0) | ((it.sy + s1) << 16);
        t.b = b0 | (b1 << 12) | FP_LB_VALID;
      }
    }
    tabs[i] = t;
  }
  __syncthreads();
  const uint8_t* frame = p.frames + (long)(item_ok ? it.src_image : 0) * p.fh * p.fw * 3;
  const long row_bytes = (long)p.fw * 3;
  const int rows = min(q.rpb, p.ch - y0);
  const int npx = rows * p.cw;
  float* out = p.canvas + ((long)item * p.ch + y0) * p.cw * 4;
  // four pixels per thread and pass: the eight window loads are issued before the first pixel is finished
  for (int i0 = tid; i0 < npx; i0 += 4 * 256) {
    fp_lb_raw raw[4];
    fp_lb_tap xt[4], yt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = min(i0 + 256 * k, npx - 1);
      const int r = (int)fp_fastdiv((unsigned)i, q.cw_div), x = i - r * p.cw;
      xt[k] = tabs[x];
      yt[k] = tabs[p.cw + r];
      raw[k] = fp_lb_issue(frame, row_bytes, xt[k], yt[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = i0 + 256 * k;
      if (i < npx) *(f32x4*)(out + (long)i * 4) = fp_lb_finish(raw[k], xt[k], yt[k], lut, p.pad_value, p.swap_rb);
    }
  }
}

}  // namespace

extern "C" int fp_resize_normalize(const uint8_t* frames, int n_frames, int frame_h, int frame_w,
                                   const fp_resize_item* items, int n_items, float* canvas, int canvas_h, int canvas_w,
                                   int canvas_c, const float* lut256, int pad_value, int swap_rb, void* stream) {
  if (!frames || !items || !canvas || !lut256) return FP_ERR_INVALID_ARG;
  if (n_frames <= 0 || frame_h <= 0 || frame_w <= 0 || n_items < 0 || canvas_h <= 0 || canvas_w <= 0 || canvas_c < 3)
    return FP_ERR_INVALID_ARG;
  if (pad_value < 0 || pad_value > 255) return FP_ERR_INVALID_ARG;
  if (n_items == 0) return FP_OK;
  ResizeArgs a{frames, items, canvas, lut256, n_frames, frame_h, frame_w, n_items, canvas_h, canvas_w, canvas_c,
               pad_value, swap_rb};
  // tabled form: 16-byte pixels, a frame row of at least 8 bytes, rows / offsets that fit the table fields (letterbox.h)
  if (canvas_c == 4 && frame_w >= 3 && frame_w <= 32767 && frame_h <= 65535 && canvas_w >= 2 && canvas_w <= 4096 &&
      ((uintptr_t)canvas) % 16 == 0 && !fp_get_knobs().resize_per_pixel) {
    ResizeRowsArgs q;
    q.a = a;
    q.rpb = (int)min((long)canvas_h, max(1L, 8192L / canvas_w));      // ~8 k pixels per workgroup
    q.bpi = (int)fp_ceil_div(canvas_h, q.rpb);
    q.cw_div = fp_make_divisor((unsigned)canvas_w);
    if ((long)n_items * q.bpi < (1L << 31)) {
      const size_t lds = 1024 + (size_t)(canvas_w + q.rpb) * sizeof(fp_lb_tap);
      hipLaunchKernelGGL(resize_normalize_rows_kernel, dim3((unsigned)((long)n_items * q.bpi)), dim3(256), lds, (hipStream_t)stream, q);
      FP_CHECK_LAUNCH();
      return FP_OK;
    }
  }
  const long total = (long)n_items * canvas_h * canvas_w;
  hipLaunchKernelGGL(resize_normalize_kernel, dim3((unsigned)fp_ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------
// Tap tables of ONE resize geometry (letterbox.h) for the stems that read u8 frames directly.
namespace {

struct LbTableArgs {
  fp_lb_tap* xtab;
  fp_lb_tap* ytab;
  int fw, cw, ch, sx, sy, sw, sh, dx, dy, dw, dh, pad_value, swap_rb, fh;
};

__global__ __launch_bounds__(256) void letterbox_tables_kernel(LbTableArgs p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < p.cw) {
    fp_lb_tap t = {0, 0};
    if (p.dw > 0 && i >= p.dx && i < p.dx + p.dw) {
      int s0, s1, a0, a1;
      fp_lb_coef(i - p.dx, (double)p.sw / (double)p.dw, p.sw, s0, s1, a0, a1);
      const int off0 = (p.sx + s0) * 3, off1 = (p.sx + s1) * 3;
      const int base = min(off0, p.fw * 3 - 8);        // the 8-byte window never leaves the frame row
      t.a = base;
      t.b = (off0 - base) | ((off1 - base) << 3) | (a0 << 6) | (a1 << 18) | FP_LB_VALID;
    }
    p.xtab[i] = t;
  } else if (i < p.cw + p.ch) {
    const int y = i - p.cw;
    fp_lb_tap t = {0, 0};
    if (p.dh > 0 && y >= p.dy && y < p.dy + p.dh) {
      int s0, s1, b0, b1;
      fp_lb_coef(y - p.dy, (double)p.sh / (double)p.dh, p.sh, s0, s1, b0, b1);
      t.a = (p.sy + s0) | ((p.sy + s1) << 16);
      t.b = b0 | (b1 << 12) | FP_LB_VALID;
    }
    p.ytab[y] = t;
  } else if (i == p.cw + p.ch) {
    const fp_lb_tap t = {p.pad_value, p.swap_rb};     // trailer: pad colour (u8 value), swap R/B
    p.xtab[i] = t;
  } else if (i == p.cw + p.ch + 1) {
    // geometry the taps were built for: the *_U8 stems refuse tables of another frame / canvas size (their 8-byte
    // window loads would leave the frames buffer)
    const fp_lb_tap t = {p.fh | (p.fw << 16), p.ch | (p.cw << 16)};
    p.xtab[i] = t;
  }
}

}  // namespace

extern "C" int fp_letterbox_tables(int frame_h, int frame_w, int canvas_h, int canvas_w, int sx, int sy, int sw, int sh,
                                   int dx, int dy, int dw, int dh, int pad_value, int swap_rb, int32_t* tables,
                                   void* stream) {
  if (!tables || frame_h <= 0 || frame_h > 65535 || frame_w < 3 || frame_w > 32767 || canvas_h <= 0 || canvas_w <= 0 ||
      canvas_h > 65535 || canvas_w > 32767)
    return FP_ERR_INVALID_ARG;
  if (sx < 0 || sy < 0 || sw <= 0 || sh <= 0 || sx + sw > frame_w || sy + sh > frame_h) return FP_ERR_INVALID_ARG;
  if (dw < 0 || dh < 0 || dx < 0 || dy < 0 || dx + dw > canvas_w || dy + dh > canvas_h) return FP_ERR_INVALID_ARG;
  if (pad_value < 0 || pad_value > 255) return FP_ERR_INVALID_ARG;
  LbTableArgs a{(fp_lb_tap*)tables, (fp_lb_tap*)tables + canvas_w, frame_w, canvas_w, canvas_h, sx, sy, sw, sh, dx, dy, dw, dh,
                pad_value, swap_rb != 0, frame_h};
  hipLaunchKernelGGL(letterbox_tables_kernel, dim3((unsigned)fp_ceil_div(canvas_w + canvas_h + 2, 256)), dim3(256), 0,
                     (hipStream_t)stream, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------
// Face crops from a float CHW image: the tail of the Triton post-process model
// (fde/modules/face_detection_trt_server/models/yolov5_face_postprocess/1/model.py:47-49,85-103):
//   image (3, H, W) fp32 RGB in [0,1]  ->  *255, RGB->BGR  ->  crop [y:yh, x:xw]  ->  cv2.resize(.., (ow, oh)) on
//   FLOAT data (INTER_LINEAR, half-pixel centres, float weights: horizontal pass then vertical)  ->  (v-127.5)/127.5
//   ->  (3, oh, ow) planes.
// One lane per output pixel, 3 channels.  cv2 is absent in the build container: "parity unpinned" like the u8 path;
// oracle/triton_postprocess_ref.py restates the same arithmetic.
namespace {

struct CropF32Args {
  const float* img;     // [3][H][W]
  const int* boxes;     // [n][4] x, y, xw, yh (already clamped to the image, xw > x, yh > y)
  float* out;           // [n][3][oh][ow]
  int H, W, n, oh, ow;
};

__global__ __launch_bounds__(256) void crop_resize_f32_kernel(CropF32Args p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const int per = p.oh * p.ow;
  if (idx >= (long)p.n * per) return;
  const int item = (int)(idx / per), pix = (int)(idx - (long)item * per);
  const int dy = pix / p.ow, dx = pix - dy * p.ow;
  const int x = p.boxes[item * 4 + 0], y = p.boxes[item * 4 + 1];
  const int sw = p.boxes[item * 4 + 2] - x, sh = p.boxes[item * 4 + 3] - y;
  // cv2 resizeGeneric_ / hresize / vresize for float: src coordinate (d + 0.5) * scale - 0.5 computed in double,
  // floor, weight in float; taps outside the crop collapse onto the border pixel with weight 0
  auto tap = [](int d, int dsize, int ssize, int& s0, int& s1, float& a1) {
    const double scale = (double)ssize / dsize;
    float fs = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(fs);
    fs -= s;
    if (s < 0) { s = 0; fs = 0.f; }
    if (s >= ssize - 1) { s = ssize - 1; fs = 0.f; }
    s0 = s;
    s1 = min(s + 1, ssize - 1);
    a1 = fs;
  };
  int sx0, sx1, sy0, sy1;
  float ax, ay;
  tap(dx, p.ow, sw, sx0, sx1, ax);
  tap(dy, p.oh, sh, sy0, sy1, ay);
  const float ax0 = 1.f - ax, ay0 = 1.f - ay;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* pl = p.img + (long)(2 - c) * p.H * p.W;   // BGR output channel c <- RGB plane 2-c
    const float* r0 = pl + (long)(y + sy0) * p.W + x;
    const float* r1 = pl + (long)(y + sy1) * p.W + x;
    const float h0 = (r0[sx0] * 255.0f) * ax0 + (r0[sx1] * 255.0f) * ax;
    const float h1 = (r1[sx0] * 255.0f) * ax0 + (r1[sx1] * 255.0f) * ax;
    const float v = h0 * ay0 + h1 * ay;
    p.out[((long)item * 3 + c) * per + pix] = (v - 127.5f) / 127.5f;
  }
}

}  // namespace

extern "C" int fp_crop_resize_f32(const float* image_chw, int H, int W, const int32_t* boxes, int n_boxes, float* out,
                                  int out_h, int out_w, void* stream) {
  if (!image_chw || H <= 0 || W <= 0 || n_boxes < 0 || out_h <= 0 || out_w <= 0) return FP_ERR_INVALID_ARG;
  if (n_boxes == 0) return FP_OK;
  if (!boxes || !out) return FP_ERR_INVALID_ARG;
  CropF32Args a{image_chw, boxes, out, H, W, n_boxes, out_h, out_w};
  const long total = (long)n_boxes * out_h * out_w;
  hipLaunchKernelGGL(crop_resize_f32_kernel, dim3((unsigned)fp_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

// ---------------------------------------------------------------------------------------------
// TF-style preprocess of the reference's FaceNet filter (similar_face_filtering/filter_faces_using_reference.py:60-68,
// SURVEY row R1): u8 RGB -> [0,1] fp32 (convert_image_dtype) -> tf.image.resize bilinear (TF2: half-pixel centres, no
// antialias) -> tf.image.per_image_standardization: (x - mean) / max(std, 1/sqrt(N)), N = oh*ow*3, population std.
// Pass 1 resizes and accumulates sum / sum of squares per image in fp64 (wave shuffle + one atomicAdd per wave);
// pass 2 normalises in place.  TF is absent offline; the standardisation formula is the reference's own test's
// (sff/tests/base/test_similar_faces_filter.py:19-27), the resize is restated (parity unpinned).
namespace {

struct StdArgs {
  const uint8_t* frames;  // [n][H][W][3]
  float* out;             // [n][oh][ow][3]
  double* stats;          // [n][2] sum, sum of squares (zeroed by the caller side of the C entry)
  int n, H, W, oh, ow;
};

__global__ __launch_bounds__(256) void resize_f01_stats_kernel(StdArgs p) {
  const int per = p.oh * p.ow;
  const int img = blockIdx.y;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  double s = 0.0, s2 = 0.0;
  if (pix < per) {
    const int dy = pix / p.ow, dx = pix - dy * p.ow;
    const float sy = (float)p.H / (float)p.oh, sx = (float)p.W / (float)p.ow;
    const float fy = ((float)dy + 0.5f) * sy - 0.5f, fx = ((float)dx + 0.5f) * sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int y0 = max((int)fly, 0), y1 = min((int)ceilf(fy), p.H - 1);
    const int x0 = max((int)flx, 0), x1 = min((int)ceilf(fx), p.W - 1);
    const float ly = fy - fly, lx = fx - flx;
    const uint8_t* f = p.frames + (long)img * p.H * p.W * 3;
    float* o = p.out + ((long)img * per + pix) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tl = f[((long)y0 * p.W + x0) * 3 + c] * (1.0f / 255.0f), tr = f[((long)y0 * p.W + x1) * 3 + c] * (1.0f / 255.0f);
      const float bl = f[((long)y1 * p.W + x0) * 3 + c] * (1.0f / 255.0f), br = f[((long)y1 * p.W + x1) * 3 + c] * (1.0f / 255.0f);
      const float top = tl + (tr - tl) * lx, bot = bl + (br - bl) * lx;
      const float v = top + (bot - top) * ly;
      o[c] = v;
      s += v;
      s2 += (double)v * v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    s2 += __shfl_xor(s2, off);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&p.stats[img * 2 + 0], s);
    atomicAdd(&p.stats[img * 2 + 1], s2);
  }
}

__global__ __launch_bounds__(256) void standardize_kernel(StdArgs p) {
  const long per = (long)p.oh * p.ow * 3;
  const int img = blockIdx.y;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= per) return;
  const double mean = p.stats[img * 2 + 0] / (double)per;
  double var = p.stats[img * 2 + 1] / (double)per - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float adj = fmaxf((float)sqrt(var), 1.0f / sqrtf((float)per));
  float* o = p.out + (long)img * per + i;
  *o = (*o - (float)mean) / adj;
}

}  // namespace

extern "C" int fp_resize_standardize(const uint8_t* frames, int n, int H, int W, float* out, int out_h, int out_w,
                                     double* stats_scratch, void* stream) {
  if (n < 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return FP_ERR_INVALID_ARG;
  if (n == 0) return FP_OK;
  if (!frames || !out || !stats_scratch) return FP_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(stats_scratch, 0, sizeof(double) * 2 * n, s) != hipSuccess) {
    fp_set_hip_error(hipGetLastError());
    return FP_ERR_LAUNCH;
  }
  StdArgs a{frames, out, stats_scratch, n, H, W, out_h, out_w};
  hipLaunchKernelGGL(resize_f01_stats_kernel, dim3((unsigned)fp_ceil_div((long)out_h * out_w, 256), (unsigned)n), dim3(256), 0, s, a);
  FP_CHECK_LAUNCH();
  hipLaunchKernelGGL(standardize_kernel, dim3((unsigned)fp_ceil_div((long)out_h * out_w * 3, 256), (unsigned)n), dim3(256), 0, s, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
