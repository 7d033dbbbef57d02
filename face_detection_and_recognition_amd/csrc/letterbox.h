// letterbox.h — device-side pieces of pad_resize_image (fde/modules/utils/image.py:31-59) shared by the stand-alone
// resize kernel (image.hip) and the network stems that read u8 frames directly (ystem.hip, stem.hip).
//
// The resize is OpenCV's INTER_LINEAR scheme for 8-bit images: 11-bit fixed-point coefficients (round-half-even of
// w * 2048), horizontal pass in int, vertical pass ((b0*(h0>>4))>>16) + ((b1*(h1>>4))>>16) + 2) >> 2.
// When every canvas of a batch has the same geometry (a letterbox), the per-column and per-row taps are computed
// ONCE into tables by the same coef() the per-pixel kernel uses, so a fused stem reproduces the stand-alone kernel
// bit for bit.  Entries are 8 bytes so that a whole table pair fits in LDS next to a stem's tiles:
//   xtab[x] = { base, sh0 | sh1 << 3 | a0 << 6 | a1 << 18 | valid << 30 }
//             base = byte offset (inside a frame row) of an 8-byte window that holds both horizontal taps,
//             sh0 / sh1 = byte position of tap 0 / tap 1 inside it, a0 / a1 = the 11-bit fixed-point weights
//   ytab[y] = { row0 | row1 << 16, b0 | b1 << 12 | valid << 30 }      frame rows of the vertical taps, weights
// valid = 0 marks a canvas column (row) outside the destination rectangle: the pixel is the pad colour.
// The entry after the last row is a trailer { pad colour (u8 value), swap R/B }, the one after it the geometry the
// tables were built for { frame_h | frame_w << 16, canvas_h | canvas_w << 16 }: a stem that is handed tables of another
// geometry writes nothing (fp_lb_geometry_ok) instead of loading outside the frames.
#pragma once
#include "common.h"

struct __attribute__((aligned(8))) fp_lb_tap {
  int a, b;
};
#define FP_LB_VALID (1 << 30)

__device__ __forceinline__ void fp_lb_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.f;
    s = 0;
  }
  if (s >= ssize - 1) {
    f = 0.f;
    s = ssize - 1;
  }
  s0 = s;
  s1 = min(s + 1, ssize - 1);
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
}

// true if the tables (trailer entry W + H + 1) were built for this frame / canvas size
__device__ __forceinline__ bool fp_lb_geometry_ok(const fp_lb_tap* tabs, int W, int H, int frame_h, int frame_w) {
  const fp_lb_tap g = tabs[W + H + 1];
  return g.a == (frame_h | (frame_w << 16)) && g.b == (H | (W << 16));
}

// Vertical pass + clamp for one channel from the two horizontal sums.
__device__ __forceinline__ int fp_lb_vpass(int h0, int h1, int b0, int b1) {
  const int o = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
  return min(max(o, 0), 255);
}

// The two 8-byte windows (row0, row1) of one canvas pixel.  frame = first byte of the pixel's frame, row_bytes = W*3.
// Unaligned 8-byte global loads (the HSA ABI runs in unaligned access mode).
struct fp_lb_raw {
  unsigned lo0, hi0, lo1, hi1;
};

__device__ __forceinline__ void fp_lb_load8(const uint8_t* p, unsigned& lo, unsigned& hi) {
  unsigned long long v;
  __builtin_memcpy(&v, p, 8);
  lo = (unsigned)v;
  hi = (unsigned)(v >> 32);
}

__device__ __forceinline__ fp_lb_raw fp_lb_issue(const uint8_t* frame, long row_bytes, const fp_lb_tap& xt,
                                                 const fp_lb_tap& yt) {
  fp_lb_raw r;
  const bool ok = (xt.b & FP_LB_VALID) && (yt.b & FP_LB_VALID);   // pad pixels load (and ignore) the frame's first bytes
  const int base = ok ? xt.a : 0;
  const int row0 = ok ? (yt.a & 0xffff) : 0, row1 = ok ? (int)((unsigned)yt.a >> 16) : 0;
  fp_lb_load8(frame + (long)row0 * row_bytes + base, r.lo0, r.hi0);
  fp_lb_load8(frame + (long)row1 * row_bytes + base, r.lo1, r.hi1);
  return r;
}

// raw windows -> normalised RGB(+0) pixel.  lut: 256 floats (LDS or global); pad: the pad colour as a u8 value.
// All products fit 24-bit operands (weights <= 2048, taps <= 255, row sums <= 522240 >> 4), hence the mul24 forms.
__device__ __forceinline__ f32x4 fp_lb_finish(const fp_lb_raw& r, const fp_lb_tap& xt, const fp_lb_tap& yt,
                                              const float* lut, int pad, int swap_rb) {
  int v[3];
  if ((xt.b & FP_LB_VALID) && (yt.b & FP_LB_VALID)) {
    const unsigned sh0 = xt.b & 7u, sh1 = ((unsigned)xt.b >> 3) & 7u;
    const unsigned a0 = ((unsigned)xt.b >> 6) & 0xfffu, a1 = ((unsigned)xt.b >> 18) & 0xfffu;
    const unsigned b0 = (unsigned)yt.b & 0xfffu, b1 = ((unsigned)yt.b >> 12) & 0xfffu;
    // one v_perm_b32 pulls a tap's three channel bytes out of the 8-byte window (selector byte k = window byte k)
    const unsigned sel0 = 0x0c020100u + sh0 * 0x010101u, sel1 = 0x0c020100u + sh1 * 0x010101u;
    const unsigned t00 = __builtin_amdgcn_perm(r.hi0, r.lo0, sel0), t01 = __builtin_amdgcn_perm(r.hi0, r.lo0, sel1);
    const unsigned t10 = __builtin_amdgcn_perm(r.hi1, r.lo1, sel0), t11 = __builtin_amdgcn_perm(r.hi1, r.lo1, sel1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned h0 = __umul24((t00 >> (8 * c)) & 255u, a0) + __umul24((t01 >> (8 * c)) & 255u, a1);
      const unsigned h1 = __umul24((t10 >> (8 * c)) & 255u, a0) + __umul24((t11 >> (8 * c)) & 255u, a1);
      const int o = (int)(((__umul24(b0, h0 >> 4) >> 16) + (__umul24(b1, h1 >> 4) >> 16) + 2) >> 2);
      v[c] = min(max(o, 0), 255);
    }
  } else {
    v[0] = v[1] = v[2] = pad;
  }
  f32x4 o;
  o[0] = lut[swap_rb ? v[2] : v[0]];
  o[1] = lut[v[1]];
  o[2] = lut[swap_rb ? v[0] : v[2]];
  o[3] = 0.f;
  return o;
}
