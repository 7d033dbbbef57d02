// jpeg.hip — baseline JPEG decode, bit-exact with libjpeg-turbo's default decompressor (gfx950 + host).
//
// The step in FRONT of the hot path: the reference reads its frames with cv2.imread
// (fde/modules/utils/inference.py:68-76, fde/face_extraction/extract_faces_from_dataset.py:393-420) and tf.io.decode_jpeg
// (sff/filter_faces_using_reference.py:62) -- both are libjpeg(-turbo).  A batch of frames decoded with PIL / cv2 on the host
// costs 3-4 ms per 576 x 1024 frame and core; here the split is the one hardware decoders use:
//   host    marker parsing + Huffman decoding (inherently serial per scan) -> quantised DCT coefficients, int16
//           (fp_jpeg_parse, fp_jpeg_entropy_decode: plain C, no GPU; callers run one thread per image)
//   device  dequantisation + 8 x 8 inverse DCT (jidctint.c's "islow" integer transform, the library's default), fancy
//           (triangle-filter) chroma upsampling for 4:2:0 / 4:2:2 (jdsample.c h2v2_fancy_upsample / h2v1_fancy_upsample),
//           YCbCr -> RGB in 16-bit fixed point (jdcolor.c) -> interleaved u8 RGB or BGR frame (fp_jpeg_reconstruct)
// All three device stages are integer arithmetic restated from the library's published algorithm; the result is compared
// byte for byte with Pillow's decode (libjpeg-turbo) of the reference's own test images (tests/golden/jpeg, tests/test_jpeg.py).
// Scope: baseline / extended sequential, 8-bit, Huffman, one interleaved scan, 1 or 3 components, luma sampling 1x1 / 2x1 /
// 2x2 with 1x1 chroma, restart intervals.  Progressive, arithmetic-coded, 12-bit, CMYK and multi-scan files are refused
// (FP_ERR_UNSUPPORTED): the caller falls back to its host decoder for those.
#include <string.h>

#include "common.h"

namespace {

const unsigned char kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  // canonical Huffman table: 9-bit lookahead (code length << 8 | symbol, 0 = longer code) + the classic maxcode walk
  unsigned short look[512];
  int maxcode[18], valptr[17], mincode[17];
  unsigned char vals[256];
  bool present;
};

bool build_huff(Huff& h, const unsigned char* bits /*[1..16]*/, const unsigned char* vals, int nvals) {
  memset(&h, 0, sizeof(h));
  int code = 0, k = 0;
  unsigned short codes[256];
  unsigned char lens[256];
  for (int l = 1; l <= 16; ++l) {
    h.valptr[l] = k;
    h.mincode[l] = code;
    for (int i = 0; i < bits[l]; ++i) {
      if (k >= nvals || k >= 256) return false;
      codes[k] = (unsigned short)code;
      lens[k] = (unsigned char)l;
      ++k;
      ++code;
    }
    h.maxcode[l] = bits[l] ? code - 1 : -1;
    if (code > (1 << l)) return false;
    code <<= 1;
  }
  h.maxcode[17] = 0x7fffffff;
  if (k != nvals) return false;
  memcpy(h.vals, vals, nvals);
  for (int i = 0; i < k; ++i) {
    if (lens[i] <= 9) {
      const int base = codes[i] << (9 - lens[i]);
      for (int j = 0; j < (1 << (9 - lens[i])); ++j) h.look[base + j] = (unsigned short)((lens[i] << 8) | vals[i]);
    }
  }
  h.present = true;
  return true;
}

struct BitReader {
  const unsigned char* p;
  const unsigned char* end;
  unsigned long long acc;   // bits left-aligned at the top
  int nbits;
  int marker;               // a marker met in the entropy-coded data (0 = none): the reader feeds zeros behind it
  void init(const unsigned char* b, const unsigned char* e) { p = b, end = e, acc = 0, nbits = 0, marker = 0; }
  void fill() {
    while (nbits <= 56) {
      unsigned v = 0;
      if (!marker && p < end) {
        v = *p;
        if (v == 0xff) {
          const unsigned n = p + 1 < end ? p[1] : 0xd9;
          if (n == 0) p += 2;                  // stuffed zero
          else {
            marker = (int)n;                   // RSTn / EOI / ...: stop consuming, feed zeros
            v = 0;
          }
        } else {
          ++p;
        }
      }
      acc |= (unsigned long long)v << (56 - nbits);
      nbits += 8;
    }
  }
  inline unsigned peek(int n) { return (unsigned)(acc >> (64 - n)); }
  inline void skip(int n) { acc <<= n, nbits -= n; }
  inline int get(int n) {                     // n in 1..16
    if (nbits < n) fill();
    const unsigned v = peek(n);
    skip(n);
    return (int)v;
  }
};

inline int huff_decode(BitReader& br, const Huff& h) {
  if (br.nbits < 16) br.fill();
  const unsigned e = h.look[br.peek(9)];
  if (e) {
    br.skip(e >> 8);
    return e & 255;
  }
  int code = (int)br.peek(9), l = 9;
  br.skip(9);
  while (l < 17 && code > h.maxcode[l]) {
    code = (code << 1) | br.get(1);
    ++l;
  }
  if (l > 16) return -1;
  return h.vals[h.valptr[l] + code - h.mincode[l]];
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

inline unsigned be16(const unsigned char* p) { return ((unsigned)p[0] << 8) | p[1]; }

// Walks the markers up to (and including) SOS; fills info.  Returns the offset of the first entropy-coded byte or a
// negative fp_status.
long parse_headers(const unsigned char* d, size_t n, fp_jpeg_info& info, Huff* dc /*[4]*/, Huff* ac /*[4]*/) {
  memset(&info, 0, sizeof(info));
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return FP_ERR_INVALID_ARG;
  size_t pos = 2;
  bool have_sof = false;
  unsigned short qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0};
  while (pos + 4 <= n) {
    if (d[pos] != 0xff) return FP_ERR_INVALID_ARG;
    unsigned m = d[pos + 1];
    pos += 2;
    if (m == 0xff) {          // fill byte
      --pos;
      continue;
    }
    if (m == 0xd8 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;
    if (m == 0xd9) return FP_ERR_INVALID_ARG;   // EOI before SOS
    if (pos + 2 > n) return FP_ERR_INVALID_ARG;
    const unsigned len = be16(d + pos);
    if (len < 2 || pos + len > n) return FP_ERR_INVALID_ARG;
    const unsigned char* s = d + pos + 2;
    const unsigned char* e = d + pos + len;
    if (m == 0xdb) {          // DQT
      while (s < e) {
        const int pq = s[0] >> 4, tq = s[0] & 15;
        if (tq > 3 || pq > 1) return FP_ERR_INVALID_ARG;
        ++s;
        if (s + (pq ? 128 : 64) > e) return FP_ERR_INVALID_ARG;
        for (int i = 0; i < 64; ++i) {
          qt[tq][kZigzag[i]] = pq ? (unsigned short)be16(s + 2 * i) : s[i];   // stored in natural order
        }
        s += pq ? 128 : 64;
        have_qt[tq] = true;
      }
    } else if (m == 0xc0 || m == 0xc1) {   // SOF0 / SOF1: baseline / extended sequential, Huffman
      if (len < 8 || s[0] != 8) return FP_ERR_UNSUPPORTED;
      info.height = (int)be16(s + 1);
      info.width = (int)be16(s + 3);
      info.ncomp = s[5];
      if (info.width <= 0 || info.height <= 0) return FP_ERR_UNSUPPORTED;
      if ((info.ncomp != 1 && info.ncomp != 3) || len < 8u + 3u * info.ncomp) return FP_ERR_UNSUPPORTED;
      for (int c = 0; c < info.ncomp; ++c) {
        comp_id[c] = s[6 + 3 * c];
        info.hs[c] = s[7 + 3 * c] >> 4;
        info.vs[c] = s[7 + 3 * c] & 15;
        comp_tq[c] = s[8 + 3 * c];
        if (comp_tq[c] > 3) return FP_ERR_INVALID_ARG;
      }
      have_sof = true;
    } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
      return FP_ERR_UNSUPPORTED;   // progressive, lossless, arithmetic, differential
    } else if (m == 0xc4) {   // DHT
      while (s < e) {
        const int tc = s[0] >> 4, th = s[0] & 15;
        if (tc > 1 || th > 3 || s + 17 > e) return FP_ERR_INVALID_ARG;
        unsigned char bits[17];
        bits[0] = 0;
        int nv = 0;
        for (int i = 1; i <= 16; ++i) bits[i] = s[i], nv += s[i];
        if (nv > 256 || s + 17 + nv > e) return FP_ERR_INVALID_ARG;
        if (!build_huff(tc ? ac[th] : dc[th], bits, s + 17, nv)) return FP_ERR_INVALID_ARG;
        s += 17 + nv;
      }
    } else if (m == 0xdd) {   // DRI
      if (len != 4) return FP_ERR_INVALID_ARG;
      info.restart_interval = (int)be16(s);
    } else if (m == 0xda) {   // SOS
      if (!have_sof) return FP_ERR_INVALID_ARG;
      const int ns = s[0];
      if (ns != info.ncomp || len != 6u + 2u * ns) return FP_ERR_UNSUPPORTED;   // one interleaved scan with every component
      for (int c = 0; c < ns; ++c) {
        if (s[1 + 2 * c] != comp_id[c]) return FP_ERR_UNSUPPORTED;
        info.td[c] = s[2 + 2 * c] >> 4;
        info.ta[c] = s[2 + 2 * c] & 15;
        if (info.td[c] > 3 || info.ta[c] > 3 || !dc[info.td[c]].present || !ac[info.ta[c]].present) return FP_ERR_INVALID_ARG;
      }
      if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return FP_ERR_UNSUPPORTED;
      // geometry: luma 1x1 / 2x1 / 2x2, chroma 1x1
      if (info.ncomp == 1) {
        info.hs[0] = info.vs[0] = 1;            // a single-component scan is never interleaved: 8 x 8 MCUs
      } else {
        if (info.hs[1] != 1 || info.vs[1] != 1 || info.hs[2] != 1 || info.vs[2] != 1) return FP_ERR_UNSUPPORTED;
        if (!((info.hs[0] == 1 && info.vs[0] == 1) || (info.hs[0] == 2 && info.vs[0] == 1) ||
              (info.hs[0] == 2 && info.vs[0] == 2)))
          return FP_ERR_UNSUPPORTED;
      }
      info.mcux = (info.width + 8 * info.hs[0] - 1) / (8 * info.hs[0]);
      info.mcuy = (info.height + 8 * info.vs[0] - 1) / (8 * info.vs[0]);
      long off = 0;
      for (int c = 0; c < info.ncomp; ++c) {
        if (!have_qt[comp_tq[c]]) return FP_ERR_INVALID_ARG;
        memcpy(info.quant[c], qt[comp_tq[c]], 128);
        info.blocks_w[c] = info.mcux * info.hs[c];
        info.blocks_h[c] = info.mcuy * info.vs[c];
        info.coef_off[c] = off;
        off += (long)info.blocks_w[c] * info.blocks_h[c] * 64;
        // the component's true size (jdmaster.c: ceil(image * samp / max_samp)): what the fancy upsampler's edges see
        info.comp_w[c] = (info.width * info.hs[c] + info.hs[0] - 1) / info.hs[0];
        info.comp_h[c] = (info.height * info.vs[c] + info.vs[0] - 1) / info.vs[0];
      }
      info.n_coefs = off;
      return (long)(pos + len);
    }
    pos += len;
  }
  return FP_ERR_INVALID_ARG;
}

// ---------------------------------------------------------------------------------------------------------------------
// device side

struct JpegDevInfo {
  int width, height, ncomp, hs0, vs0;
  int blocks_w[3], blocks_h[3], comp_w[3], comp_h[3];
  long coef_off[3], plane_off[3];
  unsigned short quant[3][64];
};

// jidctint.c jpeg_idct_islow: CONST_BITS = 13, PASS1_BITS = 2
#define JF_0_298631336 2446
#define JF_0_390180644 3196
#define JF_0_541196100 4433
#define JF_0_765366865 6270
#define JF_0_899976223 7373
#define JF_1_175875602 9633
#define JF_1_501321110 12299
#define JF_1_847759065 15137
#define JF_1_961570560 16069
#define JF_2_053119869 16819
#define JF_2_562915447 20995
#define JF_3_072711026 25172

__device__ __forceinline__ int jdescale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one 1-D pass of the islow transform on eight values; SHIFT = the pass's descale
template <int SHIFT, bool FIRST>
__device__ __forceinline__ void idct8(const int in[8], int out[8]) {
  // even part
  int z2 = in[2], z3 = in[6];
  int z1 = (z2 + z3) * JF_0_541196100;
  const int tmp2 = z1 + z3 * (-JF_1_847759065);
  const int tmp3 = z1 + z2 * JF_0_765366865;
  z2 = in[0];
  z3 = in[4];
  const int tmp0 = (z2 + z3) << 13;
  const int tmp1 = (z2 - z3) << 13;
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  // odd part
  int t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
  z1 = t0 + t3;
  z2 = t1 + t2;
  z3 = t0 + t2;
  int z4 = t1 + t3;
  const int z5 = (z3 + z4) * JF_1_175875602;
  t0 *= JF_0_298631336;
  t1 *= JF_2_053119869;
  t2 *= JF_3_072711026;
  t3 *= JF_1_501321110;
  z1 *= -JF_0_899976223;
  z2 *= -JF_2_562915447;
  z3 *= -JF_1_961570560;
  z4 *= -JF_0_390180644;
  z3 += z5;
  z4 += z5;
  t0 += z1 + z3;
  t1 += z2 + z4;
  t2 += z2 + z3;
  t3 += z1 + z4;
  out[0] = jdescale(tmp10 + t3, SHIFT);
  out[7] = jdescale(tmp10 - t3, SHIFT);
  out[1] = jdescale(tmp11 + t2, SHIFT);
  out[6] = jdescale(tmp11 - t2, SHIFT);
  out[2] = jdescale(tmp12 + t1, SHIFT);
  out[5] = jdescale(tmp12 - t1, SHIFT);
  out[3] = jdescale(tmp13 + t0, SHIFT);
  out[4] = jdescale(tmp13 - t0, SHIFT);
}

// One thread per 8 x 8 block: dequantise, columns (descale 11), rows (descale 18), + 128, clamp -> the component's sample plane.
__global__ __launch_bounds__(64) void jpeg_idct_kernel(const short* coefs, unsigned char* planes, JpegDevInfo info, long nblocks_total) {
  const long b = (long)blockIdx.x * 64 + threadIdx.x;
  if (b >= nblocks_total) return;
  int c = 0;
  long bb = b;
  while (c + 1 < info.ncomp && bb >= (long)info.blocks_w[c] * info.blocks_h[c]) bb -= (long)info.blocks_w[c] * info.blocks_h[c], ++c;
  const int bx = (int)(bb % info.blocks_w[c]), by = (int)(bb / info.blocks_w[c]);
  const short* src = coefs + info.coef_off[c] + bb * 64;
  int ws[64];
#pragma unroll
  for (int col = 0; col < 8; ++col) {
    int in[8], out[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = (int)src[r * 8 + col] * (int)info.quant[c][r * 8 + col];
    idct8<11, true>(in, out);
#pragma unroll
    for (int r = 0; r < 8; ++r) ws[r * 8 + col] = out[r];
  }
  unsigned char* dst = planes + info.plane_off[c] + ((long)by * 8) * (info.blocks_w[c] * 8) + bx * 8;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int in[8], out[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) in[k] = ws[r * 8 + k];
    idct8<18, false>(in, out);
    unsigned long long packed = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int v = min(max(out[k] + 128, 0), 255);      // range_limit (jdmaster.c prepare_range_limit_table)
      packed |= (unsigned long long)v << (8 * k);
    }
    *(unsigned long long*)(dst + (long)r * (info.blocks_w[c] * 8)) = packed;
  }
}

// chroma sample for output pixel (oy, ox) by libjpeg's fancy upsampling (jdsample.c); plane row pitch = pitch, the
// component's true size = cw x ch (edges replicate inside it)
__device__ __forceinline__ int chroma_sample(const unsigned char* pl, int pitch, int cw, int ch, int oy, int ox, int hs0, int vs0) {
  if (hs0 == 1) return pl[(long)oy * pitch + ox];                      // 4:4:4 (vs0 == 1 too)
  const int cx = ox >> 1, hodd = ox & 1;
  if (vs0 == 1) {                                                      // h2v1_fancy_upsample
    const unsigned char* r = pl + (long)oy * pitch;
    const int cur = r[cx];
    if (cw == 1) return cur;
    if (!hodd) return cx == 0 ? cur : (cur * 3 + r[cx - 1] + 1) >> 2;
    return cx == cw - 1 ? cur : (cur * 3 + r[cx + 1] + 2) >> 2;
  }
  // h2v2_fancy_upsample: the nearer row counts 3/4, the further (above for even output rows, below for odd) 1/4
  const int cy = oy >> 1;
  const int cy1 = (oy & 1) ? min(cy + 1, ch - 1) : max(cy - 1, 0);
  const unsigned char* r0 = pl + (long)cy * pitch;
  const unsigned char* r1 = pl + (long)cy1 * pitch;
  const int cur = r0[cx] * 3 + r1[cx];
  if (!hodd) {
    if (cx == 0) return (cur * 4 + 8) >> 4;
    return (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
  }
  if (cx == cw - 1) return (cur * 4 + 7) >> 4;
  return (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
}

// One thread per output pixel: upsample + ycc_rgb_convert (jdcolor.c: SCALEBITS 16) -> interleaved u8
__global__ __launch_bounds__(256) void jpeg_color_kernel(const unsigned char* planes, unsigned char* out, JpegDevInfo info, int bgr) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)info.width * info.height) return;
  const int oy = (int)(i / info.width), ox = (int)(i - (long)oy * info.width);
  const int y = planes[info.plane_off[0] + (long)oy * (info.blocks_w[0] * 8) + ox];
  int r, g, b;
  if (info.ncomp == 1) {
    r = g = b = y;
  } else {
    const int cb = chroma_sample(planes + info.plane_off[1], info.blocks_w[1] * 8, info.comp_w[1], info.comp_h[1], oy, ox, info.hs0, info.vs0) - 128;
    const int cr = chroma_sample(planes + info.plane_off[2], info.blocks_w[2] * 8, info.comp_w[2], info.comp_h[2], oy, ox, info.hs0, info.vs0) - 128;
    // FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554, ONE_HALF = 32768
    r = y + ((91881 * cr + 32768) >> 16);
    b = y + ((116130 * cb + 32768) >> 16);
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    r = min(max(r, 0), 255);
    g = min(max(g, 0), 255);
    b = min(max(b, 0), 255);
  }
  unsigned char* o = out + i * 3;
  o[0] = (unsigned char)(bgr ? b : r);
  o[1] = (unsigned char)g;
  o[2] = (unsigned char)(bgr ? r : b);
}

}  // namespace

extern "C" {

int fp_jpeg_parse(const uint8_t* data, size_t n, fp_jpeg_info* info) {
  if (!data || !info) return FP_ERR_INVALID_ARG;
  Huff dc[4], ac[4];
  memset(dc, 0, sizeof(dc));
  memset(ac, 0, sizeof(ac));
  const long r = parse_headers(data, n, *info, dc, ac);
  return r < 0 ? (int)r : FP_OK;
}

int fp_jpeg_entropy_decode(const uint8_t* data, size_t n, const fp_jpeg_info* info_in, int16_t* coefs) {
  if (!data || !info_in || !coefs) return FP_ERR_INVALID_ARG;
  fp_jpeg_info info;
  Huff dc[4], ac[4];
  memset(dc, 0, sizeof(dc));
  memset(ac, 0, sizeof(ac));
  const long start = parse_headers(data, n, info, dc, ac);
  if (start < 0) return (int)start;
  if (info.n_coefs != info_in->n_coefs || info.width != info_in->width || info.height != info_in->height) return FP_ERR_INVALID_ARG;
  memset(coefs, 0, (size_t)info.n_coefs * sizeof(int16_t));
  BitReader br;
  br.init(data + start, data + n);
  int pred[3] = {0, 0, 0};
  int until_restart = info.restart_interval, next_rst = 0;
  for (int my = 0; my < info.mcuy; ++my) {
    for (int mx = 0; mx < info.mcux; ++mx) {
      if (info.restart_interval && until_restart == 0) {
        // byte-align, expect RSTn
        br.nbits = 0;
        br.acc = 0;
        if (!br.marker) {
          // (the reader stops at markers: unread bytes before one can only be padding)
          while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] != 0 && br.p[1] != 0xff)) ++br.p;
          if (br.p + 1 < br.end) br.marker = br.p[1];
        }
        if (br.marker != 0xd0 + next_rst) return FP_ERR_INVALID_ARG;
        br.p += 2;
        br.marker = 0;
        next_rst = (next_rst + 1) & 7;
        pred[0] = pred[1] = pred[2] = 0;
        until_restart = info.restart_interval;
      }
      for (int c = 0; c < info.ncomp; ++c) {
        const Huff& hd = dc[info.td[c]];
        const Huff& ha = ac[info.ta[c]];
        for (int v = 0; v < info.vs[c]; ++v) {
          for (int hh = 0; hh < info.hs[c]; ++hh) {
            int16_t* blk = coefs + info.coef_off[c] + ((long)(my * info.vs[c] + v) * info.blocks_w[c] + (mx * info.hs[c] + hh)) * 64;
            int s = huff_decode(br, hd);
            if (s < 0 || s > 11) return FP_ERR_INVALID_ARG;
            if (s) pred[c] += extend(br.get(s), s);
            blk[0] = (int16_t)pred[c];
            for (int k = 1; k < 64;) {
              const int rs = huff_decode(br, ha);
              if (rs < 0) return FP_ERR_INVALID_ARG;
              const int r = rs >> 4;
              s = rs & 15;
              if (s == 0) {
                if (r != 15) break;            // EOB
                k += 16;
                continue;
              }
              k += r;
              if (k > 63) return FP_ERR_INVALID_ARG;
              blk[kZigzag[k]] = (int16_t)extend(br.get(s), s);
              ++k;
            }
          }
        }
      }
      if (info.restart_interval) --until_restart;
    }
  }
  return FP_OK;
}

size_t fp_jpeg_workspace_bytes(const fp_jpeg_info* info) {
  if (!info) return 0;
  size_t b = 0;
  for (int c = 0; c < info->ncomp; ++c) b += (size_t)info->blocks_w[c] * 8 * info->blocks_h[c] * 8;
  return (b + 15) / 16 * 16;
}

int fp_jpeg_reconstruct(const int16_t* coefs, const fp_jpeg_info* info, uint8_t* workspace, size_t ws_bytes, uint8_t* out, int bgr,
                        void* stream) {
  if (!coefs || !info || !workspace || !out) return FP_ERR_INVALID_ARG;
  if (info->ncomp != 1 && info->ncomp != 3) return FP_ERR_UNSUPPORTED;
  if (ws_bytes < fp_jpeg_workspace_bytes(info)) return FP_ERR_BOUNDS;
  if (((uintptr_t)workspace) % 8 || ((uintptr_t)coefs) % 2) return FP_ERR_ALIGNMENT;
  JpegDevInfo d;
  memset(&d, 0, sizeof(d));
  d.width = info->width;
  d.height = info->height;
  d.ncomp = info->ncomp;
  d.hs0 = info->hs[0];
  d.vs0 = info->vs[0];
  long nblocks = 0, poff = 0;
  for (int c = 0; c < info->ncomp; ++c) {
    if (info->blocks_w[c] <= 0 || info->blocks_h[c] <= 0) return FP_ERR_INVALID_ARG;
    d.blocks_w[c] = info->blocks_w[c];
    d.blocks_h[c] = info->blocks_h[c];
    d.comp_w[c] = info->comp_w[c];
    d.comp_h[c] = info->comp_h[c];
    d.coef_off[c] = info->coef_off[c];
    d.plane_off[c] = poff;
    poff += (long)info->blocks_w[c] * 8 * info->blocks_h[c] * 8;
    nblocks += (long)info->blocks_w[c] * info->blocks_h[c];
    memcpy(d.quant[c], info->quant[c], 128);
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, s, (const short*)coefs, workspace, d, nblocks);
  FP_CHECK_LAUNCH();
  const long npx = (long)info->width * info->height;
  hipLaunchKernelGGL(jpeg_color_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, workspace, out, d, bgr);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // extern "C"
