// jpeg.hip — JPEG decode (sequential and progressive Huffman files), bit-exact with libjpeg-turbo's default decompressor
// (gfx950 + host).
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
// Scope: sequential (SOF0 / SOF1) and PROGRESSIVE (SOF2: spectral selection + successive approximation, jdphuff.c) Huffman
// files, 8-bit, any number of scans, 1 or 3 components, luma sampling 1x1 / 2x1 / 2x2 with 1x1 chroma, restart intervals.
// Arithmetic-coded, lossless, 12-bit, CMYK and other sampling layouts are refused (FP_ERR_UNSUPPORTED): the caller falls back
// to its host decoder for those.
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
  // sequential AC decoding: 10 bits of lookahead that hold a whole run/size code AND its value bits decode in one step:
  // (value << 8) | (run << 4) | bits consumed, 0 = take the two-step path (the same idea as stb_image's fast_ac)
  short fast[1024];
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
  for (int i = 0; i < 1024; ++i) {
    const unsigned e = h.look[i >> 1];
    const int len = (int)(e >> 8), run = (int)((e >> 4) & 15), sz = (int)(e & 15);
    if (e && sz && len + sz <= 10) {
      int v = (i >> (10 - len - sz)) & ((1 << sz) - 1);
      if (v < (1 << (sz - 1))) v += 1 - (1 << sz);
      if (v >= -128 && v <= 127) h.fast[i] = (short)((v * 256) | (run << 4) | (len + sz));
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
    if (!marker && end - p >= 8) {             // eight bytes ahead with no 0xff among them: take the whole bytes that fit at once
      unsigned long long w;
      __builtin_memcpy(&w, p, 8);
      if (!((~w - 0x0101010101010101ull) & w & 0x8080808080808080ull)) {
        const int k = (64 - nbits) >> 3;
        if (k > 0) {
          w = __builtin_bswap64(w);
          acc |= (w >> nbits) & (~0ull << (64 - nbits - 8 * k));
          p += k, nbits += 8 * k;
        }
        return;
      }
    }
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

// One block of a sequential scan (jdhuff.c decode_mcu): DC difference, then run / size pairs up to EOB.
inline int decode_block_seq(BitReader& br, const Huff& hd, const Huff& ha, int& pred, int16_t* blk) {
  int s = huff_decode(br, hd);
  if (s < 0 || s > 11) return FP_ERR_INVALID_ARG;
  if (s) pred += extend(br.get(s), s);
  blk[0] = (int16_t)pred;
  for (int k = 1; k < 64;) {
    if (br.nbits < 32) br.fill();                  // a code (<= 16 bits) and its value bits (<= 11) are in the accumulator
    const int fe = ha.fast[br.peek(10)];
    if (fe) {
      k += (fe >> 4) & 15;
      if (k > 63) return FP_ERR_INVALID_ARG;
      blk[kZigzag[k++]] = (int16_t)(fe >> 8);
      br.skip(fe & 15);
      continue;
    }
    const int rs = huff_decode(br, ha);
    if (rs < 0) return FP_ERR_INVALID_ARG;
    const int r = rs >> 4;
    s = rs & 15;
    if (s == 0) {
      if (r != 15) break;                          // EOB
      k += 16;
      continue;
    }
    k += r;
    if (k > 63) return FP_ERR_INVALID_ARG;
    blk[kZigzag[k++]] = (int16_t)extend(br.get(s), s);
  }
  return FP_OK;
}

inline unsigned be16(const unsigned char* p) { return ((unsigned)p[0] << 8) | p[1]; }

struct Frame {               // what the frame header (SOF) and the tables in front of the first scan say
  int comp_id[3], comp_tq[3];
  unsigned short qt[4][64];
  bool have_qt[4];
  bool have_sof;
};

// DQT / DHT / DRI segment bodies (s .. e)
int read_dqt(const unsigned char* s, const unsigned char* e, Frame& fr) {
  while (s < e) {
    const int pq = s[0] >> 4, tq = s[0] & 15;
    if (tq > 3 || pq > 1) return FP_ERR_INVALID_ARG;
    ++s;
    if (s + (pq ? 128 : 64) > e) return FP_ERR_INVALID_ARG;
    for (int i = 0; i < 64; ++i) fr.qt[tq][kZigzag[i]] = pq ? (unsigned short)be16(s + 2 * i) : s[i];   // natural order
    s += pq ? 128 : 64;
    fr.have_qt[tq] = true;
  }
  return FP_OK;
}

int read_dht(const unsigned char* s, const unsigned char* e, Huff* dc, Huff* ac) {
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
  return FP_OK;
}

int read_sof(const unsigned char* s, unsigned len, unsigned m, fp_jpeg_info& info, Frame& fr) {
  if (len < 8 || s[0] != 8) return FP_ERR_UNSUPPORTED;       // 8-bit samples only
  info.progressive = m == 0xc2;
  info.height = (int)be16(s + 1);
  info.width = (int)be16(s + 3);
  info.ncomp = s[5];
  if (info.width <= 0 || info.height <= 0) return FP_ERR_UNSUPPORTED;
  if ((info.ncomp != 1 && info.ncomp != 3) || len < 8u + 3u * info.ncomp) return FP_ERR_UNSUPPORTED;
  for (int c = 0; c < info.ncomp; ++c) {
    fr.comp_id[c] = s[6 + 3 * c];
    info.hs[c] = s[7 + 3 * c] >> 4;
    info.vs[c] = s[7 + 3 * c] & 15;
    fr.comp_tq[c] = s[8 + 3 * c];
    if (fr.comp_tq[c] > 3) return FP_ERR_INVALID_ARG;
  }
  // geometry: luma 1x1 / 2x1 / 2x2, chroma 1x1
  if (info.ncomp == 1) {
    info.hs[0] = info.vs[0] = 1;            // a single-component image is never interleaved: 8 x 8 MCUs
  } else {
    if (info.hs[1] != 1 || info.vs[1] != 1 || info.hs[2] != 1 || info.vs[2] != 1) return FP_ERR_UNSUPPORTED;
    if (!((info.hs[0] == 1 && info.vs[0] == 1) || (info.hs[0] == 2 && info.vs[0] == 1) || (info.hs[0] == 2 && info.vs[0] == 2)))
      return FP_ERR_UNSUPPORTED;
  }
  info.mcux = (info.width + 8 * info.hs[0] - 1) / (8 * info.hs[0]);
  info.mcuy = (info.height + 8 * info.vs[0] - 1) / (8 * info.vs[0]);
  long off = 0;
  for (int c = 0; c < info.ncomp; ++c) {
    info.blocks_w[c] = info.mcux * info.hs[c];
    info.blocks_h[c] = info.mcuy * info.vs[c];
    info.coef_off[c] = off;
    off += (long)info.blocks_w[c] * info.blocks_h[c] * 64;
    // the component's true size (jdmaster.c: ceil(image * samp / max_samp)): what the fancy upsampler's edges see
    info.comp_w[c] = (info.width * info.hs[c] + info.hs[0] - 1) / info.hs[0];
    info.comp_h[c] = (info.height * info.vs[c] + info.vs[0] - 1) / info.vs[0];
  }
  info.n_coefs = off;
  fr.have_sof = true;
  return FP_OK;
}

struct Scan {
  int ns, ci[3], td[3], ta[3], Ss, Se, Ah, Al;
};

int read_sos(const unsigned char* s, unsigned len, const fp_jpeg_info& info, const Frame& fr, Scan& sc) {
  sc.ns = s[0];
  if (sc.ns < 1 || sc.ns > info.ncomp || len != 6u + 2u * sc.ns) return FP_ERR_INVALID_ARG;
  for (int i = 0; i < sc.ns; ++i) {
    int c = 0;
    while (c < info.ncomp && fr.comp_id[c] != s[1 + 2 * i]) ++c;
    if (c == info.ncomp) return FP_ERR_INVALID_ARG;
    sc.ci[i] = c;
    sc.td[i] = s[2 + 2 * i] >> 4;
    sc.ta[i] = s[2 + 2 * i] & 15;
    if (sc.td[i] > 3 || sc.ta[i] > 3) return FP_ERR_INVALID_ARG;
  }
  sc.Ss = s[1 + 2 * sc.ns];
  sc.Se = s[2 + 2 * sc.ns];
  sc.Ah = s[3 + 2 * sc.ns] >> 4;
  sc.Al = s[3 + 2 * sc.ns] & 15;
  if (!info.progressive) {
    if (sc.Ss != 0 || sc.Se != 63 || sc.Ah != 0 || sc.Al != 0) return FP_ERR_INVALID_ARG;
  } else {
    if (sc.Ss > sc.Se || sc.Se > 63 || sc.Al > 13 || (sc.Ss == 0 && sc.Se != 0) || (sc.Ss > 0 && sc.ns != 1)) return FP_ERR_INVALID_ARG;
    if (sc.Ah != 0 && sc.Ah != sc.Al + 1) return FP_ERR_INVALID_ARG;
  }
  return FP_OK;
}

// Walks the markers up to the first SOS; fills info (frame geometry, quantisation tables).  Returns the offset of that SOS
// marker's segment length or a negative fp_status.  dc / ac collect the Huffman tables seen on the way.
long parse_headers(const unsigned char* d, size_t n, fp_jpeg_info& info, Frame& fr, Huff* dc /*[4]*/, Huff* ac /*[4]*/) {
  memset(&info, 0, sizeof(info));
  memset(&fr, 0, sizeof(fr));
  if (n < 4 || d[0] != 0xff || d[1] != 0xd8) return FP_ERR_INVALID_ARG;
  size_t pos = 2;
  while (pos + 4 <= n) {
    if (d[pos] != 0xff) return FP_ERR_INVALID_ARG;
    const unsigned m = d[pos + 1];
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
    int rc = FP_OK;
    if (m == 0xdb) rc = read_dqt(s, e, fr);
    else if (m == 0xc0 || m == 0xc1 || m == 0xc2) rc = read_sof(s, len, m, info, fr);   // sequential / progressive, Huffman, 8-bit
    else if (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc) rc = FP_ERR_UNSUPPORTED;   // lossless, arithmetic, differential
    else if (m == 0xc4) rc = read_dht(s, e, dc, ac);
    else if (m == 0xdd) {
      if (len != 4) return FP_ERR_INVALID_ARG;
      info.restart_interval = (int)be16(s);
    } else if (m == 0xda) {
      if (!fr.have_sof) return FP_ERR_INVALID_ARG;
      for (int c = 0; c < info.ncomp; ++c) {
        if (!fr.have_qt[fr.comp_tq[c]]) return FP_ERR_INVALID_ARG;
        memcpy(info.quant[c], fr.qt[fr.comp_tq[c]], 128);
      }
      return (long)pos;
    }
    if (rc) return rc;
    pos += len;
  }
  return FP_ERR_INVALID_ARG;
}

// One scan's entropy-coded segment -> coefficients (jdhuff.c decode_mcu for sequential files; jdphuff.c's four MCU decoders for
// progressive ones: DC first / DC refine / AC first / AC refine).  Returns FP_OK and leaves br at the marker behind the scan.
int decode_scan(BitReader& br, const fp_jpeg_info& info, const Scan& sc, const Huff* dc, const Huff* ac, int16_t* coefs) {
  for (int i = 0; i < sc.ns; ++i) {
    if ((sc.Ss == 0 && sc.Ah == 0 && !dc[sc.td[i]].present) || (sc.Se > 0 && !ac[sc.ta[i]].present)) return FP_ERR_INVALID_ARG;
  }
  // MCU geometry: interleaved scans walk the frame's MCUs; a one-component scan walks that component's own blocks
  const int c0 = sc.ci[0];
  const bool inter = sc.ns > 1;
  const int nmx = inter ? info.mcux : (info.comp_w[c0] + 7) / 8;
  const int nmy = inter ? info.mcuy : (info.comp_h[c0] + 7) / 8;
  int pred[3] = {0, 0, 0};
  int eobrun = 0;
  int until_restart = info.restart_interval, next_rst = 0;
  const int p1 = 1 << sc.Al, m1 = -(1 << sc.Al);
  for (int my = 0; my < nmy; ++my) {
    for (int mx = 0; mx < nmx; ++mx) {
      if (info.restart_interval && until_restart == 0) {
        br.nbits = 0;                              // byte-align, expect RSTn
        br.acc = 0;
        if (!br.marker) {
          while (br.p + 1 < br.end && !(br.p[0] == 0xff && br.p[1] != 0 && br.p[1] != 0xff)) ++br.p;
          if (br.p + 1 < br.end) br.marker = br.p[1];
        }
        if (br.marker != 0xd0 + next_rst) return FP_ERR_INVALID_ARG;
        br.p += 2;
        br.marker = 0;
        next_rst = (next_rst + 1) & 7;
        pred[0] = pred[1] = pred[2] = 0;
        eobrun = 0;
        until_restart = info.restart_interval;
      }
      for (int i = 0; i < sc.ns; ++i) {
        const int c = sc.ci[i];
        const int nv = inter ? info.vs[c] : 1, nh = inter ? info.hs[c] : 1;
        for (int v = 0; v < nv; ++v) {
          for (int hh = 0; hh < nh; ++hh) {
            int16_t* blk = coefs + info.coef_off[c] + ((long)(my * nv + v) * info.blocks_w[c] + (mx * nh + hh)) * 64;
            if (!info.progressive) {
              const int rc = decode_block_seq(br, dc[sc.td[i]], ac[sc.ta[i]], pred[c], blk);
              if (rc) return rc;
            } else if (sc.Ss == 0) {
              if (sc.Ah == 0) {                    // ---- DC first scan ----
                const int s = huff_decode(br, dc[sc.td[i]]);
                if (s < 0 || s > 11) return FP_ERR_INVALID_ARG;
                if (s) pred[c] += extend(br.get(s), s);
                blk[0] = (int16_t)(pred[c] * (1 << sc.Al));
              } else if (br.get(1)) {              // ---- DC refinement: one more bit ----
                blk[0] = (int16_t)(blk[0] | p1);
              }
            } else if (sc.Ah == 0) {               // ---- AC first scan (one component) ----
              if (eobrun > 0) {
                --eobrun;
              } else {
                const Huff& ha = ac[sc.ta[i]];
                for (int k = sc.Ss; k <= sc.Se; ++k) {
                  const int rs = huff_decode(br, ha);
                  if (rs < 0) return FP_ERR_INVALID_ARG;
                  const int r = rs >> 4, s = rs & 15;
                  if (s) {
                    k += r;
                    if (k > 63) return FP_ERR_INVALID_ARG;
                    blk[kZigzag[k]] = (int16_t)(extend(br.get(s), s) * (1 << sc.Al));
                  } else if (r == 15) {
                    k += 15;                       // ZRL
                  } else {
                    eobrun = 1 << r;
                    if (r) eobrun += br.get(r);
                    --eobrun;
                    break;
                  }
                }
              }
            } else {                               // ---- AC refinement scan (one component): jdphuff.c decode_mcu_AC_refine ----
              const Huff& ha = ac[sc.ta[i]];
              int k = sc.Ss;
              if (eobrun == 0) {
                for (; k <= sc.Se; ++k) {
                  const int rs = huff_decode(br, ha);
                  if (rs < 0) return FP_ERR_INVALID_ARG;
                  int r = rs >> 4, s = rs & 15;
                  if (s) {
                    s = br.get(1) ? p1 : m1;       // (size is always 1: the new coefficient's sign)
                  } else if (r != 15) {
                    eobrun = 1 << r;
                    if (r) eobrun += br.get(r);
                    break;                         // end of band: the rest of the block gets correction bits only
                  }
                  // step over r still-zero coefficients, appending a correction bit to every nonzero one on the way
                  do {
                    int16_t* t = blk + kZigzag[k];
                    if (*t != 0) {
                      if (br.get(1) && (*t & p1) == 0) *t = (int16_t)(*t + (*t >= 0 ? p1 : m1));
                    } else if (--r < 0) {
                      break;
                    }
                    ++k;
                  } while (k <= sc.Se);
                  if (s) {
                    if (k > 63) return FP_ERR_INVALID_ARG;
                    blk[kZigzag[k]] = (int16_t)s;
                  }
                }
              }
              if (eobrun > 0) {
                for (; k <= sc.Se; ++k) {
                  int16_t* t = blk + kZigzag[k];
                  if (*t != 0 && br.get(1) && (*t & p1) == 0) *t = (int16_t)(*t + (*t >= 0 ? p1 : m1));
                }
                --eobrun;
              }
            }
          }
        }
      }
      if (info.restart_interval) --until_restart;
    }
  }
  return FP_OK;
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
  Huff* tabs = new Huff[8];
  memset(tabs, 0, 8 * sizeof(Huff));
  Frame fr;
  const long r = parse_headers(data, n, *info, fr, tabs, tabs + 4);
  delete[] tabs;
  return r < 0 ? (int)r : FP_OK;
}

int fp_jpeg_entropy_decode(const uint8_t* data, size_t n, const fp_jpeg_info* info_in, int16_t* coefs) {
  if (!data || !info_in || !coefs) return FP_ERR_INVALID_ARG;
  fp_jpeg_info info;
  Huff* dc = new Huff[8];
  Huff* ac = dc + 4;
  memset(dc, 0, 8 * sizeof(Huff));
  Frame fr;
  long pos = parse_headers(data, n, info, fr, dc, ac);
  int rc = pos < 0 ? (int)pos : FP_OK;
  if (!rc && (info.n_coefs != info_in->n_coefs || info.width != info_in->width || info.height != info_in->height)) rc = FP_ERR_INVALID_ARG;
  if (rc) {
    delete[] dc;
    return rc;
  }
  memset(coefs, 0, (size_t)info.n_coefs * sizeof(int16_t));
  // pos = the first SOS segment's length field.  One scan after the other; DHT / DRI (/ DQT, ignored: a component's table is
  // latched at its first scan) may stand between scans; EOI or the end of the data ends the image.
  int scans = 0;
  for (;;) {
    if ((size_t)pos + 2 > n) break;
    const unsigned len = be16(data + pos);
    if (len < 2 || (size_t)pos + len > n) {
      rc = FP_ERR_INVALID_ARG;
      break;
    }
    Scan sc;
    rc = read_sos(data + pos + 2, len, info, fr, sc);
    if (rc) break;
    BitReader br;
    br.init(data + pos + len, data + n);
    rc = decode_scan(br, info, sc, dc, ac, coefs);
    if (rc) break;
    ++scans;
    // the marker behind the scan
    const unsigned char* p = br.p;
    if (!br.marker) {
      while (p + 1 < data + n && !(p[0] == 0xff && p[1] != 0 && p[1] != 0xff)) ++p;
    }
    bool next_scan = false;
    while (p + 1 < data + n) {
      if (p[0] != 0xff) break;
      const unsigned m = p[1];
      if (m == 0xff) {
        ++p;
        continue;
      }
      if (m == 0xd9) break;                                        // EOI
      if (m >= 0xd0 && m <= 0xd7) {
        p += 2;
        continue;
      }
      if (p + 4 > data + n) break;
      const unsigned l2 = be16(p + 2);
      if (l2 < 2 || p + 2 + l2 > data + n) {
        rc = FP_ERR_INVALID_ARG;
        break;
      }
      if (m == 0xc4) rc = read_dht(p + 4, p + 2 + l2, dc, ac);
      else if (m == 0xdd && l2 == 4) info.restart_interval = (int)be16(p + 4);
      else if (m == 0xda) {
        pos = (long)(p + 2 - data);
        next_scan = true;
        break;
      }
      if (rc) break;
      p += 2 + l2;
    }
    if (rc || !next_scan) break;
  }
  delete[] dc;
  if (rc) return rc;
  return scans > 0 ? FP_OK : FP_ERR_INVALID_ARG;
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
