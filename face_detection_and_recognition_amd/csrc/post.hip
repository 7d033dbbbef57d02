// post.hip — anchor decode + NMS for BlazeFace and YOLOv5-face (gfx950).  Build with -ffp-contract=off:
// the keep / cluster-membership decisions compare fp32 IoUs against a threshold and must be bit-exact with
// the reference's CPU arithmetic, so every expression below keeps the reference's operation order and no
// mul+add may be contracted into an fma.
//
// Parallelisation: the scans are sequential and data dependent per image, so one workgroup owns one image
// (256 images per batch fill the 256 CUs) and the lanes parallelise the IoU row and the reductions with
// 64-wide wavefront shuffles.  Every loop is bounded by the candidate count (the reference's
// weighted NMS never terminates on a degenerate box; here such a box is emitted alone and removed).
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------------
// BlazeFace: _tensors_to_detections + _decode_boxes (blazeface.py:321-402)
__global__ __launch_bounds__(256) void blaze_decode_kernel(const float* __restrict__ raw_boxes,
                                                           const float* __restrict__ raw_scores,
                                                           const float* __restrict__ anchors, int A, float xs,
                                                           float ys, float ws, float hs, float clip, float thr,
                                                           float* __restrict__ cand, int* __restrict__ cand_count) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int a0 = 0; a0 < A; a0 += 256) {
    const int a = a0 + tid;
    float score = 0.f;
    bool pass = false;
    if (a < A) {
      float r = raw_scores[(long)b * A + a];
      r = fminf(fmaxf(r, -clip), clip);           // .clamp(-thresh, thresh)  blazeface.py:351
      score = 1.0f / (1.0f + expf(-r));            // .sigmoid()               blazeface.py:352
      pass = score >= thr;                         // blazeface.py:357
    }
    const unsigned long long m = __ballot(pass);
    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (pass) {
      const float* rb = raw_boxes + ((long)b * A + a) * 16;
      const float ax = anchors[a * 4 + 0], ay = anchors[a * 4 + 1], aw = anchors[a * 4 + 2], ah = anchors[a * 4 + 3];
      float* o = cand + ((long)b * A + off + prefix) * 17;
      const float xc = rb[0] / xs * aw + ax;       // blazeface.py:380-383
      const float yc = rb[1] / ys * ah + ay;
      const float w = rb[2] / ws * aw;
      const float h = rb[3] / hs * ah;
      o[0] = yc - h / 2.f;
      o[1] = xc - w / 2.f;
      o[2] = yc + h / 2.f;
      o[3] = xc + w / 2.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) {                // blazeface.py:393-400
        o[4 + 2 * k] = rb[4 + 2 * k] / xs * aw + ax;
        o[5 + 2 * k] = rb[5 + 2 * k] / ys * ah + ay;
      }
      o[16] = score;
    }
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (tid == 0) cand_count[b] = base_s;
}

// overlap_similarity / jaccard / intersect (blazeface.py:463-526), box = (ymin, xmin, ymax, xmax).
__device__ __forceinline__ float blaze_iou(const float4 a, const float4 b) {
  const float max_x = fminf(a.z, b.z), max_y = fminf(a.w, b.w);
  const float min_x = fmaxf(a.x, b.x), min_y = fmaxf(a.y, b.y);
  const float ix = fmaxf(max_x - min_x, 0.f), iy = fmaxf(max_y - min_y, 0.f);
  const float inter = ix * iy;
  const float area_a = (a.z - a.x) * (a.w - a.y);
  const float area_b = (b.z - b.x) * (b.w - b.y);
  const float uni = area_a + area_b - inter;
  return inter / uni;
}

constexpr int BLAZE_MAX = 896;  // num_anchors (blazeface.py:96): the largest possible candidate count

// _weighted_non_max_suppression (blazeface.py:404-458), one workgroup per image.
__global__ __launch_bounds__(256) void blaze_wnms_kernel(const float* __restrict__ dets, const int* __restrict__ counts,
                                                         int max_in, float thr, float* __restrict__ out,
                                                         int* __restrict__ out_count, int* __restrict__ member_of) {
  __shared__ float4 sbox[BLAZE_MAX];
  __shared__ float sscore[BLAZE_MAX];
  __shared__ short order[BLAZE_MAX];
  __shared__ unsigned char alive[BLAZE_MAX];
  __shared__ float red[4][18];
  __shared__ int cursor_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int n = counts[b];
  n = min(max(n, 0), min(max_in, BLAZE_MAX));
  const float* d = dets + (long)b * max_in * 17;
  float* o = out + (long)b * max_in * 17;
  int* mo = member_of ? member_of + (long)b * max_in : nullptr;
  for (int i = tid; i < n; i += 256) {
    sbox[i] = make_float4(d[i * 17 + 0], d[i * 17 + 1], d[i * 17 + 2], d[i * 17 + 3]);
    sscore[i] = d[i * 17 + 16];
  }
  __syncthreads();
  // argsort(score, descending), ties by input index (blazeface.py:426; torch's argsort is not stable,
  // the stable rule is this build's documented choice)
  for (int i = tid; i < n; i += 256) {
    const float s = sscore[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float t = sscore[j];
      rank += (t > s || (t == s && j < i)) ? 1 : 0;
    }
    order[rank] = (short)i;
    alive[i] = 1;  // indexed by sorted position below; all n positions start alive
  }
  if (tid == 0) cursor_s = 0;
  __syncthreads();
  int nout = 0;
  for (int iter = 0; iter < n; ++iter) {  // bounded: every iteration removes at least remaining[0]
    if (tid == 0) {
      int c = cursor_s;
      while (c < n && !alive[c]) ++c;
      cursor_s = c;
    }
    __syncthreads();
    const int cursor = cursor_s;
    if (cursor >= n) break;
    const int first = order[cursor];
    const float4 fb = sbox[first];
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    float ssum = 0.f, cnt = 0.f;
    for (int pos = cursor + tid; pos < n; pos += 256) {
      if (!alive[pos]) continue;
      const int j = order[pos];
      const float iou = blaze_iou(fb, sbox[j]);
      bool in = iou > thr;  // blazeface.py:441
      if (pos == cursor && !in) in = true;  // degenerate remaining[0]: emitted alone (see header)
      if (in) {
        alive[pos] = 0;
        if (mo) mo[j] = nout;
        const float s = sscore[j];
        const float* dj = d + j * 17;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] += dj[c] * s;  // (coordinates * scores).sum(0)  blazeface.py:451
        ssum += s;
        cnt += 1.f;
      }
    }
    // block reduction of 18 partials: wave shuffles, then 4 wave results through LDS (fixed order)
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = wave_sum(acc[c]);
    ssum = wave_sum(ssum);
    cnt = wave_sum(cnt);
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) red[wave][c] = acc[c];
      red[wave][16] = ssum;
      red[wave][17] = cnt;
    }
    __syncthreads();
    if (tid < 17) {
      const float total = red[0][16] + red[1][16] + red[2][16] + red[3][16];
      const float count = red[0][17] + red[1][17] + red[2][17] + red[3][17];
      float v;
      if (count > 1.5f) {  // len(overlapping) > 1  blazeface.py:448
        if (tid < 16)
          v = (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]) / total;  // blazeface.py:452
        else
          v = total / count;                                                    // blazeface.py:454
      } else {
        v = d[first * 17 + tid];
      }
      o[nout * 17 + tid] = v;
    }
    ++nout;
    __syncthreads();
  }
  if (tid == 0) out_count[b] = nout;
}

// ------------------------------------------------------------------------------------------------
// YOLOv5-face Detect decode (y5/models/yolo.py:62-108; ONNX twin onnx_utils.py:30-73)
__global__ __launch_bounds__(256) void yolo_decode_kernel(const float* __restrict__ head, int B, int ny, int nx, int na,
                                                          float stride, float aw0, float ah0, float aw1, float ah1,
                                                          float aw2, float ah2, float* __restrict__ out,
                                                          long out_image_stride, long out_row_offset) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)na * ny * nx;
  if (idx >= per * B) return;
  const int b = (int)(idx / per);
  int rem = (int)(idx - (long)b * per);
  const int a = rem / (ny * nx);
  rem -= a * ny * nx;
  const int y = rem / nx, x = rem - y * nx;
  const float aw = a == 0 ? aw0 : (a == 1 ? aw1 : aw2);
  const float ah = a == 0 ? ah0 : (a == 1 ? ah1 : ah2);
  const float* v = head + (((long)b * ny + y) * nx + x) * (na * 16) + a * 16;
  float* o = out + (long)b * out_image_stride + (out_row_offset + (long)a * ny * nx + (long)y * nx + x) * 16;
  float r[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) r[c] = v[c];
  const float gx = (float)x, gy = (float)y;
  float s0 = 1.0f / (1.0f + expf(-r[0])), s1 = 1.0f / (1.0f + expf(-r[1]));
  float s2 = 1.0f / (1.0f + expf(-r[2])), s3 = 1.0f / (1.0f + expf(-r[3]));
  o[0] = (s0 * 2.f - 0.5f + gx) * stride;  // yolo.py:83-84
  o[1] = (s1 * 2.f - 0.5f + gy) * stride;
  const float w2 = s2 * 2.f, h2 = s3 * 2.f;
  o[2] = w2 * w2 * aw;                      // yolo.py:85-86
  o[3] = h2 * h2 * ah;
  o[4] = 1.0f / (1.0f + expf(-r[4]));
#pragma unroll
  for (int k = 0; k < 5; ++k) {             // yolo.py:89-103
    o[5 + 2 * k] = r[5 + 2 * k] * aw + gx * stride;
    o[6 + 2 * k] = r[6 + 2 * k] * ah + gy * stride;
  }
  o[15] = 1.0f / (1.0f + expf(-r[15]));
}

// Greedy NMS, one workgroup per image.  MODE 0: non_max_suppression_face + torchvision.ops.nms
// (y5/utils/general.py:370-453).  MODE 1: w_non_max_suppression + w_bbox_iou (onnx_utils.py:76-163).
struct NmsScratch {  // per image, in caller-provided global scratch (max_cand entries each)
  float4* box;
  float* score;
  int* row;
  int* order;
};

template <int MODE>
__global__ __launch_bounds__(256) void yolo_nms_kernel(const float* __restrict__ pred, int n_rows, float conf_thres,
                                                       float iou_thres, int max_cand, int max_out,
                                                       float* __restrict__ out, int* __restrict__ out_count,
                                                       int* __restrict__ keep_idx, int* __restrict__ overflow,
                                                       char* __restrict__ scratch) {
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  __shared__ int cursor_s;
  constexpr int OUTC = MODE == 0 ? 16 : 7;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t per_img = (size_t)max_cand * (sizeof(float4) + sizeof(float) + 2 * sizeof(int) + sizeof(int));
  char* sc = scratch + (size_t)b * per_img;
  float4* sbox = (float4*)sc;
  float* sscore = (float*)(sbox + max_cand);
  int* srow = (int*)(sscore + max_cand);
  int* order = srow + max_cand;
  int* supp = order + max_cand;
  const float* P = pred + (long)b * n_rows * 16;
  if (tid == 0) base_s = 0;
  __syncthreads();
  bool over = false;
  // phase A: candidate compaction in row order
  for (int r0 = 0; r0 < n_rows; r0 += 256) {
    const int r = r0 + tid;
    bool pass = false;
    float score = 0.f;
    if (r < n_rows) {
      const float obj = P[(long)r * 16 + 4];
      if (MODE == 0) {
        if (obj > conf_thres) {                       // general.py:377,393
          score = P[(long)r * 16 + 15] * obj;         // general.py:410  x[:, 15:] *= x[:, 4:5]
          pass = score > conf_thres;                  // general.py:423
        }
      } else {
        pass = obj >= conf_thres;                     // onnx_utils.py:121
        score = obj;
      }
    }
    const unsigned long long m = __ballot(pass);
    const int prefix = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    const int slot = off + prefix;
    if (pass) {
      if (slot < max_cand) {
        const float cx = P[(long)r * 16 + 0], cy = P[(long)r * 16 + 1], w = P[(long)r * 16 + 2], h = P[(long)r * 16 + 3];
        // xywh2xyxy (general.py:204-211) / box_corner (onnx_utils.py:110-114)
        sbox[slot] = make_float4(cx - w / 2.f, cy - h / 2.f, cx + w / 2.f, cy + h / 2.f);
        sscore[slot] = score;
        srow[slot] = r;
      } else {
        over = true;
      }
    }
    __syncthreads();
    if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (__syncthreads_or(over ? 1 : 0)) {
    if (tid == 0) overflow[b] = 1;
  } else if (tid == 0) {
    overflow[b] = 0;
  }
  const int n = min(base_s, max_cand);
  // phase B: stable descending rank sort (torchvision nms sorts scores descending; ties by index)
  for (int i = tid; i < n; i += 256) {
    const float s = sscore[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const float t = sscore[j];
      rank += (t > s || (t == s && j < i)) ? 1 : 0;
    }
    order[rank] = i;
    supp[i] = 0;  // indexed by sorted position
  }
  if (tid == 0) cursor_s = 0;
  __threadfence_block();
  __syncthreads();
  // phase C: greedy scan
  int nout = 0;
  float* O = out + (long)b * max_out * OUTC;
  int* KI = keep_idx + (long)b * max_out;
  for (int iter = 0; iter < n; ++iter) {
    if (tid == 0) {
      int c = cursor_s;
      while (c < n && supp[c]) ++c;
      cursor_s = c;
    }
    __syncthreads();
    const int cursor = cursor_s;
    if (cursor >= n) break;
    if (nout >= max_out) {  // more survivors than the output holds: report it, the host raises
      if (tid == 0) overflow[b] = 1;
      break;
    }
    const int i = order[cursor];
    const float4 bi = sbox[i];
    if (tid < OUTC) {
      const int r = srow[i];
      float v;
      if (MODE == 0) {
        // rows = (box, conf, landmarks, cls)  general.py:422
        if (tid == 0) v = bi.x;
        else if (tid == 1) v = bi.y;
        else if (tid == 2) v = bi.z;
        else if (tid == 3) v = bi.w;
        else if (tid == 4) v = sscore[i];
        else if (tid < 15) v = P[(long)r * 16 + tid];
        else v = 0.f;
      } else {
        // (x1, y1, x2, y2, obj_conf, class_conf, class_pred)  onnx_utils.py:128-133 (class_conf = column 5)
        if (tid == 0) v = bi.x;
        else if (tid == 1) v = bi.y;
        else if (tid == 2) v = bi.z;
        else if (tid == 3) v = bi.w;
        else if (tid == 4) v = sscore[i];
        else if (tid == 5) v = P[(long)r * 16 + 5];
        else v = 0.f;
      }
      O[nout * OUTC + tid] = v;
      if (tid == 0) KI[nout] = r;
    }
    const float iarea = MODE == 0 ? (bi.z - bi.x) * (bi.w - bi.y) : (bi.z - bi.x + 1.f) * (bi.w - bi.y + 1.f);
    for (int pos = cursor + 1 + tid; pos < n; pos += 256) {
      if (supp[pos]) continue;
      const float4 bj = sbox[order[pos]];
      const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
      const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
      bool kill;
      if (MODE == 0) {
        const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
        const float inter = w * h;
        const float jarea = (bj.z - bj.x) * (bj.w - bj.y);
        const float ovr = inter / (iarea + jarea - inter);
        kill = ovr > iou_thres;
      } else {
        const float inter = fmaxf(xx2 - xx1 + 1.f, 0.f) * fmaxf(yy2 - yy1 + 1.f, 0.f);
        const float jarea = (bj.z - bj.x + 1.f) * (bj.w - bj.y + 1.f);
        const float iou = inter / (iarea + jarea - inter + 1e-16f);
        kill = !(iou < iou_thres);  // survivors are ious < nms_thres  onnx_utils.py:153
      }
      if (kill) supp[pos] = 1;
    }
    if (tid == 0) supp[cursor] = 1;
    ++nout;
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) out_count[b] = nout;
}


// ------------------------------------------------------------------------------------------------
// Detections -> per-face crop rectangles, on device (keeps detect -> embed free of a host round trip).
// fmt 0: BlazeFaceModel rows (ymin,xmin,ymax,xmax,...,score@16), normalised to the model input:
//        column reorder (blazeface/model.py:70) + get_dets_bboxes_confs_lmarks_areas (utils/inference.py:11-58).
// fmt 1: YOLOv5-face rows (x1,y1,x2,y2,conf@4,...) in model-input pixels: get_bboxes_confs_areas
//        (yolov5_face/onnx/onnx_utils.py:313-340).
// Both: conf > det_thres, area filter (fmt 0: 100*(area/total) > thr, info[6] = fraction; fmt 1: (100*area)/total > thr,
// info[6] = percent -- each in its reference's operation order), scale_coords (utils/image.py:79-99: subtract pad,
// divide by gain, clip to the frame), round half-to-even, then the crop of
// face_extraction/extract_faces_from_dataset.py:289-303: int(), offsets (tx,ty,bx,by), clamp to the frame.
// All fp32, in numpy's operation order for float32 inputs.  Faces are emitted in (frame, detection) order.
struct CropArgs {
  const float* dets;
  const int* counts;
  int B, max_dets, row, fmt, in_w, in_h, orig_w, orig_h;
  float det_thres, area_thres, gain, pad_x, pad_y;
  int tx, ty, bx, by, dst_w, dst_h, max_faces;
  fp_resize_item* items;
  float* info;
  int* n_faces;
};

__device__ __forceinline__ bool crop_one(const CropArgs& p, const float* d, float& x1, float& y1, float& x2,
                                         float& y2, float& conf, float& perc) {
  if (p.fmt == 0) {
    conf = d[16];
    if (!(conf > p.det_thres)) return false;
    x1 = d[1] * (float)p.in_w; y1 = d[0] * (float)p.in_h; x2 = d[3] * (float)p.in_w; y2 = d[2] * (float)p.in_h;
  } else {
    conf = d[4];
    if (!(conf > p.det_thres)) return false;
    x1 = d[0]; y1 = d[1]; x2 = d[2]; y2 = d[3];
  }
  const float area = (x2 - x1) * (y2 - y1);
  if (p.fmt == 0) {  // inference.py:40-42: perc = area / total (the FRACTION is reported), filter on 100 * perc
    perc = area / (float)(p.in_w * p.in_h);
    if (!(100.f * perc > p.area_thres)) return false;
  } else {           // onnx_utils.py:329-332: perc = 100 * area / total (the PERCENT is reported and compared)
    perc = (100.f * area) / (float)(p.in_w * p.in_h);
    if (!(perc > p.area_thres)) return false;
  }
  x1 = (x1 - p.pad_x) / p.gain; x2 = (x2 - p.pad_x) / p.gain;
  y1 = (y1 - p.pad_y) / p.gain; y2 = (y2 - p.pad_y) / p.gain;
  x1 = fminf(fmaxf(x1, 0.f), (float)p.orig_w); x2 = fminf(fmaxf(x2, 0.f), (float)p.orig_w);
  y1 = fminf(fmaxf(y1, 0.f), (float)p.orig_h); y2 = fminf(fmaxf(y2, 0.f), (float)p.orig_h);
  x1 = rintf(x1); y1 = rintf(y1); x2 = rintf(x2); y2 = rintf(y2);
  return true;
}

__global__ __launch_bounds__(256) void dets_to_crops_kernel(CropArgs p) {
  __shared__ int scan[256];
  __shared__ int base_s;
  const int tid = threadIdx.x;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int f0 = 0; f0 < p.B; f0 += 256) {
    const int f = f0 + tid;
    int n = 0, cnt = 0;
    const float* D = nullptr;
    if (f < p.B) {
      n = min(max(p.counts[f], 0), p.max_dets);
      D = p.dets + (long)f * p.max_dets * p.row;
      for (int i = 0; i < n; ++i) {
        float x1, y1, x2, y2, c, pc;
        if (crop_one(p, D + (long)i * p.row, x1, y1, x2, y2, c, pc)) ++cnt;
      }
    }
    scan[tid] = cnt;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive Hillis-Steele scan over the 256 frames of this chunk
      int v = tid >= off ? scan[tid - off] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    int slot = base_s + scan[tid] - cnt;
    for (int i = 0; i < n; ++i) {
      float x1, y1, x2, y2, c, pc;
      if (!crop_one(p, D + (long)i * p.row, x1, y1, x2, y2, c, pc)) continue;
      if (slot < p.max_faces) {
        int x = (int)x1 + p.tx, y = (int)y1 + p.ty, xw = (int)x2 + p.bx, yh = (int)y2 + p.by;
        x = max(x, 0); y = max(y, 0); xw = min(xw, p.orig_w); yh = min(yh, p.orig_h);
        fp_resize_item it;
        it.src_image = f;
        it.sx = x; it.sy = y; it.sw = xw - x; it.sh = yh - y;
        it.dx = 0; it.dy = 0; it.dw = p.dst_w; it.dh = p.dst_h;
        if (it.sw <= 0 || it.sh <= 0) { it.dw = 0; it.dh = 0; }  // empty crop: canvas becomes pad colour
        p.items[slot] = it;
        float* o = p.info + (long)slot * 7;
        o[0] = (float)f; o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2; o[5] = c; o[6] = pc;
      }
      ++slot;
    }
    __syncthreads();
    if (tid == 255) base_s += scan[255];
    __syncthreads();
  }
  if (tid == 0) p.n_faces[0] = base_s;  // may exceed max_faces: the host checks and raises
}

}  // namespace

extern "C" {

int fp_blaze_decode(const float* raw_boxes, const float* raw_scores, const float* anchors, int B, int A, float x_scale,
                    float y_scale, float w_scale, float h_scale, float score_clip, float min_score_thresh, float* cand,
                    int32_t* cand_count, void* stream) {
  if (!raw_boxes || !raw_scores || !anchors || !cand || !cand_count || B < 0 || A <= 0) return FP_ERR_INVALID_ARG;
  if (B == 0) return FP_OK;
  hipLaunchKernelGGL(blaze_decode_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, raw_boxes, raw_scores, anchors, A,
                     x_scale, y_scale, w_scale, h_scale, score_clip, min_score_thresh, cand, cand_count);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_blaze_weighted_nms(const float* dets, const int32_t* counts, int B, int max_in, float iou_thresh, float* out,
                          int32_t* out_count, int32_t* member_of, void* stream) {
  if (!dets || !counts || !out || !out_count || B < 0 || max_in <= 0) return FP_ERR_INVALID_ARG;
  if (max_in > BLAZE_MAX) return FP_ERR_UNSUPPORTED;
  if (B == 0) return FP_OK;
  hipLaunchKernelGGL(blaze_wnms_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dets, counts, max_in, iou_thresh,
                     out, out_count, member_of);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_yolo_decode(const float* head, int B, int ny, int nx, int na, float stride, const float* anchors_px, float* out,
                   int64_t out_image_stride, int64_t out_row_offset, void* stream) {
  if (!head || !anchors_px || !out || B < 0 || ny <= 0 || nx <= 0 || na != 3) return FP_ERR_INVALID_ARG;
  if (B == 0) return FP_OK;
  const long total = (long)B * na * ny * nx;
  hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)fp_ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     head, B, ny, nx, na, stride, anchors_px[0], anchors_px[1], anchors_px[2], anchors_px[3],
                     anchors_px[4], anchors_px[5], out, (long)out_image_stride, (long)out_row_offset);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

size_t fp_yolo_nms_scratch_bytes(int B, int max_cand) {
  if (B <= 0 || max_cand <= 0) return 0;
  return (size_t)B * (size_t)max_cand * (sizeof(float4) + sizeof(float) + 3 * sizeof(int));
}

static int yolo_nms_common(int mode, const float* pred, int B, int n_rows, float conf_thres, float iou_thres,
                           int max_cand, int max_out, float* out, int32_t* out_count, int32_t* keep_idx,
                           int32_t* overflow, void* scratch, size_t scratch_bytes, void* stream) {
  if (!pred || !out || !out_count || !keep_idx || !overflow || !scratch) return FP_ERR_INVALID_ARG;
  if (B < 0 || n_rows <= 0 || max_cand <= 0 || max_out <= 0 || max_cand % 4) return FP_ERR_INVALID_ARG;
  if (scratch_bytes < fp_yolo_nms_scratch_bytes(B, max_cand)) return FP_ERR_BOUNDS;
  if (((uintptr_t)scratch) % 16) return FP_ERR_ALIGNMENT;
  if (B == 0) return FP_OK;
  if (mode == 0)
    hipLaunchKernelGGL((yolo_nms_kernel<0>), dim3(B), dim3(256), 0, (hipStream_t)stream, pred, n_rows, conf_thres,
                       iou_thres, max_cand, max_out, out, out_count, keep_idx, overflow, (char*)scratch);
  else
    hipLaunchKernelGGL((yolo_nms_kernel<1>), dim3(B), dim3(256), 0, (hipStream_t)stream, pred, n_rows, conf_thres,
                       iou_thres, max_cand, max_out, out, out_count, keep_idx, overflow, (char*)scratch);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

int fp_yolo_nms(const float* pred, int B, int n_rows, float conf_thres, float iou_thres, int max_cand, int max_out,
                float* out, int32_t* out_count, int32_t* keep_idx, int32_t* overflow, void* scratch,
                size_t scratch_bytes, void* stream) {
  return yolo_nms_common(0, pred, B, n_rows, conf_thres, iou_thres, max_cand, max_out, out, out_count, keep_idx,
                         overflow, scratch, scratch_bytes, stream);
}

int fp_yolo_w_nms(const float* pred, int B, int n_rows, float conf_thres, float nms_thres, int max_cand, int max_out,
                  float* out, int32_t* out_count, int32_t* keep_idx, int32_t* overflow, void* scratch,
                  size_t scratch_bytes, void* stream) {
  return yolo_nms_common(1, pred, B, n_rows, conf_thres, nms_thres, max_cand, max_out, out, out_count, keep_idx,
                         overflow, scratch, scratch_bytes, stream);
}

int fp_dets_to_crops(const float* dets, const int32_t* counts, int B, int max_dets, int row_floats, int fmt, int in_w,
                     int in_h, int orig_w, int orig_h, float det_thres, float area_thres, float gain, float pad_x,
                     float pad_y, int off_tx, int off_ty, int off_bx, int off_by, int dst_w, int dst_h, int max_faces,
                     fp_resize_item* items, float* face_info, int32_t* n_faces, void* stream) {
  if (!dets || !counts || !items || !face_info || !n_faces) return FP_ERR_INVALID_ARG;
  if (B < 0 || max_dets <= 0 || max_faces <= 0 || in_w <= 0 || in_h <= 0 || orig_w <= 0 || orig_h <= 0 ||
      dst_w <= 0 || dst_h <= 0 || !(gain > 0.f))
    return FP_ERR_INVALID_ARG;
  if ((fmt == 0 && row_floats < 17) || (fmt == 1 && row_floats < 5) || fmt < 0 || fmt > 1) return FP_ERR_INVALID_ARG;
  CropArgs a{dets, counts, B, max_dets, row_floats, fmt, in_w, in_h, orig_w, orig_h, det_thres, area_thres, gain,
             pad_x, pad_y, off_tx, off_ty, off_bx, off_by, dst_w, dst_h, max_faces, items, face_info, n_faces};
  hipLaunchKernelGGL(dets_to_crops_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}

}  // extern "C"
