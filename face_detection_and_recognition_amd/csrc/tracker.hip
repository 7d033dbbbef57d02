// tracker.hip — face-tracker matching on the device (gfx950).
//
// Reference: Net.check_if_face_exists / Net.add_face
// (fde/face_extraction/extract_and_label_faces_from_dataset.py:101-121): for every new face, in detection order, walk
// the known faces IN INSERTION ORDER and take the first one with
//     (dist < normal_thres and iou > 0.1) or dist < harsh_thres
// where dist = ||feat - new||_2 (MOBILE_FACENET) or 1 - <feat,new>/(|feat||new|) (other nets) and iou is
// calculate_bbox_iou (fde/modules/utils/image.py:124-143) of the stored and the new box; on a match the stored feature
// and box are REPLACED by the new ones; otherwise the face is appended with id = number of faces so far + 1.
// The walk is sequential per face (a match rewrites the gallery the next face is compared with), so one workgroup
// processes a frame's faces one after the other; for each face the distances to all known faces are computed in
// parallel (one wave per known face, 64-wide shuffle reduction) and the first match is an LDS atomicMin over indices.
// The gallery of a video is small (tens of faces): this is latency-bound glue, not a bandwidth kernel.
// Decisions: IoU in integer arithmetic + one fp64 division exactly as Python does it (bit-exact); distances in fp32
// (numpy's reduction order is unspecified, so a distance within ~1e-6 of a threshold may decide differently).
#include "common.h"

namespace {

struct TrackArgs {
  float* feats;        // [cap][D]   gallery features (updated in place)
  int* bboxes;         // [cap][4]   gallery boxes x, y, xw, yh (updated in place)
  int* count;          // [1]        number of known faces (updated in place)
  const float* nf;     // [F][D]     new features, detection order
  const int* nb;       // [F][4]
  int* ids;            // [F]        face id assigned to each new face (1-based), 0 if the gallery is full
  unsigned char* exists;  // [F]     1 = matched a known face, 0 = appended
  int F, D, cap, mode;    // mode 0: L2 distance, 1: cosine distance
  float normal_thres, harsh_thres;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void tracker_step_kernel(TrackArgs p) {
  __shared__ int first_match;
  __shared__ int n_known;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) n_known = *p.count;
  __syncthreads();
  for (int f = 0; f < p.F; ++f) {
    if (tid == 0) first_match = 0x7fffffff;
    __syncthreads();
    const int n = n_known;
    const float* x = p.nf + (long)f * p.D;
    const int bx0 = p.nb[f * 4 + 0], by0 = p.nb[f * 4 + 1], bx1 = p.nb[f * 4 + 2], by1 = p.nb[f * 4 + 3];
    for (int i = wave; i < n; i += 4) {
      const float* g = p.feats + (long)i * p.D;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
      for (int d = lane; d < p.D; d += 64) {
        const float a = g[d], b = x[d];
        if (p.mode == 0) {
          const float df = a - b;
          s0 += df * df;
        } else {
          s0 += a * b;
          s1 += a * a;
          s2 += b * b;
        }
      }
      s0 = wave_sum(s0);
      float dist;
      if (p.mode == 0) {
        dist = sqrtf(s0);
      } else {
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        dist = 1.0f - s0 / (sqrtf(s1) * sqrtf(s2));
      }
      // calculate_bbox_iou: Python ints, one true division
      const int gx0 = p.bboxes[i * 4 + 0], gy0 = p.bboxes[i * 4 + 1], gx1 = p.bboxes[i * 4 + 2], gy1 = p.bboxes[i * 4 + 3];
      const long xd = (long)min(gx1, bx1) - max(gx0, bx0), yd = (long)min(gy1, by1) - max(gy0, by0);
      double iou = 0.0;
      if (xd >= 0 && yd >= 0) {
        const long inter = xd * yd;
        const long uni = (long)(gx1 - gx0) * (gy1 - gy0) + (long)(bx1 - bx0) * (by1 - by0) - inter;
        iou = (double)inter / (double)uni;   // 0/0 -> nan -> not > 0.1 (Python raises ZeroDivisionError there)
      }
      const bool match = (dist < p.normal_thres && iou > 0.1) || dist < p.harsh_thres;
      if (lane == 0 && match) atomicMin(&first_match, i);
    }
    __syncthreads();
    const int m = first_match;
    int slot = -1;
    if (m != 0x7fffffff) slot = m;            // replace the stored feature / box of the matched face
    else if (n < p.cap) slot = n;             // append
    if (slot >= 0) {
      for (int d = tid; d < p.D; d += 256) p.feats[(long)slot * p.D + d] = x[d];
      if (tid < 4) p.bboxes[slot * 4 + tid] = p.nb[f * 4 + tid];
    }
    if (tid == 0) {
      p.ids[f] = slot >= 0 ? slot + 1 : 0;
      p.exists[f] = m != 0x7fffffff ? 1 : 0;
      if (m == 0x7fffffff && n < p.cap) n_known = n + 1;
    }
    __threadfence_block();
    __syncthreads();   // the gallery rewrite is visible to the next face's comparisons
  }
  if (tid == 0) *p.count = n_known;
}

}  // namespace

extern "C" int fp_tracker_step(float* feats, int* bboxes, int* count, int cap, int D, const float* new_feats,
                               const int* new_bboxes, int F, int mode, float normal_thres, float harsh_thres,
                               int* ids, unsigned char* exists, void* stream) {
  if (!feats || !bboxes || !count || cap <= 0 || D <= 0 || F < 0 || (mode != 0 && mode != 1)) return FP_ERR_INVALID_ARG;
  if (F == 0) return FP_OK;
  if (!new_feats || !new_bboxes || !ids || !exists) return FP_ERR_INVALID_ARG;
  TrackArgs a;
  a.feats = feats; a.bboxes = bboxes; a.count = count; a.nf = new_feats; a.nb = new_bboxes; a.ids = ids; a.exists = exists;
  a.F = F; a.D = D; a.cap = cap; a.mode = mode; a.normal_thres = normal_thres; a.harsh_thres = harsh_thres;
  hipLaunchKernelGGL(tracker_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  FP_CHECK_LAUNCH();
  return FP_OK;
}
