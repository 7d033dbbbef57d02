// Lab harness: the production fused depthwise+projection kernel (csrc/dwpw.hip, compiled with FP_DWPW_STAMPS) on
// one Mobile-FaceNet layer shape, with s_memtime stamps per phase of the persistent kernel's first units.
// Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFP_DWPW_STAMPS -Iinclude -Iface_detection_and_recognition_amd/csrc \
//         tools/lab/dwpw_lab.hip -o tools/lab/dwpw_lab
// Run on the GPU box:  tools/lab/dwpw_lab [HW G Cout stride]
#include "dwpw.hip"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

void fp_set_hip_error(hipError_t e) { fprintf(stderr, "hip error: %s\n", hipGetErrorString(e)); }

int main(int argc, char** argv) {
  const int HW = argc > 1 ? atoi(argv[1]) : 28, G = argc > 2 ? atoi(argv[2]) : 128, Cout = argc > 3 ? atoi(argv[3]) : 64;
  const int stride = argc > 4 ? atoi(argv[4]) : 1, N = 1088;
  const int OHW = (HW + 2 - 3) / stride + 1;
  const long in_elems = (long)N * HW * HW * G, out_elems = (long)N * OHW * OHW * Cout;
  const int Npad = (Cout + 31) / 32 * 32;
  const long dw_elems = 12L * G, pw_elems = (long)G * Npad + 2 * ((Cout + 3) & ~3);
  float *arena, *weights;
  hipMalloc(&arena, (in_elems + 2 * out_elems) * 4);
  hipMalloc(&weights, (dw_elems + pw_elems) * 4);
  hipMemset(arena, 0x3c, (in_elems + 2 * out_elems) * 4);   // 0x3c3c3c3c ~ 0.0115
  hipMemset(weights, 0x3c, (dw_elems + pw_elems) * 4);
  fp_op op;
  memset(&op, 0, sizeof(op));
  op.kind = 8;
  op.N = N; op.H = HW; op.W = HW; op.OH = OHW; op.OW = OHW; op.Cin = G; op.Cout = Cout; op.KH = 3; op.KW = 3;
  op.stride = stride; op.pad_t = 1; op.pad_l = 1; op.act = FP_ACT_PRELU;
  op.res_mode = stride == 1 ? FP_RES_ADD_AFTER_ACT : FP_RES_NONE;
  op.in_off = 0; op.in_ld = G; op.in_ns = (long)HW * HW * G;
  op.out_off = in_elems; op.out_ld = Cout; op.out_ns = (long)OHW * OHW * Cout; op.out_cmul = 1;
  op.res_off = in_elems + out_elems; op.res_ld = Cout; op.res_ns = op.out_ns; op.res_C = Cout;
  op.w_off = 0; op.slope_off = dw_elems; op.bias_off = -1; op.scale_off = -1;
  unsigned long long* stamps;
  const size_t ns = 4 * 4 * FP_DWPW_NUNIT * FP_DWPW_NSTAMP;
  hipMalloc(&stamps, ns * 8);
  hipMemset(stamps, 0, ns * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  g_dwpw_stamps = nullptr;
  int rc = 0;
  for (int i = 0; i < 3; ++i) rc |= fp_launch_dwpw(op, weights, arena, 0);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) rc |= fp_launch_dwpw(op, weights, arena, 0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("dwpw %dx%d G=%d Cout=%d s%d: %.1f us per launch rc=%d err=%s\n", HW, HW, G, Cout, stride, ms * 100, rc,
         hipGetErrorString(hipGetLastError()));
  g_dwpw_stamps = stamps;
  fp_launch_dwpw(op, weights, arena, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(ns);
  hipMemcpy(h.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
  printf("per unit: [dw+lds-write | barrier | issue next | mfma | barrier] then gap to the next unit (epilogue at tile end)\n");
  for (int b = 0; b < 2; ++b)
    for (int wv = 0; wv < 4; ++wv) {
      printf("blk %d wave %d:", b, wv);
      for (int u = 1; u < 13; ++u) {
        const unsigned long long* s = &h[((b * 4 + wv) * FP_DWPW_NUNIT + u) * FP_DWPW_NSTAMP];
        const unsigned long long* n = s + FP_DWPW_NSTAMP;
        printf("  [%llu|%llu|%llu|%llu|%llu] +%llu", s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], n[0] - s[5]);
      }
      printf("\n");
    }
  return 0;
}
