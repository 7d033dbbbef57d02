// Lab harness: the production whole-Depth_Wise kernel (csrc/dwblock.hip, compiled with FP_DWB_STAMPS) on one
// Mobile-FaceNet block shape: occupancy query, launch time at several batch sizes, s_memtime stamps per phase.
// Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFP_DWB_STAMPS -Iinclude -Iface_detection_and_recognition_amd/csrc \
//         tools/lab/dwblock_lab.hip -o tools/lab/dwblock_lab
// Run on the GPU box:  tools/lab/dwblock_lab [C HW]
#include "dwblock.hip"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

void fp_set_hip_error(hipError_t e) { fprintf(stderr, "hip error: %s\n", hipGetErrorString(e)); }

template <int C, int HW, int RB, int NIMG>
static void run(int Nmax) {
  using K = DwbCfg<C, HW, RB, NIMG>;
  const int lds = K::LDS_FLOATS * 4;
  hipFuncSetAttribute((const void*)dwblock_kernel<C, HW, RB, NIMG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dwblock_kernel<C, HW, RB, NIMG>, 512, lds);
  printf("dwblock<%d,%d,%d,%d>: LDS %d B, occupancy query = %d blocks per CU\n", C, HW, RB, NIMG, lds, occ);
  const long elems = (long)Nmax * HW * HW * C;
  const int G = 2 * C;
  const long wfl = (long)C * G + 15L * G + (long)G * C + 2 * C;
  float *x, *y, *w;
  hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&w, wfl * 4);
  std::vector<float> hx(elems), hw(wfl);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) v = 0.1f * rnd();
  hipMemcpy(x, hx.data(), elems * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice);
  unsigned long long* stamps;
  const size_t ns = 8 * 8 * 8 * 5;
  hipMalloc(&stamps, ns * 8);
  DwBlockArgs a;
  a.in = x; a.out = y; a.we = w; a.par = w + (long)C * G; a.wp = a.par + 15L * G; a.has_res = 1; a.stamps = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int ns_[] = {128 * NIMG, 256 * NIMG, 512 * NIMG, 528 * NIMG / (K::NBAND > 1 ? 1 : 1), 768 * NIMG, 1024 * NIMG};
  for (int N : ns_) {
    if (N > Nmax) continue;
    a.N = N;
    for (int i = 0; i < 2; ++i) launch_variant<C, HW, RB, NIMG>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_variant<C, HW, RB, NIMG>(a, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * HW * HW * (2.0 * C * G + 9.0 * G);
    printf("  N=%5d tiles=%5d: %8.1f us per launch, %6.1f TF/s  (%s)\n", N, (N + NIMG - 1) / NIMG * K::NBAND, ms * 100,
           fl / (ms * 1e-4) / 1e12, hipGetErrorString(hipGetLastError()));
  }
  // stamps: one launch, print blocks 0..1, waves 0..3
  a.N = Nmax < 1024 ? Nmax : 1024;
  a.stamps = stamps;
  hipMemset(stamps, 0, ns * 8);
  launch_variant<C, HW, RB, NIMG>(a, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(ns);
  hipMemcpy(h.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
  printf("  cycles per step: [first (D for waves 4-7) | E | P | last (D for waves 0-3) | barrier]   (block, wave)\n");
  for (int b = 0; b < 2; ++b)
    for (int wv = 0; wv < 8; ++wv) {
      printf("  b%d w%d:", b, wv);
      for (int c = 0; c < 8 && c < G / 32; ++c) {
        const unsigned long long* t = &h[((b * 8 + wv) * 8 + c) * 5];
        const unsigned long long nxt = c + 1 < 8 && c + 1 < G / 32 ? h[((b * 8 + wv) * 8 + c + 1) * 5] : t[4];
        printf("  [%5lld %5lld %5lld %5lld %5lld]", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]),
               (long long)(t[4] - t[3]), (long long)(nxt - t[4]));
      }
      printf("\n");
    }
  hipFree(x); hipFree(y); hipFree(w); hipFree(stamps);
}

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 128, HW = argc > 2 ? atoi(argv[2]) : 14;
  if (C == 128 && HW == 14) run<128, 14, 14, 1>(1024);
  else if (C == 128 && HW == 7) run<128, 7, 7, 3>(3072);
  else run<64, 28, 7, 1>(1024);
  return 0;
}
