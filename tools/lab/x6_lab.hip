// Lab harness: the production bf16x6 Depth_Wise kernel (csrc/dwblockx6.hip, compiled with FP_X6_STAMPS) on one block
// shape: launch time at several batch sizes, s_memtime stamps per phase of every round.
// Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFP_X6_STAMPS -Iinclude -Iface_detection_and_recognition_amd/csrc \
//         tools/lab/x6_lab.hip -o tools/lab/x6_lab
// Run on the GPU box:  tools/lab/x6_lab [C HW]
#include "dwblockx6.hip"

#include <stdio.h>

#include <vector>

void fp_set_hip_error(hipError_t e) { fprintf(stderr, "hip error: %s\n", hipGetErrorString(e)); }
#include <stdlib.h>
const fp_knobs& fp_get_knobs() {   // (capi.cpp's table is not linked into the lab binary)
  static fp_knobs k = {0, 0, getenv("FP_X6_QUARTER14") ? atoi(getenv("FP_X6_QUARTER14")) : 0,
                       getenv("FP_X6_SPEC14") ? atoi(getenv("FP_X6_SPEC14")) : 0, 0};
  return k;
}

template <int C, int HW>
static void run(int Nmax) {
  using K = X6Cfg<C, HW>;
  hipFuncSetAttribute((const void*)dwblock_x6_kernel<C, HW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dwblock_x6_kernel<C, HW>, 256, K::LDS_BYTES);
  printf("dwblock_x6<%d,%d>: LDS %d B, occupancy query = %d blocks per CU\n", C, HW, K::LDS_BYTES, occ);
  const long elems = (long)Nmax * HW * HW * C;
  const int G = 2 * C;
  const long wfl = (long)C * G * 3 / 2 + 15L * G + (long)G * C * 3 / 2 + 2 * C;
  float *x, *y, *w;
  hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&w, wfl * 4);
  std::vector<float> hx(elems);
  std::vector<unsigned> hw(wfl);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) { const float f = 0.05f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (u & 0xffff0000u) | (u >> 16); }   // two sane bf16
  hipMemcpy(x, hx.data(), elems * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice);
  unsigned long long* stamps;
  const size_t ns = 4 * 4 * 9 * 8;
  hipMalloc(&stamps, ns * 8);
  DwbX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = y;
  a.we = (const unsigned short*)w;
  a.par = w + (long)C * G * 3 / 2;
  a.wp = (const unsigned short*)(a.par + 15L * G);
  a.paff = a.par + 15L * G + (long)G * C * 3 / 2;
  a.has_res = 1; a.stamps = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int ns_[] = {64, 128, 256, 512, 528, 768, 1024};
  for (int N : ns_) {
    if (N > Nmax) continue;
    a.N = N;
    for (int i = 0; i < 2; ++i) launch_x6<C, HW>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_x6<C, HW>(a, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * HW * HW * (2.0 * C * G);
    printf("  N=%5d tiles=%5d: %8.1f us per launch, %6.1f TF/s  (%s)\n", N, N * K::NBAND, ms * 100, fl / (ms * 1e-4) / 1e12,
           hipGetErrorString(hipGetLastError()));
  }
  for (int N : {256 / K::NBAND * 1, 512 / K::NBAND * 2}) {   // one / two workgroups per CU
    a.N = N;
    a.stamps = stamps;
    hipMemset(stamps, 0, ns * 8);
    launch_x6<C, HW>(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(ns);
    hipMemcpy(h.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
    printf("  N=%d (%d tiles): cycles per round: [E | barrier | D | barrier | P | rest incl. barrier]   (block, wave)\n", N, N * K::NBAND);
    for (int b = 0; b < 2; ++b)
      for (int wv = 0; wv < 4; ++wv) {
        printf("  b%d w%d:", b, wv);
        for (int c = 0; c < K::R; ++c) {
          const unsigned long long* t = &h[((b * 4 + wv) * 9 + c) * 8];
          const unsigned long long nxt = h[((b * 4 + wv) * 9 + c + 1) * 8];
          printf("  [%5lld %5lld %5lld %5lld %5lld %5lld]", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]),
                 (long long)(t[4] - t[3]), (long long)(t[5] - t[4]), (long long)(nxt - t[5]));
        }
        const unsigned long long tot = h[((b * 4 + wv) * 9 + K::R) * 8] - h[((b * 4 + wv) * 9) * 8];
        printf("   loop total %lld\n", (long long)tot);
      }
  }
  hipFree(x); hipFree(y); hipFree(w); hipFree(stamps);
}

// the 7 x 7-tile kernel (dwblock_x6q_kernel): launch times and stamps at 1, 2, 3 workgroups per CU
template <int HW>
static void runq(int Nmax) {
  using K = X6QCfg<HW>;
  constexpr int C = 128, G = 256;
  hipFuncSetAttribute((const void*)dwblock_x6q_kernel<HW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  int occ = -1;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, dwblock_x6q_kernel<HW>, 256, K::LDS_BYTES);
  printf("dwblock_x6q<%d>: LDS %d B, occupancy query = %d blocks per CU\n", HW, K::LDS_BYTES, occ);
  const long elems = (long)Nmax * HW * HW * C;
  const long wfl = (long)C * G * 3 / 2 + 15L * G + (long)G * C * 3 / 2 + 2 * C;
  float *x, *y, *w;
  hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&w, wfl * 4);
  std::vector<float> hx(elems);
  std::vector<unsigned> hw(wfl);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) { const float f = 0.05f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (u & 0xffff0000u) | (u >> 16); }
  hipMemcpy(x, hx.data(), elems * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice);
  unsigned long long* stamps;
  const size_t ns = 4 * 4 * 9 * 8;
  hipMalloc(&stamps, ns * 8);
  DwbX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = y;
  a.we = (const unsigned short*)w;
  a.par = w + (long)C * G * 3 / 2;
  a.wp = (const unsigned short*)(a.par + 15L * G);
  a.paff = a.par + 15L * G + (long)G * C * 3 / 2;
  a.has_res = 1; a.stamps = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int N : {64, 128, 192, 384, 528, 1024}) {
    if (N > Nmax) continue;
    a.N = N;
    for (int i = 0; i < 2; ++i) launch_x6q<HW>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_x6q<HW>(a, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * N * HW * HW * (2.0 * C * G);
    printf("  N=%5d tiles=%5d: %8.1f us per launch, %6.1f TF/s  (%s)\n", N, N * K::TPI, ms * 100, fl / (ms * 1e-4) / 1e12,
           hipGetErrorString(hipGetLastError()));
  }
  for (int tiles : {256, 512, 768}) {
    a.N = tiles / K::TPI;
    a.stamps = stamps;
    hipMemset(stamps, 0, ns * 8);
    launch_x6q<HW>(a, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(ns);
    hipMemcpy(h.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
    printf("  %d tiles: cycles per round: [E | barrier | D | barrier | P | rest incl. barrier]   (block, wave)\n", tiles);
    for (int b = 0; b < 1; ++b)
      for (int wv = 0; wv < 4; ++wv) {
        printf("  b%d w%d:", b, wv);
        for (int c = 0; c < K::R; ++c) {
          const unsigned long long* t = &h[((b * 4 + wv) * 9 + c) * 8];
          const unsigned long long nxt = h[((b * 4 + wv) * 9 + c + 1) * 8];
          printf("  [%5lld %5lld %5lld %5lld %5lld %5lld]", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]),
                 (long long)(t[4] - t[3]), (long long)(t[5] - t[4]), (long long)(nxt - t[5]));
        }
        printf("   loop total %lld\n", (long long)(h[((b * 4 + wv) * 9 + K::R) * 8] - h[((b * 4 + wv) * 9) * 8]));
      }
  }
  hipFree(x); hipFree(y); hipFree(w); hipFree(stamps);
}

// the wave-specialised band kernel (dwblock_x6s_kernel): per step [E (matrix) or staging (vector) | P or D | barrier]
template <int C, int HW>
static void runs(int Nmax) {
  using K = X6SCfg<C, HW>;
  constexpr int G = 2 * C;
  hipFuncSetAttribute((const void*)dwblock_x6s_kernel<C, HW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  printf("dwblock_x6s<%d,%d>: LDS %d B\n", C, HW, K::LDS_BYTES);
  const long elems = (long)Nmax * HW * HW * C;
  const long wfl = (long)C * G * 3 / 2 + 15L * G + (long)G * C * 3 / 2 + 2 * C;
  float *x, *y, *w;
  hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&w, wfl * 4);
  std::vector<float> hx(elems);
  std::vector<unsigned> hw(wfl);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = rnd();
  for (auto& v : hw) { const float f = 0.05f * rnd(); unsigned u; memcpy(&u, &f, 4); v = (u & 0xffff0000u) | (u >> 16); }
  hipMemcpy(x, hx.data(), elems * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), wfl * 4, hipMemcpyHostToDevice);
  unsigned long long* stamps;
  const size_t ns = 2 * 8 * 9 * 4;
  hipMalloc(&stamps, ns * 8);
  DwbX6Args a;
  memset(&a, 0, sizeof(a));
  a.in = x; a.out = y;
  a.we = (const unsigned short*)w;
  a.par = w + (long)C * G * 3 / 2;
  a.wp = (const unsigned short*)(a.par + 15L * G);
  a.paff = a.par + 15L * G + (long)G * C * 3 / 2;
  a.has_res = 1; a.stamps = nullptr;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int N : {64, 128, 256, 528, 1024}) {
    if (N > Nmax) continue;
    a.N = N;
    for (int i = 0; i < 2; ++i) launch_x6s<C, HW>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch_x6s<C, HW>(a, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("  N=%5d tiles=%5d: %8.1f us per launch\n", N, N * K::NBAND, ms * 100);
  }
  a.N = 128;
  a.stamps = stamps;
  hipMemset(stamps, 0, ns * 8);
  launch_x6s<C, HW>(a, 0);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(ns);
  hipMemcpy(h.data(), stamps, ns * 8, hipMemcpyDeviceToHost);
  printf("  cycles per step: matrix waves 0-3 [E | P | barrier], vector waves 4-7 [stage | D | barrier]\n");
  for (int wv = 0; wv < 8; ++wv) {
    printf("  w%d:", wv);
    for (int c = 0; c < K::R; ++c) {
      const unsigned long long* t = &h[(wv * 9 + c) * 4];
      printf("  [%5lld %5lld %5lld]", (long long)(t[1] - t[0]), (long long)(t[2] - t[1]), (long long)(t[3] - t[2]));
    }
    printf("\n");
  }
  hipFree(x); hipFree(y); hipFree(w); hipFree(stamps);
}

int main(int argc, char** argv) {
  if (argc > 1 && atoi(argv[1]) == 1) { runs<128, 14>(1024); return 0; }
  const int C = argc > 1 ? atoi(argv[1]) : 128;
  if (C == 14 || C == 7) { if (C == 14) runq<14>(1024); else runq<7>(1024); return 0; }
  if (C == 128) run<128, 14>(1024);
  else run<64, 28>(1024);
  return 0;
}
