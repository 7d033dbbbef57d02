// Mutation fuzzer for the host half of the JPEG decoder (fp_jpeg_parse + fp_jpeg_entropy_decode), built with AddressSanitizer.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "facepath.h"
static uint64_t s = 88172645463325252ull;
struct SeedInit { SeedInit() { const char* e = getenv("FUZZ_SEED"); if (e) s ^= strtoull(e, 0, 10) * 0x9e3779b97f4a7c15ull; } } seed_init;
static inline uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 16); }
int main(int argc, char** argv) {
  long iters = atol(argv[1]);
  long ok = 0, err = 0, total = 0;
  for (int f = 2; f < argc; ++f) {
    FILE* fp = fopen(argv[f], "rb");
    if (!fp) return 2;
    std::vector<unsigned char> src;
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) src.insert(src.end(), buf, buf + n);
    fclose(fp);
    for (long it = 0; it < iters; ++it) {
      // exact-size heap copy: any read past the end is an ASan report
      size_t len = src.size();
      const int mode = rnd() % 6;
      if (mode == 0) len = rnd() % (src.size() + 1);                       // truncate anywhere
      else if (mode == 1) { const size_t t = 2 + rnd() % 1200; len = t < src.size() ? t : src.size(); }   // truncate inside the headers
      unsigned char* d = (unsigned char*)malloc(len ? len : 1);
      memcpy(d, src.data(), len);
      if (len > 4) {
        const int nmut = mode == 2 ? 1 : mode == 3 ? 8 : mode == 4 ? 64 : mode == 5 ? 3 : 0;
        for (int k = 0; k < nmut; ++k) {
          size_t pos = (mode == 5 || (rnd() & 1)) ? rnd() % (len < 700 ? len : 700) : rnd() % len;   // headers get half of the hits
          const int what = rnd() % 4;
          d[pos] = what == 0 ? (unsigned char)rnd() : what == 1 ? 0xff : what == 2 ? 0x00 : (unsigned char)(d[pos] ^ (1u << (rnd() % 8)));
        }
      }
      fp_jpeg_info info;
      int rc = fp_jpeg_parse(d, len, &info);
      if (rc == 0) {
        if (info.n_coefs < 0 || info.n_coefs > (1L << 28)) { free(d); ++err; continue; }   // (a caller allocates this much)
        int16_t* co = (int16_t*)malloc((size_t)info.n_coefs * 2 + 2);
        rc = fp_jpeg_entropy_decode(d, len, &info, co);
        free(co);
      }
      rc == 0 ? ++ok : ++err;
      ++total;
      free(d);
    }
  }
  printf("%ld inputs: %ld decoded, %ld refused\n", total, ok, err);
  return 0;
}
