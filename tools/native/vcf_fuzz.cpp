// mfx_vcf_load (csrc/mfx_variants.cpp: the VCF reader and record parser) under AddressSanitizer + UBSan on damaged copies of a VCF:
//   hipcc -fsanitize=address,undefined -g -O1 -std=c++17 tools/native/vcf_fuzz.cpp merfin_amd/csrc/mfx_variants.cpp -Imerfin_amd/csrc -Iinclude -Lmerfin_amd -lmerfin_amd \
//         -Wl,-rpath,$PWD/merfin_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/vcf_fuzz && ASAN_OPTIONS=detect_leaks=0 /tmp/vcf_fuzz tests/golden/case1.vcf
#include <cstdio>
#include <random>
#include <string>
#include <vector>
#include "merfin_amd.h"
int main(int argc, char **argv) {
  std::vector<char> img;
  if (FILE *f = fopen(argv[1], "rb")) { char t[65536]; size_t n; while ((n = fread(t, 1, sizeof t, f)) > 0) img.insert(img.end(), t, t + n); fclose(f); }
  std::mt19937_64 rng(9);
  int ok = 0, bad = 0;
  for (int t = 0; t < 2000; ++t) {
    std::vector<char> d = img;
    const int kind = t % 5;
    if (kind == 0) d.resize(rng() % (d.size() + 1));                                                   // truncated (also mid-line, without final newline)
    else if (kind == 1) for (int q = 0; q < 8; ++q) d[rng() % d.size()] = "\t\n\r,/|.:0#"[rng() % 10];  // separators moved
    else if (kind == 2) for (int q = 0; q < 8; ++q) d[rng() % d.size()] = (char)(rng() & 0xff);          // arbitrary bytes (NUL included)
    else if (kind == 3) { const size_t a = rng() % d.size(), n = rng() % 200; d.erase(d.begin() + a, d.begin() + std::min(d.size(), a + n)); }
    else { const size_t a = rng() % d.size(); d.insert(d.begin() + a, 1 + rng() % 3000, "9\tA"[rng() % 3]); }   // very long fields / numbers
    FILE *o = fopen("/tmp/vcf_fuzz.vcf", "wb"); fwrite(d.data(), 1, d.size(), o); fclose(o);
    mfx_vcf *v = mfx_vcf_load("/tmp/vcf_fuzz.vcf");
    v ? ++ok : ++bad;
    mfx_vcf_free(v);
  }
  printf("2000 damaged VCFs: %d loaded (damaged lines are excluded or parsed as the reference's splitToWords would), %d refused\n", ok, bad);
  return 0;
}
