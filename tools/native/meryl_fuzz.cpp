// The meryl-directory decoder (csrc/mfx_db.cpp) under AddressSanitizer + UBSan on damaged copies of a database written by tests/meryl_layout.py:
//   hipcc -fsanitize=address,undefined -g -O1 -std=c++17 tools/native/meryl_fuzz.cpp merfin_amd/csrc/mfx_db.cpp -Imerfin_amd/csrc -Iinclude -Lmerfin_amd -lmerfin_amd \
//         -Wl,-rpath,$PWD/merfin_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/meryl_fuzz && ASAN_OPTIONS=detect_leaks=0 /tmp/meryl_fuzz <db.meryl> ...
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <random>
#include <dirent.h>
#include "merfin_amd.h"
static std::vector<char> slurp(const std::string &p) { std::vector<char> b; if (FILE *f = fopen(p.c_str(), "rb")) { char t[65536]; size_t n; while ((n = fread(t, 1, sizeof t, f)) > 0) b.insert(b.end(), t, t + n); fclose(f); } return b; }
static void spit(const std::string &p, const std::vector<char> &b, size_t n) { FILE *f = fopen(p.c_str(), "wb"); fwrite(b.data(), 1, n, f); fclose(f); }
int main(int argc, char **argv) {
  std::mt19937_64 rng(3);
  for (int a = 1; a < argc; ++a) {
    const std::string dir = argv[a];
    mfx_db_info info; uint64_t nk = 0;
    int rp = mfx_db_probe(dir.c_str(), &info), rv = mfx_db_convert(dir.c_str(), "/tmp/dbt/mconv.mfxk", &nk);
    printf("%s: probe rc=%d k=%d n=%lu convert rc=%d n=%lu\n", dir.c_str(), rp, info.k, (unsigned long)info.n_kmers, rv, (unsigned long)nk);
    std::vector<std::string> files;
    DIR *d = opendir(dir.c_str());
    while (dirent *e = readdir(d)) if (e->d_name[0] != '.') files.push_back(e->d_name);
    closedir(d);
    int errs = 0, oks = 0;
    for (int t = 0; t < 300; ++t) {
      const std::string f = dir + "/" + files[rng() % files.size()];
      const std::vector<char> img = slurp(f);
      if (img.empty()) continue;
      std::vector<char> dmg = img;
      size_t n = dmg.size();
      if (t % 4 == 0) n = rng() % dmg.size();                                  // truncated
      else for (int q = 0; q < 1 + (int)(rng() % 3); ++q) dmg[rng() % dmg.size()] ^= (char)(1 << (rng() % 8));
      spit(f, dmg, n);
      const int r1 = mfx_db_probe(dir.c_str(), &info), r2 = mfx_db_convert(dir.c_str(), "/tmp/dbt/mconv.mfxk", &nk);
      (r1 || r2) ? ++errs : ++oks;
      spit(f, img, img.size());
    }
    printf("  300 damaged copies: %d refused, %d accepted (a flipped count bit is a valid database)\n", errs, oks);
  }
  return 0;
}
