// Host side of the k-mer database code (csrc/mfx_db.cpp: probe, text / flat / delta readers, mfx_db_convert, mfx_db_write_flat) under
// AddressSanitizer + UBSan, without a device: well-formed databases of several k and damaged ones (truncated, counts in the header that
// do not fit the file, k-mers wider than 2k bits, unsorted input).  Every call must return -- an error code for the damaged ones.
//   hipcc -fsanitize=address,undefined -g -O1 -std=c++17 tools/native/db_sanitize.cpp merfin_amd/csrc/mfx_db.cpp -Imerfin_amd/csrc -Iinclude \
//         -Lmerfin_amd -lmerfin_amd -Wl,-rpath,$PWD/merfin_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/db_sanitize && ASAN_OPTIONS=detect_leaks=0 /tmp/db_sanitize /tmp/dbt
#include <algorithm>
#include <random>
#include <string>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "merfin_amd.h"

static std::vector<char> slurp(const std::string &p) { std::vector<char> b; if (FILE *f = fopen(p.c_str(), "rb")) { char t[65536]; size_t n; while ((n = fread(t, 1, sizeof t, f)) > 0) b.insert(b.end(), t, t + n); fclose(f); } return b; }
static void spit(const std::string &p, const std::vector<char> &b, size_t n) { FILE *f = fopen(p.c_str(), "wb"); fwrite(b.data(), 1, std::min(n, b.size()), f); fclose(f); }

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  std::mt19937_64 rng(11);
  int bad = 0;
  for (int k : {5, 15, 21, 22, 27, 31, 32, 41, 64}) {
    const size_t kw = k > 31 ? 2 : 1;
    const uint64_t n = k <= 5 ? 300 : 20000;
    std::vector<uint64_t> keys;
    for (uint64_t i = 0; i < n; ++i) {
      uint64_t lo = rng(), hi = rng();
      if (k <= 31) { lo &= (k == 32 ? ~0ull : ((1ull << (2 * k)) - 1)); keys.push_back(lo); }
      else { hi = k == 64 ? hi : hi & ((1ull << (2 * k - 64)) - 1); keys.push_back(lo); keys.push_back(hi); }
    }
    if (k <= 31) { std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end()); }
    const uint64_t m = keys.size() / kw;
    std::vector<uint32_t> vals(m);
    for (auto &v : vals) v = 1 + (uint32_t)(rng() % 4000);
    for (const char *env : {"", "MFX_FLAT_DELTA=0", "MFX_FLAT_PACKED=0"}) {
      if (*env) putenv(const_cast<char *>(env));
      const std::string flat = dir + "/k" + std::to_string(k) + ".mfxk", conv = dir + "/conv.mfxk";
      int rc = mfx_db_write_flat(flat.c_str(), k, keys.data(), vals.data(), m);
      mfx_db_info info;
      int rp = mfx_db_probe(flat.c_str(), &info);
      uint64_t nk = 0;
      int rv = mfx_db_convert(flat.c_str(), conv.c_str(), &nk);
      printf("k=%d %-18s write rc=%d probe rc=%d (k %d, %lu k-mers) convert rc=%d (%lu)%s\n", k, env, rc, rp, info.k, (unsigned long)info.n_kmers, rv, (unsigned long)nk,
             (rc || rp || (rv && k <= 31) || info.n_kmers != m) ? "  <-- UNEXPECTED" : "");
      bad += (rc || rp || info.n_kmers != m);
      // damaged copies: truncated at many lengths, header counts blown up, a key wider than 2k bits
      const std::vector<char> img = slurp(flat);
      for (size_t cut : {(size_t)0, (size_t)7, (size_t)16, (size_t)40, img.size() / 3, img.size() / 2, img.size() - 9, img.size() - 1}) {
        spit(dir + "/dam.mfxk", img, cut);
        (void)mfx_db_probe((dir + "/dam.mfxk").c_str(), &info);
        (void)mfx_db_convert((dir + "/dam.mfxk").c_str(), conv.c_str(), &nk);
      }
      for (int t = 0; t < 40; ++t) {                       // random bytes of the header / directory / payload flipped
        std::vector<char> d = img;
        const size_t at = t < 20 ? rng() % std::min<size_t>(d.size(), 96) : rng() % d.size();
        d[at] = (char)(d[at] ^ (1 << (rng() % 8)) ^ (t % 3 == 0 ? 0xff : 0));
        spit(dir + "/dam.mfxk", d, d.size());
        (void)mfx_db_probe((dir + "/dam.mfxk").c_str(), &info);
        (void)mfx_db_convert((dir + "/dam.mfxk").c_str(), conv.c_str(), &nk);
      }
      if (*env) { std::string e(env); e = e.substr(0, e.find('=')); unsetenv(e.c_str()); }
    }
    // `meryl print` text: well formed, then with a damaged line
    if (k <= 31) {
      const std::string txt = dir + "/k" + std::to_string(k) + ".txt";
      FILE *f = fopen(txt.c_str(), "w");
      for (uint64_t i = 0; i < m; ++i) { char s[65]; for (int b = 0; b < k; ++b) s[b] = "ACTG"[(keys[i] >> (2 * (k - 1 - b))) & 3]; s[k] = 0; fprintf(f, "%s\t%u\n", s, vals[i]); }
      fclose(f);
      mfx_db_info info; uint64_t nk = 0;
      int rp = mfx_db_probe(txt.c_str(), &info), rv = mfx_db_convert(txt.c_str(), (dir + "/conv.mfxk").c_str(), &nk);
      printf("k=%d text probe rc=%d convert rc=%d (%lu of %lu)%s\n", k, rp, rv, (unsigned long)nk, (unsigned long)m, (rp || rv || nk != m) ? "  <-- UNEXPECTED" : "");
      bad += (rp || rv || nk != m);
      std::vector<char> img = slurp(txt);
      for (int t = 0; t < 20; ++t) { std::vector<char> d = img; d[rng() % d.size()] = "X\t\n 9"[t % 5]; spit(dir + "/dam.txt", d, d.size()); (void)mfx_db_probe((dir + "/dam.txt").c_str(), &info); (void)mfx_db_convert((dir + "/dam.txt").c_str(), (dir + "/conv.mfxk").c_str(), &nk); }
    }
  }
  printf("done, unexpected: %d\n", bad);
  return bad != 0;
}
