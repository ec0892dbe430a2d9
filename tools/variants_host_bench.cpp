// Host side of the variant modes without a device: mfx_variants_run_values driven by a synthetic `values` (every path
// k-mer gets counts derived from a hash of its text), on a config-4-shaped input (one call per ~765 bases).  For profiling
// the VCF load / clustering / path enumeration / scoring / selection code on any machine:
//   g++ -O2 -std=c++17 tools/variants_host_bench.cpp -Imerfin_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -Lmerfin_amd -lmerfin_amd \
//       -Wl,-rpath,$PWD/merfin_amd -Wl,-rpath,/opt/rocm/lib -o tools/_build/variants_host_bench && MFX_VAR_TIMING=1 tools/_build/variants_host_bench 300e6
#include <chrono>
#include <functional>
#include <algorithm>
#include <random>

#include "mfx_internal.h"

using PathValues = std::function<int(const char *, uint64_t, uint32_t *, uint32_t *)>;
using PathScores = std::function<int(const char *, uint64_t, const mfx_path_table &, const mfx_trv_batch *, int, uint32_t *, double *)>;
int mfx_variants_run_values(const mfx_eval *ev, const PathValues &values, const char *vcf_path, const char *const *names, const char *const *bases,
                            const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts, const char *out_path, const char *log_path,
                            uint64_t *n_clusters, const PathScores &scores, struct mfx_vcf *loaded, uint32_t prepK = 0, struct PathClaims *claims = nullptr);
extern "C" int mfx_vcf_prepare(struct mfx_vcf *vcf, int k, const char *const *names, const char *const *bases, const uint64_t *lens, uint32_t ncontigs,
                               const mfx_variant_opts *opts);
extern "C" struct mfx_vcf *mfx_vcf_load(const char *vcf_path);
extern "C" void mfx_vcf_free(struct mfx_vcf *);

int main(int argc, char **argv) {
  const uint64_t total = argc > 1 ? (uint64_t)atof(argv[1]) : 100000000ull;
  const int mode = argc > 2 ? atoi(argv[2]) : MFX_VAR_POLISH;
  const uint32_t nc = 24;
  const bool varied = argc > 6 && atoi(argv[6]);
  std::mt19937_64 rng(7);
  std::vector<std::string> contigs(nc), names(nc);
  std::string vcf = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE\n";
  uint64_t calls = 0;
  for (uint32_t c = 0; c < nc; ++c) {
    names[c] = "ctg" + std::to_string(c);
    std::string &s = contigs[c];
    s.resize(total / nc);
    for (size_t i = 0; i < s.size(); i += 32) {
      uint64_t x = rng();
      for (size_t j = i; j < std::min(s.size(), i + 32); ++j, x >>= 2) s[j] = "ACGT"[x & 3];
    }
    // as tests/test_gpu_cfg4_fullsize.py: half of the calls alone, half in tight groups of 4 within 40 bases (DeepVariant-like)
    const size_t nv = (size_t)(s.size() / 765.0);
    std::vector<size_t> pos;
    for (size_t i = 0; i < nv / 2; ++i) pos.push_back(30 + rng() % (s.size() - 60));
    for (size_t i = 0; i < nv / 8; ++i) {
      const size_t p0 = 30 + rng() % (s.size() - 230);
      for (int q = 0; q < 4; ++q) pos.push_back(p0 + rng() % 40);
    }
    std::sort(pos.begin(), pos.end());
    pos.erase(std::unique(pos.begin(), pos.end()), pos.end());
    for (size_t p : pos) {
      char buf[512];
      const char ref = s[p];
      const char alt = "ACGT"[(std::string("ACGT").find(ref) + 1 + rng() % 3) % 4];
      const unsigned kind = varied ? (unsigned)(rng() % 20) : 0u;        // argv[6] = 1: indels, two ALTs, every genotype form
      static const char *gts[] = {"1/1", "0/1", "1|0", "1/2", "./.", "0/0", "1/1:30:12", "2/1"};
      const char *gt = varied ? gts[rng() % 8] : "1/1";
      if (kind < 12) snprintf(buf, sizeof(buf), "%s\t%zu\t.\t%c\t%c\t30\tPASS\t.\tGT\t%s\n", names[c].c_str(), p + 1, ref, alt, gt);
      else if (kind < 15) {                                              // deletion of 1..6 bases
        const size_t L = 2 + rng() % 6;
        snprintf(buf, sizeof(buf), "%s\t%zu\t.\t%s\t%c\t%.1f\tPASS\t.\tGT:DP\t%s\n", names[c].c_str(), p + 1, s.substr(p, L).c_str(), ref, 10.0 + (double)(rng() % 400) / 10, gt);
      } else if (kind < 18) {                                            // insertion
        std::string ins(1, ref);
        for (size_t q = 0, L = 1 + rng() % 5; q < L; ++q) ins += "ACGT"[rng() % 4];
        snprintf(buf, sizeof(buf), "%s\t%zu\tid%zu\t%c\t%s\t50\t.\tDP=3\tGT\t%s\n", names[c].c_str(), p + 1, p, ref, ins.c_str(), gt);
      } else {                                                           // two ALTs, one of them equal to REF now and then
        const char alt2 = (rng() % 4 == 0) ? ref : "ACGT"[rng() % 4];
        snprintf(buf, sizeof(buf), "%s\t%zu\t.\t%c\t%c,%c%c\t30\tPASS\t.\tGT\t%s\n", names[c].c_str(), p + 1, ref, alt, alt2, "ACGT"[rng() % 4], gt);
      }
      vcf += buf;
      ++calls;
    }
  }
  if (const char *dm = getenv("MFX_VHB_DAMAGE")) {               // sanitizer drives: bytes of the VCF damaged (seed = the value)
    std::mt19937_64 dr((uint64_t)atoll(dm));
    const size_t nd = 20 + dr() % 200;
    for (size_t q = 0; q < nd && !vcf.empty(); ++q) {
      const size_t at = dr() % vcf.size();
      switch (dr() % 5) {
        case 0: vcf[at] = "\t\n,/|.:0#-"[dr() % 10]; break;
        case 1: vcf[at] = (char)(dr() & 0x7f); break;
        case 2: vcf.erase(at, dr() % 40); break;
        case 3: vcf.insert(at, std::string(1 + dr() % 60, "ACGT9"[dr() % 5])); break;
        default: vcf.insert(at, "9999999999"); break;           // positions / allele indices far outside
      }
    }
  }
  const char *vp = "/tmp/mfx_vhb.vcf";
  FILE *f = fopen(vp, "w");
  fwrite(vcf.data(), 1, vcf.size(), f);
  fclose(f);
  mfx_index ix;
  ix.k = 21;
  mfx_eval ev;
  ev.ix = &ix;
  ev.peak = 26.0;
  PathValues values = [](const char *text, uint64_t len, uint32_t *rv, uint32_t *av) -> int {
    for (uint64_t i = 0; i < len; ++i) {
      uint64_t h = (i * 0x9E3779B97F4A7C15ull) ^ (uint64_t)(unsigned char)text[i] * 0xD6E8FEB86659FD93ull;
      h ^= h >> 29;
      rv[i] = (h % 37 == 0) ? 0u : 20u + (uint32_t)(h % 13);
      av[i] = 1u + (uint32_t)((h >> 20) % 7 == 0);
    }
    return 0;
  };
  // argv[3] = 1: the device-scoring form of the pipeline (what the product runs), with a stand-in for mfx_score_paths
  const bool dev = argc > 3 && atoi(argv[3]);
  // (the stand-in depends on a path's bases alone -- not on where it lies in the batch -- so that runs with the clusters enumerated by
  // mfx_traverse_cluster (MFX_VAR_DEVICE_TRAVERSE, here run on the host in the device's place) and by the host's recursion can be compared)
  PathScores scores = [](const char *text, uint64_t, const mfx_path_table &pt, const mfx_trv_batch *tb, int need_dk, uint32_t *numM, double *totdk) -> int {
    auto score1 = [&](const char *sq, uint32_t n, uint64_t q) {
      uint64_t h = 0xcbf29ce484222325ULL;
      for (uint32_t i = 0; i < n; ++i) { h ^= (unsigned char)sq[i]; h *= 0x100000001b3ULL; }
      h ^= h >> 29;
      numM[q] = n ? (uint32_t)(h % 3) : 0u;
      totdk[q] = need_dk && n ? (double)(int)(h % 17) - 8.0 : 0.0;
    };
    for (uint64_t p = 0; p < pt.npaths; ++p) score1(text + pt.off[p], pt.len[p], p);
    if (tb && tb->ncl) {
      std::vector<char> T(tb->text_end + 1, '\n');
      std::vector<uint64_t> off(tb->path_cap), voff(tb->path_cap), cfirst(tb->path_cap);
      std::vector<uint32_t> len(tb->path_cap, 0), nv(tb->path_cap), vidx(tb->row_cap), vlen(tb->row_cap);
      mfx_trv_out o;
      o.text = T.data(); o.p_off = off.data(); o.p_voff = voff.data(); o.p_cfirst = cfirst.data(); o.p_len = len.data(); o.p_nv = nv.data();
      o.gt = tb->gt; o.vidx = vidx.data(); o.vlen = vlen.data(); o.table_base = pt.npaths; o.row_base = pt.nvals;
      for (uint64_t c = 0; c < tb->ncl; ++c) {
        tb->status[c] = mfx_traverse_cluster(tb->cl[c], tb->var, tb->al, tb->win_text, tb->al_text, o, &tb->np[c]);
        for (uint32_t q = tb->np[c]; q < tb->cl[c].path_cap; ++q) len[tb->cl[c].path0 + q] = 0;
      }
      for (uint64_t q = 0; q < tb->path_cap; ++q) { tb->p_len[q] = len[q]; score1(T.data() + off[q], len[q], pt.npaths + q); }
    }
    return 0;
  };
  std::vector<const char *> nm(nc), bs(nc);
  std::vector<uint64_t> ln(nc);
  for (uint32_t c = 0; c < nc; ++c) { nm[c] = names[c].c_str(); bs[c] = contigs[c].data(); ln[c] = contigs[c].size(); }
  mfx_variant_opts vo{mode, argc > 5 ? (uint32_t)atoi(argv[5]) : 15u, 0, argc > 4 && argv[4][0] ? argv[4] : nullptr};   // argv[4] = -debug file, argv[5] = -comb
  uint64_t ncl = 0;
  auto t0 = std::chrono::steady_clock::now();
  // argv[7] = 1: the VCF loaded ahead of the run (mfx_vcf_load), as the CLI does under its index build
  // argv[7] = 2: ... and prepared (mfx_vcf_prepare: clusters merged, paths enumerated and packed ahead); argv[8]: tag of the output files
  struct mfx_vcf *ahead = (argc > 7 && atoi(argv[7])) ? mfx_vcf_load(vp) : nullptr;
  if (ahead && atoi(argv[7]) == 2) {
    if (mfx_vcf_prepare(ahead, ix.k, nm.data(), bs.data(), ln.data(), nc, &vo)) { fprintf(stderr, "prepare: %s\n", mfx_last_error()); return 1; }
    printf("prepared in %.2f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    t0 = std::chrono::steady_clock::now();
  }
  const std::string tag = argc > 8 ? argv[8] : "";
  const std::string outp = "/tmp/mfx_vhb" + tag + ".out.vcf", logp = "/tmp/mfx_vhb" + tag + ".log";
  int rc = mfx_variants_run_values(&ev, values, ahead ? nullptr : vp, nm.data(), bs.data(), ln.data(), nc, &vo, outp.c_str(), logp.c_str(), &ncl, dev ? scores : PathScores(), ahead);
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (ahead) { auto tf = std::chrono::steady_clock::now(); mfx_vcf_free(ahead); printf("mfx_vcf_free: %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tf).count()); }
  printf("rc %d: %lu bases, %lu calls, %lu clusters in %.2f s = %.0f clusters/s\n", rc, (unsigned long)total, (unsigned long)calls, (unsigned long)ncl, dt, ncl / dt);
  return rc;
}
