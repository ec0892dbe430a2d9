// mfx_variants.cpp -- merfin's variant modes (-filter / -polish / -better /
// -strict / -loose) on top of the GPU lookup path.
//
// Reference behaviour implemented (paths relative to /root/reference):
//   VCF model + clustering    src/merfin/vcfRecord.H:50-100, vcf.C:23-87, 93-149, 156-246
//   allele-combination DFS    src/merfin/merfin-variants.C:22-126   (traverse)
//   per-cluster driver        src/merfin/merfin-variants.C:131-345
//   scoring + selectors       src/merfin/varMer.C:37-659
//
// Shape of this implementation (not the reference's): MANY clusters go through the device at once.  A batch's clusters are handed over as
// TABLES (window bases, variants, alleles) and every cluster's allele combinations are enumerated there by one wave (mfx_traverse.h,
// mfx_var_traverse_kernel: the traverse recursion as a loop) straight into the batch's path text; every k-mer of every path is extracted,
// canonicalised and probed by ONE launch of the lookup kernel -dump uses, and varMer::score runs per path on the device (mfx_score_paths_trv).
// What comes back is per path: its length, its genotype row, numM and the total delta-K -- the host applies the selector of the chosen mode
// and writes the records.  Clusters beyond the device's limits (more than 8 variants or 64 combinations, strings beyond 640 bytes), -debug
// (per-position values) and the sharded index keep the host's enumeration (the same recursion, packed path text) inside the same batches.
// merfin scores cluster by cluster with four CPU probes per base (varMer.C:76-84).
#include "mfx_internal.h"
#include "mfx_pipe.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <list>
#include <map>
#include <new>
#include <type_traits>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <exception>
#include <functional>
#include <future>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---- text helpers ----------------------------------------------------------
// splitToWords semantics at the reference's call sites: runs of separators
// collapse; indexing past the end yields "no word".
std::vector<std::string> split_any(const std::string &s, const char *seps) {
  std::vector<std::string> out;
  size_t i = 0, n = s.size();
  while (i < n) {
    while (i < n && strchr(seps, s[i])) ++i;
    if (i >= n) break;
    size_t b = i;
    while (i < n && !strchr(seps, s[i])) ++i;
    out.emplace_back(s, b, i - b);
  }
  return out;
}


// ---- VCF model -------------------------------------------------------------
// Every text field is a VIEW into the VCF file image, which the database keeps (VcfDB::text): a config-4 call set has
// millions of records, and a std::string per field (ten per record, plus the ALT list) was most of the load time.
struct SV {                           // a piece of the file image
  const char *p = nullptr;
  uint32_t n = 0;
  size_t size() const { return n; }
  bool eq(const SV &o) const { return n == o.n && memcmp(p, o.p, n) == 0; }
  std::string str() const { return std::string(p, n); }
};
inline std::string &operator+=(std::string &s, const SV &v) { return s.append(v.p, v.n); }

// word `i` of s split at any char of `seps`, splitToWords semantics (runs of separators collapse); n = 0 and p = nullptr
// when there is no such word
inline SV word_at(const SV &s, const char *seps, size_t i) {
  size_t at = 0, w = 0;
  while (at < s.n) {
    while (at < s.n && strchr(seps, s.p[at])) ++at;
    if (at >= s.n) break;
    size_t b = at;
    while (at < s.n && !strchr(seps, s.p[at])) ++at;
    if (w++ == i) { SV r; r.p = s.p + b; r.n = (uint32_t)(at - b); return r; }
  }
  return SV();
}
inline size_t count_words(const SV &s, const char *seps) {
  size_t at = 0, w = 0;
  while (at < s.n) {
    while (at < s.n && strchr(seps, s.p[at])) ++at;
    if (at >= s.n) break;
    while (at < s.n && !strchr(seps, s.p[at])) ++at;
    ++w;
  }
  return w;
}

struct Record {                       // vcfRecord
  SV chr, id, ref, alts, filter, info, formats, samples;
  SV gt_field;                       // _arr_samples[0]
  uint32_t pos = 0, n_alts = 0;      // n_alts = words of ALT (_arr_alts)
  double qual = 0;
  bool ok = false;                   // false: fewer than 10 columns, "excluded" (vcfRecord.H:53-56)
  std::string line() const { std::string o; line_to(o); return o; }
  void line_to(std::string &o) const {    // vcfRecord::save, vcfRecord.H:96-97
    char q[64];
    snprintf(q, sizeof(q), "%.1f", qual);
    o += chr; o += '\t'; o += std::to_string((int)pos); o += '\t'; o += id; o += '\t'; o += ref; o += '\t'; o += alts; o += '\t';
    o += q; o += '\t'; o += filter; o += '\t'; o += info; o += '\t'; o += formats; o += '\t'; o += samples; o += '\n';
  }
};

struct Variant {                      // gtAllele
  const Record *rec = nullptr;
  uint32_t pos = 0, refLen = 0;
  double qual = 0;
  // [0] = REF; none for ./. and 0/0 genotypes.  Two or three alleles is the rule: they live in the object.
  SV few[4];
  uint32_t n_alleles = 0;
  std::vector<SV> many;              // only beyond four alleles
  size_t nalleles() const { return n_alleles; }
  const SV &allele(size_t i) const { return n_alleles <= 4 ? few[i] : many[i]; }
  void push(const SV &a) {
    if (n_alleles < 4) few[n_alleles] = a;
    else {
      if (n_alleles == 4) many.assign(few, few + 4);
      many.push_back(a);
    }
    ++n_alleles;
  }
};

// the variants of a cluster: one to three is the rule (no heap block then)
struct VarList {
  const Variant *few[3] = {nullptr, nullptr, nullptr};
  std::vector<const Variant *> many;                     // all of them, once there are more than three
  uint32_t n = 0;
  size_t size() const { return n; }
  const Variant *const *begin() const { return n <= 3 ? few : many.data(); }
  const Variant *const *end() const { return begin() + n; }
  const Variant *operator[](size_t i) const { return begin()[i]; }
  void push_back(const Variant *v) {
    if (n < 3) few[n] = v;
    else {
      if (n == 3) many.assign(few, few + 3);
      many.push_back(v);
    }
    ++n;
  }
};

struct Cluster {                      // posGT
  uint32_t rStart, rEnd;
  VarList vars;
};

// n objects whose memory is touched first by the thread that CONSTRUCTS them (placement new in the parsing loop): a config-4
// call set makes 1.1 GB of records and variants, and a std::vector's resize would zero and fault all of it in on one thread
// before the 16 parsers start.  Every element must be constructed by the owner before the array is destroyed.
template <class T> struct RawArray {
  T *p = nullptr;
  size_t n = 0;
  RawArray() = default;
  RawArray(const RawArray &) = delete;
  RawArray &operator=(const RawArray &) = delete;
  // Elements are placement-constructed by the threads that parse them (make(i)), and a worker that throws -- or a load that returns
  // early -- leaves some unconstructed: one flag byte per element (calloc: untouched pages cost nothing) says which ones the
  // destructor may destroy.
  uint8_t *made = nullptr;
  bool alloc(size_t n_) {
    p = n_ ? static_cast<T *>(malloc(n_ * sizeof(T))) : nullptr;
    made = (p && !std::is_trivially_destructible<T>::value) ? static_cast<uint8_t *>(calloc(n_, 1)) : nullptr;
    if (p && !std::is_trivially_destructible<T>::value && !made) { free(p); p = nullptr; }
    n = p ? n_ : 0;
    return p || !n_;
  }
  T *make(size_t i) {                 // constructs element i (once, by one thread)
    T *x = new (&p[i]) T();
    if (made) made[i] = 1;
    return x;
  }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
  ~RawArray() {
    if (made) {
      // (a config-4 call set: 3.8 M records and as many variants -- their destructors, one thread: 0.1 s at the end of a 1.6 s run)
      const unsigned nt = n >= (1u << 20) ? std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1u;
      auto piece = [this](size_t b, size_t e) { for (size_t i = b; i < e; ++i) if (made[i]) p[i].~T(); };
      if (nt <= 1) piece(0, n);
      else {
        std::vector<std::thread> th;
        try {
          for (unsigned t = 0; t < nt; ++t) th.emplace_back(piece, n * t / nt, n * (t + 1) / nt);
        } catch (...) {                                        // (no more threads to be had: the rest on this one)
          const size_t done = th.size();
          for (auto &x : th) x.join();
          th.clear();
          piece(n * done / nt, n);
        }
        for (auto &x : th) x.join();
      }
    }
    free(made);
    free(p);
  }
};

struct VcfDB {
  std::string text;                   // the file image every SV points into
  std::vector<std::string> headers;
  RawArray<Record> rec_store;         // one per data line (excluded ones included), file order
  RawArray<Variant> var_store;
  size_t n_records = 0;               // records loaded (not excluded)
  std::map<std::string, std::vector<Cluster *>> by_chr;
  std::map<std::string, std::vector<Cluster>> cluster_store;   // per CHROM: one Cluster per record (the merged-away ones stay unused); by_chr points into it
  std::map<std::string, std::vector<std::pair<size_t, size_t>>> runs;   // per CHROM: its runs [begin, end) of rec_store, file order
  uint64_t excluded = 0;
  int contig_ids = 0;
  ~VcfDB() {
    // the clusters of the CHROMs are released side by side (one Cluster per record: millions of small destructors)
    std::vector<std::vector<Cluster> *> big;
    for (auto &kv : cluster_store) if (kv.second.size() >= (1u << 16)) big.push_back(&kv.second);
    if (big.size() > 1) {
      std::atomic<size_t> next{0};
      auto work = [&]() { for (size_t i; (i = next.fetch_add(1)) < big.size();) std::vector<Cluster>().swap(*big[i]); };
      std::vector<std::thread> th;
      try { for (size_t t = 1; t < std::min<size_t>(big.size(), 16); ++t) th.emplace_back(work); } catch (...) {}
      work();
      for (auto &x : th) x.join();
    }
  }
};

// gtAllele::gtAllele, vcf.C:23-87
void make_variant(const Record *r, Variant *v) {
  v->rec = r;
  v->pos = r->pos - 1;
  v->refLen = (uint32_t)r->ref.size();
  v->qual = r->qual;
  const SV &g = r->gt_field;
  if (g.n >= 3 && (memcmp(g.p, "./.", 3) == 0 || memcmp(g.p, "0/0", 3) == 0))
    return;                                              // no alleles at all
  v->push(r->ref);
  int seen_alt[8];
  size_t n_seen = 0;
  std::vector<int> seen_more;
  for (size_t ti = 0;; ++ti) {
    const SV tok = word_at(g, "|/", ti);
    if (!tok.p) break;
    long altIdx = 0;
    bool plain = tok.n >= 1 && tok.n <= 9;                 // plain digits (the rule) are read here, anything else by strtol
    for (uint32_t i = 0; i < tok.n && plain; ++i) {
      const unsigned d = (unsigned char)tok.p[i] - (unsigned)'0';
      if (d > 9) plain = false;
      altIdx = altIdx * 10 + (long)d;
    }
    if (!plain) {
      char num[24];
      const size_t nn = std::min<size_t>(tok.n, sizeof(num) - 1);
      memcpy(num, tok.p, nn);
      num[nn] = 0;
      altIdx = strtol(num, nullptr, 10);
    }
    if ((int32_t)altIdx <= 0) continue;
    if ((size_t)altIdx > r->n_alts) continue;            // operator[] past the end -> nullptr
    // the reference compares POINTERS into the ALT list (same ALT index listed twice) ...
    bool dup = std::find(seen_alt, seen_alt + n_seen, (int)altIdx) != seen_alt + n_seen ||
               std::find(seen_more.begin(), seen_more.end(), (int)altIdx) != seen_more.end();
    if (dup) continue;
    const SV hap = word_at(r->alts, ",", (size_t)altIdx - 1);
    if (hap.eq(r->ref)) continue;                        // ... and the STRING against the reference allele only
    if (n_seen < 8) seen_alt[n_seen++] = (int)altIdx; else seen_more.push_back((int)altIdx);
    v->push(hap);
  }
}

// one data line -> record (ok = false: fewer than 10 columns, "excluded", vcfRecord.H:53-56)
void parse_record(const char *L, size_t n, Record *r) {
  SV line;
  line.p = L; line.n = (uint32_t)n;
  SV w[10];
  {
    // the first ten words of the line split at tabs, splitToWords semantics (runs of tabs collapse), in one pass
    size_t at = 0;
    int nw = 0;
    while (nw < 10) {
      while (at < n && L[at] == '\t') ++at;
      if (at >= n) break;
      const char *e = (const char *)memchr(L + at, '\t', n - at);
      const size_t end = e ? (size_t)(e - L) : n;
      w[nw].p = L + at; w[nw].n = (uint32_t)(end - at);
      ++nw;
      at = end;
    }
    if (nw < 10) return;
  }
  r->ok = true;
  r->chr = w[0];
  {
    // POS: plain digits that fit 32 bits are read here, anything else by strtoul as before
    bool plain = w[1].n >= 1 && w[1].n <= 9;
    uint32_t pv = 0;
    for (uint32_t i = 0; i < w[1].n && plain; ++i) {
      const unsigned d = (unsigned char)w[1].p[i] - (unsigned)'0';
      if (d > 9) plain = false;
      pv = pv * 10 + d;
    }
    if (plain) r->pos = pv;
    else {
      char num[32];
      const size_t nn = std::min<size_t>(w[1].n, sizeof(num) - 1);
      memcpy(num, w[1].p, nn); num[nn] = 0;
      r->pos = (uint32_t)strtoul(num, nullptr, 10);
    }
    // QUAL: digits[.digits] with at most 15 significant digits is an integer below 2^53 over an exact power of ten -- one
    // correctly rounded division, which is what strtod returns for it; every other spelling (sign, exponent, '.', inf, nan,
    // longer mantissas) goes to strtod
    static const double P10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
    uint64_t mant = 0;
    uint32_t nd = 0, frac = 0;
    bool fast = w[5].n >= 1 && w[5].n <= 16, dot = false;
    for (uint32_t i = 0; i < w[5].n && fast; ++i) {
      const char ch = w[5].p[i];
      if (ch == '.') { if (dot || i == 0 || i + 1 == w[5].n) fast = false; dot = true; continue; }
      const unsigned d = (unsigned char)ch - (unsigned)'0';
      if (d > 9) { fast = false; break; }
      mant = mant * 10 + d;
      ++nd;
      if (dot) ++frac;
    }
    if (fast && nd >= 1 && nd <= 15) r->qual = (double)mant / P10[frac];
    else {
      char q[64];
      const size_t nq = std::min<size_t>(w[5].n, sizeof(q) - 1);
      memcpy(q, w[5].p, nq); q[nq] = 0;
      r->qual = strtod(q, nullptr);
    }
  }
  r->id = w[2]; r->ref = w[3]; r->alts = w[4];
  r->filter = w[6]; r->info = w[7]; r->formats = w[8]; r->samples = w[9];
  r->n_alts = (uint32_t)count_words(r->alts, ",");
  r->gt_field = word_at(r->samples, ":", 0);
  if (!r->gt_field.p) r->gt_field = SV();
}

template <class F> void parallel_for(size_t n, F &&fn);

// vcfFile::loadFile, vcf.C:93-149.  The file is read whole, its data lines are parsed by the host threads (a
// config-4-sized VCF has millions of records, and the per-record string work is most of the load time), and the
// records enter the database in FILE ORDER, exactly as a sequential read would put them.
int load_vcf(const char *path, VcfDB &db, double *t_sub = nullptr) {   // t_sub[4]: read, line scan, parse, runs (seconds; diagnostics)
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t0 = now();
  auto sub = [&](int i) { const double t = now(); if (t_sub) t_sub[i] += t - t0; t0 = t; };
  mfx_file fh = mfx_open_reader(path);
  FILE *f = fh.f;
  if (!f) return mfx_fail(MFX_E_IO, "cannot open VCF '%s'", path);
  std::string &buf = db.text;
  {
    struct stat st;
    if (!fh.is_pipe() && fstat(fileno(f), &st) == 0 && S_ISREG(st.st_mode)) buf.reserve((size_t)st.st_size + 1);
    std::vector<char> blk(1 << 22);
    size_t n;
    while ((n = fread(blk.data(), 1, blk.size(), f)) > 0) buf.append(blk.data(), n);
  }
  if (mfx_close(fh)) return mfx_fail(MFX_E_IO, "reading VCF '%s' failed (stream error or the decompressor exited with an error)", path);
  sub(0);
  std::vector<std::pair<size_t, size_t>> lines;           // (offset, length) of every data line
  {
    // the line scan by the host threads: the text is cut at line starts, every thread lists the lines of its piece, the lists
    // are put together in file order (a config-4 call set: 200 MB, 3.9 M lines)
    const size_t NP = buf.size() >= (8u << 20) ? std::max<size_t>(1, std::min<size_t>(mfx_host_threads(), 64)) : 1;
    std::vector<size_t> cut(NP + 1, buf.size());
    cut[0] = 0;
    for (size_t t = 1; t < NP; ++t) {
      const size_t g = std::max(cut[t - 1], buf.size() / NP * t);
      const char *nl = g < buf.size() ? (const char *)memchr(buf.data() + g, '\n', buf.size() - g) : nullptr;
      cut[t] = nl ? (size_t)(nl - buf.data()) + 1 : buf.size();
    }
    struct Piece { std::vector<std::pair<size_t, size_t>> lines; std::vector<std::string> headers; int contig_ids = 0; };
    std::vector<Piece> pieces(NP);
    parallel_for(NP, [&](size_t t) {
      Piece &P = pieces[t];
      const size_t end = cut[t + 1];
      P.lines.reserve((end - cut[t]) / 48 + 16);
      for (size_t o = cut[t]; o < end;) {
        const char *nl = (const char *)memchr(buf.data() + o, '\n', end - o);
        size_t e = nl ? (size_t)(nl - buf.data()) : end, n = e - o;
        while (n > 0 && (buf[o + n - 1] == '\n' || buf[o + n - 1] == '\r')) --n;
        if (buf[o] == '#' && (nl || e > o)) {
          P.headers.emplace_back(buf.data() + o, n);
          if (n >= 12 && strncmp(buf.data() + o, "##contig=<ID", 12) == 0) P.contig_ids++;
        } else if (nl || e > o) {
          P.lines.emplace_back(o, n);
        }
        o = e + 1;
      }
    });
    size_t total = 0;
    for (const Piece &P : pieces) total += P.lines.size();
    lines.reserve(total);
    for (Piece &P : pieces) {
      lines.insert(lines.end(), P.lines.begin(), P.lines.end());
      for (std::string &h : P.headers) db.headers.push_back(std::move(h));
      db.contig_ids += P.contig_ids;
    }
  }
  sub(1);
  // (a config-4 call set: 1.1 GB of records and variants, constructed -- and so first touched -- by the thread that parses them)
  if (!db.rec_store.alloc(lines.size()) || !db.var_store.alloc(lines.size())) return mfx_fail(MFX_E_NOMEM, "VCF '%s': no memory for %zu records", path, lines.size());
  const size_t CH = 4096;                                  // lines per task
  parallel_for((lines.size() + CH - 1) / CH, [&](size_t c) {
    for (size_t i = c * CH, e = std::min(lines.size(), (c + 1) * CH); i < e; ++i) {
      Record *r = db.rec_store.make(i);
      Variant *v = db.var_store.make(i);
      parse_record(buf.data() + lines[i].first, lines[i].second, r);
      if (r->ok) make_variant(r, v);
    }
  });
  sub(2);
  // records of one CHROM come in runs: the runs are noted here (one map lookup per run), the clusters themselves are
  // made per CHROM by the host threads (merge_clusters).  The threads list the runs of their share of the records; runs that
  // meet at a share's border are joined.
  {
    struct Run { SV chr; size_t b, e; };
    const size_t NR = lines.size() >= (1u << 16) ? std::max<size_t>(1, std::min<size_t>(mfx_host_threads(), 64)) : 1;
    std::vector<std::vector<Run>> found(NR);
    std::vector<uint64_t> n_ok(NR, 0), n_bad(NR, 0);
    parallel_for(NR, [&](size_t t) {
      std::vector<Run> &F = found[t];
      for (size_t i = lines.size() * t / NR, e = lines.size() * (t + 1) / NR; i < e; ++i) {
        const Record *r = &db.rec_store[i];
        if (!r->ok) { n_bad[t]++; continue; }
        n_ok[t]++;
        if (F.empty() || F.back().e != i || !F.back().chr.eq(r->chr)) F.push_back(Run{r->chr, i, i});   // another CHROM, or an excluded line in between
        F.back().e = i + 1;
      }
    });
    std::vector<std::pair<size_t, size_t>> *bucket = nullptr;
    SV bucket_chr;
    for (size_t t = 0; t < NR; ++t) {
      db.n_records += n_ok[t];
      db.excluded += n_bad[t];
      for (const Run &R : found[t]) {
        if (!bucket || !bucket_chr.eq(R.chr)) {
          const std::string chr = R.chr.str();
          db.by_chr.emplace(chr, std::vector<Cluster *>());
          bucket = &db.runs[chr];
          bucket_chr = R.chr;
          bucket->emplace_back(R.b, R.e);
        } else if (bucket->back().second != R.b) bucket->emplace_back(R.b, R.e);
        else bucket->back().second = R.e;
      }
    }
  }
  sub(3);
  return MFX_OK;
}

}  // namespace

// a VCF read and parsed ahead of its run (mfx_vcf_load): host work only, so it can run under the index build
namespace { struct VarPrepared; }
struct mfx_vcf {
  VcfDB db;
  double t_load[4] = {0, 0, 0, 0};
  bool used = false;
  VarPrepared *prep = nullptr;            // mfx_vcf_prepare: the clusters merged, their paths enumerated and packed batch by batch
  ~mfx_vcf();
};

namespace {

// vcfFile::mergeChrPosGT, vcf.C:156-246: clusters whose start lies within 2k of
// the previous cluster's end are merged, unless that cluster already holds
// `comb` variants and splitting is allowed.
void merge_clusters(VcfDB &db, uint32_t k, uint32_t comb, bool nosplit, FILE *log) {
  const uint32_t K_OFFSET = 2 * k;
  // one CHROM per task: its clusters are created (one per record, file order), sorted and merged independently of the
  // others; the log lines are printed afterwards in map order, as the sequential loop printed them
  std::vector<std::pair<const std::string *, std::vector<Cluster *> *>> chrs;
  std::vector<std::vector<Cluster> *> stores;
  for (auto &kv : db.by_chr) { chrs.emplace_back(&kv.first, &kv.second); stores.push_back(&db.cluster_store[kv.first]); }
  std::vector<std::string> logs(chrs.size());
  parallel_for(chrs.size(), [&](size_t ci) {
    std::vector<Cluster *> &in = *chrs[ci].second;
    std::vector<Cluster> &store = *stores[ci];
    const auto &runs = db.runs[*chrs[ci].first];
    size_t nrec = 0;
    for (const auto &run : runs) nrec += run.second - run.first;
    store.reserve(nrec);                                   // (never grows below: the pointers into it stay valid)
    in.reserve(nrec);
    for (const auto &run : runs)
      for (size_t i = run.first; i < run.second; ++i) {
        if (!db.rec_store[i].ok) continue;
        const Variant *v = &db.var_store[i];
        store.emplace_back();
        Cluster *c = &store.back();
        c->rStart = v->pos;
        c->rEnd = v->pos + v->refLen;
        c->vars.push_back(v);
        in.push_back(c);
      }
    std::vector<Cluster *> out;
    uint32_t split = 0, merged = 0;
    // same algorithm + comparator as the reference so ties on rStart land in the same order
    std::sort(in.begin(), in.end(), [](Cluster *const &A, Cluster *const &B) { return A->rStart < B->rStart; });
    out.push_back(in[0]);
    for (size_t i = 1; i < in.size(); ++i) {
      Cluster *cur = in[i], *last = out.back();
      bool overlapping = cur->rStart < last->rEnd + K_OFFSET;
      bool toomany = last->vars.size() >= comb;
      if (!overlapping) { out.push_back(cur); continue; }
      if (toomany && !nosplit) { out.push_back(cur); split++; continue; }
      const Variant *v = cur->vars[0];
      last->vars.push_back(v);
      last->rStart = std::min(last->rStart, v->pos);
      last->rEnd = std::max(last->rEnd, v->pos + v->refLen);
      merged++;
    }
    const std::string &nm = *chrs[ci].first;
    logs[ci] = nm + " : Reduced " + std::to_string(in.size()) + " variants down to " + std::to_string(out.size()) + " combinations for evaluation:\n";
    if (split > 0) logs[ci] += nm + " :   Split   " + std::to_string(split) + " complicated combinations.\n";
    if (merged > 0) logs[ci] += nm + " :   Merged  " + std::to_string(merged) + " variants into combinations.\n";
    in.swap(out);
  });
  if (log) for (const std::string &l : logs) fputs(l.c_str(), log);
}

// ---- allele-combination enumeration (traverse) ------------------------------
// varMer's per-cluster containers (seqs, gtPaths, idxPaths, lenPaths), flat: a config-4 call set makes millions of
// paths of ~60 bases, and a std::string + three std::vectors per path was most of the host time (allocation, and the
// page faults of a heap that only grows).  Path p is text[toff[p], toff[p+1] - 1), followed by ONE separator byte ('\n',
// not ACGT: the whole arena is copied into the packed GPU buffer as it is); its genotype / offset / length snapshots
// are rows p of gt / vidx / vlen (nv = variants of the cluster columns each).
struct PathArena {                           // the paths of a RUN of consecutive clusters: one worker fills it (stage A), one reads it (stage C)
  std::string text;
  std::vector<uint64_t> toff;                // [paths + 1]: path q is text[toff[q], toff[q + 1] - 1)
  std::vector<int> gt;                       // gtPaths rows, cluster after cluster (a cluster's rows have its nv columns)
  std::vector<uint32_t> vidx, vlen;          // idxPaths / lenPaths rows (shifted offsets / post-substitution lengths snapshots)
  uint64_t off = 0;                          // where the run's text lies in the batch's packed buffer
  uint64_t p0 = 0, v0 = 0;                   // the run's first path / first row entry in the batch's path table
  uint64_t first_id = 0;                     // varMerId of the run's first path (-debug numbering)
  // the run's clusters that are enumerated ON THE DEVICE (mfx_traverse.h): their tables, and the room reserved for their paths
  std::vector<mfx_trv_cluster> tcl;
  std::vector<mfx_trv_variant> tvar;
  std::vector<mfx_trv_allele> tal;
  std::string twin, talt;                    // window bases / allele bases of those clusters
  uint64_t t_text = 0, t_paths = 0, t_rows = 0;
  uint64_t b_cl = 0, b_var = 0, b_al = 0, b_win = 0, b_alt = 0, b_text = 0, b_path = 0, b_row = 0;   // where the run's entries lie in the batch's
  void reset() {
    text.clear(); toff.clear(); toff.push_back(0); gt.clear(); vidx.clear(); vlen.clear();
    tcl.clear(); tvar.clear(); tal.clear(); twin.clear(); talt.clear(); t_text = t_paths = t_rows = 0;
  }
};

// One cluster's paths: a window of its run's arena.  (Round 3 kept a string + four vectors per cluster: their ~12 heap blocks
// per cluster -- allocated by one thread, freed by another -- were most of the host time of a config-4 call set.)
struct PathSet {
  PathArena *ar = nullptr;
  uint64_t p0 = 0, v0 = 0;                   // first path / first row entry inside the arena
  uint32_t np = 0, nv = 0;
  // a cluster enumerated on the device: what came back of its paths (lengths, genotype rows) and where they stand in the batch's path table
  const uint32_t *dlen = nullptr;
  const int32_t *dgt = nullptr;
  uint64_t dq0 = 0;
  size_t size() const { return np; }
  size_t len(size_t p) const { return dlen ? (size_t)dlen[p] : (size_t)(ar->toff[p0 + p + 1] - ar->toff[p0 + p] - 1); }
  const char *seq(size_t p) const { return ar->text.data() + ar->toff[p0 + p]; }
  uint64_t packed_off(size_t p) const { return ar->off + ar->toff[p0 + p]; }       // of path p in the batch's packed buffer
  uint64_t table_p0() const { return dlen ? dq0 : ar->p0 + p0; }                     // of path 0 in the batch's path table
  uint64_t first_id() const { return ar->first_id + p0; }
  const int *gt(size_t p) const { return dlen ? dgt + p * nv : ar->gt.data() + v0 + p * nv; }
  const uint32_t *vidx(size_t p) const { return ar->vidx.data() + v0 + p * nv; }
  const uint32_t *vlen(size_t p) const { return ar->vlen.data() + v0 + p * nv; }
  static uint64_t hash(const char *s, size_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ULL; }
    return h;
  }
  bool same(size_t p, const std::string &s) const { return len(p) == s.size() && memcmp(seq(p), s.data(), s.size()) == 0; }
  // `seen`: hash -> path indices, used once a cluster has many paths; the caller's scratch, empty at the cluster's start
  void add(const std::string &s, const std::vector<int> &g, const std::vector<uint32_t> &ix, const std::vector<uint32_t> &ln,
           std::unordered_map<uint64_t, std::vector<uint32_t>> &seen) {
    // varMer.C:39: a sequence already present is not added again.  Few paths: compare them all; many: by hash.
    if (np < 32) {
      for (size_t p = 0; p < np; ++p) if (same(p, s)) return;
    } else {
      if (seen.empty()) for (size_t p = 0; p < np; ++p) seen[hash(seq(p), len(p))].push_back((uint32_t)p);
      std::vector<uint32_t> &cand = seen[hash(s.data(), s.size())];
      for (uint32_t p : cand) if (same(p, s)) return;
      cand.push_back(np);
    }
    ar->text.append(s);
    ar->text.push_back('\n');
    ar->toff.push_back(ar->text.size());
    ar->gt.insert(ar->gt.end(), g.begin(), g.end());
    ar->vidx.insert(ar->vidx.end(), ix.begin(), ix.end());
    ar->vlen.insert(ar->vlen.end(), ln.begin(), ln.end());
    ++np;
  }
};

// merfin-variants.C:22-126.  `lens` and `cand` are per-call copies, `offs` and
// `path` are shared -- that asymmetry is part of the observable behaviour
// (the stored snapshots feed the "new k-mer" test of scoring).  `reps`: one string per recursion depth, reused for
// every candidate built at that depth (the reference copies the candidate once per allele); the caller sizes it to the
// cluster's variant count (references into it are held across the recursion: it must not grow here).
using SeenMap = std::unordered_map<uint64_t, std::vector<uint32_t>>;
void enumerate(uint32_t idx, std::vector<uint32_t> &offs, std::vector<uint32_t> lens, const Cluster &cl,
               const std::string &cand, std::vector<int> &path, PathSet &out, std::vector<std::string> &reps, size_t depth, SeenMap &seen) {
  const Variant &var = *cl.vars[idx];
  const uint32_t refLen = lens[idx];
  const uint32_t last = (uint32_t)offs.size() - 1;
  for (int j = 0; j < (int)var.nalleles(); ++j) {
    path.push_back(j);
    std::string &rep = reps[depth];
    rep = cand;
    int skipped = 0, delta = 0;
    if (j > 0) {
      const SV &hap = var.allele(j);
      lens[idx] = refLen;
      rep.replace(offs[idx], lens[idx], hap.p, hap.n);
      delta = (int)hap.size() - (int)lens[idx];
      const uint32_t affected = offs[idx] + lens[idx];
      lens[idx] = (uint32_t)hap.size();
      for (uint32_t i = idx + 1; i < offs.size() && offs[i] < affected; ++i) {   // later variants inside this REF span: forced to REF
        ++idx; path.push_back(0); ++skipped;
      }
      if (skipped > 0 && idx == last) {
        out.add(rep, path, offs, lens, seen);
        for (int q = 0; q < skipped; ++q) { path.pop_back(); --idx; }
        path.pop_back();
        continue;
      }
      for (uint32_t i = idx + 1; i < offs.size(); ++i) offs[i] += delta;
    }
    if (idx + 1 < offs.size()) enumerate(idx + 1, offs, lens, cl, reps[depth], path, out, reps, depth + 1, seen);
    if (idx == last) out.add(reps[depth], path, offs, lens, seen);
    for (uint32_t i = idx + 1; i < offs.size(); ++i) offs[i] -= delta;
    for (int q = 0; q < skipped; ++q) { path.pop_back(); --idx; }
    path.pop_back();
  }
}

// ---- scoring + selection -----------------------------------------------------
struct Scored {
  std::vector<uint32_t> numM;
  std::vector<double> totdk;                 // getTotdK of every path: the sum of its delta-K values in position order
  std::vector<std::vector<double>> ks, dks;  // per position; kept for the -debug statistics only
};

inline int base_ok(unsigned char c) {
  switch (c) { case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1; default: return 0; }
}

struct Job {                          // one cluster waiting for its GPU values
  const Cluster *cl;
  uint32_t contig;
  uint32_t rStart, rEnd;
  PathSet ps;                         // its paths, inside the arena of its run of clusters
  int64_t trv = -1;                   // >= 0: enumerated on the device -- its entry among the run's (then the batch's) traverse clusters
  std::string dbg;                    // -debug lines of this cluster (appended to its run's text by the worker)
};

// A cluster goes to the device's traverse (mfx_traverse.h) if it stays inside its limits; its tables are appended to the run's.  The
// room reserved: the product of the allele counts in paths, each as long as the window plus every variant's longest allele.
bool trv_pack(const Cluster &cl, uint32_t rStart, uint32_t rEnd, const char *contig_bases, PathArena &ar) {
  const size_t nv = cl.vars.size();
  if (nv == 0 || nv > MFX_TRV_MAX_NV) return false;
  uint64_t prod = 1, extra = 0;
  for (const Variant *v : cl.vars) {
    const size_t na = v->nalleles();
    if (na == 0) return false;
    prod *= na;
    if (prod > MFX_TRV_MAX_PATHS) return false;
    size_t longest = 0;
    for (size_t a = 1; a < na; ++a) longest = std::max<size_t>(longest, v->allele(a).size());
    extra += longest;
  }
  const uint64_t win_len = rEnd - rStart, bound = win_len + extra;
  if (bound > MFX_TRV_MAX_LEN) return false;
  mfx_trv_cluster c;
  c.win_off = ar.twin.size();
  c.win_len = (uint32_t)win_len;
  c.nv = (uint32_t)nv;
  c.var0 = (uint32_t)ar.tvar.size();
  c.path_cap = (uint32_t)prod;
  c.text0 = ar.t_text;
  c.path0 = ar.t_paths;
  c.row0 = ar.t_rows;
  c.text_cap = (uint32_t)(prod * (bound + 1));
  c.pad = 0;
  ar.twin.append(contig_bases + rStart, contig_bases + rEnd);
  for (const Variant *v : cl.vars) {
    mfx_trv_variant tv;
    tv.off = v->pos - rStart;                                  // (uint32 arithmetic, as the host's offs)
    tv.reflen = v->refLen;
    tv.na = (uint32_t)v->nalleles();
    tv.al0 = (uint32_t)ar.tal.size();
    ar.tvar.push_back(tv);
    for (size_t a = 0; a < v->nalleles(); ++a) {
      const SV &h = v->allele(a);
      mfx_trv_allele ta;
      ta.off = ar.talt.size();
      ta.len = (uint32_t)h.size();
      ta.pad = 0;
      ar.tal.push_back(ta);
      ar.talt.append(h.p, h.n);
    }
  }
  ar.t_text += c.text_cap;
  ar.t_paths += prod;
  ar.t_rows += prod * nv;
  ar.tcl.push_back(c);
  return true;
}

// A batch of clusters goes through three stages: A = its paths are enumerated and packed (host threads), B = every path k-mer
// is looked up with ONE GPU launch, C = the selectors run (host threads) and the records are written in input order.  Two
// batches are in flight: stage B of a batch runs (on its own thread: upload, kernel, download) under stage C of the batch
// before it and under the queueing of the next one.
struct VarBatch {
  std::vector<Job> jobs;
  std::string packed;
  std::vector<uint32_t> rv, av;
  // device-side scoring (`scores`): the batch's paths, one entry per path, and what came back
  std::vector<uint64_t> p_off, p_voff, p_cfirst;
  std::vector<uint32_t> p_len, p_nv, p_vidx, p_vlen, numM;
  std::vector<int32_t> p_gt;
  std::vector<double> totdk;
  std::vector<PathArena> arenas;    // one per run of RUN consecutive jobs; their capacity is kept from batch to batch
  size_t nruns = 0;
  uint64_t total = 0, npaths = 0, nvals = 0;     // of stage A: packed bytes, paths, row entries
  bool tables = false;              // stage A made the path table (p_*)
  // the clusters enumerated on the device (mfx_traverse.h): the runs' tables one behind the other, and what came back
  std::vector<mfx_trv_cluster> t_cl;
  std::vector<mfx_trv_variant> t_var;
  std::vector<mfx_trv_allele> t_al;
  std::string t_win, t_alt;
  uint64_t t_text_end = 0, t_path_cap = 0, t_row_cap = 0;
  std::vector<uint32_t> t_np, t_status, t_plen;
  std::vector<int32_t> t_gt;
  bool scored = false;              // varMer::score of this batch runs on the device (numM / totdk above)
  std::string err;                  // the error text of stage B (errors are per thread: it is carried back to the caller's)
  std::string prelude;              // (a prepared batch) what the log received while its clusters were queued
  bool live = false;
  // stage B of this batch (shared: the next batch's stage B waits for it too).  Declared LAST: a batch that is destroyed while its
  // stage B still runs (an exception unwinding the run) blocks in the future's destructor before anything the task writes
  // (err, the value arrays) is gone -- and the run's guard waits for every batch before any is destroyed.
  std::shared_future<int> gpu;
};

// What mfx_vcf_prepare leaves for the run: stage A of EVERY batch (host work that needs the VCF and the sequences but not the
// index: it runs under the index build), and the log text in the order the run would have produced it.
struct VarPrepared {
  uint32_t k = 0, comb = 0, ncontigs = 0;
  int nosplit = 0;
  bool with_tables = false;
  std::vector<uint64_t> lens;
  std::string head_log, tail_log;   // clustering's lines; what was logged after the last batch was queued
  std::deque<VarBatch> batches;
  double t_phase[3] = {0, 0, 0};    // cluster, enumerate, pack (seconds; diagnostics)
  uint64_t token = 0;               // of this prepared call set, unique in the process: what a path-only index that claimed its paths is bound to (mfx_index_claim_paths)
};

// dynamic parallel-for over [0, n) on the host threads the library may use
// Host threads that live for a whole run of the variant modes: a batch goes through five parallel loops and a config-4 call set through
// ~40 batches -- starting and joining 16 threads for each loop was 0.8 ms of its 1-3 ms (and the loops' per-thread scratch was made anew
// every time).  The run's driving thread installs its pool (t_var_pool); parallel_for on that thread hands its loop to it.
struct VarPool {
  unsigned W;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable wake, idle;
  const std::function<void()> *job = nullptr;
  uint64_t gen = 0;
  unsigned running = 0;
  bool quit = false;
  explicit VarPool(unsigned w) : W(w) {
    for (unsigned i = 0; i < W; ++i)
      th.emplace_back([this]() {
        uint64_t seen = 0;
        for (;;) {
          const std::function<void()> *f;
          {
            std::unique_lock<std::mutex> lk(m);
            wake.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            f = job;
          }
          (*f)();                                              // (never throws: parallel_for's body catches)
          std::lock_guard<std::mutex> lk(m);
          if (--running == 0) idle.notify_all();
        }
      });
  }
  void run(const std::function<void()> &f) {                   // f on every thread of the pool; returns when all are through
    {
      std::lock_guard<std::mutex> lk(m);
      job = &f;
      running = W;
      ++gen;
    }
    wake.notify_all();
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [&] { return running == 0; });
  }
  ~VarPool() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    wake.notify_all();
    for (auto &t : th) t.join();
  }
};
thread_local VarPool *t_var_pool = nullptr;

template <class F>
void parallel_for(size_t n, F &&fn) {
  VarPool *pool = t_var_pool;
  unsigned nt = std::min<size_t>(pool ? pool->W : mfx_host_threads(), n);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(64, n / (nt * 8)));
  std::atomic<size_t> next(0);
  std::exception_ptr thrown;                                   // what a worker throws (an allocation failure) is rethrown by the caller
  std::mutex thrown_mu;
  auto body = [&]() {
    try {
      for (size_t b; (b = next.fetch_add(chunk)) < n;)
        for (size_t i = b, e = std::min(n, b + chunk); i < e; ++i) fn(i);
    } catch (...) {
      std::lock_guard<std::mutex> lk(thrown_mu);
      if (!thrown) thrown = std::current_exception();
      next.store(n);                                           // the others stop at their next draw
    }
  };
  if (pool) {
    t_var_pool = nullptr;                                      // (a loop inside the loop -- there is none -- would start its own threads)
    const std::function<void()> job = body;
    pool->run(job);
    t_var_pool = pool;
  } else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(body);
    for (auto &x : th) x.join();
  }
  if (thrown) std::rethrow_exception(thrown);
}

// decimal text of an integer appended to `o` (what std::to_string gives, without the temporary)
inline void put_int(std::string &o, long long v) {
  char b[24];
  int n = 0;
  unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
  do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (v < 0) o += '-';
  while (n) o += b[--n];
}

// to_string((int)qual) of the reference (varMer.C:486,537).  A QUAL outside the int range (or NaN) makes that conversion undefined
// in C++; what the reference's binary does on x86-64 is the hardware's truncating conversion, whose answer for every such
// value is INT_MIN -- said here in defined terms.
inline int qual_as_int(double q) { return (q > -2147483649.0 && q < 2147483648.0) ? (int)q : (-2147483647 - 1); }

// the records are appended to `out` (the text of a run of clusters: one write per run)
void hom_record(std::string &out, const Cluster &cl, const int *g, size_t ng, const char *chr) {     // varMer.C:531-550
  for (size_t i = 0; i < ng; ++i) {
    int a = g[i];
    if (a <= 0) continue;
    const Variant *v = cl.vars[i];
    out += chr; out += '\t'; put_int(out, (long long)(uint32_t)(v->pos + 1)); out += "\t.\t"; out += v->allele(0); out += '\t'; out += v->allele(a); out += '\t';
    put_int(out, qual_as_int(v->qual)); out += "\tPASS\t.\tGT\t1/1\n";
  }
}

void het_record(std::string &out, const Cluster &cl, const int *g1, const int *g2, size_t ng, const char *chr) {   // varMer.C:472-529
  for (size_t i = 0; i < ng; ++i) {
    int a1 = g1[i], a2 = g2[i];
    if (a1 + a2 <= 0) continue;
    const Variant *v = cl.vars[i];
    const int q = qual_as_int(v->qual);
    out += chr; out += '\t'; put_int(out, (long long)(uint32_t)(v->pos + 1)); out += "\t.\t"; out += v->allele(0); out += '\t';
    if (a1 == a2) { out += v->allele(a1); out += '\t'; put_int(out, q); out += "\tPASS\t.\tGT\t1/1\n"; }
    else if (a1 == 0 && a2 > 0) { out += v->allele(a2); out += '\t'; put_int(out, q); out += "\tPASS\t.\tGT\t0/1\n"; }
    else if (a1 > 0 && a2 > 0) { out += v->allele(a1); out += ','; out += v->allele(a2); out += '\t'; put_int(out, q); out += "\tPASS\t.\tGT\t1/2\n"; }
    else if (a1 > 0 && a2 == 0) { out += v->allele(a1); out += '\t'; put_int(out, q); out += "\tPASS\t.\tGT\t1/0\n"; }
  }
}

double tot_dk(const std::vector<double> &d) { double s = 0; for (double x : d) s += x; return s; }   // getTotdK

// paths with the fewest missing k-mers, optionally ignoring all-missing paths (varMer.C:156-178, 406-421); `idxs`: the
// caller's scratch
void min_missing(const Job &jb, const Scored &sc, uint32_t k, bool filter_rule, uint32_t *best, std::vector<int> &idxs) {
  uint32_t numMissing = UINT32_MAX;
  idxs.clear();
  for (int i = 0; i < (int)sc.numM.size(); ++i) {
    if (sc.numM[i] == jb.ps.len(i) - k + 1) continue;                    // size_t arithmetic, as the reference
    if (filter_rule && sc.numM[i] == 0) { idxs.push_back(i); numMissing = 0; }
    if (sc.numM[i] < numMissing) { numMissing = sc.numM[i]; idxs.clear(); idxs.push_back(i); }
    else if (sc.numM[i] == numMissing) idxs.push_back(i);
  }
  *best = numMissing;
}

// the records one cluster contributes, appended to `out`
void select_records(std::string &out, const Job &jb, const Scored &sc, int mode, uint32_t k, const char *chr, std::string *log) {
  const PathSet &ps = jb.ps;
  const Cluster &cl = *jb.cl;
  const size_t nv = ps.nv;
  auto gt = [&](size_t p) { return ps.gt(p); };
  static thread_local std::vector<int> idxs;
  if (mode == MFX_VAR_FILTER) {                                          // bestFilter, varMer.C:150-199
    uint32_t best;
    min_missing(jb, sc, k, true, &best, idxs);
    if (idxs.empty()) return;
    static thread_local std::vector<int> gtIdxs;                         // the reference's list: sorted, duplicates dropped
    gtIdxs.clear();
    for (int p : idxs)
      for (int i = 0; i < (int)nv; ++i)
        if (gt(p)[i] > 0) gtIdxs.push_back(i);
    std::sort(gtIdxs.begin(), gtIdxs.end());
    gtIdxs.erase(std::unique(gtIdxs.begin(), gtIdxs.end()), gtIdxs.end());
    for (int i : gtIdxs) cl.vars[i]->rec->line_to(out);
    return;
  }
  if (mode == MFX_VAR_POLISH) {                                          // bestVariant, varMer.C:400-467
    uint32_t best;
    min_missing(jb, sc, k, false, &best, idxs);
    if (best == UINT32_MAX) return;
    if (idxs.size() == 1) { hom_record(out, cl, gt(idxs[0]), nv, chr); return; }
    // tie: order by total delta-K through the reference's own container type --
    // multimap<double,int,greater<int>> compares the keys AS INTS, descending (varMer.H:72)
    std::multimap<double, int, std::greater<int>> byDk;
    for (int p : idxs) byDk.insert(std::make_pair(sc.totdk[p], p));
    auto it = byDk.begin();
    double d1 = it->first; int p1 = it->second;
    ++it;
    double d2 = it->first; int p2 = it->second;
    if (d1 == d2) {
      if (ps.len(p1) >= ps.len(p2)) het_record(out, cl, gt(p1), gt(p2), nv, chr);
      else het_record(out, cl, gt(p2), gt(p1), nv, chr);
      return;
    }
    hom_record(out, cl, gt(p1), nv, chr);
    return;
  }
  // -better / -strict / -loose start from the reference path (varMer.C:204-395)
  if (sc.numM.empty()) return;
  const uint32_t refMissing = sc.numM[0];
  uint32_t numMissing = refMissing;
  idxs.clear();
  const bool loose = mode == MFX_VAR_LOOSE;
  for (int i = 0; i < (int)sc.numM.size(); ++i) {
    if (sc.numM[i] < numMissing) { numMissing = sc.numM[i]; idxs.clear(); idxs.push_back(i); }
    else if (sc.numM[i] == numMissing && (loose ? sc.numM[i] <= refMissing : sc.numM[i] < refMissing)) idxs.push_back(i);
  }
  if (idxs.empty()) return;
  if (idxs.size() == 1) { hom_record(out, cl, gt(idxs[0]), nv, chr); return; }
  if (!loose) {                                                          // longest path wins (first on ties)
    int idx = idxs[0];
    uint32_t longest = (uint32_t)ps.len(idx);
    for (size_t i = 1; i < idxs.size(); ++i) {
      uint32_t L = (uint32_t)ps.len(idxs[i]);
      if (L > longest) { longest = L; idx = idxs[i]; }
    }
    hom_record(out, cl, gt(idx), nv, chr);
    return;
  }
  if (idxs[0] == 0 && idxs.size() == 2) { hom_record(out, cl, gt(idxs[1]), nv, chr); return; }
  int maxVars = 0, maxIdx = idxs[0];                                     // most ALT alleles wins
  for (size_t i = 1; i < idxs.size(); ++i) {
    int cnt = 0;
    for (size_t q = 0; q < nv; ++q) if (gt(idxs[i])[q] > 0) cnt++;
    if (cnt > maxVars) { maxVars = cnt; maxIdx = idxs[i]; }
  }
  if (log) {
    *log += "[ WARNING ] :: Multiple (" + std::to_string(idxs.size()) + ") alternate pathes detected in a path beginning with variant : " + cl.vars[0]->rec->line();
    *log += "[ WARNING ] :: Max. " + std::to_string(maxVars) + " ALT variants selected\n";
  }
  hom_record(out, cl, gt(maxIdx), nv, chr);
}

// debug statistics (varMer.C:553-624)
double min_abs_k(const std::vector<double> &k) { double m = DBL_MAX; for (double x : k) { if (x < 0) continue; if (x < m) m = x; } return m == DBL_MAX ? -1 : m; }
double max_abs_k(const std::vector<double> &k) { double m = -2; for (double x : k) if (x > m) m = x; return m; }
double avg_abs_k(const std::vector<double> &k, uint32_t numM) {
  double s = 0;
  for (double x : k) if (x >= 0) s += x;
  if (k.size() == numM) return -1;
  return s / (k.size() - numM);                                          // size_t arithmetic, as the reference
}
double med_abs_k(std::vector<double> k) {
  std::sort(k.begin(), k.end());
  size_t i = 0;
  for (; i < k.size(); ++i) if (k[i] >= 0) break;
  if (i == k.size()) return -1;
  return k[i + ((k.size() - i) / 2)];
}

}  // namespace

// values(text, len, rv, av): the (readV, asmV) pair of every k-mer start of the packed path text -- one evaluator
// (mfx_dump_values) or the shards of one index (mfx_dump_values_sharded)
using PathValues = std::function<int(const char *, uint64_t, uint32_t *, uint32_t *)>;
// scores(text, len, paths, need_dk, numM, totdk): varMer::score of every path of the batch on the device (mfx_score_paths):
// what the selectors read comes back -- 12 bytes per path instead of 8 bytes per base --, the host's scoring loop does not run.
// Empty: the host scores from values() (the sharded index, -debug, tools/variants_host_bench.cpp).
// tb != nullptr: part of the batch's clusters are enumerated on the device from their tables (mfx_score_paths_trv)
using PathScores = std::function<int(const char *, uint64_t, const mfx_path_table &, const mfx_trv_batch *, int, uint32_t *, double *)>;

// (not static: tools/variants_host_bench.cpp drives the host side with a synthetic `values`, without a device)
// (prepare) the PATH-ONLY index made while the call set is prepared: on_bound is told the k-mer positions of all path text once the clusters are
// merged (it makes the table; != 0: no claims, the preparation goes on), claim runs one batch's text through mfx_claim_paths_batch -- on a thread
// of its own, under stage A of the next batch
struct PathClaims {
  std::function<int(uint64_t)> on_bound;
  std::function<int(const char *, uint64_t, const mfx_trv_batch *, uint64_t *)> claim;
  bool active = false;
  uint64_t positions = 0, bad = 0;
  int rc = MFX_OK;
  std::string err;
};
int mfx_variants_run_values(const mfx_eval *ev, const PathValues &values, const char *vcf_path, const char *const *names, const char *const *bases,
                         const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                         const char *out_path, const char *log_path, uint64_t *n_clusters, const PathScores &scores = PathScores(),
                         mfx_vcf *loaded = nullptr, uint32_t prepK = 0, PathClaims *claims = nullptr) {
  // prepK != 0: PREPARE only (mfx_vcf_prepare) -- the clusters of `loaded` are merged for k = prepK and stage A of every batch is run and
  // kept in loaded->prep; no evaluator, no output.  A later run on `loaded` starts from there.
  const bool prepare_only = prepK != 0;
  if (prepare_only ? (!loaded || !opts || (ncontigs && (!names || !bases || !lens)))
                   : (!ev || (!vcf_path && !loaded) || !opts || !out_path || (ncontigs && (!names || !bases || !lens))))
    return mfx_fail(MFX_E_INVAL, "mfx_variants_run: null argument");
  const int mode = opts->mode;
  if (mode < MFX_VAR_FILTER || mode > MFX_VAR_LOOSE) return mfx_fail(MFX_E_INVAL, "mfx_variants_run: unknown mode %d", mode);
  // the alternative paths ask for k-mers the sequence does not hold (varMer.C:76-84): a sequence-only index answers them only if it claimed
  // exactly THESE paths (mfx_index_claim_paths on the prepared call set that is run here)
  const bool path_index = !prepare_only && ev->ix->seq_only && ev->ix->paths_token != 0;
  if (!prepare_only && ev->ix->seq_only && !(path_index && loaded && loaded->prep && loaded->prep->token == ev->ix->paths_token))
    return mfx_fail(MFX_E_INVAL, path_index ? "mfx_variants_run: this path-only index claimed the paths of ANOTHER prepared call set (mfx_index_claim_paths); it answers "
                                               "mfx_variants_run_vcf on that handle only"
                                            : "mfx_variants_run: a sequence-only index holds the k-mers of one sequence; the variant modes need a full index (mfx_index_create) "
                                              "or the path-only one of their call set (mfx_vcf_prepare + mfx_index_claim_paths)");
  const uint32_t K = prepare_only ? prepK : (uint32_t)ev->ix->k;
  const uint32_t comb = opts->comb ? opts->comb : 15;
  // a prepared VCF: its stage A was run for one k / -comb / -nosplit and one set of sequences
  VarPrepared *const prep = (!prepare_only && loaded) ? loaded->prep : nullptr;
  if (prep) {
    bool same = prep->k == K && prep->comb == comb && prep->nosplit == (opts->nosplit != 0) && prep->ncontigs == ncontigs;
    for (uint32_t c = 0; c < ncontigs && same; ++c) same = prep->lens[c] == lens[c];
    if (!same) return mfx_fail(MFX_E_INVAL, "mfx_variants_run_vcf: the VCF was prepared for another k, -comb, -nosplit or set of sequences");
  }
  char *mem_log = nullptr;
  size_t mem_log_n = 0;
  FILE *log = prepare_only ? open_memstream(&mem_log, &mem_log_n) : (log_path ? fopen(log_path, "w") : stderr);
  if (!log) return mfx_fail(MFX_E_IO, "cannot open '%s'", log_path ? log_path : "(memory)");
  struct MemLog { char *&p; ~MemLog() { free(p); } } memLogGuard{mem_log};       // (destroyed after `files` below closed the stream)
  size_t mem_taken = 0;
  auto take_log = [&]() -> std::string {                                          // what the memory log received since the last call
    fflush(log);
    std::string t(mem_log + mem_taken, mem_log_n - mem_taken);
    mem_taken = mem_log_n;
    return t;
  };
  // whatever way this function is left (an exception of the host pipeline included: variants_guarded turns it into an error code),
  // the files are closed: the normal path closes them itself, checks the result and disarms the guard
  struct Files {
    FILE *log = nullptr, *out = nullptr;
    mfx_file *dbgh = nullptr;
    ~Files() {
      if (out) fclose(out);
      if (dbgh && dbgh->f) (void)mfx_close(*dbgh);
      if (log && log != stderr) fclose(log);
    }
  } files;
  files.log = log;

  // this thread drives the stages: its parallel loops run on threads that live as long as the run
  VarPool run_pool(std::max(1u, mfx_host_threads()));
  struct PoolScope { VarPool *prev; explicit PoolScope(VarPool *p) : prev(t_var_pool) { t_var_pool = p; } ~PoolScope() { t_var_pool = prev; } } poolScope(&run_pool);

  // MFX_VAR_TIMING=1: per-phase wall time on stderr (diagnostics only)
  const bool timing = getenv("MFX_VAR_TIMING") && atoi(getenv("MFX_VAR_TIMING"));
  double t_phase[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                             // load+cluster, enumerate, pack, gpu, score+select, write, queue
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int i) { double t = now(); t_phase[i] += t - t_mark; t_mark = t; };

  // the records: read and parsed here, or handed in by mfx_vcf_load (then that work ran under the caller's index build)
  VcfDB own_db;
  if (loaded && loaded->used) { return mfx_fail(MFX_E_INVAL, "mfx_variants_run_vcf: a loaded VCF serves one run (clustering rearranges it); load it again"); }
  if (prepare_only && loaded->prep) return mfx_fail(MFX_E_INVAL, "mfx_vcf_prepare: the VCF is prepared already");
  VcfDB &db = loaded ? loaded->db : own_db;
  double t_load[4] = {0, 0, 0, 0};
  int rc = MFX_OK;
  if (loaded) { loaded->used = !prepare_only; memcpy(t_load, loaded->t_load, sizeof(t_load)); }
  else rc = load_vcf(vcf_path, db, t_load);
  if (rc) return rc;
  if (prep) fwrite(prep->head_log.data(), 1, prep->head_log.size(), log);   // (the same lines, written when they were made)
  else {
    fprintf(log, "   Collected %zu header lines.\n   Loaded %zu records:\n      %-8lu unique contig%s\n      %-8u contig IDs\n   Excluded %lu invalid records\n\n",
            db.headers.size(), db.n_records, db.by_chr.size(), db.by_chr.size() == 1 ? "" : "s", (unsigned)db.contig_ids, (unsigned long)db.excluded);
    fprintf(log, "Merge variants within %u-mer bases, splitting combinations greater than %u.\n", K, comb);
    merge_clusters(db, K, comb, opts->nosplit != 0, log);
  }
  lap(0);
  VarPrepared *making = nullptr;                                           // (prepare) what this call fills
  if (prepare_only) {
    making = loaded->prep = new VarPrepared;
    making->k = K; making->comb = comb; making->nosplit = opts->nosplit != 0; making->ncontigs = ncontigs;
    making->lens.assign(lens, lens + ncontigs);
    making->with_tables = opts->debug_path == nullptr;
    making->head_log = take_log();
  }

  FILE *out = nullptr;
  mfx_file dbgh;
  FILE *dbg = nullptr;
  if (!prepare_only) {
    out = fopen(out_path, "w");
    if (!out) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", out_path);
    files.out = out;
    for (auto &h : db.headers) fprintf(out, "%s\n", h.c_str());            // merfin-variants.C:332-333
    if (opts->debug_path) {
      dbgh = mfx_open_writer(opts->debug_path, false);                     // compressedFileWriter, merfin-variants.C:149
      dbg = dbgh.f;
      files.dbgh = &dbgh;
    }
  }

  mfx_kparams kp{0.0, 0, nullptr, nullptr};
  // readK and prob depend on the read count alone (merfin-globals.C:80-97): evaluated once for the counts that occur all
  // the time, by the same routine the scoring loop would call (identical doubles)
  constexpr uint32_t KLUT = 4096;
  std::vector<double> lutK, lutP;
  if (!prepare_only) {
    kp = mfx_kparams{ev->peak, ev->n_prob, ev->probK.data(), ev->probP.data()};
    lutK.resize(KLUT); lutP.resize(KLUT);
    for (uint32_t v = 0; v < KLUT; ++v) { double a; mfx_getK(&kp, v, 0, &lutK[v], &a, &lutP[v]); }
  }
  uint64_t clusters = 0, varMerId = 0;
  const uint64_t BATCH_BYTES = (getenv("MFX_VAR_BATCH_MB") ? (uint64_t)atoi(getenv("MFX_VAR_BATCH_MB")) : 64ull) << 20;   // packed path text per GPU launch

  // the batches (VarBatch): two that take turns, or -- preparing / prepared -- one per batch of the whole call set
  std::deque<VarBatch> own_batches;
  std::deque<VarBatch> &batches = making ? making->batches : prep ? prep->batches : own_batches;
  if (!prep) { batches.resize(making ? 1 : 2); for (VarBatch &b : batches) b.jobs.reserve(65536); }
  struct StageBGuard { std::deque<VarBatch> &b; ~StageBGuard() { for (VarBatch &x : b) if (x.gpu.valid()) x.gpu.wait(); } } stageBGuard{batches};
  using Batch = VarBatch;
  size_t cur_b = 0;
  std::vector<char> out_buf(4u << 20);
  if (out) setvbuf(out, out_buf.data(), _IOFBF, out_buf.size());
  // The records of a batch (a few hundred MB for a human call set) are written by their own thread, batch by batch in order,
  // under the stages of the next batches: config 4 spent 0.6 s of its 2.5 s inside fwrite on the thread that drives the stages.
  struct Writer {
    FILE *f;
    std::mutex m;
    std::condition_variable cv;
    std::list<std::vector<std::string>> q;
    bool done = false, failed = false;
    std::thread th;
    explicit Writer(FILE *file) : f(file) {
      th = std::thread([this]() {
        for (;;) {
          std::vector<std::string> b;
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return done || !q.empty(); });
            if (q.empty()) return;
            b = std::move(q.front());
            q.pop_front();
          }
          for (const std::string &x : b) if (!x.empty() && fwrite(x.data(), 1, x.size(), f) != x.size()) failed = true;
        }
      });
    }
    void push(std::vector<std::string> &&b) { { std::lock_guard<std::mutex> lk(m); q.push_back(std::move(b)); } cv.notify_one(); }
    bool finish() { { std::lock_guard<std::mutex> lk(m); done = true; } cv.notify_one(); if (th.joinable()) th.join(); return !failed; }
    ~Writer() { finish(); }
  } writer(out);

  // stage A + the launch of stage B
  // the jobs of a batch are worked on in runs of RUN consecutive ones: a run's paths share one arena (stage A), and its
  // records / -debug lines / log lines are concatenated by the worker that scores it (stage C), so that the writer issues
  // one write per run, not per cluster
  constexpr size_t RUN = 256;
  // MFX_VAR_DEVICE_TRAVERSE=0: every cluster is enumerated on the host (A/B, tests)
  const bool trv_on = !(getenv("MFX_VAR_DEVICE_TRAVERSE") && atoi(getenv("MFX_VAR_DEVICE_TRAVERSE")) == 0);
  const bool trv_check = getenv("MFX_VAR_TRAVERSE_CHECK") && atoi(getenv("MFX_VAR_TRAVERSE_CHECK"));
  std::atomic<uint64_t> trv_checked{0};
  auto stage_a = [&](Batch &bt, bool want_tables, bool host_only = false) {
    const bool use_trv = trv_on && want_tables && !host_only;
    std::vector<Job> &jobs = bt.jobs;
    std::string &packed = bt.packed;
    lap(6);                                                              // (the clusters were queued since the last lap)
    const size_t nruns = (jobs.size() + RUN - 1) / RUN;
    bt.nruns = nruns;
    if (bt.arenas.size() < nruns) bt.arenas.resize(nruns);
    parallel_for(nruns, [&](size_t ri) {
      PathArena &ar = bt.arenas[ri];
      ar.reset();
      // per-thread scratch, reused from cluster to cluster
      static thread_local std::vector<uint32_t> offs, vl;
      static thread_local std::vector<int> path;
      static thread_local std::vector<std::string> reps;                 // candidate strings per recursion depth
      static thread_local std::string window;
      static thread_local SeenMap seen;
      for (size_t i = ri * RUN, e = std::min(jobs.size(), (ri + 1) * RUN); i < e; ++i) {
        Job &jb = jobs[i];
        offs.clear(); vl.clear(); path.clear();
        for (const Variant *v : jb.cl->vars) { offs.push_back(v->pos - jb.rStart); vl.push_back(v->refLen); }
        jb.ps = PathSet();
        jb.ps.ar = &ar;
        jb.ps.p0 = ar.toff.size() - 1;
        jb.ps.v0 = ar.gt.size();
        jb.ps.nv = (uint32_t)jb.cl->vars.size();
        jb.trv = -1;
        if (use_trv && trv_pack(*jb.cl, jb.rStart, jb.rEnd, bases[jb.contig], ar)) {      // (its paths are made on the device)
          jb.trv = (int64_t)ar.tcl.size() - 1;
          if (trv_check) {
            // MFX_VAR_TRAVERSE_CHECK=1 (tests): the shared traverse, run here on the host, against the recursion -- every path's bases,
            // genotype, offset and length rows
            PathArena tmp;
            tmp.reset();
            PathSet ps;
            ps.ar = &tmp; ps.p0 = 0; ps.v0 = 0; ps.nv = (uint32_t)jb.cl->vars.size();
            if (reps.size() < jb.cl->vars.size() + 1) reps.resize(jb.cl->vars.size() + 1);
            window.assign(bases[jb.contig] + jb.rStart, bases[jb.contig] + jb.rEnd);
            if (!seen.empty()) seen.clear();
            enumerate(0, offs, vl, *jb.cl, window, path, ps, reps, 0, seen);
            mfx_trv_cluster c = ar.tcl.back();
            c.text0 = 0; c.path0 = 0; c.row0 = 0;
            std::vector<char> T((size_t)c.text_cap + 1, '\n');
            std::vector<uint64_t> o_off(c.path_cap), o_voff(c.path_cap), o_cf(c.path_cap);
            std::vector<uint32_t> o_len(c.path_cap), o_nv(c.path_cap), o_vidx((size_t)c.path_cap * c.nv), o_vlen((size_t)c.path_cap * c.nv);
            std::vector<int32_t> o_gt((size_t)c.path_cap * c.nv);
            mfx_trv_out o;
            o.text = T.data(); o.p_off = o_off.data(); o.p_voff = o_voff.data(); o.p_cfirst = o_cf.data(); o.p_len = o_len.data(); o.p_nv = o_nv.data();
            o.gt = o_gt.data(); o.vidx = o_vidx.data(); o.vlen = o_vlen.data(); o.table_base = 0; o.row_base = 0;
            uint32_t np2 = 0;
            const uint32_t st = mfx_traverse_cluster(c, ar.tvar.data(), ar.tal.data(), ar.twin.data(), ar.talt.data(), o, &np2);
            bool same = st == MFX_TRV_OK && np2 == ps.np;
            for (uint32_t q = 0; q < np2 && same; ++q) {
              same = o_len[q] == ps.len(q) && memcmp(T.data() + o_off[q], ps.seq(q), ps.len(q)) == 0 && T[o_off[q] + o_len[q]] == '\n';
              for (uint32_t i = 0; i < c.nv && same; ++i)
                same = o_gt[(size_t)q * c.nv + i] == ps.gt(q)[i] && o_vidx[(size_t)q * c.nv + i] == ps.vidx(q)[i] && o_vlen[(size_t)q * c.nv + i] == ps.vlen(q)[i];
            }
            if (!same) throw std::runtime_error("the shared traverse and the host's recursion disagree on a cluster (status " + std::to_string(st) + ", paths " +
                                                std::to_string(np2) + " / " + std::to_string(ps.np) + ") at " + std::string(names[jb.contig]) + ":" + std::to_string(jb.rStart));
            trv_checked.fetch_add(1, std::memory_order_relaxed);
            offs.clear(); vl.clear(); path.clear();
          }
          continue;
        }
        if (reps.size() < jb.cl->vars.size() + 1) reps.resize(jb.cl->vars.size() + 1);
        window.assign(bases[jb.contig] + jb.rStart, bases[jb.contig] + jb.rEnd);
        if (!seen.empty()) seen.clear();
        enumerate(0, offs, vl, *jb.cl, window, path, jb.ps, reps, 0, seen);
      }
      if (!seen.empty()) seen.clear();
    });
    lap(1);
    uint64_t total = 0, np = 0, nvals = 0;
    for (size_t ri = 0; ri < nruns; ++ri) {
      PathArena &ar = bt.arenas[ri];
      ar.first_id = varMerId;
      varMerId += ar.toff.size() - 1;
      ar.off = total;
      total += ar.text.size();                                           // every path is followed by its '\n' (not ACGT: k-mers never span two paths)
      ar.p0 = np;
      np += ar.toff.size() - 1;
      ar.v0 = nvals;
      nvals += ar.gt.size();
    }
    packed.resize(total);
    parallel_for(nruns, [&](size_t ri) {
      const PathArena &ar = bt.arenas[ri];
      if (!ar.text.empty()) memcpy(&packed[ar.off], ar.text.data(), ar.text.size());
    });
    // the clusters that go to the device's traverse: the runs' tables one behind the other; their paths' text behind the host's, their
    // path slots and rows numbered from 0 (mfx_score_paths_trv puts them behind the host's)
    {
      uint64_t ncl = 0, nvar = 0, nal = 0, nwin = 0, nalt = 0, ntext = 0, npath = 0, nrow = 0;
      for (size_t ri = 0; ri < nruns; ++ri) {
        PathArena &ar = bt.arenas[ri];
        ar.b_cl = ncl; ncl += ar.tcl.size();
        ar.b_var = nvar; nvar += ar.tvar.size();
        ar.b_al = nal; nal += ar.tal.size();
        ar.b_win = nwin; nwin += ar.twin.size();
        ar.b_alt = nalt; nalt += ar.talt.size();
        ar.b_text = ntext; ntext += ar.t_text;
        ar.b_path = npath; npath += ar.t_paths;
        ar.b_row = nrow; nrow += ar.t_rows;
      }
      bt.t_cl.resize(ncl); bt.t_var.resize(nvar); bt.t_al.resize(nal); bt.t_win.resize(nwin); bt.t_alt.resize(nalt);
      bt.t_text_end = total + ntext; bt.t_path_cap = npath; bt.t_row_cap = nrow;
      bt.t_np.assign(ncl, 0); bt.t_status.assign(ncl, 0);
      if (!making) { bt.t_plen.resize(npath); bt.t_gt.resize(nrow); }   // (what comes BACK from stage B: a prepared batch takes the room when it gets there)
      if (ncl)
        parallel_for(nruns, [&](size_t ri) {
          const PathArena &ar = bt.arenas[ri];
          for (size_t i = 0; i < ar.tcl.size(); ++i) {
            mfx_trv_cluster c = ar.tcl[i];
            c.win_off += ar.b_win; c.var0 += (uint32_t)ar.b_var; c.text0 += total + ar.b_text; c.path0 += ar.b_path; c.row0 += ar.b_row;
            bt.t_cl[ar.b_cl + i] = c;
          }
          for (size_t i = 0; i < ar.tvar.size(); ++i) { mfx_trv_variant v = ar.tvar[i]; v.al0 += (uint32_t)ar.b_al; bt.t_var[ar.b_var + i] = v; }
          for (size_t i = 0; i < ar.tal.size(); ++i) { mfx_trv_allele a = ar.tal[i]; a.off += ar.b_alt; bt.t_al[ar.b_al + i] = a; }
          if (!ar.twin.empty()) memcpy(&bt.t_win[ar.b_win], ar.twin.data(), ar.twin.size());
          if (!ar.talt.empty()) memcpy(&bt.t_alt[ar.b_alt], ar.talt.data(), ar.talt.size());
        });
    }
    lap(2);
    bt.total = total; bt.npaths = np; bt.nvals = nvals;
    bt.tables = false;
    if ((total || !bt.t_cl.empty()) && want_tables) {
      // the path table of the batch: one entry per path, the variants' rows concatenated
      bt.tables = true;
      bt.p_off.resize(np); bt.p_voff.resize(np); bt.p_cfirst.resize(np); bt.p_len.resize(np); bt.p_nv.resize(np);
      bt.p_gt.resize(nvals); bt.p_vidx.resize(nvals); bt.p_vlen.resize(nvals);
      if (!making) { bt.numM.resize(np + bt.t_path_cap); bt.totdk.resize(np + bt.t_path_cap); }
      parallel_for(nruns, [&](size_t ri) {
        const PathArena &ar = bt.arenas[ri];
        for (size_t i = ri * RUN, e = std::min(jobs.size(), (ri + 1) * RUN); i < e; ++i) {
          const PathSet &ps = jobs[i].ps;
          const uint64_t q0 = ps.table_p0();
          for (size_t p = 0; p < ps.np; ++p) {
            const uint64_t q = q0 + p;
            bt.p_off[q] = ps.packed_off(p);
            bt.p_len[q] = (uint32_t)ps.len(p);
            bt.p_nv[q] = ps.nv;
            bt.p_voff[q] = ar.v0 + ps.v0 + p * ps.nv;
            bt.p_cfirst[q] = q0;
          }
        }
        if (!ar.gt.empty()) {
          memcpy(&bt.p_gt[ar.v0], ar.gt.data(), ar.gt.size() * sizeof(int32_t));
          memcpy(&bt.p_vidx[ar.v0], ar.vidx.data(), ar.vidx.size() * sizeof(uint32_t));
          memcpy(&bt.p_vlen[ar.v0], ar.vlen.data(), ar.vlen.size() * sizeof(uint32_t));
        }
      });
      lap(2);
    }
  };
  // the launch of stage B; prevb: the batch whose stage B runs before this one's (nullptr: none)
  auto stage_b = [&](Batch &bt, Batch *prevb) -> int {
    // (a batch prepared for the device's traverse but scored on the host after all -- -debug, MFX_VAR_HOST_SCORE --: enumerated on the host now)
    if (!((bool)scores && dbg == nullptr && bt.tables) && !bt.t_cl.empty()) stage_a(bt, false, true);
    const std::string &packed = bt.packed;
    const uint64_t total = bt.total, nvals = bt.nvals;
    bt.live = true;
    bt.scored = false;
    bt.gpu = std::shared_future<int>();                                  // (a shared future stays valid after get(): this batch has none yet)
    const bool on_device = (bool)scores && dbg == nullptr && bt.tables;  // -debug wants the per-position values: scored on the host
    if ((total || !bt.t_cl.empty()) && on_device) {
      bt.scored = true;
      if (bt.numM.size() != bt.p_off.size() + bt.t_path_cap) {           // (a prepared batch: see stage A)
        bt.t_plen.resize(bt.t_path_cap); bt.t_gt.resize(bt.t_row_cap);
        bt.numM.resize(bt.p_off.size() + bt.t_path_cap); bt.totdk.resize(bt.p_off.size() + bt.t_path_cap);
      }
      Batch *bp = &bt;
      const int need_dk = mode == MFX_VAR_POLISH ? 1 : 0;
      std::shared_future<int> prev = prevb ? prevb->gpu : std::shared_future<int>();
      bt.gpu = std::async(std::launch::async, [bp, prev, need_dk, nvals, &scores]() mutable {
        if (prev.valid()) prev.wait();                                   // one stage B at a time on the evaluator (two at once were measured: their
                                                                         // allocations and frees stall each other -- config 4: 0.68 -> 0.84 s)
        prev = std::shared_future<int>();                                // (let go of the earlier batch's state: no chain of all batches so far)
        mfx_path_table pt;
        pt.npaths = bp->p_off.size(); pt.nvals = nvals;
        pt.off = bp->p_off.data(); pt.len = bp->p_len.data(); pt.nv = bp->p_nv.data(); pt.voff = bp->p_voff.data(); pt.cfirst = bp->p_cfirst.data();
        pt.gt = bp->p_gt.data(); pt.vidx = bp->p_vidx.data(); pt.vlen = bp->p_vlen.data();
        mfx_trv_batch tb;
        if (!bp->t_cl.empty()) {
          tb.ncl = bp->t_cl.size(); tb.nvar = bp->t_var.size(); tb.nal = bp->t_al.size(); tb.win_bytes = bp->t_win.size(); tb.al_bytes = bp->t_alt.size();
          tb.cl = bp->t_cl.data(); tb.var = bp->t_var.data(); tb.al = bp->t_al.data(); tb.win_text = bp->t_win.data(); tb.al_text = bp->t_alt.data();
          tb.text_end = bp->t_text_end; tb.path_cap = bp->t_path_cap; tb.row_cap = bp->t_row_cap;
          tb.np = bp->t_np.data(); tb.status = bp->t_status.data(); tb.p_len = bp->t_plen.data(); tb.gt = bp->t_gt.data();
        }
        const int r = scores(bp->packed.data(), bp->packed.size(), pt, bp->t_cl.empty() ? nullptr : &tb, need_dk, bp->numM.data(), bp->totdk.data());
        if (r) bp->err = mfx_last_error();
        return r;
      });
    } else if (total) {
      bt.rv.resize(packed.size() + 1);
      bt.av.resize(packed.size() + 1);
      Batch *bp = &bt;
      // one stage B at a time on the evaluator(s): it still runs under stages A and C of its neighbours on the host, but two
      // values() calls never hold their device buffers (path text + two value arrays each) at once, and the sharded form
      // (mfx_dump_values_sharded: per-slot scratch, peer copies) is never entered twice
      std::shared_future<int> prev = prevb ? prevb->gpu : std::shared_future<int>();
      bt.gpu = std::async(std::launch::async, [bp, prev, &values]() mutable {
        if (prev.valid()) prev.wait();
        prev = std::shared_future<int>();
        const int r = values(bp->packed.data(), bp->packed.size(), bp->rv.data(), bp->av.data());
        if (r) bp->err = mfx_last_error();
        return r;
      });
    }
    return MFX_OK;
  };

  // the end of stage B (its values are waited for) and stage C
  auto stage_c = [&](Batch &bt) -> int {
    if (!bt.live) return MFX_OK;
    std::vector<Job> &jobs = bt.jobs;
    std::string &packed = bt.packed;
    const std::vector<uint32_t> &rv = bt.rv, &av = bt.av;
    bt.live = false;
    lap(7);
    if (bt.gpu.valid()) {
      const int r = bt.gpu.get();
      if (r) { jobs.clear(); packed.clear(); return mfx_fail(r, "%s", bt.err.c_str()); }
    }
    if (bt.scored && !bt.t_cl.empty()) {
      // a cluster the device could not enumerate (a replacement past the end of its string -- the host's std::string throws there --,
      // or more than the room reserved): the whole batch is enumerated on the host, as without the device's traverse
      bool redo = false;
      for (uint32_t st : bt.t_status) redo = redo || st != MFX_TRV_OK;
      if (redo && path_index)                                           // (mfx_index_claim_paths refuses such a call set: never reached)
        return mfx_fail(MFX_E_INVAL, "mfx_variants_run: a cluster must be enumerated on the host, whose paths this path-only index did not claim");
      if (redo) {
        stage_a(bt, true, true);
        int r = stage_b(bt, nullptr);
        if (r == MFX_OK && bt.gpu.valid()) r = bt.gpu.get();
        if (r) { jobs.clear(); packed.clear(); return mfx_fail(r, "%s", bt.err.c_str()); }
        bt.live = false;
      }
    }
    lap(3);
    const bool want_dbg = dbg != nullptr;
    const size_t nruns = bt.nruns;                                       // the runs of stage A (one arena each)
    std::vector<std::string> run_out(nruns), run_dbg(nruns), run_log(nruns);
    parallel_for(nruns, [&](size_t ri) {
     static thread_local Scored sc;                                      // per-thread scratch, reused from cluster to cluster
     run_out[ri].reserve(RUN * 96);
     for (size_t ji = ri * RUN, je = std::min(jobs.size(), (ri + 1) * RUN); ji < je; ++ji) {
      Job &jb = jobs[ji];
      // `prob` is a local of varMer::score (one per cluster) that the reference reads
      // uninitialised until the first valid k-mer writes it; before that it only
      // multiplies |0-0|, so any finite start value is equivalent.  We fix 1.0.
      double prob = 1.0;
      if (jb.trv >= 0) {                                                 // enumerated on the device: what came back of its paths
        const mfx_trv_cluster &C = bt.t_cl[jb.ps.ar->b_cl + (uint64_t)jb.trv];
        jb.ps.np = bt.t_np[jb.ps.ar->b_cl + (uint64_t)jb.trv];
        jb.ps.dlen = bt.t_plen.data() + C.path0;
        jb.ps.dgt = bt.t_gt.data() + C.row0;
        jb.ps.dq0 = bt.npaths + C.path0;
      }
      const size_t np = jb.ps.size(), nv = jb.ps.nv;
      // what the selectors read: numM always; the paths' total delta-K only in -polish (its tie-break); the per-position
      // K* and delta-K values only in the -debug statistics.  Nothing else is computed or stored.
      const bool need_dk = mode == MFX_VAR_POLISH || want_dbg, keep = want_dbg && mode != MFX_VAR_FILTER;
      sc.numM.resize(np); sc.totdk.assign(np, 0.0);
      if (want_dbg) { sc.ks.assign(np, std::vector<double>()); sc.dks.assign(np, std::vector<double>()); }
      const bool scored = bt.scored;                                     // varMer::score ran on the device (mfx_score_paths)
      if (scored) {
        const uint64_t q0 = jb.ps.table_p0();
        for (size_t p = 0; p < np; ++p) { sc.numM[p] = bt.numM[q0 + p]; if (need_dk) sc.totdk[p] = bt.totdk[q0 + p]; }
      }
      for (size_t p = 0; p < np && !scored; ++p) {                       // varMer::score, varMer.C:66-144
        const char *s = jb.ps.seq(p);
        const uint32_t slen = (uint32_t)jb.ps.len(p);
        const uint64_t o = jb.ps.packed_off(p);
        const int *gtp = jb.ps.gt(p);
        const uint32_t *vip = jb.ps.vidx(p), *vlp = jb.ps.vlen(p);
        uint32_t numM = 0, run = 0;
        double totdk = 0.0;                                              // summed in position order, as getTotdK does
        if (keep) { sc.ks[p].reserve(slen); sc.dks[p].reserve(slen); }
        for (uint32_t idx = 0; idx < slen; ++idx) {
          run = base_ok((unsigned char)s[idx]) ? run + 1 : 0;
          double readK = 0, asmK = 0;
          if (run >= K) {                                                // k-mer ENDING at idx starts at idx-k+1
            const uint64_t sp = o + idx - (K - 1);
            const uint32_t v = rv[sp];
            if (v < KLUT) { readK = lutK[v]; prob = lutP[v]; asmK = (double)av[sp]; }
            else mfx_getK(&kp, v, av[sp], &readK, &asmK, &prob);
          }
          if (readK == 0) numM++;
          if (mode == MFX_VAR_FILTER || !need_dk) continue;              // :93-96 (and: nobody reads the values below)
          const double oD = fabs(readK - asmK) * prob;                   // :99
          for (size_t j = 0; j < nv; ++j) {                              // :103-112, uint32 wrap included
            const uint32_t vi = vip[j], vl = vlp[j];
            if (gtp[j] > 0 && vi + 1 - K <= idx && idx < vi + vl + K) { asmK++; break; }
          }
          const double nD = fabs(readK - asmK) * prob;                   // :126
          totdk += oD - nD;
          if (keep) {
            double kM;
            if (readK == 0) kM = -1;                                     // :116-124
            else if (readK > asmK) kM = readK / asmK - 1;
            else kM = asmK / readK - 1;
            sc.ks[p].push_back(kM);
            sc.dks[p].push_back(oD - nD);
          }
        }
        sc.numM[p] = numM;
        sc.totdk[p] = totdk;
      }
      const char *chr = names[jb.contig];
      if (want_dbg) {                                                    // merfin-variants.C:240-276
        char buf[512];
        for (size_t p = 0; p < np; ++p) {
          snprintf(buf, sizeof(buf), "%lu\t%s:%u-%u\t", (unsigned long)(jb.ps.first_id() + p), chr, jb.rStart, jb.rEnd);
          jb.dbg += buf;
          jb.dbg.append(jb.ps.seq(p), jb.ps.len(p));
          snprintf(buf, sizeof(buf), "\t%u\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\t", sc.numM[p], min_abs_k(sc.ks[p]), max_abs_k(sc.ks[p]),
                   med_abs_k(sc.ks[p]), avg_abs_k(sc.ks[p], sc.numM[p]), sc.totdk[p]);
          jb.dbg += buf;
          for (size_t i = 0; i < nv; ++i) {
            int a = jb.ps.gt(p)[i];
            if (a > 0)
            {
              jb.dbg += chr; jb.dbg += ' '; jb.dbg += std::to_string(jb.cl->vars[i]->pos + 1); jb.dbg += " . "; jb.dbg += jb.cl->vars[i]->allele(0);
              jb.dbg += ' '; jb.dbg += jb.cl->vars[i]->allele(a); jb.dbg += " . PASS . GT 1/1  ";
            }
          }
          jb.dbg += "\n";
        }
      }
      select_records(run_out[ri], jb, sc, mode, K, chr, &run_log[ri]);
      if (want_dbg) { run_dbg[ri] += jb.dbg; std::string().swap(jb.dbg); }
     }
    });
    lap(4);
    for (size_t ri = 0; ri < nruns; ++ri) {
      if (!run_log[ri].empty()) fwrite(run_log[ri].data(), 1, run_log[ri].size(), log);
      if (dbg && !run_dbg[ri].empty()) fwrite(run_dbg[ri].data(), 1, run_dbg[ri].size(), dbg);
    }
    lap(8);
    writer.push(std::move(run_out));                                     // the records, in order, by the writer thread
    lap(9);
    clusters += jobs.size();
    jobs.clear();
    lap(10);
    packed.clear();
    lap(5);
    return MFX_OK;
  };

  // the queued clusters start their stages A and B; the batch before them finishes (its stage C).  Preparing: stage A only, and the
  // next batch gets a structure of its own.
  auto flush = [&]() -> int {
    if (batches[cur_b].jobs.empty()) return MFX_OK;
    if (making) {
      batches[cur_b].prelude = take_log();
      stage_a(batches[cur_b], making->with_tables);
      if (claims && claims->active) {
        // the batch's k-mers are claimed on the device while the host prepares the next batch; one claim at a time (they share the scratch)
        Batch *bp = &batches[cur_b];
        std::shared_future<int> prev = cur_b ? batches[cur_b - 1].gpu : std::shared_future<int>();
        bp->gpu = std::async(std::launch::async, [bp, prev, claims]() mutable {
          if (prev.valid() && prev.get() != MFX_OK) return (int)MFX_E_INVAL;       // (an earlier batch failed: its error is the one reported)
          prev = std::shared_future<int>();
          mfx_trv_batch tb;
          if (!bp->t_cl.empty()) {
            tb.ncl = bp->t_cl.size(); tb.nvar = bp->t_var.size(); tb.nal = bp->t_al.size(); tb.win_bytes = bp->t_win.size(); tb.al_bytes = bp->t_alt.size();
            tb.cl = bp->t_cl.data(); tb.var = bp->t_var.data(); tb.al = bp->t_al.data(); tb.win_text = bp->t_win.data(); tb.al_text = bp->t_alt.data();
            tb.text_end = bp->t_text_end; tb.path_cap = bp->t_path_cap; tb.row_cap = bp->t_row_cap;
            tb.np = bp->t_np.data(); tb.status = bp->t_status.data();
          }
          uint64_t bad = 0;
          const int r = claims->claim(bp->packed.data(), bp->packed.size(), bp->t_cl.empty() ? nullptr : &tb, &bad);
          if (r) bp->err = mfx_last_error();
          else if (bad) { bp->err = std::to_string(bad) + " clusters of the call set cannot be enumerated on the device (their paths are made on the host during the run)"; return (int)MFX_E_INVAL; }
          return r;
        });
      }
      batches.emplace_back();
      cur_b = batches.size() - 1;
      batches[cur_b].jobs.reserve(65536);
      return MFX_OK;
    }
    stage_a(batches[cur_b], (bool)scores && dbg == nullptr);
    int r = stage_b(batches[cur_b], &batches[cur_b ^ 1]);
    const int r2 = stage_c(batches[cur_b ^ 1]);
    cur_b ^= 1;
    return r ? r : r2;
  };

  if (prep) {
    // stage A was run ahead (mfx_vcf_prepare): every batch's stage B is launched behind the one before it, under that one's stage C
    // -- the log receives what it would have, in the same order
    t_mark = now();
    for (size_t i = 0; i < batches.size() && rc == MFX_OK; ++i) {
      if (batches[i].jobs.empty()) continue;                            // (the structure made for a batch that never came)
      fwrite(batches[i].prelude.data(), 1, batches[i].prelude.size(), log);
      rc = stage_b(batches[i], i ? &batches[i - 1] : nullptr);
      if (i) { const int r2 = stage_c(batches[i - 1]); if (rc == MFX_OK) rc = r2; }
      cur_b = i;
    }
    if (rc == MFX_OK) fwrite(prep->tail_log.data(), 1, prep->tail_log.size(), log);
    const int r2 = stage_c(batches[cur_b]);
    if (rc == MFX_OK) rc = r2;
    for (Batch &bt : batches) if (bt.gpu.valid()) (void)bt.gpu.get();
    t_phase[0] += prep->t_phase[0]; t_phase[1] += prep->t_phase[1]; t_phase[2] += prep->t_phase[2];
  } else {
  // upper bound of a cluster's path text: (product of allele counts) x (window + longest alleles)
  auto cluster_text_bound = [](const Cluster *cl, uint32_t rStart, uint32_t rEnd) -> uint64_t {
    double npaths = 1;
    uint64_t plen = (uint64_t)(rEnd - rStart) + 1;
    for (const Variant *v : cl->vars) {
      npaths *= (double)std::max<size_t>(v->nalleles(), 1);
      size_t longest = 0;
      for (size_t ai = 0; ai < v->nalleles(); ++ai) longest = std::max<size_t>(longest, v->allele(ai).size());
      plen += longest;
    }
    return (uint64_t)std::min(npaths, 4194304.0) * plen;
  };
  if (making && claims && claims->on_bound) {
    // the path-only index is made HERE, before the first batch: the k-mer positions of all path text, from the clusters alone (the loop below
    // without its log lines)
    uint64_t bound = 0;
    for (uint32_t c = 0; c < ncontigs; ++c) {
      auto it = db.by_chr.find(names[c]);
      if (it == db.by_chr.end()) continue;
      const uint64_t seqLen = lens[c];
      for (const Cluster *cl : it->second) {
        uint32_t rStart = cl->rStart, rEnd = cl->rEnd;
        const uint32_t pad = K - 1;
        rStart = rStart > pad ? rStart - pad : 0;
        if (rEnd < seqLen - pad) rEnd += pad; else rEnd = (uint32_t)seqLen;
        if (!(rStart <= rEnd && (uint64_t)rEnd <= seqLen) || cl->vars.size() > comb) continue;
        bound += cluster_text_bound(cl, rStart, rEnd);
      }
    }
    claims->positions = bound;
    const int brc = claims->on_bound(bound);
    claims->active = brc == MFX_OK;
    if (brc) { claims->rc = brc; claims->err = mfx_last_error(); }
    lap(0);
  }
  uint64_t est_bytes = 0;
  for (uint32_t c = 0; c < ncontigs && rc == MFX_OK; ++c) {
    auto it = db.by_chr.find(names[c]);
    if (it == db.by_chr.end()) continue;                                 // merfin-variants.C:141-142
    fprintf(log, "Processing sequence %s for variants\n", names[c]);
    const uint64_t seqLen = lens[c];
    for (Cluster *cl : it->second) {
      uint32_t rStart = cl->rStart, rEnd = cl->rEnd;
      const uint32_t pad = K - 1;
      rStart = rStart > pad ? rStart - pad : 0;                          // :172-173
      if (rEnd < seqLen - pad) rEnd += pad; else rEnd = (uint32_t)seqLen;   // :175-176, uint64 arithmetic as the reference
      if (!(rStart <= rEnd && (uint64_t)rEnd <= seqLen)) {               // dnaSeq::copy out of range, :208-211
        fprintf(log, "PANIC : Invalid region specified: %s : %u - %u\n", names[c], rStart, rEnd);
        continue;
      }
      if (cl->vars.size() > comb) {                                      // :213-217
        fprintf(log, "PANIC : Combination %s:%u-%u has too many variants ( found %lu > %u ) to evaluate. Consider filtering the vcf upfront. Skipping...\n",
                names[c], rStart, rEnd, cl->vars.size(), comb);
        continue;
      }
      std::vector<Job> &jobs = batches[cur_b].jobs;
      jobs.emplace_back();
      Job &jb = jobs.back();
      jb.cl = cl; jb.contig = c; jb.rStart = rStart; jb.rEnd = rEnd;
      est_bytes += cluster_text_bound(cl, rStart, rEnd);
      if (est_bytes >= BATCH_BYTES || jobs.size() >= 65536) { rc = flush(); est_bytes = 0; }
      if (rc) break;
    }
  }
  if (rc == MFX_OK) rc = flush();
  if (making && claims && claims->active) {
    for (Batch &bt : batches) {
      if (!bt.gpu.valid()) continue;
      const int r = bt.gpu.get();
      if (r && claims->rc == MFX_OK && !bt.err.empty()) { claims->rc = r; claims->err = bt.err; }
      bt.gpu = std::shared_future<int>();
    }
  }
  if (making) {
    making->tail_log = take_log();
    {
      static std::atomic<uint64_t> prepared_sets{0};
      uint64_t h = 0xcbf29ce484222325ULL;
      auto mix = [&](uint64_t x) { h ^= x; h *= 0x100000001b3ULL; };
      mix(K); mix(comb); mix(making->nosplit);
      for (const VarBatch &b : batches) { mix(b.jobs.size()); mix(b.packed.size()); mix(b.t_text_end); mix(b.t_path_cap); }
      making->token = ((prepared_sets.fetch_add(1) + 1) << 32) | (h & 0xffffffffu);
    }
    making->t_phase[0] = t_phase[0]; making->t_phase[1] = t_phase[1]; making->t_phase[2] = t_phase[2];
    if (timing) fprintf(stderr, "[mfx_variants] prepared ahead: cluster %.2fs  enumerate %.2fs  pack %.2fs  (%zu batches)\n", t_phase[0], t_phase[1], t_phase[2],
                        batches.size() - 1);
    files.log = nullptr;
    fclose(log);
    if (n_clusters) *n_clusters = 0;
    return rc;
  }
  {
    const int r2 = stage_c(batches[cur_b ^ 1]);              // the last batch launched (after an error: its GPU work is waited for, nothing is written twice)
    if (rc == MFX_OK) rc = r2;
    for (Batch &bt : batches) if (bt.gpu.valid()) (void)bt.gpu.get();
  }
  }
  lap(5);
  if (trv_check) fprintf(stderr, "[mfx_variants] traverse check: %lu clusters enumerated both ways, all equal\n", (unsigned long)trv_checked.load());
  if (timing)
    fprintf(stderr, "[mfx_variants] load+cluster %.2fs  enumerate %.2fs  pack %.2fs  gpu %.2fs  score+select %.2fs  queue %.2fs  write %.2fs\n",
            t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[6], t_phase[5]);
  if (timing) fprintf(stderr, "[mfx_variants]   load: read %.3f  lines %.3f  parse %.3f  runs %.3f  (the rest: clustering) | stage C: before %.3f  logs %.3f  hand-over %.3f  clear %.3f\n",
                      t_load[0], t_load[1], t_load[2], t_load[3], t_phase[7], t_phase[8], t_phase[9], t_phase[10]);
  if (!writer.finish() && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed", out_path);
  files.out = nullptr;
  if (fclose(out) != 0 && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed", out_path);
  files.dbgh = nullptr;
  if (dbg && mfx_close(dbgh) && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed", opts->debug_path);
  files.log = nullptr;
  if (log != stderr) fclose(log);
  if (n_clusters) *n_clusters = clusters;
  return rc;
}

// the host pipeline allocates throughout (records, paths, batches): what it throws becomes an error code at the C ABI
template <class F> static int variants_guarded(const char *who, F &&body) {
  try { return body(); }
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "%s: out of memory", who); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_HIP, "%s: %s", who, e.what()); }
}

extern "C" int mfx_variants_run(mfx_eval *ev, const char *vcf_path, const char *const *names, const char *const *bases,
                                const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                                const char *out_path, const char *log_path, uint64_t *n_clusters) {
  if (!ev) return mfx_fail(MFX_E_INVAL, "mfx_variants_run: null argument");
  PathValues values = [ev](const char *text, uint64_t len, uint32_t *rv, uint32_t *av) -> int {
    mfx_seq *ps = mfx_seq_upload(ev->device, &text, &len, 1);
    if (!ps) return mfx_last_error_code();
    struct PathLookup { PathLookup() { t_mfx_path_lookup = true; } ~PathLookup() { t_mfx_path_lookup = false; } } pathLookup;   // (a path-only index answers this text)
    int r = mfx_dump_values(ev, ps, 0, 0, len, rv, av, nullptr, nullptr);
    mfx_seq_free(ps);
    return r;
  };
  // varMer::score of every path on the device (MFX_VAR_HOST_SCORE=1: on the host threads from the per-base values, as -debug
  // and the sharded index do)
  PathScores scores;
  const char *hs = getenv("MFX_VAR_HOST_SCORE");
  if (!(hs && atoi(hs)))
    scores = [ev](const char *text, uint64_t len, const mfx_path_table &pt, const mfx_trv_batch *tb, int need_dk, uint32_t *numM, double *totdk) -> int {
      return tb ? mfx_score_paths_trv(ev, text, len, &pt, tb, need_dk, numM, totdk) : mfx_score_paths(ev, text, len, &pt, need_dk, numM, totdk);
    };
  return variants_guarded("mfx_variants_run", [&] { return mfx_variants_run_values(ev, values, vcf_path, names, bases, lens, ncontigs, opts, out_path, log_path, n_clusters, scores); });
}

// The VCF read and parsed AHEAD of its run: host work only (no device, no index), so a caller starts it on a thread of its own while
// the index is built -- merfin opens the VCF after load_Kmers (merfin-globals.C:201-219), here the 0.15 s of a 4 M-call set run
// under the build.  The handle serves one mfx_variants_run_vcf and is freed by the caller.
extern "C" mfx_vcf *mfx_vcf_load(const char *vcf_path) {
  if (!vcf_path) { mfx_fail(MFX_E_INVAL, "mfx_vcf_load: null argument"); return nullptr; }
  mfx_vcf *v = new (std::nothrow) mfx_vcf;
  if (!v) { mfx_fail(MFX_E_NOMEM, "mfx_vcf_load: no memory"); return nullptr; }
  int rc;
  try { rc = load_vcf(vcf_path, v->db, v->t_load); }          // (nothing leaves the C ABI as an exception)
  catch (const std::bad_alloc &) { rc = mfx_fail(MFX_E_NOMEM, "loading VCF '%s': out of memory", vcf_path); }
  catch (const std::exception &e) { rc = mfx_fail(MFX_E_IO, "loading VCF '%s': %s", vcf_path, e.what()); }
  if (rc != MFX_OK) { delete v; return nullptr; }
  return v;
}

mfx_vcf::~mfx_vcf() { delete prep; }

extern "C" void mfx_vcf_free(mfx_vcf *v) { delete v; }

// Stage A of the variant modes AHEAD of the run, on a loaded VCF: the clusters merged for k, their allele combinations enumerated
// (merfin-variants.C:22-126, varMer.C:39) and packed batch by batch -- host work that needs the VCF and the sequences but neither the
// index nor the device, so a caller runs it next to mfx_vcf_load under its index build (config 4 at 3 Gb: 0.3 s of the 0.8 s behind the
// build).  The run (mfx_variants_run_vcf on the same handle, the same sequences, k, -comb, -nosplit: checked) starts at stage B; its
// outputs are the unprepared run's byte for byte.  opts->debug_path set: the path tables of the device scoring are not made (-debug is
// scored on the host).
extern "C" int mfx_vcf_prepare(mfx_vcf *vcf, int k, const char *const *names, const char *const *bases, const uint64_t *lens,
                               uint32_t ncontigs, const mfx_variant_opts *opts) {
  if (!vcf || !opts || k < 1 || k > MFX_MAX_K) return mfx_fail(MFX_E_INVAL, "mfx_vcf_prepare: null argument or k out of range");
  return variants_guarded("mfx_vcf_prepare", [&] {
    return mfx_variants_run_values(nullptr, PathValues(), nullptr, names, bases, lens, ncontigs, opts, nullptr, nullptr, nullptr, PathScores(), vcf, (uint32_t)k);
  });
}

// The PATH-ONLY index of the variant modes.  -filter / -polish / -better / -loose ask the lookup tables for the k-mers of the enumerated
// paths and for nothing else (varMer::score, varMer.C:76-84; the reference loads both databases whole, merfin-globals.C:114-163): a prepared call
// set (mfx_vcf_prepare) knows those paths before any database is read, so a sequence-only index claims exactly their k-mers and both databases
// then only UPDATE them -- a few hundred million slots instead of every read k-mer (3 Gb human, 3.8 M calls: a table of ~25 GB instead of 216).
// mfx_vcf_path_bound: the k-mer positions of all batches' text (an upper bound of the distinct k-mers: the capacity of the index);
// mfx_index_claim_paths: every batch's text is made on the device as the run will make it and claimed; the index is then bound to this handle
// (another call set, or a sequence's -hist / -dump, are refused).  A cluster the device cannot enumerate makes the call fail: use the full index.
extern "C" int mfx_vcf_path_bound(const mfx_vcf *vcf, uint64_t *positions) {
  if (!vcf || !positions) return mfx_fail(MFX_E_INVAL, "mfx_vcf_path_bound: null argument");
  if (!vcf->prep) return mfx_fail(MFX_E_INVAL, "mfx_vcf_path_bound: the VCF is not prepared (mfx_vcf_prepare)");
  uint64_t n = 0;
  for (const VarBatch &b : vcf->prep->batches) n += std::max<uint64_t>(b.t_text_end, b.packed.size());
  *positions = n;
  return MFX_OK;
}
extern "C" int mfx_index_claim_paths(mfx_index *ix, mfx_vcf *vcf, uint64_t *n_positions) {
  if (!ix || !vcf) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: null argument");
  VarPrepared *prep = vcf->prep;
  if (!prep) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: the VCF is not prepared (mfx_vcf_prepare)");
  if (vcf->used) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: the prepared VCF was run already");
  if ((uint32_t)ix->k != prep->k) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: the VCF was prepared for k = %u, the index holds %d-mers", prep->k, ix->k);
  return variants_guarded("mfx_index_claim_paths", [&]() -> int {
    uint8_t *scratch = nullptr;
    uint64_t scratch_bytes = 0, positions = 0;
    struct Release { int dev; uint8_t *&p; ~Release() { mfx_claim_paths_release(dev, p); } } release{ix->device, scratch};
    for (VarBatch &b : prep->batches) {
      if (b.jobs.empty()) continue;
      mfx_trv_batch tb;
      if (!b.t_cl.empty()) {
        tb.ncl = b.t_cl.size(); tb.nvar = b.t_var.size(); tb.nal = b.t_al.size(); tb.win_bytes = b.t_win.size(); tb.al_bytes = b.t_alt.size();
        tb.cl = b.t_cl.data(); tb.var = b.t_var.data(); tb.al = b.t_al.data(); tb.win_text = b.t_win.data(); tb.al_text = b.t_alt.data();
        tb.text_end = b.t_text_end; tb.path_cap = b.t_path_cap; tb.row_cap = b.t_row_cap;
        tb.np = b.t_np.data(); tb.status = b.t_status.data(); tb.p_len = b.t_plen.data(); tb.gt = b.t_gt.data();
      }
      uint64_t bad = 0;
      if (int rc = mfx_claim_paths_batch(ix, &scratch, &scratch_bytes, b.packed.data(), b.packed.size(), b.t_cl.empty() ? nullptr : &tb, &bad)) return rc;
      if (bad) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: %lu clusters of the call set cannot be enumerated on the device (their paths are made on the host "
                                            "during the run); use the full index (mfx_index_create)", (unsigned long)bad);
      positions += std::max<uint64_t>(b.t_text_end, b.packed.size());
    }
    if (n_positions) *n_positions = positions;
    return mfx_claim_paths_finish(ix, prep->token);
  });
}

// mfx_vcf_prepare + mfx_index_create_for_seq_lf + mfx_index_claim_paths as ONE pass: the table is made as soon as the clusters are merged (its
// capacity is the bound of their path text), and every batch's k-mers are claimed on the device while the host prepares the next batch -- the
// claims (0.13 s of a 3.7 M-call human set) disappear under the preparation.  *out: the claimed index (the databases come next), or NULL with
// MFX_OK when the path-only index cannot be made for this call set (no memory, a cluster the device cannot enumerate: mfx_last_error says why) --
// the handle is prepared either way and runs on a full index as well.
extern "C" int mfx_vcf_prepare_path_index(mfx_vcf *vcf, int k, const char *const *names, const char *const *bases, const uint64_t *lens, uint32_t ncontigs,
                                          const mfx_variant_opts *opts, double max_gb, int device, double load_factor, mfx_index **out) {
  if (!vcf || !opts || !out || k < 1 || k > MFX_MAX_K_NARROW) return mfx_fail(MFX_E_INVAL, "mfx_vcf_prepare_path_index: null argument or k out of range (k <= %d)", MFX_MAX_K_NARROW);
  *out = nullptr;
  return variants_guarded("mfx_vcf_prepare_path_index", [&]() -> int {
    mfx_index *ix = nullptr;
    uint8_t *scratch = nullptr;
    uint64_t scratch_bytes = 0;
    struct Release { int dev; uint8_t *&p; mfx_index *&ix; ~Release() { mfx_claim_paths_release(dev, p); if (ix) mfx_index_free(ix); } } release{device, scratch, ix};
    PathClaims claims;
    claims.on_bound = [&](uint64_t bound) -> int {
      ix = load_factor > 0 ? mfx_index_create_for_seq_lf(k, bound + 1024, max_gb, device, load_factor) : mfx_index_create_for_seq(k, bound + 1024, max_gb, device);
      return ix ? MFX_OK : mfx_last_error_code();
    };
    claims.claim = [&](const char *text, uint64_t len, const mfx_trv_batch *tb, uint64_t *bad) -> int { return mfx_claim_paths_batch(ix, &scratch, &scratch_bytes, text, len, tb, bad); };
    const int rc = mfx_variants_run_values(nullptr, PathValues(), nullptr, names, bases, lens, ncontigs, opts, nullptr, nullptr, nullptr, PathScores(), vcf, (uint32_t)k, &claims);
    if (rc) return rc;
    if (claims.rc == MFX_OK && ix) {
      const int frc = mfx_claim_paths_finish(ix, vcf->prep->token);
      if (frc == MFX_OK) { *out = ix; ix = nullptr; return MFX_OK; }
      claims.rc = frc; claims.err = mfx_last_error();
    }
    (void)mfx_fail(claims.rc ? claims.rc : MFX_E_INVAL, "no path-only index for this call set: %s", claims.err.c_str());     // (the reason; the call itself succeeded)
    return MFX_OK;
  });
}

extern "C" int mfx_variants_run_vcf(mfx_eval *ev, mfx_vcf *vcf, const char *const *names, const char *const *bases,
                                    const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                                    const char *out_path, const char *log_path, uint64_t *n_clusters) {
  if (!ev || !vcf) return mfx_fail(MFX_E_INVAL, "mfx_variants_run_vcf: null argument");
  PathValues values = [ev](const char *text, uint64_t len, uint32_t *rv, uint32_t *av) -> int {
    mfx_seq *ps = mfx_seq_upload(ev->device, &text, &len, 1);
    if (!ps) return mfx_last_error_code();
    struct PathLookup { PathLookup() { t_mfx_path_lookup = true; } ~PathLookup() { t_mfx_path_lookup = false; } } pathLookup;   // (a path-only index answers this text)
    int r = mfx_dump_values(ev, ps, 0, 0, len, rv, av, nullptr, nullptr);
    mfx_seq_free(ps);
    return r;
  };
  PathScores scores;
  const char *hs = getenv("MFX_VAR_HOST_SCORE");
  if (!(hs && atoi(hs)))
    scores = [ev](const char *text, uint64_t len, const mfx_path_table &pt, const mfx_trv_batch *tb, int need_dk, uint32_t *numM, double *totdk) -> int {
      return tb ? mfx_score_paths_trv(ev, text, len, &pt, tb, need_dk, numM, totdk) : mfx_score_paths(ev, text, len, &pt, need_dk, numM, totdk);
    };
  return variants_guarded("mfx_variants_run_vcf", [&] { return mfx_variants_run_values(ev, values, nullptr, names, bases, lens, ncontigs, opts, out_path, log_path, n_clusters, scores, vcf); });
}

// The variant modes over an index sharded across N evaluators (read databases beyond one GPU): the packed path text of
// a batch goes to every slot's device and mfx_dump_values_sharded adds the shards' answers; everything else is the
// single-evaluator code above.
extern "C" int mfx_variants_run_sharded(mfx_eval *const *evs, uint32_t nslots, const char *vcf_path, const char *const *names,
                                        const char *const *bases, const uint64_t *lens, uint32_t ncontigs,
                                        const mfx_variant_opts *opts, const char *out_path, const char *log_path,
                                        uint64_t *n_clusters) {
  if (!evs || nslots == 0) return mfx_fail(MFX_E_INVAL, "mfx_variants_run_sharded: null argument");
  for (uint32_t d = 0; d < nslots; ++d)
    if (!evs[d]) return mfx_fail(MFX_E_INVAL, "mfx_variants_run_sharded: slot %u is null", d);
  PathValues values = [evs, nslots](const char *text, uint64_t len, uint32_t *rv, uint32_t *av) -> int {
    std::vector<mfx_seq *> sq(nslots, nullptr);
    int r = MFX_OK;
    for (uint32_t d = 0; d < nslots && r == MFX_OK; ++d) {
      for (uint32_t e = 0; e < d; ++e)
        if (evs[e]->device == evs[d]->device) { sq[d] = sq[e]; break; }           // one copy per device
      if (!sq[d]) sq[d] = d == 0 ? mfx_seq_upload(evs[0]->device, &text, &len, 1) : mfx_seq_replicate(sq[0], evs[d]->device);
      if (!sq[d]) r = mfx_last_error_code();
    }
    if (r == MFX_OK) r = mfx_dump_values_sharded(evs, sq.data(), nslots, 0, 0, len, rv, av, nullptr, nullptr);
    for (uint32_t d = 0; d < nslots; ++d) {
      bool shared = false;
      for (uint32_t e = 0; e < d; ++e) if (sq[e] == sq[d]) shared = true;
      if (sq[d] && !shared) mfx_seq_free(sq[d]);
    }
    return r;
  };
  return variants_guarded("mfx_variants_run_sharded", [&] { return mfx_variants_run_values(evs[0], values, vcf_path, names, bases, lens, ncontigs, opts, out_path, log_path, n_clusters); });
}
