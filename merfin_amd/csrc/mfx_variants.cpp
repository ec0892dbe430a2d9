// mfx_variants.cpp -- merfin's variant modes (-filter / -polish / -better /
// -strict / -loose) on top of the GPU lookup path.
//
// Reference behaviour implemented (paths relative to /root/reference):
//   VCF model + clustering    src/merfin/vcfRecord.H:50-100, vcf.C:23-87, 93-149, 156-246
//   allele-combination DFS    src/merfin/merfin-variants.C:22-126   (traverse)
//   per-cluster driver        src/merfin/merfin-variants.C:131-345
//   scoring + selectors       src/merfin/varMer.C:37-659
//
// Shape of this implementation (not the reference's): the host enumerates the
// allele-combination paths of MANY clusters first, packs all path strings into
// one buffer separated by a non-ACGT byte, and scores them with ONE launch of
// the same lookup kernel -dump uses (mfx_dump_values): every k-mer of every
// path is extracted, canonicalised and probed on the GPU.  The host then turns
// the raw (readV, asmV) per base into numM / K* / delta-K per path and applies
// the selector of the chosen mode.  merfin scores cluster by cluster with four
// CPU probes per base (varMer.C:76-84).
#include "mfx_internal.h"
#include "mfx_pipe.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <list>
#include <map>
#include <chrono>
#include <functional>
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
struct Record {                       // vcfRecord
  std::string chr, id, ref, alts, filter, info, formats, samples;
  uint32_t pos = 0;
  double qual = 0;
  std::vector<std::string> alt_list; // _arr_alts
  std::string gt_field;              // _arr_samples[0]
  std::string line() const {         // vcfRecord::save, vcfRecord.H:96-97
    char q[64];
    snprintf(q, sizeof(q), "%.1f", qual);
    return chr + "\t" + std::to_string((int)pos) + "\t" + id + "\t" + ref + "\t" + alts + "\t" + q + "\t" + filter + "\t" + info + "\t" + formats + "\t" + samples + "\n";
  }
};

struct Variant {                      // gtAllele
  const Record *rec;
  uint32_t pos, refLen;
  double qual;
  std::vector<const std::string *> alleles;   // [0] = REF; empty for ./. and 0/0 genotypes
};

struct Cluster {                      // posGT
  uint32_t rStart, rEnd;
  std::vector<const Variant *> vars;
};

struct VcfDB {
  std::vector<std::string> headers;
  std::vector<Record *> records;
  std::vector<Variant *> variants;
  std::map<std::string, std::vector<Cluster *>> by_chr;
  uint64_t excluded = 0;
  int contig_ids = 0;
  ~VcfDB() {
    for (auto r : records) delete r;
    for (auto v : variants) delete v;
    for (auto &kv : by_chr) for (auto c : kv.second) delete c;
  }
};

// gtAllele::gtAllele, vcf.C:23-87
Variant *make_variant(const Record *r) {
  Variant *v = new Variant;
  v->rec = r;
  v->pos = r->pos - 1;
  v->refLen = (uint32_t)r->ref.size();
  v->qual = r->qual;
  const std::string &g = r->gt_field;
  if (g.compare(0, 3, "./.") == 0 || g.compare(0, 3, "0/0") == 0)
    return v;                                            // no alleles at all
  v->alleles.push_back(&r->ref);
  std::vector<int> seen_alt;
  for (const std::string &tok : split_any(g, "|/")) {
    long altIdx = strtol(tok.c_str(), nullptr, 10);
    if ((int32_t)altIdx <= 0) continue;
    if ((size_t)altIdx > r->alt_list.size()) continue;   // operator[] past the end -> nullptr
    // the reference compares POINTERS into the ALT list (same ALT index listed twice) ...
    if (std::find(seen_alt.begin(), seen_alt.end(), (int)altIdx) != seen_alt.end()) continue;
    const std::string &hap = r->alt_list[altIdx - 1];
    if (hap == r->ref) continue;                         // ... and the STRING against the reference allele only
    seen_alt.push_back((int)altIdx);
    v->alleles.push_back(&hap);
  }
  return v;
}

// one data line -> record (nullptr: fewer than 10 columns, "excluded", vcfRecord.H:53-56)
Record *parse_record(const char *L, size_t n) {
  std::vector<std::string> w = split_any(std::string(L, n), "\t");
  if (w.size() < 10) return nullptr;
  Record *r = new Record;
  r->chr = w[0];
  r->pos = (uint32_t)strtoul(w[1].c_str(), nullptr, 10);
  r->id = w[2]; r->ref = w[3]; r->alts = w[4];
  r->qual = strtod(w[5].c_str(), nullptr);
  r->filter = w[6]; r->info = w[7]; r->formats = w[8]; r->samples = w[9];
  r->alt_list = split_any(r->alts, ",");
  std::vector<std::string> smp = split_any(r->samples, ":");
  r->gt_field = smp.empty() ? std::string() : smp[0];
  return r;
}

template <class F> void parallel_for(size_t n, F &&fn);

// vcfFile::loadFile, vcf.C:93-149.  The file is read whole, its data lines are parsed by the host threads (a
// config-4-sized VCF has millions of records, and the per-record string work is most of the load time), and the
// records enter the database in FILE ORDER, exactly as a sequential read would put them.
int load_vcf(const char *path, VcfDB &db) {
  mfx_file fh = mfx_open_reader(path);
  FILE *f = fh.f;
  if (!f) return mfx_fail(MFX_E_IO, "cannot open VCF '%s'", path);
  std::string buf;
  {
    std::vector<char> blk(1 << 22);
    size_t n;
    while ((n = fread(blk.data(), 1, blk.size(), f)) > 0) buf.append(blk.data(), n);
  }
  if (mfx_close(fh)) return mfx_fail(MFX_E_IO, "reading VCF '%s' failed (stream error or the decompressor exited with an error)", path);
  std::vector<std::pair<size_t, size_t>> lines;           // (offset, length) of every data line
  for (size_t o = 0; o < buf.size();) {
    const char *nl = (const char *)memchr(buf.data() + o, '\n', buf.size() - o);
    size_t e = nl ? (size_t)(nl - buf.data()) : buf.size(), n = e - o;
    while (n > 0 && (buf[o + n - 1] == '\n' || buf[o + n - 1] == '\r')) --n;
    if (buf[o] == '#' && (nl || e > o)) {
      db.headers.emplace_back(buf.data() + o, n);
      if (n >= 12 && strncmp(buf.data() + o, "##contig=<ID", 12) == 0) db.contig_ids++;
    } else if (nl || e > o) {
      lines.emplace_back(o, n);
    }
    o = e + 1;
  }
  std::vector<Record *> recs(lines.size(), nullptr);
  std::vector<Variant *> vars(lines.size(), nullptr);
  const size_t CH = 4096;                                  // lines per task
  parallel_for((lines.size() + CH - 1) / CH, [&](size_t c) {
    for (size_t i = c * CH, e = std::min(lines.size(), (c + 1) * CH); i < e; ++i) {
      recs[i] = parse_record(buf.data() + lines[i].first, lines[i].second);
      if (recs[i]) vars[i] = make_variant(recs[i]);
    }
  });
  db.records.reserve(lines.size());
  db.variants.reserve(lines.size());
  std::vector<Cluster *> *bucket = nullptr;                // records of one CHROM come in runs: one map lookup per run
  const std::string *bucket_chr = nullptr;
  for (size_t i = 0; i < lines.size(); ++i) {
    Record *r = recs[i];
    if (!r) { db.excluded++; continue; }
    db.records.push_back(r);
    Variant *v = vars[i];
    db.variants.push_back(v);
    Cluster *c = new Cluster;
    c->rStart = v->pos;
    c->rEnd = v->pos + v->refLen;
    c->vars.push_back(v);
    if (!bucket || *bucket_chr != r->chr) {
      auto it = db.by_chr.find(r->chr);
      if (it == db.by_chr.end()) it = db.by_chr.emplace(r->chr, std::vector<Cluster *>()).first;
      bucket = &it->second;
      bucket_chr = &it->first;
    }
    bucket->push_back(c);
  }
  return MFX_OK;
}

// vcfFile::mergeChrPosGT, vcf.C:156-246: clusters whose start lies within 2k of
// the previous cluster's end are merged, unless that cluster already holds
// `comb` variants and splitting is allowed.
void merge_clusters(VcfDB &db, uint32_t k, uint32_t comb, bool nosplit, FILE *log) {
  const uint32_t K_OFFSET = 2 * k;
  for (auto &kv : db.by_chr) {
    std::vector<Cluster *> &in = kv.second;
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
      delete cur;
    }
    if (log) {
      fprintf(log, "%s : Reduced %lu variants down to %lu combinations for evaluation:\n", kv.first.c_str(), in.size(), out.size());
      if (split > 0) fprintf(log, "%s :   Split   %u complicated combinations.\n", kv.first.c_str(), split);
      if (merged > 0) fprintf(log, "%s :   Merged  %u variants into combinations.\n", kv.first.c_str(), merged);
    }
    in.swap(out);
  }
}

// ---- allele-combination enumeration (traverse) ------------------------------
struct PathSet {                      // varMer's per-cluster containers
  std::vector<std::string> seqs;
  std::vector<std::vector<int>> gt;          // gtPaths
  std::vector<std::vector<uint32_t>> vidx;   // idxPaths (shifted offsets snapshot)
  std::vector<std::vector<uint32_t>> vlen;   // lenPaths (post-substitution lengths snapshot)
  void add(const std::string &s, const std::vector<int> &g, const std::vector<uint32_t> &ix, const std::vector<uint32_t> &ln) {
    if (std::find(seqs.begin(), seqs.end(), s) != seqs.end()) return;    // varMer.C:39
    seqs.push_back(s); gt.push_back(g); vidx.push_back(ix); vlen.push_back(ln);
  }
};

// merfin-variants.C:22-126.  `lens` and `cand` are per-call copies, `offs` and
// `path` are shared -- that asymmetry is part of the observable behaviour
// (the stored snapshots feed the "new k-mer" test of scoring).
void enumerate(uint32_t idx, std::vector<uint32_t> &offs, std::vector<uint32_t> lens, const Cluster &cl,
               const std::string &cand, std::vector<int> &path, PathSet &out) {
  const std::vector<const std::string *> &haps = cl.vars[idx]->alleles;
  const uint32_t refLen = lens[idx];
  const uint32_t last = (uint32_t)offs.size() - 1;
  for (int j = 0; j < (int)haps.size(); ++j) {
    path.push_back(j);
    std::string rep = cand;
    int skipped = 0, delta = 0;
    if (j > 0) {
      const std::string &hap = *haps[j];
      lens[idx] = refLen;
      rep.replace(offs[idx], lens[idx], hap);
      delta = (int)hap.size() - (int)lens[idx];
      const uint32_t affected = offs[idx] + lens[idx];
      lens[idx] = (uint32_t)hap.size();
      for (uint32_t i = idx + 1; i < offs.size() && offs[i] < affected; ++i) {   // later variants inside this REF span: forced to REF
        ++idx; path.push_back(0); ++skipped;
      }
      if (skipped > 0 && idx == last) {
        out.add(rep, path, offs, lens);
        for (int q = 0; q < skipped; ++q) { path.pop_back(); --idx; }
        path.pop_back();
        continue;
      }
      for (uint32_t i = idx + 1; i < offs.size(); ++i) offs[i] += delta;
    }
    if (idx + 1 < offs.size()) enumerate(idx + 1, offs, lens, cl, rep, path, out);
    if (idx == last) out.add(rep, path, offs, lens);
    for (uint32_t i = idx + 1; i < offs.size(); ++i) offs[i] -= delta;
    for (int q = 0; q < skipped; ++q) { path.pop_back(); --idx; }
    path.pop_back();
  }
}

// ---- scoring + selection -----------------------------------------------------
struct Scored {
  std::vector<uint32_t> numM;
  std::vector<std::vector<double>> ks, dks;
};

inline int base_ok(unsigned char c) {
  switch (c) { case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1; default: return 0; }
}

struct Job {                          // one cluster waiting for its GPU values
  const Cluster *cl;
  uint32_t contig;
  uint32_t rStart, rEnd;
  PathSet ps;
  std::vector<uint64_t> off;          // offset of each path in the packed buffer
  uint64_t first_id = 0;              // varMerId of its first path (-debug numbering)
  std::string out, dbg, log;          // produced by a worker thread, written in input order
};

// dynamic parallel-for over [0, n) on the host threads the library may use
template <class F>
void parallel_for(size_t n, F &&fn) {
  unsigned nt = std::min<size_t>(mfx_host_threads(), n);
  if (nt <= 1) { for (size_t i = 0; i < n; ++i) fn(i); return; }
  const size_t chunk = std::max<size_t>(1, std::min<size_t>(64, n / (nt * 8)));
  std::atomic<size_t> next(0);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (size_t b; (b = next.fetch_add(chunk)) < n;)
        for (size_t i = b, e = std::min(n, b + chunk); i < e; ++i) fn(i);
    });
  for (auto &x : th) x.join();
}

std::string hom_record(const Cluster &cl, const std::vector<int> &g, const char *chr) {     // varMer.C:531-550
  std::string out;
  for (size_t i = 0; i < g.size(); ++i) {
    int a = g[i];
    if (a <= 0) continue;
    const Variant *v = cl.vars[i];
    out += std::string(chr) + "\t" + std::to_string(v->pos + 1) + "\t.\t" + *v->alleles[0] + "\t" + *v->alleles[a] + "\t" +
           std::to_string((int)v->qual) + "\tPASS\t.\tGT\t1/1\n";
  }
  return out;
}

std::string het_record(const Cluster &cl, const std::vector<int> &g1, const std::vector<int> &g2, const char *chr) {   // varMer.C:472-529
  std::string out;
  for (size_t i = 0; i < g1.size(); ++i) {
    int a1 = g1[i], a2 = g2[i];
    if (a1 + a2 <= 0) continue;
    const Variant *v = cl.vars[i];
    std::string q = std::to_string((int)v->qual);
    out += std::string(chr) + "\t" + std::to_string(v->pos + 1) + "\t.\t" + *v->alleles[0] + "\t";
    if (a1 == a2) out += *v->alleles[a1] + "\t" + q + "\tPASS\t.\tGT\t1/1\n";
    else if (a1 == 0 && a2 > 0) out += *v->alleles[a2] + "\t" + q + "\tPASS\t.\tGT\t0/1\n";
    else if (a1 > 0 && a2 > 0) out += *v->alleles[a1] + "," + *v->alleles[a2] + "\t" + q + "\tPASS\t.\tGT\t1/2\n";
    else if (a1 > 0 && a2 == 0) out += *v->alleles[a1] + "\t" + q + "\tPASS\t.\tGT\t1/0\n";
  }
  return out;
}

double tot_dk(const std::vector<double> &d) { double s = 0; for (double x : d) s += x; return s; }   // getTotdK

// paths with the fewest missing k-mers, optionally ignoring all-missing paths (varMer.C:156-178, 406-421)
std::vector<int> min_missing(const Job &jb, const Scored &sc, uint32_t k, bool filter_rule, uint32_t *best) {
  uint32_t numMissing = UINT32_MAX;
  std::vector<int> idxs;
  for (int i = 0; i < (int)sc.numM.size(); ++i) {
    if (sc.numM[i] == jb.ps.seqs[i].size() - k + 1) continue;            // size_t arithmetic, as the reference
    if (filter_rule && sc.numM[i] == 0) { idxs.push_back(i); numMissing = 0; }
    if (sc.numM[i] < numMissing) { numMissing = sc.numM[i]; idxs.clear(); idxs.push_back(i); }
    else if (sc.numM[i] == numMissing) idxs.push_back(i);
  }
  *best = numMissing;
  return idxs;
}

std::string select_records(const Job &jb, const Scored &sc, int mode, uint32_t k, const char *chr, std::string *log) {
  const PathSet &ps = jb.ps;
  const Cluster &cl = *jb.cl;
  if (mode == MFX_VAR_FILTER) {                                          // bestFilter, varMer.C:150-199
    uint32_t best;
    std::vector<int> idxs = min_missing(jb, sc, k, true, &best);
    if (idxs.empty()) return "";
    std::list<int> gtIdxs;
    for (int p : idxs)
      for (int i = 0; i < (int)ps.gt[p].size(); ++i)
        if (ps.gt[p][i] > 0) gtIdxs.push_back(i);
    gtIdxs.sort();
    gtIdxs.unique();
    std::string out;
    for (int i : gtIdxs) out += cl.vars[i]->rec->line();
    return out;
  }
  if (mode == MFX_VAR_POLISH) {                                          // bestVariant, varMer.C:400-467
    uint32_t best;
    std::vector<int> idxs = min_missing(jb, sc, k, false, &best);
    if (best == UINT32_MAX) return "";
    if (idxs.size() == 1) return hom_record(cl, ps.gt[idxs[0]], chr);
    // tie: order by total delta-K through the reference's own container type --
    // multimap<double,int,greater<int>> compares the keys AS INTS, descending (varMer.H:72)
    std::multimap<double, int, std::greater<int>> byDk;
    for (int p : idxs) byDk.insert(std::make_pair(tot_dk(sc.dks[p]), p));
    auto it = byDk.begin();
    double d1 = it->first; int p1 = it->second;
    ++it;
    double d2 = it->first; int p2 = it->second;
    if (d1 == d2) {
      if (ps.seqs[p1].length() >= ps.seqs[p2].length()) return het_record(cl, ps.gt[p1], ps.gt[p2], chr);
      return het_record(cl, ps.gt[p2], ps.gt[p1], chr);
    }
    return hom_record(cl, ps.gt[p1], chr);
  }
  // -better / -strict / -loose start from the reference path (varMer.C:204-395)
  if (sc.numM.empty()) return "";
  const uint32_t refMissing = sc.numM[0];
  uint32_t numMissing = refMissing;
  std::vector<int> idxs;
  const bool loose = mode == MFX_VAR_LOOSE;
  for (int i = 0; i < (int)sc.numM.size(); ++i) {
    if (sc.numM[i] < numMissing) { numMissing = sc.numM[i]; idxs.clear(); idxs.push_back(i); }
    else if (sc.numM[i] == numMissing && (loose ? sc.numM[i] <= refMissing : sc.numM[i] < refMissing)) idxs.push_back(i);
  }
  if (idxs.empty()) return "";
  if (idxs.size() == 1) return hom_record(cl, ps.gt[idxs[0]], chr);
  if (!loose) {                                                          // longest path wins (first on ties)
    int idx = idxs[0];
    uint32_t longest = (uint32_t)ps.seqs[idx].size();
    for (size_t i = 1; i < idxs.size(); ++i) {
      uint32_t L = (uint32_t)ps.seqs[idxs[i]].length();
      if (L > longest) { longest = L; idx = idxs[i]; }
    }
    return hom_record(cl, ps.gt[idx], chr);
  }
  if (idxs[0] == 0 && idxs.size() == 2) return hom_record(cl, ps.gt[idxs[1]], chr);
  int maxVars = 0, maxIdx = idxs[0];                                     // most ALT alleles wins
  for (size_t i = 1; i < idxs.size(); ++i) {
    int cnt = 0;
    for (int a : ps.gt[idxs[i]]) if (a > 0) cnt++;
    if (cnt > maxVars) { maxVars = cnt; maxIdx = idxs[i]; }
  }
  if (log) {
    *log += "[ WARNING ] :: Multiple (" + std::to_string(idxs.size()) + ") alternate pathes detected in a path beginning with variant : " + cl.vars[0]->rec->line();
    *log += "[ WARNING ] :: Max. " + std::to_string(maxVars) + " ALT variants selected\n";
  }
  return hom_record(cl, ps.gt[maxIdx], chr);
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

static int variants_impl(const mfx_eval *ev, const PathValues &values, const char *vcf_path, const char *const *names, const char *const *bases,
                         const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                         const char *out_path, const char *log_path, uint64_t *n_clusters) {
  if (!ev || !vcf_path || !opts || !out_path || (ncontigs && (!names || !bases || !lens)))
    return mfx_fail(MFX_E_INVAL, "mfx_variants_run: null argument");
  const int mode = opts->mode;
  if (mode < MFX_VAR_FILTER || mode > MFX_VAR_LOOSE) return mfx_fail(MFX_E_INVAL, "mfx_variants_run: unknown mode %d", mode);
  if (ev->ix->seq_only)      // the alternative paths ask for k-mers the sequence does not hold (varMer.C:76-84)
    return mfx_fail(MFX_E_INVAL, "mfx_variants_run: a sequence-only index holds the k-mers of one sequence; the variant modes need a full index (mfx_index_create)");
  const uint32_t K = (uint32_t)ev->ix->k;
  const uint32_t comb = opts->comb ? opts->comb : 15;
  FILE *log = log_path ? fopen(log_path, "w") : stderr;
  if (!log) return mfx_fail(MFX_E_IO, "cannot open '%s'", log_path);

  // MFX_VAR_TIMING=1: per-phase wall time on stderr (diagnostics only)
  const bool timing = getenv("MFX_VAR_TIMING") && atoi(getenv("MFX_VAR_TIMING"));
  double t_phase[6] = {0, 0, 0, 0, 0, 0};                                // load+cluster, enumerate, pack, gpu, score+select, write
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int i) { double t = now(); t_phase[i] += t - t_mark; t_mark = t; };

  VcfDB db;
  int rc = load_vcf(vcf_path, db);
  if (rc) { if (log != stderr) fclose(log); return rc; }
  fprintf(log, "   Collected %zu header lines.\n   Loaded %zu records:\n      %-8lu unique contig%s\n      %-8u contig IDs\n   Excluded %lu invalid records\n\n",
          db.headers.size(), db.records.size(), db.by_chr.size(), db.by_chr.size() == 1 ? "" : "s", (unsigned)db.contig_ids, (unsigned long)db.excluded);
  fprintf(log, "Merge variants within %u-mer bases, splitting combinations greater than %u.\n", K, comb);
  merge_clusters(db, K, comb, opts->nosplit != 0, log);
  lap(0);

  FILE *out = fopen(out_path, "w");
  if (!out) { if (log != stderr) fclose(log); return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", out_path); }
  for (auto &h : db.headers) fprintf(out, "%s\n", h.c_str());              // merfin-variants.C:332-333
  mfx_file dbgh;
  FILE *dbg = nullptr;
  if (opts->debug_path) {
    dbgh = mfx_open_writer(opts->debug_path, false);                       // compressedFileWriter, merfin-variants.C:149
    dbg = dbgh.f;
  }

  mfx_kparams kp{ev->peak, ev->n_prob, ev->probK.data(), ev->probP.data()};
  // readK and prob depend on the read count alone (merfin-globals.C:80-97): evaluated once for the counts that occur all
  // the time, by the same routine the scoring loop would call (identical doubles)
  constexpr uint32_t KLUT = 4096;
  std::vector<double> lutK(KLUT), lutP(KLUT);
  for (uint32_t v = 0; v < KLUT; ++v) { double a; mfx_getK(&kp, v, 0, &lutK[v], &a, &lutP[v]); }
  uint64_t clusters = 0, varMerId = 0;
  const uint64_t BATCH_BYTES = 256ull << 20;                             // packed path text per GPU launch

  std::vector<Job> jobs;
  jobs.reserve(65536);
  std::vector<char> out_buf(4u << 20);
  setvbuf(out, out_buf.data(), _IOFBF, out_buf.size());
  std::string packed;
  std::vector<uint32_t> rv, av;

  // Enumerates the queued clusters' paths (host threads), scores every path k-mer
  // with ONE GPU launch, applies the selectors (host threads), writes in input order.
  auto flush = [&]() -> int {
    if (jobs.empty()) return MFX_OK;
    lap(5);
    parallel_for(jobs.size(), [&](size_t i) {
      Job &jb = jobs[i];
      std::vector<uint32_t> offs, vl;
      for (const Variant *v : jb.cl->vars) { offs.push_back(v->pos - jb.rStart); vl.push_back(v->refLen); }
      std::vector<int> path;
      enumerate(0, offs, vl, *jb.cl, std::string(bases[jb.contig] + jb.rStart, bases[jb.contig] + jb.rEnd), path, jb.ps);
    });
    lap(1);
    uint64_t total = 0;
    for (Job &jb : jobs) {
      jb.first_id = varMerId;
      varMerId += jb.ps.seqs.size();
      for (const std::string &s : jb.ps.seqs) { jb.off.push_back(total); total += s.size() + 1; }
    }
    packed.assign(total, '\n');                                          // '\n' is not ACGT: k-mers never span two paths
    parallel_for(jobs.size(), [&](size_t i) {
      Job &jb = jobs[i];
      for (size_t p = 0; p < jb.ps.seqs.size(); ++p) memcpy(&packed[jb.off[p]], jb.ps.seqs[p].data(), jb.ps.seqs[p].size());
    });
    lap(2);
    if (total) {
      rv.resize(packed.size() + 1);
      av.resize(packed.size() + 1);
      int r = values(packed.data(), packed.size(), rv.data(), av.data());
      if (r) return r;
    }
    lap(3);
    const bool want_dbg = dbg != nullptr;
    parallel_for(jobs.size(), [&](size_t ji) {
      Job &jb = jobs[ji];
      // `prob` is a local of varMer::score (one per cluster) that the reference reads
      // uninitialised until the first valid k-mer writes it; before that it only
      // multiplies |0-0|, so any finite start value is equivalent.  We fix 1.0.
      double prob = 1.0;
      Scored sc;
      const size_t np = jb.ps.seqs.size();
      sc.numM.resize(np); sc.ks.resize(np); sc.dks.resize(np);
      for (size_t p = 0; p < np; ++p) {                                  // varMer::score, varMer.C:66-144
        const std::string &s = jb.ps.seqs[p];
        const uint64_t o = jb.off[p];
        uint32_t numM = 0, run = 0;
        std::vector<double> &ks = sc.ks[p], &dks = sc.dks[p];
        if (mode != MFX_VAR_FILTER) { ks.reserve(s.size()); dks.reserve(s.size()); }
        for (uint32_t idx = 0; idx < s.size(); ++idx) {
          run = base_ok((unsigned char)s[idx]) ? run + 1 : 0;
          double readK = 0, asmK = 0;
          if (run >= K) {                                                // k-mer ENDING at idx starts at idx-k+1
            const uint64_t sp = o + idx - (K - 1);
            const uint32_t v = rv[sp];
            if (v < KLUT) { readK = lutK[v]; prob = lutP[v]; asmK = (double)av[sp]; }
            else mfx_getK(&kp, v, av[sp], &readK, &asmK, &prob);
          }
          if (readK == 0) numM++;
          if (mode == MFX_VAR_FILTER) continue;                          // :93-96
          const double oD = fabs(readK - asmK) * prob;                   // :99
          for (size_t j = 0; j < jb.ps.vidx[p].size(); ++j) {            // :103-112, uint32 wrap included
            const uint32_t vi = jb.ps.vidx[p][j], vl = jb.ps.vlen[p][j];
            if (jb.ps.gt[p][j] > 0 && vi + 1 - K <= idx && idx < vi + vl + K) { asmK++; break; }
          }
          double kM;
          if (readK == 0) kM = -1;                                       // :116-124
          else if (readK > asmK) kM = readK / asmK - 1;
          else kM = asmK / readK - 1;
          const double nD = fabs(readK - asmK) * prob;                   // :126
          ks.push_back(kM);
          dks.push_back(oD - nD);
        }
        sc.numM[p] = numM;
      }
      const char *chr = names[jb.contig];
      if (want_dbg) {                                                    // merfin-variants.C:240-276
        char buf[512];
        for (size_t p = 0; p < np; ++p) {
          snprintf(buf, sizeof(buf), "%lu\t%s:%u-%u\t", (unsigned long)(jb.first_id + p), chr, jb.rStart, jb.rEnd);
          jb.dbg += buf;
          jb.dbg += jb.ps.seqs[p];
          snprintf(buf, sizeof(buf), "\t%u\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\t", sc.numM[p], min_abs_k(sc.ks[p]), max_abs_k(sc.ks[p]),
                   med_abs_k(sc.ks[p]), avg_abs_k(sc.ks[p], sc.numM[p]), tot_dk(sc.dks[p]));
          jb.dbg += buf;
          for (size_t i = 0; i < jb.ps.gt[p].size(); ++i) {
            int a = jb.ps.gt[p][i];
            if (a > 0)
              jb.dbg += std::string(chr) + " " + std::to_string(jb.cl->vars[i]->pos + 1) + " . " + *jb.cl->vars[i]->alleles[0] + " " +
                        *jb.cl->vars[i]->alleles[a] + " . PASS . GT 1/1  ";
          }
          jb.dbg += "\n";
        }
      }
      jb.out = select_records(jb, sc, mode, K, chr, &jb.log);
      jb.ps = PathSet();                                                 // release the per-path containers here, on the worker
      std::vector<uint64_t>().swap(jb.off);
    });
    lap(4);
    for (Job &jb : jobs) {
      if (!jb.log.empty()) fputs(jb.log.c_str(), log);
      if (dbg && !jb.dbg.empty()) fputs(jb.dbg.c_str(), dbg);
      fputs(jb.out.c_str(), out);
      clusters++;
    }
    jobs.clear();
    packed.clear();
    lap(5);
    return MFX_OK;
  };

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
      jobs.emplace_back();
      Job &jb = jobs.back();
      jb.cl = cl; jb.contig = c; jb.rStart = rStart; jb.rEnd = rEnd;
      // upper bound of this cluster's path text: (product of allele counts) x (window + longest ALTs)
      double npaths = 1;
      uint64_t plen = (uint64_t)(rEnd - rStart) + 1;
      for (const Variant *v : cl->vars) {
        npaths *= (double)std::max<size_t>(v->alleles.size(), 1);
        size_t longest = 0;
        for (const std::string *a : v->alleles) longest = std::max(longest, a->size());
        plen += longest;
      }
      est_bytes += (uint64_t)std::min(npaths, 4194304.0) * plen;
      if (est_bytes >= BATCH_BYTES || jobs.size() >= 65536) { rc = flush(); est_bytes = 0; }
      if (rc) break;
    }
  }
  if (rc == MFX_OK) rc = flush();
  lap(5);
  if (timing)
    fprintf(stderr, "[mfx_variants] load+cluster %.2fs  enumerate %.2fs  pack %.2fs  gpu %.2fs  score+select %.2fs  queue+write %.2fs\n",
            t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);
  fclose(out);
  if (dbg && mfx_close(dbgh) && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed", opts->debug_path);
  if (log != stderr) fclose(log);
  if (n_clusters) *n_clusters = clusters;
  return rc;
}

extern "C" int mfx_variants_run(mfx_eval *ev, const char *vcf_path, const char *const *names, const char *const *bases,
                                const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                                const char *out_path, const char *log_path, uint64_t *n_clusters) {
  if (!ev) return mfx_fail(MFX_E_INVAL, "mfx_variants_run: null argument");
  PathValues values = [ev](const char *text, uint64_t len, uint32_t *rv, uint32_t *av) -> int {
    mfx_seq *ps = mfx_seq_upload(ev->device, &text, &len, 1);
    if (!ps) return mfx_last_error_code();
    int r = mfx_dump_values(ev, ps, 0, 0, len, rv, av, nullptr, nullptr);
    mfx_seq_free(ps);
    return r;
  };
  return variants_impl(ev, values, vcf_path, names, bases, lens, ncontigs, opts, out_path, log_path, n_clusters);
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
  return variants_impl(evs[0], values, vcf_path, names, bases, lens, ncontigs, opts, out_path, log_path, n_clusters);
}
