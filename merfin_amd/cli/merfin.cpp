// merfin (MI355X) -- command-line driver with the flag surface of the reference
// driver (src/merfin/merfin.C:79-155, validation :159-181) on top of the
// merfin_amd C ABI.  The pthread sweatShop pipeline (merfin.C:366-414) is
// replaced by: read all contigs -> pack into HBM -> one kernel launch per mode.
//
// Report types: -hist, -dump, -completeness, and the variant modes -filter /
// -polish / -better / -strict / -loose (paths enumerated on the host, every
// path k-mer scored on the GPU).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <libgen.h>
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <unistd.h>
#include <future>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "../../include/merfin_amd.h"
#include "fasta.h"

enum { OP_NONE, OP_HIST, OP_COMPL, OP_DUMP, OP_FILTER, OP_POLISH, OP_BETTER, OP_STRICT, OP_LOOSE };

struct Globals {
  const char *seqName = nullptr, *seqDBname = nullptr, *readDBname = nullptr, *pLookupTable = nullptr;
  const char *vcfName = nullptr, *outName = nullptr, *indexName = nullptr, *convertName = nullptr;
  bool placed = false;                 // -convert -placed: the records sorted by their place in the -hist table (mfx_db_convert_placed)
  double peak = 0, maxMemory = 0;
  uint64_t minV = 0, maxV = ~0ull;
  int threads = 0, reportType = OP_NONE, device = 0;
  std::vector<int> devices;          // -devices: the GPUs of this node one process drives (-hist); [0] == device
  unsigned comb = 15;
  bool nosplit = false, debug = false, skipMissing = false, sharded = false;
  std::vector<uint32_t> copyKmerK;
  std::vector<double> copyKmerP;
};

static void usage(const char *exe) {
  fprintf(stderr,
          "usage: %s <report-type> -sequence <seq.fasta> -readmers <read.meryl> -peak <haploid_peak>\n"
          "          [-prob <lookup_table>] [-seqmers <seq.meryl>] [-vcf <input.vcf>] -output <output>\n\n"
          "  Evaluates every k-mer of <seq.fasta> against the read k-mer database on an MI355X GPU.\n"
          "  Inputs: FASTA/FASTQ, plain or gz/bz2/xz.  K-mer databases: meryl directory, `meryl print`\n"
          "  text, or the flat binary written by mfx_db_write_flat.  k is taken from -readmers.\n\n"
          "    -min m / -max m   ignore read k-mers with value below / above m\n"
          "    -memory m         do not use more than m GB (of HBM) for the k-mer tables\n"
          "    -threads t        accepted for compatibility (host threads)\n"
          "    -peak m           haploid k-mer coverage peak (required except -filter)\n"
          "    -prob file        readK,prob rows; row n overrides -peak for multiplicity n\n"
          "    -seqmers db       assembly k-mer database; default: counted from -sequence on the GPU\n"
          "    -convert db       no report: rewrite the k-mer database <db> (any accepted form) as -output <file> in the flat form\n"
          "                      (sorted k-mers in delta-coded blocks; loads at the speed of the PCIe link)\n"
          "    -placed           with -convert: the records sorted by their PLACE in the table -hist / -dump build (13 <= k <= 31,\n"
          "                      canonical databases): such a database is applied to the table line after line\n"
          "    -device d         HIP device (default 0)\n"
          "    -devices list     several GPUs of this node driven by this one process, e.g. 0-7 or 0,2,5 (-hist: the index is\n"
          "                      built once and copied to the others over xGMI, every GPU evaluates its share of the\n"
          "                      sequence, the histograms are added; -dump and the variant modes: every GPU takes a contiguous\n"
          "                      run of contigs, the outputs are concatenated in order; -completeness uses the first device)\n"
          "    -sharded          with -devices: every GPU keeps only its share of the k-mer table (read databases too large\n"
          "                      for one GPU).  -hist routes every k-mer to the GPU that owns it; -dump and the variant modes\n"
          "                      look every k-mer up in all shards and add the answers; -completeness adds per-shard sums\n"
          "    -index file       cache of the built HBM index: loaded if it exists (the k-mer databases are then not\n"
          "                      read), otherwise written after the build\n\n"
          "  Report types (exactly one):\n"
          "    -hist           0-centred K* histogram to <output>; QV and QV* on stderr\n"
          "    -dump           seqName, seqPos, readK, asmK, K* per k-mer to <output>  [-skipMissing]\n"
          "    -completeness   k-mer completeness from -seqmers (or -sequence) and -readmers\n"
          "    -filter         keep variants (and combinations within k) that minimise missing k-mers -> <output>.filter.vcf\n"
          "    -polish         choose variant combinations by missing k-mers, ties by k* -> <output>.polish.vcf\n"
          "    -better -strict -loose   k*-free variants of -polish -> <output>.filter.vcf\n"
          "                    [-comb N (15)] [-nosplit] [-debug -> <output>.00.debug.gz]\n\n",
          exe);
}

// load_Kmetric, merfin-globals.C:21-62
static bool load_Kmetric(Globals &G) {
  if (!G.pLookupTable) return true;
  fprintf(stderr, "-- Loading probability table '%s'.\n\n", G.pLookupTable);
  // compressedFileReader (merfin-globals.C:34): a table named *.gz / *.bz2 / *.xz is read through its decompressor
  mfx_file pf = mfx_open_reader(G.pLookupTable);
  FILE *f = pf.f;
  if (!f) {
    fprintf(stderr, "ERROR: Probability table (-prob) file '%s' doesn't exist!\n", G.pLookupTable);
    return false;
  }
  char line[4096];
  unsigned lineNum = 0;
  while (fgets(line, sizeof(line), f)) {
    size_t L = strlen(line);
    while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
    std::string keep(line);
    std::vector<char *> w;
    for (char *s = line; *s;) {
      while (*s == ',') *s++ = 0;
      if (!*s) break;
      w.push_back(s);
      while (*s && *s != ',') ++s;
    }
    if (w.size() == 2) {
      unsigned k = (unsigned)strtoul(w[0], nullptr, 10);
      double p = strtod(w[1], nullptr);
      G.copyKmerK.push_back(k);
      G.copyKmerP.push_back(p);
      lineNum++;
      fprintf(stderr, "Copy-number: %u\t\tReadK: %u\tProbability: %f\n", lineNum, k, p);
    } else {
      fprintf(stderr, "Copy-number: invalid line %u:  '%s'\n", lineNum, keep.c_str());
    }
  }
  if (mfx_close(pf) != 0) {
    fprintf(stderr, "ERROR: reading the probability table (-prob) '%s' failed (decompressor or read error).\n", G.pLookupTable);
    return false;
  }
  return true;
}

// Digest of what an index image was built from: every input's path, size and modification time (for a meryl
// directory: of each file in it) plus -min/-max.  A changed assembly -- the iterative polish workflow -- or a
// changed filter gives another digest, and the cached image is rebuilt instead of silently reused.
static void fp_mix(uint64_t &h, const void *p, size_t n) {
  const unsigned char *b = (const unsigned char *)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ULL; }       // FNV-1a
}
static void fp_path(uint64_t &h, const char *path) {
  if (!path) { fp_mix(h, "-", 1); return; }
  fp_mix(h, path, strlen(path) + 1);
  struct stat st;
  if (stat(path, &st) != 0) return;
  if (S_ISDIR(st.st_mode)) {
    std::vector<std::string> names;
    if (DIR *d = opendir(path)) {
      while (struct dirent *e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
      closedir(d);
    }
    std::sort(names.begin(), names.end());
    for (auto &n : names) {
      std::string f = std::string(path) + "/" + n;
      struct stat fs;
      if (stat(f.c_str(), &fs) != 0) continue;
      fp_mix(h, n.c_str(), n.size() + 1);
      uint64_t v[3] = {(uint64_t)fs.st_size, (uint64_t)fs.st_mtim.tv_sec, (uint64_t)fs.st_mtim.tv_nsec};
      fp_mix(h, v, sizeof(v));
    }
  } else {
    uint64_t v[3] = {(uint64_t)st.st_size, (uint64_t)st.st_mtim.tv_sec, (uint64_t)st.st_mtim.tv_nsec};
    fp_mix(h, v, sizeof(v));
  }
}
static uint64_t input_fingerprint(const Globals &G, bool seqOnly) {
  uint64_t h = 0xcbf29ce484222325ULL;
  fp_path(h, G.readDBname);
  fp_path(h, G.seqDBname);
  if (!G.seqDBname || seqOnly) fp_path(h, G.seqName);   // the assembly side is counted from -sequence / the table holds ITS k-mers only
  if (seqOnly) fp_mix(h, "seq-only", 9);
  uint64_t v[2] = {G.minV, G.maxV};
  fp_mix(h, v, sizeof(v));
  return h ? h : 1;
}

static void print_hist(const std::vector<SeqRecord> &recs, const mfx_hist_result &r, int k, const char *outName, bool *ok) {
  uint64_t cum = 0;
  for (size_t c = 0; c < recs.size(); ++c) {   // outputHistogram's per-sequence line, in input order
    cum += r.contig_kmissing[c];
    fprintf(stderr, "%s\t%lu\t%lu\t%lu\t%.2f\n", recs[c].name.c_str(), (unsigned long)r.contig_kmissing[c], (unsigned long)cum,
            (unsigned long)r.contig_kasm[c], mfx_histoQV((double)r.contig_kmissing[c], (double)r.contig_kasm[c], k));
  }
  *ok = mfx_hist_report(&r, k, outName, "-") == 0;
}

static void print_completeness(const double *t64, const double *u64) {
  double total = 0, undrc = 0;
  for (int ii = 0; ii < 64; ii++) {                     // merfin-completeness.C:119-120, here in piece order
    fprintf(stderr, "thread %2u total %12.2f underc %15.5f completeness %0.8f\n", ii, t64[ii], u64[ii], 1.0 - u64[ii] / t64[ii]);
    total += t64[ii];
    undrc += u64[ii];
  }
  fprintf(stderr, "\n");
  fprintf(stderr, "TOTAL readK:   %15.2f\n", total);
  fprintf(stderr, "TOTAL undrcpy:    %15.5f\n", undrc);
  fprintf(stderr, "COMPLETENESS:             %0.5f\n", 1.0 - undrc / total);
}

// Upper bound of the bases in a sequence file WITHOUT reading it (0 = unknown): a plain file cannot hold more bases
// than bytes; a gzip file records its uncompressed size modulo 2^32 in its last four bytes (RFC 1952 ISIZE), and DNA
// text compresses 2-8x, which settles the multiple of 4 GiB.  Lets the k-mer table be sized -- and the read database
// be loaded -- while the sequence file is still being read.
static uint64_t bases_upper_bound(const char *path) {
  struct stat st;
  if (stat(path, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size <= 0) return 0;
  const uint64_t size = (uint64_t)st.st_size;
  const std::string p(path);
  auto ends = [&](const char *suf) { const size_t n = strlen(suf); return p.size() >= n && p.compare(p.size() - n, n, suf) == 0; };
  if (ends(".bz2") || ends(".xz") || ends(".zst")) return 0;
  if (!ends(".gz")) return size;
  if (size < 18) return 0;
  FILE *f = fopen(path, "rb");
  if (!f) return 0;
  unsigned char t[4];
  const bool ok = fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4;
  fclose(f);
  if (!ok) return 0;
  const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
  // the largest isize + n * 2^32 that is at most 8x the compressed size (a file of several gzip members only records
  // the last member's size: then the estimate may be LOW, so it must also be at least 2x the compressed size to be used)
  uint64_t best = 0;
  for (uint64_t n = 0; n < 4096; ++n) {
    const uint64_t cand = isize + (n << 32);
    if (cand > 8 * size + (1ull << 20)) break;
    best = cand;
  }
  if (best < 2 * size) return 0;
  return best;
}

// -hist and -dump with -sharded: PARTS of the assembly, one per slot (round 4).  They only ever ask the lookup tables for the
// k-mers of -sequence (merfin-histogram.C:54-64, merfin-dump.C:44-61), so a slot needs the k-mers of the contigs IT evaluates
// and nothing else: the contigs are dealt to the slots (balanced by bases), slot d claims its contigs' k-mers into a
// sequence-only index, counts them over the whole assembly (or takes them from -seqmers), the databases -- decoded once, sent
// to every slot -- update them, and every slot evaluates its contigs on its own device: no k-mer is ever exchanged, whatever
// the size of the read database (config 5: each of the 8 GPUs keeps the 1/8 of a 2 x 10^10-k-mer database that its contigs hit).
// Returns -1 when a database is not canonical (the caller falls back to the hash-sharded full tables).
static int run_parts(const Globals &G, int k, const std::vector<SeqRecord> &recs, const std::vector<const char *> &bases,
                     const std::vector<uint64_t> &lens) {
  const uint32_t N = (uint32_t)G.devices.size();
  const uint32_t NC = (uint32_t)recs.size();
  // the contigs of every slot: largest first to the least loaded slot; ascending numbers inside a slot
  std::vector<std::vector<uint32_t>> ids(N);
  {
    std::vector<uint32_t> order(NC);
    for (uint32_t c = 0; c < NC; ++c) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return lens[a] > lens[b]; });
    std::vector<uint64_t> load(N, 0);
    for (uint32_t c : order) {
      const uint32_t d = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
      ids[d].push_back(c);
      load[d] += lens[c] + 1;
    }
    for (auto &v : ids) std::sort(v.begin(), v.end());
  }
  std::vector<mfx_index *> ixs(N, nullptr);
  std::vector<mfx_seq *> own(N, nullptr), whole(N, nullptr);
  std::vector<mfx_eval *> evs(N, nullptr);
  mfx_kparams kp{G.peak, (uint32_t)G.copyKmerK.size(), G.copyKmerK.data(), G.copyKmerP.data()};
  int rc = 0;
  auto fail = [&](const char *what) { fprintf(stderr, "ERROR: %s: %s\n", what, mfx_last_error()); rc = 1; };
  auto release = [&]() {
    for (uint32_t d = 0; d < N; ++d) {
      if (evs[d]) mfx_eval_free(evs[d]);
      if (ixs[d]) mfx_index_free(ixs[d]);
      if (own[d]) mfx_seq_free(own[d]);
      bool shared = false;
      for (uint32_t e = 0; e < d; ++e) if (whole[e] == whole[d]) shared = true;
      if (whole[d] && !shared) mfx_seq_free(whole[d]);
      evs[d] = nullptr; ixs[d] = nullptr; own[d] = nullptr; whole[d] = nullptr;
    }
  };
  for (uint32_t d = 0; d < N && !rc; ++d) {
    std::vector<const char *> b2;
    std::vector<uint64_t> l2;
    uint64_t nb = 0;
    for (uint32_t c : ids[d]) { b2.push_back(bases[c]); l2.push_back(lens[c]); nb += lens[c]; }
    fprintf(stderr, "-- Part %u of %u on device %d: %zu sequences, %lu bases.\n", d, N, G.devices[d], ids[d].size(), (unsigned long)nb);
    own[d] = mfx_seq_upload(G.devices[d], b2.data(), l2.data(), (uint32_t)b2.size());
    if (!own[d]) { fail("uploading sequences"); break; }
    ixs[d] = mfx_index_create_for_seq(k, nb + 1024, G.maxMemory, G.devices[d]);
    if (!ixs[d]) { fprintf(stderr, "\n%s\n\n", mfx_last_error()); rc = 1; break; }
    if (mfx_index_claim_seq(ixs[d], own[d], nullptr)) { fail("claiming sequence k-mers"); break; }
    if (!G.seqDBname) {
      // replaces `meryl count k=.. <seq> output <seq>.meryl` (merfin-globals.C:182-186): every contig of the assembly counts
      for (uint32_t e = 0; e < d; ++e) if (G.devices[e] == G.devices[d]) { whole[d] = whole[e]; break; }
      if (!whole[d]) whole[d] = mfx_seq_upload(G.devices[d], bases.data(), lens.data(), NC);
      if (!whole[d]) { fail("uploading sequences"); break; }
      if (mfx_index_count_claimed(ixs[d], whole[d], nullptr)) { fail("counting sequence k-mers"); break; }
    }
  }
  for (uint32_t d = 0; d < N; ++d) {                              // the whole assembly was only needed for the counts
    bool shared = false;
    for (uint32_t e = 0; e < d; ++e) if (whole[e] == whole[d]) shared = true;
    if (whole[d] && !shared) mfx_seq_free(whole[d]);
  }
  std::fill(whole.begin(), whole.end(), nullptr);
  int lrc = 0;
  if (!rc && G.seqDBname) {
    fprintf(stderr, "-- Loading kmers from '%s' into the %u parts.\n", G.seqDBname, N);
    lrc = mfx_index_load_db_multi(ixs.data(), N, G.seqDBname, 1, 0, ~0ull);
    if (lrc && lrc != MFX_E_NONCANON) fail("loading -seqmers");
  }
  if (!rc && !lrc) {
    fprintf(stderr, "-- Loading kmers from '%s' into the %u parts.\n", G.readDBname, N);
    lrc = mfx_index_load_db_multi(ixs.data(), N, G.readDBname, 0, G.minV, G.maxV);
    if (lrc && lrc != MFX_E_NONCANON) fail("loading -readmers");
  }
  if (!rc && lrc == MFX_E_NONCANON) {
    fprintf(stderr, "-- A k-mer database is not canonical; building the sharded full tables instead.\n");
    release();
    return -1;
  }
  for (uint32_t d = 0; d < N && !rc; ++d) {
    evs[d] = mfx_eval_create(ixs[d], &kp, 0);
    if (!evs[d]) { fail("creating evaluator"); break; }
  }
  std::vector<const uint32_t *> idp(N);
  for (uint32_t d = 0; d < N; ++d) idp[d] = ids[d].data();
  if (!rc && (G.reportType == OP_HIST || G.skipMissing)) {
    mfx_hist_result r;
    if (G.reportType == OP_HIST) fprintf(stderr, "-- Generate histogram of the k* metric to '%s' on %u devices (one part of the sequences each).\n", G.outName, N);
    else fprintf(stderr, "-- Dump per-base k* metric to '%s' on %u devices (one part of the sequences each).\n", G.outName, N);
    if (mfx_hist_run_parts(evs.data(), own.data(), idp.data(), N, NC, &r)) fail("-hist over the parts");
    else if (G.reportType == OP_HIST) {
      bool ok = true;
      print_hist(recs, r, k, G.outName, &ok);
      if (!ok) fail("writing histogram");
      mfx_hist_result_free(&r);
    } else {                               // -dump -skipMissing (merfin-dump.C:34,81-87): no dump file, only the per-contig counts
      uint64_t cumMissing = 0, cumAsm = 0;
      for (size_t c = 0; c < recs.size(); ++c) {
        cumMissing += r.contig_kmissing[c];
        cumAsm += r.contig_kasm[c];
        fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)r.contig_kmissing[c], (unsigned long)cumMissing, (unsigned long)cumAsm);
      }
      mfx_hist_result_free(&r);
    }
  } else if (!rc) {
    fprintf(stderr, "-- Dump per-base k* metric to '%s' on %u devices (one part of the sequences each).\n", G.outName, N);
    // every contig by the slot that holds it, in input order (merfin.C:384: -dump writes in order)
    std::vector<std::pair<uint32_t, uint32_t>> where(NC);           // contig -> (slot, number inside the slot)
    for (uint32_t d = 0; d < N; ++d) for (uint32_t i = 0; i < ids[d].size(); ++i) where[ids[d][i]] = {d, i};
    uint64_t cumMissing = 0, cumAsm = 0;
    for (uint32_t c = 0; c < NC && !rc; ++c) {
      uint64_t ka = 0, km = 0;
      const uint32_t d = where[c].first;
      if (mfx_dump_contig(evs[d], own[d], where[c].second, recs[c].name.c_str(), G.outName, c > 0, &ka, &km)) { fail("-dump over the parts"); break; }
      cumMissing += km;
      cumAsm += ka;
      fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)km, (unsigned long)cumMissing, (unsigned long)cumAsm);
    }
    if (recs.empty()) { FILE *f = fopen(G.outName, "w"); if (f) fclose(f); }
  }
  release();
  if (!rc) fprintf(stderr, "Bye!\n");
  return rc;
}

// Every report over an index SHARDED across the devices of -devices (read databases beyond one GPU, BASELINE
// config 5): slot d keeps the k-mers it owns (mfx_index_set_shard), loads skip foreign k-mers, -hist routes every k-mer
// to its owner (mfx_hist_run_sharded), -completeness adds the per-piece sums of the shards in piece order
// (merfin-completeness.C:117-123; the sums are integer-valued, so the split is exact), -dump and the variant modes
// look every k-mer up in all shards and add the answers (mfx_dump_contig_sharded, mfx_variants_run_sharded).  The
// databases are decoded once: every batch is sent to all shards (mfx_index_load_db_multi).
static int run_sharded(const Globals &G, int k, const mfx_db_info &rdb, const mfx_db_info &adb, const std::vector<SeqRecord> &recs,
                       const std::vector<const char *> &bases, const std::vector<uint64_t> &lens, uint64_t totalBases) {
  const uint32_t N = (uint32_t)G.devices.size();
  std::vector<mfx_index *> ixs(N, nullptr);
  std::vector<mfx_seq *> sqs(N, nullptr);
  std::vector<mfx_eval *> evs(N, nullptr);
  std::vector<mfx_router *> rts(N, nullptr);
  mfx_kparams kp{G.peak, (uint32_t)G.copyKmerK.size(), G.copyKmerK.data(), G.copyKmerP.data()};
  const uint64_t whole = rdb.n_kmers + (G.seqDBname ? adb.n_kmers : totalBases) + 1024;
  const uint64_t capacity = (uint64_t)((double)whole / N * 1.15) + 1024;      // owners are hash-balanced
  int rc = 0;
  auto fail = [&](const char *what) { fprintf(stderr, "ERROR: %s: %s\n", what, mfx_last_error()); rc = 1; };
  for (uint32_t d = 0; d < N && !rc; ++d) {
    uint32_t same = d;
    for (uint32_t e = 0; e < d; ++e) if (G.devices[e] == G.devices[d]) { same = e; break; }
    if (!recs.empty() || G.seqName) {
      sqs[d] = same < d ? sqs[same] : mfx_seq_upload(G.devices[d], bases.data(), lens.data(), (uint32_t)recs.size());
      if (!sqs[d]) { fail("uploading sequences"); break; }
    }
    fprintf(stderr, "-- Shard %u of %u on device %d.\n", d, N, G.devices[d]);
    ixs[d] = mfx_index_create(k, capacity, G.maxMemory, G.devices[d]);
    if (!ixs[d]) { fprintf(stderr, "\n%s\n\n", mfx_last_error()); rc = 1; break; }
    if (mfx_index_set_shard(ixs[d], d, N)) { fail("creating shard"); break; }
  }
  // one pass over each database feeds every shard (each keeps the k-mers it owns)
  if (!rc) {
    fprintf(stderr, "-- Loading kmers from '%s' into the %u shards.\n", G.readDBname, N);
    if (mfx_index_load_db_multi(ixs.data(), N, G.readDBname, 0, G.minV, G.maxV)) fail("loading -readmers");
  }
  if (!rc && G.seqDBname) {
    fprintf(stderr, "-- Loading kmers from '%s' into the %u shards.\n", G.seqDBname, N);
    if (mfx_index_load_db_multi(ixs.data(), N, G.seqDBname, 1, 0, ~0ull)) fail("loading -seqmers");
  }
  for (uint32_t d = 0; d < N && !rc; ++d) {
    if (!G.seqDBname && mfx_index_count_asm(ixs[d], sqs[d], nullptr)) { fail("counting sequence k-mers"); break; }
    evs[d] = mfx_eval_create(ixs[d], &kp, 0);
    if (!evs[d]) { fail("creating evaluator"); break; }
    if (G.reportType == OP_HIST || (G.reportType == OP_DUMP && G.skipMissing)) {
      const uint64_t nt = mfx_seq_num_tiles(sqs[d]);
      rts[d] = mfx_router_create(ixs[d], N, (uint32_t)std::min<uint64_t>(16384, std::max<uint64_t>(1, nt)));
      if (!rts[d]) { fail("creating router"); break; }
    }
  }
  if (!rc && G.reportType == OP_HIST) {
    fprintf(stderr, "-- Generate histogram of the k* metric to '%s' on %u devices (sharded index).\n", G.outName, N);
    mfx_hist_result r;
    if (mfx_hist_run_sharded(evs.data(), rts.data(), sqs.data(), N, &r)) fail("-hist over the sharded index");
    else {
      bool ok = true;
      print_hist(recs, r, k, G.outName, &ok);
      if (!ok) fail("writing histogram");
      mfx_hist_result_free(&r);
    }
  } else if (!rc && G.reportType == OP_DUMP) {
    fprintf(stderr, "-- Dump per-base k* metric to '%s' on %u devices (sharded index).\n", G.outName, N);
    uint64_t cumMissing = 0, cumAsm = 0;
    if (G.skipMissing) {                 // merfin-dump.C:34,81-87: no dump file, only the per-contig counts
      mfx_hist_result r;
      if (mfx_hist_run_sharded(evs.data(), rts.data(), sqs.data(), N, &r)) fail("-dump -skipMissing over the sharded index");
      else {
        for (size_t c = 0; c < recs.size(); ++c) {
          cumMissing += r.contig_kmissing[c];
          cumAsm += r.contig_kasm[c];
          fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)r.contig_kmissing[c], (unsigned long)cumMissing, (unsigned long)cumAsm);
        }
        mfx_hist_result_free(&r);
      }
    } else {
      for (size_t c = 0; c < recs.size() && !rc; ++c) {
        uint64_t ka = 0, km = 0;
        if (mfx_dump_contig_sharded(evs.data(), sqs.data(), N, (uint32_t)c, recs[c].name.c_str(), G.outName, c > 0, &ka, &km)) { fail("-dump over the sharded index"); break; }
        cumMissing += km;
        cumAsm += ka;
        fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)km, (unsigned long)cumMissing, (unsigned long)cumAsm);
      }
      if (recs.empty()) { FILE *f = fopen(G.outName, "w"); if (f) fclose(f); }
    }
  } else if (!rc && G.reportType >= OP_FILTER) {
    fprintf(stderr, "-- Opening vcf file '%s'.\n", G.vcfName);
    fprintf(stderr, "-- Generate variant mers and score them on %u devices (sharded index).\n", N);
    const std::string outName = std::string(G.outName) + (G.reportType == OP_POLISH ? ".polish.vcf" : ".filter.vcf");   // merfin-variants.C:324-327
    const std::string dbgName = std::string(G.outName) + ".00.debug.gz";
    std::vector<const char *> names(recs.size());
    for (size_t c = 0; c < recs.size(); ++c) names[c] = recs[c].name.c_str();
    mfx_variant_opts vo;
    vo.mode = G.reportType;
    vo.comb = G.comb;
    vo.nosplit = G.nosplit ? 1 : 0;
    vo.debug_path = G.debug ? dbgName.c_str() : nullptr;
    uint64_t ncl = 0;
    if (mfx_variants_run_sharded(evs.data(), N, G.vcfName, names.data(), bases.data(), lens.data(), (uint32_t)recs.size(), &vo, outName.c_str(), nullptr, &ncl))
      fail("variant scoring over the sharded index");
  } else if (!rc) {
    fprintf(stderr, "-- Compute completeness on %u devices (sharded index).\n", N);
    double t64[64] = {0}, u64[64] = {0};
    for (uint32_t d = 0; d < N && !rc; ++d) {
      double t[64], u[64];
      if (mfx_completeness_pieces(evs[d], t, u)) { fail("-completeness"); break; }
      for (int i = 0; i < 64; ++i) { t64[i] += t[i]; u64[i] += u[i]; }
    }
    if (!rc) print_completeness(t64, u64);
  }
  for (uint32_t d = 0; d < N; ++d) {
    if (rts[d]) mfx_router_free(rts[d]);
    if (evs[d]) mfx_eval_free(evs[d]);
    if (ixs[d]) mfx_index_free(ixs[d]);
    bool shared = false;
    for (uint32_t e = 0; e < d; ++e) if (G.devices[e] == G.devices[d]) shared = true;
    if (sqs[d] && !shared) mfx_seq_free(sqs[d]);
  }
  if (!rc) fprintf(stderr, "Bye!\n");
  return rc;
}

// One evaluation context per entry of -devices: slot 0 is the index / sequence / evaluator the run built, the others
// are replicas made by peer copy (a device named twice shares table and sequence; every slot has its own evaluator).
struct Slots {
  std::vector<mfx_index *> ixs;
  std::vector<mfx_seq *> sqs;
  std::vector<mfx_eval *> evs;
  std::vector<int> devs;
  bool make(const Globals &G, mfx_index *ix, mfx_seq *seq, mfx_eval *ev, const mfx_kparams *kp) {
    const size_t N = G.devices.size();
    devs = G.devices;
    ixs.assign(N, nullptr); sqs.assign(N, nullptr); evs.assign(N, nullptr);
    ixs[0] = ix; sqs[0] = seq; evs[0] = ev;
    // the devices that need a copy (a device named twice shares table and sequence): all at once, by a doubling tree
    std::vector<int> fresh;
    std::vector<size_t> first(N, 0);
    for (size_t d = 0; d < N; ++d) {
      first[d] = d;
      for (size_t e = 0; e < d; ++e) if (devs[e] == devs[d]) { first[d] = e; break; }
      if (first[d] == d && d > 0) fresh.push_back(devs[d]);
    }
    std::vector<mfx_index *> rix(fresh.size(), nullptr);
    std::vector<mfx_seq *> rsq(fresh.size(), nullptr);
    if (!fresh.empty()) {
      if (mfx_index_replicate_many(ix, fresh.data(), (uint32_t)fresh.size(), rix.data())) return false;
      if (seq && mfx_seq_replicate_many(seq, fresh.data(), (uint32_t)fresh.size(), rsq.data())) {
        for (auto *x : rix) mfx_index_free(x);
        return false;
      }
    }
    for (size_t d = 1, f = 0; d < N; ++d) {
      if (first[d] == d) { ixs[d] = rix[f]; sqs[d] = rsq[f]; ++f; }
      else { ixs[d] = ixs[first[d]]; sqs[d] = sqs[first[d]]; }
    }
    for (size_t d = 1; d < N; ++d) {
      evs[d] = mfx_eval_create(ixs[d], kp, 0);
      if (!evs[d]) { release(); return false; }
    }
    return true;
  }
  void release() {                       // slot 0 belongs to the caller
    for (size_t d = 1; d < ixs.size(); ++d) {
      if (evs[d]) mfx_eval_free(evs[d]);
      bool shared = false;
      for (size_t e = 0; e < d; ++e) if (devs[e] == devs[d]) shared = true;
      if (!shared) { if (sqs[d]) mfx_seq_free(sqs[d]); if (ixs[d]) mfx_index_free(ixs[d]); }
    }
    evs.clear(); ixs.clear(); sqs.clear();
  }
};

// contiguous split of the contigs (input order) into `parts` runs of about equal weight; every slot works on its run
// and the outputs are concatenated in slot order -- the reference's SLURM-array recipe (scripts/parallel1/merfin.sh:68-85)
static std::vector<std::pair<size_t, size_t>> contig_partition(const std::vector<double> &w, size_t parts) {
  const size_t n = w.size();
  std::vector<double> cum(n + 1, 0.0);
  for (size_t i = 0; i < n; ++i) cum[i + 1] = cum[i] + w[i];
  std::vector<size_t> cuts(1, 0);
  for (size_t r = 1; r < parts; ++r) {
    size_t c = cum[n] > 0 ? (size_t)(std::lower_bound(cum.begin(), cum.end(), cum[n] * (double)r / (double)parts) - cum.begin()) : n * r / parts;
    cuts.push_back(std::min(n, std::max(cuts.back(), c)));
  }
  cuts.push_back(n);
  std::vector<std::pair<size_t, size_t>> out;
  for (size_t r = 0; r < parts; ++r) out.emplace_back(cuts[r], cuts[r + 1]);
  return out;
}

// parts -> out (slot order); skip_header: the '#' lines of every part but the first are dropped (VCF)
static bool concat_parts(const std::string &out, const std::vector<std::string> &parts, bool skip_header) {
  FILE *o = fopen(out.c_str(), "wb");
  if (!o) return false;
  std::vector<char> buf(1 << 22);
  bool ok = true;
  for (size_t i = 0; i < parts.size() && ok; ++i) {
    FILE *f = fopen(parts[i].c_str(), "rb");
    if (!f) continue;                                         // a slot without contigs wrote nothing
    bool at_line_start = true, in_header = false;
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0 && ok) {
      if (!(skip_header && i > 0)) { ok = fwrite(buf.data(), 1, n, o) == n; continue; }
      size_t b = 0;                                           // copy everything except lines starting with '#'
      for (size_t j = 0; j < n; ++j) {
        if (at_line_start) {
          if (!in_header && j > b) ok = ok && fwrite(buf.data() + b, 1, j - b, o) == j - b;
          in_header = buf[j] == '#';
          b = j;
          at_line_start = false;
        }
        if (buf[j] == '\n') {
          at_line_start = true;
          if (in_header) b = j + 1;
        }
      }
      if (!in_header && n > b) ok = ok && fwrite(buf.data() + b, 1, n - b, o) == n - b;
      if (in_header) b = n;
    }
    fclose(f);
    remove(parts[i].c_str());
  }
  return fclose(o) == 0 && ok;
}

#define DIE_MFX(what)                                                       \
  do {                                                                      \
    fprintf(stderr, "ERROR: %s: %s\n", what, mfx_last_error());             \
    return 1;                                                               \
  } while (0)

int main(int argc, char **argv) {
  // MFX_CLI_TIMING=3: wall-clock stamps (seconds since the epoch) at main / after the device check / at the end, so that a
  // caller who timed the whole process can tell start-up and tear-down from the phases (diagnostics)
  auto epoch = [] { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
  const double stamp_main = epoch();
  Globals G;
  std::vector<std::string> err;
  for (int arg = 1; arg < argc; arg++) {
    auto is = [&](const char *f) { return strcmp(argv[arg], f) == 0; };
    auto val = [&]() -> const char * { return arg + 1 < argc ? argv[++arg] : ""; };
    if (is("-sequence")) G.seqName = val();
    else if (is("-seqmers")) G.seqDBname = val();
    else if (is("-readmers")) G.readDBname = val();
    else if (is("-peak")) G.peak = strtod(val(), nullptr);
    else if (is("-prob")) G.pLookupTable = val();
    else if (is("-vcf")) G.vcfName = val();
    else if (is("-output")) G.outName = val();
    else if (is("-min")) G.minV = strtoull(val(), nullptr, 10);
    else if (is("-max")) G.maxV = strtoull(val(), nullptr, 10);
    else if (is("-threads")) G.threads = atoi(val());
    else if (is("-memory")) G.maxMemory = strtod(val(), nullptr);
    else if (is("-device")) G.device = atoi(val());
    else if (is("-devices")) {
      // "0-7", "0,2,5", "0-3,6": a device may be named twice (two evaluation contexts on one GPU)
      std::string spec = val();
      bool ok = !spec.empty();
      for (size_t i = 0; ok && i < spec.size();) {
        char *e = nullptr;
        long a = strtol(spec.c_str() + i, &e, 10), b = a;
        ok = e != spec.c_str() + i && a >= 0;
        i = (size_t)(e - spec.c_str());
        if (ok && i < spec.size() && spec[i] == '-') {
          const char *s2 = spec.c_str() + i + 1;
          b = strtol(s2, &e, 10);
          ok = e != s2 && b >= a;
          i = (size_t)(e - spec.c_str());
        }
        for (long d = a; ok && d <= b && d < 1024; ++d) G.devices.push_back((int)d);
        if (ok && i < spec.size()) { ok = spec[i] == ','; ++i; }
      }
      if (!ok || G.devices.empty()) { G.devices.clear(); err.push_back(std::string("Invalid device list '") + spec + "' (-devices 0-7 or 0,2,5).\n"); }
    }
    else if (is("-index")) G.indexName = val();
    else if (is("-sharded")) G.sharded = true;
    else if (is("-convert")) G.convertName = val();
    else if (is("-placed")) G.placed = true;
    else if (is("-nosplit")) G.nosplit = true;
    else if (is("-filter")) G.reportType = OP_FILTER;
    else if (is("-better")) G.reportType = OP_BETTER;
    else if (is("-strict")) G.reportType = OP_STRICT;
    else if (is("-loose")) { fprintf(stderr, "*EXPERIMENTAL* Running in -loose mode\n"); G.reportType = OP_LOOSE; }
    else if (is("-polish")) G.reportType = OP_POLISH;
    else if (is("-hist")) G.reportType = OP_HIST;
    else if (is("-dump")) G.reportType = OP_DUMP;
    else if (is("-skipMissing")) G.skipMissing = true;
    else if (is("-completeness")) G.reportType = OP_COMPL;
    else if (is("-comb")) G.comb = (unsigned)strtoul(val(), nullptr, 10);
    else if (is("-debug")) G.debug = true;
    else err.push_back(std::string("Unknown option '") + argv[arg] + "'.\n");
  }

  if (G.convertName && err.empty()) {
    // merfin -convert <db> -output <file>: a database in any accepted form rewritten as this program's flat form (sorted
    // k-mers in delta-coded blocks), on the host -- no report, no device
    if (!G.outName) { fprintf(stderr, "No output (-output) supplied.\n"); return 1; }
    fprintf(stderr, "-- Converting '%s' to '%s'.\n", G.convertName, G.outName);
    uint64_t n = 0;
    if (G.placed ? mfx_db_convert_placed(G.convertName, G.outName, &n) : mfx_db_convert(G.convertName, G.outName, &n)) {
      fprintf(stderr, "ERROR: -convert: %s\n", mfx_last_error());
      return 1;
    }
    struct stat st;
    fprintf(stderr, "-- Wrote %lu k-mers", (unsigned long)n);
    if (stat(G.outName, &st) == 0 && n) fprintf(stderr, " in %.2f GB (%.2f bytes per k-mer)", st.st_size / 1e9, (double)st.st_size / (double)n);
    fprintf(stderr, ".\nBye!\n");
    return 0;
  }

  // merfin.C:159-181
  const bool variantMode = G.reportType == OP_POLISH || G.reportType == OP_FILTER || G.reportType == OP_BETTER ||
                           G.reportType == OP_STRICT || G.reportType == OP_LOOSE;
  if (G.reportType != OP_COMPL) {
    if (!G.seqName) err.push_back("No input sequences (-sequence) supplied.\n");
    if (!G.outName) err.push_back("No output (-output) supplied.\n");
  }
  if (variantMode && !G.vcfName) err.push_back("No variant call input (-vcf) supplied; mandatory for -filter or -polish.\n");
  if (G.reportType != OP_FILTER && G.peak == 0) err.push_back("No haploid peak (-peak) supplied.\n");
  if (G.reportType == OP_COMPL && !G.seqName && !G.seqDBname)
    err.push_back("No sequence meryl database (-seqmers) nor sequence (-sequence) supplied.\n");
  if (G.reportType == OP_NONE) err.push_back("No report type (-filter, -polish, -hist, -dump, -completeness) supplied.\n");
  if (!G.readDBname) err.push_back("No read meryl database (-readmers) supplied.\n");
  if (!err.empty()) {
    usage(argv[0]);
    for (auto &e : err) fputs(e.c_str(), stderr);
    return 1;
  }
  if (!G.devices.empty()) G.device = G.devices[0];
  else G.devices.push_back(G.device);
  // load_Kmetric first (merfin-globals.C:21-62 runs before the databases are opened; host only -- a compressed table's decompressor has
  // come and gone before the HIP runtime starts)
  if (!load_Kmetric(G)) return 1;
  // (the device check stays on this thread, before anything else is started: the HIP runtime coming up on a thread of its own
  // while this one spawns a decompressor for -sequence was seen to come up with no device)
  for (int d : G.devices)
    if (mfx_device_count() <= d) {
      fprintf(stderr, "ERROR: HIP device %d not available (%d visible). This program has no CPU path.\n", d, mfx_device_count());
      return 1;
    }

  const double stamp_devices = epoch();
  // MFX_CLI_TIMING=1: wall time per phase on stderr at exit (diagnostics; not part of merfin's output)
  const bool timing = getenv("MFX_CLI_TIMING") && atoi(getenv("MFX_CLI_TIMING"));
  auto t_last = std::chrono::steady_clock::now();
  std::vector<std::pair<const char *, double>> phases;
  auto lap = [&](const char *what) {
    auto t = std::chrono::steady_clock::now();
    phases.emplace_back(what, std::chrono::duration<double>(t - t_last).count());
    t_last = t;
  };
  // MFX_CLI_TIMING=2: the steps of the index build as well
  auto t_sub = t_last;
  std::vector<std::pair<const char *, double>> steps;
  auto step = [&](const char *what) {
    auto t = std::chrono::steady_clock::now();
    steps.emplace_back(what, std::chrono::duration<double>(t - t_sub).count());
    t_sub = t;
  };

  // load_Kmers, merfin-globals.C:114-163: the read DB defines k
  mfx_db_info rdb, adb;
  if (mfx_db_probe(G.readDBname, &rdb)) DIE_MFX("opening -readmers");
  const int k = rdb.k;
  memset(&adb, 0, sizeof(adb));
  if (G.seqDBname) {
    if (mfx_db_probe(G.seqDBname, &adb)) DIE_MFX("opening -seqmers");
    if (adb.k != k) { fprintf(stderr, "ERROR: -seqmers holds %d-mers but -readmers holds %d-mers.\n", adb.k, k); return 1; }
  }

  if (rdb.format == MFX_DB_MERYL || (G.seqDBname && adb.format == MFX_DB_MERYL))
    fprintf(stderr, "-- NOTE: meryl database directories are decoded from a recalled description of the format; it has not been\n"
                    "--       checked against upstream meryl.  The decoder cross-checks every database against its own index\n"
                    "--       statistics; `meryl print` text is the verified interchange form.  To validate the decoder on this\n"
                    "--       database: python tools/meryl_conformance.py <db.meryl> <output of `meryl print db.meryl`>\n"
                    "--       (= MFX_REAL_MERYL_DB / MFX_REAL_MERYL_PRINT for tests/test_gpu_meryl_conformance.py).\n");
  // `merfin -hist / -dump -sequence s -readmers db` on one device with a delta-coded database and no -seqmers: the database's bytes start
  // moving into device memory NOW, on a thread of their own (mfx_db_stage_begin) -- under the FASTA read, the sequence upload, the
  // table's allocation and the kernel that claims the sequence's k-mers.  Any other database form, too little free memory,
  // MFX_DB_STAGE=0: no stage, the build reads the database when it gets there.
  bool pathOnly = false;                                            // the variant modes on the path-only index of their call set (decided below)
  mfx_db_stage *stage = nullptr, *stageAsm = nullptr;              // (stageAsm: -seqmers of the path-only index)
  struct StageGuard { mfx_db_stage *&s; ~StageGuard() { if (s) mfx_db_stage_free(s); s = nullptr; } } stageGuard{stage}, stageAsmGuard{stageAsm};
  // (MFX_CLI_STAGE_FIRST=0: the stage begins after the device is warmed up.  Measured, profiles/r05_stager_diag.txt: the warm-up then takes
  // 0.06 instead of 0.25 s, but the 0.15-0.2 s that the process's first allocations / queue / kernel cost move into the stager's start, the
  // claim kernel is launched at the same 0.35 s, and with stager, FASTA reader and encoder all at full speed in the first 0.3 s the process
  // runs into the CPU quota of a 16-core box in some runs: 1.09-1.13 s or 1.27-1.29 s against 1.12-1.16 s this way)
  const bool stageFirst = !(getenv("MFX_CLI_STAGE_FIRST") && atoi(getenv("MFX_CLI_STAGE_FIRST")) == 0);
  auto begin_stage = [&]() {
    const bool histLike = (G.reportType == OP_HIST || G.reportType == OP_DUMP) && !G.sharded && k <= 31 && G.seqName && !G.seqDBname && !G.indexName &&
                          G.devices.size() == 1 && !(getenv("MFX_CLI_FULL_INDEX") && atoi(getenv("MFX_CLI_FULL_INDEX")));
    if (histLike && rdb.format == MFX_DB_FLAT) stage = mfx_db_stage_begin(G.readDBname, G.device);
  };
  // the path-only index of the variant modes: both databases move while the call set is prepared and its paths are claimed -- from the moment
  // the sequences are in (under the FASTA reader the stagers' threads cost it 0.2 s of a 3 Gb file on a 16-core quota)
  auto begin_path_stages = [&]() {
    if (pathOnly && rdb.format == MFX_DB_FLAT) stage = mfx_db_stage_begin(G.readDBname, G.device);
    if (pathOnly && G.seqDBname && adb.format == MFX_DB_FLAT) stageAsm = mfx_db_stage_begin(G.seqDBname, G.device);
  };
  if (stageFirst) begin_stage();
  lap("probe k-mer databases");
  // sequences (load_Sequence, merfin-globals.C:165-197; loadSequence, merfin.C:30-53).  The file is read (and, for
  // .gz/.bz2/.xz, decompressed) by its own thread from here on; with -seqmers nothing needs the sequence before the
  // evaluation, so the read then runs under the index build and is only waited for afterwards.
  std::vector<SeqRecord> recs;
  std::vector<const char *> bases;
  std::vector<uint64_t> lens;
  uint64_t totalBases = 0;
  std::thread seqReader;
  bool seqReadFailed = false;            // written by the reader thread, read after the join
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } seqJoiner{seqReader};
  if (G.seqName) {
    fprintf(stderr, "-- Opening sequences in '%s'.\n", G.seqName);
    auto sf = std::make_shared<SeqFile>(G.seqName);
    if (!sf->ok()) { fprintf(stderr, "ERROR: cannot open '%s'.\n", G.seqName); return 1; }
    const std::string seqPath = G.seqName;
    seqReader = std::thread([&recs, &seqReadFailed, sf, seqPath]() {
      if (read_fasta_parallel(seqPath, recs)) { seqReadFailed = sf->finish() != 0; return; }     // plain FASTA: all host threads
      SeqRecord r;
      while (sf->next(r)) recs.push_back(std::move(r));
      seqReadFailed = sf->finish() != 0;      // a decompressor that died mid-stream: the records read so far are NOT the file
    });
  }
  // The variant modes ask the lookup tables for the k-mers of the enumerated PATHS and nothing else (varMer::score, varMer.C:76-84): one slot
  // on one device can prepare the call set first (host work), claim exactly those k-mers on a sequence-only index while it does
  // (mfx_vcf_prepare_path_index) and let both databases update them -- the PATH-ONLY index, a thirteenth of the full tables (3 Gb, 3.7 M calls:
  // 10.3 GB instead of 137; device work of the build 0.27 instead of 0.65 s).  Its price is the order: the VCF and the clusters' paths come
  // BEFORE the index instead of under its build -- on a 16-core host 1.9-2.1 s against 1.55-1.65 s of wall (profiles/r06_cfg4_cli.txt).  So it
  // is what a run takes when the full tables do not fit (-memory, or 90 % of the device's free memory), and MFX_CLI_PATH_INDEX=1 / 0 says
  // so explicitly.  Not with -index (the image caches the full tables), k > 31, several slots, MFX_CLI_VCF_AHEAD=0.
  {
    const char *vs = getenv("MFX_VARIANT_SLOTS");
    const size_t slots = (vs && atoi(vs) > 0) ? (size_t)atoi(vs) : G.devices.size();
    const char *pe = getenv("MFX_CLI_PATH_INDEX"), *pa = getenv("MFX_CLI_VCF_AHEAD");
    const bool can = variantMode && G.vcfName && G.seqName && slots == 1 && !G.sharded && k <= 31 && !G.indexName && !(pa && atoi(pa) == 0) &&
                     !(getenv("MFX_CLI_FULL_INDEX") && atoi(getenv("MFX_CLI_FULL_INDEX")));
    if (can && pe) pathOnly = atoi(pe) != 0;
    else if (can) {
      const uint64_t capacity = rdb.n_kmers + (G.seqDBname ? adb.n_kmers : bases_upper_bound(G.seqName)) + 1024;
      const double fullGB = mfx_index_estimate_gb(k, capacity);
      uint64_t freeB = 0, totalB = 0;
      const bool known = mfx_device_memory(G.device, &freeB, &totalB) == MFX_OK;
      pathOnly = (G.maxMemory > 0 && fullGB > G.maxMemory) || (known && fullGB * 1e9 > 0.9 * (double)freeB);
      if (pathOnly)
        fprintf(stderr, "-- The full lookup tables would take %.1f GB (%s %.1f GB): building the path-only index of the call set instead.\n", fullGB,
                G.maxMemory > 0 && fullGB > G.maxMemory ? "-memory allows" : "free on the device:", G.maxMemory > 0 && fullGB > G.maxMemory ? G.maxMemory : freeB / 1e9);
    }
  }
  // the path-only index is built FROM the call set: its VCF is read and parsed (host work only) while the FASTA file is
  std::future<mfx_vcf *> vcfLoad;
  std::string vcfAheadError;
  struct VcfLoadGuard { std::future<mfx_vcf *> &f; ~VcfLoadGuard() { if (f.valid()) mfx_vcf_free(f.get()); } } vcfLoadGuard{vcfLoad};
  if (pathOnly)
    vcfLoad = std::async(std::launch::async, [&G, &vcfAheadError]() {
      mfx_vcf *v = mfx_vcf_load(G.vcfName);
      if (!v) vcfAheadError = mfx_last_error();                    // (errors are per thread: carried to the caller's)
      return v;
    });
  bool seqDone = false;
  auto finish_seq = [&]() {
    if (seqDone) return;
    seqDone = true;
    if (seqReader.joinable()) seqReader.join();
    if (seqReadFailed) {
      fprintf(stderr, "ERROR: reading '%s' failed (read error, or the decompressor exited with an error: truncated or corrupt file?).\n", G.seqName);
      exit(1);
    }
    bases.resize(recs.size());
    lens.resize(recs.size());
    for (size_t i = 0; i < recs.size(); ++i) { bases[i] = recs[i].data(); lens[i] = recs[i].size(); totalBases += lens[i]; }
    lap("read sequences");
  };
  // ... and without -seqmers whenever the file tells an upper bound of its bases: the table is sized by the bound, the
  // read database is loaded while the file is read, the assembly k-mers are counted once it is in
  // Only compressed files are worth it: their decompression takes seconds per Gb on one core, while a plain file is parsed
  // at 2-5 GB/s and reading it under the build measurably slows the build's own readers (1 Gb: 1.8-3.0 s read first,
  // 2.4-3.2 s overlapped; MFX_CLI_OVERLAP=1 / 0 forces either).
  // -hist and -dump ask the lookup tables for the k-mers of -sequence and nothing else (merfin-histogram.C:54-64,
  // merfin-dump.C:44-61): they get a SEQUENCE-ONLY index -- the sequence's k-mers are claimed first, the databases only
  // update those (half of a 30x human read database, the error k-mers, never gets a slot; k <= 21: 8-byte slots).  The
  // other report types need the whole read database.  MFX_CLI_FULL_INDEX=1 builds the full tables for every type.
  bool seqOnly = (G.reportType == OP_HIST || G.reportType == OP_DUMP) && !G.sharded && k <= 31 &&
                 !(getenv("MFX_CLI_FULL_INDEX") && atoi(getenv("MFX_CLI_FULL_INDEX")));
  const char *ov = getenv("MFX_CLI_OVERLAP");
  const bool compressed = G.seqName && mfx_suffix_tool(G.seqName) != nullptr;
  const bool wantOverlap = ov ? atoi(ov) != 0 : compressed;
  const uint64_t basesBound = (G.seqName && !G.seqDBname && wantOverlap && !seqOnly) ? bases_upper_bound(G.seqName) : 0;
  const bool deferSeq = G.seqName && !G.sharded && !seqOnly && !pathOnly && wantOverlap && (G.seqDBname || basesBound > 0);   // (the path-only index needs the sequences first)
  // while the reader thread is on the file: the device's context, the library's code object and the pinned-memory path come
  // up here instead of inside the first upload (~0.07 s; an error here is left to that upload to report).  Not with a
  // decompressor child around (see the device check above) and not for several devices (their slots come up in parallel).
  if (G.seqName && !compressed && G.devices.size() == 1 && !(getenv("MFX_CLI_WARM") && !atoi(getenv("MFX_CLI_WARM")))) (void)mfx_device_warm(G.device);
  if (!stageFirst) begin_stage();
  if (!deferSeq) finish_seq();
  if (pathOnly) begin_path_stages();
  if (G.sharded) {
    if (G.devices.size() < 2) {
      fprintf(stderr, "ERROR: -sharded needs at least two -devices.\n");
      return 1;
    }
    if (G.indexName) { fprintf(stderr, "ERROR: -index caches a whole table; it cannot be combined with -sharded.\n"); return 1; }
    // -hist / -dump: a part of the sequences per device, each on the sequence-only index of ITS k-mers (no exchange);
    // everything else -- and a non-canonical database -- takes the hash-sharded full tables (MFX_CLI_FULL_INDEX=1: always)
    const char *fi = getenv("MFX_CLI_FULL_INDEX");
    if ((G.reportType == OP_HIST || G.reportType == OP_DUMP) && k <= 31 && !(fi && atoi(fi))) {
      const int prc = run_parts(G, k, recs, bases, lens);
      if (prc >= 0) return prc;
    }
    return run_sharded(G, k, rdb, adb, recs, bases, lens, totalBases);
  }
  mfx_index *ix = nullptr;
  mfx_seq *seq = nullptr;
  // -hist on one device with the assembly k-mers coming from -seqmers: nothing needs the sequence in HBM before the
  // evaluation, so its upload is streamed under the -hist kernel (mfx_hist_run_streamed).  Otherwise the index build
  // counts the assembly k-mers from the packed sequence and it goes up first.
  const bool streamHist = G.reportType == OP_HIST && G.seqDBname && G.devices.size() == 1 && !seqOnly;
  auto make_seq = [&]() -> bool {
    if (!recs.empty() || G.seqName) {
      seq = streamHist ? mfx_seq_create(G.device, lens.data(), (uint32_t)recs.size())
                       : mfx_seq_upload(G.device, bases.data(), lens.data(), (uint32_t)recs.size());
      if (!seq) return false;
    }
    lap("upload sequences");
    if (stage) mfx_db_stage_boost(stage);                         // the host's threads are free: the database's readers may have them all
    return true;
  };
  // The variant modes never evaluate the assembly itself on the device -- the paths around the variants go up batch by batch
  // (mfx_variants_run takes the host bases) --, so the assembly is uploaded only if its k-mers must be counted there (no
  // -seqmers: the counting branch below makes the upload when it gets there).  3 Gb: 0.2 s and 1.1 GB of HBM not spent.
  const bool seqOnDevice = !variantMode;
  if (!deferSeq && seqOnDevice && !make_seq()) DIE_MFX("uploading sequences");
  // variant modes on one slot: the VCF is read and parsed (host work only) on a thread of its own while the index is built
  // (merfin opens it after load_Kmers, merfin-globals.C:201-219; 0.15 s of a 4 M-call set).  An error is reported where the
  // VCF is opened below.  Several slots split the file per slot instead.
  std::future<mfx_vcf *> vcfAhead;
  {
    const char *vs = getenv("MFX_VARIANT_SLOTS");
    const size_t slots = (vs && atoi(vs) > 0) ? (size_t)atoi(vs) : G.devices.size();
    const char *pa = getenv("MFX_CLI_VCF_AHEAD");
    const bool dbgAhead = G.debug;
    if (variantMode && G.vcfName && slots == 1 && !G.sharded && !(pa && atoi(pa) == 0) && !pathOnly)
      vcfAhead = std::async(std::launch::async, [&G, &vcfAheadError, &recs, &bases, &lens, k, dbgAhead, deferSeq]() {
        mfx_vcf *v = mfx_vcf_load(G.vcfName);
        if (!v) { vcfAheadError = mfx_last_error(); return v; }    // (errors are per thread: carried to the caller's)
        // MFX_CLI_VCF_AHEAD=2: ... and its clusters merged, their allele combinations enumerated and packed here as well (mfx_vcf_prepare:
        // stage A of the run needs the sequences, not the index).  Not the default: on a host whose cores the index build already
        // keeps busy (the 16-core quota of the measured boxes: its readers copy 13 GB out of the page cache) the two slow each other
        // down -- config 4 at 3 Gb 2.02-2.07 s with the load alone ahead, 2.67-2.75 s with stage A too (profiles/r05_cfg4_cli_ahead.txt).
        const char *pa2 = getenv("MFX_CLI_VCF_AHEAD");
        // (never while the sequences are still being read -- deferSeq: finish_seq() fills recs / bases / lens on the main thread later)
        if (!(pa2 && atoi(pa2) == 2) || deferSeq) return v;
        std::vector<const char *> nm(recs.size());
        for (size_t c = 0; c < recs.size(); ++c) nm[c] = recs[c].name.c_str();
        mfx_variant_opts o;
        o.mode = G.reportType;
        o.comb = G.comb;
        o.nosplit = G.nosplit ? 1 : 0;
        o.debug_path = dbgAhead ? "-" : nullptr;                   // (only whether it is set matters here)
        if (mfx_vcf_prepare(v, k, nm.data(), bases.data(), lens.data(), (uint32_t)recs.size(), &o)) {
          vcfAheadError = mfx_last_error();
          mfx_vcf_free(v);
          v = nullptr;
        }
        return v;
      });
  }
  struct VcfAheadGuard { std::future<mfx_vcf *> &f; ~VcfAheadGuard() { if (f.valid()) mfx_vcf_free(f.get()); } } vcfAheadGuard{vcfAhead};
  FILE *probe = G.indexName ? fopen(G.indexName, "rb") : nullptr;
  uint64_t fingerprint = G.indexName ? input_fingerprint(G, seqOnly) : 0;
  // a run that would build the sequence-only index also accepts a FULL image of the same inputs: that is what such a run
  // saved when a database turned out not to be canonical (the fallback below) -- without this every later run would
  // build twice again and the cache would never take effect
  const uint64_t fingerprintFull = (G.indexName && seqOnly) ? input_fingerprint(G, false) : fingerprint;
  if (probe) {
    fclose(probe);
    fprintf(stderr, "-- Loading the index image '%s' (k-mer databases are not read).\n", G.indexName);
    ix = mfx_index_load(G.indexName, G.maxMemory, G.device);
    if (!ix) { fprintf(stderr, "\n%s\n\n", mfx_last_error()); return 1; }
    mfx_index_info info;
    if (mfx_index_get_info(ix, &info)) DIE_MFX("reading the index image");
    if (info.k != k) { fprintf(stderr, "ERROR: the index image holds %d-mers but -readmers holds %d-mers.\n", info.k, k); return 1; }
    uint64_t fp = 0, imin = 0, imax = 0;
    if (mfx_index_get_origin(ix, &fp, &imin, &imax)) DIE_MFX("reading the index image");
    if (seqOnly && info.seq_only == 0 && fp == fingerprintFull && imin == G.minV && imax == G.maxV) {
      seqOnly = false;                                              // the full tables of these inputs (a non-canonical database)
      fingerprint = fingerprintFull;
    } else if (fp != fingerprint || imin != G.minV || imax != G.maxV || (info.seq_only != 0) != seqOnly) {
      // the image was built from other inputs (a re-polished -sequence, another database, other -min/-max):
      // using it would give wrong asmK / readK with no diagnostic
      fprintf(stderr, "-- The index image '%s' was built from other inputs (%s); rebuilding it.\n", G.indexName,
              (imin != G.minV || imax != G.maxV) ? "-min/-max differ" : "a database or the sequence file changed");
      mfx_index_free(ix);
      ix = nullptr;
    }
  }
  mfx_vcf *vcfReady = nullptr;                                      // the call set taken from vcfAhead (prepared): the run below uses it
  struct VcfReadyGuard { mfx_vcf *&v; ~VcfReadyGuard() { if (v) mfx_vcf_free(v); v = nullptr; } } vcfReadyGuard{vcfReady};
  if (!ix && pathOnly && vcfLoad.valid()) {
    vcfReady = vcfLoad.get();
    if (!vcfReady) { fprintf(stderr, "ERROR: variant scoring: %s\n", vcfAheadError.c_str()); return 1; }
    step("(before the index: sequences, VCF)");
    // the clusters merged, the table made from the bound of their path text, then batch by batch: the paths' tables packed on the host, their
    // k-mers claimed on the device under the next batch's packing (mfx_vcf_prepare_path_index)
    fprintf(stderr, "-- Claiming the %d-mers of the variants' paths on the GPU.\n", k);
    {
      std::vector<const char *> nm(recs.size());
      for (size_t c = 0; c < recs.size(); ++c) nm[c] = recs[c].name.c_str();
      mfx_variant_opts o;
      o.mode = G.reportType;
      o.comb = G.comb;
      o.nosplit = G.nosplit ? 1 : 0;
      o.debug_path = G.debug ? "-" : nullptr;                       // (only whether it is set matters here)
      if (mfx_vcf_prepare_path_index(vcfReady, k, nm.data(), bases.data(), lens.data(), (uint32_t)recs.size(), &o, G.maxMemory, G.device, 0.7, &ix))
        DIE_MFX("preparing the call set");
      if (!ix) fprintf(stderr, "-- %s\n-- Building the full lookup tables instead.\n", mfx_last_error());
    }
    if (ix) {
      mfx_index_info info;
      if (mfx_index_get_info(ix, &info)) DIE_MFX("reading the path-only index");
      fprintf(stderr, "--\n-- Memory needed: %.3f GB (the %d-mers of the variants' paths: %lu)\n-- Memory limit:  %.3f GB%s\n--\n", info.bytes / 1e9, k,
              (unsigned long)info.distinct, G.maxMemory, G.maxMemory > 0 ? "" : " (none)");
    }
    if (stage) mfx_db_stage_boost(stage);                           // the host's threads are free: the databases' readers may have them all
    if (stageAsm) mfx_db_stage_boost(stageAsm);
    step("prepare the call set + claim the paths' k-mers");
    int lrc = 0;
    if (ix) {
      if (G.seqDBname) {
        fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.seqDBname);
        lrc = stageAsm ? mfx_index_load_db_staged(ix, stageAsm, 1, 0, ~0ull) : mfx_index_load_db(ix, G.seqDBname, 1, 0, ~0ull);
        if (stageAsm) { mfx_db_stage_free(stageAsm); stageAsm = nullptr; }
        if (lrc && lrc != MFX_E_NONCANON) DIE_MFX("loading -seqmers");
        step("load -seqmers");
      } else {
        // replaces `meryl count k=.. <seq> output <seq>.meryl` (merfin-globals.C:182-186) for the k-mers that will be asked for
        fprintf(stderr, "-- No -seqmer given. Counting the %d-mers of '%s' on the GPU.\n", k, G.seqName);
        if (!seq && !make_seq()) DIE_MFX("uploading sequences");
        if (mfx_index_count_claimed(ix, seq, nullptr)) DIE_MFX("counting sequence k-mers");
        step("count the sequence's k-mers");
      }
      if (!lrc) {
        fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.readDBname);
        lrc = stage ? mfx_index_load_db_staged(ix, stage, 0, G.minV, G.maxV) : mfx_index_load_db(ix, G.readDBname, 0, G.minV, G.maxV);
        if (lrc && lrc != MFX_E_NONCANON) DIE_MFX("loading -readmers");
        step("load -readmers");
      }
      if (lrc == MFX_E_NONCANON) {
        fprintf(stderr, "-- A k-mer database is not canonical; building the full lookup tables instead.\n");
        mfx_index_free(ix);
        ix = nullptr;
      }
    }
    // (the staged bytes are not needed by the run; a fall-back to the full tables reads the files)
    if (stage) { mfx_db_stage_free(stage); stage = nullptr; }
    if (stageAsm) { mfx_db_stage_free(stageAsm); stageAsm = nullptr; }
    if (!ix) pathOnly = false;                                      // (the prepared call set runs on the full tables as well)
  }
  if (!ix && seqOnly) {
    const uint64_t capacity = totalBases + 1024;                  // a sequence has at most one new k-mer per base
    fprintf(stderr, "--\n-- Memory needed: %.3f GB\n-- Memory limit:  %.3f GB%s\n--\n", mfx_index_estimate_gb_for_seq(k, capacity),
            G.maxMemory, G.maxMemory > 0 ? "" : " (none)");
    step("(before the index)");
    // load factor of the table: 0.4 unless the user says otherwise (MFX_LOAD_FACTOR) -- the emptier tables the library would pick probe
    // faster (3 Gb: 20 instead of 23 ms of -hist kernel) but a run that starts behind another waits seconds in hipMalloc for the driver
    // to clear what that one freed, the longer the more of the HBM both ask for (profiles/r05_e2e_lf_ab.txt)
    ix = mfx_index_create_for_seq_lf(k, capacity, G.maxMemory, G.device, 0.4);
    if (!ix && stage) {                                             // (the staged database may be what the table lacks)
      mfx_db_stage_free(stage);
      stage = nullptr;
      ix = mfx_index_create_for_seq_lf(k, capacity, G.maxMemory, G.device, 0.4);
    }
    if (!ix) {
      fprintf(stderr, "\n%s\n\n", mfx_last_error());
      return 1;
    }
    step("create the table");
    int lrc = 0;
    bool fused = false;
    if (G.seqDBname) {
      fprintf(stderr, "-- Claiming the %d-mers of '%s' on the GPU.\n", k, G.seqName);
      if (mfx_index_claim_seq(ix, seq, nullptr)) DIE_MFX("claiming sequence k-mers");
      step("claim the sequence's k-mers");
      fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.seqDBname);
      lrc = mfx_index_load_db(ix, G.seqDBname, 1, 0, ~0ull);
      if (lrc && lrc != MFX_E_NONCANON) DIE_MFX("loading -seqmers");
      step("load -seqmers");
    } else {
      // replaces `meryl count k=.. <seq> output <seq>.meryl` (merfin-globals.C:182-186)
      fprintf(stderr, "-- No -seqmer given. Counting the %d-mers of '%s' on the GPU.\n", k, G.seqName);
      // (one call for the counting and the load of -readmers: the database crosses PCIe while the k-mers are claimed)
      fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.readDBname);
      lrc = stage ? mfx_index_build_for_hist_staged(ix, seq, stage, G.minV, G.maxV) : mfx_index_build_for_hist(ix, seq, G.readDBname, G.minV, G.maxV);
      if (stage) { mfx_db_stage_free(stage); stage = nullptr; }     // its device memory is not needed by the evaluation
      if (lrc && lrc != MFX_E_NONCANON) DIE_MFX("counting sequence k-mers / loading -readmers");
      step("count the sequence's k-mers + load -readmers");
      fused = true;
    }
    if (!lrc && !fused) {
      fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.readDBname);
      lrc = mfx_index_load_db(ix, G.readDBname, 0, G.minV, G.maxV);
      if (lrc && lrc != MFX_E_NONCANON) DIE_MFX("loading -readmers");
      step("load -readmers");
    }
    if (lrc == MFX_E_NONCANON) {
      // one slot per canonical k-mer cannot answer value(fmer) + value(rmer) of a non-canonical database
      fprintf(stderr, "-- A k-mer database is not canonical; building the full lookup tables instead.\n");
      mfx_index_free(ix);
      ix = nullptr;
      seqOnly = false;
      fingerprint = fingerprintFull;                                // what is saved below is the full index of these inputs
    } else if (G.indexName) {
      fprintf(stderr, "-- Writing the index image '%s'.\n", G.indexName);
      if (mfx_index_set_fingerprint(ix, fingerprint) || mfx_index_save(ix, G.indexName)) DIE_MFX("writing the index image");
    }
  }
  if (!ix) {
    // the table is sized before the sequence is in when the file promised a bound on its bases; a file that breaks the
    // promise (several gzip members, ...) costs a second build with the true number
    for (int attempt = 0; !ix; ++attempt) {
      const uint64_t capacity = rdb.n_kmers + (G.seqDBname ? adb.n_kmers : (seqDone ? totalBases : basesBound)) + 1024;
      fprintf(stderr, "--\n-- Memory needed: %.3f GB\n-- Memory limit:  %.3f GB%s\n--\n", mfx_index_estimate_gb(k, capacity),
              G.maxMemory, G.maxMemory > 0 ? "" : " (none)");
      // (the smallest table the library makes, load factor 0.7: the device stage of these report types is a hundredth of the run, and a
      // table allocated behind another process waits for the driver to clear what that one freed -- the longer, the larger both are)
      ix = mfx_index_create_lf(k, capacity, G.maxMemory, G.device, 0.7);
      if (!ix && !G.seqDBname && !seqDone && basesBound > 0) {
        // the capacity came from the file's BOUND on its bases (a .gz of a human assembly sits near the 4 GiB step of that
        // bound): read the file to the end, as the reference does before anything else, and size the table by the truth
        fprintf(stderr, "-- The table sized by the bound on the bases of '%s' does not fit; reading the file first.\n", G.seqName);
        finish_seq();
        continue;
      }
      if (!ix) {
        fprintf(stderr, "\n%s\n\n", mfx_last_error());
        return 1;
      }
      fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.readDBname);
      if (mfx_index_load_db(ix, G.readDBname, 0, G.minV, G.maxV)) DIE_MFX("loading -readmers");
      if (G.seqDBname) {
        fprintf(stderr, "-- Loading kmers from '%s' into lookup table.\n", G.seqDBname);
        if (mfx_index_load_db(ix, G.seqDBname, 1, 0, ~0ull)) DIE_MFX("loading -seqmers");
        break;
      }
      // replaces `meryl count k=.. <seq> output <seq>.meryl` (merfin-globals.C:182-186)
      if (deferSeq && !seqDone) {            // the read database is in; now the sequence is needed
        finish_seq();
        if (totalBases > basesBound && attempt == 0) {
          fprintf(stderr, "-- NOTE: '%s' holds %lu bases, more than its size promised (%lu); rebuilding the table.\n", G.seqName,
                  (unsigned long)totalBases, (unsigned long)basesBound);
          mfx_index_free(ix);
          ix = nullptr;
          continue;
        }
      }
      if (!seq && !make_seq()) DIE_MFX("uploading sequences");
      fprintf(stderr, "-- No -seqmer given. Counting the %d-mers of '%s' on the GPU.\n", k, G.seqName);
      if (mfx_index_count_asm(ix, seq, nullptr)) DIE_MFX("counting sequence k-mers");
    }
    if (G.indexName) {
      fprintf(stderr, "-- Writing the index image '%s'.\n", G.indexName);
      if (mfx_index_set_fingerprint(ix, fingerprint) || mfx_index_save(ix, G.indexName)) DIE_MFX("writing the index image");
    }
  }

  lap("build / load index");
  if (deferSeq && !seq) {
    finish_seq();
    if (seqOnDevice && !make_seq()) DIE_MFX("uploading sequences");
  }
  mfx_kparams kp{G.peak, (uint32_t)G.copyKmerK.size(), G.copyKmerK.data(), G.copyKmerP.data()};
  mfx_eval *ev = mfx_eval_create(ix, &kp, 0);
  if (!ev) DIE_MFX("creating evaluator");

  int rc = 0;
  if (G.reportType == OP_HIST) {
    fprintf(stderr, "-- Generate histogram of the k* metric to '%s'.\n", G.outName);
    mfx_hist_result r;
    if (G.devices.size() > 1) {
      // one process, N devices (the reference drives all its workers from one binary, merfin.C:366-414): replicas of
      // the table and of the packed assembly by peer copy, every device evaluates its block-cyclic share
      const size_t N = G.devices.size();
      Slots S;
      int bad = S.make(G, ix, seq, ev, &kp) ? 0 : 1;
      lap("replicate index");
      fprintf(stderr, "-- Evaluating on %zu devices.\n", N);
      if (!bad && mfx_hist_run_multi(S.evs.data(), S.sqs.data(), (uint32_t)N, &r)) bad = 1;
      std::string why = bad ? mfx_last_error() : "";
      S.release();
      if (bad) { fprintf(stderr, "ERROR: -hist on %zu devices: %s\n", N, why.c_str()); return 1; }
    } else if (streamHist) {
      if (mfx_hist_run_streamed(ev, seq, bases.data(), &r)) DIE_MFX("-hist");
    } else if (mfx_hist_run(ev, seq, &r)) DIE_MFX("-hist");
    bool ok = true;
    print_hist(recs, r, k, G.outName, &ok);
    if (!ok) DIE_MFX("writing histogram");
    mfx_hist_result_free(&r);
  } else if (G.reportType == OP_DUMP) {
    fprintf(stderr, "-- Dump per-base k* metric to '%s'.\n", G.outName);
    uint64_t cumMissing = 0, cumAsm = 0;
    if (G.skipMissing) {
      // merfin-dump.C:34,81-87: -skipMissing suppresses the dump file entirely; only the counts remain
      mfx_hist_result r;
      if (mfx_hist_run(ev, seq, &r)) DIE_MFX("-dump -skipMissing");
      for (size_t c = 0; c < recs.size(); ++c) {
        cumMissing += r.contig_kmissing[c];
        cumAsm += r.contig_kasm[c];
        fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)r.contig_kmissing[c], (unsigned long)cumMissing, (unsigned long)cumAsm);
      }
      mfx_hist_result_free(&r);
    } else if (G.devices.size() > 1) {
      // per-contig ordered output on N devices: a contiguous run of contigs per slot (balanced by bases), one host
      // thread per slot writes its part, the parts are concatenated in slot order
      const size_t N = G.devices.size();
      Slots S;
      if (!S.make(G, ix, seq, ev, &kp)) { fprintf(stderr, "ERROR: -dump on %zu devices: %s\n", N, mfx_last_error()); return 1; }
      lap("replicate index");
      fprintf(stderr, "-- Evaluating on %zu devices.\n", N);
      std::vector<double> w(recs.size());
      for (size_t c = 0; c < recs.size(); ++c) w[c] = (double)lens[c];
      const auto runs = contig_partition(w, N);
      std::vector<std::string> parts(N), errs(N);
      std::vector<std::vector<uint64_t>> ka(N), km(N);
      std::vector<std::thread> th;
      for (size_t d = 0; d < N; ++d) {
        char suf[32];
        snprintf(suf, sizeof(suf), ".part%04zu", d);
        parts[d] = std::string(G.outName) + suf;
        th.emplace_back([&, d]() {
          mfx_host_threads_share((unsigned)N);
          for (size_t c = runs[d].first; c < runs[d].second; ++c) {
            uint64_t a = 0, m2 = 0;
            if (mfx_dump_contig(S.evs[d], S.sqs[d], (uint32_t)c, recs[c].name.c_str(), parts[d].c_str(), c > runs[d].first, &a, &m2)) { errs[d] = mfx_last_error(); return; }
            ka[d].push_back(a);
            km[d].push_back(m2);
          }
        });
      }
      for (auto &x : th) x.join();
      S.release();
      for (size_t d = 0; d < N; ++d) if (!errs[d].empty()) { fprintf(stderr, "ERROR: -dump (slot %zu): %s\n", d, errs[d].c_str()); return 1; }
      if (!concat_parts(G.outName, parts, false)) { fprintf(stderr, "ERROR: cannot write '%s'.\n", G.outName); return 1; }
      for (size_t d = 0; d < N; ++d)
        for (size_t i = 0; i < ka[d].size(); ++i) {
          cumMissing += km[d][i];
          cumAsm += ka[d][i];
          fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[runs[d].first + i].name.c_str(), (unsigned long)km[d][i], (unsigned long)cumMissing, (unsigned long)cumAsm);
        }
    } else {
      for (size_t c = 0; c < recs.size(); ++c) {
        uint64_t ka = 0, km = 0;
        if (mfx_dump_contig(ev, seq, (uint32_t)c, recs[c].name.c_str(), G.outName, c > 0, &ka, &km)) DIE_MFX("-dump");
        cumMissing += km;
        cumAsm += ka;
        fprintf(stderr, "%s\t%lu\t%lu\t%lu\n", recs[c].name.c_str(), (unsigned long)km, (unsigned long)cumMissing, (unsigned long)cumAsm);
      }
      if (recs.empty()) { FILE *f = fopen(G.outName, "w"); if (f) fclose(f); }
    }
  } else if (variantMode) {
    // open_Inputs + processVariants + outputVariants (merfin-globals.C:201-219, merfin-variants.C:131-345)
    fprintf(stderr, "-- Opening vcf file '%s'.\n", G.vcfName);
    fprintf(stderr, "-- Generate variant mers and score them.\n");
    std::string outName = std::string(G.outName) + (G.reportType == OP_POLISH ? ".polish.vcf" : ".filter.vcf");   // :324-327
    std::string dbgName = std::string(G.outName) + ".00.debug.gz";
    std::vector<const char *> names(recs.size());
    for (size_t c = 0; c < recs.size(); ++c) names[c] = recs[c].name.c_str();
    mfx_variant_opts vo;
    vo.mode = G.reportType;           // OP_* values are the MFX_VAR_* values
    vo.comb = G.comb;
    vo.nosplit = G.nosplit ? 1 : 0;
    vo.debug_path = G.debug ? dbgName.c_str() : nullptr;
    uint64_t ncl = 0;
    // One device is run as ONE slot.  (Rounds 2-3 ran it as 4 slots sharing its table: the host phases of these modes then
    // alternated between parallel and serial stretches and several slots filled one another's gaps -- 1 Gb: 2.0 s in 8 slots
    // against 2.7 s in one.  With the host side of round 4 -- arenas per run of clusters, parallel VCF load, device scoring --
    // one slot on all host threads is at least as fast and needs neither the per-slot VCF files nor the evaluator replicas:
    // config 4 at 3 Gb, evaluate + write 1.11-1.14 s in one slot, 1.16-1.19 s in four + 0.13-0.15 s to set the slots up,
    // profiles/r04_cfg4_fullsize.txt.)  MFX_VARIANT_SLOTS overrides; several -devices are one slot each.
    {
      size_t want = G.devices.size();
      const char *vs = getenv("MFX_VARIANT_SLOTS");
      if (vs && atoi(vs) > 0) want = (size_t)atoi(vs);
      const std::vector<int> real = G.devices;
      while (G.devices.size() < want) G.devices.push_back(real[G.devices.size() % real.size()]);
    }
    if (G.devices.size() > 1) {
      // BASELINE config 4 names 8 GPUs: contigs cut into one contiguous run per slot, balanced by their VCF records;
      // every slot scores its clusters on its own device and writes its part; one VCF header in the concatenation
      const size_t N = G.devices.size();
      std::vector<double> w(recs.size(), 0.0);
      // one pass over the VCF: the records per contig (for the balance), and the lines themselves, so that every slot is
      // handed a VCF of ITS contigs only (each slot parsing all 4 M records of a human call set made N slots N times the
      // host work of one; the host is what bounds these modes)
      std::string vtext;
      std::vector<std::pair<size_t, size_t>> vhead;                          // header lines (offset, length incl. newline)
      std::vector<std::vector<std::pair<size_t, size_t>>> vlines(recs.size());   // data lines per contig, in file order
      {
        mfx_file vf = mfx_open_reader(G.vcfName);
        if (!vf.f) { fprintf(stderr, "ERROR: cannot open VCF '%s'.\n", G.vcfName); return 1; }
        {
          std::vector<char> blk(1 << 22);
          size_t n;
          while ((n = fread(blk.data(), 1, blk.size(), vf.f)) > 0) vtext.append(blk.data(), n);
        }
        if (mfx_close(vf)) { fprintf(stderr, "ERROR: reading VCF '%s' failed.\n", G.vcfName); return 1; }
        std::vector<std::pair<std::string, size_t>> idx;
        for (size_t c = 0; c < recs.size(); ++c) idx.emplace_back(recs[c].name, c);
        std::sort(idx.begin(), idx.end());
        size_t last_c = recs.size();
        std::string last_chr;
        for (size_t o = 0; o < vtext.size();) {
          const char *nl = (const char *)memchr(vtext.data() + o, '\n', vtext.size() - o);
          const size_t e = nl ? (size_t)(nl - vtext.data()) + 1 : vtext.size(), n = e - o;
          if (vtext[o] == '#') vhead.emplace_back(o, n);
          else if (n > 1) {
            const char *tab = (const char *)memchr(vtext.data() + o, '\t', n);
            if (tab) {
              const size_t cl = (size_t)(tab - (vtext.data() + o));
              if (last_c == recs.size() || last_chr.size() != cl || memcmp(last_chr.data(), vtext.data() + o, cl) != 0) {
                last_chr.assign(vtext.data() + o, cl);
                auto it = std::lower_bound(idx.begin(), idx.end(), std::make_pair(last_chr, (size_t)0));
                last_c = (it != idx.end() && it->first == last_chr) ? it->second : recs.size() + 1;   // + 1: not a contig of -sequence
              }
              if (last_c < recs.size()) { w[last_c] += 1.0; vlines[last_c].emplace_back(o, n); }
            }
          }
          o = e;
        }
        for (size_t c = 0; c < recs.size(); ++c) w[c] += 1e-9 * (double)lens[c];
      }
      Slots S;
      if (!S.make(G, ix, nullptr, ev, &kp)) { fprintf(stderr, "ERROR: variant scoring on %zu devices: %s\n", N, mfx_last_error()); return 1; }
      lap("replicate index");
      fprintf(stderr, "-- Evaluating in %zu slots.\n", N);
      const auto runs = contig_partition(w, N);
      std::vector<std::string> parts(N), errs(N), dbgs(N), vins(N), logs(N);
      std::vector<uint64_t> ncls(N, 0);
      auto drop_parts = [&]() { for (size_t e = 0; e < N; ++e) { if (!parts[e].empty()) remove(parts[e].c_str()); if (!vins[e].empty()) remove(vins[e].c_str()); if (!logs[e].empty()) remove(logs[e].c_str()); } };
      std::vector<std::thread> th;
      for (size_t d = 0; d < N; ++d) {                                      // slot d's VCF: all header lines + the lines of its contigs
        char suf[48];
        snprintf(suf, sizeof(suf), ".part%04zu.in.vcf", d);
        vins[d] = outName + suf;
        FILE *vf = fopen(vins[d].c_str(), "w");
        bool ok = vf != nullptr;
        for (const auto &h : vhead) ok = ok && fwrite(vtext.data() + h.first, 1, h.second, vf) == h.second;
        for (size_t c = runs[d].first; c < runs[d].second && ok; ++c)
          for (const auto &l : vlines[c]) {
            ok = ok && fwrite(vtext.data() + l.first, 1, l.second, vf) == l.second;
            if (ok && vtext[l.first + l.second - 1] != '\n') ok = fputc('\n', vf) != EOF;     // a last line without newline
          }
        if (vf && fclose(vf) != 0) ok = false;
        if (!ok) { fprintf(stderr, "ERROR: cannot write '%s'.\n", vins[d].c_str()); drop_parts(); return 1; }
      }
      std::string().swap(vtext);
      for (size_t d = 0; d < N; ++d) {
        char suf[48];
        snprintf(suf, sizeof(suf), ".part%04zu", d);
        parts[d] = outName + suf;
        snprintf(suf, sizeof(suf), ".%02zu.debug.gz", d);
        dbgs[d] = std::string(G.outName) + suf;
        // every slot logs to its own file; they are replayed on stderr in slot order once all slots are done (N slots
        // writing PANIC / progress lines to one stderr interleave them mid-line)
        logs[d] = parts[d] + ".log";
        th.emplace_back([&, d]() {
          mfx_host_threads_share((unsigned)N);                 // N slots side by side: each takes its share of the host threads
          mfx_variant_opts o = vo;
          o.debug_path = G.debug ? dbgs[d].c_str() : nullptr;
          const size_t lo = runs[d].first, hi = runs[d].second;
          if (mfx_variants_run(S.evs[d], vins[d].c_str(), names.data() + lo, bases.data() + lo, lens.data() + lo, (uint32_t)(hi - lo), &o, parts[d].c_str(),
                               logs[d].c_str(), &ncls[d]))
            errs[d] = mfx_last_error();
        });
      }
      for (auto &x : th) x.join();
      for (size_t d = 0; d < N; ++d) {
        remove(vins[d].c_str());
        vins[d].clear();
        if (FILE *lf = fopen(logs[d].c_str(), "r")) {
          if (N > 1) fprintf(stderr, "-- slot %zu (contigs %zu..%zu):\n", d, runs[d].first, runs[d].second);
          char lb[1 << 14];
          size_t n;
          while ((n = fread(lb, 1, sizeof(lb), lf)) > 0) fwrite(lb, 1, n, stderr);
          fclose(lf);
        }
        remove(logs[d].c_str());
        logs[d].clear();
      }
      S.release();
      for (size_t d = 0; d < N; ++d) if (!errs[d].empty()) { fprintf(stderr, "ERROR: variant scoring (slot %zu): %s\n", d, errs[d].c_str()); drop_parts(); return 1; }
      if (!concat_parts(outName, parts, true)) { fprintf(stderr, "ERROR: cannot write '%s'.\n", outName.c_str()); drop_parts(); return 1; }
      for (uint64_t x : ncls) ncl += x;
    } else if (vcfReady || vcfAhead.valid()) {
      mfx_vcf *vcf = vcfReady ? vcfReady : vcfAhead.get();
      vcfReady = nullptr;
      if (!vcf) { fprintf(stderr, "ERROR: variant scoring: %s\n", vcfAheadError.c_str()); return 1; }
      const int vrc = mfx_variants_run_vcf(ev, vcf, names.data(), bases.data(), lens.data(), (uint32_t)recs.size(), &vo, outName.c_str(), nullptr, &ncl);
      // (leaving the 4 M records to the process's exit instead of freeing them here was measured: 0.1 s less in this phase, the same wall --
      // profiles/r05_exit_ab.txt)
      mfx_vcf_free(vcf);
      if (vrc) DIE_MFX("variant scoring");
    } else if (mfx_variants_run(ev, G.vcfName, names.data(), bases.data(), lens.data(), (uint32_t)recs.size(), &vo, outName.c_str(), nullptr, &ncl))
      DIE_MFX("variant scoring");
  } else if (G.reportType == OP_COMPL) {
    fprintf(stderr, "-- Compute completeness.\n");
    double t64[64], u64[64];
    if (mfx_completeness_pieces(ev, t64, u64)) DIE_MFX("-completeness");
    print_completeness(t64, u64);
  }

  lap("evaluate + write");
  mfx_eval_free(ev);
  if (seq) mfx_seq_free(seq);
  mfx_index_free(ix);
  lap("release");
  if (timing) {
    fprintf(stderr, "-- timing:");
    for (auto &ph : phases) fprintf(stderr, "  %s %.2fs", ph.first, ph.second);
    fprintf(stderr, "\n");
    if (atoi(getenv("MFX_CLI_TIMING")) > 1 && !steps.empty()) {
      fprintf(stderr, "-- timing (index):");
      for (auto &ph : steps) fprintf(stderr, "  %s %.3fs", ph.first, ph.second);
      fprintf(stderr, "\n");
    }
    if (atoi(getenv("MFX_CLI_TIMING")) > 2)
      fprintf(stderr, "-- stamps: main %.6f devices %.6f end %.6f\n", stamp_main, stamp_devices, epoch());
  }
  fprintf(stderr, "Bye!\n");
  // MFX_CLI_QUICK_EXIT=1: every output is written and closed by now; leave without the tear-down of the HIP runtime's statics
  // (not under a profiler: its tool library writes its files from an exit handler)
  if (const char *qe = getenv("MFX_CLI_QUICK_EXIT")) if (atoi(qe)) { fflush(nullptr); _exit(rc); }
  return rc;
}
