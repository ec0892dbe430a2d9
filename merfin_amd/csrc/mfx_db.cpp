// mfx_db.cpp -- k-mer database ingest: disk -> device index.
//
// Replaces merylFileReader + merylExactLookup::load as used by
// merfinGlobal::load_Kmers (src/merfin/merfin-globals.C:114-163).  Three
// on-disk forms are accepted:
//   MFX_DB_FLAT   this repo's flat binary (header + u64 k-mers + u32 counts);
//                 always verifiable, used by the tests and by fixtures.
//   MFX_DB_TEXT   `meryl print` text: "<kmer>\t<count>\n" (plain or .gz/.bz2/.xz).
//   MFX_DB_MERYL  a meryl database directory (merylIndex + 64 x 0xBBBBBB.merylData).
//                 The decoder follows the layout of marbl/meryl-utility as
//                 recalled in SURVEY.md Appendix C.  It is UNVALIDATED: the
//                 reference checkout contains no meryl source and no database,
//                 so the decoder has only been exercised against this repo's
//                 own writer of the same layout (tests/meryl_layout.py).  It
//                 refuses anything whose magic numbers / sizes do not match
//                 rather than guessing.
#include "mfx_internal.h"
#include "mfx_pipe.h"
#include "mfx_place.h"

#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <sys/mman.h>
#include <fcntl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

typedef unsigned __int128 u128;


bool is_dir(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
bool is_file(const std::string &p) {
  struct stat st;
  return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

int base_code(unsigned char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'T': case 't': return 2;
    case 'G': case 'g': return 3;
    default: return -1;
  }
}

// ---- flat binary ------------------------------------------------------------
struct FlatHeader {
  char     magic[8];     // "MFXKMER1"
  uint32_t k;
  uint32_t flags;        // bit 0: canonical; bit 1: PACKED records (k <= 21); bit 2: DELTA-coded blocks (sorted k-mers, k <= 31)
  uint64_t n;
  uint64_t n_escape;     // packed: records whose count did not fit the record (was: reserved, 0)
};
// Payload, plain : n k-mers (8 bytes; 16 for k > 31), then n uint32 counts.
// Payload, packed: n records {k-mer << 22 | count} (mfx_internal.h MFX_PACKED_*: 8 bytes per k-mer instead of 12 on disk, in
//                  the staging lanes and over PCIe; a count field of all ones = escape), then the n_escape escaped k-mers
//                  (uint64) and their counts (uint32).
// Payload, delta : what a SORTED database (strictly ascending k-mers: `meryl print` order) is written as -- blocks of
//                  MFX_DELTA_BLOCK k-mers, each {first k-mer (in the directory); count-1 differences of kbits bits; count values
//                  of vbits bits, all ones = escape}: uint64 nblocks, the directory (nblocks + 1 entries {first k-mer, file offset
//                  of the block (48 bits) | kbits << 48 | vbits << 56}; the last entry closes the last block), the blocks (8-byte
//                  aligned; bit fields LSB-first in little-endian uint64 words, the values start at a word boundary), then
//                  the n_escape escaped k-mers and counts as in the packed form.  kbits / vbits are chosen per block (vbits: what
//                  makes block + escapes smallest): 2.5-3 bytes per k-mer of a 30x human read set.  Blocks are decoded by the
//                  kernel that inserts them (mfx_table_add_delta_kernel): on disk, in the staging lanes and over PCIe the
//                  database is a third of its packed size.
constexpr uint32_t FLAT_PACKED = 2u;
constexpr uint32_t FLAT_DELTA = 4u;
// Payload, placed: a delta-coded file whose records are not the k-mers but the numbers P of mfx_place.h -- one to one with the canonical
//                  k-mers, ascending P = ascending LINE of the compact table, for a table of any size (13 <= k <= 30) -- in ascending
//                  order: what `merfin -convert -placed` makes.  The table's update kernel then walks the table line after line instead
//                  of reading a random line per record (mfx_table_add_placed_kernel).  Directory and blocks as in the delta form with P
//                  in the k-mer's place (2k + 3 bits); the escape list holds k-mers.  Bits 8-15 of `flags`: MFX_PLACE_VERSION of the
//                  functions that made P (another version is refused: convert again).
constexpr uint32_t FLAT_PLACED = 8u;
inline uint32_t flat_place_version(const FlatHeader &h) { return (h.flags >> 8) & 0xffu; }

// The header's counts are bounded by the file's size BEFORE any arithmetic uses them (a damaged or crafted header must end in
// MFX_E_FORMAT, never in a wrapped size check, a std::length_error through the C ABI or an out-of-bounds scatter): every
// escape takes 12 bytes; a k-mer takes 8 (16) + 4 bytes plain, 8 packed, and at least 2 bits delta-coded (vbits >= 2).
// nullptr = acceptable, else what is wrong.
const char *flat_header_problem(const FlatHeader &h, uint64_t fsize) {
  if (memcmp(h.magic, "MFXKMER1", 8) != 0) return "bad magic (not a flat k-mer file)";
  if (h.k < 1 || h.k > (uint32_t)MFX_MAX_K) return "k out of range";
  if (fsize < sizeof(FlatHeader)) return "truncated header";
  const uint64_t body = fsize - sizeof(FlatHeader);
  if (h.n_escape > body / 12) return "escape count beyond the file's size";
  const uint64_t rest = body - h.n_escape * 12;
  if (h.flags & FLAT_PLACED) {
    if (!(h.flags & FLAT_DELTA)) return "a placed database is delta-coded";
    if (h.k < (uint32_t)MFX_PLACE_MIN_K || h.k > (uint32_t)MFX_PLACE_MAX_K) return "placed records hold 13 <= k <= 31";
    if (flat_place_version(h) != MFX_PLACE_VERSION) return "placed by another version of the placement functions (convert the database again)";
  }
  if (h.flags & FLAT_DELTA) {
    if (h.k > (uint32_t)MFX_MAX_K_NARROW) return "delta-coded blocks hold k <= 31";
    if (h.n / 4 > rest) return "k-mer count beyond the file's size";
  } else if (h.flags & FLAT_PACKED) {
    if (h.k > (uint32_t)MFX_MAX_K_PACKED) return "packed records hold k <= 21";
    if (h.n > rest / 8) return "k-mer count beyond the file's size";
  } else {
    const uint64_t per = (h.k > (uint32_t)MFX_MAX_K_NARROW ? 16u : 8u) + 4u;
    if (h.n > rest / per) return "k-mer count beyond the file's size";
  }
  if (h.n_escape > h.n) return "more escapes than k-mers";
  return nullptr;
}
// a k-mer of k <= 31 bases has no bit at or above 2k
inline bool flat_key_fits(uint64_t key, uint32_t k) { return k >= 32 || (key >> (2 * k)) == 0; }
// bits of a record's key: the k-mer's 2k, or the 2k + 3 of a placed record
inline uint32_t flat_rec_bits(const FlatHeader &h) { return (h.flags & FLAT_PLACED) ? (uint32_t)mfx_p_bits((int)h.k) : 2u * h.k; }
inline bool flat_rec_fits(uint64_t key, const FlatHeader &h) { const uint32_t b = flat_rec_bits(h); return b >= 64 || (key >> b) == 0; }

// ---- meryl stuffedBits reader (UNVALIDATED) -----------
// A stuffedBits file image: u64 dataBlockLenMax (bits), u32 dataBlocksLen,
// u32 dataBlocksMax, u64 bgn[dataBlocksLen], u64 len[dataBlocksLen] (bits),
// then ceil(len/64) little-endian u64 words per block; values are packed
// MSB-first inside each word.
struct BitReader {
  const uint64_t *w = nullptr;
  uint64_t nbits = 0, pos = 0;
  bool ok = true;
  uint64_t get(uint32_t n) {               // n <= 64
    if (n == 0) return 0;
    if (pos + n > nbits) { ok = false; return 0; }
    uint64_t wi = pos >> 6, bo = pos & 63, v;
    if (bo + n <= 64) {
      v = (w[wi] << bo) >> (64 - n);
    } else {
      uint32_t n1 = 64 - (uint32_t)bo, n2 = n - n1;
      v = (((w[wi] << bo) >> bo) << n2) | (w[wi + 1] >> (64 - n2));
    }
    pos += n;
    return v;
  }
  uint64_t unary() {                        // number of 0 bits before the next 1 (the 1 is consumed)
    uint64_t z = 0;
    while (true) {
      if (pos >= nbits) { ok = false; return 0; }
      uint64_t wi = pos >> 6, bo = pos & 63;
      uint64_t rest = w[wi] << bo;
      if (rest) {
        uint32_t lz = (uint32_t)__builtin_clzll(rest);
        z += lz;
        pos += lz + 1;
        return z;
      }
      z += 64 - bo;
      pos += 64 - bo;
    }
  }
};

const uint64_t MERYL_IDX_MAGIC1 = 0x646e496c7972656dULL;   // "merylInd"
const uint64_t MERYL_DAT_MAGIC1 = 0x7461446c7972656dULL;   // "merylDat"
const uint64_t MERYL_DAT_MAGIC2 = 0x0a3030656c694661ULL;   // "aFile00\n"

struct StuffedFile {
  std::vector<uint64_t> words;     // all block payloads, concatenated
  std::vector<uint64_t> blk_word;  // first word of each block
  std::vector<uint64_t> blk_bits;  // bit length of each block
};

// reads ONE stuffedBits image starting at the current file offset; returns
// false at clean EOF, sets *err on a malformed image.
bool read_stuffed(FILE *f, StuffedFile &out, std::string *err) {
  uint64_t lenmax;
  uint32_t nblk, maxblk;
  size_t r = fread(&lenmax, 8, 1, f);
  if (r != 1) return false;                                   // EOF
  if (fread(&nblk, 4, 1, f) != 1 || fread(&maxblk, 4, 1, f) != 1) { *err = "truncated stuffedBits header"; return false; }
  if (nblk == 0 || nblk > maxblk || maxblk > (1u << 24)) { *err = "implausible stuffedBits block count"; return false; }
  std::vector<uint64_t> bgn(nblk), len(nblk);
  if (fread(bgn.data(), 8, nblk, f) != nblk || fread(len.data(), 8, nblk, f) != nblk) { *err = "truncated stuffedBits tables"; return false; }
  out.words.clear(); out.blk_word.clear(); out.blk_bits.clear();
  for (uint32_t b = 0; b < nblk; ++b) {
    if (len[b] > lenmax) { *err = "stuffedBits block longer than its declared maximum"; return false; }
    uint64_t nw = (len[b] + 63) / 64;
    size_t at = out.words.size();
    out.words.resize(at + nw + 1);
    if (nw && fread(out.words.data() + at, 8, nw, f) != nw) { *err = "truncated stuffedBits payload"; return false; }
    out.words[at + nw] = 0;
    out.blk_word.push_back(at);
    out.blk_bits.push_back(len[b]);
  }
  return true;
}

struct MerylIndex {
  uint32_t prefixSize = 0, suffixSize = 0, numFilesBits = 0, numBlocksBits = 0, flags = 0;
  int version = 0;
  // statistics block that follows the sizes (merylHistogram, first three 64-bit fields)
  bool     has_stats = false;
  uint64_t numUnique = 0, numDistinct = 0, numTotal = 0;
};

// What one decoded .merylData file amounts to; summed over the 64 files it must reproduce the master
// index's statistics (conformance checks of SURVEY App. C, applied on every load).
struct MerylFileSums {
  uint64_t kmers = 0;      // sum of the block headers' nKmers
  uint64_t unique = 0;     // k-mers with value 1
  uint64_t total = 0;      // sum of the values
  uint64_t blocks = 0;
};

// MFX_MERYL_LENIENT=1 turns the statistics cross-check into a warning (the statistics layout is the least
// certain part of the recalled format); the structural checks always fail hard.
bool meryl_lenient() {
  const char *e = getenv("MFX_MERYL_LENIENT");
  return e && atoi(e) != 0;
}

int read_meryl_master(const std::string &dir, MerylIndex &mi) {
  std::string p = dir + "/merylIndex";
  FILE *f = fopen(p.c_str(), "rb");
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s'", p.c_str());
  StuffedFile sf;
  std::string err;
  bool ok = read_stuffed(f, sf, &err);
  fclose(f);
  if (!ok) return mfx_fail(MFX_E_FORMAT, "'%s': %s", p.c_str(), err.empty() ? "empty file" : err.c_str());
  BitReader br;
  br.w = sf.words.data() + sf.blk_word[0];
  br.nbits = sf.blk_bits[0];
  uint64_t m1 = br.get(64), m2 = br.get(64);
  if (m1 != MERYL_IDX_MAGIC1)
    return mfx_fail(MFX_E_FORMAT, "'%s': bad magic %016lx (not a meryl index)", p.c_str(), (unsigned long)m1);
  // "ex__v.0N": 0x3N302e765f5f7865
  if ((m2 & 0xf0ffffffffffffffULL) != 0x30302e765f5f7865ULL)
    return mfx_fail(MFX_E_FORMAT, "'%s': unrecognised index version word %016lx", p.c_str(), (unsigned long)m2);
  mi.version = (int)((m2 >> 56) & 0x0f);
  if (mi.version < 1 || mi.version > 9)
    return mfx_fail(MFX_E_FORMAT, "'%s': unsupported meryl index version %d", p.c_str(), mi.version);
  mi.prefixSize = (uint32_t)br.get(32);
  mi.suffixSize = (uint32_t)br.get(32);
  mi.numFilesBits = (uint32_t)br.get(32);
  mi.numBlocksBits = (uint32_t)br.get(32);
  if (mi.version >= 2) mi.flags = (uint32_t)br.get(32);
  if (!br.ok || mi.numFilesBits != 6 || mi.prefixSize < 6 || mi.prefixSize + mi.suffixSize == 0 ||
      ((mi.prefixSize + mi.suffixSize) & 1) || mi.prefixSize + mi.suffixSize > 128 ||
      mi.numFilesBits + mi.numBlocksBits != mi.prefixSize)
    return mfx_fail(MFX_E_FORMAT, "'%s': inconsistent sizes (prefix %u suffix %u files %u blocks %u)", p.c_str(),
                    mi.prefixSize, mi.suffixSize, mi.numFilesBits, mi.numBlocksBits);
  if (br.pos + 192 <= br.nbits) {
    mi.numUnique = br.get(64);
    mi.numDistinct = br.get(64);
    mi.numTotal = br.get(64);
    mi.has_stats = br.ok;
    if (mi.has_stats && (mi.numUnique > mi.numDistinct || mi.numDistinct > mi.numTotal)) {
      if (!meryl_lenient())
        return mfx_fail(MFX_E_FORMAT, "'%s': statistics are not ordered (unique %lu <= distinct %lu <= total %lu expected); "
                        "MFX_MERYL_LENIENT=1 skips the statistics checks", p.c_str(), (unsigned long)mi.numUnique,
                        (unsigned long)mi.numDistinct, (unsigned long)mi.numTotal);
      mi.has_stats = false;
    }
  }
  return MFX_OK;
}

// Σ over the 64 files against the master statistics: a decoder (or a database) that disagrees with itself must
// not load quietly.  with_values: the values were decoded too (a full load), not only the block headers.
int check_meryl_sums(const std::string &dir, const MerylIndex &mi, const std::vector<MerylFileSums> &per, bool with_values) {
  if (!mi.has_stats) return MFX_OK;
  MerylFileSums s;
  for (const auto &x : per) { s.kmers += x.kmers; s.unique += x.unique; s.total += x.total; }
  std::string why;
  char b[256];
  if (s.kmers != mi.numDistinct) {
    snprintf(b, sizeof(b), "the data blocks hold %lu k-mers, the index statistics say %lu distinct", (unsigned long)s.kmers, (unsigned long)mi.numDistinct);
    why = b;
  } else if (with_values && s.total != mi.numTotal) {
    snprintf(b, sizeof(b), "the values sum to %lu, the index statistics say %lu total", (unsigned long)s.total, (unsigned long)mi.numTotal);
    why = b;
  } else if (with_values && s.unique != mi.numUnique) {
    snprintf(b, sizeof(b), "%lu k-mers have value 1, the index statistics say %lu unique", (unsigned long)s.unique, (unsigned long)mi.numUnique);
    why = b;
  }
  if (why.empty()) return MFX_OK;
  if (meryl_lenient()) {
    fprintf(stderr, "WARNING: meryl database '%s': %s (MFX_MERYL_LENIENT=1: continuing).\n", dir.c_str(), why.c_str());
    return MFX_OK;
  }
  return mfx_fail(MFX_E_FORMAT, "meryl database '%s' is inconsistent: %s (a truncated copy, or a layout this decoder does not "
                  "know; MFX_MERYL_LENIENT=1 downgrades this to a warning)", dir.c_str(), why.c_str());
}

std::string meryl_file_name(const std::string &dir, uint32_t file, const char *ext) {
  char b[16];
  for (int i = 0; i < 6; ++i) b[i] = ((file >> (5 - i)) & 1) ? '1' : '0';
  b[6] = 0;
  return dir + "/0x" + b + ext;
}

// decode every block of one .merylData file; emit(kmer, value)
// count_only: stop after each block's header (the k-mer count lives there)
template <class F>
int read_meryl_data(const std::string &path, uint32_t file, const MerylIndex &mi, F &&emit, MerylFileSums *sums, bool count_only = false) {
  FILE *f = fopen(path.c_str(), "rb");
  // meryl writes all 64 data files, empty pieces included: a missing one means a truncated / partially copied
  // database, which must not load as "no k-mers here"
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s': a meryl database has all 64 data files (partial copy?)", path.c_str());
  StuffedFile sf;
  std::string err;
  int rc = MFX_OK;
  bool have_prev = false;
  uint64_t prev_prefix = 0;
  while (rc == MFX_OK && read_stuffed(f, sf, &err)) {
    // a data block may be split over several stuffedBits chunks; treat them as one bit stream
    std::vector<uint64_t> stream;
    uint64_t bits = 0;
    bool aligned = true;
    for (size_t b = 0; b < sf.blk_word.size(); ++b) {
      if (!aligned) { rc = mfx_fail(MFX_E_FORMAT, "'%s': non-word-aligned interior chunk", path.c_str()); break; }
      uint64_t nw = (sf.blk_bits[b] + 63) / 64;
      stream.insert(stream.end(), sf.words.begin() + sf.blk_word[b], sf.words.begin() + sf.blk_word[b] + nw);
      bits += sf.blk_bits[b];
      aligned = (sf.blk_bits[b] % 64) == 0;
    }
    if (rc) break;
    stream.push_back(0);
    BitReader br;
    br.w = stream.data();
    br.nbits = bits;
    if (br.get(64) != MERYL_DAT_MAGIC1 || br.get(64) != MERYL_DAT_MAGIC2) { rc = mfx_fail(MFX_E_FORMAT, "'%s': bad data-block magic", path.c_str()); break; }
    uint64_t prefix = br.get(64), nk = br.get(64);
    uint32_t kcode = (uint32_t)br.get(8), ubits = (uint32_t)br.get(32), bbits = (uint32_t)br.get(32);
    (void)br.get(64);                                        // k1 (unused)
    uint32_t ccode = (uint32_t)br.get(8);
    (void)br.get(64); (void)br.get(64);                      // c1, c2 (unused)
    if (!br.ok || kcode != 1 || (ccode != 1 && ccode != 2) || ubits + bbits != mi.suffixSize || bbits > 128 ||
        (prefix >> mi.numBlocksBits) != file || mi.suffixSize > 122) {
      rc = mfx_fail(MFX_E_FORMAT, "'%s': unsupported / inconsistent data block (kCode %u cCode %u unary %u binary %u suffix %u prefix %lx)",
                    path.c_str(), kcode, ccode, ubits, bbits, mi.suffixSize, (unsigned long)prefix);
      break;
    }
    // blocks appear in increasing prefix order, so the k-mers of a file are strictly increasing across blocks too
    if (have_prev && prefix <= prev_prefix) {
      rc = mfx_fail(MFX_E_FORMAT, "'%s': block prefixes not strictly increasing (%lx after %lx)", path.c_str(), (unsigned long)prefix, (unsigned long)prev_prefix);
      break;
    }
    have_prev = true;
    prev_prefix = prefix;
    if (mi.prefixSize < 64 && (prefix >> mi.prefixSize) != 0) {      // prefixSize >= 64: every 64-bit prefix fits (and the shift would be undefined)
      rc = mfx_fail(MFX_E_FORMAT, "'%s': block prefix %lx wider than %u bits", path.c_str(), (unsigned long)prefix, mi.prefixSize);
      break;
    }
    sums->blocks++;
    // a k-mer of the block takes at least one bit of unary code, its binary part and its count: a header that claims more
    // k-mers than the block has bits for is damaged (and must not size an allocation)
    if (nk > (br.nbits - br.pos) / (1ull + bbits + (ccode == 1 ? 32u : 64u))) {
      rc = mfx_fail(MFX_E_FORMAT, "'%s': a data block of %lu bits claims %lu k-mers", path.c_str(), (unsigned long)br.nbits, (unsigned long)nk);
      break;
    }
    if (count_only) { sums->kmers += nk; continue; }
    std::vector<u128> sfx(nk);
    u128 hi = 0;
    for (uint64_t i = 0; i < nk; ++i) {
      hi += br.unary();
      // the binary part is read in two pieces when it is wider than a word (k > 32 + prefix bits)
      u128 bin = bbits > 64 ? (((u128)br.get(bbits - 64) << 64) | br.get(64)) : (u128)br.get(bbits);
      sfx[i] = (bbits >= 128 ? (u128)0 : (hi << bbits)) | bin;
      if (i && sfx[i] <= sfx[i - 1]) { rc = mfx_fail(MFX_E_FORMAT, "'%s': suffixes not strictly increasing", path.c_str()); break; }
    }
    if (rc) break;
    const u128 smask = ((u128)1 << mi.suffixSize) - 1;          // suffixSize <= 122
    if (nk && (sfx[nk - 1] & ~smask)) { rc = mfx_fail(MFX_E_FORMAT, "'%s': suffix wider than %u bits", path.c_str(), mi.suffixSize); break; }
    uint64_t uniq = 0, tot = 0;
    for (uint64_t i = 0; i < nk; ++i) {
      uint64_t v = br.get(ccode == 1 ? 32 : 64);
      const u128 km = ((u128)prefix << mi.suffixSize) | sfx[i];
      uniq += (v == 1);
      tot += v;
      emit((uint64_t)km, (uint64_t)(km >> 64), v > 0xffffffffull ? 0xffffffffu : (uint32_t)v);
    }
    if (!br.ok) { rc = mfx_fail(MFX_E_FORMAT, "'%s': data block shorter than its header claims", path.c_str()); break; }
    sums->kmers += nk;
    sums->unique += uniq;
    sums->total += tot;
  }
  fclose(f);
  if (rc == MFX_OK && !err.empty()) rc = mfx_fail(MFX_E_FORMAT, "'%s': %s", path.c_str(), err.c_str());
  return rc;
}

// The 64 files of a meryl database are independent: decode them on the host threads the library may use.
// fn(file) returns MFX_OK or a failure whose message is in the CALLING thread's mfx_last_error(); the first
// failure (lowest file number wins) is re-raised on this thread.
template <class F>
int for_each_meryl_file(F &&fn) {
  const unsigned nt = std::max(1u, std::min(64u, (unsigned)mfx_host_threads()));
  std::atomic<uint32_t> next(0);
  std::mutex emu;
  int bad_rc = MFX_OK;
  uint32_t bad_file = 64;
  std::string bad_msg;
  auto work = [&]() {
    for (uint32_t fl; (fl = next.fetch_add(1)) < 64;) {
      int rc;
      try { rc = fn(fl); }                                     // (nothing may leave a worker thread -- or the C ABI -- as an exception)
      catch (const std::bad_alloc &) { rc = mfx_fail(MFX_E_NOMEM, "meryl data file %u: out of memory", fl); }
      catch (const std::exception &e) { rc = mfx_fail(MFX_E_FORMAT, "meryl data file %u: %s", fl, e.what()); }
      if (rc != MFX_OK) {
        std::lock_guard<std::mutex> g(emu);
        if (fl < bad_file) { bad_file = fl; bad_rc = rc; bad_msg = mfx_last_error(); }
      }
    }
  };
  if (nt == 1) work();
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work);
    for (auto &x : th) x.join();
  }
  return bad_rc == MFX_OK ? MFX_OK : mfx_fail(bad_rc, "%s", bad_msg.c_str());
}

int detect(const std::string &path) {
  if (is_dir(path)) return MFX_DB_MERYL;
  if (!is_file(path)) return 0;
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return 0;
  char m[8] = {0};
  size_t r = fread(m, 1, 8, f);
  fclose(f);
  if (r == 8 && memcmp(m, "MFXKMER1", 8) == 0) return MFX_DB_FLAT;
  return MFX_DB_TEXT;
}

// batches (kmer,value) pairs into the index
struct Feeder {
  mfx_index *const *ixs;               // one table, or the shards of one process: every batch goes to all of them
  uint32_t nix;
  int side;
  uint64_t minV, maxV;
  std::mutex *mu = nullptr;            // several decoder threads feed one index: inserts are serialised
  std::vector<uint64_t> k;             // 1 word per k-mer, 2 (low, high) when the index holds k > 31
  std::vector<uint32_t> v;
  int rc = MFX_OK;
  void flush() {
    if (v.empty() || rc) return;
    std::unique_lock<std::mutex> lk;
    if (mu) lk = std::unique_lock<std::mutex>(*mu);
    const bool timing = getenv("MFX_DB_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    rc = mfx_index_add_multi(ixs, nix, k.data(), v.data(), v.size(), side, minV, maxV);
    if (timing) fprintf(stderr, "[mfx db] batch of %zu k-mers inserted in %.1f ms\n", v.size(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    k.clear(); v.clear();
  }
  void push(uint64_t lo, uint64_t hi, uint32_t val) {
    k.push_back(lo);
    if (ixs[0]->wide()) k.push_back(hi);
    v.push_back(val);
    if (v.size() >= (1u << 24)) flush();
  }
};

// One line of `meryl print` text at [p, end): "<kmer>\t<count>".  Returns 1 = k-mer parsed, 0 = blank line, -1 = malformed.
inline int parse_text_line(const char *p, const char *end, u128 &km, int &n, unsigned long long &v) {
  km = 0;
  n = 0;
  while (p < end && base_code((unsigned char)*p) >= 0) { km = (km << 2) | (u128)base_code((unsigned char)*p); ++p; ++n; }
  if (n == 0 && (p == end || *p == '\n' || *p == '\r')) return 0;
  if (n == 0 || n > MFX_MAX_K || p == end || (*p != '\t' && *p != ' ')) return -1;
  while (p < end && (*p == '\t' || *p == ' ')) ++p;
  if (p == end || *p < '0' || *p > '9') return -1;
  v = 0;
  while (p < end && *p >= '0' && *p <= '9') { v = v > (~0ull - 9) / 10 ? ~0ull : v * 10 + (unsigned)(*p - '0'); ++p; }
  return 1;
}

template <class F>
int scan_text(const std::string &path, int *k_out, F &&emit, uint64_t *count) {
  mfx_file h = mfx_open_reader(path.c_str());          // compressedFileReader: decompressor chosen by suffix, no shell
  FILE *f = h.f;
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s'", path.c_str());
  char line[512];
  int k = 0, rc = MFX_OK;
  uint64_t ln = 0;
  while (fgets(line, sizeof(line), f)) {
    ++ln;
    u128 km;
    int n;
    unsigned long long v = 0;
    const int what = parse_text_line(line, line + strlen(line), km, n, v);
    if (what == 0) continue;
    if (what < 0) { rc = mfx_fail(MFX_E_FORMAT, "'%s' line %lu: expected '<kmer>\\t<count>'", path.c_str(), (unsigned long)ln); break; }
    if (k == 0) k = n;
    if (n != k) { rc = mfx_fail(MFX_E_FORMAT, "'%s' line %lu: k-mer length %d differs from %d", path.c_str(), (unsigned long)ln, n, k); break; }
    emit((uint64_t)km, (uint64_t)(km >> 64), v > 0xffffffffull ? 0xffffffffu : (uint32_t)v);
    ++*count;
  }
  if (mfx_close(h, rc != MFX_OK) && rc == MFX_OK)
    rc = mfx_fail(MFX_E_IO, "reading '%s' failed (stream error or the decompressor exited with an error)", path.c_str());
  if (k_out) *k_out = k;
  return rc;
}

// An UNCOMPRESSED `meryl print` file parsed by the host threads: the file is cut into 32 MB pieces, a piece belongs to
// the thread that draws it and covers the lines that START in it (pread of the piece plus the tail of its last
// line).  make(t) returns thread t's sink: sink(lo, hi, v) per k-mer, sink.done() once (its return value is the
// thread's status).  A 6 G-line database (150 GB of text) is minutes of single-threaded fgets otherwise.
// on_piece(t, pc), if given, tells that thread t starts piece pc (a sink that wants the file's order back collects by piece).
inline uint64_t text_piece_bytes() {
  uint64_t PIECE = 32ull << 20;
  if (const char *e = getenv("MFX_TEXT_PIECE")) { const long v = atol(e); if (v >= 64) PIECE = (uint64_t)v; }   // tests: many pieces of a small file
  return PIECE;
}

template <class Make>
int scan_text_parallel(const std::string &path, int *k_out, uint64_t *count, Make &&make,
                       const std::function<void(unsigned, uint64_t)> &on_piece = nullptr) {
  const int fd = open(path.c_str(), O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return mfx_fail(MFX_E_IO, "cannot open '%s'", path.c_str()); }
  const uint64_t PIECE = text_piece_bytes();
  const uint64_t size = (uint64_t)st.st_size, TAIL = 4096;
  const uint64_t npieces = (size + PIECE - 1) / PIECE;
  const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({npieces, 64, (uint64_t)mfx_host_threads()}));
  std::atomic<uint64_t> next(0), total(0);
  std::atomic<int> kk(0);
  std::mutex emu;
  int bad_rc = MFX_OK;
  uint64_t bad_at = ~0ull;
  std::string bad_msg;
  auto fail_at = [&](uint64_t at, int rc, const std::string &msg) {
    std::lock_guard<std::mutex> g(emu);
    if (at < bad_at) { bad_at = at; bad_rc = rc; bad_msg = msg; }
  };
  auto work = [&](unsigned t) {
    auto sink = make(t);
    std::vector<char> buf(PIECE + TAIL + 1);
    uint64_t mine = 0;
    for (uint64_t pc; (pc = next.fetch_add(1)) < npieces;) {
      { std::lock_guard<std::mutex> g(emu); if (bad_rc != MFX_OK) break; }
      if (on_piece) on_piece(t, pc);
      const uint64_t b = pc * PIECE, want = std::min(size - (b ? b - 1 : 0), PIECE + TAIL + (b ? 1 : 0));
      // one byte before the piece tells whether a line starts exactly at b
      const uint64_t from = b ? b - 1 : 0;
      uint64_t got = 0;
      while (got < want) {
        ssize_t r = pread(fd, buf.data() + got, want - got, (off_t)(from + got));
        if (r <= 0) break;
        got += (uint64_t)r;
      }
      if (got < want) { fail_at(b, MFX_E_IO, "reading '" + path + "' failed"); break; }
      const char *p = buf.data(), *end = buf.data() + got;
      const char *lim = buf.data() + std::min<uint64_t>(got, (b ? 1 : 0) + PIECE);     // lines starting before lim are this piece's
      if (b) {                                               // skip the line that started in the previous piece
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        p = nl ? nl + 1 : end;
      }
      while (p < lim) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        if (!nl && from + got < size) { fail_at(from + (uint64_t)(p - buf.data()), MFX_E_FORMAT, "'" + path + "': a line longer than 4096 bytes"); break; }
        u128 km;
        int n;
        unsigned long long v;
        const int what = parse_text_line(p, le, km, n, v);
        if (what < 0) {
          char where[64];
          snprintf(where, sizeof(where), "%lu", (unsigned long)(from + (uint64_t)(p - buf.data())));
          fail_at(from + (uint64_t)(p - buf.data()), MFX_E_FORMAT, "'" + path + "' at byte " + where + ": expected '<kmer>\\t<count>'");
          break;
        }
        if (what > 0) {
          int k0 = kk.load(std::memory_order_relaxed);          // (a CAS per line would bounce the cache line between all threads)
          if (k0 == 0) kk.compare_exchange_strong(k0, n);
          if (k0 != 0 && k0 != n) {
            char msg[160];
            snprintf(msg, sizeof(msg), "' at byte %lu: k-mer length %d differs from %d", (unsigned long)(from + (uint64_t)(p - buf.data())), n, k0);
            fail_at(from + (uint64_t)(p - buf.data()), MFX_E_FORMAT, "'" + path + msg);
            break;
          }
          sink((uint64_t)km, (uint64_t)(km >> 64), v > 0xffffffffull ? 0xffffffffu : (uint32_t)v);
          ++mine;
        }
        p = le + 1;
      }
    }
    const int rc = sink.done();
    if (rc != MFX_OK) fail_at(~0ull - 1, rc, mfx_last_error());
    total.fetch_add(mine);
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  close(fd);
  if (k_out) *k_out = kk.load();
  if (count) *count = total.load();
  return bad_rc == MFX_OK ? MFX_OK : mfx_fail(bad_rc, "%s", bad_msg.c_str());
}

inline bool text_is_plain(const std::string &path) { return mfx_suffix_tool(path) == nullptr; }


// ---- delta-coded flat form ---------------------------------------------------
struct DeltaBlockPlan { uint8_t kb = 0, vb = 2; uint32_t nesc = 0; uint64_t bytes = 0; };

inline uint32_t bit_length(uint64_t x) { return x ? 64u - (uint32_t)__builtin_clzll(x) : 0u; }

// widths and size of block [b, b + cnt) of a strictly ascending database
DeltaBlockPlan plan_delta_block(const uint64_t *kmers, const uint32_t *values, uint32_t cnt) {
  DeltaBlockPlan p;
  uint64_t maxd = 0;
  for (uint32_t i = 1; i < cnt; ++i) maxd = std::max(maxd, kmers[i] - kmers[i - 1]);
  p.kb = (uint8_t)bit_length(maxd);
  uint32_t hist[34] = {0};                                    // bit length of value + 1: a value fits vb bits iff that is <= vb
  for (uint32_t i = 0; i < cnt; ++i) ++hist[bit_length((uint64_t)values[i] + 1)];
  uint64_t best = ~0ull;
  uint32_t above = 0;                                         // values that need more than vb bits
  for (int vb = 33; vb >= 2; --vb) {
    if (vb <= MFX_DELTA_MAX_VBITS) {
      const uint64_t cost = (uint64_t)cnt * vb + 96ull * above;
      if (cost <= best) { best = cost; p.vb = (uint8_t)vb; p.nesc = above; }
    }
    above += hist[vb];
  }
  const uint64_t kwords = ((uint64_t)(cnt - 1) * p.kb + 63) / 64, vwords = ((uint64_t)cnt * p.vb + 63) / 64;
  p.bytes = (kwords + vwords) * 8;
  return p;
}

void put_bits(uint64_t *w, uint64_t bit, uint32_t nbits, uint64_t x) {       // into zeroed words
  const uint64_t i = bit >> 6;
  const uint32_t sh = (uint32_t)bit & 63u;
  w[i] |= x << sh;
  if (sh + nbits > 64u) w[i + 1] |= x >> (64u - sh);
}

void pack_delta_block(const uint64_t *kmers, const uint32_t *values, uint32_t cnt, const DeltaBlockPlan &p, uint64_t *out) {
  memset(out, 0, p.bytes);
  for (uint32_t i = 1; i < cnt; ++i) if (p.kb) put_bits(out, (uint64_t)(i - 1) * p.kb, p.kb, kmers[i] - kmers[i - 1]);
  uint64_t *vw = out + ((uint64_t)(cnt - 1) * p.kb + 63) / 64;
  const uint32_t esc = (1u << p.vb) - 1u;
  for (uint32_t i = 0; i < cnt; ++i) put_bits(vw, (uint64_t)i * p.vb, p.vb, values[i] >= esc ? esc : values[i]);
}

template <class F>
void par_blocks(uint64_t nblocks, F &&fn) {                   // fn(b) for every block, on the library's host threads
  const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({(nblocks + 255) / 256, 64, (uint64_t)mfx_host_threads()}));
  if (nt == 1) { for (uint64_t b = 0; b < nblocks; ++b) fn(b); return; }
  std::atomic<uint64_t> nextb{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (uint64_t b0 = nextb.fetch_add(256); b0 < nblocks; b0 = nextb.fetch_add(256))
        for (uint64_t b = b0; b < std::min(nblocks, b0 + 256); ++b) fn(b);
    });
  for (auto &x : th) x.join();
}

// 0 = written; 1 = the k-mers are not strictly ascending (the caller writes another form); < 0: error
// placed: `kmers` are the numbers P of mfx_place.h (ascending); the escape list then holds the k-mers they decode to
// sbits (placed, k = 31: mfx_place.h, mfx_p_split): the strand bit of every record; `kmers` are then the numbers P >> 1 -- ascending, equal for a
// pair that differs in the strand bit alone (0 first) -- and a record's count field holds count << 1 | s (a count of 2^21 or more is an escape)
int write_flat_delta(FILE *f, const char *path, FlatHeader h, const uint64_t *kmers, const uint32_t *true_values, uint64_t n, bool placed = false,
                     const uint8_t *sbits = nullptr) {
  std::vector<uint32_t> folded;
  if (sbits) {
    folded.resize(n);
    par_blocks((n + 65535) / 65536, [&](uint64_t b) {
      for (uint64_t i = b * 65536, e = std::min<uint64_t>(n, i + 65536); i < e; ++i)
        folded[i] = true_values[i] >= (1u << (MFX_DELTA_MAX_VBITS - 1)) ? 0xffffffffu : (true_values[i] << 1) | (uint32_t)(sbits[i] & 1u);
    });
  }
  const uint32_t *values = sbits ? folded.data() : true_values;
  const uint64_t nblocks = (n + MFX_DELTA_BLOCK - 1) / MFX_DELTA_BLOCK;
  auto cnt_of = [&](uint64_t b) { return (uint32_t)std::min<uint64_t>(MFX_DELTA_BLOCK, n - b * MFX_DELTA_BLOCK); };
  const bool timing = getenv("MFX_DB_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tw0 = now();
  double t_pack = 0, t_io = 0;
  std::vector<DeltaBlockPlan> plan(nblocks);
  std::atomic<int> unsorted{0};
  par_blocks(nblocks, [&](uint64_t b) {
    const uint64_t o = b * MFX_DELTA_BLOCK;
    const uint32_t cnt = cnt_of(b);
    for (uint32_t i = (b ? 0 : 1); i < cnt; ++i)
      if (kmers[o + i] < kmers[o + i - 1] || (kmers[o + i] == kmers[o + i - 1] && !(sbits && sbits[o + i - 1] < sbits[o + i]))) { unsorted = 1; return; }
    plan[b] = plan_delta_block(kmers + o, values + o, cnt);
  });
  if (unsorted) return 1;
  const double tw1 = now();
  std::vector<uint64_t> dir(2 * (nblocks + 1));
  uint64_t at = sizeof(FlatHeader) + 8 + dir.size() * 8;
  h.n_escape = 0;
  for (uint64_t b = 0; b < nblocks; ++b) {
    dir[2 * b] = kmers[b * MFX_DELTA_BLOCK];
    dir[2 * b + 1] = at | ((uint64_t)plan[b].kb << 48) | ((uint64_t)plan[b].vb << 56);
    at += plan[b].bytes;
    h.n_escape += plan[b].nesc;
  }
  dir[2 * nblocks] = 0;
  dir[2 * nblocks + 1] = at;
  if (at >> 48) return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat: the database is too large for the delta form");
  h.flags |= FLAT_DELTA;
  if (placed) h.flags |= FLAT_PLACED | (MFX_PLACE_VERSION << 8);
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(&nblocks, 8, 1, f) == 1 && fwrite(dir.data(), 8, dir.size(), f) == dir.size();
  // the blocks, packed by all threads a piece (<= 256 MB) at a time
  std::vector<uint64_t> buf;
  for (uint64_t b0 = 0; b0 < nblocks && ok;) {
    uint64_t b1 = b0, bytes = 0;
    while (b1 < nblocks && (b1 == b0 || bytes + plan[b1].bytes <= (256ull << 20))) bytes += plan[b1++].bytes;
    buf.resize(bytes / 8);
    const uint64_t base = dir[2 * b0 + 1] & 0xffffffffffffull;
    const double tp0 = now();
    par_blocks(b1 - b0, [&](uint64_t i) {
      const uint64_t b = b0 + i;
      pack_delta_block(kmers + b * MFX_DELTA_BLOCK, values + b * MFX_DELTA_BLOCK, cnt_of(b), plan[b],
                       buf.data() + ((dir[2 * b + 1] & 0xffffffffffffull) - base) / 8);
    });
    const double tp1 = now();
    ok = fwrite(buf.data(), 8, buf.size(), f) == buf.size();
    t_pack += tp1 - tp0; t_io += now() - tp1;
    b0 = b1;
  }
  // the escapes: the counts that did not fit their block's field
  std::vector<uint64_t> ek;
  std::vector<uint32_t> ev;
  for (uint64_t b = 0; b < nblocks; ++b) {
    if (!plan[b].nesc) continue;
    const uint32_t esc = (1u << plan[b].vb) - 1u;
    for (uint64_t i = b * MFX_DELTA_BLOCK, e = i + cnt_of(b); i < e; ++i)
      if (values[i] >= esc) {
        uint32_t top, hi, pm;
        ek.push_back(placed ? mfx_p_decode_s((int)h.k, kmers[i], sbits ? sbits[i] : 0u, top, hi, pm) : kmers[i]);
        ev.push_back(true_values[i]);
      }
  }
  ok = ok && ek.size() == h.n_escape &&
       (ek.empty() || (fwrite(ek.data(), 8, ek.size(), f) == ek.size() && fwrite(ev.data(), 4, ev.size(), f) == ev.size()));
  if (timing) fprintf(stderr, "[mfx db] delta writer: plan %.2f s, pack %.2f s, write %.2f s, all %.2f s (%lu blocks)\n", tw1 - tw0, t_pack, t_io, now() - tw0,
                      (unsigned long)nblocks);
  return ok ? 0 : mfx_fail(MFX_E_IO, "short write to '%s'", path);
}

// the escape list of a packed / delta-coded file (n_escape was bounded by the file's size: flat_header_problem)
int load_flat_escapes(mfx_index *const *ixs, uint32_t nix, int fd, const char *path, const FlatHeader &h, uint64_t at, int side,
                      uint64_t minV, uint64_t maxV) {
  std::vector<uint64_t> ek;
  std::vector<uint32_t> ev;
  try { ek.resize(h.n_escape); ev.resize(h.n_escape); }
  catch (const std::exception &) { return mfx_fail(MFX_E_NOMEM, "'%s': no memory for %lu escapes", path, (unsigned long)h.n_escape); }
  if (pread(fd, ek.data(), h.n_escape * 8, (off_t)at) != (ssize_t)(h.n_escape * 8) ||
      pread(fd, ev.data(), h.n_escape * 4, (off_t)(at + h.n_escape * 8)) != (ssize_t)(h.n_escape * 4))
    return mfx_fail(MFX_E_IO, "reading '%s' failed", path);
  for (uint64_t i = 0; i < h.n_escape; ++i)
    if (!flat_key_fits(ek[i], h.k)) return mfx_fail(MFX_E_FORMAT, "'%s': an escaped k-mer is wider than 2k bits", path);
  return mfx_index_add_multi(ixs, nix, ek.data(), ev.data(), h.n_escape, side, minV, maxV);
}

// the block directory of a delta-coded file, read and checked; *expect_out: where the blocks end (the escape list follows)
int read_delta_directory(int fd, const char *path, const FlatHeader &h, uint64_t fsize, std::vector<uint64_t> &dir, uint64_t *nblocks_out,
                         uint64_t *expect_out) {
  auto bad = [&](const char *what) { return mfx_fail(MFX_E_FORMAT, "'%s': %s", path, what); };
  uint64_t nblocks = 0;
  if (h.k > (uint32_t)MFX_MAX_K_NARROW || fsize < sizeof(h) + 8 || pread(fd, &nblocks, 8, sizeof(h)) != 8) return bad("truncated delta-coded payload");
  if (nblocks != (h.n + MFX_DELTA_BLOCK - 1) / MFX_DELTA_BLOCK) return bad("block count does not match the k-mer count");
  const uint64_t dir_off = sizeof(h) + 8, dir_bytes = (nblocks + 1) * 16;
  if (fsize < dir_off + dir_bytes) return bad("truncated block directory");
  try { dir.resize(2 * (nblocks + 1)); } catch (const std::exception &) { return mfx_fail(MFX_E_NOMEM, "'%s': no memory for %lu directory entries", path, (unsigned long)nblocks); }
  for (uint64_t o = 0; o < dir_bytes;) {
    const ssize_t r = pread(fd, (char *)dir.data() + o, dir_bytes - o, (off_t)(dir_off + o));
    if (r <= 0) return mfx_fail(MFX_E_IO, "reading '%s' failed", path);
    o += (uint64_t)r;
  }
  // the directory is checked before anything reads by it: offsets ascending, 8-byte aligned, inside the file, and every
  // block exactly as long as its widths say
  uint64_t expect = dir_off + dir_bytes;
  for (uint64_t b = 0; b <= nblocks; ++b) {
    const uint64_t off = dir[2 * b + 1] & 0xffffffffffffull;
    if (off != expect) return bad("inconsistent block directory");
    if (b == nblocks) break;
    const uint32_t kb = (uint32_t)(dir[2 * b + 1] >> 48) & 0xffu, vb = (uint32_t)(dir[2 * b + 1] >> 56) & 0xffu;
    const uint64_t cnt = std::min<uint64_t>(MFX_DELTA_BLOCK, h.n - b * MFX_DELTA_BLOCK);
    if (kb > flat_rec_bits(h) || vb < 2u || vb > (uint32_t)MFX_DELTA_MAX_VBITS) return bad("block field widths out of range");
    if (!flat_rec_fits(dir[2 * b], h)) return bad("a block's first k-mer is wider than 2k bits");
    // (the k-mers inside a block exist only in the kernel that decodes them: it refuses what is wider than 2k bits, index meta[4])
    if (b + 1 < nblocks && dir[2 * (b + 1)] <= dir[2 * b]) return bad("block directory not ascending");
    expect = off + (((cnt - 1) * kb + 63) / 64 + (cnt * vb + 63) / 64) * 8;
    if (expect > fsize) return bad("truncated delta-coded payload");
  }
  if (expect > fsize || fsize - expect < h.n_escape * 12) return bad("truncated delta-coded payload");
  *nblocks_out = nblocks;
  *expect_out = expect;
  return MFX_OK;
}

int load_flat_delta(mfx_index *const *ixs, uint32_t nix, int fd, const char *path, const FlatHeader &h, uint64_t fsize, int side,
                    uint64_t minV, uint64_t maxV) {
  std::vector<uint64_t> dir;
  uint64_t nblocks = 0, expect = 0;
  int rc = read_delta_directory(fd, path, h, fsize, dir, &nblocks, &expect);
  if (rc) return rc;
  if (h.n) rc = mfx_index_add_delta_file(ixs, nix, fd, path, dir.data(), nblocks, h.n, side, minV, maxV, (h.flags & FLAT_PLACED) ? 1 : 0);
  if (rc == MFX_OK && h.n_escape) rc = load_flat_escapes(ixs, nix, fd, path, h, expect, side, minV, maxV);
  return rc;
}

}  // namespace

// ---- for the staged load (mfx_api.cpp: mfx_db_stage) ----------------------------------------------------------------------
// opens `path` if it is a delta-coded flat database and reads + checks its directory; MFX_E_FORMAT with *fd_out = -1 when it is
// another kind of database (the caller then takes the ordinary load)
int mfx_flat_delta_open(const char *path, int *fd_out, mfx_flat_delta_info *info, std::vector<uint64_t> &dir) {
  *fd_out = -1;
  if (detect(std::string(path)) != MFX_DB_FLAT) return mfx_fail(MFX_E_FORMAT, "'%s' is not a flat k-mer file", path);
  int fd = open(path, O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st) != 0) { if (fd >= 0) close(fd); return mfx_fail(MFX_E_IO, "cannot open '%s'", path); }
  FlatHeader h;
  if ((uint64_t)st.st_size < sizeof(h) || pread(fd, &h, sizeof(h), 0) != (ssize_t)sizeof(h)) { close(fd); return mfx_fail(MFX_E_FORMAT, "'%s': truncated header", path); }
  if (const char *why = flat_header_problem(h, (uint64_t)st.st_size)) { close(fd); return mfx_fail(MFX_E_FORMAT, "'%s': %s", path, why); }
  if (!(h.flags & FLAT_DELTA)) { close(fd); return mfx_fail(MFX_E_FORMAT, "'%s' is not delta-coded", path); }
  uint64_t nblocks = 0, expect = 0;
  if (int rc = read_delta_directory(fd, path, h, (uint64_t)st.st_size, dir, &nblocks, &expect)) { close(fd); return rc; }
  info->k = (int)h.k;
  info->placed = (h.flags & FLAT_PLACED) ? 1 : 0;
  info->n = h.n;
  info->n_escape = h.n_escape;
  info->nblocks = nblocks;
  info->escapes_off = expect;
  info->fsize = (uint64_t)st.st_size;
  *fd_out = fd;
  return MFX_OK;
}

// merylFileReader(path): opens the DB and reveals k (merfin-globals.C:118-119:
// "Make readDB first so we know the k size").
static int mfx_db_probe_impl(const char *path, mfx_db_info *out) {
  if (!path || !out) return mfx_fail(MFX_E_INVAL, "mfx_db_probe: null argument");
  memset(out, 0, sizeof(*out));
  std::string p(path);
  int fmt = detect(p);
  if (!fmt) return mfx_fail(MFX_E_IO, "k-mer database '%s' does not exist", path);
  out->format = fmt;
  if (fmt == MFX_DB_FLAT) {
    FILE *f = fopen(path, "rb");
    FlatHeader h;
    if (!f || fread(&h, sizeof(h), 1, f) != 1) { if (f) fclose(f); return mfx_fail(MFX_E_FORMAT, "'%s': truncated header", path); }
    fclose(f);
    out->k = (int)h.k;
    out->n_kmers = h.n;
    out->placed = (memcmp(h.magic, "MFXKMER1", 8) == 0 && (h.flags & FLAT_PLACED)) ? 1 : 0;
    return MFX_OK;
  }
  if (fmt == MFX_DB_TEXT) {
    uint64_t n = 0;
    int k = 0;
    struct Count { void operator()(uint64_t, uint64_t, uint32_t) {} int done() { return MFX_OK; } };
    int rc = text_is_plain(p) ? scan_text_parallel(p, &k, &n, [](unsigned) { return Count(); })
                              : scan_text(p, &k, [](uint64_t, uint64_t, uint32_t) {}, &n);
    if (rc) return rc;
    if (k == 0) return mfx_fail(MFX_E_FORMAT, "'%s': no k-mers found", path);
    out->k = k;
    out->n_kmers = n;
    return MFX_OK;
  }
  MerylIndex mi;
  int rc = read_meryl_master(p, mi);
  if (rc) return rc;
  out->k = (int)((mi.prefixSize + mi.suffixSize) / 2);
  // distinct k-mer count: sum of the block headers (cheap pass over the 64 data files)
  std::vector<MerylFileSums> per(64);
  rc = for_each_meryl_file([&](uint32_t fl) {
    return read_meryl_data(meryl_file_name(p, fl, ".merylData"), fl, mi, [](uint64_t, uint64_t, uint32_t) {}, &per[fl], true);
  });
  uint64_t n = 0;
  for (const auto &x : per) n += x.kmers;
  out->n_kmers = n;
  if (rc == MFX_OK) rc = check_meryl_sums(p, mi, per, false);
  return rc;
}

extern "C" int mfx_db_probe(const char *path, mfx_db_info *out) {
  try { return mfx_db_probe_impl(path, out); }                       // (nothing leaves the C ABI as an exception)
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_db_probe: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_db_probe: %s", e.what()); }
}


// merylExactLookup::load (merfin-globals.C:156,159): side 0 = read DB with the
// -min/-max filter, side 1 = assembly DB.
extern "C" int mfx_index_load_db(mfx_index *ix, const char *path, int side, uint64_t minV, uint64_t maxV) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_load_db: null argument");
  return mfx_index_load_db_multi(&ix, 1, path, side, minV, maxV);
}

// One pass over the database feeds nix tables: the shards of one process each keep the k-mers they own
// (mfx_index_set_shard), so a config-5-sized read database is decoded ONCE, not once per GPU.
static int mfx_index_load_db_multi_impl(mfx_index *const *ixs, uint32_t nix, const char *path, int side, uint64_t minV, uint64_t maxV) {
  if (!ixs || nix == 0 || !path) return mfx_fail(MFX_E_INVAL, "mfx_index_load_db: null argument");
  for (uint32_t i = 0; i < nix; ++i)
    if (!ixs[i] || ixs[i]->k != ixs[0]->k) return mfx_fail(MFX_E_INVAL, "mfx_index_load_db_multi: the indexes of one load must hold the same k");
  mfx_index *ix = ixs[0];
  std::string p(path);
  int fmt = detect(p);
  if (!fmt) return mfx_fail(MFX_E_IO, "k-mer database '%s' does not exist", path);
  Feeder fd{ixs, nix, side, minV, maxV};
  uint64_t n = 0;
  int rc = MFX_OK;
  if (fmt == MFX_DB_FLAT) {
    // Several threads pread() slices of the file straight into the index's pinned staging lanes (no mapping to
    // fault in, no per-call pinning), overlapped with the PCIe transfer and the insert kernel of the previous chunk.
    int fdn = open(path, O_RDONLY);
    struct stat st;
    if (fdn < 0 || fstat(fdn, &st) != 0) { if (fdn >= 0) close(fdn); return mfx_fail(MFX_E_IO, "cannot open '%s'", path); }
    FlatHeader h;
    if ((uint64_t)st.st_size < sizeof(h) || pread(fdn, &h, sizeof(h), 0) != (ssize_t)sizeof(h)) {
      close(fdn);
      return mfx_fail(MFX_E_FORMAT, "'%s': truncated header", path);
    }
    if (const char *why = flat_header_problem(h, (uint64_t)st.st_size)) { close(fdn); return mfx_fail(MFX_E_FORMAT, "'%s': %s", path, why); }
    if ((int)h.k != ix->k) { close(fdn); return mfx_fail(MFX_E_INVAL, "'%s' holds %u-mers but the index is built for k=%d", path, h.k, ix->k); }
    const uint64_t kw = ix->key_words();                     // k > 31: 16-byte k-mers {low, high}
    if (h.flags & FLAT_DELTA) {
      rc = load_flat_delta(ixs, nix, fdn, path, h, (uint64_t)st.st_size, side, minV, maxV);
      close(fdn);
      return rc;
    }
    if (h.flags & FLAT_PACKED) {
      if (h.k > (uint32_t)MFX_MAX_K_PACKED || (uint64_t)st.st_size < sizeof(h) + h.n * 8 + h.n_escape * 12) {
        close(fdn);
        return mfx_fail(MFX_E_FORMAT, "'%s': truncated or inconsistent packed payload", path);
      }
      if (h.n) rc = mfx_index_add_from_file(ixs, nix, fdn, path, sizeof(h), 0, h.n, side, minV, maxV);
      if (rc == MFX_OK && h.n_escape)                        // the few counts beyond the record's field
        rc = load_flat_escapes(ixs, nix, fdn, path, h, sizeof(h) + h.n * 8, side, minV, maxV);
      close(fdn);
      return rc;
    }
    if ((uint64_t)st.st_size < sizeof(h) + h.n * (8 * kw + 4)) { close(fdn); return mfx_fail(MFX_E_FORMAT, "'%s': truncated payload", path); }
    if (h.n) rc = mfx_index_add_from_file(ixs, nix, fdn, path, sizeof(h), sizeof(h) + h.n * 8 * kw, h.n, side, minV, maxV);
    close(fdn);
  } else if (fmt == MFX_DB_TEXT) {
    int k = 0;
    if (text_is_plain(p)) {
      std::mutex mu;
      struct Sink {
        Feeder f;
        void operator()(uint64_t lo, uint64_t hi, uint32_t v) { f.push(lo, hi, v); }
        int done() { f.flush(); return f.rc; }
      };
      // the k of the file is only known after its first line: a wrong k would insert garbage, so it is checked first
      {
        int k1 = 0;
        if (FILE *f = fopen(path, "rb")) {
          char line[512];
          while (k1 == 0 && fgets(line, sizeof(line), f))
            for (const char *q = line; base_code((unsigned char)*q) >= 0; ++q) ++k1;
          fclose(f);
        }
        if (k1 != ix->k) rc = k1 ? mfx_fail(MFX_E_INVAL, "'%s' holds %d-mers but the index is built for k=%d", path, k1, ix->k)
                                 : mfx_fail(MFX_E_FORMAT, "'%s': no k-mers found", path);
      }
      if (rc == MFX_OK) rc = scan_text_parallel(p, &k, &n, [&](unsigned) { return Sink{Feeder{ixs, nix, side, minV, maxV, &mu}}; });
    } else {
      rc = scan_text(p, &k, [&](uint64_t lo, uint64_t hi, uint32_t v) { fd.push(lo, hi, v); }, &n);
    }
    if (rc == MFX_OK && k != ix->k) rc = mfx_fail(MFX_E_INVAL, "'%s' holds %d-mers but the index is built for k=%d", path, k, ix->k);
  } else {
    MerylIndex mi;
    rc = read_meryl_master(p, mi);
    if (rc == MFX_OK && (int)((mi.prefixSize + mi.suffixSize) / 2) != ix->k)
      rc = mfx_fail(MFX_E_INVAL, "'%s' holds %u-mers but the index is built for k=%d", path, (mi.prefixSize + mi.suffixSize) / 2, ix->k);
    if (rc == MFX_OK) {
      std::mutex mu;
      std::vector<MerylFileSums> per(64);
      rc = for_each_meryl_file([&](uint32_t fl) {
        Feeder tf{ixs, nix, side, minV, maxV, &mu};
        int r = read_meryl_data(meryl_file_name(p, fl, ".merylData"), fl, mi, [&](uint64_t lo, uint64_t hi, uint32_t v) { tf.push(lo, hi, v); }, &per[fl]);
        if (r == MFX_OK) tf.flush();
        return r ? r : tf.rc;
      });
      if (rc == MFX_OK) rc = check_meryl_sums(p, mi, per, true);
    }
  }
  if (rc == MFX_OK) fd.flush();
  return rc ? rc : fd.rc;
}

extern "C" int mfx_index_load_db_multi(mfx_index *const *ixs, uint32_t nix, const char *path, int side, uint64_t minV, uint64_t maxV) {
  try { return mfx_index_load_db_multi_impl(ixs, nix, path, side, minV, maxV); }                       // (nothing leaves the C ABI as an exception)
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_index_load_db_multi: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_index_load_db_multi: %s", e.what()); }
}


// writes this repo's flat binary form (fixtures, interchange)
static int mfx_db_write_flat_impl(const char *path, int k, const uint64_t *kmers, const uint32_t *values, uint64_t n) {
  if (!path || (n && (!kmers || !values))) return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat: null argument");
  FILE *f = fopen(path, "wb");
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", path);
  FlatHeader h;
  memcpy(h.magic, "MFXKMER1", 8);
  h.k = (uint32_t)k;
  h.flags = 0;
  h.n = n;
  h.n_escape = 0;
  const char *de = getenv("MFX_FLAT_DELTA");
  if (k <= MFX_MAX_K_NARROW && n && !(de && atoi(de) == 0)) {   // sorted k-mers: delta-coded blocks
    const int rc = write_flat_delta(f, path, h, kmers, values, n);
    if (rc <= 0) { fclose(f); return rc; }
  }
  const char *pe = getenv("MFX_FLAT_PACKED");
  if (k <= MFX_MAX_K_PACKED && !(pe && atoi(pe) == 0)) {    // packed records: 8 bytes per k-mer
    h.flags |= FLAT_PACKED;
    std::vector<uint64_t> ek;
    std::vector<uint32_t> ev;
    bool ok = true;
    {
      // the escapes first (their number goes into the header), then the records in pieces
      for (uint64_t i = 0; i < n; ++i) if (values[i] >= MFX_PACKED_VMASK) { ek.push_back(kmers[i]); ev.push_back(values[i]); }
      h.n_escape = ek.size();
      ok = fwrite(&h, sizeof(h), 1, f) == 1;
      std::vector<uint64_t> rec(std::min<uint64_t>(n, 1u << 20));
      for (uint64_t o = 0; o < n && ok; o += rec.size()) {
        const uint64_t m = std::min<uint64_t>(rec.size(), n - o);
        for (uint64_t i = 0; i < m; ++i)
          rec[i] = (kmers[o + i] << MFX_PACKED_VBITS) | (values[o + i] >= MFX_PACKED_VMASK ? MFX_PACKED_VMASK : values[o + i]);
        ok = fwrite(rec.data(), 8, m, f) == m;
      }
      ok = ok && (ek.empty() || (fwrite(ek.data(), 8, ek.size(), f) == ek.size() && fwrite(ev.data(), 4, ev.size(), f) == ev.size()));
    }
    fclose(f);
    return ok ? MFX_OK : mfx_fail(MFX_E_IO, "short write to '%s'", path);
  }
  const size_t kw = k > MFX_MAX_K_NARROW ? 2 : 1;           // k > 31: two words per k-mer {low 64 bits, high bits}
  bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && (n == 0 || (fwrite(kmers, 8 * kw, n, f) == n && fwrite(values, 4, n, f) == n));
  fclose(f);
  return ok ? MFX_OK : mfx_fail(MFX_E_IO, "short write to '%s'", path);
}

extern "C" int mfx_db_write_flat(const char *path, int k, const uint64_t *kmers, const uint32_t *values, uint64_t n) {
  try { return mfx_db_write_flat_impl(path, k, kmers, values, n); }                       // (nothing leaves the C ABI as an exception)
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_db_write_flat: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_db_write_flat: %s", e.what()); }
}


// ---------------------------------------------------------------------------
// mfx_db_convert: any accepted database -> this library's flat form, on the host (no device).  The point is the
// delta-coded form: `meryl print` text of a 30x human read set parses at ~120 M lines/s (a minute per run), the
// same database as delta-coded blocks loads in half a second.  The k-mers are collected in input order -- which
// is ascending for `meryl print` text and for meryl directories read file by file -- and sorted (bucket by the top
// bits, then every bucket on its own) only if they are not; k > 31 keeps its order (plain 16-byte k-mers).
// ---------------------------------------------------------------------------
namespace {
struct Collected { std::vector<uint64_t> k; std::vector<uint32_t> v; };

uint64_t flat_get_bits(const uint64_t *w, uint64_t bit, uint32_t nbits) {
  const uint64_t i = bit >> 6;
  const uint32_t sh = (uint32_t)bit & 63u;
  uint64_t x = w[i] >> sh;
  if (sh + nbits > 64u) x |= w[i + 1] << (64u - sh);
  return nbits >= 64 ? x : x & ((1ull << nbits) - 1ull);
}

// a flat file into host arrays (all three encodings; k > 31: two words per k-mer)
int read_flat_host(const std::string &path, int *k_out, std::vector<uint64_t> &keys, std::vector<uint32_t> &vals) {
  FILE *f = fopen(path.c_str(), "rb");
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s'", path.c_str());
  auto bad = [&](const char *what) { fclose(f); return mfx_fail(MFX_E_FORMAT, "'%s': %s", path.c_str(), what); };
  FlatHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1) return bad("truncated header");
  struct stat fst;
  if (fstat(fileno(f), &fst) != 0) { fclose(f); return mfx_fail(MFX_E_IO, "cannot stat '%s'", path.c_str()); }
  if (const char *why = flat_header_problem(h, (uint64_t)fst.st_size)) return bad(why);
  *k_out = (int)h.k;
  const size_t kw = h.k > (uint32_t)MFX_MAX_K_NARROW ? 2 : 1;
  try { keys.resize(h.n * kw); vals.resize(h.n); }
  catch (const std::exception &) { fclose(f); return mfx_fail(MFX_E_NOMEM, "'%s': no memory for %lu k-mers", path.c_str(), (unsigned long)h.n); }
  struct EscSplit { uint64_t i, p; uint32_t s; };              // an escaped record of a placed 31-mer file: position, stored number, strand bit (2: not known yet)
  std::vector<EscSplit> esc_split;
  auto escapes = [&]() -> bool {                              // the side list of a packed / delta file: counts by k-mer
    std::vector<uint64_t> ek(h.n_escape);
    std::vector<uint32_t> ev(h.n_escape);
    if (h.n_escape && (fread(ek.data(), 8, ek.size(), f) != ek.size() || fread(ev.data(), 4, ev.size(), f) != ev.size())) return false;
    std::vector<std::pair<uint64_t, uint32_t>> e(h.n_escape);
    for (size_t i = 0; i < e.size(); ++i) e[i] = {ek[i], ev[i]};
    std::sort(e.begin(), e.end());
    for (uint64_t i = 0; i < h.n; ++i)
      if (vals[i] == 0xffffffffu && !e.empty()) {             // marked below
        auto it = std::lower_bound(e.begin(), e.end(), std::make_pair(keys[i], 0u));
        if ((it == e.end() || it->first != keys[i]) && !esc_split.empty()) {
          // a placed 31-mer whose strand bit the block does not say (no twin beside it): the other strand's k-mer then
          auto es = std::lower_bound(esc_split.begin(), esc_split.end(), i, [](const EscSplit &a, uint64_t x) { return a.i < x; });
          if (es != esc_split.end() && es->i == i && es->s == 2u) {
            uint32_t top, hi, pm;
            keys[i] = mfx_p_decode_s((int)h.k, es->p, 1u, top, hi, pm);
            it = std::lower_bound(e.begin(), e.end(), std::make_pair(keys[i], 0u));
          }
        }
        if (it == e.end() || it->first != keys[i]) return false;
        vals[i] = it->second;
      }
    return true;
  };
  if (h.flags & FLAT_DELTA) {
    uint64_t nblocks = 0;
    if (kw != 1 || fread(&nblocks, 8, 1, f) != 1 || nblocks != (h.n + MFX_DELTA_BLOCK - 1) / MFX_DELTA_BLOCK) return bad("inconsistent delta-coded payload");
    std::vector<uint64_t> dir(2 * (nblocks + 1));
    if (fread(dir.data(), 8, dir.size(), f) != dir.size()) return bad("truncated block directory");
    std::vector<uint64_t> w;
    for (uint64_t b = 0; b < nblocks; ++b) {
      const uint64_t off = dir[2 * b + 1] & 0xffffffffffffull, nxt = dir[2 * b + 3] & 0xffffffffffffull;
      const uint32_t kb = (uint32_t)(dir[2 * b + 1] >> 48) & 0xffu, vb = (uint32_t)(dir[2 * b + 1] >> 56) & 0xffu;
      const uint64_t cnt = std::min<uint64_t>(MFX_DELTA_BLOCK, h.n - b * MFX_DELTA_BLOCK);
      if (nxt < off || (nxt - off) != (((cnt - 1) * kb + 63) / 64 + (cnt * vb + 63) / 64) * 8 || vb < 2 || vb > (uint32_t)MFX_DELTA_MAX_VBITS || kb > 63)
        return bad("inconsistent block directory");
      w.assign((nxt - off) / 8 + 1, 0);
      if (fseek(f, (long)off, SEEK_SET) != 0 || fread(w.data(), 8, (nxt - off) / 8, f) != (nxt - off) / 8) return bad("truncated block");
      const uint64_t *vw = w.data() + ((cnt - 1) * kb + 63) / 64;
      uint64_t cur = dir[2 * b];
      for (uint64_t e = 0; e < cnt; ++e) {
        if (e && kb) cur += flat_get_bits(w.data(), (e - 1) * kb, kb);
        keys[b * MFX_DELTA_BLOCK + e] = cur;
        const uint32_t v = (uint32_t)flat_get_bits(vw, e * vb, vb);
        vals[b * MFX_DELTA_BLOCK + e] = v == (1u << vb) - 1u ? 0xffffffffu : v;
      }
    }
    if (h.flags & FLAT_PLACED) {                              // the records are placement numbers: back to k-mers (in the file's order)
      for (uint64_t i = 0; i < h.n; ++i) if (!flat_rec_fits(keys[i], h)) return bad("a placed record is wider than 2k + 3 bits");
      const bool split = mfx_p_split((int)h.k);               // k = 31: the count field holds count << 1 | strand bit
      // (an ESCAPED record's field is all ones: its strand bit is the second of a pair of equal numbers', or whichever of the two k-mers the
      // escape list holds -- resolved in escapes() below from the numbers kept here)
      if (split)
        for (uint64_t i = 0; i < h.n; ++i)
          if (vals[i] == 0xffffffffu)
            esc_split.push_back({i, keys[i], (i > 0 && keys[i - 1] == keys[i]) ? 1u : (i + 1 < h.n && keys[i + 1] == keys[i]) ? 0u : 2u});
      par_blocks((h.n + 65535) / 65536, [&](uint64_t b) {
        for (uint64_t i = b * 65536, e = std::min<uint64_t>(h.n, i + 65536); i < e; ++i) {
          uint32_t top, hi, pm, sb = 0;
          if (split && vals[i] != 0xffffffffu) { sb = vals[i] & 1u; vals[i] >>= 1; }
          else if (split) {
            auto it = std::lower_bound(esc_split.begin(), esc_split.end(), i, [](const EscSplit &a, uint64_t x) { return a.i < x; });
            sb = it->s == 1u ? 1u : 0u;
          }
          keys[i] = mfx_p_decode_s((int)h.k, keys[i], sb, top, hi, pm);
        }
      });
    }
    if (fseek(f, (long)(dir[2 * nblocks + 1] & 0xffffffffffffull), SEEK_SET) != 0 || !escapes()) return bad("inconsistent escape list");
  } else if (h.flags & FLAT_PACKED) {
    if (kw != 1 || (h.n && fread(keys.data(), 8, h.n, f) != h.n)) return bad("truncated packed payload");
    for (uint64_t i = 0; i < h.n; ++i) {
      const uint32_t v = (uint32_t)keys[i] & MFX_PACKED_VMASK;
      keys[i] >>= MFX_PACKED_VBITS;
      vals[i] = v == MFX_PACKED_VMASK ? 0xffffffffu : v;
    }
    if (!escapes()) return bad("inconsistent escape list");
  } else if (h.n && (fread(keys.data(), 8 * kw, h.n, f) != h.n || fread(vals.data(), 4, h.n, f) != h.n)) return bad("truncated payload");
  // every k-mer inside 2k bits: what follows (sort_pairs' bucket index, the writers' bit widths) relies on it
  if (kw == 1) {
    for (uint64_t i = 0; i < h.n; ++i)
      if (!flat_key_fits(keys[i], h.k)) return bad("a k-mer is wider than 2k bits");
  } else if (h.k < 64) {
    for (uint64_t i = 0; i < h.n; ++i)
      if (keys[2 * i + 1] >> (2 * h.k - 64)) return bad("a k-mer is wider than 2k bits");
  }
  fclose(f);
  return MFX_OK;
}

// ascending order of (k, v) for one-word k-mers: buckets by the top bits, every bucket sorted on its own, all on the host threads
void sort_pairs(int k, std::vector<uint64_t> &keys, std::vector<uint32_t> &vals, int bits = 0) {      // bits: width of the keys (0: the k-mer's 2k)
  const uint64_t n = vals.size();
  if (!bits) bits = 2 * k;
  const int shift = std::max(0, bits - 12);
  const size_t NB = (size_t)1 << std::min(12, bits);
  // histogram and scatter by slices of the input, one per host thread (a 6 G-k-mer database is two 50 GB passes)
  const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({n / (1u << 20) + 1, 64, (uint64_t)mfx_host_threads()}));
  std::vector<std::vector<uint64_t>> cnt(T, std::vector<uint64_t>(NB, 0));
  auto slices = [&](auto &&fn) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t]() { fn(t, n * t / T, n * (t + 1) / T); });
    for (auto &x : th) x.join();
  };
  slices([&](unsigned t, uint64_t b, uint64_t e) { for (uint64_t i = b; i < e; ++i) ++cnt[t][keys[i] >> shift]; });
  std::vector<uint64_t> start(NB + 1, 0);
  for (size_t b = 0; b < NB; ++b) {
    uint64_t at = start[b];
    for (unsigned t = 0; t < T; ++t) { const uint64_t c = cnt[t][b]; cnt[t][b] = at; at += c; }     // slice t's first destination in bucket b
    start[b + 1] = at;
  }
  std::vector<uint64_t> k2(n);
  std::vector<uint32_t> v2(n);
  slices([&](unsigned t, uint64_t b, uint64_t e) {
    for (uint64_t i = b; i < e; ++i) { const uint64_t d = cnt[t][keys[i] >> shift]++; k2[d] = keys[i]; v2[d] = vals[i]; }
  });
  keys.swap(k2); vals.swap(v2);
  std::vector<uint64_t>().swap(k2); std::vector<uint32_t>().swap(v2);
  par_blocks(NB, [&](uint64_t b) {
    const uint64_t lo = start[b], hi = start[b + 1];
    if (hi - lo < 2 || std::is_sorted(keys.begin() + lo, keys.begin() + hi)) return;
    std::vector<std::pair<uint64_t, uint32_t>> t(hi - lo);
    for (uint64_t i = lo; i < hi; ++i) t[i - lo] = {keys[i], vals[i]};
    std::sort(t.begin(), t.end());
    for (uint64_t i = lo; i < hi; ++i) { keys[i] = t[i - lo].first; vals[i] = t[i - lo].second; }
  });
}
}  // namespace

// any accepted database -> its k-mers (k <= 31: ascending, every k-mer once) and counts in memory: what mfx_db_convert writes and
// mfx_db_convert_placed re-keys.  t_read / t_order: seconds (diagnostics)
static int read_db_sorted(const char *in_path, int *k_out, std::vector<uint64_t> &keys, std::vector<uint32_t> &vals, double *t_read = nullptr, double *t_order = nullptr) {
  const std::string p(in_path);
  const int fmt = detect(p);
  if (!fmt) return mfx_fail(MFX_E_IO, "k-mer database '%s' does not exist", in_path);
  int k = 0, rc = MFX_OK;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  auto gather = [&](std::vector<Collected> &parts, size_t kw) {   // parts in input order -> one pair of arrays
    uint64_t n = 0;
    for (auto &c : parts) n += c.v.size();
    keys.resize(n * kw);
    vals.resize(n);
    uint64_t at = 0;
    for (auto &c : parts) {
      if (!c.v.empty()) { memcpy(keys.data() + at * kw, c.k.data(), c.k.size() * 8); memcpy(vals.data() + at, c.v.data(), c.v.size() * 4); }
      at += c.v.size();
      Collected().k.swap(c.k); Collected().v.swap(c.v);
    }
  };
  if (fmt == MFX_DB_FLAT) rc = read_flat_host(p, &k, keys, vals);
  else if (fmt == MFX_DB_TEXT) {
    // k first (the first line), then every line; plain files by all host threads, collected PER PIECE of the file so that
    // the file's order -- ascending for `meryl print` -- survives the threads
    {
      mfx_file fh = mfx_open_reader(in_path);
      char line[512];
      while (fh.f && k == 0 && fgets(line, sizeof(line), fh.f))
        for (const char *q = line; base_code((unsigned char)*q) >= 0; ++q) ++k;
      if (fh.f) (void)mfx_close(fh, true);
      if (k == 0) return mfx_fail(MFX_E_FORMAT, "'%s': no k-mers found", in_path);
      if (k > MFX_MAX_K) return mfx_fail(MFX_E_FORMAT, "'%s': k-mers of more than %d bases", in_path, MFX_MAX_K);
    }
    const size_t kw = k > MFX_MAX_K_NARROW ? 2 : 1;
    uint64_t n = 0;
    int k2 = 0;
    if (text_is_plain(p)) {
      struct stat st;
      if (stat(in_path, &st) != 0) return mfx_fail(MFX_E_IO, "cannot open '%s'", in_path);
      const uint64_t PIECE = text_piece_bytes(), npieces = ((uint64_t)st.st_size + PIECE - 1) / PIECE;
      std::vector<Collected> per(npieces ? npieces : 1);
      std::vector<Collected *> cur(64, nullptr);
      struct Sink {
        Collected **c; size_t kw;
        void operator()(uint64_t lo, uint64_t hi, uint32_t v) { (*c)->k.push_back(lo); if (kw == 2) (*c)->k.push_back(hi); (*c)->v.push_back(v); }
        int done() { return MFX_OK; }
      };
      rc = scan_text_parallel(p, &k2, &n, [&](unsigned t) { return Sink{&cur[t], kw}; },
                              [&](unsigned t, uint64_t pc) {
                                cur[t] = &per[pc];
                                per[pc].v.reserve((size_t)(PIECE / (uint64_t)(k + 3)) + 16);
                                per[pc].k.reserve(((size_t)(PIECE / (uint64_t)(k + 3)) + 16) * kw);
                              });
      if (rc == MFX_OK) gather(per, kw);
    } else {
      std::vector<Collected> one(1);
      rc = scan_text(p, &k2, [&](uint64_t lo, uint64_t hi, uint32_t v) { one[0].k.push_back(lo); if (kw == 2) one[0].k.push_back(hi); one[0].v.push_back(v); }, &n);
      if (rc == MFX_OK) gather(one, kw);
    }
    if (rc == MFX_OK && k2 != k && n) rc = mfx_fail(MFX_E_FORMAT, "'%s': k-mer length %d differs from %d", in_path, k2, k);
  } else {
    MerylIndex mi;
    rc = read_meryl_master(p, mi);
    if (rc) return rc;
    k = (int)((mi.prefixSize + mi.suffixSize) / 2);
    const size_t kw = k > MFX_MAX_K_NARROW ? 2 : 1;
    std::vector<Collected> per(64);
    std::vector<MerylFileSums> sums(64);
    rc = for_each_meryl_file([&](uint32_t fl) {
      Collected &c = per[fl];
      return read_meryl_data(meryl_file_name(p, fl, ".merylData"), fl, mi,
                             [&](uint64_t lo, uint64_t hi, uint32_t v) { c.k.push_back(lo); if (kw == 2) c.k.push_back(hi); c.v.push_back(v); }, &sums[fl]);
    });
    if (rc == MFX_OK) rc = check_meryl_sums(p, mi, sums, true);
    if (rc == MFX_OK) gather(per, kw);
  }
  if (rc) return rc;
  const double t1 = now();
  if (k <= MFX_MAX_K_NARROW) {
    bool sorted = true;
    for (uint64_t i = 1; i < vals.size() && sorted; ++i) sorted = keys[i] > keys[i - 1];
    if (!sorted) {
      sort_pairs(k, keys, vals);
      for (uint64_t i = 1; i < vals.size(); ++i)
        if (keys[i] == keys[i - 1]) return mfx_fail(MFX_E_FORMAT, "'%s' lists a k-mer twice; a database holds every k-mer once", in_path);
    }
  }
  *k_out = k;
  if (t_read) *t_read = t1 - t0;
  if (t_order) *t_order = now() - t1;
  return MFX_OK;
}

static int mfx_db_convert_impl(const char *in_path, const char *out_path, uint64_t *n_out) {
  if (!in_path || !out_path) return mfx_fail(MFX_E_INVAL, "mfx_db_convert: null argument");
  int k = 0;
  std::vector<uint64_t> keys;
  std::vector<uint32_t> vals;
  double t_read = 0, t_order = 0;
  int rc = read_db_sorted(in_path, &k, keys, vals, &t_read, &t_order);
  if (rc) return rc;
  if (n_out) *n_out = vals.size();
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t2 = now();
  rc = mfx_db_write_flat(out_path, k, keys.data(), vals.data(), vals.size());
  if (getenv("MFX_DB_TIMING")) fprintf(stderr, "[mfx db] convert: read %.2f s, order %.2f s, write %.2f s\n", t_read, t_order, now() - t2);
  return rc;
}

// the flat form of ascending placement numbers P (mfx_place.h) and their counts
// sbits: the strand bits of a k = 31 file's records (mfx_place.h, mfx_p_split), pkeys then the numbers P >> 1; the C ABI's writer (arrays of P) serves k <= 30
static int mfx_db_write_flat_placed_impl(const char *path, int k, const uint64_t *pkeys, const uint32_t *values, uint64_t n, const uint8_t *sbits = nullptr) {
  if (!path || (n && (!pkeys || !values))) return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat_placed: null argument");
  if (k < MFX_PLACE_MIN_K || k > MFX_PLACE_MAX_K) return mfx_fail(MFX_E_INVAL, "a placed database holds %d <= k <= %d (k = %d)", MFX_PLACE_MIN_K, MFX_PLACE_MAX_K, k);
  if (mfx_p_split(k) != (sbits != nullptr))
    return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat_placed: the placement number of a %d-mer takes 65 bits; its placed database is made by mfx_db_convert_placed (merfin -convert -placed)", k);
  if (!n) return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat_placed: no k-mers");
  for (uint64_t i = 0; i < n && !sbits; ++i)
    if (pkeys[i] >> mfx_p_bits(k)) return mfx_fail(MFX_E_INVAL, "mfx_db_write_flat_placed: record %lu is wider than a placement number of this k", (unsigned long)i);
  FILE *f = fopen(path, "wb");
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", path);
  FlatHeader h;
  memcpy(h.magic, "MFXKMER1", 8);
  h.k = (uint32_t)k;
  h.flags = 0;
  h.n = n;
  h.n_escape = 0;
  int rc = write_flat_delta(f, path, h, pkeys, values, n, true, sbits);
  if (fclose(f) != 0 && rc == 0) rc = mfx_fail(MFX_E_IO, "short write to '%s'", path);
  if (rc == 1) rc = mfx_fail(MFX_E_INVAL, "mfx_db_write_flat_placed: the records are not strictly ascending");
  return rc;
}
extern "C" int mfx_db_write_flat_placed(const char *path, int k, const uint64_t *pkeys, const uint32_t *values, uint64_t n) {
  try { return mfx_db_write_flat_placed_impl(path, k, pkeys, values, n); }
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_db_write_flat_placed: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_db_write_flat_placed: %s", e.what()); }
}

// P of n canonical k-mers, on the host (mfx_place.h; the device form: mfx_db_place_keys in mfx_api.cpp)
void mfx_place_keys_host(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, uint8_t *sbits_out) {      // sbits_out: k = 31 (out then holds P >> 1)
  par_blocks((n + 65535) / 65536, [&](uint64_t b) {
    for (uint64_t i = b * 65536, e = std::min<uint64_t>(n, i + 65536); i < e; ++i) {
      const uint64_t key = kmers[i], rc = mfx_p_revcomp(key, k);
      uint32_t sb;
      out[i] = mfx_p_encode_s(k, key < rc ? key : rc, sb);
      if (sbits_out) sbits_out[i] = (uint8_t)sb;
    }
  });
}

// any accepted database -> the PLACED flat form: the records sorted by where the compact table puts them
static int mfx_db_convert_placed_impl(const char *in_path, const char *out_path, uint64_t *n_out) {
  if (!in_path || !out_path) return mfx_fail(MFX_E_INVAL, "mfx_db_convert_placed: null argument");
  {
    // k first: a database this form cannot hold is refused before its tens of GB are converted and read back
    mfx_db_info pi;
    if (int prc = mfx_db_probe(in_path, &pi)) return prc;
    if (pi.k < MFX_PLACE_MIN_K || pi.k > MFX_PLACE_MAX_K)
      return mfx_fail(MFX_E_INVAL, "'%s' holds %d-mers: a placed database holds %d <= k <= %d (use -convert without -placed)", in_path, pi.k, MFX_PLACE_MIN_K, MFX_PLACE_MAX_K);
  }
  // the database's k-mers in memory (any form), then re-keyed -- no intermediate file
  int k = 0;
  std::vector<uint64_t> keys;
  std::vector<uint32_t> vals;
  if (int rc = read_db_sorted(in_path, &k, keys, vals)) return rc;
  if (k < MFX_PLACE_MIN_K || k > MFX_PLACE_MAX_K)
    return mfx_fail(MFX_E_INVAL, "'%s' holds %d-mers: a placed database holds %d <= k <= %d (use -convert without -placed)", in_path, k, MFX_PLACE_MIN_K, MFX_PLACE_MAX_K);
  std::atomic<int> noncanon{0};
  par_blocks((vals.size() + 65535) / 65536, [&](uint64_t b) {
    for (uint64_t i = b * 65536, e = std::min<uint64_t>(vals.size(), i + 65536); i < e; ++i)
      if (keys[i] > mfx_p_revcomp(keys[i], k)) { noncanon = 1; return; }
  });
  if (noncanon) return mfx_fail(MFX_E_NONCANON, "'%s' is not canonical: a placed database holds canonical k-mers (the sequence-only index it feeds does)", in_path);
  if (n_out) *n_out = vals.size();
  if (!mfx_p_split(k)) {
    mfx_place_keys_host(k, keys.data(), vals.size(), keys.data(), nullptr);
    sort_pairs(k, keys, vals, mfx_p_bits(k));
    return mfx_db_write_flat_placed(out_path, k, keys.data(), vals.data(), vals.size());
  }
  // k = 31: the stored number is P >> 1, the strand bit beside it.  The records of either strand bit are sorted on their own and merged, the
  // s = 0 one of an equal pair first (the order of the 65-bit P)
  const uint64_t N = vals.size();
  std::vector<uint8_t> sb(N);
  mfx_place_keys_host(k, keys.data(), N, keys.data(), sb.data());
  std::vector<uint64_t> k1;
  std::vector<uint32_t> v1;
  {
    uint64_t n0 = 0;
    for (uint64_t i = 0; i < N; ++i) {
      if (sb[i]) { k1.push_back(keys[i]); v1.push_back(vals[i]); }
      else { keys[n0] = keys[i]; vals[n0] = vals[i]; ++n0; }
    }
    keys.resize(n0); vals.resize(n0);
  }
  sort_pairs(k, keys, vals, 64);
  sort_pairs(k, k1, v1, 64);
  std::vector<uint64_t> mk(N);
  std::vector<uint32_t> mv(N);
  for (uint64_t a = 0, b = 0, o = 0; o < N; ++o) {
    const bool take0 = b == k1.size() || (a < keys.size() && keys[a] <= k1[b]);
    if (take0) { mk[o] = keys[a]; mv[o] = vals[a]; sb[o] = 0; ++a; }
    else { mk[o] = k1[b]; mv[o] = v1[b]; sb[o] = 1; ++b; }
  }
  std::vector<uint64_t>().swap(keys); std::vector<uint64_t>().swap(k1);
  return mfx_db_write_flat_placed_impl(out_path, k, mk.data(), mv.data(), N, sb.data());
}
extern "C" int mfx_db_convert_placed(const char *in_path, const char *out_path, uint64_t *n_out) {
  try { return mfx_db_convert_placed_impl(in_path, out_path, n_out); }
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_db_convert_placed: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_db_convert_placed: %s", e.what()); }
}

extern "C" int mfx_db_convert(const char *in_path, const char *out_path, uint64_t *n_out) {
  try { return mfx_db_convert_impl(in_path, out_path, n_out); }                       // (nothing leaves the C ABI as an exception)
  catch (const std::bad_alloc &) { return mfx_fail(MFX_E_NOMEM, "mfx_db_convert: out of memory"); }
  catch (const std::exception &e) { return mfx_fail(MFX_E_IO, "mfx_db_convert: %s", e.what()); }
}


// ---------------------------------------------------------------------------
// Device-format index image: the built table written to / read from disk as is,
// so that repeated runs (-hist, then -dump, then -polish on the same databases)
// skip the decode + insert of the k-mer databases.  Layout: IndexImageHeader,
// then the table lines (128 bytes each) in order.
// ---------------------------------------------------------------------------
struct IndexImageHeader {
  char     magic[8];         // "MFXINDX2"
  uint32_t k, mz_w;
  uint32_t shard_rank, shard_n;
  uint64_t nlines, capacity_kmers;
  uint64_t minV, maxV;
  uint64_t meta[4];          // distinct, non-canonical inserts, probe failures, reserved
  uint32_t slot_bytes, line_slots;
  uint32_t filter_set, layout;   // layout = MFX_LAYOUT_VERSION of the build that wrote the image
  uint64_t fingerprint;          // caller's digest of the inputs the table was built from (mfx_index_set_fingerprint)
  uint64_t side_nlines;          // compact layout: lines of the side table that follow the nlines main lines
  uint32_t flags, seq_digest;    // bit 0: sequence-only index, bit 1: compact layout, bit 2: frozen (counts were added), bit 3: quotient key fields; digest of the claimed sequence (0: none)
};

static_assert(sizeof(IndexImageHeader) <= MFX_INDEX_HEADER_BYTES, "index image header outgrew its public size");

static int fill_header(const mfx_index *ix, IndexImageHeader &h) {
  if (hipSetDevice(ix->device) != hipSuccess) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", ix->device);
  memset(&h, 0, sizeof(h));
  memcpy(h.magic, "MFXINDX2", 8);
  h.k = (uint32_t)ix->k; h.mz_w = (uint32_t)ix->mz_w | ((uint32_t)ix->mz_t << 8);   // (the sampling t-mer length rides in the second byte)
  h.shard_rank = ix->shard_rank; h.shard_n = ix->shard_n;
  h.nlines = ix->nlines; h.capacity_kmers = ix->capacity_kmers;
  h.minV = ix->minV; h.maxV = ix->maxV;
  h.slot_bytes = MFX_ALIGN / ix->slots_per_line(); h.line_slots = ix->slots_per_line();
  h.filter_set = ix->filter_set ? 1u : 0u;
  h.layout = MFX_LAYOUT_VERSION;
  h.fingerprint = ix->fingerprint;
  h.side_nlines = ix->side_nlines;
  h.flags = (ix->seq_only ? 1u : 0u) | (ix->compact ? 2u : 0u) | (ix->frozen ? 4u : 0u) | (ix->quot ? 8u : 0u);
  h.seq_digest = ix->seq_digest;
  if (hipMemcpy(h.meta, ix->d_meta, sizeof(h.meta), hipMemcpyDeviceToHost) != hipSuccess) return mfx_fail(MFX_E_HIP, "reading index metadata failed");
  return MFX_OK;
}

static bool header_ok(const IndexImageHeader &h) {
  const bool wide = h.k > (uint32_t)MFX_MAX_K_NARROW, compact = (h.flags & 2u) != 0;
  if (compact && (!(h.flags & 1u) || wide || h.k > (uint32_t)MFX_MAX_K_COMPACT || h.side_nlines == 0)) return false;
  if (((h.flags & 8u) != 0) != (compact && h.k > (uint32_t)MFX_MAX_K_DIRECT)) return false;      // the quotient form is the compact layout of k > 21
  if ((h.flags & 8u) && h.nlines < (1ull << (2 * ((int)h.k - 3) - 31))) return false;
  if (!compact && h.side_nlines != 0) return false;
  const uint32_t line_slots = wide ? MFX_WSLOTS_LINE : compact ? MFX_CSLOTS_LINE : MFX_SLOTS_LINE;
  return memcmp(h.magic, "MFXINDX2", 8) == 0 && h.slot_bytes == MFX_ALIGN / line_slots && h.line_slots == line_slots && h.k >= 1 &&
         h.k <= (uint32_t)MFX_MAX_K && h.nlines != 0 && h.nlines < (1ull << 32) && h.side_nlines < (1ull << 32) && h.layout == MFX_LAYOUT_VERSION;
}

// an index of exactly the header's geometry; its lines are allocated but hold nothing yet
static mfx_index *index_from_header(const IndexImageHeader &h, double max_gb, int device) {
  mfx_index *ix = mfx_index_create((int)h.k, 1, 0.0, device);
  if (!ix) return nullptr;
  const uint64_t total_lines = h.nlines + h.side_nlines;
  if (max_gb > 0 && (double)total_lines * MFX_ALIGN / 1e9 > max_gb) {
    mfx_fail(MFX_E_NOMEM, "Not enough memory to load databases.  Increase -memory. (need %.3f GB, limit %.3f GB)", (double)total_lines * MFX_ALIGN / 1e9, max_gb);
    mfx_index_free(ix);
    return nullptr;
  }
  (void)hipSetDevice(device);
  (void)hipFree(ix->d_slots);
  ix->d_slots = nullptr;
  ix->nlines = h.nlines;
  ix->capacity_kmers = h.capacity_kmers;
  ix->mz_w = (int)(h.mz_w & 0xffu);
  ix->mz_t = (int)((h.mz_w >> 8) & 0xffu);
  ix->shard_rank = h.shard_rank; ix->shard_n = h.shard_n ? h.shard_n : 1;
  ix->minV = h.minV; ix->maxV = h.maxV; ix->filter_set = h.filter_set != 0;
  ix->fingerprint = h.fingerprint;
  ix->side_nlines = h.side_nlines;
  ix->seq_only = (h.flags & 1u) != 0; ix->compact = (h.flags & 2u) != 0; ix->frozen = (h.flags & 4u) != 0; ix->quot = (h.flags & 8u) != 0;
  ix->seq_digest = ix->seq_only ? h.seq_digest : 0u;
  if (hipMalloc((void **)&ix->d_slots, total_lines * MFX_ALIGN) != hipSuccess ||
      hipMemcpy(ix->d_meta, h.meta, sizeof(h.meta), hipMemcpyHostToDevice) != hipSuccess) {
    mfx_fail(MFX_E_NOMEM, "cannot allocate %.3f GB for the index image on device %d", (double)total_lines * MFX_ALIGN / 1e9, device);
    mfx_index_free(ix);
    return nullptr;
  }
  return ix;
}

extern "C" int mfx_index_image_header(const mfx_index *ix, void *hdr) {
  if (!ix || !hdr) return mfx_fail(MFX_E_INVAL, "mfx_index_image_header: null argument");
  IndexImageHeader h;
  int rc = fill_header(ix, h);
  if (rc) return rc;
  memset(hdr, 0, MFX_INDEX_HEADER_BYTES);
  memcpy(hdr, &h, sizeof(h));
  return MFX_OK;
}

extern "C" mfx_index *mfx_index_create_from_header(const void *hdr, double max_gb, int device) {
  if (!hdr) { mfx_fail(MFX_E_INVAL, "mfx_index_create_from_header: null argument"); return nullptr; }
  IndexImageHeader h;
  memcpy(&h, hdr, sizeof(h));
  if (!header_ok(h)) { mfx_fail(MFX_E_FORMAT, "not an index image header of this build"); return nullptr; }
  return index_from_header(h, max_gb, device);
}

extern "C" int mfx_index_device_image(mfx_index *ix, void **d_lines, uint64_t *line_bytes, void **d_meta, uint64_t *meta_bytes) {
  if (!ix || !d_lines || !line_bytes || !d_meta || !meta_bytes) return mfx_fail(MFX_E_INVAL, "mfx_index_device_image: null argument");
  *d_lines = ix->d_slots;
  *line_bytes = ix->total_lines() * MFX_ALIGN;
  *d_meta = ix->d_meta;
  *meta_bytes = 4 * sizeof(uint64_t);
  return MFX_OK;
}

extern "C" int mfx_index_commit(mfx_index *ix) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_commit: null argument");
  ix->version++;                       // cached per-index facts (canonical or not) are re-read from the new contents
  return MFX_OK;
}

extern "C" int mfx_index_set_fingerprint(mfx_index *ix, uint64_t fingerprint) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_set_fingerprint: null index");
  ix->fingerprint = fingerprint;
  return MFX_OK;
}

extern "C" int mfx_index_get_origin(const mfx_index *ix, uint64_t *fingerprint, uint64_t *minV, uint64_t *maxV) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_get_origin: null index");
  if (fingerprint) *fingerprint = ix->fingerprint;
  if (minV) *minV = ix->filter_set ? ix->minV : 0;
  if (maxV) *maxV = ix->filter_set ? ix->maxV : ~0ull;
  return MFX_OK;
}

extern "C" int mfx_index_save(const mfx_index *ix, const char *path) {
  if (!ix || !path) return mfx_fail(MFX_E_INVAL, "mfx_index_save: null argument");
  IndexImageHeader h;
  int rc = fill_header(ix, h);
  if (rc) return rc;
  FILE *f = fopen(path, "wb");
  if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", path);
  if (fwrite(&h, sizeof(h), 1, f) != 1) rc = mfx_fail(MFX_E_IO, "short write to '%s'", path);
  const size_t CH = 256ull << 20;
  std::vector<char> buf(rc == MFX_OK ? CH : 1);
  const uint64_t total = ix->total_lines() * MFX_ALIGN;
  for (uint64_t o = 0; o < total && rc == MFX_OK; o += CH) {
    size_t m = (size_t)std::min<uint64_t>(CH, total - o);
    if (hipMemcpy(buf.data(), (const char *)ix->d_slots + o, m, hipMemcpyDeviceToHost) != hipSuccess) rc = mfx_fail(MFX_E_HIP, "D2H copy of the table failed");
    else if (fwrite(buf.data(), 1, m, f) != m) rc = mfx_fail(MFX_E_IO, "short write to '%s'", path);
  }
  fclose(f);
  return rc;
}

extern "C" mfx_index *mfx_index_load(const char *path, double max_gb, int device) {
  if (!path) { mfx_fail(MFX_E_INVAL, "mfx_index_load: null argument"); return nullptr; }
  FILE *f = fopen(path, "rb");
  if (!f) { mfx_fail(MFX_E_IO, "cannot open '%s'", path); return nullptr; }
  IndexImageHeader h;
  if (fread(&h, sizeof(h), 1, f) != 1 || !header_ok(h)) {
    fclose(f);
    mfx_fail(MFX_E_FORMAT, "'%s' is not an index image of this build", path);
    return nullptr;
  }
  mfx_index *ix = index_from_header(h, max_gb, device);
  if (!ix) { fclose(f); return nullptr; }
  bool ok = true;
  const size_t CH = 256ull << 20;
  std::vector<char> buf(ok ? CH : 1);
  const uint64_t total = (h.nlines + h.side_nlines) * MFX_ALIGN;
  for (uint64_t o = 0; o < total && ok; o += CH) {
    size_t m = (size_t)std::min<uint64_t>(CH, total - o);
    ok = fread(buf.data(), 1, m, f) == m && hipMemcpy((char *)ix->d_slots + o, buf.data(), m, hipMemcpyHostToDevice) == hipSuccess;
  }
  fclose(f);
  if (!ok) {
    mfx_fail(MFX_E_IO, "'%s': truncated image or device copy failed", path);
    mfx_index_free(ix);
    return nullptr;
  }
  ix->version++;
  return ix;
}
