// fasta.h -- FASTA/FASTQ record reader for the merfin CLI.
// Replaces dnaSeqFile::loadSequence / dnaSeq::{ident,bases,length} as used at
// src/merfin/merfin.C:38,45 and merfin-globals.C:194: plain or gz/bz2/xz
// input (merfin.C:195), ident() = first whitespace-delimited header token
// (it is matched against VCF CHROM at merfin-variants.C:141).
#pragma once
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <stdlib.h>
#include <unistd.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../csrc/mfx_pipe.h"

struct SeqRecord {
  std::string name;
  std::string bases;                       // the record's bases as read by SeqFile::next ...
  std::shared_ptr<char> arena;             // ... or a piece of the buffer read_fasta_parallel filled for all records
  const char *abases = nullptr;
  size_t alen = 0;
  const char *data() const { return arena ? abases : bases.data(); }
  size_t size() const { return arena ? alen : bases.size(); }
};

class SeqFile {
 public:
  explicit SeqFile(const std::string &path) {
    h_ = mfx_open_reader(path.c_str());        // decompressor by suffix, started without a shell (csrc/mfx_pipe.h)
    f_ = h_.f;
    buf_.resize(1 << 22);
    // plain files: the file size bounds every record, so the first (often only large) record never re-allocates while
    // it grows; later records reserve what is left
    struct stat st;
    if (!h_.is_pipe() && stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode)) size_hint_ = (size_t)st.st_size;
  }
  ~SeqFile() { if (f_) (void)mfx_close(h_, true); }
  bool ok() const { return f_ != nullptr; }
  // Close the input and report how it ended: 0 = clean.  Once next() has returned false the whole stream was read, so a
  // decompressor that exited non-zero (truncated / corrupt .gz) or a read error is a failure -- a shorter assembly must
  // not be evaluated as if it were the file.  Before the end of input (a reader that stopped early) the child's SIGPIPE
  // is not an error.
  int finish() {
    if (!f_) return 0;
    const bool at_end = eof_;
    f_ = nullptr;
    return mfx_close(h_, !at_end);
  }

  // one record per call; false at end of input
  bool next(SeqRecord &r) {
    r.name.clear();
    r.bases.clear();
    std::string line;
    if (!have_header_) {
      while (getline(line)) if (!line.empty() && (line[0] == '>' || line[0] == '@')) { header_ = line; have_header_ = true; break; }
      if (!have_header_) { eof_ = true; return false; }
    }
    bool fastq = header_[0] == '@';
    size_t e = 1;
    while (e < header_.size() && header_[e] != ' ' && header_[e] != '\t') ++e;
    r.name = header_.substr(1, e - 1);
    have_header_ = false;
    if (!fastq) {
      // sequence lines go straight from the read buffer into r.bases (no per-line temporary): one memchr + one append
      // per line; only a line starting with '>' is materialised, as the next record's header
      // plain files: what is left of the file bounds this record, so it never re-allocates while it grows
      const size_t left = size_hint_ > consumed() ? size_hint_ - consumed() : 0;
      if (left > r.bases.capacity()) r.bases.reserve(left);
      auto done = [&]() { if (r.bases.capacity() > r.bases.size() + (r.bases.size() >> 2) + 4096) r.bases.shrink_to_fit(); return true; };
      while (true) {
        if (pos_ == len_ && !refill()) return done();
        if (buf_[pos_] == '>') {
          if (getline(line)) { header_ = line; have_header_ = true; }
          return done();
        }
        bool eol = false;
        while (!eol) {
          if (pos_ == len_ && !refill()) return done();
          const char *p = buf_.data() + pos_;
          const char *nl = (const char *)memchr(p, '\n', len_ - pos_);
          size_t n = nl ? (size_t)(nl - p) : len_ - pos_;
          r.bases.append(p, n);
          pos_ += n + (nl ? 1 : 0);
          eol = nl != nullptr;
        }
        if (!r.bases.empty() && r.bases.back() == '\r') r.bases.pop_back();
      }
    }
    while (getline(line)) {             // sequence lines up to '+'
      if (!line.empty() && line[0] == '+') break;
      r.bases += line;
    }
    size_t q = 0;
    while (q < r.bases.size() && getline(line)) q += line.size();   // quality of the same length
    return true;
  }

 private:
  bool refill() {
    if (!f_) return false;
    len_ = fread(buf_.data(), 1, buf_.size(), f_);
    pos_ = 0;
    total_ += len_;
    return len_ != 0;
  }
  size_t consumed() const { return total_ - (len_ - pos_); }
  bool getline(std::string &out) {
    out.clear();
    while (true) {
      if (pos_ == len_) {
        if (!f_) return !out.empty();
        len_ = fread(buf_.data(), 1, buf_.size(), f_);
        pos_ = 0;
        total_ += len_;
        if (len_ == 0) return !out.empty();      // last line without a newline
      }
      const char *p = buf_.data() + pos_;
      const char *nl = (const char *)memchr(p, '\n', len_ - pos_);
      if (nl) {
        out.append(p, nl - p);
        pos_ += (nl - p) + 1;
        if (!out.empty() && out.back() == '\r') out.pop_back();
        return true;
      }
      out.append(p, len_ - pos_);
      pos_ = len_;
    }
  }
  FILE *f_ = nullptr;
  mfx_file h_;
  std::vector<char> buf_;
  size_t pos_ = 0, len_ = 0, size_hint_ = 0, total_ = 0;
  bool have_header_ = false, eof_ = false;
  std::string header_;
};

// n bytes of fresh memory that asks for transparent huge pages: touching a GB of 4 KB pages for the first time costs more
// than parsing it (a quarter of a million page faults), 2 MB pages make that 512
inline std::shared_ptr<char> mfx_big_alloc(size_t n) {
  const size_t len = (n + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
  void *p = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) return std::shared_ptr<char>();
  (void)madvise(p, len, MADV_HUGEPAGE);
  // released in pieces: unmapping a GB in one call holds the address-space lock for 50 ms, and every other thread of the
  // process that maps, unmaps or ends (a thread's stack) waits behind it
  return std::shared_ptr<char>((char *)p, [len](char *q) {
    const size_t piece = 32u << 20;
    for (size_t o = 0; o < len; o += piece) munmap(q + o, std::min(piece, len - o));
  });
}

// A plain FASTA file read by all host threads (the sequential reader parses ~5 Gb/s on one core: 0.2 s per Gb of a run
// whose GPU part takes 0.03 s): the file is pread() in slices, record headers are the '>' at line starts, and every
// record's lines are copied -- newlines dropped -- into one buffer by pieces of 8 MB, each at the offset the newline
// counts of the pieces before it give.  Same records as SeqFile::next (ident = first token of the header line; a '\r'
// before a '\n' is dropped; empty lines are nothing).  false = not applicable (compressed / piped input, a FASTQ, a file
// that does not start with '>'): the caller reads the file with SeqFile.  MFX_CLI_SEQ_THREADS=1 switches it off.
inline bool read_fasta_parallel(const std::string &path, std::vector<SeqRecord> &recs) {
  const bool timing = getenv("MFX_CLI_SEQ_TIMING") != nullptr;
  auto no = [&](const char *why) { if (timing) fprintf(stderr, "-- read_fasta_parallel: not used (%s)\n", why); return false; };
  if (mfx_suffix_tool(path.c_str()) != nullptr) return no("compressed input");
  unsigned nt = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *e = getenv("MFX_CLI_SEQ_THREADS")) nt = (unsigned)std::max(1, atoi(e));
  else if (const char *h = getenv("MFX_HOST_THREADS")) nt = (unsigned)std::max(1, atoi(h));
  if (nt < 2) return no("one thread");
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tp[6] = {now(), 0, 0, 0, 0, 0};
  const int fd = open(path.c_str(), O_RDONLY);
  struct stat st;
  const char *pm = getenv("MFX_CLI_SEQ_PAR_MIN");            // smaller files are not worth the threads (tests: 0)
  const off_t min_size = pm ? (off_t)atoll(pm) : (off_t)(1 << 20);
  if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < std::max<off_t>(min_size, 1)) { if (fd >= 0) close(fd); return no("not a regular file of the minimum size"); }
  const size_t n = (size_t)st.st_size;
  char first = 0;
  if (pread(fd, &first, 1, 0) != 1 || first != '>') { close(fd); return no("does not start with '>'"); }
  std::shared_ptr<char> raw = mfx_big_alloc(n + 1);
  if (!raw) { close(fd); return no("no memory for the file"); }
  char *buf = raw.get();
  buf[n] = 0;                                               // sentinel: every byte of the file has a successor to look at
  auto run = [nt](size_t items, auto &&fn) {                 // fn(i) for every item, handed out one at a time
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < std::min<size_t>(nt, items); ++t)
      th.emplace_back([&]() { for (size_t i = next.fetch_add(1); i < items; i = next.fetch_add(1)) fn(i); });
    for (auto &x : th) x.join();
  };
  // ---- the file, and the header positions of every slice
  const char *ps = getenv("MFX_CLI_SEQ_SLICE");              // bytes per slice / piece (tests: small and odd)
  const size_t SL = ps && atoll(ps) > 0 ? (size_t)atoll(ps) : (size_t)(8u << 20), nsl = (n + SL - 1) / SL;
  std::vector<std::vector<size_t>> heads(nsl);
  std::atomic<bool> ok{true};
  run(nsl, [&](size_t i) {
    const size_t b = i * SL, e = std::min(n, b + SL);
    for (size_t o = b; o < e;) {
      const ssize_t r = pread(fd, buf + o, e - o, (off_t)o);
      if (r <= 0) { ok = false; return; }
      o += (size_t)r;
    }
  });
  close(fd);
  if (!ok) return no("read error");
  tp[1] = now();
  run(nsl, [&](size_t i) {                                   // after all slices are in: a header needs the byte before it
    const size_t b = i * SL, e = std::min(n, b + SL);
    for (const char *p = buf + b; p < buf + e;) {
      p = (const char *)memchr(p, '>', (size_t)(buf + e - p));
      if (!p) break;
      const size_t at = (size_t)(p - buf);
      if (at == 0 || buf[at - 1] == '\n') heads[i].push_back(at);
      ++p;
    }
  });
  std::vector<size_t> hd;
  for (auto &v : heads) hd.insert(hd.end(), v.begin(), v.end());
  // ---- records: name, body = [end of the header line + 1, next header)
  struct Piece { size_t rec, b, e, drop = 0, out = 0; };
  std::vector<Piece> pieces;
  recs.clear();
  recs.resize(hd.size());
  for (size_t r = 0; r < hd.size(); ++r) {
    const size_t h = hd[r], lim = r + 1 < hd.size() ? hd[r + 1] : n;
    const char *nl = (const char *)memchr(buf + h, '\n', lim - h);
    size_t he = nl ? (size_t)(nl - buf) : lim;                // end of the header line
    const size_t body_b = std::min(lim, he + 1);
    if (nl && he > h && buf[he - 1] == '\r') --he;           // as everywhere: a CR goes only with its LF
    size_t e = h + 1;
    while (e < he && buf[e] != ' ' && buf[e] != '\t') ++e;
    recs[r].name.assign(buf + h + 1, e - h - 1);
    for (size_t b = body_b; b < lim; b += SL) pieces.push_back({r, b, std::min(lim, b + SL)});
  }
  tp[2] = now();
  // ---- what every piece drops: every '\n', and a '\r' right before a '\n' (wherever that '\n' lies: buf[e] is readable)
  run(pieces.size(), [&](size_t i) {
    Piece &p = pieces[i];
    size_t d = 0;
    for (const char *q = buf + p.b, *e = buf + p.e; q < e;) {
      q = (const char *)memchr(q, '\n', (size_t)(e - q));
      if (!q) break;
      ++d;
      if (q > buf + p.b && q[-1] == '\r') ++d;
      ++q;
    }
    if (buf[p.e - 1] == '\r' && buf[p.e] == '\n') ++d;       // its '\n' is the first byte after the piece
    p.drop = d;
  });
  tp[3] = now();
  size_t total = 0;
  std::vector<size_t> rec_off(hd.size()), rec_len(hd.size(), 0);
  for (size_t r = 0, i = 0; r < hd.size(); ++r) {
    rec_off[r] = total;
    for (; i < pieces.size() && pieces[i].rec == r; ++i) {
      pieces[i].out = total;
      total += (pieces[i].e - pieces[i].b) - pieces[i].drop;
    }
    rec_len[r] = total - rec_off[r];
  }
  std::shared_ptr<char> arena = mfx_big_alloc(total + 1);
  if (!arena) { recs.clear(); return no("no memory for the bases"); }
  char *out = arena.get();
  run(pieces.size(), [&](size_t i) {
    const Piece &p = pieces[i];
    char *w = out + p.out;
    for (const char *q = buf + p.b, *e = buf + p.e; q < e;) {
      const char *nl = (const char *)memchr(q, '\n', (size_t)(e - q));
      const char *le = nl ? nl : e;                          // end of this line's bytes inside the piece
      size_t len = (size_t)(le - q);
      if (len && le[-1] == '\r' && *le == '\n') --len;
      memcpy(w, q, len);
      w += len;
      q = nl ? nl + 1 : e;
    }
  });
  for (size_t r = 0; r < hd.size(); ++r) { recs[r].arena = arena; recs[r].abases = out + rec_off[r]; recs[r].alen = rec_len[r]; }
  tp[4] = now();
  std::thread([held = std::move(raw)]() mutable { held.reset(); }).detach();     // unmapping a GB takes 50 ms: not on the caller's time
  tp[5] = now();
  if (timing)
    fprintf(stderr, "-- read_fasta_parallel: %u threads, file %.3f s, headers %.3f s, line ends %.3f s, copy %.3f s, hand-off %.3f s\n", nt,
            tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[4] - tp[3], tp[5] - tp[4]);
  return true;
}
