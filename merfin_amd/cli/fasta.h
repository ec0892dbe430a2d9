// fasta.h -- FASTA/FASTQ record reader for the merfin CLI.
// Replaces dnaSeqFile::loadSequence / dnaSeq::{ident,bases,length} as used at
// src/merfin/merfin.C:38,45 and merfin-globals.C:194: plain or gz/bz2/xz
// input (merfin.C:195), ident() = first whitespace-delimited header token
// (it is matched against VCF CHROM at merfin-variants.C:141).
#pragma once
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <string>
#include <vector>

#include "../csrc/mfx_pipe.h"

struct SeqRecord {
  std::string name;
  std::string bases;
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
