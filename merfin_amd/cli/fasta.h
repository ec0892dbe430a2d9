// fasta.h -- FASTA/FASTQ record reader for the merfin CLI.
// Replaces dnaSeqFile::loadSequence / dnaSeq::{ident,bases,length} as used at
// src/merfin/merfin.C:38,45 and merfin-globals.C:194: plain or gz/bz2/xz
// input (merfin.C:195), ident() = first whitespace-delimited header token
// (it is matched against VCF CHROM at merfin-variants.C:141).
#pragma once
#include <stdio.h>
#include <string.h>
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
  }
  ~SeqFile() { if (f_) (void)mfx_close(h_, true); }
  bool ok() const { return f_ != nullptr; }

  // one record per call; false at end of input
  bool next(SeqRecord &r) {
    r.name.clear();
    r.bases.clear();
    std::string line;
    if (!have_header_) {
      while (getline(line)) if (!line.empty() && (line[0] == '>' || line[0] == '@')) { header_ = line; have_header_ = true; break; }
      if (!have_header_) return false;
    }
    bool fastq = header_[0] == '@';
    size_t e = 1;
    while (e < header_.size() && header_[e] != ' ' && header_[e] != '\t') ++e;
    r.name = header_.substr(1, e - 1);
    have_header_ = false;
    if (!fastq) {
      while (getline(line)) {
        if (!line.empty() && line[0] == '>') { header_ = line; have_header_ = true; break; }
        r.bases += line;
      }
      return true;
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
  bool getline(std::string &out) {
    out.clear();
    while (true) {
      if (pos_ == len_) {
        if (!f_) return !out.empty();
        len_ = fread(buf_.data(), 1, buf_.size(), f_);
        pos_ = 0;
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
  size_t pos_ = 0, len_ = 0;
  bool have_header_ = false;
  std::string header_;
};
