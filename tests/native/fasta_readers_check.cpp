// fasta_readers_check.cpp -- test program (tests/test_fasta_reader.py): reads one FASTA file with the CLI's two readers,
// the sequential SeqFile::next and the all-threads read_fasta_parallel, and prints either "same <records> <bases>" or the
// first difference.  `dump` as second argument prints the sequential reader's records (name, length, FNV-1a of the bases).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../merfin_amd/cli/fasta.h"

static uint64_t fnv(const char *p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
  return h;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::vector<SeqRecord> a, b;
  {
    SeqFile sf(argv[1]);
    if (!sf.ok()) { printf("cannot open\n"); return 1; }
    SeqRecord r;
    while (sf.next(r)) a.push_back(std::move(r));
    if (sf.finish()) { printf("reader failed\n"); return 1; }
  }
  if (argc > 2 && !strcmp(argv[2], "dump")) {
    for (auto &r : a) printf("%s\t%zu\t%016llx\n", r.name.c_str(), r.size(), (unsigned long long)fnv(r.data(), r.size()));
    return 0;
  }
  if (!read_fasta_parallel(argv[1], b)) { printf("not applicable\n"); return 0; }
  if (a.size() != b.size()) { printf("records: %zu vs %zu\n", a.size(), b.size()); return 1; }
  size_t total = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    if (a[i].name != b[i].name) { printf("record %zu: name '%s' vs '%s'\n", i, a[i].name.c_str(), b[i].name.c_str()); return 1; }
    if (a[i].size() != b[i].size()) { printf("record %zu: length %zu vs %zu\n", i, a[i].size(), b[i].size()); return 1; }
    if (memcmp(a[i].data(), b[i].data(), a[i].size())) { printf("record %zu: bases differ\n", i); return 1; }
    total += a[i].size();
  }
  printf("same %zu %zu\n", a.size(), total);
  return 0;
}
