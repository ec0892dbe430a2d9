// fasta_bench.cpp -- the CLI's two FASTA readers timed on one file, three runs each:  g++ -O2 -std=c++17 -pthread tools/native/fasta_bench.cpp -o /tmp/fasta_bench && MFX_CLI_SEQ_TIMING=1 /tmp/fasta_bench <file>
#include <chrono>
#include <stdio.h>
#include "../../merfin_amd/cli/fasta.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
  for (int rep = 0; rep < 3; ++rep) {
    { double t = now(); std::vector<SeqRecord> b; bool ok = read_fasta_parallel(argv[1], b); size_t n = 0; for (auto &r : b) n += r.size();
      printf("parallel  %d: %.3f s  ok=%d recs=%zu bases=%zu\n", rep, now() - t, ok, b.size(), n); }
    { double t = now(); std::vector<SeqRecord> a; SeqFile sf(argv[1]); SeqRecord r; while (sf.next(r)) a.push_back(std::move(r)); size_t n = 0; for (auto &x : a) n += x.size();
      printf("sequential %d: %.3f s recs=%zu bases=%zu\n", rep, now() - t, a.size(), n); }
  }
}
