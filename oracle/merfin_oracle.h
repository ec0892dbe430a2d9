/*
 * merfin_oracle.h -- CPU restatement of the reference merfin k-mer evaluation
 * path (-hist / -dump / -completeness arithmetic).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the reported CPU baseline.  Nothing under merfin_amd/
 * links, imports or executes it; the product path fails loudly when its HIP
 * library is missing instead of falling back to this.
 *
 * PARITY STATUS: "parity unpinned" at the meryl boundary.  The reference
 * (/root/reference/src/merfin) cannot be compiled here: every translation
 * unit includes meryl-utility headers (merfin-globals.H:16-20) from the
 * submodules src/meryl and src/utility (.gitmodules:1-6), which are empty in
 * the checkout and whose pinned SHAs are unrecorded (no .git).  The reference
 * ships no tests and no golden vectors.  What this oracle is pinned against:
 *   (1) hand-derived IEEE-754 known-answer values for every in-tree formula
 *       (tests/test_oracle_kat.py),
 *   (2) the reference's one data fixture, the 184-row -prob table
 *       (scripts/lookup_table/example_lookup_table.txt; committed as
 *       tests/golden/example_lookup_table.txt -- data, not source),
 *   (3) an independent pure-Python/numpy restatement of the same formulas on
 *       small random inputs (tests/test_oracle_vs_numpy.py).
 * Everything *above* merylExactLookup::value() is restated line by line from
 * in-tree source and cited below.  Everything *below* it (k-mer encoding,
 * iterator validity, lookup layout) follows the published behaviour of
 * marbl/meryl-utility as recalled in SURVEY.md Appendix C and is anchored on
 * the reference's call sites.
 */
#ifndef MERFIN_ORACLE_H
#define MERFIN_ORACLE_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- k-mer substrate (meryl-utility kmerIterator / kmerTiny; call sites
 *      merfin.C:45, merfin-histogram.C:54-64, merfin-dump.C:44-61) ---------- */

/* 2-bit code of a base: A=0 C=1 T=2 G=3 for either case ((c>>1)&3); -1 for any
 * other byte.  Complement = code ^ 2. */
int orc_base_code(unsigned char c);

typedef struct {
  int            k;
  const char    *bases;
  uint64_t       len;
  uint64_t       pos;     /* next byte to consume                       */
  uint64_t       fmer;    /* forward k-mer, 2k low bits                  */
  uint64_t       rmer;    /* reverse-complement k-mer                    */
  uint64_t       mask;
  uint32_t       run;     /* valid bases in the current run (saturates)  */
} orc_kiter;

void     orc_kiter_init(orc_kiter *it, int k, const char *bases, uint64_t len);
int      orc_kiter_next_base(orc_kiter *it);           /* 1 once per input byte, 0 at end */
int      orc_kiter_is_valid(const orc_kiter *it);      /* >= k valid bases in the run     */
uint64_t orc_kiter_position(const orc_kiter *it);      /* 0-based start of current k-mer  */

uint64_t orc_revcomp(uint64_t kmer, int k);
uint64_t orc_canonical(uint64_t kmer, int k);
/* encode an ASCII k-mer; returns 0 and sets *ok=0 on a non-ACGT byte */
uint64_t orc_encode(const char *s, int k, int *ok);

/* ---- merylExactLookup restatement (merfin-globals.C:135-159,107-108) ------
 * prefix-pointer table + sorted suffix buckets, binary search while the
 * bucket is > 8 wide then a linear scan; miss -> 0.  Built from sorted or
 * unsorted (kmer,value) pairs; pairs with value outside [minV,maxV] are
 * dropped at load ("Ignore kmers with value below/above m", merfin.C:199-200).
 * Suffixes/values are stored unpacked (uint64/uint32) -- faster on a CPU than
 * meryl's bit-packed words, so the CPU baseline is not handicapped. */
typedef struct orc_lookup orc_lookup;

orc_lookup *orc_lookup_build(int k, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                             uint64_t minV, uint64_t maxV, int prefix_bits /* 0 = auto */);
uint32_t    orc_lookup_value(const orc_lookup *L, uint64_t kmer);
uint64_t    orc_lookup_size(const orc_lookup *L);
int         orc_lookup_k(const orc_lookup *L);
void        orc_lookup_free(orc_lookup *L);
/* copy out the (sorted) contents; returns n */
uint64_t    orc_lookup_export(const orc_lookup *L, uint64_t *kmers, uint32_t *values);

/* canonical k-mer counting of a sequence set (what `meryl count` produces for
 * -seqmers, merfin-globals.C:182-186): sort-based, exact.  Appends the valid
 * canonical k-mers of (bases,len) to a growing list; finish() sorts + counts. */
typedef struct orc_counter orc_counter;
orc_counter *orc_counter_new(int k);
void         orc_counter_add(orc_counter *c, const char *bases, uint64_t len);
/* returns number of distinct k-mers; arrays are malloc'd, caller frees with orc_free */
uint64_t     orc_counter_finish(orc_counter *c, uint64_t **kmers, uint32_t **values);
void         orc_free(void *p);

/* ---- K* arithmetic (merfin-globals.C:66-110, merfin-globals.H:248-261) --- */
typedef struct {
  int             k;
  double          peak;       /* -peak                                     */
  uint32_t        n_prob;     /* rows of the -prob table (0 = none)        */
  const uint32_t *probK;      /* copyKmerK                                 */
  const double   *probP;      /* copyKmerP                                 */
} orc_params;

/* merfin-globals.C:66-98 */
void   orc_getK_values(const orc_params *p, uint32_t readV, uint32_t asmV,
                       double *readK, double *asmK, double *prob);
/* merfin-globals.C:101-110: four probes, fwd+rev in each table */
void   orc_getK_kmers(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                      uint64_t fmer, uint64_t rmer, double *readK, double *asmK, double *prob);
/* merfin-globals.H:248-261 */
double orc_getKmetric(double readK, double asmK);
/* merfin-histogram.C:22-31 */
double orc_histoQV(double kval, double ktot, int k);
/* merfin-globals.C:21-62; returns rows loaded or -1 if the file is missing.
 * arrays malloc'd (free with orc_free). */
int    orc_load_kmetric(const char *path, uint32_t **K, double **P);

/* ---- -hist (merfin-histogram.C) ------------------------------------------ */
typedef struct {
  uint64_t  kasm, kmissing;
  double    koverCpy;
  uint32_t  undrMax, overMax;
  uint64_t *undr, *over;
} orc_hist;

void orc_hist_init(orc_hist *h);
void orc_hist_free(orc_hist *h);
/* processHistogram, merfin-histogram.C:35-92, one contig */
void orc_process_histogram(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                           const char *bases, uint64_t len, orc_hist *out);
/* outputHistogram, merfin-histogram.C:96-136 (merge only; returns the QV it prints) */
double orc_output_histogram(const orc_params *p, orc_hist *global, const orc_hist *contig);
/* reportHistogram, merfin-histogram.C:140-176.  Either FILE may be NULL. */
void orc_report_histogram(const orc_params *p, const orc_hist *global, FILE *hist, FILE *summary);

/* whole-assembly drivers used for the CPU baseline: contigs[i] has lens[i]
 * bases.  mode 0: one contig per worker (merfin.C:408-410); mode 1: position
 * tiles of `tile` bases with a (k-1) halo.  Returns seconds spent in the
 * evaluate phase; result merged into *global in contig (tile) order. */
double orc_hist_run(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                    const char *const *contigs, const uint64_t *lens, uint32_t ncontigs,
                    int threads, int mode, uint64_t tile, orc_hist *global,
                    uint64_t *contig_kasm, uint64_t *contig_kmissing);

/* ---- -dump (merfin-dump.C) ------------------------------------------------ */
/* processDump, merfin-dump.C:20-68.  Arrays have len+1 entries, zeroed here. */
void orc_process_dump(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                      const char *bases, uint64_t len,
                      double *dumpReadK, double *dumpAsmK, double *dumpKMetric,
                      uint64_t *kasm, uint64_t *kmissing);
/* outputDump text, merfin-dump.C:87-93; returns lines written */
uint64_t orc_output_dump(FILE *f, const char *name, uint64_t len,
                         const double *dumpReadK, const double *dumpAsmK, const double *dumpKMetric);

/* ---- -completeness (merfin-completeness.C:48-144) -------------------------
 * sorted merge of the read and asm k-mer lists restricted to piece `ii` of 64
 * (top 6 bits of the 2k-bit k-mer = the meryl file number). */
void orc_completeness_piece(const orc_params *p,
                            const uint64_t *rk, const uint32_t *rv, uint64_t rn,
                            const uint64_t *ak, const uint32_t *av, uint64_t an,
                            double *total, double *undrc);

#ifdef __cplusplus
}
#endif
#endif
