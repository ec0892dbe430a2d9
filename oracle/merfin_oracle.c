/*
 * merfin_oracle.c -- CPU restatement of the reference merfin evaluation path.
 * TEST INFRASTRUCTURE ONLY; see merfin_oracle.h for the rules and for the
 * "parity unpinned" statement.  Every function cites the reference lines it
 * follows (paths relative to /root/reference).
 *
 * Build flags mirror the reference's FP-relevant ones (src/Makefile:420,460):
 * optimised, -funroll-loops, no -ffast-math, no -march.  See oracle/Makefile.
 */
#include "merfin_oracle.h"

#include <assert.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* k-mer substrate                                                           */
/* ======================================================================== */

/* meryl-utility kmerTiny encoding ([H]): code = (c>>1)&3
 * gives A=0 C=1 T=2 G=3 for both cases; only ACGTacgt extend a run. */
int orc_base_code(unsigned char c) {
  switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'T': case 't': return 2;
    case 'G': case 'g': return 3;
    default:            return -1;
  }
}

void orc_kiter_init(orc_kiter *it, int k, const char *bases, uint64_t len) {
  assert(k >= 1 && k <= 32);
  it->k = k;
  it->bases = bases;
  it->len = len;
  it->pos = 0;
  it->fmer = 0;
  it->rmer = 0;
  it->mask = (k == 32) ? ~(uint64_t)0 : (((uint64_t)1 << (2 * k)) - 1);
  it->run = 0;
}

/* kmerIterator::nextBase as used at merfin-histogram.C:54 / merfin-dump.C:44:
 * returns true once per input byte; a valid base shifts into the forward mer
 * on the right and its complement into the reverse mer on the left; any other
 * byte zeroes the run length. */
int orc_kiter_next_base(orc_kiter *it) {
  if (it->pos >= it->len)
    return 0;
  int c = orc_base_code((unsigned char)it->bases[it->pos++]);
  if (c < 0) {
    it->run = 0;
    return 1;
  }
  it->fmer = ((it->fmer << 2) | (uint64_t)c) & it->mask;
  it->rmer = (it->rmer >> 2) | ((uint64_t)(c ^ 2) << (2 * it->k - 2));
  if (it->run < 0xffffffffu)
    it->run++;
  return 1;
}

int orc_kiter_is_valid(const orc_kiter *it) { return it->run >= (uint32_t)it->k; }

/* "start position (0-based) of the kmer" (merfin.C:316) */
uint64_t orc_kiter_position(const orc_kiter *it) { return it->pos - (uint64_t)it->k; }

uint64_t orc_revcomp(uint64_t kmer, int k) {
  uint64_t r = 0;
  for (int i = 0; i < k; i++) {
    r = (r << 2) | ((kmer & 3) ^ 2);
    kmer >>= 2;
  }
  return r;
}

uint64_t orc_canonical(uint64_t kmer, int k) {
  uint64_t r = orc_revcomp(kmer, k);
  return kmer < r ? kmer : r;
}

uint64_t orc_encode(const char *s, int k, int *ok) {
  uint64_t m = 0;
  if (ok) *ok = 1;
  for (int i = 0; i < k; i++) {
    int c = orc_base_code((unsigned char)s[i]);
    if (c < 0) { if (ok) *ok = 0; return 0; }
    m = (m << 2) | (uint64_t)c;
  }
  return m;
}

/* ======================================================================== */
/* merylExactLookup restatement                                              */
/* ======================================================================== */

struct orc_lookup {
  int       k;
  int       prefix_bits, suffix_bits;
  uint64_t  n;
  uint64_t *bgn;      /* 2^prefix_bits + 1 bucket starts */
  uint64_t *suffix;   /* n suffixes, sorted inside each bucket */
  uint32_t *value;    /* n values */
};

typedef struct { uint64_t k; uint32_t v; } kv_t;

static int kv_cmp(const void *a, const void *b) {
  uint64_t x = ((const kv_t *)a)->k, y = ((const kv_t *)b)->k;
  return (x > y) - (x < y);
}

/* merfin-globals.C:135-136,156: load(reader, maxMem, prefixSize=0(auto), minV,
 * maxV).  Auto prefix size here: the width that minimises pointer+suffix
 * bits, p ~= log2(n / log2 n) (what the meryl memory model converges to). */
orc_lookup *orc_lookup_build(int k, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                             uint64_t minV, uint64_t maxV, int prefix_bits) {
  assert(k >= 1 && k <= 32);
  orc_lookup *L = (orc_lookup *)calloc(1, sizeof(*L));
  kv_t *kv = (kv_t *)malloc((n ? n : 1) * sizeof(kv_t));
  uint64_t m = 0;
  int sorted = 1;
  for (uint64_t i = 0; i < n; i++) {
    if ((uint64_t)values[i] < minV || (uint64_t)values[i] > maxV)
      continue;
    if (values[i] == 0)
      continue;                       /* value 0 == absent (merfin-globals.C:84) */
    if (m && kv[m - 1].k >= kmers[i]) sorted = 0;
    kv[m].k = kmers[i];
    kv[m].v = values[i];
    m++;
  }
  if (!sorted)
    qsort(kv, m, sizeof(kv_t), kv_cmp);
  /* merge duplicates (a multiset DB would add values) */
  uint64_t w = 0;
  for (uint64_t i = 0; i < m; i++) {
    if (w && kv[w - 1].k == kv[i].k) kv[w - 1].v += kv[i].v;
    else kv[w++] = kv[i];
  }
  m = w;

  if (prefix_bits <= 0) {
    double lg = m > 2 ? log2((double)m) : 1.0;
    double p = m > 2 ? floor(log2((double)m / lg)) : 0.0;
    prefix_bits = (int)p;
  }
  if (prefix_bits > 2 * k) prefix_bits = 2 * k;
  if (prefix_bits > 30) prefix_bits = 30;
  if (prefix_bits < 0) prefix_bits = 0;

  L->k = k;
  L->prefix_bits = prefix_bits;
  L->suffix_bits = 2 * k - prefix_bits;
  L->n = m;
  uint64_t nb = (uint64_t)1 << prefix_bits;
  L->bgn = (uint64_t *)calloc(nb + 1, sizeof(uint64_t));
  L->suffix = (uint64_t *)malloc((m ? m : 1) * sizeof(uint64_t));
  L->value = (uint32_t *)malloc((m ? m : 1) * sizeof(uint32_t));
  uint64_t smask = (L->suffix_bits >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << L->suffix_bits) - 1);
  for (uint64_t i = 0; i < m; i++) {
    uint64_t pfx = (L->suffix_bits >= 64) ? 0 : (kv[i].k >> L->suffix_bits);
    L->bgn[pfx + 1]++;
    L->suffix[i] = kv[i].k & smask;
    L->value[i] = kv[i].v;
  }
  for (uint64_t b = 0; b < nb; b++)
    L->bgn[b + 1] += L->bgn[b];
  free(kv);
  return L;
}

/* merylExactLookup::value (merfin-globals.C:107-108): prefix -> bucket,
 * binary search while wider than 8, then linear; miss -> 0. */
uint32_t orc_lookup_value(const orc_lookup *L, uint64_t kmer) {
  uint64_t pfx = (L->suffix_bits >= 64) ? 0 : (kmer >> L->suffix_bits);
  uint64_t sfx = (L->suffix_bits >= 64) ? kmer : (kmer & (((uint64_t)1 << L->suffix_bits) - 1));
  uint64_t lo = L->bgn[pfx], hi = L->bgn[pfx + 1];
  while (hi - lo > 8) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (L->suffix[mid] <= sfx) lo = mid; else hi = mid;
  }
  for (uint64_t i = lo; i < hi; i++)
    if (L->suffix[i] == sfx)
      return L->value[i];
  return 0;
}

uint64_t orc_lookup_size(const orc_lookup *L) { return L->n; }
int orc_lookup_k(const orc_lookup *L) { return L->k; }

void orc_lookup_free(orc_lookup *L) {
  if (!L) return;
  free(L->bgn); free(L->suffix); free(L->value); free(L);
}

uint64_t orc_lookup_export(const orc_lookup *L, uint64_t *kmers, uint32_t *values) {
  uint64_t nb = (uint64_t)1 << L->prefix_bits;
  for (uint64_t b = 0; b < nb; b++)
    for (uint64_t i = L->bgn[b]; i < L->bgn[b + 1]; i++) {
      kmers[i] = (L->suffix_bits >= 64 ? 0 : (b << L->suffix_bits)) | L->suffix[i];
      values[i] = L->value[i];
    }
  return L->n;
}

/* ---- canonical counter (`meryl count`, merfin-globals.C:182-186) --------- */
struct orc_counter { int k; uint64_t n, cap; uint64_t *mers; };

orc_counter *orc_counter_new(int k) {
  orc_counter *c = (orc_counter *)calloc(1, sizeof(*c));
  c->k = k;
  return c;
}

void orc_counter_add(orc_counter *c, const char *bases, uint64_t len) {
  orc_kiter it;
  orc_kiter_init(&it, c->k, bases, len);
  while (orc_kiter_next_base(&it)) {
    if (!orc_kiter_is_valid(&it)) continue;
    if (c->n == c->cap) {
      c->cap = c->cap ? c->cap * 2 : 1024;
      c->mers = (uint64_t *)realloc(c->mers, c->cap * sizeof(uint64_t));
    }
    c->mers[c->n++] = it.fmer < it.rmer ? it.fmer : it.rmer;
  }
}

static int u64_cmp(const void *a, const void *b) {
  uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return (x > y) - (x < y);
}

uint64_t orc_counter_finish(orc_counter *c, uint64_t **kmers, uint32_t **values) {
  qsort(c->mers, c->n, sizeof(uint64_t), u64_cmp);
  uint64_t d = 0;
  for (uint64_t i = 0; i < c->n; i++)
    if (i == 0 || c->mers[i] != c->mers[i - 1]) d++;
  *kmers = (uint64_t *)malloc((d ? d : 1) * sizeof(uint64_t));
  *values = (uint32_t *)malloc((d ? d : 1) * sizeof(uint32_t));
  uint64_t w = 0;
  for (uint64_t i = 0; i < c->n; i++) {
    if (i == 0 || c->mers[i] != c->mers[i - 1]) { (*kmers)[w] = c->mers[i]; (*values)[w] = 1; w++; }
    else (*values)[w - 1]++;
  }
  free(c->mers);
  free(c);
  return d;
}

void orc_free(void *p) { free(p); }

/* ======================================================================== */
/* K* arithmetic                                                             */
/* ======================================================================== */

/* merfin-globals.C:66-98, statement for statement. */
void orc_getK_values(const orc_params *p, uint32_t seqValue, uint32_t asmValue,
                     double *readK, double *asmK, double *prob) {
  *readK = 0.0;
  *asmK = asmValue;                                   /* :81 */
  *prob = 1.0;

  if (seqValue == 0)                                  /* :84 */
    *readK = 0;
  else if (seqValue < p->peak)                        /* :86 uint32 -> double compare */
    *readK = 1;
  else
    *readK = round(seqValue / p->peak);               /* :89 half away from zero */

  if ((seqValue > 0) && (seqValue <= p->n_prob)) {    /* :93-97 */
    *readK = p->probK[seqValue - 1];
    *prob = p->probP[seqValue - 1];
  }
}

/* merfin-globals.C:101-110: both strands probed in both tables and summed
 * (kmvalu = uint32 arithmetic). */
void orc_getK_kmers(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                    uint64_t fmer, uint64_t rmer, double *readK, double *asmK, double *prob) {
  uint32_t rv = orc_lookup_value(R, fmer) + orc_lookup_value(R, rmer);
  uint32_t av = orc_lookup_value(A, fmer) + orc_lookup_value(A, rmer);
  orc_getK_values(p, rv, av, readK, asmK, prob);
}

/* merfin-globals.H:248-261 */
double orc_getKmetric(double readK, double asmK) {
  if (readK == 0)
    return 0;
  if (asmK > readK)
    return (asmK / readK - 1) * -1;
  if (asmK < readK)
    return readK / asmK - 1;
  return 0;
}

/* merfin-histogram.C:22-31 */
double orc_histoQV(double kval, double ktot, int k) {
  double base = kval / ktot;
  double kinv = 1.0 / k;
  double qv = -10.0 * log10(1.0 - pow(1.0 - base, kinv));
  return qv;
}

/* merfin-globals.C:21-62.  Lines are split on ','; a line with exactly two
 * fields is a row (row n <-> multiplicity n), anything else is reported as
 * invalid and skipped. */
int orc_load_kmetric(const char *path, uint32_t **K, double **P) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  uint32_t cap = 256, n = 0;
  *K = (uint32_t *)malloc(cap * sizeof(uint32_t));
  *P = (double *)malloc(cap * sizeof(double));
  char line[4096];
  while (fgets(line, sizeof(line), f)) {
    size_t L = strlen(line);
    while (L && (line[L - 1] == '\n' || line[L - 1] == '\r')) line[--L] = 0;
    /* split on ',' (runs of separators collapse, as splitToWords does) */
    char *w[4]; int nw = 0; char *s = line;
    while (*s) {
      while (*s == ',') *s++ = 0;
      if (!*s) break;
      if (nw < 4) w[nw] = s;
      nw++;
      while (*s && *s != ',') s++;
    }
    if (nw == 2) {
      if (n == cap) { cap *= 2; *K = (uint32_t *)realloc(*K, cap * sizeof(uint32_t)); *P = (double *)realloc(*P, cap * sizeof(double)); }
      (*K)[n] = (uint32_t)strtoul(w[0], NULL, 10);
      (*P)[n] = strtod(w[1], NULL);
      n++;
    }
  }
  fclose(f);
  return (int)n;
}

/* ======================================================================== */
/* -hist                                                                     */
/* ======================================================================== */

void orc_hist_init(orc_hist *h) { memset(h, 0, sizeof(*h)); }
void orc_hist_free(orc_hist *h) { free(h->undr); free(h->over); memset(h, 0, sizeof(*h)); }

/* increaseArray(..., copyData|clearNew): grow to hold index `need-1`, in
 * 1024-entry steps, new entries zero (merfin-histogram.C:74,87,116,121). */
static void grow(uint64_t **a, uint32_t *max, uint64_t need) {
  if (need <= *max) return;
  uint64_t nm = ((need + 1023) / 1024) * 1024;
  *a = (uint64_t *)realloc(*a, nm * sizeof(uint64_t));
  memset(*a + *max, 0, (nm - *max) * sizeof(uint64_t));
  *max = (uint32_t)nm;
}

/* bin one evaluated k-mer: merfin-histogram.C:58-90 after the getK call */
static inline void hist_bin(orc_hist *s, double readK, double asmK, double prob) {
  if (readK == 0) {                                       /* :66-69 */
    s->kmissing++;
    return;
  }
  if (asmK > readK) {                                     /* :71-82 */
    uint32_t idx = ((asmK / readK - 1) + 0.1) / 0.2;
    grow(&s->undr, &s->undrMax, (uint64_t)idx + 1);
    s->undr[idx]++;
    s->koverCpy += (1.0 - readK / asmK) * prob;
  } else {                                                /* :84-90 */
    uint32_t idx = ((readK / asmK - 1) + 0.1) / 0.2;
    grow(&s->over, &s->overMax, (uint64_t)idx + 1);
    s->over[idx]++;
  }
}

/* processHistogram restricted to k-mers whose LAST base lies in [e0,e1);
 * the iterator is primed with the k-1 preceding bases.  e0=0,e1=len is the
 * reference's whole-contig loop (merfin-histogram.C:54-91). */
static void hist_range(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                       const char *bases, uint64_t len, uint64_t e0, uint64_t e1, orc_hist *s) {
  if (!s->undr) grow(&s->undr, &s->undrMax, 1024);        /* :45-46 */
  if (!s->over) grow(&s->over, &s->overMax, 1024);
  uint64_t b0 = e0 >= (uint64_t)(p->k - 1) ? e0 - (uint64_t)(p->k - 1) : 0;
  if (e1 > len) e1 = len;
  orc_kiter it;
  orc_kiter_init(&it, p->k, bases + b0, e1 - b0);
  double readK = 0, asmK = 0, prob = 0;
  while (orc_kiter_next_base(&it)) {                      /* :54 */
    if (!orc_kiter_is_valid(&it))                         /* :55 */
      continue;
    if (b0 + it.pos - 1 < e0)                             /* priming region (tiled mode only) */
      continue;
    s->kasm++;                                            /* :58 */
    orc_getK_kmers(p, R, A, it.fmer, it.rmer, &readK, &asmK, &prob);   /* :63 */
    hist_bin(s, readK, asmK, prob);
  }
}

void orc_process_histogram(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                           const char *bases, uint64_t len, orc_hist *out) {
  hist_range(p, R, A, bases, len, 0, len, out);
}

/* outputHistogram, merfin-histogram.C:105-133 */
double orc_output_histogram(const orc_params *p, orc_hist *g, const orc_hist *s) {
  if (!g->undr) {                                         /* :105-108 */
    grow(&g->undr, &g->undrMax, 2048);
    grow(&g->over, &g->overMax, 2048);
  }
  g->kmissing += s->kmissing;                             /* :112-114 */
  g->kasm += s->kasm;
  g->koverCpy += s->koverCpy;
  grow(&g->undr, &g->undrMax, s->undrMax);                /* :116-119 */
  for (uint32_t i = 0; i < s->undrMax; i++) g->undr[i] += s->undr[i];
  grow(&g->over, &g->overMax, s->overMax);                /* :121-124 */
  for (uint32_t i = 0; i < s->overMax; i++) g->over[i] += s->over[i];
  return orc_histoQV((double)s->kmissing, (double)s->kasm, p->k);   /* :133 */
}

/* reportHistogram, merfin-histogram.C:140-176 */
void orc_report_histogram(const orc_params *p, const orc_hist *g, FILE *hist, FILE *summary) {
  if (!g->undr || !g->over)                               /* :145-147 */
    return;
  if (hist) {
    for (uint64_t ii = g->undrMax - 1; ii > 0; ii--)      /* :153-155 */
      if (g->undr[ii] > 0)
        fprintf(hist, "%.1f\t%lu\n", ((double)ii * -0.2), (unsigned long)g->undr[ii]);
    fprintf(hist, "%.1f\t%lu\n", 0.0, (unsigned long)(g->undr[0] + g->over[0]));   /* :157 */
    for (uint64_t ii = 1; ii < g->overMax; ii++)          /* :159-161 */
      if (g->over[ii] > 0)
        fprintf(hist, "%.1f\t%lu\n", ((double)ii * 0.2), (unsigned long)g->over[ii]);
  }
  if (summary) {                                          /* :167-175 */
    fprintf(summary, "\n");
    fprintf(summary, "K-mers not found in reads (missing) : %lu\n", (unsigned long)g->kmissing);
    fprintf(summary, "K-mers overly represented in assembly: %.2f\n", g->koverCpy);
    fprintf(summary, "K-mers found in the assembly: %lu\n", (unsigned long)g->kasm);
    fprintf(summary, "Missing QV: %.2f\n", orc_histoQV((double)g->kmissing, (double)g->kasm, p->k));
    fprintf(summary, "Merfin QV*: %.2f\n", orc_histoQV(g->kmissing + g->koverCpy, (double)g->kasm, p->k));
    fprintf(summary, "*** Note this QV is valid only if -seqmer was generated with -sequence ***\n\n");
    fprintf(summary, "*** Missing QV only considers missing kmers as errors. Merfin QV* includes overrepresented kmers. ***\n\n");
    fprintf(summary, "*** When the lookup table is provided, missing QV includes weighted low frequency kmers, otherwise it is identical to Merqury QV. ***\n\n");
  }
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* The reference's scheduling (merfin.C:377-413): N workers, one contig per
 * work item (mode 0).  Mode 1 tiles positions so the CPU baseline is not
 * capped at #contigs-way parallelism. */
double orc_hist_run(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                    const char *const *contigs, const uint64_t *lens, uint32_t ncontigs,
                    int threads, int mode, uint64_t tile, orc_hist *global,
                    uint64_t *contig_kasm, uint64_t *contig_kmissing) {
  if (threads < 1) threads = 1;
  /* work items */
  uint64_t nitems = 0;
  if (mode == 0) nitems = ncontigs;
  else {
    if (tile == 0) tile = 1 << 20;
    for (uint32_t c = 0; c < ncontigs; c++) nitems += (lens[c] + tile - 1) / tile;
  }
  uint32_t *item_c = (uint32_t *)malloc((nitems ? nitems : 1) * sizeof(uint32_t));
  uint64_t *item_e0 = (uint64_t *)malloc((nitems ? nitems : 1) * sizeof(uint64_t));
  uint64_t *item_e1 = (uint64_t *)malloc((nitems ? nitems : 1) * sizeof(uint64_t));
  uint64_t w = 0;
  for (uint32_t c = 0; c < ncontigs; c++) {
    if (mode == 0) { item_c[w] = c; item_e0[w] = 0; item_e1[w] = lens[c]; w++; }
    else for (uint64_t e = 0; e < lens[c]; e += tile) {
      item_c[w] = c; item_e0[w] = e; item_e1[w] = e + tile < lens[c] ? e + tile : lens[c]; w++;
    }
  }
  orc_hist *res = (orc_hist *)calloc(nitems ? nitems : 1, sizeof(orc_hist));

  double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
  for (int64_t i = 0; i < (int64_t)nitems; i++)
    hist_range(p, R, A, contigs[item_c[i]], lens[item_c[i]], item_e0[i], item_e1[i], &res[i]);
  /* the serialised writer (outputHistogram), here in input order */
  for (uint64_t i = 0; i < nitems; i++) {
    orc_output_histogram(p, global, &res[i]);
    if (contig_kasm) contig_kasm[item_c[i]] += res[i].kasm;
    if (contig_kmissing) contig_kmissing[item_c[i]] += res[i].kmissing;
  }
  double t1 = now_s();

  for (uint64_t i = 0; i < nitems; i++) orc_hist_free(&res[i]);
  free(res); free(item_c); free(item_e0); free(item_e1);
  return t1 - t0;
}

/* ======================================================================== */
/* -dump                                                                     */
/* ======================================================================== */

/* processDump, merfin-dump.C:34-67 (skipMissing == false branch) */
void orc_process_dump(const orc_params *p, const orc_lookup *R, const orc_lookup *A,
                      const char *bases, uint64_t len,
                      double *dumpReadK, double *dumpAsmK, double *dumpKMetric,
                      uint64_t *kasm, uint64_t *kmissing) {
  if (dumpReadK) {
    memset(dumpReadK, 0, (len + 1) * sizeof(double));     /* :35-37 clearNew */
    memset(dumpAsmK, 0, (len + 1) * sizeof(double));
    memset(dumpKMetric, 0, (len + 1) * sizeof(double));
  }
  orc_kiter it;
  orc_kiter_init(&it, p->k, bases, len);
  double readK = 0, asmK = 0, prob = 0;
  while (orc_kiter_next_base(&it)) {                      /* :44 */
    if (!orc_kiter_is_valid(&it))
      continue;
    (*kasm)++;                                            /* :48 */
    orc_getK_kmers(p, R, A, it.fmer, it.rmer, &readK, &asmK, &prob);
    if (readK == 0)                                       /* :56-58 */
      (*kmissing)++;
    if (dumpReadK) {                                      /* :60-66 */
      uint64_t pp = orc_kiter_position(&it);
      dumpReadK[pp] = readK;
      dumpAsmK[pp] = asmK;
      dumpKMetric[pp] = orc_getKmetric(readK, asmK);
    }
  }
}

/* outputDump, merfin-dump.C:87-93 */
uint64_t orc_output_dump(FILE *f, const char *name, uint64_t len,
                         const double *dumpReadK, const double *dumpAsmK, const double *dumpKMetric) {
  uint64_t lines = 0;
  for (uint64_t pp = 0; pp < len; pp++)
    if ((dumpReadK[pp] != 0.0) || (dumpAsmK[pp] != 0.0) || (dumpKMetric[pp] != 0.0)) {
      fprintf(f, "%s\t%lu\t%.2f\t%.2f\t%.2f\n", name, (unsigned long)pp, dumpReadK[pp], dumpAsmK[pp], dumpKMetric[pp]);
      lines++;
    }
  return lines;
}

/* ======================================================================== */
/* -completeness                                                             */
/* ======================================================================== */

/* The loop body of merfin-completeness.C:70-117 over two sorted k-mer lists
 * (one of the 64 pieces).  Note asm-only k-mers are skipped (:106-109) and
 * -min/-max are not applied (raw merylFileReader). */
void orc_completeness_piece(const orc_params *p,
                            const uint64_t *rk, const uint32_t *rv, uint64_t rn,
                            const uint64_t *ak, const uint32_t *av, uint64_t an,
                            double *total, double *undrc) {
  uint64_t i = 0, j = 0;
  double t = 0.0, u = 0.0;
  while (i < rn || j < an) {
    double readK = 0.0, asmK = 0.0, prob = 0.0;
    if (i < rn && j < an && rk[i] == ak[j]) {             /* :79-86 */
      orc_getK_values(p, rv[i], av[j], &readK, &asmK, &prob);
      i++; j++;
    } else if (i < rn && (j >= an || rk[i] < ak[j])) {    /* :93-98 */
      orc_getK_values(p, rv[i], 0, &readK, &asmK, &prob);
      i++;
    } else {                                              /* :106-109 */
      j++;
      continue;
    }
    t += readK;                                           /* :113 */
    if (readK > asmK)                                     /* :115-116 */
      u += readK - asmK;
  }
  *total = t;
  *undrc = u;
}
