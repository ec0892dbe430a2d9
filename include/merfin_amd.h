/*
 * merfin_amd.h -- C ABI of the MI355X-native k-mer multiplicity evaluator.
 *
 * This is the drop-in boundary for merfin's evaluation hot path (reference:
 * arangrhie/merfin, paths below relative to /root/reference).  merfin has no
 * plugin/FFI interface; the seams that exist in its source are
 *   (1) the sweatShop callback triple  load / process / output
 *       (src/merfin/merfin.C:30-31,59-65, wired at :377,383,389),
 *   (2) the lookup object  merylExactLookup::{estimateMemoryUsage,load,value}
 *       (src/merfin/merfin-globals.C:135-159,107-108),
 *   (3) the K* parameter fields of merfinGlobal
 *       (src/merfin/merfin-globals.H:172,222-227,239).
 * Each entry point below names the seam it replaces.  Plain pointers and
 * sizes only: no C++/torch types cross this boundary.  INTEGRATION.md shows
 * the binding a merfin maintainer would add on the reference side.
 *
 * Conventions
 *   - every function returning int returns 0 on success or a negative MFX_E_*
 *     code; mfx_last_error() then returns a thread-local description.  The
 *     library never calls exit() (the reference does: merfin-globals.C:31,152).
 *   - objects are bound to the HIP device they were created on; calls may come
 *     from any host thread.  An mfx_eval is used by one thread at a time.  An
 *     mfx_seq and an mfx_index that are only READ (evaluations, lookups, dumps,
 *     variant runs) may be shared by several evaluators driven from several
 *     threads -- the slots of one device (`merfin -devices 0,0,0,0`): what a
 *     sequence makes on first use is made under a lock, and no result depends
 *     on what other threads have queued on the device.  Calls that CHANGE an
 *     object (uploads into a sequence, adds / loads into an index) need it to
 *     themselves.  `stream` arguments are hipStream_t passed as void* (NULL =
 *     the device's default stream).
 *   - k-mers are 2k-bit integers, A=0 C=1 T=2 G=3, first base most significant
 *     (meryl's kmerTiny encoding), 1 <= k <= 64 as in the reference (its kmer type holds
 *     2k <= 128 bits).  For k <= 31 a k-mer is ONE uint64_t in every array of this ABI; for
 *     32 <= k <= 64 it is TWO consecutive uint64_t words {low 64 bits, high bits}, so a
 *     `const uint64_t *kmers` argument then points to 2*n words.  The k <= 31 path is the tuned
 *     one (BASELINE's configurations use k = 21 and 31); 32 <= k <= 64 runs the plain per-lane
 *     kernels of csrc/mfx_wide.hip and does not support a sharded index.
 */
#ifndef MERFIN_AMD_H
#define MERFIN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFX_OK            0
#define MFX_E_INVAL      -1   /* bad argument                                  */
#define MFX_E_NOMEM      -2   /* host or device allocation failed              */
#define MFX_E_HIP        -3   /* a HIP runtime call failed                     */
#define MFX_E_FULL       -4   /* index capacity exceeded                       */
#define MFX_E_OVERFLOW   -5   /* more DISTINCT far K* bins than an evaluator holds */
#define MFX_E_IO         -6   /* file could not be read / parsed               */
#define MFX_E_FORMAT     -7   /* file magic / version / layout not recognised  */
#define MFX_E_NODEVICE   -8   /* no usable HIP device                          */
#define MFX_E_NONCANON   -9   /* a sequence-only index met a non-canonical k-mer database */

const char *mfx_last_error(void);
int         mfx_last_error_code(void);   /* code of the last failing call on this thread */
const char *mfx_version(void);
int         mfx_device_count(void);
/* Optional (no reference counterpart): bring the device's context, this library's code object and the pinned-memory path up
 * now -- what the first upload would otherwise pay (~0.07 s) -- e.g. on a helper thread while the caller reads its FASTA. */
int         mfx_device_warm(int device);
/* free / total device memory in bytes (hipMemGetInfo): what a caller sizes its choice of index by (`merfin` takes the path-only index of the
 * variant modes when the full tables would not fit) */
int         mfx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes);

/* ------------------------------------------------------------------------ */
/* Index: replaces the two merylExactLookup objects readLookup / asmLookup  */
/* (merfin-globals.H:217,220).  One device-resident table holds BOTH counts */
/* of every k-mer, so one probe serves merfinGlobal::getK(kmer,kmer,...)    */
/* (merfin-globals.C:101-110), which does four.                             */
/* ------------------------------------------------------------------------ */
typedef struct mfx_index mfx_index;

/* merylExactLookup() + estimateMemoryUsage (merfin-globals.C:134-153):
 * capacity_kmers = upper bound on distinct k-mers over both sets.  max_gb is
 * the -memory cap in GB (0 = none): creation fails with MFX_E_NOMEM when the
 * table would not fit, mirroring "Not enough memory to load databases".
 * mfx_index_estimate_gb is the SMALLEST table made for that capacity (load factor 0.7, what the cap is checked
 * against); when HBM is plentiful the table is made emptier -- down to load factor 0.45, within 0.75 of the free
 * HBM and within max_gb -- because emptier lines mean fewer second probes (MFX_LOAD_FACTOR fixes the factor). */
mfx_index *mfx_index_create(int k, uint64_t capacity_kmers, double max_gb, int device);
void       mfx_index_free(mfx_index *ix);
double     mfx_index_estimate_gb(int k, uint64_t capacity_kmers);

/* merylExactLookup::load(reader, maxMem, 0, minV, maxV), merfin-globals.C:156
 * (read DB: values outside [minV,maxV] are dropped at load, merfin.C:199-200)
 * and :159 (asm DB: unfiltered).  kmers/values: n pairs, any order; on_device
 * != 0 means the arrays already live on this index's device. */
int mfx_index_add_read(mfx_index *ix, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                       uint64_t minV, uint64_t maxV, int on_device);
int mfx_index_add_asm(mfx_index *ix, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                      int on_device);

/* merylFileReader(path) (merfin-globals.C:118-119,139): open a k-mer database
 * and learn k ("Make readDB first so we know the k size") and its size.
 * Accepted forms: a meryl database directory (decoder UNVALIDATED, see
 * csrc/mfx_db.cpp), `meryl print` text (<kmer>\t<count>, optionally .gz/.bz2/.xz)
 * and this library's flat binary (mfx_db_write_flat). */
#define MFX_DB_MERYL 1
#define MFX_DB_TEXT  2
#define MFX_DB_FLAT  3
typedef struct {
  int      k;
  int      format;         /* MFX_DB_*                                        */
  uint64_t n_kmers;        /* distinct k-mers stored                          */
  int      placed;         /* flat form: the records are sorted by their PLACE in the compact table (mfx_db_convert_placed) */
} mfx_db_info;
int mfx_db_probe(const char *path, mfx_db_info *out);
/* merylExactLookup::load from disk: side 0 = read DB (-min/-max apply,
 * merfin-globals.C:156), side 1 = assembly DB (:159). */
int mfx_index_load_db(mfx_index *ix, const char *path, int side, uint64_t minV, uint64_t maxV);
/* the same into several tables with ONE pass over the database: the shards of one process (mfx_index_set_shard), each
 * keeping the k-mers it owns -- the chunk is staged once and sent to every table's device */
int mfx_index_load_db_multi(mfx_index *const *ixs, uint32_t nix, const char *path, int side, uint64_t minV, uint64_t maxV);
/* this library's flat binary form of a database (interchange; no reference counterpart).  Strictly ascending k-mers
 * (the order `meryl print` lists them in, k <= 31) are written as delta-coded blocks -- 2.6 bytes per k-mer of a 30x read
 * set, decoded by the kernel that inserts them; any other order as 8-byte packed records (k <= 21) or plain arrays.
 * All forms load into the same tables (csrc/mfx_db.cpp holds the layouts). */
int mfx_db_write_flat(const char *path, int k, const uint64_t *kmers, const uint32_t *values, uint64_t n);
/* any accepted database (meryl directory, `meryl print` text, flat) -> the flat form, sorted (k <= 31), on the host: no
 * device is touched.  One pass makes every later load of the database a matter of its bytes over PCIe (a 30x human read
 * set: a minute of text parsing -> half a second).  CLI: merfin -convert <db> -output <file>.  n_kmers may be null. */
int mfx_db_convert(const char *in_path, const char *out_path, uint64_t *n_kmers);
/* ... -> the PLACED flat form (13 <= k <= 31, canonical databases): the records are not the k-mers but the numbers P of
 * csrc/mfx_place.h -- one to one with the canonical k-mers, and ascending P means ascending LINE of the sequence-only index's compact
 * table, whatever its size -- sorted and delta-coded (~3 bits more per record than the k-mer-sorted form).  What it buys: the kernel
 * that applies the database to the table (load_Kmers, merfin-globals.C:155-159, as a run pays it) walks the table line after line,
 * every line read and written once, instead of reading one random line per record.  Such a file loads into ANY table (the k-mers are
 * recovered from P on the device; only the compact table under its default placement takes the shortcut).  CLI: merfin -convert <db>
 * -placed -output <file>.  mfx_db_write_flat_placed writes one from ascending P and counts (mfx_db_place_keys gives P of k-mers: host
 * arrays, or device arrays on `device` for a tool that sorts on the GPU); both for k <= 30 -- the P of a 31-mer takes 65 bits, its file holds
 * P >> 1 with the strand bit in the record's count field and is made by mfx_db_convert_placed alone. */
int mfx_db_convert_placed(const char *in_path, const char *out_path, uint64_t *n_kmers);
int mfx_db_write_flat_placed(const char *path, int k, const uint64_t *pkeys, const uint32_t *values, uint64_t n);
int mfx_db_place_keys(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, int on_device, int device);

/* The built table as a device-format image on disk: later runs on the same databases skip the decode +
 * insert (no reference counterpart; merfin rebuilds its lookup tables on every start, merfin.C:361). */
int        mfx_index_save(const mfx_index *ix, const char *path);
mfx_index *mfx_index_load(const char *path, double max_gb, int device);
/* An image is only as good as the inputs it was built from: the caller stores a digest of them (database and
 * sequence files: path, size, modification time; -min/-max) with the table and compares it on the next start;
 * mfx_index_get_origin also returns the read-count filter baked into the table, so that a run with other
 * -min/-max values can refuse the image instead of silently evaluating with the stored ones. */
int        mfx_index_set_fingerprint(mfx_index *ix, uint64_t fingerprint);
int        mfx_index_get_origin(const mfx_index *ix, uint64_t *fingerprint, uint64_t *minV, uint64_t *maxV);

/* The same image in memory, for handing a built table to the other GPUs of a node (one rank decodes the
 * k-mer databases and builds, the table then travels over xGMI -- an RCCL broadcast -- instead of every rank
 * decoding and building its own; merfin_amd/distributed.py::broadcast_index).
 *   source rank : mfx_index_image_header(ix, hdr)  -> MFX_INDEX_HEADER_BYTES describing geometry + filter
 *   other ranks : ix = mfx_index_create_from_header(hdr, max_gb, device)   (empty table of that geometry)
 *   all ranks   : mfx_index_device_image(ix, &lines, &line_bytes, &meta, &meta_bytes); move `lines` and `meta`
 *                 from the source's device buffers into the others'; then mfx_index_commit(ix) on the receivers. */
#define MFX_INDEX_HEADER_BYTES 128
int        mfx_index_image_header(const mfx_index *ix, void *hdr);
mfx_index *mfx_index_create_from_header(const void *hdr, double max_gb, int device);
int        mfx_index_device_image(mfx_index *ix, void **d_lines, uint64_t *line_bytes, void **d_meta, uint64_t *meta_bytes);
int        mfx_index_commit(mfx_index *ix);

typedef struct mfx_seq mfx_seq;

/* SEQUENCE-ONLY index: the lookup object for the report types that only ever ask it for the k-mers of -sequence --
 * -hist and -dump (merfin-histogram.C:54-64 and merfin-dump.C:44-61 call getK with the kmerIterator's fmer/rmer and
 * nothing else).  It holds exactly the canonical k-mers CLAIMED from a sequence; every later add / load only UPDATES
 * the counts of those k-mers and drops the rest (for a 30x human read set: half of the database, the sequencing-error
 * k-mers, never gets a slot).  value() of a claimed k-mer is what the full tables answer; value() of any other k-mer
 * is 0, so -completeness (merfin-completeness.C:48-144 walks the whole read database) and the variant modes (alternative
 * paths, varMer.C:76-84) need the full index of mfx_index_create and are refused on this one.
 *   ix = mfx_index_create_for_seq(k, <upper bound of the sequence's distinct k-mers: its bases>, max_gb, device);
 *   mfx_index_count_asm(ix, seq, 0)      -- claims the k-mers AND counts them into the assembly side (no -seqmers), or
 *   mfx_index_claim_seq(ix, seq, 0)      -- claims them only; the assembly counts then come from -seqmers
 *   mfx_index_load_db / mfx_index_add_*  -- update-only from here on; a further claim is an error
 * For k <= 21 the table takes a compact layout -- 8-byte slots {key 42 bits | readV 11 | asmV 11}, 16 per 128-byte line,
 * exact counts of saturated fields in a side table -- built directly in that form; a human assembly takes 96 GB and
 * -hist runs on 0.42 table lines per k-mer (MFX_SEQ_COMPACT=0 keeps the 16-byte slots).  The databases must be canonical
 * (one slot per canonical k-mer cannot answer value(fmer) + value(rmer) of a non-canonical one): a load that meets a
 * non-canonical k-mer fails with MFX_E_NONCANON and the caller builds the full index instead. */
mfx_index *mfx_index_create_for_seq(int k, uint64_t capacity_kmers, double max_gb, int device);
/* ... with the table's load factor chosen by the caller (0: by the free memory, as above; MFX_LOAD_FACTOR overrides both).  The
 * emptiest table probes fastest, but a short-lived process pays for the memory it asks for: one started behind another waits in
 * hipMalloc while the driver clears what the earlier one freed.  The `merfin` CLI asks for 0.4 (61 GB for a human assembly), a
 * resident evaluator for the default (0.18, 135 GB): profiles/r05_e2e_lf_ab.txt. */
mfx_index *mfx_index_create_for_seq_lf(int k, uint64_t capacity_kmers, double max_gb, int device, double load_factor);
mfx_index *mfx_index_create_lf(int k, uint64_t capacity_kmers, double max_gb, int device, double load_factor);     /* the full table, likewise (the CLI: 0.7) */
double     mfx_index_estimate_gb_for_seq(int k, uint64_t capacity_kmers);
int        mfx_index_claim_seq(mfx_index *ix, const mfx_seq *seq, void *stream);
/* PART of an assembly per device (round 4; config 5's -hist / -dump without any exchange): a device that evaluates some
 * contigs claims THEIR k-mers (mfx_index_claim_seq on a sequence object of those contigs), takes their assembly counts
 * from the WHOLE assembly -- asmV += 1 per occurrence of a claimed k-mer in `seq`, nothing is claimed --
 *   mfx_index_count_claimed(ix, whole_assembly, 0)
 * and loads the read database update-only as above.  value() of its contigs' k-mers is then what the full tables of the
 * whole run answer (`meryl count` of -sequence counts every contig, merfin-globals.C:182-186), the device evaluates its
 * contigs alone, and mfx_hist_run_parts adds the devices' results. */
int        mfx_index_count_claimed(mfx_index *ix, const mfx_seq *seq, void *stream);

/* Native replacement of the `meryl count k=.. <seq> output <seq>.meryl` child
 * process (merfin-globals.C:182-186): counts the canonical k-mers of every
 * contig of `seq` into the assembly side of the index, on the GPU. */
int mfx_index_count_asm(mfx_index *ix, const mfx_seq *seq, void *stream);
/* mfx_index_count_asm(ix, seq, 0) followed by mfx_index_load_db(ix, read_db_path, 0, minV, maxV) as one call: the database's
 * bytes cross PCIe while the sequence's k-mers are still being claimed and counted (the inserts wait for that kernel on the
 * device).  Same table as the two calls.  Replaces load_Kmers + the `meryl count` of -sequence, merfin-globals.C:114-163,182-186. */
int mfx_index_build_for_hist(mfx_index *ix, const mfx_seq *seq, const char *read_db_path, uint64_t minV, uint64_t maxV);
/* The same with the database's bytes moving from the moment the process starts (what the reference pays per run, load_Kmers,
 * merfin-globals.C:114-163, 155-159): mfx_db_stage_begin opens a delta-coded flat database (`merfin -convert`), takes device memory
 * for all of its blocks and lets a thread of its own read the file into it over PCIe -- under the FASTA read, the sequence upload, the
 * table's allocation and the claim / count kernel; mfx_index_build_for_hist_staged launches that kernel and, behind it, the
 * decode + update kernel over the staged blocks as their copies complete.  Same table as mfx_index_build_for_hist.  _begin returns
 * NULL (mfx_last_error says why) for any other database form, when the blocks would take more than a fifth of the free device
 * memory, or with MFX_DB_STAGE=0: the caller then takes the unstaged call.  The stage is released with mfx_db_stage_free (also
 * before it was used). */
typedef struct mfx_db_stage mfx_db_stage;
mfx_db_stage *mfx_db_stage_begin(const char *read_db_path, int device);
int           mfx_index_build_for_hist_staged(mfx_index *ix, const mfx_seq *seq, mfx_db_stage *stage, uint64_t minV, uint64_t maxV);
/* mfx_index_load_db of a staged database (side 0: -readmers, the -min/-max filter applies; 1: -seqmers): only the decode + insert kernels are
 * left to run.  What the path-only index of the variant modes loads both of its databases with (mfx_index_claim_paths). */
int           mfx_index_load_db_staged(mfx_index *ix, mfx_db_stage *stage, int side, uint64_t minV, uint64_t maxV);
void          mfx_db_stage_free(mfx_db_stage *stage);
/* Until this call the stage reads the file with a few threads only (the host is busy reading and encoding the sequence; MFX_DB_STAGE_THREADS,
 * default 8); call it when the sequence is uploaded. */
void          mfx_db_stage_boost(mfx_db_stage *stage);

/* merylExactLookup::value(kmer), merfin-globals.C:107-108, batched: for each
 * query returns the stored read and asm counts (0 when absent).  Queries are
 * looked up as given (no canonicalisation). */
int mfx_index_value(const mfx_index *ix, const uint64_t *kmers, uint64_t n,
                    uint32_t *readV, uint32_t *asmV);

typedef struct {
  int      k;
  int      canonical;      /* 1: every stored k-mer <= its reverse complement */
  uint64_t capacity;       /* slots                                           */
  uint64_t distinct;       /* occupied slots                                  */
  uint64_t bytes;          /* device bytes held by the table                  */
  int      seq_only;       /* 1: sequence-only index (mfx_index_create_for_seq) */
  int      compact;        /* 1: 8-byte slots, 16 per line (sequence-only, k <= 21) */
  uint64_t dropped;        /* adds a sequence-only index dropped (k-mers never claimed) */
} mfx_index_info;
int mfx_index_get_info(const mfx_index *ix, mfx_index_info *out);

/* copy out every stored (kmer, readV, asmV); arrays sized >= info.distinct.
 * Order unspecified.  (Test / -completeness support.) */
int mfx_index_export(const mfx_index *ix, uint64_t *kmers, uint32_t *readV, uint32_t *asmV,
                     uint64_t *n_out);

/* ------------------------------------------------------------------------ */
/* Sequences: replaces the loader callback loadSequence (merfin.C:30-53) +  */
/* merfinInput::{seq,kiter} (merfin-globals.H:64-65).  All contigs are      */
/* packed into one HBM buffer once; k-mer extraction happens on the device. */
/* ------------------------------------------------------------------------ */
mfx_seq *mfx_seq_upload(int device, const char *const *bases, const uint64_t *lens, uint32_t ncontigs);
/* contigs already resident on the device, each at d_bases[i] */
mfx_seq *mfx_seq_from_device(int device, const void *const *d_bases, const uint64_t *lens, uint32_t ncontigs,
                             void *stream);
void     mfx_seq_free(mfx_seq *s);
uint32_t mfx_seq_num_contigs(const mfx_seq *s);
uint64_t mfx_seq_num_bases(const mfx_seq *s);
/* work units (contig-aligned position tiles) -- the sharding granule for
 * multi-GPU runs: rank r of N evaluates tiles [T*r/N, T*(r+1)/N). */
uint64_t mfx_seq_num_tiles(const mfx_seq *s);

/* ------------------------------------------------------------------------ */
/* Evaluator: the K* parameters of merfinGlobal (merfin-globals.H:222-239)  */
/* bound to an index.                                                       */
/* ------------------------------------------------------------------------ */
typedef struct {
  double          peak;      /* -peak (merfin.C:89-90)                         */
  uint32_t        n_prob;    /* rows of the -prob table, 0 = none              */
  const uint32_t *probK;     /* copyKmerK (merfin-globals.H:226)               */
  const double   *probP;     /* copyKmerP (merfin-globals.H:227)               */
} mfx_kparams;

typedef struct mfx_eval mfx_eval;

/* nbins: dense K* bins kept per side on the device (0 = default 65536); bins
 * beyond are aggregated in a table of far bins (mfx_hist_take_overflow), so
 * results do not depend on it. */
mfx_eval *mfx_eval_create(const mfx_index *ix, const mfx_kparams *kp, uint32_t nbins);
void      mfx_eval_free(mfx_eval *ev);
uint32_t  mfx_eval_nbins(const mfx_eval *ev);
/* Test hook (no reference counterpart): -hist launches of this evaluator run the DEBUG instance of the kernel where one
 * exists (compact layout, k = 21) and count how its probe's queries ended: out8[0] not in the first mini-bucket (first
 * cooperative pass), [1] home line full (second cooperative pass), [2] saturated count (side table), [3] per-lane
 * whole-line scans.  mfx_eval_debug_counters reads and clears them.  Never the measured configuration. */
/* Diagnostic (no reference counterpart): random 128-byte lines per second this device's HBM delivers to independent
 * 16-byte loads over a table of table_bytes (allocated and released by the call): the roof of the index probe on this
 * box; bench.py reports the -hist kernel's line rate against it (roofline.gather_ceiling). */
int       mfx_diag_gather_rate(int device, uint64_t table_bytes, double *lines_per_s);
int       mfx_eval_debug_enable(mfx_eval *ev, int on);
int       mfx_eval_debug_counters(mfx_eval *ev, uint64_t *out8);

/* merfinGlobal::getK(kmvalu,kmvalu,...) + getKmetric on the host, bit-exact
 * with the device code (merfin-globals.C:66-98, merfin-globals.H:248-261). */
void mfx_getK(const mfx_kparams *kp, uint32_t readV, uint32_t asmV,
              double *readK, double *asmK, double *prob);
double mfx_getKmetric(double readK, double asmK);
/* histoQV (merfin-histogram.C:22-31) */
double mfx_histoQV(double kval, double ktot, int k);

/* ------------------------------------------------------------------------ */
/* -hist: replaces processHistogram + outputHistogram                       */
/* (merfin-histogram.C:35-92, 96-136).                                      */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint64_t  kasm;              /* histKasm     (merfin-globals.H:184)          */
  uint64_t  kmissing;          /* histKmissing (:185)                          */
  double    koverCpy;          /* histKoverCpy (:186)                          */
  uint32_t  undrMax, overMax;  /* histUndrMax / histOverMax (:188,191)         */
  uint64_t *undr, *over;       /* histUndr / histOver                          */
  uint32_t  ncontigs;
  uint64_t *contig_kasm;       /* per-contig s->kasm     (merfin-histogram.C:58)  */
  uint64_t *contig_kmissing;   /* per-contig s->kmissing (merfin-histogram.C:67)  */
} mfx_hist_result;

/* whole assembly on this object's device; result arrays are allocated by the
 * library, release with mfx_hist_result_free. */
int  mfx_hist_run(mfx_eval *ev, const mfx_seq *seq, mfx_hist_result *out);
void mfx_hist_result_free(mfx_hist_result *r);

/* The same with the upload inside: SURVEY 8(d)'s evaluate phase, "first tile H2D start -> final reduced histogram
 * on host".  `seq` = mfx_seq_create(device, lens, n) (layout + device buffers, no bases yet); bases[i] = host
 * buffer of contig i (pinned or pageable, ASCII).  The assembly crosses PCIe PACKED: host threads encode it into
 * 2-bit codes + one validity bit per base (0.375 B/base; mfx_pack_bases below -- the kmerIterator encoding of
 * merfin.C:45 moved in front of the bus) chunk by chunk, while the previous chunk is on the copy stream and the one
 * before is evaluated; the kernel reads its tiles from the packed planes.  Chunks grow from 8 MB to 128 MB of bases;
 * consecutive launches alternate between two streams so that one's first blocks fill the CUs the other's last blocks
 * leave.  3 Gb: 36.5 ms against 33.4 ms for the resident run (56 ms with one byte per base on the bus).  The result
 * is bit-identical to mfx_seq_upload + mfx_hist_run (koverCpy included: the per-tile values are summed once, at the
 * end).  Afterwards `seq` holds the whole assembly and can be used like an uploaded one (the kernels that read one
 * byte per base unpack the planes on first use).  k > 31 and MFX_STREAM_ASCII=1: one byte per base, 64 MB chunks,
 * pinned buffers DMA'd in place, pageable ones staged through pinned memory. */
void    *mfx_host_alloc(size_t bytes);
void     mfx_host_free(void *p);
mfx_seq *mfx_seq_create(int device, const uint64_t *lens, uint32_t ncontigs);
int      mfx_hist_run_streamed(mfx_eval *ev, mfx_seq *seq, const char *const *bases, mfx_hist_result *out);
/* The same over N devices driven by one process (the reference is one binary driving its workers, merfin.C:366-414; SURVEY 8(d) asks
 * for the evaluate phase "at 1/2/4/8 GPUs"): device d receives only the packed planes of ITS share of the tiles -- the contiguous
 * stretch [T d / N, T (d + 1) / N), cut at multiples of 1024 tiles, plus the k - 1 bases of halo behind it -- over its own copy
 * stream from its own encoder threads, and evaluates the chunks as they land; the images are added on the host in slot order.  The
 * result is bit-identical to mfx_hist_run_streamed / mfx_hist_run on one device, koverCpy included (the first-level sums of the
 * (tile, wave) values are the single launch's own, and the host adds them in the device's order).  Every slot needs its own
 * evaluator and its own sequence object (mfx_seq_create on the slot's device; evaluators on replicas of one index); afterwards a
 * slot's sequence object holds its part only and refuses whole-sequence calls (MFX_E_INVAL) until something is uploaded whole. */
int      mfx_hist_run_streamed_multi(mfx_eval *const *evs, mfx_seq *const *seqs, uint32_t ndev, const char *const *bases, mfx_hist_result *out);
/* One rank's share of it for the one-process-per-GPU launcher (bench.py, merfin_amd.mgpu): the tiles [tile_begin, tile_end) are
 * encoded, uploaded and evaluated, counts and koverCpy are ADDED to the caller's device image (MFX_HIST_WORDS(nbins, ncontigs) words
 * and one double, cleared by the caller), which the launcher then all-reduces (mfx_hist_allreduce).  Returns when the device is
 * done.  mfx_hist_stream_share: the bounds rank r of n takes (the same cut as the one-process run). */
int      mfx_hist_run_streamed_range(mfx_eval *ev, mfx_seq *seq, const char *const *bases, uint64_t tile_begin, uint64_t tile_end,
                                     uint64_t *d_counts, double *d_kover);
int      mfx_hist_stream_share(uint64_t ntiles, uint32_t rank, uint32_t nranks, uint64_t *tile_begin, uint64_t *tile_end);
/* How the bases of a streamed run cross the link is chosen per call (mfx_hist_run_streamed / _multi / _range) from two measured rates: what
 * the host threads the call has ENCODE, and what the device's LINK moves from pinned memory.  While the encoders are the faster the bases
 * cross packed (above); when they are not -- a rank of N processes has 1/N of the host's cores and a link of its own -- pinned sources cross as
 * plain bytes by DMA, no host thread touches them, and the planes are made on the device (mfx_pack_kernel); the evaluation is the same
 * launches over the same planes, the result the same bit for bit.  MFX_STREAM_TRANSPORT=pack | ascii forces either.
 * mfx_diag_stream_rates measures the two rates on demand (nothing cached): `threads` host threads encoding n bases of `src` at once, and --
 * src pinned -- up to 256 MB of it over `device`'s link; 0 for a rate that could not be measured. */
int      mfx_diag_stream_rates(int device, const char *src, uint64_t n, uint32_t threads, double *enc_gbs, double *link_gbs);
/* The host-side encoder of that transport (AVX-512 / AVX2 / scalar, chosen at run time; ~28 GB/s per core of an EPYC
 * 9575F): n bases -> ceil(n/32) words.  codes[w] holds bases 32w..32w+31, the first in the two HIGHEST bits,
 * code = (c >> 1) & 3 (A 0, C 1, T 2, G 3, either case); valid[w] holds one bit per base, the first in the highest
 * bit, set iff the base is one of ACGTacgt.  A last partial word is zero-filled. */
void     mfx_pack_bases(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid);

/* Several GPUs of one node driven by ONE process -- the reference is one binary driving all its workers
 * (merfin.C:366-414).  The index is replicated (mfx_index_replicate: peer copy over xGMI, instead of N builds),
 * so is the packed assembly; slot d evaluates the block-cyclic share d of ndev of the tiles (blocks of 256 tiles)
 * on its own device and stream, all at once; the ndev counts images (~1 MB) are added on the host in slot order,
 * the overflow lists of all evaluators folded in.  Integers are exact, koverCpy is a fixed-order sum: results are
 * bit-stable and equal to the single-device ones in every integer.  Two slots may name the same device (each
 * needs its own evaluator; they may share one index and one mfx_seq). */
mfx_index *mfx_index_replicate(const mfx_index *src, int device);
mfx_seq   *mfx_seq_replicate(const mfx_seq *src, int device);
/* n replicas at once, out[i] on devices[i]: a DOUBLING TREE over xGMI -- in every round each device that holds the data
 * feeds one that does not, all copies of a round in flight together (xGMI is point to point: 7 replicas take 3 rounds with
 * 1, 2, 4 source devices instead of 7 copies through device 0's links).  The assembly travels as its packed planes
 * (0.375 B/base; mfx_seq_pack builds them on the device); a replica holds the planes and unpacks on demand.  On failure
 * nothing is left allocated. */
int        mfx_index_replicate_many(const mfx_index *src, const int *devices, uint32_t n, mfx_index **out);
int        mfx_seq_replicate_many(const mfx_seq *src, const int *devices, uint32_t n, mfx_seq **out);
int        mfx_seq_pack(mfx_seq *seq);
int        mfx_hist_run_multi(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t ndev, mfx_hist_result *out);
/* PARTS of one assembly, one per slot: slot d evaluates ITS contigs (seqs[d]) on its own sequence-only index (their k-mers
 * claimed, assembly counts from the whole assembly by mfx_index_count_claimed, the read database update-only) -- no k-mer
 * ever leaves its device, whatever the size of the read database.  contig_ids[d][i] = number of slot d's contig i in the
 * whole assembly (ncontigs_total of them); bins and counters are added, the per-contig counters placed by those numbers,
 * koverCpy summed in slot order.  The reference's own way to spread a large run -- contigs over processes,
 * scripts/parallel1/merfin.sh:68-85 -- inside one process, with the lookup object cut to each part. */
int        mfx_hist_run_parts(mfx_eval *const *evs, const mfx_seq *const *seqs, const uint32_t *const *contig_ids, uint32_t ndev,
                              uint32_t ncontigs_total, mfx_hist_result *out);

/* One process per GPU (torchrun / mpirun style launchers): the collective of the path, on RCCL over xGMI.
 * Rank 0 makes an id (mfx_comm_unique_id), the launcher hands its MFX_COMM_ID_BYTES to every rank by any
 * out-of-band channel, every rank calls mfx_comm_create (collective).  mfx_hist_allreduce turns the ranks'
 * counts images (mfx_hist_launch / mfx_hist_launch_cyclic output) into the global one IN PLACE on every rank:
 * ncclAllReduce(sum, uint64) of the image; koverCpy = the ranks' values all-gathered and added in rank order
 * (bit-stable, whatever algorithm RCCL picks).  If the reduced image's novf word (index 2*nbins + 2) is
 * non-zero, K* bins >= nbins were seen: every rank then calls mfx_hist_allgather_overflow (collective) and folds
 * the records of all ranks with mfx_hist_result_add_overflow.  Asynchronous on `stream` except where noted. */
#define MFX_COMM_ID_BYTES 128
typedef struct mfx_comm mfx_comm;
int       mfx_comm_unique_id(void *id);
mfx_comm *mfx_comm_create(const void *id, int rank, int nranks, int device);
void      mfx_comm_free(mfx_comm *c);
int       mfx_comm_rank(const mfx_comm *c);
int       mfx_comm_size(const mfx_comm *c);
int       mfx_comm_barrier(mfx_comm *c, void *stream);      /* all ranks arrived and `stream` drained (synchronous) */
int       mfx_hist_allreduce(mfx_comm *c, uint64_t *d_counts, double *d_kover, uint32_t nbins, uint32_t ncontigs, void *stream);
/* synchronises `stream`; records[] (2 * cap words) receives the two-word records of all ranks in rank order; *n_out = records */
int       mfx_hist_allgather_overflow(mfx_comm *c, mfx_eval *ev, uint64_t *records, uint64_t cap, uint64_t *n_out, void *stream);
/* The exchange step of the sharded index (config 5: "allToAllv of the routed k-mers") for the
 * one-process-per-GPU form: group r of d_send (send_counts[r] elements of elem_bytes, groups back to back in rank order:
 * what mfx_route_tiles writes) goes to rank r; the groups arrive in source-rank order.  mfx_comm_exchange_counts tells
 * every rank what it will receive (all-gather of the count rows; synchronises `stream`), mfx_comm_alltoallv moves one
 * array (one RCCL group of point-to-point sends / receives over xGMI; asynchronous on `stream`). */
int       mfx_comm_exchange_counts(mfx_comm *c, const uint64_t *send_counts, uint64_t *recv_counts, void *stream);
int       mfx_comm_alltoallv(mfx_comm *c, const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts,
                             uint32_t elem_bytes, void *stream);
/* fold n overflow records (two words each: mfx_hist_take_overflow / mfx_hist_allgather_overflow format) into a result */
int       mfx_hist_result_add_overflow(mfx_hist_result *r, const uint64_t *records, uint64_t n);

/* Device-side accumulate for sharded runs.  d_counts: uint64[MFX_HIST_WORDS]
 * caller-owned device memory, added into (zero it first), layout
 *   [0,nbins) undr | [nbins,2nbins) over | kasm | kmissing | novf |
 *   contig_kasm[ncontigs] | contig_kmissing[ncontigs]
 * d_kover: double[1], receives (+=) this launch's koverCpy.  Evaluates tiles
 * [tile_begin, tile_end).  Asynchronous on `stream`.  All-reduce d_counts and
 * d_kover over ranks to obtain the global result (the only collective). */
#define MFX_HIST_WORDS(nbins, ncontigs) (2ull * (nbins) + 3ull + 2ull * (ncontigs))
int mfx_hist_launch(mfx_eval *ev, const mfx_seq *seq, uint64_t tile_begin, uint64_t tile_end,
                    uint64_t *d_counts, double *d_kover, void *stream);
/* The same for the block-cyclic share of `rank` out of `nranks`: the tiles are cut into blocks of `block_tiles`
 * (a power of two) and rank r evaluates blocks r, r + nranks, ...  Every rank then sees every region of the assembly,
 * so a positional skew of the per-k-mer cost (repeats, k-mers that sit in overflow lines of the table) cannot make
 * one rank the straggler, which a contiguous split allows (profiles/r01_shard_timing.txt). */
int mfx_hist_launch_cyclic(mfx_eval *ev, const mfx_seq *seq, uint32_t rank, uint32_t nranks, uint32_t block_tiles,
                           uint64_t *d_counts, double *d_kover, void *stream);
/* turn an (all-reduced, host-resident) d_counts/d_kover image into a result
 * (pure host code: usable without a device, e.g. on the reducing rank) */
int mfx_hist_result_from_counts(uint32_t nbins, const uint64_t *h_counts, double kover,
                                uint32_t ncontigs, mfx_hist_result *out);
/* K* bins >= nbins seen by launches on this evaluator since the last call.  The reference's histogram arrays grow without a
 * bound (increaseArray, merfin-histogram.C:74,87); here the far bins are AGGREGATED on the device, {bin -> occurrences}, so any
 * number of k-mers may fall there (an assembly's satellite array the reads under-represent puts millions of positions into a
 * few far bins); only the number of DISTINCT far bins per evaluator is bounded (2^20; beyond: MFX_E_OVERFLOW, create the
 * evaluator with a larger nbins).  Copies up to `cap` records of TWO words each -- {bit 63 = 1 for `over` | bin index,
 * occurrences} --, sorted by the first word; *n_out = records; the evaluator's table is emptied. */
int mfx_hist_take_overflow(mfx_eval *ev, uint64_t *records, uint64_t cap, uint64_t *n_out);

/* reportHistogram (merfin-histogram.C:140-176): the histogram file text and
 * the stderr summary, byte-identical formatting.  Either path may be NULL;
 * a path ending in .gz/.bz2/.xz is piped through the matching compressor as
 * compressedFileWriter does. */
int mfx_hist_report(const mfx_hist_result *r, int k, const char *hist_path, const char *summary_path);

/* ------------------------------------------------------------------------ */
/* Sharded index (BASELINE config 5: a read DB too large for one GPU).  The  */
/* reference has no counterpart (it needs the whole DB in one host's RAM,    */
/* merfin-globals.C:148-153).  Every rank keeps the k-mers it OWNS (owner =  */
/* f(minimizer)); -hist routes each k-mer of the rank's sequence tiles to its */
/* owner, which probes, computes K* and bins; one all-reduce of the counts   */
/* image finishes.  Requires a canonical DB and odd k.                       */
/* ------------------------------------------------------------------------ */
/* call before any add/load/count: afterwards they silently skip foreign k-mers */
int mfx_index_set_shard(mfx_index *ix, uint32_t rank, uint32_t nranks);

typedef struct mfx_router mfx_router;
mfx_router *mfx_router_create(const mfx_index *ix, uint32_t nranks, uint32_t max_tiles);
void        mfx_router_free(mfx_router *r);
/* Source side: canonical k-mers of tiles [tile_begin, tile_end) grouped by owner
 * rank (sequence order kept inside a group), written to the caller's device
 * buffers d_keys_out / d_contigs_out (capacity (tile_end-tile_begin)*4096);
 * h_dest_counts[nranks] receives the group sizes.  The valid-k-mer counts
 * (kasm, per-contig kasm; merfin-histogram.C:58) are added to d_counts.
 * Synchronises `stream` once (the group sizes are needed on the host). */
int mfx_route_tiles(mfx_router *r, const mfx_seq *seq, uint64_t tile_begin, uint64_t tile_end, uint32_t nbins,
                    uint64_t *d_counts, uint64_t *d_keys_out, uint32_t *d_contigs_out, uint64_t *h_dest_counts,
                    void *stream);
/* Owner side: probe + K* + bin n received k-mers; adds bins, kmissing (global
 * and per contig) and koverCpy -- NOT kasm, which the source counted. */
int mfx_hist_keys_launch(mfx_eval *ev, const uint64_t *d_keys, const uint32_t *d_contigs, uint64_t n,
                         uint32_t ncontigs, uint64_t *d_counts, double *d_kover, void *stream);
/* The whole sharded -hist driven by ONE process: slot d = {evaluator on shard d of ndev, router on that shard,
 * the packed assembly on the shard's device}.  Rounds of route -> peer copies of the groups to their owners (xGMI)
 * -> mfx_hist_keys_launch at the owner -> counts images added on the host, overflow lists folded.  Slots may share a
 * device.  (The one-process-per-GPU form of the same loop is merfin_amd/distributed.py::sharded_hist.) */
int mfx_hist_run_sharded(mfx_eval *const *evs, mfx_router *const *routers, const mfx_seq *const *seqs, uint32_t ndev,
                         mfx_hist_result *out);

/* ------------------------------------------------------------------------ */
/* -dump: replaces processDump + outputDump (merfin-dump.C:20-68, 72-104).  */
/* ------------------------------------------------------------------------ */
/* Raw per-position values: for k-mer start positions [pos_begin,pos_end) of
 * contig `contig`, readV[i]/asmV[i] = summed read / asm counts of the k-mer
 * starting there (0,0 where no valid k-mer starts).  Host output arrays. */
int mfx_dump_values(mfx_eval *ev, const mfx_seq *seq, uint32_t contig, uint64_t pos_begin, uint64_t pos_end,
                    uint32_t *readV, uint32_t *asmV, uint64_t *kasm, uint64_t *kmissing);
/* The complete -dump of one contig appended to `path` in the reference's text
 * format "%s\t%lu\t%.2f\t%.2f\t%.2f\n" (merfin-dump.C:88-93). */
int mfx_dump_contig(mfx_eval *ev, const mfx_seq *seq, uint32_t contig, const char *name,
                    const char *path, int append, uint64_t *kasm, uint64_t *kmissing);
/* Several host threads running library calls side by side (the slots of `merfin -devices`): a thread that declares
 * itself one of `nsharers` gets 1/nsharers of the host threads in the calls it makes from then on (text formatting,
 * VCF parsing, path enumeration, staging copies) instead of all of them -- N callers each spawning a full set only
 * fight for the cores.  Thread-local; 1 = the default. */
void mfx_host_threads_share(unsigned nsharers);

/* -dump over an index SHARDED across nslots evaluators (slot d = shard d of nslots, mfx_index_set_shard; seqs[d] =
 * the same sequences resident on slot d's device; slots may share a device).  Every k-mer has one owner and the
 * other shards answer 0, so each slot looks its copy of the range up in its own shard and the value arrays are added
 * on slot 0's device (8 B per position and slot over xGMI); kmissing is counted from the sums.  Same results as
 * mfx_dump_values / mfx_dump_contig on the whole index (merfin-dump.C:44-67 sees one lookup object either way). */
int mfx_dump_values_sharded(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t nslots, uint32_t contig,
                            uint64_t pos_begin, uint64_t pos_end, uint32_t *readV, uint32_t *asmV,
                            uint64_t *kasm, uint64_t *kmissing);
int mfx_dump_contig_sharded(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t nslots, uint32_t contig,
                            const char *name, const char *path, int append, uint64_t *kasm, uint64_t *kmissing);

/* ------------------------------------------------------------------------ */
/* -filter / -polish / -better / -strict / -loose: replaces processVariants */
/* + outputVariants (merfin-variants.C:131-345), vcfFile (vcf.C) and varMer  */
/* (varMer.C).  The host enumerates the allele-combination paths of many     */
/* clusters, the GPU scores every k-mer of every path in one launch of the   */
/* -dump lookup kernel, the host applies the mode's selector.                */
/* ------------------------------------------------------------------------ */
#define MFX_VAR_FILTER 4      /* OP_FILTER .. OP_LOOSE, merfin-globals.H:34-38 */
#define MFX_VAR_POLISH 5
#define MFX_VAR_BETTER 6
#define MFX_VAR_STRICT 7
#define MFX_VAR_LOOSE  8
typedef struct {
  int         mode;         /* MFX_VAR_*                                       */
  uint32_t    comb;         /* -comb (0 = default 15), merfin.C:144-145        */
  int         nosplit;      /* -nosplit                                        */
  const char *debug_path;   /* -debug log (merfin-variants.C:240-276) or NULL; */
                            /* a name ending in .gz is gzip-compressed         */
} mfx_variant_opts;
/* names/bases/lens: the contigs of -sequence (ident() = first header token is
 * matched against VCF CHROM, merfin-variants.C:141).  Writes headers + the
 * selected records to out_path (contigs in input order).  log_path receives
 * the PANIC / WARNING / progress lines (NULL = stderr). */
int mfx_variants_run(mfx_eval *ev, const char *vcf_path, const char *const *names, const char *const *bases,
                     const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                     const char *out_path, const char *log_path, uint64_t *n_clusters);
/* The VCF of a run read and parsed ahead of it (vcfFile::loadFile, vcf.C:93-149): host work only -- no device, no index --
 * so a caller runs it on a thread of its own under the index build (merfin opens the VCF after load_Kmers,
 * merfin-globals.C:201-219).  A handle serves ONE mfx_variants_run_vcf (clustering rearranges it); NULL on error.
 * mfx_variants_run_vcf = mfx_variants_run on the loaded records: same outputs, byte for byte. */
typedef struct mfx_vcf mfx_vcf;
mfx_vcf *mfx_vcf_load(const char *vcf_path);
void     mfx_vcf_free(mfx_vcf *vcf);
int mfx_variants_run_vcf(mfx_eval *ev, mfx_vcf *vcf, const char *const *names, const char *const *bases,
                         const uint64_t *lens, uint32_t ncontigs, const mfx_variant_opts *opts,
                         const char *out_path, const char *log_path, uint64_t *n_clusters);
/* Stage A of the variant modes ahead of the run, on a loaded VCF: the clusters merged for k (vcfFile::mergeChrPosGT, vcf.C:156-246) and
 * their allele combinations enumerated and packed batch by batch (merfin-variants.C:22-126, varMer.C:39) -- host work that needs the VCF
 * and the sequences but neither the index nor a device (merfin does it after load_Kmers, merfin-variants.C:131-230; here a caller runs it
 * under its index build).  The run -- mfx_variants_run_vcf on the same handle with the same sequences, k (the index's), opts->comb and
 * opts->nosplit: anything else is refused -- then starts at the lookups; its outputs are the unprepared run's byte for byte.  The
 * sequences must stay where they are until the run is over.  Once per handle. */
int mfx_vcf_prepare(mfx_vcf *vcf, int k, const char *const *names, const char *const *bases, const uint64_t *lens,
                    uint32_t ncontigs, const mfx_variant_opts *opts);
/* The PATH-ONLY index of the variant modes (round 6).  varMer::score (varMer.C:76-84) is the only place the variant modes ask the lookup
 * tables anything, and what it asks for are the k-mers of the enumerated paths; the reference nevertheless loads both databases whole
 * (load_Kmers, merfin-globals.C:114-163).  A prepared call set knows every path before a database is opened:
 *   mfx_vcf_prepare(vcf, k, ...);  mfx_vcf_path_bound(vcf, &n);            -- n: k-mer positions of all path text >= their distinct k-mers
 *   ix = mfx_index_create_for_seq(k, n + 1024, max_gb, device);
 *   mfx_index_claim_paths(ix, vcf, NULL);                                   -- the paths made on the device as the run makes them, their k-mers claimed
 *   mfx_index_load_db(ix, seqmers, 1, ...)  (or mfx_index_count_claimed(ix, assembly, 0));   mfx_index_load_db(ix, readmers, 0, minV, maxV);
 *   ev = mfx_eval_create(ix, ...);  mfx_variants_run_vcf(ev, vcf, ...);     -- same records, byte for byte, as on the full index
 * Both databases only UPDATE the claimed k-mers (a k-mer of no path never gets a slot).  The index is bound to the handle: another call set,
 * an unprepared run, -hist / -dump of a sequence are refused on it (MFX_E_INVAL).  k <= 31; a non-canonical database makes the load
 * return MFX_E_NONCANON as for every sequence-only index (build the full one). */
int mfx_vcf_path_bound(const mfx_vcf *vcf, uint64_t *positions);
int mfx_index_claim_paths(mfx_index *ix, mfx_vcf *vcf, uint64_t *n_positions);
/* The three steps above as ONE pass (what `merfin` does): the table is made as soon as the clusters are merged and every batch's k-mers are
 * claimed on the device while the host prepares the next batch.  *out: the claimed index, or NULL (with MFX_OK: the handle is prepared and runs
 * on a full index) when this call set cannot have one -- no memory for the table, a cluster the device cannot enumerate; mfx_last_error says why.
 * load_factor 0: the library's choice. */
int mfx_vcf_prepare_path_index(mfx_vcf *vcf, int k, const char *const *names, const char *const *bases, const uint64_t *lens, uint32_t ncontigs,
                               const mfx_variant_opts *opts, double max_gb, int device, double load_factor, mfx_index **out);
/* The same over an index sharded across nslots evaluators (slot d = shard d of nslots): every batch of path text is
 * scored by mfx_dump_values_sharded; clustering, enumeration, selectors and output are the code above. */
int mfx_variants_run_sharded(mfx_eval *const *evs, uint32_t nslots, const char *vcf_path, const char *const *names,
                             const char *const *bases, const uint64_t *lens, uint32_t ncontigs,
                             const mfx_variant_opts *opts, const char *out_path, const char *log_path,
                             uint64_t *n_clusters);

/* ------------------------------------------------------------------------ */
/* -completeness: replaces computeCompleteness (merfin-completeness.C:48-144)*/
/* as one streaming pass over the joint table.                              */
/* ------------------------------------------------------------------------ */
int mfx_completeness(mfx_eval *ev, double *total, double *undrcpy);
/* the 64 per-piece sums the reference prints (piece = top 6 bits of the k-mer = meryl file number,
 * merfin-completeness.C:56-66,119-120); arrays of 64 doubles each */
int mfx_completeness_pieces(mfx_eval *ev, double *total64, double *undrcpy64);

#ifdef __cplusplus
}
#endif
#endif /* MERFIN_AMD_H */
