// mfx_internal.h -- shared declarations of the merfin_amd library (host side).
#pragma once
#include "mfx_traverse.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/merfin_amd.h"
#include "mfx_kstar.h"

// ---- error plumbing -------------------------------------------------------
void mfx_set_error(const char *fmt, ...);
int  mfx_fail(int code, const char *fmt, ...);
#define MFX_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define MFX_HIP_NULL(call)                                                                    \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return nullptr;                                                                         \
    }                                                                                         \
  } while (0)

// A fill of device memory that is COMPLETE when the call returns.  hipMemset is not: it only queues the fill on the null stream
// (tools/native/memset_probe.hip on the MI355X: it returns after 10 us while a 0.2 s kernel holds that stream, a copy issued next on
// a non-blocking stream lands first -- and is then wiped by the late fill).  Every buffer here is filled on one stream and next
// touched on another (the non-blocking copy / compute streams), so every fill is waited for.  That was how eight variant slots on
// one device lost the records of a batch now and then: slot A's path text, uploaded on its copy stream, was zeroed by its own
// plane fill, which had queued behind slot B's kernels on the null stream.
static inline hipError_t mfx_memset_now(void *p, int v, size_t n) {
#ifdef MFX_V_QUEUED_FILLS                                     // A/B build only: the behaviour before (tests/test_gpu_null_stream.py must FAIL on it)
  return hipMemset(p, v, n);
#else
  const hipError_t e = hipMemsetAsync(p, v, n, nullptr);
  return e == hipSuccess ? hipStreamSynchronize(nullptr) : e;
#endif
}

// flat-binary database -> table, read with parallel pread into the index's staging lanes (mfx_api.cpp)
// vals_off == 0: the records at keys_off are PACKED (MFX_PACKED_VBITS), there is no counts array
int  mfx_index_add_from_file(struct mfx_index *const *ixs, uint32_t nix, int fd, const char *path, uint64_t keys_off, uint64_t vals_off,
                             uint64_t n, int side, uint64_t minV, uint64_t maxV);
// the delta-coded blocks of a sorted flat database (mfx_db.cpp FLAT_DELTA); dir: (nblocks + 1) x 2 words
int  mfx_index_add_delta_file(struct mfx_index *const *ixs, uint32_t nix, int fd, const char *path, const uint64_t *dir, uint64_t nblocks,
                              uint64_t n, int side, uint64_t minV, uint64_t maxV, int placed = 0);     // placed: the records are placement numbers (mfx_place.h)
// host arrays into several tables at once (one staging, one H2D per table; sharded tables keep what they own)
int  mfx_index_add_multi(struct mfx_index *const *ixs, uint32_t nix, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side,
                         uint64_t minV, uint64_t maxV);
void mfx_index_ingest_release(struct mfx_index *ix);
// a delta-coded flat database opened for the staged load (mfx_db.cpp; mfx_api.cpp: mfx_db_stage)
struct mfx_flat_delta_info { int k = 0, placed = 0; uint64_t n = 0, n_escape = 0, nblocks = 0, escapes_off = 0, fsize = 0; };
void mfx_place_keys_host(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, uint8_t *sbits_out = nullptr);      // mfx_db.cpp: P (mfx_place.h) of canonical k-mers, host threads; k = 31: P >> 1 and the strand bits
int  mfx_flat_delta_open(const char *path, int *fd_out, mfx_flat_delta_info *info, std::vector<uint64_t> &dir);

// host threads the library may use: min(hardware, cgroup CPU quota, 64), or MFX_HOST_THREADS
unsigned mfx_host_threads();

// ---- geometry --------------------------------------------------------------
// A tile = MFX_TILE consecutive k-mer start positions of one contig.  Contigs
// start at 128-byte aligned offsets of the packed HBM buffer and are followed
// by at least one invalid byte, so no k-mer can span two contigs.
constexpr uint32_t MFX_TILE       = 4096;
constexpr uint32_t MFX_BLOCK      = 256;
constexpr uint32_t MFX_ALIGN      = 128;
constexpr uint32_t MFX_SLOTS_LINE = 8;          // 8 x 16-byte slots = one 128-byte HBM line
constexpr uint32_t MFX_NB_LDS     = 1024;       // K* bins per side privatised in LDS
#ifndef MFX_V_MAXP_LDS
#define MFX_V_MAXP_LDS 1024
#endif
constexpr uint32_t MFX_MAXP_LDS   = MFX_V_MAXP_LDS;   // read counts whose (readK, prob) is tabulated in LDS (A/B: tools/ab_build.sh -DMFX_V_MAXP_LDS=256)
constexpr uint32_t MFX_KLUT       = 32;         // (readK, asmK) pairs below this use tabulated bin index / over-copy term
// placement functions of the table (mfx_kernels.hip: mfx_minimizer, mfx_mz_line, mfx_home); index images
// written under another version are refused by mfx_index_load
constexpr uint32_t MFX_LAYOUT_VERSION = 9u;     // 9: the compact layout's line from the bijective mix of the minimizer, its first mini-bucket from the window (mfx_place.h)
constexpr uint32_t MFX_SPLIT_MAX_RANKS = 16;  // owners the sort-free router handles (a node has 8 GPUs); more: radix sort
constexpr int      MFX_MZ_W_DEFAULT = 3;        // minimizer windows of the default placement (MFX_MZ_W overrides)
constexpr uint32_t MFX_OVF_SLOTS  = 1u << 20;   // DISTINCT K* bins beyond the dense image an evaluator can hold (occurrences are unbounded: mfx_device.h, mfx_ovf_add)
constexpr uint32_t MFX_OVF_PROBES = 256;        // ... linear probes before an occurrence counts as lost (MFX_E_OVERFLOW)
constexpr size_t   MFX_OVF_WORDS  = 2 + 2 * (size_t)MFX_OVF_SLOTS;
constexpr uint32_t MFX_META_WORDS = 8;          // mfx_index::d_meta

struct mfx_slot {               // 16 bytes: one dwordx4 load per probe
  uint64_t key;                 // 2k-bit k-mer, ~0 = empty
  uint32_t readV;               // raw read-DB count (the -min/-max filter is applied at query time)
  uint32_t asmV;                // assembly count
};
constexpr uint64_t MFX_EMPTY = ~0ull;

// Packed k-mer records of the flat database form and of its transport (k <= 21: a k-mer has at most 42 bits): one uint64 =
// {k-mer << 22 | count}; a count field of all ones is an escape -- the record's real count (>= 2^22 - 1) comes in a short
// side list.  8 bytes per k-mer on disk, through the staging lanes and over PCIe instead of 12; unpacked by the lane that
// inserts it (mfx_table_add_kernel / mfx_table_update_kernel with values == nullptr).
constexpr int      MFX_PACKED_VBITS = 22;
constexpr uint32_t MFX_PACKED_VMASK = (1u << MFX_PACKED_VBITS) - 1u;
constexpr int      MFX_MAX_K_PACKED = 21;
constexpr uint32_t MFX_DELTA_BLOCK = 4096;     // k-mers per delta-coded block of a sorted flat database (mfx_db.cpp FLAT_DELTA)
constexpr int      MFX_DELTA_MAX_VBITS = 22;   // widest count field of a block; larger counts are escapes

// 32 <= k <= 64: k-mers of up to 128 bits (mfx_wide.hip).  Four 32-byte slots per 128-byte line.
struct mfx_wslot {
  uint64_t lo, hi;              // the 2k-bit k-mer
  uint32_t readV, asmV;
  uint64_t state;               // 0 empty, 1 claimed (key being written), 2 ready
};
constexpr uint32_t MFX_WSLOTS_LINE = 4;
constexpr int      MFX_MAX_K_NARROW = 31, MFX_MAX_K = 64;

// Sequence-only index, compact layout (k <= 21, see mfx_kernels.hip): 8-byte slots, 16 per 128-byte line --
// {key: 42 bits | readV: 11 | asmV: 11}; a count field of MFX_CSAT means "saturated: the exact count of that side is in
// the side table" (standard 16-byte slots, plain hashing, right behind the main lines in the same allocation).
constexpr uint32_t MFX_CSLOTS_LINE = 16;
constexpr uint32_t MFX_CSAT = 2047u;
constexpr int      MFX_MAX_K_COMPACT = 31;        // compact layout: k <= MFX_MAX_K_DIRECT holds the k-mer in the slot, above that its quotient (mfx_q_place)
constexpr int      MFX_MAX_K_DIRECT = 21;
constexpr int      MFX_MZ_W_COMPACT = 4;          // minimizer windows of the compact layout (MFX_MZ_W overrides)

struct mfx_table_view {
  mfx_slot *slots;
  uint64_t  nlines;             // 128-byte lines; slots = 8 * nlines (16 * nlines in the compact layout)
  uint32_t  minV, maxV;         // read-count filter (merfin.C:199-200), clamped to uint32
  int       k;
  int       mz_w;               // minimizer windows (0 = plain k-mer hashing; else m = k - mz_w + 1)
  int       mz_t;               // > 0: the window is SAMPLED by the k-mer's smallest mz_t-mer (mod-minimizer, mfx_kernels.hip); 0: the smallest m-mer
  uint32_t  shard_rank, shard_n;  // sharded index: this table keeps only the k-mers owned by shard_rank of shard_n
  int       wide;               // k > 31: slots are mfx_wslot (mfx_wide.hip kernels)
  int       seq_only;           // the key set is the k-mers claimed from a sequence: adds update, they never claim
  int       compact;            // 8-byte slots (above); implies seq_only
  int       quot;               // compact, k > MFX_MAX_K_DIRECT: the slot's key field holds the k-mer's QUOTIENT (mfx_kernels.hip: mfx_q_place)
  int       qshift;             // ... floor(log2(nlines))
  mfx_slot *side;               // compact: the side table of saturated counts
  uint64_t  side_nlines;
};

struct mfx_ingest;              // pinned staging lanes of the host -> table pipeline (mfx_api.cpp)

struct mfx_index {
  int       device = 0;
  mfx_ingest *ingest = nullptr; // allocated by the first host-side load, reused by the following ones
  int       k = 0;
  uint64_t  capacity_kmers = 0;
  uint64_t  nlines = 0;
  mfx_slot *d_slots = nullptr;
  uint64_t *d_meta = nullptr;   // [0] distinct  [1] non-canonical inserts  [2] probe-limit failures  [3] adds dropped by a sequence-only index
                                // [4] records wider than 2k bits (a damaged database; never inserted).  MFX_META_WORDS allocated; images and replicas carry the first 4
  uint64_t  minV = 0, maxV = ~0ull;
  bool      filter_set = false;
  int       mz_w = 0;
  int       mz_t = 0;
  uint32_t  shard_rank = 0, shard_n = 1;
  uint64_t  version = 0;        // bumped by every insert batch; lets evaluators cache index-derived facts
  uint64_t  fingerprint = 0;    // caller-supplied digest of the inputs (travels with the index image)
  // Sequence-only index (mfx_index_create_for_seq): holds exactly the k-mers claimed from a sequence
  // (mfx_index_claim_seq / mfx_index_count_asm); later adds and loads only UPDATE the counts of those k-mers, a k-mer
  // that was not claimed is dropped.  What -hist and -dump ask the lookup tables is the k-mers of -sequence and nothing
  // else (merfin-histogram.C:54-64, merfin-dump.C:44-61), so their answers are those of the full tables.
  bool      seq_only = false;
  bool      compact = false;    // seq_only, k <= 31: 8-byte slots, 16 per line (mfx_table_view)
  bool      quot = false;       // compact, k > 21: quotient form of the key field
  bool      frozen = false;     // an add / load happened: no more claims
  uint64_t  side_nlines = 0;    // compact: lines of the side table, which follows the nlines main lines in d_slots
  uint64_t  paths_token = 0;    // seq_only: the k-mers are those of a prepared VCF's PATHS (mfx_index_claim_paths), not a sequence's: the token of that call set
  uint32_t  seq_digest = 0;     // seq_only: content digest of the sequence the k-mers were claimed from (0: not recorded);
                                // evaluating another sequence on it is refused (mfx_seq_digest32, mfx_api.cpp)
  uint64_t  total_lines() const { return nlines + side_nlines; }
  bool      wide() const { return k > MFX_MAX_K_NARROW; }
  uint32_t  slots_per_line() const { return wide() ? MFX_WSLOTS_LINE : compact ? MFX_CSLOTS_LINE : MFX_SLOTS_LINE; }
  uint32_t  key_words() const { return wide() ? 2u : 1u; }      // uint64 words per k-mer at the C ABI
  mfx_table_view view() const;
};

struct mfx_seq {
  int       device = 0;
  uint32_t  ncontigs = 0;
  uint64_t  total_bases = 0;
  uint64_t  ntiles = 0;
  uint64_t  buf_bytes = 0;
  uint8_t  *d_bases = nullptr;       // padded contigs, one byte per base
  // packed planes of the same buffer (word i = bytes [32 i, 32 i + 32)), allocated and filled by a packed upload
  // (mfx_hist_run_streamed); bases_stale: the planes are newer than d_bases (mfx_seq_ensure_ascii unpacks on demand)
  uint64_t *d_codes = nullptr;
  uint32_t *d_valid = nullptr;
  bool      bases_stale = false;
  bool      planes_ok = false;       // the planes hold the current sequence (a packed upload, mfx_seq_pack, a replica)
  uint64_t *d_contig_off = nullptr;  // [ncontigs]   byte offset of each contig
  uint64_t *d_contig_len = nullptr;  // [ncontigs]
  uint64_t *d_tile_start = nullptr;  // [ncontigs+1] first tile of each contig
  uint32_t *d_tile_contig = nullptr; // [ntiles] contig of each tile
  std::vector<uint64_t> off, len, tile_start;
  bool      partial = false;         // a streamed PART was the last thing put in: only the tiles [part_lo, part_hi) are there (mfx_hist_run_streamed_multi / _range)
  uint64_t  part_lo = 0, part_hi = 0;
  mutable std::mutex lazy_mu;        // what a sequence makes on first use (bytes per base, planes, digest) is made under it; per object: slots of other sequences / devices do not queue here
  mutable uint32_t digest = 0;       // content digest (mfx_seq_digest32), computed on first use; 0: not computed / the content changed
};

// content digest of a sequence (never 0): contig lengths + the codes and validity of every base, whatever form it is held in
int mfx_seq_digest32(const mfx_seq *s, uint32_t *out);
// refuses (MFX_E_INVAL) the evaluation of a sequence other than the one a sequence-only index was claimed from
int mfx_check_seq_of_index(const mfx_index *ix, const mfx_seq *s, const char *who);
extern thread_local bool t_mfx_path_lookup;       // the calling thread looks the variant modes' own path text up: a path-only index answers (mfx_api.cpp)

// The paths of a batch of variant clusters inside their packed text, for the device-side varMer::score (mfx_api.cpp:
// mfx_score_paths; kernel mfx_var_score_kernel): host arrays
struct mfx_path_table {
  uint64_t        npaths = 0, nvals = 0;
  const uint64_t *off = nullptr;      // [npaths] first base of the path in the text
  const uint32_t *len = nullptr;      // [npaths] its bases
  const uint32_t *nv = nullptr;       // [npaths] variants of its cluster
  const uint64_t *voff = nullptr;     // [npaths] first of its nv entries below
  const uint64_t *cfirst = nullptr;   // [npaths] number of the first path of its cluster (`prob` carries from path to path inside a cluster)
  const int32_t  *gt = nullptr;       // [nvals] allele of each variant on the path
  const uint32_t *vidx = nullptr, *vlen = nullptr;   // [nvals] offset / length snapshots of the variants on the path
};
// looks every k-mer of `text` up (as mfx_dump_values) and scores every path on the device: numM[p] (missing k-mers, lead-in
// included) and, need_dk, totdk[p] (sum of the delta-K terms in position order) -- the values varMer::score computes
int mfx_score_paths(mfx_eval *ev, const char *text, uint64_t len, const mfx_path_table *pt, int need_dk, uint32_t *numM, double *totdk);

// ... with part of the batch's clusters ENUMERATED ON THE DEVICE (mfx_traverse.h; merfin-variants.C:22-126): `text` / `pt` hold the paths the
// host enumerated (clusters beyond the device's limits; may be empty), `tb` the other clusters as tables -- window bases, variants, alleles --
// and the room reserved for their paths: text behind the host's (cl[].text0 absolute in the batch's text, >= len), path slots and rows
// behind the host's (numbered from 0 in cl[].path0 / row0).  numM / totdk: [pt->npaths + tb->path_cap], the device part behind the host's;
// a slot beyond a cluster's np is an empty path.  Comes back with every cluster's status: a caller that finds one != MFX_TRV_OK discards the
// device part's results and enumerates on the host.
struct mfx_trv_batch {
  uint64_t ncl = 0, nvar = 0, nal = 0, win_bytes = 0, al_bytes = 0;
  const mfx_trv_cluster *cl = nullptr;
  const mfx_trv_variant *var = nullptr;
  const mfx_trv_allele *al = nullptr;
  const char *win_text = nullptr, *al_text = nullptr;
  uint64_t text_end = 0;              // the batch's text: [0, len) the host's, up to text_end the device's room
  uint64_t path_cap = 0, row_cap = 0;
  uint32_t *np = nullptr, *status = nullptr;     // [ncl] out
  uint32_t *p_len = nullptr;                     // [path_cap] out: the length of every path slot
  int32_t *gt = nullptr;                         // [row_cap] out: the genotype rows
};
int mfx_score_paths_trv(mfx_eval *ev, const char *text, uint64_t len, const mfx_path_table *pt, const mfx_trv_batch *tb, int need_dk, uint32_t *numM, double *totdk);

// The PATH-ONLY index of the variant modes (mfx_index_claim_paths, mfx_variants.cpp): a batch's text is put together on the device exactly as
// mfx_score_paths_trv does -- the host's paths copied in, the other clusters enumerated by the traverse kernel -- and the claim kernel of
// mfx_index_claim_seq runs over it as over ONE contig (every path is followed by '\n': no k-mer spans two).  `scratch`: the caller's device
// buffer, grown here, released with mfx_claim_paths_release; *bad: clusters the device could not enumerate (tb->status comes back).
int  mfx_claim_paths_batch(mfx_index *ix, uint8_t **scratch, uint64_t *scratch_bytes, const char *text, uint64_t len, const mfx_trv_batch *tb, uint64_t *bad);
int  mfx_claim_paths_finish(mfx_index *ix, uint64_t token);          // waits, checks the table, binds the index to the prepared call set `token`
void mfx_claim_paths_release(int device, uint8_t *scratch);

int mfx_seq_partial_error(const mfx_seq *s, const char *who);     // MFX_E_INVAL: the sequence object holds a part only
int mfx_seq_ensure_ascii(const mfx_seq *s);      // unpacks the planes into d_bases if a packed upload left them newer (mfx_api.cpp)

struct mfx_eval {
  const mfx_index *ix = nullptr;
  int       device = 0;
  double    peak = 0;
  uint32_t  n_prob = 0;
  std::vector<uint32_t> probK;
  std::vector<double>   probP;
  uint32_t *d_probK = nullptr;
  double   *d_probP = nullptr;
  uint64_t *d_underq = nullptr;      // [MFX_MAXP_LDS * MFX_KLUT] the over-copy terms of the tabulated (read count, asmV) pairs as integers (mfx_kfix; mfx_hist_eval_fx)
  uint32_t  nbins = 65536;
  int       grid = 0;
  uint64_t  canon_version = ~0ull;   // index version the cached `canon` flag belongs to
  int       canon = 0;
  double   *d_partials = nullptr;    // [2*grid] per-block koverCpy partial sums (key-driven kernel)
  uint64_t *d_tile_ctr = nullptr;    // [2] dynamic tile scheduler counters of mfx_hist_kernel (0 between launches); the second serves launches that overlap the first's
  double   *d_tile_partials = nullptr; // per-(tile,wave) koverCpy of the last launch + the chunk sums behind them
  uint64_t  tile_partials_cap = 0;   // doubles allocated
  uint64_t *d_wl[2] = {nullptr, nullptr};   // the worklists of mfx_hist_rest_kernel (mfx_kernels.h: mfx_hist_args::wl), one per launch slot (a streamed run's launches alternate between two streams)
  uint64_t  wl_cap[2] = {0, 0};      // entries each holds
  uint64_t *d_ovf = nullptr;         // the table of far K* bins: [0] distinct keys, [1] lost occurrences, [2, 2+S) occurrences, [2+S, 2+2S) keys (mfx_device.h)
  uint64_t *d_dbg = nullptr;         // [8] probe path counters of the DEBUG instance of the -hist kernel (mfx_eval_debug_enable); null: the measured instance runs
  uint8_t  *h_stage[2] = {nullptr, nullptr};   // pinned staging of the streamed upload (pageable sources), kept between calls
  size_t    h_stage_bytes = 0;
  void     *pool = nullptr;                             // host threads parked between streamed runs (mfx_api.cpp: WorkerPool)
  // device scratch of mfx_score_paths_trv (the variant modes score ~40 batches per run, one at a time: their ~20 arrays are carved out of one
  // allocation that is kept from batch to batch)
  uint8_t  *d_var_scratch = nullptr;
  uint64_t  var_scratch_bytes = 0;
  std::mutex var_scratch_mu;
  uint8_t  *h_pack[3] = {nullptr, nullptr, nullptr};   // pinned staging of the PACKED streamed upload (codes then validity words)
  size_t    h_pack_words = 0;
  // what a streamed run needs besides, kept between calls (creating and releasing it costs ~2.5 ms, 7 % of a 3 Gb run)
  struct {
    uint64_t  *d_counts = nullptr;                      // counts image
    size_t     words = 0;
    double    *d_kover = nullptr;
    uint64_t  *h_img = nullptr;                         // pinned: image + koverCpy
    hipStream_t copy = nullptr, kern[2] = {nullptr, nullptr};
    hipEvent_t up[3] = {nullptr, nullptr, nullptr}, kdone = nullptr;
    uint64_t  *d_exc[3] = {nullptr, nullptr, nullptr};  // per staging buffer: the validity words of a chunk that are not all ones
    size_t     exc_cap = 0;                             // entries each
  } sr;
};

// the evaluator's table of far K* bins (mfx_api.cpp): emptied asynchronously on a stream; collected as sorted {key, occurrences}
// pairs (header2: {distinct keys, lost occurrences} as read; MFX_E_OVERFLOW when occurrences were lost) and emptied
int mfx_ovf_reset_async(mfx_eval *ev, hipStream_t st);
int mfx_ovf_collect(mfx_eval *ev, uint64_t *header2, std::vector<uint64_t> *pairs, hipStream_t st);
