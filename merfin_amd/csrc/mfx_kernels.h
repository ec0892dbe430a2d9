// mfx_kernels.h -- kernel argument blocks and launch wrappers (mfx_kernels.hip)
#pragma once
#include "mfx_internal.h"

// K* parameters + accumulators shared by every histogram-producing kernel
struct mfx_kstar_args {
  double          peak;
  uint32_t        n_prob;
  const uint32_t *probK;
  const double   *probP;
  uint32_t        nbins;
  uint32_t        ncontigs;
  uint64_t       *counts;             // layout: include/merfin_amd.h MFX_HIST_WORDS
  double         *partials;           // [gridDim.x]
  uint64_t       *ovf;                // the evaluator's table of far K* bins (mfx_eval::d_ovf)
  const uint64_t *underq = nullptr;   // [MFX_MAXP_LDS * MFX_KLUT] mfx_kfix of the over-copy term of (read count, asmV) where the exact tables apply, else 0 (mfx_eval_create)
};

constexpr uint32_t MFX_WL_HEADER = 4096 + 2;    // words in front of the worklist's entries (<= 4096 segments)
struct mfx_hist_args {
  mfx_table_view  t;                  // t.compact: the 8-byte-slot layout of a sequence-only index (mfx_kernels.hip)
  int             canonical;          // 1: single probe of min(f,r); 0: probe both strands and sum
  const uint8_t  *bases;
  const uint64_t *codes = nullptr;    // non-null: read the tiles from the packed planes (same byte offsets / 32) instead of `bases`
  const uint32_t *valid = nullptr;
  const uint64_t *contig_off, *contig_len, *tile_start;
  uint32_t        ncontigs;
  uint64_t        tile_begin, tile_end;
  const uint32_t *tile_contig;        // [ntiles] contig of every tile
  uint64_t       *tile_ctr;           // dynamic tile scheduler: tiles handed out beyond the first gridDim.x (0 at launch)
  double         *tile_partials;      // [n_logical * MFX_BLOCK/64] koverCpy of every (tile, wave) of this launch
  // The launch evaluates n_logical tiles, numbered li = 0 .. n_logical-1:
  //   part_n == 1 : the contiguous range, tile = tile_begin + li
  //   part_n  > 1 : block-cyclic share of rank part_rank: tile = ((li >> part_shift) * part_n + part_rank << part_shift) + (li & mask)
  uint64_t        n_logical;
  uint32_t        part_rank, part_n, part_shift;
  uint64_t       *dbg = nullptr;      // non-null: the debug instance of the kernel counts the probe's endings here (mfx_eval_debug_counters)
  // the worklist of mfx_hist_rest_kernel (compact layout, canonical database; else null): words [2, 2 + wl_segs) = entries in each segment (one
  // per block of the main kernel's grid), from word MFX_WL_HEADER: 16-byte entries {k-mer, aux, slot | mode << 29 | dbl << 31}, segment g at g * wl_segcap
  uint64_t       *wl = nullptr;
  uint32_t        wl_segs = 0, wl_segcap = 0;
  mfx_kstar_args  ks;
};

struct mfx_route_args {
  mfx_table_view  t;
  const uint8_t  *bases;
  const uint64_t *contig_off, *contig_len, *tile_start;
  uint32_t        ncontigs;
  uint64_t        tile_begin, tile_end;
  uint32_t        nranks;
  uint64_t       *keys;               // [ntiles * MFX_TILE] canonical k-mer per position (~0: none)
  uint8_t        *owner;              // [ntiles * MFX_TILE] owner rank (255: none)
  uint64_t       *dest_counts;        // [nranks]
  uint64_t       *counts;             // counts image: kasm + per-contig kasm are added here
  uint32_t        nbins;
  // direct split (nranks <= MFX_SPLIT_MAX_RANKS): no sort, no position-sized scratch
  const uint32_t *tile_contig;        // [ntiles of the sequence set]
  uint32_t       *tile_cnt;           // [(tile_end - tile_begin) * nranks] k-mers of each tile per owner; then, in place,
                                      //   their exclusive prefix over the tiles (per owner)
};

constexpr uint32_t MFX_KEYS_MAX_SEGS = 16;
struct mfx_hist_keys_args {
  mfx_table_view  t;
  const uint64_t *keys = nullptr;     // nseg == 0: one array of n k-mers with their contigs; koverCpy as ordered fp64 partials (ks.partials)
  const uint32_t *contig = nullptr;
  uint64_t        n = 0;              // k-mers in all (nseg > 0: the sum of seg_n)
  // nseg > 0: the k-mers as segments, evaluated where they lie; koverCpy in fixed point (units of 2^-52) into kfix[0..1] (low, high word)
  uint32_t        nseg = 0;
  const uint64_t *seg_keys[MFX_KEYS_MAX_SEGS] = {};
  const uint32_t *seg_contig[MFX_KEYS_MAX_SEGS] = {};
  uint64_t        seg_n[MFX_KEYS_MAX_SEGS] = {};
  uint64_t       *kfix = nullptr;
  mfx_kstar_args  ks;
};

struct mfx_dump_args {
  mfx_table_view  t;
  int             canonical;
  const uint8_t  *src;         // contig base + first tile offset (128-byte aligned)
  uint64_t        npos;        // start positions to evaluate from src (tile-aligned begin)
  uint64_t        skip;        // first `skip` positions are not written (pos_begin % TILE)
  uint64_t        clen_left;   // contig bases remaining from src
  uint32_t       *readV, *asmV;// device output, index = position - skip
  double          peak;
  uint32_t        n_prob;
  const uint32_t *probK;
  const double   *probP;
  uint64_t       *stats;       // [0] kasm [1] kmissing
  int             recount = 0; // 1: readV holds final values; only the counters are taken again (sharded index)
};

struct mfx_count_args {
  mfx_table_view  t;
  const uint8_t  *bases;
  const uint64_t *codes = nullptr;    // the sequence's packed planes (2-bit codes, validity bits): read instead of `bases` when given
  const uint32_t *valid = nullptr;
  const uint64_t *contig_off, *contig_len, *tile_start;
  uint32_t        ncontigs;
  uint64_t        ntiles;
  uint64_t       *meta;
  int             count = 1;          // 1: asmV += 1 per occurrence (`meryl count`); 0: claim the k-mers only (sequence-only index); 2: asmV += 1 for the k-mers claimed BEFORE, no claims
};

// varMer::score of the paths of a batch (mfx_var_score_kernel): everything device memory
struct mfx_var_score_args {
  const uint8_t  *text;               // the packed path text (every path followed by '\n')
  const uint32_t *readV, *asmV;       // per start position of the text (mfx_dump_kernel)
  uint64_t        npaths;
  const uint64_t *off;                // [npaths] first base of the path in `text`
  const uint32_t *len, *nv;           // [npaths] bases; variants of its cluster
  const uint64_t *voff;               // [npaths] first of its nv entries in gt / vidx / vlen
  const uint64_t *cfirst;             // [npaths] number of the first path of its cluster
  const int32_t  *gt;
  const uint32_t *vidx, *vlen;
  uint32_t        k;
  int             need_dk;            // 0: numM only (-filter / -better / -strict / -loose), 1: + totdk (-polish)
  double          peak;
  uint32_t        n_prob;
  const uint32_t *probK;
  const double   *probP;
  uint32_t       *numM;               // [npaths] out
  double         *totdk;              // [npaths] out (need_dk)
};
hipError_t mfx_k_var_score(const mfx_var_score_args &a, hipStream_t st);
hipError_t mfx_k_var_traverse(const mfx_trv_cluster *cl, uint64_t ncl, const mfx_trv_variant *vars, const mfx_trv_allele *alleles, const char *win_text,
                              const char *allele_text, const mfx_trv_out &o, uint32_t *np, uint32_t *status, hipStream_t st);
hipError_t mfx_k_table_init(mfx_slot *slots, uint64_t nslots, hipStream_t st);
hipError_t mfx_k_table_add(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side,
                           uint64_t *meta, hipStream_t st);
// delta-coded blocks (mfx_db.cpp FLAT_DELTA): dir = {first k-mer, byte offset | kbits << 48 | vbits << 56} per block + one closing
// entry; payload_base = the file offset payload[0] holds; n = k-mers of the nblocks blocks
hipError_t mfx_k_table_add_delta(mfx_table_view t, const uint64_t *payload, const uint64_t *dir, uint32_t nblocks, uint64_t n,
                                 uint64_t payload_base, int side, uint64_t *meta, hipStream_t st);
// a PLACED database's blocks (records = the numbers P of mfx_place.h, ascending: the order of the table's lines)
hipError_t mfx_k_table_add_placed(mfx_table_view t, const uint64_t *payload, const uint64_t *dir, uint32_t nblocks, uint64_t n,
                                  uint64_t payload_base, int side, uint64_t *meta, hipStream_t st);
int        mfx_k_table_takes_placed(const mfx_table_view &t);
hipError_t mfx_k_place_keys(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, hipStream_t st);
hipError_t mfx_k_table_value(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV,
                             hipStream_t st);
hipError_t mfx_k_table_export(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV,
                              unsigned long long *count, hipStream_t st);
hipError_t mfx_k_hist(const mfx_hist_args &a, int grid, hipStream_t st);
hipError_t mfx_k_route(const mfx_route_args &a, hipStream_t st);
hipError_t mfx_k_route_split(const mfx_route_args &a, uint64_t *keys_out, uint32_t *contig_out, hipStream_t st);
hipError_t mfx_k_route_gather(const mfx_route_args &a, const uint32_t *idx, uint64_t nvalid, uint64_t *keys_out,
                              uint32_t *contig_out, hipStream_t st);
hipError_t mfx_k_iota(uint32_t *v, uint64_t n, hipStream_t st);
hipError_t mfx_k_hist_keys(const mfx_hist_keys_args &a, int grid, hipStream_t st);
// one-pass router (atomic reservation per tile and owner): owner d's k-mers at keys_out[d * region_cap ...), cursors[d] of them; cursors[nranks] != 0: a region overflowed
hipError_t mfx_k_route_fused(const mfx_route_args &a, uint64_t *keys_out, uint32_t *contig_out, uint64_t *cursors, uint64_t region_cap, hipStream_t st);
hipError_t mfx_k_sum_partials(const double *partials, uint32_t n, double *out, hipStream_t st);
hipError_t mfx_k_ordered_sum(const double *v, uint32_t n, double *out, hipStream_t st);
uint64_t mfx_k_tile_partials_words(uint64_t ntiles);
hipError_t mfx_k_sum_tile_partials(double *tile_partials, uint64_t ntiles, double *out, uint64_t *ctr_reset, hipStream_t st, int fixed);   // fixed: the values are integers, units of 2^-52 (k <= 31)
hipError_t mfx_k_hist_rest(const mfx_hist_args &a, int grid, hipStream_t st);
int mfx_k_hist_resident_blocks(int compact, int k);
int mfx_k_quot_supported();
hipError_t mfx_k_gather_rate(const void *table, uint64_t nlines, uint64_t *scratch, double *lines_per_s, hipStream_t st);
// 32 <= k <= 64 (mfx_wide.hip); kmers: two uint64 words per k-mer {low 64 bits, high bits}
hipError_t mfx_kw_table_add(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side, uint64_t *meta, hipStream_t st);
hipError_t mfx_kw_table_value(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV, hipStream_t st);
hipError_t mfx_kw_table_export(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV, unsigned long long *count, hipStream_t st);
hipError_t mfx_kw_hist(const mfx_hist_args &a, int grid, hipStream_t st);
hipError_t mfx_kw_dump(const mfx_dump_args &a, hipStream_t st);
hipError_t mfx_kw_count(const mfx_count_args &a, hipStream_t st);
hipError_t mfx_kw_completeness(mfx_table_view t, double peak, uint32_t n_prob, const uint32_t *probK, const double *probP, double *partials,
                               int grid, hipStream_t st);
hipError_t mfx_k_dump(const mfx_dump_args &a, hipStream_t st);
// *out += the content digest of nwords 32-base words (codes != nullptr: from the packed planes, else from the bytes)
hipError_t mfx_k_seq_digest(const uint8_t *bases, const uint64_t *codes, const uint32_t *valid, uint64_t nwords, uint64_t *out, hipStream_t st);
hipError_t mfx_k_add_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t st);
hipError_t mfx_k_pack(const uint8_t *bases, uint64_t *codes, uint32_t *valid, uint64_t nwords, hipStream_t st);
hipError_t mfx_k_unpack(const uint64_t *codes, const uint32_t *valid, uint8_t *bases, uint64_t nwords, hipStream_t st);
hipError_t mfx_k_valid_scatter(uint32_t *valid, const uint64_t *exc, uint32_t n, hipStream_t st);     // exc[i] = word index | word << 32
hipError_t mfx_k_count(const mfx_count_args &a, hipStream_t st);
hipError_t mfx_k_completeness(mfx_table_view t, double peak, uint32_t n_prob, const uint32_t *probK, const double *probP,
                              double *partials, int grid, hipStream_t st);
