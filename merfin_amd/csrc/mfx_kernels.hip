// mfx_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the merfin_amd
// evaluation path.  Written for MI355X only: 64-wide wavefronts, 160 KiB LDS,
// 128-byte HBM fetch granularity.  No MFMA: this is integer hashing and fp64
// scalar arithmetic bounded by random-access HBM line rate.
//
// Reference semantics implemented here (paths relative to /root/reference):
//   kmerIterator (meryl-utility; call sites merfin-histogram.C:54-64)
//   merfinGlobal::getK(kmer,kmer)        merfin-globals.C:101-110
//   processHistogram hot loop            merfin-histogram.C:54-91
//   processDump hot loop                 merfin-dump.C:44-67
//   computeCompleteness merge loop       merfin-completeness.C:70-117
//   `meryl count` of the assembly        merfin-globals.C:182-186
// plus the two kernels of the sharded index (no reference counterpart).
#include "mfx_internal.h"
#include "mfx_kernels.h"
#include "mfx_device.h"
#include "mfx_place.h"

#include <stdlib.h>
#include <algorithm>

// ===========================================================================
// joint k-mer table: 128-byte lines of 8 x {key, readV, asmV}
// ===========================================================================
__device__ __forceinline__ uint64_t mfx_revcomp(uint64_t fwd, int k) {
  uint64_t x = __brevll(fwd) >> (64 - 2 * k);                       // groups reversed, bits in each pair swapped
  x = ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
  uint64_t mask = (~0ULL) >> (64 - 2 * k);
  return (x ^ 0xAAAAAAAAAAAAAAAAULL) & mask;                         // complement = code ^ 2
}


// ---------------------------------------------------------------------------
// Probe sequence.  A k-mer's candidate lines are
//   region A: MFX_MZ_REGION consecutive lines starting at the line of its
//             canonical MINIMIZER (only when t.mz_w > 0), then
//   region B: consecutive lines starting at the line of the k-mer's own hash.
// Insert-only table; a k-mer lives in candidate line d only if lines 0..d-1
// were full when it was inserted, so a lookup stops at the first candidate
// line that holds the key or still has an empty slot.
//
// Why the minimizer: consecutive k-mers of a sequence share their minimizer
// for runs of ~(w+1)/2 positions, so their probes fall into the SAME 128-byte
// line -- fewer than one HBM line fetch per k-mer.  Default w = 3 windows
// (m = k-2, 0.53 line fetches per k-mer): a (k-2)-mer occurs in at most 48 k-mers,
// in practice a handful (its 2-3 genomic k-mers plus their sequencing-error
// neighbours), and a bucket that outgrows its line spills into region A's next
// lines.  w = 2 (m = k-1) bounds every bucket by 8 = one line (0.69 fetches per
// k-mer); w up to 5 is selectable (MFX_MZ_W); larger w costs more hashing per k-mer
// than the saved fetches return (profiles/r01_placement_w.txt).
// ---------------------------------------------------------------------------
constexpr uint32_t MFX_MZ_REGION = 4;
constexpr uint32_t MFX_MAX_LINES = 512;

struct mfx_probe {
  uint32_t lineA, lineB;
  uint32_t b0;                 // compact layout: the mini-bucket of the line the k-mer's slots are tried from (mfx_home)
  uint64_t fkey;               // compact layout: what the key field of the k-mer's slot holds in candidate line 0 -- the k-mer itself
                               // (k <= 21), its QUOTIENT (22 <= k <= 31, mfx_q_place); candidate line d: mfx_c_keyat
};

// Canonical minimizer of a k-mer: of its w windows of m = k-w+1 bases, the canonical m-mer
// whose ORDER hash is smallest.  Only the order matters, so the order hash is cheap (the
// m-mer folded to 32 bits, one 32-bit multiply; equal hashes: the first window wins -- a pure
// function of (key, rc), which is all insert and lookup need to agree on).
__device__ __forceinline__ uint64_t mfx_minimizer(uint64_t key, uint64_t rc, int k, int w) {
  const int m = k - w + 1;
  const uint64_t mmask = (~0ULL) >> (64 - 2 * m);
  uint64_t best = 0;
  uint32_t best_o = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    if (j < w) {
      uint64_t a = (key >> (2 * j)) & mmask;                 // m-mer starting at base w-1-j
      uint64_t b = (rc >> (2 * (w - 1 - j))) & mmask;        // its reverse complement
      uint64_t c = a < b ? a : b;                            // canonical m-mer: strand independent
      uint32_t o = ((uint32_t)c ^ (uint32_t)(c >> 32)) * 0x9E3779B1u;
      const bool take = (j == 0) || (o < best_o);
      best = take ? c : best;
      best_o = take ? o : best_o;
    }
  }
  return best;
}

// MOD-MINIMIZER (the compact layout): the window is not the one with the smallest m-mer but the one SAMPLED by the k-mer's
// smallest t-mer (t small: 4..7): of the k-t+1 t-mers of the k-mer the smallest (order hash of the canonical t-mer, ties: the
// leftmost in `key`) sits at offset x, and the k-mer's minimizer is its m-mer at offset x mod w.  Neighbouring k-mers agree
// on a t-mer for much longer than on an m-mer (a k-mer has 16 t-mers and 4 m-mers), and x mod w walks through the windows
// as the t-mer moves through the k-mer: a random sequence changes its minimizer every 3.35 positions instead of every 2.5
// (density 0.299 against 0.402 for w = 4) -- a quarter fewer table lines per k-mer at the same bucket size.  With
// (k - t) % w == w - 1 the k-mer and its reverse complement sample the same m-mer (offset x becomes k-t-x, window j becomes
// w-1-j), so that neighbours of either orientation share buckets; ties are the k-mer's own business (`key` is what the
// table stores: the canonical k-mer).  The order hash has 9 bits: it is compared together with 7 position bits in 16-bit lanes
// by the evaluation kernel, which finds the same offsets for a whole wave at once (mfx_wave_mod_lines).
__device__ __forceinline__ uint32_t mfx_tmer_order(uint32_t canonical_tmer) { return ((canonical_tmer * 0x9E3779B1u) >> 7) & 511u; }

// x: offset of the sampling t-mer; wa / wb: the m-mer of the sampled window as it stands in `key`, and its reverse complement
__device__ __forceinline__ void mfx_mod_window(uint64_t key, uint64_t rc, int k, int w, int t, uint32_t &x, uint64_t &wa, uint64_t &wb) {
  const uint32_t tmask = (1u << (2 * t)) - 1u;
  uint32_t best = 0xffffffffu;
  x = 0;
  for (int p = 0; p + t <= k; ++p) {                           // p: offset from the left (the most significant base)
    const uint32_t a = (uint32_t)(key >> (2 * (k - t - p))) & tmask, b = (uint32_t)(rc >> (2 * p)) & tmask;   // the t-mer and its reverse complement
    const uint32_t o = mfx_tmer_order(a < b ? a : b);
    if (o < best) { best = o; x = (uint32_t)p; }
  }
  const int m = k - w + 1, j = (int)(x % (uint32_t)w);          // window j from the left
  const uint64_t mmask = (~0ULL) >> (64 - 2 * m);
  wa = (key >> (2 * (w - 1 - j))) & mmask;
  wb = (rc >> (2 * j)) & mmask;
}
__device__ __forceinline__ uint64_t mfx_minimizer_mod(uint64_t key, uint64_t rc, int k, int w, int t, uint32_t &x) {
  uint64_t a, b;
  mfx_mod_window(key, rc, k, w, t, x, a, b);
  return a < b ? a : b;
}

// Line and first mini-bucket of a mod-minimizer (an m-mer of at most 36 bits: the compact layout holds k <= 21 in this form) sampled by
// the t-mer at offset x of the (canonical) k-mer.  Line: the high bits of a BIJECTION of the minimizer (mfx_place.h: mfx_p_mix -- the
// quotient form's; layout version 9; before: a 32-bit fold) scaled into the table, so that a database sorted by that mix walks the table
// line after line (the PLACED form, mfx_p_encode; +2 wave-VALU per k-mer in the -hist kernel against the fold).  Mini-bucket: (x + three
// other bits of the mix) mod 8 -- the k-mers that share a minimizer sit at consecutive positions and see its t-mer at DIFFERENT offsets,
// so they start at different mini-buckets of their common line instead of colliding at random: 3 % of the k-mers of a random sequence end
// outside their first mini-bucket at load factor 0.225 where a hash of the k-mer left 8.3 %.  x falls out of the -hist kernel's
// sliding-window minimum for free; a mini-bucket from the minimizer's WINDOW (which a placed record carries, unlike x) was tried
// for this form and costs the -hist kernel 12-25 wave-VALU per k-mer (profiles/r05_place_ab.txt): the placed update runs the t-mer
// scan on its decoded k-mers instead (it is bound by latency, not by its ALUs).  jo: that window, counted in the orientation in which the
// minimizer reads canonical -- used by the A/B build -DMFX_V_PLACE_WBUCKET=1 only.
#ifndef MFX_V_PLACE_OLDLINE
#define MFX_V_PLACE_OLDLINE 0         // A/B only (tools/ab_build.sh): the layout-8 line hash; such a table does not take placed databases in their order
#endif
#ifndef MFX_V_PLACE_WBUCKET
#define MFX_V_PLACE_WBUCKET 0         // A/B only: the mini-bucket from the oriented window instead of the t-mer's offset
#endif
__device__ __forceinline__ void mfx_mod_place(uint64_t mz, uint32_t jo, uint64_t nlines, uint32_t &line, uint32_t &b0, uint32_t x) {
#if MFX_V_PLACE_OLDLINE
  const uint32_t h = ((uint32_t)mz * 0x9E3779B1u) ^ (((uint32_t)(mz >> 32) + 0x7F4A7C15u) * 0x85EBCA77u);
  line = __umulhi(h ^ (h >> 15), (uint32_t)nlines);
  const uint32_t top = h << 3;
#else
  const uint32_t top = mfx_p_mix((uint32_t)mz, (uint32_t)(mz >> 32));
  line = __umulhi(top, (uint32_t)nlines);
#endif
#if MFX_V_PLACE_WBUCKET
  b0 = (2u * jo + (top >> 3)) & 7u;
  (void)x;
#else
  b0 = (x + (top >> 3)) & 7u;
  (void)jo;
#endif
}

// ---------------------------------------------------------------------------
// QUOTIENT form of the compact layout (22 <= k <= 31; w = 4, mod-minimizer).  A k-mer of up to 62 bits does not fit the 42-bit
// key field of an 8-byte slot -- but most of it is implied by WHERE the slot is.  The k-mer is (its minimizer: an m-mer of
// m = k - 3 bases in canonical form, the strand it stands in, the window j it occupies, the 3 bases around it); the line of the
// bucket is taken from the high bits of a BIJECTION of the minimizer, so the slot only has to keep what the line does not say:
//   top   = mix(low 32 bits of the minimizer) ^ (high bits * C)          -- invertible given the high bits, which are stored as they are
//   line  = (top * nlines) >> 32,   f = low 32 bits of top * nlines      -- two `top` of one line differ in f by >= nlines >= 2^qshift,
//   fq    = f >> qshift,  qshift = floor(log2(nlines))                      so fq tells them apart in 32 - qshift bits
//   F0    = {high bits of the minimizer : 2m - 32 | fq : 32 - qshift | strand : 1 | j : 2 | outer bases : 6}   <= 40 bits
// (the host makes the table large enough for that: nlines >= 2^(2m - 31), mfx_api.cpp).  A slot in candidate line d of its k-mer
// holds F0 | d << 40 with d <= 2: (line, key field) <-> k-mer is one to one (mfx_q_invert is the way back, used by the export),
// so a 42-bit compare is an exact match, and the all-ones word (d = 3) stays the empty slot.  A k-mer whose three candidate
// lines are full lives in the side table under its full key (mfx_c_claim).  Everything else of the layout -- 16 slots per line,
// mini-buckets, counts of 11 bits with the side table behind them, the probe -- is that of k <= 21: -hist at k = 31 reads one
// 16-byte mini-bucket per k-mer from a table half the size of the 16-byte-slot form.
// ---------------------------------------------------------------------------
constexpr uint32_t MFX_Q_LINES = 3;            // candidate lines of the quotient form (d = 0 .. 2)
constexpr int      MFX_Q_DSHIFT = 40;          // d sits above the 40 bits of F0

__device__ __forceinline__ uint32_t mfx_q_mix(uint32_t lo, uint32_t hi) { return mfx_p_mix(lo, hi); }          // (mfx_place.h: shared with the host)
__device__ __forceinline__ uint32_t mfx_q_unmix(uint32_t top, uint32_t hi) { return mfx_p_unmix(top, hi); }

// c: canonical minimizer; sbit: it stands reversed in the (canonical) k-mer; j: its window from the left; e: the j bases left and
// 3 - j bases right of it; x: offset of the sampling t-mer (first mini-bucket, as mfx_mod_place)
__device__ __forceinline__ void mfx_q_place(const mfx_table_view &t, uint64_t c, uint32_t sbit, uint32_t j, uint32_t e, uint32_t x,
                                            uint32_t &line, uint32_t &b0, uint64_t &f0) {
  const uint32_t hi = (uint32_t)(c >> 32), top = mfx_q_mix((uint32_t)c, hi), nl = (uint32_t)t.nlines;
  line = __umulhi(top, nl);
  const uint32_t fq = (top * nl) >> t.qshift;
  (void)x;
  b0 = mfx_p_bucket(top, j, sbit);
  const int R = 2 * (t.k - 3) - 32, Q = 32 - t.qshift;
  f0 = (uint64_t)hi | ((uint64_t)fq << R) | ((uint64_t)(sbit | (j << 1) | (e << 3)) << (R + Q));
}

// the pieces of a canonical k-mer `key` (rc: its reverse complement) that mfx_q_place takes
__device__ __forceinline__ void mfx_q_parts(const mfx_table_view &t, uint64_t key, uint64_t rc, uint32_t &x, uint64_t &c, uint32_t &sbit,
                                            uint32_t &j, uint32_t &e) {
  uint64_t a, b;
  mfx_mod_window(key, rc, t.k, 4, t.mz_t, x, a, b);
  j = x & 3u;
  c = a < b ? a : b;
  sbit = b < a ? 1u : 0u;
  const int m = t.k - 3;
  e = (uint32_t)(((key >> (2 * (m + 3 - (int)j))) << (2 * (3 - (int)j))) | (key & ((1ull << (2 * (3 - (int)j))) - 1ull)));
}

// (home line, F0) -> the k-mer (export, tests): the inverse of mfx_q_parts + mfx_q_place
__device__ __forceinline__ uint64_t mfx_q_invert(const mfx_table_view &t, uint32_t home, uint64_t f0) {
  const int m = t.k - 3, R = 2 * m - 32, Q = 32 - t.qshift;
  const uint32_t hi = (uint32_t)(f0 & ((1ull << R) - 1ull)), fq = (uint32_t)(f0 >> R) & (uint32_t)((1ull << Q) - 1ull);
  const uint32_t meta = (uint32_t)(f0 >> (R + Q)), sbit = meta & 1u, j = (meta >> 1) & 3u, e = (meta >> 3) & 63u;
  const uint32_t nl = (uint32_t)t.nlines;
  uint32_t top = (uint32_t)((((uint64_t)home << 32) + nl - 1u) / nl);          // the smallest `top` of this line
  const uint32_t f_first = top * nl, want = fq << t.qshift;                    // f grows by nl from one `top` of the line to the next, without wrapping
  if (want > f_first) top += (uint32_t)(((uint64_t)(want - f_first) + nl - 1u) / nl);
  const uint64_t c = ((uint64_t)hi << 32) | mfx_q_unmix(top, hi);
  const uint64_t mmer = sbit ? mfx_revcomp(c, m) : c;
  const uint64_t left = e >> (2 * (3 - j)), right = e & ((1u << (2 * (3 - j))) - 1u);
  return (left << (2 * (m + 3 - (int)j))) | (mmer << (2 * (3 - j))) | right;
}

// line of a k-mer's minimizer: one odd 64-bit multiplication (the high half of the product
// depends on every bit of the m-mer), then the multiply-range reduction.  The multiplier must
// be unrelated to the order hash's: the minimizer is the window with the SMALLEST order hash,
// so a line hash correlated with it would crowd the low lines.
__device__ __forceinline__ uint32_t mfx_mz_line(const mfx_table_view &t, uint64_t key, uint64_t krc) {
  if (t.quot) {
    uint32_t x, sbit, j, e, line, b0;
    uint64_t c, f0;
    mfx_q_parts(t, key, krc, x, c, sbit, j, e);
    mfx_q_place(t, c, sbit, j, e, x, line, b0, f0);
    return line;
  }
  if (t.mz_t) {
    uint32_t x, line, b0;
    uint64_t wa, wb;
    mfx_mod_window(key, krc, t.k, t.mz_w, t.mz_t, x, wa, wb);
    { const uint32_t jw = x % (uint32_t)t.mz_w; mfx_mod_place(wa < wb ? wa : wb, wb < wa ? (uint32_t)t.mz_w - 1u - jw : jw, t.nlines, line, b0, x); }
    return line;
  }
  return mfx_range32(mfx_minimizer(key, krc, t.k, t.mz_w) * 0xD6E8FEB86659FD93ULL, t.nlines);
}

// first mini-bucket of a compact line by a hash of the k-mer (the compact layout without the mod-minimizer placement)
__device__ __forceinline__ uint32_t mfx_c_first(uint64_t key) {
  uint32_t x = (uint32_t)key ^ (uint32_t)(key >> 32);
  x ^= x >> 15;
  x ^= x >> 7;
  return (x ^ (x >> 3)) & 7u;
}

__device__ __forceinline__ mfx_probe mfx_home(const mfx_table_view &t, uint64_t key) {
  mfx_probe pr;
  uint64_t h = mfx_hash64(key);
  pr.lineB = mfx_range32(h, t.nlines);
  pr.lineA = pr.lineB;
  pr.b0 = 0u;
  pr.fkey = key;
  if (t.quot) {                                                // compact layout, quotient form (mfx_q_place)
    uint32_t x, sbit, j, e;
    uint64_t c;
    mfx_q_parts(t, key, mfx_revcomp(key, t.k), x, c, sbit, j, e);
    mfx_q_place(t, c, sbit, j, e, x, pr.lineA, pr.b0, pr.fkey);
    return pr;
  }
  if (t.mz_t) {                                                // compact layout, mod-minimizer: line and first mini-bucket together
    uint32_t x;
    uint64_t wa, wb;
    mfx_mod_window(key, mfx_revcomp(key, t.k), t.k, t.mz_w, t.mz_t, x, wa, wb);
    { const uint32_t jw = x % (uint32_t)t.mz_w; mfx_mod_place(wa < wb ? wa : wb, wb < wa ? (uint32_t)t.mz_w - 1u - jw : jw, t.nlines, pr.lineA, pr.b0, x); }
    return pr;
  }
  if (t.mz_w > 0)
    pr.lineA = mfx_mz_line(t, key, mfx_revcomp(key, t.k));
  if (t.compact) pr.b0 = mfx_c_first(key);
  return pr;
}

// first candidate line only (the hot path needs nothing else); krc = revcomp(key)
__device__ __forceinline__ uint32_t mfx_first_line(const mfx_table_view &t, uint64_t key, uint64_t krc) {
  if (t.mz_w > 0) return mfx_mz_line(t, key, krc);
  return mfx_range32(mfx_hash64(key), t.nlines);
}

// owner rank of a k-mer in a sharded index: a hash of its minimizer (of the k-mer under plain hashing) on multipliers the line
// hash does not use, so that the owner is independent of the line inside the owner's table; 32-bit arithmetic -- the router
// evaluates it for every position of the assembly (layout version 8: before, a 64-bit finaliser and a 64-bit high product)
__device__ __forceinline__ uint32_t mfx_owner(const mfx_table_view &t, uint64_t key, uint64_t krc, uint32_t nranks) {
  const uint64_t x = t.mz_w > 0 ? mfx_minimizer(key, krc, t.k, t.mz_w) : key;
  uint32_t h = ((uint32_t)x * 0xCC9E2D51u) ^ (((uint32_t)(x >> 32) + 0x1B873593u) * 0xE6546B64u);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13;
  return __umulhi(h, nranks);
}

// d-th candidate line
__device__ __forceinline__ uint64_t mfx_probe_line(const mfx_table_view &t, const mfx_probe &pr, uint32_t d) {
  const uint32_t ra = t.mz_w > 0 ? MFX_MZ_REGION : 0u;
  uint64_t ln = d < ra ? (uint64_t)pr.lineA + d : (uint64_t)pr.lineB + (d - ra);
  if (ln >= t.nlines) ln -= t.nlines;
  return ln;
}

__device__ __forceinline__ uint4 mfx_load_slot(const mfx_table_view &t, uint64_t s) {
  return *reinterpret_cast<const uint4 *>(t.slots + s);
}

// per-lane scan of candidate lines d0, d0+1, ...  All 8 slots of a line are
// requested back to back (one memory round trip per line, not eight): a lane on
// this path stalls its whole wave, so latency matters more than the extra loads.
__device__ __forceinline__ uint2 mfx_scan_lines(const mfx_table_view &t, uint64_t key, const mfx_probe &pr, uint32_t d0) {
  for (uint32_t d = d0; d < MFX_MAX_LINES; ++d) {
    const uint4 *ln = reinterpret_cast<const uint4 *>(t.slots + mfx_probe_line(t, pr, d) * MFX_SLOTS_LINE);
    uint4 s[MFX_SLOTS_LINE];
#pragma unroll
    for (uint32_t q = 0; q < MFX_SLOTS_LINE; ++q) s[q] = ln[q];
    bool any_empty = false;
    uint2 hit = make_uint2(0u, 0u);
    bool found = false;
#pragma unroll
    for (uint32_t q = 0; q < MFX_SLOTS_LINE; ++q) {
      uint64_t sk = (uint64_t)s[q].x | ((uint64_t)s[q].y << 32);
      if (sk == key) { found = true; hit = make_uint2(s[q].z, s[q].w); }
      any_empty |= (sk == MFX_EMPTY);
    }
    if (found) {
      if (hit.x < t.minV || hit.x > t.maxV) hit.x = 0;   // -min / -max (merfin.C:199-200)
      return hit;
    }
    if (any_empty) break;                                 // the line still has room: the key was never inserted
  }
  return make_uint2(0u, 0u);                              // absent -> value 0 (merfin-globals.C:84)
}

__device__ __forceinline__ uint2 mfx_lookup(const mfx_table_view &t, uint64_t key) {
  return mfx_scan_lines(t, key, mfx_home(t, key), 0);
}

// find-or-claim the slot of `key`; nullptr when the probe limit is hit.  New k-mers are
// counted in the caller's register (`fresh`) -- one shared counter word bumped by every
// insert serialises the whole build (a single address takes ~90 M atomics/s).
__device__ __forceinline__ mfx_slot *mfx_claim(const mfx_table_view &t, uint64_t key, uint64_t *meta, uint32_t &fresh) {
  const mfx_probe pr = mfx_home(t, key);
  for (uint32_t d = 0; d < MFX_MAX_LINES; ++d) {
    const uint64_t base = mfx_probe_line(t, pr, d) * MFX_SLOTS_LINE;
    for (uint32_t q = 0; q < MFX_SLOTS_LINE; ++q) {
      mfx_slot *sl = t.slots + base + q;                 // slots fill in order: a line has room <=> its LAST slot is empty
      unsigned long long *kp = reinterpret_cast<unsigned long long *>(&sl->key);
      unsigned long long cur = __hip_atomic_load(kp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (cur == MFX_EMPTY) {
        cur = atomicCAS(kp, (unsigned long long)MFX_EMPTY, (unsigned long long)key);
        if (cur == MFX_EMPTY) {
          ++fresh;
          return sl;
        }
      }
      if (cur == key)
        return sl;
    }
  }
  atomicAdd((unsigned long long *)&meta[2], 1ull);
  return nullptr;
}

// wave-reduce the per-lane insert statistics, one atomic per wave
__device__ __forceinline__ void mfx_meta_flush(uint64_t *meta, uint32_t fresh, uint32_t noncanon) {
  uint64_t f = fresh, c = noncanon;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { f += __shfl_down(f, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63u) == 0) {
    if (f) atomicAdd((unsigned long long *)&meta[0], (unsigned long long)f);
    if (c) atomicAdd((unsigned long long *)&meta[1], (unsigned long long)c);
  }
}

// ===========================================================================
// Sequence-only index, compact layout (k <= 21): the k-mers CLAIMED from a sequence in 8-byte slots, 16 per 128-byte
// line -- {key: 42 bits | readV: 11 | asmV: 11}.  -hist and -dump ask the lookup tables for the k-mers of -sequence
// and nothing else (merfin-histogram.C:54-64, merfin-dump.C:44-61), so a table that holds exactly those answers them
// as the full tables would: half the keys of a human read database never get a slot, twice the slots fit a line, and a
// minimizer bucket of w = 4 windows (assembly k-mers only: no sequencing-error neighbours) stays in its home line --
// 0.42 lines per k-mer instead of 0.52.  The table is built DIRECTLY in this form: mfx_index_claim_seq claims the keys
// (mfx_c_claim), every later add / load only updates (mfx_c_add) and drops what was not claimed.
// A count field of MFX_CSAT = 2047 means "saturated": the exact count of that side lives in the side table (standard
// 16-byte slots, plain hashing; mfx_side_view).  Counts are moved there by the add that crosses the limit, so a field
// is either exact or saturated, never wrapped.  The all-ones word is the empty slot: its key field would be the
// poly-G 21-mer, which is never canonical (poly-C is), and only canonical k-mers are claimed.
// ===========================================================================
__device__ __forceinline__ mfx_table_view mfx_side_view(const mfx_table_view &c) {
  mfx_table_view s = c;
  s.slots = c.side; s.nlines = c.side_nlines;
  s.mz_w = 0; s.mz_t = 0; s.compact = 0; s.seq_only = 0;
  s.minV = 0u; s.maxV = 0xffffffffu;                           // the read filter is applied to the resolved count
  s.shard_rank = 0u; s.shard_n = 1u;
  return s;
}

// what the key field of a k-mer's slot holds in candidate line d (quotient form: d rides above F0; else the k-mer, whatever d)
__device__ __forceinline__ uint64_t mfx_c_keyat(const mfx_table_view &c, uint64_t fkey, uint32_t d) {
  return c.quot ? fkey | ((uint64_t)d << MFX_Q_DSHIFT) : fkey;
}
// candidate lines a compact k-mer may live in (beyond them, quotient form: the side table)
__device__ __forceinline__ uint32_t mfx_c_maxlines(const mfx_table_view &c) { return c.quot ? MFX_Q_LINES : MFX_MAX_LINES; }

// the slot of `key` in a 16-byte-slot table (the side table of a compact index), or nullptr
__device__ __forceinline__ mfx_slot *mfx_find_slot(const mfx_table_view &t, uint64_t key) {
  const mfx_probe pr = mfx_home(t, key);
  for (uint32_t d = 0; d < MFX_MAX_LINES; ++d) {
    mfx_slot *ln = t.slots + mfx_probe_line(t, pr, d) * MFX_SLOTS_LINE;
    bool any_empty = false;
#pragma unroll 1
    for (uint32_t q = 0; q < MFX_SLOTS_LINE; ++q) {
      const unsigned long long sk = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&ln[q].key), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (sk == key) return ln + q;
      any_empty |= sk == MFX_EMPTY;
    }
    if (any_empty) break;
  }
  return nullptr;
}

// the pair of a found compact slot from its low word (a saturated field: the side table has the exact count)
__device__ __forceinline__ uint2 mfx_c_fields(const mfx_table_view &c, uint64_t key, uint32_t lo) {
  uint32_t rv = (lo >> 11) & MFX_CSAT, av = lo & MFX_CSAT;
  if (rv == MFX_CSAT || av == MFX_CSAT) {
    const uint2 x = mfx_lookup(mfx_side_view(c), key);
    if (rv == MFX_CSAT) rv = x.x;
    if (av == MFX_CSAT) av = x.y;
  }
  if (rv < c.minV || rv > c.maxV) rv = 0;                      // -min / -max (merfin.C:199-200)
  return make_uint2(rv, av);
}

// per-lane scan of candidate lines d0, d0+1, ...: the slot holding the k-mer (its word in `word`), or nullptr when the
// first line with room does not hold it (never claimed).  The whole line is requested at once (8 x 16 bytes).
// beyond (quotient form): every candidate line is full of other k-mers -- if the k-mer was claimed, it is in the side table.
__device__ __forceinline__ unsigned long long *mfx_c_find(const mfx_table_view &c, const mfx_probe &pr, uint32_t d0,
                                                          unsigned long long &word, bool &beyond) {
  unsigned long long *cs = reinterpret_cast<unsigned long long *>(c.slots);
  beyond = false;
  const uint32_t dmax = mfx_c_maxlines(c);
  for (uint32_t d = d0; d < dmax; ++d) {
    unsigned long long *base = cs + mfx_probe_line(c, pr, d) * MFX_CSLOTS_LINE;
    const uint64_t key = mfx_c_keyat(c, pr.fkey, d);
    const uint4 *ln = reinterpret_cast<const uint4 *>(base);
    uint4 s[8];
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) s[q] = ln[q];
    bool any_empty = false;
    int at = -1;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
      const uint64_t x = (uint64_t)s[q].x | ((uint64_t)s[q].y << 32), y = (uint64_t)s[q].z | ((uint64_t)s[q].w << 32);
      if (x != MFX_EMPTY && (x >> 22) == key) { at = 2 * (int)q; word = x; }
      if (y != MFX_EMPTY && (y >> 22) == key) { at = 2 * (int)q + 1; word = y; }
      any_empty |= (x == MFX_EMPTY) || (y == MFX_EMPTY);
    }
    if (at >= 0) return base + at;
    if (any_empty) return nullptr;
  }
  beyond = c.quot != 0;
  return nullptr;
}

__device__ __forceinline__ uint2 mfx_c_lookup(const mfx_table_view &c, uint64_t key) {
  unsigned long long w = 0;
  bool beyond;
  if (!mfx_c_find(c, mfx_home(c, key), 0, w, beyond)) {
    if (!beyond) return make_uint2(0u, 0u);                     // absent -> value 0 (merfin-globals.C:84)
    uint2 x = mfx_lookup(mfx_side_view(c), key);                // beyond its candidate lines: under its full key in the side table
    if (x.x < c.minV || x.x > c.maxV) x.x = 0;                  // -min / -max (merfin.C:199-200)
    return x;
  }
  return mfx_c_fields(c, key, (uint32_t)w);
}

// The same two lookups for the RARE endings of the evaluation kernel's probe (a saturated count; a k-mer beyond the first two of its
// candidate lines, which the cooperative passes of the probe cover): one slot at a time, a handful of registers.  A kernel's register
// allocation is set by its hungriest path however rarely it runs; the whole-line forms above hold 32 registers of slots in flight.
__device__ __forceinline__ uint2 mfx_side_lookup_lean(const mfx_table_view &c, uint64_t key) {
  const mfx_table_view t = mfx_side_view(c);
  const mfx_probe pr = mfx_home(t, key);
  for (uint32_t d = 0; d < MFX_MAX_LINES; ++d) {
    const uint4 *ln = reinterpret_cast<const uint4 *>(t.slots + mfx_probe_line(t, pr, d) * MFX_SLOTS_LINE);
    bool any_empty = false;
#pragma unroll 1
    for (uint32_t q = 0; q < MFX_SLOTS_LINE; ++q) {
      const uint4 s = ln[q];
      const uint64_t sk = (uint64_t)s.x | ((uint64_t)s.y << 32);
      if (sk == key) return make_uint2(s.z, s.w);
      any_empty |= sk == MFX_EMPTY;
    }
    if (any_empty) break;
  }
  return make_uint2(0u, 0u);
}

// (found, the slot's low word) of the k-mer with key field `fkey` and home line `lineA`, from candidate line d0 on
__device__ __forceinline__ uint2 mfx_c_find_lean(const mfx_table_view &c, uint64_t fkey, uint32_t lineA, uint32_t d0, bool &beyond) {
  mfx_probe pr;
  pr.lineA = lineA;
  pr.lineB = c.quot ? 0u : mfx_range32(mfx_hash64(fkey), c.nlines);    // (k <= 21: fkey is the k-mer; candidate lines >= MFX_MZ_REGION follow its own hash)
  pr.b0 = 0u;
  pr.fkey = fkey;
  beyond = false;
  const uint32_t dmax = mfx_c_maxlines(c);
  for (uint32_t d = d0; d < dmax; ++d) {
    const uint4 *ln = reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned long long *>(c.slots) + mfx_probe_line(c, pr, d) * MFX_CSLOTS_LINE);
    const uint64_t key = mfx_c_keyat(c, fkey, d);
    bool any_empty = false;
#pragma unroll 1
    for (uint32_t q = 0; q < MFX_CSLOTS_LINE / 2; ++q) {
      const uint4 s = ln[q];
      const uint64_t x = (uint64_t)s.x | ((uint64_t)s.y << 32), y = (uint64_t)s.z | ((uint64_t)s.w << 32);
      if (x != MFX_EMPTY && (x >> 22) == key) return make_uint2(1u, s.x);
      if (y != MFX_EMPTY && (y >> 22) == key) return make_uint2(1u, s.z);
      any_empty |= (x == MFX_EMPTY) || (y == MFX_EMPTY);
    }
    if (any_empty) return make_uint2(0u, 0u);
  }
  beyond = c.quot != 0;
  return make_uint2(0u, 0u);
}

// the pair of a found compact slot from its low word, saturated fields from the side table (mfx_c_fields, lean)
__device__ __forceinline__ uint2 mfx_c_fields_lean(const mfx_table_view &c, uint64_t key, uint32_t lo) {
  uint32_t rv = (lo >> 11) & MFX_CSAT, av = lo & MFX_CSAT;
  if (rv == MFX_CSAT || av == MFX_CSAT) {
    const uint2 x = mfx_side_lookup_lean(c, key);
    if (rv == MFX_CSAT) rv = x.x;
    if (av == MFX_CSAT) av = x.y;
  }
  if (rv < c.minV || rv > c.maxV) rv = 0;                      // -min / -max (merfin.C:199-200)
  return make_uint2(rv, av);
}

// Where in its line a k-mer goes: the line is eight MINI-BUCKETS of two slots (16 bytes: one dwordx4 of one lane); a
// k-mer's slots are tried from mini-bucket mfx_c_first(key) on, around the line, then on through the candidate lines.
// At the load factors this layout is built at (<= 0.5, 0.25 by default) 92 % of the k-mers sit in their first mini-bucket,
// so a lookup is ONE 16-byte load of one lane for most queries (mfx_lane_lookup8) -- and neighbouring k-mers, which share
// their minimizer's line, make those loads fall into the same 128-byte lines.  A lookup stops at the key or at the first
// empty slot of its own order; scanning a WHOLE line (mfx_c_find, the cooperative probe) may stop at any empty slot: the
// k-mer's order visits every slot of a line before it leaves it.

// find-or-claim the slot of `key` (mfx_claim for 8-byte slots): a k-mer takes the first empty slot of its order, a slot
// never changes its key once written.  cur = the slot's word as seen (a fresh claim: the key with both counts 0).
// init: the counts a FRESH slot starts with (the assembly counter claims with asmV = 1: one atomic per new k-mer, not two);
// claimed = this call wrote the slot.
__device__ __forceinline__ unsigned long long *mfx_c_claim(const mfx_table_view &c, uint64_t key, uint64_t *meta, uint32_t &fresh,
                                                           unsigned long long &cur, uint32_t init, bool &claimed, mfx_slot *&side_slot) {
  const mfx_probe pr = mfx_home(c, key);
  unsigned long long *cs = reinterpret_cast<unsigned long long *>(c.slots);
  claimed = false;
  side_slot = nullptr;
  const uint32_t q0 = 2u * pr.b0;
  const uint32_t dmax = mfx_c_maxlines(c);
  // A mini-bucket is read by ONE plain 16-byte load.  It may come from this CU's L1 and be older than the table: a slot
  // seen occupied stays what it is (a slot never changes its key once written), a slot seen empty is taken by compare-and-
  // swap, whose answer is the truth -- the claim, the k-mer itself (another lane claimed it first), or another key (on).
  for (uint32_t d = 0; d < dmax; ++d) {
    unsigned long long *base = cs + mfx_probe_line(c, pr, d) * MFX_CSLOTS_LINE;
    const uint64_t kf = mfx_c_keyat(c, pr.fkey, d);
    const unsigned long long mine = ((unsigned long long)kf << 22) | init;
    for (uint32_t qi = 0; qi < MFX_CSLOTS_LINE; qi += 2) {
      const uint32_t q = (q0 + qi) & (MFX_CSLOTS_LINE - 1u);
      const uint4 s = *reinterpret_cast<const uint4 *>(base + q);
      const unsigned long long seen[2] = {(unsigned long long)s.x | ((unsigned long long)s.y << 32),
                                          (unsigned long long)s.z | ((unsigned long long)s.w << 32)};
#pragma unroll
      for (uint32_t e = 0; e < 2; ++e) {
        cur = seen[e];
        if (cur == MFX_EMPTY) {
          cur = atomicCAS(base + q + e, (unsigned long long)MFX_EMPTY, mine);
          if (cur == MFX_EMPTY) { ++fresh; cur = mine; claimed = true; return base + q + e; }
        }
        if ((cur >> 22) == kf) return base + q + e;          // cur is not the empty word here
      }
    }
  }
  if (c.quot) {
    // quotient form: the three candidate lines are full of other k-mers (and stay full: slots never empty again) -- the k-mer
    // lives in the side table under its full key, with its counts (zero at the claim: the caller adds there)
    side_slot = mfx_claim(mfx_side_view(c), key, meta, fresh);
    return nullptr;
  }
  atomicAdd((unsigned long long *)&meta[2], 1ull);
  return nullptr;
}

// counts[side] += v of the slot at w (cur = its word as last seen).  A field that would reach MFX_CSAT is set to
// MFX_CSAT by the same compare-and-swap and everything it held moves to the side table; once saturated, adds go there.
__device__ __forceinline__ void mfx_c_add(const mfx_table_view &c, unsigned long long *w, unsigned long long cur, uint64_t key, uint32_t v,
                                          int side, uint64_t *meta) {
  if (v == 0u) return;
  const int sh = side ? 0 : 11;
  uint32_t amount;
  while (true) {
    const uint32_t f = (uint32_t)(cur >> sh) & MFX_CSAT;
    if (f == MFX_CSAT) { amount = v; break; }
    const uint64_t sum = (uint64_t)f + v;
    const unsigned long long nw = sum >= MFX_CSAT ? (cur | ((unsigned long long)MFX_CSAT << sh)) : cur + ((unsigned long long)v << sh);
    const unsigned long long old = atomicCAS(w, cur, nw);
    if (old == cur) {
      if (sum < MFX_CSAT) return;
      amount = (uint32_t)sum;                                  // uint32 arithmetic, as the standard table's atomicAdd
      break;
    }
    cur = old;
  }
  uint32_t side_fresh = 0;                                     // side-table claims are not new k-mers of the index
  mfx_slot *sl = mfx_claim(mfx_side_view(c), key, meta, side_fresh);
  if (sl) atomicAdd(side ? &sl->asmV : &sl->readV, amount);
}

// ---------------------------------------------------------------------------
// Wave-cooperative insert (index build, assembly k-mer counting): the mirror image of the cooperative lookup.
// Each lane brings one key; the 8 lanes of a lane-group serve their group's 8 keys ("rounds" S = 0..7), and for
// the key of a round they read the 8 slots of its candidate line with ONE coalesced 128-byte request (lane `sub`
// reads slot `sub`), so that "is the key there" and "where is the first empty slot" are two ballots and exactly
// one lane per key issues the claiming compare-and-swap and the count update -- instead of every lane walking up
// to 8 slots of its own line with dependent atomic loads (the divergent access pattern the lookup path avoids).
//
// Nothing waits per key: a pass first DECIDES every pending round (match -> add, fire and forget; empty slot -> the
// CAS is issued; line full -> next candidate line), then collects all CAS results at once, then re-reads the lines
// of the rounds still pending, all requests of a phase in flight together.  Rounds of a group that target the SAME
// line (consecutive k-mers share their minimizer's line) are serialised: only the first pending one acts in a
// pass, the others see the line as it is afterwards.  Random keys finish in one pass (load, CAS: two round trips
// for 64 keys per wave); chains of neighbours take one more pass per link.
//
// Slots of a line fill in order: the claim goes to the LOWEST empty slot of a freshly read line, and a lost race
// re-reads the line.  A slot never changes once written, so when slot j is claimed for a key, every slot below j was
// SEEN occupied by another key -- the key cannot be in the line twice, and lookups may rely on "the line has room
// <=> its last slot is empty".  Keys are read with agent-scope atomic loads: a plain load could be served from
// this CU's L1, which other CUs' claims never refresh (a lost race would then spin on the stale line).
// ---------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ uint32_t mfx_group_bcast(uint32_t v);          // defined with the lookup path below

__device__ __forceinline__ unsigned long long mfx_slot_key_load(const mfx_table_view &t, uint32_t line, uint32_t sub) {
  return __hip_atomic_load(reinterpret_cast<unsigned long long *>(&t.slots[(uint64_t)line * MFX_SLOTS_LINE + sub].key), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

struct mfx_ins_rounds {         // what the 8 lanes of a group know about their group's 8 keys (group-uniform values)
  uint32_t klo[8], khi[8], lineA[8], lineB[8], val[8];      // val == 0: no key in this round
  uint32_t line[8], d[8];                                    // current candidate line and its number in the probe sequence
};

template <int S>
__device__ __forceinline__ void mfx_ins_announce(mfx_ins_rounds &R, uint64_t key, const mfx_probe &pr, uint32_t v) {
  R.klo[S] = mfx_group_bcast<S>((uint32_t)key);
  R.khi[S] = mfx_group_bcast<S>((uint32_t)(key >> 32));
  R.lineA[S] = mfx_group_bcast<S>(pr.lineA);
  R.lineB[S] = mfx_group_bcast<S>(pr.lineB);
  R.val[S] = mfx_group_bcast<S>(v);
}

// one key per lane (v == 0: none); adds v to the key's read (side 0) or assembly (side 1) count.
// CLAIM = false: update-only (sequence-only index) -- a key that is not in the table is dropped, not inserted, and `fresh`
// counts those instead of the new k-mers.
template <bool CLAIM = true>
__device__ __forceinline__ void mfx_group_insert(const mfx_table_view &t, uint64_t key, uint32_t v, int side, uint64_t *meta,
                                                 uint32_t &fresh) {
  const uint32_t lane = threadIdx.x & 63u, sub = lane & 7u, gsh = lane & ~7u;
  mfx_probe pr;
  pr.lineA = pr.lineB = 0u;
  if (v) pr = mfx_home(t, key);
  mfx_ins_rounds R;
  mfx_ins_announce<0>(R, key, pr, v); mfx_ins_announce<1>(R, key, pr, v); mfx_ins_announce<2>(R, key, pr, v);
  mfx_ins_announce<3>(R, key, pr, v); mfx_ins_announce<4>(R, key, pr, v); mfx_ins_announce<5>(R, key, pr, v);
  mfx_ins_announce<6>(R, key, pr, v); mfx_ins_announce<7>(R, key, pr, v);
  unsigned long long cur[8];
  uint32_t pending = 0u;                                     // bit S: round S still has to place its key (group-uniform)
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    mfx_probe p2;
    p2.lineA = R.lineA[s]; p2.lineB = R.lineB[s];
    R.d[s] = 0u;
    R.line[s] = (uint32_t)mfx_probe_line(t, p2, 0);
    if (R.val[s]) pending |= 1u << s;
    cur[s] = mfx_slot_key_load(t, R.line[s], sub);          // eight independent requests in flight
  }
  while (pending) {                                          // diverges between the groups of a wave only
    const uint32_t at_start = pending;
    uint32_t issued = 0u;
    unsigned long long old[8];
    // ---- decide: every pending round that is the first pending one on its line
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      old[s] = 0ull;
      if (!((at_start >> s) & 1u)) continue;
      bool blocked = false;
#pragma unroll
      for (int s2 = 0; s2 < s; ++s2) blocked = blocked || (((at_start >> s2) & 1u) && R.line[s2] == R.line[s]);
      if (blocked) continue;                                 // an earlier round of the group works on this line: next pass
      const unsigned long long k64 = (unsigned long long)R.klo[s] | ((unsigned long long)R.khi[s] << 32);
      const uint32_t m_match = (uint32_t)(__ballot(cur[s] == k64) >> gsh) & 0xffu;
      const uint32_t m_empty = (uint32_t)(__ballot(cur[s] == MFX_EMPTY) >> gsh) & 0xffu;
      mfx_slot *sl = t.slots + (uint64_t)R.line[s] * MFX_SLOTS_LINE + sub;
      if (m_match) {
        if (sub == (uint32_t)__ffs((int)m_match) - 1u) atomicAdd(side ? &sl->asmV : &sl->readV, R.val[s]);
        pending &= ~(1u << s);
      } else if (m_empty && !CLAIM) {                         // the line that would hold it has room and does not: never claimed
        if (sub == 0u) ++fresh;
        pending &= ~(1u << s);
      } else if (m_empty) {
        if (sub == (uint32_t)__ffs((int)m_empty) - 1u)       // lowest empty slot
          old[s] = atomicCAS(reinterpret_cast<unsigned long long *>(&sl->key), (unsigned long long)MFX_EMPTY, k64);
        issued |= 1u << s;
      } else if (++R.d[s] >= MFX_MAX_LINES) {
        if (sub == 0) atomicAdd((unsigned long long *)&meta[2], 1ull);   // probe limit: reported as MFX_E_FULL by the host
        pending &= ~(1u << s);
      } else {
        mfx_probe p2;
        p2.lineA = R.lineA[s]; p2.lineB = R.lineB[s];
        R.line[s] = (uint32_t)mfx_probe_line(t, p2, R.d[s]);  // candidate line full without the key: the next one (read below)
      }
    }
    // ---- collect the compare-and-swap results
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (!((issued >> s) & 1u)) continue;
      const unsigned long long k64 = (unsigned long long)R.klo[s] | ((unsigned long long)R.khi[s] << 32);
      // the issuing lane = the lowest empty slot of the view the round decided on (cur[s] is unchanged since)
      const uint32_t m_empty = (uint32_t)(__ballot(cur[s] == MFX_EMPTY) >> gsh) & 0xffu;
      const bool me = sub == (uint32_t)__ffs((int)m_empty) - 1u;
      const unsigned long long was = old[s];
      const bool won = me && (was == MFX_EMPTY || was == k64);   // was == key: another group inserted the same k-mer first
      if ((uint32_t)(__ballot(won) >> gsh) & 0xffu) {
        if (won) {
          mfx_slot *sl = t.slots + (uint64_t)R.line[s] * MFX_SLOTS_LINE + sub;
          atomicAdd(side ? &sl->asmV : &sl->readV, R.val[s]);
          if (was == MFX_EMPTY) ++fresh;
        }
        pending &= ~(1u << s);
      }
    }
    // ---- the rounds still pending look at their line as it is now
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if ((pending >> s) & 1u) cur[s] = mfx_slot_key_load(t, R.line[s], sub);
  }
}

__global__ void mfx_table_init_kernel(mfx_slot *slots, uint64_t nslots) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint4 e = make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u);
  for (; i < nslots; i += stride)
    reinterpret_cast<uint4 *>(slots)[i] = e;
}

// side 0: read counts, side 1: asm counts
// MODE 1 (default): cooperative batched passes (mfx_group_insert); MODE 0: per-lane walk (mfx_claim).  Measured on
// MI355X (profiles/r02_insert_modes.txt): both -- and a third variant, cooperative read + per-lane CAS -- land within
// 10 % of each other (2^28 random keys: 11.9-12.6 G fresh inserts/s, 17-18.4 G/s when every key is already present):
// an insert is bound by what it does to the memory system (the line is read, modified by one or two L2 atomics and
// written back: 268 B of HBM traffic per key), not by how the lanes find the slot.
template <int MODE>
__global__ __launch_bounds__(256) void mfx_table_add_kernel(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                                                            int side, uint64_t *meta) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t fresh = 0, noncanon = 0;
  for (uint64_t base = blockIdx.x * (uint64_t)blockDim.x; base < n; base += stride) {     // wave-uniform trip count
    const uint64_t i = base + threadIdx.x;
    uint64_t key = 0;
    uint32_t v = 0;
    if (i < n) {
      key = kmers[i];
      if (values) v = values[i];
      else { v = (uint32_t)key & MFX_PACKED_VMASK; key >>= MFX_PACKED_VBITS; if (v == MFX_PACKED_VMASK) v = 0u; }   // packed record; escape: added separately
    }
    bool ok = v != 0;
    if (ok && (key >> (2 * t.k))) { ok = false; atomicAdd((unsigned long long *)&meta[4], 1ull); }   // wider than 2k bits: a damaged record, refused by the host
    if (ok) {
      const uint64_t krc = mfx_revcomp(key, t.k);
      if (key > krc) ++noncanon;
      if (t.shard_n > 1 && mfx_owner(t, key < krc ? key : krc, key < krc ? krc : key, t.shard_n) != t.shard_rank)
        ok = false;                                          // another rank owns this k-mer
    }
    if (MODE == 1) mfx_group_insert(t, key, ok ? v : 0u, side, meta, fresh);
    else if (ok) {
      mfx_slot *sl = mfx_claim(t, key, meta, fresh);
      if (sl) atomicAdd(side ? &sl->asmV : &sl->readV, v);
    }
  }
  mfx_meta_flush(meta, fresh, noncanon);
}

// UB k-mers of every lane go into the table (v == 0: none), all lanes of the wave together.
//  * SEQUENCE-ONLY index: the key set is frozen (the k-mers claimed from the sequence), an add finds its k-mer's slot and
//    updates the count, or drops the k-mer (T.dropped -> meta[3]).  Only canonical k-mers were claimed: a non-canonical
//    k-mer of the database is counted (meta[1]) and dropped, and the host refuses the load (value(fmer) + value(rmer) of
//    the reference would need the other strand's slot too).
//    Compact layout: ONE 16-byte load per k-mer -- its first mini-bucket, where 92 % of the claimed k-mers sit and where
//    an empty slot proves that the k-mer was never claimed (mfx_c_first); the whole line (mfx_c_find) only for the rest.
//    The UB loads of a lane are in flight together.  16-byte slots: the cooperative insert without its claim.
//  * full tables: the cooperative insert (mfx_group_insert); a sharded table keeps the k-mers it owns.
struct mfx_tally { uint32_t fresh = 0, dropped = 0, noncanon = 0, wide = 0; };

__device__ __forceinline__ void mfx_tally_flush(uint64_t *meta, const mfx_tally &T) {
  uint64_t f = T.fresh, d = T.dropped, c = T.noncanon;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { f += __shfl_down(f, o, 64); d += __shfl_down(d, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63u) == 0) {
    if (f) atomicAdd((unsigned long long *)&meta[0], (unsigned long long)f);
    if (c) atomicAdd((unsigned long long *)&meta[1], (unsigned long long)c);
    if (d) atomicAdd((unsigned long long *)&meta[3], (unsigned long long)d);
  }
  if (T.wide) atomicAdd((unsigned long long *)&meta[4], (unsigned long long)T.wide);    // a damaged database only: the host refuses the load
}

// prp: the probes of the keys when the caller knows them without the minimizer scan (the quotient form's pieces are all in a PLACED
// database's record: mfx_home_placed), else nullptr.  plain: the counts are written with plain stores (the placed update, see below).
template <int UB>
__device__ __forceinline__ void mfx_apply_batch(const mfx_table_view &t, uint64_t (&key)[UB], uint32_t (&v)[UB], int side, uint64_t *meta,
                                                mfx_tally &T, const mfx_probe *prp = nullptr, bool plain = false) {
  // a k-mer has 2k bits: anything wider is a damaged record (mfx_db.cpp checks what it can see on the host; the k-mers of a
  // delta-coded block only exist here) -- never inserted, counted in meta[4], the host refuses the load (index_check)
#pragma unroll
  for (int j = 0; j < UB; ++j)
    if (v[j] && (key[j] >> (2 * t.k))) { ++T.wide; v[j] = 0u; }
  if (t.seq_only && t.compact) {
    unsigned long long *mb[UB];
    uint4 s[UB];
    mfx_probe pr[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (v[j] && key[j] > mfx_revcomp(key[j], t.k)) { ++T.noncanon; v[j] = 0u; }
      pr[j] = prp ? prp[j] : mfx_home(t, key[j]);
      mb[j] = reinterpret_cast<unsigned long long *>(t.slots) + mfx_probe_line(t, pr[j], 0) * MFX_CSLOTS_LINE + 2u * pr[j].b0;
      s[j] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
      if (v[j]) s[j] = *reinterpret_cast<const uint4 *>(mb[j]);
    }
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (v[j] == 0u) continue;
      const uint64_t x = (uint64_t)s[j].x | ((uint64_t)s[j].y << 32), y = (uint64_t)s[j].z | ((uint64_t)s[j].w << 32);
      unsigned long long *w = nullptr;
      unsigned long long cur = 0;
      bool beyond = false;
      if (x == MFX_EMPTY) { }                                  // first slot of its order empty: never claimed
      else if ((x >> 22) == pr[j].fkey) { w = mb[j]; cur = x; }
      else if (y == MFX_EMPTY) { }
      else if ((y >> 22) == pr[j].fkey) { w = mb[j] + 1; cur = y; }
      else w = mfx_c_find(t, pr[j], 0, cur, beyond);
      if (w && plain) {
        // A PLACED database's update writes the slot with a plain store.  A slot has ONE writer while a database is applied -- a
        // database holds every k-mer once, the loads of an index do not overlap, the claim kernel is ordered before them -- and an
        // atomic on this device is a 64-byte transaction at the memory side whatever the L2 holds (WRITE_SIZE: 64 B per updated
        // k-mer, sorted or placed), while a line whose slots are stored to one after the other is written back once.  A count that
        // would saturate its field takes the compare-and-swap path (it moves to the side table with it).
        const int sh = side ? 0 : 11;
        const uint32_t f = (uint32_t)(cur >> sh) & MFX_CSAT;
        if (f != MFX_CSAT && (uint64_t)f + v[j] < MFX_CSAT) { *w = cur + ((unsigned long long)v[j] << sh); continue; }
      }
      if (w) mfx_c_add(t, w, cur, key[j], v[j], side, meta);
      else {
        mfx_slot *ss = beyond ? mfx_find_slot(mfx_side_view(t), key[j]) : nullptr;     // quotient form: beyond its candidate lines
        if (ss) atomicAdd(side ? &ss->asmV : &ss->readV, v[j]); else ++T.dropped;
      }
    }
  } else if (t.seq_only) {
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (v[j] && key[j] > mfx_revcomp(key[j], t.k)) { ++T.noncanon; v[j] = 0u; }
      mfx_group_insert<false>(t, key[j], v[j], side, meta, T.dropped);
    }
  } else {
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      if (v[j]) {
        const uint64_t krc = mfx_revcomp(key[j], t.k);
        if (key[j] > krc) ++T.noncanon;
        if (t.shard_n > 1 && mfx_owner(t, key[j] < krc ? key[j] : krc, key[j] < krc ? krc : key[j], t.shard_n) != t.shard_rank)
          v[j] = 0u;                                           // another rank owns this k-mer
      }
      mfx_group_insert<true>(t, key[j], v[j], side, meta, T.fresh);
    }
  }
}

#ifndef MFX_UPD_BATCH
#define MFX_UPD_BATCH 4
#endif
// loads into a sequence-only index: MFX_UPD_BATCH k-mers per lane and pass
__global__ __launch_bounds__(256) void mfx_table_update_kernel(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                                                               int side, uint64_t *meta) {
  constexpr int UB = MFX_UPD_BATCH;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * UB;
  mfx_tally T;
  for (uint64_t base = (uint64_t)blockIdx.x * blockDim.x * UB; base < n; base += stride) {     // wave-uniform trip count
    uint64_t key[UB];
    uint32_t v[UB];
#pragma unroll
    for (int j = 0; j < UB; ++j) {
      const uint64_t i = base + (uint64_t)j * blockDim.x + threadIdx.x;
      key[j] = 0; v[j] = 0u;
      if (i < n) {
        key[j] = kmers[i];
        if (values) v[j] = values[i];
        else { v[j] = (uint32_t)key[j] & MFX_PACKED_VMASK; key[j] >>= MFX_PACKED_VBITS; if (v[j] == MFX_PACKED_VMASK) v[j] = 0u; }   // packed record; escape: added separately
      }
    }
    mfx_apply_batch<UB>(t, key, v, side, meta, T);
  }
  mfx_tally_flush(meta, T);
}

// ---------------------------------------------------------------------------
// Delta-coded blocks of a sorted database (this repo's flat format, mfx_db.cpp FLAT_DELTA): MFX_DELTA_BLOCK k-mers per
// block as {first k-mer; (count - 1) differences of kbits bits; count values of vbits bits, all ones = escape} -- 2.5-3
// bytes per k-mer of a 30x human read set instead of 8 on disk, in the staging lanes and over PCIe.  One workgroup per
// block: a lane decodes 16 consecutive entries (sum of its differences, workgroup scan, then the entries four at a time)
// and puts them into the table itself -- the decoded k-mers never exist in memory.
// dir[b] = {first k-mer, payload byte offset (48 bits) | kbits << 48 | vbits << 56}; dir[nblocks] closes the last block.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mfx_bits_at(const uint64_t *w, uint64_t bit, uint32_t nbits) {     // nbits <= 63
  const uint64_t i = bit >> 6;
  const uint32_t sh = (uint32_t)bit & 63u;
  uint64_t x = w[i] >> sh;
  if (sh + nbits > 64u) x |= w[i + 1] << (64u - sh);
  return x & ((1ull << nbits) - 1ull);
}

__global__ __launch_bounds__(256) void mfx_table_add_delta_kernel(mfx_table_view t, const uint64_t *payload, const uint64_t *dir,
                                                                  uint32_t nblocks, uint64_t n, uint64_t payload_base, int side,
                                                                  uint64_t *meta) {
  constexpr int PER = MFX_DELTA_BLOCK / 256;                   // entries per lane
  static_assert(PER == 16, "a lane decodes 16 entries, four at a time");
  __shared__ uint64_t wsum[4];
  mfx_tally T;
  const uint32_t tid = threadIdx.x, wv = tid >> 6, ln = tid & 63u;
  for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const uint64_t first = dir[2 * (uint64_t)b], info = dir[2 * (uint64_t)b + 1];
    const uint32_t kb = (uint32_t)(info >> 48) & 0xffu, vb = (uint32_t)(info >> 56) & 0xffu;
    const uint64_t left = n - (uint64_t)b * MFX_DELTA_BLOCK;
    const uint32_t cnt = left < MFX_DELTA_BLOCK ? (uint32_t)left : (uint32_t)MFX_DELTA_BLOCK;
    const uint64_t *pw = payload + (((info & 0xffffffffffffull) - payload_base) >> 3);
    const uint64_t vbit0 = (((uint64_t)(cnt - 1u) * kb + 63u) >> 6) << 6;      // the values start at a word boundary
    const uint32_t e0 = tid * PER;
    // ---- the sum of this lane's differences (entry e > 0 has difference e - 1; entry 0 is the block's first k-mer)
    uint64_t mine = 0;
#pragma unroll 4
    for (uint32_t i = 0; i < PER; ++i) {
      const uint32_t e = e0 + i;
      if (e > 0u && e < cnt && kb) mine += mfx_bits_at(pw, (uint64_t)(e - 1u) * kb, kb);
    }
    uint64_t inc = mine;                                       // inclusive scan over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o, 64); if ((int)ln >= o) inc += u; }
    __syncthreads();                                           // wsum of the previous block is read by now
    if (ln == 63u) wsum[wv] = inc;
    __syncthreads();
    uint64_t run = first + inc - mine;
    for (uint32_t w2 = 0; w2 < wv; ++w2) run += wsum[w2];
    // ---- the entries, four at a time
#pragma unroll 1
    for (uint32_t g = 0; g < PER; g += 4) {
      uint64_t key[4];
      uint32_t v[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t e = e0 + g + i;
        key[i] = 0; v[i] = 0u;
        if (e < cnt) {
          if (e > 0u && kb) run += mfx_bits_at(pw, (uint64_t)(e - 1u) * kb, kb);
          key[i] = run;
          v[i] = (uint32_t)mfx_bits_at(pw, vbit0 + (uint64_t)e * vb, vb);
          if (v[i] == (1u << vb) - 1u) v[i] = 0u;              // escape: added separately (the file's escape list)
        }
      }
      mfx_apply_batch<4>(t, key, v, side, meta, T);
    }
  }
  mfx_tally_flush(meta, T);
}

// The probe of a k-mer whose placement pieces are known (a record of a PLACED database: mfx_place.h) -- no t-mer scan, no window
// choice: the line and the first mini-bucket come from `top`, window and strand, the quotient form's key field from the same pieces.
// Valid for the QUOTIENT form of the compact layout (22 <= k <= 31) under its default placement; the direct form (k <= 21) takes its
// first mini-bucket from the t-mer's offset, which a record does not carry (mfx_mod_place).
__device__ __forceinline__ mfx_probe mfx_home_placed(const mfx_table_view &t, uint64_t key, uint32_t top, uint32_t hi, uint32_t meta) {
  mfx_probe pr;
  const uint32_t nl = (uint32_t)t.nlines, j = (meta >> 1) & 3u;
  pr.lineA = pr.lineB = __umulhi(top, nl);
  pr.b0 = mfx_p_bucket(top, j, meta & 1u);
  pr.fkey = key;
  if (t.quot) {
    const int R = 2 * (t.k - 3) - 32, Q = 32 - t.qshift;
    const uint32_t fq = (top * nl) >> t.qshift;
    pr.fkey = (uint64_t)hi | ((uint64_t)fq << R) | ((uint64_t)meta << (R + Q));
  }
  return pr;
}

// ... of the DIRECT form (k <= 21): the line is `top`'s, the first mini-bucket wants the sampling t-mer's offset x, which the record does not
// carry -- but it carries the window j = x mod 4, and the smallest t-mer of the k-mer (ties: the leftmost) is then the smallest among the
// offsets j, j + 4, j + 8, ...: four order hashes instead of the sixteen of the whole scan (mfx_mod_window).
__device__ __forceinline__ mfx_probe mfx_home_placed_direct(const mfx_table_view &t, uint64_t key, uint32_t top, uint32_t meta) {
  mfx_probe pr;
  const int k = t.k, tl = t.mz_t;
  const uint64_t rc = mfx_revcomp(key, k);
  const uint32_t tmask = (1u << (2 * tl)) - 1u, j = (meta >> 1) & 3u;
  uint32_t best = 0xffffffffu, x = j;
  for (int p = (int)j; p + tl <= k; p += MFX_PLACE_W) {
    const uint32_t a = (uint32_t)(key >> (2 * (k - tl - p))) & tmask, b = (uint32_t)(rc >> (2 * p)) & tmask;
    const uint32_t o = mfx_p_tmer_order(a < b ? a : b);
    if (o < best) { best = o; x = (uint32_t)p; }
  }
  pr.lineA = __umulhi(top, (uint32_t)t.nlines);
  pr.lineB = mfx_range32(mfx_hash64(key), t.nlines);          // (candidate lines beyond the minimizer's region follow the k-mer's own hash, as mfx_home's)
  pr.b0 = (x + (top >> 3)) & 7u;
  pr.fkey = key;
  return pr;
}

// The delta-coded blocks of a PLACED database (mfx_db.cpp FLAT_PLACED): the records are the numbers P of mfx_place.h in ascending
// order, i.e. in the order of the table's lines -- a block of 4096 records touches a few hundred CONSECUTIVE lines, each of which is
// read from HBM once, updated in the L2 and written back once, where a k-mer-sorted database reads a random line per record
// (1.09 line reads + 0.245 write-backs per record, profiles/r05_build_kernels_pmc.txt).  Same decode as mfx_table_add_delta_kernel;
// placed != 0: the table takes the records' own placement (compact layout, default placement); else the k-mer is placed anew.
__global__ __launch_bounds__(256) void mfx_table_add_placed_kernel(mfx_table_view t, const uint64_t *payload, const uint64_t *dir,
                                                                   uint32_t nblocks, uint64_t n, uint64_t payload_base, int side,
                                                                   uint64_t *meta, int placed) {
  // Records in the order of the table's lines only pay if a line's records are applied TOGETHER: a round takes 1024 consecutive
  // records (four per lane, lane after lane), i.e. ~100 consecutive lines, and never comes back to them -- with sixteen consecutive
  // records per lane (the k-mer-sorted kernel's split) every line of the block was touched in four rounds, and left the L2 in between
  // (measured: every table line written back 4.7 times, profiles/r05_e2e_placed.txt).
  constexpr uint32_t ROUND = 256u * 4u;
  const bool split = mfx_p_split(t.k);
  __shared__ uint64_t wsum[4];
  __shared__ uint64_t s_base;
  mfx_tally T;
  const uint32_t tid = threadIdx.x, wv = tid >> 6, ln = tid & 63u;
  for (uint32_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
    const uint64_t first = dir[2 * (uint64_t)b], info = dir[2 * (uint64_t)b + 1];
    const uint32_t kb = (uint32_t)(info >> 48) & 0xffu, vb = (uint32_t)(info >> 56) & 0xffu;
    const uint64_t left = n - (uint64_t)b * MFX_DELTA_BLOCK;
    const uint32_t cnt = left < MFX_DELTA_BLOCK ? (uint32_t)left : (uint32_t)MFX_DELTA_BLOCK;
    const uint64_t *pw = payload + (((info & 0xffffffffffffull) - payload_base) >> 3);
    const uint64_t vbit0 = (((uint64_t)(cnt - 1u) * kb + 63u) >> 6) << 6;
    __syncthreads();                                           // the previous block's rounds are over
    if (tid == 0) s_base = first;
    for (uint32_t r0 = 0; r0 < cnt; r0 += ROUND) {             // block-uniform
      const uint32_t e0 = r0 + tid * 4u;
      uint64_t d[4], mine = 0;
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t e = e0 + i;
        d[i] = (e > 0u && e < cnt && kb) ? mfx_bits_at(pw, (uint64_t)(e - 1u) * kb, kb) : 0ull;     // entry e > 0 has difference e - 1
        mine += d[i];
      }
      uint64_t inc = mine;                                     // inclusive scan over the workgroup
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint64_t u = __shfl_up(inc, o, 64); if ((int)ln >= o) inc += u; }
      __syncthreads();                                         // s_base and wsum of the round before are read by now
      if (ln == 63u) wsum[wv] = inc;
      __syncthreads();
      uint64_t run = s_base + inc - mine;
      for (uint32_t w2 = 0; w2 < wv; ++w2) run += wsum[w2];
      __syncthreads();
      if (tid == 255u) s_base = run + mine;                    // the value of the round's last entry: where the next round starts
      uint64_t key[4];
      uint32_t v[4];
      mfx_probe pr[4];
#pragma unroll
      for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t e = e0 + i;
        key[i] = 0; v[i] = 0u;
        pr[i].lineA = pr[i].lineB = 0u; pr[i].b0 = 0u; pr[i].fkey = 0;
        run += d[i];
        if (e < cnt) {
          uint32_t top, hi, pm;
          v[i] = (uint32_t)mfx_bits_at(pw, vbit0 + (uint64_t)e * vb, vb);
          if (v[i] == (1u << vb) - 1u) v[i] = 0u;              // escape: added separately (the file's escape list)
          uint32_t sb = 0u;
          if (split) { sb = v[i] & 1u; v[i] >>= 1; }           // k = 31: the strand bit of P travels in the count field (mfx_place.h)
          key[i] = mfx_p_decode_s(t.k, run, sb, top, hi, pm);
          if (!split && (run >> mfx_p_bits(t.k))) { if (v[i]) ++T.wide; v[i] = 0u; }      // (a damaged record: wider than any P of this k)
          // ... or one that no converter writes: a number equal to the one before it (the records ascend strictly), a number whose k-mer is not
          // canonical (every P of a canonical k-mer decodes to it; most other numbers do not).  Counted with the wide ones (MFX_E_FORMAT), not applied:
          // the plain stores below rely on one record per slot.
          if (v[i] && ((e > 0u && d[i] == 0ull && !(split && sb)) || mfx_p_revcomp(key[i], t.k) < key[i])) { ++T.wide; v[i] = 0u; }   // (k = 31: the s = 1 twin of a stored number follows it)
          if (placed) pr[i] = t.quot ? mfx_home_placed(t, key[i], top, hi, pm) : mfx_home_placed_direct(t, key[i], top, pm);
        }
      }
      mfx_apply_batch<4>(t, key, v, side, meta, T, placed ? pr : nullptr, placed != 0);
    }
  }
  mfx_tally_flush(meta, T);
}

// P of every k-mer of an array (the converter of this repo's tools sorts a database by it on the device; mfx_db.cpp does the same on the host)
__global__ void mfx_place_keys_kernel(int k, const uint64_t *kmers, uint64_t n, uint64_t *out) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint64_t key = kmers[i], rc = mfx_p_revcomp(key, k);
    out[i] = mfx_p_encode(k, key < rc ? key : rc);
  }
}

__global__ void mfx_table_value_kernel(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint2 v = t.compact ? mfx_c_lookup(t, kmers[i]) : mfx_lookup(t, kmers[i]);
    readV[i] = v.x;
    asmV[i] = v.y;
  }
}

__global__ void mfx_table_export_kernel(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV,
                                        unsigned long long *count) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  if (t.compact) {                                               // 8-byte slots; raw counts (no read filter), as below
    mfx_table_view raw = t;
    raw.minV = 0u; raw.maxV = 0xffffffffu;
    const uint64_t *cs = reinterpret_cast<const uint64_t *>(t.slots);
    for (; i < t.nlines * MFX_CSLOTS_LINE; i += stride) {
      const uint64_t x = cs[i];
      if (x == MFX_EMPTY) continue;
      uint64_t km = x >> 22;
      if (t.quot) {                                              // (line, key field) -> k-mer: the slot sits in candidate line d of its home
        const uint32_t d = (uint32_t)(km >> MFX_Q_DSHIFT), line = (uint32_t)(i / MFX_CSLOTS_LINE);
        const uint32_t home = line >= d ? line - d : line + (uint32_t)t.nlines - d;
        km = mfx_q_invert(t, home, km & ((1ull << MFX_Q_DSHIFT) - 1ull));
      }
      const uint2 v = mfx_c_fields(raw, km, (uint32_t)x);
      unsigned long long w = atomicAdd(count, 1ull);
      kmers[w] = km;
      readV[w] = v.x;
      asmV[w] = v.y;
    }
    if (t.quot) {
      // quotient form: the k-mers beyond their candidate lines live in the side table only (its other entries are the exact
      // counts of saturated fields, whose k-mers were listed above)
      const mfx_table_view sv = mfx_side_view(t);
      for (i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < sv.nlines * MFX_SLOTS_LINE; i += stride) {
        const mfx_slot sl = sv.slots[i];
        if (sl.key == MFX_EMPTY) continue;
        unsigned long long word = 0;
        bool beyond;
        if (mfx_c_find(t, mfx_home(t, sl.key), 0, word, beyond) || !beyond) continue;
        unsigned long long w = atomicAdd(count, 1ull);
        kmers[w] = sl.key;
        readV[w] = sl.readV;
        asmV[w] = sl.asmV;
      }
    }
    return;
  }
  uint64_t nslots = t.nlines * MFX_SLOTS_LINE;
  for (; i < nslots; i += stride) {
    mfx_slot s = t.slots[i];
    if (s.key == MFX_EMPTY) continue;
    unsigned long long w = atomicAdd(count, 1ull);
    kmers[w] = s.key;
    readV[w] = s.readV;
    asmV[w] = s.asmV;
  }
}


// ---------------------------------------------------------------------------
// Wave-cooperative lookup.  A 128-byte table line is what HBM delivers for any
// access into it (measured: profiles/r01_ubench_gather.txt), so a query should
// look at the WHOLE line at once.  Each 8-lane group of the wave fetches the 8
// slots of one query's home line with one coalesced 128-byte access (lane
// `sub` reads slot `sub`) and the lane holding the matching slot hands the counts
// to the owning lane through an LDS mailbox.  Per wave-instruction 8 lines are
// requested; 8 owners per group are served per round, so 64 lines are in flight
// per wave.  Queries whose home line is full without a match continue on their
// next candidate line in a second, compacted cooperative pass; the few left after
// that finish per lane.
// ---------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ uint32_t mfx_group_bcast(uint32_t v) {
  // lane' = (lane & 0x18) | S inside each 32-lane half: broadcast of sub-lane S in every 8-lane group
  return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x18 | (S << 5));
}

// Result hand-off goes through a per-wave LDS mailbox (one 16-byte record per
// lane/owner): the lane that holds the matching slot writes {readV, asmV,
// found} straight into the owner's record.  LDS requests of one wave are served in
// order, so the owner's later read needs no barrier.  This replaces three
// cross-lane permutes + ballot decoding per served query with one predicated
// store.  ("The line has room" does not go through LDS in the first pass: mfx_group_room.)
// Hand-off points of the mailbox protocol.  The hardware serves one wave's LDS requests in order; this keeps the
// COMPILER from moving an LDS access across a hand-off (wavefront-scope fence + scheduling barrier: no instructions).
__device__ __forceinline__ void mfx_wave_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct mfx_mailbox {
  uint4 rec[MFX_BLOCK];      // x = readV, y = asmV, z = found; w = "candidate line has room" in the second pass only
};

// Issue phase of a round, in two steps so that nothing serialises: first ALL the
// cross-lane broadcasts (32 LDS-crossbar ops back to back), then ALL 8 loads back
// to back.  A position without a k-mer still "owns" a query -- of line 0, whose
// outcome its lane ignores (ok[j] is false) -- so the hot sequence has no
// exec-mask branches and no dead-owner bookkeeping.
// What is broadcast is the line's ADDRESS (two words), computed once by the owner: a slot lane then needs one
// 32-bit OR for its own slot's address (lines are 128-byte aligned) instead of a 64-bit shift-and-add per load.
template <int S>
__device__ __forceinline__ void mfx_group_announce(uint32_t (&alo)[8], uint32_t (&ahi)[8], uint32_t (&klo)[8], uint32_t (&khi)[8],
                                                   uint64_t line_addr, uint32_t key_lo, uint32_t key_hi) {
  alo[S] = mfx_group_bcast<S>((uint32_t)line_addr);
  ahi[S] = mfx_group_bcast<S>((uint32_t)(line_addr >> 32));
  klo[S] = mfx_group_bcast<S>(key_lo);
  khi[S] = mfx_group_bcast<S>(key_hi);
}

typedef uint32_t mfx_u32x4 __attribute__((ext_vector_type(4)));

// The slot loads are written as instructions: left to the compiler, a slot whose value half is only used inside
// the "key matches" branch gets its load split -- key half here, value half as a second, dependent load behind a
// vmcnt(0) inside the branch.  So: eight global_load_dwordx4 back to back, and before slot S is looked at, a
// counted wait (vmcnt(7 - S): the S+1 oldest of the eight have landed; other loads in flight can only make
// that wait longer, never too short).
template <int S>
__device__ __forceinline__ void mfx_group_fetch(mfx_u32x4 (&v)[8], const uint32_t (&alo)[8], const uint32_t (&ahi)[8], uint32_t sub16) {
  const uint64_t p = ((uint64_t)ahi[S] << 32) | (uint64_t)(alo[S] | sub16);
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[S]) : "v"(p));
}
template <int S>
__device__ __forceinline__ void mfx_group_landed(mfx_u32x4 (&v)[8]) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[S]) : "n"(7 - S));
}

// Post phase.  The slot lane compares the FULL key: keys are unique in the table, so at
// most one lane of the group stores into the owner's record -- no two writers can ever
// interleave their words (a low-word-only pre-match would let that happen ~1e-9 per query,
// i.e. a few times per 3 Gb launch).  The record is the value pair alone, stored straight from the
// loaded registers: a k-mer found with both counts 0 reads like one not found, and value() of
// either is 0 (merfin-globals.C:84) -- it only takes the longer way there.
template <int S>
__device__ __forceinline__ void mfx_group_post(uint2 *rec, const mfx_u32x4 (&v)[8], const uint32_t (&klo)[8], const uint32_t (&khi)[8]) {
  const mfx_u32x4 s = v[S];
  if (s.x == klo[S] && s.y == khi[S]) rec[S] = make_uint2(s.z, s.w);     // an empty slot (key ~0) never equals a k-mer
}

// "The line still has room" travels without LDS: slots fill in order (mfx_claim), so a line has room
// exactly when its LAST slot is empty -- the lane with sub == 7 knows.  One ballot per round; bit 8g+7 of
// it belongs to the owner (g, S), i.e. to lane 8g+S: shifted there, the 8 rounds OR into ONE wave-wide
// mask whose bit `lane` is that lane's own answer.  Scalar work only.  (k <= 31: a k-mer's high word is
// below 2^30, so the high word alone tells an empty slot.)
template <int S>
__device__ __forceinline__ uint64_t mfx_group_room(const mfx_u32x4 (&v)[8]) {
  const uint64_t m = __ballot(v[S].y == 0xffffffffu);
  return ((m >> 7) & 0x0101010101010101ULL) << S;
}

// B queries per lane; ok[j] false = no query.  Results: rv[j], av[j].
template <int B>
__device__ __forceinline__ void mfx_group_lookup(const mfx_table_view &t, mfx_mailbox &M, const uint64_t (&key)[B],
                                                 const uint64_t (&krc)[B], const bool (&ok)[B], uint32_t (&rv)[B],
                                                 uint32_t (&av)[B]) {
  const uint32_t tid = threadIdx.x, sub16 = (tid & 7u) << 4, sub = tid & 7u;
  const uint32_t lane = tid & 63u, wbase = tid & ~63u;
  // first-pass records: 8 bytes per lane, dense in the front half of this wave's own 64 mailbox entries
  uint2 *const own = reinterpret_cast<uint2 *>(&M.rec[wbase]) + lane;
  uint2 *const grp = reinterpret_cast<uint2 *>(&M.rec[wbase]) + (lane & ~7u);
  const uint64_t slots0 = reinterpret_cast<uint64_t>(t.slots);
  uint64_t laddr[B];
  uint32_t pending[B];          // 0 resolved, 1 home line full: continue at candidate line 1
#pragma unroll
  for (int j = 0; j < B; ++j) { pending[j] = 0u; rv[j] = av[j] = 0u; }
  {
    const uint32_t fl = mfx_first_line(t, key[0], krc[0]);
    laddr[0] = slots0 + ((uint64_t)(ok[0] ? fl : 0u) << 7);       // no k-mer here: a dummy query of line 0, ignored below
  }
#pragma unroll
  for (int j = 0; j < B; ++j) {
    mfx_u32x4 v[8];
    uint32_t klo[8], khi[8], alo[8], ahi[8];
    const uint32_t key_lo = (uint32_t)key[j], key_hi = (uint32_t)(key[j] >> 32);
    *own = make_uint2(0u, 0u);
    mfx_wave_handoff();                                            // records cleared before any slot lane posts
    mfx_group_announce<0>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi); mfx_group_announce<1>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi);
    mfx_group_announce<2>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi); mfx_group_announce<3>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi);
    mfx_group_announce<4>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi); mfx_group_announce<5>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi);
    mfx_group_announce<6>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi); mfx_group_announce<7>(alo, ahi, klo, khi, laddr[j], key_lo, key_hi);
    mfx_group_fetch<0>(v, alo, ahi, sub16); mfx_group_fetch<1>(v, alo, ahi, sub16); mfx_group_fetch<2>(v, alo, ahi, sub16);
    mfx_group_fetch<3>(v, alo, ahi, sub16); mfx_group_fetch<4>(v, alo, ahi, sub16); mfx_group_fetch<5>(v, alo, ahi, sub16);
    mfx_group_fetch<6>(v, alo, ahi, sub16); mfx_group_fetch<7>(v, alo, ahi, sub16);
    // the next query's placement hash is computed HERE, under this query's eight loads
    if (j + 1 < B) {
      const uint32_t fl = mfx_first_line(t, key[j + 1], krc[j + 1]);
      laddr[j + 1] = slots0 + ((uint64_t)(ok[j + 1] ? fl : 0u) << 7);
    }
    uint64_t room = 0;
    mfx_group_landed<0>(v); mfx_group_post<0>(grp, v, klo, khi); room |= mfx_group_room<0>(v);
    mfx_group_landed<1>(v); mfx_group_post<1>(grp, v, klo, khi); room |= mfx_group_room<1>(v);
    mfx_group_landed<2>(v); mfx_group_post<2>(grp, v, klo, khi); room |= mfx_group_room<2>(v);
    mfx_group_landed<3>(v); mfx_group_post<3>(grp, v, klo, khi); room |= mfx_group_room<3>(v);
    mfx_group_landed<4>(v); mfx_group_post<4>(grp, v, klo, khi); room |= mfx_group_room<4>(v);
    mfx_group_landed<5>(v); mfx_group_post<5>(grp, v, klo, khi); room |= mfx_group_room<5>(v);
    mfx_group_landed<6>(v); mfx_group_post<6>(grp, v, klo, khi); room |= mfx_group_room<6>(v);
    mfx_group_landed<7>(v); mfx_group_post<7>(grp, v, klo, khi); room |= mfx_group_room<7>(v);
    mfx_wave_handoff();                                            // all posts of this round precede the owners' reads
    const uint2 r = *own;
    mfx_wave_handoff();                                            // ... which precede the next round's clear
    if (ok[j]) {
      if ((r.x | r.y) != 0u) {
        rv[j] = (r.x < t.minV || r.x > t.maxV) ? 0u : r.x;    // -min / -max (merfin.C:199-200)
        av[j] = r.y;
      } else {
        // not in the home line: absent if that line has room, else continue at the next candidate line
        pending[j] = ((room >> lane) & 1ULL) ? 0u : 1u;
      }
    }
  }
  // ---- second cooperative pass: queries whose home line was full continue at their
  // next candidate line.  They are compacted into this wave's 64 mailbox records
  // (ballot prefix), then served 8 per step exactly like the first pass, now with
  // the full key in the record (exact compare).  Without this, each such lane
  // fetched 8 slots on its own while its 63 neighbours waited.
  uint32_t qpos[B];
  uint32_t nq = 0;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const bool p = pending[j] == 1u;
    const uint64_t m = __ballot(p);
    const uint32_t pos = nq + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
    qpos[j] = 0xffffffffu;
    if (p && pos < 64u) {
      qpos[j] = pos;
      const uint32_t l1 = (uint32_t)mfx_probe_line(t, mfx_home(t, key[j]), 1);
      M.rec[wbase + pos] = make_uint4((uint32_t)key[j], (uint32_t)(key[j] >> 32), l1, 0u);
    }
    nq += (uint32_t)__popcll(m);
  }
  mfx_wave_handoff();                                              // compacted queries written before the slot lanes read them
  if (nq > 64u) nq = 64u;
  for (uint32_t q0 = 0; q0 < nq; q0 += 8u) {                 // wave-uniform trip count
    const uint32_t e = q0 + (lane >> 3);
    const bool live = e < nq;
    const uint4 ent = M.rec[wbase + (live ? e : 0u)];
    mfx_wave_handoff();                                            // every lane of the group has the entry before one answers into it
    uint4 sl = make_uint4(0u, 0u, 0u, 0u);
    if (live) sl = *reinterpret_cast<const uint4 *>(t.slots + (uint64_t)ent.z * MFX_SLOTS_LINE + sub);
    uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + e]);
    if (live) {
      if ((sl.x & sl.y) == 0xffffffffu) rec[3] = 1u;                         // candidate line has room
      else if (sl.x == ent.x && sl.y == ent.y) { rec[0] = sl.z; rec[1] = sl.w; rec[2] = 0xffffffffu; }   // found (marker: no line has this index)
    }
  }
  mfx_wave_handoff();                                              // answers posted before the owners read them
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (pending[j] == 0u) continue;
    uint32_t from = 1u;
    if (qpos[j] != 0xffffffffu) {
      const uint4 r = M.rec[wbase + qpos[j]];
      if (r.z == 0xffffffffu) { rv[j] = (r.x < t.minV || r.x > t.maxV) ? 0u : r.x; av[j] = r.y; continue; }
      if (r.w == 1u) continue;                                 // absent (rv = av = 0 already)
      from = 2u;                                               // that line was full too
    }
    uint2 v = mfx_scan_lines(t, key[j], mfx_home(t, key[j]), from);   // exact per-lane path (rare)
    rv[j] = v.x; av[j] = v.y;
  }
}

// the cooperative probe of mfx_group_lookup over 16-slot lines: a slot lane holds TWO slots of the line; the matching
// lane posts the slot's low word (counts + low key bits) into the owner's 4-byte record; the owner unpacks it.  A
// stored k-mer whose two counts are 0 reads like one that is not there -- value() of either is 0.
template <int S>
__device__ __forceinline__ void mfx_group_post8(uint32_t *rec, const mfx_u32x4 (&v)[8], const uint32_t (&klo)[8], const uint32_t (&khi)[8]) {
  const mfx_u32x4 s = v[S];
  // (key << 22) was broadcast: equal high words and low words that differ only in the 22 count bits (the empty word's
  // key field is not a canonical k-mer)
  if (s.y == khi[S] && ((s.x ^ klo[S]) >> 22) == 0u) rec[S] = s.x;
  if (s.w == khi[S] && ((s.z ^ klo[S]) >> 22) == 0u) rec[S] = s.z;
}
template <int S>
__device__ __forceinline__ uint64_t mfx_group_room8(const mfx_u32x4 (&v)[8]) {
  // any empty slot in the line (slots are not filled in line order: mfx_c_first); a stored high word is < 2^20
  uint64_t m = __ballot(v[S].y == 0xffffffffu || v[S].w == 0xffffffffu);
  m |= m >> 4; m |= m >> 2; m |= m >> 1;                     // bit 8g: any lane of group g
  return (m & 0x0101010101010101ULL) << S;
}

template <int B>
__device__ __forceinline__ void mfx_group_lookup8(const mfx_table_view &c, mfx_mailbox &M, const uint64_t (&key)[B],
                                                  const uint64_t (&krc)[B], const bool (&ok)[B], uint32_t (&rv)[B], uint32_t (&av)[B]) {
  const uint32_t tid = threadIdx.x, sub16 = (tid & 7u) << 4;
  const uint32_t lane = tid & 63u, wbase = tid & ~63u;
  uint32_t *const own = reinterpret_cast<uint32_t *>(&M.rec[wbase]) + lane;
  uint32_t *const grp = reinterpret_cast<uint32_t *>(&M.rec[wbase]) + (lane & ~7u);
  const uint64_t slots0 = reinterpret_cast<uint64_t>(c.slots);
  uint64_t laddr[B];
  uint32_t pending[B];          // 0 resolved, 1 home line full: continue at candidate line 1, 2 a count field is saturated
#pragma unroll
  for (int j = 0; j < B; ++j) { pending[j] = 0u; rv[j] = av[j] = 0u; }
  {
    const uint32_t fl = mfx_first_line(c, key[0], krc[0]);
    laddr[0] = slots0 + ((uint64_t)(ok[0] ? fl : 0u) << 7);
  }
#pragma unroll
  for (int j = 0; j < B; ++j) {
    mfx_u32x4 v[8];
    uint32_t klo[8], khi[8], alo[8], ahi[8];
    const uint64_t ks = key[j] << 22;
    *own = 0u;
    mfx_wave_handoff();
    mfx_group_announce<0>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32)); mfx_group_announce<1>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32));
    mfx_group_announce<2>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32)); mfx_group_announce<3>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32));
    mfx_group_announce<4>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32)); mfx_group_announce<5>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32));
    mfx_group_announce<6>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32)); mfx_group_announce<7>(alo, ahi, klo, khi, laddr[j], (uint32_t)ks, (uint32_t)(ks >> 32));
    mfx_group_fetch<0>(v, alo, ahi, sub16); mfx_group_fetch<1>(v, alo, ahi, sub16); mfx_group_fetch<2>(v, alo, ahi, sub16);
    mfx_group_fetch<3>(v, alo, ahi, sub16); mfx_group_fetch<4>(v, alo, ahi, sub16); mfx_group_fetch<5>(v, alo, ahi, sub16);
    mfx_group_fetch<6>(v, alo, ahi, sub16); mfx_group_fetch<7>(v, alo, ahi, sub16);
    if (j + 1 < B) {
      const uint32_t fl = mfx_first_line(c, key[j + 1], krc[j + 1]);
      laddr[j + 1] = slots0 + ((uint64_t)(ok[j + 1] ? fl : 0u) << 7);
    }
    uint64_t room = 0;
    mfx_group_landed<0>(v); mfx_group_post8<0>(grp, v, klo, khi); room |= mfx_group_room8<0>(v);
    mfx_group_landed<1>(v); mfx_group_post8<1>(grp, v, klo, khi); room |= mfx_group_room8<1>(v);
    mfx_group_landed<2>(v); mfx_group_post8<2>(grp, v, klo, khi); room |= mfx_group_room8<2>(v);
    mfx_group_landed<3>(v); mfx_group_post8<3>(grp, v, klo, khi); room |= mfx_group_room8<3>(v);
    mfx_group_landed<4>(v); mfx_group_post8<4>(grp, v, klo, khi); room |= mfx_group_room8<4>(v);
    mfx_group_landed<5>(v); mfx_group_post8<5>(grp, v, klo, khi); room |= mfx_group_room8<5>(v);
    mfx_group_landed<6>(v); mfx_group_post8<6>(grp, v, klo, khi); room |= mfx_group_room8<6>(v);
    mfx_group_landed<7>(v); mfx_group_post8<7>(grp, v, klo, khi); room |= mfx_group_room8<7>(v);
    mfx_wave_handoff();
    const uint32_t r = *own;
    mfx_wave_handoff();
    if (ok[j]) {
      const uint32_t r_rv = (r >> 11) & MFX_CSAT, r_av = r & MFX_CSAT;
      if ((r & 0x3fffffu) != 0u) {
        if (r_rv == MFX_CSAT || r_av == MFX_CSAT) { pending[j] = 2u; rv[j] = r; }           // the low word is parked in rv
        else { rv[j] = (r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv; av[j] = r_av; }
      } else {
        // no counts in the home line: absent (or claimed without counts) if that line has room, else the next candidate line
        pending[j] = ((room >> lane) & 1ULL) ? 0u : 1u;
      }
    }
  }
  // ---- second cooperative pass (as in mfx_group_lookup): queries whose home line was full continue at their next
  // candidate line, compacted into this wave's 64 mailbox records and served 8 per step
  uint32_t qpos[B];
  uint32_t nq = 0;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const bool p = pending[j] == 1u;
    const uint64_t m = __ballot(p);
    const uint32_t pos = nq + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
    qpos[j] = 0xffffffffu;
    if (p && pos < 64u) {
      qpos[j] = pos;
      const uint32_t l1 = (uint32_t)mfx_probe_line(c, mfx_home(c, key[j]), 1);
      const uint64_t ks = key[j] << 22;
      M.rec[wbase + pos] = make_uint4((uint32_t)ks, (uint32_t)(ks >> 32), l1, 0u);
    }
    nq += (uint32_t)__popcll(m);
  }
  mfx_wave_handoff();
  if (nq > 64u) nq = 64u;
  for (uint32_t q0 = 0; q0 < nq; q0 += 8u) {                 // wave-uniform trip count
    const uint32_t e = q0 + (lane >> 3);
    const bool live = e < nq;
    const uint4 ent = M.rec[wbase + (live ? e : 0u)];
    mfx_wave_handoff();
    uint4 sl = make_uint4(0u, 0u, 0u, 0u);
    if (live) sl = *reinterpret_cast<const uint4 *>(slots0 + ((uint64_t)ent.z << 7) + sub16);
    uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + e]);
    if (live) {
      if (sl.y == 0xffffffffu || sl.w == 0xffffffffu) rec[3] = 1u;                     // an empty slot: the line has room
      if (sl.y == ent.y && ((sl.x ^ ent.x) >> 22) == 0u) { rec[0] = sl.x; rec[2] = 0xffffffffu; }    // found (marker: no line has this index)
      else if (sl.w == ent.y && ((sl.z ^ ent.x) >> 22) == 0u) { rec[0] = sl.z; rec[2] = 0xffffffffu; }
    }
  }
  mfx_wave_handoff();
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (pending[j] == 0u) continue;
    uint32_t lo = rv[j];                                       // pending == 2: the found slot's low word
    rv[j] = 0u;
    if (pending[j] == 1u) {
      uint32_t from = 1u;
      bool have = false;
      if (qpos[j] != 0xffffffffu) {
        const uint4 r = M.rec[wbase + qpos[j]];
        if (r.z == 0xffffffffu) { lo = r.x; have = true; }
        else if (r.w == 1u) continue;                          // absent (rv = av = 0 already)
        else from = 2u;                                        // that line was full too
      }
      if (!have) {
        unsigned long long w = 0;
        bool beyond;                                                          // (this probe serves k <= 21 only: mfx_k_quot_supported)
        if (!mfx_c_find(c, mfx_home(c, key[j]), from, w, beyond)) continue;   // exact per-lane path (rare)
        lo = (uint32_t)w;
      }
    }
    const uint2 x = mfx_c_fields(c, key[j], lo);
    rv[j] = x.x; av[j] = x.y;
  }
}

// ---------------------------------------------------------------------------
// Per-lane probe of the compact layout.  Every lane looks its own queries up: ONE 16-byte load per query -- the k-mer's
// first mini-bucket -- and 92 % of the k-mers are there (load factor 0.25).  No hand-offs between lanes, no mailbox,
// no broadcasts: what the cooperative probe spends on them (~100 VALU / LDS-crossbar instructions per k-mer) is gone, and
// the memory system sees the same lines: neighbouring lanes hold neighbouring k-mers, which share their minimizer's line,
// so a wave's 64 loads fall into ~27 distinct 128-byte lines and run at the LINE rate, not at a lane rate
// (tools/ubench_locality.hip: 119 G 16-byte lane loads/s at 2.4 lanes per line, 50 G distinct lines/s either way).
// The B queries of a lane are in flight together; queries that did not end in their first mini-bucket go round by round
// through the next ones (a round = one more load for the lanes that need it, the others wait): 5.8 % need a second
// load, 1.3 % a third.  A line exhausted without the key or an empty slot is rare enough for the whole-line scan (mfx_c_find).
// ---------------------------------------------------------------------------
// the pair of a slot's low word with its saturated fields taken from (xr, xa)
__device__ __forceinline__ uint2 mfx_sat_fill(uint32_t lo, uint32_t xr, uint32_t xa) {
  const uint32_t f_rv = (lo >> 11) & MFX_CSAT, f_av = lo & MFX_CSAT;
  return make_uint2(f_rv == MFX_CSAT ? xr : f_rv, f_av == MFX_CSAT ? xa : f_av);
}

// The saturated queries of a batch (st == 0xfe, the found slot's low word in rv: the k-mers of repeat families -- a read count >= 2047
// means copy number >~ 79 at 26x): slot 0 of the k-mer's line of the side table, one 16-byte load per lane and query, all in flight --
// inside a satellite array or an rDNA unit EVERY lane of the wave has one.  A line of the side table fills from slot 0 on and holds
// 0.3 k-mers on average, so the k-mer is there (done: st = 0xff), the slot is empty (no exact count was ever moved: 0), or another
// k-mer is: st stays 0xfe and the query ends like the other rare ones (the worklist of mfx_hist_rest_kernel, or the per-lane scan).
#ifndef MFX_V_DIAG_NOSIDE
#define MFX_V_DIAG_NOSIDE 0           // DIAGNOSTIC builds only (wrong results): saturated fields taken as counts / rare endings dropped
#endif
#ifndef MFX_V_SIDE_LATE
#define MFX_V_SIDE_LATE 1             // the side probe of the undeferred form behind the passes, under the rare endings' ONE guard (0: in front of them, A/B: 3 % level 134.6 -> 138.1 G, i.i.d. 150.7 -> 151.0, 10 % level 111.6 -> 110.8; profiles/r06_repeats_ab.txt)
#endif
#ifndef MFX_V_DIAG_NOPUSH
#define MFX_V_DIAG_NOPUSH 0
#endif
template <int B, class KeyOf>
__device__ __forceinline__ void mfx_side_direct(const mfx_table_view &c, uint32_t (&st)[B], uint32_t (&rv)[B], uint32_t (&av)[B], KeyOf keyof, unsigned long long *dbg) {
  bool anysat = false;
#pragma unroll
  for (int j = 0; j < B; ++j) anysat |= st[j] == 0xfeu;
  if (!__any(anysat)) return;                                  // wave-uniform
#if MFX_V_DIAG_NOSIDE
#pragma unroll
  for (int j = 0; j < B; ++j) if (st[j] == 0xfeu) { st[j] = 0xffu; av[j] = rv[j] & MFX_CSAT; rv[j] = (rv[j] >> 11) & MFX_CSAT; }
  return;
#endif
  uint4 sv[B];
  uint64_t km[B];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    km[j] = 0;
    sv[j] = make_uint4(0u, 0u, 0u, 0u);
    if (st[j] == 0xfeu) {
      km[j] = keyof(j);
      sv[j] = *reinterpret_cast<const uint4 *>(c.side + (uint64_t)mfx_range32(mfx_hash64(km[j]), c.side_nlines) * MFX_SLOTS_LINE);
    }
  }
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (st[j] != 0xfeu) continue;
    if (dbg) atomicAdd(&dbg[2], 1ull);
    const uint64_t sk = (uint64_t)sv[j].x | ((uint64_t)sv[j].y << 32);
    if (sk == km[j] || sk == MFX_EMPTY) {
      const bool hit = sk == km[j];
      const uint2 x = mfx_sat_fill(rv[j], hit ? sv[j].z : 0u, hit ? sv[j].w : 0u);
      rv[j] = (x.x < c.minV || x.x > c.maxV) ? 0u : x.x;       // -min / -max (merfin.C:199-200)
      av[j] = x.y;
      st[j] = 0xffu;
    }
  }
  // another k-mer in slot 0: slot 1 the same way (a k-mer of a tandem array is asked for by millions of positions: what is
  // left after two slots -- 0.4 % of the side table's k-mers at its load -- takes the rare endings' way)
  bool more = false;
#pragma unroll
  for (int j = 0; j < B; ++j) more |= st[j] == 0xfeu;
  if (!__any(more)) return;                                    // wave-uniform
#pragma unroll
  for (int j = 0; j < B; ++j)
    if (st[j] == 0xfeu) sv[j] = *reinterpret_cast<const uint4 *>(c.side + (uint64_t)mfx_range32(mfx_hash64(km[j]), c.side_nlines) * MFX_SLOTS_LINE + 1);
#pragma unroll
  for (int j = 0; j < B; ++j) {
    if (st[j] != 0xfeu) continue;
    const uint64_t sk = (uint64_t)sv[j].x | ((uint64_t)sv[j].y << 32);
    if (sk == km[j] || sk == MFX_EMPTY) {
      const bool hit = sk == km[j];
      const uint2 x = mfx_sat_fill(rv[j], hit ? sv[j].z : 0u, hit ? sv[j].w : 0u);
      rv[j] = (x.x < c.minV || x.x > c.maxV) ? 0u : x.x;       // -min / -max (merfin.C:199-200)
      av[j] = x.y;
      st[j] = 0xffu;
    }
  }
}

// "no worklist": the probe's rare endings are per-lane scans (every kernel but -hist)
template <class F>
struct mfx_push_fn {
  static constexpr bool enabled = true;
  F f;
  __device__ __forceinline__ bool operator()(int j, bool want, uint64_t kmer, uint32_t aux, uint32_t mode) const { return f(j, want, kmer, aux, mode); }
};
struct mfx_no_push {
  static constexpr bool enabled = false;
  __device__ __forceinline__ bool operator()(int, bool, uint64_t, uint32_t, uint32_t) const { return false; }
};

#ifndef MFX_V_TAIL_STEPS
#define MFX_V_TAIL_STEPS 0            // cooperative steps of the probe's tail whose line loads are in flight together; 0: by the batch size (A/B: tools/ab_build.sh)
#endif
// fkey: what the key field of each query's slot holds in its home line (mfx_probe::fkey: the k-mer, or its quotient for k > 21);
// line / b0: its home line and first mini-bucket (mfx_home, or mfx_wave_mod_line for a whole wave at once); a query that is not
// ok must come with line 0 / b0 0 (a dummy load, ignored).  keyof(j): the canonical k-mer of query j -- asked for on the rare
// endings only (a saturated count; quotient form: a k-mer beyond its candidate lines), so that the quotient form keeps no
// k-mer alive across the loads: the -hist and -dump kernels re-extract it from the tile.
// push(j, want, kmer): the -hist kernel's worklist (mfx_hist_push) -- called by the whole wave; a lane that wants its query j
// ended by mfx_hist_rest_kernel gets true when the query is on the list (the caller then must not evaluate it: bit j of the
// result); mfx_no_push: there is no worklist, the rare endings are per-lane scans here and now.
// CAP: the mailbox entries (of the wave's 64) the passes may use; the queries of a batch beyond them take the rare endings' way
template <int B, class KeyOf, class Push = mfx_no_push, uint32_t CAP = 64u>
__device__ __forceinline__ uint32_t mfx_lane_lookup8(const mfx_table_view &c, mfx_mailbox &M, const uint64_t (&fkey)[B], const bool (&ok)[B],
                                                     uint32_t (&rv)[B], uint32_t (&av)[B], const uint32_t (&line)[B], const uint32_t (&b0)[B],
                                                     KeyOf keyof, unsigned long long *dbg = nullptr, Push push = Push()) {
  // dbg (the DEBUG instance of the -hist kernel only, mfx_hist_args::dbg): how many queries left the one-load fast path, and how --
  // [0] not in their first mini-bucket (first cooperative pass), [1] home line full of other k-mers (second cooperative pass),
  // [2] a saturated count (side table), [3] per-lane whole-line scans; tests assert that a world exercises every ending
  const uint32_t tid = threadIdx.x, sub16 = (tid & 7u) << 4, lane = tid & 63u, wbase = tid & ~63u;
  const uint4 *const slots0 = reinterpret_cast<const uint4 *>(c.slots);
  uint32_t st[B];               // 0xff done; 1 not in its first mini-bucket; 0xfe a count field is saturated (the slot's low word parked in rv);
                                // 0xfd / 0xfc whole-line scans from candidate line 2 / 0 (set below)
  uint4 v[B];
  // ---- first mini-bucket of every query: one 16-byte load per lane and query, all B in flight
#pragma unroll
  for (int j = 0; j < B; ++j) v[j] = slots0[((uint64_t)line[j] << 3) | b0[j]];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const uint4 s = v[j];
    // {key 42 | counts 22}: equal high words and low words that differ in the 22 count bits only (the empty word's key field is
    // no k-mer's: poly-G is not canonical, and a quotient never has d = 3)
    const uint64_t ks = fkey[j] << 22;
    const uint32_t klo = (uint32_t)ks, khi = (uint32_t)(ks >> 32);
    const bool ha = s.y == khi && ((s.x ^ klo) >> 22) == 0u, hb = s.w == khi && ((s.z ^ klo) >> 22) == 0u;
    const bool found = ha || hb, room = s.y == 0xffffffffu || s.w == 0xffffffffu;
    const uint32_t lo = ha ? s.x : s.z;
    const uint32_t r_rv = (lo >> 11) & MFX_CSAT, r_av = lo & MFX_CSAT;
    const bool sat = r_rv == MFX_CSAT || r_av == MFX_CSAT;
    const uint32_t f_rv = (r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv;      // -min / -max (merfin.C:199-200)
    rv[j] = found ? (sat ? lo : f_rv) : 0u;
    av[j] = (found && !sat) ? r_av : 0u;
    st[j] = !ok[j] ? 0xffu : (found ? (sat ? 0xfeu : 0xffu) : (room ? 0xffu : 1u));  // an empty slot before the key: absent (value 0, merfin-globals.C:84)
  }
  // ---- (A/B form only: the side probe of saturated count fields in front of the passes; the shipped form runs it behind them, where
  // ONE test per batch guards it together with the listing -- and a saturated field met by a PASS is served by it as well)
#if !MFX_V_SIDE_LATE
  mfx_side_direct<B>(c, st, rv, av, keyof, dbg);
#endif
  // ---- the queries that were not in their first mini-bucket (3 % at load factor 0.225): compacted into this wave's mailbox
  // and served 8 per step by the cooperative whole-line probe -- the 8 lanes of a group fetch the query's HOME line with one
  // coalesced request, so that whichever mini-bucket the k-mer went to, one more round trip finds it
  uint32_t qpos[B];
  uint32_t nq = 0;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const bool p = st[j] == 1u;
    const uint64_t m = __ballot(p);
    const uint32_t pos = nq + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
    qpos[j] = 0xffffffffu;
    if (p && pos < CAP) {
      qpos[j] = pos;
      const uint64_t ks = fkey[j] << 22;
      M.rec[wbase + pos] = make_uint4((uint32_t)ks, (uint32_t)(ks >> 32), line[j], 0u);
    }
    nq += (uint32_t)__popcll(m);
  }
  if (dbg && nq && lane == 0u) atomicAdd(&dbg[0], (unsigned long long)nq);
  if (nq) {                                                    // wave-uniform
    mfx_wave_handoff();
    if (nq > CAP) nq = CAP;
    // One pass over the wave's entries: 8 lanes per entry read its line (rec.z) with one coalesced request and answer into the
    // record -- found: the slot's low word and the marker; an empty slot seen: "room".  Up to MFX_TAIL_STEPS steps (8 entries
    // each) have their line loads in flight TOGETHER: a wave typically has ~25 displaced queries among its 256, i.e. four steps
    // -- one round trip.  second: only the entries an owner flagged for their next candidate line (rec.w == 2).
    constexpr uint32_t MFX_TAIL_STEPS = MFX_V_TAIL_STEPS ? MFX_V_TAIL_STEPS : (B <= 2 ? 2 : 4);   // ~6 displaced queries per batch element and wave
    auto tail_pass = [&](bool second) {
      for (uint32_t q0 = 0; q0 < nq; q0 += 8u * MFX_TAIL_STEPS) {
        uint4 sl[MFX_TAIL_STEPS];
        bool act[MFX_TAIL_STEPS];
#pragma unroll
        for (uint32_t sp = 0; sp < MFX_TAIL_STEPS; ++sp) {
          const uint32_t e = q0 + 8u * sp + (lane >> 3);
          sl[sp] = make_uint4(0u, 0u, 0u, 0u);
          act[sp] = false;
          if (e < nq) {
            const uint2 lw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint32_t *>(&M.rec[wbase + e]) + 2);   // {line, flag}
            act[sp] = !second || lw.y == 2u;
            if (act[sp]) sl[sp] = *reinterpret_cast<const uint4 *>(reinterpret_cast<uint64_t>(c.slots) + ((uint64_t)lw.x << 7) + sub16);
          }
        }
        mfx_wave_handoff();                                      // every lane has its entries' lines before one answers into a record
#pragma unroll
        for (uint32_t sp = 0; sp < MFX_TAIL_STEPS; ++sp) {
          const uint32_t e = q0 + 8u * sp + (lane >> 3);
          if (act[sp]) {
            uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + e]);
            const uint2 kk = *reinterpret_cast<const uint2 *>(rec);                          // the query's {key field << 22} words (a finder of this group may have replaced word 0: same key bits)
            if (sl[sp].y == 0xffffffffu || sl[sp].w == 0xffffffffu) rec[3] = 1u;               // an empty slot: the line has room
            if (sl[sp].y == kk.y && ((sl[sp].x ^ kk.x) >> 22) == 0u) { rec[0] = sl[sp].x; rec[2] = 0xffffffffu; }    // found (marker: no line has this index)
            else if (sl[sp].w == kk.y && ((sl[sp].z ^ kk.x) >> 22) == 0u) { rec[0] = sl[sp].z; rec[2] = 0xffffffffu; }
          }
        }
      }
      mfx_wave_handoff();
    };
    // what a pass left in the lane's own entries
    auto collect = [&](uint32_t pending, uint32_t full_code) {
#pragma unroll
      for (int j = 0; j < B; ++j) {
        if (st[j] != pending) continue;
        st[j] = 0xfcu;                                         // not served (more than 64 of them in this wave): whole-line scans from the home line
        if (qpos[j] != 0xffffffffu) {
          const uint4 r = M.rec[wbase + qpos[j]];
          if (r.z == 0xffffffffu) {
            const uint32_t r_rv = (r.x >> 11) & MFX_CSAT, r_av = r.x & MFX_CSAT;
            if (r_rv == MFX_CSAT || r_av == MFX_CSAT) { rv[j] = r.x; st[j] = 0xfeu; if (dbg) atomicAdd(&dbg[2], 1ull); }
            else { rv[j] = (r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv; av[j] = r_av; st[j] = 0xffu; }
          } else if (r.w == 1u) st[j] = 0xffu;                 // the line has room and does not hold the key: absent
          else st[j] = full_code;                              // the line is full of other k-mers
        }
      }
    };
    tail_pass(false);
    collect(1u, 2u);
    // ---- the queries whose home line is full of other k-mers: their next candidate line, the same way (most waves have none)
    bool again = false;
#pragma unroll
    for (int j = 0; j < B; ++j)
      if (st[j] == 2u) {
        uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + qpos[j]]);
        mfx_probe pr;
        pr.lineA = pr.lineB = line[j];
        rec[2] = (uint32_t)mfx_probe_line(c, pr, 1u);          // (candidate lines 0 .. MFX_MZ_REGION-1 follow the minimizer's line)
        rec[3] = 2u;
        if (c.quot) rec[1] |= 1u << (MFX_Q_DSHIFT + 22 - 32);  // quotient form: the key field of candidate line 1 (d = 1 above F0; the finder left word 1 alone)
        again = true;
        if (dbg) atomicAdd(&dbg[1], 1ull);
      }
    if (__any(again)) {
      mfx_wave_handoff();
      tail_pass(true);
      collect(2u, 0xfdu);                                      // still not there and no room: whole-line scans from candidate line 2
    }
  }
  // ---- the rare endings: a saturated count field whose exact count is not in slot 0 of its side-table line, further candidate
  // lines, (quotient form) a k-mer beyond them.  A lane that scans lines here holds its whole wave for up to eight dependent loads
  // per line -- 0.2 % of the queries of a genome with human-like repeat families, a fifth of the kernel's time (profiles/r06_repeats_ab.txt)
  // -- so the -hist kernel only LISTS such a query (push) and mfx_hist_rest_kernel ends it, every lane on an entry of its own.
  uint32_t pushed = 0u;
  bool rare = false;
#pragma unroll
  for (int j = 0; j < B; ++j) rare |= st[j] - 0xfcu <= 2u;   // 0xfc .. 0xfe
  if (!__any(rare)) return pushed;                            // wave-uniform: nine batches in ten of an i.i.d. genome end here
#if MFX_V_SIDE_LATE
  // ---- saturated count fields (met by the first load or by a pass): slots 0 and 1 of the side table's line, per lane; what is not
  // there stays 0xfe
  mfx_side_direct<B>(c, st, rv, av, keyof, dbg);
#endif
  if (Push::enabled) {
    {
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const bool want = st[j] - 0xfcu <= 2u;
        if (!__any(want)) continue;
        if (dbg && want) atomicAdd(&dbg[st[j] == 0xfeu ? 4 : 3], 1ull);
#if MFX_V_DIAG_NOPUSH
        if (want) { st[j] = 0xffu; rv[j] = av[j] = 0u; }
        continue;
#endif
        // (a saturated one: the main slot is known -- its low word rides in the entry, the side table's line is all that is left to read)
        if (push(j, want, want ? keyof(j) : 0ull, st[j] == 0xfeu ? rv[j] : line[j], st[j] == 0xfdu ? 1u : st[j] == 0xfeu ? 2u : 0u) && want) { pushed |= 1u << j; st[j] = 0xffu; rv[j] = av[j] = 0u; }
      }
    }
  }
  // ... or (no list, or the list is full) ONE instance of the scans' code for all B queries of the lane
  while (true) {
    int sj = -1;
    uint64_t sfkey = 0;
    uint32_t slo = 0, scode = 0, sline = 0;
#pragma unroll
    for (int j = 0; j < B; ++j)
      if (sj < 0 && st[j] >= 0xfcu && st[j] <= 0xfeu) { sj = j; sfkey = fkey[j]; slo = rv[j]; scode = st[j]; sline = line[j]; }
    if (!__any(sj >= 0)) break;
    if (sj >= 0) {
      uint2 x = make_uint2(0u, 0u);
      bool have = scode == 0xfeu, beyond = false;
      if (dbg) atomicAdd(&dbg[Push::enabled ? 5 : (have ? 4 : 3)], 1ull);
      if (!have) {
        const uint2 fd = mfx_c_find_lean(c, sfkey, sline, scode == 0xfdu ? 2u : 0u, beyond);
        have = fd.x != 0u;
        slo = fd.y;
      }
      uint32_t r_rv = (slo >> 11) & MFX_CSAT, r_av = slo & MFX_CSAT;
      const bool sat = have && (r_rv == MFX_CSAT || r_av == MFX_CSAT);
      if (sat || beyond) {                                     // the side table is keyed by the k-mer itself
        const uint2 sx = mfx_side_lookup_lean(c, keyof(sj));
        if (beyond) { r_rv = sx.x; r_av = sx.y; have = true; }
        else { if (r_rv == MFX_CSAT) r_rv = sx.x; if (r_av == MFX_CSAT) r_av = sx.y; }
      }
      if (have) x = make_uint2((r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv, r_av);      // -min / -max (merfin.C:199-200)
#pragma unroll
      for (int j = 0; j < B; ++j)
        if (sj == j) { rv[j] = x.x; av[j] = x.y; st[j] = 0xffu; }
    }
  }
  return pushed;
}

// ---------------------------------------------------------------------------
// The same probe with its tail DEFERRED (the -hist kernel; MFX_V_DEFER).  mfx_lane_lookup8 ends every batch with the cooperative
// pass over the few queries that were not in their first mini-bucket (~4 of a wave's 128 at load factor 0.18) -- two hand-offs, a
// ballot prefix, the line loads, the answers, the collection: ~150 wave instructions that cost the same for 4 entries as for 64,
// a quarter of what the wave issues per batch.  Here a batch only PARKS such a query in the wave's mailbox -- {key field << 22,
// home line, tile position} -- and goes on; the mailbox is worked off when it holds MFX_DEFER_FLUSH entries or the tile ends
// (mfx_lane_flush): one cooperative pass over up to 64 entries (8 lanes read an entry's line with one request, four steps in
// flight), then LANE e CONSUMES ENTRY e -- second candidate line, side table, further lines as before -- and evaluates it (K*, bin,
// counters) itself, all lanes busy.  Whose lane a k-mer's result lands in does not matter: the counters are integers summed over
// the block, and koverCpy of a (tile, wave) is an integer sum as well (units of 2^-52, mfx_hist_kernel), so the result does not
// depend on which queries happened to be displaced -- that is decided by the races of the table's build.
// ---------------------------------------------------------------------------
#ifndef MFX_V_DEFER
#define MFX_V_DEFER 1
#endif
#ifndef MFX_V_DEFER_FLUSH
#define MFX_V_DEFER_FLUSH 32          // entries in the wave's mailbox (of 64) from which on the next batch boundary flushes it
#endif
constexpr uint32_t MFX_DEFER_FLUSH = MFX_V_DEFER_FLUSH;
constexpr uint32_t MFX_REC_FOUND = 0xffffffffu;                // rec.z of an answered entry (no line has this index)

// phase A: the batch's loads and the one-load answers.  defer: bit j set = query j was parked (its rv / av come at the flush).
template <int B, class KeyOf, class Push = mfx_no_push, uint32_t CAP = 64u>
__device__ __forceinline__ uint32_t mfx_lane_probe_defer(const mfx_table_view &c, mfx_mailbox &M, uint32_t &nq, const uint64_t (&fkey)[B], const bool (&ok)[B],
                                                         uint32_t (&rv)[B], uint32_t (&av)[B], const uint32_t (&line)[B], const uint32_t (&b0)[B],
                                                         const uint32_t (&pos)[B], KeyOf keyof, unsigned long long *dbg = nullptr, Push push = Push()) {
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
  const uint4 *const slots0 = reinterpret_cast<const uint4 *>(c.slots);
  uint32_t st[B];               // 0xff done; 1 parked; 0xfe a count field is saturated (the slot's low word in rv); 0xfc whole-line scans from the home line
  uint4 v[B];
#pragma unroll
  for (int j = 0; j < B; ++j) v[j] = slots0[((uint64_t)line[j] << 3) | b0[j]];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const uint4 s = v[j];
    const uint64_t ks = fkey[j] << 22;
    const uint32_t klo = (uint32_t)ks, khi = (uint32_t)(ks >> 32);
    const bool ha = s.y == khi && ((s.x ^ klo) >> 22) == 0u, hb = s.w == khi && ((s.z ^ klo) >> 22) == 0u;
    const bool found = ha || hb, room = s.y == 0xffffffffu || s.w == 0xffffffffu;
    const uint32_t lo = ha ? s.x : s.z;
    const uint32_t r_rv = (lo >> 11) & MFX_CSAT, r_av = lo & MFX_CSAT;
    const bool sat = r_rv == MFX_CSAT || r_av == MFX_CSAT;
    const uint32_t f_rv = (r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv;      // -min / -max (merfin.C:199-200)
    rv[j] = found ? (sat ? lo : f_rv) : 0u;
    av[j] = (found && !sat) ? r_av : 0u;
    st[j] = !ok[j] ? 0xffu : (found ? (sat ? 0xfeu : 0xffu) : (room ? 0xffu : 1u));  // an empty slot before the key: absent (value 0, merfin-globals.C:84)
  }
  uint32_t defer = 0u;
#pragma unroll
  for (int j = 0; j < B; ++j) {
    const bool p = st[j] == 1u;
    const uint64_t m = __ballot(p);
    if (m) {                                                    // wave-uniform
      const uint32_t at = nq + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
      if (p) {
        if (at < CAP) {
          const uint64_t ks = fkey[j] << 22;
          M.rec[wbase + at] = make_uint4((uint32_t)ks, (uint32_t)(ks >> 32), line[j], pos[j] << 4);
          defer |= 1u << j;
        } else st[j] = 0xfcu;                                   // no room in the mailbox (more than 64 parked queries in this wave): scanned here and now
      }
      const uint32_t cnt = (uint32_t)__popcll(m);
      if (dbg && lane == 0u) atomicAdd(&dbg[0], (unsigned long long)cnt);
      nq = nq + cnt < CAP ? nq + cnt : CAP;
    }
  }
  // the rare endings that cannot wait (the k-mer itself is at hand only here): a saturated count whose exact value is not in slot 0
  // of its side-table line, a query the mailbox had no room for -- onto the -hist kernel's worklist (mfx_lane_lookup8), else per lane
  bool rare = false;
#pragma unroll
  for (int j = 0; j < B; ++j) rare |= (st[j] & 0xfdu) == 0xfcu;       // 0xfc, 0xfe
  if (!__any(rare)) return defer;                              // wave-uniform: ONE test per batch guards everything below
  mfx_side_direct<B>(c, st, rv, av, keyof, dbg);               // saturated count fields: slots 0 / 1 of the side table's line, per lane
  if (Push::enabled) {
    {
#pragma unroll
      for (int j = 0; j < B; ++j) {
        const bool want = st[j] == 0xfcu || st[j] == 0xfeu;
        if (!__any(want)) continue;
        if (dbg && want) atomicAdd(&dbg[st[j] == 0xfeu ? 4 : 3], 1ull);
        if (push(j, want, want ? keyof(j) : 0ull, st[j] == 0xfeu ? rv[j] : line[j], st[j] == 0xfeu ? 2u : 0u) && want) { defer |= 1u << j; st[j] = 0xffu; rv[j] = av[j] = 0u; }
      }
    }
  }
  while (true) {
    int sj = -1;
    uint64_t sfkey = 0;
    uint32_t slo = 0, scode = 0, sline = 0;
#pragma unroll
    for (int j = 0; j < B; ++j)
      if (sj < 0 && (st[j] == 0xfcu || st[j] == 0xfeu)) { sj = j; sfkey = fkey[j]; slo = rv[j]; scode = st[j]; sline = line[j]; }
    if (!__any(sj >= 0)) break;
    if (sj >= 0) {
      uint2 x = make_uint2(0u, 0u);
      bool have = scode == 0xfeu, beyond = false;
      if (dbg) atomicAdd(&dbg[Push::enabled ? 5 : (have ? 4 : 3)], 1ull);
      if (!have) {
        const uint2 fd = mfx_c_find_lean(c, sfkey, sline, 0u, beyond);
        have = fd.x != 0u;
        slo = fd.y;
      }
      uint32_t r_rv = (slo >> 11) & MFX_CSAT, r_av = slo & MFX_CSAT;
      const bool sat = have && (r_rv == MFX_CSAT || r_av == MFX_CSAT);
      if (sat || beyond) {
        const uint2 sx = mfx_side_lookup_lean(c, keyof(sj));
        if (beyond) { r_rv = sx.x; r_av = sx.y; have = true; }
        else { if (r_rv == MFX_CSAT) r_rv = sx.x; if (r_av == MFX_CSAT) r_av = sx.y; }
      }
      if (have) x = make_uint2((r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv, r_av);
#pragma unroll
      for (int j = 0; j < B; ++j)
        if (sj == j) { rv[j] = x.x; av[j] = x.y; st[j] = 0xffu; }
    }
  }
  return defer;
}

// phase B: the wave's parked queries (nq <= 64 of them) answered and handed to eval(readV, asmV), one entry per lane.
// kmer_at(p): the canonical k-mer at tile position p (quotient form: the side table is keyed by the k-mer itself).
// push1(want, kmer): the worklist again (one query per lane here)
template <class KmerAt, class Eval, class Push1>
__device__ __forceinline__ void mfx_lane_flush(const mfx_table_view &c, mfx_mailbox &M, uint32_t &nq, KmerAt kmer_at, Eval eval, Push1 push1, unsigned long long *dbg = nullptr) {
  const uint32_t tid = threadIdx.x, sub16 = (tid & 7u) << 4, lane = tid & 63u, wbase = tid & ~63u;
  const uint32_t n = nq;
  nq = 0u;
  if (n == 0u) return;                                           // wave-uniform
  mfx_wave_handoff();
  constexpr uint32_t STEPS = 4;
  auto tail_pass = [&](bool second) {
    for (uint32_t q0 = 0; q0 < n; q0 += 8u * STEPS) {
      uint4 sl[STEPS];
      uint32_t fl[STEPS];
      bool act[STEPS];
#pragma unroll
      for (uint32_t sp = 0; sp < STEPS; ++sp) {
        const uint32_t e = q0 + 8u * sp + (lane >> 3);
        sl[sp] = make_uint4(0u, 0u, 0u, 0u);
        act[sp] = false;
        fl[sp] = 0u;
        if (e < n) {
          const uint2 lw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint32_t *>(&M.rec[wbase + e]) + 2);   // {line, position << 4 | flag}
          fl[sp] = lw.y;
          act[sp] = lw.x != MFX_REC_FOUND && (!second || (lw.y & 3u) == 2u);
          if (act[sp]) sl[sp] = *reinterpret_cast<const uint4 *>(reinterpret_cast<uint64_t>(c.slots) + ((uint64_t)lw.x << 7) + sub16);
        }
      }
      mfx_wave_handoff();                                        // every lane has its entries' lines before one answers into a record
#pragma unroll
      for (uint32_t sp = 0; sp < STEPS; ++sp) {
        const uint32_t e = q0 + 8u * sp + (lane >> 3);
        if (act[sp]) {
          uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + e]);
          const uint2 kk = *reinterpret_cast<const uint2 *>(rec);                            // {key field << 22} (a finder of this group may have replaced word 0: same key bits)
          if (sl[sp].y == 0xffffffffu || sl[sp].w == 0xffffffffu) rec[3] = (fl[sp] & ~3u) | 1u;  // an empty slot: the line has room
          if (sl[sp].y == kk.y && ((sl[sp].x ^ kk.x) >> 22) == 0u) { rec[0] = sl[sp].x; rec[2] = MFX_REC_FOUND; }
          else if (sl[sp].w == kk.y && ((sl[sp].z ^ kk.x) >> 22) == 0u) { rec[0] = sl[sp].z; rec[2] = MFX_REC_FOUND; }
        }
      }
    }
    mfx_wave_handoff();
  };
  tail_pass(false);
  // lane e consumes entry e
  const bool mine = lane < n;
  uint4 r = make_uint4(0u, 0u, MFX_REC_FOUND, 0u);
  if (mine) r = M.rec[wbase + lane];
  const bool full = mine && r.z != MFX_REC_FOUND && (r.w & 3u) != 1u;          // its home line is full of other k-mers: the next candidate line
  if (__any(full)) {
    if (full) {
      uint32_t *rec = reinterpret_cast<uint32_t *>(&M.rec[wbase + lane]);
      mfx_probe pr;
      pr.lineA = pr.lineB = r.z;
      rec[2] = (uint32_t)mfx_probe_line(c, pr, 1u);              // (candidate lines 0 .. MFX_MZ_REGION-1 follow the minimizer's line)
      rec[3] = (r.w & ~3u) | 2u;
      if (c.quot) rec[1] = r.y | (1u << (MFX_Q_DSHIFT + 22 - 32));   // quotient form: the key field of candidate line 1 (d = 1 above F0)
      if (dbg) atomicAdd(&dbg[1], 1ull);
    }
    mfx_wave_handoff();
    tail_pass(true);
    if (full) {
      const uint4 r2 = M.rec[wbase + lane];
      r.x = r2.x; r.w = r2.w;
      if (r2.z == MFX_REC_FOUND) r.z = MFX_REC_FOUND;           // (else r.z stays the HOME line)
    }
  }
  // lane e's entry after the passes: closed (found with exact counts, or absent), or RARE -- a saturated count field (the exact
  // count is in the side table), both candidate lines full of other k-mers: listed for mfx_hist_rest_kernel, else ended per lane
  bool have = r.z == MFX_REC_FOUND;
  uint32_t lo = r.x;
  uint64_t fk = (((uint64_t)r.y << 32) | r.x) >> 22;             // the key field (quotient form: F0, without the candidate line's mark)
  if (c.quot) fk &= (1ull << MFX_Q_DSHIFT) - 1ull;
  const bool scan = mine && !have && (r.w & 3u) == 2u;           // neither in its home line nor in the next, both full
  const bool satd = mine && have && (((lo >> 11) & MFX_CSAT) == MFX_CSAT || (lo & MFX_CSAT) == MFX_CSAT);
  bool listed = false;
  if (__any(scan || satd)) {                                     // wave-uniform
    if (dbg && (scan || satd)) atomicAdd(&dbg[scan ? 3 : 2], 1ull);
    listed = push1(scan || satd, (scan || satd) ? (c.quot ? kmer_at(r.w >> 4) : fk) : 0ull, scan ? r.z : lo, scan ? 1u : 2u) && (scan || satd);      // (scan: r.z is still the HOME line)
  }
  if (mine && !listed) {
    bool beyond = false;
    if (scan) {                                                  // whole-line scans from candidate line 2
      if (dbg) atomicAdd(&dbg[5], 1ull);
      const uint2 fd = mfx_c_find_lean(c, fk, r.z, 2u, beyond);
      have = fd.x != 0u;
      lo = fd.y;
    }
    uint32_t r_rv = (lo >> 11) & MFX_CSAT, r_av = lo & MFX_CSAT;
    const bool sat = have && (r_rv == MFX_CSAT || r_av == MFX_CSAT);
    if (sat || beyond) {                                         // the side table is keyed by the k-mer itself
      const uint2 sx = mfx_side_lookup_lean(c, c.quot ? kmer_at(r.w >> 4) : fk);
      if (beyond) { r_rv = sx.x; r_av = sx.y; have = true; }
      else { if (r_rv == MFX_CSAT) r_rv = sx.x; if (r_av == MFX_CSAT) r_av = sx.y; }
    }
    uint32_t o_rv = 0u, o_av = 0u;
    if (have) { o_rv = (r_rv < c.minV || r_rv > c.maxV) ? 0u : r_rv; o_av = r_av; }      // -min / -max (merfin.C:199-200)
    eval(o_rv, o_av);
  }
  mfx_wave_handoff();                                            // the mailbox is free again
}

// the k-mer of query j from an array (the callers that hold their k-mers anyway): a select chain, never an indexed register array
template <int B>
struct mfx_key_from_array {
  const uint64_t (&key)[B];
  __device__ __forceinline__ uint64_t operator()(int sj) const {
    uint64_t k = 0;
#pragma unroll
    for (int j = 0; j < B; ++j) if (j == sj) k = key[j];
    return k;
  }
};

#ifndef MFX_V_LANEPROBE
#define MFX_V_LANEPROBE 1             // 1: per-lane probe of the compact layout (mfx_lane_lookup8); 0: the cooperative one (A/B: tools/ab_build.sh; k <= 21 only)
#endif

// the compact lookup of the kernels that hold their k-mers: home line, first mini-bucket and key field by mfx_home
template <int B>
__device__ __forceinline__ void mfx_compact_lookup(const mfx_table_view &t, mfx_mailbox &M, const uint64_t (&key)[B], const uint64_t (&krc)[B],
                                                   const bool (&ok)[B], uint32_t (&rv)[B], uint32_t (&av)[B]) {
#if MFX_V_LANEPROBE
  uint64_t fkey[B];
  uint32_t line[B], b0[B];
#pragma unroll
  for (int j = 0; j < B; ++j) {
    fkey[j] = key[j]; line[j] = 0u; b0[j] = 0u;
    if (ok[j]) { const mfx_probe pr = mfx_home(t, key[j]); fkey[j] = pr.fkey; line[j] = pr.lineA; b0[j] = pr.b0; }
  }
  (void)mfx_lane_lookup8<B>(t, M, fkey, ok, rv, av, line, b0, mfx_key_from_array<B>{key});
#else
  mfx_group_lookup8<B>(t, M, key, krc, ok, rv, av);
#endif
}
#define MFX_COMPACT_LOOKUP(t, MB, key, krc, ok, rv, av) mfx_compact_lookup<MFX_BATCH>(t, MB, key, krc, ok, rv, av)

// k-mer starting at tile position p; returns validity (all k bases ACGT)
__device__ __forceinline__ bool mfx_tile_kmer(const mfx_tile_lds &L, int k, uint32_t p, uint64_t &fwd) {
  uint32_t w = p >> 5, o = p & 31;
  uint64_t w0 = L.codes[w], w1 = L.codes[w + 1];
  uint32_t sh = 2 * o;
  uint64_t hi = (w0 << sh) | ((w1 >> 1) >> (63 - sh));
  fwd = hi >> (64 - 2 * k);
  uint64_t vv = (((uint64_t)L.valid[w] << 32) | L.valid[w + 1]) << o;
  return (vv >> (64 - k)) == ((~0ULL) >> (64 - k));
}

// ---------------------------------------------------------------------------
// The home lines of a wave's k-mers under the mod-minimizer placement (mfx_minimizer_mod), found for the whole wave at once.
// The lanes of a wave hold 64 consecutive positions per batch element, and the k-mer at position q samples its window by the
// smallest of the t-mers at q .. q+npos-1: a SLIDING-WINDOW MINIMUM over values that each lane computes once (the order hash
// of the t-mer at its own position) instead of npos times.  Values are 16-bit {order: 9 bits | position in the wave: 7 bits}
// so that the minimum carries its position; two of them ride in one register -- low half: ties to the leftmost (position
// q), high half: ties to the rightmost (127 - q) -- and v_pk_min_u16 takes both minima at once; a k-mer whose canonical form
// is the reverse strand reads the high half (the leftmost t-mer of the canonical k-mer is the rightmost of the forward one).
// The up to 15 positions behind the wave's 64 are computed once for all B batch elements: lane 15 * j + i takes the t-mer at
// position 64 + i of element j (mfx_wave_mod_halo).
// f, r: the forward k-mer at the lane's position and its reverse complement (whatever their validity: a valid k-mer only
// ever looks at t-mers inside itself).  p0: the tile position of the lane's first batch element.  Every lane of the wave
// takes part (the calls are wave-uniform).
// ---------------------------------------------------------------------------
#ifndef MFX_V_MIN2LEVEL
#define MFX_V_MIN2LEVEL 1             // the sliding-window minimum of mfx_wave_mod_line in two levels (A/B: tools/ab_build.sh -DMFX_V_MIN2LEVEL=0)
#endif
typedef unsigned short mfx_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mfx_pk_min_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mfx_us2, a), __builtin_bit_cast(mfx_us2, b)));
}
__device__ __forceinline__ uint32_t mfx_mod_pack(uint32_t order, uint32_t q) { return ((order << 7) | q) | (((order << 7) | (127u - q)) << 16); }

// the order values of the H = npos - 1 positions behind the wave (H <= 27), for all B batch elements: lane H * jj + i takes the
// t-mer at position 64 + i of element jj; 64 / H elements fit one pass (all four for k = 21: H = 15), the rest a second one
template <int B>
__device__ __forceinline__ void mfx_wave_mod_halo(const mfx_table_view &c, const mfx_tile_lds &L, uint32_t p0, uint32_t (&halo)[2]) {
  const int t = c.mz_t;
  const uint32_t H = (uint32_t)(c.k - t), per = 64u / H;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t jj = lane / H, ii = lane - H * jj;
#pragma unroll
  for (uint32_t ps = 0; ps < 2u; ++ps) {
    halo[ps] = 0xffffffffu;
    const uint32_t el = ps * per + jj;
    if (ps * per < (uint32_t)B && jj < per && el < (uint32_t)B) {     // (wave-uniform first test: no second pass when one holds all B)
      const uint32_t q = p0 - lane + el * MFX_BLOCK + 64u + ii, wd = q >> 5, o = q & 31u, sh = 2u * o;
      const uint64_t w0 = L.codes[wd], w1 = L.codes[wd + 1];
      const uint32_t a = (uint32_t)(((w0 << sh) | ((w1 >> 1) >> (63u - sh))) >> (64 - 2 * t));
      const uint32_t b = (uint32_t)mfx_revcomp((uint64_t)a, t);
      halo[ps] = mfx_mod_pack(mfx_tmer_order(a < b ? a : b), 64u + ii);
    }
  }
}

// the home line of the lane's k-mer of batch element j (f: forward k-mer, r: its reverse complement).  mw: 64 + npos words of LDS
// of this wave (the mailbox of the lookup, idle at this point): the wave's values are written side by side and every lane takes
// the minimum of the npos words from its own on -- npos plain LDS reads and npos - 1 packed minima (16 for k = 21); a doubling
// scheme over cross-lane reads (4 x 2 ds_bpermute, the second for the positions behind the wave) cost twice the instructions.
// fkey: what the key field of the k-mer's slot holds (the canonical k-mer; quotient form: mfx_q_place).
__device__ __forceinline__ uint32_t mfx_wave_mod_line(const mfx_table_view &c, uint32_t *mw, const uint32_t (&halo)[2], int j, uint64_t f, uint64_t r,
                                                      uint32_t &b0, uint64_t &fkey) {
  const int k = c.k, t = c.mz_t, w = c.mz_w, m = k - w + 1, npos = k - t + 1;
  const uint32_t lane = threadIdx.x & 63u, tmask = (1u << (2 * t)) - 1u;
  const uint32_t H = (uint32_t)(k - t), per = 64u / H;
  const uint64_t mmask = (~0ULL) >> (64 - 2 * m);
  const uint32_t a = (uint32_t)(f >> (2 * (k - t))) & tmask, b = (uint32_t)r & tmask;     // the t-mer at this position, and its reverse complement
  const uint32_t hsrc = (uint32_t)j < per ? halo[0] : halo[1];
  const uint32_t hv = (uint32_t)__shfl((int)hsrc, (int)(H * ((uint32_t)j < per ? (uint32_t)j : (uint32_t)j - per) + lane), 64);
  mfx_wave_handoff();                                          // the previous element's reads are done
  mw[lane] = mfx_mod_pack(mfx_tmer_order(a < b ? a : b), lane);
  if (lane <= H) mw[64u + lane] = lane < H ? hv : 0xffffffffu;
  mfx_wave_handoff();
  uint32_t v = 0xffffffffu;
#if MFX_V_MIN2LEVEL
  // The minimum over npos consecutive words in two levels: A[p] = min of the g1 words from p on -- every lane takes A[lane], the
  // first npos - g1 + 1 lanes also A[64 + lane] (the positions behind the wave) --, written back over the words, then the minimum
  // of g2 = ceil(npos / g1) values A[lane + j g1] (the last one moved back to end at the window's end: overlapping blocks do not
  // change a minimum).  k = 21: 12 reads and 9 packed minima instead of 16 and 15; k = 31 (28 t-mers): 17 and 14 instead of 28 and 27.
  {
    const int g1 = npos >= 26 ? 6 : npos >= 10 ? 4 : npos >= 5 ? 3 : 2, g2 = (npos + g1 - 1) / g1;
    uint32_t a0 = 0xffffffffu, a1 = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (i < g1) { a0 = mfx_pk_min_u16(a0, mw[lane + (uint32_t)i]); a1 = mfx_pk_min_u16(a1, mw[(lane <= (uint32_t)(npos - g1) ? 64u + lane : lane) + (uint32_t)i]); }
    mfx_wave_handoff();                                        // every lane has read its words
    mw[lane] = a0;
    if (lane <= (uint32_t)(npos - g1)) mw[64u + lane] = a1;
    mfx_wave_handoff();
#pragma unroll
    for (int j = 0; j < 7; ++j)
      if (j < g2) v = mfx_pk_min_u16(v, mw[lane + (uint32_t)(j * g1 < npos - g1 ? j * g1 : npos - g1)]);
  }
#else
#pragma unroll
  for (int i = 0; i < 28; ++i)
    if (i < npos) v = mfx_pk_min_u16(v, mw[lane + (uint32_t)i]);
#endif
  const bool fwd = f <= r;
  const uint32_t q = fwd ? (v & 127u) : 127u - ((v >> 16) & 127u);
  const uint32_t xf = q - lane, jf = xf % (uint32_t)w;        // the t-mer's offset and the window, counted on the forward strand
  const uint64_t ma = (f >> (2 * ((uint32_t)w - 1u - jf))) & mmask, mb = (r >> (2 * jf)) & mmask;
  const uint32_t xc = fwd ? xf : (uint32_t)(k - t) - xf;     // the offset in the canonical k-mer
  uint32_t line;
  fkey = fwd ? f : r;
  if (c.quot) {
    // the pieces mfx_q_parts finds from the canonical k-mer, read off the forward-strand ones ((k - t) % 4 == 3: window jf of the
    // forward k-mer is window 3 - jf of its reverse complement)
    const uint32_t jc = fwd ? jf : 3u - jf;
    const uint64_t ac = fwd ? ma : mb, bc = fwd ? mb : ma;
    const uint32_t e = (uint32_t)(((fkey >> (2 * (m + 3 - (int)jc))) << (2 * (3 - (int)jc))) | (fkey & ((1ull << (2 * (3 - (int)jc))) - 1ull)));
    mfx_q_place(c, ac < bc ? ac : bc, bc < ac ? 1u : 0u, jc, e, xc, line, b0, fkey);
  } else {
    // The oriented window, read off the forward strand: the per-k-mer form (mfx_p_parts on the canonical k-mer: window jc, reversed iff
    // b < a) gives jf where the minimizer reads canonical on the forward strand and 3 - jf where it does not, whatever strand the K-MER is
    // canonical on; a palindromic minimizer (ma == mb) counts as standing as it is in the canonical k-mer.
    const bool lt = ma < mb, keep = lt || (fwd && ma == mb);
    mfx_mod_place(lt ? ma : mb, keep ? jf : (uint32_t)w - 1u - jf, c.nlines, line, b0, xc);
  }
  return line;
}

// ===========================================================================
// -hist
// ===========================================================================

#ifndef MFX_V_BATCH
#define MFX_V_BATCH 4
#endif
constexpr int MFX_BATCH = MFX_V_BATCH;          // queries per lane and cooperative-probe sequence (tools/ab_build.sh -DMFX_V_BATCH=2: A/B)

// KF, WF: k-mer size and minimizer windows as COMPILE-TIME constants (0: taken from the table at run time).  Every shift, mask
// and window count of the extraction, the reverse complement and the placement hash is then an immediate instead of a scalar
// register -- the generic kernel keeps so many loop-invariant scalars that a third of them live spilled in VGPR lanes and are
// read back (v_readlane + hazard nops) for every query.  The launcher picks the k = 21 / w = 4 instance for the compact layout
// (meryl's default k for a human genome, BASELINE configs 1-4: 103.6 -> 108.0 G k-mers/s); any other k, and the full table --
// whose kernel sits on the HBM line rate either way (91.0 G with and without) -- run the generic instance.
#ifndef MFX_V_MINBLOCKS
#define MFX_V_MINBLOCKS 4             // blocks per CU the register allocation aims at (tools/ab_build.sh -DMFX_V_MINBLOCKS=3: A/B)
#endif
// Occupancy target (blocks per CU = waves per SIMD) and queries per lane of the -hist instances.  The per-lane probe of the
// compact layout is bound by round trips: throughput follows (waves x queries in flight) / latency, and what a wave holds per query
// decides how many waves fit.  Round 4 (profiles/r04_kernel_waves.txt, 3 Gb, same box, back to back): 4 waves x 4 queries 134.1 G,
// 5 x 4 142.1, 6 x 2 144.4, 7 x 2 147.6 (72 VGPRs, 22.5 KB of LDS per block: the prob / over-copy tables left LDS for that),
// 6 x 3 138.5, 6 x 4 107.8, 8 x 2 101.9 (spills), 8 x 1 135.6.  k = 31 (quotient form; profiles/r04_ab_k31.txt): 4 x 4 115.7 G, 5 x 4 85.8, 5 x 2 117.6,
// 6 x 2 = 7 x 2 125.7.  The generic compact instance (other k: run-time shifts) is best at 4 x 4 (k = 22 / 25 / 27 at 1 Gb: 91 / 83 / 86 G; 6 x 2: 82 / 79 / 80),
// and so is the cooperative probe of the full table.
#ifndef MFX_V_MINBLOCKS_K21
#define MFX_V_MINBLOCKS_K21 7
#endif
#ifndef MFX_V_BATCH_K21
#define MFX_V_BATCH_K21 2
#endif
#ifndef MFX_V_MINBLOCKS_K31
#define MFX_V_MINBLOCKS_K31 6
#endif
#ifndef MFX_V_BATCH_K31
#define MFX_V_BATCH_K31 2
#endif
#ifndef MFX_V_MINBLOCKS_GEN
#define MFX_V_MINBLOCKS_GEN 4
#endif
#ifndef MFX_V_BATCH_GEN
#define MFX_V_BATCH_GEN 4
#endif
// defer: the probe's tail deferred (mfx_lane_probe_defer / mfx_lane_flush).  It saves instructions (k = 21: 209 -> 200 wave-VALU per
// k-mer, k = 31: 238 -> 234) and costs table lines (a parked query's home line is fetched again when the mailbox is worked off: 0.321 ->
// 0.335 lines per k-mer) -- a gain for the instance the ALUs bound (k = 31, quotient form: 135.7 -> 139.0 G k-mers/s), a loss for the one on
// the HBM's random-line rate (k = 21: 150.4 -> 143.4 G; profiles/r05_kernel_ab.txt).
#ifndef MFX_V_DEFER_K21
#define MFX_V_DEFER_K21 0
#endif
#ifndef MFX_V_DEFER_K31
#define MFX_V_DEFER_K31 1
#endif
// kfxlds: the lane's koverCpy sum of the tile lives in LDS -- the upper half of the wave's mailbox, whose passes then have 32 entries
// (MFX_KFX_BASE) -- instead of a register pair: at 72 VGPRs the allocator kept that pair, live across the whole tile and touched by
// one k-mer in fifty, in SCRATCH, and every evaluation that met an `asmK > readK` k-mer paid a scratch load and store behind a
// full vmcnt wait (107 G k-mers/s against 150 G; profiles/r06_pmc_compare.txt: 16 M scratch stores per 10^9 k-mers).  A ds_add_u64
// without return costs one issue slot and nothing waits for it.
#ifndef MFX_V_KFXLDS_K21
#define MFX_V_KFXLDS_K21 1
#endif
constexpr uint32_t MFX_KFX_BASE = 32u;
template <bool CANON, bool COMPACT, int KF> struct mfx_hist_tune { static constexpr int blocks = MFX_V_MINBLOCKS, batch = MFX_V_BATCH, defer = 0, kfxlds = 0; };
template <> struct mfx_hist_tune<true, true, 21> { static constexpr int blocks = MFX_V_MINBLOCKS_K21, batch = MFX_V_BATCH_K21, defer = MFX_V_DEFER_K21, kfxlds = MFX_V_KFXLDS_K21; };
#ifndef MFX_V_KFXLDS_K31
#define MFX_V_KFXLDS_K31 0
#endif
template <> struct mfx_hist_tune<true, true, 31> { static constexpr int blocks = MFX_V_MINBLOCKS_K31, batch = MFX_V_BATCH_K31, defer = MFX_V_DEFER_K31, kfxlds = MFX_V_KFXLDS_K31; };
template <> struct mfx_hist_tune<true, true, 0> { static constexpr int blocks = MFX_V_MINBLOCKS_GEN, batch = MFX_V_BATCH_GEN, defer = 0, kfxlds = 0; };
template <bool CANON, bool COMPACT, int KF, int WF, int TF, bool DBG = false>
__global__ __launch_bounds__(MFX_BLOCK, (mfx_hist_tune<CANON, COMPACT, KF>::blocks)) void mfx_hist_kernel(mfx_hist_args a) {
  constexpr int BT = mfx_hist_tune<CANON, COMPACT, KF>::batch;                  // queries per lane and probe sequence of this instance
  __shared__ mfx_tile_lds L;
  __shared__ mfx_mailbox MB;
  __shared__ mfx_hist_lds H;

  const uint32_t tid = threadIdx.x;
  if (KF) { a.t.k = KF; a.t.mz_w = WF; a.t.mz_t = TF; a.t.quot = KF > MFX_MAX_K_DIRECT ? 1 : 0; }   // (the launcher checked that they are the table's)
  const int k = KF ? KF : a.t.k;
  const mfx_kstar_args &ka = a.ks;
  mfx_hist_lds_init(H, ka);
  const bool lut_ok = H.lut_ok != 0u;

  // per-lane counters: 16 positions per tile and lane, so 32 bits hold 2^28 tiles of one block (a terabase); they are
  // widened when they leave the lane.  The block's running totals live in LDS (only thread 0 touches them).
  uint32_t n_valid = 0, n_missing = 0, n_over0 = 0;
  if (tid == 0) { H.tot[0] = H.tot[1] = 0; H.wl_n = 0u; }
  uint64_t *c_glob = ka.counts + 2ull * ka.nbins;        // kasm, kmissing, novf
  uint64_t *c_kasm = c_glob + 3, *c_kmis = c_kasm + ka.ncontigs;

  // Persistent block, tiles handed out dynamically: block b starts on tile_begin + b and
  // draws every further tile from one global counter (the draw for the NEXT tile is issued
  // before the current one is processed, so its latency is never waited on).  The k-mer
  // counters are integers and koverCpy is kept per (tile, wave), so the result does not
  // depend on which block evaluated which tile.
  const uint32_t none = 0xffffffffu;
  uint32_t c = none;
  uint64_t li = blockIdx.x;                    // number of the tile within this launch
  for (uint32_t it = 0; li < a.n_logical; ++it) {
    if (tid == 0)
      H.next[it & 1] = (uint64_t)gridDim.x + atomicAdd((unsigned long long *)a.tile_ctr, 1ull);
    const uint64_t tile = a.part_n == 1 ? a.tile_begin + li
                                         : ((((li >> a.part_shift) * a.part_n + a.part_rank) << a.part_shift) | (li & ((1ull << a.part_shift) - 1ull)));
    const uint32_t cn = a.tile_contig[tile];
    if (cn != c) {
      if (c != none) {
        // contig change (block-uniform): flush the per-contig counters
        uint64_t x = n_valid, y = n_missing, z = 0;
        mfx_block_sum3(x, y, z, H.red);
        if (tid == 0 && (x | y)) {
          atomicAdd((unsigned long long *)&c_kasm[c], x);
          atomicAdd((unsigned long long *)&c_kmis[c], y);
          H.tot[0] += x;                                 // the global totals leave the block once, at its end:
          H.tot[1] += y;                                 // a fragmented assembly changes contig on every tile
        }
        n_valid = n_missing = 0;
      }
      c = cn;
    }
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    const uint8_t *src = a.bases + a.contig_off[c] + pos0;

    if (a.codes) {                           // block-uniform: the sequence arrived packed (mfx_hist_run_streamed)
      const uint64_t w0 = (a.contig_off[c] + pos0) >> 5;
      mfx_tile_fill_packed(L, a.codes + w0, a.valid + w0);
    } else {
      mfx_tile_fill(L, src);                 // the previous tile was fully consumed at the barrier below
    }
    __syncthreads();

    // This lane's koverCpy terms of this tile, in units of 2^-52 (a term is (1 - readK/asmK) * prob in [0, 1]: at most 16 of them per lane
    // and tile).  An INTEGER sum: the value of a (tile, wave) does not depend on the order its terms were added in, nor on which lane
    // evaluated which k-mer (the deferred tail of the probe hands parked queries to other lanes of the wave, mfx_lane_flush).
    uint64_t kfx = 0;
    constexpr bool kfx_lds = mfx_hist_tune<CANON, COMPACT, KF>::kfxlds != 0 && COMPACT && CANON && KF != 0 && TF != 0;
    constexpr uint32_t mb_cap = kfx_lds ? MFX_KFX_BASE : 64u;      // mailbox entries the probe has
    unsigned long long *const kfxw = reinterpret_cast<unsigned long long *>(&MB.rec[(tid & ~63u) + MFX_KFX_BASE]) + (tid & 63u);   // (kfx_lds) this lane's word
    if (kfx_lds) *kfxw = 0ull;                                   // (the mailbox was handed back at the end of the tile before)
    // ... and so does the count of the dominant bin (`over` 0: nine k-mers in ten), in the same word above the 56 bits of the sum (a lane
    // evaluates at most 16 k-mers of a tile: 16 terms below 2^52 each, a count of at most 16)
    // (a 32-bit add to the word's high half: the 64-bit constant 1 << 56 was hoisted out of the loop as a register pair -- and spilled)
    struct lds_bump { unsigned long long *w; __device__ __forceinline__ void operator++(int) { (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(w) + 1, 1u << 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } };
    struct lds_sum { unsigned long long *w; __device__ __forceinline__ void operator+=(uint64_t v) { (void)__hip_atomic_fetch_add(w, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } };
    auto eval1 = [&](uint32_t rvv, uint32_t avv) {
      if (kfx_lds) {
        lds_bump bump{kfxw};
        lds_sum sum{kfxw};                                       // (added where the term is made: the `asmK > readK` branch)
        if (mfx_hist_eval_fx(H, ka, lut_ok, rvv, avv, bump, sum)) n_missing++;
      } else if (mfx_hist_eval_fx(H, ka, lut_ok, rvv, avv, n_over0, kfx)) n_missing++;
    };
    // mod-minimizer placement: the wave finds its home lines together (mfx_wave_mod_line)
    const bool wave_lines = COMPACT && CANON && (KF ? TF != 0 : a.t.mz_t != 0);
    // the specialised instances: the probe's tail is deferred (mfx_lane_flush).  Odd k only: a parked query's counts are not seen by the
    // palindrome doubling below (an even k takes the undeferred probe)
    const bool defer_tail = MFX_V_DEFER != 0 && mfx_hist_tune<CANON, COMPACT, KF>::defer != 0 && COMPACT && CANON && KF != 0 && TF != 0 && (KF & 1) != 0;
    const bool quotf = KF ? KF > MFX_MAX_K_DIRECT : a.t.quot != 0;
    // The worklist of mfx_hist_rest_kernel: a query whose probe did not end in its two cooperative passes is LISTED -- {canonical
    // k-mer, aux, the (tile, wave) slot its koverCpy term belongs to, mode, "an even-k palindrome: both counts twice"} -- instead of
    // being scanned for by its lane while 63 others wait.  mode 0: aux = its home line; 1 (deep): ... and that line and the next are
    // known to be full of other k-mers; 2 (saturated): aux = the low word of its slot, only the side table is left to read.  The whole wave calls; the lanes that want are given consecutive entries
    // of this BLOCK's segment of the list (one LDS atomic per call); false: no list (a.wl == nullptr) or no room left in the segment,
    // the lane then scans.  kasm is counted here either way.
    auto push_wave = [&](bool want, uint64_t kmer, uint32_t aux, uint32_t mode, bool dbl) -> bool {
      if (!a.wl) return false;                                     // (kernel-uniform)
      const uint64_t m = __ballot(want);
      if (!m) return false;
      const uint32_t ln = tid & 63u, leader = (uint32_t)__ffsll((unsigned long long)m) - 1u;
      uint32_t base = 0;
      if (ln == leader) base = atomicAdd(&H.wl_n, (uint32_t)__popcll(m));       // this block's own segment of the list: an LDS counter, no global atomic
      base = (uint32_t)__shfl((int)base, (int)leader, 64);
      const uint32_t at = base + (uint32_t)__popcll(m & ((1ULL << ln) - 1ULL));
      const bool fits = want && at < a.wl_segcap;
      if (fits) reinterpret_cast<uint4 *>(a.wl + MFX_WL_HEADER)[(uint64_t)blockIdx.x * a.wl_segcap + at] =
                  make_uint4((uint32_t)kmer, (uint32_t)(kmer >> 32), aux, ((uint32_t)li * (MFX_BLOCK / 64) + (tid >> 6)) | (dbl ? 0x80000000u : 0u) | (mode << 29));
      return fits;
    };
    uint32_t nq = 0;                         // queries parked in this wave's mailbox (wave-uniform)
    uint32_t *const mwave = reinterpret_cast<uint32_t *>(H.dred) + (tid >> 6) * 128u;        // the wave's 64 + npos words of mfx_wave_mod_line (H.dred is idle in this kernel)
    for (uint32_t b = 0;; b += BT) {
      const bool last = b >= MFX_TILE / MFX_BLOCK || b * MFX_BLOCK >= n;    // (short last tile of a contig, block-uniform: nothing starts beyond n)
      nq = (uint32_t)__builtin_amdgcn_readfirstlane((int)nq);
      if (defer_tail && (last || nq >= (MFX_DEFER_FLUSH < mb_cap ? MFX_DEFER_FLUSH : mb_cap / 2u))) {
        auto kmer_at = [&](uint32_t p) -> uint64_t {
          uint64_t f;
          (void)mfx_tile_kmer(L, k, p, f);
          const uint64_t r = mfx_revcomp(f, k);
          return f < r ? f : r;
        };
        mfx_lane_flush(a.t, MB, nq, kmer_at, eval1, [&](bool want, uint64_t km, uint32_t aux, uint32_t mode) { return push_wave(want, km, aux, mode, false); }, DBG ? reinterpret_cast<unsigned long long *>(a.dbg) : nullptr);
      }
      if (last) break;
      uint32_t rv[BT], av[BT];
      bool     ok[BT];
      uint32_t parked = 0u;
      const bool even_k = KF ? (KF & 1) == 0 : (k & 1) == 0;
      if (wave_lines) {
        uint64_t fkey[BT];
        uint32_t line[BT], b0[BT];
        uint32_t halo[2], pal = 0u;
        mfx_wave_mod_halo<BT>(a.t, L, b * MFX_BLOCK + tid, halo);
#pragma unroll
        for (int j = 0; j < BT; ++j) {
          const uint32_t p = (b + j) * MFX_BLOCK + tid;     // lane-consecutive positions
          uint64_t f;
          ok[j] = mfx_tile_kmer(L, k, p, f) && (p < n);
          const uint64_t r = mfx_revcomp(f, k);
          line[j] = mfx_wave_mod_line(a.t, mwave, halo, j, f, r, b0[j], fkey[j]);
          if (!ok[j]) { line[j] = 0u; b0[j] = 0u; }            // no k-mer here: a dummy load of line 0, ignored
          if (even_k && f == r) pal |= 1u << j;
        }
        // the k-mer of query sj, for the rare endings of the probe: the key field itself (k <= 21), or again from the tile
        const bool quot = quotf;
        auto keyof = [&](int sj) -> uint64_t {
          uint64_t kk = 0;
          if (!quot) {
#pragma unroll
            for (int j = 0; j < BT; ++j) if (j == sj) kk = fkey[j];
          } else {
            uint64_t f;
            (void)mfx_tile_kmer(L, k, (b + (uint32_t)sj) * MFX_BLOCK + tid, f);
            const uint64_t r = mfx_revcomp(f, k);
            kk = f < r ? f : r;
          }
          return kk;
        };
        if (defer_tail) {
          uint32_t posn[BT];
#pragma unroll
          for (int j = 0; j < BT; ++j) posn[j] = (b + (uint32_t)j) * MFX_BLOCK + tid;
          auto pj = [&](int jj, bool want, uint64_t km, uint32_t aux, uint32_t mode) { (void)jj; return push_wave(want, km, aux, mode, false); };     // (odd k: no palindromes)
          parked = mfx_lane_probe_defer<BT, decltype(keyof), mfx_push_fn<decltype(pj)>, mb_cap>(
                       a.t, MB, nq, fkey, ok, rv, av, line, b0, posn, keyof, DBG ? reinterpret_cast<unsigned long long *>(a.dbg) : nullptr, mfx_push_fn<decltype(pj)>{pj});
        } else {
          auto pj = [&](int jj, bool want, uint64_t km, uint32_t aux, uint32_t mode) { return push_wave(want, km, aux, mode, ((pal >> jj) & 1u) != 0u); };
          parked = mfx_lane_lookup8<BT, decltype(keyof), mfx_push_fn<decltype(pj)>, mb_cap>(
                       a.t, MB, fkey, ok, rv, av, line, b0, keyof, DBG ? reinterpret_cast<unsigned long long *>(a.dbg) : nullptr, mfx_push_fn<decltype(pj)>{pj});
        }
        if (even_k) {
          // even k, canonical database: a k-mer that is its own reverse complement is looked up as fmer AND as rmer by the
          // reference -- the same slot twice (value(fmer) + value(rmer), uint32 arithmetic); every other k-mer has one strand
          // in the database, the one probed
#pragma unroll
          for (int j = 0; j < BT; ++j) if ((pal >> j) & 1u) { rv[j] += rv[j]; av[j] += av[j]; }
        }
      } else {
        uint64_t key[BT], key2[BT];
#pragma unroll
        for (int j = 0; j < BT; ++j) {
          const uint32_t p = (b + j) * MFX_BLOCK + tid;     // lane-consecutive positions
          uint64_t f;
          ok[j] = mfx_tile_kmer(L, k, p, f) && (p < n);
          const uint64_t r = mfx_revcomp(f, k);
          if (CANON) {
            key[j] = f < r ? f : r; key2[j] = f < r ? r : f;     // canonical k-mer and its reverse complement
          } else {
            key[j] = f; key2[j] = r;
          }
        }
        if (COMPACT) mfx_compact_lookup<BT>(a.t, MB, key, key2, ok, rv, av);
        else mfx_group_lookup<BT>(a.t, MB, key, key2, ok, rv, av);
        if (!CANON) {
          // value(fmer) + value(rmer), uint32 arithmetic (merfin-globals.C:107-108)
          uint32_t rv2[BT], av2[BT];
          if (COMPACT) mfx_compact_lookup<BT>(a.t, MB, key2, key, ok, rv2, av2);
          else mfx_group_lookup<BT>(a.t, MB, key2, key, ok, rv2, av2);
#pragma unroll
          for (int j = 0; j < BT; ++j) { rv[j] += rv2[j]; av[j] += av2[j]; }
        } else if (even_k) {
          // (the same slot twice for a k-mer that is its own reverse complement: see above)
#pragma unroll
          for (int j = 0; j < BT; ++j) if (key[j] == key2[j]) { rv[j] += rv[j]; av[j] += av[j]; }
        }
      }
#pragma unroll
      for (int j = 0; j < BT; ++j) {
        if (!ok[j]) continue;
        n_valid++;                                                   // merfin-histogram.C:58
        if (!((parked >> j) & 1u)) eval1(rv[j], av[j]);             // (a parked query is evaluated when the mailbox is flushed)
      }
    }
    // koverCpy of this (tile, wave): the integer sum over the 64 lanes (< 2^62)
    if (kfx_lds) { mfx_wave_handoff(); const unsigned long long w = *kfxw; kfx = w & ((1ull << 56) - 1ull); n_over0 += (uint32_t)(w >> 56); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kfx += __shfl_down(kfx, off, 64);
    // (stored as the INTEGER: mfx_hist_rest_kernel adds the terms of the k-mers it ends to the same word; the summing kernels
    // convert, mfx_sum_chunks_kernel<true>)
    if ((tid & 63u) == 0) reinterpret_cast<uint64_t *>(a.tile_partials)[li * (MFX_BLOCK / 64) + (tid >> 6)] = kfx;

    __syncthreads();                         // tile consumed; H.next[it & 1] written
    const uint64_t nx = H.next[it & 1];
    li = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(nx >> 32)) << 32) |
         (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)nx);
  }
  // last contig of this block + the register-held dominant bin
  {
    uint64_t x = n_valid, y = n_missing, z = n_over0;
    mfx_block_sum3(x, y, z, H.red);
    if (tid == 0 && a.wl) a.wl[2 + blockIdx.x] = H.wl_n < a.wl_segcap ? H.wl_n : a.wl_segcap;      // this block's segment of the worklist, written by every block of every launch (mfx_hist_rest_kernel)
    if (tid == 0) {
      if (c != none && (x | y)) {
        atomicAdd((unsigned long long *)&c_kasm[c], x);
        atomicAdd((unsigned long long *)&c_kmis[c], y);
      }
      if (H.tot[0] + x) atomicAdd((unsigned long long *)&c_glob[0], H.tot[0] + x);
      if (H.tot[1] + y) atomicAdd((unsigned long long *)&c_glob[1], H.tot[1] + y);
      if (z) atomicAdd((unsigned long long *)&ka.counts[ka.nbins], z);
    }
  }
  mfx_hist_lds_flush_bins(H, ka);
}

// ---------------------------------------------------------------------------
// The queries mfx_hist_kernel listed instead of ending them (push_wave): a wave there serves 128 positions at a time, and one lane
// scanning further candidate lines or the side table holds the other 63 -- here EVERY lane has an entry of its own, so the plain
// per-lane lookup (mfx_c_lookup: whole lines, all candidate lines, the side table, the read filter) runs with the wave full.
// Each entry is evaluated like a position of the sequence (merfin-histogram.C:63-90): bins and counters into the same image, the
// koverCpy term -- an integer, units of 2^-52 -- added to the word of its own (tile, wave), so that the launch's result is the one
// the main kernel alone would have produced, bit for bit, whatever was listed.  kasm was counted where the position was seen.
// Launched behind every launch of the main kernel on the same stream.  The list is cut into one segment per block of the main kernel
// (a block appends to its own segment through a counter in LDS: one shared counter would be a single-address atomic per listed wave,
// ~90 M/s -- 80 ms of a launch that lists 7 M of them) and writes its count when it ends; MFX_REST_SPLIT blocks here share a segment.
// ---------------------------------------------------------------------------
constexpr uint32_t MFX_REST_SPLIT = 4;
__global__ __launch_bounds__(MFX_BLOCK) void mfx_hist_rest_kernel(mfx_hist_args a) {
  __shared__ mfx_hist_lds H;
  const uint32_t tid = threadIdx.x;
  const mfx_kstar_args &ka = a.ks;
  uint64_t *c_glob = ka.counts + 2ull * ka.nbins, *c_kmis = c_glob + 3 + ka.ncontigs;
  uint64_t *kfx_words = reinterpret_cast<uint64_t *>(a.tile_partials);
  uint32_t n_missing = 0, n_over0 = 0;
  bool ready = false, lut_ok = false;                          // the K* tables of this block are made when it meets its first entry
  // MFX_REST_SPLIT blocks share a segment (block b: segment b / SPLIT, every SPLIT-th group of 256 entries from b % SPLIT on)
  for (uint32_t sb = blockIdx.x; sb < a.wl_segs * MFX_REST_SPLIT; sb += gridDim.x) {
    const uint32_t seg = sb / MFX_REST_SPLIT, part = sb % MFX_REST_SPLIT;
    const uint32_t n = (uint32_t)__hip_atomic_load(reinterpret_cast<unsigned long long *>(&a.wl[2 + seg]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (n <= part * MFX_BLOCK) continue;                       // block-uniform
    if (!ready) { mfx_hist_lds_init(H, ka); lut_ok = H.lut_ok != 0u; ready = true; }
    const uint4 *ent = reinterpret_cast<const uint4 *>(a.wl + MFX_WL_HEADER) + (uint64_t)seg * a.wl_segcap;
    for (uint32_t i0 = part * MFX_BLOCK; i0 < n; i0 += MFX_REST_SPLIT * MFX_BLOCK) {           // block-uniform trip count
      const uint32_t i = i0 + tid;
      const bool live = i < n;
      uint4 e = make_uint4(0u, 0u, 0u, 0u);
      if (live) e = ent[i];
      bool missing = false;
      if (live) {
        const uint64_t km = ((uint64_t)e.y << 32) | e.x;
        uint2 x = make_uint2(0u, 0u);
        const uint32_t mode = (e.w >> 29) & 3u;
        if (mode == 2u) x = mfx_c_fields(a.t, km, e.z);         // the slot's low word is known: the exact counts of its saturated fields (side table), the read filter
        else if (!a.t.quot) {                                    // k <= 21: the key field is the k-mer, the entry has its home line -- no placement to find again
          mfx_probe pr;
          pr.lineA = e.z; pr.lineB = a.t.mz_w > 0 ? mfx_range32(mfx_hash64(km), a.t.nlines) : e.z; pr.b0 = 0u; pr.fkey = km;
          unsigned long long word = 0;
          bool beyond;
          // deep: its home line and the next are known to be full of other k-mers: from candidate line 2 on
          if (mfx_c_find(a.t, pr, mode == 1u ? 2u : 0u, word, beyond)) x = mfx_c_fields(a.t, km, (uint32_t)word);
        } else x = mfx_c_lookup(a.t, km);
        if (e.w >> 31) { x.x += x.x; x.y += x.y; }             // an even-k palindrome: value(fmer) + value(rmer) is the same slot twice (mfx_hist_kernel)
        uint64_t kfx = 0;
        missing = mfx_hist_eval_fx(H, ka, lut_ok, x.x, x.y, n_over0, kfx);
        if (kfx) atomicAdd(reinterpret_cast<unsigned long long *>(&kfx_words[e.w & 0x1fffffffu]), (unsigned long long)kfx);
      }
      uint32_t ctg = 0u;                                         // the entry's contig, from its tile (the slot is (tile of the launch) * 4 + wave)
      if (missing) {
        const uint64_t li = (e.w & 0x1fffffffu) / (MFX_BLOCK / 64);
        const uint64_t tile = a.part_n == 1 ? a.tile_begin + li
                                             : ((((li >> a.part_shift) * a.part_n + a.part_rank) << a.part_shift) | (li & ((1ull << a.part_shift) - 1ull)));
        ctg = a.tile_contig[tile];
      }
      // the missing k-mers of a contig: the lanes of a wave that hold the same contig add together (a list runs along the sequence)
      uint64_t todo = __ballot(missing);
      while (todo) {
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const uint32_t lc = (uint32_t)__shfl((int)ctg, leader, 64);
        const uint64_t same = __ballot(missing && ctg == lc) & todo;
        if ((int)(tid & 63u) == leader) atomicAdd(reinterpret_cast<unsigned long long *>(&c_kmis[lc]), (unsigned long long)__popcll(same));
        todo &= ~same;
      }
      if (missing) n_missing++;
    }
  }
  if (!ready) return;                                          // (block-uniform)
  {
    uint64_t x = 0, y = n_missing, z = n_over0;
    mfx_block_sum3(x, y, z, H.red);
    if (tid == 0) {
      if (y) atomicAdd(reinterpret_cast<unsigned long long *>(&c_glob[1]), (unsigned long long)y);
      if (z) atomicAdd(reinterpret_cast<unsigned long long *>(&ka.counts[ka.nbins]), (unsigned long long)z);
    }
  }
  mfx_hist_lds_flush_bins(H, ka);
}

// ===========================================================================
// Sharded index (BASELINE config 5: the read DB does not fit one GPU).
// The table is split by OWNER rank = f(minimizer) (f(k-mer) under plain hashing);
// every rank keeps the k-mers it owns in an ordinary local table.  -hist then is
//   source rank : extract canonical k-mers of its tiles, label each with its owner
//                 (mfx_route_kernel), group by owner (stable radix sort), exchange;
//   owner rank  : probe + K* + bin the k-mers it receives (mfx_hist_keys_kernel);
//   all ranks   : one all-reduce of the counts image.
// ===========================================================================
__global__ __launch_bounds__(MFX_BLOCK) void mfx_route_kernel(mfx_route_args a) {
  __shared__ mfx_tile_lds L;
  __shared__ uint64_t s_red[MFX_BLOCK / 64][3];
  __shared__ uint32_t s_dest[256];
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  s_dest[tid] = 0;
  for (uint64_t tile = a.tile_begin + blockIdx.x; tile < a.tile_end; tile += gridDim.x) {
    uint32_t lo = 0, hi = a.ncontigs;
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (a.tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    const uint32_t c = lo;
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    __syncthreads();
    uint64_t n_valid = 0, z1 = 0, z2 = 0;
    const uint64_t obase = (tile - a.tile_begin) * MFX_TILE;
    for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
      const uint32_t p = b * MFX_BLOCK + tid;
      uint64_t f;
      const bool ok = mfx_tile_kmer(L, k, p, f) && p < n;
      uint64_t key = ~0ULL;
      uint32_t own = 255u;
      if (ok) {
        const uint64_t r = mfx_revcomp(f, k);
        key = f < r ? f : r;
        own = mfx_owner(a.t, key, f < r ? r : f, a.nranks);
        atomicAdd(&s_dest[own], 1u);
        n_valid++;
      }
      a.keys[obase + p] = key;
      a.owner[obase + p] = (uint8_t)own;
    }
    mfx_block_sum3(n_valid, z1, z2, s_red);
    if (tid == 0 && n_valid) {                                   // merfin-histogram.C:58, counted where the sequence lives
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 0], n_valid);
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 3 + c], n_valid);
    }
  }
  __syncthreads();
  if (tid < a.nranks && s_dest[tid]) atomicAdd((unsigned long long *)&a.dest_counts[tid], (unsigned long long)s_dest[tid]);
}

// ---------------------------------------------------------------------------
// Sort-free router for small worlds (nranks <= MFX_SPLIT_MAX_RANKS).  The output is the one the
// stable sort produces -- k-mers grouped by owner, sequence order inside an owner -- but it is
// computed as a counting split with the tile as the unit:
//   count   : k-mers of every tile per owner (also kasm, as mfx_route_kernel)
//   scan    : per owner, exclusive prefix of those counts over the tiles; owner totals
//   scatter : every tile again; wave w owns positions [1024w, 1024w+1024) of the tile, so
//             rank-in-owner = (owner's base) + (tile prefix) + (earlier waves of the tile) +
//             (earlier rounds of this wave) + (lower lanes of this round: one ballot).
// No position-sized scratch (the sort path writes 9 B, sorts 5 B and gathers 12 B per position).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MFX_BLOCK) void mfx_route_count_kernel(mfx_route_args a) {
  __shared__ mfx_tile_lds L;
  __shared__ uint64_t s_red[MFX_BLOCK / 64][3];
  __shared__ uint32_t s_cnt[MFX_SPLIT_MAX_RANKS];
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  for (uint64_t tile = a.tile_begin + blockIdx.x; tile < a.tile_end; tile += gridDim.x) {
    const uint32_t c = a.tile_contig[tile];
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();
    if (tid < MFX_SPLIT_MAX_RANKS) s_cnt[tid] = 0u;
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    __syncthreads();
    uint64_t n_valid = 0, z1 = 0, z2 = 0;
    for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
      const uint32_t p = b * MFX_BLOCK + tid;
      uint64_t f;
      if (mfx_tile_kmer(L, k, p, f) && p < n) {
        const uint64_t r = mfx_revcomp(f, k);
        atomicAdd(&s_cnt[mfx_owner(a.t, f < r ? f : r, f < r ? r : f, a.nranks)], 1u);
        n_valid++;
      }
    }
    mfx_block_sum3(n_valid, z1, z2, s_red);                       // barriers inside: s_cnt is complete after it
    if (tid == 0 && n_valid) {                                   // merfin-histogram.C:58, counted where the sequence lives
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 0], n_valid);
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 3 + c], n_valid);
    }
    if (tid < a.nranks) a.tile_cnt[(tile - a.tile_begin) * a.nranks + tid] = s_cnt[tid];
  }
}

// block d: exclusive prefix over the tiles of owner d's counts (in place), total -> dest_counts[d]
__global__ __launch_bounds__(MFX_BLOCK) void mfx_route_scan_kernel(uint32_t *tile_cnt, uint64_t ntiles, uint32_t nranks, uint64_t *dest_counts) {
  __shared__ uint64_t s[MFX_BLOCK];
  const uint32_t d = blockIdx.x, tid = threadIdx.x;
  const uint64_t per = (ntiles + MFX_BLOCK - 1) / MFX_BLOCK;
  const uint64_t t0 = (uint64_t)tid * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  uint64_t sum = 0;
  for (uint64_t t = t0; t < t1; ++t) sum += tile_cnt[t * nranks + d];
  s[tid] = sum;
  __syncthreads();
  if (tid == 0) {                                                 // 256 values: a serial pass is cheaper than it looks
    uint64_t run = 0;
    for (uint32_t i = 0; i < MFX_BLOCK; ++i) { const uint64_t v = s[i]; s[i] = run; run += v; }
    dest_counts[d] = run;
  }
  __syncthreads();
  uint64_t run = s[tid];
  for (uint64_t t = t0; t < t1; ++t) {
    const uint32_t v = tile_cnt[t * nranks + d];
    tile_cnt[t * nranks + d] = (uint32_t)run;                     // < 2^31 positions per call
    run += v;
  }
}

__global__ __launch_bounds__(MFX_BLOCK) void mfx_route_scatter_kernel(mfx_route_args a, uint64_t *keys_out, uint32_t *contig_out) {
  __shared__ mfx_tile_lds L;
  __shared__ uint32_t s_wtot[MFX_BLOCK / 64][MFX_SPLIT_MAX_RANKS];
  __shared__ uint64_t s_base[MFX_SPLIT_MAX_RANKS];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const int k = a.t.k;
  constexpr uint32_t ROUNDS = MFX_TILE / MFX_BLOCK;               // 16 rounds of 64 positions per wave
  constexpr uint32_t WSPAN = MFX_TILE / (MFX_BLOCK / 64);         // 1024 positions per wave
  if (tid == 0) {
    uint64_t run = 0;
    for (uint32_t d = 0; d < a.nranks; ++d) { s_base[d] = run; run += a.dest_counts[d]; }
  }
  for (uint64_t tile = a.tile_begin + blockIdx.x; tile < a.tile_end; tile += gridDim.x) {
    const uint32_t c = a.tile_contig[tile];
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();                                              // previous tile consumed (L, s_wtot); s_base visible
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    __syncthreads();
    uint64_t key[ROUNDS];
    uint32_t own[ROUNDS];
#pragma unroll
    for (uint32_t r = 0; r < ROUNDS; ++r) {
      const uint32_t p = wave * WSPAN + r * 64u + lane;
      uint64_t f;
      own[r] = 0xffu;
      key[r] = 0;
      if (mfx_tile_kmer(L, k, p, f) && p < n) {
        const uint64_t rc = mfx_revcomp(f, k);
        key[r] = f < rc ? f : rc;
        own[r] = mfx_owner(a.t, key[r], f < rc ? rc : f, a.nranks);
      }
    }
    for (uint32_t d = 0; d < a.nranks; ++d) {                      // this wave's k-mers per owner
      uint32_t cnt = 0;
#pragma unroll
      for (uint32_t r = 0; r < ROUNDS; ++r) cnt += (uint32_t)__popcll(__ballot(own[r] == d));
      if (lane == 0) s_wtot[wave][d] = cnt;
    }
    __syncthreads();
    const uint32_t *toff = a.tile_cnt + (tile - a.tile_begin) * a.nranks;
    for (uint32_t d = 0; d < a.nranks; ++d) {
      uint64_t run = s_base[d] + toff[d];
      for (uint32_t w = 0; w < wave; ++w) run += s_wtot[w][d];
#pragma unroll
      for (uint32_t r = 0; r < ROUNDS; ++r) {
        const bool mine = own[r] == d;
        const uint64_t m = __ballot(mine);
        if (mine) {
          const uint64_t o = run + (uint64_t)__popcll(m & ((1ULL << lane) - 1ULL));
          keys_out[o] = key[r];
          contig_out[o] = c;
        }
        run += (uint64_t)__popcll(m);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// ONE-PASS router of the one-process sharded -hist (mfx_hist_run_sharded): count, scan and scatter above read and decode every
// tile twice and need the per-tile prefix in between -- all to keep the k-mers of an owner in sequence order, which only the
// ordered fp64 koverCpy sum of the owner needs.  Here the owner sums koverCpy in FIXED POINT (mfx_hist_keys_kernel<true>: integer
// adds commute), so order is free: a block decodes a tile once, keeps its k-mers in registers, reserves room for them in every
// owner's REGION of the output with one atomic per (tile, owner), and writes them.  regions: owner d's k-mers start at
// d * region_cap; cursors[d] counts them; cursors[nranks] is raised when a region would overflow (the host then routes the round
// again with the exact, packed path -- a round so unbalanced needs a pathological sequence).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(MFX_BLOCK, 6) void mfx_route_fused_kernel(mfx_route_args a, uint64_t *keys_out, uint32_t *contig_out,
                                                                    unsigned long long *cursors, uint64_t region_cap) {
  __shared__ mfx_tile_lds L;
  __shared__ uint32_t s_cnt[MFX_SPLIT_MAX_RANKS];
  __shared__ uint64_t s_base[MFX_SPLIT_MAX_RANKS];
  __shared__ uint64_t s_red[MFX_BLOCK / 64][3];
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  constexpr uint32_t ROUNDS = MFX_TILE / MFX_BLOCK;               // 16 positions per lane and tile
  constexpr uint32_t MFX_ROUTE_PART = 4;
  for (uint64_t tile = a.tile_begin + blockIdx.x; tile < a.tile_end; tile += gridDim.x) {
    const uint32_t c = a.tile_contig[tile];
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();                                              // previous tile consumed (L)
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    uint64_t n_valid = 0, z1 = 0, z2 = 0;
    // order inside an owner's group is free (the owners sum koverCpy in fixed point), so a k-mer's place in its tile's share of
    // the group is simply what an LDS atomic hands out: no ballots, no per-wave prefixes.  A tile goes in PARTS of MFX_ROUTE_PART
    // positions per lane: the k-mers wait in registers between the count and the write, and 16 of them cost the occupancy
    for (uint32_t part = 0; part < ROUNDS; part += MFX_ROUTE_PART) {
      __syncthreads();                                            // tile filled / the previous part's s_cnt and s_base consumed
      if (tid < MFX_SPLIT_MAX_RANKS) s_cnt[tid] = 0u;
      __syncthreads();
      uint64_t key[MFX_ROUTE_PART];
      uint32_t where[MFX_ROUTE_PART];                             // owner << 16 | place in the part's share (0xffffffff: no k-mer)
#pragma unroll
      for (uint32_t r = 0; r < MFX_ROUTE_PART; ++r) {
        const uint32_t p = (part + r) * MFX_BLOCK + tid;           // lane-consecutive positions: neighbours (one minimizer, one owner) stay neighbours
        uint64_t f;
        where[r] = 0xffffffffu;
        key[r] = 0;
        if (mfx_tile_kmer(L, k, p, f) && p < n) {
          const uint64_t rc = mfx_revcomp(f, k);
          key[r] = f < rc ? f : rc;
          const uint32_t own = mfx_owner(a.t, key[r], f < rc ? rc : f, a.nranks);
          where[r] = (own << 16) | atomicAdd(&s_cnt[own], 1u);     // (at most 4096 per tile)
          n_valid++;
        }
      }
      __syncthreads();                                            // s_cnt complete
      if (tid < a.nranks) {                                       // room for this part's k-mers in every owner's region
        const uint32_t tot = s_cnt[tid];
        uint64_t base = ~0ull;
        if (tot) {
          base = atomicAdd(&cursors[tid], (unsigned long long)tot);
          if (base + tot > region_cap) { atomicAdd(&cursors[a.nranks], 1ull); base = ~0ull; }    // would overflow: nothing is written, the host re-routes
          else base += (uint64_t)tid * region_cap;
        }
        s_base[tid] = base;
      }
      __syncthreads();
#pragma unroll
      for (uint32_t r = 0; r < MFX_ROUTE_PART; ++r) {
        if (where[r] == 0xffffffffu) continue;
        const uint64_t base = s_base[where[r] >> 16];
        if (base == ~0ull) continue;
        const uint64_t o = base + (where[r] & 0xffffu);
        keys_out[o] = key[r];
        contig_out[o] = c;
      }
    }
    mfx_block_sum3(n_valid, z1, z2, s_red);
    if (tid == 0 && n_valid) {                                    // merfin-histogram.C:58, counted where the sequence lives
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 0], n_valid);
      atomicAdd((unsigned long long *)&a.counts[2ull * a.nbins + 3 + c], n_valid);
    }
  }
}

// gathers the routed k-mers into owner order (idx = stable-sorted positions) and attaches the contig id
__global__ void mfx_route_gather_kernel(mfx_route_args a, const uint32_t *idx, uint64_t nvalid, uint64_t *keys_out, uint32_t *contig_out) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nvalid; i += stride) {
    const uint32_t p = idx[i];
    keys_out[i] = a.keys[p];
    const uint64_t tile = a.tile_begin + p / MFX_TILE;
    uint32_t lo = 0, hi = a.ncontigs;
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (a.tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    contig_out[i] = lo;
  }
}

__global__ void mfx_iota_kernel(uint32_t *v, uint64_t n) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) v[i] = (uint32_t)i;
}

// owner side: canonical k-mers (grouped by source order) -> lookup -> K* -> bins.
// kasm was counted by the source; this adds kmissing (global + per contig), bins, koverCpy.
// SEGS: the k-mers come as up to MFX_KEYS_MAX_SEGS segments (the groups of the sources of a sharded run, evaluated where
// they lie: a group of the owner's own device is not copied first) and koverCpy is summed in FIXED POINT (every term
// truncated to a multiple of 2^-52, 128-bit integer accumulator a.kfix): integer adds commute, so the sum does not depend on
// the order the router's atomics left the k-mers in, nor on how the blocks cut them.  The host takes this form only when
// every prob is in [0, 4096) (terms stay below 2^64 * 2^-52); |sum - fp64 sum| <= terms * 2^-53.
template <bool SEGS>
__global__ __launch_bounds__(MFX_BLOCK, 4) void mfx_hist_keys_kernel(mfx_hist_keys_args a) {
  __shared__ mfx_mailbox MB;
  __shared__ mfx_hist_lds H;
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  const mfx_kstar_args &ka = a.ks;
  mfx_hist_lds_init(H, ka);
  const bool lut_ok = H.lut_ok != 0u;
  const uint64_t per = ((a.n + gridDim.x - 1) / gridDim.x + MFX_BLOCK * MFX_BATCH - 1) / (MFX_BLOCK * MFX_BATCH) * (MFX_BLOCK * MFX_BATCH);
  const uint64_t i0 = blockIdx.x * per, i1 = i0 + per < a.n ? i0 + per : a.n;
  uint64_t n_missing = 0, n_over0 = 0, zz = 0;
  double kover = 0.0;
  uint64_t fx_lo = 0, fx_hi = 0;                                   // SEGS: this lane's koverCpy terms, units of 2^-52
  uint64_t *c_glob = ka.counts + 2ull * ka.nbins;
  uint64_t *c_kmis = c_glob + 3 + ka.ncontigs;
  // Per-contig kmissing: the k-mers arrive in sequence order per source, so a block's slice is almost always
  // inside ONE contig.  Misses of that contig are counted in LDS (one wave-aggregated add) and flushed once;
  // only a slice that straddles contigs sends the others straight to the global counters.  (One global atomic
  // per missing k-mer made every block queue on the same few addresses: 5x the kernel time.)
  __shared__ uint32_t s_kmis;
  if (tid == 0) s_kmis = 0u;
  // the block's slice [i0, i1) of the concatenated segments, one segment piece after the other
  uint64_t seg_lo = 0;                                             // global index of the current segment's first k-mer
  bool have_bctg = false;
  uint32_t bctg = 0u;
  for (uint32_t sg = 0; sg < (SEGS ? a.nseg : 1u); ++sg) {
    const uint64_t seg_n = SEGS ? a.seg_n[sg] : a.n;
    const uint64_t *skeys = SEGS ? a.seg_keys[sg] : a.keys;
    const uint32_t *sctg = SEGS ? a.seg_contig[sg] : a.contig;
    const uint64_t lo = i0 > seg_lo ? i0 - seg_lo : 0, hi = i1 > seg_lo ? (i1 - seg_lo < seg_n ? i1 - seg_lo : seg_n) : 0;   // within the segment
    seg_lo += seg_n;
    if (hi <= lo) continue;                                        // (block-uniform)
    if (!have_bctg) {
      have_bctg = true;
      bctg = sctg[lo];                                             // block-uniform
      __syncthreads();
    }
    for (uint64_t base = lo; base < hi; base += MFX_BLOCK * MFX_BATCH) {       // block-uniform trip count
      uint64_t key[MFX_BATCH], krc[MFX_BATCH];
      uint32_t rv[MFX_BATCH], av[MFX_BATCH];
      bool ok[MFX_BATCH];
#pragma unroll
      for (int j = 0; j < MFX_BATCH; ++j) {
        const uint64_t i = base + (uint64_t)j * MFX_BLOCK + tid;
        ok[j] = i < hi;
        key[j] = ok[j] ? skeys[i] : 0ULL;
        krc[j] = mfx_revcomp(key[j], k);
      }
      mfx_group_lookup<MFX_BATCH>(a.t, MB, key, krc, ok, rv, av);
#pragma unroll
      for (int j = 0; j < MFX_BATCH; ++j) {
        double term = 0.0;
        const bool miss = ok[j] && mfx_hist_eval(H, ka, lut_ok, rv[j], av[j], n_over0, SEGS ? term : kover);
        if (SEGS && term > 0.0) {                                  // 0 + x == x exactly: `term` is this k-mer's own (1 - readK/asmK) * prob
          const uint64_t q = (uint64_t)(term * 4503599627370496.0);    // * 2^52, truncated
          fx_lo += q;
          fx_hi += fx_lo < q ? 1ull : 0ull;
        }
        const uint32_t cg = miss ? sctg[base + (uint64_t)j * MFX_BLOCK + tid] : bctg;
        const bool here = miss && cg == bctg;
        const uint64_t m = __ballot(here);                                      // wave-uniform control flow up to here
        if (m != 0ull && (tid & 63u) == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(&s_kmis, (uint32_t)__popcll(m));
        if (miss) {
          n_missing++;
          if (!here) atomicAdd((unsigned long long *)&c_kmis[cg], 1ull);
        }
      }
    }
  }
  mfx_block_sum3(n_missing, n_over0, zz, H.red);                              // has barriers: s_kmis is complete after it
  if (tid == 0) {
    if (n_missing) atomicAdd((unsigned long long *)&c_glob[1], n_missing);
    if (n_over0) atomicAdd((unsigned long long *)&ka.counts[ka.nbins], n_over0);
    if (s_kmis) atomicAdd((unsigned long long *)&c_kmis[bctg], (unsigned long long)s_kmis);
  }
  if (SEGS) {
    // 128-bit sum over the block, then into the launch's accumulator: the low word's wrap-arounds are carried into the high one
    // (each add of the low word reports the value it met)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ol = __shfl_down(fx_lo, o, 64), oh = __shfl_down(fx_hi, o, 64);
      fx_lo += ol;
      fx_hi += oh + (fx_lo < ol ? 1ull : 0ull);
    }
    __syncthreads();
    if ((tid & 63u) == 0) { H.red[tid >> 6][0] = fx_lo; H.red[tid >> 6][1] = fx_hi; }
    __syncthreads();
    if (tid == 0) {
      uint64_t lo = 0, hi = 0;
      for (uint32_t w = 0; w < MFX_BLOCK / 64; ++w) { lo += H.red[w][0]; hi += H.red[w][1] + (lo < H.red[w][0] ? 1ull : 0ull); }
      if (lo | hi) {
        const unsigned long long old = atomicAdd((unsigned long long *)&a.kfix[0], (unsigned long long)lo);
        hi += (old + lo < old) ? 1ull : 0ull;
        if (hi) atomicAdd((unsigned long long *)&a.kfix[1], (unsigned long long)hi);
      }
    }
    mfx_hist_lds_flush_bins(H, ka);
  } else {
    mfx_hist_lds_flush(H, ka, kover);
  }
}

// sums the partials in a fixed order and adds the result to *out; re-arms the tile scheduler counter
__global__ __launch_bounds__(MFX_BLOCK) void mfx_sum_partials_kernel(const double *partials, uint32_t n, double *out,
                                                                     uint64_t *ctr_reset) {
  __shared__ double s[MFX_BLOCK];
  double v = 0.0;
  for (uint32_t i = threadIdx.x; i < n; i += MFX_BLOCK) v = v + partials[i];
  s[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t st = MFX_BLOCK / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = out[0] + s[0];
    if (ctr_reset) ctr_reset[0] = 0;
  }
}

// out[0] = v[0] + v[1] + ... in index order (one thread: n = number of ranks); the multi-process koverCpy
__global__ void mfx_ordered_sum_kernel(const double *v, uint32_t n, double *out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (uint32_t i = 0; i < n; ++i) s = s + v[i];
    out[0] = s;
  }
}

// first level over the per-(tile, wave) values: block b sums in[b*MFX_SUM_CHUNK ...) in a fixed order
#define MFX_SUM_CHUNK 4096u
// FIXED: the values are integers in units of 2^-52 (mfx_hist_kernel / mfx_hist_rest_kernel), converted as they are read
template <bool FIXED>
__global__ __launch_bounds__(MFX_BLOCK) void mfx_sum_chunks_kernel(const double *in, uint64_t n, double *out) {
  __shared__ double s[MFX_BLOCK];
  const uint64_t base = (uint64_t)blockIdx.x * MFX_SUM_CHUNK;
  double v = 0.0;
  for (uint32_t i = threadIdx.x; i < MFX_SUM_CHUNK; i += MFX_BLOCK)
    if (base + i < n) v = v + (FIXED ? (double)reinterpret_cast<const uint64_t *>(in)[base + i] * 2.220446049250313e-16 : in[base + i]);
  s[threadIdx.x] = v;
  __syncthreads();
  for (uint32_t st = MFX_BLOCK / 2; st > 0; st >>= 1) {
    if (threadIdx.x < st) s[threadIdx.x] = s[threadIdx.x] + s[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = s[0];
}

// ===========================================================================
// -dump: raw (readV, asmV) per k-mer start position of one contig range
// ===========================================================================

// RECOUNT: no lookup -- readV already holds the values (summed over the shards of a sharded index by
// mfx_add_u32_kernel) and only the two counters are taken again from them: `readK == 0` is not additive over shards.
template <bool CANON, bool RECOUNT, bool COMPACT>
__global__ __launch_bounds__(MFX_BLOCK) void mfx_dump_kernel(mfx_dump_args a) {
  __shared__ mfx_tile_lds L;
  __shared__ mfx_mailbox MB;
  __shared__ uint64_t s_red[MFX_BLOCK / 64][3];
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  const uint64_t pos0 = (uint64_t)blockIdx.x * MFX_TILE;
  mfx_tile_fill(L, a.src + pos0);
  __syncthreads();
  uint64_t n_valid = 0, n_missing = 0, zz = 0;
  for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; b += MFX_BATCH) {
    if (pos0 + (uint64_t)b * MFX_BLOCK >= a.npos) break;          // last block of the range (block-uniform)
    uint64_t key[MFX_BATCH], key2[MFX_BATCH];
    uint32_t rv[MFX_BATCH], av[MFX_BATCH];
    bool     ok[MFX_BATCH], wr[MFX_BATCH];
#pragma unroll
    for (int j = 0; j < MFX_BATCH; ++j) {
      uint32_t p = (b + j) * MFX_BLOCK + tid;
      uint64_t gp = pos0 + p, f;
      wr[j] = gp < a.npos && gp >= a.skip;
      ok[j] = mfx_tile_kmer(L, k, p, f) && wr[j] && (gp < a.clen_left);
      uint64_t r = mfx_revcomp(f, k);
      if (CANON) { key[j] = f < r ? f : r; key2[j] = f < r ? r : f; }
      else { key[j] = f; key2[j] = r; }
    }
    if (RECOUNT) {
#pragma unroll
      for (int j = 0; j < MFX_BATCH; ++j) rv[j] = ok[j] ? a.readV[pos0 + (b + j) * MFX_BLOCK + tid - a.skip] : 0u;
    } else {
      if (COMPACT) MFX_COMPACT_LOOKUP(a.t, MB, key, key2, ok, rv, av);
      else mfx_group_lookup<MFX_BATCH>(a.t, MB, key, key2, ok, rv, av);
      if (!CANON) {
        uint32_t rv2[MFX_BATCH], av2[MFX_BATCH];
        if (COMPACT) MFX_COMPACT_LOOKUP(a.t, MB, key2, key, ok, rv2, av2);
        else mfx_group_lookup<MFX_BATCH>(a.t, MB, key2, key, ok, rv2, av2);
#pragma unroll
        for (int j = 0; j < MFX_BATCH; ++j) { rv[j] += rv2[j]; av[j] += av2[j]; }
      } else if ((k & 1) == 0) {                               // even k: a palindromic k-mer is value(fmer) + value(rmer) of ONE slot
#pragma unroll
        for (int j = 0; j < MFX_BATCH; ++j) if (key[j] == key2[j]) { rv[j] += rv[j]; av[j] += av[j]; }
      }
    }
#pragma unroll
    for (int j = 0; j < MFX_BATCH; ++j) {
      if (!wr[j]) continue;
      uint64_t gp = pos0 + (b + j) * MFX_BLOCK + tid;
      if (ok[j]) {
        n_valid++;                                                    // merfin-dump.C:48
        double readK, prob;
        mfx_getK_core(a.peak, a.n_prob, a.probK, a.probP, rv[j], readK, prob);
        if (readK == 0) n_missing++;                                  // :56-58
      }
      if (!RECOUNT) {
        a.readV[gp - a.skip] = ok[j] ? rv[j] : 0u;
        a.asmV[gp - a.skip] = ok[j] ? av[j] : 0u;
      }
    }
  }
  mfx_block_sum3(n_valid, n_missing, zz, s_red);
  if (tid == 0 && (n_valid | n_missing)) {
    atomicAdd((unsigned long long *)&a.stats[0], n_valid);
    atomicAdd((unsigned long long *)&a.stats[1], n_missing);
  }
}

// ===========================================================================
// varMer::score on the device (varMer.C:66-144): one lane per alternative PATH of a batch of variant clusters.  The paths'
// k-mers have been looked up by mfx_dump_kernel over the packed path text (readV / asmV per start position); this walks a
// path's bases in order exactly as the reference does -- run length of valid bases, the k-mer ENDING at idx, readK / prob by
// mfx_getK_core (prob keeps its previous value where no k-mer ends: varMer.C:77-90), missing count, and for -polish the
// delta-K terms |readK - asmK| * prob before / after the "new k-mer" bump of asmK (varMer.C:99-132, uint32 wrap of
// idxPath + 1 - k included) summed IN POSITION ORDER in fp64 -- so numM and totdk are the host's values bit for bit, and only
// 12 bytes per path come back instead of 8 bytes per base.
// ===========================================================================
__global__ __launch_bounds__(MFX_BLOCK) void mfx_var_score_kernel(mfx_var_score_args a) {
  const uint64_t p = (uint64_t)blockIdx.x * MFX_BLOCK + threadIdx.x;
  if (p >= a.npaths) return;
  const uint64_t o = a.off[p];
  const uint32_t slen = a.len[p], nv = a.nv[p], K = a.k;
  const int32_t *gtp = a.gt + a.voff[p];
  const uint32_t *vip = a.vidx + a.voff[p], *vlp = a.vlen + a.voff[p];
  const uint8_t *s = a.text + o;
  double prob = 1.0, totdk = 0.0;
  uint32_t numM = 0, run = 0;
  // `prob` is ONE variable per cluster in the reference (a local of varMer::score, which loops over the cluster's paths): a path
  // starts with what the last k-mer of the paths before it left there (1.0 before the first).  The lanes of a cluster's paths
  // run side by side, so each finds that value itself: the last valid k-mer of the nearest earlier path that has one.
  if (a.need_dk) {
    for (uint64_t q = p; q-- > a.cfirst[p];) {
      const uint8_t *sq = a.text + a.off[q];
      const uint32_t lq = a.len[q];
      uint32_t rq = 0, last = 0xffffffffu;
      for (uint32_t idx = 0; idx < lq; ++idx) {
        const uint32_t u = (uint32_t)(sq[idx] & 0xDFu) - 0x41u;
        rq = (u < 32u && ((0x00080045u >> u) & 1u)) ? rq + 1u : 0u;
        if (rq >= K) last = idx;
      }
      if (last != 0xffffffffu) {
        double rk;
        mfx_getK_core(a.peak, a.n_prob, a.probK, a.probP, a.readV[a.off[q] + last - (K - 1u)], rk, prob);
        break;
      }
    }
  }
  for (uint32_t idx = 0; idx < slen; ++idx) {
    const uint32_t u = (uint32_t)(s[idx] & 0xDFu) - 0x41u;                       // 'A' -> 0, 'C' -> 2, 'G' -> 6, 'T' -> 19, either case
    run = (u < 32u && ((0x00080045u >> u) & 1u)) ? run + 1u : 0u;
    double readK = 0.0, asmK = 0.0;
    if (run >= K) {                                                              // the k-mer ENDING at idx starts at idx - k + 1
      const uint64_t sp = o + idx - (K - 1u);
      mfx_getK_core(a.peak, a.n_prob, a.probK, a.probP, a.readV[sp], readK, prob);
      asmK = (double)a.asmV[sp];
    }
    if (readK == 0) numM++;
    if (!a.need_dk) continue;
    const double d0 = readK - asmK;
    const double oD = fabs(d0) * prob;                                           // varMer.C:99
    for (uint32_t j = 0; j < nv; ++j) {                                          // :103-112
      const uint32_t vi = vip[j], vl = vlp[j];
      if (gtp[j] > 0 && vi + 1u - K <= idx && idx < vi + vl + K) { asmK = asmK + 1.0; break; }
    }
    const double d1 = readK - asmK;
    const double nD = fabs(d1) * prob;                                           // :126
    const double dk = oD - nD;
    totdk = totdk + dk;
  }
  a.numM[p] = numM;
  if (a.need_dk) a.totdk[p] = totdk;
}
// `traverse` (merfin-variants.C:22-126) of a batch's clusters on the device: one cluster per WAVE runs mfx_traverse_cluster_t
// (mfx_traverse.h: the host's recursion as a loop; one thread per cluster with its strings in scratch took 13 ms per 65 K clusters,
// profiles/r05_cfg4_trv.txt) and writes its paths -- text, path-table entries, genotype / offset / length rows --
// into the room the host reserved (the product of the allele counts), then closes the slots it did not use (zero-length paths that
// belong to no cluster).  A few hundred bytes of work per cluster: what it saves is the host's enumeration and the upload of the path text.
struct mfx_trv_wave {                  // one wave per cluster: every lane runs the control flow, the byte loops are strided over the lanes
  static __device__ __forceinline__ uint32_t lane() { return threadIdx.x & 63u; }
  static __device__ __forceinline__ uint32_t lanes() { return 64u; }
  static __device__ __forceinline__ void sync() { __syncthreads(); }              // (the block IS the wave)
  static __device__ __forceinline__ bool all(bool p) { return __all((int)p) != 0; }
};
__global__ __launch_bounds__(64) void mfx_var_traverse_kernel(const mfx_trv_cluster *cl, uint64_t ncl, const mfx_trv_variant *vars, const mfx_trv_allele *alleles,
                                                              const char *win_text, const char *allele_text, mfx_trv_out o, uint32_t *np, uint32_t *status) {
  __shared__ char rep[MFX_TRV_MAX_NV][MFX_TRV_MAX_LEN];
  for (uint64_t c = blockIdx.x; c < ncl; c += gridDim.x) {
    const mfx_trv_cluster C = cl[c];
    uint32_t n = 0;
    const uint32_t st = mfx_traverse_cluster_t<mfx_trv_wave>(C, vars, alleles, win_text, allele_text, o, &n, rep);
    if (threadIdx.x == 0) { np[c] = n; status[c] = st; }
    for (uint32_t q = n + threadIdx.x; q < C.path_cap; q += 64u) {
      const uint64_t e = C.path0 + q;
      o.p_off[e] = C.text0;
      o.p_len[e] = 0u;
      o.p_nv[e] = 0u;
      o.p_voff[e] = 0u;
      o.p_cfirst[e] = o.table_base + e;
    }
    __syncthreads();                                                             // (the next cluster's strings reuse the LDS)
  }
}
hipError_t mfx_k_var_traverse(const mfx_trv_cluster *cl, uint64_t ncl, const mfx_trv_variant *vars, const mfx_trv_allele *alleles, const char *win_text,
                              const char *allele_text, const mfx_trv_out &o, uint32_t *np, uint32_t *status, hipStream_t st) {
  if (ncl == 0) return hipSuccess;
  const uint64_t blocks = ncl < 65536u ? ncl : 65536u;
  mfx_var_traverse_kernel<<<(unsigned)blocks, 64, 0, st>>>(cl, ncl, vars, alleles, win_text, allele_text, o, np, status);
  return hipGetLastError();
}

hipError_t mfx_k_var_score(const mfx_var_score_args &a, hipStream_t st) {
  if (a.npaths == 0) return hipSuccess;
  mfx_var_score_kernel<<<(unsigned)((a.npaths + MFX_BLOCK - 1) / MFX_BLOCK), MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}

// The streamed upload sends the 2-bit codes and, of the validity plane, only the words that are not all ones (gaps between
// contigs, the words behind a contig's end, N runs): the plane's range is filled with ones on the device and these are put in.
// exc[i] = {word index inside the range, word}
__global__ void mfx_valid_scatter_kernel(uint32_t *valid, const uint2 *exc, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) valid[exc[i].x] = exc[i].y;
}
hipError_t mfx_k_valid_scatter(uint32_t *valid, const uint64_t *exc, uint32_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  mfx_valid_scatter_kernel<<<(n + 255u) / 256u, 256, 0, st>>>(valid, reinterpret_cast<const uint2 *>(exc), n);
  return hipGetLastError();
}

// packed planes -> ASCII bases (for the kernels that read mfx_seq::d_bases after a packed upload): one thread per 16
// bases; invalid positions become 'N'
__global__ __launch_bounds__(MFX_BLOCK) void mfx_unpack_kernel(const uint64_t *codes, const uint32_t *valid, uint8_t *bases, uint64_t nwords) {
  const uint64_t stride = (uint64_t)gridDim.x * MFX_BLOCK;
  for (uint64_t i = (uint64_t)blockIdx.x * MFX_BLOCK + threadIdx.x; i < 2 * nwords; i += stride) {
    const uint64_t w = i >> 1;
    const uint32_t h = (uint32_t)i & 1u;
    const uint32_t c = (uint32_t)(codes[w] >> (h ? 0 : 32));           // 16 codes, first in the two highest bits
    const uint32_t v = (valid[w] >> (h ? 0 : 16)) & 0xffffu;
    uint32_t out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t x = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int j = q * 4 + b;
        const uint32_t code = (c >> (30 - 2 * j)) & 3u;
        const uint32_t ch = ((v >> (15 - j)) & 1u) ? ((0x47544341u >> (8 * code)) & 0xffu) : 0x4Eu;    // "ACTG"[code] or 'N'
        x |= ch << (8 * b);
      }
      out[q] = x;
    }
    *reinterpret_cast<uint4 *>(bases + 16 * i) = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

// ASCII bases -> packed planes, the whole buffer at once (the device-side twin of the host packer csrc/mfx_pack.cpp; the
// per-tile form is mfx_tile_fill): one thread per 16 bases
__global__ __launch_bounds__(MFX_BLOCK) void mfx_pack_kernel(const uint8_t *bases, uint64_t *codes, uint32_t *valid, uint64_t nwords) {
  const uint64_t stride = (uint64_t)gridDim.x * MFX_BLOCK;
  uint32_t *c32 = reinterpret_cast<uint32_t *>(codes);
  uint16_t *v16 = reinterpret_cast<uint16_t *>(valid);
  for (uint64_t i = (uint64_t)blockIdx.x * MFX_BLOCK + threadIdx.x; i < 2 * nwords; i += stride) {
    const uint4 v = *reinterpret_cast<const uint4 *>(bases + 16 * i);
    uint32_t c, ok;
    mfx_pack16(v, c, ok);
    c32[i ^ 1] = c;                      // first 16 bases of a 32-base word are its HIGH half
    v16[i ^ 1] = (uint16_t)ok;
  }
}

// Digest of a sequence's CONTENT in the library's own terms -- the 2-bit codes of the valid bases and the validity bits of
// every 32-base word of the padded buffer -- so that it does not depend on how the sequence is held (one byte per base or
// packed planes), on letter case or on what the invalid bytes were.  A wrapping sum of per-word hashes: any launch shape
// and any chunking give the same value.  A sequence-only index records the digest of the sequence it was claimed from and
// evaluations of another sequence are refused (mfx_api.cpp: seq_digest32).
__device__ __forceinline__ uint64_t mfx_double_bits(uint32_t v) {       // bit b of v -> bits 2b and 2b+1
  uint64_t x = v;
  x = (x | (x << 16)) & 0x0000FFFF0000FFFFULL;
  x = (x | (x << 8)) & 0x00FF00FF00FF00FFULL;
  x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0FULL;
  x = (x | (x << 2)) & 0x3333333333333333ULL;
  x = (x | (x << 1)) & 0x5555555555555555ULL;
  return x | (x << 1);
}
__global__ __launch_bounds__(MFX_BLOCK) void mfx_seq_digest_kernel(const uint8_t *bases, const uint64_t *codes, const uint32_t *valid,
                                                                   uint64_t nwords, unsigned long long *out) {
  const uint64_t stride = (uint64_t)gridDim.x * MFX_BLOCK;
  uint64_t acc = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * MFX_BLOCK + threadIdx.x; w < nwords; w += stride) {
    uint64_t c;
    uint32_t v;
    if (codes) { c = codes[w]; v = valid[w]; }
    else {
      uint32_t c0, v0, c1, v1;
      mfx_pack16(*reinterpret_cast<const uint4 *>(bases + 32 * w), c0, v0);
      mfx_pack16(*reinterpret_cast<const uint4 *>(bases + 32 * w + 16), c1, v1);
      c = ((uint64_t)c0 << 32) | c1;                         // first 16 bases of a 32-base word are its HIGH half
      v = (v0 << 16) | v1;
    }
    if (v) acc += mfx_hash64((c & mfx_double_bits(v)) ^ (w * 0x9E3779B97F4A7C15ULL)) + mfx_hash64((uint64_t)v + w * 0xD6E8FEB86659FD93ULL + 1ULL);
  }
  acc = mfx_wave_sum(acc);
  if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

// dst[i] += src[i]: the value arrays of the shards of one index add up to the whole index's values (every k-mer has
// exactly one owner; the other shards answer 0)
__global__ __launch_bounds__(MFX_BLOCK) void mfx_add_u32_kernel(uint32_t *dst, const uint32_t *src, uint64_t n) {
  const uint64_t stride = (uint64_t)gridDim.x * MFX_BLOCK;
  for (uint64_t i = (uint64_t)blockIdx.x * MFX_BLOCK + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// ===========================================================================
// assembly k-mer counting (`meryl count` of -sequence): canonical k-mers of
// every tile are inserted with asmV += 1.
// ===========================================================================

// MODE 0 (default): per-lane walk; MODE 1: cooperative batched passes.  Consecutive k-mers share their minimizer's
// line, so neighbouring lanes contend for the same slots: the per-lane walk simply retries on its own, while the
// cooperative passes serialise a group's same-line keys (1 Gb: 27.6 vs 14.0 G k-mers/s when the k-mers are already
// in the table -- the real case, the read database holds most assembly k-mers -- and 10.7 vs 9.0 into an empty one).
template <int MODE>
__global__ __launch_bounds__(MFX_BLOCK) void mfx_count_kernel(mfx_count_args a) {
  __shared__ mfx_tile_lds L;
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  uint32_t fresh = 0;
  for (uint64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    uint32_t lo = 0, hi = a.ncontigs;
    while (hi - lo > 1) {
      uint32_t mid = lo + (hi - lo) / 2;
      if (a.tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    // skip empty contigs that share this tile_start
    uint32_t c = lo;
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();
    if (a.codes) {                                           // block-uniform: the sequence is there as its packed planes
      const uint64_t w0 = (a.contig_off[c] + pos0) >> 5;
      mfx_tile_fill_packed(L, a.codes + w0, a.valid + w0);
    } else {
      mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    }
    __syncthreads();
    for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
      if (b * MFX_BLOCK >= n) break;                         // short last tile (block-uniform)
      uint32_t p = b * MFX_BLOCK + tid;                      // lane-consecutive positions: a group's 8 keys are neighbours
      uint64_t f;
      bool ok = mfx_tile_kmer(L, k, p, f) && p < n;
      const uint64_t r = mfx_revcomp(f, k);
      const uint64_t key = f < r ? f : r;
      if (ok && a.t.shard_n > 1 && mfx_owner(a.t, key, f < r ? r : f, a.t.shard_n) != a.t.shard_rank) ok = false;
      if (a.count == 2) {
        // count ONLY what was claimed before (a sequence-only index whose k-mers come from PART of the assembly -- the contigs one
        // device evaluates -- takes the assembly counts of those k-mers from the WHOLE assembly): find, add, never claim
        if (a.t.compact) {
          if (ok) {
            // the first mini-bucket of the k-mer's order answers for most: the k-mer, or an empty slot before it (never claimed)
            const mfx_probe pr = mfx_home(a.t, key);
            unsigned long long *mb = reinterpret_cast<unsigned long long *>(a.t.slots) + mfx_probe_line(a.t, pr, 0) * MFX_CSLOTS_LINE + 2u * pr.b0;
            const uint4 s4 = *reinterpret_cast<const uint4 *>(mb);
            const uint64_t x = (uint64_t)s4.x | ((uint64_t)s4.y << 32), y = (uint64_t)s4.z | ((uint64_t)s4.w << 32);
            unsigned long long *w = nullptr;
            unsigned long long cur = 0;
            bool beyond = false;
            if (x == MFX_EMPTY) { }
            else if ((x >> 22) == pr.fkey) { w = mb; cur = x; }
            else if (y == MFX_EMPTY) { }
            else if ((y >> 22) == pr.fkey) { w = mb + 1; cur = y; }
            else w = mfx_c_find(a.t, pr, 0, cur, beyond);
            if (w) mfx_c_add(a.t, w, cur, key, 1u, 1, a.meta);
            else if (beyond) { mfx_slot *ss = mfx_find_slot(mfx_side_view(a.t), key); if (ss) atomicAdd(&ss->asmV, 1u); }
          }
        } else {
          uint32_t dropped = 0;
          mfx_group_insert<false>(a.t, key, ok ? 1u : 0u, 1, a.meta, dropped);
        }
      } else if (a.t.compact) {                               // 8-byte slots (sequence-only index)
        if (ok) {
          unsigned long long cur = 0;
          bool claimed;
          mfx_slot *ss;
          unsigned long long *w = mfx_c_claim(a.t, key, a.meta, fresh, cur, a.count ? 1u : 0u, claimed, ss);
          if (w && a.count && !claimed) mfx_c_add(a.t, w, cur, key, 1u, 1, a.meta);
          else if (ss && a.count) atomicAdd(&ss->asmV, 1u);      // beyond its candidate lines (quotient form): counted in the side table
        }
      } else if (MODE == 1 && a.count) mfx_group_insert(a.t, key, ok ? 1u : 0u, 1, a.meta, fresh);
      else if (ok) {
        mfx_slot *sl = mfx_claim(a.t, key, a.meta, fresh);
        if (sl && a.count) atomicAdd(&sl->asmV, 1u);           // count == 0: the key is claimed, its counts come from the databases
      }
    }
  }
  mfx_meta_flush(a.meta, fresh, 0u);
}

// ===========================================================================
// -completeness: one streaming pass over the joint table
// (merfin-completeness.C:70-117; asm-only k-mers are skipped, :106-109; the
// raw read value is used -- -min/-max do not apply there).
// ===========================================================================
__global__ __launch_bounds__(MFX_BLOCK) void mfx_completeness_kernel(mfx_table_view t, double peak, uint32_t n_prob,
                                                                     const uint32_t *probK, const double *probP,
                                                                     double *pieces /* [128]: total[64], undrc[64] */) {
  // The reference merges the 64 file pieces of the two databases separately
  // (piece = top 6 bits of the 2k-bit k-mer, merfin-completeness.C:56-66) and prints one
  // line per piece; readK values are integers, so the fp64 sums are exact in any order.
  __shared__ double s_tot[64], s_und[64];
  // readK of the read counts that occur all the time, evaluated once per block by the routine the loop would call
  // (identical doubles): the fp64 division per occupied slot is what held this pass at 0.42 of the HBM peak
  constexpr uint32_t NLUT = 1024;
  __shared__ double s_rk[NLUT];
  if (threadIdx.x < 64) { s_tot[threadIdx.x] = 0.0; s_und[threadIdx.x] = 0.0; }
  for (uint32_t v = threadIdx.x; v < NLUT; v += blockDim.x) {
    double rk, pr;
    mfx_getK_core(peak, n_prob, probK, probP, v, rk, pr);
    s_rk[v] = rk;
  }
  __syncthreads();
  const uint64_t nslots = t.nlines * MFX_SLOTS_LINE;
  const int pshift = 2 * t.k >= 6 ? 2 * t.k - 6 : 0;
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nslots; i += stride) {
    uint4 s = reinterpret_cast<const uint4 *>(t.slots)[i];
    uint64_t key = (uint64_t)s.x | ((uint64_t)s.y << 32);
    if (key == MFX_EMPTY || s.z == 0) continue;          // empty slot / asm-only k-mer (:106-109)
    double readK, prob;
    if (s.z < NLUT) readK = s_rk[s.z];
    else mfx_getK_core(peak, n_prob, probK, probP, s.z, readK, prob);
    const double asmK = (double)s.w;
    const uint32_t piece = (uint32_t)(key >> pshift) & 63u;
    atomicAdd(&s_tot[piece], readK);                     // :113
    if (readK > asmK) atomicAdd(&s_und[piece], readK - asmK);   // :115-116
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    if (s_tot[threadIdx.x] != 0.0) atomicAdd(&pieces[threadIdx.x], s_tot[threadIdx.x]);
    if (s_und[threadIdx.x] != 0.0) atomicAdd(&pieces[64 + threadIdx.x], s_und[threadIdx.x]);
  }
}

// ===========================================================================
// Diagnostic: the random-access LINE rate of this device's HBM -- independent 16-byte loads at uniformly random 128-byte
// lines of a table far beyond the caches, four in flight per lane (tools/ubench_gather.hip found the rate the same for
// 8 .. 16 bytes per lane, 1 .. 8 loads in flight and 4 .. 16 blocks per CU, for tables of 8 .. 160 GiB).  It is the roof
// of the index probe: bench.py runs it on the box of the measurement and reports the kernel's line rate against it.
// ===========================================================================
template <int ILP>
__global__ __launch_bounds__(256) void mfx_gather_rate_kernel(const uint4 *__restrict__ t, uint64_t nlines, int iters, uint64_t seed, uint64_t *out) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t acc = 0, ctr = seed + tid * 0x9e3779b97f4a7c15ULL;
  for (int it = 0; it < iters; ++it) {
    uint4 v[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      ctr += 0xD1B54A32D192ED03ULL;
      v[j] = t[__umul64hi(mfx_hash64(ctr), nlines * 2) * 4];  // 16 bytes at the head of a random 64-byte half of a random line
    }
#pragma unroll
    for (int j = 0; j < ILP; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;      // all four words: ONE 16-byte load (two used words compile to two 4-byte loads, and the load unit's request rate, not the HBM, is then the limit)
  }
  if (acc == 0x1234567ULL) out[0] = acc;                       // (keeps the loads)
}
hipError_t mfx_k_gather_rate(const void *table, uint64_t nlines, uint64_t *scratch, double *lines_per_s, hipStream_t st) {
  const int grid = 256 * 8;                                    // 2^19 lanes x 2^12 loads = 2.1 G line reads per timed launch
  hipEvent_t e0, e1;
  hipError_t e = hipEventCreate(&e0);
  if (e != hipSuccess) return e;
  e = hipEventCreate(&e1);
  if (e != hipSuccess) { (void)hipEventDestroy(e0); return e; }
  const uint4 *t = reinterpret_cast<const uint4 *>(table);
  mfx_gather_rate_kernel<4><<<grid, 256, 0, st>>>(t, nlines, 1024, 1, scratch);     // warm (a full-length launch: the table's pages touched, the clocks up)
  double best = 0;
  for (int v = 0; v < 4 && e == hipSuccess; ++v) {              // 4 and 8 loads in flight per lane, twice each: the best is the roof
    (void)hipEventRecord(e0, st);
    if (v & 1) mfx_gather_rate_kernel<8><<<grid, 256, 0, st>>>(t, nlines, 512, 77 + v, scratch);
    else       mfx_gather_rate_kernel<4><<<grid, 256, 0, st>>>(t, nlines, 1024, 77 + v, scratch);
    (void)hipEventRecord(e1, st);
    e = hipEventSynchronize(e1);
    float ms = 0;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    if (e == hipSuccess && ms > 0) best = std::max(best, (double)grid * 256.0 * 4096.0 / (ms * 1e-3));
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (e == hipSuccess) *lines_per_s = best;
  return e;
}

// ===========================================================================
// launch wrappers (called from mfx_api.cpp, which is compiled as plain C++)
// ===========================================================================
hipError_t mfx_k_table_init(mfx_slot *slots, uint64_t nslots, hipStream_t st) {
  mfx_table_init_kernel<<<4096, 256, 0, st>>>(slots, nslots);
  return hipGetLastError();
}
hipError_t mfx_k_table_add(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side,
                           uint64_t *meta, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (t.seq_only) {
    blocks = (n + 256 * MFX_UPD_BATCH - 1) / (256 * MFX_UPD_BATCH);
    if (blocks > 8192) blocks = 8192;
    mfx_table_update_kernel<<<(unsigned)blocks, 256, 0, st>>>(t, kmers, values, n, side, meta);
    return hipGetLastError();
  }
  const char *me = getenv("MFX_INSERT_MODE");             // read per call: tests switch it
  const int mode = me ? atoi(me) : 1;
  if (mode == 0)      mfx_table_add_kernel<0><<<(unsigned)blocks, 256, 0, st>>>(t, kmers, values, n, side, meta);
  else                mfx_table_add_kernel<1><<<(unsigned)blocks, 256, 0, st>>>(t, kmers, values, n, side, meta);
  return hipGetLastError();
}
hipError_t mfx_k_table_add_delta(mfx_table_view t, const uint64_t *payload, const uint64_t *dir, uint32_t nblocks, uint64_t n,
                                 uint64_t payload_base, int side, uint64_t *meta, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  mfx_table_add_delta_kernel<<<nblocks < 8192u ? nblocks : 8192u, 256, 0, st>>>(t, payload, dir, nblocks, n, payload_base, side, meta);
  return hipGetLastError();
}
// does this table take a placed database's records by their own placement?  (the compact layout under its default placement)
int mfx_k_table_takes_placed(const mfx_table_view &t) {
#if MFX_V_PLACE_OLDLINE || MFX_V_PLACE_WBUCKET
  if (!t.quot) return 0;                                       // (A/B builds of the direct form's placement: the records are placed anew)
#endif
  return t.compact && t.seq_only && t.k >= MFX_PLACE_MIN_K && t.k <= MFX_PLACE_MAX_K && t.mz_w == MFX_PLACE_W && t.mz_t == mfx_p_tlen(t.k) && t.shard_n <= 1;
}
hipError_t mfx_k_table_add_placed(mfx_table_view t, const uint64_t *payload, const uint64_t *dir, uint32_t nblocks, uint64_t n,
                                  uint64_t payload_base, int side, uint64_t *meta, hipStream_t st) {
  if (nblocks == 0) return hipSuccess;
  mfx_table_add_placed_kernel<<<nblocks < 8192u ? nblocks : 8192u, 256, 0, st>>>(t, payload, dir, nblocks, n, payload_base, side, meta, mfx_k_table_takes_placed(t));
  return hipGetLastError();
}
hipError_t mfx_k_place_keys(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  mfx_place_keys_kernel<<<(unsigned)blocks, 256, 0, st>>>(k, kmers, n, out);
  return hipGetLastError();
}
hipError_t mfx_k_table_value(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV,
                             hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  mfx_table_value_kernel<<<(unsigned)blocks, 256, 0, st>>>(t, kmers, n, readV, asmV);
  return hipGetLastError();
}
hipError_t mfx_k_table_export(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV,
                              unsigned long long *count, hipStream_t st) {
  mfx_table_export_kernel<<<4096, 256, 0, st>>>(t, kmers, readV, asmV, count);
  return hipGetLastError();
}
hipError_t mfx_k_hist(const mfx_hist_args &a, int grid, hipStream_t st) {
  // MFX_DEBUG_DYN_LDS: extra dynamic LDS per block, an occupancy knob for experiments only
  static const unsigned dyn = getenv("MFX_DEBUG_DYN_LDS") ? (unsigned)atoi(getenv("MFX_DEBUG_DYN_LDS")) : 0u;
  const char *ge = getenv("MFX_HIST_GENERIC");                                                    // A/B, tests: never the specialised instances (read per launch: tests switch it)
  const bool generic = ge && atoi(ge);
  if (a.dbg && a.canonical && a.t.compact && a.t.k == 21 && a.t.mz_w == 4 && a.t.mz_t == 6 && !generic) mfx_hist_kernel<true, true, 21, 4, 6, true><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.canonical && a.t.compact && a.t.k == 21 && a.t.mz_w == 4 && a.t.mz_t == 6 && !generic) mfx_hist_kernel<true, true, 21, 4, 6><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.canonical && a.t.compact && a.t.k == 21 && a.t.mz_w == 5 && a.t.mz_t == 7 && !generic) mfx_hist_kernel<true, true, 21, 5, 7><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.canonical && a.t.compact && a.t.quot && a.t.k == 31 && a.t.mz_w == 4 && a.t.mz_t == 4 && !generic) mfx_hist_kernel<true, true, 31, 4, 4><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.canonical && a.t.compact) mfx_hist_kernel<true, true, 0, 0, 0><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.canonical)                mfx_hist_kernel<true, false, 0, 0, 0><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else if (a.t.compact)                mfx_hist_kernel<false, true, 0, 0, 0><<<grid, MFX_BLOCK, dyn, st>>>(a);
  else                                 mfx_hist_kernel<false, false, 0, 0, 0><<<grid, MFX_BLOCK, dyn, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_hist_rest(const mfx_hist_args &a, int grid, hipStream_t st) {
  if (!a.wl) return hipSuccess;
  mfx_hist_rest_kernel<<<grid, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_route(const mfx_route_args &a, hipStream_t st) {
  uint64_t nt = a.tile_end - a.tile_begin;
  if (nt == 0) return hipSuccess;
  mfx_route_kernel<<<(unsigned)(nt < 4096 ? nt : 4096), MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_route_split(const mfx_route_args &a, uint64_t *keys_out, uint32_t *contig_out, hipStream_t st) {
  const uint64_t nt = a.tile_end - a.tile_begin;
  if (nt == 0) return hipSuccess;
  const unsigned grid = (unsigned)(nt < 4096 ? nt : 4096);
  mfx_route_count_kernel<<<grid, MFX_BLOCK, 0, st>>>(a);
  mfx_route_scan_kernel<<<a.nranks, MFX_BLOCK, 0, st>>>(a.tile_cnt, nt, a.nranks, a.dest_counts);
  mfx_route_scatter_kernel<<<grid, MFX_BLOCK, 0, st>>>(a, keys_out, contig_out);
  return hipGetLastError();
}
hipError_t mfx_k_route_gather(const mfx_route_args &a, const uint32_t *idx, uint64_t nvalid, uint64_t *keys_out,
                              uint32_t *contig_out, hipStream_t st) {
  if (nvalid == 0) return hipSuccess;
  mfx_route_gather_kernel<<<2048, 256, 0, st>>>(a, idx, nvalid, keys_out, contig_out);
  return hipGetLastError();
}
hipError_t mfx_k_iota(uint32_t *v, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  mfx_iota_kernel<<<2048, 256, 0, st>>>(v, n);
  return hipGetLastError();
}
hipError_t mfx_k_hist_keys(const mfx_hist_keys_args &a, int grid, hipStream_t st) {
  if (a.nseg) mfx_hist_keys_kernel<true><<<grid, MFX_BLOCK, 0, st>>>(a);
  else        mfx_hist_keys_kernel<false><<<grid, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_route_fused(const mfx_route_args &a, uint64_t *keys_out, uint32_t *contig_out, uint64_t *cursors, uint64_t region_cap, hipStream_t st) {
  const uint64_t nt = a.tile_end - a.tile_begin;
  if (nt == 0) return hipSuccess;
  mfx_route_fused_kernel<<<(unsigned)(nt < 4096 ? nt : 4096), MFX_BLOCK, 0, st>>>(a, keys_out, contig_out, reinterpret_cast<unsigned long long *>(cursors), region_cap);
  return hipGetLastError();
}
hipError_t mfx_k_sum_partials(const double *partials, uint32_t n, double *out, hipStream_t st) {
  mfx_sum_partials_kernel<<<1, MFX_BLOCK, 0, st>>>(partials, n, out, nullptr);
  return hipGetLastError();
}
hipError_t mfx_k_ordered_sum(const double *v, uint32_t n, double *out, hipStream_t st) {
  mfx_ordered_sum_kernel<<<1, 64, 0, st>>>(v, n, out);
  return hipGetLastError();
}
uint64_t mfx_k_tile_partials_words(uint64_t ntiles) {
  const uint64_t n = ntiles * (MFX_BLOCK / 64);
  return n + (n + MFX_SUM_CHUNK - 1) / MFX_SUM_CHUNK;
}
// koverCpy of a tile-driven launch: two fixed-order levels over the per-(tile, wave) values, then += *out
hipError_t mfx_k_sum_tile_partials(double *tile_partials, uint64_t ntiles, double *out, uint64_t *ctr_reset, hipStream_t st, int fixed) {
  const uint64_t n = ntiles * (MFX_BLOCK / 64);
  const uint64_t nch = (n + MFX_SUM_CHUNK - 1) / MFX_SUM_CHUNK;
  if (nch && fixed) mfx_sum_chunks_kernel<true><<<(unsigned)nch, MFX_BLOCK, 0, st>>>(tile_partials, n, tile_partials + n);
  else if (nch) mfx_sum_chunks_kernel<false><<<(unsigned)nch, MFX_BLOCK, 0, st>>>(tile_partials, n, tile_partials + n);
  mfx_sum_partials_kernel<<<1, MFX_BLOCK, 0, st>>>(tile_partials + n, (uint32_t)nch, out, ctr_reset);
  return hipGetLastError();
}
// the quotient form of the compact layout (22 <= k <= 31) is served by the per-lane probe only (an A/B build with
// -DMFX_V_LANEPROBE=0 keeps those k in 16-byte slots)
int mfx_k_quot_supported() { return MFX_V_LANEPROBE ? 1 : 0; }
// blocks of the -hist instance this table's evaluation launches that are resident per CU (the persistent grid's size)
int mfx_k_hist_resident_blocks(int compact, int k) {
  int nb = 0;
  const hipError_t e = !compact ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfx_hist_kernel<true, false, 0, 0, 0>, MFX_BLOCK, 0)
                       : k == 21 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfx_hist_kernel<true, true, 21, 4, 6>, MFX_BLOCK, 0)
                       : k == 31 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfx_hist_kernel<true, true, 31, 4, 4>, MFX_BLOCK, 0)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, mfx_hist_kernel<true, true, 0, 0, 0>, MFX_BLOCK, 0);
  if (e != hipSuccess || nb < 1) nb = 4;
  return nb;
}
hipError_t mfx_k_unpack(const uint64_t *codes, const uint32_t *valid, uint8_t *bases, uint64_t nwords, hipStream_t st) {
  if (nwords == 0) return hipSuccess;
  uint64_t blocks = (2 * nwords + MFX_BLOCK - 1) / MFX_BLOCK;
  if (blocks > 65536) blocks = 65536;
  mfx_unpack_kernel<<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(codes, valid, bases, nwords);
  return hipGetLastError();
}
hipError_t mfx_k_pack(const uint8_t *bases, uint64_t *codes, uint32_t *valid, uint64_t nwords, hipStream_t st) {
  if (nwords == 0) return hipSuccess;
  uint64_t blocks = (2 * nwords + MFX_BLOCK - 1) / MFX_BLOCK;
  if (blocks > 65536) blocks = 65536;
  mfx_pack_kernel<<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(bases, codes, valid, nwords);
  return hipGetLastError();
}
hipError_t mfx_k_seq_digest(const uint8_t *bases, const uint64_t *codes, const uint32_t *valid, uint64_t nwords, uint64_t *out, hipStream_t st) {
  if (nwords == 0) return hipSuccess;
  uint64_t blocks = (nwords + MFX_BLOCK - 1) / MFX_BLOCK;
  if (blocks > 4096) blocks = 4096;
  mfx_seq_digest_kernel<<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(bases, codes, valid, nwords, reinterpret_cast<unsigned long long *>(out));
  return hipGetLastError();
}
hipError_t mfx_k_add_u32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 4 * MFX_BLOCK - 1) / (4 * MFX_BLOCK);
  if (blocks > 65536) blocks = 65536;
  mfx_add_u32_kernel<<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(dst, src, n);
  return hipGetLastError();
}
hipError_t mfx_k_dump(const mfx_dump_args &a, hipStream_t st) {
  uint64_t blocks = (a.npos + MFX_TILE - 1) / MFX_TILE;
  if (blocks == 0) return hipSuccess;
  if (a.recount)                       mfx_dump_kernel<true, true, false><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else if (a.canonical && a.t.compact) mfx_dump_kernel<true, false, true><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else if (a.canonical)                mfx_dump_kernel<true, false, false><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else if (a.t.compact)                mfx_dump_kernel<false, false, true><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else                                 mfx_dump_kernel<false, false, false><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_count(const mfx_count_args &a, hipStream_t st) {
  if (a.ntiles == 0) return hipSuccess;
  uint64_t blocks = a.ntiles < 8192 ? a.ntiles : 8192;
  const char *me = getenv("MFX_COUNT_MODE");
  const int mode = me ? atoi(me) : 0;
  if (mode == 1)      mfx_count_kernel<1><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else                mfx_count_kernel<0><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_k_completeness(mfx_table_view t, double peak, uint32_t n_prob, const uint32_t *probK, const double *probP,
                              double *partials, int grid, hipStream_t st) {
  mfx_completeness_kernel<<<grid, MFX_BLOCK, 0, st>>>(t, peak, n_prob, probK, probP, partials);
  return hipGetLastError();
}
