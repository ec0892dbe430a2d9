// mfx_place.h -- WHERE a k-mer lies in the compact table, as arithmetic shared by host and device code.
//
// The compact layout of a sequence-only index (mfx_kernels.hip) puts a canonical k-mer into the 128-byte line of its
// MOD-MINIMIZER: of its four windows of m = k - 3 bases the one sampled by the k-mer's smallest t-mer (mfx_mod_window), as a
// canonical m-mer c.  The line is taken from the high bits of a BIJECTION of c -- top = mix(low 32 bits of c) ^ (high bits * C),
// line = (top * nlines) >> 32 -- and the first mini-bucket from where in the k-mer the minimizer stands and three more bits of top
// (quotient form: its window and strand, mfx_p_bucket; direct form: the sampling t-mer's offset, mfx_mod_place).  So the pieces
//     c (2m bits), s (c stands reversed in the k-mer), j (its window, 0..3), e (the 3 bases around it)
// ARE the k-mer, one to one, and the 64-bit number
//     P = top : 32 | high bits of c : 2m - 32 | s : 1 | j : 2 | e : 6                      (2k + 3 bits for k >= 19: 64 bits hold it for k <= 30)
// is one to one with it as well -- and ascending P means ascending line, for a table of ANY size.  A database whose records are
// sorted by P ("placed", mfx_db.cpp FLAT_PLACED) is therefore applied to the table line after line: every line is read once and
// written once instead of one random line per record (round 5; what the reference pays per run in load_Kmers,
// merfin-globals.C:114-163).  k = 31 (round 6): P takes 65 bits; the record holds P >> 1 and the strand bit s travels in its count field
// (mfx_p_encode_s below).  The quotient form of the table (22 <= k <= 31) keeps only what the line does not say of the same
// pieces (mfx_q_place).  Everything here is integer arithmetic on uint32 / uint64: host (the converter, the tests) and device
// (the table's kernels) agree by construction.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MFX_PHD __host__ __device__ __forceinline__
#else
#define MFX_PHD inline
#endif

constexpr int MFX_PLACE_W = 4;                 // windows of the compact layout's mod-minimizer
constexpr int MFX_PLACE_MIN_K = 13, MFX_PLACE_MAX_K = 31;     // k of a placed database (k = 31: P takes 65 bits -- its strand bit travels beside it, mfx_p_encode_s)
constexpr uint32_t MFX_PLACE_VERSION = 1u;     // of the functions below; a placed database records it (another version: refused)

// the sampling t-mer's length: 4 .. 7 with (k - t) % 4 == 3, so that a k-mer and its reverse complement sample the same window
MFX_PHD int mfx_p_tlen(int k) { return ((k + 1) & 3) + 4; }

MFX_PHD uint64_t mfx_p_revcomp(uint64_t fwd, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint64_t x = __brevll(fwd) >> (64 - 2 * k);
#else
  uint64_t x = fwd;
  x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
  x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
  x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
  x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
  x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
  x = (x >> 32) | (x << 32);
  x >>= (64 - 2 * k);
#endif
  x = ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);      // bits reversed: swap back inside each base
  const uint64_t mask = (~0ULL) >> (64 - 2 * k);
  return (x ^ 0xAAAAAAAAAAAAAAAAULL) & mask;                                          // complement = code ^ 2
}

MFX_PHD uint32_t mfx_p_tmer_order(uint32_t canonical_tmer) { return ((canonical_tmer * 0x9E3779B1u) >> 7) & 511u; }

// x: offset (from the left) of the smallest t-mer of `key` (order hash of the canonical t-mer, ties: the leftmost); wa / wb: the
// m-mer of window x mod w as it stands in `key`, and its reverse complement (rc = the reverse complement of key)
MFX_PHD void mfx_p_mod_window(uint64_t key, uint64_t rc, int k, int w, int t, uint32_t &x, uint64_t &wa, uint64_t &wb) {
  const uint32_t tmask = (1u << (2 * t)) - 1u;
  uint32_t best = 0xffffffffu;
  x = 0;
  for (int p = 0; p + t <= k; ++p) {
    const uint32_t a = (uint32_t)(key >> (2 * (k - t - p))) & tmask, b = (uint32_t)(rc >> (2 * p)) & tmask;
    const uint32_t o = mfx_p_tmer_order(a < b ? a : b);
    if (o < best) { best = o; x = (uint32_t)p; }
  }
  const int m = k - w + 1, j = (int)(x % (uint32_t)w);
  const uint64_t mmask = (~0ULL) >> (64 - 2 * m);
  wa = (key >> (2 * (w - 1 - j))) & mmask;
  wb = (rc >> (2 * j)) & mmask;
}

// top <-> low 32 bits of the minimizer, given its high bits (odd multiplications and xor-shifts: invertible)
MFX_PHD uint32_t mfx_p_mix(uint32_t lo, uint32_t hi) {
  uint32_t u = lo * 0x9E3779B1u;
  u ^= u >> 15; u *= 0x85EBCA77u; u ^= u >> 13;
  return u ^ (hi * 0xC2B2AE3Du);
}
MFX_PHD uint32_t mfx_p_unmix(uint32_t top, uint32_t hi) {
  uint32_t u = top ^ (hi * 0xC2B2AE3Du);
  u ^= u >> 13; u ^= u >> 26; u *= 0xB6C92F47u;              // the inverses of the steps above, last first
  u ^= u >> 15; u ^= u >> 30;
  return u * 0x0E8B2F51u;
}

// Line of a minimizer's `top`, and the first mini-bucket of the QUOTIENT form (22 <= k <= 31; the direct form takes the sampling
// t-mer's offset instead: mfx_kernels.hip, mfx_mod_place).  The mini-bucket comes from WHERE the minimizer stands -- the window counted
// in the orientation in which the minimizer reads canonical: j if it stands as it is in the (canonical) k-mer, 3 - j if it stands
// reversed.  The up to four k-mers around one occurrence of a minimizer see it at four different such places whatever strand each of
// THEM is canonical on (with j alone a k-mer at window 3 and a reverse-canonical one at window 0 meet in one mini-bucket), so they start
// at four different mini-buckets.
MFX_PHD uint32_t mfx_p_line(uint32_t top, uint32_t nlines) { return (uint32_t)(((uint64_t)top * nlines) >> 32); }
MFX_PHD uint32_t mfx_p_bucket(uint32_t top, uint32_t j, uint32_t sbit) { return (2u * (sbit ? 3u - j : j) + (top >> 3)) & 7u; }

// the pieces of a CANONICAL k-mer `key` (rc: its reverse complement): c, s, j, e as above; k >= 13 (the mod-minimizer of the layout)
MFX_PHD void mfx_p_parts(int k, uint64_t key, uint64_t rc, uint64_t &c, uint32_t &sbit, uint32_t &j, uint32_t &e) {
  uint64_t a, b;
  uint32_t x;
  mfx_p_mod_window(key, rc, k, MFX_PLACE_W, mfx_p_tlen(k), x, a, b);
  j = x & 3u;
  c = a < b ? a : b;
  sbit = b < a ? 1u : 0u;
  const int m = k - 3;
  e = (uint32_t)(((key >> (2 * (m + 3 - (int)j))) << (2 * (3 - (int)j))) | (key & ((1ull << (2 * (3 - (int)j))) - 1ull)));
}

// bits of c above its low 32 (0 for k <= 19)
MFX_PHD int mfx_p_hibits(int k) { const int m2 = 2 * (k - 3); return m2 > 32 ? m2 - 32 : 0; }

// canonical k-mer -> P (13 <= k <= 30)
MFX_PHD uint64_t mfx_p_encode(int k, uint64_t key) {
  uint64_t c;
  uint32_t sbit, j, e;
  mfx_p_parts(k, key, mfx_p_revcomp(key, k), c, sbit, j, e);
  const int R = mfx_p_hibits(k);
  const uint32_t hi = (uint32_t)(c >> 32), top = mfx_p_mix((uint32_t)c, hi);
  return ((uint64_t)top << (R + 9)) | ((uint64_t)hi << 9) | (uint64_t)(sbit | (j << 1) | (e << 3));
}

// P -> the canonical k-mer and the pieces a table needs to place it (top, j); a P no k-mer encodes gives a k-mer that does not
// encode back to it (the loaders check what they can: the k-mer's width)
// (pieces: meta / hi / top are handed in instead of cut from P -- mfx_p_decode_s below)
MFX_PHD uint64_t mfx_p_decode(int k, uint64_t P, uint32_t &top, uint32_t &hi, uint32_t &meta, bool pieces = false, uint32_t meta_in = 0, uint32_t hi_in = 0, uint32_t top_in = 0) {
  const int m = k - 3, R = mfx_p_hibits(k);
  meta = pieces ? meta_in : (uint32_t)P & 511u;
  hi = pieces ? hi_in : (uint32_t)(P >> 9) & (uint32_t)((1ull << R) - 1ull);
  top = pieces ? top_in : (uint32_t)(P >> (R + 9));
  const uint32_t sbit = meta & 1u, j = (meta >> 1) & 3u, e = (meta >> 3) & 63u;
  const uint64_t c = ((uint64_t)hi << 32) | mfx_p_unmix(top, hi);
  const uint64_t mmer = sbit ? mfx_p_revcomp(c, m) : c;
  const uint64_t left = e >> (2 * (3 - j)), right = e & ((1u << (2 * (3 - j))) - 1u);
  return (left << (2 * (m + 3 - (int)j))) | (mmer << (2 * (3 - j))) | right;
}
// k = 31: P = top : 32 | hi : 24 | s | j | e would take 65 bits.  A placed record then holds P >> 1 -- everything but the strand bit s, 64 bits,
// ascending P >> 1 is still ascending line -- and s travels beside it: bit 0 of the record's count field, the count above it (mfx_db.cpp,
// mfx_table_add_placed_kernel).  Two k-mers that differ in s alone share the stored number (s = 0 first).  k <= 30: s_out = 0, the number is P.
MFX_PHD bool mfx_p_split(int k) { return k > 30; }
MFX_PHD uint64_t mfx_p_encode_s(int k, uint64_t key, uint32_t &s_out) {
  if (!mfx_p_split(k)) { s_out = 0u; return mfx_p_encode(k, key); }
  uint64_t c;
  uint32_t sbit, j, e;
  mfx_p_parts(k, key, mfx_p_revcomp(key, k), c, sbit, j, e);
  const int R = mfx_p_hibits(k);
  const uint32_t hi = (uint32_t)(c >> 32), top = mfx_p_mix((uint32_t)c, hi);
  s_out = sbit;
  return ((uint64_t)top << (R + 8)) | ((uint64_t)hi << 8) | (uint64_t)(j | (e << 2));
}
MFX_PHD uint64_t mfx_p_decode_s(int k, uint64_t P, uint32_t s_in, uint32_t &top, uint32_t &hi, uint32_t &meta) {
  if (!mfx_p_split(k)) return mfx_p_decode(k, P, top, hi, meta);
  const int R = mfx_p_hibits(k);
  return mfx_p_decode(k, 0ull, top, hi, meta, true, ((uint32_t)(P & 255u) << 1) | (s_in & 1u), (uint32_t)(P >> 8) & (uint32_t)((1ull << R) - 1ull), (uint32_t)(P >> (R + 8)));
}
MFX_PHD int mfx_p_bits(int k) { return mfx_p_split(k) ? 64 : 32 + mfx_p_hibits(k) + 9; }      // of the STORED number: 2k + 3 for 19 <= k <= 30 (41 below: `top` has 32 bits whatever the minimizer's length)
