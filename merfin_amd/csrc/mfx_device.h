// mfx_device.h -- device code shared by the kernel files (mfx_kernels.hip: 64-bit k-mers, k <= 31, the measured
// hot path; mfx_wide.hip: 128-bit k-mers, 32 <= k <= 64): hashing helpers, the sequence tile in LDS, block sums and
// the K* / histogram stage (merfin-histogram.C:63-90).  Include from .hip files only.
#pragma once
#include "mfx_internal.h"
#include "mfx_kernels.h"

__device__ __forceinline__ uint64_t mfx_hash64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// floor(x * n / 2^64) for n < 2^32 (the table never has more lines: mfx_index_create): two 32-bit
// multiplies instead of the four a general 64x64 high product takes.  Same value as __umul64hi(x, n).
__device__ __forceinline__ uint32_t mfx_range32(uint64_t x, uint64_t n) {
  const uint32_t n32 = (uint32_t)n;
  const uint64_t t = (uint64_t)(uint32_t)(x >> 32) * n32 + (uint64_t)__umulhi((uint32_t)x, n32);
  return (uint32_t)(t >> 32);
}

// ===========================================================================
// sequence tile in LDS: 2-bit codes packed MSB-first in 64-bit words + one
// validity bit per base, so a lane extracts the k-mer starting at ANY position
// with two LDS reads and a funnel shift (no rolling dependency between lanes).
// ===========================================================================
constexpr uint32_t MFX_TILE_BYTES  = MFX_TILE + 64;          // tile + (k-1) halo, k <= 64
constexpr uint32_t MFX_TILE_CHUNKS = MFX_TILE_BYTES / 16;    // 16-byte global loads per tile
constexpr uint32_t MFX_TILE_WORDS  = MFX_TILE_BYTES / 32;    // 64-bit code words (32 bases each)

struct mfx_tile_lds {
  uint64_t codes[MFX_TILE_WORDS + 1];
  uint32_t valid[MFX_TILE_WORDS + 1];
};

// A=0 C=1 T=2 G=3 for either case = (c >> 1) & 3; valid iff (c & 0xDF) in ACGT.
__device__ __forceinline__ void mfx_pack16(uint4 v, uint32_t &codes, uint32_t &valid) {
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  codes = 0;
  valid = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      uint32_t c = (w[q] >> (8 * b)) & 0xffu;
      uint32_t u = (c & 0xDFu) - 0x41u;                       // 'A'->0 'C'->2 'G'->6 'T'->19
      uint32_t ok = (u < 32u) ? ((0x00080045u >> u) & 1u) : 0u;
      int i = q * 4 + b;
      codes |= ((c >> 1) & 3u) << (30 - 2 * i);
      valid |= ok << (15 - i);
    }
  }
}

// cooperative fill by a 256-thread block; src is 128-byte aligned
__device__ __forceinline__ void mfx_tile_fill(mfx_tile_lds &L, const uint8_t *__restrict__ src) {
  uint32_t *c32 = reinterpret_cast<uint32_t *>(L.codes);
  uint16_t *v16 = reinterpret_cast<uint16_t *>(L.valid);
  for (uint32_t ch = threadIdx.x; ch < MFX_TILE_CHUNKS; ch += MFX_BLOCK) {
    uint4 v = *reinterpret_cast<const uint4 *>(src + 16ull * ch);
    uint32_t codes, valid;
    mfx_pack16(v, codes, valid);
    c32[ch ^ 1] = codes;                // first 16 bases of a 32-base word are its HIGH half
    v16[ch ^ 1] = (uint16_t)valid;
  }
}

// the same tile from the PACKED planes of a sequence (mfx_seq::d_codes / d_valid: the whole buffer in the tile's own
// form, written by the host packer csrc/mfx_pack.cpp): a straight copy of 130 words; w0 = first word of the tile
__device__ __forceinline__ void mfx_tile_fill_packed(mfx_tile_lds &L, const uint64_t *__restrict__ codes, const uint32_t *__restrict__ valid) {
  for (uint32_t i = threadIdx.x; i < MFX_TILE_WORDS; i += MFX_BLOCK) {
    L.codes[i] = codes[i];
    L.valid[i] = valid[i];
  }
}

__device__ __forceinline__ uint64_t mfx_wave_sum(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of up to 3 uint64 values; result valid in thread 0
__device__ __forceinline__ void mfx_block_sum3(uint64_t &a, uint64_t &b, uint64_t &c, uint64_t (*scratch)[3]) {
  a = mfx_wave_sum(a); b = mfx_wave_sum(b); c = mfx_wave_sum(c);
  uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { scratch[wave][0] = a; scratch[wave][1] = b; scratch[wave][2] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = b = c = 0;
    for (uint32_t w = 0; w < MFX_BLOCK / 64; ++w) { a += scratch[w][0]; b += scratch[w][1]; c += scratch[w][2]; }
  }
}


// LDS state of the K* / histogram stage, shared by the sequence-driven kernel
// (mfx_hist_kernel) and the key-driven one (mfx_hist_keys_kernel, sharded index).
#ifndef MFX_V_LDS_DIET
#define MFX_V_LDS_DIET 1              // 1 (default since round 4): no prob / over-copy tables in LDS (16 KB less per block: they serve the `asmK > readK` branch only; 22.5 KB per block lets 7 blocks share a CU) -- A/B: tools/ab_build.sh -DMFX_V_LDS_DIET=0
#endif
struct mfx_hist_lds {
  uint32_t hist[2 * MFX_NB_LDS];
  // Exact lookup tables, filled with the SAME fp64 routines the generic path uses:
  //   rk/pr[v]     readK and prob of read count v < MFX_MAXP_LDS (prob table and peak rule merged)
  //   bin[h][l]    bin index of the ratio h/l, term[h][l] = 1 - l/h     (h, l < MFX_KLUT)
  uint32_t rk[MFX_MAXP_LDS];
#if !MFX_V_LDS_DIET
  double   pr[MFX_MAXP_LDS];
#endif
  uint16_t bin[MFX_KLUT * MFX_KLUT];
#if !MFX_V_LDS_DIET
  double   term[MFX_KLUT * MFX_KLUT];
#endif
  uint32_t lut_ok;
  uint32_t wl_n;                          // mfx_hist_kernel: entries this block has put on its segment of the worklist
  uint64_t next[2];                       // dynamic tile scheduler: the tile fetched for the next iteration
  uint64_t tot[2];                        // mfx_hist_kernel: valid / missing k-mers of the contigs this block already flushed (thread 0)
  uint64_t red[MFX_BLOCK / 64][3];
  double   dred[MFX_BLOCK];
};

__device__ __forceinline__ void mfx_hist_lds_init(mfx_hist_lds &H, const mfx_kstar_args &ka) {
  const uint32_t tid = threadIdx.x;
  for (uint32_t i = tid; i < 2 * MFX_NB_LDS; i += MFX_BLOCK) H.hist[i] = 0;
  if (tid == 0) H.lut_ok = 1u;
  __syncthreads();
  for (uint32_t v = tid; v < MFX_MAXP_LDS; v += MFX_BLOCK) {
    double rk, pr;
    mfx_getK_core(ka.peak, ka.n_prob, ka.probK, ka.probP, v, rk, pr);
    // the table holds readK as an integer; anything else disables the fast path for this launch
    if (!(rk >= 0.0 && rk < 4294967296.0 && rk == (double)(uint32_t)rk)) H.lut_ok = 0u;
    H.rk[v] = (uint32_t)rk;
#if !MFX_V_LDS_DIET
    H.pr[v] = pr;
#endif
  }
  for (uint32_t i = tid; i < MFX_KLUT * MFX_KLUT; i += MFX_BLOCK) {
    uint32_t h = i / MFX_KLUT, l = i % MFX_KLUT;
    uint32_t b = (h >= 1 && l >= 1 && h >= l) ? mfx_bin_index((double)h, (double)l) : 0u;
    H.bin[i] = (uint16_t)b;
#if !MFX_V_LDS_DIET
    H.term[i] = (h >= 1 && l >= 1 && h > l) ? mfx_overcopy_term((double)l, (double)h, 1.0) : 0.0;
#endif
  }
  __syncthreads();
}

// K* bins beyond the dense image (idx >= nbins).  The reference's arrays simply grow (increaseArray, merfin-histogram.C:74,87), so
// any NUMBER of k-mers may fall there -- an assembly's satellite array that the reads under-represent puts millions of positions
// into a handful of far bins.  They are therefore AGGREGATED on the device: an open-addressed table {bin key -> occurrences} of
// MFX_OVF_SLOTS entries per evaluator (key: bit 63 = `over`, low bits = the bin index), so that only the number of DISTINCT far bins
// is bounded.  The lanes of a wave that hold the same key combine first (one atomic per distinct key and wave: the positions
// of a tandem array hit the same bin from every lane).  ovf: [0] distinct keys, [1] occurrences that found no slot (the host
// reports MFX_E_OVERFLOW), [2, 2 + S) occurrences per slot, [2 + S, 2 + 2 S) keys (all ones = empty).
__device__ __forceinline__ void mfx_ovf_add(uint64_t *ovf, uint64_t key, uint64_t amount) {
  unsigned long long *cnt = reinterpret_cast<unsigned long long *>(ovf) + 2, *keys = cnt + MFX_OVF_SLOTS;
  uint64_t x = key * 0x9E3779B97F4A7C15ull;
  uint32_t h = (uint32_t)(x >> 40) & (MFX_OVF_SLOTS - 1u);
  for (uint32_t probe = 0; probe < MFX_OVF_PROBES; ++probe) {
    unsigned long long cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == ~0ull) {
      cur = atomicCAS(&keys[h], ~0ull, (unsigned long long)key);
      if (cur == ~0ull) { atomicAdd(reinterpret_cast<unsigned long long *>(ovf), 1ull); cur = key; }
    }
    if (cur == key) { atomicAdd(&cnt[h], (unsigned long long)amount); return; }
    h = (h + 1u) & (MFX_OVF_SLOTS - 1u);
  }
  atomicAdd(reinterpret_cast<unsigned long long *>(ovf) + 1, (unsigned long long)amount);
}

// called by the lanes of a wave that evaluated a k-mer into a far bin (any subset of the wave: a divergent branch)
__device__ __forceinline__ void mfx_ovf_record(const mfx_kstar_args &ka, bool under, uint32_t idx) {
  const uint64_t key = (under ? 0ull : (1ull << 63)) | idx;
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t todo = __ballot(1);                                 // the lanes that are here
  while (todo) {
    const int leader = __ffsll((unsigned long long)todo) - 1;
    const uint64_t lk = ((uint64_t)(uint32_t)__shfl((int)(key >> 32), leader, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)key, leader, 64);
    const uint64_t same = __ballot(key == lk) & todo;
    if ((int)lane == leader) {
      const uint64_t n = (uint64_t)__popcll(same);
      mfx_ovf_add(ka.ovf, key, n);
      atomicAdd((unsigned long long *)&ka.counts[2ull * ka.nbins + 2], (unsigned long long)n);
    }
    todo &= ~same;                                             // (wave-uniform among the lanes here: all of them leave together)
  }
}

// One evaluated k-mer: merfin-histogram.C:63-90 after the lookups.  Returns true when the
// k-mer is "missing" (readK == 0).
template <class Counter>
__device__ __forceinline__ bool mfx_hist_eval(mfx_hist_lds &H, const mfx_kstar_args &ka, bool lut_ok, uint32_t readV,
                                              uint32_t asmV, Counter &n_over0, double &kover) {
  double readK, prob;
  uint32_t rki = 0xffffffffu;                                  // readK as an integer when the tables apply
  if (lut_ok && readV < MFX_MAXP_LDS) {
#if MFX_V_LDS_DIET
    // prob is read by the `asmK > readK` branch alone: taken from the table itself there (merfin-globals.C:93-97: the table's
    // row for 1 <= readV <= rows, else 1)
    rki = H.rk[readV]; prob = 1.0; readK = (double)rki;
#else
    rki = H.rk[readV]; prob = H.pr[readV]; readK = (double)rki;
#endif
  } else {
    mfx_getK_core(ka.peak, ka.n_prob, ka.probK, ka.probP, readV, readK, prob);
  }
  const double asmK = (double)asmV;
  if (readK == 0) return true;                                 // :66-69
  const bool under = asmK > readK;                             // :71
  uint32_t idx;
  if (rki < MFX_KLUT && asmV < MFX_KLUT && asmV >= 1) {        // exact tables (same fp64 code, evaluated once)
    const uint32_t hi = under ? asmV : rki, lo = under ? rki : asmV;
    idx = H.bin[hi * MFX_KLUT + lo];
#if MFX_V_LDS_DIET
    if (under) {
      if (readV > 0 && readV <= ka.n_prob) prob = ka.probP[readV - 1];
      kover += mfx_overcopy_term(readK, asmK, prob);           // :81  (1 - readK/asmK) * prob
    }
#else
    if (under) kover += H.term[hi * MFX_KLUT + lo] * prob;     // :81  (1 - readK/asmK) * prob
#endif
  } else {
    idx = under ? mfx_bin_index(asmK, readK) : mfx_bin_index(readK, asmK);
#if MFX_V_LDS_DIET
    if (under && rki != 0xffffffffu && readV > 0 && readV <= ka.n_prob) prob = ka.probP[readV - 1];   // (readK came from the LDS table: prob was not looked up yet)
#endif
    if (under) kover += mfx_overcopy_term(readK, asmK, prob);  // :81
  }
  if (!under && idx == 0) { n_over0++; return false; }         // the dominant bin stays in a register
  uint64_t *c_undr = ka.counts, *c_over = ka.counts + ka.nbins;
  if (idx < MFX_NB_LDS) atomicAdd(&H.hist[(under ? 0 : MFX_NB_LDS) + idx], 1u);
  else if (idx < ka.nbins) atomicAdd((unsigned long long *)&(under ? c_undr : c_over)[idx], 1ull);
  else mfx_ovf_record(ka, under, idx);
  return false;
}

// The same for the tile-driven kernel, whose koverCpy is an integer sum (mfx_kfix): the over-copy term of the (read count, asmV) pairs
// the exact tables cover comes from ka.underq -- filled once per evaluator by the SAME fp64 routines (mfx_eval_create) -- instead of a
// probability load, an fp64 division and a conversion in a branch that some lane of most waves takes (1-2 % of the k-mers have
// asmK > readK: 60 % of the waves' evaluations had to walk through it).
template <class Counter, class Sum>
__device__ __forceinline__ bool mfx_hist_eval_fx(mfx_hist_lds &H, const mfx_kstar_args &ka, bool lut_ok, uint32_t readV,
                                                 uint32_t asmV, Counter &n_over0, Sum &kfx) {
  double readK, prob;
  uint32_t rki = 0xffffffffu;                                  // readK as an integer when the tables apply
  if (lut_ok && readV < MFX_MAXP_LDS) {
    rki = H.rk[readV]; prob = 1.0; readK = (double)rki;
  } else {
    mfx_getK_core(ka.peak, ka.n_prob, ka.probK, ka.probP, readV, readK, prob);
  }
  const double asmK = (double)asmV;
  if (readK == 0) return true;                                 // :66-69
  const bool under = asmK > readK;                             // :71
  uint32_t idx;
  if (rki < MFX_KLUT && asmV < MFX_KLUT && asmV >= 1) {        // exact tables (same fp64 code, evaluated once)
    const uint32_t hi = under ? asmV : rki, lo = under ? rki : asmV;
    idx = H.bin[hi * MFX_KLUT + lo];
    if (under) kfx += ka.underq[readV * MFX_KLUT + asmV];      // :81  (1 - readK/asmK) * prob
  } else {
    idx = under ? mfx_bin_index(asmK, readK) : mfx_bin_index(readK, asmK);
    if (under && rki != 0xffffffffu && readV > 0 && readV <= ka.n_prob) prob = ka.probP[readV - 1];   // (readK came from the LDS table: prob was not looked up yet)
    if (under) kfx += mfx_kfix(mfx_overcopy_term(readK, asmK, prob));  // :81
  }
  if (!under && idx == 0) { n_over0++; return false; }         // the dominant bin stays in a register
  uint64_t *c_undr = ka.counts, *c_over = ka.counts + ka.nbins;
  if (idx < MFX_NB_LDS) atomicAdd(&H.hist[(under ? 0 : MFX_NB_LDS) + idx], 1u);
  else if (idx < ka.nbins) atomicAdd((unsigned long long *)&(under ? c_undr : c_over)[idx], 1ull);
  else mfx_ovf_record(ka, under, idx);
  return false;
}

// LDS bins -> global (non-zero only)
__device__ __forceinline__ void mfx_hist_lds_flush_bins(mfx_hist_lds &H, const mfx_kstar_args &ka) {
  const uint32_t tid = threadIdx.x;
  uint64_t *c_undr = ka.counts, *c_over = ka.counts + ka.nbins;
  __syncthreads();
  for (uint32_t i = tid; i < 2 * MFX_NB_LDS; i += MFX_BLOCK) {
    uint32_t v = H.hist[i];
    if (v) atomicAdd((unsigned long long *)&(i < MFX_NB_LDS ? c_undr[i] : c_over[i - MFX_NB_LDS]), (unsigned long long)v);
  }
}

// koverCpy of a block with a fixed launch-wide work split: fixed-order tree, one partial per block
__device__ __forceinline__ void mfx_hist_lds_flush(mfx_hist_lds &H, const mfx_kstar_args &ka, double kover) {
  const uint32_t tid = threadIdx.x;
  mfx_hist_lds_flush_bins(H, ka);
  H.dred[tid] = kover;
  __syncthreads();
  for (uint32_t s = MFX_BLOCK / 2; s > 0; s >>= 1) {
    if (tid < s) H.dred[tid] = H.dred[tid] + H.dred[tid + s];
    __syncthreads();
  }
  if (tid == 0) ka.partials[blockIdx.x] = H.dred[0];
}

