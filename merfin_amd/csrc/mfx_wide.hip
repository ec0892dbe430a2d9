// mfx_wide.hip -- the evaluation path for 32 <= k <= 64: k-mers of up to 128 bits (the reference's kmer type holds
// 2k <= 128 bits: call sites merfin-globals.C:183, varMer.C:108; meryl-utility's kmdata).
//
// Same semantics and the same K* stage as mfx_kernels.hip (shared through mfx_device.h); what differs is the table
// slot -- 32 bytes {lo, hi, readV, asmV, state}, four per 128-byte line -- and that every lane probes on its own:
// BASELINE's configurations use k = 21 and 31, so this path is written for correctness and simplicity, not tuned
// (no cooperative line probe, no minimizer placement: the home line is a hash of the k-mer).
//
// Slot protocol.  A 128-bit key cannot be claimed with one 64-bit compare-and-swap, and for k = 64 every 128-bit
// pattern is a legal k-mer, so "empty" lives in a separate word: state 0 = empty, 1 = claimed (key being written),
// 2 = ready.  An inserter claims with CAS(state, 0 -> 1), stores the key, waits for those stores, publishes state = 2; a
// lane that meets state 1 simply takes another turn of its loop (never an inner spin: the claiming lane may sit in the same
// wave and must get to its publish).  A k-mer's slots are tried in ITS order (mfx_w_home: from a hashed slot of the home
// line around the line, then the next lines), a claim takes the first empty one, so a lookup stops at the first empty slot
// of that order.
#include "mfx_device.h"

typedef unsigned __int128 mfx_u128;

constexpr uint32_t MFX_W_MAX_LINES = 512;

__device__ __forceinline__ mfx_u128 mfx_w_make(uint64_t lo, uint64_t hi) { return ((mfx_u128)hi << 64) | lo; }

__device__ __forceinline__ uint64_t mfx_w_rev64(uint64_t x) {          // base order reversed inside 64 bits
  x = __brevll(x);
  return ((x & 0x5555555555555555ULL) << 1) | ((x >> 1) & 0x5555555555555555ULL);
}

__device__ __forceinline__ mfx_u128 mfx_w_revcomp(mfx_u128 f, int k) {
  mfx_u128 r = ((mfx_u128)mfx_w_rev64((uint64_t)f) << 64) | mfx_w_rev64((uint64_t)(f >> 64));
  r >>= (128 - 2 * k);
  const mfx_u128 comp = mfx_w_make(0xAAAAAAAAAAAAAAAAULL, 0xAAAAAAAAAAAAAAAAULL);        // complement = code ^ 2
  const mfx_u128 mask = k == 64 ? ~(mfx_u128)0 : (((mfx_u128)1 << (2 * k)) - 1);
  return (r ^ comp) & mask;
}

// home line of a k-mer, and the slot of a line its probes start at: a k-mer's slots are tried from q0 on, around the line,
// then on through the following lines (each from q0 again) -- most lookups end at the first slot they read (32 bytes of one
// lane), where a walk from slot 0 read half the occupied slots of the line first
__device__ __forceinline__ uint32_t mfx_w_home(const mfx_table_view &t, mfx_u128 key, uint32_t &q0) {
  const uint64_t h = mfx_hash64((uint64_t)key ^ mfx_hash64((uint64_t)(key >> 64) + 0x9E3779B97F4A7C15ULL));
  q0 = (uint32_t)((h * 0xD6E8FEB86659FD93ULL) >> 62);
  return mfx_range32(h, t.nlines);
}

__device__ __forceinline__ mfx_wslot *mfx_w_slots(const mfx_table_view &t) { return reinterpret_cast<mfx_wslot *>(t.slots); }

// k-mer of up to 64 bases starting at tile position p; returns validity (all k bases ACGT)
__device__ __forceinline__ bool mfx_w_tile_kmer(const mfx_tile_lds &L, int k, uint32_t p, mfx_u128 &fwd) {
  const uint32_t w = p >> 5, o = p & 31, sh = 2 * o;
  const uint64_t a = L.codes[w], b = L.codes[w + 1], c = L.codes[w + 2];
  const uint64_t hi = sh ? (a << sh) | (b >> (64 - sh)) : a;
  const uint64_t lo = sh ? (b << sh) | (c >> (64 - sh)) : b;
  fwd = mfx_w_make(lo, hi) >> (128 - 2 * k);
  // 96 validity bits starting at word w, bit o of them first
  const mfx_u128 vv = ((mfx_u128)L.valid[w] << 96) | ((mfx_u128)L.valid[w + 1] << 64) | ((mfx_u128)L.valid[w + 2] << 32);
  const mfx_u128 top = (vv << o) >> (128 - k);
  const mfx_u128 want = k == 128 ? ~(mfx_u128)0 : (((mfx_u128)1 << k) - 1);
  return top == want;
}

// value(kmer): stored counts, 0 when absent (merfin-globals.C:84); -min/-max applied to the read count.
// mfx_w_lookup_from: the walk over the k-mer's slot order from slot qi0 of its home line on (qi0 = 1: the caller has looked
// at the first slot itself -- the batched -hist below, whose first-slot loads of several k-mers are in flight together).
__device__ __forceinline__ uint2 mfx_w_lookup_from(const mfx_table_view &t, uint64_t lo, uint64_t hi, uint64_t line, uint32_t q0, uint32_t qi0) {
  const mfx_wslot *S = mfx_w_slots(t);
  for (uint32_t d = 0; d < MFX_W_MAX_LINES; ++d) {
    const mfx_wslot *ln = S + line * MFX_WSLOTS_LINE;
    for (uint32_t qi = d ? 0u : qi0; qi < MFX_WSLOTS_LINE; ++qi) {
      const mfx_wslot s = ln[(q0 + qi) & (MFX_WSLOTS_LINE - 1u)];
      if (s.state == 0) return make_uint2(0u, 0u);           // the first empty slot of its order: the key was never inserted
      if (s.lo == lo && s.hi == hi) {
        uint32_t rv = s.readV;
        if (rv < t.minV || rv > t.maxV) rv = 0;              // merfin.C:199-200
        return make_uint2(rv, s.asmV);
      }
    }
    if (++line >= t.nlines) line = 0;
  }
  return make_uint2(0u, 0u);
}

__device__ __forceinline__ uint2 mfx_w_lookup(const mfx_table_view &t, mfx_u128 key) {
  uint32_t q0;
  const uint64_t line = mfx_w_home(t, key, q0);
  return mfx_w_lookup_from(t, (uint64_t)key, (uint64_t)(key >> 64), line, q0, 0u);
}

// find-or-claim; nullptr when the probe limit is hit
__device__ __forceinline__ mfx_wslot *mfx_w_claim(const mfx_table_view &t, mfx_u128 key, uint64_t *meta, uint32_t &fresh) {
  mfx_wslot *S = mfx_w_slots(t);
  uint32_t q0;
  uint64_t line = mfx_w_home(t, key, q0);
  const uint64_t lo = (uint64_t)key, hi = (uint64_t)(key >> 64);
  uint32_t d = 0, q = 0;                                     // q: slots of the current line tried so far
  mfx_wslot *found = nullptr;
  bool done = false;
  while (!done) {                                            // one slot examination per turn; state 1 = take another turn
    mfx_wslot *sl = S + line * MFX_WSLOTS_LINE + ((q0 + q) & (MFX_WSLOTS_LINE - 1u));
    unsigned long long *sp = reinterpret_cast<unsigned long long *>(&sl->state);
    unsigned long long st = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st == 0ull) {
      st = atomicCAS(sp, 0ull, 1ull);
      if (st == 0ull) {                                      // the slot is ours: write the key, then publish
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&sl->lo), (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&sl->hi), (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // The key words must be performed before the slot is published.  Every access of this protocol is an agent-scope
        // atomic (performed at the device's coherence point, past the XCD's L2), so waiting for the two stores' acknowledgements
        // orders them -- a workgroup-scope release fence is exactly that wait.  (An agent-scope release instead writes back the
        // whole L2 of the XCD on every claim: 0.4 G k-mers/s.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(sp, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ++fresh;
        found = sl;
        done = true;
        st = 3ull;                                           // nothing more to do with this slot
      }
    }
    if (st == 2ull) {
      const unsigned long long slo = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&sl->lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long shi = __hip_atomic_load(reinterpret_cast<unsigned long long *>(&sl->hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (slo == lo && shi == hi) {
        found = sl;
        done = true;
      } else if (++q == MFX_WSLOTS_LINE) {                   // next slot / next line
        q = 0;
        if (++line >= t.nlines) line = 0;
        if (++d >= MFX_W_MAX_LINES) {
          atomicAdd((unsigned long long *)&meta[2], 1ull);
          done = true;
        }
      }
    }
    // st == 1: another lane is writing its key into this slot -- look again on the next turn
  }
  return found;
}

__device__ __forceinline__ void mfx_w_meta_flush(uint64_t *meta, uint32_t fresh, uint32_t noncanon) {
  uint64_t f = fresh, c = noncanon;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { f += __shfl_down(f, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63u) == 0) {
    if (f) atomicAdd((unsigned long long *)&meta[0], (unsigned long long)f);
    if (c) atomicAdd((unsigned long long *)&meta[1], (unsigned long long)c);
  }
}

// kmers: 2 words per k-mer {low 64 bits, high bits}
__global__ void mfx_w_add_kernel(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side, uint64_t *meta) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t fresh = 0, noncanon = 0;
  for (; i < n; i += stride) {
    const uint32_t v = values[i];
    if (v == 0) continue;
    const mfx_u128 key = mfx_w_make(kmers[2 * i], kmers[2 * i + 1]);
    if (key > mfx_w_revcomp(key, t.k)) ++noncanon;
    mfx_wslot *sl = mfx_w_claim(t, key, meta, fresh);
    if (sl) atomicAdd(side ? &sl->asmV : &sl->readV, v);
  }
  mfx_w_meta_flush(meta, fresh, noncanon);
}

__global__ void mfx_w_value_kernel(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const uint2 v = mfx_w_lookup(t, mfx_w_make(kmers[2 * i], kmers[2 * i + 1]));
    readV[i] = v.x;
    asmV[i] = v.y;
  }
}

__global__ void mfx_w_export_kernel(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV, unsigned long long *count) {
  const uint64_t nslots = t.nlines * MFX_WSLOTS_LINE;
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  const mfx_wslot *S = mfx_w_slots(t);
  for (; i < nslots; i += stride) {
    const mfx_wslot s = S[i];
    if (s.state == 0) continue;
    const unsigned long long w = atomicAdd(count, 1ull);
    kmers[2 * w] = s.lo;
    kmers[2 * w + 1] = s.hi;
    readV[w] = s.readV;
    asmV[w] = s.asmV;
  }
}

// getK(kmer,kmer): value(fmer) + value(rmer) in uint32 arithmetic (merfin-globals.C:107-108); with a canonical
// database only the canonical strand can be present, so one probe of min(f, r) gives the same sum (twice the slot for
// a k-mer that is its own reverse complement: even k)
template <bool CANON>
__device__ __forceinline__ uint2 mfx_w_getV(const mfx_table_view &t, mfx_u128 f, int k) {
  const mfx_u128 r = mfx_w_revcomp(f, k);
  if (CANON) {
    uint2 v = mfx_w_lookup(t, f < r ? f : r);
    if (f == r) { v.x += v.x; v.y += v.y; }                  // even k: a palindrome is looked up as fmer and as rmer -- one slot, twice
    return v;
  }
  const uint2 a = mfx_w_lookup(t, f), b = mfx_w_lookup(t, r);
  return make_uint2(a.x + b.x, a.y + b.y);
}

// -hist (merfin-histogram.C:54-91): tile li of the launch is evaluated by block li % gridDim.x
#ifndef MFX_W_BATCH
#define MFX_W_BATCH 2          // positions of a lane whose first-slot loads are in flight together (1, 2, 4 measure alike: profiles/r04_wide_hist.txt)
#endif
#ifndef MFX_W_MINBLOCKS
#define MFX_W_MINBLOCKS 1
#endif
template <bool CANON>
__global__ __launch_bounds__(MFX_BLOCK, MFX_W_MINBLOCKS) void mfx_w_hist_kernel(mfx_hist_args a) {
  __shared__ mfx_tile_lds L;
  __shared__ mfx_hist_lds H;
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  const mfx_kstar_args &ka = a.ks;
  mfx_hist_lds_init(H, ka);
  const bool lut_ok = H.lut_ok != 0u;
  uint64_t n_over0 = 0;
  uint64_t *c_glob = ka.counts + 2ull * ka.nbins;
  uint64_t *c_kasm = c_glob + 3, *c_kmis = c_kasm + ka.ncontigs;
  for (uint64_t li = blockIdx.x; li < a.n_logical; li += gridDim.x) {
    const uint64_t tile = a.part_n == 1 ? a.tile_begin + li
                                         : ((((li >> a.part_shift) * a.part_n + a.part_rank) << a.part_shift) | (li & ((1ull << a.part_shift) - 1ull)));
    const uint32_t c = a.tile_contig[tile];
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();                                         // the previous tile is consumed
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    __syncthreads();
    uint64_t n_valid = 0, n_missing = 0, zz = 0;
    double kover = 0.0;
    if (CANON) {
      // WB positions of a lane per pass: the first slot of each k-mer's order (32 bytes: where ~3 of 4 lookups end) is loaded
      // for all of them before any is looked at -- WB independent line fetches in flight per lane instead of one; the positions
      // are evaluated in the order the one-at-a-time loop took them (the lane's koverCpy terms add up in the same order)
      constexpr uint32_t WB = MFX_W_BATCH;
      const mfx_wslot *S = mfx_w_slots(a.t);
      for (uint32_t b0 = 0; b0 < MFX_TILE / MFX_BLOCK; b0 += WB) {
        if (b0 * MFX_BLOCK >= n) break;
        uint64_t lo[WB], hi[WB], line[WB];
        uint32_t q0[WB];
        bool ok[WB], pal[WB];
        mfx_wslot s0[WB];
#pragma unroll
        for (uint32_t j = 0; j < WB; ++j) {
          const uint32_t p = (b0 + j) * MFX_BLOCK + tid;
          mfx_u128 f;
          ok[j] = mfx_w_tile_kmer(L, k, p, f) && p < n;
          const mfx_u128 r = mfx_w_revcomp(f, k);
          const mfx_u128 key = f < r ? f : r;
          pal[j] = f == r;                                   // even k: a palindrome is looked up as fmer and as rmer -- one slot, twice
          lo[j] = (uint64_t)key; hi[j] = (uint64_t)(key >> 64);
          line[j] = mfx_w_home(a.t, key, q0[j]);
          s0[j].state = 0;
          if (ok[j]) s0[j] = S[line[j] * MFX_WSLOTS_LINE + q0[j]];
        }
#pragma unroll
        for (uint32_t j = 0; j < WB; ++j) {
          if (!ok[j]) continue;
          uint2 v = make_uint2(0u, 0u);
          if (s0[j].state != 0) {
            if (s0[j].lo == lo[j] && s0[j].hi == hi[j]) {
              uint32_t rv = s0[j].readV;
              if (rv < a.t.minV || rv > a.t.maxV) rv = 0;    // merfin.C:199-200
              v = make_uint2(rv, s0[j].asmV);
            } else v = mfx_w_lookup_from(a.t, lo[j], hi[j], line[j], q0[j], 1u);
          }
          if (pal[j]) { v.x += v.x; v.y += v.y; }
          n_valid++;                                         // merfin-histogram.C:58
          if (mfx_hist_eval(H, ka, lut_ok, v.x, v.y, n_over0, kover)) n_missing++;
        }
      }
    } else {
      for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
        if (b * MFX_BLOCK >= n) break;
        const uint32_t p = b * MFX_BLOCK + tid;
        mfx_u128 f;
        if (!(mfx_w_tile_kmer(L, k, p, f) && p < n)) continue;
        const uint2 v = mfx_w_getV<CANON>(a.t, f, k);
        n_valid++;                                           // merfin-histogram.C:58
        if (mfx_hist_eval(H, ka, lut_ok, v.x, v.y, n_over0, kover)) n_missing++;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kover = kover + __shfl_down(kover, off, 64);
    if ((tid & 63u) == 0) a.tile_partials[li * (MFX_BLOCK / 64) + (tid >> 6)] = kover;
    mfx_block_sum3(n_valid, n_missing, zz, H.red);
    if (tid == 0 && (n_valid | n_missing)) {
      atomicAdd((unsigned long long *)&c_kasm[c], n_valid);
      atomicAdd((unsigned long long *)&c_kmis[c], n_missing);
      atomicAdd((unsigned long long *)&c_glob[0], n_valid);
      atomicAdd((unsigned long long *)&c_glob[1], n_missing);
    }
  }
  {
    uint64_t x = 0, y = 0, z = n_over0;
    mfx_block_sum3(x, y, z, H.red);
    if (tid == 0 && z) atomicAdd((unsigned long long *)&ka.counts[ka.nbins], z);
  }
  mfx_hist_lds_flush_bins(H, ka);
}

// -dump raw values (merfin-dump.C:44-67)
template <bool CANON>
__global__ __launch_bounds__(MFX_BLOCK) void mfx_w_dump_kernel(mfx_dump_args a) {
  __shared__ mfx_tile_lds L;
  __shared__ uint64_t s_red[MFX_BLOCK / 64][3];
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  const uint64_t pos0 = (uint64_t)blockIdx.x * MFX_TILE;
  mfx_tile_fill(L, a.src + pos0);
  __syncthreads();
  uint64_t n_valid = 0, n_missing = 0, zz = 0;
  for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
    const uint32_t p = b * MFX_BLOCK + tid;
    const uint64_t gp = pos0 + p;
    if (!(gp < a.npos && gp >= a.skip)) continue;
    mfx_u128 f;
    const bool ok = mfx_w_tile_kmer(L, k, p, f) && gp < a.clen_left;
    uint2 v = make_uint2(0u, 0u);
    if (ok) {
      v = mfx_w_getV<CANON>(a.t, f, k);
      n_valid++;                                             // merfin-dump.C:48
      double readK, prob;
      mfx_getK_core(a.peak, a.n_prob, a.probK, a.probP, v.x, readK, prob);
      if (readK == 0) n_missing++;                           // :56-58
    }
    a.readV[gp - a.skip] = v.x;
    a.asmV[gp - a.skip] = v.y;
  }
  mfx_block_sum3(n_valid, n_missing, zz, s_red);
  if (tid == 0 && (n_valid | n_missing)) {
    atomicAdd((unsigned long long *)&a.stats[0], n_valid);
    atomicAdd((unsigned long long *)&a.stats[1], n_missing);
  }
}

// `meryl count` of the assembly (merfin-globals.C:182-186)
__global__ __launch_bounds__(MFX_BLOCK) void mfx_w_count_kernel(mfx_count_args a) {
  __shared__ mfx_tile_lds L;
  const uint32_t tid = threadIdx.x;
  const int k = a.t.k;
  uint32_t fresh = 0;
  for (uint64_t tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    uint32_t lo = 0, hi = a.ncontigs;
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (a.tile_start[mid] <= tile) lo = mid; else hi = mid;
    }
    const uint32_t c = lo;
    const uint64_t pos0 = (tile - a.tile_start[c]) * MFX_TILE;
    const uint64_t clen = a.contig_len[c];
    const uint32_t n = (clen - pos0 < MFX_TILE) ? (uint32_t)(clen - pos0) : MFX_TILE;
    __syncthreads();
    mfx_tile_fill(L, a.bases + a.contig_off[c] + pos0);
    __syncthreads();
    for (uint32_t b = 0; b < MFX_TILE / MFX_BLOCK; ++b) {
      const uint32_t p = b * MFX_BLOCK + tid;
      mfx_u128 f;
      if (!(mfx_w_tile_kmer(L, k, p, f) && p < n)) continue;
      const mfx_u128 r = mfx_w_revcomp(f, k);
      mfx_wslot *sl = mfx_w_claim(a.t, f < r ? f : r, a.meta, fresh);
      if (sl) atomicAdd(&sl->asmV, 1u);
    }
  }
  mfx_w_meta_flush(a.meta, fresh, 0u);
}

// -completeness (merfin-completeness.C:70-117): piece = top 6 bits of the 2k-bit k-mer
__global__ __launch_bounds__(MFX_BLOCK) void mfx_w_completeness_kernel(mfx_table_view t, double peak, uint32_t n_prob, const uint32_t *probK,
                                                                       const double *probP, double *pieces) {
  __shared__ double s_tot[64], s_und[64];
  constexpr uint32_t NLUT = 1024;                              // readK per common read count, as in mfx_completeness_kernel
  __shared__ double s_rk[NLUT];
  if (threadIdx.x < 64) { s_tot[threadIdx.x] = 0.0; s_und[threadIdx.x] = 0.0; }
  for (uint32_t v = threadIdx.x; v < NLUT; v += blockDim.x) {
    double rk, pr;
    mfx_getK_core(peak, n_prob, probK, probP, v, rk, pr);
    s_rk[v] = rk;
  }
  __syncthreads();
  const uint64_t nslots = t.nlines * MFX_WSLOTS_LINE;
  const int pshift = 2 * t.k - 6;
  const mfx_wslot *S = mfx_w_slots(t);
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < nslots; i += stride) {
    const mfx_wslot s = S[i];
    if (s.state == 0 || s.readV == 0) continue;              // empty slot / asm-only k-mer (:106-109)
    double readK, prob;
    if (s.readV < NLUT) readK = s_rk[s.readV];
    else mfx_getK_core(peak, n_prob, probK, probP, s.readV, readK, prob);
    const double asmK = (double)s.asmV;
    const uint32_t piece = (uint32_t)(mfx_w_make(s.lo, s.hi) >> pshift) & 63u;
    atomicAdd(&s_tot[piece], readK);                         // :113
    if (readK > asmK) atomicAdd(&s_und[piece], readK - asmK);   // :115-116
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    if (s_tot[threadIdx.x] != 0.0) atomicAdd(&pieces[threadIdx.x], s_tot[threadIdx.x]);
    if (s_und[threadIdx.x] != 0.0) atomicAdd(&pieces[64 + threadIdx.x], s_und[threadIdx.x]);
  }
}

// ---- launch wrappers --------------------------------------------------------------------------------------
hipError_t mfx_kw_table_add(mfx_table_view t, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side, uint64_t *meta, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  mfx_w_add_kernel<<<(unsigned)blocks, 256, 0, st>>>(t, kmers, values, n, side, meta);
  return hipGetLastError();
}
hipError_t mfx_kw_table_value(mfx_table_view t, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV, hipStream_t st) {
  if (n == 0) return hipSuccess;
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  mfx_w_value_kernel<<<(unsigned)blocks, 256, 0, st>>>(t, kmers, n, readV, asmV);
  return hipGetLastError();
}
hipError_t mfx_kw_table_export(mfx_table_view t, uint64_t *kmers, uint32_t *readV, uint32_t *asmV, unsigned long long *count, hipStream_t st) {
  mfx_w_export_kernel<<<4096, 256, 0, st>>>(t, kmers, readV, asmV, count);
  return hipGetLastError();
}
hipError_t mfx_kw_hist(const mfx_hist_args &a, int grid, hipStream_t st) {
  if (a.canonical) mfx_w_hist_kernel<true><<<grid, MFX_BLOCK, 0, st>>>(a);
  else             mfx_w_hist_kernel<false><<<grid, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_kw_dump(const mfx_dump_args &a, hipStream_t st) {
  const uint64_t blocks = (a.npos + MFX_TILE - 1) / MFX_TILE;
  if (blocks == 0) return hipSuccess;
  if (a.canonical) mfx_w_dump_kernel<true><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  else             mfx_w_dump_kernel<false><<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_kw_count(const mfx_count_args &a, hipStream_t st) {
  if (a.ntiles == 0) return hipSuccess;
  const uint64_t blocks = a.ntiles < 8192 ? a.ntiles : 8192;
  mfx_w_count_kernel<<<(unsigned)blocks, MFX_BLOCK, 0, st>>>(a);
  return hipGetLastError();
}
hipError_t mfx_kw_completeness(mfx_table_view t, double peak, uint32_t n_prob, const uint32_t *probK, const double *probP, double *partials,
                               int grid, hipStream_t st) {
  mfx_w_completeness_kernel<<<grid, MFX_BLOCK, 0, st>>>(t, peak, n_prob, probK, probP, partials);
  return hipGetLastError();
}
