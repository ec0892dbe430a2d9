// mfx_traverse.h -- the allele combinations of ONE cluster of variants (merfin's `traverse`, merfin-variants.C:22-126, and varMer::addSeqPath,
// varMer.C:39: a path whose sequence is already there is not added again), as code shared by host and device.
//
// It restates the recursion of mfx_variants.cpp's enumerate() -- itself a restatement of the reference's, with its observable quirks: the
// variants' offsets are shared by all levels and shifted / restored around every replacement, the lengths are a copy per level, a variant
// that starts inside the REF span just replaced is forced to REF (and if that was the last one the path is added at once) -- as a loop over
// an explicit stack, one frame per level, so that the GPU can run it (mfx_var_traverse_kernel: one cluster per WAVE -- every lane runs the control
// flow, the candidate strings lie in LDS and every copy / compare is strided over the lanes: the policy P below).  The host
// calls the same function in tests/ (tools/variants_host_bench.cpp: every cluster of a synthetic call set against enumerate()).
//
// Limits (a cluster beyond them is enumerated on the host as before): MFX_TRV_MAX_NV variants, MFX_TRV_MAX_LEN bytes per candidate
// string; the caller reserves room for the product of the allele counts in paths (MFX_TRV_MAX_PATHS at most).
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define MFX_THD __host__ __device__ inline
#else
#define MFX_THD inline
#endif

constexpr uint32_t MFX_TRV_MAX_NV = 8, MFX_TRV_MAX_PATHS = 64, MFX_TRV_MAX_LEN = 640;

struct mfx_trv_variant { uint32_t off, reflen, na, al0; };       // offset in the cluster's window, REF length, alleles (incl. REF), first allele entry
struct mfx_trv_allele { uint64_t off; uint32_t len, pad; };       // in the batch's allele text
struct mfx_trv_cluster {
  uint64_t win_off;                 // the window's bases in the batch's window text
  uint32_t win_len, nv;
  uint32_t var0, path_cap;          // first variant entry; room for that many paths
  uint64_t text0;                   // where this cluster's paths go in the batch's path text (each followed by '\n'); room: text_cap
  uint64_t path0, row0;             // first path slot / first row entry (rows of nv entries per path) in the batch's path table
  uint32_t text_cap, pad;
};

// status of a cluster
constexpr uint32_t MFX_TRV_OK = 0, MFX_TRV_RANGE = 1 /* a replacement past the end of its string: the host's std::string throws there */,
                   MFX_TRV_ROOM = 2 /* a limit above, or the reserved room, exceeded */;

struct mfx_trv_out {               // the batch's path table (device-enumerated part) and path text
  char *text;                      // absolute: text[c.text0 ...]
  uint64_t *p_off, *p_voff, *p_cfirst;
  uint32_t *p_len, *p_nv;
  int32_t *gt;
  uint32_t *vidx, *vlen;
  uint64_t table_base;             // path slot q of this part is entry table_base + q of the whole table (p_cfirst, p_voff are absolute)
  uint64_t row_base;
};

// How the byte work of a cluster is spread: one thread does it all (the host, tests), or the 64 lanes of a wave that all run the same control
// flow on the same cluster (the device's kernel: candidate strings in LDS, every copy / compare strided over the lanes).
struct mfx_trv_scalar {
  static MFX_THD uint32_t lane() { return 0u; }
  static MFX_THD uint32_t lanes() { return 1u; }
  static MFX_THD void sync() {}
  static MFX_THD bool all(bool p) { return p; }
};

// the paths of cluster c; returns the status, *np_out = paths added.  rep: room for one candidate string per level (MFX_TRV_MAX_NV rows)
template <class P>
MFX_THD uint32_t mfx_traverse_cluster_t(const mfx_trv_cluster &c, const mfx_trv_variant *vars, const mfx_trv_allele *alleles, const char *win_text,
                                        const char *allele_text, const mfx_trv_out &o, uint32_t *np_out, char (*rep)[MFX_TRV_MAX_LEN]) {
  struct Frame { uint32_t idx0, idx, reflen, skipped, nj, replen; int32_t j, delta; uint32_t lens[MFX_TRV_MAX_NV]; };
  Frame f[MFX_TRV_MAX_NV];
  uint32_t offs[MFX_TRV_MAX_NV];
  int32_t path[MFX_TRV_MAX_NV];
  const uint32_t nv = c.nv, last = nv - 1, ln = P::lane(), L = P::lanes();
  uint32_t np = 0, npath = 0;
  uint64_t text_at = 0;                                            // bytes of this cluster's text written so far
  *np_out = 0;
  if (nv == 0 || nv > MFX_TRV_MAX_NV || c.win_len > MFX_TRV_MAX_LEN) return MFX_TRV_ROOM;
  const mfx_trv_variant *V = vars + c.var0;
  for (uint32_t i = 0; i < nv; ++i) { offs[i] = V[i].off; f[0].lens[i] = V[i].reflen; }
  const char *win = win_text + c.win_off;

  // out.add of the host: a sequence already present is not added again (varMer.C:39)
  auto add = [&](const char *s, uint32_t n, const uint32_t *lens) -> uint32_t {
    for (uint32_t p = 0; p < np; ++p) {
      if (o.p_len[c.path0 + p] != n) continue;
      const char *q = o.text + o.p_off[c.path0 + p];
      bool same = true;
      for (uint32_t i = ln; i < n; i += L) same = same && q[i] == s[i];
      if (P::all(same)) return MFX_TRV_OK;
    }
    if (np >= c.path_cap || text_at + n + 1 > c.text_cap) return MFX_TRV_ROOM;
    char *dst = o.text + c.text0 + text_at;
    for (uint32_t i = ln; i < n; i += L) dst[i] = s[i];
    if (ln == 0) {
      dst[n] = '\n';
      const uint64_t q = c.path0 + np;
      o.p_off[q] = c.text0 + text_at;
      o.p_len[q] = n;
      o.p_nv[q] = nv;
      o.p_voff[q] = o.row_base + c.row0 + (uint64_t)np * nv;
      o.p_cfirst[q] = o.table_base + c.path0;
      for (uint32_t i = 0; i < nv; ++i) {
        const uint64_t r = c.row0 + (uint64_t)np * nv + i;
        o.gt[r] = path[i];
        o.vidx[r] = offs[i];
        o.vlen[r] = lens[i];
      }
    }
    P::sync();                                                     // (the next add compares against what was written here)
    text_at += (uint64_t)n + 1;
    ++np;
    return MFX_TRV_OK;
  };

  uint32_t d = 0;
  f[0].idx0 = 0; f[0].j = -1; f[0].reflen = f[0].lens[0]; f[0].nj = V[0].na;
  bool returning = false;                                          // the frame above just finished: this one continues behind its call
  for (;;) {
    Frame &F = f[d];
    if (!returning) {
      // ---- the next allele of this level
      if (++F.j >= (int32_t)F.nj) {
        if (d == 0) break;
        --d;
        returning = true;
        continue;
      }
      path[npath++] = F.j;
      const char *cand = d == 0 ? win : rep[d - 1];
      const uint32_t candlen = d == 0 ? c.win_len : f[d - 1].replen;
      F.skipped = 0; F.delta = 0; F.idx = F.idx0;
      P::sync();                                                   // (whoever still reads rep[d] -- an add of the level above's last turn -- is through)
      if (F.j == 0) {
        for (uint32_t i = ln; i < candlen; i += L) rep[d][i] = cand[i];
        F.replen = candlen;
        P::sync();
      } else {
        const mfx_trv_allele &A = alleles[V[F.idx0].al0 + (uint32_t)F.j];
        const char *hap = allele_text + A.off;
        F.lens[F.idx0] = F.reflen;
        // rep = cand; rep.replace(offs[idx], lens[idx], hap): std::string semantics (pos > size throws; the count is clipped to the end) --
        // written as the three pieces of the result
        const uint32_t pos = offs[F.idx0];
        if (pos > candlen) { *np_out = np; return MFX_TRV_RANGE; }
        const uint32_t cut = F.lens[F.idx0] < candlen - pos ? F.lens[F.idx0] : candlen - pos;
        const uint32_t newlen = candlen - cut + A.len;
        if (newlen > MFX_TRV_MAX_LEN) { *np_out = np; return MFX_TRV_ROOM; }
        for (uint32_t i = ln; i < pos; i += L) rep[d][i] = cand[i];
        for (uint32_t i = ln; i < A.len; i += L) rep[d][pos + i] = hap[i];
        const uint32_t tail = candlen - pos - cut;
        for (uint32_t i = ln; i < tail; i += L) rep[d][pos + A.len + i] = cand[pos + cut + i];
        F.replen = newlen;
        P::sync();
        F.delta = (int32_t)A.len - (int32_t)F.lens[F.idx0];
        const uint32_t affected = offs[F.idx0] + F.lens[F.idx0];
        F.lens[F.idx0] = A.len;
        for (uint32_t i = F.idx + 1; i < nv && offs[i] < affected; ++i) { ++F.idx; path[npath++] = 0; ++F.skipped; }   // later variants inside this REF span: forced to REF
        if (F.skipped > 0 && F.idx == last) {
          const uint32_t st = add(rep[d], F.replen, F.lens);
          if (st) { *np_out = np; return st; }
          npath -= F.skipped + 1;
          continue;
        }
        for (uint32_t i = F.idx + 1; i < nv; ++i) offs[i] += (uint32_t)F.delta;
      }
      if (F.idx + 1 < nv) {                                        // the call of the next level
        Frame &G = f[d + 1];
        G.idx0 = F.idx + 1; G.j = -1; G.nj = V[G.idx0].na;
        for (uint32_t i = 0; i < nv; ++i) G.lens[i] = F.lens[i];   // (the lengths are a copy per call)
        G.reflen = G.lens[G.idx0];
        ++d;
        continue;
      }
    }
    returning = false;
    // ---- behind the call (or at the last variant)
    if (F.idx == last) {
      const uint32_t st = add(rep[d], F.replen, F.lens);
      if (st) { *np_out = np; return st; }
    }
    for (uint32_t i = F.idx + 1; i < nv; ++i) offs[i] -= (uint32_t)F.delta;
    npath -= F.skipped + 1;
  }
  *np_out = np;
  return MFX_TRV_OK;
}

// one thread, its own candidate strings (the host; tests)
MFX_THD uint32_t mfx_traverse_cluster(const mfx_trv_cluster &c, const mfx_trv_variant *vars, const mfx_trv_allele *alleles, const char *win_text,
                                      const char *allele_text, const mfx_trv_out &o, uint32_t *np_out) {
  char rep[MFX_TRV_MAX_NV][MFX_TRV_MAX_LEN];
  return mfx_traverse_cluster_t<mfx_trv_scalar>(c, vars, alleles, win_text, allele_text, o, np_out, rep);
}
