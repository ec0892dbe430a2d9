// mfx_kstar.h -- K* arithmetic shared by host and device code.
//
// Restates, in IEEE-754 double arithmetic with no contraction (the library is
// built with -ffp-contract=off, no fast-math), the reference formulas
//   merfinGlobal::getK(kmvalu,kmvalu,...)   src/merfin/merfin-globals.C:66-98
//   merfinGlobal::getKmetric                 src/merfin/merfin-globals.H:248-261
//   the bin index of processHistogram        src/merfin/merfin-histogram.C:72,85
// Every operation is written as a separate statement so neither hipcc nor g++
// can fuse or reassociate them; x86-64 SSE2 (the reference's target, no
// -march, src/Makefile:420) and gfx950 fp64 then agree bit for bit.
#pragma once

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define MFX_HD __host__ __device__ __forceinline__
#else
#define MFX_HD inline
#endif

struct mfx_kp_view {
  double          peak;
  uint32_t        n_prob;
  const uint32_t *probK;
  const double   *probP;
};

// merfin-globals.C:80-97.  asmK is asmV itself (:81).
MFX_HD void mfx_getK_core(double peak, uint32_t n_prob, const uint32_t *probK, const double *probP,
                          uint32_t readV, double &readK, double &prob) {
  readK = 0.0;
  prob = 1.0;
  if (readV == 0)
    readK = 0.0;
  else if ((double)readV < peak)          // :86 (uint32 promoted to double)
    readK = 1.0;
  else {
    double q = (double)readV / peak;      // :89
    readK = round(q);                     // half away from zero
  }
  if ((readV > 0) && (readV <= n_prob)) { // :93-97
    readK = (double)probK[readV - 1];
    prob = probP[readV - 1];
  }
}

// merfin-globals.H:248-261
MFX_HD double mfx_kmetric(double readK, double asmK) {
  if (readK == 0)
    return 0;
  if (asmK > readK) {
    double q = asmK / readK;
    double d = q - 1;
    return d * -1;
  }
  if (asmK < readK) {
    double q = readK / asmK;
    return q - 1;
  }
  return 0;
}

// uint32 idx = ((hi / lo - 1) + 0.1) / 0.2   (merfin-histogram.C:72,85).
// The reference converts the double to uint32 directly; for values that do not
// fit (only reachable with asmV == 0 from a foreign -seqmers, or ratios above
// 8.5e8) that is undefined behaviour which x86-64 gcc resolves as "convert to
// int64, keep the low 32 bits" (inf/NaN -> 0).  We define exactly that.
MFX_HD uint32_t mfx_bin_index(double hi, double lo) {
  double q = hi / lo;
  double d = q - 1;
  double e = d + 0.1;
  double x = e / 0.2;
  if (!(x < 9223372036854775808.0) || x < 0)
    return 0;
  return (uint32_t)(uint64_t)(int64_t)x;
}

// (1.0 - readK / asmK) * prob   (merfin-histogram.C:81)
MFX_HD double mfx_overcopy_term(double readK, double asmK, double prob) {
  double q = readK / asmK;
  double d = 1.0 - q;
  return d * prob;
}

// koverCpy as an INTEGER sum (the tile-driven -hist kernel): a term (1 - readK/asmK) * prob lies in [0, 1) -- readK >= 1, asmK <= 2^32
// -- and is counted in units of 2^-52, rounded to nearest: term + 1.0 lies in [1, 2], where a double's last bit is 2^-52, so the
// mantissa of the sum IS the rounded multiple (a term that rounds up to 1.0 gives 2.0, i.e. 2^52 units, by the same subtraction).
// Order-free, therefore: which lane of a wave evaluated a k-mer, and when, does not change the sum.
MFX_HD uint64_t mfx_kfix(double term) {
  const double y = term + 1.0;
  uint64_t b;
  __builtin_memcpy(&b, &y, sizeof(b));
  return b - 0x3FF0000000000000ull;
}
