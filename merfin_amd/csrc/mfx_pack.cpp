// merfin_amd -- host side of the packed sequence transport (mfx_hist_run_streamed).
//
// The assembly crosses PCIe as 2-bit codes + one validity bit per base (0.375 B/base instead of 1), in exactly the
// form the device keeps a sequence tile in LDS (csrc/mfx_device.h, mfx_tile_lds): 32 bases per 64-bit word, first
// base in the two HIGHEST bits, code = (c >> 1) & 3 (A 0, C 1, T 2, G 3, either case); one 32-bit validity word per
// 32 bases, first base in the highest bit, valid <=> (c & 0xDF) is one of ACGT.  This is the encoding step of the
// reference's kmerIterator (merfin.C:45, merfin-histogram.C:54-55) moved in front of the bus.
//
// Plain C++ (g++), no HIP: the AVX2 body is selected at run time; the scalar body is the definition.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

inline void pack_scalar(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid) {
  for (uint64_t w = 0; w * 32 < n; ++w) {
    const uint64_t m = n - w * 32 < 32 ? n - w * 32 : 32;
    uint64_t cw = 0;
    uint32_t vw = 0;
    for (uint64_t j = 0; j < m; ++j) {
      const uint8_t c = src[w * 32 + j], u = (uint8_t)(c & 0xDF);
      const uint32_t ok = (u == 'A' || u == 'C' || u == 'G' || u == 'T') ? 1u : 0u;
      cw |= (uint64_t)((c >> 1) & 3u) << (62 - 2 * j);
      vw |= ok << (31 - j);
    }
    codes[w] = cw;
    valid[w] = vw;
  }
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void pack_avx2(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid) {
  const uint64_t full = n / 32;
  // lut[low nibble of (c & 0xDF)] == (c & 0xDF) exactly for A C G T; entry 0 must not match the byte 0
  const __m256i lut = _mm256_setr_epi8(-1, 0x41, 0, 0x43, 0x54, 0, 0, 0x47, 0, 0, 0, 0, 0, 0, 0, 0,
                                       -1, 0x41, 0, 0x43, 0x54, 0, 0, 0x47, 0, 0, 0, 0, 0, 0, 0, 0);
  const __m256i m_df = _mm256_set1_epi8((char)0xDF), m_0f = _mm256_set1_epi8(0x0F), m_03 = _mm256_set1_epi8(0x03);
  const __m256i mul = _mm256_set1_epi32(0x01041040);    // bytes (64, 16, 4, 1): first base of a dword in the two highest bits
  const __m256i ones = _mm256_set1_epi16(1);
  const __m256i rev = _mm256_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
                                       15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
  // byte 0 of dwords 0..3 of a lane, first dword in the highest byte
  const __m256i gather = _mm256_setr_epi8(12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                          12, 8, 4, 0, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
  const __m256i word = _mm256_setr_epi32(4, 0, 0, 0, 0, 0, 0, 0);     // low qword = (lane 1 dword 0, lane 0 dword 0)
  for (uint64_t w = 0; w < full; ++w) {
    const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(src + 32 * w));
    const __m256i up = _mm256_and_si256(v, m_df);
    const __m256i ok = _mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, _mm256_and_si256(up, m_0f)), up);
    // validity, first base in the highest bit: reverse the 32 bytes, then one movemask
    const __m256i okr = _mm256_shuffle_epi8(ok, rev);
    valid[w] = (uint32_t)_mm256_movemask_epi8(_mm256_permute2x128_si256(okr, okr, 0x01));
    // codes: 4 bases -> one byte, then 8 bytes -> one word, first byte highest
    const __m256i c = _mm256_and_si256(_mm256_srli_epi16(v, 1), m_03);
    const __m256i b4 = _mm256_madd_epi16(_mm256_maddubs_epi16(c, mul), ones);   // per dword: c0*64 + c1*16 + c2*4 + c3
    const __m256i g = _mm256_permutevar8x32_epi32(_mm256_shuffle_epi8(b4, gather), word);
    codes[w] = (uint64_t)_mm_cvtsi128_si64(_mm256_castsi256_si128(g));
  }
  if (n % 32) pack_scalar(src + 32 * full, n % 32, codes + full, valid + full);
}

__attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi"))) void pack_avx512(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid) {
  const uint64_t full = n / 64;
  alignas(64) uint8_t ridx[64];
  for (int i = 0; i < 64; ++i) ridx[i] = (uint8_t)(63 - i);
  const __m512i rev = _mm512_load_si512(ridx);
  const __m512i lut = _mm512_broadcast_i32x4(_mm_setr_epi8(-1, 0x41, 0, 0x43, 0x54, 0, 0, 0x47, 0, 0, 0, 0, 0, 0, 0, 0));
  const __m512i m_df = _mm512_set1_epi8((char)0xDF), m_0f = _mm512_set1_epi8(0x0F), m_03 = _mm512_set1_epi8(0x03);
  const __m512i mul = _mm512_set1_epi32(0x40100401);    // the vector is byte-reversed: a dword holds (c3, c2, c1, c0)
  const __m512i ones = _mm512_set1_epi16(1);
  for (uint64_t i = 0; i < full; ++i) {
    const __m512i v = _mm512_permutexvar_epi8(rev, _mm512_loadu_si512(src + 64 * i));       // byte j <- base 63 - j
    const __m512i up = _mm512_and_si512(v, m_df);
    const uint64_t ok = _mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, _mm512_and_si512(up, m_0f)), up);   // bit j <- base 63 - j
    valid[2 * i] = (uint32_t)(ok >> 32);
    valid[2 * i + 1] = (uint32_t)ok;
    const __m512i c = _mm512_and_si512(_mm512_srli_epi16(v, 1), m_03);
    const __m512i b4 = _mm512_madd_epi16(_mm512_maddubs_epi16(c, mul), ones);               // dword d: the byte of bases 60-4d .. 63-4d
    const __m128i by = _mm512_cvtepi32_epi8(b4);                                            // bytes d = 0..15: low qword = bases 32..63, high = 0..31
    _mm_storeu_si128(reinterpret_cast<__m128i *>(codes + 2 * i), _mm_shuffle_epi32(by, 0x4E));
  }
  const uint64_t done = 64 * full;
  if (n > done) pack_avx2(src + done, n - done, codes + 2 * full, valid + 2 * full);
}
#endif

}  // namespace

extern "C" void mfx_pack_bases(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid) {
#if defined(__x86_64__)
  static const int level = (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi") && __builtin_cpu_supports("avx512vl")) ? 2
                           : __builtin_cpu_supports("avx2") ? 1 : 0;
  const char *cap = getenv("MFX_PACK_ISA");               // "avx2" / "scalar": cap the body (tests)
  int use = level;
  if (cap && !strcmp(cap, "avx2")) use = level < 1 ? level : 1;
  if (cap && !strcmp(cap, "scalar")) use = 0;
  if (use == 2) { pack_avx512(src, n, codes, valid); return; }
  if (use == 1) { pack_avx2(src, n, codes, valid); return; }
#endif
  pack_scalar(src, n, codes, valid);
}

// the definition, for tests of the vector body
extern "C" void mfx_pack_bases_scalar(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid) {
  pack_scalar(src, n, codes, valid);
}
