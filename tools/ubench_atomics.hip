// ubench_atomics: what a table-building kernel can ask of the memory system -- per-lane 64-bit atomics on random slots of
// a table far larger than the caches, G consecutive lanes of a wave in the same random 128-byte line (the access shape of
// claims / count updates of minimizer-keyed lines).  Modes: returning compare-and-swap (a claim), returning CAS after a
// 16-byte load of the same line (find, then claim), non-returning add (fire and forget).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_atomics.hip -o tools/_build/ubench_atomics && tools/_build/ubench_atomics [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
// MODE 0: returning CAS; 1: 16-byte load of the line, then returning CAS; 2: non-returning add; 3: 16-byte load only
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void atom_kernel(unsigned long long *t, uint64_t nlines128, int iters, uint32_t g256, uint64_t seed, uint64_t *out) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  const uint64_t grp = ((tid & 63u) * 256u) / g256 + (tid >> 6) * 64u;
  uint64_t acc = 0;
  uint64_t ctr = seed + grp * 0x9e3779b97f4a7c15ULL, own = seed ^ (tid * 0xD6E8FEB86659FD93ULL);
  for (int it = 0; it < iters; ++it) {
    unsigned long long *p[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      ctr += 0xD1B54A32D192ED03ULL;
      own += 0x9FB21C651E98DF25ULL;
      const uint64_t ln = (uint64_t)(((unsigned __int128)mix64(ctr) * nlines128) >> 64);
      p[j] = t + ln * 16 + (mix64(own) >> 60);                // a random 8-byte slot of the group's line
    }
    uint4 v[ILP];
    if (MODE == 1 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) v[j] = *reinterpret_cast<const uint4 *>((uintptr_t)p[j] & ~(uintptr_t)15);
#pragma unroll
      for (int j = 0; j < ILP; ++j) acc += v[j].x ^ v[j].w;
    }
    if (MODE == 0 || MODE == 1) {
      unsigned long long old[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) old[j] = atomicCAS(p[j], (unsigned long long)(acc & 1u) + 5ull, (unsigned long long)it);
#pragma unroll
      for (int j = 0; j < ILP; ++j) acc += old[j];
    }
    if (MODE == 2) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) atomicAdd(p[j], 1ull);
    }
  }
  if (acc == 0x1234567ULL) out[0] = acc;
}
template <int MODE, int ILP>
static void run(unsigned long long *t, uint64_t bytes, uint64_t *out, uint32_t g256, const char *what) {
  const int grid = 256 * 8, block = 256;
  int iters = (int)((1ull << 30) / ((uint64_t)grid * block * ILP));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  atom_kernel<MODE, ILP><<<grid, block>>>(t, bytes / 128, 2, g256, 1, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  atom_kernel<MODE, ILP><<<grid, block>>>(t, bytes / 128, iters, g256, 77, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)grid * block * ILP * iters;
  printf("  %-34s lanes per line %.2f ilp=%d : %7.2f G ops/s = %6.2f G distinct lines/s  %.1f ms\n", what, g256 / 256.0, ILP, n / ms * 1e-6,
         n / (g256 / 256.0) / ms * 1e-6, ms);
  fflush(stdout);
}
int main(int argc, char **argv) {
  const uint64_t gib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 32;
  const uint64_t bytes = gib << 30;
  unsigned long long *t; uint64_t *out;
  CK(hipMalloc(&t, bytes)); CK(hipMalloc(&out, 8));
  CK(hipMemset(t, 0, bytes));
  printf("table %llu GiB\n", (unsigned long long)gib);
  if (argc > 2) {                                             // only the loads, at the given lanes per line (x 256)
    for (int i = 2; i < argc; ++i) run<3, 4>(t, bytes, out, (uint32_t)atoi(argv[i]), "16-byte load");
    return 0;
  }
  for (uint32_t g : {256u, 614u}) {
    run<3, 1>(t, bytes, out, g, "16-byte load");
    run<3, 4>(t, bytes, out, g, "16-byte load");
    run<0, 1>(t, bytes, out, g, "returning CAS");
    run<0, 4>(t, bytes, out, g, "returning CAS");
    run<1, 1>(t, bytes, out, g, "16-byte load, then returning CAS");
    run<1, 4>(t, bytes, out, g, "16-byte load, then returning CAS");
    run<2, 1>(t, bytes, out, g, "non-returning add");
    run<2, 4>(t, bytes, out, g, "non-returning add");
  }
  return 0;
}
