// ubench_locality: per-lane 16-byte loads where G consecutive lanes of a wave fall into the SAME random 128-byte line
// (different / random 16-byte segments) -- the access shape of a per-lane probe of minimizer-keyed 16-byte mini-buckets,
// where neighbouring k-mers share their bucket's line.  Question: is the limit the number of LANES (as for fully divergent
// loads: ~50 G lane-loads/s) or the number of distinct LINES per instruction?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_locality.hip -o tools/_build/ubench_locality && tools/_build/ubench_locality [GiB]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__global__ void fill_kernel(uint4 *t, uint64_t n16) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) { uint64_t h = mix64(i); t[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)i, 1u); }
}
// G256 = group size x 256 (fractional averages: 614 = runs of 2.4 on average, realised as a mix of floor / ceil runs)
template <int ILP, int SECOND>
__global__ __launch_bounds__(256) void probe_kernel(const uint4 *__restrict__ t, uint64_t nlines128, int iters, uint32_t g256, uint64_t seed, uint64_t *out) {
  const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  // lane -> group id: floor(lane * 256 / g256): consecutive lanes share a group
  const uint64_t grp = ((tid & 63u) * 256u) / g256 + (tid >> 6) * 64u;
  uint64_t acc = 0;
  uint64_t ctr = seed + grp * 0x9e3779b97f4a7c15ULL, own = seed ^ (tid * 0xD6E8FEB86659FD93ULL);
  for (int it = 0; it < iters; ++it) {
    uint64_t ln[ILP];
    uint32_t seg[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      ctr += 0xD1B54A32D192ED03ULL;
      own += 0x9FB21C651E98DF25ULL;
      ln[j] = (uint64_t)(((unsigned __int128)mix64(ctr) * nlines128) >> 64);
      seg[j] = (uint32_t)(mix64(own) >> 61);                  // a random 16-byte segment of the group's line
    }
    uint4 v[ILP], v2[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) v[j] = t[ln[j] * 8 + seg[j]];
    if (SECOND) {
#pragma unroll
      for (int j = 0; j < ILP; ++j) v2[j] = t[ln[j] * 8 + ((seg[j] + 1 + (v[j].w & 0)) & 7)];    // a dependent second probe of the same line
    }
#pragma unroll
    for (int j = 0; j < ILP; ++j) { acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w; if (SECOND) acc += v2[j].x ^ v2[j].z; }
  }
  if (acc == 0x1234567ULL) out[0] = acc;
}
template <int ILP, int SECOND>
static void run(const uint4 *t, uint64_t bytes, uint64_t *out, uint32_t g256, int blocks_per_cu) {
  const int grid = 256 * blocks_per_cu, block = 256;
  int iters = (int)((1ull << 31) / ((uint64_t)grid * block * ILP));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe_kernel<ILP, SECOND><<<grid, block>>>(t, bytes / 128, 2, g256, 1, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe_kernel<ILP, SECOND><<<grid, block>>>(t, bytes / 128, iters, g256, 77, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)grid * block * ILP * iters;
  printf("  lanes per line %.2f  ilp=%d blocks/cu=%d%s : %7.2f G lane-probes/s = %6.2f G distinct lines/s  %.1f ms\n", g256 / 256.0, ILP, blocks_per_cu,
         SECOND ? " +2nd probe of the line" : "", n / ms * 1e-6, n / (g256 / 256.0) / ms * 1e-6, ms);
  fflush(stdout);
}
int main(int argc, char **argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 96.0;
  const uint64_t bytes = (uint64_t)(gib * (1ull << 30)) & ~127ull;
  uint4 *t; uint64_t *out;
  CK(hipMalloc(&t, bytes)); CK(hipMalloc(&out, 8));
  fill_kernel<<<4096, 256>>>(t, bytes / 16);
  CK(hipDeviceSynchronize());
  printf("table %.1f GiB\n", gib);
  for (uint32_t g256 : {256u, 512u, 614u, 768u, 1024u, 2048u}) {
    run<1, 0>(t, bytes, out, g256, 8);
    run<4, 0>(t, bytes, out, g256, 8);
    run<4, 0>(t, bytes, out, g256, 5);
    run<4, 1>(t, bytes, out, g256, 8);
  }
  return 0;
}
