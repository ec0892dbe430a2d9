// ubench_gather: measures the random-access line rate of MI355X HBM3E for the
// access shapes the k-mer index probe can take.  Design input for
// merfin_amd/csrc (DESIGN.md "index layout"), not part of the product.
//
//   ubench_gather [GiB ...]     default sizes: 1 8 32 96
//
// For each table size: random 8 B / 16 B / 64 B(4x16B same line) / 128 B
// (8x16B same line) loads per lane, ILP 1/4/8 independent loads in flight.
// Reports G-loads/s and GB/s at 64 B per access.

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33; return x;
}

__global__ void fill_kernel(uint4 *t, uint64_t n16) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (; i < n16; i += stride) {
    uint64_t h = mix64(i);
    t[i] = make_uint4((uint32_t)h, (uint32_t)(h >> 32), (uint32_t)i, 1u);
  }
}

// BYTES: bytes loaded per lane per access (8, 16, 64, 128), all inside one
// aligned line of max(BYTES,64).  ILP: independent accesses in flight.
template <int BYTES, int ILP>
__global__ __launch_bounds__(256) void gather_kernel(const uint4 *__restrict__ t, uint64_t nlines64,
                                                     int iters, uint64_t seed, uint64_t *out) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  uint64_t ctr = seed + tid * 0x9e3779b97f4a7c15ULL;
  for (int it = 0; it < iters; ++it) {
    uint64_t idx[ILP];
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      ctr += 0xD1B54A32D192ED03ULL;
      uint64_t h = mix64(ctr);
      // fastrange onto the number of 64 B lines
      idx[j] = (uint64_t)(((unsigned __int128)h * nlines64) >> 64);
      if (BYTES == 128) idx[j] &= ~1ULL;
    }
    if (BYTES == 8) {
      uint2 v[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) v[j] = *reinterpret_cast<const uint2 *>(t + idx[j] * 4);
#pragma unroll
      for (int j = 0; j < ILP; ++j) acc += v[j].x ^ v[j].y;
    } else if (BYTES == 16) {
      uint4 v[ILP];
#pragma unroll
      for (int j = 0; j < ILP; ++j) v[j] = t[idx[j] * 4];
#pragma unroll
      for (int j = 0; j < ILP; ++j) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    } else {
      constexpr int NV = BYTES / 16;
      uint4 v[ILP][NV];
#pragma unroll
      for (int j = 0; j < ILP; ++j)
#pragma unroll
        for (int q = 0; q < NV; ++q) v[j][q] = t[idx[j] * 4 + q];
#pragma unroll
      for (int j = 0; j < ILP; ++j)
#pragma unroll
        for (int q = 0; q < NV; ++q) acc += v[j][q].x ^ v[j][q].y ^ v[j][q].z ^ v[j][q].w;
    }
  }
  if (acc == 0x1234567ULL) out[0] = acc;   // defeat DCE
}

// two DEPENDENT-address-free 16-byte loads per lane at byte offsets 0 and OFF2 of
// a random 128-byte-aligned line: OFF2=48 same 64-B half, 64 other half, 128 next line.
template <int OFF2>
__global__ __launch_bounds__(256) void pair_kernel(const uint4 *__restrict__ t, uint64_t nlines128, int iters,
                                                   uint64_t seed, uint64_t *out) {
  uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t acc = 0;
  uint64_t ctr = seed + tid * 0x9e3779b97f4a7c15ULL;
  for (int it = 0; it < iters; ++it) {
    ctr += 0xD1B54A32D192ED03ULL;
    uint64_t h = mix64(ctr);
    uint64_t ln = (uint64_t)(((unsigned __int128)h * (nlines128 - 2)) >> 64);
    uint4 a = t[ln * 8];
    uint4 b = t[ln * 8 + OFF2 / 16];
    acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
  }
  if (acc == 0x1234567ULL) out[0] = acc;
}

template <int OFF2>
static void run_pair(const uint4 *t, uint64_t bytes, uint64_t *out) {
  int grid = 2048, block = 256, iters = 2048;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  pair_kernel<OFF2><<<grid, block>>>(t, bytes / 128, iters, 91, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double n = (double)grid * block * iters;
  printf("  pair of 16-B loads, second at +%3d B : %7.2f G-pairs/s  %.1f ms\n", OFF2, n / ms * 1e-6, ms);
  fflush(stdout);
}

__global__ __launch_bounds__(256) void stream_kernel(const uint4 *__restrict__ t, uint64_t n16, uint64_t *out) {
  uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t acc = 0;
  for (; i < n16; i += stride) { uint4 v = t[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x1234567ULL) out[0] = acc;
}

template <int BYTES, int ILP>
static void run(const uint4 *t, uint64_t bytes, uint64_t *out, int blocks_per_cu) {
  int grid = 256 * blocks_per_cu, block = 256;
  uint64_t nlines = bytes / 64;
  uint64_t total_target = 1ull << 31;            // ~2.1 G accesses
  if (BYTES >= 64) total_target >>= 1;
  int iters = (int)(total_target / ((uint64_t)grid * block * ILP));
  if (iters < 1) iters = 1;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gather_kernel<BYTES, ILP><<<grid, block>>>(t, nlines, 2, 1, out);   // warm
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  gather_kernel<BYTES, ILP><<<grid, block>>>(t, nlines, iters, 77, out);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double n = (double)grid * block * ILP * iters;
  printf("  bytes/lane=%3d ilp=%d blocks/cu=%d : %7.2f G-access/s  (%.0f GB/s @64B-line, %.0f GB/s @128B-line)  %.1f ms\n",
         BYTES, ILP, blocks_per_cu, n / ms * 1e-6, n * 64 / ms * 1e-6, n * 128 / ms * 1e-6, ms);
  fflush(stdout);
}

int main(int argc, char **argv) {
  std::vector<double> sizes;
  for (int i = 1; i < argc; ++i) sizes.push_back(atof(argv[i]));
  if (sizes.empty()) sizes = {1, 8, 32, 96};
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  size_t fr, tot; CK(hipMemGetInfo(&fr, &tot));
  printf("device %s  CUs=%d  clock=%d MHz  mem free/total = %.1f/%.1f GiB\n", p.name, p.multiProcessorCount,
         p.clockRate / 1000, fr / 1073741824.0, tot / 1073741824.0);
  uint64_t *out; CK(hipMalloc(&out, 8));
  for (double g : sizes) {
    uint64_t bytes = (uint64_t)(g * 1073741824.0);
    bytes &= ~(uint64_t)127;
    uint4 *t; CK(hipMalloc(&t, bytes));
    fill_kernel<<<8192, 256>>>(t, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("table %.1f GiB\n", g);
    {
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      stream_kernel<<<8192, 256>>>(t, bytes / 16, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  streaming read: %.0f GB/s\n", bytes / ms * 1e-6);
    }
    run<8, 1>(t, bytes, out, 8);
    run<8, 4>(t, bytes, out, 8);
    run<16, 1>(t, bytes, out, 8);
    run<16, 4>(t, bytes, out, 8);
    run<16, 8>(t, bytes, out, 8);
    run<16, 8>(t, bytes, out, 4);
    run<16, 4>(t, bytes, out, 16);
    run<64, 1>(t, bytes, out, 8);
    run<64, 4>(t, bytes, out, 8);
    run<128, 2>(t, bytes, out, 8);
    run_pair<48>(t, bytes, out);
    run_pair<64>(t, bytes, out);
    run_pair<128>(t, bytes, out);
    CK(hipFree(t));
  }
  return 0;
}
