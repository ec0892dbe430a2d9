// Is hipMemset of device memory waited for by the host?  A long kernel is put on the null stream, then hipMemset is called:
// if it returns while the kernel still runs, the fill is only QUEUED -- and a copy on a non-blocking stream issued next is
// not ordered behind it.   hipcc --offload-arch=gfx950 -O2 tools/native/memset_probe.hip -o /tmp/memset_probe && /tmp/memset_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void spin(unsigned long long cycles, unsigned long long *out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (out) *out = wall_clock64();
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t N = 64u << 20;
  unsigned char *d = nullptr, *h = nullptr;
  unsigned long long *o = nullptr;
  hipMalloc((void **)&d, N); hipMalloc((void **)&o, 8); hipHostMalloc((void **)&h, N, hipHostMallocDefault);
  memset(h, 0x5a, N);
  hipStream_t cs; hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    const double t0 = now();
    spin<<<1, 1, 0, nullptr>>>(20000000ull, o);            // 100 MHz wall clock: 0.2 s on the null stream
    const double t1 = now();
    hipMemset(d, 0, N);                                     // null stream, "synchronous" form
    const double t2 = now();
    hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, cs);     // non-blocking stream: not ordered behind the null stream
    hipStreamSynchronize(cs);
    const double t3 = now();
    hipDeviceSynchronize();
    const double t4 = now();
    std::vector<unsigned char> back(N);
    hipMemcpy(back.data(), d, N, hipMemcpyDeviceToHost);
    size_t zeros = 0;
    for (size_t i = 0; i < N; ++i) zeros += back[i] == 0;
    printf("rep %d: launch %.3f ms, hipMemset returned after %.3f ms, copy on the other stream done after %.3f ms, device idle after %.3f ms; bytes zero after all: %zu of %zu\n",
           rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, zeros, N);
  }
  return 0;
}
