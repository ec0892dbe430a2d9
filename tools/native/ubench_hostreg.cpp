// How fast can a file in the page cache (tmpfs) reach HBM?  (a) pread into pinned lanes + hipMemcpyAsync (what the stager does: the CPU copies
// every byte once); (b) mmap the file and hipHostRegister slices of the mapping from T threads, then copy straight out of the page cache's
// pages (the CPU only pins them).  usage: ubench_hostreg <file> [slice MB] [threads]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char **argv) {
  if (argc < 2) return 1;
  const size_t slice = (size_t)(argc > 2 ? atoi(argv[2]) : 256) << 20;
  const unsigned T = argc > 3 ? (unsigned)atoi(argv[3]) : 16;
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st)) { perror("open"); return 1; }
  const size_t n = (size_t)st.st_size & ~((size_t)(2u << 20) - 1);
  CK(hipSetDevice(0));
  uint8_t *d = nullptr;
  CK(hipMalloc((void **)&d, n));
  hipStream_t cs;
  CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  // (a) pread into 3 pinned lanes of 32 MB by T threads, copy
  {
    const size_t LANE = 32u << 20;
    uint8_t *lane[3];
    hipEvent_t ev[3];
    for (int i = 0; i < 3; ++i) { CK(hipHostMalloc((void **)&lane[i], LANE, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    const double t0 = now();
    bool busy[3] = {false, false, false};
    size_t c = 0;
    for (size_t o = 0; o < n; o += LANE, ++c) {
      const int li = (int)(c % 3);
      const size_t len = std::min(LANE, n - o);
      if (busy[li]) CK(hipEventSynchronize(ev[li]));
      std::vector<std::thread> th;
      const size_t per = (len + T - 1) / T;
      for (unsigned t = 0; t < T; ++t) th.emplace_back([&, t]() {
        size_t b = std::min(len, t * per), e = std::min(len, b + per);
        while (b < e) { ssize_t r = pread(fd, lane[li] + b, e - b, (off_t)(o + b)); if (r <= 0) break; b += (size_t)r; }
      });
      for (auto &x : th) x.join();
      CK(hipMemcpyAsync(d + o, lane[li], len, hipMemcpyHostToDevice, cs));
      CK(hipEventRecord(ev[li], cs));
      busy[li] = true;
    }
    CK(hipStreamSynchronize(cs));
    const double dt = now() - t0;
    printf("(a) pread -> pinned lanes -> copy: %.2f GB in %.3f s = %.1f GB/s (%u threads)\n", n / 1e9, dt, n / 1e9 / dt, T);
  }
  // (b) mmap + register slices + copy
  void *m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
  if (m == MAP_FAILED) { perror("mmap"); return 1; }
  for (int flags_i = 0; flags_i < 2; ++flags_i) {
    const unsigned flags = flags_i == 0 ? hipHostRegisterDefault : hipHostRegisterReadOnly;
    const size_t ns = (n + slice - 1) / slice;
    std::vector<hipEvent_t> done(ns);
    std::vector<char> ok(ns, 0);
    std::atomic<size_t> next{0}, reg{0};
    std::atomic<int> bad{0};
    const double t0 = now();
    double t_reg_end = 0;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; ++t) th.emplace_back([&]() {
      (void)hipSetDevice(0);
      for (size_t i = next.fetch_add(1); i < ns; i = next.fetch_add(1)) {
        uint8_t *p = (uint8_t *)m + i * slice;
        const size_t len = std::min(slice, n - i * slice);
        hipError_t e = hipHostRegister(p, len, flags);
        if (e != hipSuccess) { if (!bad.exchange(1)) fprintf(stderr, "hipHostRegister(flags %u): %s\n", flags, hipGetErrorString(e)); (void)hipGetLastError(); return; }
        ok[i] = 1;
        reg.fetch_add(1);
      }
    });
    // the copies, in order, as the slices come up
    size_t copied = 0;
    for (size_t i = 0; i < ns && !bad.load(); ++i) {
      while (!ok[i] && !bad.load()) std::this_thread::yield();
      if (!ok[i]) break;
      const size_t len = std::min(slice, n - i * slice);
      CK(hipMemcpyAsync(d + i * slice, (uint8_t *)m + i * slice, len, hipMemcpyHostToDevice, cs));
      ++copied;
    }
    for (auto &x : th) x.join();
    t_reg_end = now();
    CK(hipStreamSynchronize(cs));
    const double dt = now() - t0;
    printf("(b) mmap + hipHostRegister(flags %u, %zu MB slices, %u threads) + copy: %zu of %zu slices, registered after %.3f s, all in HBM after %.3f s = %.1f GB/s\n", flags,
           slice >> 20, T, copied, ns, t_reg_end - t0, dt, copied * (double)slice / 1e9 / dt);
    const double t1 = now();
    for (size_t i = 0; i < ns; ++i) if (ok[i]) (void)hipHostUnregister((uint8_t *)m + i * slice);
    printf("    unregister: %.3f s\n", now() - t1);
    if (!bad.load()) break;
  }
  // check one word
  uint64_t a = 0, b = 0;
  CK(hipMemcpy(&a, d + n / 2, 8, hipMemcpyDeviceToHost));
  memcpy(&b, (uint8_t *)m + n / 2, 8);
  printf("check %s\n", a == b ? "ok" : "MISMATCH");
  return 0;
}
