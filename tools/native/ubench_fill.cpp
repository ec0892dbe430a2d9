// Host side of "a file in the page cache (tmpfs) -> pinned lanes": how fast do T threads fill a pinned buffer, by pread() and by memcpy out of an
// mmap of the file, and is the pinned memory itself slow to write?  No device copies here (the link does 55 GB/s; the question is the host).
// usage: ubench_fill <file>
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
  int fd = open(argv[1], O_RDONLY);
  struct stat st;
  if (fd < 0 || fstat(fd, &st)) { perror("open"); return 1; }
  const size_t n = (size_t)st.st_size & ~((size_t)(2u << 20) - 1);
  CK(hipSetDevice(0));
  const size_t RING = 1u << 30;                  // destination: 1 GB, written round and round
  uint8_t *pinned = nullptr, *plain = (uint8_t *)aligned_alloc(4096, RING);
  CK(hipHostMalloc((void **)&pinned, RING, hipHostMallocDefault));
  memset(plain, 1, RING);
  void *m = mmap(nullptr, n, PROT_READ, MAP_SHARED, fd, 0);
  if (m == MAP_FAILED) { perror("mmap"); return 1; }
  const size_t PIECE = 1u << 20;
  for (int pass = 0; pass < 2; ++pass)           // (pass 1: the mapping's page tables exist)
    for (int method = 0; method < 2; ++method)
      for (int dst = 0; dst < 2; ++dst)
        for (unsigned T : {1u, 4u, 8u, 16u, 24u, 32u}) {
          if (pass == 1 && method == 0) continue;
          uint8_t *d = dst ? plain : pinned;
          std::atomic<size_t> next{0};
          const size_t np = n / PIECE;
          const double t0 = now();
          std::vector<std::thread> th;
          for (unsigned t = 0; t < T; ++t) th.emplace_back([&]() {
            for (size_t i = next.fetch_add(1); i < np; i = next.fetch_add(1)) {
              uint8_t *o = d + (i * PIECE) % RING;
              if (method == 0) { size_t b = 0; while (b < PIECE) { ssize_t r = pread(fd, o + b, PIECE - b, (off_t)(i * PIECE + b)); if (r <= 0) break; b += (size_t)r; } }
              else memcpy(o, (const uint8_t *)m + i * PIECE, PIECE);
            }
          });
          for (auto &x : th) x.join();
          const double dt = now() - t0;
          printf("%s -> %s, %2u threads%s: %.2f GB in %.3f s = %.1f GB/s\n", method ? "mmap+memcpy" : "pread      ", dst ? "malloc'ed" : "pinned   ", T,
                 pass ? " (second pass over the mapping)" : "", n / 1e9, dt, n / 1e9 / dt);
        }
  return 0;
}
