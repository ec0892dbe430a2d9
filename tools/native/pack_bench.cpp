// pack_bench -- what bounds the host encoder of the streamed -hist (csrc/mfx_pack.cpp) on THIS box: threads x placement x memory kind.
//   g++ -O2 -std=c++17 -pthread tools/native/pack_bench.cpp -o tools/_build/pack_bench -Lmerfin_amd -lmerfin_amd -Wl,-rpath,$PWD/merfin_amd
//   tools/_build/pack_bench [MB of bases, default 2048]
// Placement: "os" = wherever the scheduler puts the threads; "spread" = thread i pinned to L3 domain (CCD) i mod nL3 of the
// memory's NUMA node, one hardware thread per core; "packed" = the threads fill the L3 domains one after the other.
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <string>
#include <thread>
#include <vector>

extern "C" void mfx_pack_bases(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid);
extern "C" void *mfx_host_alloc(size_t bytes);

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int read_int(const std::string &p) {
  FILE *f = fopen(p.c_str(), "r");
  if (!f) return -1;
  int v = -1;
  if (fscanf(f, "%d", &v) != 1) v = -1;
  fclose(f);
  return v;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 2048;
  const size_t n = mb << 20;
  cpu_set_t allowed;
  sched_getaffinity(0, sizeof(allowed), &allowed);
  // topology of the allowed CPUs: L3 id, NUMA node, primary hardware thread of the core
  struct Cpu { int id, l3, node, core_first; };
  std::vector<Cpu> cpus;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &allowed)) continue;
    char p[160];
    snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/cache/index3/id", c);
    int l3 = read_int(p);
    snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    int first = read_int(p);
    int node = -1;
    for (int nd = 0; nd < 8 && node < 0; ++nd) {
      snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/node%d", c, nd);
      if (access(p, F_OK) == 0) node = nd;
    }
    cpus.push_back({c, l3, node, first});
  }
  std::map<int, std::vector<int>> by_l3[8];
  for (auto &c : cpus) if (c.core_first == c.id && c.node >= 0 && c.node < 8) by_l3[c.node][c.l3].push_back(c.id);
  for (int nd = 0; nd < 8; ++nd) if (!by_l3[nd].empty()) {
    printf("node %d: %zu L3 domains:", nd, by_l3[nd].size());
    for (auto &kv : by_l3[nd]) printf(" [%d: %zu cores]", kv.first, kv.second.size());
    printf("\n");
  }
  uint8_t *src_page = (uint8_t *)aligned_alloc(4096, n);
  uint8_t *src_pin = (uint8_t *)mfx_host_alloc(n);
  uint64_t *codes_pin = (uint64_t *)mfx_host_alloc(n / 4 + 4096);
  uint32_t *valid_pin = (uint32_t *)mfx_host_alloc(n / 8 + 4096);
  uint64_t *codes_page = (uint64_t *)aligned_alloc(4096, n / 4 + 4096);
  uint32_t *valid_page = (uint32_t *)aligned_alloc(4096, n / 8 + 4096);
  if (!src_pin || !codes_pin || !valid_pin) { fprintf(stderr, "pinned allocation failed\n"); return 1; }
  {  // first touch by many threads, random bases
    std::vector<std::thread> th;
    for (int t = 0; t < 16; ++t) th.emplace_back([&, t]() {
      uint64_t s = 88172645463325252ull + t;
      for (size_t i = n * t / 16; i < n * (t + 1) / 16; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; src_page[i] = "ACGT"[s & 3]; }
    });
    for (auto &x : th) x.join();
    memcpy(src_pin, src_page, n);
    memset(codes_page, 0, n / 4); memset(valid_page, 0, n / 8); memset(codes_pin, 0, n / 4); memset(valid_pin, 0, n / 8);
  }
  int node_of_src = -1;
  syscall(SYS_get_mempolicy, &node_of_src, nullptr, 0UL, src_page, 3UL);
  int node_of_pin = -1;
  syscall(SYS_get_mempolicy, &node_of_pin, nullptr, 0UL, src_pin, 3UL);
  printf("pageable source on node %d, pinned source on node %d\n", node_of_src, node_of_pin);
  const int nd = node_of_src >= 0 && !by_l3[node_of_src].empty() ? node_of_src : 0;
  std::vector<std::vector<int>> l3s;
  for (auto &kv : by_l3[nd]) l3s.push_back(kv.second);

  auto run = [&](const char *what, int T, int place, const uint8_t *src, uint64_t *codes, uint32_t *valid) {
    double best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      std::atomic<int> ready{0};
      std::atomic<bool> go{false};
      std::vector<std::thread> th;
      double t0 = 0;
      for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
        if (place != 0 && !l3s.empty()) {
          cpu_set_t s;
          CPU_ZERO(&s);
          const std::vector<int> &dom = place == 1 ? l3s[t % l3s.size()] : l3s[std::min<size_t>(l3s.size() - 1, t / std::max<size_t>(1, l3s[0].size()))];
          for (int c : dom) CPU_SET(c, &s);
          sched_setaffinity(0, sizeof(s), &s);
        }
        ready.fetch_add(1);
        while (!go.load()) {}
        const size_t per = n / T / 64 * 64;
        mfx_pack_bases(src + t * per, per, codes + t * per / 32, valid + t * per / 32);
      });
      while (ready.load() < T) {}
      t0 = now();
      go.store(true);
      for (auto &x : th) x.join();
      best = std::min(best, now() - t0);
    }
    printf("%-34s threads %2d  %-6s  %6.1f GB/s of bases (%.2f per thread)\n", what, T, place == 0 ? "os" : place == 1 ? "spread" : "packed", n / best / 1e9, n / best / 1e9 / T);
    fflush(stdout);
  };
  for (int T : {1, 2, 4, 8, 12, 15, 16, 24, 32})
    for (int place : {0, 1, 2}) {
      if (T > 16 && place == 2) continue;
      run("pageable src -> pinned dst", T, place, src_page, codes_pin, valid_pin);
    }
  for (int place : {0, 1}) {
    run("pinned src -> pinned dst", 16, place, src_pin, codes_pin, valid_pin);
    run("pageable src -> pageable dst", 16, place, src_page, codes_page, valid_page);
  }
  return 0;
}
