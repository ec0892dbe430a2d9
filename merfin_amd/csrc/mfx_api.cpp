// mfx_api.cpp -- C-ABI entry points of libmerfin_amd (include/merfin_amd.h).
// Host-side orchestration only; all evaluation work is done by the HIP
// kernels in mfx_kernels.hip.  There is deliberately NO CPU fallback: without
// a usable HIP device every entry point fails with MFX_E_NODEVICE / MFX_E_HIP.
#include "mfx_internal.h"
#include "mfx_place.h"
#include "mfx_kernels.h"
#include "mfx_pipe.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sched.h>
#include <pthread.h>
#include <sys/syscall.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <map>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";
static thread_local int  g_err_code = 0;

void mfx_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mfx_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  g_err_code = code;
  return code;
}

extern "C" const char *mfx_last_error(void) { return g_err; }
extern "C" int mfx_last_error_code(void) { return g_err_code; }
extern "C" const char *mfx_version(void) { return "merfin_amd 0.2 (gfx950)"; }

// Diagnostic (no reference counterpart): random 128-byte lines per second this device's HBM delivers to independent
// 16-byte loads over a table of `table_bytes` (allocated and released here) -- the roof of the index probe on THIS box.
extern "C" int mfx_diag_gather_rate(int device, uint64_t table_bytes, double *lines_per_s) {
  if (!lines_per_s || table_bytes < (1ull << 20)) return mfx_fail(MFX_E_INVAL, "mfx_diag_gather_rate: bad argument");
  if (device < 0 || device >= mfx_device_count()) return mfx_fail(MFX_E_NODEVICE, "HIP device %d not available", device);
  int prev = -1;
  (void)hipGetDevice(&prev);
  MFX_HIP(hipSetDevice(device));
  void *t = nullptr;
  uint64_t *scratch = nullptr;
  hipError_t e = hipMalloc(&t, table_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&scratch, 8);
  if (e == hipSuccess) e = mfx_memset_now(t, 0x5a, table_bytes);
  if (e == hipSuccess) e = mfx_k_gather_rate(t, table_bytes / MFX_ALIGN, scratch, lines_per_s, nullptr);
  if (t) (void)hipFree(t);
  if (scratch) (void)hipFree(scratch);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (e != hipSuccess) { (void)hipGetLastError(); return mfx_fail(e == hipErrorOutOfMemory ? MFX_E_NOMEM : MFX_E_HIP, "mfx_diag_gather_rate: %s", hipGetErrorString(e)); }
  return MFX_OK;
}

extern "C" int mfx_device_memory(int device, uint64_t *free_bytes, uint64_t *total_bytes) {
  if (device < 0 || device >= mfx_device_count()) return mfx_fail(MFX_E_NODEVICE, "HIP device %d not available", device);
  int prev = -1;
  (void)hipGetDevice(&prev);
  MFX_HIP(hipSetDevice(device));
  size_t f = 0, t = 0;
  const hipError_t e = hipMemGetInfo(&f, &t);
  if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
  if (e != hipSuccess) return mfx_fail(MFX_E_HIP, "hipMemGetInfo failed: %s", hipGetErrorString(e));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return MFX_OK;
}

extern "C" int mfx_device_warm(int device) {
  if (device < 0 || device >= mfx_device_count()) return mfx_fail(MFX_E_NODEVICE, "HIP device %d not available", device);
  int prev = -1;
  (void)hipGetDevice(&prev);
  MFX_HIP(hipSetDevice(device));
  // a device allocation + memset (the runtime's own kernels), one kernel of this library (its code object), a pinned
  // allocation and a stream: everything the first upload would otherwise bring up on the caller's time
  void *d = nullptr, *h = nullptr;
  hipStream_t st = nullptr;
  const bool timing = getenv("MFX_UPLOAD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tw[8] = {now(), 0, 0, 0, 0, 0, 0, 0};
  hipError_t e = hipMalloc(&d, 1 << 20);
  tw[1] = now();
  if (e == hipSuccess) e = mfx_memset_now(d, 0, 1 << 20);
  tw[2] = now();
  if (e == hipSuccess) e = mfx_k_table_init(reinterpret_cast<mfx_slot *>(d), (1 << 20) / sizeof(mfx_slot), nullptr);
  tw[3] = now();
  if (e == hipSuccess) e = hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  tw[4] = now();
  if (e == hipSuccess) e = hipMemcpyAsync(d, h, 1 << 20, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  tw[5] = now();
  // (no device-wide wait here: another thread's stream -- the database's stager -- may be busy for the whole run)
  if (st) (void)hipStreamDestroy(st);
  if (h) (void)hipHostFree(h);
  if (d) (void)hipFree(d);
  tw[6] = now();
  if (timing)
    fprintf(stderr, "-- device warm: %.3f s = first allocation %.3f + fill %.3f + first kernel of the library %.3f + pinned memory and a stream %.3f + a copy %.3f + release %.3f\n",
            tw[6] - tw[0], tw[1] - tw[0], tw[2] - tw[1], tw[3] - tw[2], tw[4] - tw[3], tw[5] - tw[4], tw[6] - tw[5]);
  if (prev >= 0) (void)hipSetDevice(prev);
  if (e != hipSuccess) { (void)hipGetLastError(); return mfx_fail(MFX_E_HIP, "mfx_device_warm: %s", hipGetErrorString(e)); }
  return MFX_OK;
}

extern "C" int mfx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void mfx_pin_spread(unsigned w);          // pins the calling thread to L3 domain w mod n (all NUMA nodes); defined with the encoder placement below

// Host threads that stay parked between the streamed runs of an evaluator: starting 16 threads costs ~0.45 ms, 1.3 % of a
// 3 Gb run, waking them ~0.05 ms.
namespace {
struct WorkerPool {
  unsigned W;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable wake, idle;
  std::function<void(unsigned)> job;
  uint64_t gen = 0;
  unsigned running = 0;
  bool quit = false;
  // spread: thread i is pinned to L3 domain i mod n of the machine (pack_spread_cpus, below) -- threads that move memory
  // share their CCD's link to it, and threads started by one parent tend to land next to each other
  explicit WorkerPool(unsigned w, bool spread = false) : W(w) {
    for (unsigned i = 0; i < W; ++i)
      th.emplace_back([this, i, spread]() {
        if (spread) mfx_pin_spread(i);
        uint64_t seen = 0;
        for (;;) {
          std::function<void(unsigned)> f;
          {
            std::unique_lock<std::mutex> lk(m);
            wake.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            f = job;
          }
          f(i);
          std::lock_guard<std::mutex> lk(m);
          if (--running == 0) idle.notify_all();
        }
      });
  }
  void start(std::function<void(unsigned)> f) {
    std::lock_guard<std::mutex> lk(m);
    job = std::move(f);
    running = W;
    ++gen;
    wake.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    idle.wait(lk, [&] { return running == 0; });
  }
  ~WorkerPool() {
    { std::lock_guard<std::mutex> lk(m); quit = true; }
    wake.notify_all();
    for (auto &t : th) t.join();
  }
};
}  // namespace

namespace {
struct DevGuard {
  int prev = -1;
  bool ok = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = (hipSetDevice(dev) == hipSuccess);
  }
  ~DevGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

template <class T>
struct DevBuf {   // scoped device scratch
  T *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
};

// Table fill target.  MFX_LOAD_FACTOR fixes it; otherwise it is chosen per index between MFX_LF_MAX (the least
// memory a table may take: what mfx_index_estimate_gb reports and -memory is checked against) and MFX_LF_MIN,
// as low as a share of the free HBM (and of max_gb) allows: emptier lines mean fewer full home lines and fewer
// second probes (3 Gb -hist, w = 3: 0.7 -> 80, 0.6 -> 85, 0.52 -> 89.8, 0.45 -> 91.5, 0.40 -> 91.2 G k-mers/s: nothing is
// gained below 0.45), and 288 GB of HBM are there to be used.
constexpr double MFX_LF_MAX = 0.7, MFX_LF_MIN = 0.45, MFX_LF_HBM_SHARE = 0.75;
// The compact layout of a sequence-only index (16 slots per line in 8 mini-buckets, buckets of w = 4 windows): the emptier the
// table, the more k-mers sit in their first mini-bucket (one 16-byte load) -- 3 Gb -hist: 102.6 / 108.0 / 111.9 G k-mers/s at load
// factors 0.30 / 0.25 / 0.20 (profiles/r03_kernel_experiments.txt).  Round 4's kernel runs at the HBM's random-line rate, so every
// line not fetched is time: 148.1 / 151.8 / 152.7 / 153.6 / 154.3 G at 0.225 / 0.18 / 0.15 / 0.125 / 0.10
// (profiles/r04_ab_load_factor.txt).  0.18 is 135 GB for a human assembly -- under two thirds of what the full table of reads +
// assembly takes at its 0.45 -- and still chosen only when that much HBM is free (lines_auto).
constexpr double MFX_CLF_MAX = 0.5, MFX_CLF_MIN = 0.18;
// A sequence-only index in 16-byte slots (22 <= k <= 31) holds the assembly's k-mers only -- half of what the full tables hold --
// and gives the memory back as speed: 500 Mb -hist at k = 31: 71.3 / 80.6 / 86.7 G k-mers/s at load factors 0.45 / 0.35 / 0.25
// (profiles/r03_hist_rates_by_k.txt).  0.30 is 160 GB for a human assembly.
constexpr double MFX_SLF_MIN = 0.30;

thread_local double t_lf_request = 0;      // mfx_index_create_for_seq_lf: the caller's load factor for the table being created (0: none)
bool load_factor_fixed(double *lf) {
  const char *e = getenv("MFX_LOAD_FACTOR");
  if (!e && t_lf_request > 0) { *lf = std::min(0.9, std::max(0.05, t_lf_request)); return true; }
  if (!e) return false;
  double v = atof(e);
  if (!(v > 0.05 && v <= 0.9)) v = MFX_LF_MAX;
  *lf = v;
  return true;
}

uint64_t lines_at(uint64_t capacity_kmers, double lf, uint32_t slots_line) {
  double slots = (double)capacity_kmers / lf;
  uint64_t nlines = (uint64_t)ceil(slots / slots_line);
  if (nlines < 1024) nlines = 1024;      // probe sequences may span 512 lines
  return nlines;
}

// smallest table this build makes for `capacity_kmers`
uint64_t lines_for(uint64_t capacity_kmers, uint32_t slots_line = MFX_SLOTS_LINE) {
  double lf = slots_line == MFX_CSLOTS_LINE ? MFX_CLF_MAX : MFX_LF_MAX;
  (void)load_factor_fixed(&lf);
  return lines_at(capacity_kmers, lf, slots_line);
}

// the table actually allocated: budget_bytes = what the table may take (0: unknown, use the smallest)
uint64_t lines_auto(uint64_t capacity_kmers, double budget_bytes, uint32_t slots_line, bool seq_only = false) {
  const double lf_max = slots_line == MFX_CSLOTS_LINE ? MFX_CLF_MAX : MFX_LF_MAX,
               lf_min = slots_line == MFX_CSLOTS_LINE ? MFX_CLF_MIN : (seq_only && slots_line == MFX_SLOTS_LINE) ? MFX_SLF_MIN : MFX_LF_MIN;
  double lf;
  if (load_factor_fixed(&lf)) return lines_at(capacity_kmers, lf, slots_line);
  lf = lf_max;
  if (budget_bytes > 0) {
    lf = (double)capacity_kmers * (MFX_ALIGN / slots_line) / budget_bytes;
    lf = std::min(lf_max, std::max(lf_min, lf));
  }
  uint64_t nl = lines_at(capacity_kmers, lf, slots_line);
  if (nl >= (1ull << 32)) nl = std::max<uint64_t>(lines_at(capacity_kmers, lf_max, slots_line), (1ull << 32) - 16);
  return nl;
}

// does a sequence-only index of k-mers of this size take the compact layout?  (MFX_SEQ_COMPACT=0: never -- A/B, tests)
// k <= 21: the slot holds the k-mer; 22 <= k <= 31: its quotient (mfx_kernels.hip: mfx_q_place), which needs the default
// placement (4 windows, mod-minimizer) and the per-lane probe
bool seq_compact(int k) {
  const char *e = getenv("MFX_SEQ_COMPACT");
  const char *hm = getenv("MFX_HOME_MODE");
  if (k > MFX_MAX_K_COMPACT || (e && atoi(e) == 0) || (hm && strcmp(hm, "plain") == 0)) return false;
  if (k > MFX_MAX_K_DIRECT) {
    const char *ws = getenv("MFX_MZ_W"), *mm = getenv("MFX_MZ_MOD"), *q = getenv("MFX_SEQ_QUOT");
    if ((ws && atoi(ws) != MFX_MZ_W_COMPACT) || (mm && atoi(mm) == 0) || (q && atoi(q) == 0) || !mfx_k_quot_supported()) return false;
  }
  return true;
}
// the fewest lines a quotient table may have: the key field keeps 2m - floor(log2(nlines)) + 9 bits of a k-mer (m = k - 3) in 40
uint64_t quot_min_lines(int k) { return k > MFX_MAX_K_DIRECT ? 1ull << (2 * (k - 3) - 31) : 0; }

// Side table of a compact index (exact counts of the saturated fields, 16-byte slots at load factor <= 0.5): room for
// 1/64 of the capacity -- a count saturates at 2047, i.e. beyond ~70 copies at 30x coverage -- and never fewer than
// 1024 lines.  MFX_SIDE_DIV overrides the divisor; a side table that fills up fails the load with MFX_E_FULL.
uint64_t side_lines_for(uint64_t capacity_kmers) {
  const char *e = getenv("MFX_SIDE_DIV");
  uint64_t div = e && atoll(e) > 0 ? (uint64_t)atoll(e) : 64;
  return std::max<uint64_t>(1024, (capacity_kmers / div) * 2 / MFX_SLOTS_LINE + 1);
}
}  // namespace

// host-side copy into the pinned staging buffer, split over a few threads for large pieces
// (a single thread moves ~12 GB/s, well under PCIe Gen5)
static void par_memcpy(uint8_t *dst, const char *src, size_t n) {
  const unsigned nt = std::min<unsigned>(8u, std::max(1u, mfx_host_threads()));
  if (n < (8u << 20) || nt == 1) { memcpy(dst, src, n); return; }
  std::vector<std::thread> th;
  const size_t per = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    size_t b = std::min(n, t * per), e = std::min(n, b + per);
    if (e > b) th.emplace_back([=]() { memcpy(dst + b, src + b, e - b); });
  }
  for (auto &x : th) x.join();
}

// ---------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------
mfx_table_view mfx_index::view() const {
  mfx_table_view v;
  v.slots = d_slots;
  v.nlines = nlines;
  v.minV = minV > 0xffffffffull ? 0xffffffffu : (uint32_t)minV;
  v.maxV = maxV > 0xffffffffull ? 0xffffffffu : (uint32_t)maxV;
  v.k = k;
  v.mz_w = mz_w;
  v.mz_t = mz_t;
  v.shard_rank = shard_rank;
  v.shard_n = shard_n;
  v.wide = wide() ? 1 : 0;
  v.seq_only = seq_only ? 1 : 0;
  v.compact = compact ? 1 : 0;
  v.quot = quot ? 1 : 0;
  v.qshift = 0;
  for (uint64_t x = nlines; x > 1; x >>= 1) ++v.qshift;      // floor(log2(nlines))
  v.side = compact ? d_slots + nlines * MFX_SLOTS_LINE : nullptr;      // the side table follows the main lines
  v.side_nlines = side_nlines;
  return v;
}

extern "C" double mfx_index_estimate_gb(int k, uint64_t capacity_kmers) {
  return (double)lines_for(capacity_kmers, k > MFX_MAX_K_NARROW ? MFX_WSLOTS_LINE : MFX_SLOTS_LINE) * MFX_ALIGN / 1e9;
}

// bytes of the smallest 16-byte-slot / compact sequence-only table for this capacity
static double seq_plain_bytes(uint64_t capacity_kmers) { return (double)lines_for(capacity_kmers, MFX_SLOTS_LINE) * MFX_ALIGN; }
static double seq_compact_bytes(int k, uint64_t capacity_kmers) {
  return (double)(std::max(lines_for(capacity_kmers, MFX_CSLOTS_LINE), quot_min_lines(k)) + side_lines_for(capacity_kmers)) * MFX_ALIGN;
}
// The quotient form needs a table of at least 2^(2(k-3)-31) lines (4 GB at k = 31) whatever the genome's size: a small genome under a
// small -memory limit (or on a nearly full device) takes the 16-byte slots, sized by the genome, instead (budget_bytes 0: no limit known).
static bool seq_compact_fits(int k, uint64_t capacity_kmers, double budget_bytes) {
  if (!seq_compact(k)) return false;
  if (k <= MFX_MAX_K_DIRECT || budget_bytes <= 0) return true;
  const double cb = seq_compact_bytes(k, capacity_kmers);
  return cb <= budget_bytes || cb <= seq_plain_bytes(capacity_kmers);
}

extern "C" double mfx_index_estimate_gb_for_seq(int k, uint64_t capacity_kmers) {
  if (k > MFX_MAX_K_NARROW) return mfx_index_estimate_gb(k, capacity_kmers);
  if (!seq_compact(k)) return seq_plain_bytes(capacity_kmers) / 1e9;
  // (the smaller of the two layouts a run may take: index_create falls back to the 16-byte slots when the quotient form's floor does not fit)
  return std::min(seq_compact_bytes(k, capacity_kmers), k > MFX_MAX_K_DIRECT ? seq_plain_bytes(capacity_kmers) : 1e300) / 1e9;
}

static mfx_index *index_create(int k, uint64_t capacity_kmers, double max_gb, int device, bool seq_only);

extern "C" mfx_index *mfx_index_create(int k, uint64_t capacity_kmers, double max_gb, int device) {
  return index_create(k, capacity_kmers, max_gb, device, false);
}

extern "C" mfx_index *mfx_index_create_for_seq(int k, uint64_t capacity_kmers, double max_gb, int device) {
  if (k > MFX_MAX_K_NARROW) {
    mfx_fail(MFX_E_INVAL, "a sequence-only index handles k <= %d; this one would hold %d-mers (use mfx_index_create)", MFX_MAX_K_NARROW, k);
    return nullptr;
  }
  return index_create(k, capacity_kmers, max_gb, device, true);
}

// The table's load factor chosen by the caller instead of by the free memory (0: as mfx_index_create_for_seq; MFX_LOAD_FACTOR still
// overrides).  The emptiest table is the fastest to probe (3 Gb, k = 21: 151.8 G k-mers/s at 0.18 = 135 GB against 136 G at 0.4 = 61 GB)
// but not the fastest RUN: a process that starts behind another one waits in hipMalloc while the driver clears what that one freed,
// and how long depends on how much of the HBM both want (profiles/r05_e2e_lf_ab.txt: `merfin -hist` at 3 Gb back to back 4.5-4.9 s at
// 0.18, 1.2-1.3 s at 0.4) -- the CLI asks for 0.4, a resident service for 0.18.
extern "C" mfx_index *mfx_index_create_for_seq_lf(int k, uint64_t capacity_kmers, double max_gb, int device, double load_factor) {
  t_lf_request = load_factor > 0 ? load_factor : 0;
  mfx_index *ix = mfx_index_create_for_seq(k, capacity_kmers, max_gb, device);
  t_lf_request = 0;
  return ix;
}

// (the full table likewise: the variant modes and -completeness of the CLI ask for the smallest table, 0.7 -- their device stage is a
// hundredth of the run, the table's allocation behind another process is seconds)
extern "C" mfx_index *mfx_index_create_lf(int k, uint64_t capacity_kmers, double max_gb, int device, double load_factor) {
  t_lf_request = load_factor > 0 ? load_factor : 0;
  mfx_index *ix = mfx_index_create(k, capacity_kmers, max_gb, device);
  t_lf_request = 0;
  return ix;
}

static mfx_index *index_create(int k, uint64_t capacity_kmers, double max_gb, int device, bool seq_only) {
  if (k < 1 || k > MFX_MAX_K) {
    mfx_fail(MFX_E_INVAL, "k=%d unsupported: k-mers hold 2k <= 128 bits, 1 <= k <= 64", k);
    return nullptr;
  }
  if (mfx_device_count() <= device || device < 0) {
    mfx_fail(MFX_E_NODEVICE, "HIP device %d not available (%d visible); merfin_amd has no CPU path", device, mfx_device_count());
    return nullptr;
  }
  bool compact = seq_only && seq_compact(k);
  if (compact && k > MFX_MAX_K_DIRECT) {                      // the quotient form's floor against -memory and the free device memory
    double budget = max_gb > 0 ? max_gb * 1e9 : 0;
    DevGuard bg(device);
    size_t free_b = 0, total_b = 0;
    if (bg.ok && hipMemGetInfo(&free_b, &total_b) == hipSuccess) { const double fb = MFX_LF_HBM_SHARE * (double)free_b; budget = budget > 0 ? std::min(budget, fb) : fb; }
    else (void)hipGetLastError();
    compact = seq_compact_fits(k, capacity_kmers, budget);
  }
  const uint32_t slots_line = k > MFX_MAX_K_NARROW ? MFX_WSLOTS_LINE : compact ? MFX_CSLOTS_LINE : MFX_SLOTS_LINE;
  if (lines_for(capacity_kmers, slots_line) >= 0xfffffff0ull) {
    mfx_fail(MFX_E_INVAL, "capacity of %lu k-mers needs more than 2^32 table lines (512 GB); shard the index instead",
             (unsigned long)capacity_kmers);
    return nullptr;
  }
  double need = !seq_only ? mfx_index_estimate_gb(k, capacity_kmers) : k > MFX_MAX_K_NARROW ? mfx_index_estimate_gb(k, capacity_kmers)
                : compact ? seq_compact_bytes(k, capacity_kmers) / 1e9 : seq_plain_bytes(capacity_kmers) / 1e9;
  if (max_gb > 0 && need > max_gb) {
    // merfin-globals.C:148-153
    mfx_fail(MFX_E_NOMEM, "Not enough memory to load databases.  Increase -memory. (need %.3f GB, limit %.3f GB)", need, max_gb);
    return nullptr;
  }
  DevGuard g(device);
  if (!g.ok) { mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
  mfx_index *ix = new mfx_index;
  ix->device = device;
  ix->k = k;
  ix->capacity_kmers = capacity_kmers;
  ix->seq_only = seq_only;
  ix->compact = compact;
  ix->quot = compact && k > MFX_MAX_K_DIRECT;
  ix->side_nlines = compact ? side_lines_for(capacity_kmers) : 0;
  {
    size_t free_b = 0, total_b = 0;
    double budget = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = MFX_LF_HBM_SHARE * (double)free_b;
    if (max_gb > 0) budget = budget > 0 ? std::min(budget, max_gb * 1e9) : max_gb * 1e9;
    if (budget > 0) budget = std::max(1.0, budget - (double)ix->side_nlines * MFX_ALIGN);
    ix->nlines = lines_auto(capacity_kmers, budget, slots_line, seq_only);
    if (ix->quot) ix->nlines = std::max(ix->nlines, quot_min_lines(k));     // (k = 31: 4 GB of table at least; a small genome's table is mostly air)
  }
  if (ix->nlines >= (1ull << 32)) {        // line numbers are 32-bit on the device (550 GB of table: beyond one GPU anyway)
    mfx_fail(MFX_E_NOMEM, "Not enough memory to load databases.  %lu k-mers need %.0f GB on one GPU; shard the index (mfx_index_set_shard).",
             (unsigned long)capacity_kmers, (double)ix->nlines * MFX_ALIGN / 1e9);
    delete ix;
    return nullptr;
  }
  {
    // placement: "mz" (default) = minimizer-keyed home line, consecutive k-mers share 128-byte lines;
    // "plain" = k-mer hash only.  w = 3 windows (m = k-2) by default; w = 2 bounds every minimizer's bucket by
    // one line (a (k-1)-mer is contained in at most 8 k-mers) at 0.69 instead of 0.53 line fetches per k-mer.
    const char *hm = getenv("MFX_HOME_MODE");
    bool mz = hm ? (strcmp(hm, "plain") != 0) : true;
    const char *ws = getenv("MFX_MZ_W");
    const int w_default = compact ? MFX_MZ_W_COMPACT : MFX_MZ_W_DEFAULT;
    int w = ws ? atoi(ws) : w_default;
    if (w < 1 || w > 5) w = w_default;
    ix->mz_w = (mz && !ix->wide()) ? std::min(w, k) : 0;    // 128-bit k-mers: plain hashing (mfx_wide.hip)
    // compact layout: the window is sampled by the k-mer's smallest t-mer (mod-minimizer: 0.30 instead of 0.40 lines per k-mer);
    // t makes (k - t) % 4 == 3, which makes the choice the same on both strands.  MFX_MZ_MOD=0: the smallest m-mer, as the other layouts.
    {
      const char *mm = getenv("MFX_MZ_MOD");
      // (w = 5, MFX_MZ_W=5, direct form only: t = 4 .. 8 with (k - t) % 5 == 4)
      const bool modw = ix->mz_w == 4 || (ix->mz_w == 5 && !ix->quot);
      ix->mz_t = (compact && modw && k >= 13 && !(mm && atoi(mm) == 0)) ? (ix->mz_w == 4 ? ((k + 1) & 3) + 4 : 4 + ((k - 8) % 5)) : 0;
    }
  }
  hipError_t e = hipMalloc((void **)&ix->d_slots, ix->total_lines() * MFX_ALIGN);
  if (e != hipSuccess && ix->nlines > std::max(lines_for(capacity_kmers, slots_line), quot_min_lines(ix->quot ? k : 0))) {
    // the roomier table did not fit after all (fragmentation, another process): fall back to the smallest one
    (void)hipGetLastError();
    ix->nlines = std::max(lines_for(capacity_kmers, slots_line), quot_min_lines(ix->quot ? k : 0));
    e = hipMalloc((void **)&ix->d_slots, ix->total_lines() * MFX_ALIGN);
  }
  if (e != hipSuccess) {
    mfx_fail(MFX_E_NOMEM, "hipMalloc of %.3f GB for the k-mer table failed: %s", need, hipGetErrorString(e));
    delete ix;
    return nullptr;
  }
  if (hipMalloc((void **)&ix->d_meta, MFX_META_WORDS * sizeof(uint64_t)) != hipSuccess ||
      mfx_memset_now(ix->d_meta, 0, MFX_META_WORDS * sizeof(uint64_t)) != hipSuccess ||
      (ix->wide() ? hipMemsetAsync(ix->d_slots, 0, ix->nlines * MFX_ALIGN, nullptr)           // state 0 = empty
       : ix->compact ? hipMemsetAsync(ix->d_slots, 0xff, ix->nlines * MFX_ALIGN, nullptr)      // 8-byte slots: the all-ones word is empty
                     : mfx_k_table_init(ix->d_slots, ix->nlines * MFX_SLOTS_LINE, nullptr)) != hipSuccess ||
      (ix->compact && mfx_k_table_init(ix->d_slots + ix->nlines * MFX_SLOTS_LINE, ix->side_nlines * MFX_SLOTS_LINE, nullptr) != hipSuccess) ||
      hipDeviceSynchronize() != hipSuccess) {
    mfx_fail(MFX_E_HIP, "k-mer table initialisation failed: %s", hipGetErrorString(hipGetLastError()));
    mfx_index_free(ix);
    return nullptr;
  }
  return ix;
}

extern "C" void mfx_index_free(mfx_index *ix) {
  if (!ix) return;
  DevGuard g(ix->device);
  mfx_index_ingest_release(ix);
  if (ix->d_slots) (void)hipFree(ix->d_slots);
  if (ix->d_meta) (void)hipFree(ix->d_meta);
  delete ix;
}

static mfx_ingest *ingest_get(mfx_index *ix, uint64_t n);

static int index_check(mfx_index *ix) {
  ix->version++;
  uint64_t meta[5];
  MFX_HIP(hipMemcpy(meta, ix->d_meta, sizeof(meta), hipMemcpyDeviceToHost));
  if (meta[4] != 0)
    return mfx_fail(MFX_E_FORMAT, "the database holds %lu records wider than 2k = %d bits: damaged or not a %d-mer database (none was inserted)",
                    (unsigned long)meta[4], 2 * ix->k, ix->k);
  if (meta[2] != 0)
    return mfx_fail(MFX_E_FULL, "k-mer table full: %lu inserts hit the probe limit (capacity %lu k-mers, %lu stored)",
                    (unsigned long)meta[2], (unsigned long)ix->capacity_kmers, (unsigned long)meta[0]);
  if ((double)meta[0] > 0.92 * (double)(ix->nlines * ix->slots_per_line()))
    return mfx_fail(MFX_E_FULL, "k-mer table over-full: %lu k-mers in %lu slots; create the index with a larger capacity",
                    (unsigned long)meta[0], (unsigned long)(ix->nlines * ix->slots_per_line()));
  if (ix->seq_only && meta[1] != 0)
    return mfx_fail(MFX_E_NONCANON, "the database holds %lu non-canonical k-mers: a sequence-only index keeps one slot per canonical k-mer of the "
                    "sequence and cannot answer value(fmer) + value(rmer) for it; build a full index (mfx_index_create)", (unsigned long)meta[1]);
  return MFX_OK;
}

// Host -> table pipeline.  LANES of pinned staging are filled by the host and emptied over PCIe into a RING of device
// buffers; the insert kernel of a chunk runs on the device buffer.  The two are decoupled (round 4): a lane is free again
// as soon as its chunk has crossed the link, a device buffer once its chunk is inserted -- so the transfers run ahead of
// the inserts by the depth of the ring (up to MFX_INGEST_RING_BYTES, 3 GB), e.g. while the kernel that claims the
// sequence's k-mers is still running (mfx_index_build_for_hist: the inserts wait for it, the link does not).
// Lanes and ring belong to the index and are REUSED by every load call (pinning memory costs ~0.4 ms per MB: allocating
// them per call was 3.5 s of a 5.1 s ingest at 1 Gb); they are released with the index (or by mfx_index_ingest_release).
constexpr uint64_t MFX_INGEST_CHUNK = 1ull << 24;           // k-mers per lane: 128 MB of keys + 64 MB of counts
constexpr int MFX_INGEST_MAX_LANES = 4, MFX_INGEST_MAX_RING = 64, MFX_INGEST_STREAMS = 4;
struct mfx_ingest {
  struct Lane {
    uint64_t *hk = nullptr;
    uint32_t *hv = nullptr;
    hipEvent_t copied = nullptr;   // the lane's chunk has left the pinned buffer (recorded on the copy stream)
    bool busy = false;
  } L[MFX_INGEST_MAX_LANES];
  struct Dev {
    uint64_t *dk = nullptr;
    uint32_t *dv = nullptr;
    hipEvent_t done = nullptr;     // the buffer's chunk is inserted
    bool busy = false;
  } D[MFX_INGEST_MAX_RING];
  hipStream_t cs = nullptr;                                  // the copy stream
  hipStream_t is[MFX_INGEST_STREAMS] = {nullptr, nullptr, nullptr, nullptr};   // insert streams, ring buffer r uses is[r % 4]
  hipEvent_t after = nullptr;  // inserts wait for this event (the claim / count kernel of mfx_index_build_for_hist); owned here
  int nl = 2;                  // lanes in use (MFX_INGEST_LANES)
  int nd = 2;                  // ring buffers in use
  uint64_t cap = 0;            // k-mers per lane
  size_t kw = 1;
};

static void ingest_free(mfx_ingest *g) {
  if (!g) return;
  if (g->cs) { (void)hipStreamSynchronize(g->cs); }
  for (auto &st : g->is) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (g->cs) (void)hipStreamDestroy(g->cs);
  if (g->after) { (void)hipEventSynchronize(g->after); (void)hipEventDestroy(g->after); }
  for (auto &l : g->L) {
    if (l.hk) (void)hipHostFree(l.hk);
    if (l.hv) (void)hipHostFree(l.hv);
    if (l.copied) (void)hipEventDestroy(l.copied);
  }
  for (auto &d : g->D) {
    if (d.dk) (void)hipFree(d.dk);
    if (d.dv) (void)hipFree(d.dv);
    if (d.done) (void)hipEventDestroy(d.done);
  }
  delete g;
}

void mfx_index_ingest_release(mfx_index *ix) {
  if (!ix || !ix->ingest) return;
  DevGuard g(ix->device);
  ingest_free(ix->ingest);
  ix->ingest = nullptr;
}

static mfx_ingest *ingest_get(mfx_index *ix, uint64_t n) {
  uint64_t chunk = MFX_INGEST_CHUNK;
  if (const char *e = getenv("MFX_INGEST_CHUNK_LOG2")) { const int lg = atoi(e); if (lg >= 16 && lg <= 28) chunk = 1ull << lg; }
  const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(n, 1), chunk);
  // lanes of 4 M k-mers or more serve any load (it goes through them in chunks): re-pinning larger ones costs more than they save
  if (ix->ingest && (ix->ingest->cap >= want || ix->ingest->cap >= (1ull << 22))) return ix->ingest;
  hipEvent_t keep_after = nullptr;                          // (an event the inserts must wait for survives a re-sizing of the lanes)
  if (ix->ingest) { keep_after = ix->ingest->after; ix->ingest->after = nullptr; }
  mfx_index_ingest_release(ix);
  // small loads (tests, single contigs) get small lanes; the first large one gets the full-size lanes
  uint64_t cap = 1ull << 16;
  while (cap < want) cap <<= 1;
  mfx_ingest *g = new mfx_ingest;
  g->after = keep_after;
  g->cap = cap;
  g->kw = ix->key_words();
  // three or more lanes let the fill of a chunk, the transfer of the one before and the insert of the one before that overlap;
  // small loads keep two (pinning costs 0.4 ms per MB)
  g->nl = cap >= (1ull << 20) ? 4 : 2;
  if (const char *e = getenv("MFX_INGEST_LANES")) { const int v = atoi(e); if (v >= 2 && v <= MFX_INGEST_MAX_LANES) g->nl = v; }
  // the ring: as many device buffers as MFX_INGEST_RING_BYTES hold (large loads), never fewer than the lanes
  const uint64_t buf_bytes = cap * 8 * g->kw + cap * 4;
  uint64_t ring_bytes = cap >= (1ull << 20) ? (3ull << 30) : 0;
  if (const char *e = getenv("MFX_INGEST_RING_MB")) ring_bytes = (uint64_t)std::max(0, atoi(e)) << 20;
  g->nd = (int)std::min<uint64_t>(MFX_INGEST_MAX_RING, std::max<uint64_t>((uint64_t)g->nl, ring_bytes / buf_bytes));
  bool ok = hipStreamCreateWithFlags(&g->cs, hipStreamNonBlocking) == hipSuccess;
  for (int si = 0; si < MFX_INGEST_STREAMS && ok; ++si) ok = hipStreamCreateWithFlags(&g->is[si], hipStreamNonBlocking) == hipSuccess;
  for (int li = 0; li < g->nl && ok; ++li) {
    auto &l = g->L[li];
    ok = hipHostMalloc((void **)&l.hk, cap * 8 * g->kw, hipHostMallocPortable) == hipSuccess &&      // DMA source for any device
         hipHostMalloc((void **)&l.hv, cap * 4, hipHostMallocPortable) == hipSuccess &&
         hipEventCreateWithFlags(&l.copied, hipEventDisableTiming) == hipSuccess;
  }
  for (int di = 0; di < g->nd && ok; ++di) {
    auto &d = g->D[di];
    const bool got = hipMalloc((void **)&d.dk, cap * 8 * g->kw) == hipSuccess && hipMalloc((void **)&d.dv, cap * 4) == hipSuccess &&
                     hipEventCreateWithFlags(&d.done, hipEventDisableTiming) == hipSuccess;
    if (!got) {
      (void)hipGetLastError();
      if (d.dk) { (void)hipFree(d.dk); d.dk = nullptr; }
      if (d.dv) { (void)hipFree(d.dv); d.dv = nullptr; }
      if (d.done) { (void)hipEventDestroy(d.done); d.done = nullptr; }
      if (di >= g->nl) { g->nd = di; break; }                // a shorter ring will do (the HBM is full of table)
      ok = false;
    }
  }
  if (!ok) { (void)hipGetLastError(); ingest_free(g); return nullptr; }
  ix->ingest = g;
  return g;
}

// The staging loop of every host-side load.  next(cap, hk, hv, d): put the next chunk of the source into the pinned lane
// buffers (room: cap * 8 * key_words bytes in hk, cap * 4 in hv) and describe it in d -- 1 = a chunk is ready, 0 = the
// source is done, -1 = it failed.  ONE source, nix tables: the chunk is staged once (in the first index's pinned lane)
// and sent to every index's device over that device's own PCIe link; each table inserts what it keeps (a sharded index
// skips the k-mers it does not own in the insert kernel).  nix = 1 is the ordinary load.
struct IngestChunk {
  size_t kbytes = 0, vbytes = 0;                             // what goes over the link from hk / hv
  // enqueue the insert of the chunk (device copies at dk / dv) on st
  std::function<hipError_t(mfx_index *ix, uint64_t *dk, uint32_t *dv, hipStream_t st)> launch;
};

template <class Next>
static int index_ingest_chunks(mfx_index *const *ixs, uint32_t nix, uint64_t n_hint, Next &&next) {
  const bool timing = getenv("MFX_INGEST_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_wait = 0, t_fill = 0;
  uint64_t nchunks = 0, nbytes = 0;
  std::vector<mfx_ingest *> gs(nix, nullptr);
  for (uint32_t i = 0; i < nix; ++i) {
    DevGuard dg(ixs[i]->device);
    ixs[i]->frozen = true;                                  // a sequence-only index takes no more claims once counts arrive
    gs[i] = ingest_get(ixs[i], n_hint);
    if (!gs[i]) return mfx_fail(MFX_E_NOMEM, "mfx_index_add: staging allocation failed");
  }
  const double t_lanes = now() - t_begin;
  uint64_t cap = gs[0]->cap;
  for (uint32_t i = 1; i < nix; ++i) cap = std::min(cap, gs[i]->cap);
  bool ok = true, src_ok = true;
  int nl = gs[0]->nl;
  for (uint32_t i = 1; i < nix; ++i) nl = std::min(nl, gs[i]->nl);
  // the inserts of a table wait for whatever still claims its k-mers (mfx_index_build_for_hist); the transfers do not
  for (uint32_t i = 0; i < nix && ok; ++i)
    if (gs[i]->after) {
      DevGuard dg(ixs[i]->device);
      for (auto &st : gs[i]->is) if (hipStreamWaitEvent(st, gs[i]->after, 0) != hipSuccess) ok = false;
    }
  std::vector<uint64_t> ring_at(nix, 0);                     // chunks sent to table i so far: chunk c lands in ring buffer c % nd
  for (int cur = 0; ok; cur = (cur + 1) % nl) {
    const double t0 = now();
    for (uint32_t i = 0; i < nix && ok; ++i) {               // the lane's previous chunk has left the pinned buffer everywhere
      mfx_ingest::Lane &l = gs[i]->L[cur];
      if (l.busy) { DevGuard dg(ixs[i]->device); if (hipEventSynchronize(l.copied) != hipSuccess) ok = false; }
      l.busy = false;
    }
    if (!ok) break;
    mfx_ingest::Lane &src = gs[0]->L[cur];
    IngestChunk d;
    const double t1 = now();
    const int got = next(cap, src.hk, src.hv, d);
    t_wait += t1 - t0; t_fill += now() - t1;
    if (got < 0) src_ok = false;
    if (got <= 0) break;
    ++nchunks; nbytes += d.kbytes + d.vbytes;
    for (uint32_t i = 0; i < nix && ok; ++i) {
      mfx_index *ix = ixs[i];
      mfx_ingest *g = gs[i];
      const int r = (int)(ring_at[i]++ % (uint64_t)g->nd);
      mfx_ingest::Dev &dv = g->D[r];
      hipStream_t ist = g->is[r % MFX_INGEST_STREAMS];
      DevGuard dg(ix->device);
      // the ring buffer is free once its previous chunk is inserted: the COPY STREAM waits for that, not the host
      ok = (!dv.busy || hipStreamWaitEvent(g->cs, dv.done, 0) == hipSuccess) &&
           (d.kbytes == 0 || hipMemcpyAsync(dv.dk, src.hk, d.kbytes, hipMemcpyHostToDevice, g->cs) == hipSuccess) &&
           (d.vbytes == 0 || hipMemcpyAsync(dv.dv, src.hv, d.vbytes, hipMemcpyHostToDevice, g->cs) == hipSuccess) &&
           hipEventRecord(g->L[cur].copied, g->cs) == hipSuccess &&
           hipStreamWaitEvent(ist, g->L[cur].copied, 0) == hipSuccess &&
           d.launch(ix, dv.dk, dv.dv, ist) == hipSuccess && hipEventRecord(dv.done, ist) == hipSuccess;
      g->L[cur].busy = true;
      dv.busy = true;
    }
  }
  for (uint32_t i = 0; i < nix; ++i) {
    DevGuard dg(ixs[i]->device);
    if (hipStreamSynchronize(gs[i]->cs) != hipSuccess) ok = false;
    for (auto &st : gs[i]->is) if (hipStreamSynchronize(st) != hipSuccess) ok = false;
    for (int li = 0; li < gs[i]->nl; ++li) gs[i]->L[li].busy = false;
    for (int di = 0; di < gs[i]->nd; ++di) gs[i]->D[di].busy = false;
    if (gs[i]->after) {                                      // what the inserts waited for is over as well: its errors are in meta
      if (hipEventSynchronize(gs[i]->after) != hipSuccess) ok = false;
      (void)hipEventDestroy(gs[i]->after);
      gs[i]->after = nullptr;
    }
  }
  if (timing)
    fprintf(stderr, "-- ingest: %.3f s = lanes %.3f (cap %llu, ring %d) + fill %.3f + waits for the link %.3f + drain; %llu chunks, %.2f GB over the link\n",
            now() - t_begin, t_lanes, (unsigned long long)cap, gs[0]->nd, t_fill, t_wait, (unsigned long long)nchunks, nbytes / 1e9);
  if (!ok) return mfx_fail(MFX_E_HIP, "mfx_index_add: transfer / insert failed: %s", hipGetErrorString(hipGetLastError()));
  if (!src_ok) return mfx_last_error_code() ? mfx_last_error_code() : MFX_E_IO;
  for (uint32_t i = 0; i < nix; ++i) {
    DevGuard dg(ixs[i]->device);
    int rc = index_check(ixs[i]);
    if (rc) return rc;
  }
  return MFX_OK;
}

// fill(o, m, hk, hv): put k-mers [o, o+m) of the source into the pinned lane buffers; false = the source failed.
// packed: the records carry their counts (MFX_PACKED_VBITS), nothing comes through hv.
template <class Fill>
static int index_ingest_multi(mfx_index *const *ixs, uint32_t nix, uint64_t n, int side, Fill &&fill, bool packed = false) {
  uint64_t o = 0;
  return index_ingest_chunks(ixs, nix, n, [&](uint64_t cap, uint64_t *hk, uint32_t *hv, IngestChunk &d) {
    if (o >= n) return 0;
    const uint64_t m = std::min(cap, n - o);
    if (!fill(o, m, hk, hv)) return -1;
    o += m;
    d.kbytes = m * 8 * ixs[0]->key_words();
    d.vbytes = packed ? 0 : m * 4;
    d.launch = [m, side, packed](mfx_index *ix, uint64_t *dk, uint32_t *dv, hipStream_t st) {
      return ix->wide() ? mfx_kw_table_add(ix->view(), dk, dv, m, side, ix->d_meta, st)
                        : mfx_k_table_add(ix->view(), dk, packed ? nullptr : dv, m, side, ix->d_meta, st);
    };
    return 1;
  });
}

template <class Fill>
static int index_ingest(mfx_index *ix, uint64_t n, int side, Fill &&fill) {
  return index_ingest_multi(&ix, 1, n, side, fill);
}

static int set_read_filter(mfx_index *ix, uint64_t minV, uint64_t maxV) {
  if (ix->filter_set && (ix->minV != minV || ix->maxV != maxV))
    return mfx_fail(MFX_E_INVAL, "mfx_index_add_read: -min/-max must be the same for every batch of one index");
  ix->minV = minV; ix->maxV = maxV; ix->filter_set = true;
  return MFX_OK;
}

static int check_same_kind(mfx_index *const *ixs, uint32_t nix, const char *who) {
  if (!ixs || nix == 0) return mfx_fail(MFX_E_INVAL, "%s: no index", who);
  for (uint32_t i = 0; i < nix; ++i)
    if (!ixs[i] || ixs[i]->k != ixs[0]->k) return mfx_fail(MFX_E_INVAL, "%s: the indexes of one load must hold the same k", who);
  return MFX_OK;
}

// host arrays into several tables at once (the shards of one process): side 0 read counts (filter applies), 1 assembly
int mfx_index_add_multi(mfx_index *const *ixs, uint32_t nix, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side,
                        uint64_t minV, uint64_t maxV) {
  int rc = check_same_kind(ixs, nix, "mfx_index_add_multi");
  if (rc) return rc;
  if (n && (!kmers || !values)) return mfx_fail(MFX_E_INVAL, "mfx_index_add_multi: null argument");
  if (side == 0) for (uint32_t i = 0; i < nix; ++i) if ((rc = set_read_filter(ixs[i], minV, maxV)) != MFX_OK) return rc;
  const size_t kw = ixs[0]->key_words();
  return index_ingest_multi(ixs, nix, n, side, [&](uint64_t o, uint64_t m, uint64_t *hk, uint32_t *hv) {
    par_memcpy((uint8_t *)hk, (const char *)(kmers + o * kw), m * 8 * kw);
    par_memcpy((uint8_t *)hv, (const char *)(values + o), m * 4);
    return true;
  });
}

// n bytes at file offset `off` into dst, read by several threads (each pread copies straight out of the page cache:
// no mapping to fault in page by page, which is what made the mmap + memcpy route top out at ~7 GB/s)
static unsigned pread_threads() {
  // these threads wait on memory, not on the ALUs: more of them than the CPU quota grants still pays (1 Gb ingest on
  // a 16-core quota: 1.9 s with 16 readers, 1.3 s with 32), unless MFX_HOST_THREADS fixes the count
  if (const char *pe = getenv("MFX_PREAD_THREADS")) if (atoi(pe) > 0) return std::min((unsigned)atoi(pe), 64u);   // (A/B)
  unsigned nt = std::max(1u, mfx_host_threads());
  if (!getenv("MFX_HOST_THREADS")) nt = std::max(nt, std::min(32u, std::max(1u, std::thread::hardware_concurrency())));
  return std::min(nt, 64u);
}

// pool: reader threads that live for the whole file (a database is read in ~100 chunks: starting and joining 32 threads for
// each was a third of the time of a chunk once the records shrank to 8 bytes); nullptr: threads of this call
static bool par_pread(int fd, uint8_t *dst, size_t n, uint64_t off, WorkerPool *pool = nullptr) {
  const unsigned nt = pool ? pool->W : pread_threads();
  auto rd = [fd](uint8_t *d, size_t len, uint64_t o) {
    while (len) {
      ssize_t r = pread(fd, d, len, (off_t)o);
      if (r <= 0) return false;
      d += r; o += (uint64_t)r; len -= (size_t)r;
    }
    return true;
  };
  if (n < (8u << 20) || nt == 1) return rd(dst, n, off);
  std::vector<char> good(nt, 1);
  const size_t per = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
  if (pool) {
    pool->start([&](unsigned t) {
      const size_t b = std::min(n, t * per), e = std::min(n, b + per);
      if (e > b) good[t] = rd(dst + b, e - b, off + b) ? 1 : 0;
    });
    pool->wait();
    for (char c : good) if (!c) return false;
    return true;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    const size_t b = std::min(n, t * per), e = std::min(n, b + per);
    if (e > b) th.emplace_back([&, t, b, e]() { good[t] = rd(dst + b, e - b, off + b) ? 1 : 0; });
  }
  for (auto &x : th) x.join();
  for (char c : good) if (!c) return false;
  return true;
}

// k-mers [0, n) of an open flat-binary database: keys at keys_off (8 * key_words bytes each), counts at vals_off
int mfx_index_add_from_file(mfx_index *const *ixs, uint32_t nix, int fd, const char *path, uint64_t keys_off, uint64_t vals_off, uint64_t n,
                            int side, uint64_t minV, uint64_t maxV) {
  int rc = check_same_kind(ixs, nix, "mfx_index_add_from_file");
  if (rc) return rc;
  if (fd < 0) return mfx_fail(MFX_E_INVAL, "mfx_index_add_from_file: bad argument");
  if (side == 0) for (uint32_t i = 0; i < nix; ++i) if ((rc = set_read_filter(ixs[i], minV, maxV)) != MFX_OK) return rc;
  const size_t kw = ixs[0]->key_words();
  const bool packed = vals_off == 0;
  if (packed && kw != 1) return mfx_fail(MFX_E_INVAL, "mfx_index_add_from_file: packed records hold k-mers of k <= %d", MFX_MAX_K_PACKED);
  std::unique_ptr<WorkerPool> pool(n * 8 * kw >= (64u << 20) ? new WorkerPool(pread_threads(), true) : nullptr);
  return index_ingest_multi(ixs, nix, n, side, [&](uint64_t o, uint64_t m, uint64_t *hk, uint32_t *hv) {
    if (par_pread(fd, (uint8_t *)hk, m * 8 * kw, keys_off + o * 8 * kw, pool.get()) &&
        (packed || par_pread(fd, (uint8_t *)hv, m * 4, vals_off + o * 4, pool.get()))) return true;
    mfx_fail(MFX_E_IO, "reading '%s' failed", path);
    return false;
  }, packed);
}

// the delta-coded blocks of an open flat database (mfx_db.cpp FLAT_DELTA): dir = (nblocks + 1) x {first k-mer, file offset |
// kbits << 48 | vbits << 56}, n k-mers in all.  Whole blocks go through the lanes as they lie in the file -- 2.5-3 bytes
// per k-mer of a 30x read set -- and are decoded by the kernel that inserts them (mfx_table_add_delta_kernel).
int mfx_index_add_delta_file(mfx_index *const *ixs, uint32_t nix, int fd, const char *path, const uint64_t *dir, uint64_t nblocks, uint64_t n,
                             int side, uint64_t minV, uint64_t maxV, int placed) {
  int rc = check_same_kind(ixs, nix, "mfx_index_add_delta_file");
  if (rc) return rc;
  if (fd < 0 || !dir || ixs[0]->key_words() != 1) return mfx_fail(MFX_E_INVAL, "mfx_index_add_delta_file: bad argument");
  if (side == 0) for (uint32_t i = 0; i < nix; ++i) if ((rc = set_read_filter(ixs[i], minV, maxV)) != MFX_OK) return rc;
  auto off = [dir](uint64_t b) { return dir[2 * b + 1] & 0xffffffffffffull; };
  const uint64_t total = off(nblocks) - off(0);
  std::unique_ptr<WorkerPool> pool(total >= (64u << 20) ? new WorkerPool(pread_threads(), true) : nullptr);
  uint64_t b0 = 0;
  // lanes sized as for total / 8 records (32 MB at most: ~13 M k-mers a chunk): a chunk is a byte range of the file, not a k-mer count
  return index_ingest_chunks(ixs, nix, std::min<uint64_t>(std::max<uint64_t>(total / 8 + 1, 2 * MFX_DELTA_BLOCK * 11), 1ull << 22), [&](uint64_t cap, uint64_t *hk, uint32_t *hv, IngestChunk &d) {
    if (b0 >= nblocks) return 0;
    uint64_t b1 = b0 + 1;                                    // at least one block (a lane holds the largest possible block)
    while (b1 < nblocks && off(b1 + 1) - off(b0) <= cap * 8 && (b1 + 2 - b0) * 16 <= cap * 4) ++b1;
    const uint64_t bytes = off(b1) - off(b0), nb = b1 - b0;
    if (bytes > cap * 8 || (nb + 1) * 16 > cap * 4) { mfx_fail(MFX_E_FORMAT, "'%s': a block larger than the format allows", path); return -1; }
    if (!par_pread(fd, (uint8_t *)hk, bytes, off(b0), pool.get())) { mfx_fail(MFX_E_IO, "reading '%s' failed", path); return -1; }
    memcpy(hv, dir + 2 * b0, (nb + 1) * 16);
    const uint64_t base = off(b0), m = std::min<uint64_t>(n - b0 * MFX_DELTA_BLOCK, nb * MFX_DELTA_BLOCK);
    d.kbytes = bytes;
    d.vbytes = (nb + 1) * 16;
    d.launch = [nb, m, base, side, placed](mfx_index *ix, uint64_t *dk, uint32_t *dv, hipStream_t st) {
      return placed ? mfx_k_table_add_placed(ix->view(), dk, reinterpret_cast<const uint64_t *>(dv), (uint32_t)nb, m, base, side, ix->d_meta, st)
                    : mfx_k_table_add_delta(ix->view(), dk, reinterpret_cast<const uint64_t *>(dv), (uint32_t)nb, m, base, side, ix->d_meta, st);
    };
    b0 = b1;
    return 1;
  });
}

static int index_add(mfx_index *ix, const uint64_t *kmers, const uint32_t *values, uint64_t n, int side, int on_device) {
  if (!ix || (n && (!kmers || !values))) return mfx_fail(MFX_E_INVAL, "mfx_index_add: null argument");
  DevGuard g(ix->device);
  const size_t kw = ix->key_words();                      // uint64 words per k-mer (2 for k > 31)
  auto table_add = [&](const uint64_t *dk, const uint32_t *dv, uint64_t m, hipStream_t s) {
    return ix->wide() ? mfx_kw_table_add(ix->view(), dk, dv, m, side, ix->d_meta, s) : mfx_k_table_add(ix->view(), dk, dv, m, side, ix->d_meta, s);
  };
  if (on_device) {
    ix->frozen = true;
    MFX_HIP(table_add(kmers, values, n, nullptr));
    MFX_HIP(hipDeviceSynchronize());
    return index_check(ix);
  }
  return index_ingest(ix, n, side, [&](uint64_t o, uint64_t m, uint64_t *hk, uint32_t *hv) {
    par_memcpy((uint8_t *)hk, (const char *)(kmers + o * kw), m * 8 * kw);
    par_memcpy((uint8_t *)hv, (const char *)(values + o), m * 4);
    return true;
  });
}

extern "C" int mfx_index_add_read(mfx_index *ix, const uint64_t *kmers, const uint32_t *values, uint64_t n,
                                  uint64_t minV, uint64_t maxV, int on_device) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_add_read: null index");
  if (ix->filter_set && (ix->minV != minV || ix->maxV != maxV))
    return mfx_fail(MFX_E_INVAL, "mfx_index_add_read: -min/-max must be the same for every batch of one index");
  ix->minV = minV;
  ix->maxV = maxV;
  ix->filter_set = true;
  return index_add(ix, kmers, values, n, 0, on_device);
}

extern "C" int mfx_index_add_asm(mfx_index *ix, const uint64_t *kmers, const uint32_t *values, uint64_t n, int on_device) {
  return index_add(ix, kmers, values, n, 1, on_device);
}

// defer: the kernel is launched and NOT waited for -- an event recorded behind it is left in the index's staging state
// (mfx_ingest::after), the inserts of the load that follows wait for it on the device, and that load's final check reads what
// both left in meta (mfx_index_build_for_hist)
// no_wait: the kernel is launched on `stream` and neither waited for nor checked -- the caller orders what follows behind it and checks
static int index_count(mfx_index *ix, const mfx_seq *seq, int count, void *stream, const char *who, bool defer = false, bool no_wait = false) {
  if (!ix || !seq) return mfx_fail(MFX_E_INVAL, "%s: null argument", who);
  if (ix->device != seq->device) return mfx_fail(MFX_E_INVAL, "index and sequence live on different devices");
  if (ix->seq_only && ix->frozen && count != 2)
    return mfx_fail(MFX_E_INVAL, "%s: this sequence-only index already took counts; its k-mers must all be claimed before the first add / load "
                    "(a k-mer claimed now would have missed them)", who);
  if (count == 2 && (!ix->seq_only || ix->wide())) return mfx_fail(MFX_E_INVAL, "%s: counting claimed k-mers needs a sequence-only index (mfx_index_create_for_seq)", who);
  // the k <= 31 kernel reads the packed planes when the sequence has them (a packed upload never makes the bytes)
  if (seq->partial) return mfx_seq_partial_error(seq, who);
  const bool from_planes = !ix->wide() && (seq->planes_ok || seq->bases_stale) && !(getenv("MFX_COUNT_ASCII") && atoi(getenv("MFX_COUNT_ASCII")));
  if (!from_planes) if (int erc = mfx_seq_ensure_ascii(seq)) return erc;
  DevGuard g(ix->device);
  mfx_count_args a;
  a.t = ix->view();
  a.bases = seq->d_bases;
  if (from_planes) { a.codes = seq->d_codes; a.valid = seq->d_valid; }
  a.contig_off = seq->d_contig_off;
  a.contig_len = seq->d_contig_len;
  a.tile_start = seq->d_tile_start;
  a.ncontigs = seq->ncontigs;
  a.ntiles = seq->ntiles;
  a.meta = ix->d_meta;
  a.count = count;
  if (ix->seq_only && count != 2) {                            // what this table can answer for: the k-mers of THIS sequence
    uint32_t d = 0;
    if (int drc = mfx_seq_digest32(seq, &d)) return drc;
    // one sequence per sequence-only index: a second claim from ANOTHER sequence would leave the index bound to the last one and the
    // evaluation of the first refused ("holds the k-mers of another sequence") -- say so here, where the mistake is made
    if (ix->seq_digest != 0 && ix->seq_digest != d)
      return mfx_fail(MFX_E_INVAL, "%s: this sequence-only index already claimed the k-mers of another sequence (content digest %08x, this one %08x); "
                      "one such index answers for ONE sequence object -- put the contigs into one mfx_seq", who, ix->seq_digest, d);
    ix->seq_digest = d;
  }
  if (count == 2) ix->frozen = true;                           // counts arrived: no more claims
  MFX_HIP(ix->wide() ? mfx_kw_count(a, (hipStream_t)stream) : mfx_k_count(a, (hipStream_t)stream));
  if (no_wait) return MFX_OK;
  // While the kernel claims / counts a large sequence's k-mers (0.11 s for 3 Gb), the host pins the staging lanes the database
  // load that follows will want (0.06 s): the lanes belong to the index and are reused by every load.
  if (defer) {
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev, (hipStream_t)stream);
    mfx_ingest *g = e == hipSuccess ? ingest_get(ix, 1ull << 22) : nullptr;      // (pins the lanes while the kernel runs)
    if (g) {
      if (g->after) { (void)hipEventSynchronize(g->after); (void)hipEventDestroy(g->after); }
      g->after = ev;
      return MFX_OK;
    }
    if (ev) (void)hipEventDestroy(ev);
    (void)hipGetLastError();                                  // no staging state: fall through to the waited form
  } else if (ix->seq_only && seq->total_bases >= (256ull << 20)) (void)ingest_get(ix, 1ull << 22);
  MFX_HIP(hipStreamSynchronize((hipStream_t)stream));
  return index_check(ix);
}

// an event the inserts were to wait for is waited for here and dropped (error paths of mfx_index_build_for_hist)
static void ingest_settle_after(mfx_index *ix) {
  if (!ix || !ix->ingest || !ix->ingest->after) return;
  DevGuard g(ix->device);
  (void)hipEventSynchronize(ix->ingest->after);
  (void)hipEventDestroy(ix->ingest->after);
  ix->ingest->after = nullptr;
}

extern "C" int mfx_index_count_asm(mfx_index *ix, const mfx_seq *seq, void *stream) {
  return index_count(ix, seq, 1, stream, "mfx_index_count_asm");
}

// The assembly counts of the k-mers claimed BEFORE (mfx_index_claim_seq), taken from another -- usually larger -- sequence:
// asmV += 1 per occurrence of a claimed k-mer, nothing is claimed.  This is how a device that evaluates PART of an assembly
// (some contigs) gets value() right for its k-mers: they are claimed from its contigs and counted over the whole assembly
// (`meryl count` of -sequence counts every contig, merfin-globals.C:182-186), and the read database then updates them.
extern "C" int mfx_index_count_claimed(mfx_index *ix, const mfx_seq *seq, void *stream) {
  return index_count(ix, seq, 2, stream, "mfx_index_count_claimed");
}

// mfx_index_count_asm + mfx_index_load_db(side 0) as ONE call -- what `merfin -hist -sequence s -readmers db` needs
// (load_Kmers, merfin-globals.C:114-163 + the `meryl count` of -sequence, :182-186) -- with the database's bytes crossing
// PCIe while the sequence's k-mers are still being claimed and counted: the inserts wait for that kernel on the device,
// the transfers fill the staging ring meanwhile.  Same table as the two calls one after the other.
extern "C" int mfx_index_build_for_hist(mfx_index *ix, const mfx_seq *seq, const char *read_db_path, uint64_t minV, uint64_t maxV) {
  if (!ix || !seq || !read_db_path) return mfx_fail(MFX_E_INVAL, "mfx_index_build_for_hist: null argument");
  const char *ov = getenv("MFX_BUILD_OVERLAP");              // 0: the two calls one after the other (A/B, tests)
  const bool defer = !(ov && atoi(ov) == 0);
  int rc = index_count(ix, seq, 1, nullptr, "mfx_index_build_for_hist", defer);
  if (rc == MFX_OK) rc = mfx_index_load_db(ix, read_db_path, 0, minV, maxV);
  ingest_settle_after(ix);                                   // (a load that failed before its staging loop leaves the event behind)
  return rc;
}

// ---------------------------------------------------------------------------
// STAGED load of the read database (load_Kmers, merfin-globals.C:114-163, as `merfin -hist` pays it per run).  The database's bytes do
// not depend on anything else of the run, so they start moving when the process starts: mfx_db_stage_begin opens a delta-coded flat
// database, takes device memory for ALL of its blocks + directory, and a thread of its own reads the file through three pinned
// lanes into that memory (one copy stream) -- under the FASTA read, the sequence upload, the table's allocation and the kernel that
// claims and counts the sequence's k-mers.  mfx_index_build_for_hist_staged then launches that kernel and, behind it, the decode +
// update kernel over the staged blocks, chunk by chunk as their copies complete (the kernels wait for the copy events on the device).
// The link is busy from the first 0.1 s of the process instead of from the moment the table exists; same table as
// mfx_index_build_for_hist.  nullptr from _begin (another database form, too little free HBM): the caller takes the unstaged call.
// ---------------------------------------------------------------------------
struct mfx_db_stage {
  int device = 0, fd = -1;
  std::string path;
  mfx_flat_delta_info info;
  std::vector<uint64_t> dir;                                  // (nblocks + 1) x {first k-mer, file offset | kbits << 48 | vbits << 56}
  uint64_t off0 = 0, payload_bytes = 0;
  uint8_t *d_payload = nullptr;
  uint64_t *d_dir = nullptr;
  uint64_t *d_esc_k = nullptr;                                // the escape list (k-mers whose count did not fit their block's field), staged like the blocks
  uint32_t *d_esc_v = nullptr;
  hipEvent_t esc_copied = nullptr;
  std::atomic<int> esc_ready{0};                              // the escape copies are enqueued and esc_copied is recorded
  std::atomic<int> boost{0};                                  // 0: a few reader threads (the FASTA reader has the host); 1: all of them (mfx_db_stage_boost)
  struct Chunk { uint64_t b0, b1; hipEvent_t copied = nullptr; };
  std::vector<Chunk> chunks;
  std::atomic<int64_t> enqueued{0};                           // chunks whose copy is enqueued and whose event is recorded
  std::atomic<int> failed{0};
  std::string error;
  std::thread worker;
  double t_begin = 0, t_first_copy = 0, t_last_enqueued = 0, t_all_copied = 0;
  double t_setup[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // begin: file + directory, memory info, device memory, events; worker: device + stream, first lane, first read, first enqueue
  double t_part[2][3] = {{0, 0, 0}, {0, 0, 0}};              // [before / after the boost][lane wait, file read, enqueue] seconds of the worker (diagnostics)
  uint64_t n_part[2] = {0, 0}, b_part[2] = {0, 0};
  uint64_t file_off(uint64_t b) const { return dir[2 * b + 1] & 0xffffffffffffull; }
};

static double stage_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void stage_worker(mfx_db_stage *S) {
  constexpr int NLMAX = 8;
  int NL = 3;
  if (const char *le = getenv("MFX_DB_STAGE_LANES")) NL = std::min(NLMAX, std::max(2, atoi(le)));
  const size_t LANE = 32u << 20;
  uint8_t *lane[NLMAX] = {nullptr};
  hipEvent_t left[NLMAX] = {nullptr};
  hipStream_t cs = nullptr;
  bool busy[NLMAX] = {false};
  auto fail = [&](const char *what, hipError_t e) {
    S->error = std::string(what) + (e != hipSuccess ? std::string(": ") + hipGetErrorString(e) : std::string());
    S->failed.store(1);
  };
  double tw0 = stage_now();
  hipError_t e = hipSetDevice(S->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
  S->t_setup[4] = stage_now() - tw0;
  // (a lane is pinned when its first chunk comes up -- 13 ms each -- and the directory goes over behind the first chunk: the first
  // bytes of the database are on the link as early as the runtime allows)
  auto lane_up = [&](int i) -> hipError_t {
    if (lane[i]) return hipSuccess;
    hipError_t le = hipHostMalloc((void **)&lane[i], LANE, hipHostMallocDefault);
    if (le == hipSuccess) le = hipEventCreateWithFlags(&left[i], hipEventDisableTiming);
    return le;
  };
  if (e != hipSuccess) fail("staging set-up failed", e);
  {
    // While the FASTA file is being read and encoded the host's threads are busy with that (a full pool here slowed the sequence
    // down by more than the database gained: profiles/r05_e2e_staged.txt); until the caller says the sequence is in
    // (mfx_db_stage_boost) a few readers keep the link fed, then all of them.
    unsigned few = 8;                                          // (2 / 4 / 8: 3 Gb wall 1.23-1.25 / 1.20-1.23 / 1.18-1.19 s spaced, profiles/r05_e2e_staged.txt)
    if (const char *fe = getenv("MFX_DB_STAGE_THREADS")) few = (unsigned)std::max(1, atoi(fe));
    std::unique_ptr<WorkerPool> pool_few(new WorkerPool(std::min(few, pread_threads()), true)), pool_all;
    auto pool_now = [&]() -> WorkerPool * {
      if (!S->boost.load(std::memory_order_relaxed)) return pool_few.get();
      if (!pool_all) pool_all.reset(new WorkerPool(pread_threads(), true));
      return pool_all.get();
    };
    for (size_t c = 0; c < S->chunks.size() && !S->failed.load(); ++c) {
      const int li = (int)(c % NL);
      const int ph = S->boost.load(std::memory_order_relaxed) ? 1 : 0;              // (diagnostics: before / after mfx_db_stage_boost)
      double tq = stage_now();
      if ((e = lane_up(li)) != hipSuccess) { fail("staging set-up failed", e); break; }
      if (c == 0) S->t_setup[5] = stage_now() - tq;
      if (busy[li] && (e = hipEventSynchronize(left[li])) != hipSuccess) { fail("staging copy failed", e); break; }
      const uint64_t o = S->file_off(S->chunks[c].b0), bytes = S->file_off(S->chunks[c].b1) - o;
      double tr = stage_now();
      S->t_part[ph][0] += tr - tq;
      if (!par_pread(S->fd, lane[li], bytes, o, pool_now())) { fail("reading the database failed", hipSuccess); break; }
      tq = stage_now();
      S->t_part[ph][1] += tq - tr;
      if (c == 0) S->t_setup[6] = tq - tr;
      S->n_part[ph] += 1;
      S->b_part[ph] += bytes;
      if (c == 0) S->t_first_copy = stage_now();
      e = hipMemcpyAsync(S->d_payload + (o - S->off0), lane[li], bytes, hipMemcpyHostToDevice, cs);
      if (e == hipSuccess && c == 0) e = hipMemcpyAsync(S->d_dir, S->dir.data(), S->dir.size() * 8, hipMemcpyHostToDevice, cs);     // (ahead of every chunk's `copied` event but the first's, which the next line records behind it)
      if (e == hipSuccess) e = hipEventRecord(left[li], cs);
      if (e == hipSuccess) e = hipEventRecord(S->chunks[c].copied, cs);
      if (e != hipSuccess) { fail("staging copy failed", e); break; }
      busy[li] = true;
      S->enqueued.store((int64_t)c + 1, std::memory_order_release);
      S->t_part[ph][2] += stage_now() - tq;
      if (c == 0) S->t_setup[7] = stage_now() - tq;
    }
    // the escape list behind the blocks: k-mers (8 bytes each), then their counts (4 bytes each)
    const uint64_t ne = S->info.n_escape;
    for (int part = 0; part < 2 && ne && !S->failed.load(); ++part) {
      const uint64_t width = part == 0 ? 8 : 4, total = ne * width, at = S->info.escapes_off + (part == 0 ? 0 : ne * 8);
      uint8_t *dst = part == 0 ? reinterpret_cast<uint8_t *>(S->d_esc_k) : reinterpret_cast<uint8_t *>(S->d_esc_v);
      for (uint64_t o = 0, c = S->chunks.size(); o < total && !S->failed.load(); o += LANE, ++c) {
        const int li = (int)(c % NL);
        const uint64_t bytes = std::min<uint64_t>(LANE, total - o);
        if ((e = lane_up(li)) != hipSuccess) { fail("staging set-up failed", e); break; }
        if (busy[li] && (e = hipEventSynchronize(left[li])) != hipSuccess) { fail("staging copy failed", e); break; }
        if (!par_pread(S->fd, lane[li], bytes, at + o, pool_now())) { fail("reading the database failed", hipSuccess); break; }
        if (part == 0) {                                      // a k-mer of k bases has no bit at or above 2k (a damaged file)
          const uint64_t *kk = reinterpret_cast<const uint64_t *>(lane[li]);
          bool wide = false;
          if (S->info.k < 32) for (uint64_t i = 0; i < bytes / 8; ++i) wide |= (kk[i] >> (2 * S->info.k)) != 0;
          if (wide) { fail("an escaped k-mer is wider than 2k bits", hipSuccess); break; }
        }
        e = hipMemcpyAsync(dst + o, lane[li], bytes, hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(left[li], cs);
        if (e != hipSuccess) { fail("staging copy failed", e); break; }
        busy[li] = true;
      }
    }
    if (!S->failed.load()) {
      e = hipEventRecord(S->esc_copied, cs);
      if (e != hipSuccess) fail("staging copy failed", e);
      else S->esc_ready.store(1, std::memory_order_release);
    }
  }
  S->t_last_enqueued = stage_now();
  if (cs) (void)hipStreamSynchronize(cs);
  S->t_all_copied = stage_now();
  for (int i = 0; i < NL; ++i) { if (lane[i]) (void)hipHostFree(lane[i]); if (left[i]) (void)hipEventDestroy(left[i]); }
  if (cs) (void)hipStreamDestroy(cs);
}

extern "C" void mfx_db_stage_free(mfx_db_stage *S) {
  if (!S) return;
  S->failed.store(1);                                          // (a worker still on the file stops at its next chunk)
  if (S->worker.joinable()) S->worker.join();
  DevGuard g(S->device);
  for (auto &c : S->chunks) if (c.copied) (void)hipEventDestroy(c.copied);
  if (S->d_payload) (void)hipFree(S->d_payload);
  if (S->d_dir) (void)hipFree(S->d_dir);
  if (S->d_esc_k) (void)hipFree(S->d_esc_k);
  if (S->d_esc_v) (void)hipFree(S->d_esc_v);
  if (S->esc_copied) (void)hipEventDestroy(S->esc_copied);
  if (S->fd >= 0) close(S->fd);
  delete S;
}

// the sequence is read and uploaded: the stager may use every reader thread from here on
extern "C" void mfx_db_stage_boost(mfx_db_stage *S) {
  if (S) S->boost.store(1, std::memory_order_relaxed);
}

extern "C" mfx_db_stage *mfx_db_stage_begin(const char *path, int device) {
  if (!path) { mfx_fail(MFX_E_INVAL, "mfx_db_stage_begin: null argument"); return nullptr; }
  if (device < 0 || device >= mfx_device_count()) { mfx_fail(MFX_E_NODEVICE, "mfx_db_stage_begin: HIP device %d not available (%d visible)", device, mfx_device_count()); return nullptr; }
  if (const char *e = getenv("MFX_DB_STAGE")) if (atoi(e) == 0) { mfx_fail(MFX_E_INVAL, "staged load disabled (MFX_DB_STAGE=0)"); return nullptr; }
  std::unique_ptr<mfx_db_stage> S(new mfx_db_stage);
  S->t_begin = stage_now();
  S->device = device;
  S->path = path;
  if (mfx_flat_delta_open(path, &S->fd, &S->info, S->dir)) return nullptr;
  S->t_setup[0] = stage_now() - S->t_begin;
  auto drop = [&](mfx_db_stage *x) { mfx_db_stage_free(x); return (mfx_db_stage *)nullptr; };
  if (S->info.n == 0 || S->info.k > MFX_MAX_K_NARROW) { mfx_fail(MFX_E_INVAL, "'%s': nothing to stage", path); return drop(S.release()); }
  S->off0 = S->file_off(0);
  S->payload_bytes = S->file_off(S->info.nblocks) - S->off0;
  DevGuard g(device);
  if (!g.ok) { mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", device); return drop(S.release()); }
  size_t free_b = 0, total_b = 0;
  double tq = stage_now();
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); mfx_fail(MFX_E_HIP, "hipMemGetInfo failed"); return drop(S.release()); }
  S->t_setup[1] = stage_now() - tq; tq = stage_now();
  // a fifth of the free HBM at most: the table is sized by what is left (a read database beyond that goes through the ring)
  if ((double)(S->payload_bytes + S->dir.size() * 8 + S->info.n_escape * 12) > 0.2 * (double)free_b) {
    mfx_fail(MFX_E_NOMEM, "'%s': %.1f GB of blocks against %.1f GB of free device memory: not staged", path, S->payload_bytes / 1e9, free_b / 1e9);
    return drop(S.release());
  }
  if (hipMalloc((void **)&S->d_payload, S->payload_bytes + 64) != hipSuccess || hipMalloc((void **)&S->d_dir, S->dir.size() * 8) != hipSuccess ||
      hipEventCreateWithFlags(&S->esc_copied, hipEventDisableTiming) != hipSuccess ||
      (S->info.n_escape && (hipMalloc((void **)&S->d_esc_k, S->info.n_escape * 8) != hipSuccess || hipMalloc((void **)&S->d_esc_v, S->info.n_escape * 4) != hipSuccess))) {
    (void)hipGetLastError();
    mfx_fail(MFX_E_NOMEM, "'%s': no device memory for the staged database", path);
    return drop(S.release());
  }
  S->t_setup[2] = stage_now() - tq; tq = stage_now();
  // chunks: whole blocks, at most 32 MB of file each
  const uint64_t LANE = 32u << 20;
  for (uint64_t b0 = 0; b0 < S->info.nblocks;) {
    uint64_t b1 = b0 + 1;
    while (b1 < S->info.nblocks && S->file_off(b1 + 1) - S->file_off(b0) <= LANE) ++b1;
    if (S->file_off(b1) - S->file_off(b0) > LANE) { mfx_fail(MFX_E_FORMAT, "'%s': a block larger than the format allows", path); return drop(S.release()); }
    mfx_db_stage::Chunk c;
    c.b0 = b0; c.b1 = b1;
    if (hipEventCreateWithFlags(&c.copied, hipEventDisableTiming) != hipSuccess) { mfx_fail(MFX_E_HIP, "event creation failed"); return drop(S.release()); }
    S->chunks.push_back(c);
    b0 = b1;
  }
  S->t_setup[3] = stage_now() - tq;
  mfx_db_stage *raw = S.release();
  raw->worker = std::thread(stage_worker, raw);
  return raw;
}

// seq != nullptr: the claim / count kernel of the sequence first (mfx_index_build_for_hist_staged); nullptr: the staged database alone, into side
// `side` of a table whose k-mers are there or claimed already (mfx_index_load_db_staged)
static int staged_load(mfx_index *ix, const mfx_seq *seq, mfx_db_stage *S, int side, uint64_t minV, uint64_t maxV, const char *who) {
  if (ix->device != S->device) return mfx_fail(MFX_E_INVAL, "%s: index and staged database live on different devices", who);
  if (S->info.k != ix->k) return mfx_fail(MFX_E_INVAL, "'%s' holds %d-mers but the index is built for k=%d", S->path.c_str(), S->info.k, ix->k);
  if (ix->wide()) return mfx_fail(MFX_E_INVAL, "%s: k <= 31 only", who);
  const bool timing = getenv("MFX_INGEST_TIMING") != nullptr;
  const double t0 = stage_now();
  int rc = side == 0 ? set_read_filter(ix, minV, maxV) : MFX_OK;
  if (rc) return rc;
  DevGuard g(ix->device);
  hipStream_t is[MFX_INGEST_STREAMS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t counted = nullptr;
  auto release = [&]() {
    for (auto &st : is) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    if (counted) (void)hipEventDestroy(counted);
  };
#define STAGED_HIP(call)                                                                                          \
  do {                                                                                                            \
    hipError_t e_ = (call);                                                                                       \
    if (e_ != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); release(); return rc; } \
  } while (0)
  for (auto &st : is) STAGED_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  STAGED_HIP(hipEventCreateWithFlags(&counted, hipEventDisableTiming));
  // the claim / count kernel on the first insert stream; every insert stream waits for it
  if (seq) {
    rc = index_count(ix, seq, 1, is[0], who, false, /*no_wait=*/true);
    if (rc) { release(); return rc; }
    STAGED_HIP(hipEventRecord(counted, is[0]));
    for (int i = 1; i < MFX_INGEST_STREAMS; ++i) STAGED_HIP(hipStreamWaitEvent(is[i], counted, 0));
  }
  ix->frozen = true;
  const double t1 = stage_now();
  double t_wait = 0;
  for (size_t c = 0; c < S->chunks.size(); ++c) {
    const double tw = stage_now();
    while (S->enqueued.load(std::memory_order_acquire) <= (int64_t)c) {
      if (S->failed.load()) { release(); return mfx_fail(MFX_E_IO, "'%s': %s", S->path.c_str(), S->error.c_str()); }
      std::this_thread::sleep_for(std::chrono::microseconds(50));     // (not a yield spin: under a CPU quota it would hold a core against the stager's readers)
    }
    t_wait += stage_now() - tw;
    const mfx_db_stage::Chunk &ch = S->chunks[c];
    hipStream_t st = is[c % MFX_INGEST_STREAMS];
    STAGED_HIP(hipStreamWaitEvent(st, ch.copied, 0));
    const uint64_t nb = ch.b1 - ch.b0, m = std::min<uint64_t>(S->info.n - ch.b0 * MFX_DELTA_BLOCK, nb * MFX_DELTA_BLOCK);
    STAGED_HIP(S->info.placed ? mfx_k_table_add_placed(ix->view(), reinterpret_cast<const uint64_t *>(S->d_payload), S->d_dir + 2 * ch.b0, (uint32_t)nb, m, S->off0, side, ix->d_meta, st)
                              : mfx_k_table_add_delta(ix->view(), reinterpret_cast<const uint64_t *>(S->d_payload), S->d_dir + 2 * ch.b0, (uint32_t)nb, m, S->off0, side, ix->d_meta, st));
  }
  // the escapes, from the staged copy: an ordinary update of (k-mer, count) arrays that are already on the device
  while (!S->esc_ready.load(std::memory_order_acquire)) {
    if (S->failed.load()) { release(); return mfx_fail(MFX_E_IO, "'%s': %s", S->path.c_str(), S->error.c_str()); }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  if (S->info.n_escape) {
    STAGED_HIP(hipStreamWaitEvent(is[0], S->esc_copied, 0));
    STAGED_HIP(mfx_k_table_add(ix->view(), S->d_esc_k, S->d_esc_v, S->info.n_escape, side, ix->d_meta, is[0]));
  }
  const double t2 = stage_now();
  for (auto &st : is) STAGED_HIP(hipStreamSynchronize(st));
#undef STAGED_HIP
  const double t3 = stage_now();
  release();
  if (S->failed.load()) return mfx_fail(MFX_E_IO, "'%s': %s", S->path.c_str(), S->error.c_str());
  rc = index_check(ix);
  if (timing)
    fprintf(stderr, "-- staged build: %.3f s = count launched %.3f + %zu update launches + escapes %.3f (of which waiting for the stager %.3f) + drain %.3f + check %.3f; "
            "stager: first copy %.3f s after its start, last copy enqueued after %.3f s (%.2f GB); the build began %.3f s after the stager\n",
            stage_now() - t0, t1 - t0, S->chunks.size(), t2 - t1, t_wait, t3 - t2, stage_now() - t3, S->t_first_copy - S->t_begin, S->t_last_enqueued - S->t_begin,
            S->payload_bytes / 1e9, t0 - S->t_begin);
  if (timing)
    fprintf(stderr, "-- stager set-up: begin = file + directory %.3f, memory info %.3f, device memory %.3f, events %.3f; worker = device + stream %.3f, first lane %.3f, first read %.3f, "
            "first enqueue %.3f s\n", S->t_setup[0], S->t_setup[1], S->t_setup[2], S->t_setup[3], S->t_setup[4], S->t_setup[5], S->t_setup[6], S->t_setup[7]);
  if (timing)
    for (int ph = 0; ph < 2; ++ph)
      fprintf(stderr, "-- stager %s the sequence was in: %lu chunks, %.2f GB; the worker waited for a lane %.3f s, read the file %.3f s (%.1f GB/s), enqueued copies %.3f s\n",
              ph ? "after" : "before", (unsigned long)S->n_part[ph], S->b_part[ph] / 1e9, S->t_part[ph][0], S->t_part[ph][1],
              S->t_part[ph][1] > 0 ? S->b_part[ph] / 1e9 / S->t_part[ph][1] : 0.0, S->t_part[ph][2]);
  return rc;
}

extern "C" int mfx_index_build_for_hist_staged(mfx_index *ix, const mfx_seq *seq, mfx_db_stage *S, uint64_t minV, uint64_t maxV) {
  if (!ix || !seq || !S) return mfx_fail(MFX_E_INVAL, "mfx_index_build_for_hist_staged: null argument");
  return staged_load(ix, seq, S, 0, minV, maxV, "mfx_index_build_for_hist_staged");
}
// mfx_index_load_db of a STAGED database: its bytes have been on their way into device memory since mfx_db_stage_begin; only the decode +
// insert kernels are left (side 0: -readmers with the -min/-max filter, 1: -seqmers).  Same table as mfx_index_load_db.
extern "C" int mfx_index_load_db_staged(mfx_index *ix, mfx_db_stage *S, int side, uint64_t minV, uint64_t maxV) {
  if (!ix || !S || (side != 0 && side != 1)) return mfx_fail(MFX_E_INVAL, "mfx_index_load_db_staged: null argument or side not 0 / 1");
  return staged_load(ix, nullptr, S, side, minV, maxV, "mfx_index_load_db_staged");
}

// P (mfx_place.h) of n k-mers -- canonicalised first --: on_device != 0: both arrays are device pointers on `device` (tools that sort a
// database on the GPU); else host arrays, host threads
extern "C" int mfx_db_place_keys(int k, const uint64_t *kmers, uint64_t n, uint64_t *out, int on_device, int device) {
  if (n && (!kmers || !out)) return mfx_fail(MFX_E_INVAL, "mfx_db_place_keys: null argument");
  if (k < MFX_PLACE_MIN_K || k > MFX_PLACE_MAX_K || mfx_p_split(k))      // (k = 31: P takes 65 bits; its placed database is made by mfx_db_convert_placed)
    return mfx_fail(MFX_E_INVAL, "mfx_db_place_keys: %d <= k <= 30 (k = %d)", MFX_PLACE_MIN_K, k);
  if (!on_device) { mfx_place_keys_host(k, kmers, n, out); return MFX_OK; }
  DevGuard g(device);
  MFX_HIP(mfx_k_place_keys(k, kmers, n, out, nullptr));
  MFX_HIP(hipStreamSynchronize(nullptr));
  return MFX_OK;
}

extern "C" int mfx_index_claim_seq(mfx_index *ix, const mfx_seq *seq, void *stream) {
  if (ix && !ix->seq_only) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_seq: not a sequence-only index (mfx_index_create_for_seq)");
  return index_count(ix, seq, 0, stream, "mfx_index_claim_seq");
}

extern "C" int mfx_index_value(const mfx_index *ix, const uint64_t *kmers, uint64_t n, uint32_t *readV, uint32_t *asmV) {
  if (!ix || (n && (!kmers || !readV || !asmV))) return mfx_fail(MFX_E_INVAL, "mfx_index_value: null argument");
  DevGuard g(ix->device);
  DevBuf<uint64_t> dk;
  DevBuf<uint32_t> dr, da;
  MFX_HIP(dk.alloc(n * ix->key_words()));
  MFX_HIP(dr.alloc(n));
  MFX_HIP(da.alloc(n));
  MFX_HIP(hipMemcpy(dk.p, kmers, n * ix->key_words() * sizeof(uint64_t), hipMemcpyHostToDevice));
  MFX_HIP(ix->wide() ? mfx_kw_table_value(ix->view(), dk.p, n, dr.p, da.p, nullptr) : mfx_k_table_value(ix->view(), dk.p, n, dr.p, da.p, nullptr));
  MFX_HIP(hipMemcpy(readV, dr.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  MFX_HIP(hipMemcpy(asmV, da.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return MFX_OK;
}

extern "C" int mfx_index_get_info(const mfx_index *ix, mfx_index_info *out) {
  if (!ix || !out) return mfx_fail(MFX_E_INVAL, "mfx_index_get_info: null argument");
  DevGuard g(ix->device);
  uint64_t meta[4];
  MFX_HIP(hipMemcpy(meta, ix->d_meta, sizeof(meta), hipMemcpyDeviceToHost));
  out->k = ix->k;
  out->canonical = (meta[1] == 0) ? 1 : 0;
  out->capacity = ix->nlines * ix->slots_per_line();
  out->distinct = meta[0];
  out->bytes = ix->total_lines() * MFX_ALIGN;
  out->seq_only = ix->seq_only ? 1 : 0;
  out->compact = ix->compact ? 1 : 0;
  out->dropped = meta[3];
  return MFX_OK;
}

extern "C" int mfx_index_export(const mfx_index *ix, uint64_t *kmers, uint32_t *readV, uint32_t *asmV, uint64_t *n_out) {
  if (!ix || !kmers || !readV || !asmV || !n_out) return mfx_fail(MFX_E_INVAL, "mfx_index_export: null argument");
  DevGuard g(ix->device);
  mfx_index_info info;
  int rc = mfx_index_get_info(ix, &info);
  if (rc) return rc;
  DevBuf<uint64_t> dk;
  DevBuf<uint32_t> dr, da;
  DevBuf<unsigned long long> dc;
  MFX_HIP(dk.alloc(info.distinct * ix->key_words()));
  MFX_HIP(dr.alloc(info.distinct));
  MFX_HIP(da.alloc(info.distinct));
  MFX_HIP(dc.alloc(1));
  MFX_HIP(mfx_memset_now(dc.p, 0, sizeof(unsigned long long)));
  MFX_HIP(ix->wide() ? mfx_kw_table_export(ix->view(), dk.p, dr.p, da.p, dc.p, nullptr) : mfx_k_table_export(ix->view(), dk.p, dr.p, da.p, dc.p, nullptr));
  unsigned long long cnt = 0;
  MFX_HIP(hipMemcpy(&cnt, dc.p, sizeof(cnt), hipMemcpyDeviceToHost));
  MFX_HIP(hipMemcpy(kmers, dk.p, cnt * ix->key_words() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  MFX_HIP(hipMemcpy(readV, dr.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
  MFX_HIP(hipMemcpy(asmV, da.p, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost));
  *n_out = cnt;
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// sequences
// ---------------------------------------------------------------------------
static mfx_seq *seq_layout(int device, const uint64_t *lens, uint32_t ncontigs) {
  mfx_seq *s = new mfx_seq;
  s->device = device;
  s->ncontigs = ncontigs;
  s->off.resize(ncontigs);
  s->len.assign(lens, lens + ncontigs);
  s->tile_start.resize((size_t)ncontigs + 1);
  uint64_t o = 0, t = 0;
  for (uint32_t c = 0; c < ncontigs; ++c) {
    s->off[c] = o;
    s->tile_start[c] = t;
    t += (lens[c] + MFX_TILE - 1) / MFX_TILE;
    s->total_bases += lens[c];
    o = (o + lens[c] + 1 + MFX_ALIGN - 1) / MFX_ALIGN * MFX_ALIGN;   // >= 1 separator byte, next contig 128-byte aligned
  }
  s->tile_start[ncontigs] = t;
  s->ntiles = t;
  s->buf_bytes = o + MFX_TILE + 2 * MFX_ALIGN;                        // tail tiles read one tile + halo past the end
  return s;
}

// One byte per base (mfx_seq::d_bases) is what the 128-bit kernels, -dump, the router and the variant modes read; a sequence
// that arrives packed and is only ever counted and evaluated by the k <= 31 -hist kernels never needs it (3 GB for a human assembly,
// 0.05-0.15 s of its upload): it is made on first use (mfx_seq_ensure_ascii).
static int seq_need_bases(mfx_seq *s) {
  if (s->d_bases) return MFX_OK;
  MFX_HIP(hipMalloc((void **)&s->d_bases, s->buf_bytes));
  MFX_HIP(mfx_memset_now(s->d_bases, 0, s->buf_bytes));                    // byte 0 is not ACGT: separators + padding
  return MFX_OK;
}

static int seq_alloc(mfx_seq *s, bool with_bases = true) {
  if (with_bases) if (int rc = seq_need_bases(s)) return rc;
  size_t nc = s->ncontigs ? s->ncontigs : 1;
  MFX_HIP(hipMalloc((void **)&s->d_contig_off, nc * sizeof(uint64_t)));
  MFX_HIP(hipMalloc((void **)&s->d_contig_len, nc * sizeof(uint64_t)));
  MFX_HIP(hipMalloc((void **)&s->d_tile_start, (nc + 1) * sizeof(uint64_t)));
  if (s->ncontigs) {
    MFX_HIP(hipMemcpy(s->d_contig_off, s->off.data(), s->ncontigs * sizeof(uint64_t), hipMemcpyHostToDevice));
    MFX_HIP(hipMemcpy(s->d_contig_len, s->len.data(), s->ncontigs * sizeof(uint64_t), hipMemcpyHostToDevice));
  }
  MFX_HIP(hipMemcpy(s->d_tile_start, s->tile_start.data(), (s->ncontigs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> tc((size_t)s->ntiles);
    for (uint32_t c = 0; c < s->ncontigs; ++c)
      std::fill(tc.begin() + (size_t)s->tile_start[c], tc.begin() + (size_t)s->tile_start[c + 1], c);
    MFX_HIP(hipMalloc((void **)&s->d_tile_contig, (tc.size() ? tc.size() : 1) * sizeof(uint32_t)));
    if (!tc.empty()) MFX_HIP(hipMemcpy(s->d_tile_contig, tc.data(), tc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  return MFX_OK;
}

// A packed upload (mfx_hist_run_streamed) leaves the sequence in its packed planes only; the kernels that read one byte
// per base get them unpacked here, once, on first use.
// What a sequence makes on first use -- its bytes per base, its packed planes, its content digest -- is made under this lock:
// N slots that share a device share its mfx_seq (merfin -dump -devices 0,0,0,0: four threads ask one sequence for its bytes at
// once; unguarded, one thread's fill of the fresh buffer wiped what another had just unpacked and that slot dumped empty contigs).

int mfx_seq_partial_error(const mfx_seq *s, const char *who) {
  return mfx_fail(MFX_E_INVAL, "%s: the sequence object holds only the tiles [%lu, %lu) of its %lu (the part one device evaluated in a streamed run over "
                  "several); upload the sequence whole", who, (unsigned long)s->part_lo, (unsigned long)s->part_hi, (unsigned long)s->ntiles);
}

int mfx_seq_ensure_ascii(const mfx_seq *cs) {
  if (cs && cs->partial) return mfx_seq_partial_error(cs, "unpacking the sequence");
  std::lock_guard<std::mutex> lazy(cs->lazy_mu);
  if (!cs->bases_stale && cs->d_bases) return MFX_OK;
  mfx_seq *s = const_cast<mfx_seq *>(cs);
  DevGuard g(s->device);
  if (int rc = seq_need_bases(s)) return rc;
  if (!s->bases_stale) return MFX_OK;                       // (a sequence that holds nothing yet: all bytes 0)
  MFX_HIP(mfx_k_unpack(s->d_codes, s->d_valid, s->d_bases, s->buf_bytes / 32, nullptr));
  MFX_HIP(hipDeviceSynchronize());
  s->bases_stale = false;
  return MFX_OK;
}

// h: the device's sum over the words (mfx_seq_digest_kernel)
static void seq_digest_finish(const mfx_seq *s, uint64_t h) {
  for (uint32_t c = 0; c < s->ncontigs; ++c) h = (h ^ s->len[c]) * 0x100000001B3ULL + c;       // the contig structure
  const uint32_t f = (uint32_t)(h ^ (h >> 32));
  s->digest = f ? f : 1u;
}

int mfx_seq_digest32(const mfx_seq *s, uint32_t *out) {
  if (s->partial) return mfx_seq_partial_error(s, "the sequence's content digest");
  std::lock_guard<std::mutex> lazy(s->lazy_mu);
  if (s->digest == 0) {
    DevGuard g(s->device);
    uint64_t *d = nullptr, h = 0;
    MFX_HIP(hipMalloc((void **)&d, sizeof(uint64_t)));
    hipError_t e = mfx_memset_now(d, 0, sizeof(uint64_t));
    const bool planes = s->planes_ok || s->bases_stale;
    if (!planes && !s->d_bases) { (void)hipFree(d); if (int rc = seq_need_bases(const_cast<mfx_seq *>(s))) return rc; MFX_HIP(hipMalloc((void **)&d, sizeof(uint64_t))); e = mfx_memset_now(d, 0, sizeof(uint64_t)); }
    if (e == hipSuccess) e = mfx_k_seq_digest(s->d_bases, planes ? s->d_codes : nullptr, planes ? s->d_valid : nullptr, s->buf_bytes / 32, d, nullptr);
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return mfx_fail(MFX_E_HIP, "sequence digest failed: %s", hipGetErrorString(e));
    seq_digest_finish(s, h);
  }
  *out = s->digest;
  return MFX_OK;
}

thread_local bool t_mfx_path_lookup = false;     // set by the variant modes around the lookups of THEIR path text (mfx_variants.cpp)
int mfx_check_seq_of_index(const mfx_index *ix, const mfx_seq *s, const char *who) {
  if (!ix->seq_only || ix->seq_digest == 0) return MFX_OK;
  if (ix->paths_token) {
    if (t_mfx_path_lookup) return MFX_OK;
    return mfx_fail(MFX_E_INVAL, "%s: this index holds the k-mers of a variant call set's PATHS (mfx_index_claim_paths) and answers the variant modes of that "
                    "call set only; build the index from the sequence (mfx_index_create_for_seq) or use a full index", who);
  }
  uint32_t d = 0;
  if (int rc = mfx_seq_digest32(s, &d)) return rc;
  if (d != ix->seq_digest)
    return mfx_fail(MFX_E_INVAL, "%s: this sequence-only index holds the k-mers of ANOTHER sequence (content digest %08x, this one %08x): "
                    "k-mers it never claimed would read as absent; build the index from this sequence (mfx_index_create_for_seq + "
                    "mfx_index_count_asm / mfx_index_claim_seq) or use a full index", who, ix->seq_digest, d);
  return MFX_OK;
}

static int seq_upload_packed(mfx_seq *seq, const char *const *bases);
static int seq_alloc_planes(mfx_seq *s);

extern "C" mfx_seq *mfx_seq_upload(int device, const char *const *bases, const uint64_t *lens, uint32_t ncontigs) {
  if ((ncontigs && (!bases || !lens)) || device < 0 || device >= mfx_device_count()) {
    mfx_fail(device < 0 || device >= mfx_device_count() ? MFX_E_NODEVICE : MFX_E_INVAL,
             "mfx_seq_upload: bad argument (device %d of %d)", device, mfx_device_count());
    return nullptr;
  }
  DevGuard g(device);
  const double t_up0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  mfx_seq *s = seq_layout(device, lens, ncontigs);
  const double t_up1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  bool packed_transport = false;
  {
    const char *asc = getenv("MFX_UPLOAD_ASCII");
    const char *pm = getenv("MFX_UPLOAD_PACKED_MIN");          // bytes from which the packed transport pays (tests: 0)
    packed_transport = !(asc && atoi(asc)) && s->buf_bytes >= (pm ? strtoull(pm, nullptr, 10) : (uint64_t)(8u << 20));
  }
  if (seq_alloc(s, !packed_transport) != MFX_OK) { mfx_seq_free(s); return nullptr; }
  if (getenv("MFX_UPLOAD_TIMING"))
    fprintf(stderr, "-- upload: layout %.3f s, device buffers %.3f s\n", t_up1 - t_up0,
            std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_up1);
  {
    // default: the sequence crosses the bus as its packed planes (0.375 B per base, encoded by the host threads); one byte per
    // base is made on the device when a kernel asks for it (mfx_seq_ensure_ascii).  MFX_UPLOAD_ASCII=1: the bytes themselves.
    if (packed_transport) {
      if (seq_upload_packed(s, bases) != MFX_OK) { mfx_seq_free(s); return nullptr; }
      return s;
    }
  }
  // The packed image (contigs at their padded offsets, zero filler in between) is assembled in a
  // pinned staging buffer and sent in large pieces: an assembly of a million small contigs must not
  // become a million tiny hipMemcpy calls.
  // Two staging buffers alternate: while one is in flight over PCIe the host fills the other.
  // (sized to the upload: pinning memory costs ~0.4 ms per MB, which a small upload should not pay 128 MB of)
  const size_t STAGE = (size_t)std::min<uint64_t>(64ull << 20, std::max<uint64_t>(1ull << 20, ((s->buf_bytes / 2 + (1ull << 20)) >> 20) << 20));
  uint8_t *stages[2] = {nullptr, nullptr};
  hipStream_t cs = nullptr;
  hipEvent_t done_ev[2] = {nullptr, nullptr};
  bool ok = hipHostMalloc((void **)&stages[0], STAGE, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void **)&stages[1], STAGE, hipHostMallocDefault) == hipSuccess &&
            hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&done_ev[0], hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&done_ev[1], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    mfx_fail(MFX_E_NOMEM, "pinned staging setup failed");
    for (int i = 0; i < 2; ++i) { if (stages[i]) (void)hipHostFree(stages[i]); if (done_ev[i]) (void)hipEventDestroy(done_ev[i]); }
    if (cs) (void)hipStreamDestroy(cs);
    mfx_seq_free(s);
    return nullptr;
  }
  int cur = 0;
  bool inflight[2] = {false, false};
  uint8_t *stage = stages[0];
  uint64_t win = 0;                      // device offset the staging buffer currently mirrors
  size_t used = 0;                       // bytes of it that are meaningful
  auto flush = [&]() {
    if (used) {
      if (hipMemcpyAsync(s->d_bases + win, stage, used, hipMemcpyHostToDevice, cs) != hipSuccess ||
          hipEventRecord(done_ev[cur], cs) != hipSuccess) ok = false;
      inflight[cur] = true;
      cur ^= 1;
      if (inflight[cur] && hipEventSynchronize(done_ev[cur]) != hipSuccess) ok = false;   // the other buffer is free again
      inflight[cur] = false;
      stage = stages[cur];
    }
    win += used;
    used = 0;
  };
  for (uint32_t c = 0; c < ncontigs && ok; ++c) {
    uint64_t done = 0;
    while (done < lens[c] && ok) {
      const uint64_t dst = s->off[c] + done;                 // device offset of the next byte of this contig
      if (dst < win + used || dst - win >= STAGE) {          // not appendable to the current window
        flush();
        win = dst;
      }
      const size_t at = (size_t)(dst - win);
      if (at > used) memset(stage + used, 0, at - used);      // inter-contig padding stays non-ACGT
      const size_t m = (size_t)std::min<uint64_t>(lens[c] - done, STAGE - at);
      par_memcpy(stage + at, bases[c] + done, m);
      used = at + m;
      done += m;
      if (used == STAGE) flush();
    }
  }
  if (ok) flush();
  if (hipStreamSynchronize(cs) != hipSuccess) ok = false;
  for (int i = 0; i < 2; ++i) { (void)hipHostFree(stages[i]); (void)hipEventDestroy(done_ev[i]); }
  (void)hipStreamDestroy(cs);
  if (!ok) {
    mfx_fail(MFX_E_HIP, "H2D copy of the packed assembly failed");
    mfx_seq_free(s);
    return nullptr;
  }
  return s;
}

extern "C" mfx_seq *mfx_seq_from_device(int device, const void *const *d_bases, const uint64_t *lens, uint32_t ncontigs,
                                        void *stream) {
  if ((ncontigs && (!d_bases || !lens)) || device < 0 || device >= mfx_device_count()) {
    mfx_fail(MFX_E_INVAL, "mfx_seq_from_device: bad argument");
    return nullptr;
  }
  DevGuard g(device);
  mfx_seq *s = seq_layout(device, lens, ncontigs);
  if (seq_alloc(s) != MFX_OK) { mfx_seq_free(s); return nullptr; }
  for (uint32_t c = 0; c < ncontigs; ++c)
    if (lens[c] && hipMemcpyAsync(s->d_bases + s->off[c], d_bases[c], lens[c], hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      mfx_fail(MFX_E_HIP, "D2D copy of contig %u failed", c);
      mfx_seq_free(s);
      return nullptr;
    }
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
    mfx_fail(MFX_E_HIP, "D2D copy sync failed");
    mfx_seq_free(s);
    return nullptr;
  }
  return s;
}

extern "C" void mfx_seq_free(mfx_seq *s) {
  if (!s) return;
  DevGuard g(s->device);
  if (s->d_bases) (void)hipFree(s->d_bases);
  if (s->d_codes) (void)hipFree(s->d_codes);
  if (s->d_valid) (void)hipFree(s->d_valid);
  if (s->d_contig_off) (void)hipFree(s->d_contig_off);
  if (s->d_contig_len) (void)hipFree(s->d_contig_len);
  if (s->d_tile_start) (void)hipFree(s->d_tile_start);
  if (s->d_tile_contig) (void)hipFree(s->d_tile_contig);
  delete s;
}

extern "C" uint32_t mfx_seq_num_contigs(const mfx_seq *s) { return s ? s->ncontigs : 0; }
extern "C" uint64_t mfx_seq_num_bases(const mfx_seq *s) { return s ? s->total_bases : 0; }
extern "C" uint64_t mfx_seq_num_tiles(const mfx_seq *s) { return s ? s->ntiles : 0; }

// ---------------------------------------------------------------------------
// evaluator
// ---------------------------------------------------------------------------
extern "C" mfx_eval *mfx_eval_create(const mfx_index *ix, const mfx_kparams *kp, uint32_t nbins) {
  if (!ix || !kp || (kp->n_prob && (!kp->probK || !kp->probP))) {
    mfx_fail(MFX_E_INVAL, "mfx_eval_create: null argument");
    return nullptr;
  }
  DevGuard g(ix->device);
  mfx_eval *ev = new mfx_eval;
  ev->ix = ix;
  ev->device = ix->device;
  ev->peak = kp->peak;
  ev->n_prob = kp->n_prob;
  ev->probK.assign(kp->probK, kp->probK + kp->n_prob);
  ev->probP.assign(kp->probP, kp->probP + kp->n_prob);
  ev->nbins = nbins ? std::max<uint32_t>(nbins, MFX_NB_LDS) : 65536;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ix->device) != hipSuccess) {
    mfx_fail(MFX_E_HIP, "hipGetDeviceProperties failed");
    delete ev;
    return nullptr;
  }
  const char *e = getenv("MFX_BLOCKS_PER_CU");
  int bpc = e ? atoi(e) : mfx_k_hist_resident_blocks(ix->compact ? 1 : 0, ix->k);   // persistent blocks: as many as are resident at once
  if (bpc < 1) bpc = 1;
  ev->grid = prop.multiProcessorCount * bpc;
  size_t np = ev->n_prob ? ev->n_prob : 1;
  if (hipMalloc((void **)&ev->d_probK, np * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void **)&ev->d_probP, np * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&ev->d_partials, 2 * (size_t)ev->grid * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&ev->d_tile_ctr, 2 * sizeof(uint64_t)) != hipSuccess ||
      mfx_memset_now(ev->d_tile_ctr, 0, 2 * sizeof(uint64_t)) != hipSuccess ||
      hipMalloc((void **)&ev->d_ovf, MFX_OVF_WORDS * sizeof(uint64_t)) != hipSuccess ||
      hipMemsetAsync(ev->d_ovf + 2 + MFX_OVF_SLOTS, 0xff, (size_t)MFX_OVF_SLOTS * sizeof(uint64_t), nullptr) != hipSuccess ||      // keys: all ones = empty
      mfx_memset_now(ev->d_ovf, 0, (2 + (size_t)MFX_OVF_SLOTS) * sizeof(uint64_t)) != hipSuccess ||
      (ev->n_prob && hipMemcpy(ev->d_probK, ev->probK.data(), ev->n_prob * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) ||
      (ev->n_prob && hipMemcpy(ev->d_probP, ev->probP.data(), ev->n_prob * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)) {
    mfx_fail(MFX_E_HIP, "mfx_eval_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
    mfx_eval_free(ev);
    return nullptr;
  }
  // (1 - readK/asmK) * prob of every (read count < MFX_MAXP_LDS, asmV < MFX_KLUT) pair the kernel's exact tables cover, as the integer
  // the tile-driven kernel adds up (mfx_kfix): the same fp64 routines the kernel's generic path runs (mfx_kstar.h, no contraction)
  {
    std::vector<uint64_t> uq((size_t)MFX_MAXP_LDS * MFX_KLUT, 0);
    for (uint32_t rv = 0; rv < MFX_MAXP_LDS; ++rv) {
      double rk, pr;
      mfx_getK_core(ev->peak, ev->n_prob, ev->probK.data(), ev->probP.data(), rv, rk, pr);
      if (!(rk >= 1.0 && rk < (double)MFX_KLUT && rk == (double)(uint32_t)rk)) continue;
      for (uint32_t av = 1; av < MFX_KLUT; ++av)
        if ((double)av > rk) uq[(size_t)rv * MFX_KLUT + av] = mfx_kfix(mfx_overcopy_term(rk, (double)av, pr));
    }
    if (hipMalloc((void **)&ev->d_underq, uq.size() * sizeof(uint64_t)) != hipSuccess ||
        hipMemcpy(ev->d_underq, uq.data(), uq.size() * sizeof(uint64_t), hipMemcpyHostToDevice) != hipSuccess) {
      mfx_fail(MFX_E_HIP, "mfx_eval_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
      mfx_eval_free(ev);
      return nullptr;
    }
  }
  return ev;
}

extern "C" void mfx_eval_free(mfx_eval *ev) {
  if (!ev) return;
  DevGuard g(ev->device);
  if (ev->d_probK) (void)hipFree(ev->d_probK);
  if (ev->d_probP) (void)hipFree(ev->d_probP);
  if (ev->d_underq) (void)hipFree(ev->d_underq);
  if (ev->d_partials) (void)hipFree(ev->d_partials);
  if (ev->d_tile_ctr) (void)hipFree(ev->d_tile_ctr);
  if (ev->d_tile_partials) (void)hipFree(ev->d_tile_partials);
  for (auto &w : ev->d_wl) if (w) (void)hipFree(w);
  if (ev->d_ovf) (void)hipFree(ev->d_ovf);
  if (ev->d_dbg) (void)hipFree(ev->d_dbg);
  for (auto &p : ev->h_stage) if (p) (void)hipHostFree(p);
  for (auto &p : ev->h_pack) if (p) (void)hipHostFree(p);
  if (ev->pool) delete static_cast<WorkerPool *>(ev->pool);
  if (ev->d_var_scratch) (void)hipFree(ev->d_var_scratch);
  if (ev->sr.d_counts) (void)hipFree(ev->sr.d_counts);
  if (ev->sr.d_kover) (void)hipFree(ev->sr.d_kover);
  if (ev->sr.h_img) (void)hipHostFree(ev->sr.h_img);
  for (auto &e : ev->sr.up) if (e) (void)hipEventDestroy(e);
  if (ev->sr.kdone) (void)hipEventDestroy(ev->sr.kdone);
  for (auto &x : ev->sr.d_exc) if (x) (void)hipFree(x);
  if (ev->sr.copy) (void)hipStreamDestroy(ev->sr.copy);
  for (auto &k : ev->sr.kern) if (k) (void)hipStreamDestroy(k);
  delete ev;
}

extern "C" uint32_t mfx_eval_nbins(const mfx_eval *ev) { return ev ? ev->nbins : 0; }

// Test hook: from now on -hist launches of this evaluator run the DEBUG instance of the kernel where one exists (the compact
// layout's specialised k = 21 kernel: same code with four counters in its probe) -- never what bench.py measures.
extern "C" int mfx_eval_debug_enable(mfx_eval *ev, int on) {
  if (!ev) return mfx_fail(MFX_E_INVAL, "mfx_eval_debug_enable: null argument");
  DevGuard g(ev->device);
  if (on && !ev->d_dbg) {
    MFX_HIP(hipMalloc((void **)&ev->d_dbg, 8 * sizeof(uint64_t)));
    MFX_HIP(mfx_memset_now(ev->d_dbg, 0, 8 * sizeof(uint64_t)));
  } else if (!on && ev->d_dbg) {
    MFX_HIP(hipDeviceSynchronize());
    (void)hipFree(ev->d_dbg);
    ev->d_dbg = nullptr;
  }
  return MFX_OK;
}

// out[0] queries that were not in their first mini-bucket (first cooperative pass), [1] home line full of other k-mers (second
// cooperative pass), [2] a saturated count (side table), [3] per-lane whole-line scans; the counters are read and cleared
extern "C" int mfx_eval_debug_counters(mfx_eval *ev, uint64_t *out8) {
  if (!ev || !out8) return mfx_fail(MFX_E_INVAL, "mfx_eval_debug_counters: null argument");
  if (!ev->d_dbg) return mfx_fail(MFX_E_INVAL, "mfx_eval_debug_counters: not enabled (mfx_eval_debug_enable)");
  DevGuard g(ev->device);
  MFX_HIP(hipDeviceSynchronize());
  MFX_HIP(hipMemcpy(out8, ev->d_dbg, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  MFX_HIP(mfx_memset_now(ev->d_dbg, 0, 8 * sizeof(uint64_t)));
  return MFX_OK;
}

extern "C" void mfx_getK(const mfx_kparams *kp, uint32_t readV, uint32_t asmV, double *readK, double *asmK, double *prob) {
  double rk, pr;
  mfx_getK_core(kp->peak, kp->n_prob, kp->probK, kp->probP, readV, rk, pr);
  *readK = rk;
  *asmK = (double)asmV;   // merfin-globals.C:81
  *prob = pr;
}

extern "C" double mfx_getKmetric(double readK, double asmK) { return mfx_kmetric(readK, asmK); }

// merfin-histogram.C:22-31
extern "C" double mfx_histoQV(double kval, double ktot, int k) {
  double base = kval / ktot;
  double kinv = 1.0 / k;
  double qv = -10.0 * log10(1.0 - pow(1.0 - base, kinv));
  return qv;
}

// ---------------------------------------------------------------------------
// -hist
// ---------------------------------------------------------------------------
static int index_canonical(const mfx_index *ix, int *canon) {
  uint64_t meta[4];
  MFX_HIP(hipMemcpy(meta, ix->d_meta, sizeof(meta), hipMemcpyDeviceToHost));
  // single probe of min(f,r) equals value(f)+value(r) for a canonical DB: the other strand is not in it (SURVEY A-7);
  // for even k the kernels count a palindromic k-mer's slot twice.  A database with non-canonical k-mers: both strands.
  *canon = meta[1] == 0 ? 1 : 0;
  return MFX_OK;
}

// the canonical/odd-k decision costs a blocking D2H read; evaluators cache it per index version so
// that mfx_hist_launch stays asynchronous (it is called once per step in the multi-GPU loop)
static int eval_canonical(mfx_eval *ev, int *canon) {
  if (ev->canon_version != ev->ix->version) {
    int rc = index_canonical(ev->ix, &ev->canon);
    if (rc) return rc;
    ev->canon_version = ev->ix->version;
  }
  *canon = ev->canon;
  return MFX_OK;
}

static int ensure_tile_partials(mfx_eval *ev, uint64_t ntiles) {
  const uint64_t need = mfx_k_tile_partials_words(ntiles);
  if (need > ev->tile_partials_cap) {
    if (ev->d_tile_partials) (void)hipFree(ev->d_tile_partials);
    ev->d_tile_partials = nullptr;
    ev->tile_partials_cap = 0;
    MFX_HIP(hipMalloc((void **)&ev->d_tile_partials, need * sizeof(double)));
    ev->tile_partials_cap = need;
  }
  return MFX_OK;
}

// The worklist of a launch over ntl tiles (mfx_hist_rest_kernel): room for one position in 32 -- a genome with human-like repeat
// families lists 1-2 % of its positions (profiles/r06_repeats_ab.txt); a launch that lists more ends the rest per lane, as every
// launch did before round 6.  0.5 bytes per position of the largest launch so far; MFX_HIST_WORKLIST=0: no lists.
static int ensure_worklist(mfx_eval *ev, int slot, uint64_t ntl) {
  static const bool off = [] { const char *e = getenv("MFX_HIST_WORKLIST"); return e && atoi(e) == 0; }();
  if (off) return MFX_OK;
  const uint64_t want = std::min<uint64_t>(std::max<uint64_t>(1u << 16, ntl * MFX_TILE / 32), 0xffffffffull);
  if (ev->d_wl[slot] && ev->wl_cap[slot] >= want) return MFX_OK;
  if (ev->d_wl[slot]) { MFX_HIP(hipDeviceSynchronize()); (void)hipFree(ev->d_wl[slot]); ev->d_wl[slot] = nullptr; ev->wl_cap[slot] = 0; }
  if (hipMalloc((void **)&ev->d_wl[slot], (MFX_WL_HEADER + 2 * want) * sizeof(uint64_t)) != hipSuccess) {
    (void)hipGetLastError();                                  // no memory for a list: the launches scan per lane
    ev->d_wl[slot] = nullptr;
    return MFX_OK;
  }
  MFX_HIP(mfx_memset_now(ev->d_wl[slot], 0, MFX_WL_HEADER * sizeof(uint64_t)));
  ev->wl_cap[slot] = want;
  return MFX_OK;
}

// part_n == 1: tiles [tile_begin, tile_end); part_n > 1: the block-cyclic share of part_rank over ALL tiles.
// chunk_of_total == 0: a complete evaluation -- the koverCpy values of its (tile, wave)s are summed into *d_kover.
// chunk_of_total  > 0: one chunk of a streamed evaluation over `chunk_of_total` tiles in all: the values land at
// their tile's place in ev->d_tile_partials (sized by the caller) and are summed ONCE after the last chunk, so
// koverCpy is bit-identical to a single launch over the whole range, however the upload was cut.
static int hist_launch(mfx_eval *ev, const mfx_seq *seq, uint64_t tile_begin, uint64_t tile_end, uint32_t part_rank, uint32_t part_n,
                       uint32_t part_shift, uint64_t *d_counts, double *d_kover, void *stream, uint64_t chunk_of_total = 0, int ctr_slot = 0,
                       uint64_t partials_tile0 = 0) {
  uint64_t ntl = tile_end - tile_begin;
  if (part_n > 1) {
    const uint64_t blk = 1ull << part_shift, nblk = (seq->ntiles + blk - 1) / blk;
    ntl = 0;
    for (uint64_t b = part_rank; b < nblk; b += part_n) ntl += std::min<uint64_t>(blk, seq->ntiles - b * blk);
  }
  if (ntl == 0) return MFX_OK;
  DevGuard g(ev->device);
  int canon = 0;
  int rc = eval_canonical(ev, &canon);
  if (rc) return rc;
  if (!chunk_of_total) {                                     // (a streamed run checks once its upload is complete)
    if (seq->partial) return mfx_seq_partial_error(seq, "-hist");
    rc = mfx_check_seq_of_index(ev->ix, seq, "-hist");
    if (rc) return rc;
  }
  const char *force = getenv("MFX_FORCE_TWO_STRAND");
  if (force && atoi(force)) canon = 0;
  if (!chunk_of_total) {
    rc = ensure_tile_partials(ev, ntl);
    if (rc) return rc;
  }
  if (seq->bases_stale && ev->ix->wide()) {                 // the 128-bit kernels read one byte per base
    rc = mfx_seq_ensure_ascii(seq);
    if (rc) return rc;
  }
  if (!(seq->planes_ok || seq->bases_stale) && !seq->d_bases) {   // (a sequence nothing was put into: every byte 0)
    rc = mfx_seq_ensure_ascii(seq);
    if (rc) return rc;
  }
  mfx_hist_args a;
  a.t = ev->ix->view();
  a.canonical = canon;
  a.bases = seq->d_bases;
  if (seq->planes_ok || seq->bases_stale) { a.codes = seq->d_codes; a.valid = seq->d_valid; }      // packed planes present: the tiles are copied, not encoded
  a.contig_off = seq->d_contig_off;
  a.contig_len = seq->d_contig_len;
  a.tile_start = seq->d_tile_start;
  a.ncontigs = seq->ncontigs;
  a.tile_begin = tile_begin;
  a.tile_end = tile_end;
  a.tile_contig = seq->d_tile_contig;
  a.tile_ctr = ev->d_tile_ctr + ctr_slot;
  a.tile_partials = ev->d_tile_partials + (chunk_of_total ? (tile_begin - partials_tile0) * (MFX_BLOCK / 64) : 0);   // partials_tile0: first tile of a streamed PART
  a.n_logical = ntl;
  a.part_rank = part_rank;
  a.part_n = part_n;
  a.part_shift = part_shift;
  a.ks.peak = ev->peak;
  a.ks.n_prob = ev->n_prob;
  a.ks.probK = ev->d_probK;
  a.ks.probP = ev->d_probP;
  a.ks.underq = ev->d_underq;
  a.ks.nbins = ev->nbins;
  a.ks.ncontigs = seq->ncontigs;
  a.ks.counts = d_counts;
  a.ks.partials = ev->d_partials;
  a.ks.ovf = ev->d_ovf;
  a.dbg = ev->d_dbg;
  if (a.t.compact && canon && ntl < (1ull << 27)) {          // the probe's rare endings are listed and ended by mfx_hist_rest_kernel (mfx_kernels.hip)
    // (a streamed run's chunks grow to 128 MB of bases: its lists are made for that size at its first chunk, not re-made as they grow)
    rc = ensure_worklist(ev, ctr_slot, chunk_of_total ? std::max<uint64_t>(ntl, std::min<uint64_t>(chunk_of_total, (128ull << 20) / MFX_TILE)) : ntl);
    if (rc) return rc;
    const uint64_t segs = std::min<uint64_t>((uint64_t)ev->grid, ntl);     // the main kernel's grid: one segment per block
    if (ev->d_wl[ctr_slot] && segs <= MFX_WL_HEADER - 2) {
      a.wl = ev->d_wl[ctr_slot];
      a.wl_segs = (uint32_t)segs;
      a.wl_segcap = (uint32_t)(ev->wl_cap[ctr_slot] / segs);
    }
  }
  MFX_HIP(ev->ix->wide() ? mfx_kw_hist(a, (int)std::min<uint64_t>((uint64_t)ev->grid, ntl), (hipStream_t)stream)
                          : mfx_k_hist(a, (int)std::min<uint64_t>((uint64_t)ev->grid, ntl), (hipStream_t)stream));
  if (a.wl) MFX_HIP(mfx_k_hist_rest(a, (int)(a.wl_segs * 4u), (hipStream_t)stream));                 // (MFX_REST_SPLIT blocks per segment)
  if (chunk_of_total) MFX_HIP(hipMemsetAsync(ev->d_tile_ctr + ctr_slot, 0, sizeof(uint64_t), (hipStream_t)stream));   // re-arm the tile scheduler
  else MFX_HIP(mfx_k_sum_tile_partials(ev->d_tile_partials, ntl, d_kover, ev->d_tile_ctr, (hipStream_t)stream, ev->ix->wide() ? 0 : 1));
  return MFX_OK;
}

extern "C" int mfx_hist_launch(mfx_eval *ev, const mfx_seq *seq, uint64_t tile_begin, uint64_t tile_end,
                               uint64_t *d_counts, double *d_kover, void *stream) {
  if (!ev || !seq || !d_counts || !d_kover) return mfx_fail(MFX_E_INVAL, "mfx_hist_launch: null argument");
  if (ev->device != seq->device) return mfx_fail(MFX_E_INVAL, "evaluator and sequence live on different devices");
  if (tile_begin > tile_end || tile_end > seq->ntiles) return mfx_fail(MFX_E_INVAL, "tile range [%lu,%lu) outside [0,%lu)",
                                                                      (unsigned long)tile_begin, (unsigned long)tile_end, (unsigned long)seq->ntiles);
  return hist_launch(ev, seq, tile_begin, tile_end, 0, 1, 0, d_counts, d_kover, stream);
}

extern "C" int mfx_hist_launch_cyclic(mfx_eval *ev, const mfx_seq *seq, uint32_t rank, uint32_t nranks, uint32_t block_tiles,
                                      uint64_t *d_counts, double *d_kover, void *stream) {
  if (!ev || !seq || !d_counts || !d_kover) return mfx_fail(MFX_E_INVAL, "mfx_hist_launch_cyclic: null argument");
  if (ev->device != seq->device) return mfx_fail(MFX_E_INVAL, "evaluator and sequence live on different devices");
  if (nranks == 0 || rank >= nranks || block_tiles == 0 || (block_tiles & (block_tiles - 1)))
    return mfx_fail(MFX_E_INVAL, "mfx_hist_launch_cyclic: rank %u of %u, block of %u tiles (a power of two)", rank, nranks, block_tiles);
  uint32_t shift = 0;
  while ((1u << shift) < block_tiles) ++shift;
  if (nranks == 1) return hist_launch(ev, seq, 0, seq->ntiles, 0, 1, 0, d_counts, d_kover, stream);
  return hist_launch(ev, seq, 0, seq->ntiles, rank, nranks, shift, d_counts, d_kover, stream);
}

// The evaluator's table of far K* bins emptied for the next launches, asynchronously on `st`: what an earlier launch left there
// and nobody collected is not the next run's.  16 MB of fills: ~10 us of device time.
static hipError_t ovf_reset_hip(mfx_eval *ev, hipStream_t st) {
  const hipError_t e = hipMemsetAsync(ev->d_ovf, 0, (2 + (size_t)MFX_OVF_SLOTS) * sizeof(uint64_t), st);
  return e != hipSuccess ? e : hipMemsetAsync(ev->d_ovf + 2 + MFX_OVF_SLOTS, 0xff, (size_t)MFX_OVF_SLOTS * sizeof(uint64_t), st);
}
int mfx_ovf_reset_async(mfx_eval *ev, hipStream_t st) {
  MFX_HIP(ovf_reset_hip(ev, st));
  return MFX_OK;
}

// the table's occupied slots as {key, occurrences} pairs in `pairs` (room for 2 * n words); the table is emptied.  Synchronises `st`.
int mfx_ovf_collect(mfx_eval *ev, uint64_t *header2, std::vector<uint64_t> *pairs, hipStream_t st) {
  uint64_t hd[2] = {0, 0};
  MFX_HIP(hipMemcpyAsync(hd, ev->d_ovf, sizeof(hd), hipMemcpyDeviceToHost, st));
  MFX_HIP(hipStreamSynchronize(st));
  if (header2) { header2[0] = hd[0]; header2[1] = hd[1]; }
  if (pairs) pairs->clear();
  if (hd[0] == 0 && hd[1] == 0) return MFX_OK;
  if (pairs && hd[0]) {
    // rare: the whole table comes over (16 MB) and is scanned here
    std::vector<uint64_t> tab(2 * (size_t)MFX_OVF_SLOTS);
    MFX_HIP(hipMemcpyAsync(tab.data(), ev->d_ovf + 2, tab.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    MFX_HIP(hipStreamSynchronize(st));
    pairs->reserve(2 * hd[0]);
    for (size_t i = 0; i < MFX_OVF_SLOTS; ++i)
      if (tab[MFX_OVF_SLOTS + i] != ~0ull) { pairs->push_back(tab[MFX_OVF_SLOTS + i]); pairs->push_back(tab[i]); }
    std::vector<std::pair<uint64_t, uint64_t>> srt(pairs->size() / 2);          // by key: the same list whatever the races of the inserts were
    for (size_t i = 0; i < srt.size(); ++i) srt[i] = {(*pairs)[2 * i], (*pairs)[2 * i + 1]};
    std::sort(srt.begin(), srt.end());
    for (size_t i = 0; i < srt.size(); ++i) { (*pairs)[2 * i] = srt[i].first; (*pairs)[2 * i + 1] = srt[i].second; }
  }
  if (int rc = mfx_ovf_reset_async(ev, st)) return rc;
  MFX_HIP(hipStreamSynchronize(st));
  if (hd[1])
    return mfx_fail(MFX_E_OVERFLOW, "%lu k-mers fell into more distinct K* bins beyond the %u dense ones than the evaluator's table of far bins "
                    "holds (%u); create the evaluator with a larger nbins", (unsigned long)hd[1], ev->nbins, MFX_OVF_SLOTS);
  return MFX_OK;
}

extern "C" int mfx_hist_take_overflow(mfx_eval *ev, uint64_t *records, uint64_t cap, uint64_t *n_out) {
  if (!ev || !n_out) return mfx_fail(MFX_E_INVAL, "mfx_hist_take_overflow: null argument");
  DevGuard g(ev->device);
  std::vector<uint64_t> pairs;
  uint64_t hd[2];
  int rc = mfx_ovf_collect(ev, hd, &pairs, nullptr);
  *n_out = pairs.size() / 2;
  if (rc) return rc;
  const uint64_t m = std::min<uint64_t>(pairs.size() / 2, cap);
  if (m && records) memcpy(records, pairs.data(), 2 * m * sizeof(uint64_t));
  return (pairs.size() / 2 > cap) ? mfx_fail(MFX_E_OVERFLOW, "%lu distinct far K* bins, caller buffer holds %lu", (unsigned long)(pairs.size() / 2), (unsigned long)cap) : MFX_OK;
}

// The reference's arrays grow in steps of 1024 under a uint32 bound (increaseArray(..., histOverMax, 1024),
// merfin-histogram.C:74,87,116,121): a bin index above 2^32 - 1025 overflows that bound there (undefined behaviour);
// here it is an error, as is running out of host memory (an index of 4e9 is 34 GB of bins, there as here).
static int result_grow(uint64_t *&a, uint32_t &max, uint64_t need) {
  if (need <= max) return MFX_OK;
  const uint64_t nm = (need + 1023) / 1024 * 1024;
  if (nm > 0xffffffffull)
    return mfx_fail(MFX_E_INVAL, "K* histogram bin %lu is beyond the 32-bit array bound of merfin's histogram (merfin-histogram.C:74,87)",
                    (unsigned long)(need - 1));
  uint64_t *b = (uint64_t *)realloc(a, nm * sizeof(uint64_t));
  if (!b) return mfx_fail(MFX_E_NOMEM, "no host memory for %lu K* histogram bins", (unsigned long)nm);
  a = b;
  memset(a + max, 0, (nm - max) * sizeof(uint64_t));
  max = (uint32_t)nm;
  return MFX_OK;
}

extern "C" int mfx_hist_result_from_counts(uint32_t nbins, const uint64_t *h, double kover, uint32_t ncontigs,
                                           mfx_hist_result *out) {
  if (!nbins || !h || !out) return mfx_fail(MFX_E_INVAL, "mfx_hist_result_from_counts: null argument");
  memset(out, 0, sizeof(*out));
  const uint32_t nb = nbins;
  uint32_t um = 0, om = 0;
  for (uint32_t i = 0; i < nb; ++i) {
    if (h[i]) um = i + 1;
    if (h[nb + i]) om = i + 1;
  }
  // merfin-histogram.C:105-108: the global arrays start at 2048 entries
  if (result_grow(out->undr, out->undrMax, std::max<uint32_t>(um, 2048)) || result_grow(out->over, out->overMax, std::max<uint32_t>(om, 2048))) {
    free(out->undr);
    free(out->over);
    memset(out, 0, sizeof(*out));
    return mfx_last_error_code();
  }
  memcpy(out->undr, h, um * sizeof(uint64_t));
  memcpy(out->over, h + nb, om * sizeof(uint64_t));
  out->kasm = h[2ull * nb + 0];
  out->kmissing = h[2ull * nb + 1];
  out->koverCpy = kover;
  out->ncontigs = ncontigs;
  out->contig_kasm = (uint64_t *)calloc(ncontigs ? ncontigs : 1, sizeof(uint64_t));
  out->contig_kmissing = (uint64_t *)calloc(ncontigs ? ncontigs : 1, sizeof(uint64_t));
  memcpy(out->contig_kasm, h + 2ull * nb + 3, ncontigs * sizeof(uint64_t));
  memcpy(out->contig_kmissing, h + 2ull * nb + 3 + ncontigs, ncontigs * sizeof(uint64_t));
  return MFX_OK;
}

// rec: {key, occurrences} pairs (mfx_hist_take_overflow's format)
static int result_add_overflow(mfx_hist_result *r, const std::vector<uint64_t> &rec) {
  uint64_t mu = 0, mo = 0;                                   // grow once, to the largest index of the batch
  for (size_t i = 0; i + 1 < rec.size(); i += 2) {
    const uint64_t x = rec[i], idx = x & ~(1ull << 63);
    if (x >> 63) mo = std::max(mo, idx + 1); else mu = std::max(mu, idx + 1);
  }
  if (int rc = result_grow(r->over, r->overMax, mo)) return rc;
  if (int rc = result_grow(r->undr, r->undrMax, mu)) return rc;
  for (size_t i = 0; i + 1 < rec.size(); i += 2) {
    const uint64_t x = rec[i], idx = x & ~(1ull << 63);
    if (x >> 63) r->over[idx] += rec[i + 1]; else r->undr[idx] += rec[i + 1];
  }
  return MFX_OK;
}

static int result_take_overflow(mfx_eval *ev, uint64_t novf, mfx_hist_result *out);

extern "C" int mfx_hist_result_add_overflow(mfx_hist_result *r, const uint64_t *records, uint64_t n) {
  if (!r || !r->undr || !r->over || (n && !records)) return mfx_fail(MFX_E_INVAL, "mfx_hist_result_add_overflow: null argument");
  return result_add_overflow(r, std::vector<uint64_t>(records, records + 2 * n));
}

// What a whole-assembly run needs besides the evaluator's tables -- a stream, the counts image, its pinned mirror -- belongs
// to the evaluator and is made once (mfx_eval::sr): creating and releasing it per call costs ~2.5 ms, which is most of an
// 8-device evaluation (4 ms of kernel per device at 3 Gb).  The device of `ev` must be current.
static int eval_run_resources(mfx_eval *ev, size_t words) {
  auto &R = ev->sr;
  if (R.words < words) {
    if (R.d_counts) (void)hipFree(R.d_counts);
    if (R.h_img) (void)hipHostFree(R.h_img);
    R.d_counts = nullptr; R.h_img = nullptr; R.words = 0;
    MFX_HIP(hipMalloc((void **)&R.d_counts, words * sizeof(uint64_t)));
    MFX_HIP(hipHostMalloc((void **)&R.h_img, (words + 2) * sizeof(uint64_t), hipHostMallocDefault));
    R.words = words;
  }
  if (!R.d_kover) MFX_HIP(hipMalloc((void **)&R.d_kover, 2 * sizeof(double)));          // ([1]: hist_run_streamed_packed's digest word)
  for (auto &k : R.kern) if (!k) MFX_HIP(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
  return MFX_OK;
}

// clears + launch (contiguous tiles, or the block-cyclic share part_rank of part_n) + D2H of image and koverCpy, all
// asynchronous on the evaluator's own stream; the caller synchronises ev->sr.kern[0] and reads ev->sr.h_img
static int eval_run_enqueue(mfx_eval *ev, const mfx_seq *seq, uint32_t part_rank, uint32_t part_n) {
  const size_t words = MFX_HIST_WORDS(ev->nbins, seq->ncontigs);
  int rc = eval_run_resources(ev, words);
  if (rc) return rc;
  auto &R = ev->sr;
  hipStream_t st = R.kern[0];
  MFX_HIP(hipMemsetAsync(R.d_counts, 0, words * sizeof(uint64_t), st));
  MFX_HIP(hipMemsetAsync(R.d_kover, 0, sizeof(double), st));
  if (int rc0 = mfx_ovf_reset_async(ev, st)) return rc0;             // far bins nobody collected from an earlier launch are not this run's
  rc = part_n > 1 ? mfx_hist_launch_cyclic(ev, seq, part_rank, part_n, 256, R.d_counts, R.d_kover, st)
                  : mfx_hist_launch(ev, seq, 0, seq->ntiles, R.d_counts, R.d_kover, st);
  if (rc) return rc;
  MFX_HIP(hipMemcpyAsync(R.h_img, R.d_counts, words * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MFX_HIP(hipMemcpyAsync(R.h_img + words, R.d_kover, sizeof(double), hipMemcpyDeviceToHost, st));
  return MFX_OK;
}

extern "C" int mfx_hist_run(mfx_eval *ev, const mfx_seq *seq, mfx_hist_result *out) {
  if (!ev || !seq || !out) return mfx_fail(MFX_E_INVAL, "mfx_hist_run: null argument");
  if (ev->device != seq->device) return mfx_fail(MFX_E_INVAL, "evaluator and sequence live on different devices");
  DevGuard g(ev->device);
  const size_t words = MFX_HIST_WORDS(ev->nbins, seq->ncontigs);
  int rc = eval_run_enqueue(ev, seq, 0, 1);
  if (rc) return rc;
  MFX_HIP(hipStreamSynchronize(ev->sr.kern[0]));
  double kover = 0;
  memcpy(&kover, ev->sr.h_img + words, sizeof(double));
  const uint64_t novf = ev->sr.h_img[2ull * ev->nbins + 2];
  rc = mfx_hist_result_from_counts(ev->nbins, ev->sr.h_img, kover, seq->ncontigs, out);
  if (rc) return rc;
  rc = result_take_overflow(ev, novf, out);
  if (rc) mfx_hist_result_free(out);
  return rc;
}

// fold the overflow list of `ev` (K* bins >= nbins) into a result
static int result_take_overflow(mfx_eval *ev, uint64_t novf, mfx_hist_result *out) {
  if (!novf) return MFX_OK;
  std::vector<uint64_t> pairs;
  if (int rc = mfx_ovf_collect(ev, nullptr, &pairs, nullptr)) return rc;
  return result_add_overflow(out, pairs);
}

// ---------------------------------------------------------------------------
// Streamed -hist: the assembly arrives as HOST buffers (what loadSequence hands over, merfin.C:30-53) and its
// upload is overlapped with the evaluation -- SURVEY 8(d)'s timed region "first tile H2D start -> final reduced
// histogram on host".  The tiles are cut into chunks; chunk i's bytes travel on a copy stream while the -hist
// kernel of chunk i-1 runs on the compute stream (one event per chunk orders them), and the counts image comes
// back with one D2H copy at the end.  Buffers from mfx_host_alloc (pinned) are DMA'd in place; pageable buffers
// go through two pinned staging buffers filled by host threads.
// ---------------------------------------------------------------------------
extern "C" void *mfx_host_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    mfx_fail(MFX_E_NOMEM, "mfx_host_alloc: cannot pin %zu bytes of host memory", bytes);
    return nullptr;
  }
  return p;
}

extern "C" void mfx_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

extern "C" mfx_seq *mfx_seq_create(int device, const uint64_t *lens, uint32_t ncontigs) {
  if ((ncontigs && !lens) || device < 0 || device >= mfx_device_count()) {
    mfx_fail(device < 0 || device >= mfx_device_count() ? MFX_E_NODEVICE : MFX_E_INVAL,
             "mfx_seq_create: bad argument (device %d of %d)", device, mfx_device_count());
    return nullptr;
  }
  DevGuard g(device);
  mfx_seq *s = seq_layout(device, lens, ncontigs);
  if (seq_alloc(s, false) != MFX_OK) { mfx_seq_free(s); return nullptr; }      // what fills it decides which form it holds
  return s;
}

static bool host_ptr_is_pinned(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

namespace {
struct Piece { uint32_t contig; uint64_t pos, n; };      // bases [pos, pos+n) of a contig

// the bases the tiles [t0, t1) read: their own positions plus the (k-1)-base halo behind the last tile of a contig
// piece (a whole 128-byte line of it: the next chunk re-sends those bytes, same values)
void chunk_pieces(const mfx_seq *s, uint64_t t0, uint64_t t1, std::vector<Piece> &out) {
  out.clear();
  uint32_t c = (uint32_t)(std::upper_bound(s->tile_start.begin(), s->tile_start.end(), t0) - s->tile_start.begin()) - 1;
  for (; c < s->ncontigs && s->tile_start[c] < t1; ++c) {
    const uint64_t cb = std::max(t0, s->tile_start[c]), ce = std::min(t1, s->tile_start[c + 1]);
    if (ce <= cb) continue;                                 // an empty contig owns no tile
    const uint64_t p0 = (cb - s->tile_start[c]) * MFX_TILE;
    const uint64_t p1 = std::min<uint64_t>(s->len[c], (ce - s->tile_start[c]) * MFX_TILE + MFX_ALIGN);
    if (p1 > p0) out.push_back({c, p0, p1 - p0});
  }
}
}  // namespace

extern "C" void mfx_pack_bases(const uint8_t *src, uint64_t n, uint64_t *codes, uint32_t *valid);      // mfx_pack.cpp

// The CPUs of the NUMA node that holds `addr` (two-socket hosts: an encoder thread on the other socket reads the assembly over
// the inter-socket links).  false: unknown (no such call in this container, one node, too few of its CPUs allowed) -- no binding.
static bool cpus_near(const void *addr, cpu_set_t *out, int *node_out = nullptr) {
  const char *e = getenv("MFX_NUMA_BIND");
  if (e && atoi(e) == 0) return false;
  int node = -1;
  if (syscall(SYS_get_mempolicy, &node, nullptr, 0UL, const_cast<void *>(addr), 3UL /* MPOL_F_NODE | MPOL_F_ADDR */) != 0 || node < 0) return false;
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE *f = fopen(path, "r");
  if (!f) return false;
  char buf[4096];
  const bool got = fgets(buf, sizeof(buf), f) != nullptr;
  fclose(f);
  if (!got) return false;
  cpu_set_t allowed, near;
  CPU_ZERO(&near);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
  for (char *p = buf; *p;) {                                  // "0-63,128-191"
    char *q;
    const long a = strtol(p, &q, 10);
    if (q == p) break;
    long b = a;
    if (*q == '-') { p = q + 1; b = strtol(p, &q, 10); }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) CPU_SET(c, &near);
    p = *q == ',' ? q + 1 : q;
    if (*q != ',' ) break;
  }
  if (node_out) *node_out = node;
  if ((unsigned)CPU_COUNT(&near) < std::max(1u, mfx_host_threads())) return false;
  if (CPU_COUNT(&near) == CPU_COUNT(&allowed)) return false;  // one node: nothing to choose
  *out = near;
  return true;
}

// Where the W encoder threads of a streamed run go.  A core reads memory through its CCD's link to the I/O die, and a few
// threads saturate one link (EPYC 9575F, tools/native/pack_bench.cpp: 8 threads in one CCD encode 55 GB/s of bases, 8 threads in 8
// CCDs 130-147 GB/s), so WHERE the scheduler happens to put the threads decides the rate of the whole pipeline: 16 threads that
// share two CCDs encode 99 GB/s, spread over the CCDs 145-193 GB/s.  MFX_PACK_PLACE:
//   spread (default) worker w is pinned to L3 domain w mod n of the source's NUMA node (every node if that is unknown), on the
//                    first hardware thread of its cores;   node  the CPUs of the source's node (the round-3 behaviour);
//   all              L3 domains of every node;   os  no binding
struct PackTopology {
  std::vector<std::vector<int>> l3_of_node[8];        // [node][domain] -> primary hardware threads
  cpu_set_t allowed;
  bool ok = false;
  PackTopology() {
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    auto read_int = [](const char *p) { int v = -1; if (FILE *f = fopen(p, "r")) { if (fscanf(f, "%d", &v) != 1) v = -1; fclose(f); } return v; };
    std::map<std::pair<int, int>, std::vector<int>> dom;
    for (int c = 0; c < CPU_SETSIZE; ++c) {
      if (!CPU_ISSET(c, &allowed)) continue;
      char p[160];
      snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
      if (read_int(p) != c) continue;                  // a second hardware thread of its core
      snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/cache/index3/id", c);
      const int l3 = read_int(p);
      int node = -1;
      for (int nd = 0; nd < 8 && node < 0; ++nd) {
        snprintf(p, sizeof(p), "/sys/devices/system/cpu/cpu%d/node%d", c, nd);
        if (access(p, F_OK) == 0) node = nd;
      }
      if (l3 < 0) continue;
      dom[{node < 0 ? 0 : node, l3}].push_back(c);
    }
    for (auto &kv : dom) l3_of_node[kv.first.first].push_back(kv.second);
    size_t n = 0;
    for (auto &v : l3_of_node) n += v.size();
    ok = n > 1;
  }
};
static const PackTopology &pack_topology() { static PackTopology t; return t; }

void mfx_pin_spread(unsigned w) {
  const char *e = getenv("MFX_POOL_SPREAD");
  if (e && atoi(e) == 0) return;
  cpu_set_t mine;
  const PackTopology &T = pack_topology();
  if (!T.ok) return;
  std::vector<const std::vector<int> *> doms;
  for (auto &nd : T.l3_of_node) for (auto &d : nd) doms.push_back(&d);
  if (doms.empty()) return;
  CPU_ZERO(&mine);
  for (int c : *doms[w % doms.size()]) CPU_SET(c, &mine);
  if (CPU_COUNT(&mine) > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof(mine), &mine);
}

// the CPU set worker w of W runs on under `mode` (0 os, 1 node, 2 spread, 3 all); false: leave the thread where it is allowed
static bool pack_cpus(int mode, int node, const cpu_set_t *near, unsigned w, cpu_set_t *out) {
  const PackTopology &T = pack_topology();
  if (mode == 0) { *out = T.allowed; return CPU_COUNT(&T.allowed) > 0; }
  if (mode == 1) { if (near) { *out = *near; return true; } *out = T.allowed; return CPU_COUNT(&T.allowed) > 0; }
  if (!T.ok) { *out = T.allowed; return CPU_COUNT(&T.allowed) > 0; }
  std::vector<const std::vector<int> *> doms;
  if (mode == 2 && node >= 0 && node < 8) for (auto &d : T.l3_of_node[node]) doms.push_back(&d);
  if (doms.empty()) for (auto &nd : T.l3_of_node) for (auto &d : nd) doms.push_back(&d);
  CPU_ZERO(out);
  for (int c : *doms[w % doms.size()]) CPU_SET(c, out);
  return CPU_COUNT(out) > 0;
}

// The streamed -hist with the assembly crossing PCIe PACKED (0.375 B per base): host threads encode every chunk into
// the tile form (2-bit codes + validity bits, csrc/mfx_pack.cpp) while the previous chunk is on the bus and the one
// before is being evaluated; the kernel reads its tiles from the packed planes (mfx_tile_fill_packed).  One byte per
// base never reaches the device (seq->bases_stale; unpacked on demand by mfx_seq_ensure_ascii).
// part != nullptr: only the tiles [part->tl, part->th) are encoded, uploaded and evaluated (one device's share of a run over several:
// mfx_hist_run_streamed_multi / mfx_hist_run_streamed_range); the sequence object then holds that part only (mfx_seq::partial).
// The two rates that decide how a streamed run's bases cross the link (hist_run_streamed_packed): what W host threads ENCODE (GB of
// bases per second, all of them at once: they share the memory system) and what the device's LINK moves from pinned host memory.
// Measured once per process and (W | device) on the caller's own buffers -- every thread encodes 4 MB of the first contig into a scratch of
// its own, 32 MB of it are copied to the device buffer they go to anyway -- ~1.5 ms together; MFX_STREAM_ENC_GBS / MFX_STREAM_LINK_GBS override.
static double stream_encode_gbs(WorkerPool *pool, unsigned W, const uint8_t *src, uint64_t n) {
  static std::mutex mu;
  static std::map<unsigned, double> cache;
  if (const char *e = getenv("MFX_STREAM_ENC_GBS")) if (atof(e) > 0) return atof(e);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(W);
    if (it != cache.end()) return it->second;
  }
  const uint64_t per = std::min<uint64_t>(4u << 20, n / std::max(1u, W) / 32 * 32);
  if (!pool || per < (1u << 16)) return 0.0;                  // too little to measure: no estimate
  std::vector<std::vector<uint64_t>> sc(W);
  for (auto &v : sc) v.assign(per / 32 + per / 64 + 2, 1);      // (written: the scratch's pages exist before the clock starts)
  // ONE pass over bases nobody has read yet (the tail of the contig: a second pass over the same 4 MB per thread would be served by the L3)
  const uint8_t *from = src + (n - (uint64_t)W * per) / 128 * 128;
  auto run = [&](unsigned w) {
    uint64_t *codes = sc[w].data();
    mfx_pack_bases(from + (uint64_t)w * per, per, codes, reinterpret_cast<uint32_t *>(codes + per / 32 + 1));
  };
  const auto t0 = std::chrono::steady_clock::now();
  pool->start(run); pool->wait();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double gbs = dt > 0 ? (double)per * W / dt / 1e9 : 0.0;
  std::lock_guard<std::mutex> lk(mu);
  cache[W] = gbs;
  return gbs;
}
static double stream_link_gbs(int device, uint8_t *d_dst, const uint8_t *pinned_src, uint64_t n, hipStream_t st) {
  static std::mutex mu;
  static std::map<int, double> cache;
  if (const char *e = getenv("MFX_STREAM_LINK_GBS")) if (atof(e) > 0) return atof(e);
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(device);
    if (it != cache.end()) return it->second;
  }
  const uint64_t m = std::min<uint64_t>(32u << 20, n);
  if (m < (4u << 20)) return 0.0;
  hipEvent_t a = nullptr, b = nullptr;
  double gbs = 0.0;
  if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess &&
      hipMemcpyAsync(d_dst, pinned_src, 1u << 20, hipMemcpyHostToDevice, st) == hipSuccess &&          // (wakes the link up)
      hipEventRecord(a, st) == hipSuccess && hipMemcpyAsync(d_dst, pinned_src, m, hipMemcpyHostToDevice, st) == hipSuccess &&
      hipEventRecord(b, st) == hipSuccess && hipEventSynchronize(b) == hipSuccess) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0) gbs = (double)m / (ms * 1e-3) / 1e9;
  } else (void)hipGetLastError();
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (gbs > 0) { std::lock_guard<std::mutex> lk(mu); cache[device] = gbs; }
  return gbs;
}

struct StreamPart {
  uint64_t tl = 0, th = 0;
  uint64_t *d_counts = nullptr;          // caller's device image / koverCpy to accumulate into (cleared by the caller); null: the evaluator's own, returned on the host
  double   *d_kover = nullptr;
  std::vector<double> *chunk_sums = nullptr;   // the first-level koverCpy sums of the part (4096 (tile, wave) values each), for a bit-stable sum over the parts
  bool      want_result = true;          // false: the image stays in ev->sr.h_img (or on the device), no mfx_hist_result is made
};
static int hist_run_streamed_packed(mfx_eval *ev, mfx_seq *seq, const char *const *bases, mfx_hist_result *out, const StreamPart *part = nullptr) {
  DevGuard g(ev->device);
  const int timing = getenv("MFX_STREAM_TIMING") ? atoi(getenv("MFX_STREAM_TIMING")) : 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_mark[6] = {0, 0, 0, 0, 0, 0};
  // MFX_STREAM_TIMING=2: the timeline of every chunk (host clocks of the packers and of the enqueue loop, device events around
  // its copy and behind its launch) -- what tools/stream_phases.py prints; costs a few events, never on by default
  struct ChunkTimes { double pack_first = 0, pack_last = 0, wait_buf = 0, enq = 0; hipEvent_t c0 = nullptr, c1 = nullptr, k1 = nullptr; };
  std::vector<ChunkTimes> ct;
  hipEvent_t ev_base = nullptr;
  const size_t words = MFX_HIST_WORDS(ev->nbins, seq->ncontigs);
  const uint64_t TL = part ? part->tl : 0, TH = part ? part->th : seq->ntiles;
  const uint64_t T = TH - TL;                                 // tiles of this run
  const bool whole = TL == 0 && TH == seq->ntiles;
  // chunks grow from 8 MB to 128 MB of bases: the first tile reaches the kernel after ~0.2 ms, and the bulk runs in few,
  // long launches (every launch pays a ramp-up and a tail of its persistent blocks: 47 launches of 64 MB cost 38 ms of
  // kernel time for 3 Gb, one launch 33.4 ms)
  const uint64_t CH0 = 2048, CH = 32768;
  constexpr int NB = 3;
  const uint64_t plane_words = seq->buf_bytes / 32 + (MFX_TILE + 64) / 32 + 1;   // a tile reads 130 words from its first one
  struct Chunk { std::vector<Piece> pieces; uint64_t lo = 0, hi = 0, t0 = 0, t1 = 0; };
  std::vector<Chunk> chunks;
  // ... and shrink again at the end (64, 32, 16 MB): what follows the last upload is the evaluation of the last chunk alone
  std::vector<uint64_t> cuts;                                 // chunk boundaries, ascending
  {
    std::vector<uint64_t> tail;                               // sizes of the closing chunks, last first
    uint64_t back = 0;
    for (uint64_t sz = 2 * CH0; sz < CH && back + sz + CH <= T / 2; sz *= 2) { tail.push_back(sz); back += sz; }
    const uint64_t front_end = T - back;
    uint64_t t0 = 0;
    for (uint64_t sz = CH0; t0 < front_end; sz = std::min(CH, sz * 2)) {
      uint64_t t1 = std::min(front_end, t0 + sz);
      if (front_end - t1 < sz / 2) t1 = front_end;            // no short launch in the middle (at most 1.5 x CH tiles)
      cuts.push_back(t1);
      t0 = t1;
    }
    for (size_t i = tail.size(); i-- > 0;) { t0 += tail[i]; cuts.push_back(t0); }
  }
  for (uint64_t t0 = 0, ci = 0; ci < cuts.size(); ++ci) {
    Chunk c;
    c.t0 = TL + t0;
    c.t1 = TL + cuts[ci];
    t0 = cuts[ci];
    chunk_pieces(seq, c.t0, c.t1, c.pieces);
    if (!c.pieces.empty()) {
      c.lo = seq->off[c.pieces.front().contig] + c.pieces.front().pos;                 // a multiple of 128
      c.hi = (seq->off[c.pieces.back().contig] + c.pieces.back().pos + c.pieces.back().n + 31) / 32 * 32;
    }
    chunks.push_back(std::move(c));
  }
  size_t STAGE_W = 0;                                         // words of the largest chunk, rounded up to 1 MB of them
  for (const Chunk &c : chunks) STAGE_W = std::max<size_t>(STAGE_W, (c.hi - c.lo) / 32);
  STAGE_W = (STAGE_W + 2 + 131071) / 131072 * 131072;
  std::atomic<int64_t> allowed{NB - 1};                     // chunks <= allowed may be packed (their staging buffer is free)
  std::atomic<bool> stop{false};
  std::vector<std::atomic<uint32_t>> done(chunks.size());
  for (auto &d : done) d.store(0);
  int rc = MFX_OK;
  auto &R = ev->sr;
  std::mutex ct_mu;
  if (timing >= 2) {
    ct.resize(chunks.size());
    (void)hipEventCreate(&ev_base);
    for (auto &x : ct) { (void)hipEventCreate(&x.c0); (void)hipEventCreate(&x.c1); (void)hipEventCreate(&x.k1); }
  }
  auto cleanup = [&]() {
    stop.store(true);
    if (ev->pool) static_cast<WorkerPool *>(ev->pool)->wait();
    if (R.copy) (void)hipStreamSynchronize(R.copy);
    for (auto &k : R.kern) if (k) (void)hipStreamSynchronize(k);
  };
#define STREAMED_HIP(call)                                                                                        \
  do {                                                                                                            \
    hipError_t e_ = (call);                                                                                       \
    if (e_ != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); cleanup(); return rc; } \
  } while (0)
  if (!seq->d_codes) {
    STREAMED_HIP(hipMalloc((void **)&seq->d_codes, plane_words * sizeof(uint64_t)));
    STREAMED_HIP(hipMalloc((void **)&seq->d_valid, plane_words * sizeof(uint32_t)));
    STREAMED_HIP(mfx_memset_now(seq->d_valid, 0, plane_words * sizeof(uint32_t)));           // no valid base outside the uploads
    STREAMED_HIP(mfx_memset_now(seq->d_codes, 0, plane_words * sizeof(uint64_t)));
  }
  // pinned staging (codes words, then validity words) belongs to the evaluator: pinning costs more than the evaluation
  if (ev->h_pack_words != STAGE_W) {
    for (auto &p : ev->h_pack) { if (p) (void)hipHostFree(p); p = nullptr; }
    ev->h_pack_words = STAGE_W;
  }
  // The validity plane crosses the link SPARSE: a word of it is all ones wherever 32 consecutive bases are ACGT, so a chunk sends
  // its code words and the few validity words that are not (contig ends, gaps, N runs); the plane's range is filled with ones on the
  // device and those are put in (mfx_k_valid_scatter): 0.25 instead of 0.375 bytes per base over PCIe, the bound of the streamed run.
  // Every packer lists the exceptional words of its share behind the staging buffer (room for 1/8 of its words: a chunk with more goes whole).
  const char *spv = getenv("MFX_STREAM_SPARSE_VALID");
  const bool sparse_valid = !(spv && atoi(spv) == 0);
  const size_t EXC_TOTAL = STAGE_W / 8 + 64 * 64;               // entries behind a staging buffer (up to 64 packers)
  for (int b = 0; b < NB && b < (int)chunks.size(); ++b)
    if (!ev->h_pack[b]) STREAMED_HIP(hipHostMalloc((void **)&ev->h_pack[b], STAGE_W * 12 + EXC_TOTAL * 8, hipHostMallocDefault));
  if (sparse_valid && R.exc_cap < EXC_TOTAL) {
    for (auto &x : R.d_exc) { if (x) (void)hipFree(x); x = nullptr; }
    for (auto &x : R.d_exc) STREAMED_HIP(hipMalloc((void **)&x, EXC_TOTAL * 8));
    R.exc_cap = EXC_TOTAL;
  }
  std::vector<uint32_t> nexc(chunks.size() * 64, 0);           // exceptional words listed by packer w of chunk ci: nexc[ci * 64 + w]
  std::vector<std::atomic<uint32_t>> dense(chunks.size());     // a packer ran out of room: the chunk's validity words go whole
  for (auto &d : dense) d.store(0);
  seq->bases_stale = true;
  seq->planes_ok = true;
  seq->digest = 0;                                           // new content
  seq->partial = !whole;
  seq->part_lo = TL;
  seq->part_hi = TH;

  // the packers: worker w encodes its share of the words of every chunk, in chunk order
  const unsigned W = std::max(1u, std::min(mfx_host_threads(), 64u));
  if (ev->pool && static_cast<WorkerPool *>(ev->pool)->W != W) { delete static_cast<WorkerPool *>(ev->pool); ev->pool = nullptr; }
  if (!ev->pool) ev->pool = new WorkerPool(W);
  uint8_t *const *stage = ev->h_pack;
  // TRANSPORT.  Host-packed (0.25-0.375 bytes per base over the link, the encoders' work in front of it) is the faster way while the
  // host threads THIS call has encode faster than its link moves plain bytes -- one process with the host's cores to itself: 150-190 GB/s
  // against 56.  A rank of N processes has 1/N of the cores but a link of its own: from N = 4 on (16-core quota) its threads are slower than
  // its link, and the run of the whole node is bound by the host's encoders whatever N.  Then the bases cross as they
  // are -- DMA straight out of the caller's pinned buffers, no host thread touches them -- and mfx_pack_kernel makes the planes on the
  // device (3 GB of bytes: 1.5 ms of its HBM); the evaluation is the same launches over the same planes, bit for bit.  Pinned sources
  // only (pageable ones would need a host copy as dear as the encoding).  MFX_STREAM_TRANSPORT=pack | ascii forces either.
  bool link_ascii = false;
  {
    bool all_pinned = seq->ncontigs > 0;
    for (uint32_t c = 0; c < seq->ncontigs && all_pinned; ++c) all_pinned = seq->len[c] == 0 || host_ptr_is_pinned(bases[c]);
    const char *tp = getenv("MFX_STREAM_TRANSPORT");
    if (tp && !strcmp(tp, "ascii")) link_ascii = all_pinned;
    else if (tp && !strcmp(tp, "pack")) link_ascii = false;
    else if (all_pinned && seq->ncontigs && seq->len[0] >= (32u << 20)) {
      if (int brc = seq_need_bases(seq)) return brc;
      if (!R.copy) MFX_HIP(hipStreamCreateWithFlags(&R.copy, hipStreamNonBlocking));
      const double enc = stream_encode_gbs(static_cast<WorkerPool *>(ev->pool), W, reinterpret_cast<const uint8_t *>(bases[0]), seq->len[0]);
      const double link = stream_link_gbs(ev->device, seq->d_bases, reinterpret_cast<const uint8_t *>(bases[0]), seq->len[0], R.copy);
      link_ascii = enc > 0 && link > 0 && enc < link;
      if (timing) fprintf(stderr, "[mfx stream] transport: %u host threads encode %.1f GB/s, the link moves %.1f GB/s -> %s\n", W, enc, link, link_ascii ? "plain bytes + device-side packing" : "host-packed planes");
    }
    if (link_ascii) if (int brc = seq_need_bases(seq)) return brc;
  }
  cpu_set_t near_cpus;
  int src_node = -1;
  const bool bind = seq->ncontigs && bases[0] && cpus_near(bases[0], &near_cpus, &src_node);      // the encoders run next to the memory they read
  int place = 2;
  if (const char *pp = getenv("MFX_PACK_PLACE")) place = !strcmp(pp, "os") ? 0 : !strcmp(pp, "node") ? 1 : !strcmp(pp, "all") ? 3 : 2;
  if (const char *nb = getenv("MFX_NUMA_BIND")) if (atoi(nb) == 0) place = 0;
  auto work = [&, W](unsigned w) {
    if (link_ascii) return;                                      // the bases cross the link as they are
    cpu_set_t mine;
    if (pack_cpus(place, src_node, bind ? &near_cpus : nullptr, w, &mine)) (void)pthread_setaffinity_np(pthread_self(), sizeof(mine), &mine);
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      while (allowed.load(std::memory_order_acquire) < (int64_t)ci) {
        if (stop.load()) return;
        std::this_thread::yield();
      }
      const Chunk &c = chunks[ci];
      const uint64_t nw = (c.hi - c.lo) / 32, w0 = nw * w / W, w1 = nw * (w + 1) / W;
      uint64_t *codes = reinterpret_cast<uint64_t *>(stage[ci % NB]);
      uint32_t *valid = reinterpret_cast<uint32_t *>(stage[ci % NB] + (size_t)STAGE_W * 8);
      const double t_p0 = timing >= 2 ? now() : 0.0;
      if (w1 > w0) {
        memset(codes + w0, 0, (w1 - w0) * 8);                // gaps between contigs and the words behind a contig's end
        memset(valid + w0, 0, (w1 - w0) * 4);
        const uint64_t my_lo = c.lo + 32 * w0, my_hi = c.lo + 32 * w1;
        for (const Piece &pc : c.pieces) {
          const uint64_t at = seq->off[pc.contig] + pc.pos;      // a multiple of 128
          const uint64_t s = std::max(my_lo, at), e = std::min(my_hi, at + pc.n);
          if (e > s) mfx_pack_bases(reinterpret_cast<const uint8_t *>(bases[pc.contig]) + pc.pos + (s - at), e - s, codes + (s - c.lo) / 32, valid + (s - c.lo) / 32);
        }
        if (sparse_valid) {
          const size_t cap_w = STAGE_W / (8 * (size_t)W) + 64;
          uint64_t *mine_exc = reinterpret_cast<uint64_t *>(stage[ci % NB] + (size_t)STAGE_W * 12) + (size_t)w * cap_w;
          uint32_t cnt = 0;
          for (uint64_t i = w0; i < w1; ++i) {
            const uint32_t vw = valid[i];
            if (vw == 0xffffffffu) continue;
            if (cnt == cap_w) { dense[ci].store(1, std::memory_order_relaxed); break; }
            mine_exc[cnt++] = (uint64_t)i | ((uint64_t)vw << 32);
          }
          nexc[ci * 64 + w] = cnt;
        }
      }
      if (timing >= 2) {
        const double t_p1 = now();
        std::lock_guard<std::mutex> lk(ct_mu);
        if (ct[ci].pack_first == 0 || t_p0 < ct[ci].pack_first) ct[ci].pack_first = t_p0;
        if (t_p1 > ct[ci].pack_last) ct[ci].pack_last = t_p1;
      }
      done[ci].fetch_add(1, std::memory_order_release);
    }
  };
  t_mark[0] = now();
  static_cast<WorkerPool *>(ev->pool)->start(work);
  t_mark[1] = now();
  // streams, events, the counts image and its pinned mirror: created once per evaluator (while the packers start)
  if (R.words < words) {
    if (R.d_counts) (void)hipFree(R.d_counts);
    if (R.h_img) (void)hipHostFree(R.h_img);
    R.d_counts = nullptr; R.h_img = nullptr; R.words = 0;
    STREAMED_HIP(hipMalloc((void **)&R.d_counts, words * sizeof(uint64_t)));
    STREAMED_HIP(hipHostMalloc((void **)&R.h_img, (words + 2) * sizeof(uint64_t), hipHostMallocDefault));
    R.words = words;
  }
  if (!R.d_kover) STREAMED_HIP(hipMalloc((void **)&R.d_kover, 2 * sizeof(double)));    // [1]: the uploaded sequence's content digest (a uint64)
  if (!R.copy) STREAMED_HIP(hipStreamCreateWithFlags(&R.copy, hipStreamNonBlocking));
  for (auto &k : R.kern) if (!k) STREAMED_HIP(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
  for (auto &e : R.up) if (!e) STREAMED_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  if (!R.kdone) STREAMED_HIP(hipEventCreateWithFlags(&R.kdone, hipEventDisableTiming));
  hipStream_t cs = R.copy, ks = R.kern[0];
  const bool own_image = !(part && part->d_counts);
  uint64_t *const d_counts = own_image ? R.d_counts : part->d_counts, *const h_img = R.h_img;
  double *const d_kover = own_image ? R.d_kover : part->d_kover;
  hipEvent_t *const up = R.up;
  if (own_image) {
    STREAMED_HIP(hipMemsetAsync(d_counts, 0, words * sizeof(uint64_t), ks));
    STREAMED_HIP(hipMemsetAsync(d_kover, 0, sizeof(double), ks));
  }
  STREAMED_HIP(hipMemsetAsync(ev->d_tile_ctr, 0, 2 * sizeof(uint64_t), ks));
  STREAMED_HIP(ovf_reset_hip(ev, ks));
  STREAMED_HIP(hipEventRecord(R.kdone, ks));
  STREAMED_HIP(hipStreamWaitEvent(R.kern[1], R.kdone, 0));          // the second kernel stream starts behind the clears
  if ((rc = ensure_tile_partials(ev, T)) != MFX_OK) { cleanup(); return rc; }
  double t_base_host = 0;
  if (timing >= 2) { (void)hipEventRecord(ev_base, cs); (void)hipEventSynchronize(ev_base); t_base_host = now(); }

  for (size_t ci = 0; ci < chunks.size(); ++ci) {
    const Chunk &c = chunks[ci];
    const int b = (int)(ci % NB);
    if (ci >= 1) {
      // the copy of chunk ci-1 has left its buffer: chunk ci-1+NB may be packed into it
      STREAMED_HIP(hipEventSynchronize(up[(ci - 1) % NB]));
      allowed.store((int64_t)(ci - 1 + NB), std::memory_order_release);
    }
    if (timing >= 2) ct[ci].wait_buf = now();
    if (!link_ascii) while (done[ci].load(std::memory_order_acquire) < W) std::this_thread::yield();
    if (timing >= 2) { ct[ci].enq = now(); (void)hipEventRecord(ct[ci].c0, cs); }
    if (link_ascii && c.hi > c.lo) {
      // the chunk's pieces by DMA out of the caller's pinned buffers (the gaps between contigs stay the buffer's zero bytes), then the
      // planes of its words on the device, behind the copies on the same stream; a word shared with the chunk before is rewritten with
      // the value it has (that chunk's kernel may still read it)
      for (const Piece &pc : c.pieces)
        STREAMED_HIP(hipMemcpyAsync(seq->d_bases + seq->off[pc.contig] + pc.pos, bases[pc.contig] + pc.pos, pc.n, hipMemcpyHostToDevice, cs));
      STREAMED_HIP(mfx_k_pack(seq->d_bases + c.lo, seq->d_codes + c.lo / 32, seq->d_valid + c.lo / 32, (c.hi - c.lo) / 32, cs));
    } else if (c.hi > c.lo) {
      const uint64_t nw = (c.hi - c.lo) / 32;
      STREAMED_HIP(hipMemcpyAsync(seq->d_codes + c.lo / 32, stage[b], nw * 8, hipMemcpyHostToDevice, cs));
      // the sparse form pays (8 bytes per listed word against 4 per word sent whole) only while fewer than half of the chunk's
      // validity words are exceptional; a chunk of short or gappy contigs goes whole
      size_t n_exc = 0;
      for (unsigned w = 0; w < W; ++w) n_exc += nexc[ci * 64 + w];
      if (sparse_valid && !dense[ci].load(std::memory_order_relaxed) && 2 * n_exc < nw) {
        // the packers' lists, one behind the other, at the head of the (now free) validity words of the staging buffer
        const size_t cap_w = STAGE_W / (8 * (size_t)W) + 64;
        uint64_t *all = reinterpret_cast<uint64_t *>(stage[b] + (size_t)STAGE_W * 8);
        const uint64_t *lists = reinterpret_cast<const uint64_t *>(stage[b] + (size_t)STAGE_W * 12);
        size_t n = 0;
        for (unsigned w = 0; w < W; ++w) {
          const uint32_t m_ = nexc[ci * 64 + w];
          if (m_) memcpy(all + n, lists + (size_t)w * cap_w, (size_t)m_ * 8);
          n += m_;
        }
        // A chunk that starts inside a contig begins with the MFX_ALIGN bases the previous chunk already sent as the halo of its last
        // tile, and that chunk's kernel may still be reading them (the copy stream does not wait for it): those words are left out of
        // the fill -- they hold their final values, and the listed ones among them are only rewritten with the same value.  Filling
        // them with ones first would let the running kernel see an N or a contig end behind the cut as valid bases for a moment.
        const uint64_t keep = (ci > 0 && c.pieces.front().pos) ? std::min<uint64_t>(nw, MFX_ALIGN / 32) : 0;   // (the first chunk of a part has nobody before it)
        if (nw > keep) STREAMED_HIP(hipMemsetAsync(seq->d_valid + c.lo / 32 + keep, 0xff, (nw - keep) * 4, cs));
        if (n) {
          STREAMED_HIP(hipMemcpyAsync(R.d_exc[b], all, n * 8, hipMemcpyHostToDevice, cs));
          STREAMED_HIP(mfx_k_valid_scatter(seq->d_valid + c.lo / 32, R.d_exc[b], (uint32_t)n, cs));
        }
      } else {
        STREAMED_HIP(hipMemcpyAsync(seq->d_valid + c.lo / 32, stage[b] + (size_t)STAGE_W * 8, nw * 4, hipMemcpyHostToDevice, cs));
      }
    }
    if (timing >= 2) (void)hipEventRecord(ct[ci].c1, cs);
    STREAMED_HIP(hipEventRecord(up[b], cs));
    // two kernel streams (and tile counters) alternate: the first blocks of a launch fill the CUs the previous launch's
    // last blocks leave behind, instead of waiting for its tail
    hipStream_t kst = R.kern[ci & 1];
    STREAMED_HIP(hipStreamWaitEvent(kst, up[b], 0));
    rc = hist_launch(ev, seq, c.t0, c.t1, 0, 1, 0, d_counts, d_kover, kst, T, (int)(ci & 1), TL);
    if (rc) { cleanup(); return rc; }
    if (timing >= 2) (void)hipEventRecord(ct[ci].k1, kst);
  }
  t_mark[2] = now();
  // a sequence-only index answers for the k-mers of ONE sequence: the content digest of what was just uploaded is taken on
  // the copy stream, behind the last chunk and under the last launches, and compared before the result is handed out
  // (a part cannot be checked: its digest is not the sequence's; the caller of the parts vouches for the sequence)
  const bool want_digest = whole && ev->ix->seq_only && ev->ix->seq_digest != 0;
  if (want_digest) {
    uint64_t *d_dig = reinterpret_cast<uint64_t *>(R.d_kover + 1);      // the evaluator's own word (eval_run_resources), never the caller's one-double buffer of a range run
    STREAMED_HIP(hipMemsetAsync(d_dig, 0, sizeof(uint64_t), cs));
    STREAMED_HIP(mfx_k_seq_digest(nullptr, seq->d_codes, seq->d_valid, seq->buf_bytes / 32, d_dig, cs));
    STREAMED_HIP(hipMemcpyAsync(h_img + words + 1, d_dig, sizeof(uint64_t), hipMemcpyDeviceToHost, cs));
  }
  STREAMED_HIP(hipEventRecord(R.kdone, R.kern[1]));
  STREAMED_HIP(hipStreamWaitEvent(ks, R.kdone, 0));
  if (T) STREAMED_HIP(mfx_k_sum_tile_partials(ev->d_tile_partials, T, d_kover, ev->d_tile_ctr, ks, ev->ix->wide() ? 0 : 1));
  if (own_image) {
    STREAMED_HIP(hipMemcpyAsync(h_img, d_counts, words * sizeof(uint64_t), hipMemcpyDeviceToHost, ks));
    STREAMED_HIP(hipMemcpyAsync(h_img + words, d_kover, sizeof(double), hipMemcpyDeviceToHost, ks));
  }
  STREAMED_HIP(hipStreamSynchronize(ks));
  if (want_digest) STREAMED_HIP(hipStreamSynchronize(cs));
  if (part && part->chunk_sums) {
    // the first-level sums behind the (tile, wave) values: what mfx_sum_partials_kernel added up; the caller adds the parts' in the same order
    const uint64_t nv = T * (MFX_BLOCK / 64), nch = (nv + 4095) / 4096;
    part->chunk_sums->assign(nch, 0.0);
    if (nch) STREAMED_HIP(hipMemcpy(part->chunk_sums->data(), ev->d_tile_partials + nv, nch * sizeof(double), hipMemcpyDeviceToHost));
  }
  t_mark[3] = now();
#undef STREAMED_HIP
  if (want_digest) {
    seq_digest_finish(seq, h_img[words + 1]);
    if ((rc = mfx_check_seq_of_index(ev->ix, seq, "-hist (streamed)")) != MFX_OK) { cleanup(); return rc; }
  }
  double kover;
  memcpy(&kover, h_img + words, sizeof(double));
  const bool want_result = !part || (part->want_result && own_image);
  if (want_result) rc = mfx_hist_result_from_counts(ev->nbins, h_img, kover, seq->ncontigs, out);
  const uint64_t novf = own_image ? h_img[2ull * ev->nbins + 2] : 0;
  t_mark[4] = now();
  cleanup();
  t_mark[5] = now();
  if (timing)
    fprintf(stderr, "[mfx stream] %zu chunks: setup %.2f ms, spawn %.2f, enqueue loop %.2f, drain %.2f, result %.2f, cleanup %.2f; total %.2f ms%s\n", chunks.size(),
            (t_mark[0] - t_begin) * 1e3, (t_mark[1] - t_mark[0]) * 1e3, (t_mark[2] - t_mark[1]) * 1e3, (t_mark[3] - t_mark[2]) * 1e3,
            (t_mark[4] - t_mark[3]) * 1e3, (t_mark[5] - t_mark[4]) * 1e3, (t_mark[5] - t_begin) * 1e3, place == 0 ? "; encoders unbound" : place == 1 ? "; encoders on the source's NUMA node" : place == 2 ? "; encoders spread over the L3 domains of the source's node" : "; encoders spread over all L3 domains");
  if (timing >= 2) {
    // all times in ms since the function was entered; device times are placed on the host clock through ev_base
    fprintf(stderr, "[mfx stream] chunk   Mbases  pack:first..last   buf-wait   enqueue   copy:start..end   kernel-end   (ms since entry; %u packer threads)\n", W);
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      float a = 0, b2 = 0, k2 = 0;
      (void)hipEventElapsedTime(&a, ev_base, ct[ci].c0);
      (void)hipEventElapsedTime(&b2, ev_base, ct[ci].c1);
      (void)hipEventElapsedTime(&k2, ev_base, ct[ci].k1);
      const double off = (t_base_host - t_begin) * 1e3;
      fprintf(stderr, "[mfx stream] %5zu %8.1f %8.2f %8.2f %10.2f %9.2f %9.2f %8.2f %11.2f\n", ci, (chunks[ci].hi - chunks[ci].lo) / 1e6,
              (ct[ci].pack_first - t_begin) * 1e3, (ct[ci].pack_last - t_begin) * 1e3, (ct[ci].wait_buf - t_begin) * 1e3, (ct[ci].enq - t_begin) * 1e3,
              off + a, off + b2, off + k2);
    }
    for (auto &x : ct) { (void)hipEventDestroy(x.c0); (void)hipEventDestroy(x.c1); (void)hipEventDestroy(x.k1); }
    (void)hipEventDestroy(ev_base);
  }
  if (rc || !want_result) return rc;
  rc = result_take_overflow(ev, novf, out);
  if (rc) mfx_hist_result_free(out);
  return rc;
}

// mfx_seq_upload's transport: the same encoder and chunking as the streamed -hist, without the evaluation -- three small
// pinned buffers (32 M bases = 12 MB each: pinning costs 0.4 ms per MB, the old 2 x 64 MB of bytes cost 50 ms before
// the first base moved), the host threads encode chunk i+1 and i+2 while chunk i is on the bus.
static int seq_upload_packed(mfx_seq *seq, const char *const *bases) {
  const bool timing = getenv("MFX_UPLOAD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double tm[5] = {now(), 0, 0, 0, 0};
  const uint64_t T = seq->ntiles, CH = 8192;                   // tiles per chunk
  constexpr int NB = 3;
  struct Chunk { std::vector<Piece> pieces; uint64_t lo = 0, hi = 0; };
  std::vector<Chunk> chunks;
  for (uint64_t t0 = 0; t0 < T; t0 += CH) {
    Chunk c;
    chunk_pieces(seq, t0, std::min(T, t0 + CH), c.pieces);
    if (c.pieces.empty()) continue;
    c.lo = seq->off[c.pieces.front().contig] + c.pieces.front().pos;                 // a multiple of 128
    c.hi = (seq->off[c.pieces.back().contig] + c.pieces.back().pos + c.pieces.back().n + 31) / 32 * 32;
    chunks.push_back(std::move(c));
  }
  tm[1] = now();
  int rc = seq_alloc_planes(seq);
  if (rc) return rc;
  seq->bases_stale = true;
  seq->planes_ok = true;
  seq->digest = 0;
  seq->partial = false;
  if (chunks.empty()) return MFX_OK;
  tm[2] = now();
  size_t STAGE_W = 0;
  for (const Chunk &c : chunks) STAGE_W = std::max<size_t>(STAGE_W, (c.hi - c.lo) / 32);
  STAGE_W = (STAGE_W + 2 + 4095) / 4096 * 4096;
  uint8_t *stage[NB] = {nullptr, nullptr, nullptr};
  hipEvent_t up[NB] = {nullptr, nullptr, nullptr};
  hipStream_t cs = nullptr;
  std::atomic<int64_t> allowed{NB - 1};
  std::atomic<bool> stop{false};
  std::vector<std::atomic<uint32_t>> done(chunks.size());
  for (auto &d : done) d.store(0);
  const unsigned W = std::max(1u, std::min(mfx_host_threads(), 64u));
  std::unique_ptr<WorkerPool> pool;
  bool ok = hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) == hipSuccess;
  for (int b = 0; b < NB && b < (int)chunks.size() && ok; ++b)
    ok = hipHostMalloc((void **)&stage[b], STAGE_W * 12, hipHostMallocDefault) == hipSuccess &&
         hipEventCreateWithFlags(&up[b], hipEventDisableTiming) == hipSuccess;
  auto work = [&, W](unsigned w) {
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      while (allowed.load(std::memory_order_acquire) < (int64_t)ci) {
        if (stop.load()) return;
        std::this_thread::yield();
      }
      const Chunk &c = chunks[ci];
      const uint64_t nw = (c.hi - c.lo) / 32, w0 = nw * w / W, w1 = nw * (w + 1) / W;
      uint64_t *codes = reinterpret_cast<uint64_t *>(stage[ci % NB]);
      uint32_t *valid = reinterpret_cast<uint32_t *>(stage[ci % NB] + (size_t)STAGE_W * 8);
      if (w1 > w0) {
        memset(codes + w0, 0, (w1 - w0) * 8);                // gaps between contigs and the words behind a contig's end
        memset(valid + w0, 0, (w1 - w0) * 4);
        const uint64_t my_lo = c.lo + 32 * w0, my_hi = c.lo + 32 * w1;
        for (const Piece &pc : c.pieces) {
          const uint64_t at = seq->off[pc.contig] + pc.pos;      // a multiple of 128
          const uint64_t s = std::max(my_lo, at), e = std::min(my_hi, at + pc.n);
          if (e > s) mfx_pack_bases(reinterpret_cast<const uint8_t *>(bases[pc.contig]) + pc.pos + (s - at), e - s, codes + (s - c.lo) / 32, valid + (s - c.lo) / 32);
        }
      }
      done[ci].fetch_add(1, std::memory_order_release);
    }
  };
  tm[3] = now();
  if (ok) {
    // (not spread over the L3 domains: 3 Gb are encoded + copied in 0.024-0.034 s as the threads fall, 0.034-0.070 s pinned; what an upload costs
    // is its device buffers -- 0.05-0.15 s for 3 GB: profiles/r04_cli_startup_1gb.txt)
    pool.reset(new WorkerPool(W));
    pool->start(work);
    for (size_t ci = 0; ci < chunks.size() && ok; ++ci) {
      const Chunk &c = chunks[ci];
      const int b = (int)(ci % NB);
      if (ci >= 1) {                                          // the copy of chunk ci-1 has left its buffer: chunk ci-1+NB may be packed into it
        ok = hipEventSynchronize(up[(ci - 1) % NB]) == hipSuccess;
        allowed.store((int64_t)(ci - 1 + NB), std::memory_order_release);
      }
      while (done[ci].load(std::memory_order_acquire) < W) std::this_thread::yield();
      const uint64_t nw = (c.hi - c.lo) / 32;
      ok = ok && hipMemcpyAsync(seq->d_codes + c.lo / 32, stage[b], nw * 8, hipMemcpyHostToDevice, cs) == hipSuccess &&
           hipMemcpyAsync(seq->d_valid + c.lo / 32, stage[b] + (size_t)STAGE_W * 8, nw * 4, hipMemcpyHostToDevice, cs) == hipSuccess &&
           hipEventRecord(up[b], cs) == hipSuccess;
    }
    stop.store(!ok);
    if (!ok) allowed.store((int64_t)chunks.size());
    pool->wait();
    if (hipStreamSynchronize(cs) != hipSuccess) ok = false;
  }
  tm[4] = now();
  for (int b = 0; b < NB; ++b) { if (stage[b]) (void)hipHostFree(stage[b]); if (up[b]) (void)hipEventDestroy(up[b]); }
  if (cs) (void)hipStreamDestroy(cs);
  if (timing)
    fprintf(stderr, "-- packed upload: chunk plan %.3f s, planes %.3f, pinned staging + stream %.3f, encode + copy %.3f (%u threads, %zu chunks), release %.3f\n",
            tm[1] - tm[0], tm[2] - tm[1], tm[3] - tm[2], tm[4] - tm[3], W, chunks.size(), now() - tm[4]);
  return ok ? MFX_OK : mfx_fail(MFX_E_HIP, "packed upload of the sequence failed: %s", hipGetErrorString(hipGetLastError()));
}

// The two rates of the transport decision, measured on demand (bench.py's model of SURVEY 8(d)'s metric at N GPUs: a rank of N has
// host_threads / N encoders and a link of its own): `threads` host threads encoding `src` (pinned or not) at once, and -- src pinned --
// 32 MB of it over this device's link.  Nothing is cached.
extern "C" int mfx_diag_stream_rates(int device, const char *src, uint64_t n, uint32_t threads, double *enc_gbs, double *link_gbs) {
  if (!src || !enc_gbs || !link_gbs || threads == 0 || threads > 256) return mfx_fail(MFX_E_INVAL, "mfx_diag_stream_rates: bad argument");
  if (device < 0 || device >= mfx_device_count()) return mfx_fail(MFX_E_NODEVICE, "mfx_diag_stream_rates: device %d of %d", device, mfx_device_count());
  *enc_gbs = *link_gbs = 0.0;
  {
    WorkerPool pool(threads, true);                              // (spread over the L3 domains, as the run's own encoders are)
    // every pass reads bases no pass before it has read (a second pass over the same bytes would be served by the L3: 256 MB on the EPYC
    // of the measured boxes): up to three passes of `threads` x 32 MB, as far as n reaches; the best one counts
    const uint64_t per = std::min<uint64_t>(32u << 20, n / threads / 128 * 128);
    if (per >= (1u << 16)) {
      std::vector<std::vector<uint64_t>> sc(threads);
      for (auto &v : sc) v.assign(per / 32 + per / 64 + 2, 1);  // (written: the scratch's pages exist before the clock starts)
      const uint64_t passes = std::max<uint64_t>(1, std::min<uint64_t>(3, n / (per * threads)));
      double best = 0;
      for (uint64_t rep = 0; rep < passes; ++rep) {
        const uint8_t *from = reinterpret_cast<const uint8_t *>(src) + rep * per * threads;
        auto run = [&](unsigned w) { mfx_pack_bases(from + (uint64_t)w * per, per, sc[w].data(), reinterpret_cast<uint32_t *>(sc[w].data() + per / 32 + 1)); };
        const auto t0 = std::chrono::steady_clock::now();
        pool.start(run); pool.wait();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dt > 0) best = std::max(best, (double)per * threads / dt / 1e9);
      }
      *enc_gbs = best;
    }
  }
  if (host_ptr_is_pinned(src) && n >= (4u << 20)) {
    DevGuard g(device);
    const uint64_t m = std::min<uint64_t>(256u << 20, n);
    uint8_t *d = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipMalloc((void **)&d, m) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&a) == hipSuccess &&
        hipEventCreate(&b) == hipSuccess && hipMemcpyAsync(d, src, 1u << 20, hipMemcpyHostToDevice, st) == hipSuccess) {
      for (int rep = 0; rep < 2; ++rep) {
        float ms = 0;
        if (hipEventRecord(a, st) == hipSuccess && hipMemcpyAsync(d, src, m, hipMemcpyHostToDevice, st) == hipSuccess && hipEventRecord(b, st) == hipSuccess &&
            hipEventSynchronize(b) == hipSuccess && hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0)
          *link_gbs = std::max(*link_gbs, (double)m / (ms * 1e-3) / 1e9);
      }
    } else (void)hipGetLastError();
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (st) (void)hipStreamDestroy(st);
    if (d) (void)hipFree(d);
  }
  return MFX_OK;
}

extern "C" int mfx_hist_run_streamed(mfx_eval *ev, mfx_seq *seq, const char *const *bases, mfx_hist_result *out) {
  if (!ev || !seq || !out || (seq->ncontigs && !bases)) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed: null argument");
  if (ev->device != seq->device) return mfx_fail(MFX_E_INVAL, "evaluator and sequence live on different devices");
  {
    // default: the assembly crosses the bus packed; MFX_STREAM_ASCII=1 (and the 128-bit k-mer kernels) send one byte per base
    const char *asc = getenv("MFX_STREAM_ASCII");
    if (!ev->ix->wide() && !(asc && atoi(asc))) return hist_run_streamed_packed(ev, seq, bases, out);
  }
  DevGuard g(ev->device);
  if (int brc = seq_need_bases(seq)) return brc;
  seq->bases_stale = false;                                 // this path writes d_bases
  seq->planes_ok = false;
  seq->digest = 0;
  seq->partial = false;
  const size_t words = MFX_HIST_WORDS(ev->nbins, seq->ncontigs);
  const uint64_t T = seq->ntiles;
  const uint64_t CH = 16384;                                // tiles per chunk: 64 MB of bases
  const size_t STAGE = (size_t)CH * (MFX_TILE + 2 * MFX_ALIGN) + MFX_TILE;   // worst case: every tile its own contig
  DevBuf<uint64_t> dc;
  DevBuf<double> dk;
  uint64_t *h_img = nullptr;                                // pinned: counts image + koverCpy
  // the pinned staging buffers (pageable sources only) belong to the evaluator: pinning 2 x 71 MB costs ~60 ms, more
  // than the whole evaluation of 3 Gb, so it is paid once per evaluator, not per call
  if (ev->h_stage_bytes != STAGE) {
    for (auto &p : ev->h_stage) { if (p) (void)hipHostFree(p); p = nullptr; }
    ev->h_stage_bytes = STAGE;
  }
  uint8_t **stage = ev->h_stage;
  hipStream_t cs = nullptr, ks = nullptr;
  hipEvent_t up[2] = {nullptr, nullptr}, staged[2] = {nullptr, nullptr};
  int rc = MFX_OK;
  auto cleanup = [&]() {
    if (cs) (void)hipStreamSynchronize(cs);
    if (ks) (void)hipStreamSynchronize(ks);
    for (int i = 0; i < 2; ++i) {
      if (up[i]) (void)hipEventDestroy(up[i]);
      if (staged[i]) (void)hipEventDestroy(staged[i]);
    }
    if (h_img) (void)hipHostFree(h_img);
    if (cs) (void)hipStreamDestroy(cs);
    if (ks) (void)hipStreamDestroy(ks);
  };
#define STREAMED_HIP(call)                                                                                        \
  do {                                                                                                            \
    hipError_t e_ = (call);                                                                                       \
    if (e_ != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); cleanup(); return rc; } \
  } while (0)
  STREAMED_HIP(dc.alloc(words));
  STREAMED_HIP(dk.alloc(1));
  STREAMED_HIP(hipHostMalloc((void **)&h_img, (words + 1) * sizeof(uint64_t), hipHostMallocDefault));
  STREAMED_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  STREAMED_HIP(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    STREAMED_HIP(hipEventCreateWithFlags(&up[i], hipEventDisableTiming));
    STREAMED_HIP(hipEventCreateWithFlags(&staged[i], hipEventDisableTiming));
  }
  STREAMED_HIP(hipMemsetAsync(dc.p, 0, words * sizeof(uint64_t), ks));
  STREAMED_HIP(hipMemsetAsync(dk.p, 0, sizeof(double), ks));
  STREAMED_HIP(hipMemsetAsync(ev->d_tile_ctr, 0, sizeof(uint64_t), ks));
  STREAMED_HIP(ovf_reset_hip(ev, ks));
  if ((rc = ensure_tile_partials(ev, T)) != MFX_OK) { cleanup(); return rc; }
  std::vector<char> pinned(seq->ncontigs);
  for (uint32_t c = 0; c < seq->ncontigs; ++c) pinned[c] = seq->len[c] && host_ptr_is_pinned(bases[c]);
  std::vector<Piece> pieces;
  bool stage_busy[2] = {false, false};
  for (uint64_t t0 = 0, ci = 0; t0 < T; t0 += CH, ++ci) {
    const uint64_t t1 = std::min(T, t0 + CH);
    const int b = (int)(ci & 1);
    chunk_pieces(seq, t0, t1, pieces);
    bool direct = pieces.size() <= 64;                       // many small contigs: one assembled copy instead of many tiny ones
    for (const Piece &pc : pieces) direct = direct && pinned[pc.contig];
    if (direct) {
      for (const Piece &pc : pieces)
        STREAMED_HIP(hipMemcpyAsync(seq->d_bases + seq->off[pc.contig] + pc.pos, bases[pc.contig] + pc.pos, pc.n, hipMemcpyHostToDevice, cs));
    } else if (!pieces.empty()) {
      if (!stage[b]) STREAMED_HIP(hipHostMalloc((void **)&stage[b], STAGE, hipHostMallocDefault));
      if (stage_busy[b]) STREAMED_HIP(hipEventSynchronize(staged[b]));   // its previous copy has left the buffer
      // the staging buffer mirrors the device range [lo, hi) of this chunk; gaps between contigs stay non-ACGT
      const uint64_t lo = seq->off[pieces.front().contig] + pieces.front().pos;
      const uint64_t hi = seq->off[pieces.back().contig] + pieces.back().pos + pieces.back().n;
      if (hi - lo > STAGE) { rc = mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed: chunk of %lu bytes exceeds the staging buffer", (unsigned long)(hi - lo)); cleanup(); return rc; }
      uint64_t filled = lo;
      for (const Piece &pc : pieces) {
        const uint64_t at = seq->off[pc.contig] + pc.pos;
        if (at > filled) memset(stage[b] + (filled - lo), 0, at - filled);
        par_memcpy(stage[b] + (at - lo), bases[pc.contig] + pc.pos, pc.n);
        filled = at + pc.n;
      }
      STREAMED_HIP(hipMemcpyAsync(seq->d_bases + lo, stage[b], hi - lo, hipMemcpyHostToDevice, cs));
      STREAMED_HIP(hipEventRecord(staged[b], cs));
      stage_busy[b] = true;
    }
    STREAMED_HIP(hipEventRecord(up[b], cs));
    STREAMED_HIP(hipStreamWaitEvent(ks, up[b], 0));
    rc = hist_launch(ev, seq, t0, t1, 0, 1, 0, dc.p, dk.p, ks, T);
    if (rc) { cleanup(); return rc; }
  }
  if (T) STREAMED_HIP(mfx_k_sum_tile_partials(ev->d_tile_partials, T, dk.p, ev->d_tile_ctr, ks, ev->ix->wide() ? 0 : 1));
  STREAMED_HIP(hipMemcpyAsync(h_img, dc.p, words * sizeof(uint64_t), hipMemcpyDeviceToHost, ks));
  STREAMED_HIP(hipMemcpyAsync(h_img + words, dk.p, sizeof(double), hipMemcpyDeviceToHost, ks));
  STREAMED_HIP(hipStreamSynchronize(ks));
#undef STREAMED_HIP
  double kover;
  memcpy(&kover, h_img + words, sizeof(double));
  if ((rc = mfx_check_seq_of_index(ev->ix, seq, "-hist (streamed)")) != MFX_OK) { cleanup(); return rc; }
  rc = mfx_hist_result_from_counts(ev->nbins, h_img, kover, seq->ncontigs, out);
  const uint64_t novf = h_img[2ull * ev->nbins + 2];
  cleanup();
  if (rc) return rc;
  rc = result_take_overflow(ev, novf, out);
  if (rc) mfx_hist_result_free(out);
  return rc;
}

// ---------------------------------------------------------------------------
// Several devices, ONE process (the reference is one binary driving all its workers, merfin.C:366-414): the index
// is replicated, device d evaluates the block-cyclic share d of N of the tiles (mfx_hist_launch_cyclic), and the
// per-device counts images (~1 MB each) are added on the host in device order -- integers exactly, koverCpy as
// a fixed-order fp64 sum, so the result is bit-stable run to run.  K* bins beyond the dense image travel in
// every evaluator's overflow list and are folded in as well.
// ---------------------------------------------------------------------------
// Doubling tree over the devices of a node: holder 0 has the data; in every round each holder feeds ONE device that does
// not have it yet, all copies of a round in flight together (xGMI is point to point: 1, 2, 4 source devices work in
// parallel, 7 replicas of a 97 GB table take 3 rounds instead of 7 copies through device 0's links one after the other).
// dev[h], ptr[h][s]: device and buffers of holder h; bytes[s]: size of segment s (the same for every holder).
static int tree_copy(const std::vector<int> &dev, const std::vector<std::vector<void *>> &ptr, const std::vector<size_t> &bytes) {
  const size_t H = dev.size();
  std::vector<hipStream_t> st(H, nullptr);
  int rc = MFX_OK;
  std::vector<size_t> have(1, 0);
  size_t next = 1;
  const size_t CH = 1ull << 30;
  while (next < H && rc == MFX_OK) {
    std::vector<size_t> fresh;
    for (size_t j = 0; j < have.size() && next < H && rc == MFX_OK; ++j, ++next) {
      const size_t from = have[j], to = next;
      DevGuard g(dev[to]);
      if (dev[from] != dev[to]) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dev[to], dev[from]) == hipSuccess && can && hipDeviceEnablePeerAccess(dev[from], 0) != hipSuccess)
          (void)hipGetLastError();                              // "already enabled" is fine
      }
      hipError_t e = st[to] ? hipSuccess : hipStreamCreateWithFlags(&st[to], hipStreamNonBlocking);
      for (size_t sgi = 0; sgi < bytes.size() && e == hipSuccess; ++sgi)
        for (size_t o = 0; o < bytes[sgi] && e == hipSuccess; o += CH)
          e = hipMemcpyPeerAsync((char *)ptr[to][sgi] + o, dev[to], (const char *)ptr[from][sgi] + o, dev[from], std::min(CH, bytes[sgi] - o), st[to]);
      if (e != hipSuccess) rc = mfx_fail(MFX_E_HIP, "peer copy from device %d to device %d failed: %s", dev[from], dev[to], hipGetErrorString(e));
      fresh.push_back(to);
    }
    for (size_t t : fresh) {
      DevGuard g(dev[t]);
      const hipError_t e = st[t] ? hipStreamSynchronize(st[t]) : hipSuccess;
      if (e != hipSuccess && rc == MFX_OK) rc = mfx_fail(MFX_E_HIP, "peer copy to device %d failed: %s", dev[t], hipGetErrorString(e));
      have.push_back(t);
    }
  }
  for (size_t h = 0; h < H; ++h)
    if (st[h]) { DevGuard g(dev[h]); (void)hipStreamDestroy(st[h]); }
  return rc;
}

extern "C" int mfx_index_replicate_many(const mfx_index *src, const int *devices, uint32_t n, mfx_index **out) {
  if (!src || !devices || !out || n == 0) return mfx_fail(MFX_E_INVAL, "mfx_index_replicate_many: null argument");
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = nullptr;
    if (devices[i] < 0 || devices[i] >= mfx_device_count()) return mfx_fail(MFX_E_INVAL, "mfx_index_replicate_many: device %d of %d", devices[i], mfx_device_count());
  }
  uint8_t hdr[MFX_INDEX_HEADER_BYTES];
  int rc = mfx_index_image_header(src, hdr);
  if (rc) return rc;
  std::vector<int> dev(1, src->device);
  std::vector<std::vector<void *>> ptr(1, std::vector<void *>{src->d_slots, src->d_meta});
  for (uint32_t i = 0; i < n && rc == MFX_OK; ++i) {
    out[i] = mfx_index_create_from_header(hdr, 0.0, devices[i]);
    if (!out[i]) { rc = mfx_last_error_code() ? mfx_last_error_code() : MFX_E_NOMEM; break; }
    dev.push_back(devices[i]);
    ptr.push_back(std::vector<void *>{out[i]->d_slots, out[i]->d_meta});
  }
  if (rc == MFX_OK) {
    { DevGuard g(src->device); (void)hipDeviceSynchronize(); }          // the source's last inserts are done
    rc = tree_copy(dev, ptr, std::vector<size_t>{(size_t)(src->total_lines() * MFX_ALIGN), 4 * sizeof(uint64_t)});
  }
  if (rc != MFX_OK) {
    const std::string why = mfx_last_error();
    for (uint32_t i = 0; i < n; ++i) { if (out[i]) mfx_index_free(out[i]); out[i] = nullptr; }
    return mfx_fail(rc, "copying the k-mer table to %u device(s) failed: %s", n, why.c_str());
  }
  for (uint32_t i = 0; i < n; ++i) {
    out[i]->fingerprint = src->fingerprint;
    out[i]->seq_digest = src->seq_digest;
    (void)mfx_index_commit(out[i]);
  }
  return MFX_OK;
}

extern "C" mfx_index *mfx_index_replicate(const mfx_index *src, int device) {
  mfx_index *out = nullptr;
  return mfx_index_replicate_many(src, &device, 1, &out) == MFX_OK ? out : nullptr;
}

static uint64_t seq_plane_words(const mfx_seq *s) { return s->buf_bytes / 32 + (MFX_TILE + 64) / 32 + 1; }   // a tile reads 130 words from its first one

static int seq_alloc_planes(mfx_seq *s) {
  if (s->d_codes) return MFX_OK;
  const uint64_t pw = seq_plane_words(s);
  MFX_HIP(hipMalloc((void **)&s->d_codes, pw * sizeof(uint64_t)));
  MFX_HIP(hipMalloc((void **)&s->d_valid, pw * sizeof(uint32_t)));
  MFX_HIP(mfx_memset_now(s->d_valid, 0, pw * sizeof(uint32_t)));           // no valid base behind the sequence
  MFX_HIP(mfx_memset_now(s->d_codes, 0, pw * sizeof(uint64_t)));
  return MFX_OK;
}

// the packed planes of a resident sequence (2-bit codes + one validity bit per base, the tile's own form), made on the device
extern "C" int mfx_seq_pack(mfx_seq *s) {
  if (!s) return mfx_fail(MFX_E_INVAL, "mfx_seq_pack: null argument");
  std::lock_guard<std::mutex> lazy(s->lazy_mu);
  if (s->planes_ok) return MFX_OK;
  DevGuard g(s->device);
  int rc = seq_alloc_planes(s);
  if (rc) return rc;
  if ((rc = seq_need_bases(s)) != MFX_OK) return rc;
  MFX_HIP(mfx_k_pack(s->d_bases, s->d_codes, s->d_valid, s->buf_bytes / 32, nullptr));
  MFX_HIP(hipDeviceSynchronize());
  s->planes_ok = true;
  return MFX_OK;
}

// The assembly travels between devices as its packed planes (0.375 B per base instead of 1); a replica holds the planes
// only and unpacks them if a kernel asks for one byte per base (mfx_seq_ensure_ascii).
extern "C" int mfx_seq_replicate_many(const mfx_seq *csrc, const int *devices, uint32_t n, mfx_seq **out) {
  if (!csrc || !devices || !out || n == 0) return mfx_fail(MFX_E_INVAL, "mfx_seq_replicate_many: null argument");
  for (uint32_t i = 0; i < n; ++i) {
    out[i] = nullptr;
    if (devices[i] < 0 || devices[i] >= mfx_device_count()) return mfx_fail(MFX_E_INVAL, "mfx_seq_replicate_many: device %d of %d", devices[i], mfx_device_count());
  }
  mfx_seq *src = const_cast<mfx_seq *>(csrc);
  if (src->partial) return mfx_seq_partial_error(src, "mfx_seq_replicate");
  int rc = mfx_seq_pack(src);
  if (rc) return rc;
  const uint64_t pw = seq_plane_words(src);
  std::vector<int> dev(1, src->device);
  std::vector<std::vector<void *>> ptr(1, std::vector<void *>{src->d_codes, src->d_valid});
  for (uint32_t i = 0; i < n && rc == MFX_OK; ++i) {
    out[i] = mfx_seq_create(devices[i], src->len.data(), src->ncontigs);
    if (!out[i]) { rc = mfx_last_error_code() ? mfx_last_error_code() : MFX_E_NOMEM; break; }
    DevGuard g(devices[i]);
    rc = seq_alloc_planes(out[i]);
    dev.push_back(devices[i]);
    ptr.push_back(std::vector<void *>{out[i]->d_codes, out[i]->d_valid});
  }
  if (rc == MFX_OK) {
    { DevGuard g(devices[n - 1]); (void)hipDeviceSynchronize(); }       // the memsets of the last replica's planes
    for (uint32_t i = 0; i + 1 < n; ++i) { DevGuard g(devices[i]); (void)hipDeviceSynchronize(); }
    rc = tree_copy(dev, ptr, std::vector<size_t>{(size_t)(pw * sizeof(uint64_t)), (size_t)(pw * sizeof(uint32_t))});
  }
  if (rc != MFX_OK) {
    const std::string why = mfx_last_error();
    for (uint32_t i = 0; i < n; ++i) { if (out[i]) mfx_seq_free(out[i]); out[i] = nullptr; }
    return mfx_fail(rc, "copying the packed assembly to %u device(s) failed: %s", n, why.c_str());
  }
  for (uint32_t i = 0; i < n; ++i) { out[i]->bases_stale = true; out[i]->planes_ok = true; }
  return MFX_OK;
}

extern "C" mfx_seq *mfx_seq_replicate(const mfx_seq *src, int device) {
  mfx_seq *out = nullptr;
  return mfx_seq_replicate_many(src, &device, 1, &out) == MFX_OK ? out : nullptr;
}

extern "C" int mfx_hist_run_multi(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t ndev, mfx_hist_result *out) {
  if (!evs || !seqs || !out || ndev == 0) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_multi: null argument");
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!evs[d] || !seqs[d]) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_multi: null evaluator / sequence for slot %u", d);
    if (evs[d]->device != seqs[d]->device) return mfx_fail(MFX_E_INVAL, "slot %u: evaluator and sequence live on different devices", d);
    if (evs[d]->nbins != evs[0]->nbins || seqs[d]->ntiles != seqs[0]->ntiles || seqs[d]->ncontigs != seqs[0]->ncontigs)
      return mfx_fail(MFX_E_INVAL, "slot %u: evaluators / sequences of one run must be replicas of each other", d);
    for (uint32_t e = 0; e < d; ++e)
      if (evs[e] == evs[d]) return mfx_fail(MFX_E_INVAL, "slots %u and %u share one evaluator (each slot launches concurrently)", e, d);
  }
  if (ndev == 1) return mfx_hist_run(evs[0], seqs[0], out);
  const uint32_t nbins = evs[0]->nbins, ncontigs = seqs[0]->ncontigs;
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs);
  int rc = MFX_OK;
  // launch on every device before waiting for any; every slot works on its evaluator's own stream and image (nothing is
  // created or released per call)
  const bool timing = getenv("MFX_MULTI_TIMING") && atoi(getenv("MFX_MULTI_TIMING"));
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  std::vector<char> launched(ndev, 0);
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    DevGuard g(evs[d]->device);
    rc = eval_run_enqueue(evs[d], seqs[d], d, ndev);
    launched[d] = rc == MFX_OK;
  }
  const double t1 = now();
  // the N images (~1 MB each) are added in slot order -- integers exactly, koverCpy as a fixed-order fp64 sum -- each as
  // soon as its slot is done, i.e. while the later slots still run; slot 0's pinned image is the accumulator
  uint64_t *sum = evs[0]->sr.h_img;
  double kover = 0.0, t_add = 0.0;
  uint64_t novf0 = 0;
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!launched[d] && !evs[d]->sr.kern[0]) continue;
    DevGuard g(evs[d]->device);
    hipError_t e = hipStreamSynchronize(evs[d]->sr.kern[0]);
    if (e != hipSuccess && rc == MFX_OK) rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_multi: slot %u failed: %s", d, hipGetErrorString(e));
    if (rc != MFX_OK) continue;
    const double ta = now();
    const uint64_t *h = evs[d]->sr.h_img;
    if (d == 0) novf0 = sum[2ull * nbins + 2];
    else for (size_t i = 0; i < words; ++i) sum[i] += h[i];
    double kv;
    memcpy(&kv, h + words, sizeof(double));
    kover = d == 0 ? kv : kover + kv;
    t_add += now() - ta;
  }
  const double t2 = now();
  if (rc == MFX_OK) rc = mfx_hist_result_from_counts(nbins, sum, kover, ncontigs, out);
  if (rc == MFX_OK) {
    rc = result_take_overflow(evs[0], novf0, out);
    for (uint32_t d = 1; d < ndev && rc == MFX_OK; ++d) rc = result_take_overflow(evs[d], evs[d]->sr.h_img[2ull * nbins + 2], out);
    if (rc) mfx_hist_result_free(out);
  }
  if (timing)
    fprintf(stderr, "[mfx multi] %u slots: enqueue %.3f ms, wait + add %.3f ms (of which adding the images %.3f), result %.3f ms\n", ndev, (t1 - t0) * 1e3,
            (t2 - t1) * 1e3, t_add * 1e3, (now() - t2) * 1e3);
  return rc;
}

// SURVEY 8(d)'s evaluate phase over N devices driven by one process: "first tile H2D start -> final reduced histogram on host" with the
// assembly in host memory.  Device d receives ONLY the packed planes of its share -- the tiles [T d / N, T (d + 1) / N), cut at
// multiples of 1024 tiles, plus the k - 1 bases of halo behind it -- through its own copy stream, from its own encoder threads (the
// host's threads are dealt to the slots), and evaluates the chunks as they land, exactly as the single-device run does
// (hist_run_streamed_packed).  The images are added on the host in slot order; koverCpy is bit-identical to the single launch: the
// first-level sums of the (tile, wave) values (4096 values = 1024 tiles each: the parts are cut there) are the single launch's own,
// and the host adds them in the order of mfx_sum_partials_kernel.  Contiguous shares, not the block-cyclic deal of the resident
// run: a device's share must be one stretch of the upload.  Every slot needs its own evaluator AND its own sequence object
// (mfx_seq_create on its device; they hold different parts afterwards and refuse whole-sequence calls: mfx_seq::partial).
static unsigned t_sharers_get();
static double sum_like_partials_kernel(const std::vector<double> &p) {
  double s[MFX_BLOCK];
  for (uint32_t t = 0; t < MFX_BLOCK; ++t) {
    double v = 0.0;
    for (size_t i = t; i < p.size(); i += MFX_BLOCK) v = v + p[i];
    s[t] = v;
  }
  for (uint32_t st = MFX_BLOCK / 2; st > 0; st >>= 1)
    for (uint32_t t = 0; t < st; ++t) s[t] = s[t] + s[t + st];
  return 0.0 + s[0];
}

static void stream_part_bounds(uint64_t T, uint32_t n, std::vector<uint64_t> &b) {
  b.assign(n + 1, 0);
  for (uint32_t d = 1; d < n; ++d) {
    uint64_t x = (uint64_t)((__uint128_t)T * d / n);
    x = (x + 512) / 1024 * 1024;
    b[d] = std::min(T, std::max(b[d - 1], x));
  }
  b[n] = T;
}

extern "C" int mfx_hist_run_streamed_multi(mfx_eval *const *evs, mfx_seq *const *seqs, uint32_t ndev, const char *const *bases, mfx_hist_result *out) {
  if (!evs || !seqs || !out || ndev == 0) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_multi: null argument");
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!evs[d] || !seqs[d]) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_multi: null evaluator / sequence for slot %u", d);
    if (evs[d]->device != seqs[d]->device) return mfx_fail(MFX_E_INVAL, "slot %u: evaluator and sequence live on different devices", d);
    if (evs[d]->nbins != evs[0]->nbins || seqs[d]->ntiles != seqs[0]->ntiles || seqs[d]->ncontigs != seqs[0]->ncontigs || seqs[d]->len != seqs[0]->len)
      return mfx_fail(MFX_E_INVAL, "slot %u: the sequence objects of one run must have the same contig lengths, the evaluators the same bins", d);
    if (evs[d]->ix->wide()) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_multi: k > 31 is not supported (the 128-bit kernels read one byte per base)");
    for (uint32_t e = 0; e < d; ++e)
      if (evs[e] == evs[d] || seqs[e] == seqs[d]) return mfx_fail(MFX_E_INVAL, "slots %u and %u share an evaluator or a sequence object (every slot streams its own part)", e, d);
  }
  if (seqs[0]->ncontigs && !bases) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_multi: null argument");
  if (ndev == 1) return mfx_hist_run_streamed(evs[0], seqs[0], bases, out);
  const uint32_t nbins = evs[0]->nbins, ncontigs = seqs[0]->ncontigs;
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs);
  std::vector<uint64_t> bound;
  stream_part_bounds(seqs[0]->ntiles, ndev, bound);
  std::vector<std::vector<double>> sums(ndev);
  std::vector<int> rcs(ndev, MFX_OK);
  std::vector<std::string> errs(ndev);
  const unsigned sharers = t_sharers_get() * ndev;
  {
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < ndev; ++d)
      th.emplace_back([&, d] {
        mfx_host_threads_share(sharers);                      // the host's encoder threads are dealt to the slots
        StreamPart p;
        p.tl = bound[d]; p.th = bound[d + 1];
        p.chunk_sums = &sums[d];
        p.want_result = false;
        rcs[d] = hist_run_streamed_packed(evs[d], seqs[d], bases, nullptr, &p);
        if (rcs[d]) errs[d] = mfx_last_error();
      });
    for (auto &t : th) t.join();
  }
  for (uint32_t d = 0; d < ndev; ++d)
    if (rcs[d]) return mfx_fail(rcs[d], "mfx_hist_run_streamed_multi: slot %u: %s", d, errs[d].c_str());
  uint64_t *sum = evs[0]->sr.h_img;                           // slot 0's pinned image is the accumulator
  std::vector<uint64_t> novf(ndev, 0);
  std::vector<double> all;
  for (uint32_t d = 0; d < ndev; ++d) {
    const uint64_t *h = evs[d]->sr.h_img;
    novf[d] = h[2ull * nbins + 2];
    if (d) for (size_t i = 0; i < words; ++i) sum[i] += h[i];
    all.insert(all.end(), sums[d].begin(), sums[d].end());
  }
  int rc = mfx_hist_result_from_counts(nbins, sum, sum_like_partials_kernel(all), ncontigs, out);
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) rc = result_take_overflow(evs[d], novf[d], out);
  if (rc && out->undr) mfx_hist_result_free(out);
  return rc;
}

// One rank's share of the same, for the one-process-per-GPU launcher: the tiles [tile_begin, tile_end) of the assembly in host memory
// are encoded, uploaded and evaluated; counts and koverCpy are ADDED to the caller's device image (cleared by the caller; the
// launcher all-reduces it over the ranks, mfx_hist_allreduce).  Returns when the device is done.  mfx_hist_stream_share gives rank
// r of n the bounds the one-process run uses.
extern "C" int mfx_hist_run_streamed_range(mfx_eval *ev, mfx_seq *seq, const char *const *bases, uint64_t tile_begin, uint64_t tile_end,
                                           uint64_t *d_counts, double *d_kover) {
  if (!ev || !seq || !d_counts || !d_kover || (seq->ncontigs && !bases)) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_range: null argument");
  if (ev->device != seq->device) return mfx_fail(MFX_E_INVAL, "evaluator and sequence live on different devices");
  if (tile_begin > tile_end || tile_end > seq->ntiles) return mfx_fail(MFX_E_INVAL, "tile range [%lu,%lu) outside [0,%lu)",
                                                                      (unsigned long)tile_begin, (unsigned long)tile_end, (unsigned long)seq->ntiles);
  if (ev->ix->wide()) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_streamed_range: k > 31 is not supported");
  StreamPart p;
  p.tl = tile_begin; p.th = tile_end;
  p.d_counts = d_counts; p.d_kover = d_kover;
  p.want_result = false;
  return hist_run_streamed_packed(ev, seq, bases, nullptr, &p);
}

extern "C" int mfx_hist_stream_share(uint64_t ntiles, uint32_t rank, uint32_t nranks, uint64_t *tile_begin, uint64_t *tile_end) {
  if (!nranks || rank >= nranks || !tile_begin || !tile_end) return mfx_fail(MFX_E_INVAL, "mfx_hist_stream_share: rank %u of %u", rank, nranks);
  std::vector<uint64_t> b;
  stream_part_bounds(ntiles, nranks, b);
  *tile_begin = b[rank];
  *tile_end = b[rank + 1];
  return MFX_OK;
}

// PARTS of one assembly, one per slot, each on its own sequence-only index (mfx_index_claim_seq on the slot's contigs,
// mfx_index_count_claimed over the whole assembly, the read database update-only): every slot evaluates ITS contigs on its
// device -- no exchange of k-mers at all, whatever the size of the read database -- and the results are put together: bins and
// counters added, the per-contig counters placed at the contigs' numbers in the whole assembly (contig_ids[d][i] = number of
// slot d's contig i), koverCpy a fixed-order sum over the slots.  What config 5's -hist (15 Gb, a read database beyond one GPU)
// runs on the 8-GPU node: each device holds the slots of its contigs' k-mers only.  Slots may share a device.
extern "C" int mfx_hist_run_parts(mfx_eval *const *evs, const mfx_seq *const *seqs, const uint32_t *const *contig_ids, uint32_t ndev,
                                  uint32_t ncontigs_total, mfx_hist_result *out) {
  if (!evs || !seqs || !contig_ids || !out || ndev == 0) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_parts: null argument");
  std::vector<char> seen(ncontigs_total, 0);
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!evs[d] || !seqs[d] || (seqs[d]->ncontigs && !contig_ids[d])) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_parts: null object for slot %u", d);
    if (evs[d]->device != seqs[d]->device) return mfx_fail(MFX_E_INVAL, "slot %u: evaluator and sequence live on different devices", d);
    if (evs[d]->nbins != evs[0]->nbins) return mfx_fail(MFX_E_INVAL, "slot %u: the evaluators of one run must have the same bins", d);
    for (uint32_t e = 0; e < d; ++e)
      if (evs[e] == evs[d]) return mfx_fail(MFX_E_INVAL, "slots %u and %u share one evaluator (each slot launches concurrently)", e, d);
    for (uint32_t i = 0; i < seqs[d]->ncontigs; ++i) {
      const uint32_t c = contig_ids[d][i];
      if (c >= ncontigs_total || seen[c]) return mfx_fail(MFX_E_INVAL, "slot %u: contig number %u is out of range or belongs to two slots", d, c);
      seen[c] = 1;
    }
  }
  const uint32_t nbins = evs[0]->nbins;
  int rc = MFX_OK;
  std::vector<char> launched(ndev, 0);
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    if (seqs[d]->ntiles == 0) continue;
    DevGuard g(evs[d]->device);
    rc = eval_run_enqueue(evs[d], seqs[d], 0, 1);
    launched[d] = rc == MFX_OK;
  }
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs_total);
  std::vector<uint64_t> sum(words, 0);
  double kover = 0.0;
  std::vector<uint64_t> novf(ndev, 0);
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!launched[d]) continue;
    DevGuard g(evs[d]->device);
    hipError_t e = hipStreamSynchronize(evs[d]->sr.kern[0]);
    if (e != hipSuccess && rc == MFX_OK) rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_parts: slot %u failed: %s", d, hipGetErrorString(e));
    if (rc != MFX_OK) continue;
    const uint32_t nc = seqs[d]->ncontigs;
    const uint64_t *h = evs[d]->sr.h_img;
    const size_t w = MFX_HIST_WORDS(nbins, nc);
    for (size_t i = 0; i < 2ull * nbins + 3; ++i) sum[i] += h[i];
    novf[d] = h[2ull * nbins + 2];
    for (uint32_t i = 0; i < nc; ++i) {
      sum[2ull * nbins + 3 + contig_ids[d][i]] += h[2ull * nbins + 3 + i];
      sum[2ull * nbins + 3 + ncontigs_total + contig_ids[d][i]] += h[2ull * nbins + 3 + nc + i];
    }
    double kv;
    memcpy(&kv, h + w, sizeof(double));
    kover = kover + kv;                                       // slot order: a fixed-order fp64 sum
  }
  if (rc == MFX_OK) rc = mfx_hist_result_from_counts(nbins, sum.data(), kover, ncontigs_total, out);
  if (rc == MFX_OK) {
    for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) if (launched[d]) rc = result_take_overflow(evs[d], novf[d], out);
    if (rc) mfx_hist_result_free(out);
  }
  return rc;
}

extern "C" void mfx_hist_result_free(mfx_hist_result *r) {
  if (!r) return;
  free(r->undr); free(r->over); free(r->contig_kasm); free(r->contig_kmissing);
  memset(r, 0, sizeof(*r));
}


// reportHistogram, merfin-histogram.C:140-176
extern "C" int mfx_hist_report(const mfx_hist_result *r, int k, const char *hist_path, const char *summary_path) {
  if (!r || !r->undr || !r->over) return mfx_fail(MFX_E_INVAL, "mfx_hist_report: empty result");
  if (hist_path) {
    mfx_file fh = mfx_open_writer(hist_path, false);     // compressedFileWriter: compressor chosen by suffix (mfx_pipe.h)
    FILE *f = fh.f;
    if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", hist_path);
    for (uint64_t ii = r->undrMax - 1; ii > 0; ii--)
      if (r->undr[ii] > 0)
        fprintf(f, "%.1f\t%lu\n", ((double)ii * -0.2), (unsigned long)r->undr[ii]);
    fprintf(f, "%.1f\t%lu\n", 0.0, (unsigned long)(r->undr[0] + r->over[0]));
    for (uint64_t ii = 1; ii < r->overMax; ii++)
      if (r->over[ii] > 0)
        fprintf(f, "%.1f\t%lu\n", ((double)ii * 0.2), (unsigned long)r->over[ii]);
    if (mfx_close(fh)) return mfx_fail(MFX_E_IO, "writing '%s' failed (stream error or the compressor exited with an error)", hist_path);
  }
  if (summary_path) {
    FILE *f = strcmp(summary_path, "-") == 0 ? stderr : fopen(summary_path, "w");
    if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", summary_path);
    fprintf(f, "\n");
    fprintf(f, "K-mers not found in reads (missing) : %lu\n", (unsigned long)r->kmissing);
    fprintf(f, "K-mers overly represented in assembly: %.2f\n", r->koverCpy);
    fprintf(f, "K-mers found in the assembly: %lu\n", (unsigned long)r->kasm);
    fprintf(f, "Missing QV: %.2f\n", mfx_histoQV((double)r->kmissing, (double)r->kasm, k));
    fprintf(f, "Merfin QV*: %.2f\n", mfx_histoQV(r->kmissing + r->koverCpy, (double)r->kasm, k));
    fprintf(f, "*** Note this QV is valid only if -seqmer was generated with -sequence ***\n\n");
    fprintf(f, "*** Missing QV only considers missing kmers as errors. Merfin QV* includes overrepresented kmers. ***\n\n");
    fprintf(f, "*** When the lookup table is provided, missing QV includes weighted low frequency kmers, otherwise it is identical to Merqury QV. ***\n\n");
    if (f != stderr) fclose(f);
  }
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// sharded index (BASELINE config 5)
// ---------------------------------------------------------------------------
extern "C" int mfx_index_set_shard(mfx_index *ix, uint32_t rank, uint32_t nranks) {
  if (!ix || nranks == 0 || rank >= nranks || nranks > 254)
    return mfx_fail(MFX_E_INVAL, "mfx_index_set_shard: need rank < nranks <= 254");
  if (ix->seq_only && nranks > 1) return mfx_fail(MFX_E_INVAL, "a sequence-only index cannot be sharded (it is the small index: shard a full one)");
  if (ix->wide() && nranks > 1) return mfx_fail(MFX_E_INVAL, "a sharded index handles k <= 31; this index holds %d-mers", ix->k);
  DevGuard g(ix->device);
  uint64_t meta[4];
  MFX_HIP(hipMemcpy(meta, ix->d_meta, sizeof(meta), hipMemcpyDeviceToHost));
  if (meta[0] != 0) return mfx_fail(MFX_E_INVAL, "mfx_index_set_shard: the index already holds k-mers");
  ix->shard_rank = rank;
  ix->shard_n = nranks;
  return MFX_OK;
}

struct mfx_router {
  const mfx_index *ix = nullptr;
  int       device = 0;
  uint32_t  nranks = 1, max_tiles = 0;
  uint64_t *d_keys = nullptr;        // [max_tiles * TILE]
  uint8_t  *d_owner = nullptr, *d_owner2 = nullptr;
  uint32_t *d_idx = nullptr, *d_idx2 = nullptr;
  uint64_t *d_dest = nullptr;        // [256]
  void     *d_tmp = nullptr;
  size_t    tmp_bytes = 0;
  uint32_t *d_tile_cnt = nullptr;    // [max_tiles * nranks] sort-free path (nranks <= MFX_SPLIT_MAX_RANKS)
  bool      split = false;
  // buffers of the one-pass routing of mfx_hist_run_sharded, made by its first run and kept for the next ones (allocating
  // gigabytes next to tables that fill the device took up to 0.3 s of a 0.08 s run)
  struct Fused {
    uint64_t *d_keys[2] = {nullptr, nullptr}, *d_rkeys = nullptr, *d_cursors = nullptr, *h_cursors = nullptr;
    uint32_t *d_ctg[2] = {nullptr, nullptr}, *d_rctg = nullptr;
    size_t region_cap = 0, rcap = 0;
    uint32_t ndev = 0;
  } fused;
};

int mfx_sort_by_owner(void *tmp, size_t &tmp_bytes, const uint8_t *kin, uint8_t *kout, const uint32_t *vin, uint32_t *vout,
                      uint64_t n, hipStream_t st);   // mfx_sort.hip (hipcub stable radix sort)

extern "C" mfx_router *mfx_router_create(const mfx_index *ix, uint32_t nranks, uint32_t max_tiles) {
  if (ix && ix->wide()) {
    mfx_fail(MFX_E_INVAL, "mfx_router_create: a sharded index handles k <= 31; this index holds %d-mers", ix->k);
    return nullptr;
  }
  if (ix && ix->seq_only) {
    mfx_fail(MFX_E_INVAL, "mfx_router_create: a sequence-only index cannot be sharded");
    return nullptr;
  }
  if (!ix || nranks == 0 || nranks > 254 || max_tiles == 0 || (uint64_t)max_tiles * MFX_TILE >= (1ull << 31)) {
    mfx_fail(MFX_E_INVAL, "mfx_router_create: bad argument (nranks <= 254, max_tiles * %u < 2^31: the sort counts items in an int)", MFX_TILE);
    return nullptr;
  }
  DevGuard g(ix->device);
  mfx_router *r = new mfx_router;
  r->ix = ix; r->device = ix->device; r->nranks = nranks; r->max_tiles = max_tiles;
  const size_t n = (size_t)max_tiles * MFX_TILE;
  // small worlds: counting split, no position-sized scratch; MFX_ROUTE_SORT=1 forces the radix-sort path (A/B, tests)
  const char *fs = getenv("MFX_ROUTE_SORT");
  r->split = nranks <= MFX_SPLIT_MAX_RANKS && !(fs && atoi(fs));
  if (r->split) {
    if (hipMalloc((void **)&r->d_tile_cnt, (size_t)max_tiles * nranks * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void **)&r->d_dest, 256 * 8) != hipSuccess) {
      mfx_fail(MFX_E_NOMEM, "mfx_router_create: device allocation failed (%u tiles)", max_tiles);
      mfx_router_free(r);
      return nullptr;
    }
    return r;
  }
  size_t tb = 0;
  mfx_sort_by_owner(nullptr, tb, nullptr, nullptr, nullptr, nullptr, n, nullptr);
  r->tmp_bytes = tb;
  if (hipMalloc((void **)&r->d_keys, n * 8) != hipSuccess || hipMalloc((void **)&r->d_owner, n) != hipSuccess ||
      hipMalloc((void **)&r->d_owner2, n) != hipSuccess || hipMalloc((void **)&r->d_idx, n * 4) != hipSuccess ||
      hipMalloc((void **)&r->d_idx2, n * 4) != hipSuccess || hipMalloc((void **)&r->d_dest, 256 * 8) != hipSuccess ||
      hipMalloc(&r->d_tmp, tb ? tb : 1) != hipSuccess) {
    mfx_fail(MFX_E_NOMEM, "mfx_router_create: device allocation failed (%zu positions)", n);
    mfx_router_free(r);
    return nullptr;
  }
  return r;
}

static void router_fused_free(mfx_router *r) {
  auto &F = r->fused;
  void *p[] = {F.d_keys[0], F.d_keys[1], F.d_rkeys, F.d_cursors, F.d_ctg[0], F.d_ctg[1], F.d_rctg};
  for (void *x : p) if (x) (void)hipFree(x);
  if (F.h_cursors) (void)hipHostFree(F.h_cursors);
  F = mfx_router::Fused();
}

extern "C" void mfx_router_free(mfx_router *r) {
  if (!r) return;
  DevGuard g(r->device);
  void *p[] = {r->d_keys, r->d_owner, r->d_owner2, r->d_idx, r->d_idx2, r->d_dest, r->d_tmp, r->d_tile_cnt};
  for (void *x : p) if (x) (void)hipFree(x);
  router_fused_free(r);
  delete r;
}

extern "C" int mfx_route_tiles(mfx_router *r, const mfx_seq *seq, uint64_t tile_begin, uint64_t tile_end, uint32_t nbins,
                               uint64_t *d_counts, uint64_t *d_keys_out, uint32_t *d_contigs_out, uint64_t *h_dest_counts,
                               void *stream) {
  if (!r || !seq || !d_counts || !d_keys_out || !d_contigs_out || !h_dest_counts)
    return mfx_fail(MFX_E_INVAL, "mfx_route_tiles: null argument");
  if (tile_begin > tile_end || tile_end > seq->ntiles || tile_end - tile_begin > r->max_tiles)
    return mfx_fail(MFX_E_INVAL, "mfx_route_tiles: tile range [%lu,%lu) invalid (max %u tiles per call)",
                    (unsigned long)tile_begin, (unsigned long)tile_end, r->max_tiles);
  DevGuard g(r->device);
  hipStream_t st = (hipStream_t)stream;
  int canon = 0;
  int rc = index_canonical(r->ix, &canon);
  if (rc) return rc;
  if (!canon || !(r->ix->k & 1)) return mfx_fail(MFX_E_INVAL, "a sharded index needs a canonical k-mer database and odd k");
  const uint64_t n = (tile_end - tile_begin) * MFX_TILE;
  if (int erc = mfx_seq_ensure_ascii(seq)) return erc;
  mfx_route_args a;
  a.t = r->ix->view();
  a.bases = seq->d_bases;
  a.contig_off = seq->d_contig_off;
  a.contig_len = seq->d_contig_len;
  a.tile_start = seq->d_tile_start;
  a.ncontigs = seq->ncontigs;
  a.tile_begin = tile_begin;
  a.tile_end = tile_end;
  a.nranks = r->nranks;
  a.keys = r->d_keys;
  a.owner = r->d_owner;
  a.dest_counts = r->d_dest;
  a.counts = d_counts;
  a.nbins = nbins;
  a.tile_contig = seq->d_tile_contig;
  a.tile_cnt = r->d_tile_cnt;
  if (r->split) {
    MFX_HIP(mfx_k_route_split(a, d_keys_out, d_contigs_out, st));
    for (uint32_t i = 0; i < r->nranks; ++i) h_dest_counts[i] = 0;
    if (tile_end > tile_begin) {
      MFX_HIP(hipMemcpyAsync(h_dest_counts, r->d_dest, r->nranks * 8, hipMemcpyDeviceToHost, st));
      MFX_HIP(hipStreamSynchronize(st));
    }
    return MFX_OK;
  }
  MFX_HIP(hipMemsetAsync(r->d_dest, 0, 256 * 8, st));
  MFX_HIP(mfx_k_route(a, st));
  MFX_HIP(mfx_k_iota(r->d_idx, n, st));
  // stable sort by owner: within a destination the k-mers keep their sequence order,
  // so the owner's fp64 koverCpy sum is reproducible
  rc = mfx_sort_by_owner(r->d_tmp, r->tmp_bytes, r->d_owner, r->d_owner2, r->d_idx, r->d_idx2, n, st);
  if (rc) return rc;
  MFX_HIP(hipMemcpyAsync(h_dest_counts, r->d_dest, r->nranks * 8, hipMemcpyDeviceToHost, st));
  MFX_HIP(hipStreamSynchronize(st));
  uint64_t nvalid = 0;
  for (uint32_t i = 0; i < r->nranks; ++i) nvalid += h_dest_counts[i];
  MFX_HIP(mfx_k_route_gather(a, r->d_idx2, nvalid, d_keys_out, d_contigs_out, st));
  // the groups are complete when this returns, as on the split path: callers hand them to OTHER streams right away
  // (mfx_hist_run_sharded: the owners' peer copies), which nothing else orders behind this gather
  MFX_HIP(hipStreamSynchronize(st));
  return MFX_OK;
}

extern "C" int mfx_hist_keys_launch(mfx_eval *ev, const uint64_t *d_keys, const uint32_t *d_contigs, uint64_t n,
                                    uint32_t ncontigs, uint64_t *d_counts, double *d_kover, void *stream) {
  if (!ev || !d_counts || !d_kover || (n && (!d_keys || !d_contigs))) return mfx_fail(MFX_E_INVAL, "mfx_hist_keys_launch: null argument");
  if (ev->ix->wide()) return mfx_fail(MFX_E_INVAL, "mfx_hist_keys_launch: a sharded index handles k <= 31");
  if (ev->ix->seq_only) return mfx_fail(MFX_E_INVAL, "mfx_hist_keys_launch: a sequence-only index holds the k-mers of one sequence, the sharded path needs a full (sharded) one; build a full index (mfx_index_create)");
  DevGuard g(ev->device);
  mfx_hist_keys_args a;
  a.t = ev->ix->view();
  a.keys = d_keys;
  a.contig = d_contigs;
  a.n = n;
  a.ks.peak = ev->peak;
  a.ks.n_prob = ev->n_prob;
  a.ks.probK = ev->d_probK;
  a.ks.probP = ev->d_probP;
  a.ks.nbins = ev->nbins;
  a.ks.ncontigs = ncontigs;
  a.ks.counts = d_counts;
  a.ks.partials = ev->d_partials;
  a.ks.ovf = ev->d_ovf;
  MFX_HIP(mfx_k_hist_keys(a, ev->grid, (hipStream_t)stream));
  MFX_HIP(mfx_k_sum_partials(ev->d_partials, (uint32_t)ev->grid, d_kover, (hipStream_t)stream));
  return MFX_OK;
}

// Sharded index driven by ONE process (BASELINE config 5 from the C++ side): slot d holds shard d of N of the
// k-mer table on its own device.  Per round, every slot routes a chunk of ITS tile range (k-mers grouped by owner,
// sequence order kept), the groups travel to their owners by peer copy over xGMI -- the all-to-all of the
// one-process-per-GPU form (merfin_amd/distributed.py::sharded_hist) -- and every owner probes / computes K* / bins
// what it received, source by source in slot order (so koverCpy is a fixed-order sum).  The N counts images are added
// on the host; kasm was counted at the sources, kmissing / bins / koverCpy at the owners.
static int hist_run_sharded_ordered(mfx_eval *const *evs, mfx_router *const *routers, const mfx_seq *const *seqs, uint32_t ndev,
                                    mfx_hist_result *out) {
  if (!evs || !routers || !seqs || !out || ndev == 0) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_sharded: null argument");
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!evs[d] || !routers[d] || !seqs[d]) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_sharded: null object for slot %u", d);
    const mfx_index *ix = evs[d]->ix;
    if (ix->shard_n != ndev || ix->shard_rank != d || routers[d]->ix != ix || routers[d]->nranks != ndev)
      return mfx_fail(MFX_E_INVAL, "slot %u: its index must be shard %u of %u (mfx_index_set_shard) and its router built on it for %u ranks", d, d, ndev, ndev);
    if (evs[d]->device != seqs[d]->device || evs[d]->nbins != evs[0]->nbins || seqs[d]->ntiles != seqs[0]->ntiles ||
        seqs[d]->ncontigs != seqs[0]->ncontigs || routers[d]->max_tiles != routers[0]->max_tiles)
      return mfx_fail(MFX_E_INVAL, "slot %u: evaluators / sequences / routers of one run must match each other", d);
  }
  const uint32_t nbins = evs[0]->nbins, ncontigs = seqs[0]->ncontigs, per = routers[0]->max_tiles;
  const uint64_t T = seqs[0]->ntiles;
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs), cap = (size_t)per * MFX_TILE;
  const size_t rcap = cap + cap / 2 + 4096;                  // receive side: a balanced owner gets ~cap k-mers per round
  struct Slot {
    uint64_t *d_counts = nullptr, *d_keys = nullptr, *d_rkeys = nullptr;
    uint32_t *d_ctg = nullptr, *d_rctg = nullptr;
    double *d_kover = nullptr;
    hipStream_t st = nullptr;
    std::vector<uint64_t> dest;      // k-mers this slot routed to each owner in the current round
    int rc = MFX_OK;
    std::string err;
  };
  std::vector<Slot> sl(ndev);
  int rc = MFX_OK;
  auto release = [&]() {
    for (uint32_t d = 0; d < ndev; ++d) {
      DevGuard g(evs[d]->device);
      if (sl[d].st) (void)hipStreamSynchronize(sl[d].st);
      void *p[] = {sl[d].d_counts, sl[d].d_keys, sl[d].d_rkeys, sl[d].d_ctg, sl[d].d_rctg, sl[d].d_kover};
      for (void *x : p) if (x) (void)hipFree(x);
      if (sl[d].st) (void)hipStreamDestroy(sl[d].st);
    }
  };
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    DevGuard g(evs[d]->device);
    sl[d].dest.assign(ndev, 0);
    if (hipStreamCreateWithFlags(&sl[d].st, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **)&sl[d].d_counts, words * sizeof(uint64_t)) != hipSuccess || hipMalloc((void **)&sl[d].d_kover, sizeof(double)) != hipSuccess ||
        hipMalloc((void **)&sl[d].d_keys, cap * 8) != hipSuccess || hipMalloc((void **)&sl[d].d_rkeys, rcap * 8) != hipSuccess ||
        hipMalloc((void **)&sl[d].d_ctg, cap * 4) != hipSuccess || hipMalloc((void **)&sl[d].d_rctg, rcap * 4) != hipSuccess ||
        hipMemsetAsync(sl[d].d_counts, 0, words * sizeof(uint64_t), sl[d].st) != hipSuccess ||
        hipMemsetAsync(sl[d].d_kover, 0, sizeof(double), sl[d].st) != hipSuccess ||
        ovf_reset_hip(evs[d], sl[d].st) != hipSuccess)
      rc = mfx_fail(MFX_E_NOMEM, "mfx_hist_run_sharded: buffers for slot %u (%zu k-mers per round) could not be set up", d, cap);
    for (uint32_t e = 0; e < d && rc == MFX_OK; ++e)
      if (evs[e]->device != evs[d]->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, evs[d]->device, evs[e]->device) == hipSuccess && can && hipDeviceEnablePeerAccess(evs[e]->device, 0) != hipSuccess)
          (void)hipGetLastError();                            // already enabled
      }
  }
  const uint64_t rounds = ((T + ndev - 1) / ndev + per - 1) / per;
  for (uint64_t r = 0; r < rounds && rc == MFX_OK; ++r) {
    // ---- route: all slots at once (each call blocks until its group sizes are on the host)
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < ndev; ++d)
      th.emplace_back([&, d]() {
        const uint64_t lo = T * d / ndev, hi = T * (d + 1) / ndev;
        const uint64_t tb = std::min(hi, lo + r * per), te = std::min(hi, tb + per);
        sl[d].rc = mfx_route_tiles(routers[d], seqs[d], tb, te, nbins, sl[d].d_counts, sl[d].d_keys, sl[d].d_ctg, sl[d].dest.data(), sl[d].st);
        if (sl[d].rc) sl[d].err = mfx_last_error();
      });
    for (auto &x : th) x.join();
    for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d)
      if (sl[d].rc) rc = mfx_fail(sl[d].rc, "slot %u: %s", d, sl[d].err.c_str());
    if (rc) break;
    // ---- exchange + evaluate: owner o takes its group from every source in slot order
    for (uint32_t o = 0; o < ndev && rc == MFX_OK; ++o) {
      DevGuard g(evs[o]->device);
      // the groups of all sources land one behind the other (slot order) and are evaluated by ONE launch: a launch per
      // source was 8x the launches, each with its own ramp-up and tail.  (Should the owners be so unbalanced that one
      // owner's share of a round outgrows its buffer, it takes its groups source by source.)
      uint64_t total = 0;
      for (uint32_t s2 = 0; s2 < ndev; ++s2) total += sl[s2].dest[o];
      const bool merged = total <= rcap;
      uint64_t at = 0;
      for (uint32_t s2 = 0; s2 < ndev && rc == MFX_OK; ++s2) {
        const uint64_t n = sl[s2].dest[o];
        if (!n) continue;
        uint64_t off = 0;
        for (uint32_t q = 0; q < o; ++q) off += sl[s2].dest[q];
        hipError_t e = hipMemcpyPeerAsync(sl[o].d_rkeys + at, evs[o]->device, sl[s2].d_keys + off, evs[s2]->device, n * 8, sl[o].st);
        if (e == hipSuccess) e = hipMemcpyPeerAsync(sl[o].d_rctg + at, evs[o]->device, sl[s2].d_ctg + off, evs[s2]->device, n * 4, sl[o].st);
        if (e != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "peer copy of %lu routed k-mers from slot %u to slot %u failed: %s", (unsigned long)n, s2, o, hipGetErrorString(e)); break; }
        if (merged) at += n;
        else rc = mfx_hist_keys_launch(evs[o], sl[o].d_rkeys, sl[o].d_rctg, n, ncontigs, sl[o].d_counts, sl[o].d_kover, sl[o].st);
      }
      if (merged && at && rc == MFX_OK)
        rc = mfx_hist_keys_launch(evs[o], sl[o].d_rkeys, sl[o].d_rctg, at, ncontigs, sl[o].d_counts, sl[o].d_kover, sl[o].st);
    }
    // the sources' buffers are rewritten by the next round's routing: every owner must have taken its groups
    for (uint32_t d = 0; d < ndev; ++d) {
      DevGuard g(evs[d]->device);
      if (hipStreamSynchronize(sl[d].st) != hipSuccess && rc == MFX_OK) rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_sharded: slot %u failed: %s", d, hipGetErrorString(hipGetLastError()));
    }
  }
  std::vector<uint64_t> sum(words, 0), h(words);
  double kover = 0.0;
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    DevGuard g(evs[d]->device);
    double kv = 0.0;
    if (hipMemcpy(h.data(), sl[d].d_counts, words * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(&kv, sl[d].d_kover, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_sharded: D2H of slot %u failed", d); break; }
    for (size_t i = 0; i < words; ++i) sum[i] += h[i];
    kover = kover + kv;                                       // slot order: fixed-order fp64 sum
    sl[d].dest.assign(1, h[2ull * nbins + 2]);                // this slot's overflow count
  }
  if (rc == MFX_OK) rc = mfx_hist_result_from_counts(nbins, sum.data(), kover, ncontigs, out);
  if (rc == MFX_OK) {
    for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) rc = result_take_overflow(evs[d], sl[d].dest[0], out);
    if (rc) mfx_hist_result_free(out);
  }
  release();
  return rc;
}


// The same run with the ONE-PASS router (mfx_route_fused_kernel) and owners that evaluate the groups WHERE THEY LIE
// (mfx_hist_keys_kernel<true>: up to 16 segments per launch): a tile is decoded once instead of twice, no per-tile prefix, a
// group whose source shares the owner's device is not copied at all, and the routing of round r + 1 runs under the owners'
// evaluation of round r (two sets of group buffers).  The price is the ORDER of an owner's k-mers, which only the fp64 sum of
// koverCpy ever needed: the owners sum it in fixed point (units of 2^-52, 128-bit integers -- adds commute), so the result is
// still bit-identical run to run, whatever the devices and the scheduling.  Taken when every slot is one of <= 16 ranks and
// every prob of the K* table lies in [0, 4096) (the fixed-point range); else hist_run_sharded_ordered.  A round so unbalanced
// that an owner's region overflows (a megabase of one repeated minimizer) is routed again by the exact counting split.
extern "C" int mfx_hist_run_sharded(mfx_eval *const *evs, mfx_router *const *routers, const mfx_seq *const *seqs, uint32_t ndev,
                                    mfx_hist_result *out) {
  if (!evs || !routers || !seqs || !out || ndev == 0) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_sharded: null argument");
  for (uint32_t d = 0; d < ndev; ++d) {
    if (!evs[d] || !routers[d] || !seqs[d]) return mfx_fail(MFX_E_INVAL, "mfx_hist_run_sharded: null object for slot %u", d);
    const mfx_index *ix = evs[d]->ix;
    if (ix->shard_n != ndev || ix->shard_rank != d || routers[d]->ix != ix || routers[d]->nranks != ndev)
      return mfx_fail(MFX_E_INVAL, "slot %u: its index must be shard %u of %u (mfx_index_set_shard) and its router built on it for %u ranks", d, d, ndev, ndev);
    if (evs[d]->device != seqs[d]->device || evs[d]->nbins != evs[0]->nbins || seqs[d]->ntiles != seqs[0]->ntiles ||
        seqs[d]->ncontigs != seqs[0]->ncontigs || routers[d]->max_tiles != routers[0]->max_tiles)
      return mfx_fail(MFX_E_INVAL, "slot %u: evaluators / sequences / routers of one run must match each other", d);
  }
  bool fused = ndev <= MFX_KEYS_MAX_SEGS && ndev <= MFX_SPLIT_MAX_RANKS;
  for (uint32_t d = 0; d < ndev && fused; ++d) {
    if (!routers[d]->split || evs[d]->ix->wide() || evs[d]->ix->seq_only) fused = false;
    for (double p : evs[d]->probP) if (!(p >= 0.0 && p < 4096.0)) fused = false;
  }
  if (const char *e = getenv("MFX_SHARDED_ORDERED")) if (atoi(e)) fused = false;      // A/B, tests: the ordered form
  if (!fused) return hist_run_sharded_ordered(evs, routers, seqs, ndev, out);

  const uint32_t nbins = evs[0]->nbins, ncontigs = seqs[0]->ncontigs, per = routers[0]->max_tiles;
  const uint64_t T = seqs[0]->ntiles;
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs), cap = (size_t)per * MFX_TILE;
  const size_t region_cap = cap / ndev + cap / (4 * ndev) + 2 * MFX_TILE;     // an owner's share of a round: 1/N of it + 25 % + two tiles
  bool any_remote = false;
  for (uint32_t d = 1; d < ndev; ++d) if (evs[d]->device != evs[0]->device) any_remote = true;
  const size_t rcap = any_remote ? cap + cap / 2 + 4096 : 0;                 // receive side of the groups that come from other devices
  struct Slot {
    uint64_t *d_counts = nullptr, *d_keys[2] = {nullptr, nullptr}, *d_rkeys = nullptr, *d_cursors = nullptr, *d_kfix = nullptr, *d_pkeys = nullptr;
    uint32_t *d_ctg[2] = {nullptr, nullptr}, *d_rctg = nullptr, *d_pctg = nullptr;
    uint64_t *h_cursors = nullptr;                               // pinned: [2][ndev + 1]
    double   *d_kover = nullptr;                                 // (the exact fallback's ordered partial sums land here)
    hipStream_t rst = nullptr, ost = nullptr;
    // the groups this slot routed in a round: [set][owner] -> where and how many
    std::vector<const uint64_t *> gkeys[2];
    std::vector<const uint32_t *> gctg[2];
    std::vector<uint64_t> gn[2];
    int rc = MFX_OK;
    std::string err;
  };
  std::vector<Slot> sl(ndev);
  int rc = MFX_OK;
  auto release = [&]() {
    for (uint32_t d = 0; d < ndev; ++d) {
      DevGuard g(evs[d]->device);
      if (sl[d].rst) (void)hipStreamSynchronize(sl[d].rst);
      if (sl[d].ost) (void)hipStreamSynchronize(sl[d].ost);
      void *p[] = {sl[d].d_counts, sl[d].d_kfix, sl[d].d_pkeys, sl[d].d_pctg, sl[d].d_kover};      // (the group buffers stay with the router)
      for (void *x : p) if (x) (void)hipFree(x);
      if (sl[d].rst) (void)hipStreamDestroy(sl[d].rst);
      if (sl[d].ost) (void)hipStreamDestroy(sl[d].ost);
    }
  };
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    DevGuard g(evs[d]->device);
    Slot &S = sl[d];
    for (int b2 = 0; b2 < 2; ++b2) { S.gkeys[b2].assign(ndev, nullptr); S.gctg[b2].assign(ndev, nullptr); S.gn[b2].assign(ndev, 0); }
    bool ok = hipStreamCreateWithFlags(&S.rst, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&S.ost, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void **)&S.d_counts, words * sizeof(uint64_t)) == hipSuccess && hipMalloc((void **)&S.d_kover, sizeof(double)) == hipSuccess &&
              hipMalloc((void **)&S.d_kfix, 2 * sizeof(uint64_t)) == hipSuccess;
    // the group buffers live in the router (made by the first run on it)
    auto &F = routers[d]->fused;
    if (ok && (F.region_cap != region_cap || F.rcap != rcap || F.ndev != ndev)) {
      router_fused_free(routers[d]);
      F.region_cap = region_cap; F.rcap = rcap; F.ndev = ndev;
      ok = hipMalloc((void **)&F.d_cursors, 2 * (ndev + 1) * sizeof(uint64_t)) == hipSuccess &&
           hipHostMalloc((void **)&F.h_cursors, 2 * (ndev + 1) * sizeof(uint64_t), hipHostMallocDefault) == hipSuccess;
      for (int b2 = 0; b2 < 2 && ok; ++b2)
        ok = hipMalloc((void **)&F.d_keys[b2], (size_t)ndev * region_cap * 8) == hipSuccess && hipMalloc((void **)&F.d_ctg[b2], (size_t)ndev * region_cap * 4) == hipSuccess;
      if (ok && rcap) ok = hipMalloc((void **)&F.d_rkeys, rcap * 8) == hipSuccess && hipMalloc((void **)&F.d_rctg, rcap * 4) == hipSuccess;
      if (!ok) router_fused_free(routers[d]);
    }
    if (ok) {
      S.d_keys[0] = F.d_keys[0]; S.d_keys[1] = F.d_keys[1]; S.d_ctg[0] = F.d_ctg[0]; S.d_ctg[1] = F.d_ctg[1];
      S.d_rkeys = F.d_rkeys; S.d_rctg = F.d_rctg; S.d_cursors = F.d_cursors; S.h_cursors = F.h_cursors;
    }
    ok = ok && hipMemsetAsync(S.d_counts, 0, words * sizeof(uint64_t), S.ost) == hipSuccess && hipMemsetAsync(S.d_kover, 0, sizeof(double), S.ost) == hipSuccess &&
         hipMemsetAsync(S.d_kfix, 0, 2 * sizeof(uint64_t), S.ost) == hipSuccess && ovf_reset_hip(evs[d], S.ost) == hipSuccess &&
         hipStreamSynchronize(S.ost) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); rc = mfx_fail(MFX_E_NOMEM, "mfx_hist_run_sharded: buffers for slot %u (%zu k-mers per round) could not be set up", d, cap); }
    for (uint32_t e = 0; e < d && rc == MFX_OK; ++e)
      if (evs[e]->device != evs[d]->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, evs[d]->device, evs[e]->device) == hipSuccess && can && hipDeviceEnablePeerAccess(evs[e]->device, 0) != hipSuccess)
          (void)hipGetLastError();                            // already enabled
        if (hipDeviceCanAccessPeer(&can, evs[e]->device, evs[d]->device) == hipSuccess && can) {
          DevGuard g2(evs[e]->device);
          if (hipDeviceEnablePeerAccess(evs[d]->device, 0) != hipSuccess) (void)hipGetLastError();
        }
      }
  }
  const uint64_t rounds = ((T + ndev - 1) / ndev + per - 1) / per;
  // ---- route round r of every slot into buffer set b2 (one host thread per slot; returns when the group sizes are on the host)
  auto route_round = [&](uint64_t r, int b2) {
    std::vector<std::thread> th;
    for (uint32_t d = 0; d < ndev; ++d)
      th.emplace_back([&, d]() {
        Slot &S = sl[d];
        S.rc = MFX_OK;
        const uint64_t lo = T * d / ndev, hi = T * (d + 1) / ndev;
        const uint64_t tb = std::min(hi, lo + r * per), te = std::min(hi, tb + per);
        for (uint32_t o = 0; o < ndev; ++o) { S.gn[b2][o] = 0; S.gkeys[b2][o] = nullptr; S.gctg[b2][o] = nullptr; }
        if (te <= tb) return;
        DevGuard g(evs[d]->device);
        mfx_router *R = routers[d];
        const mfx_seq *seq = seqs[d];
        auto fail = [&](int code, const char *what, hipError_t e) { S.rc = code; S.err = std::string(what) + ": " + hipGetErrorString(e); (void)hipGetLastError(); };
        int canon = 0;
        if (int erc = index_canonical(R->ix, &canon)) { S.rc = erc; S.err = mfx_last_error(); return; }
        if (!canon || !(R->ix->k & 1)) { S.rc = MFX_E_INVAL; S.err = "a sharded index needs a canonical k-mer database and odd k"; return; }
        if (int erc = mfx_seq_ensure_ascii(seq)) { S.rc = erc; S.err = mfx_last_error(); return; }
        mfx_route_args a;
        a.t = R->ix->view();
        a.bases = seq->d_bases;
        a.contig_off = seq->d_contig_off; a.contig_len = seq->d_contig_len; a.tile_start = seq->d_tile_start;
        a.ncontigs = seq->ncontigs;
        a.tile_begin = tb; a.tile_end = te;
        a.nranks = ndev;
        a.keys = nullptr; a.owner = nullptr; a.dest_counts = R->d_dest;
        a.counts = S.d_counts;
        a.nbins = nbins;
        a.tile_contig = seq->d_tile_contig;
        a.tile_cnt = R->d_tile_cnt;
        uint64_t *cur = S.d_cursors + (size_t)b2 * (ndev + 1), *hcur = S.h_cursors + (size_t)b2 * (ndev + 1);
        hipError_t e = hipMemsetAsync(cur, 0, (ndev + 1) * sizeof(uint64_t), S.rst);
        if (e == hipSuccess) e = mfx_k_route_fused(a, S.d_keys[b2], S.d_ctg[b2], cur, region_cap, S.rst);
        if (e == hipSuccess) e = hipMemcpyAsync(hcur, cur, (ndev + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, S.rst);
        if (e == hipSuccess) e = hipStreamSynchronize(S.rst);
        if (e != hipSuccess) { fail(MFX_E_HIP, "routing failed", e); return; }
        if (hcur[ndev] == 0) {
          for (uint32_t o = 0; o < ndev; ++o) { S.gn[b2][o] = hcur[o]; S.gkeys[b2][o] = S.d_keys[b2] + (size_t)o * region_cap; S.gctg[b2][o] = S.d_ctg[b2] + (size_t)o * region_cap; }
          return;
        }
        // an owner's region overflowed: this round of this slot again, by the exact counting split into a packed buffer.  (The one-pass
        // kernel counted the round's k-mers into kasm already: the split's own count is taken back out below.)
        if (!S.d_pkeys) {
          e = hipMalloc((void **)&S.d_pkeys, 2 * cap * 8);
          if (e == hipSuccess) e = hipMalloc((void **)&S.d_pctg, 2 * cap * 4);
          if (e != hipSuccess) { fail(MFX_E_NOMEM, "no memory for the exact re-routing of an unbalanced round", e); return; }
        }
        // kasm (global + per contig) would be counted twice: the exact split counts into a scratch image instead
        uint64_t *scratch = nullptr;
        e = hipMalloc((void **)&scratch, words * sizeof(uint64_t));
        if (e == hipSuccess) e = hipMemsetAsync(scratch, 0, words * sizeof(uint64_t), S.rst);
        a.counts = scratch;
        uint64_t *pk = S.d_pkeys + (size_t)b2 * cap;
        uint32_t *pc = S.d_pctg + (size_t)b2 * cap;
        if (e == hipSuccess) e = mfx_k_route_split(a, pk, pc, S.rst);
        uint64_t hd[MFX_SPLIT_MAX_RANKS] = {0};
        if (e == hipSuccess) e = hipMemcpyAsync(hd, R->d_dest, ndev * 8, hipMemcpyDeviceToHost, S.rst);
        if (e == hipSuccess) e = hipStreamSynchronize(S.rst);
        if (scratch) (void)hipFree(scratch);
        if (e != hipSuccess) { fail(MFX_E_HIP, "exact re-routing failed", e); return; }
        uint64_t at = 0;
        for (uint32_t o = 0; o < ndev; ++o) { S.gn[b2][o] = hd[o]; S.gkeys[b2][o] = pk + at; S.gctg[b2][o] = pc + at; at += hd[o]; }
      });
    for (auto &x : th) x.join();
    for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d)
      if (sl[d].rc) rc = mfx_fail(sl[d].rc, "slot %u: %s", d, sl[d].err.c_str());
  };
  if (rounds && rc == MFX_OK) route_round(0, 0);
  for (uint64_t r = 0; r < rounds && rc == MFX_OK; ++r) {
    const int b2 = (int)(r & 1);
    // ---- owners: one launch over the groups of all sources; a group on another device travels by peer copy first
    for (uint32_t o = 0; o < ndev && rc == MFX_OK; ++o) {
      DevGuard g(evs[o]->device);
      Slot &O = sl[o];
      mfx_hist_keys_args a;
      a.t = evs[o]->ix->view();
      a.ks.peak = evs[o]->peak; a.ks.n_prob = evs[o]->n_prob; a.ks.probK = evs[o]->d_probK; a.ks.probP = evs[o]->d_probP;
      a.ks.nbins = nbins; a.ks.ncontigs = ncontigs; a.ks.counts = O.d_counts; a.ks.partials = evs[o]->d_partials; a.ks.ovf = evs[o]->d_ovf;
      a.kfix = O.d_kfix;
      uint64_t at = 0, total = 0;
      for (uint32_t s2 = 0; s2 < ndev && rc == MFX_OK; ++s2) {
        const uint64_t n = sl[s2].gn[b2][o];
        if (!n) continue;
        const uint64_t *kp = sl[s2].gkeys[b2][o];
        const uint32_t *cp = sl[s2].gctg[b2][o];
        if (evs[s2]->device != evs[o]->device) {
          if (at + n > rcap) { rc = mfx_fail(MFX_E_FULL, "mfx_hist_run_sharded: owner %u receives more than %zu k-mers in one round", o, rcap); break; }
          hipError_t e = hipMemcpyPeerAsync(O.d_rkeys + at, evs[o]->device, kp, evs[s2]->device, n * 8, O.ost);
          if (e == hipSuccess) e = hipMemcpyPeerAsync(O.d_rctg + at, evs[o]->device, cp, evs[s2]->device, n * 4, O.ost);
          if (e != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "peer copy of %lu routed k-mers from slot %u to slot %u failed: %s", (unsigned long)n, s2, o, hipGetErrorString(e)); break; }
          kp = O.d_rkeys + at; cp = O.d_rctg + at;
          at += n;
        }
        a.seg_keys[a.nseg] = kp; a.seg_contig[a.nseg] = cp; a.seg_n[a.nseg] = n;
        ++a.nseg;
        total += n;
      }
      a.n = total;
      if (rc == MFX_OK && total) {
        hipError_t e = mfx_k_hist_keys(a, evs[o]->grid, O.ost);
        if (e != hipSuccess) rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_sharded: owner launch of slot %u failed: %s", o, hipGetErrorString(e));
      }
    }
    // ---- the next round is routed (into the other buffer set) while the owners evaluate this one
    if (r + 1 < rounds && rc == MFX_OK) route_round(r + 1, b2 ^ 1);
    // this round's groups are consumed before the round after the next overwrites their buffers
    for (uint32_t d = 0; d < ndev; ++d) {
      DevGuard g(evs[d]->device);
      if (hipStreamSynchronize(sl[d].ost) != hipSuccess && rc == MFX_OK) rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_sharded: slot %u failed: %s", d, hipGetErrorString(hipGetLastError()));
    }
  }
  std::vector<uint64_t> sum(words, 0), h(words);
  unsigned __int128 kfix = 0;
  for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) {
    DevGuard g(evs[d]->device);
    uint64_t kf[2] = {0, 0};
    if (hipMemcpy(h.data(), sl[d].d_counts, words * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(kf, sl[d].d_kfix, sizeof(kf), hipMemcpyDeviceToHost) != hipSuccess) { rc = mfx_fail(MFX_E_HIP, "mfx_hist_run_sharded: D2H of slot %u failed", d); break; }
    for (size_t i = 0; i < words; ++i) sum[i] += h[i];
    kfix += ((unsigned __int128)kf[1] << 64) | kf[0];           // integers: the order of the slots does not matter either
    sl[d].gn[0].assign(1, h[2ull * nbins + 2]);               // this slot's overflow count
  }
  // koverCpy = kfix * 2^-52 (the same three roundings whatever the run: deterministic)
  const double kover = ((double)(uint64_t)(kfix >> 64) * 18446744073709551616.0 + (double)(uint64_t)kfix) / 4503599627370496.0;
  if (rc == MFX_OK) rc = mfx_hist_result_from_counts(nbins, sum.data(), kover, ncontigs, out);
  if (rc == MFX_OK) {
    for (uint32_t d = 0; d < ndev && rc == MFX_OK; ++d) rc = result_take_overflow(evs[d], sl[d].gn[0][0], out);
    if (rc) mfx_hist_result_free(out);
  }
  release();
  return rc;
}

// ---------------------------------------------------------------------------
// -dump
// ---------------------------------------------------------------------------
extern "C" int mfx_dump_values(mfx_eval *ev, const mfx_seq *seq, uint32_t contig, uint64_t pos_begin, uint64_t pos_end,
                               uint32_t *readV, uint32_t *asmV, uint64_t *kasm, uint64_t *kmissing) {
  if (!ev || !seq || !readV || !asmV) return mfx_fail(MFX_E_INVAL, "mfx_dump_values: null argument");
  if (contig >= seq->ncontigs || pos_begin > pos_end || pos_end > seq->len[contig])
    return mfx_fail(MFX_E_INVAL, "mfx_dump_values: range [%lu,%lu) outside contig %u of length %lu",
                    (unsigned long)pos_begin, (unsigned long)pos_end, contig,
                    (unsigned long)(contig < seq->ncontigs ? seq->len[contig] : 0));
  if (int erc = mfx_seq_ensure_ascii(seq)) return erc;
  DevGuard g(ev->device);
  int canon = 0;
  int rc = index_canonical(ev->ix, &canon);
  if (rc) return rc;
  if ((rc = mfx_check_seq_of_index(ev->ix, seq, "-dump")) != MFX_OK) return rc;
  uint64_t n = pos_end - pos_begin;
  if (kasm) *kasm = 0;
  if (kmissing) *kmissing = 0;
  if (n == 0) return MFX_OK;
  uint64_t tb = pos_begin / MFX_TILE * MFX_TILE;     // tile-aligned start keeps the 16-byte loads aligned
  DevBuf<uint32_t> dr, da;
  DevBuf<uint64_t> ds;
  MFX_HIP(dr.alloc(n));
  MFX_HIP(da.alloc(n));
  MFX_HIP(ds.alloc(2));
  MFX_HIP(mfx_memset_now(ds.p, 0, 2 * sizeof(uint64_t)));
  mfx_dump_args a;
  a.t = ev->ix->view();
  a.canonical = canon;
  a.src = seq->d_bases + seq->off[contig] + tb;
  a.npos = pos_end - tb;
  a.skip = pos_begin - tb;
  a.clen_left = seq->len[contig] - tb;
  a.readV = dr.p;
  a.asmV = da.p;
  a.peak = ev->peak;
  a.n_prob = ev->n_prob;
  a.probK = ev->d_probK;
  a.probP = ev->d_probP;
  a.stats = ds.p;
  MFX_HIP(ev->ix->wide() ? mfx_kw_dump(a, nullptr) : mfx_k_dump(a, nullptr));
  MFX_HIP(hipMemcpy(readV, dr.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  MFX_HIP(hipMemcpy(asmV, da.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
  uint64_t st[2];
  MFX_HIP(hipMemcpy(st, ds.p, sizeof(st), hipMemcpyDeviceToHost));
  if (kasm) *kasm = st[0];
  if (kmissing) *kmissing = st[1];
  return MFX_OK;
}

// The same values over an index SHARDED across N evaluators (mfx_index_set_shard): every k-mer has one owner and the
// other shards answer 0, so each slot runs the lookup kernel over its copy of the sequence against its own shard and
// the N value arrays are ADDED on slot 0's device (peer copies, 8 B per position and slot).  kmissing is then
// counted from the summed values (readK == 0 is not additive over shards).
int mfx_score_paths(mfx_eval *ev, const char *text, uint64_t len, const mfx_path_table *pt, int need_dk, uint32_t *numM, double *totdk) {
  if (!ev || !pt || (len && !text) || (pt->npaths && (!numM || !pt->cfirst || (need_dk && !totdk)))) return mfx_fail(MFX_E_INVAL, "mfx_score_paths: null argument");
  if (pt->npaths == 0 || len == 0) return MFX_OK;
  mfx_seq *seq = mfx_seq_upload(ev->device, &text, &len, 1);
  if (!seq) return mfx_last_error_code() ? mfx_last_error_code() : MFX_E_HIP;
  struct Free { mfx_seq *s; ~Free() { mfx_seq_free(s); } } fr{seq};
  if (int erc = mfx_seq_ensure_ascii(seq)) return erc;
  DevGuard g(ev->device);
  int canon = 0;
  int rc = index_canonical(ev->ix, &canon);
  if (rc) return rc;
  if (ev->ix->seq_only && !ev->ix->paths_token) return mfx_fail(MFX_E_INVAL, "mfx_score_paths: a sequence-only index holds the k-mers of one sequence; alternative paths need the full index (or the path-only one: mfx_index_claim_paths)");
  DevBuf<uint32_t> dr, da, dlen, dnv, dvidx, dvlen, dnum;
  DevBuf<int32_t> dgt;
  DevBuf<uint64_t> ds, doff, dvoff, dcf;
  DevBuf<double> ddk;
  const uint64_t np = pt->npaths, nvl = pt->nvals ? pt->nvals : 1;
  MFX_HIP(dr.alloc(len)); MFX_HIP(da.alloc(len)); MFX_HIP(ds.alloc(2));
  MFX_HIP(doff.alloc(np)); MFX_HIP(dlen.alloc(np)); MFX_HIP(dnv.alloc(np)); MFX_HIP(dvoff.alloc(np)); MFX_HIP(dcf.alloc(np));
  MFX_HIP(dgt.alloc(nvl)); MFX_HIP(dvidx.alloc(nvl)); MFX_HIP(dvlen.alloc(nvl));
  MFX_HIP(dnum.alloc(np)); MFX_HIP(ddk.alloc(need_dk ? np : 1));
  hipStream_t st = nullptr;
  MFX_HIP(hipMemsetAsync(ds.p, 0, 2 * sizeof(uint64_t), st));
  MFX_HIP(hipMemcpyAsync(doff.p, pt->off, np * 8, hipMemcpyHostToDevice, st));
  MFX_HIP(hipMemcpyAsync(dlen.p, pt->len, np * 4, hipMemcpyHostToDevice, st));
  MFX_HIP(hipMemcpyAsync(dnv.p, pt->nv, np * 4, hipMemcpyHostToDevice, st));
  MFX_HIP(hipMemcpyAsync(dvoff.p, pt->voff, np * 8, hipMemcpyHostToDevice, st));
  MFX_HIP(hipMemcpyAsync(dcf.p, pt->cfirst, np * 8, hipMemcpyHostToDevice, st));
  if (pt->nvals) {
    MFX_HIP(hipMemcpyAsync(dgt.p, pt->gt, pt->nvals * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvidx.p, pt->vidx, pt->nvals * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvlen.p, pt->vlen, pt->nvals * 4, hipMemcpyHostToDevice, st));
  }
  mfx_dump_args a;
  a.t = ev->ix->view();
  a.canonical = canon;
  a.src = seq->d_bases + seq->off[0];
  a.npos = len;
  a.skip = 0;
  a.clen_left = len;
  a.readV = dr.p;
  a.asmV = da.p;
  a.peak = ev->peak;
  a.n_prob = ev->n_prob;
  a.probK = ev->d_probK;
  a.probP = ev->d_probP;
  a.stats = ds.p;
  MFX_HIP(ev->ix->wide() ? mfx_kw_dump(a, st) : mfx_k_dump(a, st));
  mfx_var_score_args sa;
  sa.text = seq->d_bases + seq->off[0];
  sa.readV = dr.p; sa.asmV = da.p;
  sa.npaths = np;
  sa.off = doff.p; sa.len = dlen.p; sa.nv = dnv.p; sa.voff = dvoff.p; sa.cfirst = dcf.p;
  sa.gt = dgt.p; sa.vidx = dvidx.p; sa.vlen = dvlen.p;
  sa.k = (uint32_t)ev->ix->k;
  sa.need_dk = need_dk ? 1 : 0;
  sa.peak = ev->peak; sa.n_prob = ev->n_prob; sa.probK = ev->d_probK; sa.probP = ev->d_probP;
  sa.numM = dnum.p; sa.totdk = ddk.p;
  MFX_HIP(mfx_k_var_score(sa, st));
  MFX_HIP(hipMemcpy(numM, dnum.p, np * 4, hipMemcpyDeviceToHost));
  if (need_dk) MFX_HIP(hipMemcpy(totdk, ddk.p, np * 8, hipMemcpyDeviceToHost));
  return MFX_OK;
}

int mfx_score_paths_trv(mfx_eval *ev, const char *text, uint64_t len, const mfx_path_table *pt, const mfx_trv_batch *tb, int need_dk, uint32_t *numM, double *totdk) {
  if (!ev || !pt || !tb || (len && !text) || !numM || (need_dk && !totdk) || (tb->ncl && (!tb->cl || !tb->var || !tb->al || !tb->np || !tb->status || !tb->p_len || !tb->gt)))
    return mfx_fail(MFX_E_INVAL, "mfx_score_paths_trv: null argument");
  if (ev->ix->seq_only && !ev->ix->paths_token) return mfx_fail(MFX_E_INVAL, "mfx_score_paths: a sequence-only index holds the k-mers of one sequence; alternative paths need the full index (or the path-only one: mfx_index_claim_paths)");
  const uint64_t hp = pt->npaths, hv = pt->nvals, NP = hp + tb->path_cap, NV = hv + tb->row_cap;
  const uint64_t total = std::max<uint64_t>(tb->text_end, len);
  if (NP == 0 || total == 0) return MFX_OK;
  DevGuard g(ev->device);
  int canon = 0;
  int rc = index_canonical(ev->ix, &canon);
  if (rc) return rc;
  // the batch's text on the device: the host's paths copied in, the rest '\n' (no k-mer) until the traverse kernel writes its paths; the
  // tile loads of the lookup kernel reach a tile past the end.  Every array of this call is a piece of ONE allocation that the evaluator
  // keeps from batch to batch (a run scores ~40 batches of the same size: ~20 allocations and as many frees -- each a device-wide wait --
  // per batch were a third of a batch's 7 ms).
  const uint64_t text_bytes = ((total + MFX_TILE - 1) / MFX_TILE + 2) * MFX_TILE + 256;
  std::lock_guard<std::mutex> scratch_lock(ev->var_scratch_mu);
  uint64_t need = 0;
  auto piece = [&](uint64_t bytes) { const uint64_t at = need; need += (bytes + 255) & ~255ull; return at; };
  const uint64_t o_text = piece(text_bytes), o_r = piece(total * 4), o_a = piece(total * 4), o_s = piece(16), o_off = piece(NP * 8), o_len = piece(NP * 4), o_nv = piece(NP * 4),
                 o_voff = piece(NP * 8), o_cf = piece(NP * 8), o_gt = piece((NV ? NV : 1) * 4), o_vidx = piece((NV ? NV : 1) * 4), o_vlen = piece((NV ? NV : 1) * 4),
                 o_num = piece(NP * 4), o_dk = piece((need_dk ? NP : 1) * 8), o_cl = piece(tb->ncl * sizeof(mfx_trv_cluster)), o_var = piece(tb->nvar * sizeof(mfx_trv_variant)),
                 o_all = piece(tb->nal * sizeof(mfx_trv_allele)), o_win = piece(tb->win_bytes), o_alt = piece(tb->al_bytes), o_np = piece(tb->ncl * 4), o_st = piece(tb->ncl * 4);
  if (need > ev->var_scratch_bytes) {
    if (ev->d_var_scratch) { (void)hipFree(ev->d_var_scratch); ev->d_var_scratch = nullptr; ev->var_scratch_bytes = 0; }
    const uint64_t want = need + need / 4;
    if (hipMalloc((void **)&ev->d_var_scratch, want) != hipSuccess) { (void)hipGetLastError(); return mfx_fail(MFX_E_NOMEM, "mfx_score_paths: no device memory for a batch of paths (%.1f GB)", want / 1e9); }
    ev->var_scratch_bytes = want;
  }
  uint8_t *const B = ev->d_var_scratch;
  struct P8 { uint8_t *p; } dtext{B + o_text}, dwin{B + o_win}, dal{B + o_alt};
  struct PC { mfx_trv_cluster *p; } dcl{reinterpret_cast<mfx_trv_cluster *>(B + o_cl)};
  struct PV { mfx_trv_variant *p; } dvar{reinterpret_cast<mfx_trv_variant *>(B + o_var)};
  struct PA { mfx_trv_allele *p; } dall{reinterpret_cast<mfx_trv_allele *>(B + o_all)};
  struct P32 { uint32_t *p; } dr{reinterpret_cast<uint32_t *>(B + o_r)}, da{reinterpret_cast<uint32_t *>(B + o_a)}, dlen{reinterpret_cast<uint32_t *>(B + o_len)},
      dnv{reinterpret_cast<uint32_t *>(B + o_nv)}, dvidx{reinterpret_cast<uint32_t *>(B + o_vidx)}, dvlen{reinterpret_cast<uint32_t *>(B + o_vlen)},
      dnum{reinterpret_cast<uint32_t *>(B + o_num)}, dnp{reinterpret_cast<uint32_t *>(B + o_np)}, dst{reinterpret_cast<uint32_t *>(B + o_st)};
  struct PI { int32_t *p; } dgt{reinterpret_cast<int32_t *>(B + o_gt)};
  struct P64 { uint64_t *p; } ds{reinterpret_cast<uint64_t *>(B + o_s)}, doff{reinterpret_cast<uint64_t *>(B + o_off)}, dvoff{reinterpret_cast<uint64_t *>(B + o_voff)},
      dcf{reinterpret_cast<uint64_t *>(B + o_cf)};
  struct PD { double *p; } ddk{reinterpret_cast<double *>(B + o_dk)};
  hipStream_t st = nullptr;
  MFX_HIP(mfx_memset_now(dtext.p + len, '\n', text_bytes - len));
  MFX_HIP(hipMemsetAsync(ds.p, 0, 2 * sizeof(uint64_t), st));
  if (len) MFX_HIP(hipMemcpyAsync(dtext.p, text, len, hipMemcpyHostToDevice, st));
  if (hp) {
    MFX_HIP(hipMemcpyAsync(doff.p, pt->off, hp * 8, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dlen.p, pt->len, hp * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dnv.p, pt->nv, hp * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvoff.p, pt->voff, hp * 8, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dcf.p, pt->cfirst, hp * 8, hipMemcpyHostToDevice, st));
  }
  if (hv) {
    MFX_HIP(hipMemcpyAsync(dgt.p, pt->gt, hv * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvidx.p, pt->vidx, hv * 4, hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvlen.p, pt->vlen, hv * 4, hipMemcpyHostToDevice, st));
  }
  if (tb->ncl) {
    MFX_HIP(hipMemcpyAsync(dcl.p, tb->cl, tb->ncl * sizeof(mfx_trv_cluster), hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(dvar.p, tb->var, tb->nvar * sizeof(mfx_trv_variant), hipMemcpyHostToDevice, st));
    if (tb->nal) MFX_HIP(hipMemcpyAsync(dall.p, tb->al, tb->nal * sizeof(mfx_trv_allele), hipMemcpyHostToDevice, st));
    if (tb->win_bytes) MFX_HIP(hipMemcpyAsync(dwin.p, tb->win_text, tb->win_bytes, hipMemcpyHostToDevice, st));
    if (tb->al_bytes) MFX_HIP(hipMemcpyAsync(dal.p, tb->al_text, tb->al_bytes, hipMemcpyHostToDevice, st));
    mfx_trv_out o;
    o.text = reinterpret_cast<char *>(dtext.p);
    o.p_off = doff.p + hp; o.p_voff = dvoff.p + hp; o.p_cfirst = dcf.p + hp; o.p_len = dlen.p + hp; o.p_nv = dnv.p + hp;
    o.gt = dgt.p + hv; o.vidx = dvidx.p + hv; o.vlen = dvlen.p + hv;
    o.table_base = hp; o.row_base = hv;
    MFX_HIP(mfx_k_var_traverse(dcl.p, tb->ncl, dvar.p, dall.p, reinterpret_cast<const char *>(dwin.p), reinterpret_cast<const char *>(dal.p), o, dnp.p, dst.p, st));
  }
  mfx_dump_args a;
  a.t = ev->ix->view();
  a.canonical = canon;
  a.src = dtext.p;
  a.npos = total;
  a.skip = 0;
  a.clen_left = total;
  a.readV = dr.p;
  a.asmV = da.p;
  a.peak = ev->peak;
  a.n_prob = ev->n_prob;
  a.probK = ev->d_probK;
  a.probP = ev->d_probP;
  a.stats = ds.p;
  MFX_HIP(ev->ix->wide() ? mfx_kw_dump(a, st) : mfx_k_dump(a, st));
  mfx_var_score_args sa;
  sa.text = dtext.p;
  sa.readV = dr.p; sa.asmV = da.p;
  sa.npaths = NP;
  sa.off = doff.p; sa.len = dlen.p; sa.nv = dnv.p; sa.voff = dvoff.p; sa.cfirst = dcf.p;
  sa.gt = dgt.p; sa.vidx = dvidx.p; sa.vlen = dvlen.p;
  sa.k = (uint32_t)ev->ix->k;
  sa.need_dk = need_dk ? 1 : 0;
  sa.peak = ev->peak; sa.n_prob = ev->n_prob; sa.probK = ev->d_probK; sa.probP = ev->d_probP;
  sa.numM = dnum.p; sa.totdk = ddk.p;
  MFX_HIP(mfx_k_var_score(sa, st));
  MFX_HIP(hipMemcpy(numM, dnum.p, NP * 4, hipMemcpyDeviceToHost));
  if (need_dk) MFX_HIP(hipMemcpy(totdk, ddk.p, NP * 8, hipMemcpyDeviceToHost));
  if (tb->ncl) {
    MFX_HIP(hipMemcpy(tb->np, dnp.p, tb->ncl * 4, hipMemcpyDeviceToHost));
    MFX_HIP(hipMemcpy(tb->status, dst.p, tb->ncl * 4, hipMemcpyDeviceToHost));
    if (tb->path_cap) MFX_HIP(hipMemcpy(tb->p_len, dlen.p + hp, tb->path_cap * 4, hipMemcpyDeviceToHost));
    if (tb->row_cap) MFX_HIP(hipMemcpy(tb->gt, dgt.p + hv, tb->row_cap * 4, hipMemcpyDeviceToHost));
  }
  return MFX_OK;
}

int mfx_claim_paths_batch(mfx_index *ix, uint8_t **scratch, uint64_t *scratch_bytes, const char *text, uint64_t len, const mfx_trv_batch *tb, uint64_t *bad) {
  if (!ix || !scratch || !scratch_bytes || (len && !text) || (tb && tb->ncl && (!tb->cl || !tb->var || !tb->al || !tb->status)))
    return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: null argument");
  if (!ix->seq_only || ix->wide()) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: not a sequence-only index of k <= %d (mfx_index_create_for_seq)", MFX_MAX_K_NARROW);
  if (ix->frozen) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: this index already took counts; its k-mers must all be claimed before the first add / load");
  if (bad) *bad = 0;
  const uint64_t ncl = tb ? tb->ncl : 0, NP = tb ? tb->path_cap : 0, NV = tb ? tb->row_cap : 0;
  const uint64_t total = std::max<uint64_t>(tb ? tb->text_end : 0, len);
  if (total == 0) return MFX_OK;
  DevGuard g(ix->device);
  const uint64_t ntiles = (total + MFX_TILE - 1) / MFX_TILE;
  const uint64_t text_bytes = (ntiles + 2) * MFX_TILE + 256;          // (the tile loads of the claim kernel reach a tile past the end)
  uint64_t need = 0;
  auto piece = [&](uint64_t bytes) { const uint64_t at = need; need += (bytes + 255) & ~255ull; return at; };
  const uint64_t o_text = piece(text_bytes), o_geo = piece(4 * 8), o_off = piece((NP ? NP : 1) * 8), o_len = piece((NP ? NP : 1) * 4), o_nv = piece((NP ? NP : 1) * 4),
                 o_voff = piece((NP ? NP : 1) * 8), o_cf = piece((NP ? NP : 1) * 8), o_gt = piece((NV ? NV : 1) * 4), o_vidx = piece((NV ? NV : 1) * 4), o_vlen = piece((NV ? NV : 1) * 4),
                 o_cl = piece((ncl ? ncl : 1) * sizeof(mfx_trv_cluster)), o_var = piece((tb ? tb->nvar : 0) * sizeof(mfx_trv_variant) + 8),
                 o_all = piece((tb ? tb->nal : 0) * sizeof(mfx_trv_allele) + 8), o_win = piece((tb ? tb->win_bytes : 0) + 8), o_alt = piece((tb ? tb->al_bytes : 0) + 8),
                 o_np = piece((ncl ? ncl : 1) * 4), o_st = piece((ncl ? ncl : 1) * 4);
  if (need > *scratch_bytes) {
    if (*scratch) { (void)hipFree(*scratch); *scratch = nullptr; *scratch_bytes = 0; }
    const uint64_t want = need + need / 4;
    if (hipMalloc((void **)scratch, want) != hipSuccess) { (void)hipGetLastError(); *scratch = nullptr; return mfx_fail(MFX_E_NOMEM, "mfx_index_claim_paths: no device memory for a batch of paths (%.1f GB)", want / 1e9); }
    *scratch_bytes = want;
  }
  uint8_t *const B = *scratch;
  hipStream_t st = nullptr;
  MFX_HIP(mfx_memset_now(B + o_text + len, '\n', text_bytes - len));
  if (len) MFX_HIP(hipMemcpyAsync(B + o_text, text, len, hipMemcpyHostToDevice, st));
  const uint64_t geo[4] = {0, total, 0, ntiles};                        // contig_off[1], contig_len[1], tile_start[2] of the one contig the text is
  MFX_HIP(hipMemcpyAsync(B + o_geo, geo, sizeof(geo), hipMemcpyHostToDevice, st));
  if (ncl) {
    MFX_HIP(hipMemcpyAsync(B + o_cl, tb->cl, ncl * sizeof(mfx_trv_cluster), hipMemcpyHostToDevice, st));
    MFX_HIP(hipMemcpyAsync(B + o_var, tb->var, tb->nvar * sizeof(mfx_trv_variant), hipMemcpyHostToDevice, st));
    if (tb->nal) MFX_HIP(hipMemcpyAsync(B + o_all, tb->al, tb->nal * sizeof(mfx_trv_allele), hipMemcpyHostToDevice, st));
    if (tb->win_bytes) MFX_HIP(hipMemcpyAsync(B + o_win, tb->win_text, tb->win_bytes, hipMemcpyHostToDevice, st));
    if (tb->al_bytes) MFX_HIP(hipMemcpyAsync(B + o_alt, tb->al_text, tb->al_bytes, hipMemcpyHostToDevice, st));
    mfx_trv_out o;
    o.text = reinterpret_cast<char *>(B + o_text);
    o.p_off = reinterpret_cast<uint64_t *>(B + o_off); o.p_voff = reinterpret_cast<uint64_t *>(B + o_voff); o.p_cfirst = reinterpret_cast<uint64_t *>(B + o_cf);
    o.p_len = reinterpret_cast<uint32_t *>(B + o_len); o.p_nv = reinterpret_cast<uint32_t *>(B + o_nv);
    o.gt = reinterpret_cast<int32_t *>(B + o_gt); o.vidx = reinterpret_cast<uint32_t *>(B + o_vidx); o.vlen = reinterpret_cast<uint32_t *>(B + o_vlen);
    o.table_base = 0; o.row_base = 0;
    MFX_HIP(mfx_k_var_traverse(reinterpret_cast<const mfx_trv_cluster *>(B + o_cl), ncl, reinterpret_cast<const mfx_trv_variant *>(B + o_var),
                               reinterpret_cast<const mfx_trv_allele *>(B + o_all), reinterpret_cast<const char *>(B + o_win), reinterpret_cast<const char *>(B + o_alt), o,
                               reinterpret_cast<uint32_t *>(B + o_np), reinterpret_cast<uint32_t *>(B + o_st), st));
  }
  mfx_count_args a;
  a.t = ix->view();
  a.bases = B + o_text;
  const uint64_t *geo_d = reinterpret_cast<const uint64_t *>(B + o_geo);
  a.contig_off = geo_d; a.contig_len = geo_d + 1; a.tile_start = geo_d + 2;
  a.ncontigs = 1;
  a.ntiles = ntiles;
  a.meta = ix->d_meta;
  a.count = 0;
  MFX_HIP(mfx_k_count(a, st));
  if (ncl) {
    MFX_HIP(hipMemcpy(tb->status, B + o_st, ncl * 4, hipMemcpyDeviceToHost));
    uint64_t nb = 0;
    for (uint64_t c = 0; c < ncl; ++c) nb += tb->status[c] != MFX_TRV_OK;
    if (bad) *bad = nb;
  } else MFX_HIP(hipStreamSynchronize(st));                             // (the host's text may go once this returns)
  return MFX_OK;
}
int mfx_claim_paths_finish(mfx_index *ix, uint64_t token) {
  if (!ix) return mfx_fail(MFX_E_INVAL, "mfx_index_claim_paths: null argument");
  DevGuard g(ix->device);
  MFX_HIP(hipDeviceSynchronize());
  if (int rc = index_check(ix)) return rc;
  ix->paths_token = token ? token : 1;
  ix->seq_digest = (uint32_t)(token >> 32) | 0x80000000u;              // (no sequence's k-mers: -hist / -dump of a sequence on this index are refused, mfx_check_seq_of_index)
  return MFX_OK;
}
void mfx_claim_paths_release(int device, uint8_t *scratch) {
  if (!scratch) return;
  DevGuard g(device);
  (void)hipFree(scratch);
}

extern "C" int mfx_dump_values_sharded(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t nslots, uint32_t contig,
                                       uint64_t pos_begin, uint64_t pos_end, uint32_t *readV, uint32_t *asmV,
                                       uint64_t *kasm, uint64_t *kmissing) {
  if (!evs || !seqs || nslots == 0 || !readV || !asmV) return mfx_fail(MFX_E_INVAL, "mfx_dump_values_sharded: null argument");
  for (uint32_t d = 0; d < nslots; ++d) {
    if (!evs[d] || !seqs[d]) return mfx_fail(MFX_E_INVAL, "mfx_dump_values_sharded: slot %u is null", d);
    const mfx_index *ix = evs[d]->ix;
    if (ix->shard_n != nslots || ix->shard_rank != d)
      return mfx_fail(MFX_E_INVAL, "slot %u: its index must be shard %u of %u (mfx_index_set_shard)", d, d, nslots);
    if (ix->wide()) return mfx_fail(MFX_E_INVAL, "a sharded index handles k <= 31");
    if (evs[d]->device != seqs[d]->device || seqs[d]->ncontigs != seqs[0]->ncontigs || contig >= seqs[d]->ncontigs ||
        seqs[d]->len[contig] != seqs[0]->len[contig])
      return mfx_fail(MFX_E_INVAL, "slot %u: the sequences must be copies of one another, each on its evaluator's device", d);
    if (int erc = mfx_seq_ensure_ascii(seqs[d])) return erc;
  }
  const mfx_seq *seq0 = seqs[0];
  if (pos_begin > pos_end || pos_end > seq0->len[contig])
    return mfx_fail(MFX_E_INVAL, "mfx_dump_values_sharded: range [%lu,%lu) outside contig %u of length %lu",
                    (unsigned long)pos_begin, (unsigned long)pos_end, contig, (unsigned long)seq0->len[contig]);
  const uint64_t n = pos_end - pos_begin;
  if (kasm) *kasm = 0;
  if (kmissing) *kmissing = 0;
  if (n == 0) return MFX_OK;
  const uint64_t tb = pos_begin / MFX_TILE * MFX_TILE;
  struct Slot { uint32_t *dr = nullptr, *da = nullptr; uint64_t *ds = nullptr; };
  std::vector<Slot> sl(nslots);
  uint32_t *tr = nullptr, *ta = nullptr;                       // slot 0's landing buffers for a peer's arrays
  auto release = [&]() {
    for (uint32_t d = 0; d < nslots; ++d) {
      DevGuard g(evs[d]->device);
      if (sl[d].dr) (void)hipFree(sl[d].dr);
      if (sl[d].da) (void)hipFree(sl[d].da);
      if (sl[d].ds) (void)hipFree(sl[d].ds);
    }
    DevGuard g(evs[0]->device);
    if (tr) (void)hipFree(tr);
    if (ta) (void)hipFree(ta);
  };
  auto args_of = [&](uint32_t d) {
    mfx_dump_args a;
    a.t = evs[d]->ix->view();
    a.canonical = 0;
    a.src = seqs[d]->d_bases + seqs[d]->off[contig] + tb;
    a.npos = pos_end - tb;
    a.skip = pos_begin - tb;
    a.clen_left = seqs[d]->len[contig] - tb;
    a.readV = sl[d].dr;
    a.asmV = sl[d].da;
    a.peak = evs[d]->peak;
    a.n_prob = evs[d]->n_prob;
    a.probK = evs[d]->d_probK;
    a.probP = evs[d]->d_probP;
    a.stats = sl[d].ds;
    return a;
  };
  int rc = MFX_OK;
  hipError_t e = hipSuccess;
  for (uint32_t d = 0; d < nslots && rc == MFX_OK && e == hipSuccess; ++d) {          // every shard's lookup, all devices at once
    DevGuard g(evs[d]->device);
    int canon = 0;
    rc = index_canonical(evs[d]->ix, &canon);
    if (rc) break;
    e = hipMalloc((void **)&sl[d].dr, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&sl[d].da, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&sl[d].ds, 2 * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemsetAsync(sl[d].ds, 0, 2 * sizeof(uint64_t), nullptr);
    if (e != hipSuccess) break;
    mfx_dump_args a = args_of(d);
    a.canonical = canon;
    e = mfx_k_dump(a, nullptr);
  }
  for (uint32_t d = 0; d < nslots && rc == MFX_OK && e == hipSuccess; ++d) {
    DevGuard g(evs[d]->device);
    e = hipDeviceSynchronize();
  }
  if (rc == MFX_OK && e == hipSuccess && nslots > 1) {
    DevGuard g(evs[0]->device);
    e = hipMalloc((void **)&tr, n * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ta, n * sizeof(uint32_t));
    for (uint32_t d = 1; d < nslots && e == hipSuccess; ++d) {
      if (evs[d]->device == evs[0]->device) {
        e = mfx_k_add_u32(sl[0].dr, sl[d].dr, n, nullptr);
        if (e == hipSuccess) e = mfx_k_add_u32(sl[0].da, sl[d].da, n, nullptr);
        continue;
      }
      e = hipMemcpyPeerAsync(tr, evs[0]->device, sl[d].dr, evs[d]->device, n * sizeof(uint32_t), nullptr);
      if (e == hipSuccess) e = hipMemcpyPeerAsync(ta, evs[0]->device, sl[d].da, evs[d]->device, n * sizeof(uint32_t), nullptr);
      if (e == hipSuccess) e = mfx_k_add_u32(sl[0].dr, tr, n, nullptr);
      if (e == hipSuccess) e = mfx_k_add_u32(sl[0].da, ta, n, nullptr);
    }
    if (e == hipSuccess) e = hipMemsetAsync(sl[0].ds, 0, 2 * sizeof(uint64_t), nullptr);
    if (e == hipSuccess) {
      mfx_dump_args a = args_of(0);
      a.recount = 1;
      e = mfx_k_dump(a, nullptr);
    }
  }
  if (rc == MFX_OK && e == hipSuccess) {
    DevGuard g(evs[0]->device);
    uint64_t st[2] = {0, 0};
    e = hipMemcpy(readV, sl[0].dr, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(asmV, sl[0].da, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(st, sl[0].ds, sizeof(st), hipMemcpyDeviceToHost);
    if (kasm) *kasm = st[0];
    if (kmissing) *kmissing = st[1];
  }
  release();
  if (rc == MFX_OK && e != hipSuccess) rc = mfx_fail(MFX_E_HIP, "mfx_dump_values_sharded: %s", hipGetErrorString(e));
  return rc;
}

// host threads the library may use for text formatting: min(hardware, cgroup quota, 64)
static thread_local unsigned t_sharers = 1;
extern "C" void mfx_host_threads_share(unsigned nsharers) { t_sharers = nsharers ? nsharers : 1; }
static unsigned t_sharers_get() { return t_sharers; }

static unsigned host_threads_total();
unsigned mfx_host_threads() { return std::max(1u, host_threads_total() / t_sharers); }

static unsigned host_threads_total() {
  const char *e = getenv("MFX_HOST_THREADS");
  if (e && atoi(e) > 0) return (unsigned)atoi(e);
  unsigned n = std::thread::hardware_concurrency();
  if (n == 0) n = 1;
  if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64], per[64];
    if (fscanf(f, "%63s %63s", q, per) == 2 && strcmp(q, "max") != 0 && atof(per) > 0) {
      unsigned lim = (unsigned)(atof(q) / atof(per));
      if (lim >= 1 && lim < n) n = lim;
    }
    fclose(f);
  }
  return std::min(n, 64u);
}

// Formats the lines of outputDump (merfin-dump.C:87-93) for positions
// [o, o+cnt) into `out`.  The text after the position depends only on the
// (readV, asmV) pair, so it is produced once per distinct pair by the very
// same printf("%.2f") the reference uses and then reused: the common pairs (read count < 256, assembly count < 16)
// from a table indexed by the pair, the others from a map.
namespace {
struct DumpTails {
  struct Small { char n = -1; char t[39]; };                // n: -1 = not made yet, 0 = no line for this pair
  std::vector<Small> lut;                                   // [rv < 256][av < 16]
  std::unordered_map<uint64_t, std::string> rest;
  const mfx_kparams &kp;
  explicit DumpTails(const mfx_kparams &k) : lut(256 * 16), kp(k) {}
  // the text after the position ("\t%.2f\t%.2f\t%.2f\n"), or n = 0 when all three values are zero
  int make(uint32_t rv, uint32_t av, char *buf, size_t cap) const {
    double readK, asmK, prob;
    mfx_getK(&kp, rv, av, &readK, &asmK, &prob);
    const double km = mfx_kmetric(readK, asmK);
    if (!((readK != 0.0) || (asmK != 0.0) || (km != 0.0))) return 0;
    return snprintf(buf, cap, "\t%.2f\t%.2f\t%.2f\n", readK, asmK, km);
  }
  const char *get(uint32_t rv, uint32_t av, size_t &n) {
    if (rv < 256u && av < 16u) {
      Small &e = lut[rv * 16u + av];
      if (e.n < 0) {
        char buf[128];
        const int m = make(rv, av, buf, sizeof(buf));
        if (m >= 0 && m < (int)sizeof(e.t)) { memcpy(e.t, buf, (size_t)m); e.n = (char)m; }
        else { n = 0; return nullptr; }                      // (cannot happen for these magnitudes; fall through to the map)
      }
      n = (size_t)e.n;
      return e.t;
    }
    const uint64_t key = ((uint64_t)rv << 32) | av;
    auto it = rest.find(key);
    if (it == rest.end()) {
      char buf[128];
      const int m = make(rv, av, buf, sizeof(buf));
      it = rest.emplace(key, std::string(buf, (size_t)std::max(m, 0))).first;
    }
    n = it->second.size();
    return it->second.data();
  }
};
}  // namespace

namespace {
struct RawBuf {                                             // a byte buffer that is never value-initialised (tens of MB per thread and chunk)
  char *p = nullptr;
  size_t cap = 0;
  char *data() { return p; }
  size_t size() const { return cap; }
  void resize(size_t n) {                                   // contents kept
    char *q = (char *)realloc(p, n);
    if (!q) throw std::bad_alloc();
    p = q; cap = n;
  }
  RawBuf() = default;
  RawBuf(const RawBuf &) = delete;
  RawBuf &operator=(const RawBuf &) = delete;
  ~RawBuf() { free(p); }
};
}  // namespace

static void dump_format_range(const mfx_kparams &kp, const char *name, size_t name_len, uint64_t o, uint64_t cnt,
                              const uint32_t *rv, const uint32_t *av, RawBuf &out, size_t &used) {
  DumpTails tails(kp);
  // the longest line: name, tab, 20 digits, tail (< 128)
  const size_t worst = name_len + 1 + 20 + 128;
  if (out.size() < cnt * 32 + worst) out.resize(cnt * 32 + worst);
  char *w = out.data(), *lim = out.data() + out.size() - worst;
  char num[24];
  for (uint64_t i = 0; i < cnt; ++i) {
    if (rv[i] == 0 && av[i] == 0) continue;            // readK = asmK = K* = 0: no line
    size_t tn = 0;
    const char *t = tails.get(rv[i], av[i], tn);
    if (tn == 0) continue;
    if (w > lim) {                                       // (lines longer than the estimate: long names)
      const size_t at = (size_t)(w - out.data());
      out.resize(out.size() * 2 + worst);
      w = out.data() + at;
      lim = out.data() + out.size() - worst;
    }
    uint64_t pos = o + i;
    int d = 0;
    do { num[d++] = (char)('0' + pos % 10); pos /= 10; } while (pos);
    memcpy(w, name, name_len);
    w += name_len;
    *w++ = '\t';
    while (d) *w++ = num[--d];
    memcpy(w, t, tn);
    w += tn;
  }
  used = (size_t)(w - out.data());
}

// outputDump, merfin-dump.C:87-93: a line for every position where any of
// readK, asmK, K* is non-zero.  Values come from the GPU in 16 M-position
// chunks -- chunk i + 1 is looked up while chunk i is formatted and written --; host threads format disjoint sub-ranges and,
// for a plain file, write them themselves at the offsets the sizes of the ranges before them give (one writer moved 2.2 GB of
// text at 2.5 GB/s: 0.9 of the 1.06 s a config-2 dump took); a compressed output goes through its pipe in order.
// values(o, e, rv, av, &kasm, &kmissing) fills the (readV, asmV) pairs of positions [o, e): one evaluator or the shards of one index
template <class Values>
static int dump_contig_impl(const mfx_eval *ev, const mfx_seq *seq, uint32_t contig, const char *name,
                            const char *path, int append, uint64_t *kasm, uint64_t *kmissing, Values values) {
  const bool timing = getenv("MFX_DUMP_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_values = 0, t_format = 0, t_write = 0;
  const bool plain = mfx_suffix_tool(path) == nullptr && !(getenv("MFX_DUMP_SERIAL") && atoi(getenv("MFX_DUMP_SERIAL")));
  mfx_file fh;
  FILE *f = nullptr;
  int fd = -1;
  uint64_t file_off = 0;
  if (plain) {
    fd = open(path, O_WRONLY | O_CREAT | O_CLOEXEC | (append ? 0 : O_TRUNC), 0666);
    if (fd < 0) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", path);
    if (append) { const off_t end = lseek(fd, 0, SEEK_END); file_off = end > 0 ? (uint64_t)end : 0; }
  } else {
    fh = mfx_open_writer(path, append != 0);
    f = fh.f;
    if (!f) return mfx_fail(MFX_E_IO, "cannot open '%s' for writing", path);
  }
  const uint64_t len = seq->len[contig];
  const uint64_t CH = 1ull << 24;
  struct U32Buf { std::unique_ptr<uint32_t[]> p; uint32_t *data() { return p.get(); } } rvb[2], avb[2];     // (not value-initialised)
  for (int b = 0; b < 2; ++b) { rvb[b].p.reset(new uint32_t[std::min(len, CH) + 1]); avb[b].p.reset(new uint32_t[std::min(len, CH) + 1]); }
  mfx_kparams kp{ev->peak, ev->n_prob, ev->probK.data(), ev->probP.data()};
  const size_t name_len = strlen(name);
  const unsigned nthr = mfx_host_threads();
  std::vector<RawBuf> parts(nthr);
  std::vector<size_t> used(nthr, 0);
  uint64_t ka = 0, km = 0;
  int rc = MFX_OK;
  // the values of chunk c (positions [c * CH, ...)) into buffer c & 1, on a helper thread
  struct Fetch { std::thread th; int rc = MFX_OK; uint64_t a = 0, m = 0; std::string err; double dt = 0; } fetch;
  auto start_fetch = [&](uint64_t o) {
    fetch.rc = MFX_OK; fetch.a = fetch.m = 0;
    const int b = (int)((o / CH) & 1);
    fetch.th = std::thread([&, o, b]() {
      const double t0 = now();
      fetch.rc = values(o, std::min(len, o + CH), rvb[b].data(), avb[b].data(), &fetch.a, &fetch.m);
      if (fetch.rc) fetch.err = mfx_last_error();          // (errors are per thread: carried back to the caller's)
      fetch.dt = now() - t0;
    });
  };
  if (len) start_fetch(0);
  for (uint64_t o = 0; o < len && rc == MFX_OK; o += CH) {
    const uint64_t e = std::min(len, o + CH);
    const int b = (int)((o / CH) & 1);
    double t0 = now();
    fetch.th.join();
    t_values += now() - t0;
    if (fetch.rc) { rc = mfx_fail(fetch.rc, "%s", fetch.err.c_str()); break; }
    ka += fetch.a;
    km += fetch.m;
    if (e < len) start_fetch(e);
    const uint64_t cnt = e - o, per = (cnt + nthr - 1) / nthr;
    const uint32_t *rv = rvb[b].data(), *av = avb[b].data();
    t0 = now();
    std::atomic<bool> fmt_ok{true};
    {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nthr; ++t) {
        const uint64_t bb = std::min(cnt, t * per), n = std::min(cnt, bb + per) - bb;
        th.emplace_back([&, t, bb, n]() {
          try { dump_format_range(kp, name, name_len, o + bb, n, rv + bb, av + bb, parts[t], used[t]); }
          catch (const std::bad_alloc &) { used[t] = 0; fmt_ok = false; }
        });
      }
      for (auto &x : th) x.join();
    }
    t_format += now() - t0;
    if (!fmt_ok) { rc = mfx_fail(MFX_E_NOMEM, "out of host memory while formatting the dump of '%s'", name); break; }
    t0 = now();
    if (plain) {
      std::vector<uint64_t> at(nthr + 1, file_off);
      for (unsigned t = 0; t < nthr; ++t) at[t + 1] = at[t] + used[t];
      std::atomic<bool> wok{true};
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nthr; ++t) {
        if (!used[t]) continue;
        th.emplace_back([&, t]() {
          const char *p = parts[t].p;
          size_t left = used[t];
          uint64_t off = at[t];
          while (left) {
            const ssize_t r = pwrite(fd, p, std::min<size_t>(left, 64u << 20), (off_t)off);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) { wok = false; return; }
            p += r; left -= (size_t)r; off += (uint64_t)r;
          }
        });
      }
      for (auto &x : th) x.join();
      file_off = at[nthr];
      if (!wok) rc = mfx_fail(MFX_E_IO, "short write to '%s'", path);
    } else {
      for (unsigned t = 0; t < nthr; ++t)
        if (used[t] && fwrite(parts[t].data(), 1, used[t], f) != used[t]) {
          rc = mfx_fail(MFX_E_IO, "short write to '%s'", path);
          break;
        }
    }
    t_write += now() - t0;
  }
  if (fetch.th.joinable()) fetch.th.join();                 // (after an error: the look-ahead is waited for, nothing of it is used)
  if (plain) {
    if (close(fd) != 0 && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed", path);
  } else if (mfx_close(fh) && rc == MFX_OK) rc = mfx_fail(MFX_E_IO, "writing '%s' failed (stream error or the compressor exited with an error)", path);
  if (timing) fprintf(stderr, "-- dump of %s: waited for values %.3f s, format %.3f s, write %.3f s (%u threads%s)\n", name, t_values, t_format, t_write, nthr, plain ? ", positional writes" : "");
  if (kasm) *kasm = ka;
  if (kmissing) *kmissing = km;
  return rc;
}

extern "C" int mfx_dump_contig(mfx_eval *ev, const mfx_seq *seq, uint32_t contig, const char *name,
                               const char *path, int append, uint64_t *kasm, uint64_t *kmissing) {
  if (!ev || !seq || !name || !path) return mfx_fail(MFX_E_INVAL, "mfx_dump_contig: null argument");
  if (contig >= seq->ncontigs) return mfx_fail(MFX_E_INVAL, "mfx_dump_contig: contig %u out of range", contig);
  return dump_contig_impl(ev, seq, contig, name, path, append, kasm, kmissing,
                          [&](uint64_t o, uint64_t e, uint32_t *rv, uint32_t *av, uint64_t *a, uint64_t *m) {
                            return mfx_dump_values(ev, seq, contig, o, e, rv, av, a, m);
                          });
}

extern "C" int mfx_dump_contig_sharded(mfx_eval *const *evs, const mfx_seq *const *seqs, uint32_t nslots, uint32_t contig,
                                       const char *name, const char *path, int append, uint64_t *kasm, uint64_t *kmissing) {
  if (!evs || !seqs || nslots == 0 || !evs[0] || !seqs[0] || !name || !path)
    return mfx_fail(MFX_E_INVAL, "mfx_dump_contig_sharded: null argument");
  if (contig >= seqs[0]->ncontigs) return mfx_fail(MFX_E_INVAL, "mfx_dump_contig_sharded: contig %u out of range", contig);
  return dump_contig_impl(evs[0], seqs[0], contig, name, path, append, kasm, kmissing,
                          [&](uint64_t o, uint64_t e, uint32_t *rv, uint32_t *av, uint64_t *a, uint64_t *m) {
                            return mfx_dump_values_sharded(evs, seqs, nslots, contig, o, e, rv, av, a, m);
                          });
}

// ---------------------------------------------------------------------------
// -completeness
// ---------------------------------------------------------------------------
extern "C" int mfx_completeness_pieces(mfx_eval *ev, double *total64, double *undrcpy64) {
  if (!ev || !total64 || !undrcpy64) return mfx_fail(MFX_E_INVAL, "mfx_completeness_pieces: null argument");
  if (ev->ix->seq_only) return mfx_fail(MFX_E_INVAL, "mfx_completeness: a sequence-only index holds the k-mers of one sequence, -completeness needs every read k-mer; build a full index (mfx_index_create)");
  DevGuard g(ev->device);
  DevBuf<double> dp;
  MFX_HIP(dp.alloc(128));
  MFX_HIP(mfx_memset_now(dp.p, 0, 128 * sizeof(double)));
  MFX_HIP(ev->ix->wide() ? mfx_kw_completeness(ev->ix->view(), ev->peak, ev->n_prob, ev->d_probK, ev->d_probP, dp.p, ev->grid, nullptr)
                          : mfx_k_completeness(ev->ix->view(), ev->peak, ev->n_prob, ev->d_probK, ev->d_probP, dp.p, ev->grid, nullptr));
  double h[128];
  MFX_HIP(hipMemcpy(h, dp.p, sizeof(h), hipMemcpyDeviceToHost));
  memcpy(total64, h, 64 * sizeof(double));
  memcpy(undrcpy64, h + 64, 64 * sizeof(double));
  return MFX_OK;
}

extern "C" int mfx_completeness(mfx_eval *ev, double *total, double *undrcpy) {
  if (!ev || !total || !undrcpy) return mfx_fail(MFX_E_INVAL, "mfx_completeness: null argument");
  double t64[64], u64[64];
  int rc = mfx_completeness_pieces(ev, t64, u64);
  if (rc) return rc;
  double t = 0, u = 0;
  for (int i = 0; i < 64; ++i) { t += t64[i]; u += u64[i]; }     // merfin-completeness.C:135-138
  *total = t;
  *undrcpy = u;
  return MFX_OK;
}
