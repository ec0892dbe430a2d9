// mfx_comm.cpp -- the one collective of the multi-process -hist path, on RCCL (xGMI inside a node).
//
// One process per GPU: every rank evaluates its block-cyclic share of the tiles into a device counts image
// (include/merfin_amd.h MFX_HIST_WORDS) and the images are all-reduced.  What the reference does with one
// writer thread merging per-contig results (merfin-histogram.C:96-136) is here
//   - ncclAllReduce(sum, uint64) of the counts image                                 (integers: exact in any order)
//   - ncclAllGather of ONE fp64 koverCpy per rank, summed in RANK ORDER on every rank (a plain fp64 all-reduce
//     is only reproducible as long as RCCL picks the same algorithm; SURVEY 8(e) asks for the ordered form)
//   - K* bins beyond the dense image (the reference's arrays are unbounded, merfin-histogram.C:74,87) travel as
//     (side, bin) records: all-gather of the per-rank record lists when the reduced image says there are any.
// Messages are ~1 MB: latency-bound, xGMI bandwidth is irrelevant here.  The unique id is exchanged by the
// caller's launcher (any out-of-band channel: a file, MPI, torch.distributed's store).
#include "mfx_internal.h"

#include <rccl/rccl.h>
#include <string.h>

#include <algorithm>
#include <vector>

static_assert(sizeof(ncclUniqueId) <= MFX_COMM_ID_BYTES, "ncclUniqueId outgrew MFX_COMM_ID_BYTES");

struct mfx_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
  double   *d_gather = nullptr;      // [nranks] one koverCpy per rank
  uint64_t *d_novf = nullptr;        // [nranks] overflow records per rank
  uint64_t *d_cnt = nullptr;         // [nranks + nranks * nranks] a rank's send counts, then every rank's (mfx_comm_exchange_counts)
};

namespace {
struct DeviceScope {             // run on `dev`, give the caller's device back afterwards
  int prev = -1;
  bool ok = false;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

#define MFX_NCCL(call)                                                                                   \
  do {                                                                                                   \
    ncclResult_t r_ = (call);                                                                            \
    if (r_ != ncclSuccess) return mfx_fail(MFX_E_HIP, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

extern "C" int mfx_comm_unique_id(void *id) {
  if (!id) return mfx_fail(MFX_E_INVAL, "mfx_comm_unique_id: null argument");
  ncclUniqueId u;
  MFX_NCCL(ncclGetUniqueId(&u));
  memset(id, 0, MFX_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return MFX_OK;
}

extern "C" mfx_comm *mfx_comm_create(const void *id, int rank, int nranks, int device) {
  if (!id || nranks < 1 || rank < 0 || rank >= nranks || device < 0 || device >= mfx_device_count()) {
    mfx_fail(device < 0 || device >= mfx_device_count() ? MFX_E_NODEVICE : MFX_E_INVAL,
             "mfx_comm_create: bad argument (rank %d of %d, device %d of %d)", rank, nranks, device, mfx_device_count());
    return nullptr;
  }
  DeviceScope ds(device);
  if (!ds.ok) { mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", device); return nullptr; }
  mfx_comm *c = new mfx_comm;
  c->rank = rank; c->nranks = nranks; c->device = device;
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    mfx_fail(MFX_E_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, nranks, device, ncclGetErrorString(r));
    delete c;
    return nullptr;
  }
  if (hipMalloc((void **)&c->d_gather, (size_t)nranks * sizeof(double)) != hipSuccess ||
      hipMalloc((void **)&c->d_novf, (size_t)nranks * sizeof(uint64_t)) != hipSuccess ||
      hipMalloc((void **)&c->d_cnt, ((size_t)nranks + (size_t)nranks * nranks) * sizeof(uint64_t)) != hipSuccess) {
    mfx_fail(MFX_E_NOMEM, "mfx_comm_create: device allocation failed");
    mfx_comm_free(c);
    return nullptr;
  }
  return c;
}

extern "C" void mfx_comm_free(mfx_comm *c) {
  if (!c) return;
  DeviceScope ds(c->device);
  if (c->d_gather) (void)hipFree(c->d_gather);
  if (c->d_novf) (void)hipFree(c->d_novf);
  if (c->d_cnt) (void)hipFree(c->d_cnt);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}

extern "C" int mfx_comm_rank(const mfx_comm *c) { return c ? c->rank : -1; }
extern "C" int mfx_comm_size(const mfx_comm *c) { return c ? c->nranks : 0; }

// all ranks have reached this point and `stream` has drained (a one-word all-reduce + a stream synchronise)
extern "C" int mfx_comm_barrier(mfx_comm *c, void *stream) {
  if (!c) return mfx_fail(MFX_E_INVAL, "mfx_comm_barrier: null argument");
  DeviceScope ds(c->device);
  if (!ds.ok) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  MFX_HIP(hipMemsetAsync(c->d_novf, 0, sizeof(uint64_t), st));
  MFX_NCCL(ncclAllReduce(c->d_novf, c->d_novf, 1, ncclUint64, ncclSum, c->comm, st));
  MFX_HIP(hipStreamSynchronize(st));
  return MFX_OK;
}

hipError_t mfx_k_ordered_sum(const double *v, uint32_t n, double *out, hipStream_t st);   // mfx_kernels.hip

// in place on every rank; asynchronous on `stream`
extern "C" int mfx_hist_allreduce(mfx_comm *c, uint64_t *d_counts, double *d_kover, uint32_t nbins, uint32_t ncontigs, void *stream) {
  if (!c || !d_counts || !d_kover || !nbins) return mfx_fail(MFX_E_INVAL, "mfx_hist_allreduce: null argument");
  DeviceScope ds(c->device);
  if (!ds.ok) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  const size_t words = MFX_HIST_WORDS(nbins, ncontigs);
  MFX_NCCL(ncclAllReduce(d_counts, d_counts, words, ncclUint64, ncclSum, c->comm, st));
  MFX_NCCL(ncclAllGather(d_kover, c->d_gather, 1, ncclDouble, c->comm, st));
  MFX_HIP(mfx_k_ordered_sum(c->d_gather, (uint32_t)c->nranks, d_kover, st));
  return MFX_OK;
}

// Every rank ends with the far K* bins of ALL ranks as {key, occurrences} pairs (rank order; a key may come from several ranks:
// mfx_hist_result_add_overflow adds).  Collective: all ranks must call it when the reduced image's novf word
// (counts[2*nbins + 2]) is non-zero -- it is the same on every rank after the all-reduce.  Every rank collects its own table on
// the host (mfx_ovf_collect: rare, small), the pair lists travel through one fixed-size all-gather.
extern "C" int mfx_hist_allgather_overflow(mfx_comm *c, mfx_eval *ev, uint64_t *records, uint64_t cap, uint64_t *n_out, void *stream) {
  if (!c || !ev || !n_out) return mfx_fail(MFX_E_INVAL, "mfx_hist_allgather_overflow: null argument");
  DeviceScope ds(c->device);
  if (!ds.ok) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  std::vector<uint64_t> mine;
  uint64_t hd[2] = {0, 0};
  const int rc_mine = mfx_ovf_collect(ev, hd, &mine, st);      // (an error here must not keep this rank out of the collectives below)
  const uint64_t lost = rc_mine ? 1ull : 0ull;
  uint64_t *d_cnt = nullptr;
  MFX_HIP(hipMalloc((void **)&d_cnt, 2 * sizeof(uint64_t)));
  const uint64_t h2[2] = {mine.size() / 2, lost};
  hipError_t e = hipMemcpyAsync(d_cnt, h2, sizeof(h2), hipMemcpyHostToDevice, st);
  uint64_t *d_cnts = nullptr;
  if (e == hipSuccess) e = hipMalloc((void **)&d_cnts, 2 * (size_t)c->nranks * sizeof(uint64_t));
  ncclResult_t r = ncclSuccess;
  std::vector<uint64_t> cnt(2 * (size_t)c->nranks, 0);
  if (e == hipSuccess) r = ncclAllGather(d_cnt, d_cnts, 2, ncclUint64, c->comm, st);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(cnt.data(), d_cnts, cnt.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_cnt);
  if (d_cnts) (void)hipFree(d_cnts);
  if (r != ncclSuccess) return mfx_fail(MFX_E_HIP, "ncclAllGather of the far-bin counts failed: %s", ncclGetErrorString(r));
  MFX_HIP(e);
  uint64_t mx = 0, total = 0, any_lost = 0;
  for (int rk = 0; rk < c->nranks; ++rk) { mx = std::max(mx, cnt[2 * rk]); total += cnt[2 * rk]; any_lost += cnt[2 * rk + 1]; }
  *n_out = total;
  if (any_lost)
    return rc_mine ? rc_mine : mfx_fail(MFX_E_OVERFLOW, "a rank saw more distinct K* bins beyond the dense ones than its table of far bins holds (%u); create the evaluators with a larger nbins", MFX_OVF_SLOTS);
  if (total == 0) return MFX_OK;
  // fixed-size exchange: every rank contributes `mx` pairs (its own, then filler)
  uint64_t *d_mine = nullptr, *d_all = nullptr;
  mine.resize(2 * mx, 0);
  MFX_HIP(hipMalloc((void **)&d_mine, 2 * mx * sizeof(uint64_t)));
  e = hipMalloc((void **)&d_all, 2 * (size_t)c->nranks * mx * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMemcpyAsync(d_mine, mine.data(), 2 * mx * sizeof(uint64_t), hipMemcpyHostToDevice, st);
  std::vector<uint64_t> all(2 * (size_t)c->nranks * mx);
  if (e == hipSuccess) r = ncclAllGather(d_mine, d_all, 2 * mx, ncclUint64, c->comm, st);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_mine);
  if (d_all) (void)hipFree(d_all);
  if (r != ncclSuccess) return mfx_fail(MFX_E_HIP, "ncclAllGather of the far bins failed: %s", ncclGetErrorString(r));
  MFX_HIP(e);
  uint64_t w = 0;
  for (int rk = 0; rk < c->nranks; ++rk)
    for (uint64_t i = 0; i < cnt[2 * rk]; ++i, ++w)
      if (records && w < cap) { records[2 * w] = all[2 * ((size_t)rk * mx + i)]; records[2 * w + 1] = all[2 * ((size_t)rk * mx + i) + 1]; }
  if (total > cap) return mfx_fail(MFX_E_OVERFLOW, "%lu far K* bins over all ranks, caller buffer holds %lu", (unsigned long)total, (unsigned long)cap);
  return MFX_OK;
}

// ---------------------------------------------------------------------------
// The exchange of the sharded index (BASELINE config 5; SURVEY 8(e): "allToAllv of 8 B k-mers"): every rank has grouped
// the k-mers of its tiles by owner (mfx_route_tiles) and sends group r to rank r.  RCCL has no all-to-all-v primitive;
// it is one group of point-to-point sends and receives, which RCCL runs concurrently over the direct xGMI links.
//   1. mfx_comm_exchange_counts: recv_counts[s] = what rank s will send to this rank (an all-gather of the count rows;
//      synchronises `stream` once -- the receive buffer has to be sized on the host);
//   2. mfx_comm_alltoallv, once per array (keys, contig ids): element r-th group of d_send -> rank r, groups received
//      in source-rank order (so the owner sees its k-mers source by source, sequence order inside a source: the order
//      the fp64 koverCpy sum depends on).  Asynchronous on `stream`.
// ---------------------------------------------------------------------------
extern "C" int mfx_comm_exchange_counts(mfx_comm *c, const uint64_t *send_counts, uint64_t *recv_counts, void *stream) {
  if (!c || !send_counts || !recv_counts) return mfx_fail(MFX_E_INVAL, "mfx_comm_exchange_counts: null argument");
  DeviceScope ds(c->device);
  if (!ds.ok) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)c->nranks;
  MFX_HIP(hipMemcpyAsync(c->d_cnt, send_counts, n * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  MFX_NCCL(ncclAllGather(c->d_cnt, c->d_cnt + n, n, ncclUint64, c->comm, st));
  std::vector<uint64_t> all(n * n);
  MFX_HIP(hipMemcpyAsync(all.data(), c->d_cnt + n, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MFX_HIP(hipStreamSynchronize(st));
  for (size_t s2 = 0; s2 < n; ++s2) recv_counts[s2] = all[s2 * n + (size_t)c->rank];
  return MFX_OK;
}

extern "C" int mfx_comm_alltoallv(mfx_comm *c, const void *d_send, const uint64_t *send_counts, void *d_recv, const uint64_t *recv_counts,
                                  uint32_t elem_bytes, void *stream) {
  if (!c || !send_counts || !recv_counts || !elem_bytes) return mfx_fail(MFX_E_INVAL, "mfx_comm_alltoallv: null argument");
  DeviceScope ds(c->device);
  if (!ds.ok) return mfx_fail(MFX_E_HIP, "hipSetDevice(%d) failed", c->device);
  hipStream_t st = (hipStream_t)stream;
  uint64_t so = 0, ro = 0;
  for (int r = 0; r < c->nranks; ++r) { so += send_counts[r]; ro += recv_counts[r]; }
  if ((so && !d_send) || (ro && !d_recv)) return mfx_fail(MFX_E_INVAL, "mfx_comm_alltoallv: null buffer");
  so = ro = 0;
  MFX_NCCL(ncclGroupStart());
  ncclResult_t bad = ncclSuccess;
  for (int r = 0; r < c->nranks; ++r) {
    if (send_counts[r] && bad == ncclSuccess)
      bad = ncclSend((const char *)d_send + so * elem_bytes, send_counts[r] * elem_bytes, ncclUint8, r, c->comm, st);
    if (recv_counts[r] && bad == ncclSuccess)
      bad = ncclRecv((char *)d_recv + ro * elem_bytes, recv_counts[r] * elem_bytes, ncclUint8, r, c->comm, st);
    so += send_counts[r];
    ro += recv_counts[r];
  }
  const ncclResult_t ge = ncclGroupEnd();
  if (bad != ncclSuccess) return mfx_fail(MFX_E_HIP, "ncclSend / ncclRecv of the routed k-mers failed: %s", ncclGetErrorString(bad));
  if (ge != ncclSuccess) return mfx_fail(MFX_E_HIP, "ncclGroupEnd failed: %s", ncclGetErrorString(ge));
  return MFX_OK;
}
