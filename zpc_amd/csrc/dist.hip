// dist.hip -- the multi-GPU exchange steps of the MPM path behind the C ABI, on RCCL directly (no Python, no torch): one process per
// GPU, one communicator per process.  The reference has no collective layer; the path shards spatially and has
// exactly these exchange steps:
//   * ghost-block halo exchange after P2G: for every peer the partial sums {m, mv, f} of the grid blocks BOTH ranks hold are swapped
//     and added -- pack kernel -> ONE ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd over the point-to-point xGMI
//     links (<= 7 peers for a 2x2x2 split, 2 for slabs: never a ring collective) -> atomic unpack-add kernel, all on the policy's
//     stream (the caller overlaps it with interior blocks by giving it a policy on a second stream);
//   * allreduce(max) of maxVelSqr for the CFL time step (simulation/grid/GridOp.hpp:71-108 computes the per-device value);
//   * particle migration: counts all-to-all + uneven all-to-all of AoS particle rows, as grouped send / recv.
// Everything is enqueued on the stream of the zs_rocm_policy passed in; nothing here synchronises the host except the uneven
// all-to-all, which needs the receive counts on the host to size the buffer (the caller passes them in).
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "common.hpp"

struct zs_rocm_dist {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  bool broken = false;  // a grouped send / recv failed half-way: peers may be left waiting, nothing more is sent on this communicator
};

namespace zsr {
static int nccl_check(ncclResult_t r, const char *what, const char *file, int line) {
  if (r == ncclSuccess) return 0;
  fprintf(stderr, "[zs_rocm | rccl] %s failed: %s at %s:%d\n", what, ncclGetErrorString(r), file, line);
  report_error(hipErrorUnknown, what, file, line);  // latched like any failed launch
  return -1;
}
#define ZSR_NCCL(expr)                                               \
  do {                                                               \
    if (::zsr::nccl_check((expr), #expr, __FILE__, __LINE__)) return -1; \
  } while (0)
// inside an open ncclGroupStart(): a failed call closes the group before returning (so that later RCCL calls of this thread do not queue
// into a group nobody ends) and marks the communicator unusable: the half-built group may have issued some of its sends / receives, a
// retry on the same communicator would pair them with the wrong messages.  The caller tears the communicator down (zs_rocm_dist_destroy
// aborts a broken one instead of draining it).
#define ZSR_NCCL_G(d, expr)                                          \
  do {                                                               \
    if (::zsr::nccl_check((expr), #expr, __FILE__, __LINE__)) {      \
      (void)ncclGroupEnd();                                          \
      (d)->broken = true;                                            \
      return -1;                                                     \
    }                                                                \
  } while (0)
// the policy must run on the communicator's device (its stream is handed to RCCL); device -1 = the calling thread's current device.
// (Resolved without constructing a Launch: its constructor / destructor wait on listened events, synchronise and print for profiling policies.)
static int same_device(const zs_rocm_dist *d, const zs_rocm_policy *pol, const char *what) {
  if (d->broken) {
    fprintf(stderr, "[zs_rocm | rccl] %s: the communicator is unusable after a failed grouped exchange\n", what);
    report_error(hipErrorUnknown, what, __FILE__, __LINE__);
    return -1;
  }
  const int dev = pol->device >= 0 ? pol->device : current_device();
  if (dev == d->device) return 0;
  fprintf(stderr, "[zs_rocm | rccl] %s: policy runs on device %d, communicator lives on device %d\n", what, dev, d->device);
  report_error(hipErrorInvalidDevice, what, __FILE__, __LINE__);
  return -1;
}
}  // namespace zsr

using namespace zsr;

extern "C" {

size_t zs_rocm_dist_unique_id_bytes(void) { return sizeof(ncclUniqueId); }
// rank 0 creates the id; the launcher hands the bytes to every rank (a file, an environment variable, MPI, a TCP store ...)
int zs_rocm_dist_unique_id(void *out) {
  ncclUniqueId id;
  ZSR_NCCL(ncclGetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
  return 0;
}
// collective over all ranks; `device` = the HIP device of this rank (-1: the current one)
zs_rocm_dist *zs_rocm_dist_create(int rank, int world, const void *uniqueId, int device) {
  if (world < 1 || rank < 0 || rank >= world || !uniqueId) return nullptr;
  auto *d = new zs_rocm_dist;
  d->rank = rank;
  d->world = world;
  d->device = device >= 0 ? device : current_device();
  DeviceGuard guard(d->device);
  ncclUniqueId id;
  std::memcpy(&id, uniqueId, sizeof(id));
  if (nccl_check(ncclCommInitRank(&d->comm, world, id, rank), "ncclCommInitRank", __FILE__, __LINE__)) {
    delete d;
    return nullptr;
  }
  // fail fast and loudly when the communicator is not the one the caller asked for (a launcher that started fewer ranks, two ranks on
  // one id slot): every later collective would hang or exchange with the wrong peer
  int count = -1, urank = -1, cudev = -1;
  (void)ncclCommCount(d->comm, &count);
  (void)ncclCommUserRank(d->comm, &urank);
  (void)ncclCommCuDevice(d->comm, &cudev);
  if (count != world || urank != rank) {
    fprintf(stderr, "[zs_rocm] dist_create: RCCL communicator has %d ranks (asked for %d), this rank is %d (asked for %d), device %d\n", count, world,
            urank, rank, cudev);
    (void)ncclCommAbort(d->comm);
    delete d;
    return nullptr;
  }
  if (getenv("ZS_ROCM_DIST_VERBOSE"))
    fprintf(stderr, "[zs_rocm] dist_create: rank %d of ncclCommCount = %d on HIP device %d\n", urank, count, cudev);
  return d;
}
// ncclCommCount of the communicator (what RCCL itself says the world is), or -1
int zs_rocm_dist_comm_count(const zs_rocm_dist *d) {
  int count = -1;
  if (!d || !d->comm || ncclCommCount(d->comm, &count) != ncclSuccess) return -1;
  return count;
}
void zs_rocm_dist_destroy(zs_rocm_dist *d) {
  if (!d) return;
  if (d->comm) (void)(d->broken ? ncclCommAbort(d->comm) : ncclCommDestroy(d->comm));
  delete d;
}
int zs_rocm_dist_rank(const zs_rocm_dist *d) { return d->rank; }
int zs_rocm_dist_world(const zs_rocm_dist *d) { return d->world; }

// Ghost-block exchange.  `blocks` (device) = the concatenated lists of local block numbers shared with each peer, peer k owning the
// slice [peerOffset[k], peerOffset[k] + peerCount[k]); both sides list a pair's blocks in the same (sorted-key) order.  sendbuf /
// recvbuf: totalBlocks * nchn * side^3 floats each.  grid[block][chn0 .. chn0 + nchn) += the peers' partial sums.
int zs_rocm_dist_halo_exchange(zs_rocm_dist *d, zs_rocm_policy *pol, float *grid, int side, int chn0, int nchn, const int *blocks,
                               size_t totalBlocks, int npeers, const int *peerRank, const size_t *peerOffset, const size_t *peerCount,
                               float *sendbuf, float *recvbuf) {
  if (!d || !totalBlocks || npeers <= 0) return 0;
  // the grid has 7 channels {m, mv, f} (geometry/Structure.hpp:37-131 as used by simulation/mpm/Simulator.cpp:116-122)
  if ((side != 4 && side != 8) || chn0 < 0 || nchn < 1 || chn0 + nchn > 7) {
    report_error(hipErrorInvalidValue, "halo_exchange: side must be 4 | 8 and [chn0, chn0 + nchn) within the 7 grid channels", __FILE__, __LINE__);
    return -1;
  }
  const size_t bf = (size_t)nchn * side * side * side;
  if (same_device(d, pol, "halo_exchange")) return -1;
  zs_rocm_mpm_halo_pack(pol, grid, blocks, totalBlocks, side, chn0, nchn, sendbuf);
  {
    Launch L(pol, "halo_exchange");
    ZSR_NCCL(ncclGroupStart());
    for (int k = 0; k < npeers; ++k) {
      ZSR_NCCL_G(d, ncclSend(sendbuf + peerOffset[k] * bf, peerCount[k] * bf, ncclFloat, peerRank[k], d->comm, L.stream));
      ZSR_NCCL_G(d, ncclRecv(recvbuf + peerOffset[k] * bf, peerCount[k] * bf, ncclFloat, peerRank[k], d->comm, L.stream));
    }
    ZSR_NCCL(ncclGroupEnd());
  }
  zs_rocm_mpm_halo_unpack(pol, grid, blocks, totalBlocks, side, chn0, nchn, recvbuf, 2);  // atomic add: corner blocks appear once per peer
  return 0;
}

// ---- the halo plan: which of this rank's grid blocks do other ranks hold as well?
// Pure host part (no GPU, no communicator: also what the CPU tests call): keysAll = the ranks' block-key lists back to back
// (counts[r] keys of 3 ints each, rank r's list in ITS block-number order), rank = this rank.  For every peer that shares at least one
// key, in rank order: the shared keys in lexicographic order (the order both sides derive independently) as positions in THIS rank's
// list.  Outputs: peerRank / peerOffset / peerCount (capacity world - 1), blocks (capacity: returned total; pass NULL to size).
// Returns the total number of (peer, block) pairs; *npeers = number of peers.
size_t zs_rocm_halo_plan_from_keys(const int *keysAll, const size_t *counts, int world, int rank, int *npeers, int *peerRank,
                                   size_t *peerOffset, size_t *peerCount, int *blocks) {
  struct Key {
    int k[3];
    bool operator<(const Key &o) const { return k[0] != o.k[0] ? k[0] < o.k[0] : (k[1] != o.k[1] ? k[1] < o.k[1] : k[2] < o.k[2]); }
    bool operator==(const Key &o) const { return k[0] == o.k[0] && k[1] == o.k[1] && k[2] == o.k[2]; }
  };
  std::vector<size_t> start(world + 1, 0);
  for (int r = 0; r < world; ++r) start[r + 1] = start[r] + counts[r];
  // this rank's keys sorted, with their block numbers
  std::vector<std::pair<Key, int>> mine(counts[rank]);
  for (size_t i = 0; i < counts[rank]; ++i) {
    const int *q = keysAll + 3 * (start[rank] + i);
    mine[i] = {Key{{q[0], q[1], q[2]}}, (int)i};
  }
  std::sort(mine.begin(), mine.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
  size_t total = 0;
  int np = 0;
  std::vector<Key> theirs;
  for (int p = 0; p < world; ++p) {
    if (p == rank || counts[p] == 0 || mine.empty()) continue;
    theirs.resize(counts[p]);
    for (size_t i = 0; i < counts[p]; ++i) {
      const int *q = keysAll + 3 * (start[p] + i);
      theirs[i] = Key{{q[0], q[1], q[2]}};
    }
    std::sort(theirs.begin(), theirs.end());
    size_t a = 0, b = 0, n = 0;
    while (a < mine.size() && b < theirs.size()) {  // merge walk: both sides visit the shared keys in ascending order
      if (mine[a].first < theirs[b]) ++a;
      else if (theirs[b] < mine[a].first) ++b;
      else {
        if (blocks) blocks[total + n] = mine[a].second;
        ++n, ++a, ++b;
      }
    }
    if (n) {
      if (peerRank) {
        peerRank[np] = p;
        peerOffset[np] = total;
        peerCount[np] = n;
      }
      ++np;
      total += n;
    }
  }
  if (npeers) *npeers = np;
  return total;
}

struct zs_rocm_halo_plan {
  hipEvent_t evBoundary = nullptr, evDone = nullptr;  // the overlapped step's two hand-overs between the compute and the exchange stream
  hipEvent_t evReady = nullptr;                       // ... and, with the two ranges side by side, the start of the exchange stream's range
  unsigned long long *signal = nullptr;               // one-launch schedule: running count of finished boundary workgroups (device) ...
  unsigned long long signalTarget = 0;                // ... and what it reads when the boundary blocks of every step so far are done
  int device = 0, side = 0, npeers = 0;
  size_t total = 0;
  std::vector<int> peerRank;
  std::vector<size_t> peerOffset, peerCount;
  int *blocks = nullptr;            // device: [total] local block numbers, per-peer slices
  float *sendbuf = nullptr, *recvbuf = nullptr;  // device: total * 7 * side^3 floats each
};
// Collective.  keys: device pointer to this rank's block keys (3 ints per block, block-number order: a bht's activeKeys), nblocks of
// them; side = 4 | 8.  All-gathers the key lists over RCCL, derives the plan on the host and allocates the exchange buffers.
zs_rocm_halo_plan *zs_rocm_dist_halo_plan_create(zs_rocm_dist *d, zs_rocm_policy *pol, const int *keys, size_t nblocks, int side) {
  if (!d || (side != 4 && side != 8)) return nullptr;
  if (same_device(d, pol, "halo_plan_create")) return nullptr;
  Launch L(pol, "halo_plan");
  auto *plan = new zs_rocm_halo_plan;
  plan->device = d->device;
  plan->side = side;
  const int world = d->world;
  // counts, then keys padded to the longest list
  unsigned long long *cntD = (unsigned long long *)L.temp(sizeof(unsigned long long) * (size_t)(world + 1));
  const unsigned long long mineCnt = nblocks;
  std::vector<unsigned long long> cntH(world);
  bool ok = hipMemcpyAsync(cntD + world, &mineCnt, sizeof(mineCnt), hipMemcpyHostToDevice, L.stream) == hipSuccess;
  ok = ok && ncclAllGather(cntD + world, cntD, 1, ncclUint64, d->comm, L.stream) == ncclSuccess;
  ok = ok && hipMemcpyAsync(cntH.data(), cntD, sizeof(unsigned long long) * world, hipMemcpyDeviceToHost, L.stream) == hipSuccess;
  ok = ok && hipStreamSynchronize(L.stream) == hipSuccess;
  if (!ok) {
    report_error(hipErrorUnknown, "halo plan: count all-gather", __FILE__, __LINE__);
    delete plan;
    return nullptr;
  }
  size_t mx = 1;
  for (int r = 0; r < world; ++r) mx = std::max<size_t>(mx, cntH[r]);
  int *padD = (int *)L.temp(sizeof(int) * 3 * mx * (size_t)(world + 1));
  int *mineD = padD + 3 * mx * (size_t)world;
  ok = hipMemsetAsync(mineD, 0, sizeof(int) * 3 * mx, L.stream) == hipSuccess;
  if (nblocks) ok = ok && hipMemcpyAsync(mineD, keys, sizeof(int) * 3 * nblocks, hipMemcpyDeviceToDevice, L.stream) == hipSuccess;
  ok = ok && ncclAllGather(mineD, padD, 3 * mx, ncclInt32, d->comm, L.stream) == ncclSuccess;
  std::vector<int> padH(3 * mx * (size_t)world);
  ok = ok && hipMemcpyAsync(padH.data(), padD, sizeof(int) * padH.size(), hipMemcpyDeviceToHost, L.stream) == hipSuccess;
  ok = ok && hipStreamSynchronize(L.stream) == hipSuccess;
  if (!ok) {
    report_error(hipErrorUnknown, "halo plan: key all-gather", __FILE__, __LINE__);
    delete plan;
    return nullptr;
  }
  std::vector<int> keysAll;
  std::vector<size_t> counts(world);
  for (int r = 0; r < world; ++r) {
    counts[r] = cntH[r];
    keysAll.insert(keysAll.end(), padH.begin() + 3 * mx * (size_t)r, padH.begin() + 3 * (mx * (size_t)r + counts[r]));
  }
  int np = 0;
  const size_t total = zs_rocm_halo_plan_from_keys(keysAll.data(), counts.data(), world, d->rank, &np, nullptr, nullptr, nullptr, nullptr);
  plan->npeers = np;
  plan->total = total;
  plan->peerRank.resize(np);
  plan->peerOffset.resize(np);
  plan->peerCount.resize(np);
  std::vector<int> blocksH(total);
  zs_rocm_halo_plan_from_keys(keysAll.data(), counts.data(), world, d->rank, &np, plan->peerRank.data(), plan->peerOffset.data(),
                              plan->peerCount.data(), blocksH.data());
  if (total) {
    DeviceGuard guard(d->device);
    const size_t bf = (size_t)7 * side * side * side;
    ok = hipMalloc((void **)&plan->blocks, sizeof(int) * total) == hipSuccess && hipMalloc((void **)&plan->sendbuf, sizeof(float) * total * bf) == hipSuccess
         && hipMalloc((void **)&plan->recvbuf, sizeof(float) * total * bf) == hipSuccess
         && hipMemcpy(plan->blocks, blocksH.data(), sizeof(int) * total, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
      report_error(hipErrorOutOfMemory, "halo plan: buffers", __FILE__, __LINE__);
      zs_rocm_dist_halo_plan_destroy(plan);
      return nullptr;
    }
  }
  return plan;
}
// A plan from explicit lists (callers with their own partition logic; the single-GPU rank proxy of bench.py, whose only "peer" is the
// rank itself): blocks[total] = local block numbers (host), slice k = [peerOffset[k], + peerCount[k]) shared with rank peerRank[k].
zs_rocm_halo_plan *zs_rocm_dist_halo_plan_from_lists(zs_rocm_dist *d, int side, int npeers, const int *peerRank, const size_t *peerOffset,
                                                     const size_t *peerCount, const int *blocks, size_t total) {
  if (!d || (side != 4 && side != 8) || npeers < 0 || (total && (!blocks || !peerRank || !peerOffset || !peerCount))) return nullptr;
  auto *plan = new zs_rocm_halo_plan;
  plan->device = d->device;
  plan->side = side;
  plan->npeers = npeers;
  plan->total = total;
  plan->peerRank.assign(peerRank, peerRank + npeers);
  plan->peerOffset.assign(peerOffset, peerOffset + npeers);
  plan->peerCount.assign(peerCount, peerCount + npeers);
  if (total) {
    DeviceGuard guard(d->device);
    const size_t bf = (size_t)7 * side * side * side;
    const bool ok = hipMalloc((void **)&plan->blocks, sizeof(int) * total) == hipSuccess && hipMalloc((void **)&plan->sendbuf, sizeof(float) * total * bf) == hipSuccess
                    && hipMalloc((void **)&plan->recvbuf, sizeof(float) * total * bf) == hipSuccess
                    && hipMemcpy(plan->blocks, blocks, sizeof(int) * total, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
      report_error(hipErrorOutOfMemory, "halo plan: buffers", __FILE__, __LINE__);
      zs_rocm_dist_halo_plan_destroy(plan);
      return nullptr;
    }
  }
  return plan;
}
void zs_rocm_dist_halo_plan_destroy(zs_rocm_halo_plan *p) {
  if (!p) return;
  DeviceGuard guard(p->device);
  if (p->evBoundary) (void)hipEventDestroy(p->evBoundary);
  if (p->evReady) (void)hipEventDestroy(p->evReady);
  if (p->signal) (void)hipFree(p->signal);
  if (p->evDone) (void)hipEventDestroy(p->evDone);
  if (p->blocks) (void)hipFree(p->blocks);
  if (p->sendbuf) (void)hipFree(p->sendbuf);
  if (p->recvbuf) (void)hipFree(p->recvbuf);
  delete p;
}
int zs_rocm_dist_halo_plan_npeers(const zs_rocm_halo_plan *p) { return p ? p->npeers : 0; }
size_t zs_rocm_dist_halo_plan_blocks(const zs_rocm_halo_plan *p) { return p ? p->total : 0; }
size_t zs_rocm_dist_halo_plan_bytes(const zs_rocm_halo_plan *p) { return p ? p->total * (size_t)7 * p->side * p->side * p->side * sizeof(float) : 0; }
const int *zs_rocm_dist_halo_plan_block_list(const zs_rocm_halo_plan *p) { return p ? p->blocks : nullptr; }
// the exchange of zs_rocm_dist_halo_exchange over the plan's own lists and buffers
int zs_rocm_dist_halo_plan_exchange(zs_rocm_halo_plan *p, zs_rocm_dist *d, zs_rocm_policy *pol, float *grid, int chn0, int nchn) {
  if (!p || !p->total) return 0;
  // the plan's buffers hold the 7 grid channels of every shared block: a wider or shifted channel window would run past them
  if (chn0 < 0 || nchn < 1 || chn0 + nchn > 7) {
    report_error(hipErrorInvalidValue, "halo_plan_exchange: [chn0, chn0 + nchn) must lie within the 7 grid channels", __FILE__, __LINE__);
    return -1;
  }
  return zs_rocm_dist_halo_exchange(d, pol, grid, p->side, chn0, nchn, p->blocks, p->total, p->npeers, p->peerRank.data(), p->peerOffset.data(),
                                    p->peerCount.data(), p->sendbuf, p->recvbuf);
}

// in-place allreduce of n floats / int64 on the device; op 0 = sum, 1 = max, 2 = min
static ncclRedOp_t red_op(int op) { return op == 1 ? ncclMax : (op == 2 ? ncclMin : ncclSum); }
int zs_rocm_dist_allreduce_f32(zs_rocm_dist *d, zs_rocm_policy *pol, float *buf, size_t n, int op) {
  if (!d || !n) return 0;
  if (same_device(d, pol, "allreduce_f32")) return -1;
  Launch L(pol, "allreduce_f32");
  ZSR_NCCL(ncclAllReduce(buf, buf, n, ncclFloat, red_op(op), d->comm, L.stream));
  return 0;
}
int zs_rocm_dist_allreduce_i64(zs_rocm_dist *d, zs_rocm_policy *pol, long long *buf, size_t n, int op) {
  if (!d || !n) return 0;
  if (same_device(d, pol, "allreduce_i64")) return -1;
  Launch L(pol, "allreduce_i64");
  ZSR_NCCL(ncclAllReduce(buf, buf, n, ncclInt64, red_op(op), d->comm, L.stream));
  return 0;
}
// recv[r] = what rank r put into its send[this rank] (one int64 each): the counts exchange in front of an uneven all-to-all
int zs_rocm_dist_alltoall_i64(zs_rocm_dist *d, zs_rocm_policy *pol, const long long *send, long long *recv) {
  if (!d) return 0;
  if (same_device(d, pol, "alltoall_i64")) return -1;
  Launch L(pol, "alltoall_i64");
  ZSR_NCCL(ncclGroupStart());
  for (int r = 0; r < d->world; ++r) {
    ZSR_NCCL_G(d, ncclSend(send + r, 1, ncclInt64, r, d->comm, L.stream));
    ZSR_NCCL_G(d, ncclRecv(recv + r, 1, ncclInt64, r, d->comm, L.stream));
  }
  ZSR_NCCL(ncclGroupEnd());
  return 0;
}
// uneven all-to-all of floats: sendCounts[r] floats starting at sendOffsets[r] go to rank r, recvCounts[r] floats from rank r land at
// recvOffsets[r] (host arrays of `world` entries): one grouped send / recv per pair over the direct xGMI link
int zs_rocm_dist_alltoallv_f32(zs_rocm_dist *d, zs_rocm_policy *pol, const float *send, const size_t *sendCounts, const size_t *sendOffsets,
                               float *recv, const size_t *recvCounts, const size_t *recvOffsets) {
  if (!d) return 0;
  if (same_device(d, pol, "alltoallv_f32")) return -1;
  Launch L(pol, "alltoallv_f32");
  ZSR_NCCL(ncclGroupStart());
  for (int r = 0; r < d->world; ++r) {
    if (sendCounts[r]) ZSR_NCCL_G(d, ncclSend(send + sendOffsets[r], sendCounts[r], ncclFloat, r, d->comm, L.stream));
    if (recvCounts[r]) ZSR_NCCL_G(d, ncclRecv(recv + recvOffsets[r], recvCounts[r], ncclFloat, r, d->comm, L.stream));
  }
  ZSR_NCCL(ncclGroupEnd());
  return 0;
}
// all ranks have reached this point AND finished their stream's work (host-blocking)
int zs_rocm_dist_barrier(zs_rocm_dist *d, zs_rocm_policy *pol) {
  if (!d) return 0;
  if (same_device(d, pol, "dist_barrier")) return -1;
  Launch L(pol, "dist_barrier");
  int *flag = (int *)L.temp(sizeof(int));
  ZSR_CHECK(hipMemsetAsync(flag, 0, sizeof(int), L.stream));
  ZSR_NCCL(ncclAllReduce(flag, flag, 1, ncclInt32, ncclSum, d->comm, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  return 0;
}

}  // extern "C"
// The gate of the one-launch schedule: ONE wave on the exchange stream that sleeps until the running count of finished boundary workgroups
// reaches `target`.  The launch that counts has been enqueued (on another stream) before the gate is, and a one-thread gate keeps no CU
// from it, so the count always arrives; should it not (a bug), the gate traps after ~10 s of the 100 MHz wall clock rather than hang the
// queue: the process then fails with a HIP error.
static __global__ void step_gate_kernel(const unsigned long long *signal, unsigned long long target) {
  const unsigned long long t0 = wall_clock64();
  // (relaxed: what is waited for are device-wide atomic adds, and the kernels behind the gate start with their own acquire)
  while (__hip_atomic_load(signal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > 1000000000ull) __builtin_trap();
  }
}
extern "C" {

// ---------------------------------------------------------------------------------------------------------------- one step, one call
// The whole sub-step of the slotted MPM path behind ONE C-ABI call: no host code of the caller runs between the kernels, so a C++ host
//  and bench.py enqueue a step for the price of one call -- at 8 ranks a rank's step is ~1 ms and a Python loop of ~8
// ctypes / RCCL calls per step is first-order.  Enqueued, in order, on the policy's stream:
//   gridB := 0;  fused G2P (from gridA) + P2G (into gridB) of the blocks [0, nBoundary)   [boundary blocks: the partition is numbered with
//   them first];  event;  the same for [nBoundary, nblocks) + re-home + commit;  wait for the exchange;  grid update of gridB (+ extf dt,
//   max |v|^2 into maxVelSqr);  allreduce(max) of maxVelSqr over the ranks (CFL).
// and on commPolicy's stream (a second stream), behind the event: pack -> grouped ncclSend / ncclRecv -> atomic unpack-add of the ghost
// blocks' partial sums (zs_rocm_dist_halo_plan_exchange) -- it overlaps the interior range.  commPolicy == NULL or nBoundary == 0 or
// == nblocks: one range, exchange on the main stream.  dist == NULL: single rank (no exchange, no allreduce).
// rangeSchedule (r06) selects how the boundary blocks get ahead: the two launches above (IN_TURN), the boundary launch on commPolicy's
// stream beside the interior launch (SIDE_BY_SIDE), or ONE launch over all blocks whose boundary workgroups count themselves off for a
// gate kernel on commPolicy's stream (ONE_LAUNCH; measured in DESIGN.md 7).
// What the reference offers for this: pol.device(i) + .listen() events between streams (cuda/execution/ExecutionPolicy.cuh:364-399);
// the schedule itself has no counterpart (zpc has no collective layer).  Returns 0, -1 on bad arguments or an RCCL error.
int zs_rocm_mpm_step_slotted(zs_rocm_policy *pol, const zs_rocm_mpm_step *a) {
  if (!pol || !a || !a->params || !a->table || !a->gridA || !a->gridB || !a->storage) return -1;
  const size_t nb = a->nblocks;
  if (!nb) return 0;
  const int side = a->params->side;
  const size_t gridBytes = nb * (size_t)7 * side * side * side * sizeof(float);
  zs_rocm_memset(pol, a->gridB, 0, gridBytes);
  float *const hgrid = a->haloGrid ? a->haloGrid : a->gridB;
  const bool exchange = a->dist && a->plan && a->plan->total;
  if (a->haloChannels < 0 || a->haloChannels > 7) return -1;
  if (a->rangeSchedule < ZS_ROCM_RANGES_IN_TURN || a->rangeSchedule > ZS_ROCM_RANGES_ONE_LAUNCH) return -1;
  if (a->rangeSchedule == ZS_ROCM_RANGES_ONE_LAUNCH && side != 8) return -1;   // (the 4^3-block kernel does not count its workgroups off)
  const int hch = a->haloChannels ? a->haloChannels : 7;
  const bool overlap = exchange && a->commPolicy && a->nBoundary > 0 && a->nBoundary < nb;
  int rc = 0;
  auto stamp = [&](void *ev) {
    if (!ev) return;
    Launch L(pol, "step: event");
    ZSR_CHECK(hipEventRecord((hipEvent_t)ev, L.stream));
  };
  auto bd = [&](int k, zs_rocm_policy *on) {  // breakdown event k on `on`'s stream
    if (!a->evBreakdown || !a->evBreakdown[k]) return;
    Launch L(on, "step: breakdown event");
    ZSR_CHECK(hipEventRecord((hipEvent_t)a->evBreakdown[k], L.stream));
  };
  stamp(a->evTransferBegin);
  bd(0, pol);
  if (overlap) {
    zs_rocm_halo_plan *p = a->plan;
    if (!p->evBoundary || !p->evDone) {  // (each creation checked by itself: a plan with only one of the two must not be used)
      DeviceGuard guard(p->device);
      if (!p->evBoundary && hipEventCreateWithFlags(&p->evBoundary, hipEventDisableTiming) != hipSuccess) p->evBoundary = nullptr;
      if (!p->evDone && hipEventCreateWithFlags(&p->evDone, hipEventDisableTiming) != hipSuccess) p->evDone = nullptr;
      if (!p->evBoundary || !p->evDone) {
        report_error(hipErrorOutOfMemory, "step_slotted: could not create the overlap events", __FILE__, __LINE__);
        return -1;
      }
    }
    const int sched = a->rangeSchedule;
    if (sched == ZS_ROCM_RANGES_ONE_LAUNCH) {
      // ONE launch over all blocks; its boundary workgroups (dispatched first: block order) count themselves off on p->signal, the gate kernel
      // on the exchange stream returns when the count says that the boundary blocks of this step are done
      if (!p->signal) {
        DeviceGuard guard(p->device);
        if (hipMalloc((void **)&p->signal, sizeof(unsigned long long)) != hipSuccess) {
          p->signal = nullptr;
          report_error(hipErrorOutOfMemory, "step_slotted: could not allocate the boundary signal", __FILE__, __LINE__);
          return -1;
        }
        Launch L(pol, "step: boundary signal := 0");
        ZSR_CHECK(hipMemsetAsync(p->signal, 0, sizeof(unsigned long long), L.stream));
        ZSR_CHECK(hipStreamSynchronize(L.stream));   // once per plan: the gate on the OTHER stream must never read the word before it is zero
        p->signalTarget = 0;
      }
      rc = mpm_g2p2g_slots_signal(pol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, 0, nb, 1, p->signal, a->nBoundary);
      if (rc) return rc;  // (bad arguments fail before any launch: nothing has been counted)
      p->signalTarget += a->nBoundary;
      {
        Launch C(a->commPolicy, "step: gate");
        hipLaunchKernelGGL(step_gate_kernel, dim3(1), dim3(1), 0, C.stream, (const unsigned long long *)p->signal, p->signalTarget);
      }
      bd(1, a->commPolicy);
    } else {
      const bool sideBySide = sched == ZS_ROCM_RANGES_SIDE_BY_SIDE;
      zs_rocm_policy *const bpol = sideBySide ? a->commPolicy : pol;   // the stream the boundary range runs on
      if (sideBySide) {  // gridB has been cleared (and gridA written) on the policy's stream: the other stream starts behind that
        if (!p->evReady) {
          DeviceGuard guard(p->device);
          if (hipEventCreateWithFlags(&p->evReady, hipEventDisableTiming) != hipSuccess) {
            p->evReady = nullptr;
            report_error(hipErrorOutOfMemory, "step_slotted: could not create the overlap events", __FILE__, __LINE__);
            return -1;
          }
        }
        {
          Launch L(pol, "step: grids ready");
          ZSR_CHECK(hipEventRecord(p->evReady, L.stream));
        }
        Launch C(a->commPolicy, "step: boundary range start");
        ZSR_CHECK(hipStreamWaitEvent(C.stream, p->evReady, 0));
      }
      rc = zs_rocm_mpm_g2p2g_slots(bpol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, 0, a->nBoundary, 0);
      if (rc) return rc;  // (nothing of the step has been committed: the range ran with finish = 0 and bad arguments fail before any launch)
      bd(1, bpol);
      {
        Launch L(bpol, "step: boundary done");
        ZSR_CHECK(hipEventRecord(p->evBoundary, L.stream));
      }
      if (!sideBySide) {
        Launch C(a->commPolicy, "step: exchange start");
        ZSR_CHECK(hipStreamWaitEvent(C.stream, p->evBoundary, 0));
      }
    }
    bd(3, a->commPolicy);
    if (a->handoverSnapshot)   // (test hook) what the exchange would read of gridB's shared blocks right now: their mass channel
      zs_rocm_mpm_halo_pack(a->commPolicy, a->gridB, p->blocks, p->total, p->side, 0, 1, a->handoverSnapshot);
    const int rcx = zs_rocm_dist_halo_plan_exchange(p, a->dist, a->commPolicy, hgrid, 0, hch);
    bd(4, a->commPolicy);
    {
      Launch C(a->commPolicy, "step: exchange done");
      ZSR_CHECK(hipEventRecord(p->evDone, C.stream));
    }
    // the second range ALWAYS runs, with its finish pass (re-home + commit): a failed exchange must not leave outbox records, claim words
    // and mover counts of the boundary range uncommitted -- the slot storage stays consistent, the error is returned afterwards
    if (sched == ZS_ROCM_RANGES_IN_TURN) {
      rc = zs_rocm_mpm_g2p2g_slots(pol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, a->nBoundary, nb, 1);
    } else if (sched == ZS_ROCM_RANGES_SIDE_BY_SIDE) {  // the interior range next to the boundary range; the finish pass needs the outbox records of both
      rc = zs_rocm_mpm_g2p2g_slots(pol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, a->nBoundary, nb, 0);
      {
        Launch L(pol, "step: wait for the boundary range");
        ZSR_CHECK(hipStreamWaitEvent(L.stream, p->evBoundary, 0));
      }
      const int rcf = zs_rocm_mpm_g2p2g_slots(pol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, nb, nb, 1);
      if (!rc) rc = rcf;
    }
    bd(2, pol);
    stamp(a->evTransferEnd);
    {
      Launch L(pol, "step: wait for the exchange");
      ZSR_CHECK(hipStreamWaitEvent(L.stream, p->evDone, 0));
    }
    bd(5, pol);
    if (rcx || rc) return rcx ? rcx : rc;
  } else {
    bd(1, pol);
    rc = zs_rocm_mpm_g2p2g_slots(pol, a->params, a->particles, a->table, a->gridA, a->gridB, nb, a->storage, a->writeAll, 0, nb, 1);
    if (rc) return rc;
    bd(2, pol);
    stamp(a->evTransferEnd);
    bd(3, pol);
    if (exchange) {
      rc = zs_rocm_dist_halo_plan_exchange(a->plan, a->dist, pol, hgrid, 0, hch);
      if (rc) return rc;
    }
    bd(4, pol);
    bd(5, pol);
  }
  if (a->maxVelSqr) zs_rocm_memset(pol, a->maxVelSqr, 0, sizeof(float));
  zs_rocm_mpm_grid_update(pol, a->params, a->gridB, nb, a->extf, a->maxVelSqr);
  if (a->collider) zs_rocm_mpm_apply_boundary(pol, a->params, a->table, a->gridB, nb, a->collider);
  bd(6, pol);
  if (a->dist && a->maxVelSqr) rc = zs_rocm_dist_allreduce_f32(a->dist, pol, a->maxVelSqr, 1, 1);
  bd(7, pol);
  return rc;
}

}  // extern "C"
