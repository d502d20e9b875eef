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

#include <cstring>

#include "common.hpp"

struct zs_rocm_dist {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
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
  return d;
}
void zs_rocm_dist_destroy(zs_rocm_dist *d) {
  if (!d) return;
  if (d->comm) (void)ncclCommDestroy(d->comm);
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
  const size_t bf = (size_t)nchn * side * side * side;
  zs_rocm_mpm_halo_pack(pol, grid, blocks, totalBlocks, side, chn0, nchn, sendbuf);
  {
    Launch L(pol, "halo_exchange");
    ZSR_NCCL(ncclGroupStart());
    for (int k = 0; k < npeers; ++k) {
      ZSR_NCCL(ncclSend(sendbuf + peerOffset[k] * bf, peerCount[k] * bf, ncclFloat, peerRank[k], d->comm, L.stream));
      ZSR_NCCL(ncclRecv(recvbuf + peerOffset[k] * bf, peerCount[k] * bf, ncclFloat, peerRank[k], d->comm, L.stream));
    }
    ZSR_NCCL(ncclGroupEnd());
  }
  zs_rocm_mpm_halo_unpack(pol, grid, blocks, totalBlocks, side, chn0, nchn, recvbuf, 2);  // atomic add: corner blocks appear once per peer
  return 0;
}

// in-place allreduce of n floats / int64 on the device; op 0 = sum, 1 = max, 2 = min
static ncclRedOp_t red_op(int op) { return op == 1 ? ncclMax : (op == 2 ? ncclMin : ncclSum); }
int zs_rocm_dist_allreduce_f32(zs_rocm_dist *d, zs_rocm_policy *pol, float *buf, size_t n, int op) {
  if (!d || !n) return 0;
  Launch L(pol, "allreduce_f32");
  ZSR_NCCL(ncclAllReduce(buf, buf, n, ncclFloat, red_op(op), d->comm, L.stream));
  return 0;
}
int zs_rocm_dist_allreduce_i64(zs_rocm_dist *d, zs_rocm_policy *pol, long long *buf, size_t n, int op) {
  if (!d || !n) return 0;
  Launch L(pol, "allreduce_i64");
  ZSR_NCCL(ncclAllReduce(buf, buf, n, ncclInt64, red_op(op), d->comm, L.stream));
  return 0;
}
// recv[r] = what rank r put into its send[this rank] (one int64 each): the counts exchange in front of an uneven all-to-all
int zs_rocm_dist_alltoall_i64(zs_rocm_dist *d, zs_rocm_policy *pol, const long long *send, long long *recv) {
  if (!d) return 0;
  Launch L(pol, "alltoall_i64");
  ZSR_NCCL(ncclGroupStart());
  for (int r = 0; r < d->world; ++r) {
    ZSR_NCCL(ncclSend(send + r, 1, ncclInt64, r, d->comm, L.stream));
    ZSR_NCCL(ncclRecv(recv + r, 1, ncclInt64, r, d->comm, L.stream));
  }
  ZSR_NCCL(ncclGroupEnd());
  return 0;
}
// uneven all-to-all of floats: sendCounts[r] floats starting at sendOffsets[r] go to rank r, recvCounts[r] floats from rank r land at
// recvOffsets[r] (host arrays of `world` entries): one grouped send / recv per pair over the direct xGMI link
int zs_rocm_dist_alltoallv_f32(zs_rocm_dist *d, zs_rocm_policy *pol, const float *send, const size_t *sendCounts, const size_t *sendOffsets,
                               float *recv, const size_t *recvCounts, const size_t *recvOffsets) {
  if (!d) return 0;
  Launch L(pol, "alltoallv_f32");
  ZSR_NCCL(ncclGroupStart());
  for (int r = 0; r < d->world; ++r) {
    if (sendCounts[r]) ZSR_NCCL(ncclSend(send + sendOffsets[r], sendCounts[r], ncclFloat, r, d->comm, L.stream));
    if (recvCounts[r]) ZSR_NCCL(ncclRecv(recv + recvOffsets[r], recvCounts[r], ncclFloat, r, d->comm, L.stream));
  }
  ZSR_NCCL(ncclGroupEnd());
  return 0;
}
// all ranks have reached this point AND finished their stream's work (host-blocking)
int zs_rocm_dist_barrier(zs_rocm_dist *d, zs_rocm_policy *pol) {
  if (!d) return 0;
  Launch L(pol, "dist_barrier");
  int *flag = (int *)L.temp(sizeof(int));
  ZSR_CHECK(hipMemsetAsync(flag, 0, sizeof(int), L.stream));
  ZSR_NCCL(ncclAllReduce(flag, flag, 1, ncclInt32, ncclSum, d->comm, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  return 0;
}

}  // extern "C"
