// common.hpp -- shared host/device plumbing of libzsrocm (gfx950 only).
//
// Runtime model (mirrors zs::Cuda, cuda/Cuda.h:27-255, redesigned for HIP): one context per device
// holding 32 lazily created spare streams + events, a latched error status, and a grow-only temporary
// arena per (device, stream) that stands in for the reference's stream-ordered cuMemAllocAsync
// temporaries (cuda/Cuda.cu:169-176) without an allocation on the hot path.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/zs_rocm.h"

namespace zsr {

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kNumSpareStreams = 32; // cuda/Cuda.h:39
// control block per (device, stream): [0, 32 KB) scan descriptors of 4-byte types ({status | value} words), [32 KB, 128 KB) those of 8-byte
// types (status words, aggregates, prefixes in separate arrays of kCtlScanTiles entries -- a region that holds VALUES is never read as
// status words by another instantiation), then the counters
constexpr size_t kCtlBytes = 132 << 10;
constexpr size_t kCtlDesc8 = 32 << 10, kCtlTicket = 128 << 10;
constexpr size_t kCtlScanTiles = 4096;   // scans of up to this many tiles run without a descriptor memset

// ------------------------------------------------------------------------------------ errors
struct DeviceContext {
  int dev = 0;
  hipStream_t streams[kNumSpareStreams] = {};
  hipEvent_t events[kNumSpareStreams + 1] = {};
  int errorStatus = 0;
  bool errorPrinted = false;
  int cuCount = 0;  // compute units of the device (0: not asked yet)
  std::mutex mtx;
  struct Block {
    char *ptr = nullptr;
    size_t cap = 0;
  };
  struct Arena {
    std::vector<Block> blocks;  // never moved or freed before zs_rocm_release_temporaries()
    // dedicated control memory of the single-launch primitives (never handed out by temp(), zeroed once): generation-tagged scan
    // descriptors and the scan's tile-ticket counter (never reset: the host keeps its value)
    char *ctl = nullptr;
    unsigned scanGen = 0, ticketShadow = 0;
  };
  std::map<hipStream_t, Arena> arenas;
};

DeviceContext &context(int dev);
int current_device();
// prints the first error per device with the source location, latches it (cuda/Cuda.h:291-312)
void report_error(hipError_t e, const char *what, const char *file, int line);

// switches the calling thread to `dev` for a scope and restores the previous device
struct DeviceGuard {
  int prev, dev;
  explicit DeviceGuard(int d) : prev(current_device()), dev(d) {
    if (dev >= 0 && dev != prev) (void)hipSetDevice(dev);
  }
  ~DeviceGuard() {
    if (dev >= 0 && dev != prev) (void)hipSetDevice(prev);
  }
};

#define ZSR_CHECK(expr)                                                  \
  do {                                                                   \
    hipError_t _e = (expr);                                              \
    if (_e != hipSuccess) ::zsr::report_error(_e, #expr, __FILE__, __LINE__); \
  } while (0)

}  // namespace zsr

// ------------------------------------------------------------------------------------ allocator handle
struct zs_rocm_allocator {
  int memsrc;    // memsrc_e: 0 host, 1 device, 2 um (types/Property.h:7)
  int8_t devid;  // ProcID, -1 = host
  // > 0: ZSPmrAllocator<true> -- a "STACK" virtual memory source (get_virtual_memory_source, resource/Resource.h): the
  // container reserves this much address space and maps physical memory as it grows; its data pointer never moves
  size_t virtualReserve = 0;
};

// ------------------------------------------------------------------------------------ policy
struct zs_rocm_policy {
  int sync = 1;          // execution/ExecutionPolicy.hpp:125
  int profile = 0;
  int device = -1;       // -1: current device (cuda/execution/ExecutionPolicy.cuh:909)
  int streamid = -1;     // -1: null stream
  int listenProc = -1, listenStream = -1;
  size_t shmem = 0;
  int block = 0;
  hipStream_t external = nullptr;
  bool hasExternal = false;
  float lastMs = 0.f;
};

namespace zsr {

struct Launch {
  // resolves device + stream of a policy, honours .listen(), brackets the op with events when
  // profiling and synchronises on destruction when the policy asks for it.
  explicit Launch(zs_rocm_policy *p, const char *what);
  ~Launch();
  hipStream_t stream = nullptr;
  int dev = 0, prevDev = 0;
  zs_rocm_policy *pol;
  const char *what;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // grow-only temporaries bound to (device, stream); valid until the next call on the same stream
  void *temp(size_t bytes);
  std::vector<size_t> tempUsed;  // bytes handed out per arena block during this call
  // control block of this (device, stream): kCtlBytes, zero-initialised on first use (stream-ordered); see DeviceContext::Arena
  DeviceContext::Arena &control();
  // reserves generation + ticket range of one scan on this stream (under the context lock); *wrapped: the generation counter started over
  char *scan_control(size_t numTiles, unsigned &gen, unsigned &ticketBase, bool &wrapped);
  // after a failed launch: control block zeroed (stream-ordered), generation and ticket shadow start over
  void scan_control_reset();
  // compute units of this launch's device (kernels with an in-launch grid barrier size their grid by it)
  unsigned cu_count();
};

// ------------------------------------------------------------------------------------ iterator ports
template <class T> struct Port {
  T *base;
  uint32_t idx, bits, mask, chns;
  __host__ __device__ __forceinline__ size_t off(size_t i) const {
    size_t j = (size_t)idx + i;
    return (((j >> bits) * (size_t)chns) << bits) | (j & (size_t)mask);
  }
  __host__ __device__ __forceinline__ T &operator[](size_t i) const { return base[off(i)]; }
  // component stride of vector attributes (GenericIterator.hpp:101)
  __host__ __device__ __forceinline__ size_t cstride() const { return (size_t)mask + 1; }
  __host__ __device__ bool contiguous() const { return chns == 1; }
};
template <class T, class P> inline Port<T> make_port(const P &p) {
  return Port<T>{(T *)p.base, p.idx, p.numTileBits, p.tileMask, p.numChns};
}
template <class T> inline Port<T> contiguous_port(T *p) { return Port<T>{p, 0u, 0u, 0u, 1u}; }

inline unsigned ceil_div(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// zs_rocm_mpm_g2p2g_slots with the boundary signal (signal == NULL: none): mpm_slotted.hip, used by dist.hip's one-launch step
int mpm_g2p2g_slots_signal(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps, const zs_rocm_bht_3 *tab, const float *gridA,
                           float *gridB, size_t nblocks, const zs_rocm_slot_storage *st, int writeAll, size_t blockBegin, size_t blockEnd, int finish,
                           unsigned long long *signal, size_t signalBlocks);

}  // namespace zsr

// ------------------------------------------------------------------------------------ device helpers
namespace zsr {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

template <class T> __device__ __forceinline__ T shfl_down(T v, int d) { return __shfl_down(v, d, 64); }
template <class T> __device__ __forceinline__ T shfl_up(T v, int d) { return __shfl_up(v, d, 64); }
template <class T> __device__ __forceinline__ T shfl(T v, int l) { return __shfl(v, l, 64); }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

// zs::plus / multiplies / getmin / getmax (ZpcFunctional.hpp)
enum { OP_PLUS = 0, OP_MUL = 1, OP_MIN = 2, OP_MAX = 3 };
template <int OP, class T> __host__ __device__ __forceinline__ T apply(T a, T b) {
  if constexpr (OP == OP_PLUS) return a + b;
  else if constexpr (OP == OP_MUL) return a * b;
  else if constexpr (OP == OP_MIN) return a < b ? a : b;
  else return a > b ? a : b;
}

}  // namespace zsr
