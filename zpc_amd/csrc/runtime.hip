// runtime.hip -- device contexts, streams/events, temporaries, error latch, policy object, launch__device.
// Replaces zs::Cuda (cuda/Cuda.h:27-383, cuda/Cuda.cu:41-461) and the CudaExecutionPolicy plumbing
// (cuda/execution/ExecutionPolicy.cuh:345-535) for HIP on gfx950.
#include "common.hpp"

namespace zsr {

static std::mutex g_ctxMutex;
static std::map<int, DeviceContext *> g_contexts;

int current_device() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return 0;
  return d;
}

DeviceContext &context(int dev) {
  std::lock_guard<std::mutex> lk(g_ctxMutex);
  auto it = g_contexts.find(dev);
  if (it != g_contexts.end()) return *it->second;
  auto *c = new DeviceContext;
  c->dev = dev;
  g_contexts[dev] = c;
  return *c;
}

void report_error(hipError_t e, const char *what, const char *file, int line) {
  int dev = current_device();
  DeviceContext &c = context(dev);
  if (c.errorStatus == 0) c.errorStatus = (int)e;
  if (!c.errorPrinted) {  // first error per context only (cuda/Cuda.h:295-311)
    c.errorPrinted = true;
    fprintf(stderr, "[zs_rocm | device %d] %s failed: %s (%d) at %s:%d\n", dev, what, hipGetErrorString(e),
            (int)e, file, line);
  }
  (void)hipGetLastError();
}

static hipStream_t spare_stream(DeviceContext &c, int id) {
  if (id < 0) return nullptr;  // streamSpare(-1) == null stream (cuda/Cuda.h:115-120)
  id %= kNumSpareStreams;
  std::lock_guard<std::mutex> lk(c.mtx);
  if (!c.streams[id]) ZSR_CHECK(hipStreamCreateWithFlags(&c.streams[id], hipStreamNonBlocking));
  return c.streams[id];
}
static hipEvent_t spare_event(DeviceContext &c, int id) {
  id = id < 0 ? kNumSpareStreams : id % kNumSpareStreams;
  std::lock_guard<std::mutex> lk(c.mtx);
  if (!c.events[id]) ZSR_CHECK(hipEventCreateWithFlags(&c.events[id], hipEventDisableTiming));
  return c.events[id];
}

Launch::Launch(zs_rocm_policy *p, const char *w) : pol(p), what(w) {
  prevDev = current_device();
  dev = p->device >= 0 ? p->device : prevDev;
  if (dev != prevDev) ZSR_CHECK(hipSetDevice(dev));
  DeviceContext &c = context(dev);
  stream = p->hasExternal ? p->external : spare_stream(c, p->streamid);
  if (p->listenProc >= 0) {  // spareStreamWaitForEvent (cuda/Cuda.cu:164-168)
    DeviceContext &src = context(p->listenProc);
    ZSR_CHECK(hipStreamWaitEvent(stream, spare_event(src, p->listenStream), 0));
  }
  if (p->profile) {
    ZSR_CHECK(hipEventCreate(&t0));
    ZSR_CHECK(hipEventCreate(&t1));
    ZSR_CHECK(hipEventRecord(t0, stream));
  }
}

Launch::~Launch() {
  ZSR_CHECK(hipGetLastError());
  DeviceContext &c = context(dev);
  if (pol->profile) {
    ZSR_CHECK(hipEventRecord(t1, stream));
    ZSR_CHECK(hipEventSynchronize(t1));
    float ms = 0.f;
    ZSR_CHECK(hipEventElapsedTime(&ms, t0, t1));
    pol->lastMs = ms;
    fprintf(stderr, "[Rocm Exec | %s]: %.4f ms\n", what, ms);
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
  }
  if (pol->sync) ZSR_CHECK(hipStreamSynchronize(stream));
  // recordEventSpare (cuda/execution/ExecutionPolicy.cuh:489): lets other policies .listen() to us
  if (!pol->hasExternal) ZSR_CHECK(hipEventRecord(spare_event(c, pol->streamid), stream));
  if (dev != prevDev) ZSR_CHECK(hipSetDevice(prevDev));  // the caller's current device is not ours to change
}

static void *arena_take(int dev, hipStream_t stream, std::vector<size_t> &tempUsed, size_t bytes);
void *Launch::temp(size_t bytes) { return arena_take(dev, stream, tempUsed, bytes); }
DeviceContext::Arena &Launch::control() {
  DeviceContext &c = context(dev);
  std::lock_guard<std::mutex> lk(c.mtx);
  auto &a = c.arenas[stream];
  if (!a.ctl) {
    ZSR_CHECK(hipMalloc((void **)&a.ctl, kCtlBytes));
    ZSR_CHECK(hipMemsetAsync(a.ctl, 0, kCtlBytes, stream));  // ordered before every later use on this stream
    a.scanGen = 0;
    a.ticketShadow = 0;
  }
  return a;
}

char *Launch::scan_control(size_t numTiles, unsigned &gen, unsigned &ticketBase, bool &wrapped) {
  DeviceContext::Arena &a = control();
  DeviceContext &c = context(dev);
  std::lock_guard<std::mutex> lk(c.mtx);
  wrapped = false;
  if (++a.scanGen >= (1u << 30)) {
    a.scanGen = 1;
    wrapped = true;
  }
  gen = a.scanGen;
  ticketBase = a.ticketShadow;
  a.ticketShadow += (unsigned)numTiles;
  return a.ctl;
}

unsigned Launch::cu_count() {
  DeviceContext &c = context(dev);
  std::lock_guard<std::mutex> lk(c.mtx);
  if (c.cuCount == 0) {
    int v = 0;
    ZSR_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    c.cuCount = v > 0 ? v : 1;
  }
  return (unsigned)c.cuCount;
}

void Launch::scan_control_reset() {
  DeviceContext::Arena &a = control();
  DeviceContext &c = context(dev);
  std::lock_guard<std::mutex> lk(c.mtx);
  if (a.ctl) ZSR_CHECK(hipMemsetAsync(a.ctl, 0, kCtlBytes, stream));
  a.scanGen = 0;
  a.ticketShadow = 0;
}

static void *arena_take(int dev, hipStream_t stream, std::vector<size_t> &tempUsed, size_t bytes) {
  DeviceContext &c = context(dev);
  bytes = (bytes + 255) & ~(size_t)255;
  if (bytes == 0) bytes = 256;
  std::lock_guard<std::mutex> lk(c.mtx);
  auto &a = c.arenas[stream];
  tempUsed.resize(a.blocks.size(), 0);
  for (size_t i = 0; i < a.blocks.size(); ++i)
    if (tempUsed[i] + bytes <= a.blocks[i].cap) {
      void *r = a.blocks[i].ptr + tempUsed[i];
      tempUsed[i] += bytes;
      return r;
    }
  // new block; geometric growth keeps the number of blocks logarithmic.  Blocks are never moved, so
  // pointers handed out earlier in this call stay valid.
  size_t total = 0;
  for (auto &b : a.blocks) total += b.cap;
  DeviceContext::Block nb;
  nb.cap = bytes > total ? bytes : total;
  if (nb.cap < ((size_t)1 << 20)) nb.cap = (size_t)1 << 20;
  ZSR_CHECK(hipMalloc((void **)&nb.ptr, nb.cap));
  a.blocks.push_back(nb);
  tempUsed.push_back(bytes);
  return nb.ptr;
}

}  // namespace zsr

using namespace zsr;

extern "C" {

zs_rocm_policy *policy__device(void) { return new zs_rocm_policy; }
void del_policy__device(zs_rocm_policy *p) { delete p; }
void zs_rocm_policy_sync(zs_rocm_policy *p, int v) { p->sync = v; }
void zs_rocm_policy_profile(zs_rocm_policy *p, int v) { p->profile = v; }
void zs_rocm_policy_device(zs_rocm_policy *p, int v) { p->device = v; }
void zs_rocm_policy_stream(zs_rocm_policy *p, int v) { p->streamid = v; }
void zs_rocm_policy_listen(zs_rocm_policy *p, int proc, int sid) {
  p->listenProc = proc;
  p->listenStream = sid;
}
// temporary_memory_resource<device_mem_tag>::do_allocate / do_deallocate (cuda/memory/Allocator.h:33-50): a stream-ordered
// allocation on the policy's stream -- every block is its own allocation, stays valid until it is handed back, and never
// overlaps another live block or the library's own per-call scratch (Launch::temp)
void *zs_rocm_policy_temporary(zs_rocm_policy *p, size_t bytes) {
  if (bytes == 0) return nullptr;
  const int dev = p->device >= 0 ? p->device : current_device();
  DeviceGuard guard(dev);
  hipStream_t stream = p->hasExternal ? p->external : spare_stream(context(dev), p->streamid);
  void *ptr = nullptr;
  ZSR_CHECK(hipMallocAsync(&ptr, bytes, stream));
  return ptr;
}
void zs_rocm_policy_temporary_free(zs_rocm_policy *p, void *ptr) {
  if (!ptr) return;
  const int dev = p->device >= 0 ? p->device : current_device();
  DeviceGuard guard(dev);
  hipStream_t stream = p->hasExternal ? p->external : spare_stream(context(dev), p->streamid);
  ZSR_CHECK(hipFreeAsync(ptr, stream));
}
void zs_rocm_policy_shmem(zs_rocm_policy *p, size_t b) { p->shmem = b; }
void zs_rocm_policy_block(zs_rocm_policy *p, int tpb) { p->block = tpb; }
void zs_rocm_policy_external_stream(zs_rocm_policy *p, void *s) {
  p->external = (hipStream_t)s;
  p->hasExternal = true;  // note: a NULL external stream means "the caller's null stream"
}
void *zs_rocm_policy_get_stream(const zs_rocm_policy *p) {
  if (p->hasExternal) return (void *)p->external;
  int dev = p->device >= 0 ? p->device : current_device();
  return (void *)spare_stream(context(dev), p->streamid);
}
int zs_rocm_policy_should_sync(const zs_rocm_policy *p) { return p->sync; }
void zs_rocm_policy_sync_ctx(const zs_rocm_policy *p) {
  ZSR_CHECK(hipStreamSynchronize((hipStream_t)zs_rocm_policy_get_stream(p)));
}
float zs_rocm_policy_last_elapsed_ms(const zs_rocm_policy *p) { return p->lastMs; }
int zs_rocm_last_error(int device) { return context(device < 0 ? current_device() : device).errorStatus; }
void zs_rocm_clear_error(int device) {
  DeviceContext &c = context(device < 0 ? current_device() : device);
  c.errorStatus = 0;
  c.errorPrinted = false;
}
int zs_rocm_current_device(void) { return current_device(); }
int zs_rocm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
// hipMemsetAsync on the policy's stream: the clear of a grid / flag word between two kernels of the same stream (callers that own
// the memory through another runtime -- torch -- would otherwise clear it on THAT runtime's stream, unordered with ours)
void zs_rocm_memset(zs_rocm_policy *p, void *ptr, int byteVal, size_t bytes) {
  if (!ptr || !bytes) return;
  Launch L(p, "memset");
  ZSR_CHECK(hipMemsetAsync(ptr, byteVal, bytes, L.stream));
}
void zs_rocm_release_temporaries(void) {
  std::lock_guard<std::mutex> lk(g_ctxMutex);
  for (auto &kv : g_contexts) {
    std::lock_guard<std::mutex> lk2(kv.second->mtx);
    for (auto &a : kv.second->arenas) {
      for (auto &b : a.second.blocks)
        if (b.ptr) (void)hipFree(b.ptr);
      if (a.second.ctl) (void)hipFree(a.second.ctl);
    }
    kv.second->arenas.clear();
  }
}

// py_interop/cuda/ExecutionPolicy.cpp:11-39
void launch__device(zs_rocm_policy *p, void *kernel, size_t dim, void **args) {
  Launch L(p, "launch__device");
  const unsigned blockDim = 128;
  const unsigned gridDim = (unsigned)((dim + blockDim - 1) / blockDim);
  if (gridDim == 0) return;
  ZSR_CHECK(hipModuleLaunchKernel((hipFunction_t)kernel, gridDim, 1, 1, blockDim, 1, 1, (unsigned)p->shmem, L.stream,
                                  args, nullptr));
}

}  // extern "C"
