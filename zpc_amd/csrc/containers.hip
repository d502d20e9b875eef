// containers.hip -- allocator handles, zs::Vector<T>, property tags and zs::TileVector<T, L> behind the
// reference's own C ABI (py_interop/{Allocator,VectorInstantiations,TileVectorInstantiations}.cpp,
// py_interop/cuda/TileVectorUtility.cpp) plus the AoSoA kernels the hot path uses.
//
// AoSoA layout (container/TileVector.hpp:73-74,108,397): one buffer of count_tiles(n) * L * C elements,
// element (chn, i) at (i / L * C + chn) * L + i % L.  Every kernel below maps lane -> i % L so that a
// wave reads/writes whole 256-byte (L = 64) or two 128-byte (L = 32) rows of one channel: fully coalesced.
#include <cstring>
#include <string>
#include <vector>

#include "common.hpp"

using namespace zsr;


namespace zsr {

static void *mem_alloc(const zs_rocm_allocator &a, size_t bytes) {
  if (bytes == 0) return nullptr;
  void *p = nullptr;
  if (a.memsrc == 0) return std::malloc(bytes);
  int prev = current_device();
  if (a.devid >= 0 && a.devid != prev) ZSR_CHECK(hipSetDevice(a.devid));
  if (a.memsrc == 1) ZSR_CHECK(hipMalloc(&p, bytes));
  else ZSR_CHECK(hipMallocManaged(&p, bytes));
  if (a.devid >= 0 && a.devid != prev) ZSR_CHECK(hipSetDevice(prev));
  return p;
}
static void mem_free(const zs_rocm_allocator &a, void *p) {
  if (!p) return;
  if (a.memsrc == 0) std::free(p);
  else ZSR_CHECK(hipFree(p));
}
// Resource::copy (resource/Resource.h:273-300)
static void mem_copy(const zs_rocm_allocator &da, void *dst, const zs_rocm_allocator &sa, const void *src, size_t bytes) {
  if (!bytes) return;
  if (da.memsrc == 0 && sa.memsrc == 0) std::memcpy(dst, src, bytes);
  else ZSR_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
}
static void mem_set(const zs_rocm_allocator &a, void *p, int ch, size_t bytes) {
  if (!bytes) return;
  if (a.memsrc == 0) std::memset(p, ch, bytes);
  else {
    ZSR_CHECK(hipMemset(p, ch, bytes));
    ZSR_CHECK(hipDeviceSynchronize());
  }
}

// HIP virtual memory management behind ZSPmrAllocator<true> (cuda/memory/Allocator.cpp:120-420 uses cuMemAddressReserve /
// cuMemCreate / cuMemMap): reserve address space once, map granule-sized physical chunks at the end as the container grows.
struct VmmRange {
  char *base = nullptr;
  size_t reserved = 0, mapped = 0, gran = 0;
  int dev = 0;
  std::vector<hipMemGenericAllocationHandle_t> chunks;
  std::vector<size_t> chunkBytes;
  bool reserve(size_t bytes, int device) {
    dev = device;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || g == 0) return false;
    // uniform power-of-two chunks, the range aligned to the chunk size (mixed-size mappings at unaligned offsets made
    // hipMemSetAccess fail intermittently on ROCm 7): reserved / 1024 clamped to [2 MiB, 1 GiB]
    gran = (size_t)2 << 20;
    while (gran < g) gran <<= 1;
    while (gran < ((size_t)1 << 30) && gran * 1024 < bytes) gran <<= 1;
    reserved = (bytes + gran - 1) / gran * gran;
    if (hipMemAddressReserve((void **)&base, reserved, gran, nullptr, 0) != hipSuccess) {
      base = nullptr;
      return false;
    }
    return true;
  }
  bool grow(size_t bytes) {  // make [0, bytes) backed by physical memory
    if (bytes <= mapped) return true;
    const size_t want = (bytes + gran - 1) / gran * gran;
    if (want > reserved) {
      fprintf(stderr, "[zs_rocm] virtual range exhausted: need %zu of %zu reserved bytes\n", want, reserved);
      return false;
    }
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    while (mapped < want) {
      hipMemGenericAllocationHandle_t h;
      hipError_t e = hipMemCreate(&h, gran, &prop, 0);
      if (e != hipSuccess) {
        fprintf(stderr, "[zs_rocm] hipMemCreate(%zu) failed: %s\n", gran, hipGetErrorString(e));
        return false;
      }
      if ((e = hipMemMap(base + mapped, gran, 0, h, 0)) != hipSuccess) {
        fprintf(stderr, "[zs_rocm] hipMemMap failed: %s\n", hipGetErrorString(e));
        (void)hipMemRelease(h);
        return false;
      }
      if ((e = hipMemSetAccess(base + mapped, gran, &acc, 1)) != hipSuccess) {
        fprintf(stderr, "[zs_rocm] hipMemSetAccess failed: %s\n", hipGetErrorString(e));
        (void)hipMemUnmap(base + mapped, gran);
        (void)hipMemRelease(h);
        return false;
      }
      chunks.push_back(h);
      chunkBytes.push_back(gran);
      mapped += gran;
    }
    return true;
  }
  void release() {
    if (!base) return;
    (void)hipDeviceSynchronize();
    size_t off = 0;
    for (size_t i = 0; i < chunks.size(); ++i) {
      (void)hipMemUnmap(base + off, chunkBytes[i]);
      (void)hipMemRelease(chunks[i]);
      off += chunkBytes[i];
    }
    (void)hipMemAddressFree(base, reserved);
    base = nullptr;
    chunks.clear();
    chunkBytes.clear();
    mapped = reserved = 0;
  }
};

// untyped zs::Vector (container/Vector.hpp:11-421)
struct VecBuf {
  zs_rocm_allocator alloc{1, 0};
  char *data = nullptr;
  size_t size = 0, cap = 0, esize = 4;
  VmmRange *vmm = nullptr;  // non-null: storage is a mapped prefix of a reserved address range
  VecBuf() = default;
  VecBuf(const zs_rocm_allocator &a, size_t n, size_t es) : alloc(a), size(n), cap(n), esize(es) {
    if (a.virtualReserve && a.memsrc == 1) {
      vmm = new VmmRange;
      const int dev = a.devid >= 0 ? a.devid : current_device();
      const size_t need = n * es;
      if (vmm->reserve(a.virtualReserve > need ? a.virtualReserve : need, dev) && vmm->grow(need)) {
        data = vmm->base;
        cap = vmm->mapped / es;
        return;
      }
      static bool warned = false;
      if (!warned) fprintf(stderr, "[zs_rocm] virtual memory source unavailable on this device/runtime: using a plain allocation\n");
      warned = true;
      vmm->release();
      delete vmm;
      vmm = nullptr;
      (void)hipGetLastError();
    }
    data = (char *)mem_alloc(a, n * es);
  }
  ~VecBuf() {
    if (vmm) {
      vmm->release();
      delete vmm;
    } else
      mem_free(alloc, data);
  }
  VecBuf(const VecBuf &) = delete;
  VecBuf &operator=(const VecBuf &) = delete;
  void swap(VecBuf &o) {
    std::swap(alloc, o.alloc);
    std::swap(data, o.data);
    std::swap(size, o.size);
    std::swap(cap, o.cap);
    std::swap(esize, o.esize);
    std::swap(vmm, o.vmm);
  }
  size_t growth(size_t newSize) const {  // geometric_size_growth, Vector.hpp:407-416
    size_t g = cap + cap / 2;
    return newSize > g ? newSize : g;
  }
  void resize(size_t newSize, size_t alignment = 1) {  // resize / alignedResize, Vector.hpp:228-288
    if (newSize <= size) {
      size = newSize;
      return;
    }
    if (newSize > cap) {
      if (vmm && vmm->grow(newSize * esize)) {  // virtual: map more pages behind the same pointer, nothing moves
        cap = vmm->mapped / esize;
        size = newSize;
        return;
      }
      size_t ncap = growth(newSize);
      if (size_t r = ncap % alignment) ncap += alignment - r;
      VecBuf tmp(alloc, ncap, esize);
      if (size) mem_copy(tmp.alloc, tmp.data, alloc, data, size * esize);
      tmp.size = newSize;
      swap(tmp);
      return;
    }
    size = newSize;
  }
  void relocate(int memsrc, int8_t devid) {  // clone(mloc) + swap
    zs_rocm_allocator na{memsrc, devid, alloc.virtualReserve};
    VecBuf tmp(na, cap, esize);
    tmp.size = size;
    if (size) mem_copy(na, tmp.data, alloc, data, size * esize);
    swap(tmp);
  }
  void reset(int ch) { mem_set(alloc, data, ch, size * esize); }  // Vector.hpp:225-227
};

}  // namespace zsr

// ------------------------------------------------------------------------------------ property tags
struct zs_rocm_property_tags {
  std::vector<std::string> names;
  std::vector<int> sizes;
};

namespace zsr {

struct TileVec {
  std::vector<std::string> names;
  std::vector<int> sizes, offsets;
  int numChannels = 0;
  size_t L = 32, size = 0;
  VecBuf buf;
  // property tables mirrored in the container's memory space for in-kernel name lookup (TileVector.hpp:76-88,503-509):
  // tagNames[i] is a 32-byte SmallString, then offsets and sizes
  char *tagNames = nullptr;
  int *tagOffsets = nullptr, *tagSizes = nullptr;
  void mirror_tags() {
    const size_t n = names.size();
    zs_rocm_allocator h{0, -1};
    std::vector<char> nb(32 * (n ? n : 1), 0);
    for (size_t i = 0; i < n; ++i) std::strncpy(nb.data() + 32 * i, names[i].c_str(), 31);
    mem_free(buf.alloc, tagNames); mem_free(buf.alloc, tagOffsets); mem_free(buf.alloc, tagSizes);
    zs_rocm_allocator plain{buf.alloc.memsrc, buf.alloc.devid};
    tagNames = (char *)mem_alloc(plain, nb.size());
    tagOffsets = (int *)mem_alloc(plain, sizeof(int) * (n ? n : 1));
    tagSizes = (int *)mem_alloc(plain, sizeof(int) * (n ? n : 1));
    mem_copy(plain, tagNames, h, nb.data(), nb.size());
    if (n) {
      mem_copy(plain, tagOffsets, h, offsets.data(), sizeof(int) * n);
      mem_copy(plain, tagSizes, h, sizes.data(), sizeof(int) * n);
    }
  }
  ~TileVec() { mem_free(buf.alloc, tagNames); mem_free(buf.alloc, tagOffsets); mem_free(buf.alloc, tagSizes); }
  TileVec(const zs_rocm_allocator &a, const zs_rocm_property_tags &t, size_t n, size_t L_, size_t es)
      : names(t.names), sizes(t.sizes), L(L_), size(n) {
    for (int s : sizes) {  // running sums in declaration order (TileVector.hpp:79-85)
      offsets.push_back(numChannels);
      numChannels += s;
    }
    VecBuf b(a, tiles(n) * L * (size_t)numChannels, es);
    buf.swap(b);
    mirror_tags();
  }
  size_t tiles(size_t n) const { return (n + L - 1) / L; }
  int find(const char *name) const {
    for (size_t i = 0; i < names.size(); ++i)
      if (names[i] == name) return (int)i;
    return -1;
  }
  void resize(size_t n) {  // TileVector.hpp:477-482
    size = n;
    buf.resize(tiles(n) * L * (size_t)numChannels, L * (size_t)numChannels);
  }
  void relocate(int m, int8_t d) {
    mem_free(buf.alloc, tagNames); mem_free(buf.alloc, tagOffsets); mem_free(buf.alloc, tagSizes);
    tagNames = nullptr;
    tagOffsets = tagSizes = nullptr;
    buf.relocate(m, d);
    mirror_tags();
  }
};

// ------------------------------------------------------------------------------------ kernels
// one thread per (tile, channel, lane): consecutive threads -> consecutive lanes of one channel row
template <class W> __global__ void tv_fill_kernel(W *buf, size_t total, W val) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) buf[i] = val;
}

// append_channels copy (TileVector.hpp:583-623): dst has Cd >= Cs channels, the first Cs channels are the
// old ones, new channels zero-filled.
template <class W>
__global__ void tv_append_copy_kernel(const W *src, W *dst, size_t ntiles, int Cs, int Cd, int lbits) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t L = (size_t)1 << lbits;
  const size_t total = ntiles * (size_t)Cd * L;
  if (i >= total) return;
  size_t lane = i & (L - 1);
  size_t rc = i >> lbits;
  size_t chn = rc % (size_t)Cd, tile = rc / (size_t)Cd;
  dst[i] = chn < (size_t)Cs ? src[((tile * (size_t)Cs + chn) << lbits) | lane] : (W)0;
}

// dst(:, i) = src(:, map[i]) (gather) or dst(:, map[i]) = src(:, i) (scatter), all C channels.
// Thread = (i, chunk of channels): lanes walk i so the contiguous side is coalesced per channel row.
template <class W, bool GATHER>
__global__ void tv_reorder_kernel(const W *src, W *dst, size_t n, int C, int lbits, const int *map) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t L = (size_t)1 << lbits;
  size_t j = (size_t)map[i];
  size_t si = GATHER ? j : i, di = GATHER ? i : j;
  const W *s = src + (((si >> lbits) * (size_t)C) << lbits) + (si & (L - 1));
  W *d = dst + (((di >> lbits) * (size_t)C) << lbits) + (di & (L - 1));
#pragma unroll 5
  for (int c = 0; c < C; ++c) d[(size_t)c << lbits] = s[(size_t)c << lbits];
}

// gather of a subset of the channels (bit c of `mask`): re-binning between fused MPM steps only has to carry the step's inputs
__global__ void tv_gather_channels_kernel(const uint32_t *src, uint32_t *dst, size_t n, int C, int lbits, const int *map,
                                          unsigned long long mask) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t L = (size_t)1 << lbits;
  const size_t j = (size_t)map[i];
  const uint32_t *s = src + (((j >> lbits) * (size_t)C) << lbits) + (j & (L - 1));
  uint32_t *d = dst + (((i >> lbits) * (size_t)C) << lbits) + (i & (L - 1));
  for (int c = 0; c < C; ++c)
    if ((mask >> c) & 1ull) d[(size_t)c << lbits] = s[(size_t)c << lbits];  // mask is uniform: no divergence
}

__global__ void tv_from_aos_kernel(const float *aos, size_t n, int C, int lbits, float *tv) {
  // thread per (i, c) with c fastest on the AoS side would uncoalesce the AoSoA side; stage through LDS:
  // block handles 64 particles x C channels
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const size_t i0 = (size_t)blockIdx.x * 64;
  const int cnt = (int)(n - i0 < 64 ? n - i0 : 64);
  for (int k = threadIdx.x; k < cnt * C; k += blockDim.x) sm[k] = aos[i0 * (size_t)C + k];  // coalesced AoS read
  __syncthreads();
  const size_t L = (size_t)1 << lbits;
  for (int k = threadIdx.x; k < 64 * C; k += blockDim.x) {
    int c = k >> 6, l = k & 63;
    if (l < cnt) {
      size_t i = i0 + l;
      tv[(((i >> lbits) * (size_t)C + c) << lbits) | (i & (L - 1))] = sm[l * C + c];
    }
  }
}
__global__ void tv_to_aos_kernel(const float *tv, size_t n, int C, int lbits, float *aos) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const size_t i0 = (size_t)blockIdx.x * 64;
  const int cnt = (int)(n - i0 < 64 ? n - i0 : 64);
  const size_t L = (size_t)1 << lbits;
  for (int k = threadIdx.x; k < 64 * C; k += blockDim.x) {
    int c = k >> 6, l = k & 63;
    if (l < cnt) {
      size_t i = i0 + l;
      sm[l * C + c] = tv[(((i >> lbits) * (size_t)C + c) << lbits) | (i & (L - 1))];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cnt * C; k += blockDim.x) aos[i0 * (size_t)C + k] = sm[k];
}

// rows map[j] of an AoSoA buffer -> AoS rows j (send buffers of the particle migration); 64 rows per block through LDS so that
// both sides are coalesced
__global__ void tv_gather_rows_kernel(const float *tv, const int *map, size_t m, int C, int lbits, float *aos) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const size_t j0 = (size_t)blockIdx.x * 64;
  const int cnt = (int)(m - j0 < 64 ? m - j0 : 64);
  const size_t L = (size_t)1 << lbits;
  for (int k = threadIdx.x; k < 64 * C; k += blockDim.x) {
    const int c = k >> 6, l = k & 63;
    if (l < cnt) {
      const size_t i = (size_t)map[j0 + l];
      sm[l * C + c] = tv[(((i >> lbits) * (size_t)C + c) << lbits) | (i & (L - 1))];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < cnt * C; k += blockDim.x) aos[j0 * (size_t)C + k] = sm[k];
}
// AoS rows j -> elements dstOffset + j of an AoSoA buffer (received particles appended behind the kept ones)
__global__ void tv_scatter_rows_kernel(const float *aos, size_t m, int C, int lbits, float *tv, size_t dstOffset) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const size_t j0 = (size_t)blockIdx.x * 64;
  const int cnt = (int)(m - j0 < 64 ? m - j0 : 64);
  const size_t L = (size_t)1 << lbits;
  for (int k = threadIdx.x; k < cnt * C; k += blockDim.x) sm[k] = aos[j0 * (size_t)C + k];
  __syncthreads();
  for (int k = threadIdx.x; k < 64 * C; k += blockDim.x) {
    const int c = k >> 6, l = k & 63;
    if (l < cnt) {
      const size_t i = dstOffset + j0 + l;
      tv[(((i >> lbits) * (size_t)C + c) << lbits) | (i & (L - 1))] = sm[l * C + c];
    }
  }
}

// BASELINE config 2 "TileVector AoSoA load/store": read every channel of every element, scale, write back.
// The buffer of whole tiles is contiguous, so the AoSoA sweep is a flat 16-byte-per-lane stream.
// NT: non-temporal stores for buffers of >= 128 MB (a template parameter: see scan_kernel in primitives.hip)
template <bool NT> __global__ __launch_bounds__(256) void tv_scale_kernel(float4 *buf, size_t nvec, float alpha) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 *b = reinterpret_cast<f4 *>(buf);
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < nvec; i += stride) {
    f4 v = b[i];
    v *= alpha;
    if constexpr (NT) __builtin_nontemporal_store(v, b + i);
    else b[i] = v;
  }
}

static int log2i(size_t L) {
  int b = 0;
  while (((size_t)1 << b) < L) ++b;
  return b;
}

template <class W> static void tv_fill(Launch &L, void *buf, size_t total, W val) {
  if (!total) return;
  hipLaunchKernelGGL((tv_fill_kernel<W>), dim3(ceil_div(total, 256)), dim3(256), 0, L.stream, (W *)buf, total, val);
}

}  // namespace zsr

// ======================================================================================= C ABI
extern "C" {

zs_rocm_allocator *allocator(int memsrc, int8_t devid) { return new zs_rocm_allocator{memsrc, devid, 0}; }
// py_interop/Allocator.cpp:14-19: get_virtual_memory_source(mre, devid, reservedSpace, "STACK")
zs_rocm_allocator *allocator_virtual(int memsrc, int8_t devid, size_t reservedSpace) {
  return new zs_rocm_allocator{memsrc, devid, reservedSpace ? reservedSpace : (size_t)1 << 36};
}
void del_allocator(zs_rocm_allocator *a) { delete a; }
void del_allocator_virtual(zs_rocm_allocator *a) { delete a; }
int mem_enum__host(void) { return 0; }
int mem_enum__device(void) { return 1; }
int mem_enum__um(void) { return 2; }

zs_rocm_property_tags *property_tags(const char *const *names, const int *sizes, size_t n) {
  auto *t = new zs_rocm_property_tags;
  for (size_t i = 0; i < n; ++i) {
    t->names.emplace_back(names[i]);
    t->sizes.push_back(sizes[i]);
  }
  return t;
}
void del_property_tags(zs_rocm_property_tags *t) { delete t; }
void property_tags_get_item(zs_rocm_property_tags *t, size_t index, const char **name, size_t *size) {
  *name = t->names[index].c_str();
  *size = (size_t)t->sizes[index];
}
size_t property_tags_get_size(zs_rocm_property_tags *t) { return t->names.size(); }

// ---- Vector<T>: py_interop/VectorInstantiations.cpp:8-170.  SFX is empty or _virtual: both name the same object type, the
// allocator the container was created with decides how its storage grows.
#define ZSR_DEFINE_VECTOR_TYPE(T) struct zs_rocm_vector_##T { VecBuf b; };
#define ZSR_DEFINE_VECTOR(T, SFX)                                                                          \
  zs_rocm_vector_##T *container__v_##T##SFX(zs_rocm_allocator *a, size_t n) {                              \
    auto *v = new zs_rocm_vector_##T;                                                                      \
    VecBuf tmp(*a, n, sizeof(T));                                                                          \
    v->b.swap(tmp);                                                                                        \
    return v;                                                                                              \
  }                                                                                                        \
  void del_container__v_##T##SFX(zs_rocm_vector_##T *v) { delete v; }                                      \
  void relocate_container__v_##T##SFX(zs_rocm_vector_##T *v, int m, int8_t d) { v->b.relocate(m, d); }     \
  void resize_container__v_##T##SFX(zs_rocm_vector_##T *v, size_t n) { v->b.resize(n); }                   \
  void reset_container__v_##T##SFX(zs_rocm_vector_##T *v, int ch) { v->b.reset(ch); }                      \
  size_t container_size__v_##T##SFX(const zs_rocm_vector_##T *v) { return v->b.size; }                     \
  size_t container_capacity__v_##T##SFX(const zs_rocm_vector_##T *v) { return v->b.cap; }                  \
  T get_val_i_container__v_##T##SFX(zs_rocm_vector_##T *v, size_t i) { /* getVal(i): 1-element copy */     \
    T r{};                                                                                                 \
    zs_rocm_allocator h{0, -1};                                                                            \
    mem_copy(h, &r, v->b.alloc, v->b.data + i * sizeof(T), sizeof(T));                                     \
    return r;                                                                                              \
  }                                                                                                        \
  void set_val_i_container__v_##T##SFX(zs_rocm_vector_##T *v, size_t i, T x) {                             \
    zs_rocm_allocator h{0, -1};                                                                            \
    mem_copy(v->b.alloc, v->b.data + i * sizeof(T), h, &x, sizeof(T));                                     \
  }                                                                                                        \
  T get_val_container__v_##T##SFX(zs_rocm_vector_##T *v) { return get_val_i_container__v_##T##SFX(v, 0); } \
  void set_val_container__v_##T##SFX(zs_rocm_vector_##T *v, T x) { set_val_i_container__v_##T##SFX(v, 0, x); } \
  void copy_to_container__v_##T##SFX(zs_rocm_vector_##T *v, void *src) { /* assignVals(host src) */        \
    zs_rocm_allocator h{0, -1};                                                                            \
    mem_copy(v->b.alloc, v->b.data, h, src, v->b.size * sizeof(T));                                        \
  }                                                                                                        \
  void copy_from_container__v_##T##SFX(zs_rocm_vector_##T *v, void *dst) { /* retrieveVals(host dst) */    \
    zs_rocm_allocator h{0, -1};                                                                            \
    mem_copy(h, dst, v->b.alloc, v->b.data, v->b.size * sizeof(T));                                        \
  }                                                                                                        \
  T *get_handle_container__v_##T##SFX(zs_rocm_vector_##T *v) { return (T *)v->b.data; }                    \
  zs_rocm_vector_view_lite *pyview__v_##T##SFX(zs_rocm_vector_##T *v) { return new zs_rocm_vector_view_lite{v->b.data}; } \
  zs_rocm_vector_view_lite *pyview__v_const_##T##SFX(const zs_rocm_vector_##T *v) {                        \
    return new zs_rocm_vector_view_lite{v->b.data};                                                        \
  }                                                                                                        \
  aosoa_iterator_##T##_1 get_iterator_1__v_##T##SFX(zs_rocm_vector_##T *v, uint32_t id) {                  \
    return aosoa_iterator_##T##_1{(T *)v->b.data, id, 0u, 0u, 1u}; /* aos ctor, GenericIterator.hpp:71-72 */ \
  }                                                                                                        \
  aosoa_iterator_const_##T##_1 get_iterator_1__v_const_##T##SFX(const zs_rocm_vector_##T *v, uint32_t id) { \
    return aosoa_iterator_const_##T##_1{(const T *)v->b.data, id, 0u, 0u, 1u};                             \
  }                                                                                                        \
  aosoa_iterator_##T##_3 get_iterator_3__v_##T##SFX(zs_rocm_vector_##T *v, uint32_t id) {                  \
    return aosoa_iterator_##T##_3{(T *)v->b.data, id, 0u, 0u, 3u};                                         \
  }                                                                                                        \
  aosoa_iterator_const_##T##_3 get_iterator_3__v_const_##T##SFX(const zs_rocm_vector_##T *v, uint32_t id) { \
    return aosoa_iterator_const_##T##_3{(const T *)v->b.data, id, 0u, 0u, 3u};                             \
  }
#define ZSR_DEFINE_VECTOR_BOTH(T)                                                     \
  ZSR_DEFINE_VECTOR_TYPE(T)                                                           \
  ZSR_DEFINE_VECTOR(T, )                                                              \
  ZSR_DEFINE_VECTOR(T, _virtual)                                                      \
  void del_pyview__v_##T(zs_rocm_vector_view_lite *v) { delete v; }                   \
  void del_pyview__v_const_##T(zs_rocm_vector_view_lite *v) { delete v; }             \
  T *container_data__v_##T(zs_rocm_vector_##T *v) { return (T *)v->b.data; }
ZSR_DEFINE_VECTOR_BOTH(int)
ZSR_DEFINE_VECTOR_BOTH(float)
ZSR_DEFINE_VECTOR_BOTH(double)

// ---- TileVector<T, L>
#define ZSR_DEFINE_TILEVECTOR_TYPE(T, LW) struct zs_rocm_tv_##T##_##LW { TileVec *tv; };
#define ZSR_DEFINE_TILEVECTOR(T, LW, SFX)                                                                     \
  zs_rocm_tv_##T##_##LW *container__tv_##T##_##LW##SFX(zs_rocm_allocator *a, const zs_rocm_property_tags *t,       \
                                                  size_t n) {                                                 \
    return new zs_rocm_tv_##T##_##LW{new TileVec(*a, *t, n, LW, sizeof(T))};                                  \
  }                                                                                                           \
  void del_container__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v) {                                               \
    delete v->tv;                                                                                             \
    delete v;                                                                                                 \
  }                                                                                                           \
  void relocate_container__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v, int m, int8_t d) { v->tv->relocate(m, d); } \
  void resize_container__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v, size_t n) { v->tv->resize(n); }              \
  void reset_container__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v, int ch) { v->tv->buf.reset(ch); }             \
  size_t container_size__tv_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v) { return v->tv->size; }                \
  size_t container_capacity__tv_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v) {                                  \
    return v->tv->numChannels ? v->tv->buf.cap / (size_t)v->tv->numChannels : 0; /* TileVector.hpp:382 */     \
  }                                                                                                           \
  int property_offset__tv_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v, const char *name) {                      \
    int i = v->tv->find(name);                                                                                \
    return i < 0 ? -1 : v->tv->offsets[i]; /* getPropertyOffset, TileVector.hpp:528-535 */                    \
  }                                                                                                           \
  int property_size__tv_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v, const char *name) {                        \
    int i = v->tv->find(name);                                                                                \
    return i < 0 ? -1 : v->tv->sizes[i];                                                                      \
  }                                                                                                           \
  aosoa_iterator_##T##_1 get_iterator_1__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v, uint32_t id,                 \
                                                       uint32_t chnOffset) {                                  \
    /* aosoa_iterator(aosoa, ptr, id, tileSize, chnOffset, numChns), GenericIterator.hpp:76-82 */             \
    aosoa_iterator_##T##_1 it;                                                                                \
    it.base = (T *)v->tv->buf.data + (size_t)chnOffset * LW;                                                  \
    it.idx = id;                                                                                              \
    it.numTileBits = (uint32_t)log2i(LW);                                                                     \
    it.tileMask = LW - 1;                                                                                     \
    it.numChns = (uint32_t)v->tv->numChannels;                                                                \
    return it;                                                                                                \
  }                                                                                                           \
  aosoa_iterator_const_##T##_1 get_iterator_1__tv_const_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v, uint32_t id, \
                                                                       uint32_t chnOffset) {                  \
    return aosoa_iterator_const_##T##_1{(const T *)v->tv->buf.data + (size_t)chnOffset * LW, id, (uint32_t)log2i(LW), LW - 1, \
                                       (uint32_t)v->tv->numChannels};                                         \
  }                                                                                                           \
  aosoa_iterator_##T##_3 get_iterator_3__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v, uint32_t id, uint32_t chnOffset) { \
    return aosoa_iterator_##T##_3{(T *)v->tv->buf.data + (size_t)chnOffset * LW, id, (uint32_t)log2i(LW), LW - 1, \
                                 (uint32_t)v->tv->numChannels};                                               \
  }                                                                                                           \
  aosoa_iterator_const_##T##_3 get_iterator_3__tv_const_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v, uint32_t id, \
                                                                       uint32_t chnOffset) {                  \
    return aosoa_iterator_const_##T##_3{(const T *)v->tv->buf.data + (size_t)chnOffset * LW, id, (uint32_t)log2i(LW), LW - 1, \
                                       (uint32_t)v->tv->numChannels};                                         \
  }                                                                                                           \
  zs_rocm_tv_view_lite *pyview__tv_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v) {                                \
    return new zs_rocm_tv_view_lite{v->tv->buf.data, v->tv->numChannels};                                     \
  }                                                                                                           \
  zs_rocm_tv_view_lite *pyview__tv_const_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v) {                    \
    return new zs_rocm_tv_view_lite{v->tv->buf.data, v->tv->numChannels};                                     \
  }                                                                                                           \
  zs_rocm_tv_named_view_lite *pyview__tvn_##T##_##LW##SFX(zs_rocm_tv_##T##_##LW *v) {                         \
    return new zs_rocm_tv_named_view_lite{v->tv->buf.data, v->tv->numChannels, v->tv->tagNames, v->tv->tagOffsets, \
                                          v->tv->tagSizes, (int)v->tv->names.size()};                         \
  }                                                                                                           \
  zs_rocm_tv_named_view_lite *pyview__tvn_const_##T##_##LW##SFX(const zs_rocm_tv_##T##_##LW *v) {             \
    return new zs_rocm_tv_named_view_lite{v->tv->buf.data, v->tv->numChannels, v->tv->tagNames, v->tv->tagOffsets, \
                                          v->tv->tagSizes, (int)v->tv->names.size()};                         \
  }                                                                                                           \
  void append_properties__rocm_tv_##T##_##LW##SFX(zs_rocm_policy *pol, zs_rocm_tv_##T##_##LW *v,                   \
                                             const zs_rocm_property_tags *tags) {                             \
    TileVec &o = *v->tv;                                                                                      \
    zs_rocm_property_tags merged{o.names, o.sizes};                                                           \
    bool grew = false;                                                                                        \
    for (size_t k = 0; k < tags->names.size(); ++k) {                                                         \
      int i = o.find(tags->names[k].c_str());                                                                 \
      if (i >= 0) {                                                                                           \
        if (o.sizes[i] != tags->sizes[k]) { /* the reference throws here (TileVector.hpp:613-617) */          \
          fprintf(stderr, "[zs_rocm] append_channels: property '%s' changes width %d -> %d, ignored\n",    \
                  tags->names[k].c_str(), o.sizes[i], tags->sizes[k]);                                        \
          return;                                                                                             \
        }                                                                                                     \
        continue;                                                                                             \
      }                                                                                                       \
      merged.names.push_back(tags->names[k]);                                                                 \
      merged.sizes.push_back(tags->sizes[k]);                                                                 \
      grew = true;                                                                                            \
    }                                                                                                         \
    if (!grew) return;                                                                                        \
    auto *nt = new TileVec(o.buf.alloc, merged, o.size, LW, sizeof(T));                                       \
    {                                                                                                         \
      Launch L(pol, "append_channels");                                                                       \
      size_t ntiles = o.tiles(o.size);                                                                        \
      size_t total = ntiles * (size_t)nt->numChannels * LW;                                                   \
      using W = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;                                       \
      if (total)                                                                                              \
        hipLaunchKernelGGL((tv_append_copy_kernel<W>), dim3(ceil_div(total, 256)), dim3(256), 0, L.stream,    \
                           (const W *)o.buf.data, (W *)nt->buf.data, ntiles, o.numChannels, nt->numChannels,  \
                           log2i(LW));                                                                        \
      /* the old buffer is released below: the copy must have finished */                                     \
      ZSR_CHECK(hipStreamSynchronize(L.stream));                                                              \
    }                                                                                                         \
    delete v->tv;                                                                                             \
    v->tv = nt;                                                                                               \
  }
#define ZSR_DEFINE_TILEVECTOR_ONCE(T, LW)                                                                     \
  size_t container_num_channels__tv_##T##_##LW(const zs_rocm_tv_##T##_##LW *v) { return (size_t)v->tv->numChannels; } \
  T *container_data__tv_##T##_##LW(zs_rocm_tv_##T##_##LW *v) { return (T *)v->tv->buf.data; }                 \
  void zs_rocm_fill__tv_##T##_##LW(zs_rocm_policy *pol, zs_rocm_tv_##T##_##LW *v, T val) {                    \
    Launch L(pol, "tv_reset");                                                                                \
    using W = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;                                         \
    W w;                                                                                                      \
    std::memcpy(&w, &val, sizeof(T));                                                                         \
    tv_fill<W>(L, v->tv->buf.data, v->tv->tiles(v->tv->size) * LW * (size_t)v->tv->numChannels, w);           \
  }                                                                                                           \
  void zs_rocm_reorder__tv_##T##_##LW(zs_rocm_policy *pol, zs_rocm_tv_##T##_##LW *v, const int *map,          \
                                      int gather) {                                                           \
    TileVec &o = *v->tv;                                                                                      \
    zs_rocm_property_tags same{o.names, o.sizes};                                                             \
    auto *nt = new TileVec(o.buf.alloc, same, o.size, LW, sizeof(T));                                         \
    {                                                                                                         \
      Launch L(pol, "tv_reorder");                                                                            \
      using W = std::conditional_t<sizeof(T) == 4, uint32_t, uint64_t>;                                       \
      if (o.size) {                                                                                           \
        if (gather)                                                                                           \
          hipLaunchKernelGGL((tv_reorder_kernel<W, true>), dim3(ceil_div(o.size, 256)), dim3(256), 0,         \
                             L.stream, (const W *)o.buf.data, (W *)nt->buf.data, o.size, o.numChannels,       \
                             log2i(LW), map);                                                                 \
        else                                                                                                  \
          hipLaunchKernelGGL((tv_reorder_kernel<W, false>), dim3(ceil_div(o.size, 256)), dim3(256), 0,        \
                             L.stream, (const W *)o.buf.data, (W *)nt->buf.data, o.size, o.numChannels,       \
                             log2i(LW), map);                                                                 \
      }                                                                                                       \
      ZSR_CHECK(hipStreamSynchronize(L.stream));                                                              \
    }                                                                                                         \
    delete v->tv;                                                                                             \
    v->tv = nt;                                                                                               \
  }
#define ZSR_DEFINE_TILEVECTOR_BOTH(T, LW)                                                   \
  ZSR_DEFINE_TILEVECTOR_TYPE(T, LW)                                                         \
  ZSR_DEFINE_TILEVECTOR(T, LW, )                                                            \
  ZSR_DEFINE_TILEVECTOR(T, LW, _virtual)                                                    \
  ZSR_DEFINE_TILEVECTOR_ONCE(T, LW)                                                         \
  void del_pyview__tv_##T##_##LW(zs_rocm_tv_view_lite *v) { delete v; }                     \
  void del_pyview__tv_const_##T##_##LW(zs_rocm_tv_view_lite *v) { delete v; }               \
  void del_pyview__tvn_##T##_##LW(zs_rocm_tv_named_view_lite *v) { delete v; }              \
  void del_pyview__tvn_const_##T##_##LW(zs_rocm_tv_named_view_lite *v) { delete v; }

ZSR_DEFINE_TILEVECTOR_BOTH(int, 8)
ZSR_DEFINE_TILEVECTOR_BOTH(int, 32)
ZSR_DEFINE_TILEVECTOR_BOTH(int, 64)
ZSR_DEFINE_TILEVECTOR_BOTH(int, 512)
ZSR_DEFINE_TILEVECTOR_BOTH(float, 8)
ZSR_DEFINE_TILEVECTOR_BOTH(float, 32)
ZSR_DEFINE_TILEVECTOR_BOTH(float, 64)
ZSR_DEFINE_TILEVECTOR_BOTH(float, 512)
ZSR_DEFINE_TILEVECTOR_BOTH(double, 8)
ZSR_DEFINE_TILEVECTOR_BOTH(double, 32)
ZSR_DEFINE_TILEVECTOR_BOTH(double, 64)
ZSR_DEFINE_TILEVECTOR_BOTH(double, 512)

// ---- raw AoSoA kernels
void zs_rocm_tv_from_aos_f32(zs_rocm_policy *pol, const float *aos, size_t n, int C, int Lw, float *tv) {
  Launch L(pol, "tv_from_aos");
  if (!n) return;
  hipLaunchKernelGGL(tv_from_aos_kernel, dim3(ceil_div(n, 64)), dim3(256), 64 * C * sizeof(float), L.stream, aos, n, C,
                     log2i((size_t)Lw), tv);
}
void zs_rocm_tv_to_aos_f32(zs_rocm_policy *pol, const float *tv, size_t n, int C, int Lw, float *aos) {
  Launch L(pol, "tv_to_aos");
  if (!n) return;
  hipLaunchKernelGGL(tv_to_aos_kernel, dim3(ceil_div(n, 64)), dim3(256), 64 * C * sizeof(float), L.stream, tv, n, C,
                     log2i((size_t)Lw), aos);
}
void zs_rocm_tv_scale_f32(zs_rocm_policy *pol, float *tv, size_t n, int C, int Lw, float alpha) {
  Launch L(pol, "tv_scale");
  size_t total = ((n + Lw - 1) / Lw) * (size_t)Lw * (size_t)C;  // whole tiles; padding lanes are scaled too (harmless)
  size_t nvec = total / 4;                                       // L >= 8 -> total % 4 == 0
  if (!nvec) return;
  unsigned grid = (unsigned)std::min<size_t>(ceil_div(nvec, 256), 256 * 16);
  if (nvec * sizeof(float4) >= ((size_t)128 << 20)) hipLaunchKernelGGL(tv_scale_kernel<true>, dim3(grid), dim3(256), 0, L.stream, (float4 *)tv, nvec, alpha);
  else hipLaunchKernelGGL(tv_scale_kernel<false>, dim3(grid), dim3(256), 0, L.stream, (float4 *)tv, nvec, alpha);
}
void zs_rocm_tv_gather_rows_f32(zs_rocm_policy *pol, const float *tv, const int *map, size_t m, int C, int Lw, float *aos) {
  Launch L(pol, "tv_gather_rows");
  if (!m) return;
  hipLaunchKernelGGL(tv_gather_rows_kernel, dim3(ceil_div(m, 64)), dim3(256), 64 * C * sizeof(float), L.stream, tv, map, m, C,
                     log2i((size_t)Lw), aos);
}
void zs_rocm_tv_scatter_rows_f32(zs_rocm_policy *pol, const float *aos, size_t m, int C, int Lw, float *tv, size_t dstOffset) {
  Launch L(pol, "tv_scatter_rows");
  if (!m) return;
  hipLaunchKernelGGL(tv_scatter_rows_kernel, dim3(ceil_div(m, 64)), dim3(256), 64 * C * sizeof(float), L.stream, aos, m, C,
                     log2i((size_t)Lw), tv, dstOffset);
}
void zs_rocm_tv_gather_channels_f32(zs_rocm_policy *pol, const float *src, float *dst, size_t n, int C, int Lw, const int *map,
                                    unsigned long long channelMask) {
  Launch L(pol, "tv_gather_channels");
  if (!n || C > 64) return;
  hipLaunchKernelGGL(tv_gather_channels_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, (const uint32_t *)src, (uint32_t *)dst, n,
                     C, log2i((size_t)Lw), map, channelMask);
}
void zs_rocm_tv_gather_f32(zs_rocm_policy *pol, const float *src, float *dst, size_t n, int C, int Lw, const int *map) {
  Launch L(pol, "tv_gather");
  if (!n) return;
  hipLaunchKernelGGL((tv_reorder_kernel<uint32_t, true>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream,
                     (const uint32_t *)src, (uint32_t *)dst, n, C, log2i((size_t)Lw), map);
}

}  // extern "C"
