// primitives.hip -- zs::reduce / inclusive_scan / exclusive_scan / radix_sort(_pair) for gfx950.
//
// Replaces the cub calls of CudaExecutionPolicy (cuda/execution/ExecutionPolicy.cuh:560-867):
//   reduce      two-level: 16-byte vector loads, wave64 shuffle tree, LDS across the 4 waves of a
//               block, <= 2048 partials, one finishing block.                       4 B/elem (i32)
//   scan        single pass, decoupled look-back (Merrill & Garland) over 4096-element tiles:
//               tiles take a dynamic ticket (no dispatch-order assumption), publish
//               {status,value} as ONE 8-byte agent-scope store (write-through, no fence needed;
//               guide G16 "R2 granule"), wave 0 looks back 64 tiles at a time.        8 B/elem (i32)
//   radix sort  8-bit LSD passes on the bit window [sbit, ebit); signed keys are ordered by
//               XOR-ing the sign bit inside the digit extraction (execution/ExecutionPolicy.hpp:
//               485-490) so no copy-in/copy-out kernels exist (the reference spends 4 extra
//               kernels + 4 temp vectors on that, ExecutionPolicy.cuh:794-820); the first pass
//               reads the caller's iterator and the last pass writes the caller's iterator.
//               Per pass: tile histogram -> scan of (digit, tile) counts -> stable scatter with
//               wave-level multisplit ranking (ballot match) + per-wave LDS digit counters.
// All kernels address memory through `Port`s (py_interop/GenericIterator.hpp:88-104), so Vector,
// AoS and TileVector-channel iterators take the same path.
#include <algorithm>
#include <limits>
#include <type_traits>

#include "common.hpp"
#include "../../include/zensim_rocm/merge_sort.hpp"

namespace zsr {

template <int OP, class T> __host__ __device__ __forceinline__ T combine(T a, T b) {
  if constexpr (std::is_integral_v<T> && (OP == OP_PLUS || OP == OP_MUL)) {
    using U = std::make_unsigned_t<T>;  // wrap-around, no signed-overflow UB
    if constexpr (OP == OP_PLUS) return (T)((U)a + (U)b);
    else return (T)((U)a * (U)b);
  } else
    return apply<OP>(a, b);
}

template <int OP, class T> constexpr T identity_of() {
  if constexpr (OP == OP_PLUS) return (T)0;
  else if constexpr (OP == OP_MUL) return (T)1;
  else if constexpr (OP == OP_MIN) return std::numeric_limits<T>::max();
  else return std::numeric_limits<T>::lowest();
}

template <int OP, class T> __device__ __forceinline__ T wave_reduce(T v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = combine<OP>(v, shfl_down(v, d));
  return v;  // lane 0 holds the result
}

// ======================================================================================= reduce
constexpr int RED_BLOCK = 256;
constexpr int RED_MAX_BLOCKS = 2048;  // 256 CUs x 8 blocks (guide G11)

template <int OP, class T> __device__ __forceinline__ T block_reduce(T v) {
  __shared__ T sm[RED_BLOCK / 64];
  v = wave_reduce<OP>(v);
  if (lane_id() == 0) sm[wave_id()] = v;
  __syncthreads();
  T r = identity_of<OP, T>();
  if (threadIdx.x < RED_BLOCK / 64) r = sm[threadIdx.x];
  if (wave_id() == 0) {
#pragma unroll
    for (int d = RED_BLOCK / 128; d > 0; d >>= 1) r = combine<OP>(r, shfl_down(r, d));
  }
  return r;  // thread 0
}

// (a single-launch form -- last workgroup to arrive folds the partials -- was measured in r02: the agent-scope release every workgroup needs
// before it counts itself in costs more than the second launch: 64 M ints 0.100 ms instead of 0.043, 1 M 8.8 us instead of 7.5)
template <int OP, class T, bool VEC>
__global__ __launch_bounds__(RED_BLOCK) void reduce_partial_kernel(Port<const T> in, size_t n, T *partials) {
  T acc = identity_of<OP, T>();
  const size_t tid = (size_t)blockIdx.x * RED_BLOCK + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * RED_BLOCK;
  if constexpr (VEC) {  // contiguous + 16-byte aligned: 16 B per lane per load
    constexpr int V = 16 / sizeof(T);
    struct alignas(16) Vec { T v[V]; };
    const Vec *p = reinterpret_cast<const Vec *>(in.base + in.idx);
    const size_t nv = n / V;
    for (size_t i = tid; i < nv; i += nthreads) {
      Vec x = p[i];
#pragma unroll
      for (int j = 0; j < V; ++j) acc = combine<OP>(acc, x.v[j]);
    }
    for (size_t i = nv * V + tid; i < n; i += nthreads) acc = combine<OP>(acc, in[i]);
  } else {
    for (size_t i = tid; i < n; i += nthreads) acc = combine<OP>(acc, in[i]);
  }
  acc = block_reduce<OP>(acc);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

template <int OP, class T>
__global__ __launch_bounds__(RED_BLOCK) void reduce_final_kernel(const T *partials, int np, T init, Port<T> out) {
  T acc = identity_of<OP, T>();
  for (int i = threadIdx.x; i < np; i += RED_BLOCK) acc = combine<OP>(acc, partials[i]);
  acc = block_reduce<OP>(acc);
  if (threadIdx.x == 0) out[0] = combine<OP>(init, acc);
}

template <int OP, class T> static void reduce_impl(Launch &L, Port<const T> in, size_t n, Port<T> out, T init) {
  size_t perBlock = (size_t)RED_BLOCK * (64 / sizeof(T));  // 64 B per thread per grid-stride round
  int nb = (int)std::min<size_t>(RED_MAX_BLOCKS, std::max<size_t>(1, (n + perBlock - 1) / perBlock));
  T *partials = (T *)L.temp(sizeof(T) * RED_MAX_BLOCKS);
  const bool vec = in.contiguous() && (((uintptr_t)(in.base + in.idx)) % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((reduce_partial_kernel<OP, T, true>), dim3(nb), dim3(RED_BLOCK), 0, L.stream, in, n, partials);
  else
    hipLaunchKernelGGL((reduce_partial_kernel<OP, T, false>), dim3(nb), dim3(RED_BLOCK), 0, L.stream, in, n, partials);
  hipLaunchKernelGGL((reduce_final_kernel<OP, T>), dim3(1), dim3(RED_BLOCK), 0, L.stream, (const T *)partials, nb, init, out);
}

template <class T> static void reduce_dispatch(Launch &L, Port<const T> in, size_t n, Port<T> out, T init, int op) {
  switch (op) {
    case OP_PLUS: reduce_impl<OP_PLUS>(L, in, n, out, init); break;
    case OP_MUL: reduce_impl<OP_MUL>(L, in, n, out, init); break;
    case OP_MIN: reduce_impl<OP_MIN>(L, in, n, out, init); break;
    default: reduce_impl<OP_MAX>(L, in, n, out, init); break;
  }
}

// ======================================================================================= scan
constexpr int SCAN_BLOCK = 512;
enum : unsigned { ST_INVALID = 0, ST_AGG = 1, ST_PREFIX = 2 };

// tile descriptor storage.  4-byte values: one packed u64 {status<<32 | bits}.  8-byte values: a flag
// word plus separate aggregate / prefix slots, value written through and drained before the flag.
template <class T, int W = sizeof(T)> struct Desc;
// The status word carries the GENERATION of the call (gen << 2 | state): descriptors left behind by earlier calls in the stream's
// dedicated control block read as invalid, so a scan of up to kCtlScanTiles tiles needs no memset launch in front of it.  (Temporary
// memory is zeroed and used with generation 0.)
template <class T> struct Desc<T, 4> {
  unsigned long long *d;
  unsigned gen;
  static size_t bytes(size_t tiles) { return tiles * 8; }
  __host__ __device__ explicit Desc(void *p, size_t, unsigned g) : d((unsigned long long *)p), gen(g << 2) {}
  __device__ __forceinline__ void publish(size_t tile, unsigned st, T v) const {
    unsigned bits;
    __builtin_memcpy(&bits, &v, 4);
    __hip_atomic_store(&d[tile], ((unsigned long long)(gen | st) << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ unsigned poll(size_t tile, T &v) const {
    unsigned long long x = __hip_atomic_load(&d[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned bits = (unsigned)x;
    __builtin_memcpy(&v, &bits, 4);
    const unsigned s = (unsigned)(x >> 32);
    return (s & ~3u) == gen ? (s & 3u) : (unsigned)ST_INVALID;
  }
};
template <class T> struct Desc<T, 8> {
  unsigned *flag;
  unsigned long long *agg, *pre;
  unsigned gen;
  static size_t bytes(size_t tiles) { return tiles * 24; }
  __host__ __device__ explicit Desc(void *p, size_t tiles, unsigned g)
      : flag((unsigned *)p + 0), agg((unsigned long long *)p + tiles), pre((unsigned long long *)p + 2 * tiles), gen(g << 2) {}
  // layout: [tiles x u64 region used for flags (first 4 B of each 8)] [agg] [pre]; flags indexed densely
  __device__ __forceinline__ void publish(size_t tile, unsigned st, T v) const {
    unsigned long long bits;
    __builtin_memcpy(&bits, &v, 8);
    __hip_atomic_store(st == ST_AGG ? &agg[tile] : &pre[tile], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // value is at the coherence point before the flag
    __hip_atomic_store(&flag[tile], gen | st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ unsigned poll(size_t tile, T &v) const {
    unsigned st = __hip_atomic_load(&flag[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    st = (st & ~3u) == gen ? (st & 3u) : (unsigned)ST_INVALID;
    if (st != ST_INVALID) {
      unsigned long long bits =
          __hip_atomic_load(st == ST_AGG ? &agg[tile] : &pre[tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_memcpy(&v, &bits, 8);
    }
    return st;
  }
};

// SCAN_ROWS = rows of 512 x (16 B / sizeof(T)) elements per tile.  Large inputs use 8 rows (16384 4-byte elements per
// tile): the dynamic tile ticket is one device-wide atomic counter, which MI355X serves at ~90 increments/us, so 4096-
// element tiles cap a 64M-element scan at ~0.18 ms of pure ticket time (measured 2.2 TB/s); small inputs keep 2 rows
// so that a 1M-element scan still spreads over all 256 CUs.  512 threads x 8 rows instead of 256 x 16 (same tile): half
// the registers per thread and twice the loads in flight per tile, 3.0 -> 3.35 TB/s at 64 M (1024 x 4: the same).
template <int OP, class T, bool EXCL, int SCAN_ROWS, bool NTSTORE>
__global__ __launch_bounds__(SCAN_BLOCK) void scan_kernel(Port<const T> in, Port<T> out, size_t n, T init, void *descMem,
                                                          size_t layoutTiles, unsigned *ticket, unsigned gen, unsigned ticketBase) {
  constexpr int V = 16 / sizeof(T);
  constexpr int ROWW = SCAN_BLOCK * V;       // elements per row
  constexpr int TILE = ROWW * SCAN_ROWS;     // 4096 (4-byte) / 2048 (8-byte)
  constexpr int NW = SCAN_BLOCK / 64;
  const T ident = identity_of<OP, T>();
  Desc<T> desc(descMem, layoutTiles, gen);  // (layoutTiles: array length of the 8-byte layout -- the call's tile count, or the fixed
                                            // capacity of the control block)

  __shared__ unsigned sTile;
  __shared__ T sWave[SCAN_ROWS][NW];
  __shared__ T sTilePrefix;
  if (threadIdx.x == 0) sTile = atomicAdd(ticket, 1u) - ticketBase;  // (the dedicated counter is never reset: the host knows its value)
  __syncthreads();
  const size_t tile = sTile;
  const size_t tileBase = tile * (size_t)TILE;
  const int t = threadIdx.x, lane = lane_id(), w = wave_id();

  // ---- load: row k, thread t holds V consecutive elements at tileBase + k*ROWW + t*V
  T x[SCAN_ROWS][V];
  const bool full = tileBase + TILE <= n;
  const bool vec = in.contiguous() && (((uintptr_t)(in.base + in.idx)) % 16 == 0);
#pragma unroll
  for (int k = 0; k < SCAN_ROWS; ++k) {
    const size_t e0 = tileBase + (size_t)k * ROWW + (size_t)t * V;
    if (full && vec) {
      struct alignas(16) Vec { T v[V]; };
      Vec q = *reinterpret_cast<const Vec *>(in.base + in.idx + e0);
#pragma unroll
      for (int j = 0; j < V; ++j) x[k][j] = q.v[j];
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) x[k][j] = (e0 + j < n) ? in[e0 + j] : ident;
    }
  }
  // ---- thread-local inclusive scan per row, wave scan of the row totals
  T inc[SCAN_ROWS];  // inclusive scan over lanes of thread totals
  T tot[SCAN_ROWS];
#pragma unroll
  for (int k = 0; k < SCAN_ROWS; ++k) {
#pragma unroll
    for (int j = 1; j < V; ++j) x[k][j] = combine<OP>(x[k][j - 1], x[k][j]);
    tot[k] = x[k][V - 1];
    T s = tot[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      T o = shfl_up(s, d);
      if (lane >= d) s = combine<OP>(o, s);
    }
    inc[k] = s;
    if (lane == 63) sWave[k][w] = s;
  }
  __syncthreads();
  // ---- exclusive prefix of (row k, wave w) in row-major order + tile aggregate (16 LDS broadcasts)
  T base[SCAN_ROWS];
  T run = ident;
#pragma unroll
  for (int k = 0; k < SCAN_ROWS; ++k) {
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      if (ww == w) base[k] = run;
      run = combine<OP>(run, sWave[k][ww]);
    }
  }
  const T aggregate = run;
  // ---- decoupled look-back by wave 0
  if (w == 0) {
    if (tile == 0) {
      if (lane == 0) {
        desc.publish(0, ST_PREFIX, aggregate);
        sTilePrefix = ident;
      }
    } else {
      if (lane == 0) desc.publish(tile, ST_AGG, aggregate);
      T running = ident;
      long long pred = (long long)tile - 1 - lane;
      while (true) {
        T v = ident;
        unsigned st = ST_PREFIX;  // before the first tile: prefix = identity
        if (pred >= 0) st = desc.poll((size_t)pred, v);
        if (__any(st == ST_INVALID)) {
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        unsigned long long pm = __ballot(st == ST_PREFIX);
        if (pm) {
          int first = __ffsll((long long)pm) - 1;  // nearest predecessor holding an inclusive prefix
          T c = lane <= first ? v : ident;
          c = wave_reduce<OP>(c);
          running = combine<OP>(running, shfl(c, 0));
          break;
        }
        T c = wave_reduce<OP>(v);
        running = combine<OP>(running, shfl(c, 0));
        pred -= 64;
      }
      if (lane == 0) {
        desc.publish(tile, ST_PREFIX, combine<OP>(running, aggregate));
        sTilePrefix = running;
      }
    }
  }
  __syncthreads();
  const T tp = EXCL ? combine<OP>(init, sTilePrefix) : sTilePrefix;
  // ---- write
#pragma unroll
  for (int k = 0; k < SCAN_ROWS; ++k) {
    T lanePrev = shfl_up(inc[k], 1);
    if (lane == 0) lanePrev = ident;
    const T pfx = combine<OP>(tp, combine<OP>(base[k], lanePrev));  // everything before this thread's V elements
    T y[V];
    if constexpr (EXCL) {
      y[0] = pfx;
#pragma unroll
      for (int j = 1; j < V; ++j) y[j] = combine<OP>(pfx, x[k][j - 1]);
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) y[j] = combine<OP>(pfx, x[k][j]);
    }
    const size_t e0 = tileBase + (size_t)k * ROWW + (size_t)t * V;
    const bool ovec = out.contiguous() && (((uintptr_t)(out.base + out.idx)) % 16 == 0);
    if (full && ovec) {
      // large outputs (>= 128 MB: beyond what the memory-side cache keeps for the next kernel anyway) are written with non-temporal
      // stores: 64 M ints 0.147 -> 0.125 ms; 16 M unchanged; non-temporal LOADS of the input measured slower with them (0.137)
      typedef T VecT __attribute__((ext_vector_type(V)));
      VecT q;
#pragma unroll
      for (int j = 0; j < V; ++j) q[j] = y[j];
      // (a template parameter, not a run-time flag: the compiler merges the two stores of a run-time branch into a plain one)
      if constexpr (NTSTORE) __builtin_nontemporal_store(q, reinterpret_cast<VecT *>(out.base + out.idx + e0));
      else *reinterpret_cast<VecT *>(out.base + out.idx + e0) = q;
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j)
        if (e0 + j < n) out[e0 + j] = y[j];
    }
  }
}

template <int OP, class T, bool EXCL, int ROWS> static void scan_launch(Launch &L, Port<const T> in, size_t n, Port<T> out, T init) {
  constexpr size_t TILE = (size_t)SCAN_BLOCK * (16 / sizeof(T)) * ROWS;
  const int ntStore = n * sizeof(T) >= ((size_t)128 << 20) ? 1 : 0;
  const size_t numTiles = (n + TILE - 1) / TILE;
  const size_t dbytes = Desc<T>::bytes(numTiles);
  if (numTiles <= kCtlScanTiles) {
    // the stream's control block: descriptors tagged with a fresh generation, a ticket counter that only ever counts up -> one launch
    unsigned gen, base;
    bool wrapped;
    char *ctl = L.scan_control(numTiles, gen, base, wrapped);
    if (wrapped) ZSR_CHECK(hipMemsetAsync(ctl, 0, kCtlTicket, L.stream));  // generation wrap: start over from clean descriptors
    if (ntStore)
      hipLaunchKernelGGL((scan_kernel<OP, T, EXCL, ROWS, true>), dim3((unsigned)numTiles), dim3(SCAN_BLOCK), 0, L.stream, in, out, n, init,
                         (void *)(ctl + (sizeof(T) == 8 ? kCtlDesc8 : 0)), kCtlScanTiles, (unsigned *)(ctl + kCtlTicket), gen, base);
    else
      hipLaunchKernelGGL((scan_kernel<OP, T, EXCL, ROWS, false>), dim3((unsigned)numTiles), dim3(SCAN_BLOCK), 0, L.stream, in, out, n, init,
                         (void *)(ctl + (sizeof(T) == 8 ? kCtlDesc8 : 0)), kCtlScanTiles, (unsigned *)(ctl + kCtlTicket), gen, base);
    // The host's shadow of the ticket counter was advanced for this launch.  A launch that never ran leaves the device counter behind
    // it, and every later scan on the stream would derive wrong tile numbers: report, and bring both back to a clean control block.
    // (One stream must not be driven by two host threads at once -- reservation and launch are not one atomic step -- nor be captured
    // into a graph: the generation and ticket base are baked into the kernel arguments.)
    if (hipError_t e = hipGetLastError(); e != hipSuccess) {
      report_error(e, "scan_kernel launch", __FILE__, __LINE__);
      L.scan_control_reset();
    }
    return;
  }
  char *mem = (char *)L.temp(dbytes + 256);
  ZSR_CHECK(hipMemsetAsync(mem, 0, dbytes + 256, L.stream));  // descriptors + ticket re-initialised every call
  unsigned *ticket = (unsigned *)(mem + dbytes);
  if (ntStore)
    hipLaunchKernelGGL((scan_kernel<OP, T, EXCL, ROWS, true>), dim3((unsigned)numTiles), dim3(SCAN_BLOCK), 0, L.stream, in, out, n, init,
                       (void *)mem, numTiles, ticket, 0u, 0u);
  else
    hipLaunchKernelGGL((scan_kernel<OP, T, EXCL, ROWS, false>), dim3((unsigned)numTiles), dim3(SCAN_BLOCK), 0, L.stream, in, out, n, init,
                       (void *)mem, numTiles, ticket, 0u, 0u);
}
template <int OP, class T, bool EXCL> static void scan_impl(Launch &L, Port<const T> in, size_t n, Port<T> out, T init) {
  if (n == 0) return;
  // (r02: 4-row and 16-row tiles re-measured at 16 M / 64 M / 256 M elements: 2.5 / 2.9 / 3.2 and 3.0 / 3.0 / 3.2 TB/s against 2.9 / 3.4 / 3.8 for 8 rows)
  // rows per tile by size (r03, i32 exclusive scan on one MI355X, us for 2 / 4 / 8 rows): 250 K 5.4 / 6.5 / 8.6, 500 K 6.5 / 6.3 / 9.5,
  // 1 M 8.6 / 7.7 / 9.1, 2 M 14.2 / 10.2 / 10.9, 4 M 24.3 / 16.5 / 13.7, 8 M 38.7 / 29.7 / 21.3, 16 M 63.9 / 48.2 / 38.7
  static const int rowsOverride = [] { const char *e = getenv("ZS_ROCM_SCAN_ROWS"); return e ? atoi(e) : 0; }();  // measurement only
  const int rows = rowsOverride ? rowsOverride : (n >= (size_t)3000000 ? 8 : (n >= (size_t)400000 ? 4 : 2));
  if (rows == 8) scan_launch<OP, T, EXCL, 8>(L, in, n, out, init);
  else if (rows == 4) scan_launch<OP, T, EXCL, 4>(L, in, n, out, init);
  else scan_launch<OP, T, EXCL, 2>(L, in, n, out, init);
}

template <class T>
static void scan_dispatch(Launch &L, Port<const T> in, size_t n, Port<T> out, T init, int op, bool excl) {
  // the reference's scans are only instantiated for plus / multiplies through the C ABI; min/max are
  // reachable through the C++ face (any associative op)
  switch (op) {
    case OP_PLUS: excl ? scan_impl<OP_PLUS, T, true>(L, in, n, out, init) : scan_impl<OP_PLUS, T, false>(L, in, n, out, init); break;
    case OP_MUL: excl ? scan_impl<OP_MUL, T, true>(L, in, n, out, init) : scan_impl<OP_MUL, T, false>(L, in, n, out, init); break;
    case OP_MIN: excl ? scan_impl<OP_MIN, T, true>(L, in, n, out, init) : scan_impl<OP_MIN, T, false>(L, in, n, out, init); break;
    default: excl ? scan_impl<OP_MAX, T, true>(L, in, n, out, init) : scan_impl<OP_MAX, T, false>(L, in, n, out, init); break;
  }
}

// internal entry used by other translation units (bht canonicalisation, MPM binning)
void exclusive_scan_u32(Launch &L, const unsigned *in, size_t n, unsigned *out) {
  scan_impl<OP_PLUS, unsigned, true>(L, contiguous_port<const unsigned>(in), n, contiguous_port<unsigned>(out), 0u);
}

// ======================================================================================= radix sort
constexpr int RS_BLOCK = 512;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_BLOCK * RS_ITEMS;  // 8192 keys per workgroup
constexpr int RS_NW = RS_BLOCK / 64;

template <class K> struct KeyBits {
  using U = std::make_unsigned_t<K>;
  static constexpr U flip = std::is_signed_v<K> ? (U)((U)1 << (sizeof(K) * 8 - 1)) : (U)0;
  __device__ __forceinline__ static unsigned digit(K k, int st, unsigned mask) {
    return (unsigned)((((U)k) ^ flip) >> st) & mask;
  }
};

// position of (wave w, item k, lane l) inside a tile: w*(64*ITEMS) + k*64 + l -> coalesced and order preserving
// ---------------------------------------------------------------------------------------- onesweep
// One kernel per 8-bit pass (Adinets & Merrill, "Onesweep"): the global digit histograms of ALL passes are taken in one
// upfront read of the keys; in a pass every tile ranks its keys (ballot multisplit, as above), obtains the start of each of
// its 256 digit runs by a chained scan over per-(tile, digit) descriptors (decoupled look-back, dynamic tile ticket) and
// scatters.  Keys are first permuted into tile-local digit order in LDS so that a wave writes contiguous runs.
// Traffic: 4 B/key (histograms) + passes x (4 R + 4 W) instead of passes x (4 + 4 + 4) + count scans.
// small-input path (below, "split + finish")
constexpr int RSS_BLOCK = 1024, RSS_ITEMS = RS_TILE / RSS_BLOCK, RSS_NW = RSS_BLOCK / 64;  // its workgroups: 16 waves x 8 keys per lane -- a
                                                                                         // workgroup has a CU to itself there, waves hide latency
// keys one workgroup of the finish kernel holds in registers and sorts in LDS: 16 per lane (16384) for 4-byte keys, 8 per lane for 8-byte keys
template <class K> constexpr int rss_fitems() { return sizeof(K) == 4 ? 16 : 8; }
template <class K> constexpr unsigned rs_small_cap() { return (unsigned)(RSS_BLOCK * rss_fitems<K>()); }
// largest input: keys that fill only 128 buckets still fit (n/128 +- a few sqrt(n/128): 16 000 +- 500 at 2 048 000 4-byte keys)
template <class K> constexpr size_t rs_small_max_n() { return sizeof(K) == 4 ? (size_t)2048 * 1000 : (size_t)1024 * 1000; }
constexpr size_t RS_SMALL_MAX_TILES = 256;
constexpr int RS_CTL_MODE = 257, RS_CTL_TOP = 259, RS_CTL_EBIT = 260, RS_CTL_BIG = 261, RS_CTL_TICKET = 264, RS_CTL_DONE = 272 /* [16] */,
              RS_CTL_WORDS = 320;
enum : unsigned { RS_FAST = 0, RS_LSD = 1, RS_COPY_IN = 2, RS_COPY_SPLIT = 3, RS_ONE_BIG = 4 };
constexpr unsigned OS_FLAG_AGG = 1u << 30, OS_FLAG_PREFIX = 2u << 30, OS_VAL_MASK = (1u << 30) - 1u;

template <class K, int NPASS>
__global__ __launch_bounds__(256) void radix_global_hist_kernel(Port<const K> keys, size_t n, int sbit, int ebit, unsigned *ghist) {
  __shared__ unsigned h[NPASS][256];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) h[p][threadIdx.x] = 0;
  __syncthreads();
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto count = [&](K k) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int st = sbit + 8 * p;
      if (st < ebit) {
        const int bits = ebit - st < 8 ? ebit - st : 8;
        atomicAdd(&h[p][KeyBits<K>::digit(k, st, (1u << bits) - 1u)], 1u);  // ds_add_u32: full rate (unlike ds_add_f32)
      }
    }
  };
  // eight independent loads in flight per thread (one load per trip left the kernel at 1.9 TB/s: 64 M keys in 140 us)
  constexpr int U = 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    K k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) k[u] = keys[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) count(k[u]);
  }
  for (; i < n; i += stride) count(keys[i]);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < NPASS; ++p)
    if (h[p][threadIdx.x]) atomicAdd(&ghist[p * 256 + threadIdx.x], h[p][threadIdx.x]);
}
template <class K, bool PAIR, int BLOCK, int ITEMS, int LBN>
__global__ __launch_bounds__(BLOCK) void radix_onesweep_kernel(Port<const K> kin, Port<const int> vin, Port<K> kout, Port<int> vout,
                                                               size_t n, int st, unsigned mask, const unsigned *ghistPass,
                                                               unsigned *desc, unsigned *ticket) {
  constexpr int NW = BLOCK / 64, TILE = BLOCK * ITEMS;
  static_assert(BLOCK >= 256 && BLOCK % 256 == 0, "one thread per digit in the first 256 threads");
  __shared__ unsigned cnt[NW][256];      // per-wave digit counters -> per-wave exclusive offsets inside the digit run
  __shared__ unsigned tileStart[256];    // start of digit d inside the tile-local sorted order
  __shared__ unsigned globalStart[256];  // start of this tile's run of digit d in the output
  __shared__ unsigned sWave[4], sWave2[4];
  __shared__ unsigned sTile;
  __shared__ unsigned thist[256];        // the tile's digit histogram, taken BEFORE the ranking (see below)
  __shared__ K keyS[TILE];
  __shared__ int valS[PAIR ? TILE : 1];
  const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
  if (t == 0) sTile = atomicAdd(ticket, 1u);
  for (int i = t; i < NW * 256; i += BLOCK) (&cnt[0][0])[i] = 0;
  if (t < 256) thist[t] = 0;
  // global start of digit t = exclusive scan of this pass's 256-bin histogram (every tile redoes the 256-element scan: cheaper than
  // a launch of its own); partial sums of the four waves go through sWave
  // (the load is issued here and consumed after the ranking)
  const unsigned ghv = t < 256 ? ghistPass[t] : 0u;
  __syncthreads();
  const unsigned tile = sTile;
  const size_t tileBase = (size_t)tile * TILE;
  const unsigned tileCount = (unsigned)(n - tileBase < TILE ? n - tileBase : TILE);

  K key[ITEMS];
  int val[ITEMS];
  unsigned rank[ITEMS];
  const size_t base = tileBase + (size_t)w * (64 * ITEMS) + lane;
  // full tile of plain arrays (the common case): raw pointers with immediate offsets, no bounds checks.  The generic path
  // keeps the TileVector channel addressing (Port::off) and the ragged last tile.
  const bool fast = tileCount == TILE && kin.contiguous() && kout.contiguous() && (!PAIR || (vin.contiguous() && vout.contiguous()));
  if (fast) {
    const K *kp = kin.base + kin.idx + base;
    const int *vp = PAIR ? vin.base + vin.idx + base : nullptr;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      // keys-only passes read their input with non-temporal loads (64 M keys: 0.92-0.93 -> 0.89-0.90 ms, 16 M unchanged; with values
      // alongside it is slower: 1.26 -> 1.36 ms)
      if constexpr (!PAIR) key[k] = __builtin_nontemporal_load(kp + k * 64);
      else key[k] = kp[k * 64];
      if constexpr (PAIR) val[k] = vp[k * 64];
    }
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const size_t i = base + (size_t)k * 64;
      if (i < n) {
        key[k] = kin[i];
        if constexpr (PAIR) val[k] = vin[i];
      }
    }
  }
  // The tile's aggregate is published as early as possible: a plain LDS histogram of the digits (16 ds_add_u32 per thread) right after
  // the loads, ~1 us into the tile, instead of after the ranking (~5 us later).  Measured before this change at 64 M keys: a look-back
  // consumed 42 predecessor descriptors in 20 polling round trips -- most polls ran into tiles that were still ranking -- and the
  // look-back cost 0.36 of the sort's 0.98 ms.  With early aggregates the successors' look-backs find their predecessors published.
  if (fast) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) atomicAdd(&thist[KeyBits<K>::digit(key[k], st, mask)], 1u);
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
      if (base + (size_t)k * 64 < n) atomicAdd(&thist[KeyBits<K>::digit(key[k], st, mask)], 1u);
  }
  __syncthreads();
  unsigned *myDesc = desc + (size_t)tile * 256 + t;
  if (t < 256)
    __hip_atomic_store(myDesc, (tile == 0 ? OS_FLAG_PREFIX : OS_FLAG_AGG) | thist[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  volatile unsigned *wc = cnt[w];
  const unsigned long long lt = lanemask_lt();
  const unsigned ltlo = (unsigned)lt, lthi = (unsigned)(lt >> 32);
  // ballot multisplit, 4 VALU per bit: t = -(bit) (v_bfe_i32), its ballot m (v_cmp), peers &= ~(m ^ t) on both halves
  // (one v_bitop3_b32 each: truth table of a & ~(b ^ c) with a = 0xF0, b = 0xCC, c = 0xAA -> 0x90)
  auto rank_item = [&](int k, bool valid) {
    const unsigned d = valid ? KeyBits<K>::digit(key[k], st, mask) : 0u;
    const unsigned long long vb = __ballot(valid);
    unsigned plo = (unsigned)vb, phi = (unsigned)(vb >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const unsigned tb = (unsigned)__builtin_amdgcn_sbfe((int)d, b, 1);
      const unsigned long long m = __ballot(tb != 0u);
      plo = __builtin_amdgcn_bitop3_b32(plo, (unsigned)m, tb, 0x90);
      phi = __builtin_amdgcn_bitop3_b32(phi, (unsigned)(m >> 32), tb, 0x90);
    }
    // every lane reads its digit's running count (same-address broadcast), then the lowest peer lane bumps it: LDS
    // operations of one wave execute in order, so no cross-lane shuffle of the old value is needed
    const unsigned below = (unsigned)__popc(plo & ltlo) + (unsigned)__popc(phi & lthi);
    const unsigned old = wc[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0) wc[d] = old + (unsigned)__popc(plo) + (unsigned)__popc(phi);
    rank[k] = old + below;
    __builtin_amdgcn_wave_barrier();
  };
  if (fast) {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) rank_item(k, true);
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) rank_item(k, base + (size_t)k * 64 < n);
  }
  __syncthreads();
  // thread t < 256 owns digit t: wave offsets, tile count, chained scan
  unsigned myCount = 0, excl = 0;
  if (t < 256) {
    unsigned run = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const unsigned c = cnt[i][t];
      cnt[i][t] = run;
      run += c;
    }
    myCount = run;  // (== thist[t], published above)
    // exclusive scan of the 256 tile counts -> tileStart (wave level, combined below)
    unsigned s = myCount;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned o = shfl_up(s, d);
      if (lane >= d) s += o;
    }
    if (lane == 63) sWave2[w] = s;
    tileStart[t] = s - myCount;
    unsigned sc = ghv;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned o = shfl_up(sc, d);
      if (lane >= d) sc += o;
    }
    if (lane == 63) sWave[w] = sc;
    const unsigned gstart = sc - ghv;  // + the sums of the lower waves, added after the barrier
    // look back over the predecessors' descriptors of digit t
    if (tile != 0) {
      // look back over the predecessors' descriptors of digit t, LB tiles per round trip: the loads of one batch are
      // independent, so a walk over many aggregate-only tiles (the tiles that started together with this one) costs one L2
      // latency per batch instead of one per tile
      // The first batch is short (in a long launch the tiles are staggered and an inclusive prefix is a few tiles away); when the
      // whole launch starts together (small inputs: every tile is aggregate-only until tile 0's chain reaches it) the walk continues
      // LBN tiles per round trip (64 in the small-tile instantiation).
      long long p = (long long)tile - 1;
      bool done = false;
      auto batch = [&](auto lbTag) {
        constexpr int LB = decltype(lbTag)::value;
        unsigned v[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j)
          v[j] = p - j >= 0 ? __hip_atomic_load(desc + (size_t)(p - j) * 256 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : OS_FLAG_PREFIX;  // before tile 0: an empty inclusive prefix
        // branch-free walk over the batch: `alive` while every descriptor so far was published and none was an inclusive prefix
        int used = 0;
        bool alive = true;
#pragma unroll
        for (int j = 0; j < LB; ++j) {
          const unsigned f = v[j] >> 30;  // 0 not published, 1 aggregate, 2 inclusive prefix
          const bool take = alive && f != 0u;
          excl += take ? (v[j] & OS_VAL_MASK) : 0u;
          used += take ? 1 : 0;
          done = done || (take && f == 2u);
          alive = take && f != 2u;
        }
        p -= used;
        if (!done && used < LB) __builtin_amdgcn_s_sleep(1);  // ran into a tile that has not published yet
      };
      batch(std::integral_constant<int, 8>{});
      while (!done) batch(std::integral_constant<int, LBN>{});
      __hip_atomic_store(myDesc, OS_FLAG_PREFIX | (excl + myCount), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    globalStart[t] = gstart + excl;
  }
  __syncthreads();
  if (t < 256) {
    unsigned b = 0, gb = 0;
    for (int i = 0; i < w; ++i) b += sWave2[i], gb += sWave[i];
    const unsigned ts = tileStart[t] + b;
    tileStart[t] = ts;
    globalStart[t] += gb - ts;  // (wrapping) so that dst = globalStart[d] + position in the tile-sorted order
#pragma unroll
    for (int i = 0; i < NW; ++i) cnt[i][t] += ts;  // tile-sorted position = cnt[w][d] + rank
  }
  __syncthreads();
  // tile-local digit order in LDS
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    if (fast || base + (size_t)k * 64 < n) {
      const unsigned d = KeyBits<K>::digit(key[k], st, mask);
      const unsigned lp = cnt[w][d] + rank[k];
      keyS[lp] = key[k];
      if constexpr (PAIR) valS[lp] = val[k];
    }
  }
  __syncthreads();
  // coalesced runs out
  if (fast) {
    K *ko = kout.base + kout.idx;
    int *vo = PAIR ? vout.base + vout.idx : nullptr;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const unsigned lp = (unsigned)t + (unsigned)k * BLOCK;
      const K kk = keyS[lp];
      const unsigned d = KeyBits<K>::digit(kk, st, mask);
      const unsigned dst = globalStart[d] + lp;
      // (non-temporal stores of these 4-byte run pieces were measured: 64 M keys 0.94 -> 1.02 ms, pairs 1.26 -> 2.06 ms -- they defeat
      // the write combining in L2; non-temporal loads of the pass input: keys 0.88, pairs 1.36 ms: not adopted either)
      ko[dst] = kk;
      if constexpr (PAIR) vo[dst] = valS[lp];
    }
  } else {
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const unsigned lp = (unsigned)t + (unsigned)k * BLOCK;
      if (lp < tileCount) {
        const K kk = keyS[lp];
        const unsigned d = KeyBits<K>::digit(kk, st, mask);
        const size_t dst = (size_t)(globalStart[d] + lp);
        kout[dst] = kk;
        if constexpr (PAIR) vout[dst] = valS[lp];
      }
    }
  }
}

template <class K, bool PAIR> __global__ void radix_copy_kernel(Port<const K> kin, Port<const int> vin, Port<K> kout,
                                                                Port<int> vout, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    kout[i] = kin[i];
    if constexpr (PAIR) vout[i] = vin[i];
  }
}

// ---------------------------------------------------------------------------------------- small inputs: split + finish
// Up to ~2 M 4-byte keys (1 M 8-byte keys) every tile of a pass is resident at once and a pass costs its latency chain (loads -> ranking -> look-back over
// all predecessors -> scatter) four times over, plus six launches.  The small path sorts in three launches, with no look-back and no memset:
//   1. radix_small_hist_kernel: per-tile histogram of the top 9 bits that actually differ between the keys (below), by plain stores;
//   2. radix_small_split_kernel: every tile sums the histograms of the tiles before it (independent loads, no waiting on anybody) and
//      scatters its keys into 256 buckets (stable): the 512 bins taken in pairs -- the top 8-bit digit -- or, when the bins in use span
//      fewer than 256, one bin per bucket counted from the first bin in use;
//   3. radix_small_finish_kernel: one workgroup per bucket sorts it by the remaining low bits inside LDS (LSD passes over <= 16384 keys
//      -- 8192 8-byte keys -- held in registers, 4 / 8 / 16 per lane by the bucket's size) and writes it out.
// Workgroups are 1024 threads: at these sizes a workgroup has a CU to itself, and 16 waves hide each other's latencies.
// Which bits differ: a tile ORs (key ^ keys[0]) over its own keys and over 1024 keys sampled across the whole input, and counts the window
// under the highest differing bit hb_j it sees.  The split kernel takes the maximum hb over the tiles -- exact, every key is in some tile.
// A tile that counted a lower window (the sample missed the top bit: outliers) still yields its row when all its keys share the digit of
// keys[0] in the true window (hb_j below it); otherwise the input goes the slow way.  So narrow key ranges under a wide [sbit, ebit), morton
// codes, sorted inputs and equal keys all take the three launches.
// The slow ways, all inside the finish launch:
//   * one bucket above the LDS capacity (a sentinel value, say): the other buckets are finished as usual, then the workgroups sort that one
//     by LSD passes over its own range -- per pass the tiles' histograms, then the tiles' histogram-sum splits, handed out as tickets
//     (rs_coop_lsd: no co-residency assumed, no grid barrier);
//   * several: the same LSD passes over the whole input and only the differing bits.  Slower than the ordinary passes (waiting for a phase
//     inside a launch costs more than a launch boundary), the price of not launching passes that would return at once in the common case.
// Same stable order every way.
template <class K, bool PAIR> struct RsLds {
  unsigned cnt[RSS_NW][256];  // per-wave digit counters -> offsets
  unsigned tileStart[256], globalStart[256];
  unsigned sTot9[512], sBelow9[512];  // sums of the tiles' histogram rows (all tiles / the tiles before this one), by LDS atomics
  int hbS[256];              // highest differing bit seen by tile j
  unsigned sWave[RSS_NW], sWave2[4];
  unsigned sBad, sOverCnt, sBig, sLo, sHi;
  K keyS[RS_TILE];
  int valS[PAIR ? RS_TILE : 1];
};

// ballot multisplit of one item per lane (the ranking step of radix_onesweep_kernel, see there)
__device__ __forceinline__ unsigned rs_rank_one(unsigned d, bool valid, volatile unsigned *wc, unsigned ltlo, unsigned lthi) {
  if (!valid) d = 0u;
  const unsigned long long vb = __ballot(valid);
  unsigned plo = (unsigned)vb, phi = (unsigned)(vb >> 32);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const unsigned tb = (unsigned)__builtin_amdgcn_sbfe((int)d, b, 1);
    const unsigned long long m = __ballot(tb != 0u);
    plo = __builtin_amdgcn_bitop3_b32(plo, (unsigned)m, tb, 0x90);
    phi = __builtin_amdgcn_bitop3_b32(phi, (unsigned)(m >> 32), tb, 0x90);
  }
  const unsigned below = (unsigned)__popc(plo & ltlo) + (unsigned)__popc(phi & lthi);
  const unsigned old = wc[d];
  __builtin_amdgcn_wave_barrier();
  if (valid && below == 0) wc[d] = old + (unsigned)__popc(plo) + (unsigned)__popc(phi);
  __builtin_amdgcn_wave_barrier();
  return old + below;
}

__device__ __forceinline__ int rs_top_of(int hb, int sbit) {  // start of the 9-bit window under the highest differing bit
  const int tp = hb + 1 - 9;
  return tp < sbit ? sbit : tp;
}

// histogram of the digit (st, mask) over one tile -> row[256] (plain stores); h: 256 words of LDS
template <class K>
__device__ __forceinline__ void rs_count_tile(unsigned *h, const K *keys, unsigned n, unsigned tile, int st, unsigned mask, unsigned *row) {
  const int t = threadIdx.x;
  if (t < 256) h[t] = 0u;
  __syncthreads();
  const unsigned base = tile * RS_TILE + (unsigned)wave_id() * (64 * RSS_ITEMS) + lane_id();
#pragma unroll
  for (int k = 0; k < RSS_ITEMS; ++k)
    if (base + k * 64 < n) atomicAdd(&h[KeyBits<K>::digit(keys[base + k * 64], st, mask)], 1u);
  __syncthreads();
  if (t < 256) row[t] = h[t];
}

// part[tile][2][256]: row 0 = the tile's top window (9 bits, 512 counts packed two per word), row 1 = lowest digit [sbit, sbit + 8)
// (first pass of the whole-input LSD fallback);
// meta[tile] = highest differing bit the tile saw (-1: none)
template <class K>
__global__ __launch_bounds__(RSS_BLOCK) void radix_small_hist_kernel(const K *keys, unsigned n, int sbit, int ebit, unsigned *part, int *meta,
                                                                    unsigned *ctl) {
  using U = typename KeyBits<K>::U;
  __shared__ unsigned h[512 + 256];  // [0, 512): the 9-bit window; [512, 768): the lowest 8-bit digit
  __shared__ U sOr[RSS_NW];
  const int t = threadIdx.x, lane = lane_id(), w = wave_id();
  if (t < 768) h[t] = 0u;
  const unsigned tile = blockIdx.x;
  const unsigned base = tile * RS_TILE + (unsigned)w * (64 * RSS_ITEMS) + lane;
  K key[RSS_ITEMS];
#pragma unroll
  for (int k = 0; k < RSS_ITEMS; ++k)
    if (base + k * 64 < n) key[k] = keys[base + k * 64];
  const U k0 = (U)keys[0];
  const U ks = (U)keys[(size_t)t * n / RSS_BLOCK];
  if (tile == 0 && t < 24) ctl[RS_CTL_TICKET + t] = 0u;  // ticket counter and phase-completion counters of the in-launch LSD passes
  U diff = ks ^ k0;
#pragma unroll
  for (int k = 0; k < RSS_ITEMS; ++k)
    if (base + k * 64 < n) diff |= (U)key[k] ^ k0;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    if constexpr (sizeof(U) == 8)
      diff |= (U)__shfl_xor((unsigned)diff, d, 64) | ((U)__shfl_xor((unsigned)(diff >> 32), d, 64) << 32);
    else
      diff |= (U)__shfl_xor((unsigned)diff, d, 64);
  }
  if (lane == 0) sOr[w] = diff;
  __syncthreads();
  U all = 0;
#pragma unroll
  for (int i = 0; i < RSS_NW; ++i) all |= sOr[i];
  constexpr int KB = (int)sizeof(K) * 8;
  all &= (ebit >= KB ? ~(U)0 : (U)(((U)1 << ebit) - 1u)) & (U) ~(U)(((U)1 << sbit) - 1u);
  const int hb = all ? 63 - __clzll((long long)(unsigned long long)all) : -1;
  const int top = rs_top_of(hb, sbit);
#pragma unroll
  for (int k = 0; k < RSS_ITEMS; ++k)
    if (base + k * 64 < n) {
      atomicAdd(&h[KeyBits<K>::digit(key[k], top, 0x1FFu)], 1u);
      atomicAdd(&h[512 + KeyBits<K>::digit(key[k], sbit, 0xFFu)], 1u);
    }
  __syncthreads();
  // row 0: 512 counts of at most 8192 each, two per word.  The split kernel reads a whole row with one 16-byte load per lane; lane l's four
  // words hold the bins l + 64 i (i = 0 .. 7), so that its LDS accumulation of a row is free of bank conflicts
  if (t < 256) {
    const int l = t >> 2, q = t & 3;
    part[(size_t)tile * 512 + t] = h[l + 128 * q] | (h[l + 128 * q + 64] << 16);
  }
  else if (t < 512) part[(size_t)tile * 512 + t] = h[256 + t];
  if (t == 0) meta[tile] = hb;
}

// One tile of a stable split of kin by the digit (st, mask): start of each digit's run for this tile = sum of the tiles' histogram rows
// (row of tile j at rows + j * 512), then rank, tile-local digit order in LDS, coalesced runs out.
// FIRST (the split kernel): the digit is the window under the highest differing bit of all tiles (meta), rows of tiles that counted a
// lower window are rebuilt (see the head comment); tile 0 records bucket starts, window and mode in ctl; nothing moves unless the mode
// is RS_FAST / RS_ONE_BIG / RS_COPY_SPLIT.
template <class K, bool PAIR, bool FIRST>
__device__ __forceinline__ void rs_split_tile(RsLds<K, PAIR> &S, const K *kin, const int *vin, K *kout, int *vout, unsigned n, int st,
                                              unsigned mask, const unsigned *rows, unsigned numTiles, unsigned tile, const int *meta,
                                              int sbit, unsigned *ctl) {
  constexpr int NW = RSS_NW, ITEMS = RSS_ITEMS, TILE = RS_TILE, BLOCK = RSS_BLOCK;
  const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
  const unsigned tileBase = tile * TILE;
  const unsigned tileCount = n - tileBase < (unsigned)TILE ? n - tileBase : (unsigned)TILE;
  const bool full = tileCount == (unsigned)TILE;
  for (int i = t; i < NW * 256; i += BLOCK) (&S.cnt[0][0])[i] = 0;
  if (t < 512) S.sTot9[t] = 0u, S.sBelow9[t] = 0u;
  K key[ITEMS];
  int val[ITEMS];
  unsigned rank[ITEMS];
  const unsigned base = tileBase + (unsigned)w * (64 * ITEMS) + lane;
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (full || base + k * 64 < n) {
      key[k] = kin[base + k * 64];
      if constexpr (PAIR) val[k] = vin[base + k * 64];
    }
  // histogram rows: wave w reads the rows of the tiles j = w (mod 16), a whole 1 KB row per load, all of a batch in flight at once -- one
  // memory round trip for the lot (workgroups have a CU each here: registers are free, latency is not).  Batches of NW * RB = 128 rows:
  // one batch up to 1 M keys, two above.  FIRST: 512 two-byte counts per row (lane q: bins q + 64 i); else 256 words (digits 4q .. 4q + 3).
  constexpr int RB = sizeof(K) == 4 ? 8 : 4;  // (8-byte keys: registers are short at 1024 threads, and no build of these kernels may spill --
                                              // see DESIGN, the trap under the small-input sort)
  uint4 rv[RB];
#pragma unroll
  for (int u = 0; u < RB; ++u) {
    const unsigned j = (unsigned)w + (unsigned)NW * u;
    rv[u] = j < numTiles ? *reinterpret_cast<const uint4 *>(rows + (size_t)j * 512 + 4 * lane) : uint4{0u, 0u, 0u, 0u};
  }
  int hbG = -1;
  unsigned dref = 0;
  if constexpr (FIRST) {
    if (t == 0) S.sBad = 0u, S.sOverCnt = 0u, S.sBig = 0u, S.sLo = 512u, S.sHi = 0u;
    int hj = -1;
    if (t < 256) {
      hj = t < (int)numTiles ? meta[t] : -1;
      S.hbS[t] = hj;
    }
    const K k0 = kin[0];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const int o = __shfl_xor(hj, d, 64);
      hj = o > hj ? o : hj;
    }
    if (lane == 0) S.sWave[w] = (unsigned)hj;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NW; ++i) hbG = (int)S.sWave[i] > hbG ? (int)S.sWave[i] : hbG;
    st = rs_top_of(hbG, sbit);
    mask = 0x1FFu;
    dref = KeyBits<K>::digit(k0, st, mask);
  } else {
    __syncthreads();
  }
  {
    constexpr int NV = FIRST ? 8 : 4;
    unsigned tot[NV], below[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) tot[i] = 0u, below[i] = 0u;
    bool bad = false;
    for (unsigned j0 = 0; j0 < numTiles; j0 += (unsigned)(NW * RB)) {
      if (j0 != 0u) {
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const unsigned j = j0 + (unsigned)w + (unsigned)NW * u;
          rv[u] = j < numTiles ? *reinterpret_cast<const uint4 *>(rows + (size_t)j * 512 + 4 * lane) : uint4{0u, 0u, 0u, 0u};
        }
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const unsigned j = j0 + (unsigned)w + (unsigned)NW * u;
        if (j < numTiles) {  // (wave-uniform)
          unsigned v[NV];
          if constexpr (FIRST) {
            const unsigned q[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[2 * i] = q[i] & 0xFFFFu, v[2 * i + 1] = q[i] >> 16;
            const int hj = S.hbS[j];
            if (rs_top_of(hj, sbit) != st) {  // this tile counted a lower window: all its keys sit in keys[0]'s bin, or the row is lost
              bad = bad || hj >= st;
              const unsigned cj = j == numTiles - 1 ? n - j * TILE : (unsigned)TILE;
#pragma unroll
              for (int i = 0; i < NV; ++i) v[i] = (unsigned)(lane + 64 * i) == dref ? cj : 0u;
            }
          } else {
            v[0] = rv[u].x, v[1] = rv[u].y, v[2] = rv[u].z, v[3] = rv[u].w;
          }
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            tot[i] += v[i];
            below[i] += j < tile ? v[i] : 0u;
          }
        }
      }
    }
    // (FIRST: this lane's bins are lane + 64 i, see radix_small_hist_kernel; else the digits 4 lane + i)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int bin = FIRST ? lane + 64 * i : NV * lane + i;
      if (tot[i]) atomicAdd(&S.sTot9[bin], tot[i]);
      if (below[i]) atomicAdd(&S.sBelow9[bin], below[i]);
    }
    if (FIRST && bad) S.sBad = 1u;
    if constexpr (FIRST) {
      // first / last bin in use: this lane's bins of this wave's rows, folded over the wave, one LDS atomic pair per wave
      unsigned lo = 512u, hi = 0u;  // hi = last bin + 1
#pragma unroll
      for (int i = NV - 1; i >= 0; --i)
        if (tot[i]) lo = (unsigned)(lane + 64 * i);
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (tot[i]) hi = (unsigned)(lane + 64 * i) + 1u;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const unsigned a = __shfl_xor(lo, d, 64), b = __shfl_xor(hi, d, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
      }
      if (lane == 0 && hi != 0u) atomicMin(&S.sLo, lo), atomicMax(&S.sHi, hi);
    }
  }
  __syncthreads();
  // FIRST: 512 fine bins -> 256 buckets.  Bins in use spanning fewer than 256: bucket = bin - first bin in use (keys that straddle a
  // power of two, e.g. ints in [-2^30, 2^30), get 256 buckets instead of 128); otherwise bucket = bin / 2, the 8-bit digit.
  bool fine = false;
  unsigned off = 0;
  if constexpr (FIRST) {
    const int lo = (int)S.sLo, hi = (int)S.sHi - 1;
    fine = hi - lo < 256;
    off = fine ? (unsigned)lo : 0u;
  }
  auto bucket_of = [&](K k) -> unsigned {
    const unsigned d = KeyBits<K>::digit(k, st, mask);
    if constexpr (FIRST) return fine ? d - off : d >> 1;
    else return d;
  };
  unsigned bstart = 0, below = 0;
  if (t < 256) {
    unsigned tot = 0;
    if constexpr (FIRST) {
      if (fine) {
        if ((unsigned)t + off < 512u) tot = S.sTot9[t + off], below = S.sBelow9[t + off];
      } else {
        tot = S.sTot9[2 * t] + S.sTot9[2 * t + 1], below = S.sBelow9[2 * t] + S.sBelow9[2 * t + 1];
      }
    } else {
      tot = S.sTot9[t], below = S.sBelow9[t];
    }
    if (FIRST && tot > rs_small_cap<K>()) {
      atomicAdd(&S.sOverCnt, 1u);
      S.sBig = (unsigned)t;
    }
    unsigned sc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned o = shfl_up(sc, d);
      if (lane >= d) sc += o;
    }
    if (lane == 63) S.sWave2[w] = sc;
    bstart = sc - tot;
  }
  __syncthreads();
  if (t < 256)
    for (int i = 0; i < w; ++i) bstart += S.sWave2[i];
  if constexpr (FIRST) {
    const int below_top = fine ? st : st + 1;  // the finish kernel sorts the bits [sbit, below_top)
    unsigned mode = RS_FAST;
    if (hbG < 0) mode = RS_COPY_IN;                   // no key differs from keys[0] inside the window: the input is its own sorted order
    else if (S.sBad) mode = RS_LSD;
    else if (below_top == sbit) mode = RS_COPY_SPLIT;  // nothing left under the buckets: the split is the sort
    else if (S.sOverCnt == 1u) mode = RS_ONE_BIG;
    else if (S.sOverCnt > 1u) mode = RS_LSD;
    if (tile == 0 && t < 256) {
      ctl[t] = bstart;
      if (t == 255) ctl[256] = n;
      if (t == 0) {
        ctl[RS_CTL_MODE] = mode;
        ctl[RS_CTL_TOP] = (unsigned)below_top;
        ctl[RS_CTL_EBIT] = (unsigned)(hbG + 1);
        ctl[RS_CTL_BIG] = S.sBig;
      }
    }
    if (mode == RS_COPY_IN || mode == RS_LSD) return;  // (uniform)
  }
  __syncthreads();  // (sWave2 is reused below)
  volatile unsigned *wc = S.cnt[w];
  const unsigned long long lt = lanemask_lt();
  const unsigned ltlo = (unsigned)lt, lthi = (unsigned)(lt >> 32);
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const bool valid = full || base + k * 64 < n;
    rank[k] = rs_rank_one(valid ? bucket_of(key[k]) : 0u, valid, wc, ltlo, lthi);
  }
  __syncthreads();
  if (t < 256) {
    unsigned run = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const unsigned c = S.cnt[i][t];
      S.cnt[i][t] = run;
      run += c;
    }
    unsigned s = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      unsigned o = shfl_up(s, d);
      if (lane >= d) s += o;
    }
    if (lane == 63) S.sWave2[w] = s;
    S.tileStart[t] = s - run;
  }
  __syncthreads();
  if (t < 256) {
    unsigned b = 0;
    for (int i = 0; i < w; ++i) b += S.sWave2[i];
    const unsigned ts = S.tileStart[t] + b;
    S.globalStart[t] = bstart + below - ts;  // (wrapping) dst = globalStart[d] + position in the tile-sorted order
#pragma unroll
    for (int i = 0; i < NW; ++i) S.cnt[i][t] += ts;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)
    if (full || base + k * 64 < n) {
      const unsigned lp = S.cnt[w][bucket_of(key[k])] + rank[k];
      S.keyS[lp] = key[k];
      if constexpr (PAIR) S.valS[lp] = val[k];
    }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const unsigned lp = (unsigned)t + (unsigned)k * BLOCK;
    if (full || lp < tileCount) {
      const K kk = S.keyS[lp];
      const unsigned dst = S.globalStart[bucket_of(kk)] + lp;
      kout[dst] = kk;
      if constexpr (PAIR) vout[dst] = S.valS[lp];
    }
  }
}

template <class K, bool PAIR>
__global__ __launch_bounds__(RSS_BLOCK) void radix_small_split_kernel(const K *kin, const int *vin, K *kout, int *vout, unsigned n, int sbit,
                                                                     const unsigned *part, const int *meta, unsigned numTiles,
                                                                     unsigned *ctl) {
  __shared__ RsLds<K, PAIR> S;
  rs_split_tile<K, PAIR, true>(S, kin, vin, kout, vout, n, 0, 0u, part, numTiles, blockIdx.x, meta, sbit, ctl);
}

// LSD passes over src[0, n) by the bits [sbit, ebit), all inside the calling launch, by whichever of its workgroups are running: hop p
// writes `out` on the last pass, else bufA / bufB alternately (B first).  Row 1 of `part` carries the tile histograms; rowValid: it already
// holds the first pass's (radix_small_hist_kernel counted it over the input).
// The work is a list of phases -- per pass: the tiles' histograms of the digit (unless rowValid covers it), then the tiles' splits -- and a
// workgroup draws (phase, tile) tickets from one counter until the list is exhausted.  Before a phase's tile it waits until every tile of
// the previous phase is done (a completion counter per phase).  No co-residency is assumed: a ticket is only ever held by a running
// workgroup, and it waits only for tickets drawn before its own, so any number of resident workgroups >= 1 makes progress -- two such sorts
// on two streams, or a busy device, cannot deadlock it the way a grid barrier over a fixed set of workgroups could.
template <class K, bool PAIR>
__device__ __forceinline__ void rs_coop_lsd(RsLds<K, PAIR> &S, const K *src, const int *srcV, K *bufA, int *bufAV, K *bufB, int *bufBV, K *out,
                                            int *outV, unsigned n, int sbit, int ebit, unsigned *part, unsigned *ctl, bool rowValid) {
  const int passes = (ebit - sbit + 7) / 8;
  const unsigned numTiles = (n + RS_TILE - 1) / RS_TILE;
  const unsigned first = rowValid ? 1u : 0u;          // phases q = first .. 2 passes - 1: q even = histograms of pass q / 2, q odd = its splits
  const unsigned numPhases = 2u * (unsigned)passes - first;
  const int t = threadIdx.x;
  if (blockIdx.x >= numTiles) return;  // a phase has numTiles tickets: more workgroups than that would only poll
  // (drawing the next ticket while the current one is worked on was measured: slower -- 102 -> 108 us on the outlier case at 1 M keys)
  for (;;) {
    __syncthreads();  // (the previous ticket's work is finished with S)
    if (t == 0) S.sBig = __hip_atomic_fetch_add(ctl + RS_CTL_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned ticket = S.sBig;
    const unsigned ph = ticket / numTiles, tile = ticket % numTiles;
    if (ph >= numPhases) break;
    if (ph > 0u) {
      if (t == 0) {
        while (__hip_atomic_load(ctl + RS_CTL_DONE + (ph - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < numTiles) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    const unsigned q = ph + first;
    const int p = (int)(q >> 1);
    const int st = sbit + 8 * p;
    const int bits = ebit - st < 8 ? ebit - st : 8;
    const unsigned mask = (1u << bits) - 1u;
    const bool last = p == passes - 1;
    const K *sK = p == 0 ? src : (((p - 1) & 1) ? bufA : bufB);
    const int *sV = p == 0 ? srcV : (((p - 1) & 1) ? bufAV : bufBV);
    if ((q & 1u) == 0u) {
      rs_count_tile<K>(S.tileStart, sK, n, tile, st, mask, part + (size_t)tile * 512 + 256);
    } else {
      K *dK = last ? out : ((p & 1) ? bufA : bufB);
      int *dV = last ? outV : ((p & 1) ? bufAV : bufBV);
      rs_split_tile<K, PAIR, false>(S, sK, sV, dK, dV, n, st, mask, part + 256, numTiles, tile, nullptr, sbit, nullptr);
    }
    // this tile of the phase is done: stores drained and written back, then counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(ctl + RS_CTL_DONE + ph, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One bucket of c <= MAXI * 1024 keys: into registers (positions stay (wave, item, lane)-ordered), LSD passes over the bits [sbit, top)
// through LDS, out.
#ifdef ZS_RS_NOINLINE  // repro builds only (tools/repro/): a real call with a stack in the finish kernel, one of the two r03 builds behind the "scratch trap"
#define ZS_RS_BUCKET_INLINE __noinline__
#else
#define ZS_RS_BUCKET_INLINE __forceinline__
#endif
template <class K, bool PAIR, int MAXI>
__device__ ZS_RS_BUCKET_INLINE void rs_finish_bucket(unsigned (*cnt)[256], unsigned *sWave2, K *keyS, int *valS, const K *bk, const int *bv, K *ok,
                                                 int *ov, unsigned c, int sbit, int top) {
  constexpr int NW = RSS_NW, BLOCK = RSS_BLOCK;
  const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
  const int KI = (int)((c + BLOCK - 1) / BLOCK);
  const unsigned wpos = (unsigned)w * 64u * (unsigned)KI + (unsigned)lane;
  volatile unsigned *wc = cnt[w];
  const unsigned long long lt = lanemask_lt();
  const unsigned ltlo = (unsigned)lt, lthi = (unsigned)(lt >> 32);
  K key[MAXI];
  int val[MAXI];
  unsigned rank[MAXI];
#pragma unroll
  for (int k = 0; k < MAXI; ++k)
    if (k < KI && wpos + k * 64 < c) {
      key[k] = bk[wpos + k * 64];
      if constexpr (PAIR) val[k] = bv[wpos + k * 64];
    }
  for (int st = sbit; st < top; st += 8) {
    const int bits = top - st < 8 ? top - st : 8;
    const unsigned mask = (1u << bits) - 1u;
    for (int i = t; i < NW * 256; i += BLOCK) (&cnt[0][0])[i] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXI; ++k)
      if (k < KI) {
        const bool valid = wpos + k * 64 < c;
        rank[k] = rs_rank_one(valid ? KeyBits<K>::digit(key[k], st, mask) : 0u, valid, wc, ltlo, lthi);
      }
    __syncthreads();
    unsigned excl = 0;
    if (t < 256) {
      unsigned run = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        const unsigned cc = cnt[i][t];
        cnt[i][t] = run;
        run += cc;
      }
      unsigned s = run;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        unsigned o = shfl_up(s, d);
        if (lane >= d) s += o;
      }
      if (lane == 63) sWave2[w] = s;
      excl = s - run;
    }
    __syncthreads();
    if (t < 256) {
      for (int i = 0; i < w; ++i) excl += sWave2[i];
#pragma unroll
      for (int i = 0; i < NW; ++i) cnt[i][t] += excl;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXI; ++k)
      if (k < KI && wpos + k * 64 < c) {
        const unsigned lp = cnt[w][KeyBits<K>::digit(key[k], st, mask)] + rank[k];
        keyS[lp] = key[k];
        if constexpr (PAIR) valS[lp] = val[k];
      }
    __syncthreads();
    if (st + 8 < top) {
#pragma unroll
      for (int k = 0; k < MAXI; ++k)
        if (k < KI && wpos + k * 64 < c) {
          key[k] = keyS[wpos + k * 64];
          if constexpr (PAIR) val[k] = valS[wpos + k * 64];
        }
    } else {
      for (unsigned i = t; i < c; i += BLOCK) {
        ok[i] = keyS[i];
        if constexpr (PAIR) ov[i] = valS[i];
      }
    }
  }
}

// buckets: tk0 (written by the split kernel) -> kout; the other modes as the head comment says
template <class K, bool PAIR>
__global__ __launch_bounds__(RSS_BLOCK) void radix_small_finish_kernel(const K *kin, const int *vin, K *tk0, int *tv0, K *tk1, int *tv1, K *kout,
                                                                      int *vout, unsigned n, int sbit, unsigned *part, unsigned *ctl) {
  constexpr int NW = RSS_NW, BLOCK = RSS_BLOCK;
  constexpr unsigned CAP = rs_small_cap<K>();
  // LDS: the bucket layout (counters + up to 16384 keys and values) and the tile layout of the in-launch LSD passes share the bytes
  struct FinLds {
    unsigned cnt[NW][256];
    unsigned sWave2[4];
    K keyS[CAP];
    int valS[PAIR ? CAP : 1];
  };
  constexpr size_t ldsBytes = sizeof(FinLds) > sizeof(RsLds<K, PAIR>) ? sizeof(FinLds) : sizeof(RsLds<K, PAIR>);
  __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[ldsBytes];
  RsLds<K, PAIR> &S = *reinterpret_cast<RsLds<K, PAIR> *>(ldsRaw);
  FinLds &F = *reinterpret_cast<FinLds *>(ldsRaw);
  const int lane = lane_id(), w = wave_id(), t = threadIdx.x;
#ifdef ZS_RS_FORCE_SCRATCH  // measurement builds only (tools/repro/): makes this kernel use private memory, see DESIGN "a trap met on the way"
  {
    volatile unsigned junk[ZS_RS_FORCE_SCRATCH];
    for (int i = 0; i < ZS_RS_FORCE_SCRATCH; ++i) junk[(i + threadIdx.x) % ZS_RS_FORCE_SCRATCH] = (unsigned)i;
    if (junk[threadIdx.x % ZS_RS_FORCE_SCRATCH] == 0xFFFFFFFFu) ctl[RS_CTL_WORDS - 1] = 1u;
  }
#endif
  const unsigned mode = ctl[RS_CTL_MODE];
  const int top = (int)ctl[RS_CTL_TOP];
  const unsigned start0 = ctl[blockIdx.x], end0 = ctl[blockIdx.x + 1];  // (gridDim.x <= 256) this workgroup's first bucket, fetched with the mode
  if (mode == RS_COPY_IN || mode == RS_COPY_SPLIT) {
    const K *sk = mode == RS_COPY_IN ? kin : tk0;
    const int *sv = mode == RS_COPY_IN ? vin : tv0;
    // keys and values are copied independently: a pair sort with the keys in place but separate value arrays
    // (radix_sort_pair(keys, iota, keys, perm) on constant keys, or a bit window in which all keys agree) must still write vout
    const bool copyK = sk != kout;
    bool copyV = false;
    if constexpr (PAIR) copyV = sv != vout;
    if (copyK || copyV)
      for (unsigned i = blockIdx.x * BLOCK + t; i < n; i += gridDim.x * BLOCK) {
        if (copyK) kout[i] = sk[i];
        if constexpr (PAIR) {
          if (copyV) vout[i] = sv[i];
        }
      }
    return;
  }
  const unsigned big = mode == RS_ONE_BIG ? ctl[RS_CTL_BIG] : 256u;
  for (unsigned b = blockIdx.x; b < (mode == RS_LSD ? 0u : 256u); b += gridDim.x) {
    const unsigned start = b == blockIdx.x ? start0 : ctl[b], c = (b == blockIdx.x ? end0 : ctl[b + 1]) - start;
    if (c == 0u || b == big) continue;
    __syncthreads();  // (a second bucket of this workgroup: the first one's copy out of keyS is finished)
    // items per lane by the bucket's size (each instantiation unrolled for exactly its count)
    if (c <= 4u * BLOCK) rs_finish_bucket<K, PAIR, 4>(F.cnt, F.sWave2, F.keyS, F.valS, tk0 + start, PAIR ? tv0 + start : nullptr, kout + start,
                                                       PAIR ? vout + start : nullptr, c, sbit, top);
    else if (c <= 8u * BLOCK) rs_finish_bucket<K, PAIR, 8>(F.cnt, F.sWave2, F.keyS, F.valS, tk0 + start, PAIR ? tv0 + start : nullptr, kout + start,
                                                            PAIR ? vout + start : nullptr, c, sbit, top);
    else if constexpr (CAP > 8u * BLOCK)
      rs_finish_bucket<K, PAIR, 16>(F.cnt, F.sWave2, F.keyS, F.valS, tk0 + start, PAIR ? tv0 + start : nullptr, kout + start,
                                    PAIR ? vout + start : nullptr, c, sbit, top);
  }
  if (mode == RS_ONE_BIG || mode == RS_LSD) {
    // RS_ONE_BIG: the one bucket too large for LDS -- its range of tk0 -> ... -> the same range of kout, by the bits below the top window.
    // RS_LSD: the whole input -> ... -> kout by all differing bits.  (One call site: the body is inlined once.)
    const bool whole = mode == RS_LSD;
    const unsigned start = whole ? 0u : ctl[big], c = whole ? n : ctl[big + 1] - start;
    __syncthreads();
    rs_coop_lsd<K, PAIR>(S, whole ? kin : tk0 + start, !PAIR ? nullptr : (whole ? vin : tv0 + start), tk0 + start,
                         PAIR ? tv0 + start : nullptr, tk1 + start, PAIR ? tv1 + start : nullptr, kout + start,
                         PAIR ? vout + start : nullptr, c, sbit, whole ? (int)ctl[RS_CTL_EBIT] : top, part, ctl, whole);
  }
}

// host side of the small path (see above).  grid of the finish launch: one workgroup per bucket, at most one per CU
template <class K, bool PAIR>
static void radix_sort_small(Launch &L, const K *kin, const int *vin, K *kout, int *vout, unsigned n, int sbit, int ebit, unsigned cus) {
  const unsigned numTiles = ceil_div(n, RS_TILE);
  const size_t partWords = (size_t)numTiles * 512;
  unsigned *mem = (unsigned *)L.temp(sizeof(unsigned) * (RS_CTL_WORDS + 256 + partWords));
  unsigned *ctl = mem, *part = mem + RS_CTL_WORDS + 256;
  int *meta = (int *)(mem + RS_CTL_WORDS);
  K *tk[2] = {(K *)L.temp(sizeof(K) * (size_t)n), (K *)L.temp(sizeof(K) * (size_t)n)};
  int *tv[2] = {nullptr, nullptr};
  if (PAIR) tv[0] = (int *)L.temp(sizeof(int) * (size_t)n), tv[1] = (int *)L.temp(sizeof(int) * (size_t)n);
  hipLaunchKernelGGL((radix_small_hist_kernel<K>), dim3(numTiles), dim3(RSS_BLOCK), 0, L.stream, kin, n, sbit, ebit, part, meta, ctl);
  hipLaunchKernelGGL((radix_small_split_kernel<K, PAIR>), dim3(numTiles), dim3(RSS_BLOCK), 0, L.stream, kin, vin, tk[0], tv[0], n, sbit,
                     (const unsigned *)part, (const int *)meta, numTiles, ctl);
  hipLaunchKernelGGL((radix_small_finish_kernel<K, PAIR>), dim3(std::min(256u, cus)), dim3(RSS_BLOCK), 0, L.stream, kin, vin, tk[0], tv[0],
                     tk[1], tv[1], kout, vout, n, sbit, part, ctl);
}

template <class K, bool PAIR>
static void radix_sort_impl(Launch &L, Port<const K> kin, Port<const int> vin, Port<K> kout, Port<int> vout, size_t n,
                            int sbit, int ebit) {
  if (n == 0) return;
  if (n > (size_t)OS_VAL_MASK) {
    // tile descriptors carry a 30-bit running count next to their 2 flag bits, scatter positions are 32-bit: a larger
    // input would be corrupted silently, so it is refused (latched error, like any failed launch)
    report_error(hipErrorInvalidValue, "radix_sort: more than 2^30 - 1 keys per call", __FILE__, __LINE__);
    return;
  }
  if (sbit < 0) sbit = 0;
  if (ebit > (int)sizeof(K) * 8) ebit = (int)sizeof(K) * 8;
  const int passes = ebit > sbit ? (ebit - sbit + 7) / 8 : 0;
  if (passes == 0) {
    hipLaunchKernelGGL((radix_copy_kernel<K, PAIR>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, kin, vin, kout, vout, n);
    return;
  }
  if (passes == 1 && (const void *)kin.base == (const void *)kout.base) {
    // single pass with output aliasing the input: sort into a temporary, then copy (the reference always
    // stages through temporaries, so in-place calls are legal there)
    K *tmpK = (K *)L.temp(sizeof(K) * n);
    int *tmpV = PAIR ? (int *)L.temp(sizeof(int) * n) : nullptr;
    radix_sort_impl<K, PAIR>(L, kin, vin, contiguous_port<K>(tmpK), contiguous_port<int>(tmpV), n, sbit, ebit);
    hipLaunchKernelGGL((radix_copy_kernel<K, PAIR>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream,
                       contiguous_port<const K>(tmpK), contiguous_port<const int>(tmpV), kout, vout, n);
    return;
  }
  {
    static const int smallOff = [] { const char *e = getenv("ZS_ROCM_SORT_SMALL"); return e && atoi(e) == 0 ? 1 : 0; }();  // measurement only
    if (!smallOff && passes >= 2 && n <= rs_small_max_n<K>() && kin.contiguous() && kout.contiguous() &&
        (!PAIR || (vin.contiguous() && vout.contiguous()))) {
      radix_sort_small<K, PAIR>(L, kin.base + kin.idx, PAIR ? vin.base + vin.idx : nullptr, kout.base + kout.idx,
                                PAIR ? vout.base + vout.idx : nullptr, (unsigned)n, sbit, ebit, L.cu_count());
      return;
    }
  }
  // 8192-key tiles at every size.  Measured at 1 M keys (122 tiles, all resident at once, so the look-back walks aggregates): smaller
  // tiles (2048 / 4096 keys) or wider look-back batches (16 / 32 / 64 descriptors per round trip) are all slower -- a pass costs
  // ~10 us + ~45 ns per tile of chain (r02 measurements).
  const unsigned numTiles = ceil_div(n, RS_TILE);
  constexpr int MAXPASS = (int)sizeof(K);
  // [MAXPASS][256] global digit starts + per pass: ticket + [numTiles][256] descriptors (re-initialised every call)
  const size_t descBytes = sizeof(unsigned) * (256 * (size_t)numTiles + 64);
  const size_t ghistBytes = sizeof(unsigned) * 256 * MAXPASS;
  char *ctl = (char *)L.temp(ghistBytes + descBytes * (size_t)passes);  // one block, one memset
  unsigned *ghist = (unsigned *)ctl;
  char *descMem = ctl + ghistBytes;
  ZSR_CHECK(hipMemsetAsync(ctl, 0, ghistBytes + descBytes * (size_t)passes, L.stream));
  K *tk[2] = {nullptr, nullptr};
  int *tv[2] = {nullptr, nullptr};
  const int ntemp = passes >= 3 ? 2 : passes - 1;
  for (int i = 0; i < ntemp; ++i) {
    tk[i] = (K *)L.temp(sizeof(K) * n);
    if (PAIR) tv[i] = (int *)L.temp(sizeof(int) * n);
  }
  {
    // every block ends with 256 x passes global atomics on the same addresses: few, fat blocks for small inputs
    const unsigned hb = (unsigned)std::min<size_t>(ceil_div(n, 256 * 16), 2048);
    hipLaunchKernelGGL((radix_global_hist_kernel<K, MAXPASS>), dim3(hb), dim3(256), 0, L.stream, kin, n, sbit, ebit, ghist);
  }
  Port<const K> srcK = kin;
  Port<const int> srcV = vin;
  for (int p = 0; p < passes; ++p) {
    const int st = sbit + 8 * p;
    const int bits = std::min(8, ebit - st);  // narrowed last pass (execution/ExecutionPolicy.hpp:493-496)
    const unsigned mask = (1u << bits) - 1u;
    Port<K> dstK = kout;
    Port<int> dstV = vout;
    if (p != passes - 1) {
      // temporaries alternate so that a pass never writes the buffer it reads
      int which = p & 1;
      dstK = contiguous_port<K>(tk[which]);
      if (PAIR) dstV = contiguous_port<int>(tv[which]);
    }
    unsigned *desc = (unsigned *)(descMem + descBytes * (size_t)p);
    unsigned *ticket = desc + 256 * (size_t)numTiles;
    hipLaunchKernelGGL((radix_onesweep_kernel<K, PAIR, RS_BLOCK, RS_ITEMS, 8>), dim3(numTiles), dim3(RS_BLOCK), 0, L.stream, srcK, srcV, dstK, dstV,
                         n, st, mask, (const unsigned *)(ghist + 256 * p), desc, ticket);
    srcK = Port<const K>{dstK.base, dstK.idx, dstK.bits, dstK.mask, dstK.chns};
    if (PAIR) srcV = Port<const int>{dstV.base, dstV.idx, dstV.bits, dstV.mask, dstV.chns};
  }
}

// internal entry for other translation units
void radix_sort_pair_u32(Launch &L, const unsigned *kin, const int *vin, unsigned *kout, int *vout, size_t n, int sbit,
                         int ebit) {
  radix_sort_impl<unsigned, true>(L, contiguous_port<const unsigned>(kin), contiguous_port<const int>(vin),
                                  contiguous_port<unsigned>(kout), contiguous_port<int>(vout), n, sbit, ebit);
}
void radix_sort_pair_u64(Launch &L, const unsigned long long *kin, const int *vin, unsigned long long *kout, int *vout,
                         size_t n, int sbit, int ebit) {
  radix_sort_impl<unsigned long long, true>(L, contiguous_port<const unsigned long long>(kin),
                                            contiguous_port<const int>(vin), contiguous_port<unsigned long long>(kout),
                                            contiguous_port<int>(vout), n, sbit, ebit);
}

template <class K>
static void radix_sort_api(zs_rocm_policy *pol, const K *kin, const int *vin, K *kout, int *vout, size_t n, int sbit, int ebit) {
  Launch L(pol, "radix_sort");
  if (vin && vout)
    radix_sort_impl<K, true>(L, contiguous_port<const K>(kin), contiguous_port<const int>(vin), contiguous_port<K>(kout),
                             contiguous_port<int>(vout), n, sbit, ebit);
  else
    radix_sort_impl<K, false>(L, contiguous_port<const K>(kin), Port<const int>{}, contiguous_port<K>(kout), Port<int>{}, n,
                              sbit, ebit);
}

// ======================================================================================= merge sort
template <class T> struct LessOp {
  __device__ __forceinline__ bool operator()(const T &a, const T &b) const { return a < b; }
};
template <class T> struct GreaterOp {
  __device__ __forceinline__ bool operator()(const T &a, const T &b) const { return a > b; }
};
template <class T, class Comp, class KIt, class VIt>
static void merge_sort_impl(Launch &L, KIt keys, VIt vals, bool pair, size_t n, Comp comp) {
  using namespace zs_rocm_ms;
  if (pair) {
    const size_t b = scratch_bytes<T, int, true>(n);
    merge_sort_run<T, int, true>(L.stream, keys, vals, n, comp, b ? L.temp(b) : nullptr);
  } else {
    const size_t b = scratch_bytes<T, NoVal, false>(n);
    merge_sort_run<T, NoVal, false>(L.stream, keys, (NoVal *)nullptr, n, comp, b ? L.temp(b) : nullptr);
  }
}
template <class T> static void merge_sort_api(zs_rocm_policy *pol, T *keys, int *vals, size_t n, int descending) {
  Launch L(pol, "merge_sort");
  if (descending) merge_sort_impl<T>(L, keys, vals, vals != nullptr, n, GreaterOp<T>{});
  else merge_sort_impl<T>(L, keys, vals, vals != nullptr, n, LessOp<T>{});
}

}  // namespace zsr

using namespace zsr;

// ======================================================================================= C ABI
extern "C" {

// ---- (A) py_interop/cuda/ExecutionPolicy.cpp:41-131
#define ZSR_DEFINE_PRIMITIVES(T)                                                                                   \
  void reduce_sum__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                           \
                                aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {                   \
    Launch L(pol, "reduce_sum");                                                                                   \
    reduce_impl<OP_PLUS, T>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)0);  \
  }                                                                                                                \
  void reduce_prod__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                          \
                                 aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {                  \
    Launch L(pol, "reduce_prod");                                                                                  \
    reduce_impl<OP_MUL, T>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)1); \
  }                                                                                                                \
  void reduce_min__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                           \
                                aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {                   \
    Launch L(pol, "reduce_min");                                                                                   \
    reduce_impl<OP_MIN, T>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out),        \
                           std::numeric_limits<T>::max());                                                         \
  }                                                                                                                \
  void reduce_max__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                           \
                                aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {                   \
    Launch L(pol, "reduce_max");                                                                                   \
    reduce_impl<OP_MAX, T>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out),        \
                           std::numeric_limits<T>::lowest());                                                      \
  }                                                                                                                \
  void exclusive_scan_sum__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                   \
                                        aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {           \
    Launch L(pol, "exclusive_scan_sum");                                                                           \
    scan_impl<OP_PLUS, T, true>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)0); \
  }                                                                                                                \
  void exclusive_scan_prod__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                  \
                                         aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {          \
    Launch L(pol, "exclusive_scan_prod");                                                                          \
    scan_impl<OP_MUL, T, true>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)1); \
  }                                                                                                                \
  void inclusive_scan_sum__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                   \
                                        aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {           \
    Launch L(pol, "inclusive_scan_sum");                                                                           \
    scan_impl<OP_PLUS, T, false>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)0); \
  }                                                                                                                \
  void inclusive_scan_prod__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_const_##T##_1 first,                  \
                                         aosoa_iterator_const_##T##_1 last, aosoa_iterator_##T##_1 out) {          \
    Launch L(pol, "inclusive_scan_prod");                                                                          \
    scan_impl<OP_MUL, T, false>(L, make_port<const T>(first), (size_t)(last.idx - first.idx), make_port<T>(out), (T)1); \
  }                                                                                                                \
  /* merge sort: py_interop/cuda/ExecutionPolicy.cpp:99-111 */                                                     \
  void merge_sort__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_##T##_1 first, aosoa_iterator_##T##_1 last) {  \
    Launch L(pol, "merge_sort");                                                                                   \
    merge_sort_impl<T>(L, make_port<T>(first), Port<int>{}, false, (size_t)(last.idx - first.idx), LessOp<T>{});   \
  }                                                                                                                \
  void merge_sort_pair__rocm_##T##_1(zs_rocm_policy *pol, aosoa_iterator_##T##_1 keys, aosoa_iterator_int_1 vals,  \
                                     size_t count) {                                                               \
    Launch L(pol, "merge_sort_pair");                                                                              \
    merge_sort_impl<T>(L, make_port<T>(keys), make_port<int>(vals), true, count, LessOp<T>{});                     \
  }

ZSR_DEFINE_PRIMITIVES(int)
ZSR_DEFINE_PRIMITIVES(float)
ZSR_DEFINE_PRIMITIVES(double)

// radix sort: integral T only; for float/double the reference instantiates an empty body
// (py_interop/cuda/ExecutionPolicy.cpp:112-126, `if constexpr (is_integral_v<T>)`)
void radix_sort__rocm_int_1(zs_rocm_policy *pol, aosoa_iterator_int_1 first, aosoa_iterator_int_1 last,
                            aosoa_iterator_int_1 out) {
  Launch L(pol, "radix_sort");
  radix_sort_impl<int, false>(L, make_port<const int>(first), Port<const int>{}, make_port<int>(out), Port<int>{},
                              (size_t)(last.idx - first.idx), 0, 32);
}
void radix_sort_pair__rocm_int_1(zs_rocm_policy *pol, aosoa_iterator_int_1 keysIn, aosoa_iterator_int_1 valsIn,
                                 aosoa_iterator_int_1 keysOut, aosoa_iterator_int_1 valsOut, size_t count) {
  Launch L(pol, "radix_sort_pair");
  radix_sort_impl<int, true>(L, make_port<const int>(keysIn), make_port<const int>(valsIn), make_port<int>(keysOut),
                             make_port<int>(valsOut), count, 0, 32);
}
void radix_sort__rocm_float_1(zs_rocm_policy *, aosoa_iterator_float_1, aosoa_iterator_float_1, aosoa_iterator_float_1) {}
void radix_sort_pair__rocm_float_1(zs_rocm_policy *, aosoa_iterator_float_1, aosoa_iterator_int_1, aosoa_iterator_float_1,
                                   aosoa_iterator_int_1, size_t) {}
void radix_sort__rocm_double_1(zs_rocm_policy *, aosoa_iterator_double_1, aosoa_iterator_double_1, aosoa_iterator_double_1) {}
void radix_sort_pair__rocm_double_1(zs_rocm_policy *, aosoa_iterator_double_1, aosoa_iterator_int_1,
                                    aosoa_iterator_double_1, aosoa_iterator_int_1, size_t) {}

// ---- (B)
#define ZSR_DEFINE_RAW(T, S)                                                                                      \
  void zs_rocm_reduce_##S(zs_rocm_policy *pol, const T *in, size_t n, T *out, T init, int op) {                   \
    Launch L(pol, "reduce");                                                                                      \
    reduce_dispatch<T>(L, contiguous_port<const T>(in), n, contiguous_port<T>(out), init, op);                    \
  }                                                                                                               \
  void zs_rocm_scan_##S(zs_rocm_policy *pol, const T *in, size_t n, T *out, T init, int op, int exclusive) {      \
    Launch L(pol, exclusive ? "exclusive_scan" : "inclusive_scan");                                               \
    scan_dispatch<T>(L, contiguous_port<const T>(in), n, contiguous_port<T>(out), init, op, exclusive != 0);      \
  }
ZSR_DEFINE_RAW(int32_t, i32)
ZSR_DEFINE_RAW(int64_t, i64)
ZSR_DEFINE_RAW(float, f32)
ZSR_DEFINE_RAW(double, f64)

void zs_rocm_radix_sort_i32(zs_rocm_policy *p, const int32_t *kin, const int32_t *vin, int32_t *kout, int32_t *vout,
                            size_t n, int sbit, int ebit) {
  radix_sort_api<int32_t>(p, kin, vin, kout, vout, n, sbit, ebit);
}
void zs_rocm_radix_sort_u32(zs_rocm_policy *p, const uint32_t *kin, const int32_t *vin, uint32_t *kout, int32_t *vout,
                            size_t n, int sbit, int ebit) {
  radix_sort_api<uint32_t>(p, kin, vin, kout, vout, n, sbit, ebit);
}
void zs_rocm_radix_sort_i64(zs_rocm_policy *p, const int64_t *kin, const int32_t *vin, int64_t *kout, int32_t *vout,
                            size_t n, int sbit, int ebit) {
  radix_sort_api<int64_t>(p, kin, vin, kout, vout, n, sbit, ebit);
}
void zs_rocm_radix_sort_u64(zs_rocm_policy *p, const uint64_t *kin, const int32_t *vin, uint64_t *kout, int32_t *vout,
                            size_t n, int sbit, int ebit) {
  radix_sort_api<uint64_t>(p, kin, vin, kout, vout, n, sbit, ebit);
}

void zs_rocm_merge_sort_i32(zs_rocm_policy *p, int32_t *k, int32_t *v, size_t n, int d) { merge_sort_api<int32_t>(p, k, v, n, d); }
void zs_rocm_merge_sort_u32(zs_rocm_policy *p, uint32_t *k, int32_t *v, size_t n, int d) { merge_sort_api<uint32_t>(p, k, v, n, d); }
void zs_rocm_merge_sort_i64(zs_rocm_policy *p, int64_t *k, int32_t *v, size_t n, int d) { merge_sort_api<int64_t>(p, k, v, n, d); }
void zs_rocm_merge_sort_u64(zs_rocm_policy *p, uint64_t *k, int32_t *v, size_t n, int d) { merge_sort_api<uint64_t>(p, k, v, n, d); }
void zs_rocm_merge_sort_f32(zs_rocm_policy *p, float *k, int32_t *v, size_t n, int d) { merge_sort_api<float>(p, k, v, n, d); }
void zs_rocm_merge_sort_f64(zs_rocm_policy *p, double *k, int32_t *v, size_t n, int d) { merge_sort_api<double>(p, k, v, n, d); }

}  // extern "C"
