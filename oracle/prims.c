/*
 * prims.c -- CPU restatement of zs::reduce / exclusive_scan / inclusive_scan / radix_sort(_pair).
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).  Two flavours per primitive:
 *   orc_*      SequentialExecutionPolicy semantics (execution/ExecutionPolicy.hpp:245-274,457-608)
 *   orc_omp_*  OmpExecutionPolicy semantics: one contiguous chunk per thread, Hillis-Steele over
 *              chunk totals, per-thread histograms + backward stable scatter
 *              (omp/execution/ExecutionPolicy.hpp:264-473, 891-1160).  These are the CPU baseline.
 */
#include "zpc_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#  include <omp.h>
#endif

#define OP_SUM(a, b) ((a) + (b))
#define OP_PROD(a, b) ((a) * (b))
#define OP_MIN(a, b) ((a) < (b) ? (a) : (b))
#define OP_MAX(a, b) ((a) > (b) ? (a) : (b))

/* unsigned arithmetic for the integer sum/prod so that wrap-around is defined behaviour */
#define DEF_SEQ(T, UT, S, LOWEST, HIGHEST)                                                     \
  /* reduce: left fold from init (ExecutionPolicy.hpp:267-274); inits as the C ABI passes them \
     (py_interop/cuda/ExecutionPolicy.cpp:43-70): 0, 1, numeric max, numeric lowest */         \
  void orc_reduce_sum_##S(const T *in, size_t n, T *out) {                                     \
    UT r = (UT)0;                                                                              \
    for (size_t i = 0; i < n; ++i) r = (UT)(r + (UT)in[i]);                                    \
    *out = (T)r;                                                                               \
  }                                                                                            \
  void orc_reduce_prod_##S(const T *in, size_t n, T *out) {                                    \
    UT r = (UT)1;                                                                              \
    for (size_t i = 0; i < n; ++i) r = (UT)(r * (UT)in[i]);                                    \
    *out = (T)r;                                                                               \
  }                                                                                            \
  void orc_reduce_min_##S(const T *in, size_t n, T *out) {                                     \
    T r = HIGHEST;                                                                             \
    for (size_t i = 0; i < n; ++i) r = OP_MIN(r, in[i]);                                       \
    *out = r;                                                                                  \
  }                                                                                            \
  void orc_reduce_max_##S(const T *in, size_t n, T *out) {                                     \
    T r = LOWEST;                                                                              \
    for (size_t i = 0; i < n; ++i) r = OP_MAX(r, in[i]);                                       \
    *out = r;                                                                                  \
  }                                                                                            \
  /* exclusive: d[0]=init; d[i]=op(d[i-1],a[i-1]) (ExecutionPolicy.hpp:257-264) */             \
  void orc_exclusive_scan_sum_##S(const T *in, size_t n, T *out) {                             \
    UT r = (UT)0;                                                                              \
    for (size_t i = 0; i < n; ++i) {                                                           \
      UT v = (UT)in[i];                                                                        \
      out[i] = (T)r;                                                                           \
      r = (UT)(r + v);                                                                         \
    }                                                                                          \
  }                                                                                            \
  void orc_exclusive_scan_prod_##S(const T *in, size_t n, T *out) {                            \
    UT r = (UT)1;                                                                              \
    for (size_t i = 0; i < n; ++i) {                                                           \
      UT v = (UT)in[i];                                                                        \
      out[i] = (T)r;                                                                           \
      r = (UT)(r * v);                                                                         \
    }                                                                                          \
  }                                                                                            \
  /* inclusive: d[0]=a[0]; d[i]=op(d[i-1],a[i]) (ExecutionPolicy.hpp:247-253) */               \
  void orc_inclusive_scan_sum_##S(const T *in, size_t n, T *out) {                             \
    UT r = (UT)0;                                                                              \
    for (size_t i = 0; i < n; ++i) {                                                           \
      r = i ? (UT)(r + (UT)in[i]) : (UT)in[i];                                                 \
      out[i] = (T)r;                                                                           \
    }                                                                                          \
  }                                                                                            \
  void orc_inclusive_scan_prod_##S(const T *in, size_t n, T *out) {                            \
    UT r = (UT)1;                                                                              \
    for (size_t i = 0; i < n; ++i) {                                                           \
      r = i ? (UT)(r * (UT)in[i]) : (UT)in[i];                                                 \
      out[i] = (T)r;                                                                           \
    }                                                                                          \
  }

DEF_SEQ(int32_t, uint32_t, i32, INT32_MIN, INT32_MAX)
DEF_SEQ(int64_t, uint64_t, i64, INT64_MIN, INT64_MAX)
DEF_SEQ(float, float, f32, -3.402823466e+38f, 3.402823466e+38f)
DEF_SEQ(double, double, f64, -1.7976931348623157e+308, 1.7976931348623157e+308)

/* ------------------------------------------------------------------ OMP-semantic reduce / scan */
static int clamp_threads(int nth, size_t n) {
  if (nth < 1) nth = 1;
  /* reference: "#pragma omp parallel if (_dop < dist)": serial region when n <= dop */
  if ((size_t)nth >= n) nth = 1;
  return nth;
}

#define DEF_OMP(T, UT, S)                                                                      \
  /* omp reduce_impl, omp/execution/ExecutionPolicy.hpp:420-473: every thread folds its chunk  \
     starting from init (0), then a log-step tree tmp = op(tmp, local[t+stride]) */            \
  void orc_omp_reduce_sum_##S(const T *in, size_t n, T *out, int nthreads) {                   \
    int nth = clamp_threads(nthreads, n);                                                      \
    UT *local = (UT *)calloc((size_t)nth, sizeof(UT));                                         \
    UT *tmpv = (UT *)calloc((size_t)nth, sizeof(UT));                                          \
    size_t nwork = (n + (size_t)nth - 1) / (size_t)nth;                                        \
    _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                          \
    for (int t = 0; t < nth; ++t) {                                                            \
      size_t st = nwork * (size_t)t, ed = st + nwork;                                          \
      if (ed > n) ed = n;                                                                      \
      UT r = (UT)0;                                                                            \
      for (size_t i = st; i < ed; ++i) r = (UT)(r + (UT)in[i]);                                \
      local[t] = r;                                                                            \
      tmpv[t] = r;                                                                             \
    }                                                                                          \
    for (int stride = 1; stride < nth; stride *= 2) {                                          \
      for (int t = 0; t + stride < nth; ++t) tmpv[t] = (UT)(tmpv[t] + local[t + stride]);      \
      for (int t = 0; t + stride < nth; ++t) local[t] = tmpv[t];                               \
    }                                                                                          \
    *out = n ? (T)tmpv[0] : (T)0;                                                              \
    free(local);                                                                               \
    free(tmpv);                                                                                \
  }                                                                                            \
  /* omp exclusive_scan_impl, :340-402 (init = identity) */                                    \
  void orc_omp_exclusive_scan_sum_##S(const T *in, size_t n, T *out, int nthreads) {           \
    int nth = clamp_threads(nthreads, n);                                                      \
    UT *local = (UT *)calloc((size_t)nth, sizeof(UT));                                         \
    UT *tmpv = (UT *)calloc((size_t)nth, sizeof(UT));                                          \
    size_t nwork = (n + (size_t)nth - 1) / (size_t)nth;                                        \
    _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                          \
    for (int t = 0; t < nth; ++t) {                                                            \
      size_t st = nwork * (size_t)t, ed = st + nwork;                                          \
      if (ed > n) ed = n;                                                                      \
      if (st < ed) {                                                                           \
        UT r = (UT)in[st];                                                                     \
        out[st] = (T)0;                                                                        \
        for (size_t i = st + 1; i < ed; ++i) {                                                 \
          UT v = (UT)in[i];                                                                    \
          out[i] = (T)r;                                                                       \
          r = (UT)(r + v);                                                                     \
        }                                                                                      \
        local[t] = r;                                                                          \
        tmpv[t] = r;                                                                           \
      }                                                                                        \
    }                                                                                          \
    for (int stride = 1; stride < nth; stride *= 2) {                                          \
      for (int t = nth - 1; t >= stride; --t)                                                  \
        if (nwork * (size_t)t < n) tmpv[t] = (UT)(tmpv[t] + local[t - stride]);                \
      for (int t = stride; t < nth; ++t)                                                       \
        if (nwork * (size_t)t < n) local[t] = tmpv[t];                                         \
    }                                                                                          \
    _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                          \
    for (int t = 1; t < nth; ++t) {                                                            \
      size_t st = nwork * (size_t)t, ed = st + nwork;                                          \
      if (ed > n) ed = n;                                                                      \
      if (st < ed) {                                                                           \
        UT add = local[t - 1];                                                                 \
        for (size_t i = st; i < ed; ++i) out[i] = (T)((UT)out[i] + add);                       \
      }                                                                                        \
    }                                                                                          \
    free(local);                                                                               \
    free(tmpv);                                                                                \
  }                                                                                            \
  /* omp inclusive_scan_impl, :264-339 */                                                      \
  void orc_omp_inclusive_scan_sum_##S(const T *in, size_t n, T *out, int nthreads) {           \
    int nth = clamp_threads(nthreads, n);                                                      \
    UT *local = (UT *)calloc((size_t)nth, sizeof(UT));                                         \
    UT *tmpv = (UT *)calloc((size_t)nth, sizeof(UT));                                          \
    size_t nwork = (n + (size_t)nth - 1) / (size_t)nth;                                        \
    _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                          \
    for (int t = 0; t < nth; ++t) {                                                            \
      size_t st = nwork * (size_t)t, ed = st + nwork;                                          \
      if (ed > n) ed = n;                                                                      \
      if (st < ed) {                                                                           \
        UT r = (UT)in[st];                                                                     \
        out[st] = (T)r;                                                                        \
        for (size_t i = st + 1; i < ed; ++i) {                                                 \
          r = (UT)(r + (UT)in[i]);                                                             \
          out[i] = (T)r;                                                                       \
        }                                                                                      \
        local[t] = r;                                                                          \
        tmpv[t] = r;                                                                           \
      }                                                                                        \
    }                                                                                          \
    for (int stride = 1; stride < nth; stride *= 2) {                                          \
      for (int t = nth - 1; t >= stride; --t)                                                  \
        if (nwork * (size_t)t < n) tmpv[t] = (UT)(tmpv[t] + local[t - stride]);                \
      for (int t = stride; t < nth; ++t)                                                       \
        if (nwork * (size_t)t < n) local[t] = tmpv[t];                                         \
    }                                                                                          \
    _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                          \
    for (int t = 1; t < nth; ++t) {                                                            \
      size_t st = nwork * (size_t)t, ed = st + nwork;                                          \
      if (ed > n) ed = n;                                                                      \
      if (st < ed) {                                                                           \
        UT add = local[t - 1];                                                                 \
        for (size_t i = st; i < ed; ++i) out[i] = (T)((UT)out[i] + add);                       \
      }                                                                                        \
    }                                                                                          \
    free(local);                                                                               \
    free(tmpv);                                                                                \
  }

DEF_OMP(int32_t, uint32_t, i32)
DEF_OMP(int64_t, uint64_t, i64)
DEF_OMP(float, float, f32)
DEF_OMP(double, double, f64)

/* ------------------------------------------------------------------ radix sort */
/* UT = unsigned type of the same width; SIGNBIT = 0 for unsigned keys */
#define DEF_SORT(T, UT, S, SIGNBIT)                                                            \
  static void seq_radix_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,         \
                            size_t n, int sbit, int ebit) {                                    \
    if (n == 0) return;                                                                        \
    UT *cur = (UT *)malloc(n * sizeof(UT)), *nxt = (UT *)malloc(n * sizeof(UT));               \
    int32_t *vcur = NULL, *vnxt = NULL;                                                        \
    if (vin) {                                                                                 \
      vcur = (int32_t *)malloc(n * sizeof(int32_t));                                           \
      vnxt = (int32_t *)malloc(n * sizeof(int32_t));                                           \
      memcpy(vcur, vin, n * sizeof(int32_t));                                                  \
    }                                                                                          \
    /* sign handling on copy-in (ExecutionPolicy.hpp:485-490) */                               \
    for (size_t i = 0; i < n; ++i) cur[i] = (UT)kin[i] ^ (UT)(SIGNBIT);                        \
    int binCount = 256;                                                                        \
    UT binMask = 255;                                                                          \
    size_t sizes[256], offs[256];                                                              \
    for (int st = sbit; st < ebit; st += 8) {                                                  \
      if (st + 8 > ebit) { /* narrowed last pass (:493-496) */                                 \
        binMask >>= (st + 8 - ebit);                                                           \
        binCount >>= (st + 8 - ebit);                                                          \
      }                                                                                        \
      for (int b = 0; b < binCount; ++b) sizes[b] = 0;                                         \
      for (size_t i = 0; i < n; ++i) sizes[(cur[i] >> st) & binMask]++;                        \
      int skip = sizes[0] == n; /* all-in-one-bin pass is skipped (:502-509) */                \
      offs[0] = 0;                                                                             \
      for (int b = 1; b < binCount && !skip; ++b) {                                            \
        if (sizes[b] == n) skip = 1;                                                           \
        offs[b] = offs[b - 1] + sizes[b - 1];                                                  \
      }                                                                                        \
      if (skip) continue;                                                                      \
      for (int b = 0; b < binCount; ++b) sizes[b] += offs[b]; /* end offsets */                \
      for (size_t i = n; i-- > 0;) { /* placed from the back => stable (:511-515) */           \
        size_t d = --sizes[(cur[i] >> st) & binMask];                                          \
        nxt[d] = cur[i];                                                                       \
        if (vin) vnxt[d] = vcur[i];                                                            \
      }                                                                                        \
      UT *t = cur; cur = nxt; nxt = t;                                                         \
      int32_t *tv = vcur; vcur = vnxt; vnxt = tv;                                              \
    }                                                                                          \
    for (size_t i = 0; i < n; ++i) kout[i] = (T)(cur[i] ^ (UT)(SIGNBIT));                      \
    if (vin) memcpy(vout, vcur, n * sizeof(int32_t));                                          \
    free(cur); free(nxt); free(vcur); free(vnxt);                                              \
  }                                                                                            \
  void orc_radix_sort_##S(const T *in, T *out, size_t n, int sbit, int ebit) {                 \
    seq_radix_##S(in, NULL, out, NULL, n, sbit, ebit);                                         \
  }                                                                                            \
  void orc_radix_sort_pair_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,      \
                               size_t n, int sbit, int ebit) {                                 \
    seq_radix_##S(kin, vin, kout, vout, n, sbit, ebit);                                        \
  }                                                                                            \
  /* omp radix_sort(_pair)_impl, omp/execution/ExecutionPolicy.hpp:891-1160 */                 \
  static void omp_radix_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,         \
                            size_t n, int sbit, int ebit, int nthreads) {                      \
    if (n == 0) return;                                                                        \
    int nth = clamp_threads(nthreads, n);                                                      \
    UT *cur = (UT *)malloc(n * sizeof(UT)), *nxt = (UT *)malloc(n * sizeof(UT));               \
    int32_t *vcur = NULL, *vnxt = NULL;                                                        \
    if (vin) {                                                                                 \
      vcur = (int32_t *)malloc(n * sizeof(int32_t));                                           \
      vnxt = (int32_t *)malloc(n * sizeof(int32_t));                                           \
    }                                                                                          \
    size_t nwork = (n + (size_t)nth - 1) / (size_t)nth;                                        \
    size_t(*bins)[256] = (size_t(*)[256])malloc((size_t)nth * sizeof(size_t[256]));            \
    _Pragma("omp parallel for num_threads(nth) schedule(static)")                             \
    for (size_t i = 0; i < n; ++i) {                                                           \
      cur[i] = (UT)kin[i] ^ (UT)(SIGNBIT);                                                     \
      if (vin) vcur[i] = vin[i];                                                               \
    }                                                                                          \
    int binCount = 256;                                                                        \
    UT binMask = 255;                                                                          \
    for (int st = sbit; st < ebit; st += 8) {                                                  \
      if (st + 8 > ebit) {                                                                     \
        binMask >>= (st + 8 - ebit);                                                           \
        binCount >>= (st + 8 - ebit);                                                          \
      }                                                                                        \
      /* per-thread histogram over its chunk (:961-963) */                                     \
      _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                        \
      for (int t = 0; t < nth; ++t) {                                                          \
        size_t stt = nwork * (size_t)t, ed = stt + nwork;                                      \
        if (ed > n) ed = n;                                                                    \
        for (int b = 0; b < binCount; ++b) bins[t][b] = 0;                                     \
        for (size_t i = stt; i < ed; ++i) bins[t][(cur[i] >> st) & binMask]++;                 \
      }                                                                                        \
      /* one thread: totals, skip detection, exclusive scan, per-thread END offsets (:969-989) */ \
      size_t tot[256], off[256];                                                               \
      int skip = 0;                                                                            \
      for (int b = 0; b < binCount; ++b) {                                                     \
        size_t s = 0;                                                                          \
        for (int t = 0; t < nth; ++t) s += bins[t][b];                                         \
        tot[b] = s;                                                                            \
        if (s == n) skip = 1;                                                                  \
      }                                                                                        \
      if (skip) continue;                                                                      \
      off[0] = 0;                                                                              \
      for (int b = 1; b < binCount; ++b) off[b] = off[b - 1] + tot[b - 1];                     \
      for (int b = 0; b < binCount; ++b) {                                                     \
        bins[0][b] += off[b];                                                                  \
        for (int t = 1; t < nth; ++t) bins[t][b] += bins[t - 1][b];                            \
      }                                                                                        \
      /* each thread scatters its chunk backwards with --end offset (:996-998) */              \
      _Pragma("omp parallel for num_threads(nth) schedule(static, 1)")                        \
      for (int t = 0; t < nth; ++t) {                                                          \
        size_t stt = nwork * (size_t)t, ed = stt + nwork;                                      \
        if (ed > n) ed = n;                                                                    \
        for (size_t i = ed; i-- > stt;) {                                                      \
          size_t d = --bins[t][(cur[i] >> st) & binMask];                                      \
          nxt[d] = cur[i];                                                                     \
          if (vin) vnxt[d] = vcur[i];                                                          \
        }                                                                                      \
      }                                                                                        \
      UT *tk = cur; cur = nxt; nxt = tk;                                                       \
      int32_t *tv = vcur; vcur = vnxt; vnxt = tv;                                              \
    }                                                                                          \
    _Pragma("omp parallel for num_threads(nth) schedule(static)")                             \
    for (size_t i = 0; i < n; ++i) {                                                           \
      kout[i] = (T)(cur[i] ^ (UT)(SIGNBIT));                                                   \
      if (vin) vout[i] = vcur[i];                                                              \
    }                                                                                          \
    free(cur); free(nxt); free(vcur); free(vnxt); free(bins);                                  \
  }                                                                                            \
  void orc_omp_radix_sort_##S(const T *in, T *out, size_t n, int sbit, int ebit, int nth) {    \
    omp_radix_##S(in, NULL, out, NULL, n, sbit, ebit, nth);                                    \
  }                                                                                            \
  void orc_omp_radix_sort_pair_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,  \
                                   size_t n, int sbit, int ebit, int nth) {                    \
    omp_radix_##S(kin, vin, kout, vout, n, sbit, ebit, nth);                                   \
  }

DEF_SORT(int32_t, uint32_t, i32, 0x80000000u)
DEF_SORT(uint32_t, uint32_t, u32, 0u)
DEF_SORT(int64_t, uint64_t, i64, 0x8000000000000000ull)
DEF_SORT(uint64_t, uint64_t, u64, 0ull)

/* ------------------------------------------------------------------------------------------------
 * merge_sort / merge_sort_pair, SequentialExecutionPolicy (execution/ExecutionPolicy.hpp:341-420):
 * insertion sort of every run of 16 (`:347-361`: swap while comp(keys[k], keys[k-1])), then bottom-up
 * passes halfStride = 16, 32, ... merging [ll, mid) and [mid, rr) into the other buffer, left element
 * first unless comp(b, a) (`:378-386`), tails copied (`:387-394`); result copied back when the ping-pong
 * ended in the temporary (`switched`, `:420`). */
#define DEF_MSORT(T, S)                                                                            \
  static int msort_lt_##S(T a, T b, int desc) { return desc ? (a > b) : (a < b); }                \
  void orc_merge_sort_##S(T *keys, int32_t *vals, size_t n, int desc) {                            \
    if (n == 0) return;                                                                            \
    for (size_t ll = 0; ll < n;) {                                                                 \
      size_t rr = ll + 16 < n ? ll + 16 : n;                                                       \
      for (size_t i = ll + 1; i != rr; ++i)                                                        \
        for (size_t k = i; k != ll; --k) {                                                         \
          size_t j = k - 1;                                                                        \
          if (!msort_lt_##S(keys[k], keys[j], desc)) break;                                        \
          T tk = keys[k]; keys[k] = keys[j]; keys[j] = tk;                                         \
          if (vals) { int32_t tv = vals[k]; vals[k] = vals[j]; vals[j] = tv; }                     \
        }                                                                                          \
      ll = rr;                                                                                     \
    }                                                                                              \
    T *ok = (T *)malloc(n * sizeof(T));                                                            \
    int32_t *ov = vals ? (int32_t *)malloc(n * sizeof(int32_t)) : NULL;                            \
    T *ck = keys, *nk = ok;                                                                        \
    int32_t *cv = vals, *nv = ov;                                                                  \
    for (size_t half = 16; half < n; half *= 2) {                                                  \
      size_t stride = half * 2;                                                                    \
      for (size_t ll = 0; ll < n; ll += stride) {                                                  \
        size_t mid = ll + half < n ? ll + half : n, rr = ll + stride < n ? ll + stride : n;        \
        size_t left = ll, right = mid, k = ll;                                                     \
        while (left < mid && right < rr) {                                                         \
          if (!msort_lt_##S(ck[right], ck[left], desc)) {                                          \
            nk[k] = ck[left]; if (vals) nv[k] = cv[left]; ++k; ++left;                             \
          } else {                                                                                 \
            nk[k] = ck[right]; if (vals) nv[k] = cv[right]; ++k; ++right;                          \
          }                                                                                        \
        }                                                                                          \
        while (left < mid) { nk[k] = ck[left]; if (vals) nv[k] = cv[left]; ++k; ++left; }          \
        while (right < rr) { nk[k] = ck[right]; if (vals) nv[k] = cv[right]; ++k; ++right; }       \
      }                                                                                            \
      T *tk = ck; ck = nk; nk = tk;                                                                \
      int32_t *tv = cv; cv = nv; nv = tv;                                                          \
    }                                                                                              \
    if (ck != keys) {                                                                              \
      memcpy(keys, ck, n * sizeof(T));                                                             \
      if (vals) memcpy(vals, cv, n * sizeof(int32_t));                                             \
    }                                                                                              \
    free(ok); free(ov);                                                                            \
  }
DEF_MSORT(int32_t, i32)
DEF_MSORT(uint32_t, u32)
DEF_MSORT(int64_t, i64)
DEF_MSORT(uint64_t, u64)
DEF_MSORT(float, f32)
DEF_MSORT(double, f64)
