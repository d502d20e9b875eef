/*
 * zpc_oracle.h -- CPU restatement of the zpc hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the MI355X implementation under zpc_amd/.  It is never
 * linked, imported or called by the product path: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  Every function cites the reference file:line whose
 * behaviour it restates (paths relative to the zenustech/zpc tree, include/zensim/...).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - 3x3 SVD, compute_stress_{fixedcorotated,sand}, quadratic B-spline weights, universal_hash /
 *     hash_combine, next_2pow / morton codes are pinned against the reference's own header-only
 *     code compiled in place (oracle/_ref, recipe oracle/Makefile) and against golden vectors
 *     generated from that build (tests/golden, script tools/gen_golden.py).
 *   - reduce / scan / radix sort on integers are pinned by uniqueness of the result (wrap-around
 *     integer add, stable sort) and cross-checked against numpy in tests; the reference's only
 *     in-tree test for them (test/utils/parallel_primitives.hpp:9-32, reduce == serial fold) is
 *     replayed in tests/test_oracle_primitives.py.
 *   - P2G / G2P (and through them the fused G2P2G step) as whole functions: PINNED.  The functor
 *     bodies (simulation/transfer/P2G.hpp:51-125, G2P.hpp:44-83) are spelled in oracle/ref_shim.cpp over
 *     the reference's own make_local_arena / unpack_coord_in_grid / compute_stress_* /
 *     matrixMatrixMultiplication3d (all header-only, built in place); tools/gen_golden.py writes
 *     tests/golden/p2g_g2p.npz (4096 particles, 5 models, block sides 4 and 8) and
 *     tests/test_oracle_cpu.py compares mpm.c with it.  The von Mises / NACC models exist in a host
 *     and a CUDA spelling that differ (sqrt iteration, NACC yield pressure): orc_mpm_params.hostVariant
 *     = 1 follows the host header (what the golden vectors were made with), 0 the CUDA header.
 *   - bht / HashTable (sequential insert + query) and P2C2G / G2C2P as whole functions: PINNED (r03).  The
 *     reference's containers, policies and functor headers need the un-vendored magic_enum / plog
 *     submodule headers and are unbuildable in this image, so -- as for P2G -- the bodies are spelled in
 *     oracle/ref_shim.cpp over the reference's own pieces (universal_hash_base, hash_combine, next_2pow,
 *     the mt19937 seeds; vec, lower_trunc, compute_stress_*, unpack_coord_in_grid); tools/gen_golden.py
 *     writes tests/golden/containers_seq.npz (tables byte for byte) and c2.npz, tests/test_oracle_cpu.py
 *     compares bht.c / hashtable.c / mpm.c with them.  Not covered by a fixture: P2C2G with the plastic
 *     models that carry logJp (the reference functor re-runs their update per (cell, particle) pair, this
 *     restatement once per particle), and the parallel policies' arrival-order-dependent table layouts
 *     (set semantics are what is compared there).
 *   - GridArena (math/curve/InterpolationKernel.hpp:272-470): the reference's own template instantiated
 *     over a dense box in ref_shim.cpp -> tests/golden/grid_arena.npz, compared with the C++ face on the GPU.
 */
#ifndef ZPC_ORACLE_H
#define ZPC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- primitives (prims.c) */
/* SequentialExecutionPolicy semantics, execution/ExecutionPolicy.hpp:245-274 */
#define ORC_DECL_PRIMS(T, S)                                                                  \
  void orc_reduce_sum_##S(const T *in, size_t n, T *out);                                     \
  void orc_reduce_prod_##S(const T *in, size_t n, T *out);                                    \
  void orc_reduce_min_##S(const T *in, size_t n, T *out);                                     \
  void orc_reduce_max_##S(const T *in, size_t n, T *out);                                     \
  void orc_exclusive_scan_sum_##S(const T *in, size_t n, T *out);                             \
  void orc_exclusive_scan_prod_##S(const T *in, size_t n, T *out);                            \
  void orc_inclusive_scan_sum_##S(const T *in, size_t n, T *out);                             \
  void orc_inclusive_scan_prod_##S(const T *in, size_t n, T *out);                            \
  /* OmpExecutionPolicy semantics (chunk per thread), omp/execution/ExecutionPolicy.hpp */    \
  void orc_omp_reduce_sum_##S(const T *in, size_t n, T *out, int nthreads);                   \
  void orc_omp_exclusive_scan_sum_##S(const T *in, size_t n, T *out, int nthreads);           \
  void orc_omp_inclusive_scan_sum_##S(const T *in, size_t n, T *out, int nthreads);
ORC_DECL_PRIMS(int32_t, i32)
ORC_DECL_PRIMS(int64_t, i64)
ORC_DECL_PRIMS(float, f32)
ORC_DECL_PRIMS(double, f64)

/* 8-bit LSD radix sort on bit window [sbit, ebit); signed keys ordered through a sign-bit XOR.
 * execution/ExecutionPolicy.hpp:457-608 (sequential), omp/...:891-1160 (omp variant). */
#define ORC_DECL_SORT(T, S)                                                                   \
  void orc_radix_sort_##S(const T *in, T *out, size_t n, int sbit, int ebit);                 \
  void orc_radix_sort_pair_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,      \
                               size_t n, int sbit, int ebit);                                 \
  void orc_omp_radix_sort_##S(const T *in, T *out, size_t n, int sbit, int ebit, int nth);    \
  void orc_omp_radix_sort_pair_##S(const T *kin, const int32_t *vin, T *kout, int32_t *vout,  \
                                   size_t n, int sbit, int ebit, int nth);
ORC_DECL_SORT(int32_t, i32)
ORC_DECL_SORT(uint32_t, u32)
ORC_DECL_SORT(int64_t, i64)
ORC_DECL_SORT(uint64_t, u64)

/* stable merge sort (in place), keys by `<` (descending != 0: by `>`), optional int32 values:
 * SequentialExecutionPolicy::merge_sort_pair, execution/ExecutionPolicy.hpp:341-420 -- insertion sort on runs of
 * 16, then bottom-up merges that take the LEFT element on ties (`!comp(b, a)`). */
#define ORC_DECL_MSORT(T, S) void orc_merge_sort_##S(T *keys, int32_t *vals, size_t n, int descending);
ORC_DECL_MSORT(int32_t, i32)
ORC_DECL_MSORT(uint32_t, u32)
ORC_DECL_MSORT(int64_t, i64)
ORC_DECL_MSORT(uint64_t, u64)
ORC_DECL_MSORT(float, f32)
ORC_DECL_MSORT(double, f64)

/* ---------------------------------------------------------------- TileVector (tilevector.c) */
/* element (chn, i) of TileVector<T, L> with C channels: container/TileVector.hpp:108,397 */
size_t orc_tv_offset(size_t i, size_t chn, size_t L, size_t C);
size_t orc_tv_num_tiles(size_t n, size_t L); /* TileVector.hpp:73-74 */
/* aosoa_iterator_port address: py_interop/GenericIterator.hpp:88-104 */
size_t orc_aosoa_offset(uint32_t idx, uint32_t numTileBits, uint32_t tileMask, uint32_t numChns);
/* AoS [n][C] <-> AoSoA pack/unpack of all channels */
void orc_tv_from_aos_f32(const float *aos, size_t n, size_t C, size_t L, float *tv);
void orc_tv_to_aos_f32(const float *tv, size_t n, size_t C, size_t L, float *aos);

/* ---------------------------------------------------------------- bht (bht.c) */
/* bht<int, dim, int, 16>: container/Bht.hpp:16-272, view :403-1072 */
typedef struct orc_bht orc_bht;
uint32_t orc_universal_hash_i32(uint32_t hx, uint32_t hy, int32_t k);          /* HashUtils.hpp:23-25 */
uint32_t orc_universal_hash_vec(uint32_t hx, uint32_t hy, const int32_t *k, int dim); /* :26-43 */
void orc_bht_hash_params(uint32_t out[6]); /* std::mt19937(2): Bht.hpp:165-169, Bcht.hpp:39-43 */
size_t orc_bht_table_size(size_t nExpected); /* evaluateTableSize, Bht.hpp:154-158 */
orc_bht *orc_bht_create(int dim, size_t nExpected);                 /* B = 16 */
orc_bht *orc_bht_create_b(int dim, size_t nExpected, int bucket);  /* B = 16 | 32 */
size_t orc_bht_table_size_b(size_t nExpected, int bucket);
void orc_bht_destroy(orc_bht *);
void orc_bht_reset(orc_bht *, int clearCnt);                 /* Bht.hpp:306-318 */
int32_t orc_bht_insert(orc_bht *, const int32_t *key);      /* host insert, Bht.hpp:612-664 */
int32_t orc_bht_query(const orc_bht *, const int32_t *key); /* Bht.hpp:667-698 */
void orc_bht_insert_many(orc_bht *, const int32_t *keys, size_t n, int32_t *ret);
void orc_bht_query_many(const orc_bht *, const int32_t *keys, size_t n, int32_t *ret);
int32_t orc_bht_size(const orc_bht *);
size_t orc_bht_get_table_size(const orc_bht *);
int32_t orc_bht_build_success(const orc_bht *);
const int32_t *orc_bht_active_keys(const orc_bht *); /* [size][dim] */
const int32_t *orc_bht_keys(const orc_bht *);        /* [tableSize][4 or dim-padded] storage keys */
const int32_t *orc_bht_indices(const orc_bht *);
const int32_t *orc_bht_status(const orc_bht *);       /* [tableSize], all -1 outside an insertion */
int orc_bht_key_stride(const orc_bht *);
void orc_bht_resize(orc_bht *, size_t newCapacity); /* Bht.hpp:320-340 */

/* ---------------------------------------------------------------- HashTable (hashtable.c) */
/* zs::HashTable<i32, dim, int>, container/HashTable.hpp:16-592: hash_combine hash, linear probing stride 127 */
typedef struct orc_hashtable orc_hashtable;
size_t orc_hashtable_table_size(size_t nExpected);                 /* :87-90 */
int32_t orc_hashtable_do_hash(const int32_t *key, int dim);        /* :496-500 */
orc_hashtable *orc_hashtable_create(int dim, size_t nExpected);
void orc_hashtable_destroy(orc_hashtable *);
void orc_hashtable_reset(orc_hashtable *, int clearCnt);           /* :212-228,294-299 */
int32_t orc_hashtable_insert(orc_hashtable *, const int32_t *key); /* :376-397 */
int orc_hashtable_insert_id(orc_hashtable *, const int32_t *key, int32_t id); /* :423-443 */
int32_t orc_hashtable_query(const orc_hashtable *, const int32_t *key);       /* :445-457 */
int32_t orc_hashtable_entry(const orc_hashtable *, const int32_t *key);       /* :458-470 */
void orc_hashtable_insert_many(orc_hashtable *, const int32_t *keys, size_t n, int32_t *ret);
void orc_hashtable_query_many(const orc_hashtable *, const int32_t *keys, size_t n, int32_t *ret);
int32_t orc_hashtable_size(const orc_hashtable *);
int32_t orc_hashtable_get_table_size(const orc_hashtable *);
const int32_t *orc_hashtable_active_keys(const orc_hashtable *);
const int32_t *orc_hashtable_keys(const orc_hashtable *);    /* [tableSize][dim] */
const int32_t *orc_hashtable_indices(const orc_hashtable *); /* [tableSize] */
void orc_hashtable_resize(orc_hashtable *, size_t nExpected);      /* :281-292 */
void orc_hashtable_preserve(orc_hashtable *, size_t nExpected);    /* :258-279 */
/* zs::IndexBuckets<3,i32,i32> via index_buckets_for_particles (simulation/particle/Query.tpp:9-58), sequential policy */
orc_hashtable *orc_index_buckets_for_particles(const float *pos, size_t n, float dx, float displacement, size_t expectedCells,
                                               int32_t **counts, int32_t **offsets, int32_t **indices);
void orc_index_buckets_free(int32_t *counts, int32_t *offsets, int32_t *indices);

/* ---------------------------------------------------------------- LBvh (lbvh.c) */
/* zs::LBvh<3, int, f32>, container/Bvh.hpp:86-492 (structure), :810-1082 (build), :1219-1248 (refit), :644-680 (traversal).
 * Boxes are [n][6] floats {min xyz, max xyz} = AABBBox<3, f32>. */
typedef struct orc_lbvh orc_lbvh;
uint32_t orc_morton_3d_32(float x, float y, float z);                       /* math/bit/Bits.h:122-125 */
void orc_lbvh_whole_box(const float *bvs, size_t n, float box[6]);          /* compute_bounding_box, Bvh.hpp:39-84 */
uint32_t orc_lbvh_morton(const float whole[6], const float bv[6]);          /* _build_init_mc_id, :177-188 */
orc_lbvh *orc_lbvh_create(void);
void orc_lbvh_destroy(orc_lbvh *);
void orc_lbvh_build(orc_lbvh *, const float *primBvs, size_t n, int refit);
void orc_lbvh_refit(orc_lbvh *, const float *primBvs);
size_t orc_lbvh_num_leaves(const orc_lbvh *);
size_t orc_lbvh_num_nodes(const orc_lbvh *);
const float *orc_lbvh_bvs(const orc_lbvh *);
const int32_t *orc_lbvh_parents(const orc_lbvh *);
const int32_t *orc_lbvh_levels(const orc_lbvh *);
const int32_t *orc_lbvh_leaf_inds(const orc_lbvh *);
const int32_t *orc_lbvh_aux_indices(const orc_lbvh *);
size_t orc_lbvh_iter_neighbors(const orc_lbvh *, const float bv[6], int32_t *out, size_t cap);

/* ---------------------------------------------------------------- MPM (mpm.c) */
/* McAdams 3x3 SVD, column-major 9-vectors: math/matrix/SVD.hpp:15-1030 */
void orc_svd3(const float F[9], float U[9], float S[3], float V[9]);
void orc_lame(float E, float nu, float *mu, float *lam); /* physics/ConstitutiveModel.hpp:34-38 */
/* physics/ConstitutiveModel_Vol_dP.hpp:10-47 */
void orc_stress_fixedcorotated(float volume, float mu, float lam, const float F[9], float PF[9]);
/* physics/ConstitutiveModel_Vol_dP.hpp:246-326 (F is projected in place, logJp updated) */
void orc_stress_sand(float volume, float mu, float lam, float cohesion, float beta,
                     float yieldSurface, int volCorrection, float *logJp, float F[9], float PF[9]);
/* quadratic B-spline arena: simulation/Utils.hpp:47-75, math/curve/InterpolationKernel.hpp:47-55,93-130 */
void orc_arena(float dx, const float pos[3], int32_t corner[3], float localPos[3], float w[9]);

typedef struct {
  int model;          /* 0 = FixedCorotated, 1 = DruckerPrager(sand), 2 = VonMisesFixedCorotated, 3 = NACC, 4 = EquationOfState */
  float dx, dt;
  float volume, E, nu;
  float cohesion, beta, yieldSurface;
  int volCorrection;
  int side;           /* block side length in cells: 4 (Grids<f32,3,4>) or 8 (SparseGrid<3,f32,8>) */
  int nthreads;       /* 1 = SequentialExecutionPolicy order, >1 = OmpExecutionPolicy with float CAS atomics */
  float yieldStress;  /* model 2 = VonMisesFixedCorotated */
  float xi, Msqr;     /* model 3 = NACC (also E, nu, beta) */
  int hardeningOn;
  float bulk, viscosity; /* model 4 = EquationOfState: J lives in component 0 of the F slot */
  int hostVariant;    /* von Mises / NACC only: 0 = cuda/physics/ConstitutiveModel.hpp (what CudaExecutionPolicy callers run, the
                         form the GPU path is compared with), 1 = physics/ConstitutiveModel_Vol_dP.hpp (the host header
                         simulation/transfer/P2G.hpp includes; the form tests/golden/p2g_g2p.npz holds) */
} orc_mpm_params;
void orc_stress_vonmises(float volume, float mu, float lam, float yieldStress, int hostVariant, float F[9], float PF[9]);
void orc_stress_nacc(float volume, float mu, float lam, float bm, float xi, float beta, float Msqr, int hardeningOn, int hostVariant,
                     float *logJp, float F[9], float PF[9]);
float orc_nacc_bulk(float E, float nu);
float orc_nacc_msqr(float fa);

/* sparsity/SparsityOp.hpp:59-115: ComputeSparsity + EnlargeSparsity on a bht keyed by block coord */
void orc_mpm_build_partition(orc_bht *table, const float *pos, size_t n, float dx, int side);
/* simulation/transfer/P2G.hpp:51-125.  grid: [nblocks][7][side^3] (m, mv xyz, rhs xyz). */
void orc_mpm_p2g(const orc_mpm_params *p, const orc_bht *table, size_t n, const float *mass,
                 const float *pos, const float *vel, const float *C, const float *F, float *logJp,
                 float *grid);
/* simulation/grid/GridOp.hpp:71-108 (v = mv/m + extf*dt; max |v|^2) */
void orc_mpm_grid_update(const orc_mpm_params *p, size_t nblocks, float *grid, const float extf[3],
                         float *maxVelSqr);
/* simulation/transfer/P2C2G.hpp:53-189 (kind 0 P2C2GTransfer), :346-439 (1 ...Momentum), :547-679 (2 ...Force); buckets = IndexBuckets of
 * cell size dx, displacement 0; each particle's constitutive update is evaluated once (see mpm.c) */
void orc_mpm_p2c2g(const orc_mpm_params *p, int kind, const orc_bht *table, const orc_hashtable *buckets, const int32_t *offsets,
                   const int32_t *indices, size_t n, const float *mass, const float *pos, const float *vel, const float *B,
                   const float *F, float *logJp, float *grid);
/* simulation/transfer/G2C2P.hpp:59-135 (accumulates into vel / B), :235-270 */
void orc_mpm_g2c2p(const orc_mpm_params *p, const orc_bht *table, const orc_hashtable *buckets, const int32_t *offsets,
                   const int32_t *indices, const float *pos, float *vel, float *B, const float *grid);
void orc_mpm_post_g2c2p(const orc_mpm_params *p, size_t n, float *pos, const float *vel, const float *B, float *F);
/* simulation/transfer/G2P.hpp:44-83 */
void orc_mpm_g2p(const orc_mpm_params *p, const orc_bht *table, size_t n, float *pos, float *vel,
                 float *C, float *F, const float *grid);

#ifdef __cplusplus
}
#endif
#endif
