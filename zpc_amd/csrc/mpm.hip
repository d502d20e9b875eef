// mpm.hip -- partition, index buckets, binning, grid update, constitutive test entries, owner classification, halo pack/unpack (see mpm_device.hpp for the kernels)
#include "mpm_device.hpp"

using namespace zsr;

extern "C" {


void zs_rocm_mpm_compute_sparsity(zs_rocm_policy *pol, zs_rocm_bht_3 *tab, zs_rocm_attr pos, size_t n, float dx, int side,
                                  int keyIsOrigin) {
  Launch L(pol, "ComputeSparsity");
  if (!n) return;
  hipLaunchKernelGGL(compute_sparsity_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, tab->t.dev(), make_port<float>(pos), n,
                     1.0f / dx, side, keyIsOrigin ? side : 1);
}
void zs_rocm_mpm_partition_for_particles(zs_rocm_policy *pol, zs_rocm_hashtable *tab, zs_rocm_attr pos, size_t n, float dx,
                                         int blocklen) {
  if (tab->dim != 3) return;
  zs_rocm_hashtable_reset(pol, tab, 1);  // CleanSparsity (SparsityCompute.tpp:19)
  Launch L(pol, "partition_for_particles");
  if (!n) return;
  hipLaunchKernelGGL(compute_sparsity_ht_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, tab->dev(), make_port<float>(pos), n,
                     1.0f / dx, blocklen);
}
zs_rocm_index_buckets *zs_rocm_index_buckets_create(void) { return new zs_rocm_index_buckets; }
void zs_rocm_index_buckets_destroy(zs_rocm_index_buckets *ib) {
  if (!ib) return;
  if (ib->table) zs_rocm_hashtable_destroy(ib->table);
  (void)hipFree(ib->indices); (void)hipFree(ib->offsets); (void)hipFree(ib->counts);
  delete ib;
}
void zs_rocm_index_buckets_get_view(const zs_rocm_index_buckets *ib, zs_rocm_index_buckets_view *v) {
  v->table = ib->table; v->indices = ib->indices; v->offsets = ib->offsets; v->counts = ib->counts;
  v->numBuckets = ib->numBuckets; v->numEntries = ib->numEntries; v->dx = ib->dx;
}
void zs_rocm_index_buckets_for_particles(zs_rocm_policy *pol, zs_rocm_index_buckets *ib, zs_rocm_attr pos, size_t n, float dx,
                                         float displacement, size_t expectedCells) {
  ib->dx = dx;
  ib->displacement = displacement;
  ib->dense = 0;
  // a time loop rebuilds the buckets every step: table and arrays are kept while they are large enough (hipMalloc / hipFree
  // synchronise the device and cost more than the kernels below)
  size_t want = expectedCells ? expectedCells : n;
  // the table lives on the device the policy runs on (one process per GPU: that is the rank's device, not device 0)
  const int tdev = pol->device >= 0 ? pol->device : current_device();
  if (ib->table && ib->table->devid != tdev) { zs_rocm_hashtable_destroy(ib->table); ib->table = nullptr; ib->tableFor = 0; }
  if (ib->table && ib->tableFor >= want && ib->tableFor <= 4 * want) {
    want = ib->tableFor;
    zs_rocm_hashtable_reset(pol, ib->table, 1);
  } else {
    if (ib->table) zs_rocm_hashtable_destroy(ib->table);
    ib->table = zs_rocm_hashtable_create(3, want, 1, tdev);  // Query.tpp:27 (created reset)
    ib->tableFor = want;
  }
  ib->numEntries = (int)n;
  ib->numBuckets = 0;
  if (!n) return;
  Launch L(pol, "index_buckets_for_particles");
  const float dxinv = 1.0f / dx;
  int nc = 0;
  int *full = (int *)L.temp(sizeof(int));
  for (;;) {
    ZSR_CHECK(hipMemsetAsync(full, 0, sizeof(int), L.stream));
    hipLaunchKernelGGL(ib_cells_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, ib->table->dev(), make_port<float>(pos), n, dxinv,
                       displacement, full);
    int isFull = 0;
    ZSR_CHECK(hipMemcpyAsync(&nc, ib->table->cnt, sizeof(int), hipMemcpyDeviceToHost, L.stream));
    ZSR_CHECK(hipMemcpyAsync(&isFull, full, sizeof(int), hipMemcpyDeviceToHost, L.stream));
    ZSR_CHECK(hipStreamSynchronize(L.stream));
    if (!isFull) break;
    // `expectedCells` underestimated the occupied cells by more than the table's 16x headroom: a larger table, again.  The
    // reference sizes the table by the particle count (an upper bound of the cells), which ends the loop at the latest.
    if (want >= n) {
      report_error(hipErrorOutOfMemory, "index_buckets_for_particles: hash table full at one slot per particle", __FILE__, __LINE__);
      ib->numEntries = 0;
      return;
    }
    want = std::min(n, want * 8);
    zs_rocm_hashtable_destroy(ib->table);
    ib->table = zs_rocm_hashtable_create(3, want, 1, tdev);
    ib->tableFor = want;
  }
  ib->numBuckets = nc;
  const size_t numCells = (size_t)nc + 1;  // Query.tpp:36
  if (numCells > ib->capCells) {
    (void)hipFree(ib->offsets); (void)hipFree(ib->counts);
    ib->capCells = numCells + numCells / 2;
    ZSR_CHECK(hipMalloc((void **)&ib->counts, ib->capCells * sizeof(int)));
    ZSR_CHECK(hipMalloc((void **)&ib->offsets, ib->capCells * sizeof(int)));
  }
  if (n > ib->capEntries) {
    (void)hipFree(ib->indices);
    ib->capEntries = n;
    ZSR_CHECK(hipMalloc((void **)&ib->indices, n * sizeof(int)));
  }
  ZSR_CHECK(hipMemsetAsync(ib->counts, 0, numCells * sizeof(int), L.stream));
  unsigned *cellOf = (unsigned *)L.temp(sizeof(unsigned) * n), *cellSorted = (unsigned *)L.temp(sizeof(unsigned) * n);
  int *ids = (int *)L.temp(sizeof(int) * n);
  hipLaunchKernelGGL(ib_count_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, ib->table->dev(), make_port<float>(pos), n, dxinv,
                     displacement, (unsigned *)ib->counts, cellOf, ids);
  exclusive_scan_u32(L, (const unsigned *)ib->counts, numCells, (unsigned *)ib->offsets);
  // SpatiallyDistribute as a stable sort of (bucket, particle id): ids ascend inside a bucket
  int bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < numCells) ++bits;
  radix_sort_pair_u32(L, cellOf, ids, cellSorted, ib->indices, n, 0, bits);
}
void zs_rocm_index_buckets_for_partition(zs_rocm_policy *pol, zs_rocm_index_buckets *ib, zs_rocm_attr pos, size_t n, float dx,
                                         const zs_rocm_bht_3 *tab, int side, int keyIsOrigin) {
  ib->dx = dx;
  ib->displacement = 0.f;
  if (ib->table) { zs_rocm_hashtable_destroy(ib->table); ib->table = nullptr; ib->tableFor = 0; }
  ib->dense = 1;
  ib->denseSide = side;
  ib->numEntries = (int)n;
  ib->numBuckets = 0;
  Launch L(pol, "index_buckets_for_partition");
  const int nb = bht_size(tab->t, L.stream);
  if (!n || !nb || (side != 4 && side != 8)) return;
  const size_t nbuckets = (size_t)nb * side * side * side, numCells = nbuckets + 2;  // + the bucket of the unlisted particles + end
  ib->numBuckets = (int)nbuckets;
  if (numCells > ib->capCells) {
    (void)hipFree(ib->offsets); (void)hipFree(ib->counts);
    ib->capCells = numCells + numCells / 2;
    ZSR_CHECK(hipMalloc((void **)&ib->counts, ib->capCells * sizeof(int)));
    ZSR_CHECK(hipMalloc((void **)&ib->offsets, ib->capCells * sizeof(int)));
  }
  if (n > ib->capEntries) {
    (void)hipFree(ib->indices);
    ib->capEntries = n;
    ZSR_CHECK(hipMalloc((void **)&ib->indices, n * sizeof(int)));
  }
  ZSR_CHECK(hipMemsetAsync(ib->counts, 0, numCells * sizeof(int), L.stream));
  unsigned *cellOf = (unsigned *)L.temp(sizeof(unsigned) * n), *cellSorted = (unsigned *)L.temp(sizeof(unsigned) * n);
  int *ids = (int *)L.temp(sizeof(int) * n);
  hipLaunchKernelGGL(ib_dense_count_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, tab->t.dev(), make_port<float>(pos), n, 1.0f / dx,
                     side, keyIsOrigin ? side : 1, (int)nbuckets, (unsigned *)ib->counts, cellOf, ids);
  exclusive_scan_u32(L, (const unsigned *)ib->counts, numCells, (unsigned *)ib->offsets);
  int bits = 1;
  while (bits < 32 && ((size_t)1 << bits) < numCells) ++bits;
  radix_sort_pair_u32(L, cellOf, ids, cellSorted, ib->indices, n, 0, bits);  // stable: ids ascend inside a bucket
}
void zs_rocm_mpm_enlarge_sparsity__hashtable(zs_rocm_policy *pol, zs_rocm_hashtable *tab, const int lo[3], const int hi[3]) {
  if (tab->dim != 3) return;
  Launch L(pol, "enlarge_sparsity");
  int nb = 0;
  ZSR_CHECK(hipMemcpyAsync(&nb, tab->cnt, sizeof(int), hipMemcpyDeviceToHost, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  const int e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
  if (nb <= 0 || e0 <= 0 || e1 <= 0 || e2 <= 0) return;
  hipLaunchKernelGGL(enlarge_sparsity_ht_kernel, dim3(ceil_div((size_t)nb * e0 * e1 * e2, 256)), dim3(256), 0, L.stream, tab->dev(), nb,
                     lo[0], lo[1], lo[2], e0, e1, e2);
}
void zs_rocm_mpm_enlarge_sparsity(zs_rocm_policy *pol, zs_rocm_bht_3 *tab, const int lo[3], const int hi[3], int keyStride) {
  Launch L(pol, "EnlargeSparsity");
  const int nb = bht_size(tab->t, L.stream);
  const int e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
  if (nb == 0 || e0 <= 0 || e1 <= 0 || e2 <= 0) return;
  hipLaunchKernelGGL(enlarge_sparsity_kernel, dim3(ceil_div((size_t)nb * e0 * e1 * e2, 256)), dim3(256), 0, L.stream, tab->t.dev(), nb,
                     lo[0], lo[1], lo[2], e0, e1, e2, keyStride > 0 ? keyStride : 1);
}
void zs_rocm_mpm_build_neighbors(zs_rocm_policy *pol, const zs_rocm_bht_3 *tab, int *nbr, int keyStride) {
  Launch L(pol, "build_neighbors");
  const int nb = bht_size(tab->t, L.stream);
  if (!nb) return;
  hipLaunchKernelGGL(build_neighbors_kernel, dim3(ceil_div((size_t)nb * 8, 256)), dim3(256), 0, L.stream, tab->t.dev(), nb, nbr, keyStride > 0 ? keyStride : 1);
}

void zs_rocm_mpm_bin_particles(zs_rocm_policy *pol, const zs_rocm_bht_3 *tab, zs_rocm_attr pos, size_t n, float dx, int side,
                               int keyIsOrigin, int *order, int *binStart, unsigned *cellCount) {
  Launch L(pol, "bin_particles");
  const int nb = bht_size(tab->t, L.stream);
  if (nb == 0) return;
  const int nbins = nb * (side == 4 ? 1 : 8);
  const size_t ncells = (size_t)nbins * 64;
  unsigned *cellStart = (unsigned *)L.temp(sizeof(unsigned) * (ncells + 1));
  unsigned *cellOf = (unsigned *)L.temp(sizeof(unsigned) * (n + 1));
  unsigned *rankOf = (unsigned *)L.temp(sizeof(unsigned) * (n + 1));
  int *byCell = (int *)L.temp(sizeof(int) * (n + 1));
  int *err = (int *)L.temp(sizeof(int));
  ZSR_CHECK(hipMemsetAsync(cellCount, 0, sizeof(unsigned) * ncells, L.stream));
  ZSR_CHECK(hipMemsetAsync(err, 0, sizeof(int), L.stream));
  BhtDev t = tab->t.dev();
  Port<float> pp = make_port<float>(pos);
  if (n) {
    if (side == 4)
      hipLaunchKernelGGL((bin_count_kernel<4>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t, pp, n, dx, cellCount, cellOf, rankOf, err, keyIsOrigin ? side : 1);
    else
      hipLaunchKernelGGL((bin_count_kernel<8>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, t, pp, n, dx, cellCount, cellOf, rankOf, err, keyIsOrigin ? side : 1);
  }
  exclusive_scan_u32(L, cellCount, ncells, cellStart);
  if (n)
    hipLaunchKernelGGL(bin_place_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, n, (const unsigned *)cellStart,
                       (const unsigned *)cellOf, (const unsigned *)rankOf, byCell);
  hipLaunchKernelGGL(bin_roundrobin_kernel, dim3(nbins), dim3(64), 0, L.stream, nbins, (const unsigned *)cellStart,
                     (const unsigned *)cellCount, (const int *)byCell, order, binStart, (unsigned)n);
  int herr = 0;
  ZSR_CHECK(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, L.stream));
  ZSR_CHECK(hipStreamSynchronize(L.stream));
  if (herr) fprintf(stderr, "[zs_rocm] bin_particles: particles outside the partition were dropped from the bins\n");
}


void zs_rocm_mpm_grid_update(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, float *grid, size_t nblocks, const float extf[3],
                             float *maxVelSqr) {
  Launch L(pol, "ComputeGridBlockVelocity");
  if (!nblocks) return;
  const size_t nc = (size_t)p->side * p->side * p->side;
  if (p->side == 4)
    hipLaunchKernelGGL((grid_update_kernel<4>), dim3(ceil_div(nblocks * nc, 256)), dim3(256), 0, L.stream, grid, nblocks, p->dt, extf[0],
                       extf[1], extf[2], maxVelSqr);
  else
    hipLaunchKernelGGL((grid_update_kernel<8>), dim3(ceil_div(nblocks * nc, 256)), dim3(256), 0, L.stream, grid, nblocks, p->dt, extf[0],
                       extf[1], extf[2], maxVelSqr);
}



int zs_rocm_mpm_stress_channels(void) { return STRESS_N; }
void zs_rocm_mpm_update_stress(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, zs_rocm_particles ps) {
  Launch L(pol, "update_stress");
  if (!ps.n || !ps.stress.base) return;
  MpmDev mp = make_dev(p);
  ParticlesDev pd = make_particles(ps);
#define CALL_UPDATE_STRESS(S, M) hipLaunchKernelGGL((update_stress_kernel<M>), dim3(ceil_div(ps.n, 256)), dim3(256), 0, L.stream, mp, pd)
  if (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE) return;
  ZSR_DISPATCH_PURE_(0, p->model, CALL_UPDATE_STRESS)
}

void zs_rocm_mpm_stress(zs_rocm_policy *pol, const zs_rocm_mpm_params *p, float *F, float *logJp, size_t n, float *PF) {
  Launch L(pol, "compute_stress");
  if (!n) return;
  MpmDev mp = make_dev(p);
#define CALL_STRESS(S, M) hipLaunchKernelGGL((stress_kernel<M>), dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, mp, F, logJp, n, PF)
  if (p->model < ZS_MPM_FIXED_COROTATED || p->model > ZS_MPM_EQUATION_OF_STATE) return;
  ZSR_DISPATCH_PURE_(0, p->model, CALL_STRESS)
}
float zs_rocm_nacc_msqr(float fa) {  // NACCConfig::mohrColumbFriction / M / Msqr, dim = 3 (physics/ConstitutiveModel.hpp:771-785)
  const int dim = 3;
  const float sin_phi = std::sin(fa);  // the reference passes `fa` to sin() as it is
  const float mcf = std::sqrt(2.f / 3.f) * 2.f * sin_phi / (3.f - sin_phi);
  const float M = mcf * dim / std::sqrt(2.f / (6.f - dim));
  return M * M;
}
void zs_rocm_svd3(zs_rocm_policy *pol, const float *F, size_t n, float *U, float *S, float *V) {
  Launch L(pol, "svd3");
  if (!n) return;
  hipLaunchKernelGGL(svd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, F, n, U, S, V);
}

void zs_rocm_mpm_owner_rank(zs_rocm_policy *pol, zs_rocm_attr pos, size_t n, float dx, const int lo[3], const int hi[3], const int dims[3],
                            int align, int *owner) {
  Launch L(pol, "owner_rank");
  if (!n) return;
  OwnerSplit sp;
  for (int d = 0; d < 3; ++d) { sp.lo[d] = lo[d]; sp.hi[d] = hi[d]; sp.dims[d] = dims[d]; }
  sp.align = align < 1 ? 1 : align;
  hipLaunchKernelGGL(owner_rank_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, L.stream, make_port<float>(pos), n, 1.0f / dx, sp, owner);
}
// counts[r] = number of i with owner[i] == r, r in [0, world) (world <= 1024): how many particles a migration sends to each rank
void zs_rocm_mpm_owner_counts(zs_rocm_policy *pol, const int *owner, size_t n, int world, int *counts) {
  Launch L(pol, "owner_counts");
  if (world < 1 || world > 1024) return;
  ZSR_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)world, L.stream));
  if (!n) return;
  const unsigned blocks = ceil_div(n, 256 * 16) < 2048u ? ceil_div(n, 256 * 16) : 2048u;
  hipLaunchKernelGGL(owner_count_kernel, dim3(blocks), dim3(256), sizeof(int) * (size_t)world, L.stream, owner, n, world, counts);
}
void zs_rocm_mpm_halo_pack(zs_rocm_policy *pol, const float *grid, const int *blocks, size_t nb, int side, int chn0, int nchn, float *buf) {
  Launch L(pol, "halo_pack");
  const int nc = side * side * side;
  if (!nb) return;
  hipLaunchKernelGGL(halo_pack_kernel, dim3(ceil_div(nb * nchn * nc, 256)), dim3(256), 0, L.stream, grid, blocks, nb, nc, chn0, nchn, buf);
}
void zs_rocm_mpm_halo_unpack(zs_rocm_policy *pol, float *grid, const int *blocks, size_t nb, int side, int chn0, int nchn, const float *buf,
                             int add) {
  Launch L(pol, "halo_unpack");
  const int nc = side * side * side;
  if (!nb) return;
  if (add == 2)
    hipLaunchKernelGGL((halo_unpack_kernel<2>), dim3(ceil_div(nb * nchn * nc, 256)), dim3(256), 0, L.stream, grid, blocks, nb, nc, chn0,
                       nchn, buf);
  else if (add)
    hipLaunchKernelGGL((halo_unpack_kernel<1>), dim3(ceil_div(nb * nchn * nc, 256)), dim3(256), 0, L.stream, grid, blocks, nb, nc, chn0,
                       nchn, buf);
  else
    hipLaunchKernelGGL((halo_unpack_kernel<0>), dim3(ceil_div(nb * nchn * nc, 256)), dim3(256), 0, L.stream, grid, blocks, nb, nc, chn0,
                       nchn, buf);
}


}  // extern "C"
