// C++ template face test, mirroring the reference's own CUDA tests on the rocm slot:
//   test/cuda/main.cu:7-32    reduce(getmax/getmin/plus) over the "b" channel of TileVector<int,32>{a:3,b:2,c:1}
//   test/cuda/basic.cu:53-161 Vector fill on device + clone compare; TileVector<float,32> named channels, pack / tuple
// plus the launcher shapes (Collapse 1/2/3-D, shmem-first lambdas), bht insert/query inside lambdas, atomics.
// Build: hipcc --offload-arch=gfx950 -std=c++17 -I include tests/cpp/test_cpp_face.hip -L zpc_amd/lib -lzsrocm
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <cmath>
#include <numeric>
#include <string>
#include <vector>

#include <hip/hip_cooperative_groups.h>

#include "zensim_rocm/zs_rocm.hpp"
#include "zensim_rocm/collider_device.hpp"

#define CHECK(c)                                                        \
  do {                                                                  \
    if (!(c)) {                                                         \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);        \
      std::exit(1);                                                     \
    }                                                                   \
  } while (0)

using namespace zs;
constexpr auto space = execspace_e::rocm;


// The reference's own primitive test, with its statements (test/utils/parallel_primitives.hpp:9-32, test/parallel_primitives.cpp:9-33):
// reduce over range(tv, "b") into a Vector built from the policy's temporary allocator, a zip-range copy through the policy, a clone
// to host memory and a serial fold.  It compiles against this header unchanged except for the policy type.
template <typename Pol, typename Range, typename Op> bool test_reduction(Pol &&policy, Range &&r, Op op) {
  using namespace zs;
  static_assert(std::is_lvalue_reference_v<decltype(*std::begin(r))>, "deref of the iterator shall be a lvalue");
  using value_t = std::remove_reference_t<decltype(*std::begin(r))>;
  static_assert(std::is_fundamental_v<value_t>, "value_t should be a fundamental type");
  const auto sz = range_size(r);
  auto mop = make_monoid(op);
  auto allocator = get_temporary_memory_source(policy);
  Vector<value_t> res{allocator, 1};
  reduce(policy, std::begin(r), std::end(r), std::begin(res), mop.identity(), op);
  Vector<value_t> vals{allocator, (size_t)sz};
  policy(zip(r, vals), [] ZS_LAMBDA(const value_t &src, value_t &dst) mutable { dst = src; });
  vals = vals.clone({memsrc_e::host, -1});
  value_t e = mop.identity();
  for (size_t i = 0; i != sz; ++i) e = op(e, vals[i]);
  if constexpr (std::is_integral_v<value_t>) return e == res.getVal();
  else return std::abs(e - res.getVal()) / e < 1e-6;
}
// gen_rnd_tv_ints (test/utils/initialization.hpp): TileVector<int, 32> {a:3, b:2, c:1} filled with pseudo-random ints < bound
static zs::TileVector<int, 32> gen_rnd_tv_ints(size_t n, int bound) {
  using namespace zs;
  TileVector<int, 32> tv({{"a", 3}, {"b", 2}, {"c", 1}}, n, memsrc_e::um);
  unsigned long long x = 88172645463325252ull;
  auto v = view<execspace_e::rocm>(tv);
  for (size_t i = 0; i < n; ++i)
    for (int c = 0; c < 6; ++c) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      v(c, i) = (int)(x % (unsigned long long)(bound > 0 ? bound : 1 << 30)) - (bound > 0 ? 0 : (1 << 29));
    }
  return tv;
}
// ---- GridArena against the reference's own GridArena (tests/golden/grid_arena.npz, flattened to a binary file by the python wrapper):
// 58 floats per (case, point) -- stencil corner, local position, weights / first / second derivatives per axis (4 slots each), isample of
// two channels, minimum, maximum, weight + weightsGradient at three stencil nodes (oracle/ref_shim.cpp: ref_grid_arena)
template <zs::kernel_e kt, int order, class View>
__device__ void arena_record(const View &g, const float *X, int f, float dflt, float *o) {
  using namespace zs;
  const small_vec<float, 3> x{{X[0], X[1], X[2]}};
  auto ar = f < 0 ? g.template iArena<kt, order>(x) : g.template iArena<kt, order>(x, f);
  constexpr int W = decltype(ar)::width;
  for (int d = 0; d < 3; ++d) *o++ = (float)ar.iCorner[d];
  for (int d = 0; d < 3; ++d) *o++ = ar.iLocalPos[d];
  for (int q = 0; q < 3; ++q)
    for (int d = 0; d < 3; ++d)
      for (int k = 0; k < 4; ++k) *o++ = (q <= order && k < W) ? ar.w[q <= order ? q : 0][d][k < W ? k : 0] : 0.f;
  *o++ = ar.isample(0, dflt);
  *o++ = ar.isample(1, dflt);
  *o++ = ar.minimum(0);
  *o++ = ar.maximum(1);
  const int locs[3][3] = {{0, 0, 0}, {W - 1, 0, 1 % W}, {1 % W, W - 1, W - 1}};
  for (int l = 0; l < 3; ++l) {
    const small_vec<int, 3> loc{{locs[l][0], locs[l][1], locs[l][2]}};
    *o++ = ar.weight(loc);
    if constexpr (order > 0) {
      const auto gw = ar.weightsGradient(loc);
      for (int d = 0; d < 3; ++d) *o++ = gw[d];
    } else {
      for (int d = 0; d < 3; ++d) *o++ = 0.f;
    }
  }
}
static int grid_arena_against_reference(const char *path) {
  using namespace zs;
  constexpr auto space = execspace_e::rocm;
  std::FILE *fp = std::fopen(path, "rb");
  if (!fp) return 1;
  int hdr[6];  // ncases, npts, ext, lo[3]
  float fh[2];  // dx, default
  if (std::fread(hdr, 4, 6, fp) != 6 || std::fread(fh, 4, 2, fp) != 2) return 1;
  const int ncases = hdr[0], npts = hdr[1], ext = hdr[2];
  const int lo[3] = {hdr[3], hdr[4], hdr[5]};
  const float dx = fh[0], dflt = fh[1];
  std::vector<int> cases(3 * ncases);
  std::vector<float> data((size_t)2 * ext * ext * ext), X(3 * (size_t)npts), want((size_t)ncases * npts * 58);
  if (std::fread(cases.data(), 4, cases.size(), fp) != cases.size() || std::fread(data.data(), 4, data.size(), fp) != data.size() ||
      std::fread(X.data(), 4, X.size(), fp) != X.size() || std::fread(want.data(), 4, want.size(), fp) != want.size())
    return 1;
  std::fclose(fp);
  auto pol = rocm_exec();
  // the box [lo, lo + ext)^3 inside a SparseGrid<3, f32, 8>: every block the box touches is allocated; cells of those blocks outside
  // the box hold the default value, which is what valueOr returns for unallocated cells (the reference's dense view returns it for both)
  SparseGrid<3, float, 8> sg(std::vector<PropertyTag>{{"a", 1}, {"b", 1}}, 64);
  sg.scale(dx);
  sg._background = dflt;
  Vector<float> dd(data.size(), memsrc_e::um), dX(X.size(), memsrc_e::um), got((size_t)npts * 58, memsrc_e::um);
  std::copy(data.begin(), data.end(), dd.data());
  std::copy(X.begin(), X.end(), dX.data());
  const int nc = ext * ext * ext;
  pol(range(nc), [g = view<space>(sg), dx, ext, l0 = lo[0], l1 = lo[1], l2 = lo[2]] ZS_LAMBDA(long long c) {
    const int i = (int)(c / (ext * ext)), j = (int)(c / ext) % ext, k = (int)(c % ext);
    g.insert(small_vec<float, 3>{{(l0 + i + 0.25f) * dx, (l1 + j + 0.25f) * dx, (l2 + k + 0.25f) * dx}});
  });
  const std::size_t nb = sg.numBlocks();
  CHECK(nb >= 8 && nb <= 64);
  pol(range((long long)nb * 512), [g = view<space>(sg), d = view<space>(dd), dflt, ext, l0 = lo[0], l1 = lo[1], l2 = lo[2]] ZS_LAMBDA(long long c) {
    const int b = (int)(c / 512), k = (int)(c % 512);
    const auto ic = g.iCoord(b, k);
    const int i = ic[0] - l0, j = ic[1] - l1, q = ic[2] - l2;
    const bool in = i >= 0 && i < ext && j >= 0 && j < ext && q >= 0 && q < ext;
    for (int ch = 0; ch < 2; ++ch) g(ch, b, k) = in ? d[(((std::size_t)ch * ext + i) * ext + j) * ext + q] : dflt;
  });
  int bad = 0;
  for (int ci = 0; ci < ncases; ++ci) {
    const int kt = cases[3 * ci], order = cases[3 * ci + 1], f = cases[3 * ci + 2];
    pol(range(npts), [g = view<space>(sg), x = view<space>(dX), o = view<space>(got), kt, order, f, dflt] ZS_LAMBDA(long long i) {
      float *out = &o[(std::size_t)i * 58];
      const float *X = &x[3 * i];
#define GA(K, KT, O) if (kt == K && order == O) { arena_record<KT, O>(g, X, f, dflt, out); return; }
      GA(0, kernel_e::linear, 0) GA(0, kernel_e::linear, 1) GA(0, kernel_e::linear, 2)
      GA(1, kernel_e::quadratic, 0) GA(1, kernel_e::quadratic, 1) GA(1, kernel_e::quadratic, 2)
      GA(2, kernel_e::cubic, 0) GA(2, kernel_e::cubic, 1) GA(2, kernel_e::cubic, 2)
      GA(3, kernel_e::delta2, 0) GA(4, kernel_e::delta3, 0) GA(5, kernel_e::delta4, 0)
#undef GA
    });
    const int widthOf[6] = {2, 3, 4, 2, 3, 4};  // linear, quadratic, cubic, delta2, delta3, delta4
    for (int i = 0; i < npts; ++i) {
      // minimum / maximum pad missing cells with +-max, not with the caller's default: the sparse grid of this test stores the default in the
      // cells of allocated blocks outside the box, the reference's dense view has no such cells -- compared only where the arena stays inside
      bool inside = true;
      for (int d = 0; d < 3; ++d) {
        const int c0 = (int)want[((std::size_t)ci * npts + i) * 58 + d];
        inside = inside && c0 >= lo[d] && c0 + widthOf[kt] <= lo[d] + ext;
      }
      for (int k = 0; k < 58; ++k) {
        if (!inside && (k == 44 || k == 45)) continue;
        const float a = got.data()[(std::size_t)i * 58 + k], b = want[((std::size_t)ci * npts + i) * 58 + k];
        // corner / local position: exact and 1e-6; weights: 2e-6 of values in [-6, 6] (second derivatives of the cubic);
        // samples: 27 / 64-term sums of values of size ~3: 2e-5
        const float tol = k < 3 ? 0.f : (k < 6 ? 1e-6f : (k < 42 ? 2e-6f * std::max(1.f, std::fabs(b)) : 2e-5f * std::max(1.f, std::fabs(b))));
        if (!(std::fabs(a - b) <= tol)) {
          if (bad < 8) std::printf("grid arena mismatch: case %d (kernel %d, order %d, face %d) point %d slot %d: %g vs reference %g\n", ci, kt, order, f, i, k, a, b);
          ++bad;
        }
      }
    }
  }
  std::printf("grid arena vs reference: %d cases x %d points, %d mismatches\n", ncases, npts, bad);
  return bad ? 1 : 0;
}

int main(int argc, char **argv) {
  if (argc > 2 && std::string(argv[1]) == "--grid-arena") return grid_arena_against_reference(argv[2]);
  auto pol = rocm_exec();
  CHECK(pol.shouldSync());
  // ---- Vector: fill on device, clone to host, compare (basic.cu:65-107)
  {
    const int n = 100000;
    Vector<int> v(n, memsrc_e::device);
    pol(range(n), [vv = view<space>(v)] ZS_LAMBDA(long long i) { vv[i] = (int)(i * 3 + 1); });
    auto h = v.clone(memsrc_e::host);
    for (int i = 0; i < n; ++i) CHECK(h.data()[i] == i * 3 + 1);
    v.resize(120000);
    CHECK(v.size() == 120000 && v.capacity() == 150000 && v.getVal(7) == 22);
  }
  // ---- TileVector<float,32>: named channels, pack / set (basic.cu:110-151)
  {
    const int n = 1000;
    TileVector<float, 32> tv({{"m", 1}, {"x", 3}, {"v", 3}, {"F", 9}}, n, memsrc_e::um);
    CHECK(tv.numChannels() == 16 && tv.getPropertyOffset("v") == 4 && tv.getPropertyOffset("nope") == -1);
    const int xo = tv.getPropertyOffset("x"), mo = tv.getPropertyOffset("m");
    pol(range(n), [t = view<space>({}, tv), xo, mo] ZS_LAMBDA(long long i) {
      t(mo, i) = (float)i;
      t.set(xo, i, small_vec<float, 3>{{(float)i, (float)i + 0.5f, -(float)i}});
    });
    for (int i = 0; i < n; ++i) {  // um: read on the host through the layout formula
      const float *b = tv.data() + ((size_t)(i / 32) * 16 + xo) * 32 + i % 32;
      CHECK(b[0] == (float)i && b[32] == (float)i + 0.5f && b[64] == -(float)i);
    }
    Vector<float> sum(1);
    sum.setVal(0.f);
    pol(range(n), [t = view<space>(tv), xo, s = view<space>(sum)] ZS_LAMBDA(long long i) {
      auto p = t.pack(dim_c<3>, xo, i);
      atomic_add(exec_rocm, &s[0], p[0] + p[1] + p[2]);
    });
    CHECK(std::abs(sum.getVal() - (0.5f * n + (float)n * (n - 1) / 2)) < 1.0f);
  }
  // ---- zs::Particles / ParticlesView and zs::Grids / GridsView (geometry/Structurefree.hpp:21-300, geometry/Structure.hpp:131-265, 811-1090):
  // the containers the reference's transfer functors take (simulation/transfer/P2G.hpp:27-49), used the way those functors use them
  {
    using pars_t = Particles<float, 3>;
    using grids_t = Grids<float, 3, 4>;
    const int n = 4096, nblocks = 8;   // a 2 x 2 x 2 arrangement of 4^3 blocks: block (bx, by, bz) has number (bx * 2 + by) * 2 + bz
    const float dx = 0.125f;
    pars_t pars(n, memsrc_e::um, 0);
    pars.addAttr("m", attrib_e::scalar);
    pars.addAttr("v", attrib_e::vector);
    pars.addAttr("F", attrib_e::matrix);
    pars.addAttr("C", attrib_e::matrix);
    pars.addAttr("logJp", attrib_e::scalar);
    CHECK(pars.size() == (size_t)n && pars.hasAttr("F") && !pars.hasAttr("J") && pars.space() == memsrc_e::um);
    CHECK(pars_t::get_attribute_enum(pars.attr("C")) == attrib_e::matrix && &pars.addAttr("v", attrib_e::vector) == &pars.attr("v"));
    pol(range(n), [pv = proxy<space>(pars), dx] ZS_LAMBDA(long long i) mutable {
      pv.mass(i) = 1.0f + (float)(i % 7);
      // a lattice of 16^3 points, half a cell off the nodes, inside the 8^3-cell box
      pv.pos(i) = pars_t::TV{{((float)(i % 16) * 0.5f + 0.25f) * dx, ((float)((i / 16) % 16) * 0.5f + 0.25f) * dx, ((float)(i / 256) * 0.5f + 0.25f) * dx}};
      pv.vel(i) = pars_t::TV{{1.f, -2.f, 0.5f}};
      auto &F = pv.F(i);
      for (int d = 0; d < 9; ++d) F(d) = (d & 3) ? 0.f : 1.f;
      pv.C(i) = pars_t::TM{};
      pv.logJp(i) = 0.f;
    });
    {
      const zs_rocm_particles ports = pars.ports();  // the same memory as the C ABI sees it: AoS iterator ports
      CHECK(ports.pos.base == (float *)pars.attrVector("x").data() && ports.pos.numChns == 3 && ports.pos.tileMask == 0 && ports.F.numChns == 9 &&
            ports.mass.base == pars.attrScalar("m").data() && ports.n == (size_t)n);
      const auto X = pars.retrievePositions();
      CHECK(X.size() == (size_t)n && std::abs(X[17][0] - (1 * 0.5f + 0.25f) * dx) < 1e-7f && std::abs(X[17][1] - (1 * 0.5f + 0.25f) * dx) < 1e-7f);
    }
    grids_t grids({{"m", 1}, {"v", 3}}, dx, nblocks, memsrc_e::um, 0);
    CHECK(grids.numBlocks() == (size_t)nblocks && grids_t::block_space() == 64 && grids.grid(collocated_c).numChannels() == 4 &&
          grids.numCells() == (size_t)nblocks * 64 * 4 && grids.grid().hasProperty("v") && !grids.grid().hasProperty("rhs"));
    grids.grid(collocated_c).blocks.reset(0);
    using gv_t = GridsView<space, grids_t>;
    {
      constexpr auto c = gv_t::coord_to_cellid(vec<int, 3>{{1, 2, 3}});
      static_assert(c == (1 * 4 + 2) * 4 + 3, "");
      const auto back = gv_t::cellid_to_coord(c);
      CHECK(back[0] == 1 && back[1] == 2 && back[2] == 3 && gv_t::global_coord_to_cellid(vec<int, 3>{{5, -2, 11}}) == (1 * 4 + 2) * 4 + 3);
    }
    // a nearest-node splat written like the reference's P2GTransfer::operator() (P2G.hpp:51-125): particles.pos / mass / vel, the
    // collocated grid of `grids`, a block by number, cell ids from local coordinates
    pol(range(n), [particles = proxy<space>(pars), grids = proxy<space>(grids), dx] ZS_LAMBDA(long long parid) mutable {
      const auto pos = particles.pos(parid);
      const auto vel = particles.vel(parid);
      const float mass = particles.mass(parid);
      vec<int, 3> coord{}, blockid{}, local{};
      for (int d = 0; d < 3; ++d) {
        coord[d] = (int)(pos[d] / dx);
        blockid[d] = coord[d] / gv_t::side_length;
        local[d] = coord[d] % gv_t::side_length;
      }
      const size_t blockno = (size_t)((blockid[0] * 2 + blockid[1]) * 2 + blockid[2]);
      auto grid = grids.grid(collocated_c);
      auto grid_block = grid.block(blockno);
      atomic_add(exec_rocm, &grid_block(0, local), mass);
      for (int d = 0; d < 3; ++d) atomic_add(exec_rocm, &grid("v", blockno, local) + d * gv_t::block_space(), mass * vel[d]);
    });
    pol(Collapse{nblocks, grids_t::block_space()}, [grids = proxy<space>(grids)] ZS_LAMBDA(int bi, int ci) mutable {  // momentum -> velocity
      auto block = grids[bi];
      const float m = block(0, ci);
      if (m > 0.f) {
        auto mv = block.pack<3>(1, ci);
        block.set<3>("v", ci, vec<float, 3>{{mv[0] / m, mv[1] / m, mv[2] / m}});
      }
    });
    double msum = 0, want = 0;
    for (int i = 0; i < n; ++i) want += 1.0 + (i % 7);
    const float *g = grids.grid().data();
    bool vel_ok = true;
    for (int b = 0; b < nblocks; ++b)
      for (int c = 0; c < 64; ++c) {
        const float m = g[((size_t)b * 4 + 0) * 64 + c];
        msum += m;
        CHECK(m > 0.f);  // eight particles per cell
        vel_ok = vel_ok && std::abs(g[((size_t)b * 4 + 1) * 64 + c] - 1.f) < 1e-5f && std::abs(g[((size_t)b * 4 + 2) * 64 + c] + 2.f) < 1e-5f &&
                 std::abs(g[((size_t)b * 4 + 3) * 64 + c] - 0.5f) < 1e-5f;
      }
    CHECK(std::abs(msum - want) < 1e-3 * want && vel_ok);
    // the const views and the host-side container operations
    const pars_t &cpars = pars;
    auto cpv = proxy<space>(cpars);
    CHECK(cpv.size() == (size_t)n && cpv.mass(3) == 4.0f && cpv.F(5)(4) == 1.f);
    pars_t more(16, memsrc_e::um, 0);
    more.addAttr("m", attrib_e::scalar);
    pars.append(more);
    CHECK(pars.size() == (size_t)n + 16 && pars.attrScalar("m").size() == (size_t)n + 16);
    pars.resize(n);
    CHECK(pars.attrMatrix("F").size() == (size_t)n);
    grids.align(grid_e::staggered);
    CHECK(grids.grid(staggered_c).numBlocks() == (size_t)nblocks);
  }
  // ---- reduce over a TileVector channel through the iterator ABI (main.cu:7-32)
  for (int n : {1, 2, 7, 16, 128, 1024, 200000}) {
    TileVector<int, 32> tv({{"a", 3}, {"b", 2}, {"c", 1}}, n, memsrc_e::um);
    std::srand(n);
    const int C = 6, bo = tv.getPropertyOffset("b");
    int mx = std::numeric_limits<int>::lowest(), mn = std::numeric_limits<int>::max();
    long long s = 0;
    for (int i = 0; i < n; ++i)
      for (int c = 0; c < C; ++c) {
        int val = std::rand() % 2001 - 1000;
        tv.data()[((size_t)(i / 32) * C + c) * 32 + i % 32] = val;
        if (c == bo) { mx = std::max(mx, val); mn = std::min(mn, val); s += val; }
      }
    Vector<int> out(1, memsrc_e::um);
    aosoa_iterator_const_int_1 first{tv.data() + bo * 32, 0u, 5u, 31u, (unsigned)C}, last = first;
    last.idx = (unsigned)n;
    aosoa_iterator_int_1 o{out.data(), 0u, 0u, 0u, 1u};
    reduce_max__rocm_int_1(pol.handle(), first, last, o);
    CHECK(out.data()[0] == mx);
    reduce_min__rocm_int_1(pol.handle(), first, last, o);
    CHECK(out.data()[0] == mn);
    reduce_sum__rocm_int_1(pol.handle(), first, last, o);
    CHECK(out.data()[0] == (int)s);
  }
  // ---- primitives on contiguous device ranges through the free functions
  {
    const int n = 300001;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (i * 7919) % 100003 - 50000;
    Vector<int> a(n), b(n);
    Vector<int> out(1, memsrc_e::um);
    (void)hipMemcpy(a.data(), h.data(), n * 4, hipMemcpyHostToDevice);
    reduce(pol, a.data(), a.data() + n, out.data(), 0, plus<int>{});
    CHECK(out.data()[0] == (int)std::accumulate(h.begin(), h.end(), 0ll));
    reduce(pol, a.data(), a.data() + n, out.data(), std::numeric_limits<int>::lowest(), getmax<int>{});
    CHECK(out.data()[0] == *std::max_element(h.begin(), h.end()));
    exclusive_scan(pol, a.data(), a.data() + n, b.data());
    std::vector<int> r(n);
    (void)hipMemcpy(r.data(), b.data(), n * 4, hipMemcpyDeviceToHost);
    int acc = 0;
    for (int i = 0; i < n; ++i) { CHECK(r[i] == acc); acc += h[i]; }
    radix_sort(pol, a.data(), a.data() + n, b.data());
    (void)hipMemcpy(r.data(), b.data(), n * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    CHECK(r == h);
  }
  // ---- merge_sort / merge_sort_pair with user comparators (ExecutionPolicy.cuh:698-752): stable, in place
  {
    const int n = 300007;
    std::vector<int> hk(n), hv(n);
    unsigned s = 12345u;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hk[i] = (int)(s >> 8) % 1000 - 500; hv[i] = i; }
    Vector<int> k(n, memsrc_e::device), v(n, memsrc_e::device);
    (void)hipMemcpy(k.data(), hk.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(v.data(), hv.data(), n * 4, hipMemcpyHostToDevice);
    // order by |key| descending: a comparator no radix sort can express directly
    auto comp = [] ZS_LAMBDA(int a, int b) { return (a < 0 ? -a : a) > (b < 0 ? -b : b); };
    merge_sort_pair(pol, k.data(), v.data(), (std::size_t)n, comp);
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return std::abs(hk[a]) > std::abs(hk[b]); });
    std::vector<int> rk(n), rv(n);
    (void)hipMemcpy(rk.data(), k.data(), n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(rv.data(), v.data(), n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) { CHECK(rv[i] == idx[i]); CHECK(rk[i] == hk[idx[i]]); }
    // indirect sort: permutation ordered by an external key array captured in the comparator (LBvh-style)
    (void)hipMemcpy(k.data(), hk.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(v.data(), hv.data(), n * 4, hipMemcpyHostToDevice);
    merge_sort(pol, v.data(), v.data() + n, [key = k.data()] ZS_LAMBDA(int a, int b) { return key[a] < key[b]; });
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return hk[a] < hk[b]; });
    (void)hipMemcpy(rv.data(), v.data(), n * 4, hipMemcpyDeviceToHost);
    CHECK(rv == idx);
    // struct keys (16 bytes) with the default zs::less through operator<
    struct Key { double d; int tag; int pad;
      __host__ __device__ bool operator<(const Key &o) const { return d < o.d; } };
    std::vector<Key> hs(5000);
    for (int i = 0; i < 5000; ++i) hs[i] = Key{(double)(hk[i] % 17), i, 0};
    Key *ds;
    (void)hipMalloc((void **)&ds, sizeof(Key) * hs.size());
    (void)hipMemcpy(ds, hs.data(), sizeof(Key) * hs.size(), hipMemcpyHostToDevice);
    sort(pol, ds, ds + hs.size());
    std::stable_sort(hs.begin(), hs.end());
    std::vector<Key> rs(hs.size());
    (void)hipMemcpy(rs.data(), ds, sizeof(Key) * hs.size(), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < hs.size(); ++i) CHECK(rs[i].tag == hs[i].tag);
    (void)hipFree(ds);
  }
  // ---- launcher shapes: Collapse{nb, nt}, Collapse{nb, ntiles, tileSize}, shmem-first lambdas
  {
    Vector<int> cnt(4, memsrc_e::um);
    cnt.reset(0);
    pol(Collapse{10, 128}, [c = view<space>(cnt)] ZS_LAMBDA(int block, int thread) { atomic_add(exec_rocm, &c[0], block * 128 + thread); });
    CHECK(cnt.data()[0] == (1280 * 1279) / 2);
    pol(Collapse{5, 4, 16}, [c = view<space>(cnt)] ZS_LAMBDA(int block, int tile, int lane) { atomic_add(exec_rocm, &c[1], block * 64 + tile * 16 + lane); });
    CHECK(cnt.data()[1] == (320 * 319) / 2);
    pol.shmem(256 * sizeof(int));
    pol(Collapse{3, 256}, [c = view<space>(cnt)] ZS_LAMBDA(char *shm, int block, int thread) {
      int *s = (int *)shm;
      s[thread] = thread;
      __syncthreads();
      if (thread == 0) {
        int t = 0;
        for (int i = 0; i < 256; ++i) t += s[i];
        atomic_add(exec_rocm, &c[2], t);
      }
    });
    CHECK(cnt.data()[2] == 3 * (256 * 255) / 2);
    pol.shmem(0);
    pol(range(1000), [c = view<space>(cnt)] ZS_LAMBDA(long long i) { if (i % 3 == 0) atomic_inc(exec_rocm, &c[3]); });
    CHECK(cnt.data()[3] == 334);
  }
  // ---- bht: insert / query inside lambdas (Bht.hpp:490-542, 667-698), set semantics
  {
    const int n = 50000;
    bht<int, 3, int, 16> tab(n);
    Vector<int> ret(n, memsrc_e::um);
    pol(range(n), [tb = view<space>(tab), r = view<space>(ret)] ZS_LAMBDA(long long i) {
      small_vec<int, 3> k{{(int)(i % 37) - 18, (int)((i / 37) % 11), (int)(i % 5)}};
      r[i] = tb.insert(k);
    });
    std::vector<std::array<int, 3>> uniq;
    {
      std::vector<std::array<int, 3>> all;
      for (int i = 0; i < n; ++i) all.push_back({i % 37 - 18, (i / 37) % 11, i % 5});
      std::sort(all.begin(), all.end());
      all.erase(std::unique(all.begin(), all.end()), all.end());
      uniq = all;
    }
    CHECK(tab.size() == uniq.size());
    int winners = 0;
    for (int i = 0; i < n; ++i) winners += ret.data()[i] >= 0;
    CHECK(winners == (int)uniq.size());
    pol(range(n), [tb = view<space>(tab), r = view<space>(ret)] ZS_LAMBDA(long long i) {
      small_vec<int, 3> k{{(int)(i % 37) - 18, (int)((i / 37) % 11), (int)(i % 5)}};
      small_vec<int, 3> miss{{1000 + (int)i, 0, 0}};
      r[i] = (tb.query(k) >= 0 && tb.query(miss) == -1) ? 1 : 0;
    });
    for (int i = 0; i < n; ++i) CHECK(ret.data()[i] == 1);
  }
  // ---- HashTable: insert / query inside lambdas (HashTable.hpp:353-470), resize keeps indices
  {
    const int n = 40000;
    HashTable<3> tab(n);
    Vector<int> ret(n, memsrc_e::um);
    pol(range(n), [tb = proxy<space>(tab), r = view<space>(ret)] ZS_LAMBDA(long long i) {
      small_vec<int, 3> k{{(int)(i % 41) - 20, (int)((i / 41) % 13), (int)(i % 7)}};
      r[i] = tb.insert(k);
    });
    std::vector<std::array<int, 3>> all;
    for (int i = 0; i < n; ++i) all.push_back({i % 41 - 20, (i / 41) % 13, i % 7});
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    CHECK(tab.size() == (int)all.size());
    int winners = 0;
    for (int i = 0; i < n; ++i) winners += ret.data()[i] >= 0;
    CHECK(winners == (int)all.size());
    tab.resize(pol, 4 * n);
    pol(range(n), [tb = proxy<space>(tab), r = view<space>(ret)] ZS_LAMBDA(long long i) {
      small_vec<int, 3> k{{(int)(i % 41) - 20, (int)((i / 41) % 13), (int)(i % 7)}};
      small_vec<int, 3> miss{{1000 + (int)i, 0, 0}};
      const int q = tb.query(k);
      r[i] = (q >= 0 && q < tb.size() && tb.query(miss) == -1 && !tb.insert(k, 5)) ? 1 : 0;
    });
    for (int i = 0; i < n; ++i) CHECK(ret.data()[i] == 1);
  }
  // ---- TileVector::reorderTiles: gather and scatter of whole tiles by an index map (TileVector.hpp:641-691)
  {
    TileVector<float, 32> tvr(std::vector<PropertyTag>{{"a", 2}, {"b", 1}}, 32 * 7 - 5, memsrc_e::um);
    const int nt = (int)tvr.numTiles();
    CHECK(nt == 7);
    auto hv = view<space>(tvr);
    for (int i = 0; i < 32 * 7 - 5; ++i)
      for (int c = 0; c < 3; ++c) hv(c, (std::size_t)i) = (float)(1000 * c + i);
    Vector<int> map(nt, memsrc_e::um);
    const int perm[7] = {3, 0, 6, 1, 5, 2, 4};
    for (int i = 0; i < nt; ++i) map.data()[i] = perm[i];
    tvr.reorderTiles(pol, map);  // gather: new tile i = old tile perm[i]
    hv = view<space>(tvr);
    for (int t = 0; t < nt; ++t)
      for (int l = 0; l < 32; ++l)
        if (perm[t] * 32 + l < 32 * 7 - 5)
          for (int c = 0; c < 3; ++c) CHECK(hv(c, (std::size_t)t, l) == (float)(1000 * c + perm[t] * 32 + l));
    tvr.reorderTiles(pol, map, wrapv<true>{});  // scatter by the same map undoes the gather
    hv = view<space>(tvr);
    for (int i = 0; i < 32 * 7 - 5; ++i)
      for (int c = 0; c < 3; ++c) CHECK(hv(c, (std::size_t)i) == (float)(1000 * c + i));
  }
  // ---- SparseGrid<3, f32, 8>: world-space insert / query, decomposeCoord, valueOr, trilinear wSample of a linear field
  {
    const float dx = 0.125f;
    SparseGrid<3, float, 8> sg(std::vector<PropertyTag>{{"sdf", 1}, {"v", 3}}, 64);
    sg.scale(dx);
    sg.translate(-1.f, -1.f, -1.f);
    sg._background = 7.f;
    const int np = 5000;
    Vector<float> px(3 * np, memsrc_e::um);
    unsigned s = 99u;
    for (int i = 0; i < 3 * np; ++i) { s = s * 1664525u + 1013904223u; px.data()[i] = -0.5f + 1.0f * (float)(s >> 8) / (float)(1u << 24); }
    pol(range(np), [g = view<space>(sg), p = view<space>(px)] ZS_LAMBDA(long long i) {
      // a 2x2x2-cell neighbourhood of every sample so that the linear stencil of the point is allocated
      for (int o = 0; o < 8; ++o)
        g.insert(small_vec<float, 3>{{p[3 * i] + ((o >> 2) - 0.5f) * 0.125f, p[3 * i + 1] + (((o >> 1) & 1) - 0.5f) * 0.125f, p[3 * i + 2] + ((o & 1) - 0.5f) * 0.125f}});
    });
    const std::size_t nb = sg.numBlocks();
    CHECK(nb > 0 && nb <= 64);
    // fill: sdf = 2x - y + 0.5z + 1 at every node of every active block, v = world coordinate
    pol(range((long long)nb * 512), [g = view<space>(sg)] ZS_LAMBDA(long long c) {
      const int b = (int)(c / 512), k = (int)(c % 512);
      const auto w = g.wCoord(b, k);
      g(0, b, k) = 2.f * w[0] - w[1] + 0.5f * w[2] + 1.f;
      for (int d = 0; d < 3; ++d) g(1 + d, b, k) = w[d];
      const auto ic = g.iCoord(b, k);
      const auto bc = g.decomposeCoord(ic);  // round trip through the table
      if (bc.bno != b || bc.cno != k) g(0, b, k) = -1e30f;
    });
    Vector<float> err(np, memsrc_e::um);
    pol(range(np), [g = view<space>(sg), p = view<space>(px), e = view<space>(err)] ZS_LAMBDA(long long i) {
      const small_vec<float, 3> x{{p[3 * i], p[3 * i + 1], p[3 * i + 2]}};
      const float f = g.wSample(0, x), ref = 2.f * x[0] - x[1] + 0.5f * x[2] + 1.f;
      const auto v = g.wPack(dim_c<3>, 1, x);
      float m = fabsf(f - ref);
      for (int d = 0; d < 3; ++d) m = fmaxf(m, fabsf(v[d] - x[d]));
      if (g.query(x) < 0) m = 1e30f;
      // far outside: background through valueOr / sample
      if (fabsf(g.wSample(0, small_vec<float, 3>{{50.f, 50.f, 50.f}}) - 7.f) > 0.f) m = 1e30f;
      e[i] = m;
    });
    float worst = 0.f;
    for (int i = 0; i < np; ++i) worst = std::max(worst, err.data()[i]);
    CHECK(worst < 2e-5f);  // trilinear interpolation reproduces linear fields
    // the same grid through the other kernels of GridArena: every B-spline reproduces linear fields, the quadratic and cubic ones with
    // exact gradients (sum_i grad w_i = 0, sum_i grad w_i * f_i = grad f); the delta kernels are partitions of unity; name-keyed access
    pol(range(np), [g = view<space>(sg), p = view<space>(px), e = view<space>(err)] ZS_LAMBDA(long long i) {
      using V3 = small_vec<float, 3>;
      const V3 x{{p[3 * i] * 0.5f, p[3 * i + 1] * 0.5f, p[3 * i + 2] * 0.5f}};  // keep the wider stencils inside the allocated blocks
      const float ref = 2.f * x[0] - x[1] + 0.5f * x[2] + 1.f;
      float m = 0.f;
      auto aq = g.wArena<kernel_e::quadratic, 1>(x);
      auto ac = g.wArena<kernel_e::cubic, 2>(x);
      bool covered = true;  // all nodes of the widest stencil exist
      for (auto loc : ac.range()) covered = covered && g.decomposeCoord(ac.coord(loc)).bno >= 0;
      if (covered) {
        m = fmaxf(m, fabsf(aq.isample(0, 7.f) - ref));
        m = fmaxf(m, fabsf(ac.isample("sdf", 0, 7.f) - ref));
        m = fmaxf(m, fabsf(g.wSample(0, x, kernel_c<kernel_e::quadratic>) - ref));
        float sw = 0.f, sw3 = 0.f, sw4 = 0.f, gsum[3] = {0.f, 0.f, 0.f}, gf[3] = {0.f, 0.f, 0.f}, lap = 0.f;
        for (auto loc : aq.range()) {
          sw += aq.weight(loc);
          const V3 gw = aq.weightsGradient(loc);
          const float f = aq.val(0, loc);
          for (int d = 0; d < 3; ++d) { gsum[d] += gw[d]; gf[d] += gw[d] * f; }
        }
        for (auto loc : ac.range()) lap += ac.w[2][0][loc[0]] * ac.w[0][1][loc[1]] * ac.w[0][2][loc[2]] * ac.val(0, loc);  // d2/dX2 of a linear field
        m = fmaxf(m, fabsf(sw - 1.f));
        const float want[3] = {2.f, -1.f, 0.5f};  // d sdf / d world; the arena's gradients are per index unit: * dx
        for (int d = 0; d < 3; ++d) m = fmaxf(m, fmaxf(fabsf(gsum[d]), fabsf(gf[d] / 0.125f - want[d]) * 0.02f));
        m = fmaxf(m, fabsf(lap) * 1e-2f);
        for (auto loc : g.wArena<kernel_e::delta3>(x).range()) sw3 += g.wArena<kernel_e::delta3>(x).weight(loc);
        for (auto loc : g.wArena<kernel_e::delta4>(x).range()) sw4 += g.wArena<kernel_e::delta4>(x).weight(loc);
        m = fmaxf(m, fmaxf(fabsf(sw3 - 1.f), fabsf(sw4 - 1.f)));
        const float lo = aq.minimum(0), hi = aq.maximum(0);
        if (!(lo <= ref + 1e-5f && ref <= hi + 1e-5f)) m = 1e30f;
        // staggered: "v" holds the world coordinate of the NODE; component f sampled as a face value sits half a cell lower along f
        const V3 vs = g.wStaggeredPack(1, x);
        for (int d = 0; d < 3; ++d) m = fmaxf(m, fabsf(vs[d] - (x[d] + 0.5f * 0.125f)));
        if (g.propertyOffset("v") != 1 || g.propertyOffset("nope") != -1 || !g.hasProperty("sdf")) m = 1e30f;
      }
      e[i] = covered ? m : -1.f;
    });
    worst = 0.f;
    int ncov = 0;
    for (int i = 0; i < np; ++i) { worst = std::max(worst, err.data()[i]); ncov += err.data()[i] >= 0.f; }
    CHECK(ncov > np / 2);
    CHECK(worst < 5e-5f);
  }
  // ---- SparseGrid<2, f64, 8> (64-cell blocks) and SparseGrid<1, f32, 32>: the generic head; cell <-> coordinate maps, staggered cell
  //      averages of a MAC field, cubic sampling of a linear field in double precision
  {
    SparseGrid<2, double, 8> g2(std::vector<PropertyTag>{{"u", 2}, {"p", 1}}, 32);
    g2.scale(0.25);
    g2.translate(-2.0, 1.0);
    static_assert(SparseGrid<2, double, 8>::block_size == 64 && SparseGrid<1, float, 32>::block_size == 32 && SparseGrid<3, float, 4>::block_size == 64, "");
    Vector<double> e2(1, memsrc_e::um);
    e2.reset(0);
    pol(range(16 * 16), [g = view<space>(g2)] ZS_LAMBDA(long long i) {
      g.insert(small_vec<double, 2>{{-2.0 + 0.25 * (double)(i / 16) * 2.0, 1.0 + 0.25 * (double)(i % 16) * 2.0}});  // 32 x 32 cells
    });
    const std::size_t nb2 = g2.numBlocks();
    CHECK(nb2 == 16);
    pol(range((long long)nb2 * 64), [g = view<space>(g2)] ZS_LAMBDA(long long c) {
      const int b = (int)(c / 64), k = (int)(c % 64);
      const auto ic = g.iCoord(b, k);
      const auto w = g.wCoord(b, k);
      // MAC field u = (3 x - y, x + 2 y) stored at the face centres, p = 1 - x + 4 y at the nodes
      const auto f0 = g.wStaggeredCoord(b, k, 0), f1 = g.wStaggeredCoord(b, k, 1);
      g(0, b, k) = 3.0 * f0[0] - f0[1];
      g(1, b, k) = f1[0] + 2.0 * f1[1];
      g("p", 0, b, k) = 1.0 - w[0] + 4.0 * w[1];
      const auto bc = g.decomposeCoord(ic);
      if (bc.bno != b || bc.cno != k || g.local_coord_to_offset(g.local_offset_to_coord(k)) != k) g(2, b, k) = 1e300;
    });
    pol(range(2000), [g = view<space>(g2), e = view<space>(e2)] ZS_LAMBDA(long long i) {
      unsigned s = 7u + (unsigned)i * 2654435761u;
      auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (double)(1u << 24); };
      const small_vec<double, 2> x{{-2.0 + 1.0 + 5.5 * rnd(), 1.0 + 1.0 + 5.5 * rnd()}};  // one cell away from the border at least
      double m = fabs(g.wSample("p", 0, x, kernel_c<kernel_e::cubic>) - (1.0 - x[0] + 4.0 * x[1]));
      const auto u = g.wStaggeredPack(0, x);
      m = fmax(m, fabs(u[0] - (3.0 * x[0] - x[1])));
      m = fmax(m, fabs(u[1] - (x[0] + 2.0 * x[1])));
      // component 1 of the MAC field at the centre of face 0 of the cell that holds x: mean of the four surrounding 1-faces
      const auto X = g.worldToIndex(x);
      const small_vec<int, 2> c{{(int)floor(X[0]), (int)floor(X[1])}};
      const auto fc = g.indexToWorld(small_vec<double, 2>{{(double)c[0] - 0.5, (double)c[1]}});
      m = fmax(m, fabs(g.iStaggeredCellSample(0, 1, c, 0) - (fc[0] + 2.0 * fc[1])));
      m = fmax(m, fabs(g.iStaggeredCellPack(0, c, 0)[0] - (3.0 * fc[0] - fc[1])));
      // orientation >= dim reads the opposite face = the same face of the next cell
      m = fmax(m, fabs(g.valueOr(true_c, 0, c, 2, -1.0) - g.valueOr(false_c, 0, small_vec<int, 2>{{c[0] + 1, c[1]}}, -1.0)));
      atomicMax((unsigned long long *)&e[0], (unsigned long long)__double_as_longlong(m));  // non-negative doubles order like integers
    });
    CHECK(e2.data()[0] < 1e-12);
    SparseGrid<1, float, 32> g1(1, 8);
    pol(range(4), [g = view<space>(g1)] ZS_LAMBDA(long long i) { g.insert(small_vec<float, 1>{{(float)(32 * i)}}); });
    CHECK(g1.numBlocks() == 4);
  }
  // ---- LBvh: build, iter_neighbors / self_iter_neighbors inside lambdas (Bvh.hpp:644-728) vs brute force
  {
    const int n = 6000;
    std::vector<AABBBox3f> hb(n);
    unsigned s = 4242u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / (float)(1u << 24); };
    for (auto &b : hb) {
      const float c[3] = {rnd(), rnd(), rnd()}, e = 0.004f + 0.02f * rnd();
      for (int d = 0; d < 3; ++d) { b.lo[d] = c[d] - e; b.hi[d] = c[d] + e; }
    }
    Vector<AABBBox3f> bvs(n, memsrc_e::device);
    (void)hipMemcpy(bvs.data(), hb.data(), sizeof(AABBBox3f) * n, hipMemcpyHostToDevice);
    LBvh bvh;
    bvh.build(pol, bvs);
    CHECK(bvh.getNumLeaves() == (std::size_t)n && bvh.getNumNodes() == (std::size_t)(2 * n - 1));
    Vector<int> cnt(n, memsrc_e::um), selfcnt(1, memsrc_e::um);
    selfcnt.reset(0);
    pol(range(n), [bv = view<space>(bvh), q = view<space>(bvs), c = view<space>(cnt), sc = view<space>(selfcnt)] ZS_LAMBDA(long long i) {
      int k = 0;
      bv.iter_neighbors(q[i], [&](int) { ++k; });
      c[i] = k;
      int pairs = 0;
      bv.self_iter_neighbors((int)i, [&](int) { ++pairs; });
      atomic_add(exec_rocm, &sc[0], pairs);
    });
    long long total = 0;
    for (int i = 0; i < n; ++i) total += cnt.data()[i];
    long long brute = 0;
    for (int i = 0; i < n; i += 40) {  // spot-check rows exactly
      int k = 0;
      for (int j = 0; j < n; ++j) k += zsr::aabb_overlaps(hb[j], hb[i]);
      CHECK(k == cnt.data()[i]);
    }
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) brute += zsr::aabb_overlaps(hb[j], hb[i]);
    CHECK(2 * brute - n == total);                 // ordered pairs = 2 * unordered (incl. self) - self
    CHECK(selfcnt.data()[0] == (int)brute);        // self iteration reports every unordered pair (and self) once
    // ray_intersect (Bvh.hpp:521-545) and find_nearest_point (:622-661) vs brute force over the leaves
    const int nq = 512;
    Vector<int> hits(nq, memsrc_e::um), nearest(nq, memsrc_e::um);
    pol(range(nq), [bv = view<space>(bvh), h = view<space>(hits), nn = view<space>(nearest)] ZS_LAMBDA(long long i) {
      const float t = (float)i / 512.f;
      const float ro[3] = {-0.1f, t, 0.5f * t + 0.1f}, rd[3] = {1.f, 0.3f - 0.6f * t, 0.2f};
      int k = 0;
      bv.ray_intersect(ro, rd, [&](int) { ++k; });
      h[i] = k;
      const float p[3] = {t, 1.f - t, 0.5f};
      int best = -1;
      bv.find_nearest_point(p, 3.402823466e+38f, &best);
      nn[i] = best;
    });
    for (int i = 0; i < nq; i += 7) {
      const float t = (float)i / 512.f;
      const float ro[3] = {-0.1f, t, 0.5f * t + 0.1f}, rd[3] = {1.f, 0.3f - 0.6f * t, 0.2f}, p[3] = {t, 1.f - t, 0.5f};
      int k = 0, best = -1;
      float bd = 3.4e38f;
      for (int j = 0; j < n; ++j) {
        k += LBvhView::ray_box_intersect(ro, rd, hb[j]);
        const float x = p[0] - hb[j].lo[0], y = p[1] - hb[j].lo[1], z = p[2] - hb[j].lo[2], d2 = x * x + y * y + z * z;
        if (d2 < bd) { bd = d2; best = j; }
      }
      CHECK(k == hits.data()[i]);
      const int g = nearest.data()[i];
      const float gx = p[0] - hb[g].lo[0], gy = p[1] - hb[g].lo[1], gz = p[2] - hb[g].lo[2];
      CHECK(g >= 0 && gx * gx + gy * gy + gz * gz <= bd * (1.f + 1e-6f));
    }
  }
  {  // Collider<AnalyticLevelSet<Sphere>> (slip, moving) inside a user lambda == the same struct evaluated on the host
    zs_rocm_collider c;
    const float par[4] = {0.1f, -0.05f, 0.02f, 0.45f};
    zs_rocm_collider_init(&c, ZS_ROCM_GEOM_SPHERE, ZS_ROCM_COLLIDER_SLIP, par, 4);
    c.dbdt[0] = 0.3f; c.omega[2] = 0.7f; c.b[1] = 0.05f;
    const zsr::ColliderDev col(c);
    const int nq = 4096;
    Vector<float> xs(3 * nq, memsrc_e::um), vs(3 * nq, memsrc_e::um);
    Vector<int> ins(nq, memsrc_e::um);
    for (int i = 0; i < nq; ++i)
      for (int d = 0; d < 3; ++d) {
        xs.data()[3 * i + d] = -0.8f + 1.6f * (float)((i * 37 + d * 101 + (i / 7) * 13) % 997) / 997.f;
        vs.data()[3 * i + d] = -1.f + 2.f * (float)((i * 53 + d * 211) % 499) / 499.f;
      }
    std::vector<float> v0(vs.data(), vs.data() + 3 * nq);
    pol(range(nq), [col, x = view<space>(xs), v = view<space>(vs), in = view<space>(ins)] ZS_LAMBDA(long long i) {
      const float p[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
      float u[3] = {v[3 * i], v[3 * i + 1], v[3 * i + 2]};
      in[i] = col.resolveCollision(p, u) ? 1 : 0;
      v[3 * i] = u[0]; v[3 * i + 1] = u[1]; v[3 * i + 2] = u[2];
    });
    int inside = 0;
    for (int i = 0; i < nq; ++i) {
      const float p[3] = {xs.data()[3 * i], xs.data()[3 * i + 1], xs.data()[3 * i + 2]};
      float u[3] = {v0[3 * i], v0[3 * i + 1], v0[3 * i + 2]};
      float xmb[3], X[3];
      col.to_material(p, xmb, X);
      if (fabsf(col.signed_distance(X)) < 1e-4f) continue;  // on the surface the two roundings may disagree
      const bool h = col.resolveCollision(p, u);
      CHECK((int)h == ins.data()[i]);
      inside += h;
      for (int d = 0; d < 3; ++d) CHECK(fabsf(u[d] - vs.data()[3 * i + d]) < 1e-5f);
    }
    CHECK(inside > 200 && inside < nq - 200);
  }
  CHECK(zs_rocm_last_error(-1) == 0);
  // ---- Appendix-B odds and ends: append_channels / reset(pol, val) / clone, Vector bulk copies, for_each, par_exec, make_monoid
  {
    const int n = 1000;
    TileVector<float, 32> tv({{"m", 1}, {"x", 3}}, n, memsrc_e::um);
    tv.reset(pol, 2.5f);
    CHECK(tv.data()[0] == 2.5f && tv.data()[tv.tiles() * 32 * 4 - 1] == 2.5f);
    pol(range(n), [t = view<space>(tv)] ZS_LAMBDA(long long i) { t(1, i) = (float)i; });
    tv.append_channels(pol, {{"x", 3}, {"v", 3}, {"J", 1}});  // "x" exists with the same width; "v" and "J" are new, zero-filled
    CHECK(tv.numChannels() == 8 && tv.getPropertyOffset("v") == 4 && tv.getPropertyOffset("J") == 7 && tv.size() == (size_t)n);
    pol.syncCtx();
    for (int i = 0; i < n; ++i) {
      const float *b = tv.data() + ((size_t)(i / 32) * 8) * 32 + i % 32;
      CHECK(b[0] == 2.5f && b[32] == (float)i && b[4 * 32] == 0.f && b[7 * 32] == 0.f);
    }
    bool threw = false;
    try { tv.append_channels(pol, {{"x", 2}}); } catch (const std::runtime_error &) { threw = true; }
    CHECK(threw);
    auto h = tv.clone(memsrc_e::host);
    CHECK(h.data()[32 + 5] == 5.f && h.numChannels() == 8);
    Vector<int> a(4), b(3);
    const int av[4] = {1, 2, 3, 4}, bv[3] = {7, 8, 9};
    a.assignVals(av);
    b.assignVals(bv);
    a.append(b);
    a.push_back(11);
    int out[8];
    a.retrieveVals(out);
    CHECK(a.size() == 8 && out[0] == 1 && out[3] == 4 && out[4] == 7 && out[6] == 9 && out[7] == 11);
    Vector<int> cnt(1);
    cnt.setVal(0);
    for_each(par_exec(rocm_c), range(1000), [c = view<space>(cnt)] ZS_LAMBDA(long long i) { atomic_add(exec_rocm, &c[0], (int)(i & 1)); });
    CHECK(cnt.getVal() == 500);
    CHECK(make_monoid(plus<int>{}).identity() == 0 && make_monoid(multiplies<float>{}).identity() == 1.f);
    CHECK(make_monoid(getmin<int>{}).identity() == 2147483647 && make_monoid(getmax<int>{}).identity() == (-2147483647 - 1));
    CHECK(make_monoid(getmax<int>{})(3, 9) == 9);
    CHECK(valid_memspace_for_execution(pol, memsrc_e::device) && !valid_memspace_for_execution(pol, memsrc_e::host));
    CHECK(pol.getProcid() == -1);
    {  // ndrange<3>(3): 27 stencil offsets, first index slowest, on the host and inside a kernel
      int k = 0, okh = 1;
      for (auto loc : ndrange<3>(3)) { okh &= (get<0>(loc) == k / 9 && loc[1] == (k / 3) % 3 && get<2>(loc) == k % 3); ++k; }
      CHECK(k == 27 && okh);
      Vector<int> s3(1);
      s3.setVal(0);
      pol(range(64), [s = view<space>(s3)] ZS_LAMBDA(long long) {
        int acc = 0;
        for (auto loc : ndrange<3>(3)) acc += get<0>(loc) * 9 + get<1>(loc) * 3 + get<2>(loc);
        atomic_add(exec_rocm, &s[0], acc);
      });
      CHECK(s3.getVal() == 64 * 351);
    }
    // two live temporaries + a multi-block reduce over one of them: the blocks must not alias each other or the reduce's own scratch
    auto tms = get_temporary_memory_source(pol);
    const long long nT = 1 << 20;
    int *scratch = (int *)tms.allocate(nT * sizeof(int));
    int *scratch2 = (int *)tms.allocate(nT * sizeof(int));
    CHECK(scratch != nullptr && scratch2 != nullptr && scratch != scratch2);
    pol(range(nT), [scratch, scratch2] ZS_LAMBDA(long long i) { scratch[i] = (int)(i & 1023); scratch2[i] = -1; });
    Vector<int> total(2);
    reduce(pol, (const int *)scratch, (const int *)scratch + nT, total.data(), 0, plus<int>{});
    reduce(pol, (const int *)scratch2, (const int *)scratch2 + nT, total.data() + 1, 0, plus<int>{});
    CHECK(total.getVal(0) == (int)((nT / 1024) * (1023 * 1024 / 2)));
    CHECK(total.getVal(1) == -(int)nT);
    tms.deallocate(scratch, nT * sizeof(int));
    tms.deallocate(scratch2, nT * sizeof(int));
    // tile_insert / tile_query: 16-lane tiles, every lane of a tile carries the tile's key
    bht<int, 3, int, 16> tb(4096);
    Vector<int> bad(1);
    bad.setVal(0);
    pol(range(64 * 16), [t = view<space>(tb), b = view<space>(bad)] ZS_LAMBDA(long long i) {
      auto tile = cooperative_groups::tiled_partition<16>(cooperative_groups::this_thread_block());
      const int g = (int)(i / 16);
      small_vec<int, 3> key{{g % 4, (g / 4) % 4, g / 16}};
      const int no = t.tile_insert(tile, key);
      if (no < 0 || no >= 64) atomic_add(exec_rocm, &b[0], 1);
    });
    CHECK(tb.size() == 64 && bad.getVal() == 0);
    pol(range(64 * 16), [t = view<space>(tb), b = view<space>(bad)] ZS_LAMBDA(long long i) {
      auto tile = cooperative_groups::tiled_partition<16>(cooperative_groups::this_thread_block());
      const int g = (int)(i / 16);
      small_vec<int, 3> key{{g % 4, (g / 4) % 4, g / 16}}, absent{{9, g, 9}};
      if (t.tile_query(tile, key) < 0 || t.tile_query(tile, absent) != -1) atomic_add(exec_rocm, &b[0], 1);
    });
    CHECK(bad.getVal() == 0);
  }
  {  // ---- the reference's reduction test over range(tv, "b") (test/parallel_primitives.cpp:9-33)
    for (size_t n : {(size_t)1, (size_t)2, (size_t)7, (size_t)16, (size_t)128, (size_t)1024, (size_t)200000}) {
      auto vals = gen_rnd_tv_ints(n, 0);
      CHECK(test_reduction(pol, range(vals, "b"), getmax<int>()));
      CHECK(test_reduction(pol, range(vals, "b"), getmin<int>()));
      vals = gen_rnd_tv_ints(n, 100);
      CHECK(test_reduction(pol, range(vals, "b"), plus<int>()));
    }
  }
  {  // ---- range launches: arity dispatch of range_launch / range_launch_with_params (ExecutionPolicy.cuh:245-322, 40-155)
    const size_t n = 5000;
    TileVector<float, 64> tv({{"x", 3}, {"m", 1}}, n, memsrc_e::device);
    Vector<float> a(n), b(n);
    Vector<int> cnt(1);
    cnt.setVal(0);
    pol(enumerate(a, b), [] ZS_LAMBDA(long long i, float &x, float &y) { x = (float)i; y = 2.f * (float)i; });          // (i, refs...)
    pol(zip(a, b), [] ZS_LAMBDA(float &x, const float &y) { x += y; });                                                   // (refs...)
    pol(range(tv, "m"), [] ZS_LAMBDA(float &m) { m = 1.5f; });                                                            // single range
    pol(zip(range(tv, "m"), a), std::make_tuple(0.5f, 3), [] ZS_LAMBDA(float &m, const float &x, const std::tuple<float, int> &p) {
      m = m * std::get<0>(p) + x * (float)std::get<1>(p);                                                                 // (refs..., params)
    });
    pol.shmem(256 * sizeof(float));
    pol(enumerate(a), [c = view<space>(cnt)] ZS_LAMBDA(float *shm, long long i, float &x) {                               // (shmem*, i, refs...)
      shm[threadIdx.x] = x;
      __syncthreads();
      if (shm[threadIdx.x] != 3.f * (float)i) atomic_add(exec_rocm, &c[0], 1);
    });
    pol.shmem(0);
    CHECK(cnt.getVal() == 0);
    auto mh = tv.clone(memsrc_e::um);
    auto mv = view<space>(mh);
    bool ok = true;
    for (size_t i = 0; i < n; i += 97) ok = ok && mv(3, i) == 0.75f + 9.f * (float)i;
    CHECK(ok);
    // named view: tv("name", d, i), pack / set by name (TileVector.hpp:1150-1540)
    pol(range(n), [t = proxy<space>({"x", "m"}, tv)] ZS_LAMBDA(long long i) {
      t("x", 0, (size_t)i) = t("m", (size_t)i);
      t("x", 1, (size_t)i) = 2.f;
      t.set("x", (size_t)i, small_vec<float, 3>{{t("x", 0, (size_t)i), 4.f, (float)t.hasProperty("nope")}});
    });
    auto th = tv.clone(memsrc_e::um);
    auto tvw = view<space>(th);
    CHECK(tvw(0, 4850) == 0.75f + 9.f * 4850.f && tvw(1, 4850) == 4.f && tvw(2, 4850) == 0.f);
    // generic-iterator scans: a TileVector channel as input, a Vector as output
    TileVector<int, 32> ti({{"k", 1}, {"pad", 2}}, 3000, memsrc_e::um);
    auto tiv = view<space>(ti);
    for (size_t i = 0; i < 3000; ++i) tiv(0, i) = (int)(i % 7);
    Vector<int> sc(3000, memsrc_e::um);
    exclusive_scan(pol, ti.begin("k"), ti.end("k"), sc.begin(), 0, plus<int>{});
    int run = 0;
    bool sok = true;
    for (size_t i = 0; i < 3000; ++i) { sok = sok && sc[i] == run; run += (int)(i % 7); }
    CHECK(sok);
    // iterator-taking radix sorts (execution/ExecutionPolicy.hpp:765-781): a TileVector channel in, a Vector out; pairs with an index channel
    for (size_t i = 0; i < 3000; ++i) tiv(0, i) = (int)((i * 2654435761u) % 100003u) - 50000, tiv(1, i) = (int)i;
    Vector<int> so(3000, memsrc_e::um), vo(3000, memsrc_e::um);
    radix_sort(pol, ti.begin("k"), ti.end("k"), so.begin());
    bool rok = true;
    for (size_t i = 1; i < 3000; ++i) rok = rok && so[i - 1] <= so[i];
    CHECK(rok);
    radix_sort_pair(pol, ti.begin("k"), ti.begin("pad"), so.begin(), vo.begin(), 3000);
    for (size_t i = 0; i < 3000; ++i) rok = rok && so[i] == tiv(0, (size_t)vo[i]) && (i == 0 || so[i - 1] <= so[i]);
    CHECK(rok);
    // zs::tuple / zs::make_tuple / zs::get as the parameter pack of pol(range, params, f) (cuda/execution/ExecutionPolicy.cuh:281-322)
    {
      constexpr auto tp = zs::make_tuple(2, 0.25f, 'c');
      static_assert(zs::tuple_size_v<std::decay_t<decltype(tp)>> == 3 && std::is_same_v<zs::tuple_element_t<1, std::decay_t<decltype(tp)>>, float>);
      static_assert(zs::get<0>(tp) == 2 && tp.get<2>() == 'c' && std::is_trivially_copyable_v<std::decay_t<decltype(tp)>>);
      auto [ta, tb, tc] = tp;
      CHECK(ta == 2 && tb == 0.25f && tc == 'c');
      Vector<float> pa(n), pb(n);
      pol(enumerate(pa, pb), [] ZS_LAMBDA(long long i, float &x, float &y) { x = (float)i; y = 1.f; });
      pol(zip(pa, pb), zs::make_tuple(0.5f, 3), [] ZS_LAMBDA(float &x, const float &y, const zs::tuple<float, int> &p) {
        x = x * zs::get<0>(p) + y * (float)zs::get<1>(p);
      });
      pol(enumerate(pa), zs::make_tuple(7.f), [] ZS_LAMBDA(long long i, float &x, zs::tuple<float> p) { x += p.get<0>() * (float)(i & 1); });
      auto ph = pa.clone(memsrc_e::um);
      bool pok = true;
      for (size_t i = 0; i < n; i += 101) pok = pok && ph[i] == 0.5f * (float)i + 3.f + 7.f * (float)(i & 1);
      CHECK(pok);
    }
    // Vector{allocator, n} / get_memory_source / MemoryLocation
    Vector<double> vd{get_memory_source(memsrc_e::um, 0), 16};
    CHECK(vd.size() == 16 && vd.memspace() == memsrc_e::um);
    bht<int, 2, int, 32> t2{get_memory_source(memsrc_e::device, 0), 100};
    CHECK(t2.size() == 0);
  }
  std::printf("cpp face ok\n");
  return 0;
}
