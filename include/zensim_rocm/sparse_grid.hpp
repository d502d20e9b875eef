// sparse_grid.hpp -- zs::SparseGrid<dim, T, Side>, its view and GridArena for gfx950 (part of zs_rocm.hpp; do not include alone).
//
// What the reference offers (geometry/SparseGrid.hpp:16-188 the container, :199-916 the view; math/curve/InterpolationKernel.hpp:
// 271-560 GridArena): a bht<int, dim, int, 16> keyed by block ORIGIN coordinates (multiples of Side) next to a TileVector<T, Side^dim>
// of cell values, an index<->world transform, collocated and staggered (face-centred) sampling through a small per-query stencil
// object -- the arena -- whose width and weights follow an interpolation kernel (linear / quadratic / cubic B-spline with 0-2
// derivative orders, or the 2/3/4-point discrete delta kernels of the immersed-boundary literature).
//
// Here: dim 1-3, T float or double, any power-of-two Side whose Side^dim is a lane width TileVector supports.  The transform is
// uniform scale + translation (what scale(dx) / translate(t) build, SparseGrid.hpp:170-182).  Everything a kernel calls is a member of
// the POD view; the arena keeps per-axis weight rows and evaluates products on the fly (dim * width floats instead of width^dim).
#pragma once

namespace zs {

enum struct kernel_e : int { linear = 2, quadratic = 3, cubic = 4, delta2 = 5, delta3 = 6, delta4 = 7 };  // types/Property.h
template <kernel_e k> using kernel_t = wrapv<k>;
template <kernel_e k> constexpr kernel_t<k> kernel_c{};
struct false_type_t {};
struct true_type_t {};
constexpr false_type_t false_c{};  // index-space argument
constexpr true_type_t true_c{};    // world-space argument

namespace detail {
  constexpr int ipow(int b, int e) { return e == 0 ? 1 : b * ipow(b, e - 1); }
  ZS_FUNCTION constexpr int kernel_width(kernel_e kt) {
    return (kt == kernel_e::linear || kt == kernel_e::delta2) ? 2 : ((kt == kernel_e::quadratic || kt == kernel_e::delta3) ? 3 : 4);
  }
  // base_node<degree>(x) = floor(x - degree / 2) (InterpolationKernel.hpp:46-55): lowest node of the stencil around x
  template <class T> ZS_FUNCTION int stencil_base(T x, int degree) { return (int)floor(x - (T)0.5 * (T)degree); }

  // one axis of weights for local position t measured from the stencil's lowest node:
  //   linear t in [0,1), quadratic t in [0.5,1.5), cubic t in [1,2)        (InterpolationKernel.hpp:57-180)
  //   delta kernels: r = |t - node|                                         (:181-268)
  // order 0 = value, 1 = first derivative, 2 = second derivative (B-splines only), all in index-space units
  template <kernel_e kt, int order, class T> ZS_FUNCTION void axis_weights(T t, T *w) {
    if constexpr (kt == kernel_e::linear) {
      if constexpr (order == 0) { w[0] = (T)1 - t; w[1] = t; }
      else if constexpr (order == 1) { w[0] = (T)-1; w[1] = (T)1; }
      else { w[0] = (T)0; w[1] = (T)0; }
    } else if constexpr (kt == kernel_e::quadratic) {
      const T a = (T)1.5 - t, b = t - (T)1, c = t - (T)0.5;
      if constexpr (order == 0) { w[0] = (T)0.5 * a * a; w[1] = (T)0.75 - b * b; w[2] = (T)0.5 * c * c; }
      else if constexpr (order == 1) { w[0] = -a; w[1] = -(b + b); w[2] = c; }
      else { w[0] = (T)1; w[1] = (T)-2; w[2] = (T)1; }
    } else if constexpr (kt == kernel_e::cubic) {
      const T z = (T)2 - t, p = t - (T)1, q = (T)2 - t, u = t - (T)1;  // distances to nodes 0..3: t, p = t-1, q = 2-t, 3-t
      if constexpr (order == 0) {
        w[0] = z * z * z / (T)6;
        w[1] = ((T)0.5 * p - (T)1) * p * p + (T)2 / (T)3;
        w[2] = ((T)0.5 * q - (T)1) * q * q + (T)2 / (T)3;
        w[3] = u * u * u / (T)6;
      } else if constexpr (order == 1) {
        w[0] = (T)-0.5 * z * z;
        w[1] = ((T)1.5 * p - (T)2) * p;
        w[2] = ((T)-1.5 * q + (T)2) * q;
        w[3] = (T)0.5 * u * u;
      } else {
        w[0] = z;
        w[1] = (T)-2 + (T)3 * p;
        w[2] = (T)-2 + (T)3 * q;
        w[3] = u;
      }
    } else {
      static_assert(order == 0, "the delta kernels carry no derivatives");
      constexpr int W = kernel_width(kt);
#pragma unroll
      for (int o = 0; o < W; ++o) {
        const T r = fabs(t - (T)o);
        T v = (T)0;
        if constexpr (kt == kernel_e::delta2) {
          if (r < (T)1) v = (T)1 - r;
        } else if constexpr (kt == kernel_e::delta3) {
          if (r <= (T)0.5) v = ((T)1 + sqrt((T)1 - (T)3 * r * r)) / (T)3;
          else if (r < (T)1.5) v = ((T)5 - (T)3 * r - sqrt((T)1 - (T)3 * ((T)1 - r) * ((T)1 - r))) / (T)6;
        } else {
          if (r <= (T)1) v = ((T)3 - (T)2 * r + sqrt((T)1 + (T)4 * r - (T)4 * r * r)) / (T)8;
          else if (r < (T)2) v = ((T)5 - (T)2 * r - sqrt((T)-7 + (T)12 * r - (T)4 * r * r)) / (T)8;
        }
        w[o] = v;
      }
    }
  }
}  // namespace detail

template <int dim, class T, int Side> struct SparseGridView;

// GridArena<GridView, kernel, deriv_order> (InterpolationKernel.hpp:271-560): the stencil of one sample point.
template <class GridViewT, kernel_e kt_ = kernel_e::linear, int drv_order = 0> struct GridArena {
  using grid_view_type = GridViewT;
  using value_type = typename GridViewT::value_type;
  static constexpr int dim = GridViewT::dim;
  static constexpr kernel_e kt = kt_;
  static constexpr int width = detail::kernel_width(kt_);
  static constexpr int deriv_order = drv_order;
  static_assert(drv_order >= 0 && drv_order <= 2, "weight derivative order is 0, 1 or 2");
  static_assert(drv_order == 0 || kt_ == kernel_e::linear || kt_ == kernel_e::quadratic || kt_ == kernel_e::cubic,
                "only the B-spline kernels have derivatives");
  using coord_type = small_vec<value_type, dim>;
  using integer_coord_type = small_vec<int, dim>;
  static constexpr int degree = width - 2;  // lerp degree: 0 linear / delta2, 1 quadratic / delta3, 2 cubic / delta4

  const GridViewT *gridPtr;
  value_type w[drv_order + 1][dim][width];  // [derivative order][axis][node]
  coord_type iLocalPos;                     // position measured from iCorner, index space
  integer_coord_type iCorner;               // lowest node of the stencil

  // collocated sample point X (index space); staggered: the values of face f sit at -0.5 along axis f % dim
  __device__ __forceinline__ GridArena(false_type_t, const GridViewT *gv, const coord_type &X, int f = -1) : gridPtr(gv) {
#pragma unroll
    for (int d = 0; d < dim; ++d) {
      const value_type shift = (f >= 0 && d == f % dim) ? (value_type)-0.5 : (value_type)0;
      iCorner[d] = detail::stencil_base(X[d] - shift, degree);
      iLocalPos[d] = X[d] - ((value_type)iCorner[d] + shift);
      detail::axis_weights<kt_, 0>(iLocalPos[d], w[0][d]);
      if constexpr (drv_order > 0) detail::axis_weights<kt_, 1>(iLocalPos[d], w[1][d]);
      if constexpr (drv_order > 1) detail::axis_weights<kt_, 2>(iLocalPos[d], w[2][d]);
    }
  }
  __device__ __forceinline__ GridArena(true_type_t, const GridViewT *gv, const coord_type &x, int f = -1)
      : GridArena(false_c, gv, gv->worldToIndex(x), f) {}

  ZS_FUNCTION static ndrange_t<dim> range() { return ndrange<dim>(width); }
  ZS_FUNCTION static integer_coord_type offset(const integer_coord_type &loc) { return loc; }
  ZS_FUNCTION integer_coord_type coord(const integer_coord_type &loc) const {
    integer_coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = iCorner[d] + loc[d];
    return r;
  }
  // weight(loc) = prod_d w_d[loc_d]; weightGradient<I>(loc): derivative along axis I; weightsGradient(loc): all axes
  ZS_FUNCTION value_type weight(const integer_coord_type &loc) const {
    value_type r = (value_type)1;
#pragma unroll
    for (int d = 0; d < dim; ++d) r *= w[0][d][loc[d]];
    return r;
  }
  template <class... Is, std::enable_if_t<sizeof...(Is) == dim && (std::is_integral_v<Is> && ...), int> = 0>
  ZS_FUNCTION value_type weight(Is... is) const { return weight(integer_coord_type{{(int)is...}}); }
  template <int I> ZS_FUNCTION value_type weightGradient(const integer_coord_type &loc) const {
    static_assert(drv_order > 0 && I < dim, "construct the arena with deriv_order >= 1");
    value_type r = (value_type)1;
#pragma unroll
    for (int d = 0; d < dim; ++d) r *= w[d == I ? 1 : 0][d][loc[d]];
    return r;
  }
  ZS_FUNCTION coord_type weightsGradient(const integer_coord_type &loc) const {
    static_assert(drv_order > 0, "construct the arena with deriv_order >= 1");
    coord_type g;
#pragma unroll
    for (int i = 0; i < dim; ++i) {
      value_type r = (value_type)1;
#pragma unroll
      for (int d = 0; d < dim; ++d) r *= w[d == i ? 1 : 0][d][loc[d]];
      g[i] = r;
    }
    return g;
  }
  // node value (the reference materialises all width^dim of them as `arena(chn, default)`; here they are read on demand)
  __device__ __forceinline__ value_type val(int chn, const integer_coord_type &loc, value_type defaultVal = {}) const {
    return gridPtr->valueOr(false_c, chn, coord(loc), defaultVal);
  }
  __device__ __forceinline__ value_type isample(int chn, value_type defaultVal = {}) const {
    value_type ret = (value_type)0;
    for (auto loc : range()) ret += weight(loc) * val(chn, loc, defaultVal);
    return ret;
  }
  __device__ __forceinline__ value_type isample(const char *prop, int chn, value_type defaultVal = {}) const {
    return isample(gridPtr->propertyOffset(prop) + chn, defaultVal);
  }
  __device__ __forceinline__ value_type minimum(int chn = 0) const {
    value_type r = std::numeric_limits<value_type>::max();
    for (auto loc : range()) {
      const value_type v = val(chn, loc, std::numeric_limits<value_type>::max());
      if (v < r) r = v;
    }
    return r;
  }
  __device__ __forceinline__ value_type maximum(int chn = 0) const {
    value_type r = std::numeric_limits<value_type>::lowest();
    for (auto loc : range()) {
      const value_type v = val(chn, loc, std::numeric_limits<value_type>::lowest());
      if (v > r) r = v;
    }
    return r;
  }
};

// SparseGridView (geometry/SparseGrid.hpp:199-916)
template <int dim_, class T, int Side> struct SparseGridView {
  static_assert(dim_ >= 1 && dim_ <= 3, "dim 1-3");
  static_assert(Side > 0 && (Side & (Side - 1)) == 0, "side length must be a power of two");
  static constexpr int dim = dim_, side_length = Side, block_size = detail::ipow(Side, dim_);
  static constexpr int sentinel_v = -1;
  static constexpr int max_props = 16, max_name = 16;
  using value_type = T;
  using coord_type = small_vec<T, dim_>;
  using integer_coord_type = small_vec<int, dim_>;
  using coord_t = coord_type;    // round-1 spellings
  using icoord_t = integer_coord_type;
  template <int N> using packed_t = small_vec<T, N>;

  BHTView<dim_> _table;
  TileVectorView<T, block_size> _grid;
  T _dx;
  coord_type _origin;
  T _background;
  // property directory for name-keyed access from kernels (the reference's SmallString lookups, TileVector.hpp:1150-1200)
  char _names[max_props][max_name];
  int _offs[max_props], _sizes[max_props], _np;

  ZS_FUNCTION int propertyOffset(const char *name) const {
    for (int p = 0; p < _np; ++p) {
      int c = 0;
      while (c < max_name && _names[p][c] == name[c] && name[c]) ++c;
      if (_names[p][c] == name[c]) return _offs[p];
    }
    return -1;
  }
  ZS_FUNCTION bool hasProperty(const char *name) const { return propertyOffset(name) >= 0; }
  ZS_FUNCTION coord_type indexToWorld(const coord_type &X) const {
    coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = X[d] * _dx + _origin[d];
    return r;
  }
  ZS_FUNCTION coord_type indexToWorld(const integer_coord_type &X) const {
    coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = (T)X[d] * _dx + _origin[d];
    return r;
  }
  ZS_FUNCTION coord_type worldToIndex(const coord_type &x) const {
    coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = (x[d] - _origin[d]) / _dx;
    return r;
  }
  ZS_FUNCTION T voxelSize(int = 0) const { return _dx; }
  __device__ __forceinline__ int numActiveBlocks() const { return _table.size(); }
  // linear cell index <-> in-block coordinate (:266-292): first axis slowest
  ZS_FUNCTION static integer_coord_type local_offset_to_coord(int offset) {
    integer_coord_type r;
    for (int d = dim - 1; d >= 0; --d, offset /= Side) r[d] = offset % Side;
    return r;
  }
  ZS_FUNCTION static int local_coord_to_offset(const integer_coord_type &c) {
    int r = c[0];
#pragma unroll
    for (int d = 1; d < dim; ++d) r = r * Side + c[d];
    return r;
  }
  ZS_FUNCTION static int global_coord_to_local_offset(const integer_coord_type &c) {
    int r = c[0] & (Side - 1);
#pragma unroll
    for (int d = 1; d < dim; ++d) r = r * Side + (c[d] & (Side - 1));
    return r;
  }
  struct BlockCell { int bno, cno; };
  // decomposeCoord (:305-309): cell = coord & (Side-1), block origin = coord - cell -> (table.query(origin), offset)
  __device__ __forceinline__ BlockCell decomposeCoord(const integer_coord_type &c) const {
    integer_coord_type org;
#pragma unroll
    for (int d = 0; d < dim; ++d) org[d] = c[d] - (c[d] & (Side - 1));
    return {_table.query(org), global_coord_to_local_offset(c)};
  }
  ZS_FUNCTION T &operator()(int chn, int bno, int cno) const { return _grid(chn, (std::size_t)bno, cno); }
  ZS_FUNCTION T &operator()(const char *prop, int chn, int bno, int cno) const { return _grid(propertyOffset(prop) + chn, (std::size_t)bno, cno); }
  // valueOr (:344-367)
  ZS_FUNCTION T valueOr(int chn, int bno, int cno, T defaultVal) const { return bno == sentinel_v ? defaultVal : _grid(chn, (std::size_t)bno, cno); }
  __device__ __forceinline__ T valueOr(false_type_t, int chn, const integer_coord_type &c, T defaultVal) const {
    const BlockCell bc = decomposeCoord(c);
    return valueOr(chn, bc.bno, bc.cno, defaultVal);
  }
  // orientation 0..dim-1: the face of this cell; dim..2dim-1: the same face of the next cell along that axis
  __device__ __forceinline__ T valueOr(true_type_t, int chn, const integer_coord_type &c, int orientation, T defaultVal) const {
    integer_coord_type cc = c;
    const int f = orientation % (2 * dim);
    if (f >= dim) ++cc[f - dim];
    return valueOr(false_c, chn, cc, defaultVal);
  }
  __device__ __forceinline__ T valueOr(int chn, const integer_coord_type &c, T defaultVal) const { return valueOr(false_c, chn, c, defaultVal); }
  __device__ __forceinline__ T valueOr(int chn, const integer_coord_type &c, int orientation, T defaultVal) const {
    return valueOr(true_c, chn, c, orientation, defaultVal);
  }
  // iCoord / wCoord (:404-417), staggered (:419-440)
  __device__ __forceinline__ integer_coord_type iCoord(int bno, int cno) const {
    const integer_coord_type l = local_offset_to_coord(cno);
    const int *k = _table.t.activeKeys + dim * (std::size_t)bno;
    integer_coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = k[d] + l[d];
    return r;
  }
  __device__ __forceinline__ integer_coord_type iCoord(std::size_t cellno) const { return iCoord((int)(cellno / block_size), (int)(cellno % block_size)); }
  __device__ __forceinline__ coord_type wCoord(int bno, int cno) const { return indexToWorld(iCoord(bno, cno)); }
  __device__ __forceinline__ coord_type wCoord(std::size_t cellno) const { return indexToWorld(iCoord(cellno)); }
  __device__ __forceinline__ coord_type iStaggeredCoord(int bno, int cno, int f) const {
    const integer_coord_type c = iCoord(bno, cno);
    coord_type r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = (T)c[d];
    r[f] -= (T)0.5;
    return r;
  }
  __device__ __forceinline__ coord_type wStaggeredCoord(int bno, int cno, int f) const { return indexToWorld(iStaggeredCoord(bno, cno, f)); }
  // insert / query by world position (:442-457): X = floor(worldToIndex(x) + 0.5), block origin = X - (X & (Side-1))
  __device__ __forceinline__ integer_coord_type blockOriginOf(const coord_type &x) const {
    const coord_type X_ = worldToIndex(x);
    integer_coord_type X;
#pragma unroll
    for (int d = 0; d < dim; ++d) {
      X[d] = (int)floor(X_[d] + (T)0.5);
      X[d] -= X[d] & (Side - 1);
    }
    return X;
  }
  __device__ __forceinline__ int insert(const coord_type &x) const { return _table.insert(blockOriginOf(x)); }
  __device__ __forceinline__ int query(const coord_type &x) const { return _table.query(blockOriginOf(x)); }

  // arenas (:369-388)
  template <kernel_e kt = kernel_e::linear, int order = 0> __device__ __forceinline__ auto iArena(const coord_type &X, kernel_t<kt> = {}) const {
    return GridArena<SparseGridView, kt, order>(false_c, this, X);
  }
  template <kernel_e kt = kernel_e::linear, int order = 0> __device__ __forceinline__ auto wArena(const coord_type &x, kernel_t<kt> = {}) const {
    return GridArena<SparseGridView, kt, order>(true_c, this, x);
  }
  template <kernel_e kt = kernel_e::linear, int order = 0> __device__ __forceinline__ auto iArena(const coord_type &X, int f, kernel_t<kt> = {}) const {
    return GridArena<SparseGridView, kt, order>(false_c, this, X, f);
  }
  template <kernel_e kt = kernel_e::linear, int order = 0> __device__ __forceinline__ auto wArena(const coord_type &x, int f, kernel_t<kt> = {}) const {
    return GridArena<SparseGridView, kt, order>(true_c, this, x, f);
  }
  // collocated sampling (:459-520)
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T iSample(int chn, const coord_type &X, kernel_t<kt> = {}) const {
    return iArena<kt>(X).isample(chn, _background);
  }
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T iSample(const char *prop, int chn, const coord_type &X, kernel_t<kt> = {}) const {
    return iSample<kt>(propertyOffset(prop) + chn, X);
  }
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T wSample(int chn, const coord_type &x, kernel_t<kt> = {}) const {
    return iSample<kt>(chn, worldToIndex(x));
  }
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T wSample(const char *prop, int chn, const coord_type &x, kernel_t<kt> = {}) const {
    return iSample<kt>(propertyOffset(prop) + chn, worldToIndex(x));
  }
  template <int N, kernel_e kt = kernel_e::linear> __device__ __forceinline__ packed_t<N> iPack(dim_t<N>, int chn, const coord_type &X, kernel_t<kt> = {}) const {
    const auto pad = iArena<kt>(X);  // one arena for all N channels (:703-712)
    packed_t<N> r;
#pragma unroll
    for (int d = 0; d < N; ++d) r[d] = pad.isample(chn + d, _background);
    return r;
  }
  template <int N, kernel_e kt = kernel_e::linear> __device__ __forceinline__ packed_t<N> wPack(dim_t<N> t, int chn, const coord_type &x, kernel_t<kt> k = {}) const {
    return iPack(t, chn, worldToIndex(x), k);
  }
  template <int N, kernel_e kt = kernel_e::linear> __device__ __forceinline__ packed_t<N> wPack(dim_t<N> t, const char *prop, const coord_type &x, kernel_t<kt> k = {}) const {
    return iPack(t, propertyOffset(prop), worldToIndex(x), k);
  }
  // staggered (MAC) sampling: channel chn + f lives on the faces normal to axis f (:584-609, :657-701)
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T iStaggeredSample(int chn, int f, const coord_type &X, kernel_t<kt> = {}) const {
    return iArena<kt>(X, f).isample(chn + f, _background);
  }
  template <kernel_e kt = kernel_e::linear> __device__ __forceinline__ T wStaggeredSample(int chn, int f, const coord_type &x, kernel_t<kt> = {}) const {
    return iStaggeredSample<kt>(chn, f, worldToIndex(x));
  }
  template <int N = dim_, kernel_e kt = kernel_e::linear> __device__ __forceinline__ packed_t<N> iStaggeredPack(int chn, const coord_type &X, dim_t<N> = {}, kernel_t<kt> = {}) const {
    static_assert(N <= dim_, "one component per axis");
    packed_t<N> r;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = iArena<kt>(X, i).isample(chn + i, _background);
    return r;
  }
  template <int N = dim_, kernel_e kt = kernel_e::linear> __device__ __forceinline__ packed_t<N> wStaggeredPack(int chn, const coord_type &x, dim_t<N> t = {}, kernel_t<kt> k = {}) const {
    return iStaggeredPack(chn, worldToIndex(x), t, k);
  }
  // value of component d of a staggered vector field at the centre of face f of cell X (:546-572): d == f reads the face itself, any
  // other component is the mean of the four d-faces around it
  __device__ __forceinline__ T iStaggeredCellSample(int propOffset, int d, const integer_coord_type &X, int bno, int cno, int f) const {
    static_assert(dim_ == 2 || dim_ == 3, "2-D and 3-D only");
    if (d == f) return valueOr(propOffset + d, bno, cno, _background);
    integer_coord_type a = X, b = X, c = X;
    a[d] += 1;
    b[f] -= 1;
    c[d] += 1;
    c[f] -= 1;
    return (valueOr(propOffset + d, bno, cno, _background) + valueOr(false_c, propOffset + d, a, _background)
            + valueOr(false_c, propOffset + d, b, _background) + valueOr(false_c, propOffset + d, c, _background)) * (T)0.25;
  }
  __device__ __forceinline__ T iStaggeredCellSample(int propOffset, int d, const integer_coord_type &X, int f) const {
    integer_coord_type cc = X;
    if (f >= dim) ++cc[f -= dim];
    const BlockCell bc = decomposeCoord(cc);
    return iStaggeredCellSample(propOffset, d, cc, bc.bno, bc.cno, f);
  }
  __device__ __forceinline__ packed_t<dim_> iStaggeredCellPack(int propOffset, const integer_coord_type &X, int f) const {
    integer_coord_type cc = X;
    if (f >= dim) ++cc[f -= dim];
    const BlockCell bc = decomposeCoord(cc);
    packed_t<dim_> r;
#pragma unroll
    for (int d = 0; d < dim; ++d) r[d] = iStaggeredCellSample(propOffset, d, cc, bc.bno, bc.cno, f);
    return r;
  }
};

// SparseGrid<dim, T, Side> (geometry/SparseGrid.hpp:16-188)
template <int dim_ = 3, class T = float, int Side = 8> struct SparseGrid {
  static constexpr int dim = dim_, side_length = Side, block_size = detail::ipow(Side, dim_);
  using value_type = T;
  using table_type = bht<int, dim_, int, 16>;
  using grid_storage_type = TileVector<T, block_size>;
  using view_type = SparseGridView<dim_, T, Side>;
  SparseGrid(const std::vector<PropertyTag> &tags, std::size_t numBlocks, memsrc_e mre = memsrc_e::device)
      : _table(numBlocks), _grid(tags, numBlocks * (std::size_t)block_size, mre) {
    for (int d = 0; d < dim; ++d) _origin[d] = (T)0;
  }
  SparseGrid(int numChns, std::size_t numBlocks, memsrc_e mre = memsrc_e::device)
      : SparseGrid(std::vector<PropertyTag>{{"unnamed", numChns}}, numBlocks, mre) {}
  template <class Alloc> SparseGrid(const Alloc &, const std::vector<PropertyTag> &tags, std::size_t numBlocks)
      : SparseGrid(tags, numBlocks, memsrc_e::device) {}
  std::size_t numBlocks() const { return _table.size(); }
  std::size_t numReservedBlocks() const { return _grid.size() / block_size; }
  int numChannels() const { return _grid.numChannels(); }
  int getPropertyOffset(const std::string &n) const { return _grid.getPropertyOffset(n); }
  int getPropertySize(const std::string &n) const { return _grid.getPropertySize(n); }
  bool hasProperty(const std::string &n) const { return _grid.hasProperty(n); }
  void scale(T s) { _dx *= s; }  // :181-182
  template <class... Ts> void translate(Ts... t) {  // :170-172
    static_assert(sizeof...(Ts) == dim_, "one offset per axis");
    const T v[dim_] = {(T)t...};
    for (int d = 0; d < dim; ++d) _origin[d] += v[d];
  }
  T voxelSize() const { return _dx; }
  void reset(int ch = 0) { _grid.reset(ch); }
  view_type view() {
    view_type v{};
    v._table = _table.view();
    v._grid = TileVectorView<T, block_size>{_grid.data(), _grid.size(), _grid.numChannels()};
    v._dx = _dx;
    for (int d = 0; d < dim; ++d) v._origin[d] = _origin[d];
    v._background = _background;
    v._np = 0;
    for (std::size_t i = 0; i < _grid._tags.size() && v._np < view_type::max_props; ++i, ++v._np) {
      const std::string &nm = _grid._tags[i].name;
      int c = 0;
      for (; c < (int)nm.size() && c < view_type::max_name - 1; ++c) v._names[v._np][c] = nm[c];
      v._names[v._np][c] = 0;
      v._offs[v._np] = _grid._offsets[i];
      v._sizes[v._np] = _grid._tags[i].numChannels;
    }
    return v;
  }
  table_type _table;
  grid_storage_type _grid;
  T _dx = (T)1, _origin[dim_];
  T _background = (T)0;
};

}  // namespace zs
