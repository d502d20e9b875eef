// structures.hpp -- zs::Particles<T, d> / ParticlesView and zs::Grids<T, d, Side> / GridsView of the C++ face (included by zs_rocm.hpp).
//
// The types the reference's MPM transfer functors take (simulation/transfer/P2G.hpp:27-49: P2GTransfer{wrapv<space>, wrapv<scheme>, dt,
// model, table, particles, grids}; geometry/Structurefree.hpp:21-300; geometry/Structure.hpp:131-265, 811-1090), over this face's
// Vector / TileVector: a user translation unit written against them (`pars.pos(i)`, `pars.F(i)`, `grids.grid(collocated_c)(chn, block, cell)`,
// `grid_block.pack<3>(1, cellid)`) compiles with hipcc unchanged.  Storage is the reference's: one Vector per named particle attribute
// (array of structs per attribute: Vector<vec3>, Vector<vec9>), one TileVector<T, Side^d> per grid category whose tile is a block.
// `ports()` hands the same memory to the C ABI (zs_rocm_particles / the grid pointer of zs_rocm_mpm_p2g ...): the library's kernels read
// AoS attributes through their iterator ports ("attributes are passed as aosoa_iterator_ports").
#pragma once

#include <array>
#include <map>
#include <variant>

namespace zs {

enum class attrib_e : unsigned char { scalar = 0, vector, matrix, affine };
enum class grid_e : unsigned char { collocated = 0, cellcentered, staggered, total };
constexpr wrapv<grid_e::collocated> collocated_c{};
constexpr wrapv<grid_e::cellcentered> cellcentered_c{};
constexpr wrapv<grid_e::staggered> staggered_c{};
template <class T, int N> using vec = small_vec<T, N>;

// ------------------------------------------------------------------------------------------------------------------------- Particles
template <class ValueT = float, int d = 3> struct Particles {  // geometry/Structurefree.hpp:21-224
  using T = ValueT;
  using TV = vec<T, d>;
  using TM = vec<T, d * d>;
  using TMAffine = vec<T, (d + 1) * (d + 1)>;
  using Attribute = std::variant<Vector<T>, Vector<TV>, Vector<TM>, Vector<TMAffine>>;
  using allocator_type = ZSPmrAllocator;
  using size_type = std::size_t;
  static constexpr int dim = d;

  template <class TT> static constexpr attrib_e get_attribute_enum(const Vector<TT> &) {
    if constexpr (std::is_same_v<TT, T>) return attrib_e::scalar;
    else if constexpr (std::is_same_v<TT, TV>) return attrib_e::vector;
    else if constexpr (std::is_same_v<TT, TM>) return attrib_e::matrix;
    else return attrib_e::affine;
  }
  static attrib_e get_attribute_enum(const Attribute &att) {
    return std::visit([](const auto &a) { return get_attribute_enum(a); }, att);
  }

  Particles(const allocator_type &allocator, size_type count) : _alloc(allocator) { _attributes["x"] = Vector<TV>{allocator, count}; }  // "x" is a reserved key
  Particles(memsrc_e mre = memsrc_e::host, ProcID devid = -1) : Particles{get_memory_source(mre, devid), 0} {}
  Particles(size_type count, memsrc_e mre = memsrc_e::host, ProcID devid = -1) : Particles{get_memory_source(mre, devid), count} {}

  MemoryLocation memoryLocation() const noexcept { return _alloc.location; }
  memsrc_e space() const noexcept { return _alloc.location.memspace(); }
  ProcID devid() const noexcept { return _alloc.location.devid(); }
  size_type size() const noexcept {
    return std::visit([](const auto &a) { return a.size(); }, _attributes.at("x"));
  }
  const allocator_type &get_allocator() const noexcept { return _alloc; }

  auto &attrs() { return _attributes; }
  const auto &attrs() const { return _attributes; }
  const Attribute *tryGet(const std::string &attrib) const noexcept {
    auto it = _attributes.find(attrib);
    return it != _attributes.end() ? &it->second : nullptr;
  }
  Attribute *tryGet(const std::string &attrib) noexcept {
    auto it = _attributes.find(attrib);
    return it != _attributes.end() ? &it->second : nullptr;
  }
  bool hasAttr(const std::string &attrib, bool checkEmpty = false) const noexcept {
    if (auto obj = tryGet(attrib))
      if (!checkEmpty || std::visit([](const auto &a) -> bool { return a.size() > 0; }, *obj)) return true;
    return false;
  }
  Attribute &attr(const std::string &attrib) { return _attributes.at(attrib); }
  const Attribute &attr(const std::string &attrib) const { return _attributes.at(attrib); }
  template <class AT> Vector<AT> &attr(const std::string &attrib) { return std::get<Vector<AT>>(attr(attrib)); }
  template <class AT> const Vector<AT> &attr(const std::string &attrib) const { return std::get<Vector<AT>>(attr(attrib)); }
  Vector<T> &attrScalar(const std::string &a) { return attr<T>(a); }
  const Vector<T> &attrScalar(const std::string &a) const { return attr<T>(a); }
  Vector<TV> &attrVector(const std::string &a) { return attr<TV>(a); }
  const Vector<TV> &attrVector(const std::string &a) const { return attr<TV>(a); }
  Vector<TM> &attrMatrix(const std::string &a) { return attr<TM>(a); }
  const Vector<TM> &attrMatrix(const std::string &a) const { return attr<TM>(a); }
  Vector<TMAffine> &attrAffine(const std::string &a) { return attr<TMAffine>(a); }
  const Vector<TMAffine> &attrAffine(const std::string &a) const { return attr<TMAffine>(a); }

  template <class TT = void> TT *getAttrAddress(const std::string &attrib) {
    if (hasAttr(attrib)) return std::visit([](auto &a) -> TT * { return (TT *)a.data(); }, attr(attrib));
    return (TT *)nullptr;
  }
  template <class TT = void> const TT *getAttrAddress(const std::string &attrib) const {
    if (hasAttr(attrib)) return std::visit([](const auto &a) -> const TT * { return (const TT *)a.data(); }, attr(attrib));
    return (const TT *)nullptr;
  }

  Attribute &addAttr(const std::string &attrib, attrib_e ae) {  // :172-193: an existing attribute of the same kind is kept
    if (auto obj = tryGet(attrib))
      if (get_attribute_enum(*obj) == ae) return *obj;
    auto &att = _attributes[attrib];
    switch (ae) {
      case attrib_e::scalar: att = Vector<T>{_alloc, size()}; break;
      case attrib_e::vector: att = Vector<TV>{_alloc, size()}; break;
      case attrib_e::matrix: att = Vector<TM>{_alloc, size()}; break;
      default: att = Vector<TMAffine>{_alloc, size()}; break;
    }
    return att;
  }
  void append(const Particles &other) {  // :195-214
    for (auto &&kv : other.attrs()) {
      if (auto obj = tryGet(kv.first)) {
        if (obj->index() != kv.second.index())
          throw std::runtime_error("attributes of the same name \"" + kv.first + "\" are of different types");
        std::visit([&](auto &dst) { dst.append(std::get<std::decay_t<decltype(dst)>>(kv.second)); }, *obj);
      } else
        _attributes[kv.first] = kv.second;
    }
  }
  void resize(size_type newSize) {
    for (auto &&kv : _attributes) std::visit([newSize](auto &a) { a.resize(newSize); }, kv.second);
  }
  std::vector<std::array<ValueT, dim>> retrievePositions() const {  // :72-80
    const auto &X = attr<TV>("x");
    std::vector<std::array<ValueT, dim>> ret(X.size());
    if (X.size()) (void)hipMemcpy(ret.data(), X.data(), X.size() * sizeof(TV), hipMemcpyDefault);
    return ret;
  }

  // the attributes as the C ABI takes them: AoS iterator ports (tile width 1: element i, component c at base[i * N + c]).  Names follow
  // ParticlesView: "m", "x", "v", "C", "F" or "J" (the fluid's volume ratio travels in the F slot, zs_rocm.h), "logJp"
  zs_rocm_particles ports() {
    static_assert(sizeof(T) == 4 && d == 3, "the C ABI's MPM entry points are f32, 3-D");
    auto port = [&](const char *name, unsigned comps) -> zs_rocm_attr {
      zs_rocm_attr a{};
      a.base = hasAttr(name) ? getAttrAddress<float>(name) : nullptr;
      a.idx = 0;
      a.numTileBits = 0;
      a.tileMask = 0;
      a.numChns = comps;
      return a;
    };
    zs_rocm_particles p{};
    p.mass = port("m", 1);
    p.pos = port("x", 3);
    p.vel = port("v", 3);
    p.C = port("C", 9);
    p.F = hasAttr("F") ? port("F", 9) : port("J", 1);
    p.logJp = port("logJp", 1);
    p.stress = zs_rocm_attr{};
    p.n = size();
    return p;
  }

protected:
  allocator_type _alloc;
  std::map<std::string, Attribute> _attributes;
};

// ParticlesView (geometry/Structurefree.hpp:226-330): the accessor names below are the interface transfer functors are written against
// (mass / pos / vel / Dinv / J / F / C / B / logJp / size); the storage behind them is this backend's own -- one table of attribute base
// pointers filled from the container by name, shared by the mutable and the const view.
namespace detail {
  enum particle_attr : int { pa_m, pa_x, pa_v, pa_dinv, pa_j, pa_f, pa_c, pa_logjp, pa_count };
  template <class Byte> struct particle_attr_table {
    Byte *slot[pa_count] = {};
    std::size_t count = 0;
    particle_attr_table() = default;
    template <class ParticlesT> explicit particle_attr_table(ParticlesT &particles) : count(particles.size()) {
      static constexpr const char *names[pa_count] = {"m", "x", "v", "Dinv", "J", "F", "C", "logJp"};
      for (int k = 0; k < pa_count; ++k) slot[k] = (Byte *)particles.getAttrAddress(names[k]);
    }
    template <class E> ZS_FUNCTION E &at(int k, std::size_t i) const { return reinterpret_cast<E *>(slot[k])[i]; }
  };
}  // namespace detail
template <execspace_e space, class ParticlesT, class = void> struct ParticlesView {
  using T = typename ParticlesT::T;
  using TV = typename ParticlesT::TV;
  using TM = typename ParticlesT::TM;
  static constexpr int dim = ParticlesT::dim;
  using size_type = typename ParticlesT::size_type;
  ParticlesView() = default;
  explicit ParticlesView(ParticlesT &particles) : _tab(particles) {}
  ZS_FUNCTION T &mass(size_type i) const { return _tab.template at<T>(detail::pa_m, i); }
  ZS_FUNCTION TV &pos(size_type i) const { return _tab.template at<TV>(detail::pa_x, i); }
  ZS_FUNCTION TV &vel(size_type i) const { return _tab.template at<TV>(detail::pa_v, i); }
  ZS_FUNCTION TV &Dinv(size_type i) const { return _tab.template at<TV>(detail::pa_dinv, i); }
  ZS_FUNCTION T &J(size_type i) const { return _tab.template at<T>(detail::pa_j, i); }          // volume ratio (EquationOfState fluid)
  ZS_FUNCTION TM &F(size_type i) const { return _tab.template at<TM>(detail::pa_f, i); }        // deformation gradient (solids)
  ZS_FUNCTION TM &C(size_type i) const { return _tab.template at<TM>(detail::pa_c, i); }        // affine velocity field of the APIC transfer
  ZS_FUNCTION TM &B(size_type i) const { return C(i); }                                         // (the reference aliases B to C)
  ZS_FUNCTION T &logJp(size_type i) const { return _tab.template at<T>(detail::pa_logjp, i); }  // plastic volume (DruckerPrager / NACC)
  ZS_FUNCTION size_type size() const noexcept { return (size_type)_tab.count; }

private:
  detail::particle_attr_table<char> _tab;
};
template <execspace_e space, class ParticlesT> struct ParticlesView<space, const ParticlesT> {
  using T = typename ParticlesT::T;
  using TV = typename ParticlesT::TV;
  using TM = typename ParticlesT::TM;
  static constexpr int dim = ParticlesT::dim;
  using size_type = typename ParticlesT::size_type;
  ParticlesView() = default;
  explicit ParticlesView(const ParticlesT &particles) : _tab(particles) {}
  ZS_FUNCTION T mass(size_type i) const { return _tab.template at<const T>(detail::pa_m, i); }
  ZS_FUNCTION const TV &pos(size_type i) const { return _tab.template at<const TV>(detail::pa_x, i); }
  ZS_FUNCTION const TV &vel(size_type i) const { return _tab.template at<const TV>(detail::pa_v, i); }
  ZS_FUNCTION const TV &Dinv(size_type i) const { return _tab.template at<const TV>(detail::pa_dinv, i); }
  ZS_FUNCTION const T &J(size_type i) const { return _tab.template at<const T>(detail::pa_j, i); }
  ZS_FUNCTION const TM &F(size_type i) const { return _tab.template at<const TM>(detail::pa_f, i); }
  ZS_FUNCTION const TM &C(size_type i) const { return _tab.template at<const TM>(detail::pa_c, i); }
  ZS_FUNCTION const TM &B(size_type i) const { return C(i); }
  ZS_FUNCTION const T &logJp(size_type i) const { return _tab.template at<const T>(detail::pa_logjp, i); }
  ZS_FUNCTION size_type size() const noexcept { return (size_type)_tab.count; }

private:
  detail::particle_attr_table<const char> _tab;
};
template <execspace_e space, class V, int d> ParticlesView<space, Particles<V, d>> proxy(Particles<V, d> &p) { return ParticlesView<space, Particles<V, d>>{p}; }
template <execspace_e space, class V, int d> ParticlesView<space, const Particles<V, d>> proxy(const Particles<V, d> &p) {
  return ParticlesView<space, const Particles<V, d>>{p};
}

// ------------------------------------------------------------------------------------------------------------------------------ Grids
namespace detail {
constexpr int pow_integral(int b, int e) { return e == 0 ? 1 : b * pow_integral(b, e - 1); }
// proxy<space>({}, tilevector): the named view over ALL properties (the reference's meaning of an empty tag list, TileVector.hpp:1513-1540)
template <execspace_e space, class T, int L> TileVectorNamedView<T, L> all_props_view(TileVector<T, L> &v) {
  TileVectorNamedView<T, L> r{};
  static_cast<TileVectorView<T, L> &>(r) = view<space>(v);
  for (std::size_t k = 0; k < v._tags.size() && r._np < r.max_props; ++k) {
    const std::string &nm = v._tags[k].name;
    int c = 0;
    for (; c < (int)nm.size() && c < r.max_name - 1; ++c) r._names[r._np][c] = nm[c];
    r._names[r._np][c] = 0;
    r._offs[r._np] = v._offsets[k];
    r._sizes[r._np] = v._tags[k].numChannels;
    ++r._np;
  }
  return r;
}
}  // namespace detail
template <class ValueT = float, int d_ = 3, int SideLength = 4, grid_e category_ = grid_e::collocated> struct Grid {  // Structure.hpp:18-128
  using value_type = ValueT;
  using size_type = std::size_t;
  using channel_counter_type = int;
  using cell_index_type = int;
  static constexpr int dim = d_;
  static constexpr int side_length = SideLength;
  static constexpr int block_size = detail::pow_integral(SideLength, d_);
  static constexpr grid_e category = category_;
  static constexpr bool is_power_of_two = (SideLength & (SideLength - 1)) == 0;
  static constexpr int num_cell_bits = SideLength == 1 ? 0 : (SideLength == 2 ? 1 : (SideLength == 4 ? 2 : (SideLength == 8 ? 3 : 4)));
  using grid_storage_t = TileVector<value_type, block_size>;
  using IV = vec<int, dim>;
  using TV = vec<value_type, dim>;
  using CellIV = vec<int, dim>;

  Grid(const ZSPmrAllocator &allocator, const std::vector<PropertyTag> &channelTags, value_type dx, size_type count = 0)
      : blocks{channelTags, count * (size_type)block_size, allocator.location.memspace()}, dx{dx} {}
  Grid(const std::vector<PropertyTag> &channelTags, value_type dx, size_type count, memsrc_e mre = memsrc_e::host, ProcID devid = -1)
      : Grid{get_memory_source(mre, devid), channelTags, dx, count} {}
  Grid(value_type dx = 1.f, memsrc_e mre = memsrc_e::host, ProcID devid = -1) : Grid{get_memory_source(mre, devid), {{"m", 1}, {"v", dim}}, dx, 0} {}

  memsrc_e memspace() const noexcept { return blocks._mre; }
  size_type size() const noexcept { return blocks.size(); }
  size_type numBlocks() const noexcept { return blocks.numTiles(); }
  size_type numTiles() const noexcept { return blocks.numTiles(); }
  channel_counter_type numChannels() const noexcept { return blocks.numChannels(); }
  bool hasProperty(const std::string &s) const noexcept { return blocks.hasProperty(s); }
  int getPropertySize(const std::string &s) const { return blocks.getPropertySize(s); }
  int getPropertyOffset(const std::string &s) const { return blocks.getPropertyOffset(s); }
  void resize(size_type numBlocks) {  // (contents of the first min(old, new) blocks are kept, like TileVector::resize)
    grid_storage_t nb{blocks._tags, numBlocks * (size_type)block_size, blocks._mre};
    const std::size_t keep = (blocks._buf.size() < nb._buf.size() ? blocks._buf.size() : nb._buf.size()) * sizeof(value_type);
    if (keep) (void)hipMemcpy(nb.data(), blocks.data(), keep, hipMemcpyDefault);
    blocks = std::move(nb);
  }
  template <class Policy> void append_channels(Policy &&policy, const std::vector<PropertyTag> &tags) { blocks.append_channels(policy, tags); }
  template <class Policy> void reset(Policy &&policy, value_type val) { blocks.reset(policy, val); }
  value_type *data() { return blocks.data(); }  // [block][channel][cell]: the `grid` pointer of the C ABI's MPM entry points
  const value_type *data() const { return blocks.data(); }

  grid_storage_t blocks;
  value_type dx;
};

template <class ValueT = float, int d_ = 3, int SideLength = 4> struct Grids {  // Structure.hpp:131-262
  template <grid_e category = grid_e::collocated> using grid_t = Grid<ValueT, d_, SideLength, category>;
  using collocated_grid_t = grid_t<grid_e::collocated>;
  using value_type = ValueT;
  using allocator_type = ZSPmrAllocator;
  using cell_index_type = int;
  using coord_index_type = int;
  using size_type = std::size_t;
  using channel_counter_type = int;
  static constexpr int dim = d_;
  static constexpr int side_length = SideLength;
  static constexpr int block_space() noexcept { return collocated_grid_t::block_size; }
  static constexpr bool is_power_of_two = collocated_grid_t::is_power_of_two;
  static constexpr int num_cell_bits = collocated_grid_t::num_cell_bits;
  using grid_storage_t = typename collocated_grid_t::grid_storage_t;
  using CellIV = vec<int, dim>;
  using IV = vec<int, dim>;
  using TV = vec<value_type, dim>;

  Grids(const allocator_type &allocator, const std::vector<PropertyTag> &channelTags = {{"m", 1}, {"v", dim}}, value_type dx = 1.f,
        size_type numBlocks = 0, grid_e ge = grid_e::collocated)
      : _collocatedGrid{allocator, channelTags, dx}, _cellcenteredGrid{allocator, channelTags, dx}, _staggeredGrid{allocator, channelTags, dx},
        _dx{dx}, _primaryGrid{ge} {
    if (ge == grid_e::collocated) _collocatedGrid.resize(numBlocks);
    else if (ge == grid_e::cellcentered) _cellcenteredGrid.resize(numBlocks);
    else if (ge == grid_e::staggered) _staggeredGrid.resize(numBlocks);
  }
  Grids(const std::vector<PropertyTag> &channelTags = {{"m", 1}, {"v", dim}}, value_type dx = 1.f, size_type numBlocks = 0,
        memsrc_e mre = memsrc_e::host, ProcID devid = -1, grid_e ge = grid_e::collocated)
      : Grids{get_memory_source(mre, devid), channelTags, dx, numBlocks, ge} {}

  template <class F> decltype(auto) gridApply(grid_e category, F &&f) {
    if (category == grid_e::collocated) return f(_collocatedGrid.blocks);
    else if (category == grid_e::cellcentered) return f(_cellcenteredGrid.blocks);
    return f(_staggeredGrid.blocks);
  }
  template <class F> decltype(auto) gridApply(grid_e category, F &&f) const {
    if (category == grid_e::collocated) return f(_collocatedGrid.blocks);
    else if (category == grid_e::cellcentered) return f(_cellcenteredGrid.blocks);
    return f(_staggeredGrid.blocks);
  }
  memsrc_e space() const noexcept { return _collocatedGrid.memspace(); }
  size_type size() const noexcept { return gridApply(_primaryGrid, [](const auto &b) -> size_type { return b.size(); }); }
  size_type numBlocks() const noexcept { return gridApply(_primaryGrid, [](const auto &b) -> size_type { return b.numTiles(); }); }
  void align(grid_e targetGrid) {  // :223-227
    if (targetGrid == _primaryGrid) return;
    const auto nb = numBlocks();
    if (targetGrid == grid_e::collocated) _collocatedGrid.resize(nb);
    else if (targetGrid == grid_e::cellcentered) _cellcenteredGrid.resize(nb);
    else _staggeredGrid.resize(nb);
  }
  template <grid_e category = grid_e::collocated> auto &grid(wrapv<category> = {}) noexcept {
    if constexpr (category == grid_e::collocated) return _collocatedGrid;
    else if constexpr (category == grid_e::cellcentered) return _cellcenteredGrid;
    else return _staggeredGrid;
  }
  template <grid_e category = grid_e::collocated> const auto &grid(wrapv<category> = {}) const noexcept {
    if constexpr (category == grid_e::collocated) return _collocatedGrid;
    else if constexpr (category == grid_e::cellcentered) return _cellcenteredGrid;
    else return _staggeredGrid;
  }
  template <grid_e category = grid_e::collocated> size_type numCells(wrapv<category> c = {}) const noexcept { return grid(c).size() * grid(c).numChannels(); }

  grid_t<grid_e::collocated> _collocatedGrid;
  grid_t<grid_e::cellcentered> _cellcenteredGrid;
  grid_t<grid_e::staggered> _staggeredGrid;
  value_type _dx;
  grid_e _primaryGrid;
};

template <execspace_e space, class GridsT, class = void> struct GridsView {  // Structure.hpp:811-1120
  static constexpr bool is_const_structure = std::is_const_v<GridsT>;
  using grids_t = std::remove_const_t<GridsT>;
  using value_type = typename grids_t::value_type;
  static constexpr int dim = grids_t::dim;
  static constexpr int side_length = grids_t::side_length;
  static constexpr int block_space() noexcept { return grids_t::block_space(); }
  static constexpr bool is_power_of_two = grids_t::is_power_of_two;
  static constexpr int num_cell_bits = grids_t::num_cell_bits;
  using grid_view_t = TileVectorNamedView<value_type, grids_t::block_space()>;
  using size_type = std::size_t;
  using channel_counter_type = int;
  using cell_index_type = int;
  using coord_index_type = int;
  using CellIV = typename grids_t::CellIV;
  using IV = typename grids_t::IV;
  using TV = typename grids_t::TV;

  ZS_FUNCTION static constexpr CellIV cellid_to_coord(cell_index_type cellid) noexcept {
    CellIV ret{};
    for (int d = dim - 1; d >= 0; --d, cellid /= side_length) ret.v[d] = cellid % side_length;
    return ret;
  }
  template <class Ti> ZS_FUNCTION static constexpr cell_index_type coord_to_cellid(const vec<Ti, dim> &coord) noexcept {
    cell_index_type ret{0};
    for (int d = 0; d != dim; ++d) ret = ret * side_length + (cell_index_type)coord.v[d];
    return ret;
  }
  template <class Ti> ZS_FUNCTION static constexpr cell_index_type global_coord_to_cellid(const vec<Ti, dim> &coord) noexcept {
    cell_index_type ret{0};
    for (int d = 0; d != dim; ++d) {
      const int c = (int)coord.v[d] % side_length;
      ret = ret * side_length + (c < 0 ? c + side_length : c);  // (power-of-two sides: coord & (side - 1), the reference's form)
    }
    return ret;
  }

  GridsView() = default;
  explicit GridsView(GridsT &grids)
      : _collocatedGrid{detail::all_props_view<space>(const_cast<grids_t &>(grids).grid(collocated_c).blocks)},
        _cellcenteredGrid{detail::all_props_view<space>(const_cast<grids_t &>(grids).grid(cellcentered_c).blocks)},
        _staggeredGrid{detail::all_props_view<space>(const_cast<grids_t &>(grids).grid(staggered_c).blocks)}, _dx{grids._dx} {}

  // one block of a grid: a tile of the TileVector
  template <grid_e category = grid_e::collocated> struct Block {
    static constexpr int block_space() noexcept { return GridsView::block_space(); }
    grid_view_t grid;
    size_type blockno;
    value_type dx;
    ZS_FUNCTION bool hasProperty(const char *name) const { return grid.hasProperty(name); }
    template <class Ti> ZS_FUNCTION value_type &operator()(channel_counter_type c, const vec<Ti, dim> &loc) const { return grid(c, blockno, coord_to_cellid(loc)); }
    template <class Ti> ZS_FUNCTION value_type &operator()(const char *name, const vec<Ti, dim> &loc) const {
      return grid(grid.propertyOffset(name), blockno, coord_to_cellid(loc));
    }
    ZS_FUNCTION value_type &operator()(channel_counter_type c, cell_index_type cellid) const { return grid(c, blockno, cellid); }
    ZS_FUNCTION value_type &operator()(const char *name, cell_index_type cellid) const { return grid(grid.propertyOffset(name), blockno, cellid); }
    template <int N> ZS_FUNCTION vec<value_type, N> pack(channel_counter_type chn, cell_index_type cellid) const {
      return grid.pack(dim_c<N>, chn, blockno * (size_type)block_space() + (size_type)cellid);
    }
    template <int N> ZS_FUNCTION vec<value_type, N> pack(const char *name, cell_index_type cellid) const { return pack<N>(grid.propertyOffset(name), cellid); }
    template <int N, class V> ZS_FUNCTION void set(channel_counter_type chn, cell_index_type cellid, const vec<V, N> &val) const {
      static_assert(!is_const_structure, "");
      vec<value_type, N> v;
      for (int k = 0; k < N; ++k) v.v[k] = (value_type)val.v[k];
      grid.set(chn, blockno * (size_type)block_space() + (size_type)cellid, v);
    }
    template <int N, class V> ZS_FUNCTION void set(const char *name, cell_index_type cellid, const vec<V, N> &val) const {
      set<N>(grid.propertyOffset(name), cellid, val);
    }
    ZS_FUNCTION size_type size() const noexcept { return (size_type)block_space(); }
  };
  template <grid_e category = grid_e::collocated> struct Grid {
    using size_type = typename GridsView::size_type;
    using cell_index_type = typename GridsView::cell_index_type;
    using value_type = typename GridsView::value_type;
    using channel_counter_type = typename GridsView::channel_counter_type;
    static constexpr int dim = GridsView::dim;
    static constexpr int side_length = GridsView::side_length;
    static constexpr int block_space() noexcept { return GridsView::block_space(); }
    ZS_FUNCTION static constexpr auto cellid_to_coord(cell_index_type cellid) noexcept { return GridsView::cellid_to_coord(cellid); }
    template <class Ti> ZS_FUNCTION static constexpr auto coord_to_cellid(const vec<Ti, dim> &c) noexcept { return GridsView::coord_to_cellid(c); }
    template <class Ti> ZS_FUNCTION static constexpr auto global_coord_to_cellid(const vec<Ti, dim> &c) noexcept { return GridsView::global_coord_to_cellid(c); }
    grid_view_t grid;
    value_type dx;
    ZS_FUNCTION bool hasProperty(const char *name) const { return grid.hasProperty(name); }
    ZS_FUNCTION Block<category> block(size_type i) const { return Block<category>{grid, i, dx}; }
    ZS_FUNCTION Block<category> operator[](size_type i) const { return block(i); }
    template <class Ti> ZS_FUNCTION value_type &operator()(channel_counter_type c, size_type blockid, const vec<Ti, dim> &loc) const {
      return grid(c, blockid * (size_type)block_space() + (size_type)coord_to_cellid(loc));
    }
    template <class Ti> ZS_FUNCTION value_type &operator()(const char *name, size_type blockid, const vec<Ti, dim> &loc) const {
      return grid(grid.propertyOffset(name), blockid * (size_type)block_space() + (size_type)coord_to_cellid(loc));
    }
    ZS_FUNCTION value_type &operator()(channel_counter_type chn, size_type cellid) const { return grid(chn, cellid); }
    ZS_FUNCTION value_type &operator()(const char *name, size_type cellid) const { return grid(grid.propertyOffset(name), cellid); }
    ZS_FUNCTION value_type &operator()(channel_counter_type chn, size_type blockid, cell_index_type cellid) const { return grid(chn, blockid, cellid); }
    template <int N> ZS_FUNCTION vec<value_type, N> pack(channel_counter_type chn, size_type cellid) const { return grid.pack(dim_c<N>, chn, cellid); }
    template <int N> ZS_FUNCTION vec<value_type, N> pack(const char *name, size_type cellid) const { return grid.pack(dim_c<N>, grid.propertyOffset(name), cellid); }
    template <int N, class Ti> ZS_FUNCTION vec<value_type, N> pack(channel_counter_type chn, size_type blockid, const vec<Ti, dim> &loc) const {
      return grid.pack(dim_c<N>, chn, blockid * (size_type)block_space() + (size_type)coord_to_cellid(loc));
    }
    template <int N, class V> ZS_FUNCTION void set(channel_counter_type chn, size_type cellid, const vec<V, N> &val) const {
      static_assert(!is_const_structure, "");
      vec<value_type, N> v;
      for (int k = 0; k < N; ++k) v.v[k] = (value_type)val.v[k];
      grid.set(chn, cellid, v);
    }
    template <int N, class V> ZS_FUNCTION void set(const char *name, size_type cellid, const vec<V, N> &val) const { set<N>(grid.propertyOffset(name), cellid, val); }
    ZS_FUNCTION size_type size() const noexcept { return grid.size(); }
    ZS_FUNCTION int numChannels() const noexcept { return grid.numChannels(); }
  };

  template <grid_e category = grid_e::collocated> ZS_FUNCTION Grid<category> grid(wrapv<category> = {}) const {
    if constexpr (category == grid_e::collocated) return Grid<category>{_collocatedGrid, _dx};
    else if constexpr (category == grid_e::cellcentered) return Grid<category>{_cellcenteredGrid, _dx};
    else return Grid<category>{_staggeredGrid, _dx};
  }
  template <grid_e category = grid_e::collocated> ZS_FUNCTION Block<category> block(size_type i) const { return grid(wrapv<category>{}).block(i); }
  ZS_FUNCTION Block<grid_e::collocated> operator[](size_type i) const { return block(i); }
  template <grid_e category = grid_e::collocated> ZS_FUNCTION value_type &cell(channel_counter_type chn, size_type bid, cell_index_type cid) const {
    return grid(wrapv<category>{})(chn, bid, cid);
  }
  template <grid_e category = grid_e::collocated> ZS_FUNCTION value_type &operator()(channel_counter_type chn, size_type bid, cell_index_type cid) const {
    return cell<category>(chn, bid, cid);
  }

  grid_view_t _collocatedGrid, _cellcenteredGrid, _staggeredGrid;
  value_type _dx;
};
template <execspace_e space, class V, int d, int S> GridsView<space, Grids<V, d, S>> proxy(Grids<V, d, S> &g) { return GridsView<space, Grids<V, d, S>>{g}; }
template <execspace_e space, class V, int d, int S> GridsView<space, const Grids<V, d, S>> proxy(const Grids<V, d, S> &g) {
  return GridsView<space, const Grids<V, d, S>>{g};
}

}  // namespace zs
