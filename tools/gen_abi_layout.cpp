/* tools/gen_abi_layout.cpp -- TEST INFRASTRUCTURE.  Compiled IN PLACE against the reference's py_interop headers (build container
 * only; recipe: tools/gen_abi_layout.sh) to derive the byte layout of every POD the reference's C ABI hands across the boundary
 * by value or by pointer: VectorViewLite, TileVectorViewLite, TileVectorNamedViewLite, BhtViewLite (dim 1-4, B 16/32) and
 * aosoa_iterator_port.  Output = tests/golden/abi_layout.json, which tests/test_host_cpu.py compares with the layout of the
 * structs include/zs_rocm.h declares (a JIT kernel compiled against the reference's view headers reads these objects directly). */
#include "zensim/zpc_tpls/fmt/format.h"
#include "zensim/math/Hash.hpp"
#include "zensim/py_interop/HashUtils.hpp"
#include "zensim/math/bit/Bits.h"
#include "zensim/py_interop/VectorView.hpp"
#include "zensim/py_interop/TileVectorView.hpp"
#include "zensim/py_interop/BhtView.hpp"
#include "zensim/py_interop/GenericIterator.hpp"
#include <cstddef>
#include <cstdio>
using namespace zs;
static bool first_struct = true;
#define BEGIN(name, T)                                                              \
  {                                                                                 \
    using S = T;                                                                    \
    printf("%s\n  \"%s\": {\"size\": %zu, \"align\": %zu, \"members\": {", first_struct ? "" : ",", name, sizeof(S), alignof(S)); \
    first_struct = false;                                                           \
    bool first = true;                                                              \
    S obj{};
#define MEM(label, expr)                                                            \
    printf("%s\"%s\": [%zu, %zu]", first ? "" : ", ", label, (size_t)((const char *)&(obj.expr) - (const char *)&obj), sizeof(obj.expr)); \
    first = false;
#define END() \
    printf("}}"); \
  }
template <int dim, int B> static void bht(const char *name) {
  using V = BhtViewLite<int, dim, int, B>;
  BEGIN(name, V)
  MEM("keys", _table.keys._vector)
  MEM("indices", _table.indices._vector)
  MEM("status", _table.status._vector)
  MEM("activeKeys", _activeKeys._vector)
  MEM("cnt", _cnt._vector)
  MEM("success", _success._vector)
  MEM("tableSize", _tableSize)
  MEM("numBuckets", _numBuckets)
  MEM("hf0x", _hf0._hashx)
  MEM("hf0y", _hf0._hashy)
  MEM("hf1x", _hf1._hashx)
  MEM("hf1y", _hf1._hashy)
  MEM("hf2x", _hf2._hashx)
  MEM("hf2y", _hf2._hashy)
  END()
}
using TVL = TileVectorViewLite<float, 32>;
using TVNL = TileVectorNamedViewLite<float, 32>;
using PortF1 = aosoa_iterator_port<float, 1>;
using PortF3 = aosoa_iterator_port<float, 3>;
using PortCD1 = aosoa_iterator_port<const double, 1>;
int main() {
  printf("{");
  BEGIN("VectorViewLite<int>", VectorViewLite<int>)
  MEM("_vector", _vector)
  END()
  BEGIN("TileVectorViewLite<float,32>", TVL)
  MEM("_vector", _vector)
  MEM("_numChannels", _numChannels)
  END()
  BEGIN("TileVectorNamedViewLite<float,32>", TVNL)
  MEM("_vector", _vector)
  MEM("_numChannels", _numChannels)
  MEM("_tagNames", _tagNames)
  MEM("_tagOffsets", _tagOffsets)
  MEM("_tagSizes", _tagSizes)
  MEM("_N", _N)
  END()
  bht<1, 16>("BhtViewLite<int,1,int,16>");
  bht<2, 16>("BhtViewLite<int,2,int,16>");
  bht<3, 16>("BhtViewLite<int,3,int,16>");
  bht<4, 16>("BhtViewLite<int,4,int,16>");
  bht<1, 32>("BhtViewLite<int,1,int,32>");
  bht<2, 32>("BhtViewLite<int,2,int,32>");
  bht<3, 32>("BhtViewLite<int,3,int,32>");
  bht<4, 32>("BhtViewLite<int,4,int,32>");
  BEGIN("aosoa_iterator_port<float,1>", PortF1)
  MEM("base", base)
  MEM("idx", idx)
  MEM("numTileBits", numTileBits)
  MEM("tileMask", tileMask)
  MEM("numChns", numChns)
  END()
  BEGIN("aosoa_iterator_port<float,3>", PortF3)
  MEM("base", base)
  MEM("idx", idx)
  MEM("numTileBits", numTileBits)
  MEM("tileMask", tileMask)
  MEM("numChns", numChns)
  END()
  BEGIN("aosoa_iterator_port<const double,1>", PortCD1)
  MEM("base", base)
  MEM("idx", idx)
  MEM("numTileBits", numTileBits)
  MEM("tileMask", tileMask)
  MEM("numChns", numChns)
  END()
  printf("\n}\n");
  return 0;
}
