// bht.hpp -- host-side object of zs::bht<int, dim, int, B> (dim 1-4, B = 16 or 32); the device view and the probe/insert protocol live in
// include/zensim_rocm/bht_device.hpp (shared with the header-only C++ face).
#pragma once
#include "common.hpp"
#include "../../include/zensim_rocm/bht_device.hpp"

namespace zsr {

// host-side object shared by bht.hip and mpm.hip
struct BhtHost {
  int dim = 3, memsrc = 1, bucket = BHT_BUCKET;
  int8_t devid = 0;
  size_t tableSize = 0;
  int *keys = nullptr, *indices = nullptr, *status = nullptr, *activeKeys = nullptr, *cnt = nullptr, *success = nullptr;
  unsigned hf[6];
  BhtDev dev() const {
    BhtDev d;
    d.keys = keys; d.indices = indices; d.status = status; d.activeKeys = activeKeys; d.cnt = cnt; d.success = success;
    d.tableSize = (unsigned)tableSize;
    d.bucket = (unsigned)bucket;
    d.numBuckets = (unsigned)(tableSize / (size_t)bucket);
    for (int i = 0; i < 6; ++i) d.hf[i] = hf[i];
    return d;
  }
};
int bht_size(const BhtHost &t, hipStream_t s);  // device -> host 4-byte copy (Bht.hpp:248)

}  // namespace zsr

struct zs_rocm_bht_1 { zsr::BhtHost t; };
struct zs_rocm_bht_2 { zsr::BhtHost t; };
struct zs_rocm_bht_3 { zsr::BhtHost t; };
struct zs_rocm_bht_4 { zsr::BhtHost t; };
