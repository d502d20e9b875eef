/*
 * tilevector.c -- CPU restatement of the TileVector<T, L> AoSoA addressing.
 * TEST INFRASTRUCTURE ONLY (see zpc_oracle.h).
 */
#include "zpc_oracle.h"

/* container/TileVector.hpp:73-74  count_tiles(n) = (n + L - 1) / L */
size_t orc_tv_num_tiles(size_t n, size_t L) { return (n + L - 1) / L; }

/* container/TileVector.hpp:108,397,768-769: (i / L * numChannels + chn) * L + i % L */
size_t orc_tv_offset(size_t i, size_t chn, size_t L, size_t C) { return (i / L * C + chn) * L + i % L; }

/* py_interop/GenericIterator.hpp:88-93: ((idx >> bits) * numChns << bits) | (idx & mask) */
size_t orc_aosoa_offset(uint32_t idx, uint32_t numTileBits, uint32_t tileMask, uint32_t numChns) {
  return (size_t)((((idx >> numTileBits) * numChns) << numTileBits) | (idx & tileMask));
}

void orc_tv_from_aos_f32(const float *aos, size_t n, size_t C, size_t L, float *tv) {
  for (size_t i = 0; i < n; ++i)
    for (size_t c = 0; c < C; ++c) tv[orc_tv_offset(i, c, L, C)] = aos[i * C + c];
}
void orc_tv_to_aos_f32(const float *tv, size_t n, size_t C, size_t L, float *aos) {
  for (size_t i = 0; i < n; ++i)
    for (size_t c = 0; c < C; ++c) aos[i * C + c] = tv[orc_tv_offset(i, c, L, C)];
}
