// rc_null_t.hip -- k_tiled_dp<KT, SEM> for the tile sizes 12..15 (see rc_null_tiled.h)
#include "rc_null_tiled.h"

namespace rc {

size_t null_tiled_codes_bytes(int NK, int KT, int L) { return TiledLayout(NK, KT).codes_bytes(L); }
size_t null_tiled_state_bytes(int L) { return tiled_state_bytes(L); }

template <int KT> static int occ_rec(int kt, size_t lds) {
  if constexpr (KT > kTiledMaxKT) return 0;
  else {
    if (kt != KT) return occ_rec<KT + 1>(kt, lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_tiled_dp<KT, false>, 64, lds) != hipSuccess) return 0;
    return nb;
  }
}
int tiled_dp_occupancy(int KT, size_t ldsBytes) { return occ_rec<kTiledMinKT>(KT, ldsBytes); }

template <int KT> static bool launch_rec(int kt, bool nanSem, const NullArgs &a, int grid, size_t lds, uint8_t *scratch, hipStream_t st) {
  if constexpr (KT > kTiledMaxKT) return false;
  else {
    if (kt != KT) return launch_rec<KT + 1>(kt, nanSem, a, grid, lds, scratch, st);
    if (nanSem) hipLaunchKernelGGL((k_tiled_dp<KT, true>), dim3(grid), dim3(64), lds, st, a, a.blob, a.dblocks, a.classBlocks, a.flags, scratch, a.maxima);
    else hipLaunchKernelGGL((k_tiled_dp<KT, false>), dim3(grid), dim3(64), lds, st, a, a.blob, a.dblocks, a.classBlocks, a.flags, scratch, a.maxima);
    return true;
  }
}
bool launch_tiled_dp(int KT, bool nanSem, const NullArgs &a, int grid, size_t ldsBytes, uint8_t *scratchBytes, hipStream_t stream) {
  return launch_rec<kTiledMinKT>(KT, nanSem, a, grid, ldsBytes, scratchBytes, stream);
}

}  // namespace rc
