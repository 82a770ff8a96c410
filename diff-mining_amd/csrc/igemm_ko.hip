// igemm_ko.hip — the 128-row implicit-GEMM tile with the (dy, 64-channel slab, dx) k order of the tap-reuse layers (igemm_tile.h,
// template parameter KO): the partner of igemm_pers_tr.hip for launches / row tails the persistent tile does not take.
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm_tile_ko(const IGemmParams& p, hipStream_t s) {
    if (!igemm_ko_layer(p) || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    return (p.Cout % 320 == 0) ? launch_t<4, 5, false, false, false, true>(p, s) : launch_t<2, 5, false, false, false, true>(p, s);
}

}  // namespace dm
