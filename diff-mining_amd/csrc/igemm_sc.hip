// igemm_sc.hip — the 128-row implicit-GEMM tile with a ResNet block's conv_shortcut folded into its conv2 (igemm_tile.h,
// template parameter SC): the partner of igemm_pers_sc.hip for launches / row tails the persistent tile does not take.
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm_tile_sc(const IGemmParams& p, hipStream_t s) {
    if ((p.mode != IG_CONV3 && p.mode != IG_DENSE) || p.epi != EPI_PLAIN || p.ln_s || p.temb || p.X2 || !p.X3 || p.Csc <= 0 || p.Csc % BK || p.C3 % BK ||
        (p.C3 < p.Csc && !p.X4) || p.Cout % 160 != 0 || p.Cin % BK != 0 || p.M <= 0 || p.ksplit > 1) return hipErrorInvalidValue;
    return (p.Cout % 320 == 0) ? launch_t<4, 5, false, false, true>(p, s) : launch_t<2, 5, false, false, true>(p, s);
}

}  // namespace dm
