// igemm_ln.hip — the 128 px x 320 / 160 ch implicit-GEMM tiles with LayerNorm folded into the epilogue
// (transformer blocks: LN -> Linear), see IGemmParams::ln_stats in dm_kernels.h.  Own translation unit
// (co-compiling instantiations perturbs the main kernel's register allocation).
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm_tile_ln(const IGemmParams& p, hipStream_t s) {
    if (p.Cout % 160 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    return (p.Cout % 320 == 0) ? launch_t<4, 5, true>(p, s) : launch_t<2, 5, true>(p, s);
}

// A/B: the 128 x 160 tile (4 waves, two blocks per CU) whatever Cout is
hipError_t launch_igemm_tile_ln_half(const IGemmParams& p, hipStream_t s) {
    if (p.Cout % 160 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0) return hipErrorInvalidValue;
    return launch_t<2, 5, true>(p, s);
}

}  // namespace dm
