// igemm_ws.hip — the 128-row implicit-GEMM tile with per-sample weights and bias row (igemm_tile.h, template parameter WS): the
// partner of igemm_pers_ws.hip for launches the persistent tile does not take (few tiles).  Own translation unit.
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm_tile_ws(const IGemmParams& p, hipStream_t s) {
    if (p.mode != IG_DENSE || p.epi != EPI_PLAIN || !p.ln_t || p.w_sample_stride <= 0 || p.rows_per_sample <= 0 ||
        p.rows_per_sample % 128 != 0 || p.M % p.rows_per_sample != 0 || p.Cout % 160 != 0 || p.Cin % BK != 0 || p.res || p.temb)
        return hipErrorInvalidValue;
    return (p.Cout % 320 == 0) ? launch_t<4, 5, true, true>(p, s) : launch_t<2, 5, true, true>(p, s);
}

}  // namespace dm
