// igemm_pers_ln.hip — LayerNorm-folded instantiations of the persistent 256 x 320 tile (igemm_pers_tile.h).
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_ln(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_t<true>(p, s); }

#ifdef DM_IGEMM_TIMING
extern "C" int dm_debug_pers_ln_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pers_dbg), sizeof(long long) * 8) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace dm
