// igemm_pers.hip — plain instantiations of the persistent 256 px x 320 ch implicit-GEMM tile (igemm_pers_tile.h).
// Own translation unit like the other tile kernels (co-compiling large kernels perturbs their register allocation).
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_t<false>(p, s); }

#ifdef DM_IGEMM_TIMING
extern "C" int dm_debug_pers_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pers_dbg), sizeof(long long) * 8) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace dm
