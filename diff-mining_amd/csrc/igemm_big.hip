// igemm_big.hip — plain instantiations of the 256 px x 320 ch implicit-GEMM tile (igemm_big_tile.h).
// Kept in its own translation unit: co-compiling it with the 128x320 kernel cost that kernel ~5 %
// (register allocation / scheduling drift).
#include "igemm_big_tile.h"

namespace dm {

hipError_t launch_igemm_big(const IGemmParams& p, hipStream_t s) { return launch_igemm_big_t<false>(p, s); }

#ifdef DM_IGEMM_TIMING
extern "C" int dm_debug_igemm_timing(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_igemm_dbg), sizeof(long long) * 16) == hipSuccess ? 0 : 1;
}
#endif

}  // namespace dm
