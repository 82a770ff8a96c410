// igemm_big_ln.hip — the 256 px x 320 ch implicit-GEMM tile with LayerNorm folded into the epilogue
// (transformer blocks: LN -> Linear / GEGLU projection), see IGemmParams::ln_stats in dm_kernels.h.
#include "igemm_big_tile.h"

namespace dm {

hipError_t launch_igemm_big_ln(const IGemmParams& p, hipStream_t s) { return launch_igemm_big_t<true>(p, s); }

}  // namespace dm
