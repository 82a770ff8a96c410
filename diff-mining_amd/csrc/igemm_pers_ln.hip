// igemm_pers_ln.hip — LayerNorm-folded instantiations of the persistent 256 x 320 tile (igemm_pers_tile.h).
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_ln(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_t<true>(p, s); }

}  // namespace dm
