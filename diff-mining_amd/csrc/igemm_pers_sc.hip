// igemm_pers_sc.hip — the persistent 256 x 320 tile with a ResNet block's `conv_shortcut` (1x1 on the block's, possibly
// concatenated, input) folded into its conv2 as extra k steps (igemm_pers_tile.h, template parameter SC).  Own translation unit.
#define DM_IGEMM_PERS_SC 1
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_sc(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_sc_t(p, s); }

}  // namespace dm
