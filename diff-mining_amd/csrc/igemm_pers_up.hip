// igemm_pers_up.hip — `Upsample2D` (F.interpolate nearest, exact 2x) + its 3x3 convolution as FOUR 2x2 convolutions on the
// low-resolution source, one per output parity class, on the persistent 256 x 320 tile (igemm_pers_tile.h, template parameter UP4):
// 4 k taps instead of 9 (the taps of the 3x3 kernel that read the same source pixel are pre-summed), the up-sampled tensor never
// exists.  The layer behind it: diffusers `Upsample2D.forward` inside `UNet2DConditionModel.up_blocks[0..2]`, reached from
// diffmining/typicality/compute.py:100 and dift.py:191.  Own translation unit (the other instantiations do not change by a byte).
#define DM_IGEMM_PERS_UP 1
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_up4(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_up4_t(p, s); }

}  // namespace dm
