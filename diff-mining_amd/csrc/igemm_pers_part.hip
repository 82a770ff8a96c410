// igemm_pers_part.hip — the split-K instantiation of the persistent 256 x 320 implicit-GEMM tile (igemm_pers_tile.h,
// EPI_PARTIAL) for the 3x3 convolutions of the 8x8 level (M = 10 240 rows at the bench batch = 160 tiles for 256 CUs):
// three k parts of three taps each give 480 (tile, part) units = two rounds of 60 k steps instead of one round of 180
// on 62 % of the chip, at the big tile's 13.8 LDS-DMA bytes per kMAC; the fp32 partials go through the reduction kernel
// of igemm_splitk.hip.  Own translation unit, like every instantiation of that header.
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_partial(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_partial_t(p, s); }

}  // namespace dm
