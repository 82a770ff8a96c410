// igemm_pers_ws.hip — the persistent 256 x 320 tile with per-sample weights and bias row (igemm_pers_tile.h, template
// parameter WS): `Transformer2DModel.norm` (GroupNorm, no activation) folded into `proj_in` (1x1 convolution) — the GEMM runs
// on the raw residual stream, the normalised tensor is never written or read.  Own translation unit.
#define DM_IGEMM_PERS_WS 1
#include "igemm_pers_tile.h"

namespace dm {

hipError_t launch_igemm_pers_ws(const IGemmParams& p, hipStream_t s) { return launch_igemm_pers_ws_t(p, s); }

}  // namespace dm
