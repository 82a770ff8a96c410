// igemm64.hip — the implicit-GEMM tile kernel instantiated with 64-channel waves (tiles of
// 128 px x 128 or 256 ch) for the SDv1.5 VAE encoder's channel counts 128 / 256 / 512, which the
// U-Net's 80-channel waves do not divide (`vae.encode`, diffmining/typicality/compute.py:91-93).
// Adds the `Downsample2D(padding=0)` conv mode: F.pad(x, (0,1,0,1)) + 3x3 stride 2.
#include "igemm_tile.h"

namespace dm {

hipError_t launch_igemm64(const IGemmParams& p, hipStream_t s) {
    if (p.Cout % 128 != 0 || p.Cin % BK != 0 || p.C1 % BK != 0 || p.M <= 0 || p.epi != EPI_PLAIN) return hipErrorInvalidValue;
    return (p.Cout % 256 == 0) ? launch_t<4, 4>(p, s) : launch_t<2, 4>(p, s);
}

}  // namespace dm
