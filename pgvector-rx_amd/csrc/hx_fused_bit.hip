// hx_fused_bit.hip -- k_fused instantiations for one element type (see hx_fused_kernel.h).
#include "hx_fused_kernel.h"

hipError_t hx_launch_fused_bit(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    if (metric == HX_HAMMING) return launch_fused_mode<OpHamming>(e, p, grid, lds, mode);
    return launch_fused_mode<OpJaccard>(e, p, grid, lds, mode);
}
