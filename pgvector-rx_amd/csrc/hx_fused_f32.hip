// hx_fused_f32.hip -- k_fused instantiations for one element type (see hx_fused_kernel.h).
#include "hx_fused_kernel.h"

hipError_t hx_launch_fused_f32(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    switch (metric) {
    case HX_L2SQ: return launch_fused_mode<OpF32<K_L2>>(e, p, grid, lds, mode);
    case HX_NEG_IP: return launch_fused_mode<OpF32<K_IP>>(e, p, grid, lds, mode);
    default: return launch_fused_mode<OpF32<K_L1>>(e, p, grid, lds, mode);
    }
}
