// hx_fused_sparse.hip -- k_fused instantiations for sparsevec rows (see hx_fused_kernel.h; the distance path is f_dist_batch's OpSparse branch).
// Scans (MODE 0, 2) and search-only inserts (MODE 3: builds and aminsert; select_neighbors and the back-links of a build follow in hx_biglist.hip's
// list kernels, aminsert's back-connections in hx_links.hip's k_update_runs; hx_index_set_fused(0): the lock-step driver, whose pair kernels walk the merge joins).  One wavefront per search, one LANE per row of an expansion; always the 64-lanes-per-task build.
#include "hx_fused_kernel.h"

template <int KIND>
static hipError_t launch_sparse_mode(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    if (mode == 0) return launch_fused<OpSparse<KIND>, 0, 64>(e, p, grid, lds);
    if (mode == 2) return launch_fused<OpSparse<KIND>, 2, 64>(e, p, grid, lds);
    if (mode == 3) return launch_fused<OpSparse<KIND>, 3, 64>(e, p, grid, lds);
    return hipErrorInvalidValue;
}

hipError_t hx_launch_fused_sparse(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    switch (metric) {
    case HX_L2SQ: return launch_sparse_mode<K_L2>(e, p, grid, lds, mode);
    case HX_NEG_IP: return launch_sparse_mode<K_IP>(e, p, grid, lds, mode);
    default: return launch_sparse_mode<K_L1>(e, p, grid, lds, mode);
    }
}
