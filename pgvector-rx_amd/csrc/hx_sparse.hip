// hx_sparse.hip -- sparsevec distances for the DistanceFn seam (src/graph/mod.rs:144-145) and their normalisation.
//
// A sparsevec distance is a merge join of two ascending index lists (src/types/sparsevec.rs:873-950, 1038-1088): irregular, short (an indexed
// sparsevec holds at most 1000 non-zero elements) and order-sensitive -- the reference adds its f32 terms in merged-index order.  The kernels below
// keep exactly that order: ONE LANE walks one pair with the reference's own loop, and a wavefront works on 64 pairs at a time.  They read the
// request arrays of K1 / K2 (hx_engine.hip), so hx_distances / hx_distances_batch / hx_pairwise / hx_pairwise_many and with them the lock-step
// graph driver serve the type unchanged.  Row record: { int32 nnz; int32 pad[3]; int32 index[cap]; float value[cap] }, cap = min(dim, 1000).
// Built with -ffp-contract=off like every float kernel of the engine: mul and add are rounded separately, as in the reference's unfused Rust.
#include "hx_ops.h"

namespace {

// query-vs-rows: one wavefront per expansion group, the query record parked in LDS, lane l walks rows l, l + 64, ...
template <int KIND>
__global__ void __launch_bounds__(64)
k_sparse_groups(const uint8_t *__restrict__ rows, const uint8_t *__restrict__ queries, uint32_t pitch, uint32_t cap,
                const uint32_t *__restrict__ grp_q, const uint32_t *__restrict__ grp_off, const uint32_t *__restrict__ ids, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t q_lds[];
    const uint32_t g = blockIdx.x;
    const uint32_t beg = grp_off[g], n = grp_off[g + 1] - beg;
    if (n == 0) return;
    const uint32_t qsel = grp_q[g];
    const uint8_t *qsrc = (qsel & HX_QUERY_SLOT) ? queries + (size_t)(qsel & 0x7fffffffu) * pitch : rows + (size_t)qsel * pitch;
    for (uint32_t o = threadIdx.x * 16u; o < pitch; o += 64u * 16u) *(u4 *)(q_lds + o) = *(const u4 *)(qsrc + o);
    __syncthreads();
    const SpRow q = sp_row((const uint8_t *)q_lds, cap);
    for (uint32_t r = threadIdx.x; r < n; r += 64u) {
        const SpRow b = sp_row(rows + (size_t)ids[beg + r] * pitch, cap);
        out[beg + r] = sp_merge<KIND>(q, b);                  // build_callback / scan: distance(query, element), graph/mod.rs:221
    }
}

// pair blocks: the workgroup table of K2 (group, first pair of the slab); thread t walks pairs p0 + t, p0 + t + 512 of the slab
template <int KIND>
__global__ void __launch_bounds__(HX_PAIR_WG)
k_sparse_pairs(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t cap,
               const uint32_t *__restrict__ pg_off, const uint16_t *__restrict__ pg_na, const uint16_t *__restrict__ pg_nb,
               const uint32_t *__restrict__ pids, const uint64_t *__restrict__ pg_out_off, const uint32_t *__restrict__ wg_tab, float *__restrict__ out)
{
    const uint32_t g = wg_tab[2 * blockIdx.x], p0 = wg_tab[2 * blockIdx.x + 1];
    const uint32_t na = pg_na[g], nb = pg_nb[g];
    const uint32_t *ids = pids + pg_off[g];
    const uint32_t P = nb ? na * nb : na * (na - 1) / 2;
    const uint32_t end = p0 + HX_PAIR_SLAB < P ? p0 + HX_PAIR_SLAB : P;
    for (uint32_t p = p0 + threadIdx.x; p < end; p += HX_PAIR_WG) {
        uint32_t i, j;
        if (nb) { i = p / nb; j = na + p % nb; } else tri_decode(p, i, j);
        const SpRow a = sp_row(rows + (size_t)ids[i] * pitch, cap), b = sp_row(rows + (size_t)ids[j] * pitch, cap);
        out[pg_out_off[g] + p] = sp_merge<KIND>(a, b);
    }
}

// sparsevec_l2_normalize_raw (sparsevec.rs:1123-1178): f64 norm, every value divided in f64 and rounded to f32, zeros dropped
__global__ void k_sparse_normalize(uint8_t *rows, uint32_t pitch, uint32_t cap, uint64_t n, double *norms)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    uint8_t *rec = rows + r * pitch;
    int32_t *idx = (int32_t *)(rec + 16); float *val = (float *)(rec + 16 + (size_t)cap * 4);
    const int nnz = *(const int32_t *)rec;
    double norm = 0.0;
    for (int i = 0; i < nnz; i++) { const double v = (double)val[i]; norm += v * v; }
    norm = sqrt(norm);
    if (norms) norms[r] = norm;
    int k = 0;
    if (norm > 0.0)
        for (int i = 0; i < nnz; i++) {
            const float v = (float)((double)val[i] / norm);
            if (v != 0.0f) { idx[k] = idx[i]; val[k] = v; k++; }
        }
    for (int i = k; i < nnz; i++) { idx[i] = 0; val[i] = 0.0f; }       // unused slots stay zero: rows are compared bytewise
    *(int32_t *)rec = k;
}

}  // namespace

static inline uint32_t sparse_cap(const hx_engine *e) { return (uint32_t)(e->dim < HX_SPARSE_MAX_NNZ ? e->dim : HX_SPARSE_MAX_NNZ); }

hipError_t hx_launch_sparse_dist(hx_engine *e, uint32_t n_groups)
{
    const HxChannel &c = e->ch;
    const uint32_t pitch = (uint32_t)e->pitch, cap = sparse_cap(e);
#define SPD(K) hipLaunchKernelGGL((k_sparse_groups<K>), dim3(n_groups), dim3(64), e->pitch, e->stream, e->d_rows, e->d_queries, pitch, cap, c.d_grp_q, c.d_grp_off, c.d_ids, c.d_out)
    if (e->metric == HX_L2SQ) SPD(K_L2); else if (e->metric == HX_NEG_IP) SPD(K_IP); else SPD(K_L1);
#undef SPD
    return hipGetLastError();
}

hipError_t hx_launch_sparse_pairs(hx_engine *e, uint32_t n_wgs)
{
    const HxChannel &c = e->ch;
    const uint32_t pitch = (uint32_t)e->pitch, cap = sparse_cap(e);
#define SPP(K) hipLaunchKernelGGL((k_sparse_pairs<K>), dim3(n_wgs), dim3(HX_PAIR_WG), 0, e->stream, e->d_rows, pitch, cap, c.d_pg_off, c.d_pg_na, c.d_pg_nb, c.d_pids, c.d_pg_out_off, c.d_wg_tab, c.d_pout)
    if (e->metric == HX_L2SQ) SPP(K_L2); else if (e->metric == HX_NEG_IP) SPP(K_IP); else SPP(K_L1);
#undef SPP
    return hipGetLastError();
}

hipError_t hx_launch_sparse_normalize(hx_engine *e, uint8_t *base, uint64_t n, double *d_norms)
{
    hipLaunchKernelGGL(k_sparse_normalize, dim3((uint32_t)((n + 63) / 64)), dim3(64), 0, e->stream, base, (uint32_t)e->pitch, sparse_cap(e), n, d_norms);
    return hipGetLastError();
}
