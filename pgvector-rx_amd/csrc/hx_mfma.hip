// hx_mfma.hip -- the batched-build distance GEMM on the matrix cores (SURVEY 8 row g, BASELINE configs[3]: halfvec inner product).
//
// The operand of select_neighbors / check_element_closer (src/graph/mod.rs:284-297, 324-336) is a block of pairwise inner products among
// <= 64 candidate rows: G = X . X^T, a true f16 GEMM (halfvec.rs:687-733 arithmetic: f16 inputs, f32 accumulation).  k_pair_mfma_f16
// computes the pair groups of hx_pairwise_many with v_mfma_f32_32x32x16_f16: one wavefront per 32 x 32 tile of a group, operands straight
// from HBM/L2 into registers (the k index of an MFMA step may be ANY permutation as long as A and B use the same one, so lane (r, h) reads a
// contiguous 64-byte piece of its row per super-step = four MFMA steps: full cache lines per lane, no LDS staging, no transposes).
//
// Products of two halves are exact in f32, so an MFMA value differs from the canonical-order value (hx_engine.hip) only by the order of the
// f32 additions: |g - c| <= 2 K 2^-24 sum_k |a_k b_k| <= 2 K 2^-24 |a| |b|.  The graph driver uses MFMA values only for decisions
// `d(e, r) <= d(e, q)` (mod.rs:333) that fall outside that band and re-evaluates the others in the canonical order, which keeps graphs
// bit-identical to the oracle's (hx_index.cpp, SelectTask).  k_row_norm2 provides |a|^2 per row for the band.
#include "hx_ops.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

namespace {

// one 256-thread workgroup per group; wave w computes tile w of the group (lower triangle: (0,0) (1,0) (1,1); rectangle: A-block x B-block)
__global__ void __launch_bounds__(256, 2)
k_pair_mfma_f16(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t dim,
                const uint32_t *__restrict__ pg_off, const uint16_t *__restrict__ pg_na, const uint16_t *__restrict__ pg_nb,
                const uint32_t *__restrict__ pids, const uint64_t *__restrict__ pg_out_off, const uint32_t *__restrict__ glist,
                float *__restrict__ out)
{
    const uint32_t g = glist[blockIdx.x];
    const uint32_t na = pg_na[g], nb = pg_nb[g];
    const uint32_t *ids = pids + pg_off[g];
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    uint32_t ti, tj;                                              // tile: A rows [32 ti, 32 ti + 32), B rows [32 tj, 32 tj + 32)
    const uint32_t ab = (na + 31u) / 32u, bb = nb ? (nb + 31u) / 32u : ab;
    if (nb == 0u) { ti = wave == 0u ? 0u : 1u; tj = wave == 2u ? 1u : 0u; if (wave > 2u || ti >= ab) return; }
    else { ti = wave / bb; tj = wave % bb; if (ti >= ab) return; }
    const uint32_t r = lane & 31u, h = lane >> 5;
    const uint32_t ia = ti * 32u + r, ib = tj * 32u + r;
    const uint32_t a_id = ids[ia < na ? ia : 0u];
    const uint32_t b_id = nb ? ids[na + (ib < nb ? ib : 0u)] : ids[ib < na ? ib : 0u];
    const uint8_t *pa = rows + (size_t)a_id * pitch, *pb = rows + (size_t)b_id * pitch;
    float16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    const uint32_t nsteps = (dim + 63u) / 64u;                    // super-steps of 64 halves: lane half h owns bytes [128 t + 64 h, + 64)
    const uint32_t nfull = pitch / 128u;                          // super-steps that lie inside the row for every lane
    // loads are issued unconditionally (a lane past the row re-reads its first bytes) and masked only where they are consumed, in the one
    // ragged step: a select right behind a load would make the wave wait for it and undo the prefetch
    auto load = [&](uint32_t t, u4 (&a)[4], u4 (&b)[4]) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t off = t * 128u + h * 64u + (uint32_t)q * 16u;
            const uint32_t o = off < pitch ? off : 0u;
            a[q] = *(const u4 *)(pa + o); b[q] = *(const u4 *)(pb + o);
        }
    };
    auto mma = [&](uint32_t t, u4 (&a)[4], u4 (&b)[4]) {
        if (t >= nfull) {
#pragma unroll
            for (int q = 0; q < 4; q++) if (t * 128u + h * 64u + (uint32_t)q * 16u >= pitch) { a[q] = u4{0u, 0u, 0u, 0u}; b[q] = u4{0u, 0u, 0u, 0u}; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a[q]), __builtin_bit_cast(half8, b[q]), acc, 0, 0, 0);
    };
    // three register sets: two super-steps of loads in flight behind the one being multiplied
    u4 a0[4], b0[4], a1[4], b1[4], a2[4], b2[4];
    load(0u, a0, b0);
    if (nsteps > 1u) load(1u, a1, b1);
    for (uint32_t t = 0; t < nsteps; t += 3u) {
        if (t + 2u < nsteps) load(t + 2u, a2, b2);
        mma(t, a0, b0);
        if (t + 1u < nsteps) {
            if (t + 3u < nsteps) load(t + 3u, a0, b0);
            mma(t + 1u, a1, b1);
        }
        if (t + 2u < nsteps) {
            if (t + 4u < nsteps) load(t + 4u, a1, b1);
            mma(t + 2u, a2, b2);
        }
    }
    // C/D layout of the 32x32 forms: column = lane & 31 (B row), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A row)
    const uint64_t ob = pg_out_off[g];
    const uint32_t J = tj * 32u + r;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const uint32_t I = ti * 32u + (uint32_t)(reg & 3) + 8u * (uint32_t)(reg >> 2) + 4u * h;
        const float v = -acc[reg];                                // negative inner product, halfvec.rs:786-791
        if (nb == 0u) { if (I < na && J < I) out[ob + (uint64_t)I * (I - 1u) / 2u + J] = v; }
        else if (I < na && J < nb) out[ob + (uint64_t)I * nb + J] = v;
    }
}

// |row|^2 in f32 (f64 accumulation, rounded up): the Cauchy-Schwarz factor of the MFMA band
__global__ void k_row_norm2_f16(const uint8_t *__restrict__ rows, uint32_t pitch, uint32_t dim, uint64_t first, uint64_t n, float *__restrict__ norm2)
{
    const uint64_t i = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); const uint32_t lane = threadIdx.x & 63u;
    if (i >= n) return;
    const unsigned short *r = (const unsigned short *)(rows + (first + i) * pitch);
    double s = 0.0;
    for (uint32_t k = lane; k < dim; k += 64u) { const double v = (double)half2f(r[k]); s += v * v; }
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) norm2[first + i] = (float)(s * (1.0 + 1e-6));
}

}  // namespace

// launches the MFMA kernel over the groups listed in d_glist (indices into the channel's pair-group arrays, already on the device)
hipError_t hx_launch_pair_mfma(hx_engine *e, uint32_t n_groups, const uint32_t *d_glist)
{
    const HxChannel &c = e->ch;
    hipLaunchKernelGGL(k_pair_mfma_f16, dim3(n_groups), dim3(256), 0, e->stream, e->d_rows, (uint32_t)e->pitch, (uint32_t)e->dim,
                       c.d_pg_off, c.d_pg_na, c.d_pg_nb, c.d_pids, c.d_pg_out_off, d_glist, c.d_pout);
    return hipGetLastError();
}

// |row|^2 of rows [first, first + n) into the engine's norm array (device + host copy); halfvec only
int hx_engine::mfma_norms(uint64_t upto)
{
    if (dtype != HX_F16) return fail(HX_E_ARG, "the MFMA pair path serves halfvec rows");
    if (upto > n_rows) upto = n_rows;
    if (upto <= mf_norm_rows) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    if (capacity > mf_cap) {
        float *nd = nullptr;
        HX_HIP(this, hipMalloc((void **)&nd, capacity * sizeof(float)));
        if (d_mf_norm2 && mf_norm_rows) HX_HIP(this, hipMemcpyAsync(nd, d_mf_norm2, mf_norm_rows * sizeof(float), hipMemcpyDeviceToDevice, stream));
        HX_HIP(this, hipStreamSynchronize(stream));
        if (d_mf_norm2) (void)hipFree(d_mf_norm2);
        d_mf_norm2 = nd; mf_cap = capacity;
    }
    const uint64_t first = mf_norm_rows, n = upto - first;
    hipLaunchKernelGGL(k_row_norm2_f16, dim3((uint32_t)((n + 3) / 4)), dim3(256), 0, stream, d_rows, (uint32_t)pitch, (uint32_t)dim, first, n, d_mf_norm2);
    HX_HIP(this, hipGetLastError());
    h_mf_norm2.resize(upto);
    HX_HIP(this, hipMemcpyAsync(h_mf_norm2.data() + first, d_mf_norm2 + first, n * sizeof(float), hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    mf_norm_rows = upto;
    return HX_OK;
}
