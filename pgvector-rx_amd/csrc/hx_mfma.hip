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
#include "hx_fused_core.h"

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


// =================================================================================================
// select_neighbors on the matrix cores for the device-resident build (SURVEY 8 row g on the default placement).
//
// k_fused MODE 3 leaves, per (member, layer) PROBLEM, the sorted result set W of the layer's search: n <= ef_construction candidates.
// select_neighbors (src/graph/mod.rs:269-308) then asks for d(e, r) between candidates only (mod.rs:324-336): entries of the n x n Gram matrix of
// the candidates' rows -- for halfvec inner product a true f16 GEMM X . X^T with K = dim (halfvec.rs:687-733: f16 inputs, f32 accumulation),
// W^2 d / (W d 2 B) = W/2 flop per byte of rows: ~100 flop/B at W = 200 (the lower triangle alone: 143 flop/B of HBM traffic, below the f16 ridge
// of ~310 flop/B, so the kernel is bound by streaming the W rows once -- 1.6 MB per problem at halfvec(4000)).
//
//   k_wgemm_f16   one 512-thread workgroup per problem: the n rows stream through LDS in K-chunks of 64 halves (128 B per row per chunk, fetched by
//                 LDS-DMA with a per-lane source address: one wave instruction fills 8 rows), a ring of three chunk images (two in flight); the lower-triangular 32 x 32 tiles
//                 (<= 36 at n <= 256) are dealt to the 8 waves in row-major runs (consecutive tiles share their A rows), v_mfma_f32_32x32x16_f16,
//                 accumulators in registers for the whole K loop; -acc goes to G[problem][i (i - 1) / 2 + j], j < i.
//                 LDS image of a chunk: row r at r * 128 B, its 16-byte piece kp in slot kp ^ ((r >> 1) & 7): the 16 lanes that one ds_read_b128
//                 phase serves then touch 16 different 16-byte columns of the 256-byte bank row (conflict-free), and the DMA side needs no padding
//                 (each lane simply fetches the piece that belongs in ITS slot).
//   k_wselect     one wavefront per problem replays the heuristic on G.  An MFMA value differs from the canonical-order value (hx_fused_core.h) only
//                 by the order of the f32 additions, |g - c| <= band = 2 K 2^-24 |a| |b|; a decision `d(e, r) <= d(e, q)` (mod.rs:333) whose two sides
//                 are further apart than the band is taken from G, the others are re-evaluated in the canonical order (the candidate parked in LDS,
//                 the partners streamed: f_dist_batch, the traversal kernel's own code) -- lists and distance bits equal the oracle's.
// =================================================================================================
#define WG_NR 256u            /* rows (candidates) a problem may have: ef_construction <= 256 on this path */
#define WG_KC 128u            /* bytes of each row per K-chunk (64 halves = four MFMA k-steps) */
#define WG_TPW 5              /* tiles per wave: 36 lower-triangular tiles / 8 waves */

struct WgParams {
    const uint8_t *rows; uint32_t pitch, ef, m;
    const uint2 *wl; const uint32_t *wl_cnt; const uint8_t *prob_layer; uint32_t n_prob;
    float *G; uint64_t g_stride;                 // floats per problem: ef (ef - 1) / 2
    uint32_t stage_bytes;                        // LDS bytes of one K-chunk image: round_up(ef, 8) rows x WG_KC (tile rows past it read the next stage / the pad: never stored)
};

__global__ void __launch_bounds__(512, 2)
k_wgemm_f16(const WgParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // a ring of three chunk images of stage_bytes each + a pad of 31 rows
    const uint32_t pr = blockIdx.x;
    const uint32_t n = p.wl_cnt[pr];
    const uint32_t lm = p.prob_layer[pr] == 0 ? 2u * p.m : p.m;
    if (n <= lm || n > WG_NR) return;                              // all candidates are taken (mod.rs:276-278): nothing to compare
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    const uint2 *W = p.wl + (size_t)pr * p.ef;
    // DMA duty of this wave: row groups (8 rows) wave, wave + 8, ...; lane -> (row in group, slot)
    const uint32_t n_grp = (n + 7u) / 8u;
    const uint32_t my_grps = n_grp > wave ? (n_grp - wave + 7u) / 8u : 0u;   // <= 4
    const uint32_t sub = lane >> 3, slot = lane & 7u;
    const uint8_t *src[4]; uint32_t kpl[4]; bool rowok[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t row = (wave + 8u * (uint32_t)j) * 8u + sub;
        rowok[j] = (uint32_t)j < my_grps && row < n;
        const uint32_t id = rowok[j] ? W[row].y : 0u;
        kpl[j] = slot ^ ((row >> 1) & 7u);                         // the piece that belongs in this lane's slot
        src[j] = p.rows + (size_t)id * p.pitch + kpl[j] * 16u;
    }
    const uint32_t nchunks = (p.pitch + WG_KC - 1u) / WG_KC;
    auto dma = [&](uint32_t c) {
        uint8_t *stage = lds + (c % 3u) * p.stage_bytes;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if ((uint32_t)j < my_grps) {                           // wave-uniform
                uint8_t *dst = stage + (wave + 8u * (uint32_t)j) * 8u * WG_KC;   // this group's 1 KiB
                const bool inrow = c * WG_KC + kpl[j] * 16u < p.pitch;
                if (rowok[j] && inrow)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)(src[j] + (size_t)c * WG_KC),
                                                     (__attribute__((address_space(3))) uint32_t *)dst, 16, 0, 0);
                else if (rowok[j]) *(lds_u4 *)((lds_u8 *)dst + lane * 16u) = u4{0u, 0u, 0u, 0u};   // past the row's end: zeros (only in the last chunk)
            }
        }
    };
    // my tiles: the lower triangle (diagonal included) of the T x T tile grid in row-major order, dealt in runs
    const uint32_t T = (n + 31u) / 32u, ntiles = T * (T + 1u) / 2u;
    const uint32_t per = (ntiles + 7u) / 8u;
    const uint32_t t0 = wave * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    uint32_t ti[WG_TPW], tj[WG_TPW];
    {
        uint32_t a = 0, b = 0, idx = 0;                            // walk the enumeration to t0 (<= 36 steps, scalar)
        while (idx < t0) { if (b == a) { a++; b = 0; } else b++; idx++; }
#pragma unroll
        for (int t = 0; t < WG_TPW; t++) { ti[t] = a; tj[t] = b; if (b == a) { a++; b = 0; } else b++; }
    }
    const uint32_t ntl = t1 > t0 ? t1 - t0 : 0u;
    float16v acc[WG_TPW];
#pragma unroll
    for (int t = 0; t < WG_TPW; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
    const uint32_t r = lane & 31u, h = lane >> 5;
    // Ring of three chunk images: while chunk c is multiplied the requests of chunks c + 1 and c + 2 are in flight (two images = 51 KB per workgroup at
    // ef_construction 200, two workgroups per CU), and one barrier per chunk suffices: a wave that passes barrier(c) has finished chunk c - 1, whose image is
    // the one the requests of chunk c + 2 overwrite.
    dma(0u);
    if (nchunks > 1u) dma(1u);
    for (uint32_t c = 0; c < nchunks; c++) {
        // chunk c has landed for this wave when at most the my_grps requests of chunk c + 1 are outstanding (zero-filling lanes issue LDS stores instead of
        // requests in the last chunk; the count is an upper bound then, which only makes the wait stricter)
        if (c + 1u < nchunks) {
            switch (my_grps) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            }
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                           // every wave's share of chunk c has landed, and every wave is done with chunk c - 1
        if (c + 2u < nchunks) dma(c + 2u);
        const lds_u8 *stage = (const lds_u8 *)(lds + (c % 3u) * p.stage_bytes);
#pragma unroll
        for (uint32_t ks = 0; ks < WG_KC / 32u; ks++) {            // four k-steps of 16 halves: lane half h owns piece 2 ks + h
            const uint32_t kp = 2u * ks + h;
            u4 af = {0u, 0u, 0u, 0u}; uint32_t a_of = 0xffffffffu;
#pragma unroll
            for (int t = 0; t < WG_TPW; t++) {
                if ((uint32_t)t < ntl) {                            // wave-uniform
                    if (ti[t] != a_of) { const uint32_t ra = ti[t] * 32u + r; af = *(const lds_u4 *)(stage + ra * WG_KC + ((kp ^ ((ra >> 1) & 7u)) * 16u)); a_of = ti[t]; }
                    const uint32_t rb = tj[t] * 32u + r;
                    const u4 bf = *(const lds_u4 *)(stage + rb * WG_KC + ((kp ^ ((rb >> 1) & 7u)) * 16u));
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, af), __builtin_bit_cast(half8, bf), acc[t], 0, 0, 0);
                }
            }
        }
    }
    // C/D layout of the 32x32 forms: column = lane & 31 (B row = j), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (A row = i)
    float *G = p.G + (size_t)pr * p.g_stride;
#pragma unroll
    for (int t = 0; t < WG_TPW; t++) {
        if ((uint32_t)t < ntl) {
            const uint32_t J = tj[t] * 32u + r;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const uint32_t I = ti[t] * 32u + (uint32_t)(reg & 3) + 8u * (uint32_t)(reg >> 2) + 4u * h;
                if (I < n && J < I) G[(size_t)I * (I - 1u) / 2u + J] = -acc[t][reg];      // negative inner product, halfvec.rs:786-791
            }
        }
    }
}

struct WsParams {
    const uint8_t *rows; uint32_t pitch, nch, ef, m;
    const uint2 *wl; const uint32_t *wl_cnt; const uint8_t *prob_layer; const uint32_t *prob_slot, *prob_task; uint32_t n_prob;
    const uint32_t *status;                       // per task (k_fused MODE 3): only FS_OK tasks are selected for
    const float *G; uint64_t g_stride; const float *norm2; float band_k;
    uint32_t *out_cnt, *out_ids; float *out_d; uint32_t o_cst, o_lst;        // the members' records (FusedParams' MODE 1 outputs)
    unsigned long long *counters;                 // [0] decisions taken from G, [1] pairs re-evaluated in the canonical order, [2] pairs the GEMM evaluated (lower triangles)
};

// select_neighbors(W, lm) (mod.rs:269-308) for one problem per wavefront; LDS: IDS[64] | dsc[64] | RI[64] | RLv[64] (8 B) | DL[ef] (4 B) | parked row
template <class OP>
__global__ void __launch_bounds__(64)
k_wselect(const WsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t pr = blockIdx.x, lane = threadIdx.x;
    if (p.status[p.prob_task[pr]] != FS_OK) return;
    const uint32_t n = p.wl_cnt[pr], lc = p.prob_layer[pr], os = p.prob_slot[pr];
    const uint32_t lm0 = 2u * p.m, lm = lc == 0 ? lm0 : p.m;
    const uint2 *W = p.wl + (size_t)pr * p.ef;
    const size_t lb = (size_t)os * p.o_lst + (size_t)lc * lm0;
    uint32_t *IDS = (uint32_t *)lds; float *DSC = (float *)(IDS + 64); uint32_t *RI = (uint32_t *)(DSC + 64);
    uint2 *RLv = (uint2 *)(RI + 64); uint32_t *DL = (uint32_t *)(RLv + 64);
    lds_u8 *QV = (lds_u8 *)(DL + ((p.ef + 3u) & ~3u));
    if (n <= lm) {                                                  // mod.rs:276-278
        for (uint32_t i = lane; i < n; i += 64) { const uint2 v = W[i]; p.out_ids[lb + i] = v.y; p.out_d[lb + i] = __builtin_bit_cast(float, v.x); }
        if (lane == 0) p.out_cnt[(size_t)os * p.o_cst + lc] = n;
        return;
    }
    const FRows fr{p.rows, p.pitch, p.nch, DSC};
    const float *G = p.G + (size_t)pr * p.g_stride;
    uint32_t r = 0, nd = 0; unsigned long long n_g = 0, n_x = 0;
    for (uint32_t i = 0; i < n && r < lm; i++) {                     // mod.rs:284-297
        const uint2 e = W[i]; const float thr = __builtin_bit_cast(float, e.x);
        bool closer = true;
        if (r > 0) {                                                 // check_element_closer, mod.rs:315-339: is any d(e, r_j) <= d(e, q)?
            bool sure = false, unsure = false; uint32_t rid = 0;
            if (lane < r) {
                const uint2 rv = RLv[lane]; rid = rv.y;
                const float g = G[(size_t)i * (i - 1u) / 2u + RI[lane]];
                const double band = (double)p.band_k * __builtin_sqrt((double)p.norm2[e.y] * (double)p.norm2[rid]);
                sure = (double)g + band <= (double)thr;              // the canonical value is <= thr whatever the summation order did
                unsure = !sure && (double)g - band <= (double)thr;
            }
            n_g += r;
            if (__ballot(sure) != 0ull) closer = false;
            else {
                const unsigned long long um = __ballot(unsure);
                if (um != 0ull) {                                    // the band holds the decision: the canonical order decides
                    const uint32_t cnt = (uint32_t)__popcll(um);
                    if (unsure) IDS[__popcll(um & ((1ull << lane) - 1ull))] = rid;
                    f_park_w(fr, p.rows + (size_t)e.y * p.pitch, lane, QV);
                    const float d = f_dist_batch<OP, 64>(fr, QV, IDS, cnt, lane);
                    n_x += cnt;
                    if (__ballot(lane < cnt && d <= thr) != 0ull) closer = false;
                    F_WSYNC();
                }
            }
        }
        if (lane == 0) { if (closer) { RI[r] = i; RLv[r] = e; } else DL[nd] = i; }
        if (closer) r++; else nd++;
        F_WSYNC();
    }
    // keep-pruned back-fill (mod.rs:300-305), then the list
    for (uint32_t i = lane; i < r; i += 64) { const uint2 v = RLv[i]; p.out_ids[lb + i] = v.y; p.out_d[lb + i] = __builtin_bit_cast(float, v.x); }
    const uint32_t fill = nd < lm - r ? nd : lm - r;
    for (uint32_t j = lane; j < fill; j += 64) { const uint2 v = W[DL[j]]; p.out_ids[lb + r + j] = v.y; p.out_d[lb + r + j] = __builtin_bit_cast(float, v.x); }
    if (lane == 0) {
        p.out_cnt[(size_t)os * p.o_cst + lc] = r + fill;
        atomicAdd(&p.counters[0], n_g); atomicAdd(&p.counters[1], n_x); atomicAdd(&p.counters[2], (unsigned long long)n * (n - 1u) / 2u);
    }
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


// select_neighbors for the n_prob problems a k_fused MODE 3 launch left (hx_fused.inc.h); everything on the engine's stream, nothing waits here.
// prob_*: device arrays of n_prob entries; d_status: the launch's per-task statuses; record outputs as FusedParams' MODE 1 outputs.
int hx_engine::mfma_select(uint32_t n_prob, uint32_t ef, const void *d_wl, const uint32_t *d_wl_cnt, const uint8_t *d_prob_layer, const uint32_t *d_prob_slot,
                           const uint32_t *d_prob_task, const uint32_t *d_status, uint32_t *d_rec, uint32_t rec_words, unsigned long long *d_counters)
{
    if (n_prob == 0) return HX_OK;
    if (dtype != HX_F16 || metric != HX_NEG_IP) return fail(HX_E_ARG, "the matrix-core select serves halfvec inner product");
    if (ef > WG_NR) return fail(HX_E_ARG, "ef_construction > 256: the matrix-core select is not built for it");
    HX_HIP(this, hipSetDevice(device));
    const uint64_t g_stride = (uint64_t)ef * (ef - 1u) / 2u;
    const size_t need = (size_t)n_prob * g_stride * sizeof(float);
    if (need > cap_wg) {
        HX_HIP(this, hipStreamSynchronize(stream));
        if (d_wg) (void)hipFree(d_wg);
        d_wg = nullptr; cap_wg = 0;
        HX_HIP(this, hipMalloc((void **)&d_wg, need + need / 4));
        cap_wg = need + need / 4;
    }
    static thread_local bool attr_set = false;
    if (!attr_set) {
        HX_HIP(this, hipFuncSetAttribute((const void *)k_wgemm_f16, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * WG_NR * WG_KC + 32 * WG_KC));
        HX_HIP(this, hipFuncSetAttribute((const void *)k_wselect<OpF16<K_IP>>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr_set = true;
    }
    WgParams g; g.rows = d_rows; g.pitch = (uint32_t)pitch; g.ef = ef; g.m = mirror.m; g.wl = (const uint2 *)d_wl; g.wl_cnt = d_wl_cnt; g.prob_layer = d_prob_layer;
    g.n_prob = n_prob; g.G = d_wg; g.g_stride = g_stride; g.stage_bytes = ((ef + 7u) & ~7u) * WG_KC;
    const size_t lds_gemm = 3 * (size_t)g.stage_bytes + 31 * WG_KC;        // ef_construction 200: 80 768 B -> two workgroups per CU
    if (timing) HX_HIP(this, hipEventRecord(ev4, stream));
    hipLaunchKernelGGL(k_wgemm_f16, dim3(n_prob), dim3(512), lds_gemm, stream, g);
    HX_HIP(this, hipGetLastError());
    if (timing) HX_HIP(this, hipEventRecord(ev5, stream));
    WsParams w; w.rows = d_rows; w.pitch = (uint32_t)pitch; w.nch = (uint32_t)((pitch + 1023) / 1024); w.ef = ef; w.m = mirror.m; w.wl = (const uint2 *)d_wl; w.wl_cnt = d_wl_cnt;
    w.prob_layer = d_prob_layer; w.prob_slot = d_prob_slot; w.prob_task = d_prob_task; w.n_prob = n_prob; w.status = d_status; w.G = d_wg; w.g_stride = g_stride;
    w.norm2 = d_mf_norm2; w.band_k = 2.0f * (float)dim * 5.9604645e-08f * 1.001f;
    w.out_cnt = d_rec; w.out_ids = d_rec + HX_FUSED_MAXL; w.out_d = (float *)(d_rec + HX_FUSED_MAXL + HX_FUSED_MAXL * 2 * mirror.m); w.o_cst = w.o_lst = rec_words;
    w.counters = d_counters;
    const size_t lds_sel = (64 + 64 + 64) * 4 + 64 * 8 + (((size_t)ef + 3) & ~(size_t)3) * 4 + (size_t)w.nch * 1024;
    hipLaunchKernelGGL((k_wselect<OpF16<K_IP>>), dim3(n_prob), dim3(64), lds_sel, stream, w);
    HX_HIP(this, hipGetLastError());
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    wg_pending = timing;
    return HX_OK;
}

int hx_engine::wsel_reserve(uint32_t n_prob, uint32_t ef)
{
    HxWselWork &w = wsel;
    if (n_prob <= w.cap_prob && ef <= w.cap_ef) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipStreamSynchronize(stream));
    for (void *q : {(void *)w.d_wl, (void *)w.d_cnt, (void *)w.d_slot, (void *)w.d_task, (void *)w.d_layer, (void *)w.d_counters}) if (q) (void)hipFree(q);
    w = HxWselWork();
    const size_t np = (size_t)n_prob + n_prob / 4 + 64;
    HX_HIP(this, hipMalloc(&w.d_wl, np * ef * 8));
    HX_HIP(this, hipMalloc((void **)&w.d_cnt, np * 4));
    HX_HIP(this, hipMalloc((void **)&w.d_slot, np * 4));
    HX_HIP(this, hipMalloc((void **)&w.d_task, np * 4));
    HX_HIP(this, hipMalloc((void **)&w.d_layer, np));
    HX_HIP(this, hipMalloc((void **)&w.d_counters, 64));
    w.cap_prob = np; w.cap_ef = ef;
    return HX_OK;
}

// after the stream was synchronised: the GEMM's duration and pair count (lower triangles: 2 dim flops each) into the MFMA statistics (hx_kernel_stats kind 4)
int hx_engine::mfma_select_done(uint64_t gemm_pairs)
{
    if (!wg_pending) return HX_OK;
    wg_pending = false;
    float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev4, ev5));
    stat_mfma.launches++; stat_mfma.ms += ms; stat_mfma.units += gemm_pairs;
    HX_HIP(this, hipEventElapsedTime(&ms, ev5, ev3));
    stat_wsel.launches++; stat_wsel.ms += ms; stat_wsel.units += gemm_pairs;
    return HX_OK;
}
