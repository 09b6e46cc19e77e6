// hx_biglist.hip -- select_neighbors (graph/mod.rs:269-339) and update_neighbor_connections (mod.rs:442-489) for lists of ANY legal size
// (options.rs:203-225: m <= 100, so layer-0 lists of up to 200 and result sets W of up to ef_construction <= 1000 candidates).  The kernels of
// hx_links.hip keep a list in one wavefront's lanes and its pair matrix in LDS (m <= 32); here a list lives in LDS arrays walked 64 entries at a
// time and the pair matrix is never formed as a whole: a candidate's row is parked in LDS and the rows selected so far are streamed past it with the
// reference's early exit (FUSED_RB rows per memory round trip), i.e. the distance evaluations check_element_closer makes, in its order, a few
// rows further at most; k_list_ops remembers them per list across the back-links of a batch (pair memo, below).  Stateless: the host hands the lists over as they stand and takes the new ones back (it is the master
// copy for these index shapes); distances are the canonical ones, so lists and distance bits equal the lock-step driver's and the oracle's.
//   k_select_w   one wavefront per (new element, layer): select_neighbors over the result set W the traversal kernel (MODE 3) left on the device
//   k_list_ops   one wavefront per (neighbour, layer) list: its back-links of the batch applied in insertion order
//   k_update_runs_big   the same for aminsert's back-connections (get_update_index + write_neighbor_update, insert.rs:500-871)
#include "hx_fused_core.h"

#define BL_MAX 208            /* lm <= 200, + the new element, rounded up */

// select_neighbors over n candidates in ascending order (cid(h), cd(h)); RI[0..return) = positions h of the new list, in its order
template <class OP, int LPR, class IDF, class DF>
__device__ __forceinline__ uint32_t bl_select(const FRows &fp, lds_u8 *QV, uint32_t *SEL, uint32_t *RI, uint32_t *DI, const uint32_t n, const uint32_t maxn,
                                              IDF cid, DF cd, const uint32_t lane, unsigned long long &ndist)
{
    if (n <= maxn) {                                                                // mod.rs:276-278
        for (uint32_t i = lane; i < n; i += 64) RI[i] = i;
        F_WSYNC();
        return n;
    }
    uint32_t nR = 0, nD = 0;
    for (uint32_t h = 0; h < n; h++) {
        if (nR >= maxn) break;                                                      // mod.rs:285-287
        const uint32_t id = (uint32_t)__builtin_amdgcn_readfirstlane((int)cid(h));
        bool closer = true;                                                         // check_element_closer, mod.rs:315-339
        if (nR) {
            f_park_w(fp, fp.rows + (size_t)id * fp.pitch, lane, QV);
            closer = !f_any_le<OP, LPR>(fp, QV, SEL, nR, lane, cd(h), ndist);
        }
        if (lane == 0) { if (closer) { SEL[nR] = id; RI[nR] = h; } else DI[nD] = h; }
        if (closer) nR++; else nD++;
        F_WSYNC();
    }
    for (uint32_t k = 0; k < nD && nR < maxn; k++, nR++) if (lane == 0) RI[nR] = DI[k];   // mod.rs:300-305
    F_WSYNC();
    return nR;
}

struct SelWParams {
    const uint8_t *rows; uint32_t pitch, cap, n_prob, stride, lm0;                 // cap: sparsevec records' entry capacity (0: dense rows)
    const uint2 *wl; const uint32_t *wl_cnt, *lm;                                   // W of problem pr: wl[pr * stride ..), ascending, {distance bits, id}
    uint32_t *out_ids; float *out_d; uint32_t *out_cnt; unsigned long long *n_pairs;
};

template <class OP, int LPR>
__global__ void __launch_bounds__(64, 4)
k_select_w(const SelWParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *SEL = (uint32_t *)lds, *RI = SEL + BL_MAX, *DI = RI + BL_MAX;        // DI: stride entries
    float *DSC = (float *)(DI + ((p.stride + 15u) & ~15u));
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t lane = threadIdx.x, pr = blockIdx.x;
    if (pr >= p.n_prob) return;
    const FRows fp{p.rows, p.pitch, (p.pitch + 1023u) / 1024u, DSC, p.cap};
    const uint2 *W = p.wl + (size_t)pr * p.stride;
    const uint32_t n = p.wl_cnt[pr] < p.stride ? p.wl_cnt[pr] : p.stride, lm = p.lm[pr];
    unsigned long long ndist = 0;
    const uint32_t nR = bl_select<OP, LPR>(fp, QV, SEL, RI, DI, n, lm,
                                           [&](uint32_t h) { return W[h].y; }, [&](uint32_t h) { return __builtin_bit_cast(float, W[h].x); }, lane, ndist);
    for (uint32_t i = lane; i < nR; i += 64) {
        const uint2 c = W[RI[i]];
        p.out_ids[(size_t)pr * p.lm0 + i] = c.y; p.out_d[(size_t)pr * p.lm0 + i] = __builtin_bit_cast(float, c.x);
    }
    if (lane == 0) { p.out_cnt[pr] = nR; atomicAdd(p.n_pairs, ndist); }
}

struct ListOpsParams {
    const uint8_t *rows; uint32_t pitch, cap, n_groups, lm0;
    uint32_t *ids; float *d; uint32_t *cnt;                                         // list of group g: ids/d[g * lm0 ..), cnt[g]  (in and out)
    const uint32_t *lm, *op_off, *op_new; const float *op_d; unsigned long long *n_pairs;
    // mirror mode (mir.l0_ids != nullptr): the lists are the device mirror's own (group g = (mir.target[g], mir.layer[g]), launch index -> group through mir.gmap)
    ListMirrorArgs mir;
};

// Pair memo of k_list_ops: the candidates of one list carry HANDLES (0..lm; a list position keeps its handle while the element stays, the element a
// back-link drops hands its handle to the next newcomer) and M[tri(ha, hb)] (LDS, packed lower triangle over lm + 1 handles) holds every pair distance
// evaluated so far for the list, NaN = not yet.  Consecutive back-links to one list re-walk nearly the same candidates, so after the first op of a run
// almost every check_element_closer is answered from M and only the newcomer's pairs are evaluated -- the job the resident pair matrix does for
// lists of <= 32 slots (hx_links.hip), without ever forming the whole matrix.
__device__ __forceinline__ uint32_t bl_tri(uint32_t a, uint32_t b) { const uint32_t hi = a > b ? a : b, lo = a > b ? b : a; return hi * (hi - 1u) / 2u + lo; }

template <class OP, int LPR>
__global__ void __launch_bounds__(64, 1)
k_list_ops(const ListOpsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *LID = (uint32_t *)lds; float *LD = (float *)(LID + BL_MAX);
    uint32_t *HND = (uint32_t *)(LD + BL_MAX), *ORD = HND + BL_MAX, *SEL = ORD + BL_MAX, *SELH = SEL + BL_MAX, *RI = SELH + BL_MAX, *DI = RI + BL_MAX;
    uint32_t *TI = DI + BL_MAX; float *TD = (float *)(TI + BL_MAX); uint32_t *TH = (uint32_t *)(TD + BL_MAX), *USED = TH + BL_MAX, *MID = USED + BL_MAX, *MH = MID + BL_MAX;
    float *DSC = (float *)(MH + BL_MAX);
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t nch = (p.pitch + 1023u) / 1024u;
    float *M = (float *)((uint8_t *)QV + nch * 1024u);                              // (lm0 + 1) lm0 / 2 entries
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= p.n_groups) return;
    const bool mirror = p.mir.l0_ids != nullptr;
    const uint32_t g = mirror && p.mir.gmap ? p.mir.gmap[blockIdx.x] : blockIdx.x;
    const FRows fp{p.rows, p.pitch, nch, DSC, p.cap};
    uint32_t *gl_ids; float *gl_d; uint16_t *gl_cnt16 = nullptr; uint32_t lm, cnt, target = 0, layer = 0;
    if (mirror) {
        target = p.mir.target[g]; layer = p.mir.layer[g];
        lm = layer == 0 ? 2u * p.mir.m : p.mir.m;
        if (layer == 0) { gl_ids = p.mir.l0_ids + (size_t)target * 2u * p.mir.m; gl_d = p.mir.l0_d + (size_t)target * 2u * p.mir.m; gl_cnt16 = p.mir.l0_cnt + target; }
        else { const uint32_t blk = p.mir.up_block[target] + layer - 1; gl_ids = p.mir.up_ids + (size_t)blk * p.mir.m; gl_d = p.mir.up_d + (size_t)blk * p.mir.m; gl_cnt16 = p.mir.up_cnt + blk; }
        cnt = *gl_cnt16;
    } else { gl_ids = p.ids + (size_t)g * p.lm0; gl_d = p.d + (size_t)g * p.lm0; lm = p.lm[g]; cnt = p.cnt[g]; }
    for (uint32_t i = lane; i < cnt; i += 64) { LID[i] = gl_ids[i]; LD[i] = gl_d[i]; HND[i] = i; }
    const float qnan = __builtin_nanf("");
    for (uint32_t i = lane; i < (lm + 1u) * lm / 2u; i += 64) M[i] = qnan;
    uint32_t hfree = lm;                                                            // the handle no list position holds once the list is full
    F_WSYNC();
    unsigned long long ndist = 0;
    constexpr uint32_t B = OP::kSparse ? 64u : f_step_rows<LPR>();               // sparsevec: one lane per row, so a step holds a lane-full of rows
    const uint32_t *op_off = mirror ? p.mir.op_off : p.op_off, *op_new = mirror ? p.mir.op_new : p.op_new; const float *op_d = mirror ? p.mir.op_d : p.op_d;
    for (uint32_t op = op_off[g]; op < op_off[g + 1]; op++) {
        if (cnt < lm) {                                                             // mod.rs:469-471
            if (lane == 0) { LID[cnt] = op_new[op]; LD[cnt] = op_d[op]; HND[cnt] = cnt; }
            cnt++; F_WSYNC();
            continue;
        }
        const uint32_t n = cnt + 1;                                                 // mod.rs:474-482: the list + the new element, stable sort by distance
        if (lane == 0) { LID[cnt] = op_new[op]; LD[cnt] = op_d[op]; HND[cnt] = hfree; }
        for (uint32_t j = lane; j <= lm; j += 64) if (j != hfree) M[bl_tri(hfree, j)] = qnan;   // the handle's previous owner is gone
        F_WSYNC();
        for (uint32_t i = lane; i < n; i += 64) {
            const float di = LD[i]; uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) { const float dj = LD[j]; rank += (dj < di || (dj == di && j < i)) ? 1u : 0u; }
            ORD[rank] = i; USED[i] = 0u;
        }
        F_WSYNC();
        uint32_t nR = 0, nD = 0;
        for (uint32_t h = 0; h < n; h++) {                                          // select_neighbors, mod.rs:269-308 (n > lm always)
            if (nR >= lm) break;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)ORD[h]);
            const uint32_t hc = (uint32_t)__builtin_amdgcn_readfirstlane((int)HND[c]);
            const float dc = LD[c];
            bool hit = false; uint32_t nmiss = 0;
            for (uint32_t k0 = 0; k0 < nR && !hit; k0 += 64) {                      // what the memo already knows
                const uint32_t k = k0 + lane; const bool in = k < nR;
                const float v = in ? M[bl_tri(hc, SELH[k])] : 0.0f;
                const bool known = in && v == v;
                if (__ballot(known && v <= dc) != 0ull) { hit = true; break; }      // mod.rs:333-335
                const unsigned long long um = __ballot(in && !known);
                if (in && !known) { const uint32_t at = nmiss + (uint32_t)__popcll(um & ((1ull << lane) - 1ull)); MID[at] = SEL[k]; MH[at] = SELH[k]; }
                nmiss += (uint32_t)__popcll(um);
            }
            F_WSYNC();
            if (!hit && nmiss) {                                                    // the rest is evaluated, in R's order, FUSED_RB rows at a time, until one is close enough
                f_park_w(fp, p.rows + (size_t)LID[c] * p.pitch, lane, QV);
                for (uint32_t j0 = 0; j0 < nmiss; j0 += B) {
                    const uint32_t nb = nmiss - j0 < B ? nmiss - j0 : B;
                    const float d = f_dist_batch<OP, LPR>(fp, QV, MID + j0, nb, lane);
                    ndist += nb;
                    if (lane < nb) M[bl_tri(hc, MH[j0 + lane])] = d;
                    if (__ballot(lane < nb && d <= dc) != 0ull) { hit = true; break; }
                }
            }
            if (lane == 0) { if (!hit) { SEL[nR] = LID[c]; SELH[nR] = hc; RI[nR] = c; } else DI[nD] = c; }
            if (!hit) nR++; else nD++;
            F_WSYNC();
        }
        for (uint32_t k = 0; k < nD && nR < lm; k++, nR++) if (lane == 0) RI[nR] = DI[k];   // mod.rs:300-305
        F_WSYNC();
        for (uint32_t i = lane; i < nR; i += 64) { const uint32_t c = RI[i]; TI[i] = LID[c]; TD[i] = LD[c]; TH[i] = HND[c]; USED[c] = 1u; }
        F_WSYNC();
        for (uint32_t c0 = 0; c0 < n; c0 += 64) {                                   // the one candidate left out hands its handle on
            const unsigned long long out = __ballot(c0 + lane < n && USED[c0 + lane < n ? c0 + lane : 0] == 0u);
            if (out) { hfree = HND[c0 + (uint32_t)__builtin_ctzll(out)]; break; }
        }
        hfree = (uint32_t)__builtin_amdgcn_readfirstlane((int)hfree);
        F_WSYNC();
        for (uint32_t i = lane; i < nR; i += 64) { LID[i] = TI[i]; LD[i] = TD[i]; HND[i] = TH[i]; }   // mod.rs:484-485
        cnt = nR;
        F_WSYNC();
    }
    for (uint32_t i = lane; i < cnt; i += 64) { gl_ids[i] = LID[i]; gl_d[i] = LD[i]; }
    if (mirror && p.mir.xrec) {                                                     // multi-GPU builds: the updated list as a self-describing record
        uint32_t *xr = p.mir.xrec + (size_t)g * p.mir.xrec_words;
        for (uint32_t i = lane; i < cnt; i += 64) { xr[3 + i] = LID[i]; xr[3 + 2u * p.mir.m + i] = __builtin_bit_cast(unsigned int, LD[i]); }
        if (lane == 0) { xr[0] = target; xr[1] = layer; xr[2] = cnt; }
    }
    if (lane == 0) { if (mirror) *gl_cnt16 = (uint16_t)cnt; else p.cnt[g] = cnt; atomicAdd(p.n_pairs, ndist); }
}

// k_update_runs (hx_links.hip) for lists of more than 64 slots: every back-connection a batch of aminserts makes to one list, applied in order --
// "already connected" (insert.rs:805-812), the free slot (:556-559), else get_update_index's walk (insert.rs:630-737): the members stable-sorted by
// distance with the new element behind every member that is not farther, a member kept when no kept MEMBER lies at least as close to it as the
// owner does (pairs with the new element are skipped), pruned members filling up in order, the new element replacing the first member not kept.
template <class OP, int LPR>
__global__ void __launch_bounds__(64, 4)
k_update_runs_big(const ListOpsParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *LID = (uint32_t *)lds; float *LD = (float *)(LID + BL_MAX);
    uint32_t *ORD = (uint32_t *)(LD + BL_MAX), *SEL = ORD + BL_MAX, *KEPT = SEL + BL_MAX, *PRN = KEPT + BL_MAX;
    float *DSC = (float *)(PRN + 3 * BL_MAX);                                       // same carve as k_list_ops (8 arrays)
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    if (g >= p.n_groups) return;
    const FRows fp{p.rows, p.pitch, (p.pitch + 1023u) / 1024u, DSC, p.cap};
    const uint32_t lm = p.lm[g];
    uint32_t cnt = p.cnt[g];
    for (uint32_t i = lane; i < cnt; i += 64) { LID[i] = p.ids[(size_t)g * p.lm0 + i]; LD[i] = p.d[(size_t)g * p.lm0 + i]; }
    F_WSYNC();
    unsigned long long ndist = 0;
    for (uint32_t op = p.op_off[g]; op < p.op_off[g + 1]; op++) {
        const uint32_t nid = p.op_new[op]; const float nd = p.op_d[op];
        bool ex = false;
        for (uint32_t i = lane; i < cnt; i += 64) ex = ex || LID[i] == nid;
        if (__ballot(ex) != 0ull) continue;                                         // connection already exists
        if (cnt < lm) { if (lane == 0) { LID[cnt] = nid; LD[cnt] = nd; } cnt++; F_WSYNC(); continue; }
        uint32_t pn = 0;                                                            // the new element sorts behind every member with d <= new_d
        for (uint32_t c0 = 0; c0 < cnt; c0 += 64) pn += (uint32_t)__popcll(__ballot(c0 + lane < cnt && !(LD[c0 + lane < cnt ? c0 + lane : 0] > nd)));
        for (uint32_t i = lane; i < cnt; i += 64) {
            const float di = LD[i]; uint32_t rank = 0;
            for (uint32_t j = 0; j < cnt; j++) { const float dj = LD[j]; rank += (dj < di || (dj == di && j < i)) ? 1u : 0u; }
            ORD[rank + (di > nd ? 1u : 0u)] = i; KEPT[i] = 0u;
        }
        if (lane == 0) ORD[pn] = 0xFFFFFFFFu;
        F_WSYNC();
        uint32_t nsel = 0, nselm = 0, npr = 0; bool new_sel = false;
        for (uint32_t h = 0; h <= cnt; h++) {
            if (nsel >= lm) break;
            const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)ORD[h]);
            if (s == 0xFFFFFFFFu) { new_sel = true; nsel++; continue; }
            bool hit = false;
            if (nselm) {
                f_park_w(fp, p.rows + (size_t)LID[s] * p.pitch, lane, QV);
                hit = f_any_le<OP, LPR>(fp, QV, SEL, nselm, lane, LD[s], ndist);
            }
            if (!hit) { if (lane == 0) { SEL[nselm] = LID[s]; KEPT[s] = 1u; } nselm++; nsel++; }
            else { if (lane == 0) PRN[npr] = s; npr++; }
            F_WSYNC();
        }
        for (uint32_t k = 0; k < npr && nsel < lm; k++, nsel++) if (lane == 0) KEPT[PRN[k]] = 1u;   // insert.rs:707-712
        F_WSYNC();
        if (new_sel) {
            for (uint32_t c0 = 0; c0 < cnt; c0 += 64) {
                const unsigned long long out = __ballot(c0 + lane < cnt && KEPT[c0 + lane < cnt ? c0 + lane : 0] == 0u);
                if (out) { if (lane == 0) { const uint32_t slot = c0 + (uint32_t)__builtin_ctzll(out); LID[slot] = nid; LD[slot] = nd; } break; }   // insert.rs:722-737
            }
        }
        F_WSYNC();
    }
    for (uint32_t i = lane; i < cnt; i += 64) { p.ids[(size_t)g * p.lm0 + i] = LID[i]; p.d[(size_t)g * p.lm0 + i] = LD[i]; }
    if (lane == 0) { p.cnt[g] = cnt; atomicAdd(p.n_pairs, ndist); }
}

template <class OP>
static hipError_t launch_select_w(hx_engine *e, const SelWParams &p)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (2 * BL_MAX + ((p.stride + 15u) & ~15u) + 64) * 4 + nch * 1024;
    if constexpr (OP::kSparse) hipLaunchKernelGGL((k_select_w<OP, 64>), dim3(p.n_prob), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 128) hipLaunchKernelGGL((k_select_w<OP, 8>), dim3(p.n_prob), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 512) hipLaunchKernelGGL((k_select_w<OP, 32>), dim3(p.n_prob), dim3(64), lds, e->stream, p);
    else hipLaunchKernelGGL((k_select_w<OP, 64>), dim3(p.n_prob), dim3(64), lds, e->stream, p);
    return hipGetLastError();
}
template <class OP>
static hipError_t launch_list_ops(hx_engine *e, const ListOpsParams &p, bool disk)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (8 * BL_MAX + 64) * 4 + nch * 1024;
    if (disk) {
        if constexpr (OP::kSparse) hipLaunchKernelGGL((k_update_runs_big<OP, 64>), dim3(p.n_groups), dim3(64), lds, e->stream, p);
        else if (e->pitch <= 128) hipLaunchKernelGGL((k_update_runs_big<OP, 8>), dim3(p.n_groups), dim3(64), lds, e->stream, p);
        else if (e->pitch <= 512) hipLaunchKernelGGL((k_update_runs_big<OP, 32>), dim3(p.n_groups), dim3(64), lds, e->stream, p);
        else hipLaunchKernelGGL((k_update_runs_big<OP, 64>), dim3(p.n_groups), dim3(64), lds, e->stream, p);
        return hipGetLastError();
    }
    const size_t ldm = (14 * BL_MAX + 64) * 4 + nch * 1024 + ((size_t)p.lm0 + 1) * p.lm0 / 2 * 4;   // + the pair memo
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t st = hipFuncSetAttribute((const void *)k_list_ops<OP, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if constexpr (!OP::kSparse) {
            if (st == hipSuccess) st = hipFuncSetAttribute((const void *)k_list_ops<OP, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
            if (st == hipSuccess) st = hipFuncSetAttribute((const void *)k_list_ops<OP, 32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        }
        if (st != hipSuccess) return st;
        attr_set = true;
    }
    if constexpr (OP::kSparse) hipLaunchKernelGGL((k_list_ops<OP, 64>), dim3(p.n_groups), dim3(64), ldm, e->stream, p);
    else if (e->pitch <= 128) hipLaunchKernelGGL((k_list_ops<OP, 8>), dim3(p.n_groups), dim3(64), ldm, e->stream, p);
    else if (e->pitch <= 512) hipLaunchKernelGGL((k_list_ops<OP, 32>), dim3(p.n_groups), dim3(64), ldm, e->stream, p);
    else hipLaunchKernelGGL((k_list_ops<OP, 64>), dim3(p.n_groups), dim3(64), ldm, e->stream, p);
    return hipGetLastError();
}

static int bl_reserve(hx_engine *e, size_t bytes)
{
    HxMirror &mr = e->mirror;
    if (bytes <= mr.cap_lk) return HX_OK;
    HX_HIP(e, hipStreamSynchronize(e->stream));
    if (mr.h_lk) (void)hipHostFree(mr.h_lk);
    if (mr.d_lk) (void)hipFree(mr.d_lk);
    mr.h_lk = mr.d_lk = nullptr; mr.cap_lk = 0;
    const size_t n = bytes + bytes / 2;
    HX_HIP(e, hipHostMalloc((void **)&mr.h_lk, n, hipHostMallocDefault));
    HX_HIP(e, hipMalloc((void **)&mr.d_lk, n));
    mr.cap_lk = n;
    return HX_OK;
}

// select_neighbors for n_prob result sets that lie in wsel (d_wl: stride entries per problem, ascending; d_cnt); lm[pr] = list size of the problem's layer.
// On return out_* point into pinned memory: ids / d with lm0 entries per problem, cnt per problem.
int hx_engine::biglist_select(uint32_t n_prob, uint32_t stride, const uint32_t *lm, uint32_t lm0,
                              const uint32_t **out_ids, const float **out_d, const uint32_t **out_cnt, uint64_t *n_pairs)
{
    if (n_prob == 0) return HX_OK;
    if (lm0 == 0 || lm0 > 200 || stride == 0 || stride > 1000) return fail(HX_E_ARG, "biglist_select: list size / ef_construction out of range");
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "k_select_w serves rows <= 8 KiB");
    if (!wsel.d_wl || n_prob > wsel.cap_prob || stride > wsel.cap_ef) return fail(HX_E_STATE, "biglist_select: no result sets on the device");
    HX_HIP(this, hipSetDevice(device));
    size_t o = 64;
    const size_t o_lm = o; o += al16((size_t)n_prob * 4);
    const size_t in_bytes = o;
    const size_t o_cnt = o; o += al16((size_t)n_prob * 4);
    const size_t o_ids = o; o += al16((size_t)n_prob * lm0 * 4);
    const size_t o_d = o; o += al16((size_t)n_prob * lm0 * 4);
    int rc = bl_reserve(this, o);
    if (rc) return rc;
    HxMirror &mr = mirror;
    memset(mr.h_lk, 0, 64);
    memcpy(mr.h_lk + o_lm, lm, (size_t)n_prob * 4);
    HX_HIP(this, hipMemcpyAsync(mr.d_lk, mr.h_lk, in_bytes, hipMemcpyHostToDevice, stream));
    SelWParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.cap = dtype == HX_SPARSE ? (uint32_t)std::min(dim, HX_SPARSE_MAX_NNZ) : 0u; p.n_prob = n_prob; p.stride = stride; p.lm0 = lm0;
    p.wl = (const uint2 *)wsel.d_wl; p.wl_cnt = wsel.d_cnt; p.lm = (const uint32_t *)(mr.d_lk + o_lm);
    p.out_ids = (uint32_t *)(mr.d_lk + o_ids); p.out_d = (float *)(mr.d_lk + o_d); p.out_cnt = (uint32_t *)(mr.d_lk + o_cnt);
    p.n_pairs = (unsigned long long *)mr.d_lk;
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_select_w<OpF32<K>>(this, p)
#define F16C(K) ls = launch_select_w<OpF16<K>>(this, p)
    if (dtype == HX_SPARSE) ls = metric == HX_L2SQ ? launch_select_w<OpSparse<K_L2>>(this, p) : metric == HX_NEG_IP ? launch_select_w<OpSparse<K_IP>>(this, p) : launch_select_w<OpSparse<K_L1>>(this, p);
    else HX_DISPATCH(this, F32C, F16C, ls = launch_select_w<OpHamming>(this, p), ls = launch_select_w<OpJaccard>(this, p));
#undef F32C
#undef F16C
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk + o_cnt, mr.d_lk + o_cnt, o - o_cnt, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    *out_cnt = (const uint32_t *)(mr.h_lk + o_cnt); *out_ids = (const uint32_t *)(mr.h_lk + o_ids); *out_d = (const float *)(mr.h_lk + o_d);
    return HX_OK;
}

// back-links of one batch, grouped per list: stage (pinned arrays the caller fills: lists with lm0 entries per group, their sizes and counts, the
// ops of group g at [op_off[g], op_off[g+1])), then biglist_ops_run -> the new lists in the same arrays
int hx_engine::biglist_ops_stage(uint32_t n_groups, uint32_t n_ops, uint32_t lm0, uint32_t **ids, float **d, uint32_t **cnt, uint32_t **lm, uint32_t **op_off,
                                 uint32_t **op_new, float **op_d)
{
    if (n_groups == 0 || lm0 == 0 || lm0 > 200) return fail(HX_E_ARG, "biglist_ops_stage: bad sizes");
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "k_list_ops serves rows <= 8 KiB");
    HX_HIP(this, hipSetDevice(device));
    size_t o = 64;
    bl_o_lm = o; o += al16((size_t)n_groups * 4);
    bl_o_off = o; o += al16(((size_t)n_groups + 1) * 4);
    bl_o_new = o; o += al16((size_t)n_ops * 4);
    bl_o_od = o; o += al16((size_t)n_ops * 4);
    bl_o_cnt = o; o += al16((size_t)n_groups * 4);
    bl_o_ids = o; o += al16((size_t)n_groups * lm0 * 4);
    bl_o_d = o; o += al16((size_t)n_groups * lm0 * 4);
    bl_end = o;
    int rc = bl_reserve(this, o);
    if (rc) return rc;
    uint8_t *h = mirror.h_lk;
    memset(h, 0, 64);
    *ids = (uint32_t *)(h + bl_o_ids); *d = (float *)(h + bl_o_d); *cnt = (uint32_t *)(h + bl_o_cnt); *lm = (uint32_t *)(h + bl_o_lm);
    *op_off = (uint32_t *)(h + bl_o_off); *op_new = (uint32_t *)(h + bl_o_new); *op_d = (float *)(h + bl_o_od);
    bl_groups = n_groups; bl_lm0 = lm0;
    return HX_OK;
}

int hx_engine::biglist_ops_run(uint64_t *n_pairs, bool disk)
{
    if (bl_groups == 0) return fail(HX_E_STATE, "biglist_ops_run without biglist_ops_stage");
    HxMirror &mr = mirror;
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipMemcpyAsync(mr.d_lk, mr.h_lk, bl_end, hipMemcpyHostToDevice, stream));
    ListOpsParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.cap = dtype == HX_SPARSE ? (uint32_t)std::min(dim, HX_SPARSE_MAX_NNZ) : 0u; p.n_groups = bl_groups; p.lm0 = bl_lm0;
    p.ids = (uint32_t *)(mr.d_lk + bl_o_ids); p.d = (float *)(mr.d_lk + bl_o_d); p.cnt = (uint32_t *)(mr.d_lk + bl_o_cnt);
    p.lm = (const uint32_t *)(mr.d_lk + bl_o_lm); p.op_off = (const uint32_t *)(mr.d_lk + bl_o_off); p.op_new = (const uint32_t *)(mr.d_lk + bl_o_new);
    p.op_d = (const float *)(mr.d_lk + bl_o_od); p.n_pairs = (unsigned long long *)mr.d_lk;
    p.mir = ListMirrorArgs{};
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_list_ops<OpF32<K>>(this, p, disk)
#define F16C(K) ls = launch_list_ops<OpF16<K>>(this, p, disk)
    if (dtype == HX_SPARSE) ls = metric == HX_L2SQ ? launch_list_ops<OpSparse<K_L2>>(this, p, disk) : metric == HX_NEG_IP ? launch_list_ops<OpSparse<K_IP>>(this, p, disk) : launch_list_ops<OpSparse<K_L1>>(this, p, disk);
    else HX_DISPATCH(this, F32C, F16C, ls = launch_list_ops<OpHamming>(this, p, disk), ls = launch_list_ops<OpJaccard>(this, p, disk));
#undef F32C
#undef F16C
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk + bl_o_cnt, mr.d_lk + bl_o_cnt, bl_end - bl_o_cnt, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    bl_groups = 0;
    return HX_OK;
}

// the batch pipeline's ordinary groups of lists with 33..64 slots (m = 17..32), in place on the device mirror: the lazily filled pair memo instead of
// k_links_cached's eagerly completed pair matrix (2 016 pairs per touched list and batch when there is no resident matrix to start from)
hipError_t hx_launch_list_ops_mirror(hx_engine *e, const ListMirrorArgs &a)
{
    ListOpsParams p{};
    p.rows = e->d_rows; p.pitch = (uint32_t)e->pitch; p.cap = 0u; p.n_groups = a.n_groups; p.lm0 = 2u * a.m;
    p.n_pairs = a.n_pairs; p.mir = a;
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_list_ops<OpF32<K>>(e, p, false)
#define F16C(K) ls = launch_list_ops<OpF16<K>>(e, p, false)
    HX_DISPATCH(e, F32C, F16C, ls = launch_list_ops<OpHamming>(e, p, false), ls = launch_list_ops<OpJaccard>(e, p, false));
#undef F32C
#undef F16C
    return ls;
}
