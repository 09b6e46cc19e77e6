// hx_links.hip -- update_neighbor_connections (graph/mod.rs:442-489) on the device: k_links (uncached, 512 threads per list),
// k_links_cached (one wavefront per list, pair matrix resident in HBM) and k_links_hub (speculating waves for hub lists),
// plus their host-side launch code.  Own translation unit: compiles in parallel with the traversal kernel.
#include "hx_fused_core.h"

// =================================================================================================
// K3 k_links: update_neighbor_connections (graph/mod.rs:442-489) for one (target, layer) list per 512-thread workgroup.
//   The list (ids + distances) comes from the device mirror into LDS; the group's back-link ops are applied in
//   insertion order: append while there is room (mod.rs:469-471); otherwise candidates = list + new, stable sort by
//   distance (rank sort), the <= 33 candidate rows are staged through LDS exactly as in K2 (1 KiB chunks, register
//   prefetch, double buffer) to form the lower-triangular pair matrix in LDS, and one wavefront runs
//   select_neighbors / check_element_closer (mod.rs:269-339) on that matrix.  The new list goes back to the mirror
//   and to the host.  Distances use the canonical order, so the result is bit-identical to the lock-step path.
// =================================================================================================
#define LK_MAXN 33            /* lm + 1 with lm <= 32 */
#define LK_STAGE 5
#ifndef LK_MINW
#define LK_MINW 4
#endif

struct LinksParams {
    const uint8_t *rows; uint32_t pitch, m;
    uint32_t *l0_ids; float *l0_d; uint16_t *l0_cnt; const uint32_t *up_block; uint32_t *up_ids; float *up_d; uint16_t *up_cnt;
    uint32_t n_groups; const uint32_t *target, *layer, *op_off, *op_new; const float *op_d;
    const uint32_t *gmap;   // launch index -> group (nullptr: identity); the groups of a batch are split between k_links_cached and k_links_hub
    uint32_t *out_ids; float *out_d; uint32_t *out_cnt; unsigned long long *n_pairs;
    const uint2 *wtab; uint32_t wt_size, wt_base, wt_n; const uint8_t *wt_valid;   // W tables of the open batch's members (FusedParams::wtab); nullptr: off
    uint32_t *xrec; uint32_t xrec_words;   // multi-GPU builds: the updated list of group g as a self-describing record {target, layer, cnt, ids[2m], d[2m]} at xrec + g * xrec_words
    uint32_t dbg;   // timing experiments only (HX_LK_DBG): 1 skip pair math, 2 skip row loads, 4 skip select
};

// Pair matrix (packed lower triangle, LDS) of the n <= 33 rows sid[0..n): every thread of the 512-thread workgroup calls this.
template <class OP>
__device__ __forceinline__ void lk_pair_matrix(const LinksParams &p, uint8_t *bufs, const uint32_t buf_bytes, const uint32_t *sid, const uint32_t n,
                                               float *tri, const uint32_t lane, const uint32_t wave)
{
    // ---- pair matrix of the n candidate rows.  Rows 1..n-1 of the triangle are paired (r, n-r) -- r + (n-r) = n <= 33
    // pairs per row-pair -- and wave w owns row-pairs q = w and q = 15 - w (r = q + 1).  For a row-pair, accumulator
    // A[j] is pair (n-r, j), j < n-r, and A[32-j] is pair (r, j), j < r (disjoint because n <= 33).  A wave reads
    // its 4 "a" fragments once per chunk and each b_j fragment once for up to 4 pairs: 36 LDS reads per chunk
    // instead of 132, and 4 independent accumulation chains per read. ----
    uint32_t rr[2], hh[2]; bool use_r[2], use_h[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; q2++) {
        const uint32_t q = q2 == 0 ? wave : 15u - wave;
        const uint32_t r = q + 1, h = n - r;                   // r <= 16
        use_h[q2] = r < n && h > r;                            // partner row strictly above r
        use_r[q2] = r < n && h >= r;                           // r itself (also the lone middle row when h == r)
        rr[q2] = r; hh[q2] = use_h[q2] ? h : 0u;
    }
    uint32_t rid[LK_STAGE];
#pragma unroll
    for (int t = 0; t < LK_STAGE; t++) { const uint32_t r = wave + t * HX_PAIR_WAVES; rid[t] = r < n ? sid[r] : 0u; }
    typename OP::acc_t acc[HX_PAIRS_PER_WAVE];                 // [0..32] row-pair 0, [33..65] row-pair 1
#pragma unroll
    for (int s2 = 0; s2 < HX_PAIRS_PER_WAVE; s2++) OP::init(acc[s2]);
    u4 pre[LK_STAGE];
    auto prefetch = [&](uint32_t c0) {
        const uint32_t off = c0 + lane * 16u;
#pragma unroll
        for (int t = 0; t < LK_STAGE; t++) {
            u4 v = {0u, 0u, 0u, 0u};
            if (wave + t * HX_PAIR_WAVES < n && off < p.pitch && !(p.dbg & 2u)) v = *(const u4 *)(p.rows + (size_t)rid[t] * p.pitch + off);
            pre[t] = v;
        }
    };
    prefetch(0);
    uint32_t bufsel = 0;
    for (uint32_t c0 = 0; c0 < p.pitch; c0 += 1024u, bufsel ^= 1u) {
        uint8_t *buf = bufs + bufsel * buf_bytes;
#pragma unroll
        for (int t = 0; t < LK_STAGE; t++) {
            const uint32_t r = wave + t * HX_PAIR_WAVES;
            if (r < n) *(u4 *)(buf + r * 1024u + lane * 16u) = pre[t];
        }
        __syncthreads();
        if (c0 + 1024u < p.pitch) prefetch(c0 + 1024u);
        if (p.dbg & 1u) continue;
        const u4 ar0 = *(const u4 *)(buf + rr[0] * 1024u + lane * 16u), ah0 = *(const u4 *)(buf + hh[0] * 1024u + lane * 16u);
        const u4 ar1 = *(const u4 *)(buf + rr[1] * 1024u + lane * 16u), ah1 = *(const u4 *)(buf + hh[1] * 1024u + lane * 16u);
        const uint32_t jmax = n - 1u;                          // largest row index any wave needs as "b" is n-2
#pragma unroll
        for (int j = 0; j < LK_MAXN - 1; j++) {
            if ((uint32_t)j < jmax) {
                const u4 bj = *(const u4 *)(buf + (uint32_t)j * 1024u + lane * 16u);
                if (use_h[0] && (uint32_t)j < hh[0]) OP::add(acc[j], ah0, bj);
                if (use_r[0] && (uint32_t)j < rr[0]) OP::add(acc[32 - j], ar0, bj);
                if (use_h[1] && (uint32_t)j < hh[1]) OP::add(acc[33 + j], ah1, bj);
                if (use_r[1] && (uint32_t)j < rr[1]) OP::add(acc[33 + 32 - j], ar1, bj);
            }
        }
    }
    {
        float res0 = 0.f, res1 = 0.f;
        if (!(p.dbg & 32u)) reduce_pairs<OP, HX_PAIRS_PER_WAVE>(acc, lane, res0, res1);
        // lane l holds accumulator l (< 64); lanes 0,1 also hold accumulators 64, 65
#pragma unroll
        for (int part = 0; part < 2; part++) {
            const uint32_t sidx = part == 0 ? lane : 64u + lane;
            const float val = part == 0 ? res0 : res1;
            if (part == 1 && lane >= HX_PAIRS_PER_WAVE - 64) continue;
            const uint32_t q2 = sidx >= 33u ? 1u : 0u, k = sidx - 33u * q2;
            const uint32_t r = q2 ? rr[1] : rr[0], h = q2 ? hh[1] : hh[0];
            const bool uh = q2 ? use_h[1] : use_h[0], ur = q2 ? use_r[1] : use_r[0];
            if (uh && k < h) tri[h * (h - 1) / 2 + k] = val;
            else if (ur && 32u - k < r) tri[r * (r - 1) / 2 + (32u - k)] = val;
        }
    }
}

template <class OP>
__global__ void __launch_bounds__(HX_PAIR_WG, LK_MINW)
k_links(const LinksParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t buf_bytes = LK_MAXN * 1024u;
    uint8_t *bufs = lds;                                         // 2 x 33 KiB row-chunk buffers
    uint32_t *lid = (uint32_t *)(lds + 2 * buf_bytes);           // current list ids [40]
    float *ld = (float *)(lid + 40);                             // current list distances [40]
    uint32_t *sid = (uint32_t *)(ld + 40);                       // sorted candidates [40]
    float *sd = (float *)(sid + 40);
    float *tri = sd + 40;                                        // pair matrix, packed lower triangle [528]
    uint32_t *sel = (uint32_t *)(tri + HX_PAIR_SLAB);            // R indices [40], discarded indices [40], ctl [8]
    uint32_t *dis = sel + 40; uint32_t *ctl = dis + 40;

    const uint32_t g = blockIdx.x;
    if (g >= p.n_groups) return;
    if (p.dbg & 8u) return;
    const uint32_t target = p.target[g], layer = p.layer[g];
    const uint32_t lm = layer == 0 ? 2u * p.m : p.m;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *gl_ids; float *gl_d; uint16_t *gl_cnt;
    if (layer == 0) { gl_ids = p.l0_ids + (size_t)target * 2u * p.m; gl_d = p.l0_d + (size_t)target * 2u * p.m; gl_cnt = p.l0_cnt + target; }
    else { const uint32_t blk = p.up_block[target] + layer - 1; gl_ids = p.up_ids + (size_t)blk * p.m; gl_d = p.up_d + (size_t)blk * p.m; gl_cnt = p.up_cnt + blk; }
    uint32_t cnt = *gl_cnt;
    if (threadIdx.x < cnt) { lid[threadIdx.x] = gl_ids[threadIdx.x]; ld[threadIdx.x] = gl_d[threadIdx.x]; }
    __syncthreads();
    unsigned long long pairs = 0;

    for (uint32_t op = p.op_off[g]; op < p.op_off[g + 1]; op++) {
        if (p.dbg & 16u) break;
        const uint32_t new_id = p.op_new[op]; const float new_d = p.op_d[op];
        if (cnt < lm) {                                                            // mod.rs:469-471
            if (threadIdx.x == 0) { lid[cnt] = new_id; ld[cnt] = new_d; }
            cnt++;
            __syncthreads();
            continue;
        }
        const uint32_t n = cnt + 1;                                                // mod.rs:474-482: items + new, stable sort by distance
        if (threadIdx.x == 0) { lid[cnt] = new_id; ld[cnt] = new_d; }
        __syncthreads();
        if (threadIdx.x < n) {
            const float d = ld[threadIdx.x]; uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) { const float dj = ld[j]; rank += (dj < d) || (dj == d && j < threadIdx.x); }
            sid[rank] = lid[threadIdx.x]; sd[rank] = d;
        }
        __syncthreads();
        lk_pair_matrix<OP>(p, bufs, buf_bytes, sid, n, tri, lane, wave);
        const uint32_t P = n * (n - 1) / 2;
        pairs += P;
        __syncthreads();
        // ---- select_neighbors(candidates, lm) on the matrix: mod.rs:284-305 ----
        if (wave == 0) {
            uint32_t r = 0, nd = 0;
            for (uint32_t i = 0; i < n; i++) {
                if (r >= lm) break;                                                // mod.rs:285-287
                if (p.dbg & 4u) { if (lane == 0) sel[r] = i; r++; continue; }
                const float ed = sd[i];
                bool hit = false;
                if (lane < r) { const uint32_t rj = sel[lane]; hit = tri[i * (i - 1) / 2 + rj] <= ed; }   // mod.rs:333-335
                const bool closer = __ballot(hit) == 0ull;
                if (lane == 0) { if (closer) sel[r] = i; else dis[nd] = i; }
                if (closer) r++; else nd++;
            }
            if (lane == 0) { for (uint32_t j = 0; j < nd && r < lm; j++) sel[r++] = dis[j]; ctl[0] = r; }   // mod.rs:300-305
        }
        __syncthreads();
        cnt = ctl[0];
        if (threadIdx.x < cnt) { const uint32_t k = sel[threadIdx.x]; lid[threadIdx.x] = sid[k]; ld[threadIdx.x] = sd[k]; }
        __syncthreads();
    }
    if (threadIdx.x < cnt) {
        gl_ids[threadIdx.x] = lid[threadIdx.x]; gl_d[threadIdx.x] = ld[threadIdx.x];
        if (p.out_ids) { p.out_ids[(size_t)g * 2u * p.m + threadIdx.x] = lid[threadIdx.x]; p.out_d[(size_t)g * 2u * p.m + threadIdx.x] = ld[threadIdx.x]; }
    }
    if (p.xrec) {
        uint32_t *xr = p.xrec + (size_t)g * p.xrec_words;
        if (threadIdx.x < cnt) { xr[3 + threadIdx.x] = lid[threadIdx.x]; xr[3 + 2u * p.m + threadIdx.x] = __builtin_bit_cast(unsigned int, ld[threadIdx.x]); }
        if (threadIdx.x == 0) { xr[0] = target; xr[1] = layer; xr[2] = cnt; }
    }
    if (threadIdx.x == 0) { *gl_cnt = (uint16_t)cnt; if (p.out_cnt) p.out_cnt[g] = cnt; atomicAdd(p.n_pairs, pairs); }
}

// =================================================================================================
// K4b k_links_cached: the same update_neighbor_connections, one WAVEFRONT per (target, layer) list, with the list's pair
//   matrix kept resident in HBM between batches (layer 0: 496 f32 per element = 2 GB per 1M rows -- cheap on 288 GB).
//   A prune of a full list then needs only the 32 distances new-row <-> current neighbours (streamed exactly like an
//   expansion: new row parked in LDS, neighbour rows 4 x 3 KiB at a time, canonical order) instead of all 528 pairs;
//   select_neighbors runs on the cached matrix + those 32, and the matrix of the surviving list is written back.
//   Pairs the cache does not hold yet (a list's first prune, or after the host rewrote the list) are computed first,
//   row by row.  Cached values are the very bits a recomputation would give, so results equal k_links / the lock-step path.
// =================================================================================================
#define LC_SLOTS 32
#ifndef LC_RB
#define LC_RB FUSED_RB         /* rows in flight when a whole list is streamed; 8 was measured slower (spills at 4 waves/SIMD) */
#endif
#define LC_TRI (LC_SLOTS * (LC_SLOTS - 1) / 2)     /* 496 */
__device__ __forceinline__ uint32_t lc_tri(uint32_t i, uint32_t j) { return i > j ? i * (i - 1) / 2 + j : j * (j - 1) / 2 + i; }

// =================================================================================================
// k_pm_fill: the resident pair matrix of a FULL layer-0 list that is about to see its first prune (or whose cache the host
//   invalidated), computed the K2 way -- the 32 rows staged through LDS once, 496 pairs from LDS -- instead of by k_links_cached's
//   in-wave fill, which streams slot s against slots < s (496 row reads for the same 32 rows: measured as 60 % of the back-link
//   kernels' row traffic on the 1M x 768 build).  One 512-thread workgroup per group of the batch; groups that need nothing return at once.
// =================================================================================================
template <class OP>
__global__ void __launch_bounds__(HX_PAIR_WG, LK_MINW)
k_pm_fill(const LinksParams p, float *pm, uint8_t *pm_valid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t buf_bytes = LK_MAXN * 1024u;
    uint8_t *bufs = lds;
    uint32_t *lid = (uint32_t *)(lds + 2 * buf_bytes);           // list ids [40]
    float *tri = (float *)(lid + 40);                            // packed lower triangle [528]
    if (blockIdx.x >= p.n_groups) return;
    const uint32_t g = p.gmap ? p.gmap[blockIdx.x] : blockIdx.x;
    if (p.layer[g] != 0u) return;
    const uint32_t target = p.target[g], lm = 2u * p.m;
    const uint32_t cnt = p.l0_cnt[target];
    if (cnt != lm || pm_valid[target] >= cnt) return;            // not full yet (appends come first), or already cached
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < cnt) lid[threadIdx.x] = p.l0_ids[(size_t)target * lm + threadIdx.x];
    __syncthreads();
    lk_pair_matrix<OP>(p, bufs, buf_bytes, lid, cnt, tri, lane, wave);
    __syncthreads();
    float *dst = pm + (size_t)target * LC_TRI;
    for (uint32_t i = threadIdx.x; i < cnt * (cnt - 1u) / 2u; i += HX_PAIR_WG) dst[i] = tri[i];
    if (threadIdx.x == 0) { pm_valid[target] = (uint8_t)cnt; atomicAdd(p.n_pairs, (unsigned long long)(cnt * (cnt - 1u) / 2u)); }
}


// One back-link op (new_id at distance new_d) on the list held in LDS: update_neighbor_connections' body, mod.rs:458-487.
// Runs in ONE wave (wave-level ordering only), on the list state (M, lid, ld, cnt, v) and the scratch arrays it is given.
// SPEC == false: applies the op; returns true when the list was pruned (it is then in select order, its matrix complete).
// SPEC == true: touches only the scratch arrays and answers "would this op change the list?" -- false only when it is certain
// that the new row is the one left out and every survivor keeps its slot (then list, distances and matrix stay as they are).
template <class OP, int LPR, bool SPEC, int SLOTS = LC_SLOTS>   // SLOTS: list capacity the kernel is built for (32: m <= 16, 64: m <= 32); also the index that stands for the new row
__device__ bool lc_op(const FRows &fp, const LinksParams &p, float *M, float *M2, uint32_t *lid, float *ld, uint32_t *lid2, float *ld2,
                      float *nd, uint32_t *pos, float *sd, uint32_t *sel, uint32_t *dis, uint32_t *ORD, uint32_t *IDS, lds_u8 *QV,
                      uint32_t &cnt, uint32_t &v, const uint32_t lm, const uint32_t new_id, const float new_d, const uint32_t lane, unsigned long long &ndist, unsigned long long *tk = nullptr,
                      const uint2 *wt = nullptr, const uint32_t wt_mask = 0)
{
    unsigned long long tq = tk ? __builtin_amdgcn_s_memtime() : 0ull;
#define LC_TICK(k) do { if (tk) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk[k] += t_ - tq; tq = t_; } } while (0)


        if (cnt < lm) {                                                            // mod.rs:469-471
            if (SPEC) return true;
            if (lane == 0) { lid[cnt] = new_id; ld[cnt] = new_d; }
            cnt++;
            F_WSYNC();
            return false;
        }
        // pairs among the current list that the cache lacks: slot s against slots < s
        if (SPEC && v < cnt) return true;
        for (uint32_t sl = v < 1 ? 1 : v; sl < cnt; sl++) {
            if (lane < sl) IDS[lane] = lid[lane];
            f_park_w(fp, p.rows + (size_t)lid[sl] * p.pitch, lane, QV);
            const float d = f_dist_batch<OP, LPR, LC_RB>(fp, QV, IDS, sl, lane);
            if (lane < sl) M[sl * (sl - 1) / 2 + lane] = d;
            ndist += sl;
            F_WSYNC();
        }
        v = cnt;
        LC_TICK(0);                                                                // missing pairs of the matrix
        // d(new row, slot) is evaluated LAZILY: a hub list (inner product on unnormalised rows sends thousands of back-links
        // per batch to one list, all applied by this one wave in order) drops most newcomers after a few comparisons, so
        // streaming all `cnt` neighbour rows per op would be a long serial chain of wasted loads.  Order of evaluation:
        // the slots already accepted when the walk reaches the new row first (FUSED_RB at a time, stop at the first batch
        // with a hit, like check_element_closer's early return), the remaining slots only if the new row stays in the list.
        const uint32_t n = cnt + 1;                                                // mod.rs:474-482: items + new, stable sort by distance
        {   // stable rank sort of list + new row (the new row is the LAST candidate, so equal distances keep list entries first): every lane
            // compares its distance with lane j's, read through the scalar unit (no LDS traffic); the new row has no lane of its own when cnt = 64
            const float d = lane < cnt ? ld[lane] : 0.0f; uint32_t rank = 0;
            const unsigned int dbits = __builtin_bit_cast(unsigned int, d);
            for (uint32_t j = 0; j < cnt; j++) {
                const float dj = __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_readlane((int)dbits, (int)j));
                rank += (dj < d) || (dj == d && j < lane);
            }
            if (lane < cnt) { rank += new_d < d ? 1u : 0u; pos[rank] = lane; sd[rank] = d; }
            const uint32_t rank_new = (uint32_t)__popcll(__ballot(lane < cnt && !(new_d < d)));      // list entries with d <= new_d go first
            if (lane == 0) { pos[rank_new] = SLOTS; sd[rank_new] = new_d; }
        }
        F_WSYNC();
        LC_TICK(1);                                                                // sort
        // d(new row, slot) the new row's own layer-0 search already evaluated: one probe of its W table per slot (a 1-hop gather of 8 bytes
        // instead of streaming a row); only the slots it never met are streamed below.  Same bits either way.
        unsigned long long known = 0ull;
        if (wt) {
            bool f = false;
            if (lane < cnt) { float dv = 0.0f; f = wt_lookup(wt, wt_mask, lid[lane], dv); if (f) nd[lane] = dv; }
            known = __ballot(f);
            F_WSYNC();
        }
        const unsigned long long cmask = cnt >= 64u ? ~0ull : (1ull << cnt) - 1ull;
        const uint32_t n_unk = (uint32_t)__popcll(~known & cmask);
        // select_neighbors(candidates, lm): mod.rs:284-305.  D(k1,k2) = cached pair or the new row's distance
        uint32_t r = 0, ndc = 0, n_done = 0;      // n_done: how many entries of the evaluation order ORD have their nd[]
        bool ordered = false; unsigned long long amask = 0ull;   // slots accepted so far
        uint32_t my_slot = 0;                      // lane j: slot of the j-th accepted candidate
        auto order_unknown = [&](const unsigned long long first) {   // evaluation order of the slots still to be streamed: those in `first` first
            const unsigned long long u = ~known & cmask, a = u & first, rest = u & ~first, below = (1ull << lane) - 1ull;
            if (lane < cnt && ((u >> lane) & 1ull)) {
                const uint32_t o = ((a >> lane) & 1ull) ? (uint32_t)__popcll(a & below) : (uint32_t)__popcll(a) + (uint32_t)__popcll(rest & below);
                IDS[o] = lid[lane]; ORD[o] = lane;
            }
            if (n_unk) f_park_w(fp, p.rows + (size_t)new_id * p.pitch, lane, QV); else F_WSYNC();
            ordered = true;
        };
        auto finish_nd = [&]() {                  // all remaining d(new, slot)
            if (!ordered) order_unknown(0ull);
            if (n_done < n_unk) {
                const float d = f_dist_batch<OP, LPR, LC_RB>(fp, QV, IDS + n_done, n_unk - n_done, lane);
                if (lane < n_unk - n_done) nd[ORD[n_done + lane]] = d;
                ndist += n_unk - n_done; n_done = n_unk;
            }
            F_WSYNC();
        };
        for (uint32_t i = 0; i < n; i++) {
            if (r >= lm) break;
            const float ed = sd[i]; const uint32_t si = pos[i];
            bool closer;
            if (si == SLOTS) {
                LC_TICK(2);                                                        // walk so far
                // accepted slots decide (mod.rs:324-336): first the ones whose distance is already known, then the rest streamed
                // FUSED_RB at a time, stopping at the first batch with a hit like check_element_closer's early return
                const unsigned long long am = amask;
                bool hit = __ballot(lane < cnt && ((am >> lane) & 1ull) && ((known >> lane) & 1ull) && nd[lane] <= ed) != 0ull;
                if (!hit) {
                    order_unknown(am);
                    const uint32_t na = (uint32_t)__popcll(am & ~known & cmask);
                    constexpr uint32_t B = f_step_rows<LPR>();
                    for (uint32_t j0 = 0; j0 < na && !hit; j0 += B) {
                        const uint32_t nb = na - j0 < B ? na - j0 : B;
                        const float d = f_dist_batch<OP, LPR>(fp, QV, IDS + j0, nb, lane);
                        if (lane < nb) nd[ORD[j0 + lane]] = d;
                        ndist += nb; n_done = j0 + nb;
                        hit = __ballot(lane < nb && d <= ed) != 0ull;                  // mod.rs:333-335
                    }
                }
                closer = !hit;
                LC_TICK(3);                                                        // lazy distances of the new row
                if (SPEC && closer) return true;                                   // the new row enters the list
                if (closer) finish_nd();                                           // later candidates are compared with the new row
            } else {
                bool hit = false;
                if (lane < r) {
                    const uint32_t sj = my_slot;                                   // slot of the lane-th accepted candidate (== pos[sel[lane]])
                    const float dij = sj == SLOTS ? nd[si] : M[lc_tri(si, sj)];
                    hit = dij <= ed;                                               // mod.rs:333-335
                }
                closer = __ballot(hit) == 0ull;
                if (closer) amask |= 1ull << si;
            }
            if (closer && lane == r) my_slot = si;
            if (lane == 0) { if (closer) sel[r] = i; else dis[ndc] = i; }
            if (closer) r++; else ndc++;
            F_WSYNC();
        }
        if (lane == 0) for (uint32_t j = 0; j < ndc && r < lm; j++) sel[r++] = dis[j];   // mod.rs:300-305
        r = __shfl(r, 0, 64);
        F_WSYNC();
        LC_TICK(2);
        {   // unchanged iff the new row is the one left out AND the survivors keep their slots, in order: nothing to rebuild then
            const bool ok = lane >= r || pos[sel[lane]] == lane;
            const bool changed = __ballot(!ok) != 0ull;
            if (SPEC) return changed;
            if (!changed) { LC_TICK(4); return true; }            // (a pruned list stays in select order: still "pruned" for the caller)
        }
        {   // the new row's distances to every slot are needed only if it stays in the list
            bool mine = lane < r && pos[sel[lane]] == SLOTS;
            if (__ballot(mine) != 0ull) finish_nd();
        }
        // surviving list and its pair matrix
        if (lane < r) { const uint32_t sa = pos[sel[lane]]; lid2[lane] = sa == SLOTS ? new_id : lid[sa]; ld2[lane] = sd[sel[lane]]; }
        for (uint32_t idx = lane; idx < r * (r - 1) / 2; idx += 64) {
            uint32_t a, b; tri_decode(idx, a, b);
            const uint32_t sa = pos[sel[a]], sb = pos[sel[b]];
            M2[idx] = sa == SLOTS ? nd[sb] : (sb == SLOTS ? nd[sa] : M[lc_tri(sa, sb)]);
        }
        F_WSYNC();
        if (lane < r) { lid[lane] = lid2[lane]; ld[lane] = ld2[lane]; }
        for (uint32_t idx = lane; idx < r * (r - 1) / 2; idx += 64) M[idx] = M2[idx];
        cnt = r; v = r;
        F_WSYNC();
        LC_TICK(4);                                                                // remaining distances + list / matrix rebuild
            return true;
#undef LC_TICK
}

// the W table of the element a back-link op adds, if it is a member of the open batch whose layer-0 search ran on this GPU
__device__ __forceinline__ const uint2 *lk_wtab(const LinksParams &p, uint32_t layer, uint32_t new_id)
{
    if (!p.wtab || layer != 0u) return nullptr;
    const uint32_t k = new_id - p.wt_base;
    if (k >= p.wt_n || (p.wt_valid && !p.wt_valid[k])) return nullptr;
    return p.wtab + (size_t)k * p.wt_size;
}

template <class OP, int LPR, int SLOTS>
__global__ void __launch_bounds__(64, 4)
k_links_cached(const LinksParams p, float *pm, uint8_t *pm_valid)
{
    constexpr int AR = SLOTS + 8, MT = (SLOTS + 1) * SLOTS / 2;      // list-sized arrays; packed lower triangle over SLOTS + 1 candidates
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float *M = (float *)lds;                       // pair matrix over list slots, packed lower triangle [528]
    float *M2 = M + MT;
    uint32_t *lid = (uint32_t *)(M2 + MT);        // list ids / distances to the target [40]
    float *ld = (float *)(lid + AR);
    uint32_t *lid2 = (uint32_t *)(ld + AR); float *ld2 = (float *)(lid2 + AR);
    float *nd = ld2 + AR;                          // d(new row, slot j)
    uint32_t *pos = (uint32_t *)(nd + AR);         // sorted candidate k -> slot (LC_SLOTS = the new row)
    float *sd = (float *)(pos + AR);
    uint32_t *sel = (uint32_t *)(sd + AR), *dis = sel + AR, *ORD = dis + AR, *IDS = ORD + AR;   // IDS[64]
    float *DSC = (float *)(IDS + 64);
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= p.n_groups) return;
    const uint32_t g = p.gmap ? p.gmap[blockIdx.x] : blockIdx.x;
    const FRows fp{p.rows, p.pitch, (p.pitch + 1023u) / 1024u, DSC};
    const uint32_t target = p.target[g], layer = p.layer[g];
    const uint32_t lm = layer == 0 ? 2u * p.m : p.m;
    uint32_t *gl_ids; float *gl_d; uint16_t *gl_cnt;
    if (layer == 0) { gl_ids = p.l0_ids + (size_t)target * 2u * p.m; gl_d = p.l0_d + (size_t)target * 2u * p.m; gl_cnt = p.l0_cnt + target; }
    else { const uint32_t blk = p.up_block[target] + layer - 1; gl_ids = p.up_ids + (size_t)blk * p.m; gl_d = p.up_d + (size_t)blk * p.m; gl_cnt = p.up_cnt + blk; }
    uint32_t cnt = *gl_cnt;
    const bool cached = layer == 0 && lm == LC_SLOTS && pm != nullptr;
    uint32_t v = cached ? pm_valid[target] : 0u;                 // slots [0, v) have their pairs in the cache
    if (v > cnt) v = 0;
    if (lane < cnt) { lid[lane] = gl_ids[lane]; ld[lane] = gl_d[lane]; }
    if (v > 1) { const float *src = pm + (size_t)target * LC_TRI; for (uint32_t i = lane; i < v * (v - 1) / 2; i += 64) M[i] = src[i]; }
    __syncthreads();
    unsigned long long ndist = 0;

    unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
    const bool tm = FUSED_TIMERS_ON && (p.dbg & 8u) != 0; const unsigned long long tk0 = tm ? __builtin_amdgcn_s_memtime() : 0ull;   // phase clocks: -DFUSED_TIMERS builds only
    for (uint32_t op = p.op_off[g]; op < p.op_off[g + 1]; op++)
        (void)lc_op<OP, LPR, false, SLOTS>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op], p.op_d[op], lane, ndist, tm ? tk : nullptr,
                                    lk_wtab(p, layer, p.op_new[op]), p.wt_size - 1u);
    if (lane < cnt) {
        gl_ids[lane] = lid[lane]; gl_d[lane] = ld[lane];
        if (p.out_ids) { p.out_ids[(size_t)g * 2u * p.m + lane] = lid[lane]; p.out_d[(size_t)g * 2u * p.m + lane] = ld[lane]; }
    }
    if (cached) {
        float *dst = pm + (size_t)target * LC_TRI;
        for (uint32_t i = lane; i < (v > 1 ? v * (v - 1) / 2 : 0u); i += 64) dst[i] = M[i];
        if (lane == 0) pm_valid[target] = (uint8_t)v;
    }
    if (p.xrec) {
        uint32_t *xr = p.xrec + (size_t)g * p.xrec_words;
        if (lane < cnt) { xr[3 + lane] = lid[lane]; xr[3 + 2u * p.m + lane] = __builtin_bit_cast(unsigned int, ld[lane]); }
        if (lane == 0) { xr[0] = target; xr[1] = layer; xr[2] = cnt; }
    }
    if (lane == 0) { *gl_cnt = (uint16_t)cnt; if (p.out_cnt) p.out_cnt[g] = cnt; atomicAdd(p.n_pairs, ndist); }
    if (tm && lane == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tk[5] = __builtin_amdgcn_s_memtime() - tk0; for (int i = 0; i < 6; i++) atomicAdd(p.n_pairs + 1 + i, tk[i]); }
}

// =================================================================================================
// K4c k_links_hub: the same per-list work for a list that receives a LONG chain of back-links in one batch (inner product on
//   unnormalised rows: thousands of ops for one hub list, which one wave would apply one after the other).  Almost all of a
//   hub's newcomers are left out again, so HUB_W waves evaluate the next HUB_W ops speculatively, each against the
//   current list (lc_op<SPEC>: scratch only); the ops before the first one that would change the list are no-ops by
//   construction, that one is applied by wave 0 with the ordinary code, and the rest are re-evaluated.  Same result as the
//   one-wave kernel, op for op.
// =================================================================================================
#ifndef HUB_W
#define HUB_W 8
#endif
template <class OP, int LPR, int SLOTS>
__global__ void __launch_bounds__(64 * HUB_W, 1)
k_links_hub(const LinksParams p, float *pm, uint8_t *pm_valid)
{
    constexpr int AR = SLOTS + 8, MT = (SLOTS + 1) * SLOTS / 2;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t nchb = ((p.pitch + 1023u) / 1024u) * 1024u;
    // shared: M[528] M2[528] lid[40] ld[40] lid2[40] ld2[40] ctl[16]; per wave: nd pos sd sel dis ORD [40 each] IDS[64] DSC[64] QV[nchb]
    float *M = (float *)lds, *M2 = M + MT;
    uint32_t *lid = (uint32_t *)(M2 + MT); float *ld = (float *)(lid + AR);
    uint32_t *lid2 = (uint32_t *)(ld + AR); float *ld2 = (float *)(lid2 + AR);
    uint32_t *ctl = (uint32_t *)(ld2 + AR);
    uint8_t *wbase = (uint8_t *)(ctl + 16) + (size_t)wave * ((AR * 6 + 64 + 64) * 4 + nchb);
    float *nd = (float *)wbase; uint32_t *pos = (uint32_t *)(nd + AR); float *sd = (float *)(pos + AR);
    uint32_t *sel = (uint32_t *)(sd + AR), *dis = sel + AR, *ORD = dis + AR, *IDS = ORD + AR;
    float *DSC = (float *)(IDS + 64);
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    if (blockIdx.x >= p.n_groups) return;
    const uint32_t g = p.gmap ? p.gmap[blockIdx.x] : blockIdx.x;
    const FRows fp{p.rows, p.pitch, (p.pitch + 1023u) / 1024u, DSC};
    const uint32_t target = p.target[g], layer = p.layer[g];
    const uint32_t lm = layer == 0 ? 2u * p.m : p.m;
    uint32_t *gl_ids; float *gl_d; uint16_t *gl_cnt;
    if (layer == 0) { gl_ids = p.l0_ids + (size_t)target * 2u * p.m; gl_d = p.l0_d + (size_t)target * 2u * p.m; gl_cnt = p.l0_cnt + target; }
    else { const uint32_t blk = p.up_block[target] + layer - 1; gl_ids = p.up_ids + (size_t)blk * p.m; gl_d = p.up_d + (size_t)blk * p.m; gl_cnt = p.up_cnt + blk; }
    const bool cached = layer == 0 && lm == LC_SLOTS && pm != nullptr;
    uint32_t cnt = *gl_cnt;
    uint32_t v = cached ? pm_valid[target] : 0u;
    if (v > cnt) v = 0;
    if (threadIdx.x < cnt) { lid[threadIdx.x] = gl_ids[threadIdx.x]; ld[threadIdx.x] = gl_d[threadIdx.x]; }
    if (v > 1) { const float *src = pm + (size_t)target * LC_TRI; for (uint32_t i = threadIdx.x; i < v * (v - 1) / 2; i += 64 * HUB_W) M[i] = src[i]; }
    const uint32_t op_end = p.op_off[g + 1];
    if (threadIdx.x == 0) { ctl[0] = p.op_off[g]; ctl[1] = cnt; ctl[2] = v; ctl[3] = 0; }   // next op, |list|, cached slots, list is in select order
    unsigned long long ndist = 0;
    for (;;) {
        __syncthreads();
        const uint32_t op = ctl[0]; cnt = ctl[1]; v = ctl[2]; const bool canon = ctl[3] != 0;
        if (op >= op_end) break;
        if (!canon || cnt < lm || v < cnt) {                       // appends, the first prune, missing pairs: the ordinary path, one op
            __syncthreads();
            if (wave == 0) {
                const bool pruned = lc_op<OP, LPR, false, SLOTS>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op], p.op_d[op], lane, ndist, nullptr,
                                                          lk_wtab(p, layer, p.op_new[op]), p.wt_size - 1u);
                if (lane == 0) { ctl[0] = op + 1u; ctl[1] = cnt; ctl[2] = v; if (pruned) ctl[3] = 1u; }
            }
            continue;
        }
        const uint32_t nv = op_end - op < HUB_W ? op_end - op : HUB_W;   // ops evaluated this round
        bool changed = false;
        if (wave < nv) {
            uint32_t c2 = cnt, v2 = v;
            changed = lc_op<OP, LPR, true, SLOTS>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, c2, v2, lm, p.op_new[op + wave], p.op_d[op + wave], lane, ndist, nullptr,
                                           lk_wtab(p, layer, p.op_new[op + wave]), p.wt_size - 1u);
        }
        if (lane == 0) ctl[4 + wave] = changed ? 1u : 0u;
        __syncthreads();
        uint32_t first = nv;
        for (uint32_t w = 0; w < nv; w++) if (ctl[4 + w]) { first = w; break; }
        __syncthreads();
        if (first == nv) { if (threadIdx.x == 0) ctl[0] = op + nv; continue; }
        if (wave == 0) {
            (void)lc_op<OP, LPR, false, SLOTS>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op + first], p.op_d[op + first], lane, ndist, nullptr,
                                        lk_wtab(p, layer, p.op_new[op + first]), p.wt_size - 1u);
            if (lane == 0) { ctl[0] = op + first + 1u; ctl[1] = cnt; ctl[2] = v; }
        }
    }
    cnt = ctl[1]; v = ctl[2];
    if (threadIdx.x < cnt) {
        gl_ids[threadIdx.x] = lid[threadIdx.x]; gl_d[threadIdx.x] = ld[threadIdx.x];
        if (p.out_ids) { p.out_ids[(size_t)g * 2u * p.m + threadIdx.x] = lid[threadIdx.x]; p.out_d[(size_t)g * 2u * p.m + threadIdx.x] = ld[threadIdx.x]; }
    }
    if (cached) {
        float *dst = pm + (size_t)target * LC_TRI;
        for (uint32_t i = threadIdx.x; i < (v > 1 ? v * (v - 1) / 2 : 0u); i += 64 * HUB_W) dst[i] = M[i];
        if (threadIdx.x == 0) pm_valid[target] = (uint8_t)v;
    }
    if (p.xrec) {
        uint32_t *xr = p.xrec + (size_t)g * p.xrec_words;
        if (threadIdx.x < cnt) { xr[3 + threadIdx.x] = lid[threadIdx.x]; xr[3 + 2u * p.m + threadIdx.x] = __builtin_bit_cast(unsigned int, ld[threadIdx.x]); }
        if (threadIdx.x == 0) { xr[0] = target; xr[1] = layer; xr[2] = cnt; }
    }
    if (threadIdx.x == 0) { *gl_cnt = (uint16_t)cnt; if (p.out_cnt) p.out_cnt[g] = cnt; }
    if (lane == 0) atomicAdd(p.n_pairs, ndist);
}

// =================================================================================================
// K4d k_update_index: get_update_index of aminsert (src/index/insert.rs:500-739) for FULL lists, one wavefront per (neighbour, layer, new element) op.
//   Stateless: the host hands over the list as it stands (ids + the stored distances to the list's owner: the very bits insert.rs:606 recomputes) and
//   d(owner, new element); the kernel answers the slot the new element takes (-3: none).  Candidates = the lm members, stable-sorted by distance, with
//   the new element behind every member that is not farther (insert.rs:630-665: two stable sorts); the heuristic walks them (insert.rs:673-712): a
//   member is kept when no already-kept MEMBER lies at least as close to it as the owner does (pairs with the new element are skipped, :680-693) --
//   its row parked in LDS, the kept members' rows streamed FUSED_RB at a time with the reference's early exit (f_any_le) -- and pruned members fill
//   the remaining places in order; the new element replaces the first member that is not kept (:722-737).  The pair distances are evaluated only
//   when the walk asks for them (the lock-step driver orders all lm (lm - 1) / 2 up front); the answer is the same.
// =================================================================================================
struct UpdParams {
    const uint8_t *rows; uint32_t pitch, cap, n_ops, stride;                       // cap: sparsevec records' entry capacity (0: dense rows)
    const uint32_t *ids; const float *d; const float *new_d; const uint32_t *cnt;
    int32_t *slot; unsigned long long *n_pairs;
};

template <class OP, int LPR>
__global__ void __launch_bounds__(64, 4)
k_update_index(const UpdParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *LID = (uint32_t *)lds; float *LD = (float *)(LID + 64);
    uint32_t *ORD = (uint32_t *)(LD + 64);                      // sorted position -> slot; 64 = the new element  [<= 65 entries]
    uint32_t *SEL = ORD + 72, *PRN = SEL + 64;                  // kept members' row ids (walk order); pruned members' slots
    float *DSC = (float *)(PRN + 64);
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t lane = threadIdx.x, op = blockIdx.x;
    if (op >= p.n_ops) return;
    const FRows fp{p.rows, p.pitch, (p.pitch + 1023u) / 1024u, DSC, p.cap};
    const uint32_t cnt = p.cnt[op], lm = cnt;                   // full list
    const float nd = p.new_d[op];
    float di = 0.0f;
    if (lane < cnt) { LID[lane] = p.ids[(size_t)op * p.stride + lane]; di = p.d[(size_t)op * p.stride + lane]; LD[lane] = di; }
    F_WSYNC();
    uint32_t rank = 0;                                          // stable rank of member `lane` among the members
    for (uint32_t j = 0; j < cnt; j++) { const float dj = LD[j]; rank += (dj < di || (dj == di && j < lane)) ? 1u : 0u; }
    const uint32_t pn = (uint32_t)__popcll(__ballot(lane < cnt && !(di > nd)));   // the new element sorts behind every member with d <= new_d
    if (lane < cnt) ORD[rank + (di > nd ? 1u : 0u)] = lane;
    if (lane == 0) ORD[pn] = 64u;
    F_WSYNC();
    uint32_t nsel = 0, nselm = 0, npr = 0; bool new_sel = false, kept = false;
    unsigned long long ndist = 0;
    for (uint32_t h = 0; h <= cnt; h++) {
        if (nsel >= lm) break;
        const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)ORD[h]);
        if (s == 64u) { new_sel = true; nsel++; continue; }
        bool hit = false;
        if (nselm) {
            f_park_w(fp, p.rows + (size_t)LID[s] * p.pitch, lane, QV);
            hit = f_any_le<OP, LPR>(fp, QV, SEL, nselm, lane, LD[s], ndist);
        }
        if (!hit) { if (lane == 0) SEL[nselm] = LID[s]; nselm++; nsel++; if (lane == s) kept = true; }
        else { if (lane == 0) PRN[npr] = s; npr++; }
        F_WSYNC();
    }
    for (uint32_t k = 0; k < npr && nsel < lm; k++, nsel++) if (lane == PRN[k]) kept = true;   // insert.rs:707-712
    int res = -3;
    if (new_sel) { const unsigned long long out = __ballot(lane < cnt && !kept); if (out) res = (int)__builtin_ctzll(out); }
    if (lane == 0) { p.slot[op] = res; atomicAdd(p.n_pairs, ndist); }
}

// k_update_runs: every back-connection a batch of aminserts makes to ONE list, applied in order by one wavefront -- write_neighbor_update's "already connected"
// test (insert.rs:805-812), the free slot (:556-559, :826-838) and k_update_index's walk for a full list, the list kept in LDS between the ops.  For indexes
// without deleted / TID-less elements (get_update_index's other exits, insert.rs:524-527, 566-625, never fire there); the new list goes back to the host.
struct UpdRunParams {
    const uint8_t *rows; uint32_t pitch, cap, n_runs, stride;
    uint32_t *ids; float *d; uint32_t *cnt;                       // list of run r: ids/d[r * stride ..), cnt[r]  (in and out)
    const uint32_t *lm, *op_off, *op_new; const float *op_d; unsigned long long *n_pairs;
};

template <class OP, int LPR>
__global__ void __launch_bounds__(64, 4)
k_update_runs(const UpdRunParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t *LID = (uint32_t *)lds; float *LD = (float *)(LID + 64);
    uint32_t *ORD = (uint32_t *)(LD + 64);
    uint32_t *SEL = ORD + 72, *SELS = SEL + 64, *PRN = SELS + 64, *MID = PRN + 64, *MS = MID + 64;   // kept members: row ids, slots; pruned slots; members whose pair is not in the memo yet
    float *DSC = (float *)(MS + 64);
    lds_u8 *QV = (lds_u8 *)(DSC + 64);
    const uint32_t nch = (p.pitch + 1023u) / 1024u;
    // pair memo of the run: M[tri(slot a, slot b)] = d(member a, member b) once the walk has asked for it, NaN before; a replaced slot forgets its row.
    // (pairs with the new element are never evaluated, insert.rs:680-693, so slots are all the key there is)
    float *M = (float *)((uint8_t *)QV + nch * 1024u);
    const uint32_t lane = threadIdx.x, r = blockIdx.x;
    if (r >= p.n_runs) return;
    const FRows fp{p.rows, p.pitch, nch, DSC, p.cap};
    const uint32_t lm = p.lm[r];
    uint32_t cnt = p.cnt[r];
    if (lane < cnt) { LID[lane] = p.ids[(size_t)r * p.stride + lane]; LD[lane] = p.d[(size_t)r * p.stride + lane]; }
    const float qnan = __builtin_nanf("");
    const bool memo = p.op_off[r + 1] - p.op_off[r] > 1u;                           // a single op has nothing to remember for
    if (memo) for (uint32_t i = lane; i < lm * (lm - 1u) / 2u; i += 64) M[i] = qnan;
    F_WSYNC();
    unsigned long long ndist = 0;
    constexpr uint32_t B = OP::kSparse ? 64u : f_step_rows<LPR>();               // sparsevec: one lane per row, so a step holds a lane-full of rows
    auto tri = [](uint32_t a, uint32_t b) { const uint32_t hi = a > b ? a : b, lo = a > b ? b : a; return hi * (hi - 1u) / 2u + lo; };
    for (uint32_t op = p.op_off[r]; op < p.op_off[r + 1]; op++) {
        const uint32_t nid = p.op_new[op]; const float nd = p.op_d[op];
        if (__ballot(lane < cnt && LID[lane] == nid) != 0ull) continue;             // connection already exists
        if (cnt < lm) { if (lane == 0) { LID[cnt] = nid; LD[cnt] = nd; } cnt++; F_WSYNC(); continue; }
        const float di = lane < cnt ? LD[lane] : 0.0f;
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; j++) { const float dj = LD[j]; rank += (dj < di || (dj == di && j < lane)) ? 1u : 0u; }
        const uint32_t pn = (uint32_t)__popcll(__ballot(lane < cnt && !(di > nd)));
        if (lane < cnt) ORD[rank + (di > nd ? 1u : 0u)] = lane;
        if (lane == 0) ORD[pn] = 64u;
        F_WSYNC();
        uint32_t nsel = 0, nselm = 0, npr = 0; bool new_sel = false, kept = false;
        for (uint32_t h = 0; h <= cnt; h++) {
            if (nsel >= lm) break;
            const uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)ORD[h]);
            if (s == 64u) { new_sel = true; nsel++; continue; }
            const float ds = LD[s];
            bool hit = false;
            if (nselm) {
                uint32_t nmiss = nselm;
                if (memo) {                                                         // what the run already knows about member s
                    const bool in = lane < nselm;
                    const float v = in ? M[tri(s, SELS[lane])] : 0.0f;
                    const bool known = in && v == v;
                    hit = __ballot(known && v <= ds) != 0ull;
                    const unsigned long long um = __ballot(in && !known);
                    if (in && !known) { const uint32_t at = (uint32_t)__popcll(um & ((1ull << lane) - 1ull)); MID[at] = SEL[lane]; MS[at] = SELS[lane]; }
                    nmiss = (uint32_t)__popcll(um);
                    F_WSYNC();
                }
                if (!hit && nmiss) {
                    const uint32_t *ids = memo ? MID : SEL;
                    f_park_w(fp, p.rows + (size_t)LID[s] * p.pitch, lane, QV);
                    for (uint32_t j0 = 0; j0 < nmiss; j0 += B) {
                        const uint32_t nb = nmiss - j0 < B ? nmiss - j0 : B;
                        const float d = f_dist_batch<OP, LPR>(fp, QV, ids + j0, nb, lane);
                        ndist += nb;
                        if (memo && lane < nb) M[tri(s, MS[j0 + lane])] = d;
                        if (__ballot(lane < nb && d <= ds) != 0ull) { hit = true; break; }
                    }
                }
            }
            if (!hit) { if (lane == 0) { SEL[nselm] = LID[s]; SELS[nselm] = s; } nselm++; nsel++; if (lane == s) kept = true; }
            else { if (lane == 0) PRN[npr] = s; npr++; }
            F_WSYNC();
        }
        for (uint32_t k = 0; k < npr && nsel < lm; k++, nsel++) if (lane == PRN[k]) kept = true;
        if (new_sel) {
            const unsigned long long out = __ballot(lane < cnt && !kept);
            if (out) {
                const uint32_t slot = (uint32_t)__builtin_ctzll(out);
                if (lane == 0) { LID[slot] = nid; LD[slot] = nd; }
                if (memo && lane < cnt && lane != slot) M[tri(slot, lane)] = qnan;  // the slot's previous owner is gone
            }
        }
        F_WSYNC();
    }
    if (lane < cnt) { p.ids[(size_t)r * p.stride + lane] = LID[lane]; p.d[(size_t)r * p.stride + lane] = LD[lane]; }
    if (lane == 0) { p.cnt[r] = cnt; atomicAdd(p.n_pairs, ndist); }
}

template <class OP>
static hipError_t launch_update_runs(hx_engine *e, const UpdRunParams &p)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (64 * 8 + 72) * 4 + nch * 1024 + (size_t)p.stride * (p.stride - 1) / 2 * 4;   // + the run's pair memo
    if constexpr (OP::kSparse) hipLaunchKernelGGL((k_update_runs<OP, 64>), dim3(p.n_runs), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 128) hipLaunchKernelGGL((k_update_runs<OP, 8>), dim3(p.n_runs), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 512) hipLaunchKernelGGL((k_update_runs<OP, 32>), dim3(p.n_runs), dim3(64), lds, e->stream, p);
    else hipLaunchKernelGGL((k_update_runs<OP, 64>), dim3(p.n_runs), dim3(64), lds, e->stream, p);
    return hipGetLastError();
}

int hx_engine::update_runs_stage(uint32_t n_runs, uint32_t n_ops, uint32_t stride, uint32_t **ids, float **d, uint32_t **cnt, uint32_t **lm, uint32_t **op_off,
                                 uint32_t **op_new, float **op_d)
{
    HxMirror &mr = mirror;
    if (n_runs == 0 || stride == 0 || stride > 64) return fail(HX_E_ARG, "update_runs_stage: bad sizes");
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "k_update_runs serves rows <= 8 KiB");
    HX_HIP(this, hipSetDevice(device));
    size_t o = 64;
    ur_o_lm = o; o += al16((size_t)n_runs * 4);
    ur_o_off = o; o += al16(((size_t)n_runs + 1) * 4);
    ur_o_new = o; o += al16((size_t)n_ops * 4);
    ur_o_od = o; o += al16((size_t)n_ops * 4);
    ur_o_cnt = o; o += al16((size_t)n_runs * 4);
    ur_o_ids = o; o += al16((size_t)n_runs * stride * 4);
    ur_o_d = o; o += al16((size_t)n_runs * stride * 4);
    ur_end = o;
    if (o > mr.cap_lk) {
        HX_HIP(this, hipStreamSynchronize(stream));
        if (mr.h_lk) (void)hipHostFree(mr.h_lk);
        if (mr.d_lk) (void)hipFree(mr.d_lk);
        mr.h_lk = mr.d_lk = nullptr; mr.cap_lk = 0;
        const size_t n = o * 2;
        HX_HIP(this, hipHostMalloc((void **)&mr.h_lk, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&mr.d_lk, n));
        mr.cap_lk = n;
    }
    uint8_t *h = mr.h_lk;
    memset(h, 0, 64);
    *ids = (uint32_t *)(h + ur_o_ids); *d = (float *)(h + ur_o_d); *cnt = (uint32_t *)(h + ur_o_cnt); *lm = (uint32_t *)(h + ur_o_lm);
    *op_off = (uint32_t *)(h + ur_o_off); *op_new = (uint32_t *)(h + ur_o_new); *op_d = (float *)(h + ur_o_od);
    ur_n = n_runs; ur_stride = stride;
    return HX_OK;
}

int hx_engine::update_runs_run(uint64_t *n_pairs)
{
    HxMirror &mr = mirror;
    if (ur_n == 0) return fail(HX_E_STATE, "update_runs_run without update_runs_stage");
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipMemcpyAsync(mr.d_lk, mr.h_lk, ur_end, hipMemcpyHostToDevice, stream));
    UpdRunParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.cap = dtype == HX_SPARSE ? (uint32_t)std::min(dim, HX_SPARSE_MAX_NNZ) : 0u; p.n_runs = ur_n; p.stride = ur_stride;
    p.ids = (uint32_t *)(mr.d_lk + ur_o_ids); p.d = (float *)(mr.d_lk + ur_o_d); p.cnt = (uint32_t *)(mr.d_lk + ur_o_cnt);
    p.lm = (const uint32_t *)(mr.d_lk + ur_o_lm); p.op_off = (const uint32_t *)(mr.d_lk + ur_o_off); p.op_new = (const uint32_t *)(mr.d_lk + ur_o_new);
    p.op_d = (const float *)(mr.d_lk + ur_o_od); p.n_pairs = (unsigned long long *)mr.d_lk;
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_update_runs<OpF32<K>>(this, p)
#define F16C(K) ls = launch_update_runs<OpF16<K>>(this, p)
    if (dtype == HX_SPARSE) ls = metric == HX_L2SQ ? launch_update_runs<OpSparse<K_L2>>(this, p) : metric == HX_NEG_IP ? launch_update_runs<OpSparse<K_IP>>(this, p) : launch_update_runs<OpSparse<K_L1>>(this, p);
    else HX_DISPATCH(this, F32C, F16C, ls = launch_update_runs<OpHamming>(this, p), ls = launch_update_runs<OpJaccard>(this, p));
#undef F32C
#undef F16C
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk + ur_o_cnt, mr.d_lk + ur_o_cnt, ur_end - ur_o_cnt, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    ur_n = 0;
    return HX_OK;
}

template <class OP>
static hipError_t launch_update_index(hx_engine *e, const UpdParams &p)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (64 * 5 + 72) * 4 + nch * 1024;
    if constexpr (OP::kSparse) hipLaunchKernelGGL((k_update_index<OP, 64>), dim3(p.n_ops), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 128) hipLaunchKernelGGL((k_update_index<OP, 8>), dim3(p.n_ops), dim3(64), lds, e->stream, p);
    else if (e->pitch <= 512) hipLaunchKernelGGL((k_update_index<OP, 32>), dim3(p.n_ops), dim3(64), lds, e->stream, p);
    else hipLaunchKernelGGL((k_update_index<OP, 64>), dim3(p.n_ops), dim3(64), lds, e->stream, p);
    return hipGetLastError();
}

// staging of one wave of get_update_index ops: pinned arrays the caller fills (ids / d: `stride` entries per op), then update_index_run
int hx_engine::update_index_stage(uint32_t n_ops, uint32_t stride, uint32_t **ids, float **d, float **new_d, uint32_t **cnt)
{
    HxMirror &mr = mirror;
    if (n_ops == 0 || stride == 0 || stride > 64) return fail(HX_E_ARG, "update_index_stage: bad sizes");
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "k_update_index serves rows <= 8 KiB");
    HX_HIP(this, hipSetDevice(device));
    size_t o = 64;
    ui_o_nd = o; o += al16((size_t)n_ops * 4);
    ui_o_cnt = o; o += al16((size_t)n_ops * 4);
    ui_o_ids = o; o += al16((size_t)n_ops * stride * 4);
    ui_o_d = o; o += al16((size_t)n_ops * stride * 4);
    ui_in = o;
    ui_o_slot = o; o += al16((size_t)n_ops * 4);
    if (o > mr.cap_lk) {
        HX_HIP(this, hipStreamSynchronize(stream));
        if (mr.h_lk) (void)hipHostFree(mr.h_lk);
        if (mr.d_lk) (void)hipFree(mr.d_lk);
        mr.h_lk = mr.d_lk = nullptr; mr.cap_lk = 0;
        const size_t n = o * 2;
        HX_HIP(this, hipHostMalloc((void **)&mr.h_lk, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&mr.d_lk, n));
        mr.cap_lk = n;
    }
    uint8_t *h = mr.h_lk;
    memset(h, 0, 64);
    *new_d = (float *)(h + ui_o_nd); *cnt = (uint32_t *)(h + ui_o_cnt); *ids = (uint32_t *)(h + ui_o_ids); *d = (float *)(h + ui_o_d);
    ui_n = n_ops; ui_stride = stride;
    return HX_OK;
}

int hx_engine::update_index_run(const int32_t **slot_out, uint64_t *n_pairs)
{
    HxMirror &mr = mirror;
    if (ui_n == 0) return fail(HX_E_STATE, "update_index_run without update_index_stage");
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipMemcpyAsync(mr.d_lk, mr.h_lk, ui_in, hipMemcpyHostToDevice, stream));
    UpdParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.cap = dtype == HX_SPARSE ? (uint32_t)std::min(dim, HX_SPARSE_MAX_NNZ) : 0u; p.n_ops = ui_n; p.stride = ui_stride;
    p.ids = (const uint32_t *)(mr.d_lk + ui_o_ids); p.d = (const float *)(mr.d_lk + ui_o_d); p.new_d = (const float *)(mr.d_lk + ui_o_nd);
    p.cnt = (const uint32_t *)(mr.d_lk + ui_o_cnt); p.slot = (int32_t *)(mr.d_lk + ui_o_slot); p.n_pairs = (unsigned long long *)mr.d_lk;
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_update_index<OpF32<K>>(this, p)
#define F16C(K) ls = launch_update_index<OpF16<K>>(this, p)
    if (dtype == HX_SPARSE) ls = metric == HX_L2SQ ? launch_update_index<OpSparse<K_L2>>(this, p) : metric == HX_NEG_IP ? launch_update_index<OpSparse<K_IP>>(this, p) : launch_update_index<OpSparse<K_L1>>(this, p);
    else HX_DISPATCH(this, F32C, F16C, ls = launch_update_index<OpHamming>(this, p), ls = launch_update_index<OpJaccard>(this, p));
#undef F32C
#undef F16C
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk + ui_o_slot, mr.d_lk + ui_o_slot, (size_t)ui_n * 4, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    *slot_out = (const int32_t *)(mr.h_lk + ui_o_slot);
    ui_n = 0;
    return HX_OK;
}

template <class OP, int LPR, int SLOTS>
static hipError_t launch_links_cached_s(hx_engine *e, const LinksParams &p)
{
    constexpr size_t AR = SLOTS + 8, MT = (SLOTS + 1) * SLOTS / 2;
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (MT * 2 + AR * 11 + 64 + 64) * 4 + nch * 1024;
    hipLaunchKernelGGL((k_links_cached<OP, LPR, SLOTS>), dim3(p.n_groups), dim3(64), lds, e->stream, p, e->mirror.d_pm, e->mirror.d_pm_valid);
    return hipGetLastError();
}
template <class OP, int LPR>
static hipError_t launch_links_cached_lpr(hx_engine *e, const LinksParams &p)
{   // lists of up to 32 slots (m <= 16) or 64 (m <= 32)
    return 2 * p.m <= 32 ? launch_links_cached_s<OP, LPR, 32>(e, p) : launch_links_cached_s<OP, LPR, 64>(e, p);
}
template <class OP, int LPR, int SLOTS>
static hipError_t launch_links_hub_s(hx_engine *e, const LinksParams &p)
{
    constexpr size_t AR = SLOTS + 8, MT = (SLOTS + 1) * SLOTS / 2;
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (MT * 2 + AR * 4 + 16) * 4 + (size_t)HUB_W * ((AR * 6 + 64 + 64) * 4 + nch * 1024);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t st = hipFuncSetAttribute((const void *)k_links_hub<OP, LPR, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (st != hipSuccess) return st;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_links_hub<OP, LPR, SLOTS>), dim3(p.n_groups), dim3(64 * HUB_W), lds, e->stream, p, e->mirror.d_pm, e->mirror.d_pm_valid);
    return hipGetLastError();
}
template <class OP, int LPR>
static hipError_t launch_links_hub_lpr(hx_engine *e, const LinksParams &p)
{
    return 2 * p.m <= 32 ? launch_links_hub_s<OP, LPR, 32>(e, p) : launch_links_hub_s<OP, LPR, 64>(e, p);
}
template <class OP>
static hipError_t launch_links_hub(hx_engine *e, const LinksParams &p)
{
    if (e->pitch <= 128) return launch_links_hub_lpr<OP, 8>(e, p);
    if (e->pitch <= 512) return launch_links_hub_lpr<OP, 32>(e, p);
    return launch_links_hub_lpr<OP, 64>(e, p);
}
template <class OP>
static hipError_t launch_links_cached(hx_engine *e, const LinksParams &p)
{   // lanes per row by payload: <= 128 B (bit(1024), tiny test vectors) 8, <= 512 B (vector(128)) 32, else the whole wave
    if (e->pitch <= 128) return launch_links_cached_lpr<OP, 8>(e, p);
    if (e->pitch <= 512) return launch_links_cached_lpr<OP, 32>(e, p);
    return launch_links_cached_lpr<OP, 64>(e, p);
}

template <class OP>
static hipError_t launch_pm_fill(hx_engine *e, const LinksParams &p)
{
    const size_t lds = 2 * (size_t)LK_MAXN * 1024u + (40 + HX_PAIR_SLAB) * 4;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t s = hipFuncSetAttribute((const void *)k_pm_fill<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (s != hipSuccess) return s;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_pm_fill<OP>), dim3(p.n_groups), dim3(HX_PAIR_WG), lds, e->stream, p, e->mirror.d_pm, e->mirror.d_pm_valid);
    return hipGetLastError();
}

template <class OP>
static hipError_t launch_links(hx_engine *e, const LinksParams &p)
{
    const size_t lds = 2 * (size_t)LK_MAXN * 1024u + (40 * 4 + HX_PAIR_SLAB + 40 * 2 + 8) * 4;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t s = hipFuncSetAttribute((const void *)k_links<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (s != hipSuccess) return s;
        attr_set = true;
    }
    if (getenv("HX_DEBUG")) {
        static thread_local bool once = false;
        if (!once) { once = true; int nb = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_links<OP>, HX_PAIR_WG, lds);
            fprintf(stderr, "[hx] k_links: dynamic LDS %zu B, occupancy API says %d blocks/CU\n", lds, nb); }
    }
    hipLaunchKernelGGL((k_links<OP>), dim3(p.n_groups), dim3(HX_PAIR_WG), lds, e->stream, p);
    return hipGetLastError();
}

int hx_engine::links_run(uint32_t n_groups, const uint32_t *target, const uint32_t *layer, const uint32_t *op_off,
                         const uint32_t *op_new, const float *op_d, const uint32_t **out_ids, const float **out_d, const uint32_t **out_cnt, uint64_t *n_pairs,
                         bool want_lists)
{
    HxMirror &mr = mirror;
    if (n_groups == 0) return HX_OK;
    if (2 * mr.m + 1 > LK_MAXN) return fail(HX_E_ARG, "k_links handles m <= 16");
    HX_HIP(this, hipSetDevice(device));
    static const bool use_cache = !(getenv("HX_LINKS_NOCACHE") && atoi(getenv("HX_LINKS_NOCACHE")));
    if (use_cache && 2 * mr.m == LC_SLOTS && pitch <= FUSED_MAXCH * 1024u && mr.cap_pm < mr.cap) {
        // pair-matrix cache for every layer-0 list: 496 f32 per element
        float *npm = nullptr; uint8_t *nv = nullptr;
        HX_HIP(this, hipMalloc((void **)&npm, (size_t)mr.cap * LC_TRI * sizeof(float)));
        HX_HIP(this, hipMalloc((void **)&nv, mr.cap));
        HX_HIP(this, hipMemsetAsync(nv, 0, mr.cap, stream));
        if (mr.d_pm && mr.cap_pm) {
            HX_HIP(this, hipMemcpyAsync(npm, mr.d_pm, (size_t)mr.cap_pm * LC_TRI * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HX_HIP(this, hipMemcpyAsync(nv, mr.d_pm_valid, mr.cap_pm, hipMemcpyDeviceToDevice, stream));
        }
        HX_HIP(this, hipStreamSynchronize(stream));
        if (mr.d_pm) (void)hipFree(mr.d_pm);
        if (mr.d_pm_valid) (void)hipFree(mr.d_pm_valid);
        mr.d_pm = npm; mr.d_pm_valid = nv; mr.cap_pm = mr.cap;
    }
    const bool cached_kernel = use_cache && mr.d_pm != nullptr && 2 * mr.m == LC_SLOTS;
    const uint32_t n_ops = op_off[n_groups], lm0 = 2 * mr.m;
    size_t o = 0;
    const size_t o_ctr = o; o += 64;
    const size_t o_tg = o; o += al16((size_t)n_groups * 4);
    const size_t o_ly = o; o += al16((size_t)n_groups * 4);
    const size_t o_off = o; o += al16(((size_t)n_groups + 1) * 4);
    const size_t o_new = o; o += al16((size_t)n_ops * 4);
    const size_t o_od = o; o += al16((size_t)n_ops * 4);
    const size_t o_gmap = o; o += al16((size_t)n_groups * 4);   // launch order: hub lists first, then the rest
    const size_t in_bytes = o;
    const size_t o_cnt = o; o += al16((size_t)n_groups * 4);
    const size_t o_ids = o; o += al16((size_t)n_groups * lm0 * 4);
    const size_t o_d = o; o += al16((size_t)n_groups * lm0 * 4);
    if (o > mr.cap_lk) {
        if (mr.h_lk) (void)hipHostFree(mr.h_lk);
        if (mr.d_lk) (void)hipFree(mr.d_lk);
        mr.h_lk = mr.d_lk = nullptr; mr.cap_lk = 0;
        const size_t n = o * 2;
        HX_HIP(this, hipHostMalloc((void **)&mr.h_lk, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&mr.d_lk, n));
        mr.cap_lk = n;
    }
    uint8_t *h = mr.h_lk;
    memset(h + o_ctr, 0, 64);
    memcpy(h + o_tg, target, (size_t)n_groups * 4); memcpy(h + o_ly, layer, (size_t)n_groups * 4);
    memcpy(h + o_off, op_off, ((size_t)n_groups + 1) * 4);
    memcpy(h + o_new, op_new, (size_t)n_ops * 4); memcpy(h + o_od, op_d, (size_t)n_ops * 4);
    uint32_t n_hub = 0;
    {   // lists with a long chain of ops go to the speculative multi-wave kernel (k_links_hub); HX_HUB_MIN=0 disables it
        static const uint32_t hub_min = getenv("HX_HUB_MIN") ? (uint32_t)atoi(getenv("HX_HUB_MIN")) : 48u;
        uint32_t *gm = (uint32_t *)(h + o_gmap);
        if (cached_kernel && hub_min) for (uint32_t g = 0; g < n_groups; g++) if (op_off[g + 1] - op_off[g] >= hub_min) gm[n_hub++] = g;
        uint32_t k = n_hub;
        if (n_hub) { for (uint32_t g = 0; g < n_groups; g++) if (op_off[g + 1] - op_off[g] < hub_min) gm[k++] = g; }
        else for (uint32_t g = 0; g < n_groups; g++) gm[g] = g;
    }
    HX_HIP(this, hipMemcpyAsync(mr.d_lk, h, in_bytes, hipMemcpyHostToDevice, stream));
    LinksParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.m = mr.m;
    p.l0_ids = mr.d_l0_ids; p.l0_d = mr.d_l0_d; p.l0_cnt = mr.d_l0_cnt; p.up_block = mr.d_up_block; p.up_ids = mr.d_up_ids; p.up_d = mr.d_up_d; p.up_cnt = mr.d_up_cnt;
    p.n_groups = n_groups; p.target = (const uint32_t *)(mr.d_lk + o_tg); p.layer = (const uint32_t *)(mr.d_lk + o_ly);
    p.op_off = (const uint32_t *)(mr.d_lk + o_off); p.op_new = (const uint32_t *)(mr.d_lk + o_new); p.op_d = (const float *)(mr.d_lk + o_od);
    p.out_ids = (uint32_t *)(mr.d_lk + o_ids); p.out_d = (float *)(mr.d_lk + o_d); p.out_cnt = (uint32_t *)(mr.d_lk + o_cnt);
    p.n_pairs = (unsigned long long *)(mr.d_lk + o_ctr);
    p.gmap = nullptr; p.xrec = nullptr; p.xrec_words = 0; p.wtab = nullptr; p.wt_size = 0; p.wt_base = 0; p.wt_n = 0; p.wt_valid = nullptr;
    { const char *dv = getenv("HX_LK_DBG"); p.dbg = dv ? (uint32_t)atoi(dv) : 0u; }
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
    if (cached_kernel) {
        if (n_hub) {
            LinksParams ph = p; ph.n_groups = n_hub; ph.gmap = (const uint32_t *)(mr.d_lk + o_gmap);
#define F32C(K) ls = launch_links_hub<OpF32<K>>(this, ph)
#define F16C(K) ls = launch_links_hub<OpF16<K>>(this, ph)
            HX_DISPATCH(this, F32C, F16C, ls = launch_links_hub<OpHamming>(this, ph), ls = launch_links_hub<OpJaccard>(this, ph));
#undef F32C
#undef F16C
            HX_HIP(this, ls);
            p.gmap = (const uint32_t *)(mr.d_lk + o_gmap) + n_hub; p.n_groups = n_groups - n_hub;
        }
        if (p.n_groups) {
#define F32C(K) ls = launch_links_cached<OpF32<K>>(this, p)
#define F16C(K) ls = launch_links_cached<OpF16<K>>(this, p)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links_cached<OpHamming>(this, p), ls = launch_links_cached<OpJaccard>(this, p));
#undef F32C
#undef F16C
        }
    } else {
#define F32C(K) ls = launch_links<OpF32<K>>(this, p)
#define F16C(K) ls = launch_links<OpF16<K>>(this, p)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links<OpHamming>(this, p), ls = launch_links<OpJaccard>(this, p));
#undef F32C
#undef F16C
    }
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(h + o_ctr, mr.d_lk + o_ctr, 64, hipMemcpyDeviceToHost, stream));
    if (want_lists) HX_HIP(this, hipMemcpyAsync(h + o_cnt, mr.d_lk + o_cnt, o - o_cnt, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    *out_cnt = (const uint32_t *)(h + o_cnt); *out_ids = (const uint32_t *)(h + o_ids); *out_d = (const float *)(h + o_d);
    unsigned long long np; memcpy(&np, h + o_ctr, 8);
    if (FUSED_TIMERS_ON && (p.dbg & 8u)) { unsigned long long t[7]; memcpy(t, h + o_ctr, 56); fprintf(stderr, "[hx] k_links_cached groups %u ops %u: ticks matrix-fill %llu sort %llu walk %llu lazy-nd %llu rebuild %llu; whole kernel per wave %llu\n", n_groups, n_ops, t[1], t[2], t[3], t[4], t[5], t[6]); }
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    return HX_OK;
}

int hx_engine::links_run_grouped(uint32_t n_ops, const unsigned long long *keys, const uint32_t *op_new, const float *op_d, uint64_t *n_pairs, uint32_t stats[2],
                                 bool on_device, bool want_xrec)
{
    HxMirror &mr = mirror;
    stats[0] = stats[1] = 0;
    if (n_pairs) *n_pairs = 0;
    xl_records = 0;
    if (n_ops == 0) return HX_OK;
    if (2 * mr.m > 64 || pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "device-side op grouping serves m <= 32 and rows <= 8 KiB");
    HX_HIP(this, hipSetDevice(device));
    const bool use_pm = 2 * mr.m == LC_SLOTS;                     // the resident pair-matrix cache is laid out for 32-slot lists
    if (use_pm && mr.cap_pm < mr.cap) {                           // pair-matrix cache for every layer-0 list (as in links_run)
        float *npm = nullptr; uint8_t *nv = nullptr;
        HX_HIP(this, hipMalloc((void **)&npm, (size_t)mr.cap * LC_TRI * sizeof(float)));
        HX_HIP(this, hipMalloc((void **)&nv, mr.cap));
        HX_HIP(this, hipMemsetAsync(nv, 0, mr.cap, stream));
        if (mr.d_pm && mr.cap_pm) {
            HX_HIP(this, hipMemcpyAsync(npm, mr.d_pm, (size_t)mr.cap_pm * LC_TRI * sizeof(float), hipMemcpyDeviceToDevice, stream));
            HX_HIP(this, hipMemcpyAsync(nv, mr.d_pm_valid, mr.cap_pm, hipMemcpyDeviceToDevice, stream));
        }
        HX_HIP(this, hipStreamSynchronize(stream));
        if (mr.d_pm) (void)hipFree(mr.d_pm);
        if (mr.d_pm_valid) (void)hipFree(mr.d_pm_valid);
        mr.d_pm = npm; mr.d_pm_valid = nv; mr.cap_pm = mr.cap;
    }
    if (mr.cap_lk < 256) {                                      // counters live in the links staging buffers
        if (mr.h_lk) (void)hipHostFree(mr.h_lk);
        if (mr.d_lk) (void)hipFree(mr.d_lk);
        mr.h_lk = mr.d_lk = nullptr; mr.cap_lk = 0;
        HX_HIP(this, hipHostMalloc((void **)&mr.h_lk, 4096, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&mr.d_lk, 4096));
        mr.cap_lk = 4096;
    }
    static const uint32_t hub_min = getenv("HX_HUB_MIN") ? (uint32_t)atoi(getenv("HX_HUB_MIN")) : 48u;
    static const bool prefill = !(getenv("HX_PM_PREFILL") && atoi(getenv("HX_PM_PREFILL")) == 0);
    uint32_t c[5];
    int rc = on_device ? hx_group_run(this, n_ops, hub_min, grp, c, use_pm && prefill) : hx_group_ops(this, n_ops, keys, op_new, op_d, hub_min, grp, c, use_pm && prefill);
    if (rc) return rc;
    const uint32_t n_groups = c[0], n_hub = c[1], n_norm = c[2];
    stats[0] = n_groups; stats[1] = c[3];
    LinksParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.m = mr.m;
    p.l0_ids = mr.d_l0_ids; p.l0_d = mr.d_l0_d; p.l0_cnt = mr.d_l0_cnt; p.up_block = mr.d_up_block; p.up_ids = mr.d_up_ids; p.up_d = mr.d_up_d; p.up_cnt = mr.d_up_cnt;
    p.n_groups = n_groups; p.target = grp.tg; p.layer = grp.ly; p.op_off = grp.off; p.op_new = grp.op_new; p.op_d = grp.op_d; p.gmap = nullptr;
    p.out_ids = nullptr; p.out_d = nullptr; p.out_cnt = nullptr;
    p.xrec = nullptr; p.xrec_words = 0;
    p.wtab = nullptr; p.wt_size = 0; p.wt_base = 0; p.wt_n = 0; p.wt_valid = nullptr;
    if (on_device && bw.wt_size) { p.wtab = (const uint2 *)bw.d_wtab; p.wt_size = bw.wt_size; p.wt_base = bw.wt_base; p.wt_n = bw.wt_n; p.wt_valid = bw.d_wt_valid; }
    if (want_xrec) {
        const uint32_t xw = hx_xrec_words(mr.m);
        const size_t need = (size_t)n_groups * xw * 4;
        if (need > cap_xl) {
            HX_HIP(this, hipStreamSynchronize(stream));
            if (d_xl) (void)hipFree(d_xl);
            d_xl = nullptr; cap_xl = 0;
            HX_HIP(this, hipMalloc((void **)&d_xl, need * 2));
            cap_xl = need * 2;
        }
        p.xrec = d_xl; p.xrec_words = xw; xl_records = n_groups;
    }
    p.n_pairs = (unsigned long long *)mr.d_lk;
    { const char *dv = getenv("HX_LK_DBG"); p.dbg = dv ? (uint32_t)atoi(dv) : 0u; }
    HX_HIP(this, hipMemsetAsync(mr.d_lk, 0, 64, stream));
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
    float *pm_save = mr.d_pm;
    if (use_pm && prefill && c[4]) {                             // pair matrices of the full lists that are pruned for the first time (compacted by k_split)
        LinksParams pf = p; pf.n_groups = c[4]; pf.gmap = grp.gmap_fill;
#define F32C(K) ls = launch_pm_fill<OpF32<K>>(this, pf)
#define F16C(K) ls = launch_pm_fill<OpF16<K>>(this, pf)
        HX_DISPATCH(this, F32C, F16C, ls = launch_pm_fill<OpHamming>(this, pf), ls = launch_pm_fill<OpJaccard>(this, pf));
#undef F32C
#undef F16C
        HX_HIP(this, ls);
    }
    if (!use_pm) mr.d_pm = nullptr;                              // lists of other sizes are pruned from scratch (matrix in LDS only)
    if (n_hub) {
        LinksParams ph = p; ph.n_groups = n_hub; ph.gmap = grp.gmap_hub;
#define F32C(K) ls = launch_links_hub<OpF32<K>>(this, ph)
#define F16C(K) ls = launch_links_hub<OpF16<K>>(this, ph)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links_hub<OpHamming>(this, ph), ls = launch_links_hub<OpJaccard>(this, ph));
#undef F32C
#undef F16C
    }
    static const bool memo_lists = !(getenv("HX_LINKS_MEMO") && atoi(getenv("HX_LINKS_MEMO")) == 0);
    if (ls == hipSuccess && n_norm && !use_pm && 2 * mr.m > 32 && memo_lists) {
        // lists of 33..64 slots have no resident pair matrix: k_list_ops' lazily filled memo (hx_biglist.hip) instead of completing 2 016 pairs per touched list
        ListMirrorArgs a{};
        a.n_groups = n_norm; a.m = mr.m; a.target = grp.tg; a.layer = grp.ly; a.op_off = grp.off; a.op_new = grp.op_new; a.gmap = grp.gmap_norm; a.op_d = grp.op_d;
        a.l0_ids = mr.d_l0_ids; a.l0_d = mr.d_l0_d; a.l0_cnt = mr.d_l0_cnt; a.up_block = mr.d_up_block; a.up_ids = mr.d_up_ids; a.up_d = mr.d_up_d; a.up_cnt = mr.d_up_cnt;
        a.xrec = p.xrec; a.xrec_words = p.xrec_words; a.n_pairs = p.n_pairs;
        ls = hx_launch_list_ops_mirror(this, a);
    } else
    if (ls == hipSuccess && n_norm) {
        p.n_groups = n_norm; p.gmap = grp.gmap_norm;
#define F32C(K) ls = launch_links_cached<OpF32<K>>(this, p)
#define F16C(K) ls = launch_links_cached<OpF16<K>>(this, p)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links_cached<OpHamming>(this, p), ls = launch_links_cached<OpJaccard>(this, p));
#undef F32C
#undef F16C
    }
    mr.d_pm = pm_save;
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    return HX_OK;
}
