// hx_fused.inc.h -- included by hx_engine.hip.  Device-resident HNSW traversal: one wavefront per search.
//
// The lock-step host driver (hx_index.cpp) pays one host round trip per candidate expansion.  This file moves the
// whole of search_layer (graph/mod.rs:161-255 / scan.rs:302-448), the greedy descent and per-layer loop of
// find_element_neighbors (graph/mod.rs:355-427) / get_scan_items (scan.rs:458-530) and select_neighbors
// (graph/mod.rs:269-339) into ONE persistent kernel:
//   * a 64-thread workgroup (one wavefront) owns one insert or query from start to finish; workgroups pull tasks
//     from an atomic counter until none are left (every wave reaches the exit: the counter only grows);
//   * the candidate heap C, the result heap W, the entry-point / sorted-candidate array and the select lists live
//     in LDS; the heaps use the same Rust-std sift order as the host driver and the oracle, executed by lane 0;
//   * the query's 16-byte fragments stay in registers; an expansion reads the candidate's neighbour ids (coalesced),
//     test-and-sets a per-workgroup visited bitmap with L2 atomics, and evaluates the unvisited rows with the SAME
//     canonical summation order as K1/K2 (lane l owns bytes chunk*1024+16*l, xor butterfly), 4 rows in flight;
//   * the graph is read from a device mirror (ids only) that the host refreshes after each batch.
// Results (neighbour lists with distances / top-k) are therefore bit-identical to the lock-step path; the
// tests compare the two paths and the oracle.  A task whose candidate heap would overflow its LDS budget reports
// FS_OVERFLOW and is re-run by the lock-step path (still on the GPU kernels: there is no CPU fallback).

#include <cstdio>
#define FUSED_MAXCH 8          /* 1 KiB chunks per row: pitch <= 8192 B covers vector(2000), halfvec(4000), bit(64000) */
#ifndef FUSED_RB
#define FUSED_RB 4             /* rows in flight per wave ... */
#endif
#ifndef FUSED_MINW
#define FUSED_MINW 4
#endif
#ifndef FUSED_MINW_ITER
#define FUSED_MINW_ITER 3          /* the iterative-scan kernel carries more state: 168 VGPRs instead of spilling 230 */
#endif
#define FUSED_CG 3             /* ... times chunks of each requested at once */
#define FUSED_CCAP 8192u       /* candidate-heap capacity per search (LDS head + global spill) */
#define FUSED_MAXL 8           /* layers 0..7 handled on the device (P(level >= 8) = 16^-8 at m=16) */
enum { FS_OK = 0, FS_OVERFLOW = 1, FS_HOST = 2 };

struct FusedParams {
    const uint8_t *rows, *queries; uint32_t pitch, nch; uint64_t n_rows;
    const uint32_t *l0_ids; const uint16_t *l0_cnt; const int32_t *level;
    const uint32_t *up_block, *up_ids; const uint16_t *up_cnt;
    const float *l0_d, *up_d;                    // stored neighbour distances (the select phase reuses them)
    uint32_t m, entry; int32_t entry_level;
    uint32_t ntasks; const uint32_t *t_qsel; const int32_t *t_level;
    uint32_t ef, k, ccap, clds;                  // ccap: capacity of the candidate heap, its first clds entries in LDS
    uint2 *spill; uint32_t spill_stride;         // per-workgroup spill area of the candidate heap (entries)
    uint32_t *vis; uint64_t vis_words;            // per-workgroup visited set: open-addressing table of vis_words (power of 2) row ids
    uint32_t *next_task;
    uint32_t *out_ids; float *out_d; uint32_t *out_cnt; uint32_t *status;
    unsigned long long *n_dist;                   // [0] query-vs-row distances, [1] select distances, [2] max |C| seen
    // iterative scan (k_fused MODE 2): hnsw.iterative_scan relaxed_order (1) / strict_order (2), scan.rs:794-875
    uint32_t iter_mode, limit; long long max_tuples;
    const uint16_t *emask;                        // per element: bits 0-9 = which of its heap TIDs pass the filter, bits 12-15 = number of heap TIDs
    unsigned long long *disc; uint32_t disc_stride, disc_lds;   // per-workgroup tail of the `discarded` heap (entries), its LDS head
    uint32_t *out_tix;                            // which heap TID of the element each output is
    float *dsc;                                   // 64 floats of LDS scratch for the short-row distance path (set inside the kernels)
    uint32_t fdbg;                                // experiments (HX_F_DBG): 1 no pre-filter, 4 phase timers into n_dist[3..7]
};

struct FHeapItem { float d; uint32_t id; };
__device__ __forceinline__ uint2 fh_pack(float d, uint32_t id) { return make_uint2(__builtin_bit_cast(unsigned int, d), id); }
__device__ __forceinline__ float fh_d(const uint2 &v) { return __builtin_bit_cast(float, v.x); }

// Heap storage: the first `L` entries live in LDS, the rest in this workgroup's spill area in global memory (only lane 0
// touches a heap, and a thread sees its own stores in program order).  Deep heaps are rare and only their bottom level
// spills, so the common case never leaves LDS while the LDS budget per search stays small.
typedef __attribute__((address_space(3))) uint2 lds_uint2;     // LDS-qualified: keeps heap accesses ds_read/ds_write, never FLAT
struct HStore {
    lds_uint2 *lds; uint2 *glob; uint32_t L;
    __device__ __forceinline__ uint2 get(uint32_t i) const
    {
        if (i < L) return make_uint2(lds[i].x, lds[i].y);
        return glob[i - L];
    }
    __device__ __forceinline__ void set(uint32_t i, uint2 v) const
    {
        if (i < L) { lds[i].x = v.x; lds[i].y = v.y; } else glob[i - L] = v;
    }
};

// Rust std BinaryHeap; NEAREST: smallest distance on top.  Called by ONE lane.
template <bool NEAREST> struct FHeap {
    static __device__ __forceinline__ bool le(float a, float b) { return NEAREST ? !(b > a) : !(a > b); }
    static __device__ void sift_up(const HStore &h, uint32_t start, uint32_t pos)
    {
        const uint2 e = h.get(pos); const float ed = fh_d(e);
        while (pos > start) {
            const uint32_t parent = (pos - 1) >> 1;
            const uint2 pv = h.get(parent);
            if (le(ed, fh_d(pv))) break;
            h.set(pos, pv); pos = parent;
        }
        h.set(pos, e);
    }
    static __device__ void push(const HStore &h, uint32_t &len, uint2 c) { h.set(len, c); len++; sift_up(h, 0, len - 1); }
    static __device__ uint2 pop(const HStore &h, uint32_t &len)      // len > 0
    {
        uint2 item = h.get(len - 1); len--;
        if (len > 0) {
            const uint2 top = h.get(0); h.set(0, item); item = top;
            // sift_down_to_bottom(0)
            const uint32_t end = len; uint32_t pos = 0;
            const uint2 e = h.get(0);
            uint32_t child = 1;
            while (end >= 2 && child <= end - 2) {
                const uint2 a = h.get(child), b = h.get(child + 1);
                const bool right = le(fh_d(a), fh_d(b));
                h.set(pos, right ? b : a); pos = child + (right ? 1u : 0u); child = 2 * pos + 1;
            }
            if (child == end - 1) { h.set(pos, h.get(child)); pos = child; }
            h.set(pos, e);
            sift_up(h, 0, pos);
        }
        return item;
    }
};

// The same Rust std BinaryHeap, executed by the WHOLE wavefront on an LDS array (every lane calls with identical
// arguments and gets identical results).  Serial heap code is what a GPU is worst at -- measured 3 us per pop, a third of
// an expansion step -- but the heap's moves are more parallel than they look:
//   * sift_up's path (the ancestors of the new slot) is known up front: lane k reads ancestor k, one ballot finds where
//     the walk stops, and the lanes below shift their ancestors down in one store;
//   * sift_down_to_bottom's path depends only on the heap's contents, never on the moving element: it is a pointer
//     chase of one LDS read (both children) per level with lane k latching level k; the closing sift_up along that
//     same path is again one ballot, and all the moves are one store.
// The resulting array is the one the serial algorithm leaves, element for element (ties included).
#define F_WSYNC() asm volatile("" ::: "memory")     /* LDS ops of one wave execute in order; only the compiler must not reorder */
// heap storage for PHeap: plain LDS, or an LDS head + a tail in this workgroup's global area (the `discarded` heap of an
// iterative scan holds every visited element that is not a result: tens of thousands of entries).  The global part is read
// and written with L1-bypassing 64-bit accesses because different lanes of the wave read what other lanes wrote.
struct LStore {
    lds_uint2 *A;
    static constexpr bool kGlobal = false;
    __device__ __forceinline__ uint2 ld(uint32_t i) const { return make_uint2(A[i].x, A[i].y); }
    __device__ __forceinline__ void st(uint32_t i, uint2 v) const { A[i].x = v.x; A[i].y = v.y; }
};
struct GStore {
    lds_uint2 *A; unsigned long long *G; uint32_t L;
    static constexpr bool kGlobal = true;
    __device__ __forceinline__ uint2 ld(uint32_t i) const
    {
        if (i < L) return make_uint2(A[i].x, A[i].y);
        const unsigned long long v = __hip_atomic_load(G + (i - L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
    }
    __device__ __forceinline__ void st(uint32_t i, uint2 v) const
    {
        if (i < L) { A[i].x = v.x; A[i].y = v.y; }
        else __hip_atomic_store(G + (i - L), (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
template <bool NEAREST> struct PHeap {
    static __device__ __forceinline__ bool le(float a, float b) { return NEAREST ? !(b > a) : !(a > b); }
    static __device__ __forceinline__ uint2 ld(lds_uint2 *A, uint32_t i) { return make_uint2(A[i].x, A[i].y); }
    template <class ST> static __device__ __forceinline__ void sync(const ST &)
    {   // LDS ops of one wave execute in order, and so do its vector-memory ops to one address at the L2 both sides go to
        // (L1 bypassed): wavefront-scope ordering needs no wait on gfx9 (LLVM AMDGPU memory model), only a compiler barrier
        F_WSYNC();
    }
    template <class ST> static __device__ __forceinline__ void push(const ST &S, uint32_t &len, uint2 c, uint32_t lane)
    {
        sync(S);
        const uint32_t pos1 = len + 1u; len++;                         // 1-based slot of the new element
        const uint32_t depth = 31u - (uint32_t)__builtin_clz(pos1);    // number of ancestors
        const bool anc = lane >= 1u && lane <= depth;                  // lane k holds the k-th ancestor
        uint2 v = make_uint2(0u, 0u);
        if (anc) v = S.ld((pos1 >> lane) - 1u);
        const unsigned long long sm = __ballot(anc && le(fh_d(c), fh_d(v)));   // sift_up breaks at the first such ancestor
        const uint32_t t = sm ? (uint32_t)__builtin_ctzll(sm) : depth + 1u;
        if (anc && lane < t) S.st((pos1 >> (lane - 1u)) - 1u, v);      // ancestors below the stop move down one level
        if (lane == 0u) S.st((pos1 >> (t - 1u)) - 1u, c);
        sync(S);
    }
    template <class ST> static __device__ __forceinline__ uint2 pop(const ST &S, uint32_t &len, uint32_t lane)   // len > 0
    {
        sync(S);
        const uint2 last = S.ld(len - 1u); len--;
        if (len == 0u) return last;
        const uint2 top = S.ld(0u);
        const uint32_t end = len;
        uint32_t pos = 0u, child = 1u, k = 0u;
        uint32_t myP = 0u, myC = 0u; uint2 myV = make_uint2(0u, 0u);   // lane k: path slot k, path slot k+1 and its old value
        while (end >= 2u && child <= end - 2u) {
            const uint2 a = S.ld(child), b = S.ld(child + 1u);
            const bool right = le(fh_d(a), fh_d(b));
            const uint2 cv = right ? b : a; const uint32_t cp = child + (right ? 1u : 0u);
            if (lane == k) { myP = pos; myC = cp; myV = cv; }
            pos = cp; child = 2u * pos + 1u; k++;
        }
        if (child == end - 1u) {
            const uint2 a = S.ld(child);
            if (lane == k) { myP = pos; myC = child; myV = a; }
            k++;
        }
        // the moved element climbs back from the bottom of the path while it beats its parent: it ends in path slot t
        const unsigned long long sm = __ballot(lane < k && le(fh_d(last), fh_d(myV)));
        const uint32_t t = sm ? 64u - (uint32_t)__builtin_clzll(sm) : 0u;
        if (lane < t) S.st(myP, myV);
        if (t == 0u) { if (lane == 0u) S.st(0u, last); }
        else if (lane == t - 1u) S.st(myC, last);
        sync(S);
        return top;
    }
    static __device__ __forceinline__ void push(lds_uint2 *A, uint32_t &len, uint2 c, uint32_t lane) { push(LStore{A}, len, c, lane); }
    static __device__ __forceinline__ uint2 pop(lds_uint2 *A, uint32_t &len, uint32_t lane) { return pop(LStore{A}, len, lane); }
};

// visited set (HashSet<usize> of graph/mod.rs:171): a per-workgroup open-addressing table of row ids in global memory
// (32-64 KB per wave), organised as 16-byte BUCKETS of four ids.  A membership test is ONE 16-byte load (bypassing the
// vector L1, because inserts are L2 atomics) of the key's bucket: the key is there, or the bucket still has an empty
// slot (=> the key is absent: buckets only ever fill up, slots x,y,z,w in order), or -- rarely -- the bucket is full and
// the next one is probed.  An insert is an atomicCAS on the first empty slot; its result is needed only to detect that
// another lane of the same instruction took the slot, so the caller may look at it later (after the row loads of the
// expansion have been issued) and re-insert then: the test costs one memory hop instead of one per probe.
#define VIS_EMPTY 0xffffffffu
__device__ __forceinline__ uint32_t vis_mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// true: key present.  false: *slot = the empty slot the key belongs in
__device__ __forceinline__ bool vis_lookup(uint32_t *tab, uint32_t bmask, uint32_t key, uint32_t *&slot)
{
    uint32_t b = vis_mix(key) & bmask;
    for (;;) {
        const unsigned long long *bp = (const unsigned long long *)(tab + 4u * b);
        const unsigned long long lo = __hip_atomic_load(bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long hi = __hip_atomic_load(bp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t x = (uint32_t)lo, y = (uint32_t)(lo >> 32), z = (uint32_t)hi, w = (uint32_t)(hi >> 32);
        if (x == key || y == key || z == key || w == key) return true;
        const int e = x == VIS_EMPTY ? 0 : (y == VIS_EMPTY ? 1 : (z == VIS_EMPTY ? 2 : (w == VIS_EMPTY ? 3 : -1)));
        if (e >= 0) { slot = tab + 4u * b + (uint32_t)e; return false; }
        b = (b + 1u) & bmask;
    }
}
// deferred half of an insert: `old` is what the atomicCAS on `slot` returned; re-insert while another lane won the slot
__device__ __forceinline__ void vis_settle(uint32_t *tab, uint32_t bmask, uint32_t key, uint32_t *slot, uint32_t old)
{
    while (old != VIS_EMPTY) {
        (void)vis_lookup(tab, bmask, key, slot);
        old = atomicCAS(slot, VIS_EMPTY, key);
    }
}
__device__ __forceinline__ bool vis_test_and_set(uint32_t *tab, uint32_t bmask, uint32_t key)
{
    uint32_t *slot = nullptr;
    if (vis_lookup(tab, bmask, key, slot)) return true;
    vis_settle(tab, bmask, key, slot, atomicCAS(slot, VIS_EMPTY, key));
    return false;
}

struct FusedCtx {
    uint2 *C, *W, *EP, *RES, *RL, *DL; uint32_t *IDS, *CTL; uint8_t *QV, *EV; HStore CH, WH;
    uint32_t *vis; uint32_t lane; uint32_t status;
    GStore DS; lds_uint2 *DP, *WS; uint32_t *LV; uint32_t dlen, vcount;           // iterative scan: `discarded` min-heap, visited ids so far (the set survives resumes)
    unsigned long long nd0, nd1; uint32_t cmax;
    uint32_t tph[14];  // [13] select phase; diagnostic phase clocks (HX_F_DBG & 4): pop, list fetch, visited, compaction, distances, settle+prefilter, replay; [7] expansions, [8] heap pushes
};

// parks one vector (row or query slot) in LDS, chunk-major: bytes [c*1024 + 16*lane, +16); zero past the pitch
__device__ __forceinline__ void f_park(const FusedParams &p, const uint8_t *src, uint32_t lane, uint8_t *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        u4 v = {0u, 0u, 0u, 0u};
        if (off < p.pitch) v = *(const u4 *)(src + off);
        *(u4 *)(dst + off) = v;
    }
    __syncthreads();
}

// f_park for code that runs in ONE wave of a multi-wave workgroup (no workgroup barrier)
__device__ __forceinline__ void f_park_w(const FusedParams &p, const uint8_t *src, uint32_t lane, uint8_t *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        u4 v = {0u, 0u, 0u, 0u};
        if (off < p.pitch) v = *(const u4 *)(src + off);
        *(u4 *)(dst + off) = v;
    }
    F_WSYNC();
}

// f_park without the trip through registers: gfx950's global_load_lds writes each lane's 16 bytes straight to
// LDS (destination = wave-uniform base + 16 * lane: exactly the parked layout) and completes asynchronously under vmcnt, so the
// NEXT select candidate's row travels while the current one is being compared.  Lanes past the row's end store zeros themselves.
__device__ __forceinline__ void f_park_async(const FusedParams &p, const uint8_t *src, uint32_t lane, uint8_t *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        if (off < p.pitch)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)(src + off),
                                             (__attribute__((address_space(3))) uint32_t *)(dst + c * 1024u), 16, 0, 0);
        else *(u4 *)(dst + off) = u4{0u, 0u, 0u, 0u};
    }
}

// Short rows (payload <= 512 B: bit(1024), vector(128), the reference's 3-d tests): a 64-lane wave per row would leave
// most lanes idle and cost one memory hop per FUSED_RB rows.  Here LPR = 8/16/32 lanes share a row (as K1 does), 64/LPR rows
// are read by ONE load instruction, and up to eight such instructions are in flight, so a whole neighbour list is one hop.
// The bits are the canonical ones: the lanes a short row does not reach contribute +0.0 partials in the 64-lane order.
template <class OP, int LPR>
__device__ __forceinline__ float f_dist_small(const FusedParams &p, const uint8_t *qv, const uint32_t *ids, uint32_t n, uint32_t lane)
{
    constexpr int R = 64 / LPR, PF = 8;
    const uint32_t g = lane / LPR, loff = (lane % LPR) * 16u;
    const bool in = loff < p.pitch;
    u4 q = {0u, 0u, 0u, 0u};
    if (in) q = *(const u4 *)(qv + loff);
    for (uint32_t j0 = 0; j0 < n; j0 += R * PF) {
        u4 rv[PF];
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const uint32_t j = j0 + (uint32_t)k * R + g;
            u4 v = {0u, 0u, 0u, 0u};
            if (in && j < n) v = *(const u4 *)(p.rows + (size_t)ids[j] * p.pitch + loff);
            rv[k] = v;
        }
#pragma unroll
        for (int k = 0; k < PF; k++) {
            const uint32_t j = j0 + (uint32_t)k * R + g;
            if (j0 + (uint32_t)k * R < n) {                       // wave-uniform: this pass holds at least one row
                typename OP::acc_t acc; OP::init(acc); OP::add(acc, q, rv[k]);
                const float d = OP::template finish<LPR>(acc);
                if (loff == 0u && j < n) p.dsc[j] = d;
            }
        }
    }
    F_WSYNC();
    const float mine = lane < n ? p.dsc[lane] : 0.0f;
    F_WSYNC();
    return mine;
}
// rows evaluated per early-exit step of check_element_closer: eight on the short-row path, else FUSED_RB
template <int LPR> __device__ __forceinline__ constexpr uint32_t f_step_rows() { return LPR < 64 ? 8u : (uint32_t)FUSED_RB; }

// distances from the vector parked at `qv` (LDS) to rows ids[0..n) (LDS); lane j (< 64) returns d(q, ids[j]); n <= 64.
// FUSED_RB rows x FUSED_CG 1-KiB chunks are requested at once (one HBM latency per row batch at d <= 768 f32), then
// consumed chunk by chunk in ascending order -- the canonical per-lane order.
template <class OP, int LPR, int RB = FUSED_RB>
__device__ __forceinline__ float f_dist_batch(const FusedParams &p, const uint8_t *qv, const uint32_t *ids, uint32_t n, uint32_t lane, uint32_t *tk = nullptr)
{
    if constexpr (LPR < 64) return f_dist_small<OP, LPR>(p, qv, ids, n, lane);
    float mine = 0.0f;
    unsigned long long tq = tk ? __builtin_amdgcn_s_memtime() : 0ull;
#define FD_TICK(k) do { if (tk) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk[k] += (uint32_t)(t_ - tq); tq = t_; } } while (0)
    const uint32_t loff = lane * 16u;
    for (uint32_t j0 = 0; j0 < n; j0 += RB) {
        const uint8_t *rp[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) rp[r] = p.rows + (size_t)ids[j0 + r < n ? j0 + r : j0] * p.pitch + loff;
        typename OP::acc_t acc[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) OP::init(acc[r]);
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < p.nch; c0 += FUSED_CG) {
            u4 rv[RB][FUSED_CG];
#pragma unroll
            for (int k = 0; k < FUSED_CG; k++) {
                const uint32_t off = (c0 + k) * 1024u;
                const bool in = c0 + k < p.nch && off + loff < p.pitch;
#pragma unroll
                for (int r = 0; r < RB; r++) { u4 v = {0u, 0u, 0u, 0u}; if (in) v = *(const u4 *)(rp[r] + off); rv[r][k] = v; }
            }
            FD_TICK(9);                                   // addresses + load issue
            if (tk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FD_TICK(10);                                  // waiting for the rows
#pragma unroll
            for (int k = 0; k < FUSED_CG; k++) {
                if (c0 + k < p.nch) {
                    const u4 q = *(const u4 *)(qv + (c0 + k) * 1024u + loff);
#pragma unroll
                    for (int r = 0; r < RB; r++) OP::add(acc[r], q, rv[r][k]);
                }
            }
            if (tk) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FD_TICK(11);                                  // arithmetic
        }
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const float d = OP::template finish<64>(acc[r]);
            if (j0 + r < n && lane == j0 + r) mine = d;
        }
        FD_TICK(12);                                      // reductions
    }
    return mine;
#undef FD_TICK
}

// check_element_closer (graph/mod.rs:315-339): is any d(q, ids[j]) <= thr?  Rows are evaluated FUSED_RB at a time in list
// order and the scan stops at the first batch that contains a hit, like the reference's early `return false`
// (the answer is the same; fewer rows are streamed for rejected candidates).  *n_eval += rows evaluated.
template <class OP, int LPR>
__device__ __forceinline__ bool f_any_le(const FusedParams &p, const uint8_t *qv, const uint32_t *ids, uint32_t n, uint32_t lane, float thr,
                                         unsigned long long &n_eval)
{
    constexpr uint32_t B = f_step_rows<LPR>();
    for (uint32_t j0 = 0; j0 < n; j0 += B) {
        const uint32_t nb = n - j0 < B ? n - j0 : B;
        const float d = f_dist_batch<OP, LPR>(p, qv, ids + j0, nb, lane);
        n_eval += nb;
        if (__ballot(lane < nb && d <= thr) != 0ull) return true;
    }
    return false;
}

// Algorithm 2 with entry points EP[0..n_ep); leaves the result set in the W heap (cx.CTL[1] = |W|).
// scan == false: search_layer (graph/mod.rs:161-255); scan == true: search_layer_disk without `discarded` (scan.rs:302-448).
#define F_TICK(k) do { if (tm) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); cx.tph[k] += (uint32_t)(t1_ - t0); t0 = t1_; } } while (0)
// ITER: search_layer_disk WITH the iterative scan's state (scan.rs:302-448): the visited set is the caller's and survives
// resumes (fresh == false keeps it; eps_visited == false: resume_scan_items' entry points are already in it), and every
// visited element that does not end in W goes to the `discarded` min-heap, in the reference's order of pushes.
template <class OP, int LPR, bool ITER = false>
__device__ void f_search_layer(const FusedParams &p, FusedCtx &cx, uint32_t n_ep, uint32_t ef, int layer, bool scan, bool fresh = true, bool eps_visited = true)
{
    const uint32_t lane = cx.lane;
    if (fresh) {   // fresh visited set
        for (uint64_t w = (uint64_t)lane * 4; w < p.vis_words; w += 256) *(u4 *)(cx.vis + w) = u4{VIS_EMPTY, VIS_EMPTY, VIS_EMPTY, VIS_EMPTY};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t vcount = fresh ? 0u : cx.vcount;
    if (eps_visited) {
        vcount += n_ep;
        for (uint32_t i = lane; i < n_ep; i += 64) (void)vis_test_and_set(cx.vis, (uint32_t)(p.vis_words >> 2) - 1u, cx.EP[i].y);
    }
    // Pushes into `discarded` are only read back by a later resume, so they are queued (in the reference's order) and applied
    // per expansion.  The new slots are consecutive, so at every level their ancestors form ONE contiguous index range: those
    // ranges (< 2c + depth entries for c queued pushes) are gathered into an LDS working set with one round of loads, the c
    // sift-ups run on the working set in queue order -- LDS latency instead of a memory hop each -- and the ranges are stored
    // back.  Same array as pushing one by one into the heap itself.
    uint32_t ndp = 0;
    // queue members [j0, j0 + c) -> heap slots [dlen, dlen + c); all of them lie on the same tree level (the caller splits a
    // queue that crosses a power of two), so "k levels up" is the same tree depth for every member and the level ranges are disjoint
    auto d_flush_range = [&](const uint32_t j0, const uint32_t c) {
        const uint32_t p0 = cx.dlen;
        const uint32_t depth = 31u - (uint32_t)__builtin_clz(p0 + c);       // levels 1..depth above the new slots (1-based heap indices)
        auto lo_of = [&](uint32_t k) { const uint32_t v = (p0 + 1u) >> k; return v ? v : 1u; };
        auto hi_of = [&](uint32_t k) { return (p0 + c) >> k; };
        if (lane >= 1u && lane <= depth) {                                 // per level: first index, offset of its range in the working set
            uint32_t base = 0;
            for (uint32_t k = 1; k < lane; k++) base += hi_of(k) - lo_of(k) + 1u;
            cx.LV[2u * lane] = lo_of(lane); cx.LV[2u * lane + 1u] = base;
        }
        uint32_t total = 0;
        for (uint32_t k = 1; k <= depth; k++) total += hi_of(k) - lo_of(k) + 1u;
        F_WSYNC();
        // gather: working-set slot f <-> (level, index)
        uint32_t my_idx[2] = {0u, 0u};
        for (int h = 0; h < 2; h++) {
            const uint32_t f = lane + 64u * (uint32_t)h;
            if (f < total) {
                uint32_t k = 1, base = 0;
                for (;; k++) { const uint32_t len = hi_of(k) - lo_of(k) + 1u; if (f < base + len) break; base += len; }
                my_idx[h] = lo_of(k) + (f - base);
                const uint2 v = cx.DS.ld(my_idx[h] - 1u);
                cx.WS[f].x = v.x; cx.WS[f].y = v.y;
            }
        }
        F_WSYNC();
        for (uint32_t j = 0; j < c; j++) {                                 // the c sift-ups, in queue order, on the working set
            const uint32_t pos1 = p0 + 1u + j;
            const uint2 it = make_uint2(cx.DP[j0 + j].x, cx.DP[j0 + j].y);
            const uint32_t dj = 31u - (uint32_t)__builtin_clz(pos1);
            const bool anc = lane >= 1u && lane <= dj;
            uint2 v = make_uint2(0u, 0u);
            if (anc) { const uint32_t s0 = cx.LV[2u * lane + 1u] + ((pos1 >> lane) - cx.LV[2u * lane]); v = make_uint2(cx.WS[s0].x, cx.WS[s0].y); }
            const unsigned long long sm = __ballot(anc && PHeap<true>::le(fh_d(it), fh_d(v)));
            const uint32_t t = sm ? (uint32_t)__builtin_ctzll(sm) : dj + 1u;
            F_WSYNC();
            if (anc && lane < t) {                                         // ancestor `lane` moves down to the path's slot one level below
                if (lane == 1u) { cx.DP[j0 + j].x = v.x; cx.DP[j0 + j].y = v.y; }
                else { const uint32_t s1 = cx.LV[2u * (lane - 1u) + 1u] + ((pos1 >> (lane - 1u)) - cx.LV[2u * (lane - 1u)]); cx.WS[s1].x = v.x; cx.WS[s1].y = v.y; }
            }
            if (lane == 0u && t > 1u) {                                    // the new element lands at level t-1 (t == 1: it stays in its own slot, DP[j])
                const uint32_t s1 = cx.LV[2u * (t - 1u) + 1u] + ((pos1 >> (t - 1u)) - cx.LV[2u * (t - 1u)]);
                cx.WS[s1].x = it.x; cx.WS[s1].y = it.y;
            }
            F_WSYNC();
        }
        // store back: the ranges, then the new slots
        for (int h = 0; h < 2; h++) {
            const uint32_t f = lane + 64u * (uint32_t)h;
            if (f < total) cx.DS.st(my_idx[h] - 1u, make_uint2(cx.WS[f].x, cx.WS[f].y));
        }
        if (lane < c) cx.DS.st(p0 + lane, make_uint2(cx.DP[j0 + lane].x, cx.DP[j0 + lane].y));
        cx.dlen = p0 + c;
        PHeap<true>::sync(cx.DS);
    };
    auto d_flush = [&]() {
        const uint32_t c = ndp;
        ndp = 0;
        if (c == 0 || cx.status != FS_OK) return;
        if (cx.dlen + c > p.disc_stride + p.disc_lds) { cx.status = FS_OVERFLOW; return; }
        uint32_t j0 = 0;
        while (j0 < c) {
            if (cx.dlen < 64u) {                                           // small heap: ancestors may be queue members themselves
                PHeap<true>::push(cx.DS, cx.dlen, make_uint2(cx.DP[j0].x, cx.DP[j0].y), lane); j0++; continue;
            }
            const uint32_t first1 = cx.dlen + 1u;                          // 1-based slot of the next member
            const uint32_t level_end = (2u << (31u - (uint32_t)__builtin_clz(first1))) - 1u;   // last slot of its tree level
            const uint32_t cs = (c - j0) < (level_end - first1 + 1u) ? (c - j0) : (level_end - first1 + 1u);
            d_flush_range(j0, cs);
            j0 += cs;
        }
    };
    auto d_push = [&](uint2 it) {
        if (lane == 0) { cx.DP[ndp].x = it.x; cx.DP[ndp].y = it.y; }
        ndp++;
        F_WSYNC();
        if (ndp == 64u) d_flush();
    };
    // heaps are driven by the whole wave (PHeap) while the candidate heap fits its LDS part; a heap that outgrows it
    // (rare) is handed to the serial hybrid LDS+spill code on lane 0.  clen/wl/rlen: |C|, |W|, result_len -- wave-uniform.
    uint32_t clen = 0, wl = 0, rlen = 0;
    lds_uint2 *const CA = cx.CH.lds, *const WA = cx.WH.lds;
    auto c_push = [&](uint2 it) {
        if (clen < cx.CH.L) PHeap<true>::push(CA, clen, it, lane);
        else { __syncthreads(); if (lane == 0) { uint32_t l = clen; FHeap<true>::push(cx.CH, l, it); } clen++; __syncthreads(); }
    };
    auto c_pop = [&]() -> uint2 {
        if (clen <= cx.CH.L) return PHeap<true>::pop(CA, clen, lane);
        __syncthreads();
        if (lane == 0) { uint32_t l = clen; const uint2 c = FHeap<true>::pop(cx.CH, l); cx.RES[0] = c; }
        clen--; __syncthreads();
        const uint2 c = cx.RES[0]; __syncthreads();
        return c;
    };
    for (uint32_t i = 0; i < n_ep; i++) {
        if (clen >= p.ccap) { cx.status = FS_OVERFLOW; break; }
        const uint2 it = cx.EP[i];
        c_push(it); PHeap<false>::push(WA, wl, it, lane);
    }
    rlen = wl;
    __syncthreads();
    for (;;) {
        if (cx.status != FS_OK) break;
        // pop the nearest candidate, decide whether to stop
        const bool tm = (p.fdbg & 4u) != 0; unsigned long long t0 = tm ? __builtin_amdgcn_s_memtime() : 0ull;
        // The candidate about to be popped is the heap's root: read it, decide, and put its neighbour list's loads in
        // flight BEFORE the pop's heap maintenance, which then hides that memory hop.
        uint32_t go = 0, cid = 0;
        if (clen > 0) {
            const uint2 c = PHeap<true>::ld(CA, 0u);
            const float cd = fh_d(c);
            bool stop;
            if (!scan) { const float f = wl ? __builtin_bit_cast(float, (unsigned int)WA[0].x) : 3.402823466e+38f; stop = cd > f; }                     // mod.rs:188-193
            else { const double f = wl ? (double)__builtin_bit_cast(float, (unsigned int)WA[0].x) : 1.7976931348623157e+308; stop = (double)cd > f; }    // scan.rs:339-346
            if (!stop) { go = 1; cid = c.y; }
        }
        go = __builtin_amdgcn_readfirstlane(go); cid = __builtin_amdgcn_readfirstlane(cid);
        const uint32_t *nb = p.l0_ids; uint32_t n = 0, lmax = 0, e_first = 0; int32_t clevel = 0x7fffffff;
        if (go) {
            if (layer == 0) { nb = p.l0_ids + (size_t)cid * 2u * p.m; lmax = 2u * p.m; }
            else { nb = p.up_ids + (size_t)(p.up_block[cid] + (uint32_t)(layer - 1)) * p.m; lmax = p.m; clevel = p.level[cid]; }
            e_first = lane < lmax ? nb[lane] : 0u;                                   // issued together with the count: one memory hop
            if (layer == 0) n = p.l0_cnt[cid]; else n = p.up_cnt[p.up_block[cid] + (uint32_t)(layer - 1)];
        }
        uint2 popped = make_uint2(0u, 0u); const bool had = clen > 0;
        if (had) popped = c_pop();                                                   // mod.rs:187 (the popped element is the root read above)
        F_TICK(0);
        if (!go) { if (ITER && had) d_push(popped); break; }                         // scan.rs:341-345 (flushed after the loop)
        if (tm) cx.tph[7]++;
        // a linked element at layer 0 always has level >= 0, so the check of mod.rs:198-200 needs no load there
        if (layer > 0 && clevel < layer) continue;
        F_TICK(1);
        for (uint32_t n0 = 0; n0 < n; n0 += 64) {                                    // lists longer than a wave (m > 32) go in order
            const uint32_t idx = n0 + lane;
            uint32_t e = 0; bool unvis = false;
            const uint32_t bmask = (uint32_t)(p.vis_words >> 2) - 1u;
            uint32_t *vslot = nullptr; uint32_t vold = VIS_EMPTY;                    // insert in flight (settled below)
            if (idx < n) {
                e = n0 == 0 ? e_first : nb[idx];
                unvis = !vis_lookup(cx.vis, bmask, e, vslot);                        // visited.contains / insert, mod.rs:206-209
                if (unvis) vold = atomicCAS(vslot, VIS_EMPTY, e);
                if (unvis && layer > 0 && p.level[e] < layer) unvis = false;         // mod.rs:213-216
            }
            F_TICK(2);
            const unsigned long long mask = __ballot(unvis);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            vcount += cnt;
            if (vcount * 4u > (uint32_t)p.vis_words * 3u) { cx.status = FS_OVERFLOW; break; }   // table too full: re-run in the lock-step path
            if (cnt == 0) { vis_settle(cx.vis, bmask, e, vslot, vold); continue; }
            if (unvis) cx.IDS[__popcll(mask & ((1ull << lane) - 1ull))] = e;
            __syncthreads();
            F_TICK(3);
            const float mine = f_dist_batch<OP, LPR>(p, cx.QV, cx.IDS, cnt, lane, tm ? cx.tph : nullptr);
            F_TICK(4);
            vis_settle(cx.vis, bmask, e, vslot, vold);                               // the CAS results came back with the rows
            if (lane < cnt) cx.RES[lane] = fh_pack(mine, cx.IDS[lane]);
            cx.nd0 += cnt;
            // Pre-filter in parallel: once W is full (result_len >= ef) its furthest distance f only shrinks while this
            // list is replayed, so a row with d >= f NOW can never be added later in the replay; lane 0 then visits only the
            // survivors, in list order, and re-tests each against the current f -- same pushes, same order, as mod.rs:226-243.
            bool keep = lane < cnt;
            if (keep && rlen >= ef && wl && !(p.fdbg & 1u)) {
                const float f0 = __builtin_bit_cast(float, (unsigned int)WA[0].x);
                keep = scan ? !((double)mine >= (double)f0) : (mine < f0);
            }
            unsigned long long km = ITER ? __ballot(lane < cnt) : __ballot(keep);     // ITER: rejected rows are visited too (they go to `discarded`)
            F_TICK(5);
            __syncthreads();
            while (km) {                                                             // replay in list order, mod.rs:226-243 / scan.rs:372-429
                const uint32_t j = (uint32_t)__builtin_ctzll(km); km &= km - 1ull;
                const uint2 it = cx.RES[j]; const float d = fh_d(it);
                const bool always_add = rlen < ef;
                const float wtop = wl ? __builtin_bit_cast(float, (unsigned int)WA[0].x) : 0.0f;
                bool add;
                if (!scan) { const float f = wl ? wtop : 3.402823466e+38f; add = d < f || always_add; }
                else { const double f = wl ? (double)wtop : 1.7976931348623157e+308; add = !(!always_add && (double)d >= f); }
                if (!add) { if (ITER) { d_push(it); if (cx.status != FS_OK) break; } continue; }   // scan.rs:385-404
                if (clen >= p.ccap) { cx.status = FS_OVERFLOW; break; }
                c_push(it); PHeap<false>::push(WA, wl, it, lane); rlen++;
                if (tm) cx.tph[8]++;
                if (clen > cx.cmax) cx.cmax = clen;
                if (rlen > ef) {
                    const uint2 ev = PHeap<false>::pop(WA, wl, lane); rlen--;
                    if (ITER) { d_push(ev); if (cx.status != FS_OK) break; }        // scan.rs:423-428
                }
            }
            if (ITER) d_flush();
            __syncthreads();
            F_TICK(6);
            cx.status = __shfl(cx.status, 0, 64);
            if (cx.status != FS_OK) break;
        }
    }
    if (ITER) {
        while (clen > 0 && cx.status == FS_OK) d_push(c_pop());                      // scan.rs:432-438: what is left of C
        d_flush();
        cx.vcount = vcount;
    }
    if (lane == 0) cx.CTL[1] = wl;
    __syncthreads();
}

// stable sort of the W heap's internal array into EP: ascending (build, mod.rs:248-254) or descending (scan.rs:441-446);
// rank sort: ties keep their order in W's array, exactly what a stable sort of that array does
__device__ void f_sort_results(FusedCtx &cx, uint32_t n, bool desc)
{
    for (uint32_t i = cx.lane; i < n; i += 64) {
        const uint2 me = cx.W[i]; const float d = fh_d(me);
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j++) {
            const float dj = fh_d(cx.W[j]);
            rank += (desc ? dj > d : dj < d) || (dj == d && j < i);
        }
        cx.EP[rank] = me;
    }
    __syncthreads();
}

template <class OP, int MODE, int LPR>   // MODE 0: query (get_scan_items), 1: insert (find_element_neighbors); LPR: lanes per row (64, or 8/32 for short rows)
__global__ void __launch_bounds__(64, (MODE == 2 ? FUSED_MINW_ITER : FUSED_MINW))
k_fused(const FusedParams p_in)
{
    FusedParams p = p_in;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    FusedCtx cx;
    const uint32_t lm0 = 2u * p.m;
    // LDS carve: C[ccap] | W[ef+2] | EP[ef+2] | RES[64] | RL[2m] | IDS[64] CTL[16] | QV[nch KiB].  The select phase runs
    // after the layer's search is over, so its scratch (the candidate under test EV and the discarded list DL) reuses C.
    cx.C = (uint2 *)lds;
    cx.W = cx.C + p.clds;
    cx.EP = cx.W + (p.ef + 2);
    cx.RES = cx.EP + (p.ef + 2);
    cx.RL = cx.RES + 64;
    cx.IDS = (uint32_t *)(cx.RL + lm0);
    cx.CTL = cx.IDS + 64;
    p.dsc = (float *)(cx.CTL + 16);
    cx.QV = (uint8_t *)(p.dsc + 64);                      // query parked in LDS (nch KiB)
    cx.DP = (lds_uint2 *)(cx.QV + p.nch * 1024u);          // MODE 2: queue of pending `discarded` pushes (64 entries), then the heap's LDS head
    cx.WS = cx.DP + 64; cx.LV = (uint32_t *)(cx.WS + 160);   // working set of a flush (<= 2*64 + depth entries), per-level {first index, offset}
    cx.DS.A = cx.WS + 160 + 32; cx.DS.L = MODE == 2 ? p.disc_lds : 0u;   // MODE 2: LDS head of the `discarded` heap
    cx.DS.G = MODE == 2 ? p.disc + (size_t)blockIdx.x * p.disc_stride : nullptr; cx.dlen = 0; cx.vcount = 0;
    cx.EV = (uint8_t *)cx.C;                              // host guarantees clds*8 >= nch*1024 + (ef+2)*8
    cx.DL = (uint2 *)(cx.EV + p.nch * 1024u);
    cx.CH.lds = (lds_uint2 *)cx.C; cx.CH.glob = p.spill + (size_t)blockIdx.x * p.spill_stride; cx.CH.L = p.clds;
    cx.WH.lds = (lds_uint2 *)cx.W; cx.WH.glob = nullptr; cx.WH.L = 0xffffffffu;
    cx.lane = threadIdx.x;
    cx.vis = p.vis + (size_t)blockIdx.x * p.vis_words;
    cx.nd0 = cx.nd1 = 0; cx.cmax = 0;
    for (int i = 0; i < 14; i++) cx.tph[i] = 0;
    const uint32_t lane = cx.lane;

    for (;;) {
        if (lane == 0) cx.CTL[5] = atomicAdd(p.next_task, 1u);
        __syncthreads();
        const uint32_t t = cx.CTL[5];
        __syncthreads();
        if (t >= p.ntasks) break;
        cx.status = FS_OK;
        const uint32_t qsel = p.t_qsel[t];
        const uint8_t *qsrc = (qsel & HX_QUERY_SLOT) ? p.queries + (size_t)(qsel & 0x7fffffffu) * p.pitch : p.rows + (size_t)qsel * p.pitch;
        f_park(p, qsrc, lane, cx.QV);
        const int new_level = MODE == 1 ? p.t_level[t] : -1;
        if (MODE == 1 && new_level >= FUSED_MAXL) { if (lane == 0) p.status[t] = FS_HOST; continue; }

        // d(q, entry point): mod.rs:371-377 / scan.rs:475
        if (lane == 0) cx.IDS[0] = p.entry;
        __syncthreads();
        const float d0 = f_dist_batch<OP, LPR>(p, cx.QV, cx.IDS, 1, lane);
        cx.nd0 += 1;
        if (lane == 0) cx.EP[0] = fh_pack(d0, p.entry);
        __syncthreads();
        uint32_t n_ep = 1;

        // greedy descent with ef = 1: mod.rs:385-399 (down to new_level+1) / scan.rs:491-512 (down to 1)
        const int stop_above = MODE == 1 ? new_level : 0;
        for (int lc = p.entry_level; lc > stop_above && cx.status == FS_OK; lc--) {
            f_search_layer<OP, LPR>(p, cx, n_ep, 1u, lc, MODE != 1);
            const uint32_t wl = cx.CTL[1];
            if (wl > 0) {
                f_sort_results(cx, wl, MODE != 1);
                if (MODE != 1) { const uint2 best = cx.EP[wl - 1]; __syncthreads(); if (lane == 0) cx.EP[0] = best; __syncthreads(); }
                n_ep = 1;                         // MODE 1: ep = vec![w[0]]; EP[0] already is the nearest
            } else if (MODE != 1) { n_ep = 0; break; }
        }

        if (MODE == 2) {
            // get_scan_items + the amgettuple loop of an iterative scan (scan.rs:458-577, 794-875) for one query
            uint32_t outc = 0; long long tuples = 0; double prev = -__builtin_inf();
            cx.dlen = 0; cx.vcount = 0;
            const size_t obase = (size_t)t * p.limit;
            if (cx.status == FS_OK && n_ep > 0) {
                f_search_layer<OP, LPR, true>(p, cx, n_ep, p.ef, 0, true, true, true);           // scan.rs:515-528
                bool single = false;
                for (;;) {
                    if (cx.status != FS_OK) break;
                    const uint32_t wl = single ? 1u : cx.CTL[1];
                    if (!single) f_sort_results(cx, wl, true);                                   // EP[0..wl): nearest LAST
                    // emit from the back (scan.rs:796-815, 860-874); the elements' TID masks are fetched 64 at a time
                    uint32_t left = wl;
                    while (left > 0 && outc < p.limit) {
                        const uint32_t c = left < 64u ? left : 64u, base = left - c;
                        if (lane < c) cx.IDS[lane] = p.emask[cx.EP[base + lane].y];
                        __syncthreads();
                        for (uint32_t i = c; i-- > 0 && outc < p.limit;) {
                            const uint32_t em = cx.IDS[i]; const uint32_t nt = em >> 12;
                            if (nt == 0) continue;                                               // scan.rs:866-868
                            tuples++;
                            const uint2 v = cx.EP[base + i]; const double dv = (double)fh_d(v);
                            for (int ti = (int)nt - 1; ti >= 0 && outc < p.limit; ti--) {        // heaptids.pop()
                                if (p.iter_mode == 2u) { if (dv < prev) continue; prev = dv; }   // strict_order, scan.rs:801-806
                                if (!((em >> ti) & 1u)) continue;                                // the executor's filter rejects this tuple
                                if (lane == 0) { p.out_ids[obase + outc] = v.y; p.out_d[obase + outc] = fh_d(v); p.out_tix[obase + outc] = (uint32_t)ti; }
                                outc++;
                            }
                        }
                        __syncthreads();
                        left = base;
                    }
                    if (outc >= p.limit) break;
                    if (tuples >= p.max_tuples) {                                                // scan.rs:831-841: drain `discarded` one by one
                        if (cx.dlen == 0) break;
                        const uint2 one = PHeap<true>::pop(cx.DS, cx.dlen, lane);
                        __syncthreads(); if (lane == 0) cx.EP[0] = one; __syncthreads();
                        single = true;
                        continue;
                    }
                    if (cx.dlen == 0) break;                                                     // resume_scan_items, scan.rs:548-550
                    single = false;
                    n_ep = 0;
                    while (n_ep < p.ef && cx.dlen > 0) {
                        const uint2 x = PHeap<true>::pop(cx.DS, cx.dlen, lane);
                        __syncthreads(); if (lane == 0) cx.EP[n_ep] = x; __syncthreads();
                        n_ep++;
                    }
                    f_search_layer<OP, LPR, true>(p, cx, n_ep, p.ef, 0, true, false, false);
                }
            }
            if (lane == 0) { p.out_cnt[t] = outc; p.status[t] = cx.status; }
        } else if (MODE == 0) {
            uint32_t cnt = 0;
            if (cx.status == FS_OK && n_ep > 0) {
                f_search_layer<OP, LPR>(p, cx, n_ep, p.ef, 0, true);                  // scan.rs:515-528
                const uint32_t wl = cx.CTL[1];
                f_sort_results(cx, wl, true);                                        // nearest LAST
                cnt = wl < p.k ? wl : p.k;
                for (uint32_t i = lane; i < cnt; i += 64) {                          // amgettuple pops from the back
                    const uint2 v = cx.EP[wl - 1 - i];
                    p.out_ids[(size_t)t * p.k + i] = v.y; p.out_d[(size_t)t * p.k + i] = fh_d(v);
                }
            }
            if (lane == 0) { p.out_cnt[t] = cnt; p.status[t] = cx.status; }
        } else {
            const int start = new_level < p.entry_level ? new_level : p.entry_level;
            const size_t obase = (size_t)t * FUSED_MAXL;
            for (uint32_t i = lane; i < FUSED_MAXL; i += 64) p.out_cnt[obase + i] = 0;
            for (int lc = start; lc >= 0 && cx.status == FS_OK; lc--) {
                const uint32_t lm = lc == 0 ? lm0 : p.m;
                f_search_layer<OP, LPR>(p, cx, n_ep, p.ef, lc, false);                // mod.rs:407-416
                if (cx.status != FS_OK) break;
                const uint32_t wl = cx.CTL[1];
                f_sort_results(cx, wl, false);                                       // W ascending; also the next layer's entry points (mod.rs:425)
                n_ep = wl;
                // select_neighbors(W, lm): mod.rs:269-308
                const unsigned long long ts0 = (p.fdbg & 4u) ? __builtin_amdgcn_s_memtime() : 0ull;
                uint32_t r = 0, nd = 0;
                if (wl <= lm) {
                    for (uint32_t i = lane; i < wl; i += 64) cx.RL[i] = cx.EP[i];
                    r = wl;
                } else {
                    // The candidate under test is parked in LDS; the NEXT candidate's row is fetched into the other of two
                    // buffers (the select scratch in C's LDS part, and the query's slot: d(e, q) is already known, so the query is
                    // not needed until the next layer's search and is parked again afterwards) while this one is compared.
                    // Its neighbour list comes along: d(e, r) for an accepted r that is already one of e's neighbours is stored in
                    // the mirror (the very bits a fresh evaluation gives: every term is symmetric in its operands), so a candidate
                    // that one of those rules out costs no row traffic at all.
                    uint8_t *const evb[2] = {cx.EV, cx.QV};
                    uint32_t nx_id = 0xFFFFFFFFu; float nx_d = 0.0f;                  // lane's slot of the NEXT candidate's list (id, stored distance)
                    auto list_prefetch = [&](uint32_t el) {
                        const uint32_t *li; const float *ld; uint32_t lc_n;
                        if (lc == 0) { li = p.l0_ids + (size_t)el * lm0; ld = p.l0_d + (size_t)el * lm0; lc_n = p.l0_cnt[el]; }
                        else { const uint32_t blk = p.up_block[el] + (uint32_t)(lc - 1); li = p.up_ids + (size_t)blk * p.m; ld = p.up_d + (size_t)blk * p.m; lc_n = p.up_cnt[blk]; }
                        nx_id = 0xFFFFFFFFu; nx_d = 0.0f;
                        if (lane < lc_n) { nx_id = li[lane]; nx_d = ld[lane]; }
                    };
                    for (uint32_t i = 0; i < wl; i++) {
                        if (r >= lm) break;                                          // mod.rs:285-287
                        const uint2 e = cx.EP[i];
                        bool closer = true;                                          // check_element_closer, mod.rs:315-339
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // e's own row and list (requested one iteration ago) have landed
                        __syncthreads();
                        const uint32_t my_id = nx_id; const float my_d = nx_d;       // e's list slot of this lane
                        if (i + 1u < wl) { const uint32_t en = cx.EP[i + 1u].y; f_park_async(p, p.rows + (size_t)en * p.pitch, lane, evb[(i + 1u) & 1u]); list_prefetch(en); }
                        if (r > 0 && i > 0) {
                            bool known_hit = false;
                            if (my_id != 0xFFFFFFFFu && my_d <= fh_d(e)) for (uint32_t j = 0; j < r; j++) known_hit |= cx.RL[j].y == my_id;
                            if (__ballot(known_hit) != 0ull) closer = false;         // mod.rs:333-335 with a distance we already hold
                        }
                        if (r > 0 && closer) {
                            if (lane < r) cx.IDS[lane] = cx.RL[lane].y;
                            __syncthreads();
                            closer = !f_any_le<OP, LPR>(p, evb[i & 1u], cx.IDS, r, lane, fh_d(e), cx.nd1);   // mod.rs:324-336
                            __syncthreads();
                        }
                        if (lane == 0) { if (closer) cx.RL[r] = e; else cx.DL[nd] = e; }
                        if (closer) r++; else nd++;
                        __syncthreads();
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // no row may still be in flight towards the query's slot
                    __syncthreads();
                    f_park(p, qsrc, lane, cx.QV);                                    // the query again, for the next layer's search
                    if (lane == 0) for (uint32_t j = 0; j < nd && r < lm; j++) cx.RL[r++] = cx.DL[j];   // mod.rs:300-305
                    r = __shfl(r, 0, 64);
                }
                __syncthreads();
                const size_t lb = ((size_t)t * FUSED_MAXL + (size_t)lc) * lm0;
                for (uint32_t i = lane; i < r; i += 64) { const uint2 v = cx.RL[i]; p.out_ids[lb + i] = v.y; p.out_d[lb + i] = fh_d(v); }
                if (lane == 0) p.out_cnt[obase + lc] = r;
                __syncthreads();
                if (p.fdbg & 4u) cx.tph[13] += (uint32_t)(__builtin_amdgcn_s_memtime() - ts0);
            }
            if (lane == 0) p.status[t] = cx.status;
        }
        __syncthreads();
    }
    if (lane == 0) { atomicAdd(&p.n_dist[0], cx.nd0); atomicAdd(&p.n_dist[1], cx.nd1); atomicMax(&p.n_dist[2], (unsigned long long)cx.cmax);
                     if (p.fdbg & 4u) { for (int i = 0; i < 14; i++) atomicAdd(&p.n_dist[3 + i], (unsigned long long)cx.tph[i]); } }
}

// ---- device graph mirror maintenance ------------------------------------------------------------------------
__global__ void k_mirror_levels(int32_t *level, uint32_t *up_block, uint32_t first, uint32_t n, const int32_t *lv, const uint32_t *blk)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { level[first + i] = lv[i]; up_block[first + i] = blk[i]; }
}
// one 64-thread block per record: {elem, layer, cnt, block} + ids[stride] + dists[stride]
__global__ void k_mirror_lists(uint32_t *l0_ids, float *l0_d, uint16_t *l0_cnt, uint32_t *up_ids, float *up_d, uint16_t *up_cnt, uint32_t m,
                               uint32_t n_rec, const uint32_t *hdr, const uint32_t *ids, const float *dd, uint32_t stride, uint8_t *pm_valid)
{
    const uint32_t r = blockIdx.x;
    if (r >= n_rec) return;
    const uint32_t elem = hdr[4 * r], layer = hdr[4 * r + 1], cnt = hdr[4 * r + 2], blk = hdr[4 * r + 3];
    const uint32_t *src = ids + (size_t)r * stride; const float *sd = dd + (size_t)r * stride;
    if (layer == 0) {
        for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) { l0_ids[(size_t)elem * 2u * m + k] = src[k]; l0_d[(size_t)elem * 2u * m + k] = sd[k]; }
        if (threadIdx.x == 0) { l0_cnt[elem] = (uint16_t)cnt; if (pm_valid) pm_valid[elem] = 0; }   // a list written by the host: cached pair matrix is stale
    } else {
        for (uint32_t k = threadIdx.x; k < cnt; k += blockDim.x) { up_ids[(size_t)blk * m + k] = src[k]; up_d[(size_t)blk * m + k] = sd[k]; }
        if (threadIdx.x == 0) up_cnt[blk] = (uint16_t)cnt;
    }
}

template <class T> static int mirror_grow(hx_engine *e, T *&p, size_t old_n, size_t new_n)
{
    T *q = nullptr;
    HX_HIP(e, hipMalloc((void **)&q, new_n * sizeof(T)));
    HX_HIP(e, hipMemsetAsync(q, 0, new_n * sizeof(T), e->stream));
    if (p && old_n) HX_HIP(e, hipMemcpyAsync(q, p, old_n * sizeof(T), hipMemcpyDeviceToDevice, e->stream));
    HX_HIP(e, hipStreamSynchronize(e->stream));
    if (p) (void)hipFree(p);
    p = q;
    return HX_OK;
}

int hx_engine::mirror_reserve(uint32_t m, uint64_t n_elems, uint64_t n_blocks)
{
    HxMirror &mr = mirror;
    int rc;
    if (mr.m == 0) mr.m = m;
    if (mr.m != m) return fail(HX_E_STATE, "mirror m mismatch");
    if (n_elems > mr.cap) {
        const uint64_t nc = std::max<uint64_t>(std::max<uint64_t>(n_elems, capacity), mr.cap * 2);
        if ((rc = mirror_grow(this, mr.d_l0_ids, mr.cap * 2 * m, nc * 2 * m))) return rc;
        if ((rc = mirror_grow(this, mr.d_l0_d, mr.cap * 2 * m, nc * 2 * m))) return rc;
        if ((rc = mirror_grow(this, mr.d_l0_cnt, mr.cap, nc))) return rc;
        if ((rc = mirror_grow(this, mr.d_level, mr.cap, nc))) return rc;
        if ((rc = mirror_grow(this, mr.d_up_block, mr.cap, nc))) return rc;
        mr.cap = nc;
    }
    if (n_blocks > mr.cap_blocks) {
        const uint64_t nb = std::max<uint64_t>(std::max<uint64_t>(n_blocks, capacity / 8 + 1024), mr.cap_blocks * 2);
        if ((rc = mirror_grow(this, mr.d_up_ids, mr.cap_blocks * m, nb * m))) return rc;
        if ((rc = mirror_grow(this, mr.d_up_d, mr.cap_blocks * m, nb * m))) return rc;
        if ((rc = mirror_grow(this, mr.d_up_cnt, mr.cap_blocks, nb))) return rc;
        mr.cap_blocks = nb;
    }
    return HX_OK;
}

// levels/up_block of elements [first, first+n_new) and n_rec list records (hdr: elem, layer, cnt, block; ids, dists: stride 2m)
int hx_engine::mirror_update(uint32_t first, uint32_t n_new, const int32_t *levels, const uint32_t *blocks,
                             uint32_t n_rec, const uint32_t *hdr, const uint32_t *ids, const float *dists)
{
    HxMirror &mr = mirror;
    const uint32_t stride = 2 * mr.m;
    const size_t bytes = (size_t)n_new * 8 + (size_t)n_rec * (16 + (size_t)stride * 8);
    if (bytes == 0) return HX_OK;
    if (bytes > mr.cap_stage) {
        if (mr.h_stage) (void)hipHostFree(mr.h_stage);
        if (mr.d_stage) (void)hipFree(mr.d_stage);
        mr.h_stage = mr.d_stage = nullptr; mr.cap_stage = 0;
        const size_t n = bytes * 2 + 4096;
        HX_HIP(this, hipHostMalloc((void **)&mr.h_stage, n, hipHostMallocDefault));
        HX_HIP(this, hipMalloc((void **)&mr.d_stage, n));
        mr.cap_stage = n;
    }
    uint8_t *h = mr.h_stage; size_t o = 0;
    const size_t o_lv = o; memcpy(h + o, levels, (size_t)n_new * 4); o += (size_t)n_new * 4;
    const size_t o_bk = o; memcpy(h + o, blocks, (size_t)n_new * 4); o += (size_t)n_new * 4;
    const size_t o_hdr = o; memcpy(h + o, hdr, (size_t)n_rec * 16); o += (size_t)n_rec * 16;
    const size_t o_ids = o; memcpy(h + o, ids, (size_t)n_rec * stride * 4); o += (size_t)n_rec * stride * 4;
    const size_t o_dd = o; memcpy(h + o, dists, (size_t)n_rec * stride * 4); o += (size_t)n_rec * stride * 4;
    HX_HIP(this, hipMemcpyAsync(mr.d_stage, h, o, hipMemcpyHostToDevice, stream));
    if (n_new) hipLaunchKernelGGL(k_mirror_levels, dim3((n_new + 255) / 256), dim3(256), 0, stream, mr.d_level, mr.d_up_block, first, n_new,
                                  (const int32_t *)(mr.d_stage + o_lv), (const uint32_t *)(mr.d_stage + o_bk));
    if (n_rec) hipLaunchKernelGGL(k_mirror_lists, dim3(n_rec), dim3(64), 0, stream, mr.d_l0_ids, mr.d_l0_d, mr.d_l0_cnt, mr.d_up_ids, mr.d_up_d, mr.d_up_cnt, mr.m, n_rec,
                                  (const uint32_t *)(mr.d_stage + o_hdr), (const uint32_t *)(mr.d_stage + o_ids), (const float *)(mr.d_stage + o_dd), stride, mr.d_pm_valid);
    HX_HIP(this, hipGetLastError());
    HX_HIP(this, hipStreamSynchronize(stream));
    return HX_OK;
}

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
    uint32_t dbg;   // timing experiments only (HX_LK_DBG): 1 skip pair math, 2 skip row loads, 4 skip select
};

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
        // ---- pair matrix of the n candidate rows.  Rows 1..n-1 of the triangle are paired (r, n-r) -- r + (n-r) = n <= 33
        // pairs per row-pair -- and wave w owns row-pairs q = w and q = 15 - w (r = q + 1).  For a row-pair, accumulator
        // A[j] is pair (n-r, j), j < n-r, and A[32-j] is pair (r, j), j < r (disjoint because n <= 33).  A wave reads
        // its 4 "a" fragments once per chunk and each b_j fragment once for up to 4 pairs: 36 LDS reads per chunk
        // instead of 132, and 4 independent accumulation chains per read. ----
        const uint32_t P = n * (n - 1) / 2;
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
    if (threadIdx.x == 0) { *gl_cnt = (uint16_t)cnt; p.out_cnt[g] = cnt; atomicAdd(p.n_pairs, pairs); }
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

// One back-link op (new_id at distance new_d) on the list held in LDS: update_neighbor_connections' body, mod.rs:458-487.
// Runs in ONE wave (wave-level ordering only), on the list state (M, lid, ld, cnt, v) and the scratch arrays it is given.
// SPEC == false: applies the op; returns true when the list was pruned (it is then in select order, its matrix complete).
// SPEC == true: touches only the scratch arrays and answers "would this op change the list?" -- false only when it is certain
// that the new row is the one left out and every survivor keeps its slot (then list, distances and matrix stay as they are).
template <class OP, int LPR, bool SPEC>
__device__ bool lc_op(const FusedParams &fp, const LinksParams &p, float *M, float *M2, uint32_t *lid, float *ld, uint32_t *lid2, float *ld2,
                      float *nd, uint32_t *pos, float *sd, uint32_t *sel, uint32_t *dis, uint32_t *ORD, uint32_t *IDS, uint8_t *QV,
                      uint32_t &cnt, uint32_t &v, const uint32_t lm, const uint32_t new_id, const float new_d, const uint32_t lane, unsigned long long &ndist, unsigned long long *tk = nullptr)
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
        {   // rank sort: every lane compares its distance with lane j's, read through the scalar unit (no LDS traffic)
            const float d = lane < cnt ? ld[lane] : new_d; uint32_t rank = 0;
            const unsigned int dbits = __builtin_bit_cast(unsigned int, d);
            for (uint32_t j = 0; j < n; j++) {
                const float dj = __builtin_bit_cast(float, (unsigned int)__builtin_amdgcn_readlane((int)dbits, (int)j));
                rank += (dj < d) || (dj == d && j < lane);
            }
            if (lane < n) { pos[rank] = lane < cnt ? lane : LC_SLOTS; sd[rank] = d; }
        }
        F_WSYNC();
        LC_TICK(1);                                                                // sort
        // select_neighbors(candidates, lm): mod.rs:284-305.  D(k1,k2) = cached pair or the new row's distance
        uint32_t r = 0, ndc = 0, n_done = 0;      // n_done: how many entries of the evaluation order ORD have their nd[]
        bool ordered = false; unsigned long long amask = 0ull;   // slots accepted so far
        uint32_t my_slot = 0;                      // lane j: slot of the j-th accepted candidate
        auto finish_nd = [&]() {                  // all remaining d(new, slot)
            if (!ordered) {
                if (lane < cnt) { IDS[lane] = lid[lane]; ORD[lane] = lane; }
                f_park_w(fp, p.rows + (size_t)new_id * p.pitch, lane, QV);
                ordered = true;
            }
            if (n_done < cnt) {
                const float d = f_dist_batch<OP, LPR, LC_RB>(fp, QV, IDS + n_done, cnt - n_done, lane);
                if (lane < cnt - n_done) nd[ORD[n_done + lane]] = d;
                ndist += cnt - n_done; n_done = cnt;
            }
            F_WSYNC();
        };
        for (uint32_t i = 0; i < n; i++) {
            if (r >= lm) break;
            const float ed = sd[i]; const uint32_t si = pos[i];
            bool closer;
            if (si == LC_SLOTS) {
                LC_TICK(2);                                                        // walk so far
                // accepted slots first in the evaluation order
                const bool acc = lane < cnt && ((amask >> lane) & 1ull) != 0ull;
                const unsigned long long am = amask, below = (1ull << lane) - 1ull;
                const uint32_t na = (uint32_t)__popcll(am);
                if (lane < cnt) {
                    const uint32_t o = acc ? (uint32_t)__popcll(am & below) : na + (uint32_t)__popcll(~am & below);
                    IDS[o] = lid[lane]; ORD[o] = lane;
                }
                f_park_w(fp, p.rows + (size_t)new_id * p.pitch, lane, QV);
                ordered = true;
                bool hit = false;
                constexpr uint32_t B = f_step_rows<LPR>();
                for (uint32_t j0 = 0; j0 < na && !hit; j0 += B) {
                    const uint32_t nb = na - j0 < B ? na - j0 : B;
                    const float d = f_dist_batch<OP, LPR>(fp, QV, IDS + j0, nb, lane);
                    if (lane < nb) nd[ORD[j0 + lane]] = d;
                    ndist += nb; n_done = j0 + nb;
                    hit = __ballot(lane < nb && d <= ed) != 0ull;                  // mod.rs:333-335
                }
                closer = !hit;
                LC_TICK(3);                                                        // lazy distances of the new row
                if (SPEC && closer) return true;                                   // the new row enters the list
                if (closer) finish_nd();                                           // later candidates are compared with the new row
            } else {
                bool hit = false;
                if (lane < r) {
                    const uint32_t sj = my_slot;                                   // slot of the lane-th accepted candidate (== pos[sel[lane]])
                    const float dij = sj == LC_SLOTS ? nd[si] : M[lc_tri(si, sj)];
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
            bool mine = lane < r && pos[sel[lane]] == LC_SLOTS;
            if (__ballot(mine) != 0ull) finish_nd();
        }
        // surviving list and its pair matrix
        if (lane < r) { const uint32_t sa = pos[sel[lane]]; lid2[lane] = sa == LC_SLOTS ? new_id : lid[sa]; ld2[lane] = sd[sel[lane]]; }
        for (uint32_t idx = lane; idx < r * (r - 1) / 2; idx += 64) {
            uint32_t a, b; tri_decode(idx, a, b);
            const uint32_t sa = pos[sel[a]], sb = pos[sel[b]];
            M2[idx] = sa == LC_SLOTS ? nd[sb] : (sb == LC_SLOTS ? nd[sa] : M[lc_tri(sa, sb)]);
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

template <class OP, int LPR>
__global__ void __launch_bounds__(64, 4)
k_links_cached(const LinksParams p, float *pm, uint8_t *pm_valid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float *M = (float *)lds;                       // pair matrix over list slots, packed lower triangle [528]
    float *M2 = M + 528;
    uint32_t *lid = (uint32_t *)(M2 + 528);        // list ids / distances to the target [40]
    float *ld = (float *)(lid + 40);
    uint32_t *lid2 = (uint32_t *)(ld + 40); float *ld2 = (float *)(lid2 + 40);
    float *nd = ld2 + 40;                          // d(new row, slot j)
    uint32_t *pos = (uint32_t *)(nd + 40);         // sorted candidate k -> slot (LC_SLOTS = the new row)
    float *sd = (float *)(pos + 40);
    uint32_t *sel = (uint32_t *)(sd + 40), *dis = sel + 40, *ORD = dis + 40, *IDS = ORD + 40;   // IDS[64]
    float *DSC = (float *)(IDS + 64);
    uint8_t *QV = (uint8_t *)(DSC + 64);
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x >= p.n_groups) return;
    const uint32_t g = p.gmap ? p.gmap[blockIdx.x] : blockIdx.x;
    FusedParams fp; fp.rows = p.rows; fp.pitch = p.pitch; fp.nch = (p.pitch + 1023u) / 1024u; fp.dsc = DSC;
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
    const bool tm = (p.dbg & 8u) != 0; const unsigned long long tk0 = tm ? __builtin_amdgcn_s_memtime() : 0ull;
    for (uint32_t op = p.op_off[g]; op < p.op_off[g + 1]; op++)
        (void)lc_op<OP, LPR, false>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op], p.op_d[op], lane, ndist, tm ? tk : nullptr);
    if (lane < cnt) {
        gl_ids[lane] = lid[lane]; gl_d[lane] = ld[lane];
        if (p.out_ids) { p.out_ids[(size_t)g * 2u * p.m + lane] = lid[lane]; p.out_d[(size_t)g * 2u * p.m + lane] = ld[lane]; }
    }
    if (cached) {
        float *dst = pm + (size_t)target * LC_TRI;
        for (uint32_t i = lane; i < (v > 1 ? v * (v - 1) / 2 : 0u); i += 64) dst[i] = M[i];
        if (lane == 0) pm_valid[target] = (uint8_t)v;
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
template <class OP, int LPR>
__global__ void __launch_bounds__(64 * HUB_W, 1)
k_links_hub(const LinksParams p, float *pm, uint8_t *pm_valid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t nchb = ((p.pitch + 1023u) / 1024u) * 1024u;
    // shared: M[528] M2[528] lid[40] ld[40] lid2[40] ld2[40] ctl[16]; per wave: nd pos sd sel dis ORD [40 each] IDS[64] DSC[64] QV[nchb]
    float *M = (float *)lds, *M2 = M + 528;
    uint32_t *lid = (uint32_t *)(M2 + 528); float *ld = (float *)(lid + 40);
    uint32_t *lid2 = (uint32_t *)(ld + 40); float *ld2 = (float *)(lid2 + 40);
    uint32_t *ctl = (uint32_t *)(ld2 + 40);
    uint8_t *wbase = (uint8_t *)(ctl + 16) + (size_t)wave * ((40 * 6 + 64 + 64) * 4 + nchb);
    float *nd = (float *)wbase; uint32_t *pos = (uint32_t *)(nd + 40); float *sd = (float *)(pos + 40);
    uint32_t *sel = (uint32_t *)(sd + 40), *dis = sel + 40, *ORD = dis + 40, *IDS = ORD + 40;
    float *DSC = (float *)(IDS + 64);
    uint8_t *QV = (uint8_t *)(DSC + 64);
    if (blockIdx.x >= p.n_groups) return;
    const uint32_t g = p.gmap ? p.gmap[blockIdx.x] : blockIdx.x;
    FusedParams fp; fp.rows = p.rows; fp.pitch = p.pitch; fp.nch = (p.pitch + 1023u) / 1024u; fp.dsc = DSC;
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
                const bool pruned = lc_op<OP, LPR, false>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op], p.op_d[op], lane, ndist);
                if (lane == 0) { ctl[0] = op + 1u; ctl[1] = cnt; ctl[2] = v; if (pruned) ctl[3] = 1u; }
            }
            continue;
        }
        const uint32_t nv = op_end - op < HUB_W ? op_end - op : HUB_W;   // ops evaluated this round
        bool changed = false;
        if (wave < nv) {
            uint32_t c2 = cnt, v2 = v;
            changed = lc_op<OP, LPR, true>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, c2, v2, lm, p.op_new[op + wave], p.op_d[op + wave], lane, ndist);
        }
        if (lane == 0) ctl[4 + wave] = changed ? 1u : 0u;
        __syncthreads();
        uint32_t first = nv;
        for (uint32_t w = 0; w < nv; w++) if (ctl[4 + w]) { first = w; break; }
        __syncthreads();
        if (first == nv) { if (threadIdx.x == 0) ctl[0] = op + nv; continue; }
        if (wave == 0) {
            (void)lc_op<OP, LPR, false>(fp, p, M, M2, lid, ld, lid2, ld2, nd, pos, sd, sel, dis, ORD, IDS, QV, cnt, v, lm, p.op_new[op + first], p.op_d[op + first], lane, ndist);
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
    if (threadIdx.x == 0) { *gl_cnt = (uint16_t)cnt; if (p.out_cnt) p.out_cnt[g] = cnt; }
    if (lane == 0) atomicAdd(p.n_pairs, ndist);
}

template <class OP, int LPR>
static hipError_t launch_links_cached_lpr(hx_engine *e, const LinksParams &p)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (528 * 2 + 40 * 11 + 64 + 64) * 4 + nch * 1024;
    hipLaunchKernelGGL((k_links_cached<OP, LPR>), dim3(p.n_groups), dim3(64), lds, e->stream, p, e->mirror.d_pm, e->mirror.d_pm_valid);
    return hipGetLastError();
}
template <class OP, int LPR>
static hipError_t launch_links_hub_lpr(hx_engine *e, const LinksParams &p)
{
    const size_t nch = (e->pitch + 1023) / 1024;
    const size_t lds = (528 * 2 + 40 * 4 + 16) * 4 + (size_t)HUB_W * ((40 * 6 + 64 + 64) * 4 + nch * 1024);
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t st = hipFuncSetAttribute((const void *)k_links_hub<OP, LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (st != hipSuccess) return st;
        attr_set = true;
    }
    hipLaunchKernelGGL((k_links_hub<OP, LPR>), dim3(p.n_groups), dim3(64 * HUB_W), lds, e->stream, p, e->mirror.d_pm, e->mirror.d_pm_valid);
    return hipGetLastError();
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
    p.gmap = nullptr;
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
    if (p.dbg & 8u) { unsigned long long t[7]; memcpy(t, h + o_ctr, 56); fprintf(stderr, "[hx] k_links_cached groups %u ops %u: ticks matrix-fill %llu sort %llu walk %llu lazy-nd %llu rebuild %llu; whole kernel per wave %llu\n", n_groups, n_ops, t[1], t[2], t[3], t[4], t[5], t[6]); }
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    return HX_OK;
}

int hx_engine::links_run_grouped(uint32_t n_ops, const unsigned long long *keys, const uint32_t *op_new, const float *op_d, uint64_t *n_pairs, uint32_t stats[2])
{
    HxMirror &mr = mirror;
    stats[0] = stats[1] = 0;
    if (n_pairs) *n_pairs = 0;
    if (n_ops == 0) return HX_OK;
    if (2 * mr.m != LC_SLOTS || pitch > FUSED_MAXCH * 1024u) return fail(HX_E_STATE, "device-side op grouping serves m = 16 and rows <= 8 KiB");
    HX_HIP(this, hipSetDevice(device));
    if (mr.cap_pm < mr.cap) {                                   // pair-matrix cache for every layer-0 list (as in links_run)
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
    uint32_t c[4];
    int rc = hx_group_ops(this, n_ops, keys, op_new, op_d, hub_min, grp, c);
    if (rc) return rc;
    const uint32_t n_groups = c[0], n_hub = c[1], n_norm = c[2];
    stats[0] = n_groups; stats[1] = c[3];
    LinksParams p;
    p.rows = d_rows; p.pitch = (uint32_t)pitch; p.m = mr.m;
    p.l0_ids = mr.d_l0_ids; p.l0_d = mr.d_l0_d; p.l0_cnt = mr.d_l0_cnt; p.up_block = mr.d_up_block; p.up_ids = mr.d_up_ids; p.up_d = mr.d_up_d; p.up_cnt = mr.d_up_cnt;
    p.n_groups = n_groups; p.target = grp.tg; p.layer = grp.ly; p.op_off = grp.off; p.op_new = grp.op_new; p.op_d = grp.op_d; p.gmap = nullptr;
    p.out_ids = nullptr; p.out_d = nullptr; p.out_cnt = nullptr;
    p.n_pairs = (unsigned long long *)mr.d_lk;
    { const char *dv = getenv("HX_LK_DBG"); p.dbg = dv ? (uint32_t)atoi(dv) : 0u; }
    HX_HIP(this, hipMemsetAsync(mr.d_lk, 0, 64, stream));
    if (timing) HX_HIP(this, hipEventRecord(ev2, stream));
    hipError_t ls = hipSuccess;
    if (n_hub) {
        LinksParams ph = p; ph.n_groups = n_hub; ph.gmap = grp.gmap_hub;
#define F32C(K) ls = launch_links_hub<OpF32<K>>(this, ph)
#define F16C(K) ls = launch_links_hub<OpF16<K>>(this, ph)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links_hub<OpHamming>(this, ph), ls = launch_links_hub<OpJaccard>(this, ph));
#undef F32C
#undef F16C
        HX_HIP(this, ls);
    }
    if (n_norm) {
        p.n_groups = n_norm; p.gmap = grp.gmap_norm;
#define F32C(K) ls = launch_links_cached<OpF32<K>>(this, p)
#define F16C(K) ls = launch_links_cached<OpF16<K>>(this, p)
        HX_DISPATCH(this, F32C, F16C, ls = launch_links_cached<OpHamming>(this, p), ls = launch_links_cached<OpJaccard>(this, p));
#undef F32C
#undef F16C
        HX_HIP(this, ls);
    }
    if (timing) HX_HIP(this, hipEventRecord(ev3, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_lk, mr.d_lk, 64, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    unsigned long long np; memcpy(&np, mr.h_lk, 8);
    if (n_pairs) *n_pairs = np;
    if (timing) { float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev2, ev3)); last_ms = ms; stat_links.launches++; stat_links.units += np; stat_links.ms += ms; }
    return HX_OK;
}

int hx_engine::mirror_download(uint64_t n_elems, uint64_t n_blocks, uint32_t *l0_ids, float *l0_d, uint16_t *l0_cnt, uint32_t *up_ids, float *up_d, uint16_t *up_cnt)
{
    HxMirror &mr = mirror;
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipStreamSynchronize(stream));
    const size_t lm0 = 2u * (size_t)mr.m;
    if (n_elems) {
        HX_HIP(this, hipMemcpy(l0_ids, mr.d_l0_ids, n_elems * lm0 * 4, hipMemcpyDeviceToHost));
        HX_HIP(this, hipMemcpy(l0_d, mr.d_l0_d, n_elems * lm0 * 4, hipMemcpyDeviceToHost));
        HX_HIP(this, hipMemcpy(l0_cnt, mr.d_l0_cnt, n_elems * 2, hipMemcpyDeviceToHost));
    }
    if (n_blocks) {
        HX_HIP(this, hipMemcpy(up_ids, mr.d_up_ids, n_blocks * mr.m * 4, hipMemcpyDeviceToHost));
        HX_HIP(this, hipMemcpy(up_d, mr.d_up_d, n_blocks * mr.m * 4, hipMemcpyDeviceToHost));
        HX_HIP(this, hipMemcpy(up_cnt, mr.d_up_cnt, n_blocks * 2, hipMemcpyDeviceToHost));
    }
    return HX_OK;
}

template <class OP, int MODE, int LPR>
static hipError_t launch_fused(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds)
{
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t s = hipFuncSetAttribute((const void *)k_fused<OP, MODE, LPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (s != hipSuccess) return s;
        attr_set = true;
    }
    if (getenv("HX_DEBUG")) {
        static thread_local bool once = false;
        if (!once) { once = true; int nb = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_fused<OP, MODE, LPR>, 64, lds);
            fprintf(stderr, "[hx] k_fused<mode %d, %d lanes/row>: dynamic LDS %zu B, grid %u, occupancy API says %d blocks/CU\n", MODE, LPR, lds, grid, nb); }
    }
    hipLaunchKernelGGL((k_fused<OP, MODE, LPR>), dim3(grid), dim3(64), lds, e->stream, p);
    return hipGetLastError();
}

template <class OP, int LPR>
static hipError_t launch_fused_lpr(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    if (mode == 2) return launch_fused<OP, 2, LPR>(e, p, grid, lds);
    return mode == 0 ? launch_fused<OP, 0, LPR>(e, p, grid, lds) : launch_fused<OP, 1, LPR>(e, p, grid, lds);
}
template <class OP>
static hipError_t launch_fused_mode(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{   // lanes per row by payload, as in launch_links_cached
    if (e->pitch <= 128) return launch_fused_lpr<OP, 8>(e, p, grid, lds, mode);
    if (e->pitch <= 512) return launch_fused_lpr<OP, 32>(e, p, grid, lds, mode);
    return launch_fused_lpr<OP, 64>(e, p, grid, lds, mode);
}

// mode 0: ntasks queries -> out_ids/out_d [ntasks][k], out_cnt[ntasks]; mode 1: ntasks inserts -> out_ids/out_d
// [ntasks][FUSED_MAXL][2m], out_cnt [ntasks][FUSED_MAXL].  status[ntasks].  All host pointers.
int hx_engine::fused_run(int mode, uint32_t ntasks, const uint32_t *q_sel, const int32_t *t_level, uint32_t ef, uint32_t k,
                         uint32_t entry, int entry_level, uint32_t *out_ids, float *out_d, uint32_t *out_cnt, uint32_t *status,
                         uint64_t counts[2], const HxFusedIter *it, HxFusedView *view, uint32_t roomy)
{
    HxMirror &mr = mirror;
    if (ntasks == 0) return HX_OK;
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_ARG, "row too wide for the fused kernel");
    if (mode == 1 && 2 * mr.m > 64) return fail(HX_E_ARG, "m > 32 is served by the lock-step path");
    if (mode == 2 && (!it || !it->emask || !it->out_tix)) return fail(HX_E_ARG, "iterative scan arguments missing");
    HX_HIP(this, hipSetDevice(device));
    // LDS: C[ccap] W[ef+2] EP[ef+2] RES[64] RL[2m] DL[ef+2] (8 B each) + IDS[64] + CTL[16] (4 B each)
    // candidate heap: up to FUSED_CCAP entries, the first `clds` in LDS and the tail in a per-workgroup spill area
    // (the largest heap seen on 1M x 768 builds was 1552 entries at ef = 200); beyond FUSED_CCAP a task reports
    // FS_OVERFLOW and is re-run by the lock-step path
    const size_t nch_ = (pitch + 1023) / 1024;
    if (roomy < 1 || mode == 2) roomy = 1;
    const uint32_t ccap = FUSED_CCAP * roomy;
    uint32_t clds = mode == 1 ? 1024u : 512u;
    uint32_t disc_lds = mode == 2 ? 512u : 0u;
    uint32_t iter_per_cu = 14u;
    if (mode == 2) { const char *a = getenv("HX_DISC_LDS"), *b = getenv("HX_ITER_PER_CU"); if (a && atoi(a) > 0) disc_lds = (uint32_t)atoi(a); if (b && atoi(b) > 0) iter_per_cu = (uint32_t)atoi(b); }   // tuning knobs
    { const char *cv = getenv(mode == 1 ? "HX_CLDS_INSERT" : "HX_CLDS_QUERY"); if (cv && atoi(cv) > 0) clds = (uint32_t)atoi(cv); }   // tuning knob
    clds = std::max<uint32_t>(clds, (uint32_t)((nch_ * 1024 + ((size_t)ef + 2) * 8 + 7) / 8));   // select scratch aliases C's LDS part
    auto lds_bytes = [&](uint32_t cc) { return ((size_t)cc + 2 * ((size_t)ef + 2) + 64 + 2 * mr.m) * 8 + (64 + 16 + 64) * 4 + nch_ * 1024 + (size_t)(disc_lds ? disc_lds + 64 + 160 + 32 : 0) * 8; };
    const size_t lds = lds_bytes(clds);
    // residency: one wave per workgroup, LDS-limited
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(16, (160 * 1024) / (lds + 512)));
    { const char *pv = getenv(mode == 0 ? "HX_QUERY_PER_CU" : "HX_INSERT_PER_CU"); if (mode != 2 && pv && atoi(pv) > 0) per_cu = std::min<uint32_t>(per_cu, (uint32_t)atoi(pv)); }   // tuning knob
    uint32_t grid = std::min<uint32_t>(ntasks, 256u * per_cu);
    // >= 2x the ids a search touches at its usual ~ef expansions; twice that on indexes of >= 4M rows, where some searches reach further.
    // Measured on 1M x 768: a table twice as large costs 3-6 % of the scan rate (cache footprint), one half as large overflows and retries.
    const uint64_t vis_need = ((uint64_t)ef * 2 * mr.m * 2 + 1024) * (n_rows >= 4000000ull ? 2 : 1);
    uint64_t vis_words = 4096; while (vis_words < vis_need) vis_words <<= 1;
    { const char *vv = getenv("HX_VIS_SHIFT"); if (vv && mode != 2) { const int sh = atoi(vv); if (sh < 0) vis_words >>= -sh; else vis_words <<= sh; if (vis_words < 4096) vis_words = 4096; } }   // tuning knob
    vis_words *= roomy;
    uint64_t disc_stride = 0;
    if (mode == 2) {
        // an iterative scan keeps its visited set and `discarded` heap across resumes: sized for max_scan_tuples (a query that
        // outgrows them reports FS_OVERFLOW and is re-run by the lock-step path); fewer resident workgroups bound the footprint
        const uint64_t mt = (uint64_t)std::min<long long>(std::max<long long>(it->max_tuples, 1), 1 << 20);
        disc_stride = std::max<uint64_t>(4 * mt, 16384);
        while (vis_words < 8 * mt + 4096) vis_words <<= 1;
        { const char *vv = getenv("HX_ITER_VIS_SHIFT"); if (vv) { const int sh = atoi(vv); if (sh < 0) vis_words >>= -sh; else vis_words <<= sh; if (vis_words < 8192) vis_words = 8192; } }   // tuning knob
        // at most ~12 GB of per-query state: a huge max_scan_tuples gets fewer resident queries, never less than one per CU pair
        const uint64_t per_wg = disc_stride * 8 + vis_words * 4;
        const uint32_t fit = (uint32_t)std::max<uint64_t>(128, (12ull << 30) / per_wg);
        grid = std::min<uint32_t>(grid, std::min<uint32_t>(256u * iter_per_cu, fit));
        const size_t need_disc = (size_t)grid * disc_stride * 8;
        if (need_disc > mr.cap_disc) {
            if (mr.d_disc) (void)hipFree(mr.d_disc);
            mr.d_disc = nullptr; mr.cap_disc = 0;
            HX_HIP(this, hipMalloc((void **)&mr.d_disc, need_disc));
            mr.cap_disc = need_disc;
        }
        if (mr.cap > mr.cap_emask) {
            if (mr.d_emask) (void)hipFree(mr.d_emask);
            mr.d_emask = nullptr; mr.cap_emask = 0;
            HX_HIP(this, hipMalloc((void **)&mr.d_emask, (size_t)mr.cap * 2));
            mr.cap_emask = mr.cap;
        }
        HX_HIP(this, hipMemcpyAsync(mr.d_emask, it->emask, (size_t)it->n_elems * 2, hipMemcpyHostToDevice, stream));
    }
    if (!mr.d_spill) HX_HIP(this, hipMalloc((void **)&mr.d_spill, (size_t)256 * 16 * FUSED_CCAP * 8));
    uint32_t *vis_ptr = nullptr; void *spill_ptr = mr.d_spill;
    if (roomy > 1) {   // a retry launch of a few overflowed tasks: private, larger tables sized for exactly this grid
        grid = std::min<uint32_t>(grid, 1024u);
        const size_t need_sp = (size_t)grid * ccap * 8; const uint64_t need_vis = (uint64_t)grid * vis_words;
        if (need_sp > mr.cap_spill_big) { if (mr.d_spill_big) (void)hipFree(mr.d_spill_big); mr.d_spill_big = nullptr; mr.cap_spill_big = 0; HX_HIP(this, hipMalloc(&mr.d_spill_big, need_sp)); mr.cap_spill_big = need_sp; }
        if (need_vis > mr.cap_vis_big) { if (mr.d_vis_big) (void)hipFree(mr.d_vis_big); mr.d_vis_big = nullptr; mr.cap_vis_big = 0; HX_HIP(this, hipMalloc((void **)&mr.d_vis_big, need_vis * 4)); mr.cap_vis_big = need_vis; }
        vis_ptr = mr.d_vis_big; spill_ptr = mr.d_spill_big;
    }
    if (roomy == 1 && (uint64_t)grid * vis_words > mr.cap_vis) {
        if (mr.d_vis) (void)hipFree(mr.d_vis);
        mr.d_vis = nullptr; mr.cap_vis = 0;
        const uint64_t n = (uint64_t)(mode == 2 ? grid : 256u * 16u) * vis_words;
        HX_HIP(this, hipMalloc((void **)&mr.d_vis, n * 4));
        mr.cap_vis = n;
    }
    const size_t out_n = mode != 1 ? (size_t)ntasks * k : (size_t)ntasks * FUSED_MAXL * 2 * mr.m;      // mode 2: k = limit
    const size_t cnt_n = mode != 1 ? (size_t)ntasks : (size_t)ntasks * FUSED_MAXL;
    // device task/in/out buffers (one allocation, reused)
    const size_t need = al16((size_t)ntasks * 4) * 3 + al16(out_n * 4) * 3 + al16(cnt_n * 4) + 256;
    if (need > mr.cap_io) {
        if (mr.d_io) (void)hipFree(mr.d_io);
        if (mr.h_io) (void)hipHostFree(mr.h_io);
        mr.d_io = mr.h_io = nullptr; mr.cap_io = 0;
        const size_t n = need * 2;
        HX_HIP(this, hipMalloc((void **)&mr.d_io, n));
        HX_HIP(this, hipHostMalloc((void **)&mr.h_io, n, hipHostMallocDefault));
        mr.cap_io = n;
    }
    size_t o = 0;
    const size_t o_ctr = o; o += 256;
    const size_t o_q = o; o += al16((size_t)ntasks * 4);
    const size_t o_lv = o; o += al16((size_t)ntasks * 4);
    const size_t in_bytes = o;
    const size_t o_st = o; o += al16((size_t)ntasks * 4);
    const size_t o_cnt = o; o += al16(cnt_n * 4);
    const size_t o_ids = o; o += al16(out_n * 4);
    const size_t o_d = o; o += al16(out_n * 4);
    const size_t o_tix = o; o += al16(out_n * 4);
    memset(mr.h_io + o_ctr, 0, 256);
    memcpy(mr.h_io + o_q, q_sel, (size_t)ntasks * 4);
    if (t_level) memcpy(mr.h_io + o_lv, t_level, (size_t)ntasks * 4); else memset(mr.h_io + o_lv, 0, (size_t)ntasks * 4);
    HX_HIP(this, hipMemcpyAsync(mr.d_io, mr.h_io, in_bytes, hipMemcpyHostToDevice, stream));
    FusedParams p;
    p.rows = d_rows; p.queries = d_queries; p.pitch = (uint32_t)pitch; p.nch = (uint32_t)((pitch + 1023) / 1024); p.n_rows = n_rows;
    p.l0_ids = mr.d_l0_ids; p.l0_cnt = mr.d_l0_cnt; p.level = mr.d_level; p.up_block = mr.d_up_block; p.up_ids = mr.d_up_ids; p.up_cnt = mr.d_up_cnt;
    p.l0_d = mr.d_l0_d; p.up_d = mr.d_up_d;
    p.m = mr.m; p.entry = entry; p.entry_level = entry_level;
    p.ntasks = ntasks; p.t_qsel = (const uint32_t *)(mr.d_io + o_q); p.t_level = (const int32_t *)(mr.d_io + o_lv);
    p.ef = ef; p.k = k; p.ccap = ccap; p.clds = clds;
    p.iter_mode = 0; p.limit = k; p.max_tuples = 0; p.emask = nullptr; p.disc = nullptr; p.disc_stride = 0; p.disc_lds = disc_lds; p.out_tix = (uint32_t *)(mr.d_io + o_tix);
    if (mode == 2) { p.iter_mode = (uint32_t)it->iter_mode; p.max_tuples = it->max_tuples; p.emask = mr.d_emask; p.disc = (unsigned long long *)mr.d_disc; p.disc_stride = (uint32_t)disc_stride; }
    p.spill = (uint2 *)spill_ptr; p.spill_stride = ccap;
    { const char *dv = getenv("HX_F_DBG"); p.fdbg = dv ? (uint32_t)atoi(dv) : 0u; }
    p.vis = vis_ptr ? vis_ptr : mr.d_vis; p.vis_words = vis_words;
    p.next_task = (uint32_t *)(mr.d_io + o_ctr);
    p.n_dist = (unsigned long long *)(mr.d_io + o_ctr + 8);
    p.out_ids = (uint32_t *)(mr.d_io + o_ids); p.out_d = (float *)(mr.d_io + o_d); p.out_cnt = (uint32_t *)(mr.d_io + o_cnt);
    p.status = (uint32_t *)(mr.d_io + o_st);
    if (timing) HX_HIP(this, hipEventRecord(ev0, stream));
    hipError_t ls = hipSuccess;
#define F32C(K) ls = launch_fused_mode<OpF32<K>>(this, p, grid, lds, mode)
#define F16C(K) ls = launch_fused_mode<OpF16<K>>(this, p, grid, lds, mode)
    HX_DISPATCH(this, F32C, F16C, ls = launch_fused_mode<OpHamming>(this, p, grid, lds, mode), ls = launch_fused_mode<OpJaccard>(this, p, grid, lds, mode));
#undef F32C
#undef F16C
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(ev1, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_io + o_ctr, mr.d_io + o_ctr, 256, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(mr.h_io + o_st, mr.d_io + o_st, (mode == 2 ? o : o_tix) - o_st, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipStreamSynchronize(stream));
    if (roomy == 1 && mode != 2) {   // test hook: pretend every k-th task overflowed, so that the roomy retry path is exercised
        const char *fv = getenv("HX_FORCE_OVERFLOW_MOD"); const uint32_t k = fv ? (uint32_t)atoi(fv) : 0u;
        if (k) for (uint32_t t = 0; t < ntasks; t += k) ((uint32_t *)(mr.h_io + o_st))[t] = FS_OVERFLOW;
    }
    if (view) {   // the caller reads the pinned staging buffer in place
        view->status = (const uint32_t *)(mr.h_io + o_st); view->cnt = (const uint32_t *)(mr.h_io + o_cnt);
        view->ids = (const uint32_t *)(mr.h_io + o_ids); view->d = (const float *)(mr.h_io + o_d);
    } else {
        memcpy(status, mr.h_io + o_st, (size_t)ntasks * 4);
        memcpy(out_cnt, mr.h_io + o_cnt, cnt_n * 4);
        memcpy(out_ids, mr.h_io + o_ids, out_n * 4);
        memcpy(out_d, mr.h_io + o_d, out_n * 4);
    }
    if (mode == 2) memcpy(it->out_tix, mr.h_io + o_tix, out_n * 4);
    unsigned long long nd[17]; memcpy(nd, mr.h_io + o_ctr + 8, 136);
    if (getenv("HX_F_DBG") && (atoi(getenv("HX_F_DBG")) & 4))
        fprintf(stderr, "[hx] k_fused mode %d tasks %u: 100 MHz ticks summed over waves: pop %llu list %llu visited %llu compact %llu dist %llu settle+filter %llu replay %llu; expansions %llu pushes %llu; inside dist: issue %llu wait %llu math %llu reduce %llu; select phase %llu\n",
                mode, ntasks, nd[3], nd[4], nd[5], nd[6], nd[7], nd[8], nd[9], nd[10], nd[11], nd[12], nd[13], nd[14], nd[15], nd[16]);
    if (counts) { counts[0] = nd[0]; counts[1] = nd[1]; }
    if (nd[2] > fused_cmax) fused_cmax = nd[2];
    if (timing) {
        float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, ev0, ev1));
        last_ms = ms; stat_fused.launches++; stat_fused.units += nd[0] + nd[1]; stat_fused.ms += ms;
    }
    return HX_OK;
}
