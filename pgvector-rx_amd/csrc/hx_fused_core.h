// hx_fused_core.h -- device code shared by the traversal kernel (hx_fused.inc.h) and the back-link kernels (hx_links.hip):
// kernel parameters, the wave-parallel Rust-std heaps, the visited table, and the query-vs-rows distance batches in the
// canonical summation order.
#pragma once
#include "hx_ops.h"

#include <cstdio>
#define FUSED_MAXCH 8          /* 1 KiB chunks per row: pitch <= 8192 B covers vector(2000), halfvec(4000), bit(64000) */
#ifndef FUSED_RB
#define FUSED_RB 4             /* rows in flight per wave ... */
#endif
#ifndef FUSED_MINW
#define FUSED_MINW 4
#endif
// In-kernel phase clocks (HX_F_DBG=4): compiled in only with -DFUSED_TIMERS (HX_CFLAGS=-DFUSED_TIMERS python pgvector-rx_amd/build.py --force) -- a dozen
// `if (timers)` branches per expansion and fourteen live counters cost the shipped kernels registers and issue slots for a diagnostic.
#ifdef FUSED_TIMERS
#define FUSED_TIMERS_ON true
#else
#define FUSED_TIMERS_ON false
#endif
#ifndef FUSED_BAR_BEFORE_ROWS
#define FUSED_BAR_BEFORE_ROWS false   /* true: wait for the visited-set CAS before the row loads of an expansion are issued (the behaviour up to round 2; A/B knob) */
#endif
#define FUSED_SLOTS_PER_CU 24u     /* the per-wave spill areas and visited tables are sized for this many resident searches per CU */
#ifndef FUSED_MINW_SA
#define FUSED_MINW_SA 4             /* the sorted-array query kernel */
#endif
#ifndef FUSED_MINW_INS
#define FUSED_MINW_INS 3            /* insert-mode kernel: 168 VGPRs, no spills (13 searches per CU are LDS-limited anyway); measured +2 % over 4 */
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
    uint32_t wcap;                               // entries of the W heap and of the sorted-result array in LDS: ef + 2, more for repair searches (skip-set members are not counted, so W outgrows ef)
    uint2 *spill; uint32_t spill_stride;         // per-workgroup spill area of the candidate heap (entries)
    uint32_t *vis; uint64_t vis_words;            // per-workgroup visited set: open-addressing table of vis_words (power of 2) row ids
    uint32_t *next_task;
    uint32_t *out_ids; float *out_d; uint32_t *out_cnt; uint32_t *status;
    uint32_t o_cst, o_lst;                        // MODE 1 output strides in 32-bit words: out_cnt[s*o_cst + layer], out_ids/out_d[s*o_lst + layer*2m + k] (status[t] stays per task)
    const uint32_t *t_oslot;                      // MODE 1: output slot s of task t (nullptr: s = t)
    uint2 *wtab; uint32_t wt_size, wt_slot0; uint8_t *wt_valid;   // MODE 1: the layer-0 result set W of task t as an open-addressing table of wt_size {d, id} entries at
                                                  // wtab + (wt_slot0 + s) * wt_size (the back-link kernels look d(new row, x) up there instead of streaming row x); nullptr: off
    // MODE 3 (insert, search only: select_neighbors runs on the matrix cores afterwards, hx_mfma.hip): the sorted result set W of layer lc of task t goes to
    // problem t_prob[t] + lc: wl_out[problem * wcap_out + i] = {distance bits, id}, wl_cnt[problem] = |W| (wcap_out = ef; wcap for repair searches)
    uint2 *wl_out; uint32_t *wl_cnt; const uint32_t *t_prob; uint32_t wcap_out;
    uint32_t sparse_cap;                          // sparsevec (OpSparse): index / value slots per row record
    // MODE 3 for vacuum's repair_graph_element (vacuum.rs:288-407): per task its own entry point (nullptr: entry / entry_level above) and the skip set --
    // elements with skip[e] != 0 and the repaired element itself are traversed but not counted into the result set (scan.rs:330-336, 416-419)
    const uint32_t *t_entry; const uint8_t *skip;
    uint32_t ondisk;                              // MODE 3 for aminsert: search_layer_disk semantics (W handed out nearest LAST, as scan.rs:441-446 sorts it)
    unsigned long long *n_dist;                   // [0] query-vs-row distances, [1] select distances, [2] max |C| seen
    // iterative scan (k_fused MODE 2): hnsw.iterative_scan relaxed_order (1) / strict_order (2), scan.rs:794-875
    uint32_t iter_mode, limit; long long max_tuples;
    const uint16_t *emask;                        // per element: bits 0-9 = which of its heap TIDs pass the filter, bits 12-15 = number of heap TIDs
    unsigned long long *disc; uint32_t disc_stride, disc_lds;   // per-workgroup tail of the `discarded` heap (entries), its LDS head
    uint32_t *out_tix;                            // which heap TID of the element each output is
    uint32_t sa;                                  // 1: searches with ef > 1 run on one sorted array (f_search_layer_sa); ties are redone by the heap kernel
    uint32_t fdbg;                                // experiments (HX_F_DBG): 1 no pre-filter, 4 phase timers into n_dist[3..7]
};

// The traversal kernels read their parameter block IN PLACE from the kernarg segment (constant address space: scalar loads), through a
// pointer the compiler cannot see through (k_fused launders it): a field is then s_load-ed where it is used -- hoisted out of a loop only when
// the loop always reads it -- instead of all ~60 dwords being loaded at kernel entry and held live (or spilled) to the end.
typedef const __attribute__((address_space(4))) FusedParams KParams;
// A fresh, opaque copy of the parameter pointer: loads through it cannot be merged with (or hoisted above) loads through an older copy, so what a
// phase derives from the parameters is computed where the phase starts and is dead when it ends, instead of living from kernel entry to exit.
__device__ __forceinline__ KParams &f_params_here(KParams &p) { KParams *q = &p; asm volatile("" : "+s"(q)); return *q; }

// one translation unit per element type holds the k_fused instantiations (hx_fused_f32.hip / _f16.hip / _bit.hip)
hipError_t hx_launch_fused_f32(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode);
hipError_t hx_launch_fused_f16(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode);
hipError_t hx_launch_fused_bit(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode);
// hx_biglist.hip: k_list_ops working on the device mirror's lists (the batch pipeline's groups of lists of 33..64 slots, m = 17..32)
struct ListMirrorArgs {
    uint32_t n_groups, m; const uint32_t *target, *layer, *op_off, *op_new, *gmap; const float *op_d;
    uint32_t *l0_ids; float *l0_d; uint16_t *l0_cnt; const uint32_t *up_block; uint32_t *up_ids; float *up_d; uint16_t *up_cnt;
    uint32_t *xrec; uint32_t xrec_words; unsigned long long *n_pairs;
};
hipError_t hx_launch_list_ops_mirror(hx_engine *e, const ListMirrorArgs &a);
hipError_t hx_launch_fused_sparse(hx_engine *e, int metric, const FusedParams &p, uint32_t grid, size_t lds, int mode);   // modes 0, 2, 3 (scans and search-only inserts)

// what the query-vs-rows helpers below need: the row store and 64 floats of LDS scratch for the short-row path.  Kept apart from FusedParams
// so that the kernel's parameters stay an immutable kernel argument (fields are s_load-ed from the kernarg segment where they are used instead of
// the whole struct being held live in scalar registers -- it was copied to a mutable local for the sake of `dsc`, and spilled)
struct FRows { const uint8_t *rows; uint32_t pitch, nch; float *dsc; uint32_t cap = 0; };   // cap: sparsevec records only (slots per record)

struct FHeapItem { float d; uint32_t id; };
__device__ __forceinline__ uint2 fh_pack(float d, uint32_t id) { return make_uint2(__builtin_bit_cast(unsigned int, d), id); }
__device__ __forceinline__ float fh_d(const uint2 &v) { return __builtin_bit_cast(float, v.x); }

// Heap storage: the first `L` entries live in LDS, the rest in this workgroup's spill area in global memory (only lane 0
// touches a heap, and a thread sees its own stores in program order).  Deep heaps are rare and only their bottom level
// spills, so the common case never leaves LDS while the LDS budget per search stays small.
typedef __attribute__((address_space(3))) uint2 lds_uint2;     // LDS-qualified: keeps heap accesses ds_read/ds_write, never FLAT
typedef __attribute__((address_space(3))) uint8_t lds_u8;       // a vector parked in LDS (query, select candidate): read with ds_read_b128 also where the pointer is picked at run time
typedef __attribute__((address_space(3))) u4 lds_u4;
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
/* A search is driven by ONE wavefront: what the traversal code needs between its steps is ordering inside that wave -- its earlier LDS and
   memory operations issued before its later ones -- never a workgroup barrier. */
#define F_BAR() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); } while (0)
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
// One 16-byte load of a bucket that bypasses the vector L1 (sc1 = agent scope: the inserts are L2 atomics, an L1 copy could be stale).  A relaxed
// agent-scope atomic load is at most 8 bytes wide, and two of them per bucket doubled the requests the visited test puts into the memory pipeline
// (a third of an expansion's requests); a buffer load takes the cache policy as an operand.  Only this wave writes its table and never while a
// test is in flight, so the 16 bytes need not be read atomically.  The table base is wave-uniform (one table per wave).
__device__ __forceinline__ u4 vis_load_bucket(const uint32_t *tab, uint32_t b)
{
    const uint64_t a = (uint64_t)tab;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(b * 16u), 0, 16 /* sc1 */);
}
// true: key present.  false: *slot = the empty slot the key belongs in
__device__ __forceinline__ bool vis_lookup(uint32_t *tab, uint32_t bmask, uint32_t key, uint32_t *&slot)
{
    uint32_t b = vis_mix(key) & bmask;
    for (;;) {
        const u4 bk = vis_load_bucket(tab, b);
        const uint32_t x = bk.x, y = bk.y, z = bk.z, w = bk.w;
        if (x == key || y == key || z == key || w == key) return true;
        const int e = x == VIS_EMPTY ? 0 : (y == VIS_EMPTY ? 1 : (z == VIS_EMPTY ? 2 : (w == VIS_EMPTY ? 3 : -1)));
        if (e >= 0) { slot = tab + 4u * b + (uint32_t)e; return false; }
        b = (b + 1u) & bmask;
    }
}
// vis_lookup whose home bucket `bk` was requested earlier (vis_load_bucket); valid as long as no insert of this wave happened in between
__device__ __forceinline__ bool vis_lookup_from(uint32_t *tab, uint32_t bmask, uint32_t key, uint32_t *&slot, u4 bk)
{
    uint32_t b = vis_mix(key) & bmask;
    for (;;) {
        const uint32_t x = bk.x, y = bk.y, z = bk.z, w = bk.w;
        if (x == key || y == key || z == key || w == key) return true;
        const int e = x == VIS_EMPTY ? 0 : (y == VIS_EMPTY ? 1 : (z == VIS_EMPTY ? 2 : (w == VIS_EMPTY ? 3 : -1)));
        if (e >= 0) { slot = tab + 4u * b + (uint32_t)e; return false; }
        b = (b + 1u) & bmask;
        bk = vis_load_bucket(tab, b);
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

// W table of an insert (written by k_fused<insert>, read by the back-link kernels): d(new row, key) if the new row's layer-0 search evaluated it
__device__ __forceinline__ bool wt_lookup(const uint2 *tab, uint32_t mask, uint32_t key, float &d)
{
    uint32_t s = vis_mix(key) & mask;
    for (;;) {
        const uint2 e = tab[s];
        if (e.y == key) { d = __builtin_bit_cast(float, e.x); return true; }
        if (e.y == VIS_EMPTY) return false;
        s = (s + 1u) & mask;
    }
}

struct FusedCtx {
    FRows fr;
    uint2 *C, *W, *EP, *RES, *RL, *DL; uint32_t *IDS, *CTL; lds_u8 *QV, *EV; HStore CH, WH;
    uint32_t *vis; uint32_t lane; uint32_t status;
    const uint8_t *skip; uint32_t skip_self;                                        // MODE 3 repair searches (nullptr: no skip set)
    GStore DS; lds_uint2 *DP, *WS; uint32_t *LV; uint32_t dlen, vcount;           // iterative scan: `discarded` min-heap, visited ids so far (the set survives resumes)
    unsigned long long nd0, nd1; uint32_t cmax;
    uint32_t tph[14];  // [13] select phase; diagnostic phase clocks (HX_F_DBG & 4): pop, list fetch, visited, compaction, distances, settle+prefilter, replay; [7] expansions, [8] heap pushes
};

// parks one vector (row or query slot) in LDS, chunk-major: bytes [c*1024 + 16*lane, +16); zero past the pitch
__device__ __forceinline__ void f_park(const FRows &p, const uint8_t *src, uint32_t lane, lds_u8 *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        u4 v = {0u, 0u, 0u, 0u};
        if (off < p.pitch) v = *(const u4 *)(src + off);
        *(lds_u4 *)(dst + off) = v;
    }
    F_BAR();
}

// f_park for code that runs in ONE wave of a multi-wave workgroup (no workgroup barrier)
__device__ __forceinline__ void f_park_w(const FRows &p, const uint8_t *src, uint32_t lane, lds_u8 *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        u4 v = {0u, 0u, 0u, 0u};
        if (off < p.pitch) v = *(const u4 *)(src + off);
        *(lds_u4 *)(dst + off) = v;
    }
    F_WSYNC();
}

// f_park without the trip through registers: gfx950's global_load_lds writes each lane's 16 bytes straight to
// LDS (destination = wave-uniform base + 16 * lane: exactly the parked layout) and completes asynchronously under vmcnt, so the
// NEXT select candidate's row travels while the current one is being compared.  Lanes past the row's end store zeros themselves.
__device__ __forceinline__ void f_park_async(const FRows &p, const uint8_t *src, uint32_t lane, lds_u8 *dst)
{
    for (uint32_t c = 0; c < p.nch; c++) {
        const uint32_t off = c * 1024u + lane * 16u;
        if (off < p.pitch)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t *)(src + off),
                                             (__attribute__((address_space(3))) uint32_t *)(dst + c * 1024u), 16, 0, 0);
        else *(lds_u4 *)(dst + off) = u4{0u, 0u, 0u, 0u};
    }
}

// Short rows (payload <= 512 B: bit(1024), vector(128), the reference's 3-d tests): a 64-lane wave per row would leave
// most lanes idle and cost one memory hop per FUSED_RB rows.  Here LPR = 8/16/32 lanes share a row (as K1 does), 64/LPR rows
// are read by ONE load instruction, and up to eight such instructions are in flight, so a whole neighbour list is one hop.
// The bits are the canonical ones: the lanes a short row does not reach contribute +0.0 partials in the 64-lane order.
template <class OP, int LPR>
__device__ __forceinline__ float f_dist_small(const FRows &p, const lds_u8 *qv, const uint32_t *ids, uint32_t n, uint32_t lane)
{
    constexpr int R = 64 / LPR, PF = 8;
    const uint32_t g = lane / LPR, loff = (lane % LPR) * 16u;
    const bool in = loff < p.pitch;
    u4 q = {0u, 0u, 0u, 0u};
    if (in) q = *(const lds_u4 *)(qv + loff);
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
__device__ __forceinline__ float f_dist_batch(const FRows &p, const lds_u8 *qv, const uint32_t *ids, uint32_t n, uint32_t lane, uint32_t *tk = nullptr)
{
    if constexpr (OP::kSparse) {     // sparsevec: lane j walks the merge join of (parked query record, row ids[j]) in the reference's own order (hx_ops.h: sp_merge)
        float d = 0.0f;
        if (lane < n) d = sp_merge<OP::kind>(sp_row((const uint8_t *)qv, p.cap), sp_row(p.rows + (size_t)ids[lane] * p.pitch, p.cap));   // distance(query, element), graph/mod.rs:221
        return d;
    } else {
    if constexpr (LPR < 64) return f_dist_small<OP, LPR>(p, qv, ids, n, lane);
    float mine = 0.0f;
    unsigned long long tq = tk ? __builtin_amdgcn_s_memtime() : 0ull;
#define FD_TICK(k) do { if (tk) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tk[k] += (uint32_t)(t_ - tq); tq = t_; } } while (0)
    const uint32_t loff = lane * 16u;
    const bool ragged = (p.pitch & 1023u) != 0u;              // the last chunk of a row is partial: lanes past the payload must contribute zeros
    for (uint32_t j0 = 0; j0 < n; j0 += RB) {
        // Row bases are wave-uniform (an id is one value for the whole wave): kept in scalar registers, so a load is base + per-lane offset and
        // costs no address VGPRs.  Every load of a round is issued unconditionally, back to back, into its own registers: a lane past the payload
        // re-reads the row's first bytes and its fragment is zeroed afterwards (a conditional load per fragment made the compiler serialise the
        // round -- measured as two extra memory latencies per row batch).
        const uint8_t *rb[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const uint32_t id = (uint32_t)__builtin_amdgcn_readfirstlane((int)ids[j0 + r < n ? j0 + r : j0]);
            rb[r] = p.rows + (size_t)id * p.pitch;
        }
        typename OP::acc_t acc[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) OP::init(acc[r]);
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < p.nch; c0 += FUSED_CG) {
            const uint32_t kc = p.nch - c0 < (uint32_t)FUSED_CG ? p.nch - c0 : (uint32_t)FUSED_CG;   // chunks of this round (uniform)
            u4 rv[RB][FUSED_CG];
            uint32_t offk[FUSED_CG]; bool ink[FUSED_CG];
#pragma unroll
            for (int k = 0; k < FUSED_CG; k++) { const uint32_t o = (c0 + (uint32_t)k) * 1024u + loff; ink[k] = o < p.pitch; offk[k] = ink[k] ? o : 0u; }
            if (kc == (uint32_t)FUSED_CG) {
#pragma unroll
                for (int k = 0; k < FUSED_CG; k++)
#pragma unroll
                    for (int r = 0; r < RB; r++) rv[r][k] = *(const u4 *)(rb[r] + offk[k]);
            } else {
#pragma unroll
                for (int k = 0; k < FUSED_CG - 1; k++)
                    if ((uint32_t)k < kc) {
#pragma unroll
                        for (int r = 0; r < RB; r++) rv[r][k] = *(const u4 *)(rb[r] + offk[k]);
                    }
            }
            FD_TICK(9);                                   // addresses + load issue
            if (tk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FD_TICK(10);                                  // waiting for the rows
#pragma unroll
            for (int k = 0; k < FUSED_CG; k++) {
                if ((uint32_t)k < kc) {
                    const u4 q = *(const lds_u4 *)(qv + (c0 + (uint32_t)k) * 1024u + loff);
                    if (ragged && c0 + (uint32_t)k + 1u == p.nch) {                 // the row's partial last chunk: lanes past the payload contribute zeros
#pragma unroll
                        for (int r = 0; r < RB; r++) {
                            u4 v = rv[r][k];
                            if (!ink[k]) v = u4{0u, 0u, 0u, 0u};
                            OP::add(acc[r], q, v);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < RB; r++) OP::add(acc[r], q, rv[r][k]);
                    }
                }
            }
            if (tk) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FD_TICK(11);                                  // arithmetic
        }
        if constexpr (OP::kFloatAcc && RB == 4) {           // the four rows' butterflies share their steps (hx_ops.h: lanes_sum4_f)
            const float z = lanes_sum4_f(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float d = OP::post(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, z), lanes_sum4_lane(r))));
                if (j0 + r < n && lane == j0 + r) mine = d;
            }
        } else {
#pragma unroll
            for (int r = 0; r < RB; r++) {
                const float d = OP::template finish<64>(acc[r]);
                if (j0 + r < n && lane == j0 + r) mine = d;
            }
        }
        FD_TICK(12);                                      // reductions
    }
    return mine;
#undef FD_TICK
    }
}

// check_element_closer (graph/mod.rs:315-339): is any d(q, ids[j]) <= thr?  Rows are evaluated FUSED_RB at a time in list
// order and the scan stops at the first batch that contains a hit, like the reference's early `return false`
// (the answer is the same; fewer rows are streamed for rejected candidates).  *n_eval += rows evaluated.
template <class OP, int LPR>
__device__ __forceinline__ bool f_any_le(const FRows &p, const lds_u8 *qv, const uint32_t *ids, uint32_t n, uint32_t lane, float thr,
                                         unsigned long long &n_eval)
{
    constexpr uint32_t B = OP::kSparse ? 64u : f_step_rows<LPR>();   // sparsevec: one lane per row, so a step holds a lane-full of rows
    for (uint32_t j0 = 0; j0 < n; j0 += B) {
        const uint32_t nb = n - j0 < B ? n - j0 : B;
        const float d = f_dist_batch<OP, LPR>(p, qv, ids + j0, nb, lane);
        n_eval += nb;
        if (__ballot(lane < nb && d <= thr) != 0ull) return true;
    }
    return false;
}


// query-vs-rows distances of one expansion / one check_element_closer step
template <class OP, int LPR, int RB = FUSED_RB>
__device__ __forceinline__ float f_dist(FusedCtx &cx, const lds_u8 *qv, const uint32_t *ids, uint32_t n, uint32_t lane, uint32_t *tk = nullptr)
{
    return f_dist_batch<OP, LPR, RB>(cx.fr, qv, ids, n, lane, tk);
}

template <class OP, int LPR>
__device__ __forceinline__ bool f_any_le_x(FusedCtx &cx, const lds_u8 *qv, const uint32_t *ids, uint32_t n, uint32_t lane, float thr, unsigned long long &n_eval)
{
    return f_any_le<OP, LPR>(cx.fr, qv, ids, n, lane, thr, n_eval);
}
