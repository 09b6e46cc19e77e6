// hx_index.cpp -- host-side HNSW graph driver: the reference's graph functions re-designed as resumable
// state machines that run MANY inserts / queries in lock-step and hand each step's candidate rows to the
// device kernels of hx_engine.hip in one launch.
//
// Mirrors (same names, same control flow per insert/query; every distance comes from the engine):
//   search_layer               src/graph/mod.rs:161-255      -> SearchCore (build mode)
//   select_neighbors           src/graph/mod.rs:269-308      -> SelectTask
//   check_element_closer       src/graph/mod.rs:315-339      -> SelectTask::process_block
//   find_element_neighbors     src/graph/mod.rs:355-427      -> InsertTask
//   update_neighbor_connections src/graph/mod.rs:442-489     -> BacklinkTask
//   build_callback             src/index/build.rs:400-535    -> hx_index_insert
//   search_layer_disk          src/index/scan.rs:302-448     -> SearchCore (scan mode)
//   get_scan_items / resume_scan_items / amgettuple  src/index/scan.rs:458-577, 709-876 -> QueryTask
//
// Why lock-step: one expansion evaluates <= 2M rows (32 rows x 3 KB at d=768), far too little to keep an
// MI355X busy.  The distances of one expansion do not depend on heap state, so the host computes them
// for thousands of independent searches at once and then replays each search's heap logic in the
// reference's order: per search the control flow is identical to the sequential code.
//
// The heaps restate Rust std::collections::BinaryHeap (sift_up / sift_down_to_bottom) so that the order
// among equal distances matches; see DESIGN.md "tie order".
#include "hx_internal.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <chrono>
#include <unordered_map>

namespace {

struct Cand { float d; uint32_t id; };

// ------------------------------------------------------------------------------------------------
// BinaryHeap with Rust std's exact sift order.  NEAREST=true: smallest distance on top
// (NearestCandidate, graph/mod.rs:103-112); false: largest on top (FurthestCandidate, :131-139).
// ------------------------------------------------------------------------------------------------
template <bool NEAREST> struct RHeap {
    std::vector<Cand> v;
    static bool le(const Cand &a, const Cand &b) { return NEAREST ? !(b.d > a.d) : !(a.d > b.d); }   // cmp(a,b) != Greater
    bool empty() const { return v.empty(); }
    size_t size() const { return v.size(); }
    const Cand &top() const { return v[0]; }
    void clear() { v.clear(); }
    size_t sift_up(size_t start, size_t pos)
    {
        Cand e = v[pos];
        while (pos > start) {
            size_t parent = (pos - 1) / 2;
            if (le(e, v[parent])) break;
            v[pos] = v[parent]; pos = parent;
        }
        v[pos] = e;
        return pos;
    }
    void push(Cand c) { v.push_back(c); sift_up(0, v.size() - 1); }
    void sift_down_to_bottom(size_t pos)
    {
        const size_t end = v.size(), start = pos;
        Cand e = v[pos];
        size_t child = 2 * pos + 1;
        while (end >= 2 && child <= end - 2) {
            if (le(v[child], v[child + 1])) child += 1;
            v[pos] = v[child]; pos = child; child = 2 * pos + 1;
        }
        if (child == end - 1) { v[pos] = v[child]; pos = child; }
        v[pos] = e;
        sift_up(start, pos);
    }
    bool pop(Cand &out)
    {
        if (v.empty()) return false;
        Cand item = v.back(); v.pop_back();
        if (!v.empty()) { std::swap(item, v[0]); sift_down_to_bottom(0); }
        out = item;
        return true;
    }
};

// visited set: open addressing on row ids (only membership matters: HashSet<usize>, graph/mod.rs:171)
struct VisitedSet {
    std::vector<uint32_t> tab; size_t n = 0, mask = 0;
    static constexpr uint32_t EMPTY = 0xffffffffu;
    void reset(size_t expect)
    {
        size_t cap = 256; while (cap < expect * 2) cap <<= 1;
        if (tab.size() != cap) tab.assign(cap, EMPTY); else std::fill(tab.begin(), tab.end(), EMPTY);
        mask = cap - 1; n = 0;
    }
    static inline uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
    void grow()
    {
        std::vector<uint32_t> old; old.swap(tab);
        tab.assign(old.size() * 2, EMPTY); mask = tab.size() - 1; n = 0;
        for (uint32_t k : old) if (k != EMPTY) test_and_set(k);
    }
    // returns true if already present
    bool test_and_set(uint32_t k)
    {
        if ((n + 1) * 2 > tab.size()) grow();
        size_t i = mix(k) & mask;
        while (true) {
            uint32_t t = tab[i];
            if (t == k) return true;
            if (t == EMPTY) { tab[i] = k; n++; return false; }
            i = (i + 1) & mask;
        }
    }
    bool contains(uint32_t k) const
    {
        if (tab.empty()) return false;
        size_t i = mix(k) & mask;
        while (true) { uint32_t t = tab[i]; if (t == k) return true; if (t == EMPTY) return false; i = (i + 1) & mask; }
    }
};

// ------------------------------------------------------------------------------------------------
// graph storage (GraphElement / NeighborArray, graph/mod.rs:24-84), flat arrays indexed by row id
// ------------------------------------------------------------------------------------------------
constexpr int HEAPTIDS = 10;   // hnsw_constants.rs:85

// Allocator whose value-less construct() default-initialises: growing the list / TID arrays by a batch then touches no memory (a list's slots beyond
// its count and a TID array's beyond ntids are never read), so a 10M-row build does not page in and zero 3 GB of host arrays it may never look at --
// during device batches the lists live in the mirror (the replicated O(batch) host work is what bounds the multi-GPU speed-up).
template <class T> struct NoInitAlloc {
    using value_type = T;
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(::operator new(n * sizeof(T))); }
    void deallocate(T *p, size_t) { ::operator delete(p); }
    template <class U, class... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U; else ::new ((void *)p) U(std::forward<A>(a)...);
    }
    template <class U> bool operator==(const NoInitAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const NoInitAlloc<U> &) const { return false; }
};

struct Graph {
    int m = 16;
    std::vector<int32_t> level;              // < 0: tombstoned duplicate
    std::vector<uint16_t> n0_cnt; std::vector<Cand, NoInitAlloc<Cand>> n0;   // layer 0: stride 2m
    std::vector<uint64_t> up_off;                                    // first slot of layers 1..level in `up`
    std::vector<Cand, NoInitAlloc<Cand>> up; std::vector<uint16_t> up_cnt;   // up: stride m per layer; up_cnt per (elem, layer)
    std::vector<uint64_t> upc_off;
    std::vector<std::array<int64_t, HEAPTIDS>, NoInitAlloc<std::array<int64_t, HEAPTIDS>>> tids; std::vector<uint8_t> ntids;
    std::vector<uint8_t> deleted;           // HnswElementTupleData.deleted (types/hnsw.rs:112-137): set by vacuum's mark_deleted
    int64_t entry = -1;

    uint32_t size() const { return (uint32_t)level.size(); }
    void reserve(size_t n)      // no reallocation (and re-copy of gigabytes) while a bulk insert appends n more elements
    {
        const size_t t = level.size() + n;
        level.reserve(t); deleted.reserve(t); n0_cnt.reserve(t); n0.reserve(t * 2 * (size_t)m); up_off.reserve(t); upc_off.reserve(t); tids.reserve(t); ntids.reserve(t);
        up.reserve(up.size() + n / 8 * (size_t)m + 64); up_cnt.reserve(up_cnt.size() + n / 8 + 64);
    }
    int lm(int layer) const { return layer == 0 ? 2 * m : m; }       // hnsw_get_layer_m, hnsw_constants.rs:122-128
    uint32_t add(int lv)
    {
        uint32_t id = size();
        level.push_back(lv);
        n0_cnt.push_back(0); n0.resize(n0.size() + 2 * (size_t)m);
        up_off.push_back(up.size()); upc_off.push_back(up_cnt.size());
        if (lv > 0) { up.resize(up.size() + (size_t)lv * m); up_cnt.resize(up_cnt.size() + lv, 0); }
        tids.emplace_back(); ntids.push_back(0); deleted.push_back(0);
        return id;
    }
    // b elements at once (a batch): every array grows once instead of b times
    void add_bulk(const int32_t *levels, uint32_t b, int max_level)
    {
        const size_t n0sz = level.size();
        size_t up_add = 0;
        for (uint32_t i = 0; i < b; i++) { int lv = std::min(levels[i], max_level); if (lv < 0) lv = 0; up_add += (size_t)lv; }
        level.resize(n0sz + b); n0_cnt.resize(n0sz + b, 0); n0.resize((n0sz + b) * 2 * (size_t)m);
        up_off.resize(n0sz + b); upc_off.resize(n0sz + b); tids.resize(n0sz + b); ntids.resize(n0sz + b, 0); deleted.resize(n0sz + b, 0);
        size_t uo = up.size(), co = up_cnt.size();
        up.resize(uo + up_add * (size_t)m); up_cnt.resize(co + up_add, 0);
        for (uint32_t i = 0; i < b; i++) {
            int lv = std::min(levels[i], max_level); if (lv < 0) lv = 0;
            level[n0sz + i] = lv; up_off[n0sz + i] = uo; upc_off[n0sz + i] = co;
            uo += (size_t)lv * m; co += (size_t)lv;
        }
    }
    Cand *list(uint32_t e, int layer) { return layer == 0 ? &n0[(size_t)e * 2 * m] : &up[up_off[e] + (size_t)(layer - 1) * m]; }
    const Cand *list(uint32_t e, int layer) const { return const_cast<Graph *>(this)->list(e, layer); }
    uint16_t &cnt(uint32_t e, int layer) { return layer == 0 ? n0_cnt[e] : up_cnt[upc_off[e] + (layer - 1)]; }
    uint16_t cnt(uint32_t e, int layer) const { return const_cast<Graph *>(this)->cnt(e, layer); }
};

static void stable_sort_asc(std::vector<Cand> &v) { std::stable_sort(v.begin(), v.end(), [](const Cand &a, const Cand &b) { return a.d < b.d; }); }
static void stable_sort_desc(std::vector<Cand> &v) { std::stable_sort(v.begin(), v.end(), [](const Cand &a, const Cand &b) { return b.d < a.d; }); }

// ------------------------------------------------------------------------------------------------
// lock-step task protocol
// ------------------------------------------------------------------------------------------------
struct PairGroup { uint16_t na, nb; uint8_t mfma = 0; };   // mfma: evaluate on the matrix cores (hx_mfma.hip), values within a known band of the canonical ones

struct LsTask {
    // request produced by advance(): at most one distance group and any number of pair groups
    uint32_t q_sel = 0;
    std::vector<uint32_t> dist_ids;
    std::vector<PairGroup> pgroups; std::vector<uint32_t> pair_ids;
    // where the scheduler put the results of the last request
    size_t dist_off = 0, pair_out_off = 0;
    bool fresh = true;
    uint64_t n_dist = 0, n_pair = 0;          // distance evaluations requested by this task
    virtual ~LsTask() {}
    // consumes the results of its previous request (dres/pres point at this task's slices) and either
    // posts a new request (returns true) or finishes (returns false)
    virtual bool advance(const float *dres, const float *pres) = 0;
    void clear_req() { dist_ids.clear(); pgroups.clear(); pair_ids.clear(); }
};

// ------------------------------------------------------------------------------------------------
// Algorithm 2, resumable.  Build mode = search_layer (graph/mod.rs:161-255); scan mode =
// search_layer_disk (scan.rs:302-448) with its `discarded` heap for iterative scans.
// ------------------------------------------------------------------------------------------------
struct SearchCore {
    const Graph *g = nullptr;
    int layer = 0; size_t ef = 1; bool scan_mode = false;
    RHeap<true> C; RHeap<false> W; size_t wlen = 0;
    VisitedSet own_vis; VisitedSet *vis = nullptr;
    RHeap<true> *discarded = nullptr;
    std::vector<uint32_t> pend;
    bool finished = true;
    // vacuum's repair search (search_layer_disk's skip_count, scan.rs:331-336,416-419): members of `skip` and `skip_self` are traversed and kept in W
    // but do not count towards ef; elements flagged in `deleted` are visited and dropped (load_element -> None, scan.rs:178-181)
    const uint8_t *skip = nullptr, *deleted = nullptr; uint32_t skip_self = 0xFFFFFFFFu;
    inline bool counted(uint32_t e) const { return e != skip_self && !(skip && skip[e]); }

    void start(const Graph *gr, const std::vector<Cand> &ep, size_t ef_, int layer_, bool scan, VisitedSet *shared_vis,
               RHeap<true> *disc, bool add_entry_to_visited)
    {
        g = gr; ef = ef_; layer = layer_; scan_mode = scan; discarded = disc;
        C.clear(); W.clear(); wlen = 0; pend.clear(); finished = false;
        if (shared_vis) vis = shared_vis; else { own_vis.reset(ef * 8 + 64); vis = &own_vis; }
        for (const Cand &e : ep) {
            if (add_entry_to_visited) vis->test_and_set(e.id);
            C.push(e); W.push(e); if (counted(e.id)) wlen++;
        }
    }
    inline void apply(uint32_t e, float d)
    {
        const bool always_add = wlen < ef;
        bool add;
        if (!scan_mode) { const float f = W.empty() ? FLT_MAX : W.top().d; add = d < f || always_add; }          // mod.rs:226-229
        else { const double f = W.empty() ? DBL_MAX : (double)W.top().d; add = !(!always_add && (double)d >= f); } // scan.rs:372-383,195-200
        if (add) {
            Cand c{d, e};
            C.push(c); W.push(c); if (counted(e)) wlen++;
            if (wlen > ef) { Cand ev; W.pop(ev); wlen--; if (discarded) discarded->push(ev); }   // mod.rs:239-242 / scan.rs:420-429
        } else if (discarded) {
            discarded->push(Cand{d, e});                                                        // scan.rs:385-404
        }
    }
    // returns true when `pend` holds rows whose distances are needed
    bool run(const float *res)
    {
        if (!pend.empty()) { for (size_t k = 0; k < pend.size(); k++) apply(pend[k], res[k]); pend.clear(); }
        Cand c;
        while (C.pop(c)) {
            if (!scan_mode) { const float f = W.empty() ? FLT_MAX : W.top().d; if (c.d > f) break; }
            else { const double f = W.empty() ? DBL_MAX : (double)W.top().d; if ((double)c.d > f) { if (discarded) discarded->push(c); break; } }
            if (g->level[c.id] < layer) continue;                                    // mod.rs:198-200
            const Cand *nb = g->list(c.id, layer); const uint16_t n = g->cnt(c.id, layer);
            for (uint16_t k = 0; k < n; k++) {
                const uint32_t e = nb[k].id;
                if (vis->test_and_set(e)) continue;                                  // mod.rs:206-209
                if (deleted && deleted[e]) continue;
                if (layer > 0 && g->level[e] < layer) continue;                      // mod.rs:213-216; at layer 0 every linked element qualifies (tombstones are never linked)
                pend.push_back(e);
            }
            if (!pend.empty()) return true;
        }
        if (discarded) { Cand r; while (C.pop(r)) discarded->push(r); }              // scan.rs:433-438
        finished = true;
        return false;
    }
    void results_asc(std::vector<Cand> &out) const { out = W.v; stable_sort_asc(out); }    // mod.rs:248-254
    void results_desc(std::vector<Cand> &out) const { out = W.v; stable_sort_desc(out); }  // scan.rs:441-446 (nearest last)
};

// ------------------------------------------------------------------------------------------------
// select_neighbors (graph/mod.rs:269-308), resumable: candidates are processed in blocks; the
// operands of check_element_closer for a whole block are fetched as pair groups in one step.
// ------------------------------------------------------------------------------------------------
struct SelectTask {
    const std::vector<Cand> *cands = nullptr; size_t maxn = 0;
    std::vector<Cand> R, disc; std::vector<int> r_blk;     // r_blk[i] = index inside the current block of R[i], or -1
    size_t pos = 0, blk = 0, r0 = 0;
    bool finished = true;
    static constexpr size_t FIRST_BLOCK = 48, NEXT_BLOCK = 32, RCHUNK = 32;
    // MFMA mode (halfvec inner product): the block's pair values come from the matrix cores first; a decision `d(e, r) <= d(e, q)` (mod.rs:333)
    // whose two sides are closer than the summation-order band is re-requested in the canonical order before the block is processed
    bool mfma = false; const float *norm2 = nullptr; float band_k = 0.0f;
    int phase = 0;                          // 1: approximate values requested, 2: exact fix-ups requested
    std::vector<float> vals; std::vector<uint32_t> fix;
    uint64_t n_mfma = 0, n_exact = 0;

    void start(const std::vector<Cand> *c, size_t maxn_)
    {
        cands = c; maxn = maxn_; R.clear(); disc.clear(); r_blk.clear(); pos = 0; blk = 0; r0 = 0; finished = false; phase = 0;
        if (c->size() <= maxn) { R = *c; finished = true; }                          // mod.rs:276-278
    }
    // posts the pair groups of the next block into t; returns false when there is nothing left to ask
    bool post(LsTask &t)
    {
        const size_t n = cands->size();
        if (finished) return false;
        if (R.size() >= maxn || pos >= n) { finish(); return false; }
        r0 = R.size();
        blk = std::min(n - pos, r0 == 0 ? FIRST_BLOCK : NEXT_BLOCK);
        for (auto &x : r_blk) x = -1;
        if (blk >= 2) {                                                              // triangle inside the block
            t.pgroups.push_back(PairGroup{(uint16_t)blk, 0, (uint8_t)mfma});
            for (size_t k = 0; k < blk; k++) t.pair_ids.push_back((*cands)[pos + k].id);
            t.n_pair += blk * (blk - 1) / 2;
        }
        for (size_t c0 = 0; c0 < r0; c0 += RCHUNK) {                                 // block x (R as of block start)
            const size_t nbk = std::min(RCHUNK, r0 - c0);
            t.pgroups.push_back(PairGroup{(uint16_t)blk, (uint16_t)nbk, (uint8_t)mfma});
            for (size_t k = 0; k < blk; k++) t.pair_ids.push_back((*cands)[pos + k].id);
            for (size_t j = 0; j < nbk; j++) t.pair_ids.push_back(R[c0 + j].id);
            t.n_pair += blk * nbk;
        }
        if (t.pgroups.empty()) {   // single candidate and empty R: nothing to compare against
            process_block(nullptr);
            return post(t);
        }
        if (mfma) phase = 1;
        return true;
    }
    // consumes the results of the previous request (if any) and posts the next one; false: select_neighbors is complete
    bool step(const float *pres, LsTask &t)
    {
        if (pres) {
            if (!mfma) process_block(pres);
            else if (phase == 1) {
                const size_t n_tri = blk >= 2 ? blk * (blk - 1) / 2 : 0;
                size_t n_all = n_tri;
                for (size_t c0 = 0; c0 < r0; c0 += RCHUNK) n_all += blk * std::min(RCHUNK, r0 - c0);
                vals.assign(pres, pres + n_all); n_mfma += n_all;
                // pairs whose decision the summation order could flip: |value - threshold| within the band
                fix.clear();
                auto in_band = [&](size_t idx, const Cand &a, uint32_t b_id) {
                    const float band = band_k * std::sqrt(norm2[a.id] * norm2[b_id]);
                    return std::fabs(vals[idx] - a.d) <= band;
                };
                for (size_t k = 0; k < blk; k++) {
                    const Cand &a = (*cands)[pos + k];
                    const size_t first = fix.size();
                    for (size_t kk = 0; kk < k; kk++) if (in_band(k * (k - 1) / 2 + kk, a, (*cands)[pos + kk].id)) fix.push_back((uint32_t)(k * (k - 1) / 2 + kk));
                    size_t base = n_tri;
                    for (size_t c0 = 0; c0 < r0; c0 += RCHUNK) {
                        const size_t nbk = std::min(RCHUNK, r0 - c0);
                        for (size_t j = 0; j < nbk; j++) if (in_band(base + k * nbk + j, a, R[c0 + j].id)) fix.push_back((uint32_t)(base + k * nbk + j));
                        base += blk * nbk;
                    }
                    // exact request for row a against its in-band partners: 1 x nb rectangles (<= 63 partners each)
                    for (size_t f0 = first; f0 < fix.size(); f0 += 63) {
                        const size_t nbk = std::min<size_t>(63, fix.size() - f0);
                        t.pgroups.push_back(PairGroup{1, (uint16_t)nbk, 0});
                        t.pair_ids.push_back(a.id);
                        for (size_t f = f0; f < f0 + nbk; f++) {
                            const size_t idx = fix[f];
                            uint32_t b_id;
                            if (idx < n_tri) b_id = (*cands)[pos + (idx - k * (k - 1) / 2)].id;
                            else { size_t rem = idx - n_tri, c0 = 0; for (;; c0 += RCHUNK) { const size_t nbk2 = std::min(RCHUNK, r0 - c0); if (rem < blk * nbk2) { b_id = R[c0 + rem % nbk2].id; break; } rem -= blk * nbk2; } }
                            t.pair_ids.push_back(b_id);
                        }
                        t.n_pair += nbk;
                    }
                }
                if (!fix.empty()) { n_exact += fix.size(); phase = 2; return true; }
                process_block(vals.data()); phase = 0;
            } else if (phase == 2) {
                for (size_t f = 0; f < fix.size(); f++) vals[fix[f]] = pres[f];      // the exact groups were posted in `fix` order, one value each
                process_block(vals.data()); phase = 0;
            }
        }
        return post(t);
    }
    void process_block(const float *res)
    {
        const float *tri = nullptr; const float *rect = res;
        if (blk >= 2) { tri = res; rect = res + blk * (blk - 1) / 2; }
        for (size_t k = 0; k < blk; k++) {
            if (R.size() >= maxn) break;                                             // mod.rs:285-287
            const Cand e = (*cands)[pos + k];
            bool closer = true;                                                      // check_element_closer, mod.rs:315-339
            for (size_t ri = 0; ri < R.size(); ri++) {
                float d;
                if (ri < r0) { const size_t chunk = ri / RCHUNK, j = ri % RCHUNK, nbk = std::min(RCHUNK, r0 - chunk * RCHUNK);
                               d = rect[chunk * RCHUNK * blk + k * nbk + j]; }
                else { const size_t kk = (size_t)r_blk[ri]; d = tri[k * (k - 1) / 2 + kk]; }
                if (d <= e.d) { closer = false; break; }                             // mod.rs:333-335
            }
            if (closer) { R.push_back(e); r_blk.push_back((int)k); } else disc.push_back(e);
        }
        pos += blk;
    }
    void finish()
    {
        for (const Cand &d : disc) { if (R.size() >= maxn) break; R.push_back(d); }  // mod.rs:300-305
        finished = true;
    }
};

// ------------------------------------------------------------------------------------------------
// find_element_neighbors (graph/mod.rs:355-427) for one new element
// ------------------------------------------------------------------------------------------------
struct InsertTask : LsTask {
    const Graph *g; uint32_t id; int new_level, entry_level; uint32_t entry; int efc;
    enum { S_INIT, S_ENTRY, S_GREEDY, S_SEARCH, S_SELECT, S_DONE } st = S_INIT;
    int lc = 0;
    std::vector<Cand> ep, w;
    SearchCore sc; SelectTask sel;
    std::vector<std::vector<Cand>> nb;     // selected neighbours per layer 0..new_level
    // the layers' result sets W (ascending) when the traversal kernel has already searched (MODE 3: m > 32, whose select and back-links the device
    // kernels do not serve): the task then only runs select_neighbors per layer
    std::vector<std::vector<Cand>> preset; bool has_preset = false;

    bool post_search() { dist_ids = sc.pend; q_sel = id; n_dist += dist_ids.size(); return true; }
    bool advance(const float *dres, const float *pres) override
    {
        clear_req();
        for (;;) {
            switch (st) {
            case S_INIT:
                nb.assign(new_level + 1, {});
                if (has_preset) {                                                    // mod.rs:403-427 with W given
                    lc = std::min(new_level, entry_level);
                    if (lc < 0) { st = S_DONE; break; }
                    w = preset[lc]; sel.start(&w, (size_t)g->lm(lc)); st = S_SELECT; pres = nullptr;
                    break;
                }
                dist_ids.push_back(entry); q_sel = id; n_dist += 1; st = S_ENTRY;    // mod.rs:371-377
                return true;
            case S_ENTRY:
                ep.assign(1, Cand{dres[0], entry});
                lc = entry_level; st = S_GREEDY;
                if (lc >= new_level + 1) sc.start(g, ep, 1, lc, false, nullptr, nullptr, true);
                dres = nullptr;
                break;
            case S_GREEDY:                                                           // mod.rs:385-399
                if (lc < new_level + 1) {
                    lc = std::min(new_level, entry_level); st = S_SEARCH;
                    if (lc >= 0) sc.start(g, ep, (size_t)efc, lc, false, nullptr, nullptr, true);
                    break;
                }
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_asc(w);
                if (!w.empty()) ep.assign(1, w[0]);
                lc--;
                if (lc >= new_level + 1) sc.start(g, ep, 1, lc, false, nullptr, nullptr, true);
                break;
            case S_SEARCH:                                                           // mod.rs:403-416
                if (lc < 0) { st = S_DONE; break; }
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_asc(w);
                sel.start(&w, (size_t)g->lm(lc));
                st = S_SELECT; pres = nullptr;
                break;
            case S_SELECT:                                                           // mod.rs:419-425
                { const float *pp = pres; pres = nullptr; if (sel.step(pp, *this)) return true; }
                nb[lc] = sel.R;
                if (has_preset) {
                    lc--;
                    if (lc < 0) { st = S_DONE; break; }
                    w = preset[lc]; sel.start(&w, (size_t)g->lm(lc)); pres = nullptr;
                    break;
                }
                ep = w;
                lc--;
                st = S_SEARCH;
                if (lc >= 0) sc.start(g, ep, (size_t)efc, lc, false, nullptr, nullptr, true);
                break;
            case S_DONE:
                return false;
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// update_neighbor_connections (graph/mod.rs:442-489), regrouped: one task per (neighbour, layer) list,
// applying that list's back-links in insertion order.  Lists are independent of each other.
// ------------------------------------------------------------------------------------------------
struct BackOp { uint32_t target; int layer; uint32_t new_id; float d; };

struct BacklinkTask : LsTask {
    Graph *g; uint32_t target; int layer; std::vector<BackOp> ops; size_t k = 0;
    std::vector<Cand> all; SelectTask sel; bool selecting = false;
    bool advance(const float *, const float *pres) override
    {
        clear_req();
        const size_t lm = (size_t)g->lm(layer);
        for (;;) {
            if (selecting) {
                { const float *pp = pres; pres = nullptr; if (sel.step(pp, *this)) return true; }
                Cand *lst = g->list(target, layer);
                for (size_t i = 0; i < sel.R.size(); i++) lst[i] = sel.R[i];         // mod.rs:484-485
                g->cnt(target, layer) = (uint16_t)sel.R.size();
                selecting = false; k++;
            }
            if (k >= ops.size()) return false;
            const BackOp &op = ops[k];
            Cand *lst = g->list(target, layer); uint16_t &c = g->cnt(target, layer);
            if (c < lm) { lst[c++] = Cand{op.d, op.new_id}; k++; continue; }         // mod.rs:469-471
            all.assign(lst, lst + c); all.push_back(Cand{op.d, op.new_id});          // mod.rs:474-482
            stable_sort_asc(all);
            sel.start(&all, lm);
            selecting = true;
        }
    }
};

// ------------------------------------------------------------------------------------------------
// get_scan_items + amgettuple (scan.rs:458-530, 709-876) for one query slot
// ------------------------------------------------------------------------------------------------
struct QueryTask : LsTask {
    const Graph *g; uint32_t slot; size_t ef_search; int mode; int64_t max_scan_tuples; uint32_t limit;
    const uint8_t *filter = nullptr; uint64_t n_filter = 0;
    enum { Q_INIT, Q_ENTRY, Q_GREEDY, Q_GROUND, Q_EMIT, Q_RESUME, Q_DONE } st = Q_INIT;
    int lc = 0;
    std::vector<Cand> ep, results;        // results: nearest LAST
    SearchCore sc; VisitedSet visited; RHeap<true> discarded;
    int64_t tuples = 0; double previous_distance = -HUGE_VAL;
    // output
    std::vector<int64_t> out_tid; std::vector<float> out_d; std::vector<uint32_t> out_elem;

    bool post_search() { dist_ids = sc.pend; q_sel = HX_QUERY_SLOT | slot; n_dist += dist_ids.size(); return true; }
    bool pass(int64_t tid) const { return !filter || (tid >= 0 && (uint64_t)tid < n_filter && filter[tid]); }
    bool advance(const float *dres, const float *) override
    {
        clear_req();
        const bool iterative = mode != 0;
        for (;;) {
            switch (st) {
            case Q_INIT:
                if (g->entry < 0) { st = Q_DONE; break; }                            // scan.rs:469-472
                sc.deleted = g->deleted.data();                                      // load_element -> None for a deleted tuple (scan.rs:178-181): visited, not a candidate
                dist_ids.push_back((uint32_t)g->entry); q_sel = HX_QUERY_SLOT | slot; n_dist += 1; st = Q_ENTRY;
                return true;
            case Q_ENTRY:
                ep.assign(1, Cand{dres[0], (uint32_t)g->entry}); dres = nullptr;
                lc = g->level[g->entry]; st = Q_GREEDY;
                if (lc >= 1) sc.start(g, ep, 1, lc, true, nullptr, nullptr, true);
                break;
            case Q_GREEDY:                                                           // scan.rs:491-512
                if (lc < 1) {
                    if (iterative) visited.reset(ef_search * (size_t)g->m * 2);
                    sc.start(g, ep, ef_search, 0, true, iterative ? &visited : nullptr, iterative ? &discarded : nullptr, true);
                    st = Q_GROUND; break;
                }
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_desc(results);
                if (results.empty()) { st = Q_DONE; break; }
                ep.assign(1, results.back());
                lc--;
                if (lc >= 1) sc.start(g, ep, 1, lc, true, nullptr, nullptr, true);
                break;
            case Q_GROUND:                                                           // scan.rs:515-528
            case Q_RESUME:
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_desc(results);
                st = Q_EMIT;
                break;
            case Q_EMIT:                                                             // scan.rs:794-875
                while (!results.empty() && out_tid.size() < limit) {
                    const Cand scd = results.back(); results.pop_back();
                    const uint8_t nt = g->ntids[scd.id];
                    if (nt == 0) continue;                                           // scan.rs:866-868
                    tuples++;
                    for (int t = (int)nt - 1; t >= 0 && out_tid.size() < limit; t--) {   // heaptids.pop()
                        if (mode == 2) { if ((double)scd.d < previous_distance) continue; previous_distance = (double)scd.d; }
                        const int64_t tid = g->tids[scd.id][t];
                        if (!pass(tid)) continue;
                        out_tid.push_back(tid); out_d.push_back(scd.d); out_elem.push_back(scd.id);
                    }
                }
                if (out_tid.size() >= limit || !iterative) { st = Q_DONE; break; }
                // results exhausted, iterative scan: scan.rs:817-858
                if (tuples >= max_scan_tuples) {
                    Cand one;
                    if (!discarded.pop(one)) { st = Q_DONE; break; }
                    results.assign(1, one);
                    break;
                }
                if (discarded.empty()) { st = Q_DONE; break; }                       // resume_scan_items, scan.rs:548-550
                ep.clear();
                { Cand x; while (ep.size() < ef_search && discarded.pop(x)) ep.push_back(x); }
                sc.start(g, ep, ef_search, 0, true, &visited, &discarded, false);
                st = Q_RESUME;
                break;
            case Q_DONE:
                return false;
            }
        }
    }
};


// ------------------------------------------------------------------------------------------------
// f3: the on-disk paths.  find_element_neighbors_on_disk (src/index/insert.rs:1021-1123): Algorithm 1 over search_layer_disk
// (f64 comparisons, nearest-last results), the `lm` NEAREST of each layer taken WITHOUT the heuristic (:1111-1117); with a skip
// set (vacuum's repair, src/index/vacuum.rs:288-407) ef + 1 and skip members neither counted nor selected.
// ------------------------------------------------------------------------------------------------
struct DiskNeighborsTask : LsTask {
    const Graph *g; uint32_t query_sel = 0; int new_level = 0, entry_level = 0; uint32_t entry = 0; int efc = 64;
    const uint8_t *skip = nullptr; uint32_t skip_self = 0xFFFFFFFFu; bool repair = false;
    enum { S_INIT, S_ENTRY, S_GREEDY, S_SEARCH, S_DONE } st = S_INIT;
    int lc = 0;
    std::vector<Cand> ep, w;
    SearchCore sc;
    std::vector<std::vector<Cand>> nb;     // per layer, nearest first
    void reset() { st = S_INIT; lc = 0; n_dist = n_pair = 0; clear_req(); ep.clear(); w.clear(); }
    void begin_search(size_t ef, int layer)
    {
        sc.skip = skip; sc.skip_self = skip_self; sc.deleted = g->deleted.data();
        sc.start(g, ep, ef, layer, true, nullptr, nullptr, true);
    }
    bool post_search() { dist_ids = sc.pend; q_sel = query_sel; n_dist += dist_ids.size(); return true; }
    bool advance(const float *dres, const float *) override
    {
        clear_req();
        for (;;) {
            switch (st) {
            case S_INIT:
                nb.assign(new_level + 1, {});
                if (g->deleted[entry]) { st = S_DONE; break; }                       // load_element(entry) -> None, insert.rs:1037-1048
                dist_ids.push_back(entry); q_sel = query_sel; n_dist += 1; st = S_ENTRY;
                return true;
            case S_ENTRY:
                ep.assign(1, Cand{dres[0], entry}); dres = nullptr;
                lc = entry_level; st = S_GREEDY;
                if (lc >= new_level + 1) begin_search(1, lc);
                break;
            case S_GREEDY:                                                           // insert.rs:1053-1074
                if (lc < new_level + 1) {
                    lc = std::min(new_level, entry_level); st = S_SEARCH;
                    if (lc >= 0) begin_search((size_t)efc + (repair ? 1 : 0), lc);   // insert.rs:1081-1086
                    break;
                }
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_desc(w);
                if (w.empty()) { st = S_DONE; break; }
                ep.assign(1, w.back());                                              // w.into_iter().last(): the nearest
                lc--;
                if (lc >= new_level + 1) begin_search(1, lc);
                break;
            case S_SEARCH:                                                           // insert.rs:1077-1120
                if (lc < 0) { st = S_DONE; break; }
                if (sc.run(dres)) return post_search();
                dres = nullptr;
                sc.results_desc(w);
                {
                    const size_t lm = (size_t)g->lm(lc);
                    std::vector<Cand> &out = nb[lc];
                    for (size_t i = w.size(); i-- > 0 && out.size() < lm;) {         // filtered.iter().rev().take(lm)
                        if (w[i].id == skip_self || (skip && skip[w[i].id])) continue;
                        out.push_back(w[i]);
                    }
                }
                ep = w;                                                              // ep_list = w
                lc--;
                if (lc >= 0) begin_search((size_t)efc + (repair ? 1 : 0), lc);
                break;
            case S_DONE:
                return false;
            }
        }
    }
};

// get_update_index (insert.rs:500-739) for one (neighbour element, layer): where the new element goes in that list, if anywhere.
// One request: d(neighbour, each connected element) as a distance group and the pair block among the connected elements.
struct UpdateIndexTask : LsTask {
    const Graph *g; uint32_t nbr = 0; int layer = 0; float new_d = 0.0f;
    int result = -3;                       // -3 None, -2 free slot, >= 0 slot to overwrite
    std::vector<uint32_t> ids;
    int stage = 0;
    void reset() { stage = 0; result = -3; n_dist = n_pair = 0; clear_req(); ids.clear(); }
    bool advance(const float *dres, const float *pres) override
    {
        clear_req();
        const size_t lm = (size_t)g->lm(layer);
        if (stage == 0) {
            stage = 1;
            if (g->deleted[nbr]) { result = -3; return false; }                     // insert.rs:524-527
            const Cand *lst = g->list(nbr, layer); const size_t cnt = g->cnt(nbr, layer);
            if (cnt < lm) { result = -2; return false; }                            // insert.rs:556-559
            for (size_t i = 0; i < cnt; i++) {                                      // insert.rs:566-625: an element being deleted gives up its slot
                const uint32_t c = lst[i].id;
                if (g->deleted[c] || g->ntids[c] == 0) { result = (int)i; return false; }
            }
            ids.resize(cnt);
            for (size_t i = 0; i < cnt; i++) ids[i] = lst[i].id;
            dist_ids = ids; q_sel = nbr; n_dist += cnt;
            if (cnt >= 2) { pgroups.push_back(PairGroup{(uint16_t)cnt, 0}); pair_ids = ids; n_pair += cnt * (cnt - 1) / 2; }
            return true;
        }
        // candidates sorted by distance (stable), then the new element added and sorted again (stable): insert.rs:630-665
        const size_t cnt = ids.size();
        struct HC { float d; int slot; };                                           // slot -1 = the new element
        std::vector<HC> all(cnt);
        for (size_t i = 0; i < cnt; i++) all[i] = HC{dres[i], (int)i};
        std::stable_sort(all.begin(), all.end(), [](const HC &a, const HC &b) { return a.d < b.d; });
        all.push_back(HC{new_d, -1});
        std::stable_sort(all.begin(), all.end(), [](const HC &a, const HC &b) { return a.d < b.d; });
        auto pair_d = [&](int a, int b) { const int hi = a > b ? a : b, lo = a > b ? b : a; return pres[(size_t)hi * (hi - 1) / 2 + lo]; };
        std::vector<int> sel, pruned;                                               // indices into `all`
        for (size_t h = 0; h < all.size(); h++) {                                   // insert.rs:673-704
            if (sel.size() >= lm) break;
            bool closer = true;
            for (int k : sel) {
                if (all[h].slot >= 0 && all[k].slot >= 0 && pair_d(all[h].slot, all[k].slot) <= all[h].d) { closer = false; break; }   // pairs with the new element are skipped (:680-693)
            }
            if (closer) sel.push_back((int)h); else pruned.push_back((int)h);
        }
        for (int k : pruned) { if (sel.size() >= lm) break; sel.push_back(k); }    // insert.rs:707-712
        bool new_selected = false; std::vector<uint8_t> kept(cnt, 0);
        for (int k : sel) { if (all[k].slot < 0) new_selected = true; else kept[all[k].slot] = 1; }
        result = -3;
        if (new_selected) for (size_t i = 0; i < cnt; i++) if (!kept[i]) { result = (int)i; break; }   // insert.rs:722-737
        return false;
    }
};

// ------------------------------------------------------------------------------------------------
// tiny persistent thread pool: parallel_for over task indices
// ------------------------------------------------------------------------------------------------
struct Pool {
    // Persistent workers.  While a lock-step run is in progress (`hot`), workers spin on `epoch` instead of sleeping
    // on a condition variable: a run issues thousands of short parallel phases back to back and a futex wake-up
    // (~20-50 us) per phase would dominate them.
    std::vector<std::thread> th; std::mutex mu; std::condition_variable cv;
    std::function<void(size_t)> fn; size_t n = 0; std::atomic<size_t> next{0};
    std::atomic<uint64_t> epoch{0}; std::atomic<int> done{0}; std::atomic<bool> stop{false}, hot{false};
    explicit Pool(int nt) { for (int i = 0; i < nt; i++) th.emplace_back([this] { loop(); }); }
    ~Pool() { stop = true; { std::lock_guard<std::mutex> l(mu); } cv.notify_all(); for (auto &t : th) t.join(); }
    void work() { for (;;) { size_t i = next.fetch_add(1, std::memory_order_relaxed); if (i >= n) break; fn(i); } }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while (epoch.load(std::memory_order_acquire) == seen && !stop.load(std::memory_order_relaxed)) {
                // brief spin only: the box gives this process a CPU *quota*, so cycles burnt spinning are cycles the
                // task logic does not get
                if (hot.load(std::memory_order_relaxed) && spins < 200) { spins++; __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> l(mu);
                cv.wait_for(l, std::chrono::milliseconds(200), [&] { return stop.load() || epoch.load() != seen; });
            }
            if (stop.load()) return;
            seen = epoch.load(std::memory_order_acquire);
            work();
            done.fetch_add(1, std::memory_order_release);
        }
    }
    void set_hot(bool h) { hot = h; if (h) { std::lock_guard<std::mutex> l(mu); } cv.notify_all(); }
    // runs f(0..count-1) on the workers + the calling thread; items are handed out one at a time (callers pass chunks)
    void parallel_for(size_t count, const std::function<void(size_t)> &f)
    {
        if (count == 0) return;
        if (th.empty() || count == 1) { for (size_t i = 0; i < count; i++) f(i); return; }
        fn = f; n = count; next.store(0); done.store(0);
        { std::lock_guard<std::mutex> l(mu); epoch.fetch_add(1, std::memory_order_release); }
        cv.notify_all();
        work();
        for (int spins = 0; done.load(std::memory_order_acquire) != (int)th.size(); spins++) { if (spins < 2000) __builtin_ia32_pause(); else std::this_thread::yield(); }
    }
};

} // namespace

// ================================================================================================
struct BatchState {
    bool open = false, linked = false;
    bool lazy_lists = false;     // single-process insert: lists pruned on the device are not copied back per batch (hx_index::ensure_host_lists)
    bool merged = false;         // merge_duplicates ran (rollback undoes its TID merges)
    bool dev = false;            // device-resident batch (hx_index_dbatch_*): the members' lists never visit the host
    uint32_t base = 0, b = 0, entry = 0; int entry_level = 0;
    std::vector<int64_t> tids; std::vector<uint32_t> elem; std::vector<uint8_t> searched;
    std::vector<BackOp> ops; std::vector<std::pair<size_t, size_t>> grp;
};

struct hx_index {
    hx_engine *e = nullptr;
    BatchState bs;
    Graph g; int efc = 64;
    std::unique_ptr<Pool> pool; int n_threads = 0;
    uint64_t counters[8] = {0};
    // true while some live element's list may still name a deleted element (vacuum.rs:300-303 leaves an un-repairable entry point as it is): searches then
    // meet load_element -> None (scan.rs:178-181) and stay on the lock-step driver.  hx_index_vacuum verifies and clears it; a page image never loads
    // deleted tuples and hx_index_invalidate unlinks what it drops, so nothing else can set it.
    bool dead_refs = false;
    std::vector<std::unique_ptr<InsertTask>> insert_pool;      // task objects are reused across batches (their heaps,
    std::vector<std::unique_ptr<BacklinkTask>> backlink_pool;  // visited tables and request vectors keep their capacity)
    std::vector<std::unique_ptr<QueryTask>> query_pool;
    std::vector<std::unique_ptr<DiskNeighborsTask>> disk_pool;
    std::vector<std::unique_ptr<UpdateIndexTask>> update_pool;
    bool fused = true;                                         // device-resident traversal (hx_fused.inc.h) for searches
    bool mfma = false; uint64_t mfma_pairs = 0, mfma_exact = 0; // lock-step select blocks on the matrix cores (halfvec inner product)
    bool mfma_on() const { return mfma && e->dtype == HX_F16 && e->metric == HX_NEG_IP; }
    void arm_select(SelectTask &sel) { sel.mfma = mfma_on(); sel.norm2 = e->h_mf_norm2.data(); sel.band_k = 2.0f * (float)e->dim * 5.9604645e-08f * 1.001f; sel.n_mfma = sel.n_exact = 0; }
    // index loaded from its page image (hx_index_load_pages): where each element tuple sits and the version it carried (types/hnsw.rs:120)
    std::vector<uint32_t> loc_blk; std::vector<uint16_t> loc_off; std::vector<uint8_t> loc_ver; std::unordered_map<uint64_t, uint32_t> loc2elem;
    std::vector<std::pair<uint32_t, int>> dirty;               // (element, layer) lists the device mirror has not seen yet
    uint32_t mirror_elems = 0;
    uint64_t fused_tasks = 0, fused_redo = 0;
    struct ScanSlot { bool busy = false; uint32_t first = 0, nq = 0, ef = 0, k = 0; } scan_slot[HX_SCAN_SLOTS];   // pipelined scans in flight (hx_index_search_submit)
    uint32_t scans_in_flight = 0;
    double prof[16] = {0};   // seconds: [0] advance, [1] compact, [2] fill, [3] dist launch+wait, [4] pair launch+wait, [5] rounds
    std::string err;
    int fail(int code, const std::string &m) { err = m; return code; }

    bool fused_ok() const { return fused && 2 * g.m <= 64 && e->pitch <= 8192 && e->dtype != HX_SPARSE; }   // the insert kernel / batch pipeline; sparsevec and m > 32: MODE 3 + hx_biglist.hip
    // scans: the traversal kernel walks lists longer than a wavefront 64 ids at a time, so every m the reference allows (options.rs:203-225: m <= 100) is served;
    // only the insert-mode kernel and the back-link kernels are built for lists of <= 64
    bool fused_scan_ok() const { return fused && e->pitch <= 8192; }   // round 3: sparsevec too (hx_fused_sparse.hip: one lane per row walks the merge join)
    // scans: the kernels do not model load_element -> None for a deleted element (scan.rs:178-181: skipped, not counted), so while one can be met
    // (dead_refs: only the un-repairable entry point of vacuum.rs:300-303) scans stay on the lock-step driver, which does
    bool device_scan_ok() const { return fused_scan_ok() && !dead_refs; }
    // device-resident batches (hx_batch.hip): the traversal kernel and the back-link kernels both serve this m
    bool dbatch_ok() const { static const bool off = getenv("HX_DEVICE_BATCH") && atoi(getenv("HX_DEVICE_BATCH")) == 0; return !off && fused_ok() && 2 * g.m <= 64; }
    void mark_dirty(uint32_t elem) { for (int lc = 0; lc <= g.level[elem]; lc++) dirty.emplace_back(elem, lc); }
    // brings the device copy of the graph up to date: levels of new elements + every list written since the last sync
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    struct Timer { double &acc; double t0; explicit Timer(double &a) : acc(a), t0(now_s()) {} ~Timer() { acc += now_s() - t0; } };
    int sync_mirror()
    {
        Timer tm(prof[7]);
        const uint32_t m = (uint32_t)g.m;
        int rc = e->mirror_reserve(m, g.size(), g.up.size() / m + 1);
        if (rc) return fail(rc, e->err);
        // a list recorded twice is scattered twice with the same (current) content: harmless, and cheaper than sorting
        const uint32_t n_new = g.size() - mirror_elems, n_rec = (uint32_t)dirty.size();
        if (n_new == 0 && n_rec == 0) return HX_OK;
        std::vector<int32_t> lv(n_new); std::vector<uint32_t> blk(n_new);
        for (uint32_t i = 0; i < n_new; i++) { lv[i] = g.level[mirror_elems + i]; blk[i] = (uint32_t)(g.up_off[mirror_elems + i] / m); }
        std::vector<uint32_t> hdr((size_t)n_rec * 4), ids((size_t)n_rec * 2 * m, 0u); std::vector<float> dd((size_t)n_rec * 2 * m, 0.0f);
        pool->parallel_for((n_rec + 1023) / 1024, [&](size_t ci) {
            for (size_t r = ci * 1024; r < std::min<size_t>(n_rec, ci * 1024 + 1024); r++) {
                const uint32_t el = dirty[r].first; const int layer = dirty[r].second;
                const bool live = g.level[el] >= layer;
                const uint16_t c = live ? g.cnt(el, layer) : 0; const Cand *l = live ? g.list(el, layer) : nullptr;
                hdr[4 * r] = el; hdr[4 * r + 1] = (uint32_t)layer; hdr[4 * r + 2] = c;
                hdr[4 * r + 3] = layer > 0 ? (uint32_t)(g.up_off[el] / m) + (uint32_t)(layer - 1) : 0u;
                for (uint16_t k = 0; k < c; k++) { ids[r * 2 * m + k] = l[k].id; dd[r * 2 * m + k] = l[k].d; }
            }
        });
        if ((rc = e->mirror_update(mirror_elems, n_new, lv.data(), blk.data(), n_rec, hdr.data(), ids.data(), dd.data()))) return fail(rc, e->err);
        mirror_elems = g.size(); dirty.clear();
        return HX_OK;
    }

    // Runs tasks to completion in lock-step: every round, every live task's request goes into ONE distance launch
    // and ONE pair launch.  At most `window` tasks are live at a time; finished ones are replaced from the rest of
    // `tasks` (continuous admission keeps the launches full until the tail).
    struct ChunkSum { size_t alive = 0, dgroups = 0, dids = 0, pgroups = 0, pids = 0, pout = 0; };
    // Lists pruned by k_links live in the device mirror first; the host copy is refreshed in one sweep when something on
    // the host needs to read lists (lock-step tasks, export, serialisation).  sync_mirror() goes first so that lists the
    // host wrote since the last launch are in the mirror too, which makes the sweep a plain overwrite.
    bool host_stale = false, in_insert = false;
    // scratch of hx_index_batch_links, kept between batches (a batch allocates and page-faults ~10 MB otherwise)
    struct LinkScratch { std::vector<BackOp> raw; std::vector<uint32_t> hist, own, tg, ly, off, onew, opstart, da, db, dstart; std::vector<float> od; std::vector<uint8_t> deq;
                         std::vector<unsigned long long> keys;
                         std::vector<uint32_t> za, zb, ha, hb, cls; std::vector<int32_t> cur; std::vector<uint8_t> dupflag;
                         std::vector<std::vector<std::pair<size_t, size_t>>> bgrp; } ls;
    // build.rs:482-525 for the open batch, in row order: a member merges into the first byte-identical zero-distance layer-0 neighbour that
    // has room for another heap TID ((za, zb): confirmed (member, neighbour) pairs in list order), else into the earliest identical member of
    // this batch that is still an element of its own and has room ((ha, hb): confirmed (later, earlier) pairs, ascending); otherwise it stays
    // an element and may become the entry point.  A merged member is tombstoned (level -1 - level); ls.dupflag marks them.  Returns their number.
    uint32_t merge_duplicates(const std::vector<uint32_t> &za, const std::vector<uint32_t> &zb, const std::vector<uint32_t> &ha, const std::vector<uint32_t> &hb)
    {
        const uint32_t b = bs.b, base = bs.base;
        auto &cls = ls.cls; auto &cur = ls.cur; auto &dupflag = ls.dupflag;
        dupflag.assign(b, 0);
        const bool classes = !ha.empty();
        if (classes) {
            cls.resize(b); cur.assign(b, -1);
            for (uint32_t i = 0; i < b; i++) cls[i] = i;
            for (size_t k = 0; k < ha.size(); k++) cls[ha[k] - base] = cls[hb[k] - base];      // ascending in ha: the earlier member's class is final
        }
        size_t zp = 0; uint32_t n_dup = 0;
        bs.merged = true;
        for (uint32_t i = 0; i < b; i++) {
            const uint32_t id = base + i;
            int64_t dup = -1;
            while (zp < za.size() && za[zp] < id) zp++;
            for (size_t k = zp; k < za.size() && za[k] == id; k++)
                if (g.ntids[zb[k]] < HEAPTIDS && g.level[zb[k]] >= 0) { dup = zb[k]; break; }
            if (dup < 0 && classes) { const int32_t c = cur[cls[i]]; if (c >= 0 && g.ntids[base + (uint32_t)c] < HEAPTIDS) dup = base + (uint32_t)c; }
            if (dup >= 0) {                                     // merge into the existing element; this row becomes a tombstone
                g.tids[dup][g.ntids[dup]++] = bs.tids[i];
                g.level[id] = -1 - g.level[id];
                bs.elem[i] = (uint32_t)dup; dupflag[i] = 1; n_dup++;
            } else {
                if (g.level[id] > g.level[g.entry]) g.entry = id;   // build.rs:523-525
                g.tids[id][0] = bs.tids[i]; g.ntids[id] = 1;
                bs.elem[i] = id;
                if (classes) cur[cls[i]] = (int32_t)i;
            }
        }
        return n_dup;
    }
    // a stage of the open batch failed: the elements batch_begin added hold no lists or TIDs and must not stay in the graph, or the rows
    // could never be inserted again (first_row != index size); dropped here, together with the entry point / TIDs the batch had not touched yet
    void rollback_batch()
    {
        if (bs.open) {
            const uint32_t keep = bs.base;
            if (!bs.linked && bs.merged) {                       // TIDs merged into older elements, entry point moved to a member
                for (uint32_t i = bs.b; i-- > 0;) if (ls.dupflag[i] && bs.elem[i] < keep && g.ntids[bs.elem[i]] > 0) g.ntids[bs.elem[i]]--;
                g.entry = bs.entry;
            }
            if (!bs.linked && g.size() > keep) {
                g.level.resize(keep); g.n0_cnt.resize(keep); g.n0.resize((size_t)keep * 2 * g.m);
                g.up.resize(g.up_off[keep]); g.up_cnt.resize(g.upc_off[keep]); g.up_off.resize(keep); g.upc_off.resize(keep);
                g.tids.resize(keep); g.ntids.resize(keep); g.deleted.resize(keep);
                if (mirror_elems > keep) mirror_elems = keep;
                size_t w = 0; for (size_t k = 0; k < dirty.size(); k++) if (dirty[k].first < keep) dirty[w++] = dirty[k]; dirty.resize(w);
            }
        }
        bs = BatchState();
    }
    int ensure_host_lists()
    {
        if (!host_stale) return HX_OK;
        int rc = sync_mirror();
        if (rc) return rc;
        const uint32_t n = mirror_elems, m = (uint32_t)g.m; const size_t lm0 = 2u * (size_t)m, nb = g.up_cnt.size();
        std::vector<uint32_t> ids((size_t)n * lm0), uids(nb * m); std::vector<float> dd((size_t)n * lm0), ud(nb * m);
        std::vector<uint16_t> ucnt(nb);
        if ((rc = e->mirror_download(n, nb, ids.data(), dd.data(), g.n0_cnt.data(), uids.data(), ud.data(), ucnt.data()))) return fail(rc, e->err);
        pool->parallel_for((n + 8191) / 8192, [&](size_t ci) {
            for (size_t el = ci * 8192; el < std::min<size_t>(n, ci * 8192 + 8192); el++)
                for (size_t k = 0; k < lm0; k++) g.n0[el * lm0 + k] = Cand{dd[el * lm0 + k], ids[el * lm0 + k]};
        });
        for (size_t b = 0; b < nb; b++) { g.up_cnt[b] = ucnt[b]; for (uint32_t k = 0; k < m; k++) g.up[b * m + k] = Cand{ud[b * m + k], uids[b * m + k]}; }
        host_stale = false;
        return HX_OK;
    }
    int run_lockstep(std::vector<LsTask *> &tasks, size_t window = 0)
    {
        if (tasks.empty()) return HX_OK;
        { int rc0 = ensure_host_lists(); if (rc0) return rc0; }
        if (mfma_on()) { int rc0 = e->mfma_norms(e->n_rows); if (rc0) return fail(rc0, e->err); }
        if (window == 0 || window > tasks.size()) window = tasks.size();
        std::vector<LsTask *> live(tasks.begin(), tasks.begin() + window), live2;
        size_t admitted = window;
        for (LsTask *t : live) t->fresh = true;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        pool->set_hot(true);
        int rc = HX_OK;
        std::vector<ChunkSum> sums, base;
        std::vector<uint8_t> alive;
        while (!live.empty()) {
            const float *dbase = e->ch.h_out, *pbase = e->ch.h_pout;
            const size_t nlive = live.size();
            const size_t csize = std::max<size_t>(16, nlive / ((size_t)n_threads * 6) + 1), nchunks = (nlive + csize - 1) / csize;
            sums.assign(nchunks, ChunkSum()); alive.assign(nlive, 0);
            double t0 = now();
            // phase A: state machines consume the previous round's results and post their next request
            pool->parallel_for(nchunks, [&](size_t ci) {
                ChunkSum cs;
                const size_t lo = ci * csize, hi = std::min(nlive, lo + csize);
                for (size_t i = lo; i < hi; i++) {
                    LsTask *t = live[i];
                    const bool fr = t->fresh; t->fresh = false;
                    if (!t->advance(fr ? nullptr : dbase + t->dist_off, fr ? nullptr : pbase + t->pair_out_off)) continue;
                    alive[i] = 1; cs.alive++;
                    if (!t->dist_ids.empty()) { cs.dgroups++; cs.dids += t->dist_ids.size(); }
                    for (const PairGroup &pg : t->pgroups) {
                        cs.pgroups++; cs.pids += (size_t)pg.na + pg.nb;
                        cs.pout += pg.nb ? (size_t)pg.na * pg.nb : (size_t)pg.na * (pg.na - 1) / 2;
                    }
                }
                sums[ci] = cs;
            });
            double t1 = now(); prof[0] += t1 - t0;
            base.assign(nchunks + 1, ChunkSum());
            for (size_t c = 0; c < nchunks; c++) {
                base[c + 1].alive = base[c].alive + sums[c].alive; base[c + 1].dgroups = base[c].dgroups + sums[c].dgroups;
                base[c + 1].dids = base[c].dids + sums[c].dids; base[c + 1].pgroups = base[c].pgroups + sums[c].pgroups;
                base[c + 1].pids = base[c].pids + sums[c].pids; base[c + 1].pout = base[c].pout + sums[c].pout;
            }
            const ChunkSum &tot = base[nchunks];
            // admit fresh tasks into the freed slots (they post their first request next round)
            size_t n_new = std::min(tasks.size() - admitted, window - tot.alive);
            live2.resize(tot.alive + n_new);
            if (tot.alive == 0 && n_new == 0) break;
            HxRound rd; rd.n_dgroups = (uint32_t)tot.dgroups; rd.n_dids = (uint32_t)tot.dids; rd.n_pgroups = (uint32_t)tot.pgroups;
            rd.n_pids = (uint32_t)tot.pids; rd.n_pout = tot.pout;
            if ((rc = e->layout_round(rd))) { rc = fail(rc, e->err); break; }
            double t2 = now(); prof[1] += t2 - t1;
            HxChannel &c = e->ch;
            // phase B: compaction + request fill, each chunk at its prefix offsets
            pool->parallel_for(nchunks, [&](size_t ci) {
                ChunkSum o = base[ci];
                const size_t lo = ci * csize, hi = std::min(nlive, lo + csize);
                for (size_t i = lo; i < hi; i++) {
                    if (!alive[i]) continue;
                    LsTask *t = live[i];
                    live2[o.alive++] = t;
                    t->dist_off = o.dids; t->pair_out_off = o.pout;
                    if (!t->dist_ids.empty()) {
                        c.h_grp_q[o.dgroups] = t->q_sel; c.h_grp_off[o.dgroups] = (uint32_t)o.dids; o.dgroups++;
                        memcpy(c.h_ids + o.dids, t->dist_ids.data(), t->dist_ids.size() * sizeof(uint32_t));
                        o.dids += t->dist_ids.size();
                    }
                    if (!t->pgroups.empty()) {
                        memcpy(c.h_pids + o.pids, t->pair_ids.data(), t->pair_ids.size() * sizeof(uint32_t));
                        for (const PairGroup &pg : t->pgroups) {
                            c.h_pg_off[o.pgroups] = (uint32_t)o.pids; c.h_pg_na[o.pgroups] = pg.na; c.h_pg_nb[o.pgroups] = pg.nb; c.h_pg_flag[o.pgroups] = pg.mfma;
                            c.h_pg_out_off[o.pgroups] = o.pout; o.pgroups++;
                            o.pids += (size_t)pg.na + pg.nb;
                            o.pout += pg.nb ? (size_t)pg.na * pg.nb : (size_t)pg.na * (pg.na - 1) / 2;
                        }
                    }
                }
            });
            c.h_grp_off[tot.dgroups] = (uint32_t)tot.dids;
            c.h_pg_off[tot.pgroups] = (uint32_t)tot.pids;
            for (size_t k = 0; k < n_new; k++) { LsTask *t = tasks[admitted++]; t->fresh = true; live2[tot.alive + k] = t; }
            live.swap(live2);
            double t3 = now(); prof[2] += t3 - t2;
            if ((rc = e->run_round())) { rc = fail(rc, e->err); break; }
            prof[3] += now() - t3; prof[5] += 1.0;
        }
        pool->set_hot(false);
        return rc;
    }
};

// types/hnsw.rs:337-349 with BLCKSZ 8192: (8192 - 24 - 8 - 4 - 4) / 6 / m - 2, capped at 255
static int max_level_for(int m) { int v = (8192 - 24 - 8 - 4 - 4) / 6 / m - 2; return v < 255 ? v : 255; }

extern "C" {

int hx_index_create(hx_engine *e, int m, int ef_construction, hx_index **out)
{
    if (!e || !out) return HX_E_ARG;
    *out = nullptr;
    // options.rs:203-225 ranges; build.rs:865-867 ef_construction >= 2m
    if (m < 2 || m > 100) return e->fail(HX_E_ARG, "m must be between 2 and 100");
    if (ef_construction < 4 || ef_construction > 1000) return e->fail(HX_E_ARG, "ef_construction must be between 4 and 1000");
    if (ef_construction < 2 * m) return e->fail(HX_E_ARG, "ef_construction must be greater than or equal to 2 * m");
    hx_index *ix = new (std::nothrow) hx_index();
    if (!ix) return e->fail(HX_E_NOMEM, "out of host memory");
    ix->e = e; ix->g.m = m; ix->efc = ef_construction;
    ix->mfma = e->dtype == HX_F16 && e->metric == HX_NEG_IP;      // halfvec inner product: select_neighbors' pair distances are a true f16 GEMM -> matrix cores (hx_index_set_mfma(0): VALU)
    int nt = (int)std::thread::hardware_concurrency(); if (nt <= 0) nt = 4; if (nt > 16) nt = 16;
    ix->n_threads = nt; ix->pool.reset(new Pool(nt - 1));
    *out = ix;
    return HX_OK;
}

int hx_index_destroy(hx_index *ix) { delete ix; return HX_OK; }
const char *hx_index_last_error(const hx_index *ix) { return ix ? ix->err.c_str() : ""; }

int hx_index_set_threads(hx_index *ix, int n_threads)
{
    if (!ix || n_threads < 1 || n_threads > 256) return HX_E_ARG;
    ix->n_threads = n_threads; ix->pool.reset(new Pool(n_threads - 1));
    return HX_OK;
}


// ---- staged form of one lock-step batch (single GPU: begin, search(0,b), links(0,1), end; several GPUs: every
// ---- rank holds a replica of rows and graph, searches its slice, prunes the lists it owns, and the serialized
// ---- lists are all-gathered between the stages -- pgvector-rx_amd/dist_build.py) --------------------------------
int hx_index_batch_begin(hx_index *ix, uint64_t first_row, uint32_t b, const int32_t *levels, const int64_t *tids)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix) return HX_E_ARG;
    hx_index::Timer t_bb(ix->prof[12]);
    if (b == 0 || !levels || !tids) return ix->fail(HX_E_ARG, "empty batch or NULL argument");
    Graph &g = ix->g; BatchState &bs = ix->bs;
    if (bs.open) return ix->fail(HX_E_STATE, "a batch is already open");
    if (g.entry < 0) return ix->fail(HX_E_STATE, "insert the first element with hx_index_insert before opening batches");
    if (first_row != g.size()) return ix->fail(HX_E_STATE, "rows must be inserted in append order: first_row != index size");
    if (first_row + b > hx_num_rows(ix->e)) return ix->fail(HX_E_ARG, "rows not present in the engine");
    const int mxl = max_level_for(g.m);
    if (!ix->in_insert) { int rc0 = ix->ensure_host_lists(); if (rc0) return rc0; }   // an external (multi-GPU) driver exports lists from the host copy
    {   // keep the vectors' capacity from batch to batch
        BatchState fresh;
        fresh.tids.swap(bs.tids); fresh.elem.swap(bs.elem); fresh.searched.swap(bs.searched); fresh.ops.swap(bs.ops); fresh.grp.swap(bs.grp);
        fresh.tids.clear(); fresh.elem.clear(); fresh.searched.clear(); fresh.ops.clear(); fresh.grp.clear();
        bs = std::move(fresh);
    }
    bs.open = true; bs.base = g.size(); bs.b = b; bs.entry = (uint32_t)g.entry; bs.entry_level = g.level[g.entry];
    bs.tids.assign(tids, tids + b); bs.elem.assign(b, 0); bs.searched.assign(b, 0);
    g.add_bulk(levels, b, mxl);
    return HX_OK;
}

// find_element_neighbors for batch members [lo, hi) against the graph as of batch_begin
int hx_index_batch_search(hx_index *ix, uint32_t lo, uint32_t hi)
{
    if (!ix) return HX_E_ARG;
    hx_index::Timer t_bs(ix->prof[11]);
    BatchState &bs = ix->bs; Graph &g = ix->g;
    if (!bs.open || lo > hi || hi > bs.b) return ix->fail(HX_E_STATE, "no open batch / bad member range");
    std::vector<uint32_t> todo;                                  // batch members still to be searched by the lock-step path
    if (ix->fused_ok() && hi > lo) {
        int rc = ix->sync_mirror();
        if (rc) return rc;
        const uint32_t n = hi - lo, lm0 = 2u * (uint32_t)g.m;
        std::vector<uint32_t> qsel(n); std::vector<int32_t> tl(n);
        for (uint32_t i = 0; i < n; i++) { qsel[i] = bs.base + lo + i; tl[i] = g.level[bs.base + lo + i]; }
        uint64_t cnts[2] = {0, 0};
        HxFusedView v;                                          // results are read in place from the pinned staging buffer
        auto t0 = std::chrono::steady_clock::now();
        if ((rc = ix->e->fused_run(1, n, qsel.data(), tl.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                                   nullptr, nullptr, nullptr, nullptr, cnts, nullptr, &v))) return ix->fail(rc, ix->e->err);
        ix->prof[6] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ix->counters[1] += cnts[0]; ix->counters[2] += cnts[1];
        const uint32_t *status = v.status, *ocnt = v.cnt, *oids = v.ids; const float *od = v.d;
        ix->pool->parallel_for((n + 511) / 512, [&](size_t ci) {
            for (uint32_t i = (uint32_t)ci * 512; i < std::min<uint32_t>(n, (uint32_t)ci * 512 + 512); i++) {
                const uint32_t id = bs.base + lo + i;
                if (status[i] != 0) continue;
                for (int lc = 0; lc <= g.level[id]; lc++) {
                    const uint32_t c = ocnt[(size_t)i * HX_FUSED_MAXL + lc]; Cand *lst = g.list(id, lc);
                    const size_t ob = ((size_t)i * HX_FUSED_MAXL + lc) * lm0;
                    for (uint32_t k = 0; k < c; k++) lst[k] = Cand{od[ob + k], oids[ob + k]};
                    g.cnt(id, lc) = (uint16_t)c;
                }
                bs.searched[lo + i] = 1;
            }
        });
        std::vector<uint32_t> again;                             // members whose tables overflowed: one more try on the device with roomier tables
        for (uint32_t i = 0; i < n; i++) { if (status[i] == 0) ix->mark_dirty(bs.base + lo + i); else if (status[i] == 1) again.push_back(lo + i); else todo.push_back(lo + i); }
        if (!again.empty()) {
            const uint32_t na = (uint32_t)again.size();
            std::vector<uint32_t> q2(na); std::vector<int32_t> l2(na);
            for (uint32_t k = 0; k < na; k++) { q2[k] = bs.base + again[k]; l2[k] = g.level[bs.base + again[k]]; }
            HxFusedView v2;
            if ((rc = ix->e->fused_run(1, na, q2.data(), l2.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                                       nullptr, nullptr, nullptr, nullptr, cnts, nullptr, &v2, 8))) return ix->fail(rc, ix->e->err);
            ix->counters[1] += cnts[0]; ix->counters[2] += cnts[1];
            for (uint32_t k = 0; k < na; k++) {
                const uint32_t id = bs.base + again[k];
                if (v2.status[k] != 0) { todo.push_back(again[k]); continue; }
                for (int lc = 0; lc <= g.level[id]; lc++) {
                    const uint32_t c = v2.cnt[(size_t)k * HX_FUSED_MAXL + lc]; Cand *lst = g.list(id, lc);
                    const size_t ob = ((size_t)k * HX_FUSED_MAXL + lc) * lm0;
                    for (uint32_t j = 0; j < c; j++) lst[j] = Cand{v2.d[ob + j], v2.ids[ob + j]};
                    g.cnt(id, lc) = (uint16_t)c;
                }
                ix->mark_dirty(id);
                bs.searched[again[k]] = 1;
            }
        }
        ix->fused_tasks += n; ix->fused_redo += todo.size();
        if (todo.empty()) return HX_OK;
    } else if (ix->fused && (2 * g.m > 64 || ix->e->dtype == HX_SPARSE) && ix->e->pitch <= 8192 && hi > lo) {
        // m > 32 (the reference allows m <= 100, options.rs:203-225) and sparsevec rows: the insert-mode select and the back-link kernels are built for lists of
        // <= 64 and for dense rows, but
        // the searches -- the bulk of the distance evaluations -- run in the traversal kernel all the same (MODE 3 walks lists longer than a wavefront
        // 64 ids at a time and hands out every layer's W); select_neighbors and update_neighbor_connections follow on the lock-step driver.
        int rc = ix->sync_mirror();
        if (rc) return rc;
        hx_engine *e = ix->e;
        const uint32_t n = hi - lo;
        std::vector<uint32_t> qsel(n), prob(n), tstat(n); std::vector<int32_t> tl(n);
        uint32_t P = 0;
        for (uint32_t i = 0; i < n; i++) { qsel[i] = bs.base + lo + i; tl[i] = g.level[bs.base + lo + i]; prob[i] = P; P += (uint32_t)std::min(tl[i], bs.entry_level) + 1u; }
        if ((rc = e->wsel_reserve(P, (uint32_t)ix->efc)) || (rc = e->db_reserve_records(n))) return ix->fail(rc, e->err);
        HxWselWork &w = e->wsel;
        if (hipMemsetAsync(w.d_cnt, 0, (size_t)P * 4, e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "clearing the W counts failed");
        HxFusedDev dev; dev.d_rec = e->bw.d_rec; dev.rec_words = hx_rec_words((uint32_t)g.m); dev.h_slots = nullptr;
        dev.d_wl_out = w.d_wl; dev.d_wl_cnt = w.d_cnt; dev.h_prob = prob.data();
        uint64_t cnts[2] = {0, 0};
        const double t0 = hx_index::now_s();
        if ((rc = e->fused_run(3, n, qsel.data(), tl.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                               nullptr, nullptr, nullptr, tstat.data(), cnts, nullptr, nullptr, 1, &dev))) return ix->fail(rc, e->err);
        ix->prof[6] += hx_index::now_s() - t0;
        ix->counters[1] += cnts[0];
        static const bool biglist_off = getenv("HX_BIGLIST") && atoi(getenv("HX_BIGLIST")) == 0;     // experiments: select / back-links of these shapes on the lock-step driver
        if (!biglist_off) {
            // select_neighbors per (member, layer) in k_select_w (hx_biglist.hip), straight from the result sets the traversal kernel left on the device
            const uint32_t lm0 = 2u * (uint32_t)g.m;
            std::vector<uint32_t> lmv(P, lm0);
            for (uint32_t i = 0; i < n; i++) for (int lc = 0; lc <= std::min(tl[i], bs.entry_level); lc++) lmv[prob[i] + (uint32_t)lc] = (uint32_t)g.lm(lc);
            const uint32_t *oids = nullptr, *ocnt = nullptr; const float *od = nullptr; uint64_t np = 0;
            if ((rc = e->biglist_select(P, (uint32_t)ix->efc, lmv.data(), lm0, &oids, &od, &ocnt, &np))) return ix->fail(rc, e->err);
            ix->counters[2] += np;
            ix->pool->parallel_for((n + 255) / 256, [&](size_t ci) {
                for (uint32_t i = (uint32_t)ci * 256; i < std::min<uint32_t>(n, (uint32_t)ci * 256 + 256); i++) {
                    if (tstat[i] != 0) continue;
                    const uint32_t id = bs.base + lo + i; const int start = std::min(tl[i], bs.entry_level);
                    for (int lc = 0; lc <= tl[i]; lc++) {
                        Cand *lst = g.list(id, lc); uint32_t c = 0;
                        if (lc <= start) {
                            const size_t pr = (size_t)prob[i] + (size_t)lc; c = std::min<uint32_t>(ocnt[pr], (uint32_t)g.lm(lc));
                            for (uint32_t k = 0; k < c; k++) lst[k] = Cand{od[pr * lm0 + k], oids[pr * lm0 + k]};
                        }
                        g.cnt(id, lc) = (uint16_t)c;
                    }
                    bs.searched[lo + i] = 1;
                }
            });
            for (uint32_t i = 0; i < n; i++) { if (tstat[i] != 0) todo.push_back(lo + i); else ix->mark_dirty(bs.base + lo + i); }
            ix->fused_tasks += n; ix->fused_redo += todo.size();
            if (todo.empty()) return HX_OK;
        } else {
        std::vector<uint32_t> wcnt(P); std::vector<uint2> wl((size_t)P * ix->efc);
        if (hipMemcpyAsync(wcnt.data(), w.d_cnt, (size_t)P * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipMemcpyAsync(wl.data(), w.d_wl, (size_t)P * ix->efc * 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "reading the W lists failed");
        std::vector<std::unique_ptr<InsertTask>> &its = ix->insert_pool;
        while (its.size() < n) its.emplace_back(new InsertTask());
        std::vector<LsTask *> tasks; std::vector<uint32_t> members;
        for (uint32_t i = 0; i < n; i++) {
            if (tstat[i] != 0) { todo.push_back(lo + i); continue; }
            InsertTask &t = *its[tasks.size()];
            t.st = InsertTask::S_INIT; t.lc = 0; t.n_dist = t.n_pair = 0; t.clear_req();
            t.g = &g; t.id = bs.base + lo + i; t.new_level = tl[i]; t.entry = bs.entry; t.entry_level = bs.entry_level; t.efc = ix->efc;
            const int start = std::min(tl[i], bs.entry_level);
            t.preset.assign(start + 1, {}); t.has_preset = true;
            for (int lc = 0; lc <= start; lc++) {
                const size_t pr = (size_t)prob[i] + (size_t)lc; const uint2 *W = wl.data() + pr * ix->efc;
                t.preset[lc].resize(wcnt[pr]);
                for (uint32_t k = 0; k < wcnt[pr]; k++) { memcpy(&t.preset[lc][k].d, &W[k].x, 4); t.preset[lc][k].id = W[k].y; }
            }
            if (ix->mfma_on()) { int rcn = e->mfma_norms(e->n_rows); if (rcn) return ix->fail(rcn, e->err); }
            ix->arm_select(t.sel);
            tasks.push_back(&t); members.push_back(lo + i);
        }
        if ((rc = ix->run_lockstep(tasks))) return rc;
        for (size_t ti = 0; ti < tasks.size(); ti++) {
            InsertTask &t = *its[ti];
            for (int lc = 0; lc <= t.new_level; lc++) {
                Cand *lst = g.list(t.id, lc);
                for (size_t k = 0; k < t.nb[lc].size(); k++) lst[k] = t.nb[lc][k];
                g.cnt(t.id, lc) = (uint16_t)t.nb[lc].size();
            }
            t.has_preset = false; t.preset.clear();
            ix->counters[1] += t.n_dist; ix->counters[2] += t.n_pair;
            ix->mfma_pairs += t.sel.n_mfma; ix->mfma_exact += t.sel.n_exact;
            ix->mark_dirty(t.id);
            bs.searched[members[ti]] = 1;
        }
        ix->fused_tasks += n; ix->fused_redo += todo.size();
        if (todo.empty()) return HX_OK;
        }
    } else {
        for (uint32_t i = lo; i < hi; i++) todo.push_back(i);
    }
    std::vector<std::unique_ptr<InsertTask>> &its = ix->insert_pool;
    while (its.size() < todo.size()) its.emplace_back(new InsertTask());
    std::vector<LsTask *> tasks(todo.size());
    for (size_t ti = 0; ti < todo.size(); ti++) {
        const uint32_t i = todo[ti];
        InsertTask &t = *its[ti];
        t.has_preset = false;
        t.st = InsertTask::S_INIT; t.lc = 0; t.n_dist = t.n_pair = 0; t.clear_req();
        t.g = &g; t.id = bs.base + i; t.new_level = g.level[t.id]; t.entry = bs.entry; t.entry_level = bs.entry_level; t.efc = ix->efc;
        if (ix->mfma_on()) { int rcn = ix->e->mfma_norms(ix->e->n_rows); if (rcn) return ix->fail(rcn, ix->e->err); }
        ix->arm_select(t.sel);
        tasks[ti] = &t;
    }
    int rc = ix->run_lockstep(tasks);
    if (rc) return rc;
    for (size_t ti = 0; ti < todo.size(); ti++) {               // elements[new_idx].neighbors[lc].items = neighbors (mod.rs:422)
        const uint32_t i = todo[ti];
        InsertTask &t = *its[ti];
        for (int lc = 0; lc <= t.new_level; lc++) {
            Cand *lst = g.list(t.id, lc);
            for (size_t k = 0; k < t.nb[lc].size(); k++) lst[k] = t.nb[lc][k];
            g.cnt(t.id, lc) = (uint16_t)t.nb[lc].size();
        }
        ix->counters[1] += t.n_dist; ix->counters[2] += t.n_pair;
        ix->mfma_pairs += t.sel.n_mfma; ix->mfma_exact += t.sel.n_exact;
        ix->mark_dirty(t.id);
        bs.searched[i] = 1;
    }
    return HX_OK;
}

// serialized neighbour lists of the new members [lo, hi): per member, per layer 0..level: u32 count, lm x {u32 id, f32 d}
static size_t list_bytes(const Graph &g, int layer) { return 4 + (size_t)g.lm(layer) * 8; }
uint64_t hx_index_batch_new_bytes(const hx_index *ix, uint32_t lo, uint32_t hi)
{
    if (!ix || !ix->bs.open || lo > hi || hi > ix->bs.b) return 0;
    uint64_t n = 0;
    for (uint32_t i = lo; i < hi; i++) for (int lc = 0; lc <= ix->g.level[ix->bs.base + i]; lc++) n += list_bytes(ix->g, lc);
    return n;
}
static uint8_t *put_list(const Graph &g, uint32_t e, int layer, uint8_t *p)
{
    const uint32_t c = g.cnt(e, layer); const Cand *l = g.list(e, layer); const int lm = g.lm(layer);
    memcpy(p, &c, 4); p += 4;
    for (int k = 0; k < lm; k++) { Cand x = k < (int)c ? l[k] : Cand{0.0f, 0u}; memcpy(p, &x.id, 4); memcpy(p + 4, &x.d, 4); p += 8; }
    return p;
}
static const uint8_t *get_list(Graph &g, uint32_t e, int layer, const uint8_t *p)
{
    uint32_t c; memcpy(&c, p, 4); p += 4; const int lm = g.lm(layer);
    if (c > (uint32_t)lm) c = (uint32_t)lm;
    Cand *l = g.list(e, layer);
    for (int k = 0; k < lm; k++) { if (k < (int)c) { memcpy(&l[k].id, p, 4); memcpy(&l[k].d, p + 4, 4); } p += 8; }
    g.cnt(e, layer) = (uint16_t)c;
    return p;
}
int hx_index_batch_export_new(const hx_index *ix, uint32_t lo, uint32_t hi, void *buf)
{
    if (!ix || !buf || !ix->bs.open || lo > hi || hi > ix->bs.b) return HX_E_ARG;
    uint8_t *p = (uint8_t *)buf;
    for (uint32_t i = lo; i < hi; i++) { if (!ix->bs.searched[i]) return HX_E_STATE; for (int lc = 0; lc <= ix->g.level[ix->bs.base + i]; lc++) p = put_list(ix->g, ix->bs.base + i, lc, p); }
    return HX_OK;
}
int hx_index_batch_import_new(hx_index *ix, uint32_t lo, uint32_t hi, const void *buf)
{
    if (!ix || !buf || !ix->bs.open || lo > hi || hi > ix->bs.b) return HX_E_ARG;
    const uint8_t *p = (const uint8_t *)buf;
    for (uint32_t i = lo; i < hi; i++) { for (int lc = 0; lc <= ix->g.level[ix->bs.base + i]; lc++) p = get_list(ix->g, ix->bs.base + i, lc, p); ix->bs.searched[i] = 1; ix->mark_dirty(ix->bs.base + i); }
    return HX_OK;
}

// duplicate merge + entry-point update (every rank, identical), then back-link pruning of the lists this rank owns
// (owner of a list = target row id % world)
int hx_index_batch_links(hx_index *ix, uint32_t rank, uint32_t world)
{
    if (!ix) return HX_E_ARG;
    BatchState &bs = ix->bs; Graph &g = ix->g;
    if (!bs.open || world == 0 || rank >= world) return ix->fail(HX_E_STATE, "no open batch / bad rank");
    for (uint32_t i = 0; i < bs.b; i++) if (!bs.searched[i]) return ix->fail(HX_E_STATE, "batch member without neighbour lists (search or import it first)");
    const uint32_t b = bs.b, base = bs.base;
    int rc;
    double t_links0 = hx_index::now_s();
    // duplicate detection (build.rs:482-512): byte-compare the leading zero-distance layer-0 neighbours; rows that are identical to
    // an EARLIER member of this batch are found by hashing on the device (they are not linked yet, so the graph cannot show them)
    auto &da = ix->ls.da; auto &db = ix->ls.db; da.clear(); db.clear();
    for (uint32_t i = 0; i < b; i++) {
        const uint32_t id = base + i; const Cand *l0 = g.list(id, 0);
        for (uint16_t k = 0; k < g.cnt(id, 0); k++) { if (l0[k].d != 0.0f) break; da.push_back(id); db.push_back(l0[k].id); }
    }
    auto &deq = ix->ls.deq; deq.assign(da.size(), 0);
    if (!da.empty() && (rc = hx_rows_equal(ix->e, (uint32_t)da.size(), da.data(), db.data(), deq.data()))) return ix->fail(rc, ix->e->err);
    auto &za = ix->ls.za; auto &zb = ix->ls.zb; auto &ha = ix->ls.ha; auto &hb = ix->ls.hb;
    if ((rc = ix->e->db_dup_candidates(base, b, nullptr, za, zb, ha, hb))) return ix->fail(rc, ix->e->err);
    za.clear(); zb.clear();
    for (size_t k = 0; k < da.size(); k++) if (deq[k]) { za.push_back(da[k]); zb.push_back(db[k]); }
    ix->merge_duplicates(za, zb, ha, hb);
    std::vector<BackOp> &ops = bs.ops; ops.clear();
    auto &opstart = ix->ls.opstart; opstart.assign(b + 1, 0u);   // ops of member i land at [opstart[i], opstart[i+1])
    for (uint32_t i = 0; i < b; i++) {
        const uint32_t id = base + i;
        uint32_t nops = 0;
        if (g.level[id] < 0) { const int lv = -1 - g.level[id]; for (int lc = 0; lc <= lv; lc++) g.cnt(id, lc) = 0; }   // tombstone: no links
        else for (int lc = g.level[id]; lc >= 0; lc--) nops += g.cnt(id, lc);
        opstart[i + 1] = opstart[i] + nops;
    }
    // Single-process build on the device path: the ops are only listed here (update_neighbor_connections order, mod.rs:451-458);
    // grouping them per (target, layer) list, hub split and the prunes all happen on the device (hx_group.hip), and the
    // updated lists stay in the mirror until the host needs them.
    static const bool device_grouping = !(getenv("HX_DEVICE_GROUPING") && atoi(getenv("HX_DEVICE_GROUPING")) == 0);
    if (device_grouping && bs.lazy_lists && world == 1 && ix->fused_ok() && g.m == 16 && ix->e->pitch <= 8192) {
        const uint32_t n_ops = opstart[b];
        unsigned long long *keys = nullptr; uint32_t *onew = nullptr; float *od = nullptr;   // pinned staging; key = target << 7 | layer (levels reach 82 at m = 16)
        if ((rc = ix->e->links_stage_ops(n_ops, &keys, &onew, &od))) return ix->fail(rc, ix->e->err);
        const uint32_t csz = 128, nck = (b + csz - 1) / csz;
        ix->pool->parallel_for(nck, [&](size_t ci) {
            for (uint32_t i = (uint32_t)ci * csz; i < std::min(b, (uint32_t)(ci + 1) * csz); i++) {
                const uint32_t id = base + i; uint32_t o = opstart[i];
                if (opstart[i + 1] == o) continue;
                for (int lc = g.level[id]; lc >= 0; lc--) {
                    const Cand *lst = g.list(id, lc);
                    for (uint16_t k = 0; k < g.cnt(id, lc); k++) { keys[o] = ((unsigned long long)lst[k].id << 7) | (unsigned long long)lc; onew[o] = id; od[o] = lst[k].d; o++; }
                }
            }
        });
        if ((rc = ix->sync_mirror())) return rc;
        ix->prof[8] += hx_index::now_s() - t_links0;
        {
            hx_index::Timer tl(ix->prof[9]);
            uint64_t np = 0; uint32_t st[2] = {0, 0};
            if ((rc = ix->e->links_run_grouped(n_ops, keys, onew, od, &np, st))) return ix->fail(rc, ix->e->err);
            ix->counters[3] += np;
            ix->prof[13] = std::max(ix->prof[13], (double)st[1]); ix->prof[14] += n_ops;
            if (n_ops) ix->host_stale = true;
        }
        bs.ops.clear(); bs.grp.clear();
        bs.linked = true;
        return HX_OK;
    }
    // Back-link ops in update_neighbor_connections order (mod.rs:451-458), then grouped per (target, layer) with the
    // insertion order kept inside a group: parallel stable bucket sort (bucket = target & 255; buckets are independent).
    {
        const uint32_t n_ops = opstart[b];
        constexpr uint32_t NB = 256;
        const uint32_t csz = 128, nck = (b + csz - 1) / csz;
        auto &raw = ix->ls.raw; raw.resize(n_ops);
        auto &hist = ix->ls.hist; hist.assign((size_t)nck * NB, 0u);
        ix->pool->parallel_for(nck, [&](size_t ci) {
            uint32_t *h = &hist[ci * NB];
            for (uint32_t i = (uint32_t)ci * csz; i < std::min(b, (uint32_t)(ci + 1) * csz); i++) {
                const uint32_t id = base + i; uint32_t o = opstart[i];
                if (opstart[i + 1] == o) continue;
                for (int lc = g.level[id]; lc >= 0; lc--) {
                    const Cand *lst = g.list(id, lc);
                    for (uint16_t k = 0; k < g.cnt(id, lc); k++) {
                        raw[o] = BackOp{lst[k].id, lc, id, lst[k].d};
                        if (lst[k].id % world == rank) h[lst[k].id & (NB - 1)]++; else raw[o].layer = -1;   // another rank prunes that list
                        o++;
                    }
                }
            }
        });
        std::vector<uint32_t> bstart(NB + 1, 0u);
        for (uint32_t bk = 0; bk < NB; bk++) { uint32_t t = 0; for (uint32_t c = 0; c < nck; c++) t += hist[(size_t)c * NB + bk]; bstart[bk + 1] = bstart[bk] + t; }
        // per (chunk, bucket) write cursor = bucket start + ops of earlier chunks in that bucket (keeps insertion order)
        for (uint32_t bk = 0; bk < NB; bk++) { uint32_t run = bstart[bk]; for (uint32_t c = 0; c < nck; c++) { const uint32_t t = hist[(size_t)c * NB + bk]; hist[(size_t)c * NB + bk] = run; run += t; } }
        ops.resize(bstart[NB]);
        ix->pool->parallel_for(nck, [&](size_t ci) {
            uint32_t *cur = &hist[ci * NB];
            const uint32_t lo = opstart[std::min<size_t>(b, ci * csz)], hi = opstart[std::min<size_t>(b, (ci + 1) * csz)];
            for (uint32_t o = lo; o < hi; o++) if (raw[o].layer >= 0) ops[cur[raw[o].target & (NB - 1)]++] = raw[o];
        });
        auto &bgrp = ix->ls.bgrp; bgrp.resize(NB); for (auto &v : bgrp) v.clear();
        ix->pool->parallel_for(NB, [&](size_t bk) {
            std::stable_sort(ops.begin() + bstart[bk], ops.begin() + bstart[bk + 1],
                             [](const BackOp &a, const BackOp &c) { return a.target != c.target ? a.target < c.target : a.layer < c.layer; });
            for (size_t s0 = bstart[bk]; s0 < bstart[bk + 1];) {
                size_t t = s0; while (t < bstart[bk + 1] && ops[t].target == ops[s0].target && ops[t].layer == ops[s0].layer) t++;
                bgrp[bk].push_back({s0, t});
                s0 = t;
            }
        });
        bs.grp.clear();
        for (uint32_t bk = 0; bk < NB; bk++) bs.grp.insert(bs.grp.end(), bgrp[bk].begin(), bgrp[bk].end());
    }
    if (ix->fused_ok() && g.lm(0) + 1 <= 33) {
        // device path: k_links applies every owned list's back-links (append / prune) in one launch
        if ((rc = ix->sync_mirror())) return rc;
        auto &own = ix->ls.own; own.clear();                     // indices into bs.grp of the lists this rank owns
        own.reserve(bs.grp.size());
        for (uint32_t gi = 0; gi < bs.grp.size(); gi++) if (ops[bs.grp[gi].first].target % world == rank) own.push_back(gi);
        const uint32_t ng = (uint32_t)own.size(), lm0 = 2u * (uint32_t)g.m;
        auto &tg = ix->ls.tg; auto &ly = ix->ls.ly; auto &off = ix->ls.off; tg.resize(ng); ly.resize(ng); off.assign(ng + 1, 0u);
        for (uint32_t k = 0; k < ng; k++) off[k + 1] = off[k] + (uint32_t)(bs.grp[own[k]].second - bs.grp[own[k]].first);
        for (uint32_t k = 0; k < ng; k++) ix->prof[13] = std::max(ix->prof[13], (double)(off[k + 1] - off[k]));   // longest serial op chain of one list
        ix->prof[14] += off[ng];
        auto &onew = ix->ls.onew; auto &od = ix->ls.od; onew.resize(off[ng]); od.resize(off[ng]);
        ix->pool->parallel_for((ng + 4095) / 4096, [&](size_t ci) {
            for (size_t k = ci * 4096; k < std::min<size_t>(ng, ci * 4096 + 4096); k++) {
                const auto &gr = bs.grp[own[k]];
                tg[k] = ops[gr.first].target; ly[k] = (uint32_t)ops[gr.first].layer;
                uint32_t o = off[k];
                for (size_t q = gr.first; q < gr.second; q++) { onew[o] = ops[q].new_id; od[o] = ops[q].d; o++; }
            }
        });
        ix->prof[8] += hx_index::now_s() - t_links0;
        if (ng) {
            hx_index::Timer tl(ix->prof[9]);
            const uint32_t *oids = nullptr, *ocnt = nullptr; const float *odd = nullptr;
            uint64_t np = 0;
            const bool lazy = bs.lazy_lists;
            if ((rc = ix->e->links_run(ng, tg.data(), ly.data(), off.data(), onew.data(), od.data(), &oids, &odd, &ocnt, &np, !lazy))) return ix->fail(rc, ix->e->err);
            ix->counters[3] += np;
            if (lazy) ix->host_stale = true;
            else
            ix->pool->parallel_for((ng + 2047) / 2048, [&](size_t ci) {
                for (size_t gi = ci * 2048; gi < std::min<size_t>(ng, ci * 2048 + 2048); gi++) {
                    Cand *lst = g.list(tg[gi], (int)ly[gi]); const uint32_t c = ocnt[gi];
                    for (uint32_t k = 0; k < c; k++) lst[k] = Cand{odd[gi * lm0 + k], oids[gi * lm0 + k]};
                    g.cnt(tg[gi], (int)ly[gi]) = (uint16_t)c;
                }
            });
        }
        bs.linked = true;
        return HX_OK;
    }
    static const bool biglist_off = getenv("HX_BIGLIST") && atoi(getenv("HX_BIGLIST")) == 0;
    if (ix->fused && ix->e->pitch <= 8192 && !biglist_off) {
        // lists the lane-per-slot kernels do not serve (m > 32: up to 200 slots; sparsevec rows): k_list_ops (hx_biglist.hip), one wavefront per owned list, from the
        // list contents the host hands over; the new lists come back into the host copy (the master for these index shapes)
        auto &own = ix->ls.own; own.clear();
        for (uint32_t gi = 0; gi < bs.grp.size(); gi++) if (ops[bs.grp[gi].first].target % world == rank) own.push_back(gi);
        const uint32_t ng = (uint32_t)own.size(), lm0 = 2u * (uint32_t)g.m;
        if (ng) {
            auto &off = ix->ls.off; off.assign(ng + 1, 0u);
            for (uint32_t k = 0; k < ng; k++) off[k + 1] = off[k] + (uint32_t)(bs.grp[own[k]].second - bs.grp[own[k]].first);
            for (uint32_t k = 0; k < ng; k++) ix->prof[13] = std::max(ix->prof[13], (double)(off[k + 1] - off[k]));
            ix->prof[14] += off[ng];
            uint32_t *h_ids, *h_cnt, *h_lm, *h_off, *h_new; float *h_d, *h_od;
            if ((rc = ix->e->biglist_ops_stage(ng, off[ng], lm0, &h_ids, &h_d, &h_cnt, &h_lm, &h_off, &h_new, &h_od))) return ix->fail(rc, ix->e->err);
            memcpy(h_off, off.data(), ((size_t)ng + 1) * 4);
            ix->pool->parallel_for((ng + 1023) / 1024, [&](size_t ci) {
                for (size_t k = ci * 1024; k < std::min<size_t>(ng, ci * 1024 + 1024); k++) {
                    const auto &gr = bs.grp[own[k]];
                    const uint32_t t = ops[gr.first].target; const int ly = ops[gr.first].layer;
                    const Cand *lst = g.list(t, ly); const uint32_t c = g.cnt(t, ly);
                    for (uint32_t i = 0; i < c; i++) { h_ids[k * lm0 + i] = lst[i].id; h_d[k * lm0 + i] = lst[i].d; }
                    h_cnt[k] = c; h_lm[k] = (uint32_t)g.lm(ly);
                    uint32_t o = off[k];
                    for (size_t q = gr.first; q < gr.second; q++) { h_new[o] = ops[q].new_id; h_od[o] = ops[q].d; o++; }
                }
            });
            ix->prof[8] += hx_index::now_s() - t_links0;
            uint64_t np = 0;
            { hx_index::Timer tl(ix->prof[9]); if ((rc = ix->e->biglist_ops_run(&np))) return ix->fail(rc, ix->e->err); }
            ix->counters[3] += np;
            ix->pool->parallel_for((ng + 1023) / 1024, [&](size_t ci) {
                for (size_t k = ci * 1024; k < std::min<size_t>(ng, ci * 1024 + 1024); k++) {
                    const auto &gr = bs.grp[own[k]];
                    const uint32_t t = ops[gr.first].target; const int ly = ops[gr.first].layer;
                    Cand *lst = g.list(t, ly); const uint32_t c = std::min<uint32_t>(h_cnt[k], (uint32_t)g.lm(ly));
                    for (uint32_t i = 0; i < c; i++) lst[i] = Cand{h_d[k * lm0 + i], h_ids[k * lm0 + i]};
                    g.cnt(t, ly) = (uint16_t)c;
                }
            });
            for (uint32_t k = 0; k < ng; k++) ix->dirty.emplace_back(ops[bs.grp[own[k]].first].target, ops[bs.grp[own[k]].first].layer);
        }
        bs.linked = true;
        return HX_OK;
    }
    std::vector<std::unique_ptr<BacklinkTask>> &bts = ix->backlink_pool; std::vector<LsTask *> btasks;
    size_t nbt = 0;
    for (const auto &gr : bs.grp) {
        if (ops[gr.first].target % world != rank) continue;
        if (nbt == bts.size()) bts.emplace_back(new BacklinkTask());
        BacklinkTask &bt = *bts[nbt++];
        bt.k = 0; bt.selecting = false; bt.n_dist = bt.n_pair = 0; bt.clear_req();
        bt.g = &g; bt.target = ops[gr.first].target; bt.layer = ops[gr.first].layer; bt.ops.assign(ops.begin() + gr.first, ops.begin() + gr.second);
        if (ix->mfma_on()) { int rcn = ix->e->mfma_norms(ix->e->n_rows); if (rcn) return ix->fail(rcn, ix->e->err); }
        ix->arm_select(bt.sel);
        btasks.push_back(&bt);
    }
    ix->prof[8] += hx_index::now_s() - t_links0;      // duplicates + op grouping + task setup
    { hx_index::Timer tl(ix->prof[9]); if ((rc = ix->run_lockstep(btasks))) return rc; }
    for (size_t i = 0; i < nbt; i++) { ix->counters[3] += bts[i]->n_pair; ix->mfma_pairs += bts[i]->sel.n_mfma; ix->mfma_exact += bts[i]->sel.n_exact; ix->dirty.emplace_back(bts[i]->target, bts[i]->layer); }
    bs.linked = true;
    return HX_OK;
}

// serialized lists owned by `owner` after the links stage, in (target, layer) order -- every rank derives the same order
// serialized lists this rank pruned in the links stage (it grouped only the ops whose target it owns), self-describing:
// per list u32 target, u32 layer, then the list record (u32 count, lm x {u32 id, f32 d})
uint64_t hx_index_batch_links_bytes(const hx_index *ix)
{
    if (!ix || !ix->bs.open || !ix->bs.linked) return 0;
    uint64_t n = 0;
    for (const auto &gr : ix->bs.grp) n += 8 + list_bytes(ix->g, ix->bs.ops[gr.first].layer);
    return n;
}
int hx_index_batch_export_links(const hx_index *ix, void *buf)
{
    if (!ix || !buf || !ix->bs.open || !ix->bs.linked) return HX_E_ARG;
    const BatchState &bs = ix->bs; const size_t ng = bs.grp.size();
    std::vector<size_t> off(ng + 1, 0);
    for (size_t k = 0; k < ng; k++) off[k + 1] = off[k] + 8 + list_bytes(ix->g, bs.ops[bs.grp[k].first].layer);
    uint8_t *base = (uint8_t *)buf;
    ix->pool->parallel_for((ng + 4095) / 4096, [&](size_t ci) {
        for (size_t k = ci * 4096; k < std::min(ng, ci * 4096 + 4096); k++) {
            const BackOp &o = bs.ops[bs.grp[k].first];
            uint8_t *p = base + off[k];
            const uint32_t ly = (uint32_t)o.layer;
            memcpy(p, &o.target, 4); memcpy(p + 4, &ly, 4);
            put_list(ix->g, o.target, o.layer, p + 8);
        }
    });
    return HX_OK;
}
int hx_index_batch_import_links(hx_index *ix, const void *buf, uint64_t nbytes)
{
    if (!ix || (!buf && nbytes) || !ix->bs.open || !ix->bs.linked) return HX_E_ARG;
    Graph &g = ix->g;
    const uint8_t *base = (const uint8_t *)buf;
    std::vector<size_t> off;                                   // record boundaries (the record size depends on its layer)
    for (size_t o = 0; o + 8 <= nbytes;) {
        uint32_t tg, ly; memcpy(&tg, base + o, 4); memcpy(&ly, base + o + 4, 4);
        if (tg >= g.size() || g.level[tg] < 0 || ly > (uint32_t)g.level[tg]) return ix->fail(HX_E_ARG, "corrupt link record");   // unsigned: a layer word >= 2^31 must not pass as negative
        off.push_back(o);
        o += 8 + list_bytes(g, (int)ly);
        if (o > nbytes) return ix->fail(HX_E_ARG, "truncated link record");
    }
    const size_t nr = off.size(), d0 = ix->dirty.size();
    ix->dirty.resize(d0 + nr);
    ix->pool->parallel_for((nr + 4095) / 4096, [&](size_t ci) {
        for (size_t k = ci * 4096; k < std::min(nr, ci * 4096 + 4096); k++) {
            uint32_t tg, ly; memcpy(&tg, base + off[k], 4); memcpy(&ly, base + off[k] + 4, 4);
            get_list(g, tg, (int)ly, base + off[k] + 8);
            ix->dirty[d0 + k] = {tg, (int)ly};
        }
    });
    return HX_OK;
}

int hx_index_batch_end(hx_index *ix, uint32_t *elem_out)
{
    if (!ix) return HX_E_ARG;
    BatchState &bs = ix->bs;
    if (!bs.open || !bs.linked) return ix->fail(HX_E_STATE, "batch not linked yet");
    if (elem_out) memcpy(elem_out, bs.elem.data(), (size_t)bs.b * sizeof(uint32_t));
    bs.open = false; bs.linked = false; bs.lazy_lists = false; bs.b = 0;      // vectors keep their capacity for the next batch
    return HX_OK;
}


// ================================================================================================
// Device-resident batches (hx_batch.hip): the same batch as hx_index_batch_*, but the members' neighbour lists stay in device
// records from k_fused<insert> to the back-link kernels; the host only sees statuses, duplicate candidates and counters.
// hx_index_insert runs begin, search(0, b), links(0, 1), end on an engine-owned record buffer; a multi-GPU build passes its
// exchange buffers (pgvector-rx_amd/dist_build.py) and all-gathers them between the stages.
// ================================================================================================
int hx_index_dbatch_supported(const hx_index *ix, const int32_t *levels, uint32_t b)
{
    if (!ix || !levels || !ix->dbatch_ok() || ix->g.entry < 0) return 0;
    const int mxl = max_level_for(ix->g.m);
    for (uint32_t i = 0; i < b; i++) if (std::min(levels[i], mxl) >= HX_FUSED_MAXL) return 0;   // a member above the kernel's layers: host path for this batch
    return 1;
}
uint64_t hx_index_dbatch_record_bytes(const hx_index *ix) { return ix ? (uint64_t)hx_rec_words((uint32_t)ix->g.m) * 4 : 0; }
uint64_t hx_index_dbatch_list_record_bytes(const hx_index *ix) { return ix ? (uint64_t)hx_xrec_words((uint32_t)ix->g.m) * 4 : 0; }

int hx_index_dbatch_begin(hx_index *ix, uint64_t first_row, uint32_t b, const int32_t *levels, const int64_t *tids)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix) return HX_E_ARG;
    hx_index::Timer t_bb(ix->prof[12]);
    if (b == 0 || !levels || !tids) return ix->fail(HX_E_ARG, "empty batch or NULL argument");
    Graph &g = ix->g; BatchState &bs = ix->bs;
    if (bs.open) return ix->fail(HX_E_STATE, "a batch is already open");
    if (!hx_index_dbatch_supported(ix, levels, b)) return ix->fail(HX_E_STATE, "this batch cannot run device-resident (hx_index_dbatch_supported)");
    if (first_row != g.size()) return ix->fail(HX_E_STATE, "rows must be inserted in append order: first_row != index size");
    if (first_row + b > hx_num_rows(ix->e)) return ix->fail(HX_E_ARG, "rows not present in the engine");
    const int mxl = max_level_for(g.m);
    {   // keep the vectors' capacity from batch to batch
        BatchState fresh;
        fresh.tids.swap(bs.tids); fresh.elem.swap(bs.elem); fresh.searched.swap(bs.searched); fresh.ops.swap(bs.ops); fresh.grp.swap(bs.grp);
        fresh.tids.clear(); fresh.elem.clear(); fresh.searched.clear(); fresh.ops.clear(); fresh.grp.clear();
        bs = std::move(fresh);
    }
    bs.open = true; bs.dev = true; bs.base = g.size(); bs.b = b; bs.entry = (uint32_t)g.entry; bs.entry_level = g.level[g.entry];
    bs.tids.assign(tids, tids + b); bs.elem.assign(b, 0); bs.searched.assign(b, 0);
    g.add_bulk(levels, b, mxl);
    // levels / upper-layer blocks of the new elements and every list the host wrote since the last launch go to the mirror now;
    // from here on the mirror is the only place this batch's lists exist
    int rc = ix->sync_mirror();
    if (rc) { ix->rollback_batch(); return rc; }
    if ((rc = ix->e->db_begin_wtabs(bs.base, b, (uint32_t)ix->efc))) { ix->rollback_batch(); return ix->fail(rc, ix->e->err); }
    return HX_OK;
}

// find_element_neighbors for members [lo, hi); their records are written at d_records + (member - lo) * record_bytes
int hx_index_dbatch_search(hx_index *ix, uint32_t lo, uint32_t hi, void *d_records)
{
    if (!ix) return HX_E_ARG;
    hx_index::Timer t_bs(ix->prof[11]);
    BatchState &bs = ix->bs; Graph &g = ix->g;
    if (!bs.open || !bs.dev || lo > hi || hi > bs.b) return ix->fail(HX_E_STATE, "no open device batch / bad member range");
    if (hi == lo) return HX_OK;
    if (!d_records) return ix->fail(HX_E_ARG, "d_records is NULL");
    const uint32_t n = hi - lo, rw = hx_rec_words((uint32_t)g.m);
    // Longest tasks first: a member with a higher level searches (and selects on) more layers, and a long task that starts in the launch's last round
    // is what the whole chip then waits for.  The kernel pulls tasks in list order and addresses every output through the task's slot, so the order
    // is free (HX_TASK_ORDER=0: member order).
    std::vector<uint32_t> qsel(n), status(n), tstat(n), slots(n); std::vector<int32_t> tl(n);
    for (uint32_t i = 0; i < n; i++) slots[i] = i;
    static const int order_env = getenv("HX_TASK_ORDER") ? atoi(getenv("HX_TASK_ORDER")) : 1;
    if (order_env) std::stable_sort(slots.begin(), slots.end(), [&](uint32_t a, uint32_t b) { return g.level[bs.base + lo + a] > g.level[bs.base + lo + b]; });
    for (uint32_t k = 0; k < n; k++) { qsel[k] = bs.base + lo + slots[k]; tl[k] = g.level[bs.base + lo + slots[k]]; }
    uint64_t cnts[2] = {0, 0};
    HxFusedDev dev; dev.d_rec = (uint32_t *)d_records; dev.rec_words = rw; dev.h_slots = slots.data();
    if (ix->e->bw.wt_size) { dev.d_wtab = ix->e->bw.d_wtab; dev.wt_size = ix->e->bw.wt_size; dev.wt_slot0 = lo; dev.d_wt_valid = ix->e->bw.d_wt_valid; }
    int rc;
    const double t0 = hx_index::now_s();
    // halfvec inner product with hx_index_set_mfma(1): the traversal kernel stops after each layer's search (MODE 3) and select_neighbors runs on the
    // matrix cores (hx_mfma.hip: k_wgemm_f16 + k_wselect) -- SURVEY 8 row g on the default placement.  Same lists, same distance bits.
    const bool split = ix->mfma_on() && ix->efc <= 256 && ix->e->pitch > 512;
    if (split) {
        hx_engine *e = ix->e;
        if ((rc = e->mfma_norms(e->n_rows))) return ix->fail(rc, e->err);
        std::vector<uint32_t> prob(n), pslot, ptask; std::vector<uint8_t> player;
        for (uint32_t k = 0; k < n; k++) {
            prob[k] = (uint32_t)player.size();
            const int start = std::min<int>(tl[k], bs.entry_level);
            for (int lc = 0; lc <= start; lc++) { player.push_back((uint8_t)lc); pslot.push_back(slots[k]); ptask.push_back(k); }
        }
        const uint32_t P = (uint32_t)player.size();
        if ((rc = e->wsel_reserve(P, (uint32_t)ix->efc))) return ix->fail(rc, e->err);
        HxWselWork &w = e->wsel;
        if (hipMemcpyAsync(w.d_layer, player.data(), P, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(w.d_slot, pslot.data(), (size_t)P * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemcpyAsync(w.d_task, ptask.data(), (size_t)P * 4, hipMemcpyHostToDevice, e->stream) != hipSuccess ||
            hipMemsetAsync(w.d_cnt, 0, (size_t)P * 4, e->stream) != hipSuccess || hipMemsetAsync(w.d_counters, 0, 32, e->stream) != hipSuccess ||
            hipStreamSynchronize(e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "matrix-core select: staging the problem table failed");
        dev.d_wl_out = w.d_wl; dev.d_wl_cnt = w.d_cnt; dev.h_prob = prob.data();
        if ((rc = e->fused_run(3, n, qsel.data(), tl.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                               nullptr, nullptr, nullptr, tstat.data(), cnts, nullptr, nullptr, 1, &dev))) return ix->fail(rc, e->err);
        const uint32_t *d_status = (const uint32_t *)(e->mirror.io.d_io + e->mirror.io.o_st);      // the launch's statuses stay there until the next launch
        if ((rc = e->mfma_select(P, (uint32_t)ix->efc, w.d_wl, w.d_cnt, w.d_layer, w.d_slot, w.d_task, d_status, dev.d_rec, rw, w.d_counters))) return ix->fail(rc, e->err);
        unsigned long long wc[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(wc, w.d_counters, 32, hipMemcpyDeviceToHost, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
            return ix->fail(HX_E_HIP, "matrix-core select failed");
        if ((rc = e->mfma_select_done(wc[2]))) return ix->fail(rc, e->err);
        ix->mfma_pairs += wc[0]; ix->mfma_exact += wc[1];
        cnts[1] += wc[1];                                          // select distances evaluated by streaming rows (the canonical re-evaluations)
        dev.d_wl_out = nullptr; dev.d_wl_cnt = nullptr; dev.h_prob = nullptr;   // the overflow retry below is a plain MODE 1 launch
    } else
    if ((rc = ix->e->fused_run(1, n, qsel.data(), tl.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                               nullptr, nullptr, nullptr, tstat.data(), cnts, nullptr, nullptr, 1, &dev))) return ix->fail(rc, ix->e->err);
    for (uint32_t k = 0; k < n; k++) status[slots[k]] = tstat[k];
    ix->prof[6] += hx_index::now_s() - t0;
    ix->counters[1] += cnts[0]; ix->counters[2] += cnts[1];
    std::vector<uint32_t> again, todo;
    for (uint32_t i = 0; i < n; i++) { if (status[i] == 0) bs.searched[lo + i] = 1; else if (status[i] == 1) again.push_back(i); else todo.push_back(i); }
    if (!again.empty()) {                                        // overflowed tables: one more try on the device with roomier ones
        const uint32_t na = (uint32_t)again.size();
        std::vector<uint32_t> q2(na), st2(na); std::vector<int32_t> l2(na);
        for (uint32_t k = 0; k < na; k++) { q2[k] = bs.base + lo + again[k]; l2[k] = g.level[bs.base + lo + again[k]]; }
        HxFusedDev dev2 = dev; dev2.h_slots = again.data();
        if ((rc = ix->e->fused_run(1, na, q2.data(), l2.data(), (uint32_t)ix->efc, 0, bs.entry, bs.entry_level,
                                   nullptr, nullptr, nullptr, st2.data(), cnts, nullptr, nullptr, 8, &dev2))) return ix->fail(rc, ix->e->err);
        ix->counters[1] += cnts[0]; ix->counters[2] += cnts[1];
        for (uint32_t k = 0; k < na; k++) { if (st2[k] == 0) bs.searched[lo + again[k]] = 1; else todo.push_back(again[k]); }
    }
    ix->fused_tasks += n; ix->fused_redo += todo.size();
    if (todo.empty()) return HX_OK;
    // still overflowing: the lock-step driver on the host copy of the graph as of the batch start (the mirror holds exactly that)
    ix->host_stale = true;
    std::vector<std::unique_ptr<InsertTask>> &its = ix->insert_pool;
    while (its.size() < todo.size()) its.emplace_back(new InsertTask());
    std::vector<LsTask *> tasks(todo.size());
    for (size_t ti = 0; ti < todo.size(); ti++) {
        InsertTask &t = *its[ti];
        t.has_preset = false;
        t.st = InsertTask::S_INIT; t.lc = 0; t.n_dist = t.n_pair = 0; t.clear_req();
        t.g = &g; t.id = bs.base + lo + todo[ti]; t.new_level = g.level[t.id]; t.entry = bs.entry; t.entry_level = bs.entry_level; t.efc = ix->efc;
        if (ix->mfma_on()) { int rcn = ix->e->mfma_norms(ix->e->n_rows); if (rcn) return ix->fail(rcn, ix->e->err); }
        ix->arm_select(t.sel);
        tasks[ti] = &t;
    }
    if ((rc = ix->run_lockstep(tasks))) return rc;
    const uint32_t lm0 = 2u * (uint32_t)g.m;
    std::vector<uint32_t> src(HX_FUSED_MAXL + 2u * HX_FUSED_MAXL * lm0);
    for (size_t ti = 0; ti < todo.size(); ti++) {
        InsertTask &t = *its[ti];
        std::fill(src.begin(), src.end(), 0u);
        for (int lc = 0; lc <= t.new_level; lc++) {
            src[lc] = (uint32_t)t.nb[lc].size();
            for (size_t k = 0; k < t.nb[lc].size(); k++) {
                src[HX_FUSED_MAXL + (size_t)lc * lm0 + k] = t.nb[lc][k].id;
                memcpy(&src[HX_FUSED_MAXL + HX_FUSED_MAXL * lm0 + (size_t)lc * lm0 + k], &t.nb[lc][k].d, 4);
            }
        }
        if ((rc = ix->e->db_fill_record((uint32_t *)d_records, todo[ti], src.data()))) return ix->fail(rc, ix->e->err);
        ix->counters[1] += t.n_dist; ix->counters[2] += t.n_pair;
        bs.searched[lo + todo[ti]] = 1;
    }
    return HX_OK;
}

// every rank: duplicate merge + entry point + the members' lists into the mirror; then update_neighbor_connections for the lists this rank owns.
// d_records holds ALL b member records (record i = member i).  *n_list_records = lists this rank pruned (their records: export_links).
int hx_index_dbatch_links(hx_index *ix, uint32_t rank, uint32_t world, const void *d_records, uint64_t *n_list_records)
{
    if (!ix) return HX_E_ARG;
    BatchState &bs = ix->bs;
    if (n_list_records) *n_list_records = 0;
    if (!bs.open || !bs.dev || world == 0 || rank >= world || !d_records) return ix->fail(HX_E_STATE, "no open device batch / bad rank");
    const uint32_t b = bs.b, base = bs.base;
    const double t0 = hx_index::now_s();
    int rc;
    auto &za = ix->ls.za; auto &zb = ix->ls.zb; auto &ha = ix->ls.ha; auto &hb = ix->ls.hb;
    if ((rc = ix->e->db_dup_candidates(base, b, (const uint32_t *)d_records, za, zb, ha, hb))) return ix->fail(rc, ix->e->err);
    const uint32_t n_dup = ix->merge_duplicates(za, zb, ha, hb);
    uint32_t n_ops = 0;
    if ((rc = ix->e->db_apply(base, b, (const uint32_t *)d_records, n_dup ? ix->ls.dupflag.data() : nullptr, rank, world, &n_ops))) return ix->fail(rc, ix->e->err);
    ix->prof[8] += hx_index::now_s() - t0;
    {
        hx_index::Timer tl(ix->prof[9]);
        uint64_t np = 0; uint32_t st[2] = {0, 0};
        if ((rc = ix->e->links_run_grouped(n_ops, nullptr, nullptr, nullptr, &np, st, true, world > 1))) return ix->fail(rc, ix->e->err);
        ix->counters[3] += np;
        ix->prof[13] = std::max(ix->prof[13], (double)st[1]); ix->prof[14] += n_ops;
    }
    ix->host_stale = true;
    bs.linked = true;
    if (n_list_records) *n_list_records = world > 1 ? ix->e->xl_records : 0;
    return HX_OK;
}

int hx_index_dbatch_export_links(hx_index *ix, void *d_out)
{
    if (!ix || !ix->bs.open || !ix->bs.dev || !ix->bs.linked) return HX_E_ARG;
    hx_engine *e = ix->e;
    if (e->xl_records == 0) return HX_OK;
    if (!d_out) return ix->fail(HX_E_ARG, "d_out is NULL");
    const size_t bytes = (size_t)e->xl_records * hx_xrec_words((uint32_t)ix->g.m) * 4;
    if (hipMemcpyAsync(d_out, e->d_xl, bytes, hipMemcpyDeviceToDevice, e->stream) != hipSuccess || hipStreamSynchronize(e->stream) != hipSuccess)
        return ix->fail(HX_E_HIP, "export of the pruned-list records failed");
    return HX_OK;
}

int hx_index_dbatch_import_links(hx_index *ix, const void *d_list_records, uint64_t n)
{
    if (!ix || !ix->bs.open || !ix->bs.dev || !ix->bs.linked) return HX_E_ARG;
    if (n == 0) return HX_OK;
    if (!d_list_records || n > 0xFFFFFFFFull) return ix->fail(HX_E_ARG, "bad list records");
    int rc = ix->e->db_import_lists((const uint32_t *)d_list_records, (uint32_t)n);
    if (rc) return ix->fail(rc, ix->e->err);
    ix->host_stale = true;
    return HX_OK;
}

// The members' W tables (d(new row, x) for every x the member's layer-0 search kept: what the back-link kernels look up instead of streaming row x)
// exist only on the rank that searched the member.  Exchange format, per member: the table (wt_size x {distance bits, id}) + 16 bytes whose first is
// the valid flag.  A rank that imports the other ranks' tables prunes its own lists with the same look-ups a single GPU has.
uint64_t hx_index_dbatch_wtab_bytes(const hx_index *ix)
{
    if (!ix || !ix->bs.open || !ix->bs.dev || ix->e->bw.wt_size == 0) return 0;
    return (uint64_t)ix->e->bw.wt_size * 8 + 16;
}

static int wtab_copy(hx_index *ix, uint32_t lo, uint32_t hi, void *d_buf, bool out)
{
    if (!ix || !ix->bs.open || !ix->bs.dev) return HX_E_ARG;
    hx_engine *e = ix->e;
    if (hi < lo || hi > ix->bs.b) return ix->fail(HX_E_ARG, "bad member range");
    if (hi == lo || e->bw.wt_size == 0) return HX_OK;
    if (!d_buf) return ix->fail(HX_E_ARG, "buffer is NULL");
    const size_t tb = (size_t)e->bw.wt_size * 8, stride = tb + 16, n = hi - lo;
    uint8_t *tab = (uint8_t *)e->bw.d_wtab + (size_t)lo * tb, *val = e->bw.d_wt_valid + lo, *buf = (uint8_t *)d_buf;
    hipError_t s;
    if (out) {
        s = hipMemcpy2DAsync(buf, stride, tab, tb, tb, n, hipMemcpyDeviceToDevice, e->stream);
        if (s == hipSuccess) s = hipMemcpy2DAsync(buf + tb, stride, val, 1, 1, n, hipMemcpyDeviceToDevice, e->stream);
    } else {
        s = hipMemcpy2DAsync(tab, tb, buf, stride, tb, n, hipMemcpyDeviceToDevice, e->stream);
        if (s == hipSuccess) s = hipMemcpy2DAsync(val, 1, buf + tb, stride, 1, n, hipMemcpyDeviceToDevice, e->stream);
    }
    if (s == hipSuccess && out) s = hipStreamSynchronize(e->stream);      // the caller hands the buffer to a collective on another stream
    if (s != hipSuccess) return ix->fail(HX_E_HIP, std::string("W table exchange: ") + hipGetErrorString(s));
    return HX_OK;
}
int hx_index_dbatch_export_wtabs(hx_index *ix, uint32_t lo, uint32_t hi, void *d_out) { return wtab_copy(ix, lo, hi, d_out, true); }
int hx_index_dbatch_import_wtabs(hx_index *ix, uint32_t lo, uint32_t hi, const void *d_in) { return wtab_copy(ix, lo, hi, (void *)d_in, false); }

int hx_index_dbatch_end(hx_index *ix, uint32_t *elem_out)
{
    if (!ix) return HX_E_ARG;
    BatchState &bs = ix->bs;
    if (!bs.open || !bs.dev || !bs.linked) return ix->fail(HX_E_STATE, "device batch not linked yet");
    if (hipStreamSynchronize(ix->e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "stream synchronisation failed");   // imports have landed before the caller reuses its buffers
    if (elem_out) memcpy(elem_out, bs.elem.data(), (size_t)bs.b * sizeof(uint32_t));
    bs.open = false; bs.linked = false; bs.dev = false; bs.lazy_lists = false; bs.b = 0;
    return HX_OK;
}

int hx_index_insert(hx_index *ix, uint64_t first_row, uint32_t n, const int32_t *levels, const int64_t *tids,
                    uint32_t batch, uint32_t *elem_out)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix) return HX_E_ARG;
    if (n == 0) return HX_OK;
    hx_index::Timer t_all(ix->prof[10]);
    if (!levels || !tids) return ix->fail(HX_E_ARG, "NULL argument");
    Graph &g = ix->g;
    if (ix->bs.open) return ix->fail(HX_E_STATE, "a staged batch is open");
    if (first_row != g.size()) return ix->fail(HX_E_STATE, "rows must be inserted in append order: first_row != index size");
    if (first_row + n > hx_num_rows(ix->e)) return ix->fail(HX_E_ARG, "rows not present in the engine");
    if (batch == 0) batch = 1;
    g.reserve(n);
    const int mxl = max_level_for(g.m);
    uint32_t done = 0;
    while (done < n) {
        // ramp-up: a batch never exceeds 1/8 of the graph it searches, so early batches (which would be a
        // large share of a tiny graph and cannot see each other) stay small.  Same rule as
        // pgvector-rx_amd/levels.py:batch_schedule, which the oracle-side tests follow.
        uint32_t b = std::min(std::min(batch, n - done), std::max<uint32_t>(1u, g.size() / 8u));
        // the very first element has nothing to search (build.rs:526-529); it still counts as a member of
        // its batch so that batch boundaries do not depend on whether the index was empty
        if (g.entry < 0) {
            int lv = std::min(levels[done], mxl); if (lv < 0) lv = 0;
            uint32_t id = g.add(lv);
            g.entry = id; g.tids[id][0] = tids[done]; g.ntids[id] = 1;
            if (elem_out) elem_out[done] = id;
            done++; b--;
            if (b == 0) continue;
        }
        int rc;
        if (hx_index_dbatch_supported(ix, levels + done, b)) {
            // device-resident batch: lists go from k_fused<insert> to the back-link kernels without visiting the host
            if ((rc = ix->e->db_reserve_records(b))) return ix->fail(rc, ix->e->err);
            if ((rc = hx_index_dbatch_begin(ix, first_row + done, b, levels + done, tids + done))) return rc;
            if ((rc = hx_index_dbatch_search(ix, 0, b, ix->e->bw.d_rec)) || (rc = hx_index_dbatch_links(ix, 0, 1, ix->e->bw.d_rec, nullptr)) ||
                (rc = hx_index_dbatch_end(ix, elem_out ? elem_out + done : nullptr))) { ix->rollback_batch(); return rc; }
            done += b;
            continue;
        }
        ix->in_insert = true;
        rc = hx_index_batch_begin(ix, first_row + done, b, levels + done, tids + done);
        ix->in_insert = false;
        if (rc) return rc;
        ix->bs.lazy_lists = true;
        if ((rc = hx_index_batch_search(ix, 0, b)) || (rc = hx_index_batch_links(ix, 0, 1)) ||
            (rc = hx_index_batch_end(ix, elem_out ? elem_out + done : nullptr))) { ix->rollback_batch(); return rc; }
        done += b;
    }
    return HX_OK;
}


// ================================================================================================
// f3: aminsert (src/index/insert.rs:1227-1480) and vacuum (src/index/vacuum.rs) on the engine
// ================================================================================================
namespace {
// write_neighbor_update insert.rs:793-871
void write_neighbor_update(hx_index *ix, uint32_t n, int layer, uint32_t new_id, float new_d, int update_idx)
{
    Graph &g = ix->g;
    Cand *lst = g.list(n, layer); uint16_t &cnt = g.cnt(n, layer); const int lm = g.lm(layer);
    for (uint16_t i = 0; i < cnt; i++) if (lst[i].id == new_id) return;            // connection already exists
    if (update_idx == -2) { if (cnt < lm) { lst[cnt++] = Cand{new_d, new_id}; ix->dirty.emplace_back(n, layer); } }
    else if (update_idx >= 0 && update_idx < (int)cnt) { lst[update_idx] = Cand{new_d, new_id}; ix->dirty.emplace_back(n, layer); }
}
}  // namespace

int hx_index_insert_ondisk(hx_index *ix, uint64_t first_row, uint32_t n, const int32_t *levels, const int64_t *tids, uint32_t batch, uint32_t *elem_out)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix) return HX_E_ARG;
    if (n == 0) return HX_OK;
    if (!levels || !tids) return ix->fail(HX_E_ARG, "NULL argument");
    Graph &g = ix->g;
    if (ix->bs.open) return ix->fail(HX_E_STATE, "a staged batch is open");
    if (first_row != g.size()) return ix->fail(HX_E_STATE, "rows must be inserted in append order: first_row != index size");
    if (first_row + n > hx_num_rows(ix->e)) return ix->fail(HX_E_ARG, "rows not present in the engine");
    const bool big = 2 * g.m > (int)HX_PAIR_MAX_ROWS;                               // lists of more than 64 slots: device kernels only (hx_biglist.hip)
    if (big && !(ix->fused && ix->e->pitch <= 8192))
        return ix->fail(HX_E_ARG, "m > 32: the on-disk insert path runs in the device kernels only (rows <= 8 KiB, hx_index_set_fused(1))");
    if (batch == 0) batch = 1;
    hx_index::Timer t_all(ix->prof[10]);
    int rc = ix->ensure_host_lists();
    if (rc) return rc;
    const int mxl = max_level_for(g.m);
    const bool any_deleted = std::find(g.deleted.begin(), g.deleted.end(), (uint8_t)1) != g.deleted.end();
    const bool dead_reachable = any_deleted && (ix->dead_refs || (g.entry >= 0 && g.deleted[g.entry]));   // deleted elements nobody links to are never met
    bool any_unlinkable = dead_reachable;                                           // an element get_update_index would give up a slot for (insert.rs:566-625)
    for (uint32_t i = 0; i < g.size() && !any_unlinkable; i++) if (g.level[i] >= 0 && g.ntids[i] == 0 && !g.deleted[i]) any_unlinkable = true;
    if (big && any_unlinkable) return ix->fail(HX_E_STATE, "m > 32: the on-disk insert path needs an index whose deleted elements are unlinked (VACUUM to completion first)");
    uint32_t done = 0;
    while (done < n) {
        if (g.entry < 0) {                                                          // first element: insert.rs:1320-1338 (no entry point yet)
            int lv = std::min(levels[done], mxl); if (lv < 0) lv = 0;
            const uint32_t id = g.add(lv);
            g.entry = id; g.tids[id][0] = tids[done]; g.ntids[id] = 1;
            if (elem_out) elem_out[done] = id;
            done++;
            continue;
        }
        const uint32_t b = std::min(batch, n - done), base = g.size();
        const uint32_t entry = (uint32_t)g.entry; const int entry_level = g.level[entry];
        // stage 1: every member's neighbour search against the graph as it stands (batch == 1: the reference's one insert at a time;
        // batch > 1: what concurrent backends do -- each searches without seeing the others' uncommitted elements, e.g. 013's 10 pgbench clients).
        // Device-resident placement: find_element_neighbors_on_disk (insert.rs:1021-1123) is the traversal kernel's MODE 3 -- the greedy descent and a
        // search_layer per layer, the sorted result set W of every layer handed out -- and its "filtered.iter().rev().take(lm)" (:1111-1117) the first lm
        // entries of W.  (An index in which a deleted element can still be reached -- load_element skips those, scan.rs:178-181; hx_index::dead_refs -- and what
        // the kernel does not serve go through the lock-step driver, as does every member whose tables overflow.)
        std::vector<std::vector<std::vector<Cand>>> nbs(b);                          // [member][layer]: nearest first
        std::vector<int> mlv(b);
        for (uint32_t i = 0; i < b; i++) { int lv = std::min(levels[done + i], mxl); if (lv < 0) lv = 0; mlv[i] = lv; nbs[i].assign(lv + 1, {}); }
        std::vector<uint32_t> ls_members;                                            // members for the lock-step driver
        if (ix->fused_scan_ok() && !dead_reachable && b >= 1) {
            if ((rc = ix->sync_mirror())) return rc;
            hx_engine *e = ix->e;
            std::vector<uint32_t> qsel(b), prob(b), tstat(b); std::vector<int32_t> tl(b);
            uint32_t P = 0;
            for (uint32_t i = 0; i < b; i++) { qsel[i] = base + i; tl[i] = mlv[i]; prob[i] = P; P += (uint32_t)std::min(mlv[i], entry_level) + 1u; }
            if ((rc = e->wsel_reserve(P, (uint32_t)ix->efc)) || (rc = e->db_reserve_records(b))) return ix->fail(rc, e->err);
            HxWselWork &w = e->wsel;
            if (hipMemsetAsync(w.d_cnt, 0, (size_t)P * 4, e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "on-disk insert: clearing the W counts failed");
            HxFusedDev dev; dev.d_rec = e->bw.d_rec; dev.rec_words = hx_rec_words((uint32_t)g.m); dev.h_slots = nullptr;
            dev.d_wl_out = w.d_wl; dev.d_wl_cnt = w.d_cnt; dev.h_prob = prob.data(); dev.ondisk = true;
            uint64_t cnts[2] = {0, 0};
            const double t0 = hx_index::now_s();
            if ((rc = e->fused_run(3, b, qsel.data(), tl.data(), (uint32_t)ix->efc, 0, entry, entry_level,
                                   nullptr, nullptr, nullptr, tstat.data(), cnts, nullptr, nullptr, 1, &dev))) return ix->fail(rc, e->err);
            ix->prof[6] += hx_index::now_s() - t0;
            ix->counters[4] += cnts[0];
            std::vector<uint32_t> wcnt(P); std::vector<uint2> wl((size_t)P * ix->efc);
            if (hipMemcpyAsync(wcnt.data(), w.d_cnt, (size_t)P * 4, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                hipMemcpyAsync(wl.data(), w.d_wl, (size_t)P * ix->efc * 8, hipMemcpyDeviceToHost, e->stream) != hipSuccess ||
                hipStreamSynchronize(e->stream) != hipSuccess) return ix->fail(HX_E_HIP, "on-disk insert: reading the W lists failed");
            for (uint32_t i = 0; i < b; i++) {
                if (tstat[i] != 0) { ls_members.push_back(i); continue; }
                const int start = std::min(mlv[i], entry_level);
                for (int lc = 0; lc <= start; lc++) {
                    const size_t pr = (size_t)prob[i] + (size_t)lc, lm = (size_t)g.lm(lc);
                    const uint2 *W = wl.data() + pr * ix->efc;                       // nearest LAST (scan.rs:441-446)
                    std::vector<Cand> &out = nbs[i][lc];
                    for (size_t k = wcnt[pr]; k-- > 0 && out.size() < lm;) { Cand c; memcpy(&c.d, &W[k].x, 4); c.id = W[k].y; out.push_back(c); }   // filtered.iter().rev().take(lm), insert.rs:1111-1117
                }
            }
            ix->fused_tasks += b; ix->fused_redo += ls_members.size();
        } else {
            for (uint32_t i = 0; i < b; i++) ls_members.push_back(i);
        }
        if (!ls_members.empty()) {
            auto &dts = ix->disk_pool;
            while (dts.size() < ls_members.size()) dts.emplace_back(new DiskNeighborsTask());
            std::vector<LsTask *> tasks(ls_members.size());
            for (size_t k = 0; k < ls_members.size(); k++) {
                DiskNeighborsTask &t = *dts[k];
                t.reset();
                t.g = &g; t.query_sel = base + ls_members[k]; t.new_level = mlv[ls_members[k]]; t.entry = entry; t.entry_level = entry_level; t.efc = ix->efc;
                t.skip = nullptr; t.skip_self = 0xFFFFFFFFu; t.repair = false;
                tasks[k] = &t;
            }
            if ((rc = ix->run_lockstep(tasks))) return rc;
            for (size_t k = 0; k < ls_members.size(); k++) {
                DiskNeighborsTask &t = *dts[k];
                ix->counters[4] += t.n_dist;
                for (int lc = 0; lc <= t.new_level && lc < (int)t.nb.size(); lc++) nbs[ls_members[k]][lc] = t.nb[lc];
            }
        }
        // duplicate candidates: leading zero-distance layer-0 neighbours, compared byte for byte (find_duplicate_on_disk insert.rs:1180-1214)
        const double t_2a = hx_index::now_s();
        std::vector<uint32_t> da, db; std::vector<uint32_t> dstart(b + 1, 0u);
        for (uint32_t i = 0; i < b; i++) {
            if (!nbs[i].empty()) for (const Cand &c : nbs[i][0]) { if (c.d != 0.0f) break; da.push_back(base + i); db.push_back(c.id); }
            dstart[i + 1] = (uint32_t)da.size();
        }
        std::vector<uint8_t> deq(da.size(), 0);
        if (!da.empty() && (rc = hx_rows_equal(ix->e, (uint32_t)da.size(), da.data(), db.data(), deq.data()))) return ix->fail(rc, ix->e->err);
        // stage 2a: members in order -- duplicate merge, or the element, its own lists, the entry point; its back-connections are listed
        struct UOp { uint32_t nbr; int layer; uint32_t id; float d; };
        std::vector<UOp> ops;
        for (uint32_t i = 0; i < b; i++) {
            const uint32_t id = g.add(mlv[i]);                                      // element id == row id, also for a merged row (tombstone)
            int64_t dup = -1;
            for (uint32_t k = dstart[i]; k < dstart[i + 1]; k++)
                if (deq[k] && g.ntids[db[k]] > 0 && g.ntids[db[k]] < HEAPTIDS && !g.deleted[db[k]]) { dup = db[k]; break; }   // add_duplicate_on_disk :1136-1171
            if (dup >= 0) {
                g.tids[dup][g.ntids[dup]++] = tids[done + i];
                g.level[id] = -1 - g.level[id];
                if (elem_out) elem_out[done + i] = (uint32_t)dup;
                continue;
            }
            for (int lc = 0; lc <= mlv[i]; lc++) {                                  // the new element's neighbour tuple, insert.rs:1384-1410
                Cand *lst = g.list(id, lc); const size_t c = std::min(nbs[i][lc].size(), (size_t)g.lm(lc));
                for (size_t k = 0; k < c; k++) lst[k] = nbs[i][lc][k];
                g.cnt(id, lc) = (uint16_t)c;
            }
            g.tids[id][0] = tids[done + i]; g.ntids[id] = 1;
            ix->mark_dirty(id);
            for (int lc = mlv[i]; lc >= 0; lc--) {                                  // update_neighbors_on_disk insert.rs:883-958: one get_update_index per (neighbour, layer)
                const size_t lm = (size_t)g.lm(lc);
                for (size_t k = 0; k < nbs[i][lc].size() && k < lm; k++) ops.push_back(UOp{nbs[i][lc][k].id, lc, id, nbs[i][lc][k].d});
            }
            if (mlv[i] > g.level[g.entry]) g.entry = id;                            // insert.rs:1453-1470 (HNSW_UPDATE_ENTRY_GREATER)
            if (elem_out) elem_out[done + i] = id;
        }
        // stage 2b: get_update_index reads and write_neighbor_update writes ONE list (and row data, deleted flags and heap-TID counts, which the
        // back-connections do not change), so updates of different lists commute: the batch's updates run as waves -- wave k holds the k-th update of
        // every list, in member order per list -- one lock-step round per wave instead of one per member.  batch == 1 is the reference's order exactly.
        ix->prof[8] += hx_index::now_s() - t_2a;
        hx_index::Timer t_2b(ix->prof[9]);
        if (!ops.empty()) {
            std::vector<uint32_t> order(ops.size());
            for (uint32_t k = 0; k < ops.size(); k++) order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
                return ops[x].nbr != ops[y].nbr ? ops[x].nbr < ops[y].nbr : ops[x].layer < ops[y].layer; });
            std::vector<std::pair<uint32_t, uint32_t>> runs;                          // [first, end) of every list's updates in `order`
            for (uint32_t k = 0; k < order.size();) {
                uint32_t j = k + 1;
                while (j < order.size() && ops[order[j]].nbr == ops[order[k]].nbr && ops[order[j]].layer == ops[order[k]].layer) j++;
                runs.emplace_back(k, j); k = j;
            }
            auto &uts = ix->update_pool;
            // Device placement of a wave (round 3): the host settles what needs no distance (insert.rs:524-527, 556-559, 566-625) and k_update_index
            // (hx_links.hip) answers the full lists, one wavefront per op, from the list contents the host hands over.
            const bool dev_upd = ix->fused && ix->e->pitch <= 8192;
            std::vector<uint32_t> wops, devq; std::vector<int> wres;
            // no deleted / TID-less element in the index: get_update_index's early exits cannot fire, and k_update_runs applies a list's whole run in one go
            bool runs_done = false;
            if (dev_upd && !any_unlinkable) {
                const uint32_t nr = (uint32_t)runs.size(), stride = 2u * (uint32_t)g.m;
                uint32_t *h_ids, *h_cnt, *h_lm, *h_off, *h_new; float *h_d, *h_od;
                if ((rc = big ? ix->e->biglist_ops_stage(nr, (uint32_t)ops.size(), stride, &h_ids, &h_d, &h_cnt, &h_lm, &h_off, &h_new, &h_od)
                              : ix->e->update_runs_stage(nr, (uint32_t)ops.size(), stride, &h_ids, &h_d, &h_cnt, &h_lm, &h_off, &h_new, &h_od))) return ix->fail(rc, ix->e->err);
                ix->pool->parallel_for((nr + 1023) / 1024, [&](size_t ci) {
                    for (size_t k = ci * 1024; k < std::min<size_t>(nr, ci * 1024 + 1024); k++) {
                        const UOp &o0 = ops[order[runs[k].first]];
                        const Cand *lst = g.list(o0.nbr, o0.layer); const uint32_t c = g.cnt(o0.nbr, o0.layer);
                        for (uint32_t i = 0; i < c; i++) { h_ids[k * stride + i] = lst[i].id; h_d[k * stride + i] = lst[i].d; }
                        h_cnt[k] = c; h_lm[k] = (uint32_t)g.lm(o0.layer); h_off[k] = runs[k].first;
                        for (uint32_t q = runs[k].first; q < runs[k].second; q++) { h_new[q] = ops[order[q]].id; h_od[q] = ops[order[q]].d; }
                    }
                });
                h_off[nr] = (uint32_t)ops.size();
                uint64_t np = 0;
                const double t_k = hx_index::now_s();
                if ((rc = big ? ix->e->biglist_ops_run(&np, true) : ix->e->update_runs_run(&np))) return ix->fail(rc, ix->e->err);
                ix->prof[3] += hx_index::now_s() - t_k; ix->prof[5] += 1.0;
                ix->counters[3] += np;
                ix->pool->parallel_for((nr + 1023) / 1024, [&](size_t ci) {
                    for (size_t k = ci * 1024; k < std::min<size_t>(nr, ci * 1024 + 1024); k++) {
                        const UOp &o0 = ops[order[runs[k].first]];
                        Cand *lst = g.list(o0.nbr, o0.layer); const uint32_t c = std::min<uint32_t>(h_cnt[k], (uint32_t)g.lm(o0.layer));
                        for (uint32_t i = 0; i < c; i++) lst[i] = Cand{h_d[k * stride + i], h_ids[k * stride + i]};
                        g.cnt(o0.nbr, o0.layer) = (uint16_t)c;
                    }
                });
                for (uint32_t k = 0; k < nr; k++) ix->dirty.emplace_back(ops[order[runs[k].first]].nbr, ops[order[runs[k].first]].layer);
                runs_done = true;
            }
            for (uint32_t wave = 0; dev_upd && !runs_done; wave++) {
                wops.clear(); devq.clear();
                for (const auto &r : runs) if (r.first + wave < r.second) wops.push_back(order[r.first + wave]);
                if (wops.empty()) break;
                wres.assign(wops.size(), -4);                                        // -4: the device decides
                for (size_t k = 0; k < wops.size(); k++) {
                    const UOp &o = ops[wops[k]];
                    const Cand *lst = g.list(o.nbr, o.layer); const size_t cnt = g.cnt(o.nbr, o.layer), lm = (size_t)g.lm(o.layer);
                    if (g.deleted[o.nbr]) { wres[k] = -3; continue; }
                    if (cnt < lm) { wres[k] = -2; continue; }
                    for (size_t i = 0; i < cnt; i++) if (g.deleted[lst[i].id] || g.ntids[lst[i].id] == 0) { wres[k] = (int)i; break; }
                    if (wres[k] == -4) devq.push_back((uint32_t)k);
                }
                if (!devq.empty()) {
                    const uint32_t nd = (uint32_t)devq.size(), stride = 2u * (uint32_t)g.m;
                    uint32_t *h_ids = nullptr, *h_cnt = nullptr; float *h_d = nullptr, *h_nd = nullptr;
                    if ((rc = ix->e->update_index_stage(nd, stride, &h_ids, &h_d, &h_nd, &h_cnt))) return ix->fail(rc, ix->e->err);
                    ix->pool->parallel_for((nd + 1023) / 1024, [&](size_t ci) {
                        for (uint32_t j = (uint32_t)ci * 1024; j < std::min<uint32_t>(nd, (uint32_t)ci * 1024 + 1024); j++) {
                            const UOp &o = ops[wops[devq[j]]];
                            const Cand *lst = g.list(o.nbr, o.layer); const uint32_t cnt = g.cnt(o.nbr, o.layer);
                            for (uint32_t i = 0; i < cnt; i++) { h_ids[(size_t)j * stride + i] = lst[i].id; h_d[(size_t)j * stride + i] = lst[i].d; }
                            h_cnt[j] = cnt; h_nd[j] = o.d;
                        }
                    });
                    const int32_t *slot = nullptr; uint64_t np = 0;
                    const double t_k = hx_index::now_s();
                    if ((rc = ix->e->update_index_run(&slot, &np))) return ix->fail(rc, ix->e->err);
                    ix->prof[3] += hx_index::now_s() - t_k; ix->prof[5] += 1.0;
                    ix->counters[3] += np;
                    for (uint32_t j = 0; j < nd; j++) wres[devq[j]] = slot[j];
                }
                for (size_t k = 0; k < wops.size(); k++) {
                    if (wres[k] == -3) continue;
                    const UOp &o = ops[wops[k]];
                    write_neighbor_update(ix, o.nbr, o.layer, o.id, o.d, wres[k]);
                }
            }
            for (uint32_t wave = 0; !dev_upd && !runs_done; wave++) {
                std::vector<LsTask *> utasks; std::vector<uint32_t> which;
                for (const auto &r : runs) {
                    if (r.first + wave >= r.second) continue;
                    const UOp &o = ops[order[r.first + wave]];
                    if (utasks.size() == uts.size()) uts.emplace_back(new UpdateIndexTask());
                    UpdateIndexTask &u = *uts[utasks.size()];
                    u.reset(); u.g = &g; u.nbr = o.nbr; u.layer = o.layer; u.new_d = o.d;
                    utasks.push_back(&u); which.push_back(order[r.first + wave]);
                }
                if (utasks.empty()) break;
                if ((rc = ix->run_lockstep(utasks))) return rc;
                for (size_t k = 0; k < utasks.size(); k++) {
                    UpdateIndexTask &u = *uts[k];
                    ix->counters[3] += u.n_dist + u.n_pair;
                    if (u.result == -3) continue;
                    const UOp &o = ops[which[k]];
                    write_neighbor_update(ix, o.nbr, o.layer, o.id, o.d, u.result);
                }
            }
        }
        done += b;
    }
    return HX_OK;
}

// vacuum.rs: pass 1 remove_heap_tids, pass 2 repair_graph, pass 3 mark_deleted.  batch == 1: elements are repaired one after the other as the
// reference does; batch > 1: that many repair searches run in lock-step against the same state of the graph.
int hx_index_vacuum(hx_index *ix, const int64_t *dead_tids, uint64_t n_dead, uint32_t batch, uint64_t *n_deleted_out, uint64_t *n_repaired_out)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix || (!dead_tids && n_dead)) return HX_E_ARG;
    Graph &g = ix->g;
    if (ix->bs.open) return ix->fail(HX_E_STATE, "a staged batch is open");
    if (batch == 0) batch = 1;
    int rc = ix->ensure_host_lists();
    if (rc) return rc;
    std::vector<int64_t> dead(dead_tids, dead_tids + n_dead);
    std::sort(dead.begin(), dead.end());
    const uint32_t n = g.size();
    std::vector<uint8_t> del(n, 0);
    int64_t highest = -1; int highest_level = -1;
    for (uint32_t e = 0; e < n; e++) {                                              // pass 1, vacuum.rs:118-217
        if (g.level[e] < 0) continue;                                               // merged duplicate: has no tuple
        if (g.ntids[e] > 0) {
            uint8_t k = 0;
            for (uint8_t i = 0; i < g.ntids[e]; i++) if (!std::binary_search(dead.begin(), dead.end(), g.tids[e][i])) g.tids[e][k++] = g.tids[e][i];
            g.ntids[e] = k;
        }
        if (g.ntids[e] == 0) del[e] = 1;
        else if (g.level[e] > highest_level && (int64_t)e != g.entry) { highest = e; highest_level = g.level[e]; }
    }
    auto needs_updated = [&](uint32_t e) {                                          // vacuum.rs:230-285
        for (int lc = 0; lc <= g.level[e]; lc++) { const Cand *l = g.list(e, lc); for (uint16_t k = 0; k < g.cnt(e, lc); k++) if (del[l[k].id]) return true; }
        return g.cnt(e, 0) < (uint16_t)g.lm(0);
    };
    uint64_t n_rep = 0;
    auto &dts = ix->disk_pool;
    // repairs `elems` (each against `entries[i]`) in lock-step, then overwrites their neighbour tuples (vacuum.rs:288-407)
    // Device placement of the repair searches (round 3): k_fused MODE 3 with search_layer_disk's semantics, the skip set (the dead elements + the repaired
    // element itself: traversed, not counted, not selected; ef_construction + 1) and a per-task entry point; the lm nearest of every layer's W that are
    // not in the skip set become the element's new lists.  Used while no element deleted by an EARLIER vacuum can be reached (load_element would skip
    // those, scan.rs:178-181: the closing check of the vacuum that deleted them found no list naming one -- hx_index::dead_refs), for rows the kernel
    // serves.
    const bool dev_repair = ix->fused_scan_ok() && (!ix->dead_refs || std::find(g.deleted.begin(), g.deleted.end(), (uint8_t)1) == g.deleted.end());
    ix->dead_refs = true;                                                           // until the closing check below says otherwise
    uint8_t *d_skip = nullptr;
    struct SkipFree { uint8_t *&p; ~SkipFree() { if (p) (void)hipFree(p); } } skip_free{d_skip};
    if (dev_repair && n) {
        if (hipSetDevice(ix->e->device) != hipSuccess || hipMalloc((void **)&d_skip, n) != hipSuccess ||
            hipMemcpy(d_skip, del.data(), n, hipMemcpyHostToDevice) != hipSuccess) return ix->fail(HX_E_HIP, "vacuum: uploading the skip set failed");
    }
    auto repair_on_device = [&](const std::vector<uint32_t> &elems, const std::vector<int64_t> &entries, std::vector<uint32_t> &left, std::vector<int64_t> &left_ent) -> int {
        std::vector<uint32_t> who, qsel, prob, h_entry; std::vector<int32_t> tl;
        uint32_t P = 0;
        for (size_t i = 0; i < elems.size(); i++) {
            if (entries[i] < 0 || (int64_t)elems[i] == entries[i]) continue;       // vacuum.rs:300-303
            const uint32_t e = elems[i], ent = (uint32_t)entries[i];
            if (g.deleted[ent]) {                                                   // load_element(entry) -> None: every list comes out empty (insert.rs:1037-1048)
                for (int lc = 0; lc <= g.level[e]; lc++) g.cnt(e, lc) = 0;
                ix->mark_dirty(e); n_rep++;
                continue;
            }
            who.push_back(e); qsel.push_back(e); tl.push_back(g.level[e]); h_entry.push_back(ent); prob.push_back(P);
            P += (uint32_t)std::min(g.level[e], g.level[ent]) + 1u;
        }
        if (who.empty()) return HX_OK;
        int r = ix->sync_mirror();
        if (r) return r;
        hx_engine *en = ix->e;
        const uint32_t nt = (uint32_t)who.size(), ef = (uint32_t)ix->efc + 1u, wst = ef + 2u + 254u;     // insert.rs:1081-1086; wst: the W lists' stride (hx_fused.inc.h: wcap of a repair launch)
        if ((r = en->wsel_reserve(P, wst)) || (r = en->db_reserve_records(nt))) return ix->fail(r, en->err);
        HxWselWork &w = en->wsel;
        if (hipMemsetAsync(w.d_cnt, 0, (size_t)P * 4, en->stream) != hipSuccess) return ix->fail(HX_E_HIP, "vacuum: clearing the W counts failed");
        HxFusedDev dev; dev.d_rec = en->bw.d_rec; dev.rec_words = hx_rec_words((uint32_t)g.m); dev.h_slots = nullptr;
        dev.d_wl_out = w.d_wl; dev.d_wl_cnt = w.d_cnt; dev.h_prob = prob.data(); dev.ondisk = true; dev.h_entry = h_entry.data(); dev.d_skip = d_skip;
        std::vector<uint32_t> tstat(nt); uint64_t cnts[2] = {0, 0};
        if ((r = en->fused_run(3, nt, qsel.data(), tl.data(), ef, 0, (uint32_t)h_entry[0], g.level[h_entry[0]],
                               nullptr, nullptr, nullptr, tstat.data(), cnts, nullptr, nullptr, 1, &dev))) return ix->fail(r, en->err);
        ix->counters[4] += cnts[0];
        std::vector<uint32_t> wcnt(P); std::vector<uint2> wl((size_t)P * wst);
        if (hipMemcpyAsync(wcnt.data(), w.d_cnt, (size_t)P * 4, hipMemcpyDeviceToHost, en->stream) != hipSuccess ||
            hipMemcpyAsync(wl.data(), w.d_wl, (size_t)P * wst * 8, hipMemcpyDeviceToHost, en->stream) != hipSuccess ||
            hipStreamSynchronize(en->stream) != hipSuccess) return ix->fail(HX_E_HIP, "vacuum: reading the W lists failed");
        for (uint32_t k = 0; k < nt; k++) {
            const uint32_t e = who[k];
            if (tstat[k] != 0) { left.push_back(e); left_ent.push_back((int64_t)h_entry[k]); continue; }   // outgrew its tables: the lock-step driver
            const int start = std::min(g.level[e], g.level[h_entry[k]]);
            for (int lc = 0; lc <= g.level[e]; lc++) {
                Cand *lst = g.list(e, lc); size_t c = 0; const size_t lm = (size_t)g.lm(lc);
                if (lc <= start) {
                    const size_t pr = (size_t)prob[k] + (size_t)lc; const uint2 *W = wl.data() + pr * wst;   // nearest LAST
                    for (size_t j = wcnt[pr]; j-- > 0 && c < lm;) {                  // filtered.iter().rev().take(lm), insert.rs:1103-1117
                        if (W[j].y == e || del[W[j].y]) continue;
                        memcpy(&lst[c].d, &W[j].x, 4); lst[c].id = W[j].y; c++;
                    }
                }
                g.cnt(e, lc) = (uint16_t)c;
            }
            ix->mark_dirty(e); n_rep++;
        }
        ix->fused_tasks += nt; ix->fused_redo += left.size();
        return HX_OK;
    };
    auto repair = [&](const std::vector<uint32_t> &elems_in, const std::vector<int64_t> &entries_in) -> int {
        std::vector<uint32_t> elems_l; std::vector<int64_t> entries_l;
        if (dev_repair) {
            int r0 = repair_on_device(elems_in, entries_in, elems_l, entries_l);
            if (r0) return r0;
            if (elems_l.empty()) return HX_OK;
        }
        const std::vector<uint32_t> &elems = dev_repair ? elems_l : elems_in;
        const std::vector<int64_t> &entries = dev_repair ? entries_l : entries_in;
        std::vector<LsTask *> tasks; std::vector<uint32_t> who;
        while (dts.size() < elems.size()) dts.emplace_back(new DiskNeighborsTask());
        for (size_t i = 0; i < elems.size(); i++) {
            if (entries[i] < 0 || (int64_t)elems[i] == entries[i]) continue;       // vacuum.rs:300-303
            DiskNeighborsTask &t = *dts[tasks.size()];
            t.reset();
            t.g = &g; t.query_sel = elems[i]; t.new_level = g.level[elems[i]]; t.entry = (uint32_t)entries[i]; t.entry_level = g.level[entries[i]]; t.efc = ix->efc;
            t.skip = del.data(); t.skip_self = elems[i]; t.repair = true;
            tasks.push_back(&t); who.push_back(elems[i]);
        }
        int r = ix->run_lockstep(tasks);
        if (r) return r;
        for (size_t i = 0; i < who.size(); i++) {
            DiskNeighborsTask &t = *static_cast<DiskNeighborsTask *>(tasks[i]);
            const uint32_t e = who[i];
            ix->counters[4] += t.n_dist;
            for (int lc = 0; lc <= g.level[e]; lc++) {
                Cand *lst = g.list(e, lc); const size_t c = lc < (int)t.nb.size() ? std::min(t.nb[lc].size(), (size_t)g.lm(lc)) : 0;
                for (size_t k = 0; k < c; k++) lst[k] = t.nb[lc][k];
                g.cnt(e, lc) = (uint16_t)c;
            }
            ix->mark_dirty(e);
            n_rep++;
        }
        return HX_OK;
    };
    // pass 2: the entry point first (vacuum.rs:411-520)
    if (highest >= 0 && !g.deleted[highest] && needs_updated((uint32_t)highest) && (rc = repair({(uint32_t)highest}, {g.entry}))) return rc;
    if (g.entry >= 0) {
        if (del[g.entry]) g.entry = highest;
        else if (!g.deleted[g.entry] && needs_updated((uint32_t)g.entry) && (rc = repair({(uint32_t)g.entry}, {highest >= 0 ? highest : g.entry}))) return rc;
    }
    std::vector<uint32_t> grp; std::vector<int64_t> ent;
    auto flush = [&]() -> int { if (grp.empty()) return HX_OK; int r = repair(grp, ent); grp.clear(); ent.clear(); return r; };
    for (uint32_t e = 0; e < n; e++) {                                              // vacuum.rs:540-640
        if (g.level[e] < 0 || g.ntids[e] == 0 || g.deleted[e]) continue;
        if (!needs_updated(e)) continue;
        if (g.entry < 0 || g.level[e] > g.level[g.entry]) {                         // may become the entry point: on its own, in order
            if ((rc = flush())) return rc;
            if ((rc = repair({e}, {g.entry}))) return rc;
            if (g.entry < 0 || g.level[e] > g.level[g.entry]) g.entry = e;
            continue;
        }
        grp.push_back(e); ent.push_back(g.entry);
        if (grp.size() >= batch && (rc = flush())) return rc;
    }
    if ((rc = flush())) return rc;
    uint64_t n_del = 0;
    for (uint32_t e = 0; e < n; e++) {                                              // pass 3, vacuum.rs:655-793
        if (g.level[e] < 0 || g.deleted[e] || g.ntids[e] > 0) continue;
        for (int lc = 0; lc <= g.level[e]; lc++) g.cnt(e, lc) = 0;
        g.deleted[e] = 1; ix->mark_dirty(e); n_del++;
    }
    if (n_deleted_out) *n_deleted_out = n_del;
    if (n_repaired_out) *n_repaired_out = n_rep;
    {   // closing check: does any live element still name a deleted one?  (only an entry point vacuum.rs:300-303 could not repair)
        std::atomic<bool> refs{false};
        ix->pool->parallel_for((n + 4095) / 4096, [&](size_t ci) {
            for (uint32_t e = (uint32_t)ci * 4096; e < std::min<uint32_t>(n, (uint32_t)ci * 4096 + 4096); e++) {
                if (g.level[e] < 0 || g.deleted[e]) continue;
                for (int lc = 0; lc <= g.level[e]; lc++) { const Cand *l = g.list(e, lc); for (uint16_t k = 0; k < g.cnt(e, lc); k++) if (g.deleted[l[k].id]) refs.store(true, std::memory_order_relaxed); }
            }
        });
        ix->dead_refs = refs.load();
    }
    return HX_OK;
}


// f2: version-checked invalidation of an index loaded from its page image.  The scan side trusts a neighbour tuple only while its version
// equals the element's (load_neighbor_tids, scan.rs:262-265; HnswElementTupleData.version, types/hnsw.rs:120: vacuum bumps it when it frees
// a tuple, so a reused slot no longer matches).  When the host changes pages under a loaded mirror it reports the tuples it touched:
// every listed (blkno, offno) whose CURRENT version differs from the one loaded (versions == NULL: unconditionally) is dropped from the
// mirror -- no heap TIDs, no lists, unlinked from every neighbour list (order of the survivors kept), entry point re-picked like
// vacuum does (highest remaining level) -- so that device scans never return or traverse a tuple whose slot now holds something else.
int hx_index_invalidate(hx_index *ix, uint32_t n, const uint32_t *blkno, const uint16_t *offno, const uint8_t *versions, uint32_t *n_dropped_out)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix || (n && (!blkno || !offno))) return HX_E_ARG;
    if (n_dropped_out) *n_dropped_out = 0;
    Graph &g = ix->g;
    if (ix->loc2elem.empty()) return ix->fail(HX_E_STATE, "the index was not loaded from pages (hx_index_load_pages)");
    int rc = ix->ensure_host_lists();
    if (rc) return rc;
    std::vector<uint8_t> drop(g.size(), 0); uint32_t nd = 0;
    for (uint32_t i = 0; i < n; i++) {
        auto it = ix->loc2elem.find(((uint64_t)blkno[i] << 16) | offno[i]);
        if (it == ix->loc2elem.end()) continue;                                   // a tuple the mirror never held
        const uint32_t e = it->second;
        if (g.deleted[e] || drop[e]) continue;
        if (versions && versions[i] == ix->loc_ver[e]) continue;                  // still the tuple that was loaded
        drop[e] = 1; nd++;
    }
    if (n_dropped_out) *n_dropped_out = nd;
    if (!nd) return HX_OK;
    const uint32_t N = g.size();
    for (uint32_t e = 0; e < N; e++) {
        if (g.level[e] < 0) continue;
        if (drop[e]) {
            for (int lc = 0; lc <= g.level[e]; lc++) g.cnt(e, lc) = 0;
            g.ntids[e] = 0; g.deleted[e] = 1; ix->mark_dirty(e);
            continue;
        }
        for (int lc = 0; lc <= g.level[e]; lc++) {
            Cand *l = g.list(e, lc); uint16_t &c = g.cnt(e, lc); uint16_t w = 0;
            for (uint16_t k = 0; k < c; k++) if (!drop[l[k].id]) l[w++] = l[k];
            if (w != c) { c = w; ix->dirty.emplace_back(e, lc); }
        }
    }
    if (g.entry >= 0 && drop[g.entry]) {
        int64_t best = -1;
        for (uint32_t e = 0; e < N; e++) if (g.level[e] >= 0 && !g.deleted[e] && g.ntids[e] > 0 && (best < 0 || g.level[e] > g.level[best])) best = e;
        g.entry = best;
    }
    return HX_OK;
}

int hx_index_deleted(const hx_index *ix, uint32_t elem) { if (!ix || elem >= ix->g.size()) return HX_E_ARG; return ix->g.deleted[elem]; }

uint32_t hx_index_size(const hx_index *ix) { return ix ? ix->g.size() : 0; }
int64_t hx_index_entry(const hx_index *ix) { return ix ? ix->g.entry : -1; }
int hx_index_level(const hx_index *ix, uint32_t elem) { if (!ix || elem >= ix->g.size()) return HX_E_ARG - 1000; return ix->g.level[elem]; }

int hx_index_neighbors(const hx_index *ix, uint32_t elem, int layer, uint32_t *ids_out, float *dist_out)
{
    if (ix && const_cast<hx_index *>(ix)->ensure_host_lists()) return HX_E_HIP - 1000;
    if (!ix || elem >= ix->g.size()) return HX_E_ARG;
    const Graph &g = ix->g;
    const int lv = g.level[elem] < 0 ? -1 - g.level[elem] : g.level[elem];
    if (layer < 0 || layer > lv) return HX_E_ARG;
    const Cand *l = g.list(elem, layer); const uint16_t n = g.cnt(elem, layer);
    for (uint16_t k = 0; k < n; k++) { if (ids_out) ids_out[k] = l[k].id; if (dist_out) dist_out[k] = l[k].d; }
    return n;
}

int hx_index_heaptids(const hx_index *ix, uint32_t elem, int64_t *tids_out)
{
    if (!ix || elem >= ix->g.size()) return HX_E_ARG;
    const int n = ix->g.ntids[elem];
    for (int k = 0; k < n; k++) if (tids_out) tids_out[k] = ix->g.tids[elem][k];
    return n;
}

int hx_index_export_levels(const hx_index *ix, uint32_t first, uint32_t n, int32_t *levels_out)
{
    if (!ix || (!levels_out && n) || (uint64_t)first + n > ix->g.size()) return HX_E_ARG;
    if (n) memcpy(levels_out, ix->g.level.data() + first, (size_t)n * sizeof(int32_t));
    return HX_OK;
}

int hx_index_export_layer(const hx_index *ix, int layer, uint32_t first, uint32_t n, uint32_t *ids_out, float *dist_out, uint16_t *cnt_out)
{
    if (ix) { int rc0 = const_cast<hx_index *>(ix)->ensure_host_lists(); if (rc0) return rc0; }
    if (!ix || layer < 0 || (n && (!ids_out || !cnt_out)) || (uint64_t)first + n > ix->g.size()) return HX_E_ARG;
    const Graph &g = ix->g; const size_t lm = (size_t)g.lm(layer);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t e = first + i;
        if (g.level[e] < layer) { cnt_out[i] = 0; continue; }
        const Cand *l = g.list(e, layer); const uint16_t c = g.cnt(e, layer);
        cnt_out[i] = c;
        for (uint16_t k = 0; k < c; k++) { ids_out[i * lm + k] = l[k].id; if (dist_out) dist_out[i * lm + k] = l[k].d; }
    }
    return HX_OK;
}

int hx_index_set_neighbors(hx_index *ix, uint32_t elem, int layer, uint32_t count, const uint32_t *ids, const float *dist)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (ix) { int rc0 = ix->ensure_host_lists(); if (rc0) return rc0; }
    if (!ix || elem >= ix->g.size()) return HX_E_ARG;
    Graph &g = ix->g;
    if (layer < 0 || g.level[elem] < layer || count > (uint32_t)g.lm(layer) || (count && (!ids || !dist))) return ix->fail(HX_E_ARG, "bad neighbour list");
    Cand *l = g.list(elem, layer);
    for (uint32_t k = 0; k < count; k++) l[k] = Cand{dist[k], ids[k]};
    g.cnt(elem, layer) = (uint16_t)count;
    ix->dirty.emplace_back(elem, layer);
    return HX_OK;
}

int hx_index_set_fused(hx_index *ix, int enabled) { if (!ix) return HX_E_ARG; ix->fused = enabled != 0; return HX_OK; }
int hx_index_set_mfma(hx_index *ix, int enabled)
{
    if (!ix) return HX_E_ARG;
    if (enabled && (ix->e->dtype != HX_F16 || ix->e->metric != HX_NEG_IP)) return ix->fail(HX_E_ARG, "the MFMA pair path serves halfvec inner product");
    ix->mfma = enabled != 0;
    return HX_OK;
}
int hx_index_mfma_stats(const hx_index *ix, uint64_t *mfma_pairs, uint64_t *exact_pairs)
{
    if (!ix) return HX_E_ARG;
    if (mfma_pairs) *mfma_pairs = ix->mfma_pairs;
    if (exact_pairs) *exact_pairs = ix->mfma_exact;
    return HX_OK;
}
int hx_index_fused_stats(const hx_index *ix, uint64_t *tasks, uint64_t *redone)
{
    if (!ix) return HX_E_ARG;
    if (tasks) *tasks = ix->fused_tasks;
    if (redone) *redone = ix->fused_redo | (ix->e->fused_cmax << 32);   // high half: max candidate-heap length seen (diagnostic)
    return HX_OK;
}

int hx_index_profile(const hx_index *ix, double seconds_out[16], int reset)
{
    if (!ix || !seconds_out) return HX_E_ARG;
    memcpy(seconds_out, ix->prof, sizeof ix->prof);
    if (reset) memset(const_cast<hx_index *>(ix)->prof, 0, sizeof ix->prof);
    return HX_OK;
}

int hx_index_counters(const hx_index *ix, uint64_t counters_out[8])
{
    if (!ix || !counters_out) return HX_E_ARG;
    memcpy(counters_out, ix->counters, sizeof ix->counters);
    return HX_OK;
}

// the scans `todo` (indices into the caller's output arrays; engine query slot = q0 + index) on the lock-step host driver
static int lockstep_scan(hx_index *ix, const std::vector<uint32_t> &todo, uint32_t q0, uint32_t ef_search, int mode, int64_t max_scan_tuples, uint32_t limit,
                         const uint8_t *filter, uint64_t n_filter, int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out)
{
    const uint32_t nt = (uint32_t)todo.size();
    std::vector<std::unique_ptr<QueryTask>> &qs = ix->query_pool; std::vector<LsTask *> tasks(nt);
    while (qs.size() < nt) qs.emplace_back(new QueryTask());
    for (uint32_t qi = 0; qi < nt; qi++) {
        const uint32_t q = todo[qi];
        QueryTask &t = *qs[qi];
        t.st = QueryTask::Q_INIT; t.lc = 0; t.n_dist = t.n_pair = 0; t.clear_req(); t.tuples = 0; t.previous_distance = -HUGE_VAL;
        t.out_tid.clear(); t.out_d.clear(); t.out_elem.clear(); t.discarded.clear(); t.results.clear();
        t.g = &ix->g; t.slot = q0 + q; t.ef_search = ef_search; t.mode = mode; t.max_scan_tuples = max_scan_tuples; t.limit = limit;
        t.filter = filter; t.n_filter = n_filter;
        tasks[qi] = &t;
    }
    int rc = ix->run_lockstep(tasks);
    if (rc) return rc;
    for (uint32_t qi = 0; qi < nt; qi++) {
        const uint32_t q = todo[qi];
        QueryTask &t = *qs[qi];
        const uint32_t c = (uint32_t)t.out_tid.size();
        counts_out[q] = c;
        for (uint32_t k = 0; k < c; k++) {
            tids_out[(size_t)q * limit + k] = t.out_tid[k];
            if (dist_out) dist_out[(size_t)q * limit + k] = t.out_d[k];
            if (elems_out) elems_out[(size_t)q * limit + k] = t.out_elem[k];
        }
        ix->counters[4] += t.n_dist;
    }
    return HX_OK;
}


// amgettuple's expansion of a device scan's results (per query: every heap TID of each element, nearest first, scan.rs:794-875) and the
// one retry of overflowed queries on the device with roomier tables; queries that still fail are appended to `todo` (lock-step driver).
// q0: engine query slot of the view's first query (pipelined scans address a window of the uploaded queries)
static int finish_device_scan(hx_index *ix, const HxFusedView &v, uint32_t q0, uint32_t nq, uint32_t ef_search, uint32_t limit,
                          int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out, std::vector<uint32_t> &todo)
{
    const Graph &g = ix->g;
    const uint32_t ke = std::min<uint32_t>(limit, ef_search);
    int rc;
    uint64_t cnts[2] = {0, 0};
    const uint32_t *status = v.status, *ocnt = v.cnt, *oids = v.ids; const float *od = v.d;
    const double t_exp0 = hx_index::now_s();
    ix->pool->parallel_for((nq + 1023) / 1024, [&](size_t ci) {
        for (uint32_t q = (uint32_t)ci * 1024; q < std::min<uint32_t>(nq, (uint32_t)ci * 1024 + 1024); q++) {
            if (status[q] != 0) continue;
            if (q + 1 < nq) for (uint32_t i = 0; i < ocnt[q + 1] && i < ke; i++) {      // the heap TIDs of the next query's elements: random host reads
                const uint32_t eln = oids[(size_t)(q + 1) * ke + i]; __builtin_prefetch(&g.tids[eln]); __builtin_prefetch(&g.ntids[eln]);
            }
            uint32_t c = 0;                                  // amgettuple: every heap TID of each element, nearest first (scan.rs:794-875)
            for (uint32_t i = 0; i < ocnt[q] && c < limit; i++) {
                const uint32_t el = oids[(size_t)q * ke + i];
                for (int t = (int)g.ntids[el] - 1; t >= 0 && c < limit; t--) {
                    tids_out[(size_t)q * limit + c] = g.tids[el][t];
                    if (dist_out) dist_out[(size_t)q * limit + c] = od[(size_t)q * ke + i];
                    if (elems_out) elems_out[(size_t)q * limit + c] = el;
                    c++;
                }
            }
            counts_out[q] = c;
        }
    });
    ix->prof[15] += hx_index::now_s() - t_exp0;
    std::vector<uint32_t> again;                             // overflowed queries: one more try on the device with roomier tables
    for (uint32_t q = 0; q < nq; q++) { if (status[q] == 1) again.push_back(q); else if (status[q] != 0) todo.push_back(q); }
    if (!again.empty()) {
        const uint32_t na = (uint32_t)again.size();
        std::vector<uint32_t> q2(na);
        for (uint32_t k = 0; k < na; k++) q2[k] = HX_QUERY_SLOT | (q0 + again[k]);
        HxFusedView v2;
        if ((rc = ix->e->fused_run(0, na, q2.data(), nullptr, ef_search, ke, (uint32_t)g.entry, g.level[g.entry],
                                   nullptr, nullptr, nullptr, nullptr, cnts, nullptr, &v2, 8))) return ix->fail(rc, ix->e->err);
        ix->counters[4] += cnts[0];
        for (uint32_t k = 0; k < na; k++) {
            const uint32_t q = again[k];
            if (v2.status[k] != 0) { todo.push_back(q); continue; }
            uint32_t c = 0;
            for (uint32_t i = 0; i < v2.cnt[k] && c < limit; i++) {
                const uint32_t el = v2.ids[(size_t)k * ke + i];
                for (int t = (int)g.ntids[el] - 1; t >= 0 && c < limit; t--) {
                    tids_out[(size_t)q * limit + c] = g.tids[el][t];
                    if (dist_out) dist_out[(size_t)q * limit + c] = v2.d[(size_t)k * ke + i];
                    if (elems_out) elems_out[(size_t)q * limit + c] = el;
                    c++;
                }
            }
            counts_out[q] = c;
        }
    }
    return HX_OK;
}

static int search_impl(hx_index *ix, uint32_t nq, uint32_t ef_search, int mode, int64_t max_scan_tuples, uint32_t limit,
                       const uint8_t *filter, uint64_t n_filter, int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out)
{
    if (!ix) return HX_E_ARG;
    if (nq == 0) return HX_OK;
    if (!tids_out || !counts_out) return ix->fail(HX_E_ARG, "NULL argument");
    if (ef_search < 1 || ef_search > 1000) return ix->fail(HX_E_ARG, "hnsw.ef_search must be between 1 and 1000");   // options.rs:156-166
    if (nq > ix->e->n_queries) return ix->fail(HX_E_STATE, "upload the queries with hx_set_queries first");
    std::vector<uint32_t> todo;                                  // query slots for the lock-step path
    if (mode == 0 && ix->device_scan_ok() && ix->g.entry >= 0) {
        const double t_sm0 = hx_index::now_s();
        int rc = ix->sync_mirror();
        if (rc) return rc;
        ix->prof[4] += hx_index::now_s() - t_sm0;
        const Graph &g = ix->g;
        const uint32_t ke = std::min<uint32_t>(limit, ef_search);
        std::vector<uint32_t> qsel(nq);
        for (uint32_t q = 0; q < nq; q++) qsel[q] = HX_QUERY_SLOT | q;
        uint64_t cnts[2] = {0, 0};
        HxFusedView v;                                          // results are read in place from the pinned staging buffer
        auto t0 = std::chrono::steady_clock::now();
        if ((rc = ix->e->fused_run(0, nq, qsel.data(), nullptr, ef_search, ke, (uint32_t)g.entry, g.level[g.entry],
                                   nullptr, nullptr, nullptr, nullptr, cnts, nullptr, &v))) return ix->fail(rc, ix->e->err);
        ix->prof[6] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ix->counters[4] += cnts[0];
        if ((rc = finish_device_scan(ix, v, 0, nq, ef_search, limit, tids_out, dist_out, elems_out, counts_out, todo))) return rc;
        ix->fused_tasks += nq; ix->fused_redo += todo.size();
        if (todo.empty()) return HX_OK;
    } else if (mode != 0 && ix->device_scan_ok() && ix->g.entry >= 0 && limit <= 4096) {
        // iterative scan on the device (k_fused MODE 2): the filter reaches the kernel as a per-element mask of passing heap TIDs
        int rc = ix->sync_mirror();
        if (rc) return rc;
        const Graph &g = ix->g;
        const uint32_t n = g.size();
        std::vector<uint16_t> emask(n);
        ix->pool->parallel_for((n + 65535) / 65536, [&](size_t ci) {
            for (size_t el = ci * 65536; el < std::min<size_t>(n, ci * 65536 + 65536); el++) {
                const uint32_t nt = g.level[el] < 0 ? 0u : g.ntids[el]; uint32_t msk = 0;
                for (uint32_t t = 0; t < nt; t++) { const int64_t tid = g.tids[el][t]; if (!filter || (tid >= 0 && (uint64_t)tid < n_filter && filter[tid])) msk |= 1u << t; }
                emask[el] = (uint16_t)(msk | (nt << 12));
            }
        });
        std::vector<uint32_t> qsel(nq), status(nq), ocnt(nq), oids((size_t)nq * limit), otix((size_t)nq * limit);
        std::vector<float> od((size_t)nq * limit);
        for (uint32_t q = 0; q < nq; q++) qsel[q] = HX_QUERY_SLOT | q;
        HxFusedIter it; it.iter_mode = mode; it.max_tuples = max_scan_tuples; it.emask = emask.data(); it.n_elems = n; it.out_tix = otix.data();
        uint64_t cnts[2] = {0, 0};
        auto t0 = std::chrono::steady_clock::now();
        if ((rc = ix->e->fused_run(2, nq, qsel.data(), nullptr, ef_search, limit, (uint32_t)g.entry, g.level[g.entry],
                                   oids.data(), od.data(), ocnt.data(), status.data(), cnts, &it))) return ix->fail(rc, ix->e->err);
        ix->prof[6] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ix->counters[4] += cnts[0];
        auto take = [&](uint32_t q, size_t src, const uint32_t *cn, const uint32_t *ids, const uint32_t *tix, const float *dd) {
            const uint32_t c = std::min(cn[src], limit);
            for (uint32_t i = 0; i < c; i++) {
                const uint32_t el = ids[src * limit + i];
                tids_out[(size_t)q * limit + i] = g.tids[el][tix[src * limit + i]];
                if (dist_out) dist_out[(size_t)q * limit + i] = dd[src * limit + i];
                if (elems_out) elems_out[(size_t)q * limit + i] = el;
            }
            counts_out[q] = c;
        };
        std::vector<uint32_t> again;                             // scans that outgrew their visited table / discarded heap: once more on the device, 8x the room
        for (uint32_t q = 0; q < nq; q++) {
            if (status[q] == 1) again.push_back(q); else if (status[q] != 0) todo.push_back(q); else take(q, q, ocnt.data(), oids.data(), otix.data(), od.data());
        }
        if (!again.empty()) {
            const uint32_t na = (uint32_t)again.size();
            std::vector<uint32_t> q2(na), st2(na), c2(na), i2((size_t)na * limit), x2((size_t)na * limit); std::vector<float> d2((size_t)na * limit);
            for (uint32_t k = 0; k < na; k++) q2[k] = HX_QUERY_SLOT | again[k];
            HxFusedIter it2 = it; it2.out_tix = x2.data();
            if ((rc = ix->e->fused_run(2, na, q2.data(), nullptr, ef_search, limit, (uint32_t)g.entry, g.level[g.entry],
                                       i2.data(), d2.data(), c2.data(), st2.data(), cnts, &it2, nullptr, 8))) return ix->fail(rc, ix->e->err);
            ix->counters[4] += cnts[0];
            for (uint32_t k = 0; k < na; k++) { if (st2[k] != 0) todo.push_back(again[k]); else take(again[k], k, c2.data(), i2.data(), x2.data(), d2.data()); }
        }
        ix->fused_tasks += nq; ix->fused_redo += todo.size();
        if (todo.empty()) return HX_OK;
    } else {
        for (uint32_t q = 0; q < nq; q++) todo.push_back(q);
    }
    return lockstep_scan(ix, todo, 0, ef_search, mode, max_scan_tuples, limit, filter, n_filter, tids_out, dist_out, elems_out, counts_out);
}

int hx_index_search(hx_index *ix, uint32_t nq, uint32_t ef_search, uint32_t k,
                    int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out)
{
    return search_impl(ix, nq, ef_search, 0, 0, k, nullptr, 0, tids_out, dist_out, elems_out, counts_out);
}

int hx_index_search_submit(hx_index *ix, uint32_t slot, uint32_t first_query, uint32_t nq, uint32_t ef_search, uint32_t k)
{
    if (!ix) return HX_E_ARG;
    if (slot >= HX_SCAN_SLOTS) return ix->fail(HX_E_ARG, "scan slot out of range");
    if (nq == 0 || k == 0) return ix->fail(HX_E_ARG, "nq and k must be positive");
    if (ef_search < 1 || ef_search > 1000) return ix->fail(HX_E_ARG, "hnsw.ef_search must be between 1 and 1000");   // options.rs:156-166
    if ((uint64_t)first_query + nq > ix->e->n_queries) return ix->fail(HX_E_STATE, "upload the queries with hx_set_queries first");
    hx_index::ScanSlot &ss = ix->scan_slot[slot];
    if (ss.busy) return ix->fail(HX_E_STATE, "scan slot busy: hx_index_search_wait first");
    if (!ix->device_scan_ok() || ix->g.entry < 0) return ix->fail(HX_E_STATE, "this index does not scan on the device (empty, hx_index_set_fused(0), or a deleted element can still be met): use hx_index_search");
    int rc;
    if (ix->scans_in_flight == 0) { if ((rc = ix->sync_mirror())) return rc; }
    else if (ix->g.size() != ix->mirror_elems || !ix->dirty.empty()) return ix->fail(HX_E_STATE, "the index was modified while a scan is in flight");
    hx_engine *e = ix->e;
    if ((rc = e->scan_io_init(slot))) return ix->fail(rc, e->err);
    HxFusedIo &io = e->scan_io[slot];
    // the mirror and the queries were written on the engine's stream: order this slot's stream behind it
    if (hipEventRecord(io.ev_dep, e->stream) != hipSuccess || hipStreamWaitEvent(io.stream, io.ev_dep, 0) != hipSuccess) return ix->fail(HX_E_HIP, "stream dependency");
    const Graph &g = ix->g;
    const uint32_t ke = std::min<uint32_t>(k, ef_search);
    std::vector<uint32_t> qsel(nq);
    for (uint32_t q = 0; q < nq; q++) qsel[q] = HX_QUERY_SLOT | (first_query + q);
    if ((rc = e->fused_launch(io, 0, nq, qsel.data(), nullptr, ef_search, ke, (uint32_t)g.entry, g.level[g.entry], nullptr, 1, nullptr))) return ix->fail(rc, e->err);
    ss.busy = true; ss.first = first_query; ss.nq = nq; ss.ef = ef_search; ss.k = k;
    ix->scans_in_flight++;
    return HX_OK;
}

int hx_index_search_wait(hx_index *ix, uint32_t slot, int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out)
{
    if (!ix) return HX_E_ARG;
    if (slot >= HX_SCAN_SLOTS) return ix->fail(HX_E_ARG, "scan slot out of range");
    if (!tids_out || !counts_out) return ix->fail(HX_E_ARG, "NULL argument");
    hx_index::ScanSlot &ss = ix->scan_slot[slot];
    if (!ss.busy) return ix->fail(HX_E_STATE, "nothing submitted on this scan slot");
    hx_engine *e = ix->e;
    HxFusedView v; uint64_t cnts[2] = {0, 0};
    ss.busy = false; ix->scans_in_flight--;
    int rc = e->fused_collect(e->scan_io[slot], nullptr, nullptr, nullptr, nullptr, cnts, &v);
    if (rc) return ix->fail(rc, e->err);
    ix->counters[4] += cnts[0];
    std::vector<uint32_t> todo;
    if ((rc = finish_device_scan(ix, v, ss.first, ss.nq, ss.ef, ss.k, tids_out, dist_out, elems_out, counts_out, todo))) return rc;
    ix->fused_tasks += ss.nq; ix->fused_redo += todo.size();
    if (todo.empty()) return HX_OK;
    // a query that outgrew even the roomy tables: the lock-step driver (still on the GPU kernels) serves it from its engine query slot
    return lockstep_scan(ix, todo, ss.first, ss.ef, 0, 0, ss.k, nullptr, 0, tids_out, dist_out, elems_out, counts_out);
}

int hx_index_search_iterative(hx_index *ix, uint32_t nq, uint32_t ef_search, int mode, int64_t max_scan_tuples,
                              uint32_t limit, const uint8_t *filter_pass, uint64_t n_filter,
                              int64_t *tids_out, float *dist_out, uint32_t *counts_out)
{
    if (mode != 1 && mode != 2) return ix ? ix->fail(HX_E_ARG, "mode must be 1 (relaxed_order) or 2 (strict_order)") : HX_E_ARG;
    if (max_scan_tuples < 1) return ix->fail(HX_E_ARG, "hnsw.max_scan_tuples must be at least 1");
    return search_impl(ix, nq, ef_search, mode, max_scan_tuples, limit, filter_pass, n_filter, tids_out, dist_out, nullptr, counts_out);
}

// A NULL order-by value (`ORDER BY val <-> NULL`): load_element reports distance 0.0 for every element without calling the distance procedure
// (scan.rs:186-187), so the scan is graph traversal only -- which elements it reaches, and in which order, follows from Rust's BinaryHeap on equal
// keys.  No arithmetic, hence no kernel: the lock-step scan task runs on the host with zeros for every distance it asks for.
int hx_index_search_null(hx_index *ix, uint32_t ef_search, int mode, int64_t max_scan_tuples, uint32_t limit,
                         const uint8_t *filter_pass, uint64_t n_filter, int64_t *tids_out, uint32_t *elems_out, uint32_t *count_out)
{
    if (!ix || !tids_out || !count_out) return HX_E_ARG;
    if (mode < 0 || mode > 2) return ix->fail(HX_E_ARG, "mode must be 0 (off), 1 (relaxed_order) or 2 (strict_order)");
    if (ef_search < 1 || ef_search > 1000) return ix->fail(HX_E_ARG, "hnsw.ef_search must be between 1 and 1000");
    if (mode != 0 && max_scan_tuples < 1) return ix->fail(HX_E_ARG, "hnsw.max_scan_tuples must be at least 1");
    { int rc0 = ix->ensure_host_lists(); if (rc0) return rc0; }
    QueryTask t;
    t.g = &ix->g; t.slot = 0; t.ef_search = ef_search; t.mode = mode; t.max_scan_tuples = max_scan_tuples; t.limit = limit;
    t.filter = filter_pass; t.n_filter = n_filter;
    std::vector<float> zeros;
    const float *dres = nullptr;
    while (t.advance(dres, nullptr)) { zeros.assign(t.dist_ids.size(), 0.0f); dres = zeros.data(); }
    const uint32_t c = (uint32_t)t.out_tid.size();
    for (uint32_t k = 0; k < c; k++) { tids_out[k] = t.out_tid[k]; if (elems_out) elems_out[k] = t.out_elem[k]; }
    *count_out = c;
    return HX_OK;
}

// ------------------------------------------------------------------------------------------------
// f1: graph -> PostgreSQL HNSW index pages.  One forward pass over the elements: a cursor (current block, its
// pd_lower / pd_upper) places the element tuple and then the neighbour tuple exactly where PageAddItemExtended would
// (line pointer grows from the front, MAXALIGNed tuple data from the back); since every element's location is known
// only after the pass, neighbour TIDs are filled in a second sweep over the already-placed tuples (the reference
// overwrites its placeholders the same way, build.rs:727-795).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr uint32_t PG_BLCKSZ = HX_PAGE_SIZE, PG_PAGE_HDR = 24, PG_ITEMID = 4, PG_SPECIAL = 8, PG_LAYOUT_VERSION = 4;
constexpr uint32_t ETUP_HDR = 72, NTUP_HDR = 4, TID_BYTES = 6, INVALID_BLOCK = 0xFFFFFFFFu;
inline uint32_t pg_maxalign(uint32_t x) { return (x + 7u) & ~7u; }
inline void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
inline void put32(uint8_t *p, uint32_t v) { put16(p, v & 0xffffu); put16(p + 2, v >> 16); }
inline void put_tid(uint8_t *p, uint32_t blk, uint32_t off) { put16(p, blk >> 16); put16(p + 2, blk & 0xffffu); put16(p + 4, off); }   // ItemPointerSet: bi_hi, bi_lo, ip_posid
struct PageCursor {
    uint8_t *base; uint64_t cap; uint32_t blk; uint32_t lower, upper;
    uint8_t *page() const { return base ? base + (size_t)blk * PG_BLCKSZ : nullptr; }
    bool start(uint32_t b)                     // PageInit(page, BLCKSZ, sizeof(HnswPageOpaqueData)) + hnsw_init_page, build.rs:59-65
    {
        blk = b; lower = PG_PAGE_HDR; upper = PG_BLCKSZ - PG_SPECIAL;
        if (!base) return true;
        if (b >= cap) return false;
        uint8_t *pg = page();
        memset(pg, 0, PG_BLCKSZ);
        put16(pg + 12, lower); put16(pg + 14, upper); put16(pg + 16, upper); put16(pg + 18, PG_BLCKSZ | PG_LAYOUT_VERSION);
        put32(pg + upper, INVALID_BLOCK); put16(pg + upper + 4, 0); put16(pg + upper + 6, 0xFF90);
        return true;
    }
    uint32_t free_space() const { const uint32_t sp = upper - lower; return sp < PG_ITEMID ? 0u : sp - PG_ITEMID; }   // PageGetFreeSpace
    uint32_t max_off() const { return (lower - PG_PAGE_HDR) / PG_ITEMID; }                                              // PageGetMaxOffsetNumber
    // PageAddItemExtended(page, item, size, InvalidOffsetNumber, 0): returns the tuple's byte offset in the page, 0 if it does not fit
    uint32_t add(uint32_t size)
    {
        const uint32_t nl = lower + PG_ITEMID, al = pg_maxalign(size);
        if (al > upper || nl > upper - al) return 0;
        const uint32_t nu = upper - al;
        if (base) {
            uint8_t *pg = page();
            put32(pg + lower, nu | (1u << 15) | (size << 17));           // ItemIdSetNormal: lp_off:15 | lp_flags=LP_NORMAL:2 | lp_len:15
            put16(pg + 12, nl); put16(pg + 14, nu);
        }
        lower = nl; upper = nu;
        return nu;
    }
    void link_next(uint32_t next) { if (base) put32(page() + (PG_BLCKSZ - PG_SPECIAL), next); }                          // hnsw_build_append_page, build.rs:92-112
};
}  // namespace

int hx_index_serialize_pages(const hx_index *cix, uint8_t *pages_out, uint64_t cap_pages, uint64_t *n_pages_out,
                             uint32_t *elem_blkno_out, uint16_t *elem_offno_out)
{
    if (!cix || !n_pages_out) return HX_E_ARG;
    hx_index *ix = const_cast<hx_index *>(cix);
    if (ix->e->dtype == HX_SPARSE) return ix->fail(HX_E_ARG, "no page image for sparsevec (its varlena is variable-length; the engine holds fixed-size records)");
    { int rc0 = ix->ensure_host_lists(); if (rc0) return rc0; }
    const Graph &g = ix->g;
    const hx_engine *e = ix->e;
    const uint32_t n = g.size(), m = (uint32_t)g.m;
    const uint32_t payload = (uint32_t)hx_row_bytes(e), value_size = 8u + payload;
    const uint32_t max_size = PG_BLCKSZ - pg_maxalign(PG_PAGE_HDR) - pg_maxalign(PG_SPECIAL) - PG_ITEMID;               // hnsw_max_size, types/hnsw.rs:325-331
    const uint32_t etup_size = pg_maxalign(ETUP_HDR + value_size);                                                         // hnsw_element_tuple_size
    if (etup_size > max_size) return ix->fail(HX_E_ARG, "index tuple too large");                                          // build.rs:611-613
    if (pages_out && cap_pages < 2) return ix->fail(HX_E_ARG, "cap_pages too small");

    struct Loc { uint32_t blk, nblk; uint16_t off, noff; uint32_t nbyte; };   // element tuple (blk, off); neighbour tuple (nblk, noff) at byte nbyte of its page
    std::vector<Loc> loc(n, Loc{INVALID_BLOCK, INVALID_BLOCK, 0, 0, 0});
    std::vector<uint8_t> rows;                                                // payload of a chunk of elements, read back from the device
    const uint32_t CH = 16384; uint32_t rows_first = 0, rows_n = 0;

    // ---- meta page (block 0), create_meta_page build.rs:545-568 ----
    PageCursor cur{pages_out, cap_pages, 0, 0, 0};
    if (!cur.start(0)) return ix->fail(HX_E_ARG, "cap_pages too small");
    uint8_t *meta = pages_out ? pages_out + PG_PAGE_HDR : nullptr;
    if (meta) {
        put32(meta + 0, 0xA953A953u); put32(meta + 4, 1u); put32(meta + 8, (uint32_t)e->dim);
        put16(meta + 12, m); put16(meta + 14, (uint32_t)ix->efc);
        put32(meta + 16, INVALID_BLOCK); put16(meta + 20, 0); put16(meta + 22, 0xFFFFu /* entry_level -1 */); put32(meta + 24, INVALID_BLOCK);
        put16(pages_out + 12, PG_PAGE_HDR + 28u);                                                                          // pd_lower = end of HnswMetaPageData
    }
    // ---- data pages, create_graph_pages build.rs:576-712 ----
    if (!cur.start(1)) return ix->fail(HX_E_ARG, "cap_pages too small");
    for (uint32_t idx = 0; idx < n; idx++) {
        if (g.level[idx] < 0) continue;                                                                                    // merged duplicate: no element of its own
        const uint32_t level = (uint32_t)g.level[idx];
        if (level > 255u) return ix->fail(HX_E_STATE, "level does not fit the element tuple");
        const uint32_t ntup_size = pg_maxalign(NTUP_HDR + (level + 2u) * m * TID_BYTES);                                   // hnsw_neighbor_tuple_size
        const uint32_t combined = etup_size + ntup_size + PG_ITEMID;
        uint32_t fs = cur.free_space();
        if (fs < etup_size || (combined <= max_size && fs < combined)) {                                                   // build.rs:644-650
            cur.link_next(cur.blk + 1);
            if (!cur.start(cur.blk + 1)) return ix->fail(HX_E_ARG, "cap_pages too small");
        }
        Loc &L = loc[idx];
        L.blk = cur.blk; L.off = (uint16_t)(cur.max_off() + 1);
        if (combined <= max_size) { L.nblk = L.blk; L.noff = (uint16_t)(L.off + 1); } else { L.nblk = L.blk + 1; L.noff = 1; }   // build.rs:657-661
        const uint32_t eb = cur.add(etup_size);
        if (!eb) return ix->fail(HX_E_STATE, "failed to add element tuple to page");
        if (pages_out) {
            if (idx < rows_first || idx >= rows_first + rows_n) {
                rows_first = idx; rows_n = std::min<uint32_t>(CH, n - idx);
                rows.resize((size_t)rows_n * payload);
                int rc = hx_read_rows(ix->e, rows_first, rows_n, rows.data());
                if (rc) return ix->fail(rc, hx_last_error(ix->e));
            }
            uint8_t *t = cur.page() + eb;                              // page was zeroed: padding and `unused` stay 0
            t[0] = 1; t[1] = (uint8_t)level; t[2] = 0; t[3] = 0;
            for (uint32_t k = 0; k < HEAPTIDS; k++) {
                if (k < g.ntids[idx]) { const uint64_t v = (uint64_t)g.tids[idx][k]; put_tid(t + 4 + k * TID_BYTES, (uint32_t)(v >> 16), (uint32_t)(v & 0xffffu)); }
                else put_tid(t + 4 + k * TID_BYTES, INVALID_BLOCK, 0);                                                     // ItemPointerSetInvalid
            }
            put_tid(t + 64, L.nblk, L.noff);
            uint8_t *v = t + ETUP_HDR;
            put32(v, value_size << 2);                                                                                     // SET_VARSIZE, 4-byte header
            if (e->dtype == HX_BIT) put32(v + 4, (uint32_t)e->dim); else { put16(v + 4, (uint32_t)e->dim); put16(v + 6, 0); }
            memcpy(v + 8, rows.data() + (size_t)(idx - rows_first) * payload, payload);
        }
        if (cur.free_space() < ntup_size) {                                                                                // build.rs:684-688
            cur.link_next(cur.blk + 1);
            if (!cur.start(cur.blk + 1)) return ix->fail(HX_E_ARG, "cap_pages too small");
        }
        if (cur.blk != L.nblk || cur.max_off() + 1 != L.noff) return ix->fail(HX_E_STATE, "failed to add neighbor tuple to page");
        L.nbyte = cur.add(ntup_size);
        if (!L.nbyte) return ix->fail(HX_E_STATE, "failed to add neighbor tuple to page");
    }
    const uint32_t insert_page = cur.blk, n_pages = cur.blk + 1;
    *n_pages_out = n_pages;
    for (uint32_t i = 0; i < n; i++) { if (elem_blkno_out) elem_blkno_out[i] = loc[i].blk; if (elem_offno_out) elem_offno_out[i] = loc[i].off; }
    if (!pages_out) return HX_OK;
    // ---- neighbour tuples, write_neighbor_tuples build.rs:719-795 ----
    for (uint32_t idx = 0; idx < n; idx++) {
        if (g.level[idx] < 0) continue;
        const Loc &L = loc[idx];
        uint8_t *t = pages_out + (size_t)L.nblk * PG_BLCKSZ + L.nbyte;
        t[0] = 2; t[1] = 0;
        uint32_t k = 0;
        for (int lc = g.level[idx]; lc >= 0; lc--) {
            const uint32_t lm = (uint32_t)g.lm(lc), c = g.cnt(idx, lc); const Cand *lst = g.list(idx, lc);
            for (uint32_t i = 0; i < lm; i++, k++) {
                if (i < c) {
                    const Loc &N = loc[lst[i].id];
                    if (N.blk == INVALID_BLOCK) return ix->fail(HX_E_STATE, "neighbour list references a merged duplicate");
                    put_tid(t + NTUP_HDR + k * TID_BYTES, N.blk, N.off);
                } else put_tid(t + NTUP_HDR + k * TID_BYTES, INVALID_BLOCK, 0);
            }
        }
        put16(t + 2, k);
    }
    // ---- update_meta_page build.rs:801-821 ----
    if (g.entry >= 0) {
        const Loc &E = loc[(size_t)g.entry];
        put32(meta + 16, E.blk); put16(meta + 20, E.off); put16(meta + 22, (uint32_t)g.level[(size_t)g.entry] & 0xffffu);
    }
    put32(meta + 24, insert_page);
    return HX_OK;
}

// f2: pages -> graph + rows (see include/hnswrx.h).  Reads are bounds-checked: a malformed image yields HX_E_ARG, never a fault.
namespace {
inline uint32_t get16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
inline uint32_t get32(const uint8_t *p) { return get16(p) | (get16(p + 2) << 16); }
struct TidRef { uint32_t blk, off; bool valid() const { return !(blk == INVALID_BLOCK) && off != 0; } };   // ItemPointerIsValid: ip_posid != 0
inline TidRef get_tid(const uint8_t *p) { return TidRef{(get16(p) << 16) | get16(p + 2), get16(p + 4)}; }
}

int hx_index_load_pages(hx_index *ix, const uint8_t *pages, uint64_t n_pages, uint32_t *elem_blkno_out, uint16_t *elem_offno_out,
                        uint64_t cap_elems, uint64_t *n_elems_out)
{
    if (ix && ix->scans_in_flight) return ix->fail(HX_E_STATE, "a pipelined scan is in flight: hx_index_search_wait first");
    if (!ix || !pages || !n_elems_out) return HX_E_ARG;
    Graph &g = ix->g; hx_engine *e = ix->e;
    if (e->dtype == HX_SPARSE) return ix->fail(HX_E_ARG, "no page image for sparsevec");
    if (g.size() != 0 || hx_num_rows(e) != 0) return ix->fail(HX_E_STATE, "hx_index_load_pages needs an empty index on an empty engine");
    if (n_pages < 2 || n_pages > 0xFFFFFFFEull) return ix->fail(HX_E_ARG, "page image too short");
    const uint8_t *meta = pages + PG_PAGE_HDR;
    if (get32(meta) != 0xA953A953u || get32(meta + 4) != 1u) return ix->fail(HX_E_ARG, "not an HNSW meta page (magic / version)");
    const uint32_t m = get16(meta + 12);
    if (get32(meta + 8) != (uint32_t)e->dim) return ix->fail(HX_E_ARG, "meta page dimensions differ from the engine's");
    if (m != (uint32_t)g.m) return ix->fail(HX_E_ARG, "meta page m differs from the index's");
    const uint32_t payload = (uint32_t)hx_row_bytes(e), value_size = 8u + payload;
    // item table of every page on the chain: (page, offset number) -> byte offset, length
    auto item = [&](uint32_t blk, uint32_t off, uint32_t &lo, uint32_t &ln) -> bool {
        if (blk == 0 || blk >= n_pages || off == 0) return false;
        const uint8_t *pg = pages + (size_t)blk * PG_BLCKSZ;
        const uint32_t lower = get16(pg + 12);
        if (lower < PG_PAGE_HDR || lower > PG_BLCKSZ || off > (lower - PG_PAGE_HDR) / PG_ITEMID) return false;
        const uint32_t lp = get32(pg + PG_PAGE_HDR + (off - 1) * PG_ITEMID);
        lo = lp & 0x7fffu; ln = lp >> 17;
        return ((lp >> 15) & 3u) == 1u && ln != 0 && lo + ln <= PG_BLCKSZ;          // LP_NORMAL, scan.rs:170-174
    };
    // pass 1: elements in chain order
    struct El { uint32_t blk, off, lo; };
    std::vector<El> els;
    std::vector<uint32_t> page_base(n_pages + 1, 0u);                              // index into item_idx
    for (uint64_t b = 0; b < n_pages; b++) {
        const uint32_t lower = get16(pages + b * PG_BLCKSZ + 12);
        page_base[b + 1] = page_base[b] + (b == 0 || lower < PG_PAGE_HDR || lower > PG_BLCKSZ ? 0u : (lower - PG_PAGE_HDR) / PG_ITEMID);
    }
    std::vector<uint32_t> item_idx(page_base[n_pages], 0xFFFFFFFFu);
    {
        std::vector<uint8_t> seen(n_pages, 0);
        for (uint32_t blk = 1; blk != INVALID_BLOCK;) {
            if (blk >= n_pages || seen[blk]) return ix->fail(HX_E_ARG, "page chain leaves the image or loops");
            seen[blk] = 1;
            const uint8_t *pg = pages + (size_t)blk * PG_BLCKSZ;
            const uint32_t special = get16(pg + 16);
            if (special != PG_BLCKSZ - PG_SPECIAL || get16(pg + special + 6) != 0xFF90u) return ix->fail(HX_E_ARG, "not an HNSW page (special area / page id)");
            const uint32_t nitems = page_base[blk + 1] - page_base[blk];
            for (uint32_t off = 1; off <= nitems; off++) {
                uint32_t lo, ln;
                if (!item(blk, off, lo, ln)) continue;
                const uint8_t *t = pg + lo;
                if (t[0] != 1 || t[2] != 0) continue;                              // element tuples that are not deleted, scan.rs:178-181
                if (ln < ETUP_HDR + value_size || (get32(t + ETUP_HDR) >> 2) != value_size) return ix->fail(HX_E_ARG, "element value size differs from the engine's row size");
                item_idx[page_base[blk] + off - 1] = (uint32_t)els.size();
                els.push_back(El{blk, off, lo});
            }
            blk = get32(pg + special);
        }
    }
    const uint32_t n = (uint32_t)els.size();
    if ((elem_blkno_out || elem_offno_out) && cap_elems < n) return ix->fail(HX_E_ARG, "cap_elems too small");
    auto lookup = [&](const TidRef &t) -> uint32_t {
        if (!t.valid() || t.blk >= n_pages) return 0xFFFFFFFFu;
        if (t.off > page_base[t.blk + 1] - page_base[t.blk]) return 0xFFFFFFFFu;
        return item_idx[page_base[t.blk] + t.off - 1];
    };
    // rows
    {
        const uint32_t CH = 16384; std::vector<uint8_t> buf((size_t)std::min(CH, std::max(n, 1u)) * payload);
        for (uint32_t i0 = 0; i0 < n; i0 += CH) {
            const uint32_t c = std::min(CH, n - i0);
            for (uint32_t i = 0; i < c; i++) memcpy(buf.data() + (size_t)i * payload, pages + (size_t)els[i0 + i].blk * PG_BLCKSZ + els[i0 + i].lo + ETUP_HDR + 8, payload);
            uint64_t first = 0;
            int rc = hx_append_rows(e, buf.data(), c, &first);
            if (rc) return ix->fail(rc, hx_last_error(e));
        }
    }
    // graph
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *t = pages + (size_t)els[i].blk * PG_BLCKSZ + els[i].lo;
        const int level = t[1];
        g.add(level);
        ix->loc_blk.push_back(els[i].blk); ix->loc_off.push_back((uint16_t)els[i].off); ix->loc_ver.push_back(t[3]);
        ix->loc2elem[((uint64_t)els[i].blk << 16) | els[i].off] = i;
        uint8_t nt = 0;
        for (uint32_t k = 0; k < HEAPTIDS; k++) {
            const TidRef h = get_tid(t + 4 + k * TID_BYTES);
            if (!h.valid()) break;                                                 // scan.rs:201-207
            g.tids[i][nt++] = (int64_t)(((uint64_t)h.blk << 16) | h.off);
        }
        g.ntids[i] = nt;
        const TidRef nref = get_tid(t + 64);
        uint32_t lo, ln;
        if (!item(nref.blk, nref.off, lo, ln)) continue;
        const uint8_t *nt_ = pages + (size_t)nref.blk * PG_BLCKSZ + lo;
        if (nt_[0] != 2 || nt_[1] != t[3] || get16(nt_ + 2) != (uint32_t)(level + 2) * m || ln < NTUP_HDR + (uint32_t)(level + 2) * m * TID_BYTES) continue;   // scan.rs:262-266
        for (int lc = level; lc >= 0; lc--) {
            const uint32_t lm = (uint32_t)g.lm(lc), start = (uint32_t)(level - lc) * m;
            Cand *lst = g.list(i, lc); uint16_t c = 0;
            for (uint32_t k = 0; k < lm; k++) {
                const TidRef r = get_tid(nt_ + NTUP_HDR + (start + k) * TID_BYTES);
                if (!r.valid()) break;                                             // scan.rs:275-277
                const uint32_t id = lookup(r);
                if (id == 0xFFFFFFFFu) continue;                                   // deleted / stale target: load_element would return None
                lst[c++] = Cand{0.0f, id};
            }
            g.cnt(i, lc) = c;
        }
        if (elem_blkno_out) elem_blkno_out[i] = els[i].blk;
        if (elem_offno_out) elem_offno_out[i] = (uint16_t)els[i].off;
    }
    for (uint32_t i = 0; i < n && (elem_blkno_out || elem_offno_out); i++) { if (elem_blkno_out) elem_blkno_out[i] = els[i].blk; if (elem_offno_out) elem_offno_out[i] = (uint16_t)els[i].off; }
    const TidRef ent{get32(meta + 16), get16(meta + 20)};
    const uint32_t ent_idx = lookup(ent);
    g.entry = ent_idx == 0xFFFFFFFFu ? -1 : (int64_t)ent_idx;
    // distances of every list, batched through K1 (query = the element's own row)
    {
        const uint32_t GCH = 1u << 18;
        std::vector<uint32_t> gq, goff, gids; std::vector<std::pair<uint32_t, int>> gl; std::vector<float> out;
        auto flush = [&]() -> int {
            if (gq.empty()) return HX_OK;
            out.resize(gids.size());
            int rc = hx_distances_batch(e, (uint32_t)gq.size(), gq.data(), goff.data(), gids.data(), out.data());
            if (rc) return ix->fail(rc, hx_last_error(e));
            for (size_t k = 0; k < gl.size(); k++) { Cand *lst = g.list(gl[k].first, gl[k].second); for (uint32_t j = goff[k]; j < goff[k + 1]; j++) lst[j - goff[k]].d = out[j]; }
            gq.clear(); goff.assign(1, 0u); gids.clear(); gl.clear();
            return HX_OK;
        };
        goff.assign(1, 0u);
        for (uint32_t i = 0; i < n; i++) {
            for (int lc = g.level[i]; lc >= 0; lc--) {
                const uint16_t c = g.cnt(i, lc); if (!c) continue;
                const Cand *lst = g.list(i, lc);
                gq.push_back(i); gl.emplace_back(i, lc);
                for (uint16_t k = 0; k < c; k++) gids.push_back(lst[k].id);
                goff.push_back((uint32_t)gids.size());
                ix->dirty.emplace_back(i, lc);
            }
            if (gq.size() >= GCH) { int rc = flush(); if (rc) return rc; }
        }
        int rc = flush(); if (rc) return rc;
    }
    *n_elems_out = n;
    return HX_OK;
}

} // extern "C"
