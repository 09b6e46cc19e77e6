// hx_fused_kernel.h -- the device-resident HNSW traversal kernel (one wavefront per search) and its launchers; included by one
// translation unit per element type (hx_fused_f32.hip / _f16.hip / _bit.hip) so that the instantiations compile in parallel.
#pragma once
#include "hx_fused_core.h"

// Algorithm 2 with entry points EP[0..n_ep); leaves the result set in the W heap (cx.CTL[1] = |W|).
// scan == false: search_layer (graph/mod.rs:161-255); scan == true: search_layer_disk without `discarded` (scan.rs:302-448).
#define F_TICK(k) do { if (tm) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t1_ = __builtin_amdgcn_s_memtime(); cx.tph[k] += (uint32_t)(t1_ - t0); t0 = t1_; } } while (0)
// ITER: search_layer_disk WITH the iterative scan's state (scan.rs:302-448): the visited set is the caller's and survives
// resumes (fresh == false keeps it; eps_visited == false: resume_scan_items' entry points are already in it), and every
// visited element that does not end in W goes to the `discarded` min-heap, in the reference's order of pushes.
template <class OP, int LPR, bool ITER = false, bool SKIP = false>
__device__ __forceinline__ void f_search_layer(KParams &p_in, FusedCtx &cx, uint32_t n_ep, uint32_t ef, int layer, bool scan, bool fresh = true, bool eps_visited = true)
{
    KParams &p = f_params_here(p_in);
    const uint32_t lane = cx.lane;
    // A greedy step (ef = 1) visits a few dozen ids: it uses the first 1 024 words of the table, so clearing costs 4 KB instead of 32-64 KB per
    // layer of the descent (the clears were 4 % of a query's memory traffic).  Any table size gives the same visited SET; one that fills up reports
    // FS_OVERFLOW and the task is retried with roomier tables, as for the full-size table.
    const uint64_t vis_words = (ef == 1u && !ITER && p.vis_words > 1024u) ? 1024u : p.vis_words;
    if (fresh) {   // fresh visited set
        for (uint64_t w = (uint64_t)lane * 4; w < vis_words; w += 256) *(u4 *)(cx.vis + w) = u4{VIS_EMPTY, VIS_EMPTY, VIS_EMPTY, VIS_EMPTY};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t vcount = fresh ? 0u : cx.vcount;
    if (eps_visited) {
        vcount += n_ep;
        for (uint32_t i = lane; i < n_ep; i += 64) (void)vis_test_and_set(cx.vis, (uint32_t)(vis_words >> 2) - 1u, cx.EP[i].y);
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
        else { F_BAR(); if (lane == 0) { uint32_t l = clen; FHeap<true>::push(cx.CH, l, it); } clen++; F_BAR(); }
    };
    auto c_pop = [&]() -> uint2 {
        if (clen <= cx.CH.L) return PHeap<true>::pop(CA, clen, lane);
        F_BAR();
        if (lane == 0) { uint32_t l = clen; const uint2 c = FHeap<true>::pop(cx.CH, l); cx.RES[0] = c; }
        clen--; F_BAR();
        const uint2 c = cx.RES[0]; F_BAR();
        return c;
    };
    // SKIP (vacuum's repair searches): members of the skip set are traversed but not counted (`should_count`, scan.rs:330-336, 416-419)
    auto counted = [&](uint32_t id) -> bool {
        if constexpr (!SKIP) return true;
        else return cx.skip == nullptr || !(id == cx.skip_self || cx.skip[id] != 0);
    };
    for (uint32_t i = 0; i < n_ep; i++) {
        if (clen >= p.ccap) { cx.status = FS_OVERFLOW; break; }
        const uint2 it = cx.EP[i];
        if (SKIP && wl + 1u >= p.wcap) { cx.status = FS_OVERFLOW; break; }          // uncounted members made W outgrow its LDS array: the lock-step driver takes the task
        c_push(it); PHeap<false>::push(WA, wl, it, lane);
        if (counted(it.y)) rlen++;
    }
    F_BAR();
    const uint32_t bmask = (uint32_t)(vis_words >> 2) - 1u;
    const bool ahead = layer == 0 && (p.fdbg & 8u);
    uint32_t pf_cid = 0xffffffffu, pf_e = 0u, pf_n = 0u;
    for (;;) {
        if (cx.status != FS_OK) break;
        // pop the nearest candidate, decide whether to stop
        const bool tm = FUSED_TIMERS_ON && (p.fdbg & 4u) != 0; unsigned long long t0 = tm ? __builtin_amdgcn_s_memtime() : 0ull;
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
        bool probed = false; u4 pbk = {0u, 0u, 0u, 0u};
        if (go) {
            if (layer == 0) { nb = p.l0_ids + (size_t)cid * 2u * p.m; lmax = 2u * p.m; }
            else { nb = p.up_ids + (size_t)(p.up_block[cid] + (uint32_t)(layer - 1)) * p.m; lmax = p.m; clevel = p.level[cid]; }
            if (ahead && cid == pf_cid) {                                            // the list came in during the previous replay
                e_first = pf_e; n = pf_n; probed = true;
                if (lane < n) pbk = vis_load_bucket(cx.vis, vis_mix(e_first) & bmask);   // in flight during the pop
            } else {
                e_first = lane < lmax ? nb[lane] : 0u;                               // issued together with the count: one memory hop
                if (layer == 0) n = p.l0_cnt[cid]; else n = p.up_cnt[p.up_block[cid] + (uint32_t)(layer - 1)];
            }
        }
        pf_cid = 0xffffffffu;
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
            uint32_t *vslot = nullptr; uint32_t vold = VIS_EMPTY;                    // insert in flight (settled below)
            if (idx < n) {
                e = n0 == 0 ? e_first : nb[idx];
                if (probed && n0 == 0) unvis = !vis_lookup_from(cx.vis, bmask, e, vslot, pbk);
                else unvis = !vis_lookup(cx.vis, bmask, e, vslot);                   // visited.contains / insert, mod.rs:206-209
                if (unvis) vold = atomicCAS(vslot, VIS_EMPTY, e);
                if (unvis && layer > 0 && p.level[e] < layer) unvis = false;         // mod.rs:213-216
            }
            F_TICK(2);
            const unsigned long long mask = __ballot(unvis);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            vcount += cnt;
            if (vcount * 4u > (uint32_t)vis_words * 3u) { cx.status = FS_OVERFLOW; break; }   // table too full: re-run in the lock-step path
            if (cnt == 0) { vis_settle(cx.vis, bmask, e, vslot, vold); continue; }
            if (unvis) cx.IDS[__popcll(mask & ((1ull << lane) - 1ull))] = e;
            if constexpr (FUSED_BAR_BEFORE_ROWS) F_BAR(); else F_WSYNC();   // the ids are read back by this wave (LDS is in order); no wait for the visited-set CAS still in flight: it is settled after the row loads
            F_TICK(3);
            const float mine = f_dist<OP, LPR>(cx, cx.QV, cx.IDS, cnt, lane, tm ? cx.tph : nullptr);
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
            F_BAR();
            if (ahead && n0 + 64u >= n && clen <= cx.CH.L) {                         // look-ahead: who is popped next?
                float best = 0.0f; uint32_t bid = 0u; bool have = false;
                if (clen > 0) { const uint2 r = PHeap<true>::ld(CA, 0u); best = fh_d(r); bid = r.y; have = true; }
                unsigned long long cm = __ballot(keep && (!have || mine < best));
                while (cm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(cm); cm &= cm - 1ull;
                    const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), (int)j));
                    if (!have || dj < best) { best = dj; bid = cx.IDS[j]; have = true; }
                }
                if (have) {
                    bid = (uint32_t)__builtin_amdgcn_readfirstlane((int)bid);
                    pf_cid = bid;
                    pf_e = lane < 2u * p.m ? p.l0_ids[(size_t)bid * 2u * p.m + lane] : 0u;
                    pf_n = p.l0_cnt[bid];
                }
            }
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
                if (SKIP && wl + 1u >= p.wcap) { cx.status = FS_OVERFLOW; break; }
                c_push(it); PHeap<false>::push(WA, wl, it, lane);
                if (counted(it.y)) rlen++;
                if (tm) cx.tph[8]++;
                if (clen > cx.cmax) cx.cmax = clen;
                if (rlen > ef) {
                    const uint2 ev = PHeap<false>::pop(WA, wl, lane); rlen--;
                    if (ITER) { d_push(ev); if (cx.status != FS_OK) break; }        // scan.rs:423-428
                }
            }
            if (ITER) d_flush();
            F_BAR();
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
    F_BAR();
}

#ifdef HX_EXPERIMENTS
// search_layer / search_layer_disk on ONE sorted array (the `SA` kernels; SURVEY 8 row a1 / a4, same results as f_search_layer).
//
// Without ties the two heaps of Algorithm 2 carry redundant state: every element of C was pushed to W at the same moment (mod.rs:236-241), an element
// that W evicted is at least as far as everything W keeps and W's furthest only moves closer, so it fails the stop test `c.d > f.d` (mod.rs:188-193)
// whenever it reaches C's root -- the candidates that can still be expanded are exactly the unexpanded members of W.  The kernel therefore keeps W as
// an ascending array in LDS with an "expanded" flag per entry (bit 31 of the id): the next candidate is the first unflagged entry (K = ceil(ef/64)
// ballots), the search ends when there is none, and the rows of an expansion are merged in together -- each entry counts the new rows in front of it,
// each new row the entries and new rows in front of it, one round of LDS writes -- instead of a sift per push and per eviction.  Replaying the list in
// order (mod.rs:226-243) leaves the ef nearest of W + the rows that pass `d < f.d`; so does the merge.
// A TIE changes that: which of two equal distances a binary heap pops first depends on its array, and an evicted element at exactly W's furthest
// distance would still be expanded.  Every pair of equal distances that meets inside W is adjacent in the array when the second one arrives, so the
// merge sees it; the search then reports FS_OVERFLOW and the task is redone by the heap kernel (k_fused<.., SA = false>, the retry launch), which
// reproduces Rust's BinaryHeap bit by bit.  Integer-valued metrics (Hamming, Jaccard) tie all the time and never take this path.
//
// Measured (1M x 768, 10 000 queries, profiles/r02_rocprofv3_pmc_insts_sa*.txt): 39 % fewer VALU and 34 % fewer SALU instructions per expansion, 102
// VGPRs without scratch, no candidate heap in LDS; the launch takes 7.5 ms instead of 9.3.  But 4.5 % of the queries meet a tie -- two of a search's
// ~1 700 f32 distances coincide more often than one would think -- and their retry launch is one search latency long (3.9 ms) however few they are;
// redoing them inside the same kernel needs the heap code and its registers next to this one and was slower than the heap kernel alone.
// Opt-in: HX_SORTED_ARRAY=1.
template <class OP, int LPR>
__device__ void f_search_layer_sa(KParams &p, FusedCtx &cx, uint32_t n_ep, uint32_t ef, int layer)
{
    constexpr uint32_t XF = 0x80000000u;
    const uint32_t lane = cx.lane;
    const uint64_t vis_words = (ef == 1u && p.vis_words > 1024u) ? 1024u : p.vis_words;      // as f_search_layer: a greedy step touches a few dozen ids
    for (uint64_t w = (uint64_t)lane * 4; w < vis_words; w += 256) *(u4 *)(cx.vis + w) = u4{VIS_EMPTY, VIS_EMPTY, VIS_EMPTY, VIS_EMPTY};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t bmask = (uint32_t)(vis_words >> 2) - 1u;
    uint32_t vcount = n_ep;
    for (uint32_t i = lane; i < n_ep; i += 64) (void)vis_test_and_set(cx.vis, bmask, cx.EP[i].y);
    lds_uint2 *const A = (lds_uint2 *)cx.W;
    const uint32_t K = (ef + 63u) >> 6;                                  // <= 4 (the host keeps ef <= 256 on this path)
    // the entry points arrive ascending (one element, or the previous layer's result); anything else goes to the heap kernel
    bool bad = n_ep > ef;
    for (uint32_t i = lane; i < n_ep; i += 64) {
        const uint2 e = cx.EP[i];
        if (i + 1u < n_ep && !(fh_d(e) < fh_d(cx.EP[i + 1u]))) bad = true;
        A[i].x = e.x; A[i].y = e.y;
    }
    if (__ballot(bad) != 0ull) cx.status = FS_OVERFLOW;
    uint32_t len = n_ep;
    F_BAR();
    while (cx.status == FS_OK) {
        const bool tm = FUSED_TIMERS_ON && (p.fdbg & 4u) != 0; unsigned long long t0 = tm ? __builtin_amdgcn_s_memtime() : 0ull;
        // the nearest unexpanded member of W (mod.rs:187-193)
        uint32_t pos = 0xffffffffu;
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            if (k < K) {
                const uint32_t i = lane + 64u * k;
                const bool un = i < len && !(A[i].y & XF);
                const unsigned long long um = __ballot(un);
                if (um != 0ull && pos == 0xffffffffu) pos = 64u * k + (uint32_t)__builtin_ctzll(um);
            }
        }
        if (pos == 0xffffffffu) break;
        const uint32_t cid = (uint32_t)__builtin_amdgcn_readfirstlane((int)A[pos].y);
        const uint32_t *nb; uint32_t n, lmax; int32_t clevel = 0x7fffffff;
        if (layer == 0) { nb = p.l0_ids + (size_t)cid * 2u * p.m; lmax = 2u * p.m; }
        else { nb = p.up_ids + (size_t)(p.up_block[cid] + (uint32_t)(layer - 1)) * p.m; lmax = p.m; clevel = p.level[cid]; }
        const uint32_t e_first = lane < lmax ? nb[lane] : 0u;
        if (layer == 0) n = p.l0_cnt[cid]; else n = p.up_cnt[p.up_block[cid] + (uint32_t)(layer - 1)];
        if (lane == 0) A[pos].y = cid | XF;
        F_TICK(0);
        if (tm) cx.tph[7]++;
        if (layer > 0 && clevel < layer) continue;                               // mod.rs:198-200
        for (uint32_t n0 = 0; n0 < n; n0 += 64) {
            const uint32_t idx = n0 + lane;
            uint32_t e = 0; bool unvis = false;
            uint32_t *vslot = nullptr; uint32_t vold = VIS_EMPTY;
            if (idx < n) {
                e = n0 == 0 ? e_first : nb[idx];
                unvis = !vis_lookup(cx.vis, bmask, e, vslot);                    // mod.rs:206-209
                if (unvis) vold = atomicCAS(vslot, VIS_EMPTY, e);
                if (unvis && layer > 0 && p.level[e] < layer) unvis = false;     // mod.rs:213-216
            }
            F_TICK(2);
            const unsigned long long mask = __ballot(unvis);
            const uint32_t cnt = (uint32_t)__popcll(mask);
            vcount += cnt;
            if (vcount * 4u > (uint32_t)vis_words * 3u) { cx.status = FS_OVERFLOW; break; }
            if (cnt == 0) { vis_settle(cx.vis, bmask, e, vslot, vold); continue; }
            if (unvis) cx.IDS[__popcll(mask & ((1ull << lane) - 1ull))] = e;
            if constexpr (FUSED_BAR_BEFORE_ROWS) F_BAR(); else F_WSYNC();   // the ids are read back by this wave (LDS is in order); no wait for the visited-set CAS still in flight: it is settled after the row loads
            F_TICK(3);
            const float mine = f_dist<OP, LPR>(cx, cx.QV, cx.IDS, cnt, lane, tm ? cx.tph : nullptr);
            F_TICK(4);
            vis_settle(cx.vis, bmask, e, vslot, vold);
            cx.nd0 += cnt;
            // rows that can enter W: all of them while W is short of ef, else those nearer than its furthest member (mod.rs:226-232)
            const bool full = len >= ef;
            const float fmax = len ? __builtin_bit_cast(float, (unsigned int)A[len - 1u].x) : 0.0f;
            const bool keep = lane < cnt && (!full || mine < fmax);
            unsigned long long km = __ballot(keep);
            F_TICK(5);
            if (km != 0ull) {
                const uint32_t nk = (uint32_t)__popcll(km);
                const uint32_t myid = lane < cnt ? cx.IDS[lane] : 0u;
                uint2 v[4]; uint32_t sh[4]; bool valid[4];
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint32_t i = lane + 64u * k;
                    valid[k] = k < K && i < len; sh[k] = 0u;
                    v[k] = valid[k] ? make_uint2(A[i].x, A[i].y) : make_uint2(0u, 0u);
                }
                bool tie = false; uint32_t mypos = 0u;
                while (km) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(km); km &= km - 1ull;
                    const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mine), (int)j));
                    uint32_t before = 0u;                                         // members of W in front of row j
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++) {
                        if (k < K) {
                            const float dk = fh_d(v[k]);
                            const bool gt = valid[k] && dk > dj;
                            tie |= valid[k] && dk == dj;
                            sh[k] += gt ? 1u : 0u;
                            before += (uint32_t)__popcll(__ballot(valid[k] && !gt));
                        }
                    }
                    before += (uint32_t)__popcll(__ballot(keep && mine < dj));    // new rows in front of row j
                    tie |= keep && mine == dj && lane != j;
                    if (lane == j) mypos = before;
                }
                if (__ballot(tie) != 0ull) { cx.status = FS_OVERFLOW; break; }
                F_WSYNC();
#pragma unroll
                for (uint32_t k = 0; k < 4u; k++) {
                    const uint32_t i = lane + 64u * k + sh[k];
                    if (valid[k] && sh[k] != 0u && i < ef) { A[i].x = v[k].x; A[i].y = v[k].y; }
                }
                if (keep && mypos < ef) { A[mypos].x = __builtin_bit_cast(unsigned int, mine); A[mypos].y = myid; }
                len = len + nk < ef ? len + nk : ef;
                if (tm) cx.tph[8] += nk;
                F_WSYNC();
            }
            F_TICK(6);
        }
    }
    for (uint32_t i = lane; i < len; i += 64) A[i].y &= ~XF;
    if (lane == 0) cx.CTL[1] = len;
    F_BAR();
}

#endif  // HX_EXPERIMENTS

// stable sort of the W heap's internal array into EP: ascending (build, mod.rs:248-254) or descending (scan.rs:441-446);
// rank sort: ties keep their order in W's array, exactly what a stable sort of that array does
__device__ __forceinline__ void f_sort_results(FusedCtx &cx, uint32_t n, bool desc)
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
    F_BAR();
}

// One search after the other, from the task counter, driven by ONE wavefront whose state lives in the LDS slot `lds` (`slot` selects its
// visited table / spill area / discarded heap in global memory).  MODE 0: query (get_scan_items), 1: insert (find_element_neighbors),
// 2: iterative scan; LPR: lanes per row (64, or 8/32 for short rows).
template <class OP, int MODE, int LPR, bool SA = false>
__device__ __forceinline__ void f_worker(KParams &p_in, uint8_t *lds, const uint32_t slot, const uint32_t lane_in)
{
    FusedCtx cx;
    KParams &p = f_params_here(p_in);
    constexpr bool INS = MODE == 1 || MODE == 3;           // find_element_neighbors; MODE 3 stops after each layer's search (select on the matrix cores)
    const uint32_t lm0 = 2u * p.m;
    // LDS carve, fixed-size regions first so that their addresses are compile-time constants (no scalar register holds them):
    //   IDS[64] u32 | CTL[32] u32 | dsc[64] f32 | RES[64] | RL[64] | QV[nch KiB] | W[wcap] | EP[wcap] | C[clds] | MODE 2: DP[64] WS[160] LV[32] DS[disc_lds]   (wcap = ef + 2, more for repair searches)
    // (8-byte entries unless noted; hx_fused.inc.h sizes the allocation).  The select phase runs after the layer's search is over, so its scratch
    // (the candidate under test EV and the discarded list DL) reuses C.
    cx.IDS = (uint32_t *)lds;
    cx.CTL = cx.IDS + 64;
    cx.fr.dsc = (float *)(cx.CTL + 32);
    cx.RES = (uint2 *)(cx.fr.dsc + 64);
    cx.RL = cx.RES + 64;
    cx.QV = (lds_u8 *)(cx.RL + 64);                       // query parked in LDS (nch KiB)
    cx.W = (uint2 *)(cx.QV + p.nch * 1024u);
    cx.EP = cx.W + p.wcap;
    cx.C = cx.EP + p.wcap;
    cx.DP = (lds_uint2 *)(cx.C + p.clds);                 // MODE 2: queue of pending `discarded` pushes (64 entries), then the heap's LDS head
    cx.WS = cx.DP + 64; cx.LV = (uint32_t *)(cx.WS + 160);   // working set of a flush (<= 2*64 + depth entries), per-level {first index, offset}
    cx.DS.A = cx.WS + 160 + 32; cx.DS.L = MODE == 2 ? p.disc_lds : 0u;   // MODE 2: LDS head of the `discarded` heap
    cx.DS.G = MODE == 2 ? p.disc + (size_t)slot * p.disc_stride : nullptr; cx.dlen = 0; cx.vcount = 0;
    cx.EV = (lds_u8 *)cx.C;                               // host guarantees clds*8 >= nch*1024 + (ef+2)*8
    cx.DL = (uint2 *)(cx.EV + p.nch * 1024u);
    cx.CH.lds = (lds_uint2 *)cx.C; cx.CH.glob = p.spill + (size_t)slot * p.spill_stride; cx.CH.L = p.clds;
    cx.WH.lds = (lds_uint2 *)cx.W; cx.WH.glob = nullptr; cx.WH.L = 0xffffffffu;
    cx.lane = lane_in;
    cx.vis = p.vis + (size_t)slot * p.vis_words;
    cx.nd0 = cx.nd1 = 0; cx.cmax = 0;
    for (int i = 0; i < 14; i++) cx.tph[i] = 0;
    const uint32_t lane = cx.lane;

    for (;;) {
        KParams &p = f_params_here(p_in);                  // per task: what the task derives from the parameters does not outlive it
        cx.fr.rows = p.rows; cx.fr.pitch = p.pitch; cx.fr.nch = p.nch; cx.fr.cap = p.sparse_cap;
        if (lane == 0) cx.CTL[5] = atomicAdd(p.next_task, 1u);
        F_BAR();
        const uint32_t t = cx.CTL[5];
        F_BAR();
        if (t >= p.ntasks) break;
        cx.status = FS_OK;
        const uint32_t qsel = p.t_qsel[t];
        const uint8_t *qsrc = (qsel & HX_QUERY_SLOT) ? p.queries + (size_t)(qsel & 0x7fffffffu) * p.pitch : p.rows + (size_t)qsel * p.pitch;
        f_park(cx.fr, qsrc, lane, cx.QV);
        const int new_level = INS ? p.t_level[t] : -1;
        // MODE 1 outputs are addressed through strides so that one launch can fill either the SoA staging arrays or the AoS records
        // of a batch's exchange buffer (hx_batch.hip); os = output slot of this task
        const uint32_t os = (INS && p.t_oslot) ? p.t_oslot[t] : t;
        if (INS && new_level >= FUSED_MAXL) { if (lane == 0) p.status[t] = FS_HOST; continue; }

        // d(q, entry point): mod.rs:371-377 / scan.rs:475
        const uint32_t entry = (MODE == 3 && p.t_entry) ? p.t_entry[t] : p.entry;
        const int entry_level = (MODE == 3 && p.t_entry) ? p.level[entry] : p.entry_level;
        cx.skip = MODE == 3 ? p.skip : nullptr; cx.skip_self = qsel;               // the repaired element is its own query row (vacuum.rs:331-333)
        if (lane == 0) cx.IDS[0] = entry;
        F_BAR();
        const float d0 = f_dist<OP, LPR>(cx, cx.QV, cx.IDS, 1, lane);
        cx.nd0 += 1;
        if (lane == 0) cx.EP[0] = fh_pack(d0, entry);
        F_BAR();
        uint32_t n_ep = 1;

        // greedy descent with ef = 1: mod.rs:385-399 (down to new_level+1) / scan.rs:491-512 (down to 1)
        const int stop_above = INS ? new_level : 0;
        // search_layer_disk's semantics (f64 comparisons, results nearest LAST, that order as the next layer's entry points): scans, and MODE 3 when it
        // serves aminsert's find_element_neighbors_on_disk (insert.rs:1021-1123; p.ondisk) instead of the build's find_element_neighbors
        const bool scan_sem = !INS || (MODE == 3 && p.ondisk != 0u);
        for (int lc = entry_level; lc > stop_above && cx.status == FS_OK; lc--) {
#ifdef HX_EXPERIMENTS
            if constexpr (SA) {
                f_search_layer_sa<OP, LPR>(p, cx, n_ep, 1u, lc);
                if (cx.CTL[1] > 0) { const uint2 best = cx.W[0]; F_BAR(); if (lane == 0) cx.EP[0] = best; F_BAR(); n_ep = 1; }
                else if (!INS) { n_ep = 0; break; }
                continue;
            }
#endif
            f_search_layer<OP, LPR, false, MODE == 3>(p, cx, n_ep, 1u, lc, scan_sem);
            const uint32_t wl = cx.CTL[1];
            if (wl > 0) {
                f_sort_results(cx, wl, scan_sem);
                if (scan_sem) { const uint2 best = cx.EP[wl - 1]; F_BAR(); if (lane == 0) cx.EP[0] = best; F_BAR(); }   // w.into_iter().last(): scan.rs:506-510 / insert.rs:1069-1073
                n_ep = 1;                         // MODE 1: ep = vec![w[0]]; EP[0] already is the nearest
            } else if (scan_sem) { n_ep = 0; break; }
        }

        if (MODE == 2) {
            // get_scan_items + the amgettuple loop of an iterative scan (scan.rs:458-577, 794-875) for one query
            uint32_t outc = 0; long long tuples = 0; double prev = -__builtin_inf();
            cx.dlen = 0; cx.vcount = 0;
            const size_t obase = (size_t)t * p.limit;
            if (cx.status == FS_OK && n_ep > 0) {
                // ONE call site for the layer-0 search with the scan's state (the first search, scan.rs:515-528, and every resume, scan.rs:553-575): the
                // function is inlined once -- called from two places it stayed an out-of-line function, and its call frame was the kernel's scratch
                bool first = true, single = false;
                for (;;) {
                    if (!single) f_search_layer<OP, LPR, true>(p, cx, n_ep, p.ef, 0, true, first, first);
                    first = false;
                    if (cx.status != FS_OK) break;
                    const uint32_t wl = single ? 1u : cx.CTL[1];
                    if (!single) f_sort_results(cx, wl, true);                                   // EP[0..wl): nearest LAST
                    // emit from the back (scan.rs:796-815, 860-874); the elements' TID masks are fetched 64 at a time
                    uint32_t left = wl;
                    while (left > 0 && outc < p.limit) {
                        const uint32_t c = left < 64u ? left : 64u, base = left - c;
                        if (lane < c) cx.IDS[lane] = p.emask[cx.EP[base + lane].y];
                        F_BAR();
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
                        F_BAR();
                        left = base;
                    }
                    if (outc >= p.limit) break;
                    if (tuples >= p.max_tuples) {                                                // scan.rs:831-841: drain `discarded` one by one
                        if (cx.dlen == 0) break;
                        const uint2 one = PHeap<true>::pop(cx.DS, cx.dlen, lane);
                        F_BAR(); if (lane == 0) cx.EP[0] = one; F_BAR();
                        single = true;
                        continue;
                    }
                    if (cx.dlen == 0) break;                                                     // resume_scan_items, scan.rs:548-550
                    single = false;
                    n_ep = 0;
                    while (n_ep < p.ef && cx.dlen > 0) {
                        const uint2 x = PHeap<true>::pop(cx.DS, cx.dlen, lane);
                        F_BAR(); if (lane == 0) cx.EP[n_ep] = x; F_BAR();
                        n_ep++;
                    }
                }
            }
            if (lane == 0) { p.out_cnt[t] = outc; p.status[t] = cx.status; }
        } else if (MODE == 0) {
            uint32_t cnt = 0;
            if (cx.status == FS_OK && n_ep > 0) {
#ifdef HX_EXPERIMENTS
                if constexpr (SA) {
                    f_search_layer_sa<OP, LPR>(p, cx, n_ep, p.ef, 0);                        // scan.rs:515-528; W comes out ascending
                    const uint32_t wl = cx.CTL[1];
                    cnt = cx.status != FS_OK ? 0u : wl < p.k ? wl : p.k;
                    for (uint32_t i = lane; i < cnt; i += 64) {
                        const uint2 v = cx.W[i];
                        p.out_ids[(size_t)t * p.k + i] = v.y; p.out_d[(size_t)t * p.k + i] = fh_d(v);
                    }
                } else
#endif
                {
                f_search_layer<OP, LPR, false>(p, cx, n_ep, p.ef, 0, true);                  // scan.rs:515-528
                const uint32_t wl = cx.CTL[1];
                f_sort_results(cx, wl, true);                                        // nearest LAST
                cnt = wl < p.k ? wl : p.k;
                for (uint32_t i = lane; i < cnt; i += 64) {                          // amgettuple pops from the back
                    const uint2 v = cx.EP[wl - 1 - i];
                    p.out_ids[(size_t)t * p.k + i] = v.y; p.out_d[(size_t)t * p.k + i] = fh_d(v);
                }
                }
            }
            if (lane == 0) { p.out_cnt[t] = cnt; p.status[t] = cx.status; }
        } else {
            const int start = new_level < entry_level ? new_level : entry_level;
            const size_t obase = (size_t)os * p.o_cst;
            for (uint32_t i = lane; i < FUSED_MAXL; i += 64) p.out_cnt[obase + i] = 0;
            KParams &p_task = p;
            for (int lc = start; lc >= 0 && cx.status == FS_OK; lc--) {
                f_search_layer<OP, LPR, false, MODE == 3>(p_task, cx, n_ep, p_task.ef, lc, scan_sem);   // mod.rs:407-416 / insert.rs:1088-1101
                if (cx.status != FS_OK) break;
                KParams &p = f_params_here(p_task);            // the select phase derives its own view of the parameters (dead again when the layer is done)
                const uint32_t lm = lc == 0 ? lm0 : p.m;
                const uint32_t wl = cx.CTL[1];
                f_sort_results(cx, wl, scan_sem);                                    // W ascending (build; on-disk: nearest last); also the next layer's entry points (mod.rs:425, insert.rs:1119)
                n_ep = wl;
                if (lc == 0 && p.wtab) {
                    // W (ids + distance bits) as a hash table for the back-link kernels: built in the candidate heap's LDS (dead until the
                    // next search), copied out coalesced.  d(new, x) there is the very value a back-link prune would recompute.
                    lds_uint2 *T = cx.CH.lds; const uint32_t tm = p.wt_size - 1u;
                    for (uint32_t i = lane; i < p.wt_size; i += 64) { T[i].x = 0u; T[i].y = VIS_EMPTY; }
                    F_BAR();
                    for (uint32_t i = lane; i < wl; i += 64) {
                        const uint2 e = cx.EP[i];
                        uint32_t s = vis_mix(e.y) & tm;
                        while (atomicCAS((uint32_t *)&T[s].y, VIS_EMPTY, e.y) != VIS_EMPTY) s = (s + 1u) & tm;
                        T[s].x = e.x;
                    }
                    F_BAR();
                    uint2 *dst = p.wtab + (size_t)(p.wt_slot0 + os) * p.wt_size;
                    for (uint32_t i = lane; i < p.wt_size; i += 64) dst[i] = make_uint2(T[i].x, T[i].y);
                    if (lane == 0 && p.wt_valid) p.wt_valid[p.wt_slot0 + os] = 1;
                    F_BAR();
                }
                if constexpr (MODE == 3) {                                           // W (ascending) leaves the kernel; select_neighbors follows in hx_mfma.hip
                    const size_t pr = (size_t)p.t_prob[t] + (size_t)lc;
                    uint2 *dst = p.wl_out + pr * p.wcap_out;
                    for (uint32_t i = lane; i < wl; i += 64) dst[i] = cx.EP[i];
                    if (lane == 0) p.wl_cnt[pr] = wl;
                    F_BAR();
                    continue;
                }
                // select_neighbors(W, lm): mod.rs:269-308
                const unsigned long long ts0 = (FUSED_TIMERS_ON && (p.fdbg & 4u)) ? __builtin_amdgcn_s_memtime() : 0ull;
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
                    lds_u8 *const evb[2] = {cx.EV, cx.QV};
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
                        F_BAR();
                        const uint32_t my_id = nx_id; const float my_d = nx_d;       // e's list slot of this lane
                        if (i + 1u < wl) { const uint32_t en = cx.EP[i + 1u].y; f_park_async(cx.fr, p.rows + (size_t)en * p.pitch, lane, evb[(i + 1u) & 1u]); list_prefetch(en); }
                        if (r > 0 && i > 0) {
                            bool known_hit = false;
                            if (my_id != 0xFFFFFFFFu && my_d <= fh_d(e)) for (uint32_t j = 0; j < r; j++) known_hit |= cx.RL[j].y == my_id;
                            if (__ballot(known_hit) != 0ull) closer = false;         // mod.rs:333-335 with a distance we already hold
                        }
                        if (r > 0 && closer) {
                            if (lane < r) cx.IDS[lane] = cx.RL[lane].y;
                            F_BAR();
                            closer = !f_any_le_x<OP, LPR>(cx, evb[i & 1u], cx.IDS, r, lane, fh_d(e), cx.nd1);   // mod.rs:324-336
                            F_BAR();
                        }
                        if (lane == 0) { if (closer) cx.RL[r] = e; else cx.DL[nd] = e; }
                        if (closer) r++; else nd++;
                        F_BAR();
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // no row may still be in flight towards the query's slot
                    F_BAR();
                    f_park(cx.fr, qsrc, lane, cx.QV);                                    // the query again, for the next layer's search
                    if (lane == 0) for (uint32_t j = 0; j < nd && r < lm; j++) cx.RL[r++] = cx.DL[j];   // mod.rs:300-305
                    r = __shfl(r, 0, 64);
                }
                F_BAR();
                const size_t lb = (size_t)os * p.o_lst + (size_t)lc * lm0;
                for (uint32_t i = lane; i < r; i += 64) { const uint2 v = cx.RL[i]; p.out_ids[lb + i] = v.y; p.out_d[lb + i] = fh_d(v); }
                if (lane == 0) p.out_cnt[obase + lc] = r;
                F_BAR();
                if (FUSED_TIMERS_ON && (p.fdbg & 4u)) cx.tph[13] += (uint32_t)(__builtin_amdgcn_s_memtime() - ts0);
            }
            if (lane == 0) p.status[t] = cx.status;
        }
        F_BAR();
    }
    if (lane == 0) { atomicAdd(&p.n_dist[0], cx.nd0); atomicAdd(&p.n_dist[1], cx.nd1); atomicMax(&p.n_dist[2], (unsigned long long)cx.cmax);
                     if (FUSED_TIMERS_ON && (p.fdbg & 4u)) { for (int i = 0; i < 14; i++) atomicAdd(&p.n_dist[3 + i], (unsigned long long)cx.tph[i]); } }
}

template <class OP, int MODE, int LPR, bool SA = false>
__global__ void __launch_bounds__(64, (MODE == 2 ? FUSED_MINW_ITER : (MODE == 1 || MODE == 3) ? FUSED_MINW_INS : SA ? FUSED_MINW_SA : OP::query_minw ? OP::query_minw : FUSED_MINW))
k_fused(const FusedParams p_unused)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    KParams *kp = (KParams *)__builtin_amdgcn_kernarg_segment_ptr();      // the parameter block is the kernel's only argument: offset 0 of the kernarg segment
    asm volatile("" : "+s"(kp));                                          // laundered (hx_fused_core.h: KParams)
    f_worker<OP, MODE, LPR, SA>(*kp, lds, blockIdx.x, threadIdx.x);
}


template <class OP, int MODE, int LPR, bool SA = false>
static hipError_t launch_fused(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds)
{
    static thread_local bool attr_set = false;
    if (!attr_set) {
        hipError_t s = hipFuncSetAttribute((const void *)k_fused<OP, MODE, LPR, SA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        if (s != hipSuccess) return s;
        attr_set = true;
    }
    if (getenv("HX_DEBUG")) {
        static thread_local bool once = false;
        if (!once) { once = true; int nb = -1; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)k_fused<OP, MODE, LPR, SA>, 64, lds);
            fprintf(stderr, "[hx] k_fused<mode %d, %d lanes/row, sorted-array %d>: dynamic LDS %zu B, grid %u, occupancy API says %d blocks/CU\n", MODE, LPR, (int)SA, lds, grid, nb); }
    }
    hipLaunchKernelGGL((k_fused<OP, MODE, LPR, SA>), dim3(grid), dim3(64), lds, e->fused_stream ? e->fused_stream : e->stream, p);
    return hipGetLastError();
}


template <class OP, int LPR>
static hipError_t launch_fused_lpr(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{
    if (mode == 2) return launch_fused<OP, 2, LPR>(e, p, grid, lds);
#ifdef HX_EXPERIMENTS
    if constexpr (OP::sorted_array_ok) { if (p.sa && mode == 0) return launch_fused<OP, 0, LPR, true>(e, p, grid, lds); }
#endif
    if (mode == 3) return launch_fused<OP, 3, LPR>(e, p, grid, lds);      // search only: the matrix-core select (hx_mfma.hip) and aminsert's nearest-lm (insert.rs:1111-1117) take the W lists
    return mode == 0 ? launch_fused<OP, 0, LPR>(e, p, grid, lds) : launch_fused<OP, 1, LPR>(e, p, grid, lds);
}
template <class OP>
static hipError_t launch_fused_mode(hx_engine *e, const FusedParams &p, uint32_t grid, size_t lds, int mode)
{   // lanes per row by payload, as in launch_links_cached
    if (e->pitch <= 128) return launch_fused_lpr<OP, 8>(e, p, grid, lds, mode);
    if (e->pitch <= 512) return launch_fused_lpr<OP, 32>(e, p, grid, lds, mode);
    return launch_fused_lpr<OP, 64>(e, p, grid, lds, mode);
}

