// hx_fused.inc.h -- included by hx_engine.hip: the host side of the device-resident traversal (graph mirror maintenance, fused_run).
//
// The lock-step host driver (hx_index.cpp) pays one host round trip per candidate expansion.  The traversal kernel
// (hx_fused_kernel.h, compiled per element type in hx_fused_{f32,f16,bit}.hip) moves the whole of search_layer
// (graph/mod.rs:161-255 / scan.rs:302-448), the greedy descent and per-layer loop of find_element_neighbors (graph/mod.rs:355-427) /
// get_scan_items (scan.rs:458-530), the iterative-scan loop (scan.rs:794-875) and select_neighbors (graph/mod.rs:269-339) into ONE
// persistent kernel:
//   * one wavefront owns one insert or query from start to finish; waves pull tasks from an atomic counter until none are left
//     (every wave reaches the exit: the counter only grows);
//   * the candidate heap C (LDS head, tail spilled to a per-wave area in global memory), the result heap W, the entry-point /
//     sorted-candidate array and the select lists live in LDS; both heaps are driven by the WHOLE wave (PHeap: ancestor gather + ballot +
//     scatter) to the array states Rust's BinaryHeap produces, as the host driver and the oracle do;
//   * the query is parked in LDS; an expansion reads the candidate's neighbour ids (coalesced), tests them against a per-wave visited
//     hash table in global memory (16-byte buckets, one L1-bypassing load per test, the insert's CAS settled after the row loads) and
//     evaluates the unvisited rows in the canonical summation order of K1/K2 (lane l owns bytes chunk*1024+16*l, xor butterfly),
//     4 rows x 3 chunks in flight, loads issued unconditionally from scalar row bases;
//   * the graph is read from the device mirror, which during a build is the authoritative copy (hx_batch.hip).
// Results (neighbour lists with distances / top-k) are bit-identical to the lock-step path; the tests compare the two paths and the oracle.
// A task whose tables overflow reports FS_OVERFLOW, is retried once on the device with 8x the tables and only then re-run by the lock-step
// path (still on the GPU kernels: there is no CPU fallback).

#include "hx_fused_core.h"

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

// The sorted-array search is a tie-inexact experiment: compiled only with HX_CFLAGS=-DHX_EXPERIMENTS, selected by HX_SORTED_ARRAY=1.
// (k_fused2, the pooled-stream-waves kernel of round 2 -- measured 0.29-0.38 of peak against 0.50 -- was removed in round 3; it is in the history.)
#ifdef HX_EXPERIMENTS
static int sa_env_on() { return getenv("HX_SORTED_ARRAY") ? atoi(getenv("HX_SORTED_ARRAY")) : 0; }
#else
static int sa_env_on() { return 0; }
#endif

// mode 0: ntasks queries -> out_ids/out_d [ntasks][k], out_cnt[ntasks]; mode 1: ntasks inserts -> out_ids/out_d
// [ntasks][FUSED_MAXL][2m], out_cnt [ntasks][FUSED_MAXL].  status[ntasks].  All host pointers.
int hx_engine::fused_run(int mode, uint32_t ntasks, const uint32_t *q_sel, const int32_t *t_level, uint32_t ef, uint32_t k,
                         uint32_t entry, int entry_level, uint32_t *out_ids, float *out_d, uint32_t *out_cnt, uint32_t *status,
                         uint64_t counts[2], const HxFusedIter *it, HxFusedView *view, uint32_t roomy, const HxFusedDev *dev)
{
    if (ntasks == 0) return HX_OK;
    HxFusedIo &io = mirror.io;
    if (!io.stream) { io.stream = stream; io.ev0 = ev0; io.ev1 = ev1; }     // the engine's own stream and events
    int rc = fused_launch(io, mode, ntasks, q_sel, t_level, ef, k, entry, entry_level, it, roomy, dev);
    if (rc) return rc;
    return fused_collect(io, out_ids, out_d, out_cnt, status, counts, view);
}

// a pipelined scan slot: a stream, events and buffers of its own (created on first use)
int hx_engine::scan_io_init(uint32_t slot)
{
    if (slot >= HX_SCAN_SLOTS) return fail(HX_E_ARG, "scan slot out of range");
    HxFusedIo &io = scan_io[slot];
    if (io.stream) return HX_OK;
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipStreamCreateWithFlags(&io.stream, hipStreamNonBlocking)); io.own_stream = true;
    HX_HIP(this, hipEventCreate(&io.ev0)); HX_HIP(this, hipEventCreate(&io.ev1));
    HX_HIP(this, hipEventCreateWithFlags(&io.ev_dep, hipEventDisableTiming));
    return HX_OK;
}

// Everything of a launch up to its asynchronous result copies, on io.stream; nothing here waits for the device.
int hx_engine::fused_launch(HxFusedIo &io, int mode, uint32_t ntasks, const uint32_t *q_sel, const int32_t *t_level, uint32_t ef, uint32_t k,
                            uint32_t entry, int entry_level, const HxFusedIter *it, uint32_t roomy, const HxFusedDev *dev)
{
    HxMirror &mr = mirror;
    hipStream_t stream = io.stream;                                  // shadows the engine's: every call below goes to this launch's stream
    if (io.busy) return fail(HX_E_STATE, "a launch is still in flight on this slot");
    if (ntasks == 0) return fail(HX_E_ARG, "no tasks");
    if (pitch > FUSED_MAXCH * 1024u) return fail(HX_E_ARG, "row too wide for the fused kernel");
    const bool ins = mode == 1 || mode == 3;                       // find_element_neighbors (3: search only, W lists out)
    if (dtype == HX_SPARSE && mode == 1) return fail(HX_E_ARG, "sparsevec: the traversal kernel serves scans and search-only inserts (modes 0, 2, 3)");
    if (mode == 1 && 2 * mr.m > 64) return fail(HX_E_ARG, "m > 32: the insert kernel's select phase is built for lists of <= 64 (search-only MODE 3 serves every m)");
    if (mode == 3 && (!dev || !dev->d_wl_out || !dev->d_wl_cnt || !dev->h_prob)) return fail(HX_E_ARG, "mode 3 needs the W-list buffers");
    if (mode == 2 && (!it || !it->emask || !it->out_tix)) return fail(HX_E_ARG, "iterative scan arguments missing");
    if (dev && !ins) return fail(HX_E_ARG, "device-resident results are an insert-mode feature");
    HX_HIP(this, hipSetDevice(device));
    // LDS (carved in hx_fused_kernel.h f_worker): IDS[64] CTL[32] dsc[64] (4 B each) | RES[64] RL[64] (8 B each) | query (nch KiB) | W[ef+2] EP[ef+2] C[clds] (8 B each)
    // candidate heap: up to FUSED_CCAP entries, the first `clds` in LDS and the tail in a per-workgroup spill area
    // (the largest heap seen on 1M x 768 builds was 1552 entries at ef = 200); beyond FUSED_CCAP a task reports
    // FS_OVERFLOW and is re-run by the lock-step path
    const size_t nch_ = (pitch + 1023) / 1024;
    if (roomy < 1) roomy = 1;
    const uint32_t ccap = FUSED_CCAP * roomy;
    uint32_t clds = ins ? 600u : 512u;     // insert: 600 entries measured +8 % over 1024 (13 instead of 10 searches per CU at ef_construction 200)
    uint32_t disc_lds = mode == 2 ? 512u : 0u;
    uint32_t iter_per_cu = 14u;
    if (mode == 2) { const char *a = getenv("HX_DISC_LDS"), *b = getenv("HX_ITER_PER_CU"); if (a && atoi(a) > 0) disc_lds = (uint32_t)atoi(a); if (b && atoi(b) > 0) iter_per_cu = (uint32_t)atoi(b); }   // tuning knobs
    { const char *cv = getenv(ins ? "HX_CLDS_INSERT" : "HX_CLDS_QUERY"); if (cv && atoi(cv) > 0) clds = (uint32_t)atoi(cv); }   // tuning knob
    // HX_SORTED_ARRAY=1 (opt-in experiment, hx_fused_kernel.h: f_search_layer_sa): first launches of queries search on one sorted array; a query that
    // meets a tie reports FS_OVERFLOW and its retry launch (roomy > 1) uses the heap kernel, which is exact for any input
    const int sa_env = sa_env_on();
    const bool sa = sa_env && roomy == 1 && mode == 0 && dtype != HX_BIT && ef > 1 && ef <= 256;
    if (sa) clds = 0;                                            // no candidate heap
    if (mode == 1) clds = std::max<uint32_t>(clds, (uint32_t)((nch_ * 1024 + ((size_t)ef + 2) * 8 + 7) / 8));   // (mode 3 has no select phase)
      // select scratch aliases C's LDS part
    if (dev && dev->d_wtab) clds = std::max<uint32_t>(clds, dev->wt_size);                     // so does the W table
    const uint32_t wcap = ef + 2u + ((mode == 3 && dev && dev->d_skip) ? 254u : 0u);   // repair searches: room for 254 uncounted skip-set members in W, beyond that FS_OVERFLOW
    auto lds_bytes = [&](uint32_t cc) { return ((size_t)cc + 2 * (size_t)wcap + 64 + 64) * 8 + (64 + 32 + 64) * 4 + nch_ * 1024 + (size_t)(disc_lds ? disc_lds + 64 + 160 + 32 : 0) * 8; };   // f_worker's carve
    const size_t lds = lds_bytes(clds);
    // residency: one wave per workgroup, LDS-limited
    const size_t waves_cap = 4u * (size_t)(mode == 2 ? FUSED_MINW_ITER : ins ? FUSED_MINW_INS : FUSED_MINW);   // register-file limit: launch_bounds waves per SIMD x 4 SIMDs
    uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::max<size_t>(16, waves_cap), (160 * 1024) / (lds + 512)));
    { const char *pv = getenv(mode == 0 ? "HX_QUERY_PER_CU" : "HX_INSERT_PER_CU"); if (mode != 2 && pv && atoi(pv) > 0) per_cu = std::min<uint32_t>(per_cu, (uint32_t)atoi(pv)); }   // tuning knob
    per_cu = std::min<uint32_t>(per_cu, FUSED_SLOTS_PER_CU);
    uint32_t grid = std::min<uint32_t>(ntasks, 256u * per_cu);
    // >= 2x the ids a search touches at its usual ~ef expansions; twice that on indexes of >= 4M rows, where some searches reach further.
    // Measured on 1M x 768: a table twice as large costs 3-6 % of the scan rate (cache footprint), one half as large overflows and retries.
    const uint64_t vis_need = ((uint64_t)ef * 2 * mr.m * 2 + 1024) * (n_rows >= 4000000ull ? 2 : 1);
    uint64_t vis_words = 4096; while (vis_words < vis_need) vis_words <<= 1;
    { const char *vv = getenv("HX_VIS_SHIFT"); if (vv && mode != 2) { const int sh = atoi(vv); if (sh < 0) vis_words >>= -sh; else vis_words <<= sh; if (vis_words < 4096) vis_words = 4096; } }   // tuning knob
    if (mode != 2) vis_words *= roomy;
    uint64_t disc_stride = 0;
    if (mode == 2) {
        // an iterative scan keeps its visited set and `discarded` heap across resumes: sized for max_scan_tuples (a query that
        // outgrows them reports FS_OVERFLOW, is retried with `roomy` x the tables and only then re-run by the lock-step path); fewer resident
        // workgroups bound the footprint
        const uint64_t mt = (uint64_t)std::min<long long>(std::max<long long>(it->max_tuples, 1), 1 << 20);
        disc_stride = std::max<uint64_t>(4 * mt, 16384) * roomy;
        while (vis_words < 8 * mt + 4096) vis_words <<= 1;
        vis_words *= roomy;
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
    if (!io.d_spill) HX_HIP(this, hipMalloc((void **)&io.d_spill, (size_t)256 * FUSED_SLOTS_PER_CU * FUSED_CCAP * 8));
    uint32_t *vis_ptr = nullptr; void *spill_ptr = io.d_spill;
    if (roomy > 1) {   // a retry launch of a few overflowed tasks: private, larger tables sized for exactly this grid
        grid = std::min<uint32_t>(grid, 1024u);
        const size_t need_sp = (size_t)grid * ccap * 8; const uint64_t need_vis = (uint64_t)grid * vis_words;
        if (need_sp > mr.cap_spill_big) { if (mr.d_spill_big) (void)hipFree(mr.d_spill_big); mr.d_spill_big = nullptr; mr.cap_spill_big = 0; HX_HIP(this, hipMalloc(&mr.d_spill_big, need_sp)); mr.cap_spill_big = need_sp; }
        if (need_vis > mr.cap_vis_big) { if (mr.d_vis_big) (void)hipFree(mr.d_vis_big); mr.d_vis_big = nullptr; mr.cap_vis_big = 0; HX_HIP(this, hipMalloc((void **)&mr.d_vis_big, need_vis * 4)); mr.cap_vis_big = need_vis; }
        vis_ptr = mr.d_vis_big; spill_ptr = mr.d_spill_big;
    }
    if (roomy == 1 && (uint64_t)grid * vis_words > io.cap_vis) {
        if (io.d_vis) (void)hipFree(io.d_vis);
        io.d_vis = nullptr; io.cap_vis = 0;
        const uint64_t n = (uint64_t)(mode == 2 ? grid : 256u * FUSED_SLOTS_PER_CU) * vis_words;
        HX_HIP(this, hipMalloc((void **)&io.d_vis, n * 4));
        io.cap_vis = n;
    }
    // dev: the neighbour lists go straight into the caller's device records (a batch's exchange buffer); only the statuses come back
    const size_t out_n = dev ? 0 : !ins ? (size_t)ntasks * k : (size_t)ntasks * FUSED_MAXL * 2 * mr.m;      // mode 2: k = limit
    const size_t cnt_n = dev ? 0 : !ins ? (size_t)ntasks : (size_t)ntasks * FUSED_MAXL;
    // device task/in/out buffers (one allocation, reused)
    const size_t need = al16((size_t)ntasks * 4) * 6 + al16(out_n * 4) * 3 + al16(cnt_n * 4) + 256;
    if (need > io.cap_io) {
        if (io.d_io) (void)hipFree(io.d_io);
        if (io.h_io) (void)hipHostFree(io.h_io);
        io.d_io = io.h_io = nullptr; io.cap_io = 0;
        const size_t n = need * 2;
        HX_HIP(this, hipMalloc((void **)&io.d_io, n));
        HX_HIP(this, hipHostMalloc((void **)&io.h_io, n, hipHostMallocDefault));
        io.cap_io = n;
    }
    size_t o = 0;
    const size_t o_ctr = o; o += 256;
    const size_t o_q = o; o += al16((size_t)ntasks * 4);
    const size_t o_lv = o; o += al16((size_t)ntasks * 4);
    const size_t o_slot = o; o += al16((size_t)ntasks * 4);
    const size_t o_prob = o; o += al16((size_t)ntasks * 4);
    const size_t o_ent = o; o += al16((size_t)ntasks * 4);
    const size_t in_bytes = o;
    const size_t o_st = o; o += al16((size_t)ntasks * 4);
    const size_t o_cnt = o; o += al16(cnt_n * 4);
    const size_t o_ids = o; o += al16(out_n * 4);
    const size_t o_d = o; o += al16(out_n * 4);
    const size_t o_tix = o; o += al16(out_n * 4);
    memset(io.h_io + o_ctr, 0, 256);
    memcpy(io.h_io + o_q, q_sel, (size_t)ntasks * 4);
    if (t_level) memcpy(io.h_io + o_lv, t_level, (size_t)ntasks * 4); else memset(io.h_io + o_lv, 0, (size_t)ntasks * 4);
    if (dev && dev->h_slots) memcpy(io.h_io + o_slot, dev->h_slots, (size_t)ntasks * 4);
    if (mode == 3) memcpy(io.h_io + o_prob, dev->h_prob, (size_t)ntasks * 4);
    if (mode == 3 && dev->h_entry) memcpy(io.h_io + o_ent, dev->h_entry, (size_t)ntasks * 4);
    HX_HIP(this, hipMemcpyAsync(io.d_io, io.h_io, in_bytes, hipMemcpyHostToDevice, stream));
    FusedParams p;
    p.rows = d_rows; p.queries = d_queries; p.pitch = (uint32_t)pitch; p.nch = (uint32_t)((pitch + 1023) / 1024); p.n_rows = n_rows;
    p.l0_ids = mr.d_l0_ids; p.l0_cnt = mr.d_l0_cnt; p.level = mr.d_level; p.up_block = mr.d_up_block; p.up_ids = mr.d_up_ids; p.up_cnt = mr.d_up_cnt;
    p.l0_d = mr.d_l0_d; p.up_d = mr.d_up_d;
    p.m = mr.m; p.entry = entry; p.entry_level = entry_level;
    p.ntasks = ntasks; p.t_qsel = (const uint32_t *)(io.d_io + o_q); p.t_level = (const int32_t *)(io.d_io + o_lv);
    p.ef = ef; p.k = k; p.ccap = ccap; p.clds = clds; p.wcap = wcap;
    p.iter_mode = 0; p.limit = k; p.max_tuples = 0; p.emask = nullptr; p.disc = nullptr; p.disc_stride = 0; p.disc_lds = disc_lds; p.out_tix = (uint32_t *)(io.d_io + o_tix);
    if (mode == 2) { p.iter_mode = (uint32_t)it->iter_mode; p.max_tuples = it->max_tuples; p.emask = mr.d_emask; p.disc = (unsigned long long *)mr.d_disc; p.disc_stride = (uint32_t)disc_stride; }
    p.spill = (uint2 *)spill_ptr; p.spill_stride = ccap;
    { const char *dv = getenv("HX_F_DBG"); p.fdbg = dv ? (uint32_t)atoi(dv) : 0u; }
    p.sa = sa ? 1u : 0u;
    p.vis = vis_ptr ? vis_ptr : io.d_vis; p.vis_words = vis_words;
    p.next_task = (uint32_t *)(io.d_io + o_ctr);
    p.n_dist = (unsigned long long *)(io.d_io + o_ctr + 8);
    p.out_ids = (uint32_t *)(io.d_io + o_ids); p.out_d = (float *)(io.d_io + o_d); p.out_cnt = (uint32_t *)(io.d_io + o_cnt);
    p.status = (uint32_t *)(io.d_io + o_st);
    p.o_cst = FUSED_MAXL; p.o_lst = FUSED_MAXL * 2 * mr.m; p.t_oslot = nullptr;
    p.wtab = nullptr; p.wt_size = 0; p.wt_slot0 = 0; p.wt_valid = nullptr;
    p.sparse_cap = dtype == HX_SPARSE ? (uint32_t)std::min(dim, HX_SPARSE_MAX_NNZ) : 0u;
    p.t_entry = (mode == 3 && dev && dev->h_entry) ? (const uint32_t *)(io.d_io + o_ent) : nullptr; p.skip = (mode == 3 && dev) ? dev->d_skip : nullptr;
    p.wl_out = nullptr; p.wl_cnt = nullptr; p.t_prob = nullptr; p.ondisk = (mode == 3 && dev && dev->ondisk) ? 1u : 0u;
    p.wcap_out = ef;
    if (mode == 3) { p.wl_out = (uint2 *)dev->d_wl_out; p.wl_cnt = dev->d_wl_cnt; p.t_prob = (const uint32_t *)(io.d_io + o_prob); if (dev->d_skip) p.wcap_out = wcap; }
    if (dev && dev->d_wtab) { p.wtab = (uint2 *)dev->d_wtab; p.wt_size = dev->wt_size; p.wt_slot0 = dev->wt_slot0; p.wt_valid = dev->d_wt_valid; }
    if (dev) {   // record = cnt[FUSED_MAXL] | ids[FUSED_MAXL][2m] | d[FUSED_MAXL][2m]  (hx_batch.hip reads the same layout)
        p.out_cnt = dev->d_rec; p.out_ids = dev->d_rec + FUSED_MAXL; p.out_d = (float *)(dev->d_rec + FUSED_MAXL + FUSED_MAXL * 2 * mr.m);
        p.o_cst = p.o_lst = dev->rec_words;
        if (dev->h_slots) p.t_oslot = (const uint32_t *)(io.d_io + o_slot);
    }
    if (timing && io.own_stream && !scan_epoch_set) {   // time zero of the pipelined scans' busy-time union
        if (!ev_scan_epoch) HX_HIP(this, hipEventCreate(&ev_scan_epoch));
        HX_HIP(this, hipEventRecord(ev_scan_epoch, this->stream)); HX_HIP(this, hipEventSynchronize(ev_scan_epoch));
        scan_epoch_set = true; scan_last_end = 0.0;
    }
    if (timing) HX_HIP(this, hipEventRecord(io.ev0, stream));
    hipError_t ls = hipSuccess;
    fused_stream = stream;
    ls = dtype == HX_F32 ? hx_launch_fused_f32(this, metric, p, grid, lds, mode)
            : dtype == HX_F16 ? hx_launch_fused_f16(this, metric, p, grid, lds, mode)
            : dtype == HX_SPARSE ? hx_launch_fused_sparse(this, metric, p, grid, lds, mode) : hx_launch_fused_bit(this, metric, p, grid, lds, mode);
    HX_HIP(this, ls);
    if (timing) HX_HIP(this, hipEventRecord(io.ev1, stream));
    HX_HIP(this, hipMemcpyAsync(io.h_io + o_ctr, io.d_io + o_ctr, 256, hipMemcpyDeviceToHost, stream));
    HX_HIP(this, hipMemcpyAsync(io.h_io + o_st, io.d_io + o_st, (dev ? o_cnt : mode == 2 ? o : o_tix) - o_st, hipMemcpyDeviceToHost, stream));
    io.busy = true; io.mode = mode; io.ntasks = ntasks; io.roomy = roomy; io.timed = timing; io.it = it; io.has_dev = dev != nullptr;
    io.o_ctr = o_ctr; io.o_st = o_st; io.o_cnt = o_cnt; io.o_ids = o_ids; io.o_d = o_d; io.o_tix = o_tix; io.out_n = out_n; io.cnt_n = cnt_n;
    return HX_OK;
}

// waits for the launch in flight on `io` and hands its results over (view: read in place from the pinned staging buffer)
int hx_engine::fused_collect(HxFusedIo &io, uint32_t *out_ids, float *out_d, uint32_t *out_cnt, uint32_t *status, uint64_t counts[2], HxFusedView *view)
{
    if (!io.busy) return fail(HX_E_STATE, "no launch in flight on this slot");
    io.busy = false;
    const int mode = io.mode; const uint32_t ntasks = io.ntasks, roomy = io.roomy; const bool dev = io.has_dev; const HxFusedIter *it = io.it;
    const size_t o_ctr = io.o_ctr, o_st = io.o_st, o_cnt = io.o_cnt, o_ids = io.o_ids, o_d = io.o_d, o_tix = io.o_tix, out_n = io.out_n, cnt_n = io.cnt_n;
    HX_HIP(this, hipSetDevice(device));
    HX_HIP(this, hipStreamSynchronize(io.stream));
    if (roomy == 1 && mode != 2) {   // test hook: pretend every k-th task overflowed, so that the roomy retry path is exercised
        const char *fv = getenv("HX_FORCE_OVERFLOW_MOD"); const uint32_t k = fv ? (uint32_t)atoi(fv) : 0u;
        if (k) for (uint32_t t = 0; t < ntasks; t += k) ((uint32_t *)(io.h_io + o_st))[t] = FS_OVERFLOW;
    }
    if (dev) { if (status) memcpy(status, io.h_io + o_st, (size_t)ntasks * 4); }
    else if (view) {   // the caller reads the pinned staging buffer in place
        view->status = (const uint32_t *)(io.h_io + o_st); view->cnt = (const uint32_t *)(io.h_io + o_cnt);
        view->ids = (const uint32_t *)(io.h_io + o_ids); view->d = (const float *)(io.h_io + o_d);
    } else {
        memcpy(status, io.h_io + o_st, (size_t)ntasks * 4);
        memcpy(out_cnt, io.h_io + o_cnt, cnt_n * 4);
        memcpy(out_ids, io.h_io + o_ids, out_n * 4);
        memcpy(out_d, io.h_io + o_d, out_n * 4);
    }
    if (mode == 2) memcpy(it->out_tix, io.h_io + o_tix, out_n * 4);
    unsigned long long nd[17]; memcpy(nd, io.h_io + o_ctr + 8, 136);
    if (getenv("HX_F_DBG") && (atoi(getenv("HX_F_DBG")) & 4) && !FUSED_TIMERS_ON) { static bool once = false; if (!once) { once = true; fprintf(stderr, "[hx] HX_F_DBG=4: this library was built without -DFUSED_TIMERS (HX_CFLAGS=-DFUSED_TIMERS python pgvector-rx_amd/build.py --force)\n"); } }
    if (FUSED_TIMERS_ON && getenv("HX_F_DBG") && (atoi(getenv("HX_F_DBG")) & 4))
        fprintf(stderr, "[hx] k_fused mode %d tasks %u: shader-clock ticks (s_memtime) summed over waves: pop %llu list %llu visited %llu compact %llu dist %llu settle+filter %llu replay %llu; expansions %llu pushes %llu; inside dist: issue %llu wait %llu math %llu reduce %llu; select phase %llu\n",
                mode, ntasks, nd[3], nd[4], nd[5], nd[6], nd[7], nd[8], nd[9], nd[10], nd[11], nd[12], nd[13], nd[14], nd[15], nd[16]);
    if (counts) { counts[0] = nd[0]; counts[1] = nd[1]; }
    if (nd[2] > fused_cmax) fused_cmax = nd[2];
    if (io.timed) {
        float ms = 0.f; HX_HIP(this, hipEventElapsedTime(&ms, io.ev0, io.ev1));
        last_ms = ms; stat_fused.launches++; stat_fused.units += nd[0] + nd[1]; stat_fused.ms += ms;
        if (io.own_stream && scan_epoch_set) {
            float t_end = 0.f; HX_HIP(this, hipEventElapsedTime(&t_end, ev_scan_epoch, io.ev1));
            const double b = (double)t_end, a = std::max(b - (double)ms, scan_last_end);
            stat_scan.launches++; stat_scan.units += nd[0]; if (b > a) stat_scan.ms += b - a;
            if (b > scan_last_end) scan_last_end = b;
        }
    }
    return HX_OK;
}
