// hx_internal.h -- shared between the HIP engine (hx_engine.hip) and the host graph driver (hx_index.cpp).
// Not part of the ABI; the ABI is include/hnswrx.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/hnswrx.h"

#define HX_FUSED_MAXL 8   /* layers handled by the fused kernel; mirrors FUSED_MAXL */

// One request channel = ONE pinned host buffer + ONE device mirror for everything a lock-step round sends
// (group query selectors, offsets, row ids, pair-group descriptors, workgroup table) and ONE pair of buffers for
// what it gets back (K1 distances, K2 pair distances): a round costs one H2D copy, <= 2 kernels, one D2H copy,
// one stream sync.  The pointers below are sub-arrays of those packed buffers, valid after layout_round().
// The lock-step driver writes its requests straight into the h_* arrays (no extra copy).
struct HxRound { uint32_t n_dgroups = 0, n_dids = 0, n_pgroups = 0, n_pids = 0; uint64_t n_pout = 0; };

struct HxChannel {
    uint8_t *h_req = nullptr, *d_req = nullptr; size_t cap_req = 0, req_bytes = 0;
    uint8_t *h_res = nullptr, *d_res = nullptr; size_t cap_res = 0, res_bytes = 0;
    HxRound round; uint32_t max_wgs = 0;
    // distance groups (hx_distances_batch shape)
    uint32_t *h_grp_q = nullptr, *h_grp_off = nullptr, *h_ids = nullptr;
    uint32_t *d_grp_q = nullptr, *d_grp_off = nullptr, *d_ids = nullptr;
    float *h_out = nullptr, *d_out = nullptr;
    // pair groups (hx_pairwise_many shape); wg_tab: {group, first pair} per workgroup
    uint32_t *h_pg_off = nullptr, *h_pids = nullptr, *h_wg_tab = nullptr;
    uint8_t *h_pg_flag = nullptr; uint32_t *h_glist = nullptr, *d_glist = nullptr;   // per group: 1 = evaluate on the matrix cores (hx_mfma.hip); the list of those groups
    uint16_t *h_pg_na = nullptr, *h_pg_nb = nullptr;
    uint64_t *h_pg_out_off = nullptr;
    uint32_t *d_pg_off = nullptr, *d_pids = nullptr, *d_wg_tab = nullptr;
    uint16_t *d_pg_na = nullptr, *d_pg_nb = nullptr;
    uint64_t *d_pg_out_off = nullptr;
    float *h_pout = nullptr, *d_pout = nullptr;
};

struct HxKernelStat { uint64_t launches = 0, units = 0; double ms = 0.0; };

// arguments of an iterative scan on the device (k_fused MODE 2)
struct HxFusedIter {
    int iter_mode = 1; long long max_tuples = 0;       // 1 relaxed_order, 2 strict_order; hnsw.max_scan_tuples
    const uint16_t *emask = nullptr; uint64_t n_elems = 0;   // per element: bits 0-9 heap TIDs that pass the filter, bits 12-15 number of heap TIDs
    uint32_t *out_tix = nullptr;                       // [ntasks][limit]: which heap TID of out_ids' element
};
// zero-copy view of a fused launch's results in its pinned staging buffer (valid until the next launch on the same HxFusedIo)
struct HxFusedView { const uint32_t *ids = nullptr, *cnt = nullptr, *status = nullptr; const float *d = nullptr; };
struct HxFusedDev;
// Everything ONE k_fused launch in flight owns: its stream and events, the pinned + device staging of tasks and results, the per-workgroup visited
// tables and candidate-heap spill areas, and what fused_collect needs to finish it.  The engine's synchronous launches use mirror.io on the engine
// stream; pipelined scans (hx_index_search_submit / _wait) use scan_io[slot], each on a stream of its own, so that the first round of launch N+1
// fills the CUs the last round of launch N leaves idle.
struct HxFusedIo {
    hipStream_t stream = nullptr; bool own_stream = false; hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_dep = nullptr;
    uint8_t *h_io = nullptr, *d_io = nullptr; size_t cap_io = 0;
    uint32_t *d_vis = nullptr; uint64_t cap_vis = 0;
    void *d_spill = nullptr;                 // candidate-heap spill areas of the fused kernel's workgroups
    // the launch in flight (fused_launch -> fused_collect)
    bool busy = false; int mode = 0; uint32_t ntasks = 0, roomy = 1; bool timed = false;
    size_t o_ctr = 0, o_st = 0, o_cnt = 0, o_ids = 0, o_d = 0, o_tix = 0, out_n = 0, cnt_n = 0;
    const HxFusedIter *it = nullptr; bool has_dev = false;
};

// device copy of the graph (neighbour ids only) + scratch of the fused traversal kernel (hx_fused.inc.h)
struct HxMirror {
    uint32_t m = 0; uint64_t cap = 0, cap_blocks = 0;
    uint32_t *d_l0_ids = nullptr; float *d_l0_d = nullptr; uint16_t *d_l0_cnt = nullptr; int32_t *d_level = nullptr;
    uint32_t *d_up_block = nullptr, *d_up_ids = nullptr; float *d_up_d = nullptr; uint16_t *d_up_cnt = nullptr;
    uint8_t *h_lk = nullptr, *d_lk = nullptr; size_t cap_lk = 0;
    float *d_pm = nullptr; uint8_t *d_pm_valid = nullptr; uint64_t cap_pm = 0;   // resident pair matrices of the layer-0 lists
    HxFusedIo io;                            // task / result staging, visited tables and spill areas of the synchronous fused_run launches (engine stream)
    void *d_spill_big = nullptr; size_t cap_spill_big = 0; uint32_t *d_vis_big = nullptr; uint64_t cap_vis_big = 0;   // tables of a retry launch (fused_run roomy > 1)
    void *d_disc = nullptr; size_t cap_disc = 0;             // `discarded` heaps of iterative scans (k_fused MODE 2)
    uint16_t *d_emask = nullptr; uint64_t cap_emask = 0;     // per-element heap-TID filter masks of the current iterative scan
    uint8_t *h_stage = nullptr, *d_stage = nullptr; size_t cap_stage = 0;
};

// insert-mode results written straight into device records (a batch's exchange buffer, hx_batch.hip): record r =
// cnt[HX_FUSED_MAXL] | ids[HX_FUSED_MAXL][2m] | d[HX_FUSED_MAXL][2m], rec_words 32-bit words apart; task t fills record h_slots[t] (nullptr: t)
struct HxFusedDev { uint32_t *d_rec = nullptr; uint32_t rec_words = 0; const uint32_t *h_slots = nullptr;
                    void *d_wtab = nullptr; uint32_t wt_size = 0, wt_slot0 = 0; uint8_t *d_wt_valid = nullptr;
                    // mode 3 (search only; hx_mfma.hip selects): layer lc of task t is problem h_prob[t] + lc; its W goes to d_wl_out[problem * ef ..], |W| to d_wl_cnt[problem]
                    void *d_wl_out = nullptr; uint32_t *d_wl_cnt = nullptr; const uint32_t *h_prob = nullptr; bool ondisk = false;
                    const uint32_t *h_entry = nullptr; const uint8_t *d_skip = nullptr; };   // mode 3 repair searches: per-task entry points (host array), skip flags per element (device)   // ondisk: search_layer_disk semantics (aminsert)   // W tables of the members (hx_fused_core.h FusedParams::wtab)

// device-side grouping of a batch's back-link ops (hx_group.hip): workspace + the grouped arrays it leaves on the device
struct HxGroupWork {
    uint8_t *d = nullptr; size_t cap = 0; uint32_t *h_ctr = nullptr; uint8_t *h = nullptr; size_t cap_h = 0;   // device workspace, pinned counters, pinned op staging
    const uint32_t *tg = nullptr, *ly = nullptr, *off = nullptr, *op_new = nullptr, *gmap_hub = nullptr, *gmap_norm = nullptr, *gmap_fill = nullptr; const float *op_d = nullptr;
    unsigned long long *d_keys = nullptr; uint32_t *d_new = nullptr; float *d_d = nullptr;   // where the ungrouped ops go (after hx_group_reserve)
};
// workspace of the device-resident batch pipeline (hx_batch.hip)
struct HxBatchWork {
    uint8_t *d = nullptr; size_t cap = 0;               // device scratch (hashes, candidate pairs, op counts)
    uint8_t *h = nullptr; size_t cap_h = 0; uint32_t *h_ctr = nullptr;   // pinned staging and counters
    uint32_t *d_rec = nullptr; size_t cap_rec = 0;      // member records of single-process batches (multi-GPU builds pass their exchange buffer instead)
    void *d_wtab = nullptr; size_t cap_wtab = 0; uint8_t *d_wt_valid = nullptr; size_t cap_wtv = 0;   // the members' W tables (d(new, x) of their layer-0 searches) for the back-link kernels
    uint32_t wt_size = 0, wt_base = 0, wt_n = 0;        // entries per table (0: off); element ids [wt_base, wt_base + wt_n) are the open batch's members
};
uint32_t hx_rec_words(uint32_t m);                      // 32-bit words per member record: cnt[HX_FUSED_MAXL] | ids[HX_FUSED_MAXL][2m] | d[HX_FUSED_MAXL][2m]
// W lists and problem table of a k_fused MODE 3 launch (hx_mfma.hip)
struct HxWselWork { void *d_wl = nullptr; uint32_t *d_cnt = nullptr, *d_slot = nullptr, *d_task = nullptr; uint8_t *d_layer = nullptr; unsigned long long *d_counters = nullptr;
                    size_t cap_prob = 0; uint32_t cap_ef = 0; };
struct hx_engine;
static inline uint32_t hx_xrec_words(uint32_t m) { return (3u + 4u * m + 3u) & ~3u; }   // list record of the multi-GPU exchange: target, layer, cnt, ids[2m], d[2m]
int hx_group_stage(hx_engine *e, uint32_t n_ops, HxGroupWork &w, unsigned long long **keys, uint32_t **op_new, float **op_d);
int hx_group_reserve(hx_engine *e, uint32_t n_ops, HxGroupWork &w);
int hx_group_run(hx_engine *e, uint32_t n_ops, uint32_t hub_min, HxGroupWork &w, uint32_t counters_out[5], bool want_fill = false);   // counters: groups, hub lists, other lists, longest chain, lists whose pair matrix k_pm_fill must compute
int hx_group_ops(hx_engine *e, uint32_t n_ops, const unsigned long long *h_keys, const uint32_t *h_new, const float *h_d, uint32_t hub_min,
                 HxGroupWork &w, uint32_t counters_out[5], bool want_fill = false);

struct hx_engine {
    int device = 0, dtype = 0, metric = 0, dim = 0;
    uint64_t row_bytes = 0, pitch = 0, capacity = 0, n_rows = 0;
    uint8_t *d_rows = nullptr;
    uint8_t *d_queries = nullptr; uint32_t cap_queries = 0, n_queries = 0;
    hipStream_t stream = nullptr;
    hipStream_t fused_stream = nullptr;   // the stream the k_fused launchers use (set by fused_launch: the engine's, or a pipelined scan slot's)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    bool timing = false; float last_ms = 0.f;
    HxKernelStat stat_dist, stat_pair, stat_fused, stat_links, stat_mfma, stat_wsel;
    // pipelined scans: launches overlap, so their busy time is the UNION of the launches' [start, end] intervals (HIP events of each slot's stream,
    // measured from ev_scan_epoch on the engine's stream); stat_scan.ms = that union
    HxKernelStat stat_scan; hipEvent_t ev_scan_epoch = nullptr; bool scan_epoch_set = false; double scan_last_end = 0.0;
    hipEvent_t ev4 = nullptr, ev5 = nullptr;
    // |row|^2 of halfvec rows for the MFMA band (hx_mfma.hip)
    float *d_mf_norm2 = nullptr; uint64_t mf_cap = 0, mf_norm_rows = 0; std::vector<float> h_mf_norm2;
    int mfma_norms(uint64_t upto);
    HxWselWork wsel; int wsel_reserve(uint32_t n_prob, uint32_t ef); int mfma_select_done(uint64_t gemm_pairs);
    // select_neighbors on the matrix cores behind k_fused MODE 3 (hx_mfma.hip): Gram matrices of the problems' candidates, then the heuristic on them
    float *d_wg = nullptr; size_t cap_wg = 0; bool wg_pending = false;
    int mfma_select(uint32_t n_prob, uint32_t ef, const void *d_wl, const uint32_t *d_wl_cnt, const uint8_t *d_prob_layer, const uint32_t *d_prob_slot,
                    const uint32_t *d_prob_task, const uint32_t *d_status, uint32_t *d_rec, uint32_t rec_words, unsigned long long *d_counters);
    HxMirror mirror;
    uint64_t fused_cmax = 0;      // largest candidate-heap length any fused task reached (sizing the LDS budget)
    HxChannel ch;
    std::string err;

    // internal (driver-facing) entry points: size the packed buffers for a round and point ch.h_*/d_* into them;
    // then (after the caller filled the h_* request arrays) copy, launch K1 and/or K2, copy back, sync.
    int layout_round(const HxRound &r);
    int run_round();
    // device-resident traversal (hx_fused.inc.h)
    int mirror_reserve(uint32_t m, uint64_t n_elems, uint64_t n_blocks);
    int mirror_update(uint32_t first, uint32_t n_new, const int32_t *levels, const uint32_t *blocks,
                      uint32_t n_rec, const uint32_t *hdr, const uint32_t *ids, const float *dists);
    // update_neighbor_connections on the device for n_groups (target, layer) lists: group g applies ops
    // [op_off[g], op_off[g+1]) = (new element, its distance to the target) in order; the lists are read from and written
    // back to the mirror, and returned: out_cnt[g], out_ids/out_d [g][2m]
    // copies the mirror's lists of elements [0, n_elems) / upper-layer blocks [0, n_blocks) to host arrays (ids and distances SoA, counts)
    int mirror_download(uint64_t n_elems, uint64_t n_blocks, uint32_t *l0_ids, float *l0_d, uint16_t *l0_cnt, uint32_t *up_ids, float *up_d, uint16_t *up_cnt);
    // update_neighbor_connections for a batch whose ops (key = target << 7 | layer, new element, distance; op order) are grouped on the device;
    // the updated lists stay in the mirror.  stats: [0] groups, [1] longest chain
    int links_stage_ops(uint32_t n_ops, unsigned long long **keys, uint32_t **op_new, float **op_d) { return hx_group_stage(this, n_ops, grp, keys, op_new, op_d); }
    // on_device: the ops already sit in grp.d_keys / d_new / d_d (hx_group_reserve + the batch pipeline's emission kernel); want_xrec: every updated
    // list is also written as a record {target, layer, cnt, ids[2m], d[2m]} into d_xl (xl_records of hx_xrec_words(m) words) for the multi-GPU exchange
    int links_run_grouped(uint32_t n_ops, const unsigned long long *keys, const uint32_t *op_new, const float *op_d, uint64_t *n_pairs, uint32_t stats[2],
                          bool on_device = false, bool want_xrec = false);
    uint32_t *d_xl = nullptr; size_t cap_xl = 0; uint32_t xl_records = 0;
    // device-resident batch pipeline (hx_batch.hip)
    HxBatchWork bw;
    int db_reserve_records(uint64_t n_records);
    int db_begin_wtabs(uint32_t base, uint32_t b, uint32_t ef_construction);   // sizes and clears the W tables of a new batch
    int db_fill_record(uint32_t *d_rec, uint32_t slot, const uint32_t *h_src);
    int db_dup_candidates(uint32_t base, uint32_t b, const uint32_t *d_rec, std::vector<uint32_t> &za, std::vector<uint32_t> &zb,
                          std::vector<uint32_t> &ha, std::vector<uint32_t> &hb);
    int db_apply(uint32_t base, uint32_t b, const uint32_t *d_rec, const uint8_t *h_dup, uint32_t rank, uint32_t world, uint32_t *n_ops_out);
    int db_import_lists(const uint32_t *d_xrec, uint32_t n_records);
    HxGroupWork grp;
    // lists of any legal size (hx_biglist.hip): select_neighbors over the result sets in wsel; a batch's back-links per list, stateless
    int biglist_select(uint32_t n_prob, uint32_t stride, const uint32_t *lm, uint32_t lm0, const uint32_t **out_ids, const float **out_d, const uint32_t **out_cnt, uint64_t *n_pairs);
    int biglist_ops_stage(uint32_t n_groups, uint32_t n_ops, uint32_t lm0, uint32_t **ids, float **d, uint32_t **cnt, uint32_t **lm, uint32_t **op_off, uint32_t **op_new, float **op_d);
    int biglist_ops_run(uint64_t *n_pairs, bool disk = false);   // disk: aminsert's get_update_index / write_neighbor_update instead of update_neighbor_connections
    size_t bl_o_lm = 0, bl_o_off = 0, bl_o_new = 0, bl_o_od = 0, bl_o_cnt = 0, bl_o_ids = 0, bl_o_d = 0, bl_end = 0; uint32_t bl_groups = 0, bl_lm0 = 0;
    // aminsert: all back-connections of a batch, one wavefront per list (hx_links.hip: k_update_runs): stage -> fill -> run -> the new lists in the same arrays
    int update_runs_stage(uint32_t n_runs, uint32_t n_ops, uint32_t stride, uint32_t **ids, float **d, uint32_t **cnt, uint32_t **lm, uint32_t **op_off, uint32_t **op_new, float **op_d);
    int update_runs_run(uint64_t *n_pairs);
    size_t ur_o_lm = 0, ur_o_off = 0, ur_o_new = 0, ur_o_od = 0, ur_o_cnt = 0, ur_o_ids = 0, ur_o_d = 0, ur_end = 0; uint32_t ur_n = 0, ur_stride = 0;
    // aminsert's get_update_index for one wave of full lists (hx_links.hip: k_update_index): stage -> fill the pinned arrays -> run
    int update_index_stage(uint32_t n_ops, uint32_t stride, uint32_t **ids, float **d, float **new_d, uint32_t **cnt);
    int update_index_run(const int32_t **slot_out, uint64_t *n_pairs);
    size_t ui_o_nd = 0, ui_o_cnt = 0, ui_o_ids = 0, ui_o_d = 0, ui_o_slot = 0, ui_in = 0; uint32_t ui_n = 0, ui_stride = 0;
    int links_run(uint32_t n_groups, const uint32_t *target, const uint32_t *layer, const uint32_t *op_off,
                  const uint32_t *op_new, const float *op_d, const uint32_t **out_ids, const float **out_d, const uint32_t **out_cnt, uint64_t *n_pairs,
                  bool want_lists = true);   // false: the updated lists stay in the mirror only (the host pulls them when it needs them)
    // fused_run = fused_launch (everything up to the asynchronous result copies, on io.stream) + fused_collect (stream sync, results, counters)
    HxFusedIo scan_io[HX_SCAN_SLOTS];
    int scan_io_init(uint32_t slot);
    int fused_launch(HxFusedIo &io, int mode, uint32_t ntasks, const uint32_t *q_sel, const int32_t *t_level, uint32_t ef, uint32_t k,
                     uint32_t entry, int entry_level, const HxFusedIter *it, uint32_t roomy, const HxFusedDev *dev);
    int fused_collect(HxFusedIo &io, uint32_t *out_ids, float *out_d, uint32_t *out_cnt, uint32_t *status, uint64_t counts[2], HxFusedView *view);
    int fused_run(int mode, uint32_t ntasks, const uint32_t *q_sel, const int32_t *t_level, uint32_t ef, uint32_t k,
                  uint32_t entry, int entry_level, uint32_t *out_ids, float *out_d, uint32_t *out_cnt, uint32_t *status,
                  uint64_t counts[2], const HxFusedIter *it = nullptr, HxFusedView *view = nullptr, uint32_t roomy = 1, const HxFusedDev *dev = nullptr);   // roomy > 1: retry of overflowed tasks with that many times the visited table and candidate heap
    int fail(int code, const std::string &msg) { err = msg; return code; }
};

// hx_sparse.hip: the sparsevec kernels (same request arrays as K1 / K2)
#define HX_SPARSE_MAX_NNZ 1000
hipError_t hx_launch_sparse_dist(hx_engine *e, uint32_t n_groups);
hipError_t hx_launch_sparse_pairs(hx_engine *e, uint32_t n_wgs);
hipError_t hx_launch_sparse_normalize(hx_engine *e, uint8_t *base, uint64_t n, double *d_norms);
hipError_t hx_launch_pair_mfma(hx_engine *e, uint32_t n_groups, const uint32_t *d_glist);

#define HX_HIP(e, call)                                                                             \
    do {                                                                                            \
        hipError_t _s = (call);                                                                     \
        if (_s != hipSuccess) return (e)->fail(HX_E_HIP, std::string(#call) + ": " + hipGetErrorString(_s)); \
    } while (0)
