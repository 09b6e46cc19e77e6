/*
 * hnswrx.h -- C ABI of the MI355X-native HNSW distance engine for pgvector-rx.
 *
 * This is the drop-in boundary for ONE path of the reference (maropu/pgvector-rx): the distance
 * evaluations inside HNSW search-layer candidate expansion and the ef_construction neighbour-selection
 * loop of index build.  A Rust/pgrx host binds these symbols with an `extern "C"` block
 * (INTEGRATION.md shows it) in place of the per-pair fmgr trampoline; everything PostgreSQL-side
 * (access-method handler, operator classes, pages, WAL) stays in the host.
 *
 * Conventions
 *   - plain C, POD arguments only; no exceptions/longjmp/abort cross the boundary.
 *   - every entry returns int: 0 = ok, <0 = HX_E_*; hx_last_error() gives the text
 *     (the host turns non-zero into pgrx::error!, mirroring ereport(ERROR) -- build.rs:399, scan.rs:708).
 *   - a handle owns one HIP stream and is NOT thread-safe (the reference is one single-threaded backend
 *     per connection, build.rs:385); host pointers are never retained after a call returns.
 *   - distances come back as f32: the build path consumes `f64 as f32` (build.rs:366-367) and the scan
 *     path's f64 (scan.rs:190-191) is a widening of the same f32 accumulator, so f32 is lossless for
 *     L2/IP/L1/Hamming.  Jaccard is computed in f64 on the device and rounded once to f32.
 *   - rows are dense payloads WITHOUT the varlena header: dim f32 (vector.rs:43-48 `x[dim]`),
 *     dim u16 IEEE binary16 (halfvec.rs:41-46), or ceil(dim/8) bytes MSB-first (bitvec.rs:28-37).
 *   - row ids are dense uint32 in append order (the host keeps the TID <-> row-id map).
 */
#ifndef HNSWRX_H
#define HNSWRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HX_ABI_VERSION 1

/* element type of the indexed column */
enum hx_dtype { HX_F32 = 0 /* vector */, HX_F16 = 1 /* halfvec */, HX_BIT = 2 /* bit(n) */, HX_SPARSE = 3 /* sparsevec */ };
/* HX_SPARSE (src/types/sparsevec.rs): `dim` is the sparsevec's dimension count (<= 1e9) and a row is the fixed-size record
 *   { int32 nnz; int32 pad[3]; int32 index[cap]; float value[cap]; } padded to a multiple of 16 bytes, cap = min(dim, 1000)
 * (an indexed sparsevec holds at most 1000 non-zero elements), indices ascending and 0-based as in SparseVecHeader, unused slots zero (rows are
 * compared bytewise for the duplicate test).  hx_row_bytes gives the record size.  Metrics: HX_L2SQ, HX_NEG_IP, HX_L1 -- the merge joins of
 * sparse_l2_squared_distance / sparse_inner_product / sparsevec_l1_distance (sparsevec.rs:873-950, 1038-1088) with the reference's own f32
 * accumulation order, one lane per pair; hx_normalize_rows = sparsevec_l2_normalize_raw (:1123-1178, zeros dropped).  The index runs on the
 * lock-step driver for this type (the traversal and back-link kernels stream dense rows); no page image. */

/* which opclass support FUNCTION 1 the index names (hnsw_constants.rs:12) */
enum hx_metric {
    HX_L2SQ = 0,    /* vector_l2_squared_distance  vector.rs:598-607, halfvec.rs:780-791 */
    HX_NEG_IP = 1,  /* vector_negative_inner_product vector.rs:624-633; also the cosine opclasses (rows pre-normalised, vector.rs:852-856) */
    HX_L1 = 2,      /* l1_distance                 vector.rs:652-659 */
    HX_HAMMING = 3, /* hamming_distance            bitvec.rs:144-153 */
    HX_JACCARD = 4  /* jaccard_distance            bitvec.rs:156-167 */
};

enum hx_status {
    HX_OK = 0,
    HX_E_ARG = -1,      /* bad argument (null, out of range id, dtype/metric mismatch) */
    HX_E_DIM = -2,      /* "different vector dimensions" vector.rs:510-514 / max dims hnsw_constants.rs:4 */
    HX_E_NOMEM = -3,    /* host or device allocation failed / capacity exceeded */
    HX_E_HIP = -4,      /* a HIP runtime call failed */
    HX_E_NODEVICE = -5, /* no usable GPU: the engine never falls back to the CPU */
    HX_E_STATE = -6     /* call sequence error */
};

typedef struct hx_engine hx_engine;   /* device row store + kernels (replaces DistanceFn, graph/mod.rs:144-145) */
typedef struct hx_index hx_index;     /* host-side HNSW graph driven in lock-step over the engine (graph/mod.rs, build.rs, scan.rs) */

/* group query source: a row already in the store, or a slot of the uploaded query set */
#define HX_QUERY_SLOT 0x80000000u

int hx_abi_version(void);
const char *hx_last_error(const hx_engine *e);     /* e may be NULL: error of the last failed hx_create on this thread */

/* ------------------------------------------------------------------------------------------------
 * Engine: device-resident row store + batched distance kernels
 * ------------------------------------------------------------------------------------------------ */

/* Creates an engine on HIP device `device` for rows of `dim` elements; capacity_rows is reserved up
 * front in HBM (the reference's Vec<u8> arena, build.rs:249, grows instead).  Fails with
 * HX_E_NODEVICE when no GPU is visible. */
int hx_create(int device, int dtype, int metric, int dim, uint64_t capacity_rows, hx_engine **out);
int hx_destroy(hx_engine *e);

int hx_dim(const hx_engine *e);
uint64_t hx_row_bytes(const hx_engine *e);   /* payload bytes per row as the caller supplies them */
uint64_t hx_num_rows(const hx_engine *e);
void *hx_stream(const hx_engine *e);         /* the hipStream_t every kernel of this handle runs on */

/* Appends n rows (host memory, row-major, hx_row_bytes each); *first_row_id = id of the first.
 * Mirrors bs.values.extend_from_slice, build.rs:451-454. */
int hx_append_rows(hx_engine *e, const void *rows_host, uint64_t n, uint64_t *first_row_id);
/* Same, rows already in device memory (synthetic benches keep data in HBM).  The device is synchronised first (hipDeviceSynchronize): whatever stream
 * produced the buffer has finished before it is copied.  hx_set_queries_device does the same. */
int hx_append_rows_device(hx_engine *e, const void *rows_dev, uint64_t n, uint64_t *first_row_id);
/* Drops the last n rows: bs.values.truncate on a duplicate, build.rs:507-509. */
int hx_pop_rows(hx_engine *e, uint64_t n);
/* Copies rows back (tests / page serialiser). */
int hx_read_rows(hx_engine *e, uint64_t first, uint64_t n, void *rows_host);

/* L2-normalises rows [first, first+n) in place with the reference's f64 procedure
 * (l2_normalize_raw vector.rs:106-126, halfvec.rs:204-233 incl. its f32->half rounding);
 * norms_host (may be NULL) receives the f64 norms (vector_norm vector.rs:672-683) so the host can
 * skip zero-norm rows (build.rs:433-435). */
int hx_normalize_rows(hx_engine *e, uint64_t first, uint64_t n, double *norms_host);

/* Uploads a set of nq query vectors (host memory) into the engine's query slots 0..nq-1;
 * with normalize!=0 they are normalised first (scan.rs:749-751). */
int hx_set_queries(hx_engine *e, const void *queries_host, uint32_t nq, int normalize);
int hx_set_queries_device(hx_engine *e, const void *queries_dev, uint32_t nq, int normalize);

/* One query vector (host) against n rows: the batched form of the <=2M FunctionCall2Coll calls of one
 * candidate expansion (graph/mod.rs:205-224, scan.rs:362-383).  out[i] = d(query, row_ids[i]). */
int hx_distances(hx_engine *e, const void *query_host, const uint32_t *row_ids, uint32_t n, float *out);

/* Lock-step form: n_groups expansions in one launch.  Group g evaluates query group_query[g]
 * (a row id, or HX_QUERY_SLOT|slot) against row_ids[group_offsets[g] .. group_offsets[g+1]);
 * out is indexed like row_ids. */
int hx_distances_batch(hx_engine *e, uint32_t n_groups, const uint32_t *group_query,
                       const uint32_t *group_offsets, const uint32_t *row_ids, float *out);

/* Pairwise distances among w rows: out[i*w + j] = d(ids[i], ids[j]) (symmetric, 0 diagonal is
 * computed, not assumed).  The operand set of select_neighbors/check_element_closer,
 * graph/mod.rs:284-297,324-336. */
int hx_pairwise(hx_engine *e, const uint32_t *ids, uint32_t w, float *out_wxw);

/* Many small pair blocks in one launch (back-link pruning of one insert touches <= lm neighbours, each a
 * (lm+1)-row block, graph/mod.rs:458-486).  Group g has na[g] "A" rows followed by nb[g] "B" rows in
 * ids[group_offsets[g]..]:
 *    nb[g] == 0 : lower triangle of A x A, packed: out[out_offsets[g] + i*(i-1)/2 + j] = d(A_i, A_j), j < i
 *    nb[g]  > 0 : full rectangle:          out[out_offsets[g] + i*nb + j]      = d(A_i, B_j)
 * na+nb <= HX_PAIR_MAX_ROWS per group. */
#define HX_PAIR_MAX_ROWS 64
int hx_pairwise_many(hx_engine *e, uint32_t n_groups, const uint32_t *group_offsets,
                     const uint16_t *na, const uint16_t *nb, const uint32_t *ids,
                     const uint64_t *out_offsets, float *out);

/* The same pair blocks on the matrix cores (v_mfma_f32_32x32x16_f16), halfvec inner product only: BASELINE configs[3]'s "fp16 MFMA
 * batched-build distance GEMM".  Products of halves are exact in f32, so a value differs from hx_pairwise_many's only by the order of
 * the f32 additions: |delta| <= 2 * dim * 2^-24 * |a| |b|; norm2_out (nullable, one float per entry of ids) receives |row|^2 for
 * that bound.  hx_index_set_mfma makes the graph driver use these values for the decisions of check_element_closer
 * (graph/mod.rs:333) that fall outside the band and re-evaluate the others exactly, so graphs stay identical. */
int hx_pairwise_many_mfma(hx_engine *e, uint32_t n_groups, const uint32_t *group_offsets, const uint16_t *na, const uint16_t *nb,
                          const uint32_t *ids, const uint64_t *out_offsets, float *out, float *norm2_out);

/* Byte equality of row pairs (the datumIsEqual-style duplicate test, build.rs:491-500). */
int hx_rows_equal(hx_engine *e, uint32_t n_pairs, const uint32_t *a_ids, const uint32_t *b_ids, uint8_t *equal_out);

/* Kernel timing of the last hx_distances_batch / hx_pairwise_many launch on this handle, measured
 * with HIP events on the handle's stream (bench.py's roofline leg). */
int hx_set_timing(hx_engine *e, int enabled);
int hx_last_kernel_ms(hx_engine *e, float *ms);
/* Accumulated since the last reset, while timing is enabled: kind 0 = query-vs-rows kernel (units =
 * distances), kind 1 = pair-block kernel (units = pairs), 2 = traversal kernel, 3 = back-link kernels, 4 = MFMA pair kernel (units = pairs),
 * 5 = pipelined scans (hx_index_search_submit): units = distances, ms = the UNION of the overlapping launches' busy intervals,
 * 6 = select_neighbors on the Gram matrices of the matrix-core build path (k_wselect; kind 4 then is its GEMM, k_wgemm_f16). */
int hx_kernel_stats(hx_engine *e, int kind, uint64_t *launches, uint64_t *units, double *ms, int reset);

/* ------------------------------------------------------------------------------------------------
 * Index: host-side mirror of the reference's graph functions, issuing the batches above
 * ------------------------------------------------------------------------------------------------ */

/* HnswBuildState::new, build.rs:295-343: m in [2,100], ef_construction in [4,1000] and >= 2m
 * (options.rs:203-225, build.rs:865-867).  The index borrows the engine (which must outlive it). */
int hx_index_create(hx_engine *e, int m, int ef_construction, hx_index **out);
int hx_index_destroy(hx_index *ix);
const char *hx_index_last_error(const hx_index *ix);

/* Host worker threads used by the lock-step driver (default: hardware concurrency, capped at 16). */
int hx_index_set_threads(hx_index *ix, int n_threads);

/* build_callback (build.rs:400-535) for n rows that were already appended to the engine as rows
 * [first_row, first_row+n) (and normalised for cosine opclasses).  levels[i] is the level draw of row i
 * (the reference draws it with an unseeded rand::random, build.rs:373-377); tids[i] its heap TID.
 * batch == 1 reproduces the reference's strictly sequential schedule; batch > 1 runs `batch`
 * find_element_neighbors searches in lock-step against the graph as of the batch start, then applies
 * duplicate merge / back-links / entry-point update in row order (DESIGN.md "snapshot schedule").
 * elem_out[i] (may be NULL) = element (row id) that holds tid i. */
int hx_index_insert(hx_index *ix, uint64_t first_row, uint32_t n, const int32_t *levels,
                    const int64_t *tids, uint32_t batch, uint32_t *elem_out);

/* The same batch in stages, for sharing one batch between several GPUs of a node (each rank keeps a replica
 * of rows and graph; DESIGN.md "multi-GPU build"):
 *   begin  : add the b rows [first_row, first_row+b) as elements (every rank)
 *   search : find_element_neighbors for members [lo, hi) of the batch (this rank's slice)
 *   export_new / import_new : serialized neighbour lists of members, exchanged with an all-gather
 *   links  : duplicate merge + entry point (every rank), then update_neighbor_connections for the lists this
 *            rank owns (owner = target row id % world; only those ops are grouped and pruned here)
 *   export_links / import_links : the lists a rank pruned, as self-describing records, exchanged with an all-gather
 *   end    : closes the batch; elem_out[i] = element holding tid i
 * hx_index_insert is begin, search(0,b), links(0,1), end. */
int hx_index_batch_begin(hx_index *ix, uint64_t first_row, uint32_t b, const int32_t *levels, const int64_t *tids);
int hx_index_batch_search(hx_index *ix, uint32_t lo, uint32_t hi);
uint64_t hx_index_batch_new_bytes(const hx_index *ix, uint32_t lo, uint32_t hi);
int hx_index_batch_export_new(const hx_index *ix, uint32_t lo, uint32_t hi, void *buf);
int hx_index_batch_import_new(hx_index *ix, uint32_t lo, uint32_t hi, const void *buf);
int hx_index_batch_links(hx_index *ix, uint32_t rank, uint32_t world);
uint64_t hx_index_batch_links_bytes(const hx_index *ix);                          /* size of this rank's own export */
int hx_index_batch_export_links(const hx_index *ix, void *buf);                  /* self-describing records: target, layer, list */
int hx_index_batch_import_links(hx_index *ix, const void *buf, uint64_t nbytes); /* any rank's export */
int hx_index_batch_end(hx_index *ix, uint32_t *elem_out);

/* The same batch with the members' neighbour lists kept in DEVICE memory from the traversal kernel to the back-link kernels
 * (what hx_index_insert does by itself whenever hx_index_dbatch_supported; the reference builds one row at a time on one core,
 * build.rs:400-535, handler.rs:153-154).  Buffers are device pointers owned by the caller -- a multi-GPU build all-gathers them
 * (RCCL) between the stages without a host copy:
 *   supported          : 1 when a batch with these level draws can run device-resident (m <= 32, rows <= 8 KiB, no level beyond the
 *                        traversal kernel's 8 layers); otherwise use hx_index_batch_* for that batch
 *   record_bytes       : bytes of one member record (cnt[8] | ids[8][2m] | d[8][2m], 32-bit words)
 *   list_record_bytes  : bytes of one pruned-list record {target, layer, cnt, ids[2m], d[2m]}
 *   begin              : as hx_index_batch_begin (every rank)
 *   search             : find_element_neighbors for members [lo, hi); record of member i at d_records + (i - lo) * record_bytes
 *   links              : d_records = all b member records (record i = member i).  Every rank merges duplicates / moves the entry point and
 *                        scatters the members' lists into its graph copy; then prunes the back-link lists it owns (target % world == rank).
 *                        *n_list_records = lists it pruned (0 when world == 1)
 *   export_links       : copies this rank's list records to d_out (device)
 *   import_links       : scatters n list records of another rank into the graph copy
 *   end                : closes the batch; elem_out[i] = element holding tid i */
int hx_index_dbatch_supported(const hx_index *ix, const int32_t *levels, uint32_t b);
uint64_t hx_index_dbatch_record_bytes(const hx_index *ix);
uint64_t hx_index_dbatch_list_record_bytes(const hx_index *ix);
int hx_index_dbatch_begin(hx_index *ix, uint64_t first_row, uint32_t b, const int32_t *levels, const int64_t *tids);
int hx_index_dbatch_search(hx_index *ix, uint32_t lo, uint32_t hi, void *d_records);
int hx_index_dbatch_links(hx_index *ix, uint32_t rank, uint32_t world, const void *d_records, uint64_t *n_list_records);
int hx_index_dbatch_export_links(hx_index *ix, void *d_out);
int hx_index_dbatch_import_links(hx_index *ix, const void *d_list_records, uint64_t n);
/* Optional exchange between _search and _links: the W tables of members [lo, hi) (hx_index_dbatch_wtab_bytes per member; 0 = tables off) as written by
 * the rank that searched them -> d_out, and another rank's -> the engine.  With them a rank prunes the lists it owns with the look-ups a single GPU has
 * instead of streaming the rows again; results do not depend on it (a look-up returns the very bits a recomputation gives). */
uint64_t hx_index_dbatch_wtab_bytes(const hx_index *ix);
int hx_index_dbatch_export_wtabs(hx_index *ix, uint32_t lo, uint32_t hi, void *d_out);
int hx_index_dbatch_import_wtabs(hx_index *ix, uint32_t lo, uint32_t hi, const void *d_in);
int hx_index_dbatch_end(hx_index *ix, uint32_t *elem_out);

/* ---- f3: the on-disk paths on the engine ---------------------------------------------------------
 * aminsert (src/index/insert.rs:1227-1480) for n rows that were already appended to the engine: find_element_neighbors_on_disk
 * (:1021-1123 -- search_layer_disk per layer, the lm NEAREST taken without the heuristic, :1111-1117), find_duplicate_on_disk
 * (:1180-1214), the new element's tuple, update_neighbors_on_disk / get_update_index (:883-958, :500-739 -- incl. its skipped
 * new-vs-existing check :680-693) and the entry-point update.  batch == 1: one insert at a time, exactly the reference's result;
 * batch > 1: the neighbour searches of `batch` rows run against the same graph, as concurrent backends would.
 * Placement: searches in the traversal kernel (MODE 3), back-connections in k_update_runs (one wavefront per list) while no deleted / TID-less
 * element can be met, else k_update_index in waves; hx_index_set_fused(0): everything on the lock-step driver.  Every legal m; m > 32 needs
 * the device placement (rows <= 8 KiB) and an index whose deleted elements are unlinked (HX_E_ARG / HX_E_STATE otherwise). */
int hx_index_insert_ondisk(hx_index *ix, uint64_t first_row, uint32_t n, const int32_t *levels, const int64_t *tids,
                           uint32_t batch, uint32_t *elem_out);
/* ambulkdelete + amvacuumcleanup (src/index/vacuum.rs): dead_tids = the heap TIDs the bulk-delete callback reports dead.  Pass 1
 * removes them (:118-217); pass 2 repairs the entry point and every element with a deleted neighbour or a layer-0 list that is
 * not full (:230-285, :411-644) with repair_graph_element's search (:288-407: the deleted set and the element itself skipped,
 * ef_construction + 1); pass 3 marks the emptied elements deleted (:655-793).  batch as above. */
int hx_index_vacuum(hx_index *ix, const int64_t *dead_tids, uint64_t n_dead, uint32_t batch, uint64_t *n_deleted_out, uint64_t *n_repaired_out);
int hx_index_deleted(const hx_index *ix, uint32_t elem);      /* 1 after vacuum marked the element deleted */

/* graph export (what create_graph_pages/write_neighbor_tuples serialise, build.rs:545-821) */
uint32_t hx_index_size(const hx_index *ix);
int64_t hx_index_entry(const hx_index *ix);                   /* -1 = empty */
int hx_index_level(const hx_index *ix, uint32_t elem);        /* <0 = tombstoned duplicate */
int hx_index_neighbors(const hx_index *ix, uint32_t elem, int layer, uint32_t *ids_out, float *dist_out); /* returns count */
int hx_index_heaptids(const hx_index *ix, uint32_t elem, int64_t *tids_out);                              /* returns count (<= 10) */
/* bulk export: levels (negative = tombstoned duplicate) and one layer's neighbour lists for elements
 * [first, first+n): ids_out/dist_out are [n][lm] (lm = 2m at layer 0, m above), cnt_out[n]; elements whose level
 * is below `layer` get cnt 0.  dist_out may be NULL. */
int hx_index_export_levels(const hx_index *ix, uint32_t first, uint32_t n, int32_t *levels_out);
int hx_index_export_layer(const hx_index *ix, int layer, uint32_t first, uint32_t n,
                          uint32_t *ids_out, float *dist_out, uint16_t *cnt_out);
/* graph import: lets a rank install lists computed elsewhere (multi-GPU exchange) */
int hx_index_set_neighbors(hx_index *ix, uint32_t elem, int layer, uint32_t count, const uint32_t *ids, const float *dist);

/* distance-evaluation counters per hot loop (SURVEY 3 "hot-loop summary"): [0] entry point,
 * [1] search_layer expansions, [2] select_neighbors in find_element_neighbors, [3] back-link pruning, [4] scan */
int hx_index_counters(const hx_index *ix, uint64_t counters_out[8]);

/* Traversal placement.  enabled (default): find_element_neighbors and non-iterative scans run in the device-resident
 * fused kernel (one wavefront per search, heaps in LDS); a task that overflows its LDS budget is re-run by the
 * lock-step host driver.  disabled: everything runs in the lock-step driver.  Both produce identical results.
 * fused_stats: tasks given to the fused kernel and how many of them had to be re-run. */
int hx_index_set_fused(hx_index *ix, int enabled);
/* halfvec inner product (BASELINE configs[3]): select_neighbors' candidate-vs-candidate distances (graph/mod.rs:284-297, 324-336 over halfvec.rs:687-733)
 * are a true f16 GEMM and run on the matrix cores -- ON by default for that operator class, refused for every other.  Device-resident placement (default):
 * the traversal kernel stops after each layer's search, k_wgemm_f16 computes each member's W x W Gram matrix (W <= ef_construction <= 256; larger
 * ef_construction keeps the VALU select), k_wselect replays the heuristic on it.  Lock-step placement: pair blocks of select_neighbors / back-link pruning
 * through hx_pairwise_many_mfma.  Either way a decision whose two sides lie within the summation-order band 2 dim 2^-24 |a||b| is re-evaluated in the
 * canonical order, so lists and distance bits do not depend on the setting.  stats: decisions taken from MFMA values, pairs re-evaluated exactly. */
int hx_index_set_mfma(hx_index *ix, int enabled);
int hx_index_mfma_stats(const hx_index *ix, uint64_t *mfma_pairs, uint64_t *exact_pairs);
int hx_index_fused_stats(const hx_index *ix, uint64_t *tasks, uint64_t *redone);

/* host-side wall time of the lock-step driver since the last reset, seconds: [0] task state machines,
 * [1] request compaction, [2] request fill, [3] round copies+launches+wait (also the aminsert update kernels), [4] mirror sync of plain scans, [5] rounds (lock-step rounds + aminsert update-kernel launches), [6] fused kernel calls, [7] mirror sync, [8] link-stage setup (aminsert: the member stage), [9] link-stage kernels / lock-step (aminsert: the update stage), [10] hx_index_insert / hx_index_insert_ondisk total,
 * [11] batch_search total, [12] batch_begin total */
int hx_index_profile(const hx_index *ix, double seconds_out[16], int reset);

/* get_scan_items + amgettuple (scan.rs:458-530, 709-876), iterative_scan = off, for nq queries in
 * lock-step: the queries are the engine's query slots 0..nq-1 (hx_set_queries).  Per query, up to k heap
 * TIDs nearest first with their distances; counts_out[q] = number returned. */
int hx_index_search(hx_index *ix, uint32_t nq, uint32_t ef_search, uint32_t k,
                    int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out);

/* The same scan, pipelined.  A launch of the traversal kernel ends with a round of searches that no longer fills the chip (one search
 * long, ~2 ms of a 9 ms launch at 10 000 queries on 1M x 768); a host that has the NEXT batch of queries ready (one backend per connection:
 * scan.rs:709-876 is called per query, batches arrive continuously) submits it before it collects the previous one, each batch on a slot
 * (0 .. HX_SCAN_SLOTS-1) with a stream, staging buffers, visited tables and spill areas of its own, and the first round of batch N+1 fills the
 * CUs the last round of batch N leaves idle.
 *   hx_index_search_submit: queries = engine query slots first_query .. first_query+nq-1 (hx_set_queries uploads all batches' queries);
 *                           returns at once (HX_E_STATE if the slot is busy or the index cannot scan on the device: use hx_index_search then).
 *   hx_index_search_wait:   blocks until that batch is done; outputs as hx_index_search (per query k heap TIDs nearest first).
 * Results are those of hx_index_search bit for bit.  The index must not be modified between a submit and its wait (mutators return HX_E_STATE). */
#define HX_SCAN_SLOTS 4
int hx_index_search_submit(hx_index *ix, uint32_t slot, uint32_t first_query, uint32_t nq, uint32_t ef_search, uint32_t k);
int hx_index_search_wait(hx_index *ix, uint32_t slot, int64_t *tids_out, float *dist_out, uint32_t *elems_out, uint32_t *counts_out);

/* Iterative scan (hnsw.iterative_scan = relaxed_order | strict_order, scan.rs:794-875): per query, keeps
 * resuming from the discarded heap until `limit` tuples for which filter_pass[tid] != 0 were produced,
 * max_scan_tuples is exhausted, or the graph is. mode: 1 = relaxed_order, 2 = strict_order.
 * filter_pass may be NULL (every tuple passes); it is indexed by tid, n_filter entries. */
int hx_index_search_iterative(hx_index *ix, uint32_t nq, uint32_t ef_search, int mode, int64_t max_scan_tuples,
                              uint32_t limit, const uint8_t *filter_pass, uint64_t n_filter,
                              int64_t *tids_out, float *dist_out, uint32_t *counts_out);

/* One scan with a NULL order-by value (`ORDER BY val <-> NULL`; scan.rs:186-187: every element gets distance 0.0, the distance procedure is
 * never called).  Pure graph traversal on the host -- there is nothing to compute -- in the order Rust's BinaryHeap gives equal keys.
 * mode 0: plain scan (k = limit), 1 / 2: iterative as above.  tids_out / elems_out (may be NULL): `limit` entries; *count_out = tuples returned. */
int hx_index_search_null(hx_index *ix, uint32_t ef_search, int mode, int64_t max_scan_tuples, uint32_t limit,
                         const uint8_t *filter_pass, uint64_t n_filter, int64_t *tids_out, uint32_t *elems_out, uint32_t *count_out);

/* ---- graph -> PostgreSQL index pages -----------------------------------------
 * Serialises the built graph as the byte image of the HNSW relation fork, exactly as ambuild flushes it:
 * create_meta_page + create_graph_pages + write_neighbor_tuples + update_meta_page (src/index/build.rs:545-821)
 * with the page structures of src/types/hnsw.rs:20-169 (meta page: magic 0xA953A953, version 1; element tuple:
 * 72-byte header {type 1, level, deleted, version, 10 heap TIDs, neighbour TID} + the value's varlena; neighbour
 * tuple: {type 2, version, count} + (level+2)*m index TIDs, layers top -> 0, invalid-TID padding; element and
 * neighbour tuple co-located when both fit; pages chained through the special area's nextblkno, page_id 0xFF90).
 * Page 0 is the meta page, data pages follow; every page is HX_PAGE_SIZE bytes (BLCKSZ 8192, layout version 4).
 * The value varlena is rebuilt from the engine's payload: {vl_len_ = size << 2, dim:i16, unused:i16} for
 * vector / halfvec (vector.rs:43-48, halfvec.rs:41-46), {vl_len_, bit_len:i32} for bit (bitvec.rs:28-37).
 * Heap TIDs: the int64 tids given to hx_index_insert are read as (block << 16) | offset; offsets start at 1
 * (FirstOffsetNumber) -- offset 0 is PostgreSQL's invalid TID and reads back as "no TID".
 * Tombstoned duplicates (merged into another element's heap TIDs) get no tuples, as in build.rs:482-512.
 *   pages_out == NULL: only computes *n_pages_out (and the element locations, if asked);
 *   else writes min(cap_pages, needed) pages and fails with HX_E_ARG when cap_pages is too small.
 * elem_blkno_out / elem_offno_out (nullable, one per element): where each element tuple went (0xFFFFFFFF / 0 for
 * tombstones) -- the host's disk_locs (build.rs:283-285). */
#define HX_PAGE_SIZE 8192
int hx_index_serialize_pages(const hx_index *ix, uint8_t *pages_out, uint64_t cap_pages, uint64_t *n_pages_out,
                             uint32_t *elem_blkno_out, uint16_t *elem_offno_out);

/* ---- PostgreSQL index pages -> engine (the scan side's (blkno, offno) -> row map) -----------
 * The inverse of hx_index_serialize_pages, and what a scan-side host does once per index: walks the page chain from
 * HNSW_HEAD_BLKNO (block 1), reads element tuples the way load_element does (type 1, not deleted, payload at
 * etup + 72, heap TIDs up to the first invalid one: scan.rs:155-228) and neighbour tuples the way load_neighbor_tids
 * does (version / count check, layer slots at 4 + ((level - layer) * m + i) * 6, up to the first invalid TID:
 * scan.rs:236-283), appends the payloads to the (empty) engine in (block, offset) order -- element index == row id ==
 * order of appearance -- and rebuilds the graph, heap TIDs ((block << 16) | offset) and the entry point of the meta
 * page.  Deleted elements are not loaded and TIDs that point at them are dropped (the scan skips them the same way).
 * The page format stores no distances; they are recomputed with the batched distance kernel, so a loaded index is
 * indistinguishable from the one that was serialised (same bits), and inserts may continue on it.
 * Requires an empty index on an empty engine whose dtype/dim match the stored varlenas and whose m matches the meta
 * page.  elem_blkno_out / elem_offno_out (nullable, capacity `cap_elems`): the (blkno, offno) -> row map, by row;
 * *n_elems_out = elements loaded. */
int hx_index_load_pages(hx_index *ix, const uint8_t *pages, uint64_t n_pages, uint32_t *elem_blkno_out, uint16_t *elem_offno_out,
                        uint64_t cap_elems, uint64_t *n_elems_out);

/* Version-checked invalidation of a loaded index.  The scan side trusts a neighbour tuple only while its version equals
 * the element's (load_neighbor_tids, scan.rs:262-265; HnswElementTupleData.version, types/hnsw.rs:120 -- vacuum bumps it when it frees a
 * tuple, so a reused slot no longer matches).  A host that changes pages under a loaded mirror reports the element tuples it touched with
 * their CURRENT versions: every (blkno[i], offno[i]) whose version differs from the one that was loaded (versions == NULL: every listed
 * tuple) is dropped from the mirror -- no heap TIDs, no lists, unlinked from all neighbour lists, entry point re-picked -- so device scans
 * neither return nor traverse a tuple whose slot now holds something else.  *n_dropped_out = elements dropped. */
int hx_index_invalidate(hx_index *ix, uint32_t n, const uint32_t *blkno, const uint16_t *offno, const uint8_t *versions, uint32_t *n_dropped_out);

#ifdef __cplusplus
}
#endif
#endif /* HNSWRX_H */
