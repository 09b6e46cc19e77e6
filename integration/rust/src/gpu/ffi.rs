//! `extern "C"` bindings of `include/hnswrx.h` (libhnswrx.so: the MI355X HNSW distance engine) for the pgvector-rx crate.
//!
//! Reference-side file: add as `src/gpu/ffi.rs` (+ `pub mod gpu;` in `src/lib.rs`); `build.rs` of the crate adds
//! `println!("cargo:rustc-link-lib=dylib=hnswrx");` and the library's search path.  This is what `bindgen include/hnswrx.h`
//! produces, written out by hand because the build image has no Rust toolchain: keep it in step with the header
//! (`tests/test_abi.py` checks that every symbol the header declares is exported by the library).
#![allow(non_camel_case_types, dead_code)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct hx_engine {
    _private: [u8; 0],
}
#[repr(C)]
pub struct hx_index {
    _private: [u8; 0],
}

pub const HX_ABI_VERSION: c_int = 1;
// enum hx_dtype
pub const HX_F32: c_int = 0;
pub const HX_F16: c_int = 1;
pub const HX_BIT: c_int = 2;
pub const HX_SPARSE: c_int = 3; // sparsevec as fixed-size records { nnz, pad[3], index[cap], value[cap] }, cap = min(dim, 1000): include/hnswrx.h
// enum hx_metric: which opclass support FUNCTION 1 the index names (src/hnsw_constants.rs:12)
pub const HX_L2SQ: c_int = 0;
pub const HX_NEG_IP: c_int = 1;
pub const HX_L1: c_int = 2;
pub const HX_HAMMING: c_int = 3;
pub const HX_JACCARD: c_int = 4;
// enum hx_status
pub const HX_OK: c_int = 0;
pub const HX_E_ARG: c_int = -1;
pub const HX_E_DIM: c_int = -2;
pub const HX_E_NOMEM: c_int = -3;
pub const HX_E_HIP: c_int = -4;
pub const HX_E_NODEVICE: c_int = -5;
pub const HX_E_STATE: c_int = -6;

pub const HX_QUERY_SLOT: u32 = 0x8000_0000;
pub const HX_PAIR_MAX_ROWS: u32 = 64;
pub const HX_PAGE_SIZE: usize = 8192;

#[link(name = "hnswrx")]
extern "C" {
    pub fn hx_abi_version() -> c_int;
    pub fn hx_last_error(e: *const hx_engine) -> *const c_char;

    // ---- engine: device row store + batched distance kernels (replaces graph::DistanceFn, src/graph/mod.rs:144-145) ----
    pub fn hx_create(device: c_int, dtype: c_int, metric: c_int, dim: c_int, capacity_rows: u64, out: *mut *mut hx_engine) -> c_int;
    pub fn hx_destroy(e: *mut hx_engine) -> c_int;
    pub fn hx_dim(e: *const hx_engine) -> c_int;
    pub fn hx_row_bytes(e: *const hx_engine) -> u64;
    pub fn hx_num_rows(e: *const hx_engine) -> u64;
    pub fn hx_stream(e: *const hx_engine) -> *mut c_void;
    pub fn hx_append_rows(e: *mut hx_engine, rows_host: *const c_void, n: u64, first_row_id: *mut u64) -> c_int;
    pub fn hx_append_rows_device(e: *mut hx_engine, rows_dev: *const c_void, n: u64, first_row_id: *mut u64) -> c_int;
    pub fn hx_pop_rows(e: *mut hx_engine, n: u64) -> c_int;
    pub fn hx_read_rows(e: *mut hx_engine, first: u64, n: u64, rows_host: *mut c_void) -> c_int;
    pub fn hx_normalize_rows(e: *mut hx_engine, first: u64, n: u64, norms_host: *mut f64) -> c_int;
    pub fn hx_set_queries(e: *mut hx_engine, queries_host: *const c_void, nq: u32, normalize: c_int) -> c_int;
    pub fn hx_set_queries_device(e: *mut hx_engine, queries_dev: *const c_void, nq: u32, normalize: c_int) -> c_int;
    pub fn hx_distances(e: *mut hx_engine, query_host: *const c_void, row_ids: *const u32, n: u32, out: *mut f32) -> c_int;
    pub fn hx_distances_batch(e: *mut hx_engine, n_groups: u32, group_query: *const u32, group_offsets: *const u32,
                              row_ids: *const u32, out: *mut f32) -> c_int;
    pub fn hx_pairwise(e: *mut hx_engine, ids: *const u32, w: u32, out_wxw: *mut f32) -> c_int;
    pub fn hx_pairwise_many(e: *mut hx_engine, n_groups: u32, group_offsets: *const u32, na: *const u16, nb: *const u16,
                            ids: *const u32, out_offsets: *const u64, out: *mut f32) -> c_int;
    pub fn hx_pairwise_many_mfma(e: *mut hx_engine, n_groups: u32, group_offsets: *const u32, na: *const u16, nb: *const u16,
                                 ids: *const u32, out_offsets: *const u64, out: *mut f32, norm2_out: *mut f32) -> c_int;
    pub fn hx_rows_equal(e: *mut hx_engine, n_pairs: u32, a_ids: *const u32, b_ids: *const u32, equal_out: *mut u8) -> c_int;
    pub fn hx_set_timing(e: *mut hx_engine, enabled: c_int) -> c_int;
    pub fn hx_last_kernel_ms(e: *mut hx_engine, ms: *mut f32) -> c_int;
    pub fn hx_kernel_stats(e: *mut hx_engine, kind: c_int, launches: *mut u64, units: *mut u64, ms: *mut f64, reset: c_int) -> c_int;

    // ---- index: the graph functions of src/graph/mod.rs, src/index/build.rs, scan.rs, insert.rs, vacuum.rs over the engine ----
    pub fn hx_index_create(e: *mut hx_engine, m: c_int, ef_construction: c_int, out: *mut *mut hx_index) -> c_int;
    pub fn hx_index_destroy(ix: *mut hx_index) -> c_int;
    pub fn hx_index_last_error(ix: *const hx_index) -> *const c_char;
    pub fn hx_index_set_threads(ix: *mut hx_index, n_threads: c_int) -> c_int;
    pub fn hx_index_insert(ix: *mut hx_index, first_row: u64, n: u32, levels: *const i32, tids: *const i64, batch: u32,
                           elem_out: *mut u32) -> c_int;
    pub fn hx_index_insert_ondisk(ix: *mut hx_index, first_row: u64, n: u32, levels: *const i32, tids: *const i64, batch: u32,
                                  elem_out: *mut u32) -> c_int;
    pub fn hx_index_vacuum(ix: *mut hx_index, dead_tids: *const i64, n_dead: u64, batch: u32, n_deleted_out: *mut u64,
                           n_repaired_out: *mut u64) -> c_int;
    pub fn hx_index_deleted(ix: *const hx_index, elem: u32) -> c_int;
    // staged batches (several GPUs share one batch): host buffers ...
    pub fn hx_index_batch_begin(ix: *mut hx_index, first_row: u64, b: u32, levels: *const i32, tids: *const i64) -> c_int;
    pub fn hx_index_batch_search(ix: *mut hx_index, lo: u32, hi: u32) -> c_int;
    pub fn hx_index_batch_new_bytes(ix: *const hx_index, lo: u32, hi: u32) -> u64;
    pub fn hx_index_batch_export_new(ix: *const hx_index, lo: u32, hi: u32, buf: *mut c_void) -> c_int;
    pub fn hx_index_batch_import_new(ix: *mut hx_index, lo: u32, hi: u32, buf: *const c_void) -> c_int;
    pub fn hx_index_batch_links(ix: *mut hx_index, rank: u32, world: u32) -> c_int;
    pub fn hx_index_batch_links_bytes(ix: *const hx_index) -> u64;
    pub fn hx_index_batch_export_links(ix: *const hx_index, buf: *mut c_void) -> c_int;
    pub fn hx_index_batch_import_links(ix: *mut hx_index, buf: *const c_void, nbytes: u64) -> c_int;
    pub fn hx_index_batch_end(ix: *mut hx_index, elem_out: *mut u32) -> c_int;
    // ... and device buffers (RCCL all-gathers them without a host copy)
    pub fn hx_index_dbatch_supported(ix: *const hx_index, levels: *const i32, b: u32) -> c_int;
    pub fn hx_index_dbatch_record_bytes(ix: *const hx_index) -> u64;
    pub fn hx_index_dbatch_list_record_bytes(ix: *const hx_index) -> u64;
    pub fn hx_index_dbatch_begin(ix: *mut hx_index, first_row: u64, b: u32, levels: *const i32, tids: *const i64) -> c_int;
    pub fn hx_index_dbatch_search(ix: *mut hx_index, lo: u32, hi: u32, d_records: *mut c_void) -> c_int;
    pub fn hx_index_dbatch_links(ix: *mut hx_index, rank: u32, world: u32, d_records: *const c_void, n_list_records: *mut u64) -> c_int;
    pub fn hx_index_dbatch_export_links(ix: *mut hx_index, d_out: *mut c_void) -> c_int;
    pub fn hx_index_dbatch_import_links(ix: *mut hx_index, d_list_records: *const c_void, n: u64) -> c_int;
    pub fn hx_index_dbatch_wtab_bytes(ix: *const hx_index) -> u64;
    pub fn hx_index_dbatch_export_wtabs(ix: *mut hx_index, lo: u32, hi: u32, d_out: *mut c_void) -> c_int;
    pub fn hx_index_dbatch_import_wtabs(ix: *mut hx_index, lo: u32, hi: u32, d_in: *const c_void) -> c_int;
    pub fn hx_index_dbatch_end(ix: *mut hx_index, elem_out: *mut u32) -> c_int;
    // graph export / import
    pub fn hx_index_size(ix: *const hx_index) -> u32;
    pub fn hx_index_entry(ix: *const hx_index) -> i64;
    pub fn hx_index_level(ix: *const hx_index, elem: u32) -> c_int;
    pub fn hx_index_neighbors(ix: *const hx_index, elem: u32, layer: c_int, ids_out: *mut u32, dist_out: *mut f32) -> c_int;
    pub fn hx_index_heaptids(ix: *const hx_index, elem: u32, tids_out: *mut i64) -> c_int;
    pub fn hx_index_export_levels(ix: *const hx_index, first: u32, n: u32, levels_out: *mut i32) -> c_int;
    pub fn hx_index_export_layer(ix: *const hx_index, layer: c_int, first: u32, n: u32, ids_out: *mut u32, dist_out: *mut f32,
                                 cnt_out: *mut u16) -> c_int;
    pub fn hx_index_set_neighbors(ix: *mut hx_index, elem: u32, layer: c_int, count: u32, ids: *const u32, dist: *const f32) -> c_int;
    pub fn hx_index_counters(ix: *const hx_index, counters_out: *mut u64) -> c_int;
    pub fn hx_index_set_fused(ix: *mut hx_index, enabled: c_int) -> c_int;
    pub fn hx_index_set_mfma(ix: *mut hx_index, enabled: c_int) -> c_int;
    pub fn hx_index_mfma_stats(ix: *const hx_index, mfma_pairs: *mut u64, exact_pairs: *mut u64) -> c_int;
    pub fn hx_index_fused_stats(ix: *const hx_index, tasks: *mut u64, redone: *mut u64) -> c_int;
    pub fn hx_index_profile(ix: *const hx_index, seconds_out: *mut f64, reset: c_int) -> c_int;
    // scans
    pub fn hx_index_search(ix: *mut hx_index, nq: u32, ef_search: u32, k: u32, tids_out: *mut i64, dist_out: *mut f32,
                           elems_out: *mut u32, counts_out: *mut u32) -> c_int;
    /// Pipelined scan (include/hnswrx.h): submit the next batch of queries before collecting the previous one; slots 0..HX_SCAN_SLOTS-1.
    pub fn hx_index_search_submit(ix: *mut hx_index, slot: u32, first_query: u32, nq: u32, ef_search: u32, k: u32) -> c_int;
    pub fn hx_index_search_wait(ix: *mut hx_index, slot: u32, tids_out: *mut i64, dist_out: *mut f32, elems_out: *mut u32,
                                counts_out: *mut u32) -> c_int;
    pub fn hx_index_search_iterative(ix: *mut hx_index, nq: u32, ef_search: u32, mode: c_int, max_scan_tuples: i64, limit: u32,
                                     filter_pass: *const u8, n_filter: u64, tids_out: *mut i64, dist_out: *mut f32,
                                     counts_out: *mut u32) -> c_int;
    /// `ORDER BY val <-> NULL` (scan.rs:186-187): traversal with every distance 0.0, no kernel involved.
    pub fn hx_index_search_null(ix: *mut hx_index, ef_search: u32, mode: c_int, max_scan_tuples: i64, limit: u32, filter_pass: *const u8,
                                n_filter: u64, tids_out: *mut i64, elems_out: *mut u32, count_out: *mut u32) -> c_int;
    // PostgreSQL page image <-> engine
    pub fn hx_index_serialize_pages(ix: *const hx_index, pages_out: *mut u8, cap_pages: u64, n_pages_out: *mut u64,
                                    elem_blkno_out: *mut u32, elem_offno_out: *mut u16) -> c_int;
    pub fn hx_index_load_pages(ix: *mut hx_index, pages: *const u8, n_pages: u64, elem_blkno_out: *mut u32, elem_offno_out: *mut u16,
                               cap_elems: u64, n_elems_out: *mut u64) -> c_int;
    pub fn hx_index_invalidate(ix: *mut hx_index, n: u32, blkno: *const u32, offno: *const u16, versions: *const u8,
                               n_dropped_out: *mut u32) -> c_int;
}

/// Every non-zero status becomes a PostgreSQL ERROR, as `pgrx::error!` does today (src/index/build.rs:399, src/index/scan.rs:708).
///
/// # Safety
/// `e` must be null or a live engine handle.
pub unsafe fn ck(e: *const hx_engine, rc: c_int) {
    if rc != HX_OK {
        let msg = CStr::from_ptr(hx_last_error(e)).to_string_lossy().into_owned();
        pgrx::error!("hnswrx: {} (status {})", msg, rc);
    }
}

/// The same for index calls.
///
/// # Safety
/// `ix` must be a live index handle.
pub unsafe fn ck_index(ix: *const hx_index, rc: c_int) {
    if rc != HX_OK {
        let msg = CStr::from_ptr(hx_index_last_error(ix)).to_string_lossy().into_owned();
        pgrx::error!("hnswrx: {} (status {})", msg, rc);
    }
}

/// `(hx_dtype, hx_metric, normalise)` for an operator class, from the name of its support FUNCTION 1
/// (src/types/vector.rs:842-861, halfvec.rs:1046-1069, bitvec.rs:223-233).
pub fn engine_kind(distance_proc: &str) -> Option<(c_int, c_int, bool)> {
    Some(match distance_proc {
        "vector_l2_squared_distance" => (HX_F32, HX_L2SQ, false),
        "vector_negative_inner_product" => (HX_F32, HX_NEG_IP, false), // cosine opclasses: same FUNCTION 1 plus FUNCTION 2 (norm) => normalise = true
        "l1_distance" => (HX_F32, HX_L1, false),
        "halfvec_l2_squared_distance" => (HX_F16, HX_L2SQ, false),
        "halfvec_negative_inner_product" => (HX_F16, HX_NEG_IP, false),
        "halfvec_l1_distance" => (HX_F16, HX_L1, false),
        "hamming_distance" => (HX_BIT, HX_HAMMING, false),
        "jaccard_distance" => (HX_BIT, HX_JACCARD, false),
        "sparsevec_l2_squared_distance" => (HX_SPARSE, HX_L2SQ, false), // sparsevec.rs:1552-1582
        "sparsevec_negative_inner_product" => (HX_SPARSE, HX_NEG_IP, false),
        "sparsevec_l1_distance" => (HX_SPARSE, HX_L1, false),
        _ => return None,
    })
}
