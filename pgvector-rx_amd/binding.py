"""ctypes binding of include/hnswrx.h (no arithmetic here; every call lands in libhnswrx.so)."""
import ctypes as C
import os

import numpy as np

from . import build as _build

F32, F16, BIT, SPARSE = 0, 1, 2, 3
L2SQ, NEG_IP, L1, HAMMING, JACCARD = 0, 1, 2, 3, 4
QUERY_SLOT = 0x80000000
_NP = {F32: np.float32, F16: np.uint16, BIT: np.uint8, SPARSE: np.uint8}
SPARSE_MAX_NNZ = 1000


def sparse_record_bytes(dim):
    return (16 + 8 * min(dim, SPARSE_MAX_NNZ) + 15) & ~15


def pack_sparse(dim, rows):
    """rows: iterable of (indices, values) -- ascending 0-based indices, at most 1000 of them -> uint8[n, record] in the engine's HX_SPARSE
    record layout {int32 nnz; int32 pad[3]; int32 index[cap]; float32 value[cap]} (include/hnswrx.h), unused slots zero."""
    cap, rb = min(dim, SPARSE_MAX_NNZ), sparse_record_bytes(dim)
    rows = list(rows)
    out = np.zeros((len(rows), rb), np.uint8)
    for r, (idx, val) in enumerate(rows):
        idx, val = np.asarray(idx, np.int32), np.asarray(val, np.float32)
        k = len(idx)
        assert k == len(val) and k <= cap and (k < 2 or np.all(np.diff(idx) > 0)) and (k == 0 or (idx[0] >= 0 and idx[-1] < dim))
        out[r, 0:4] = np.array([k], np.int32).view(np.uint8)
        out[r, 16:16 + 4 * k] = idx.view(np.uint8)
        out[r, 16 + 4 * cap:16 + 4 * cap + 4 * k] = val.view(np.uint8)
    return out

_lib = None


class HxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("hnswrx error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return _build.LIB


def lib():
    """Loads libhnswrx.so, (re)building it when hipcc is available and the sources are newer.
    There is no fallback: if the library cannot be built or loaded this raises."""
    global _lib
    if _lib is not None:
        return _lib
    if _build.stale():
        _build.build()
    L = C.CDLL(_build.LIB)
    vp, i32, u32, u64, i64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_int64
    sig = {
        "hx_abi_version": (i32, []),
        "hx_last_error": (C.c_char_p, [vp]),
        "hx_create": (i32, [i32, i32, i32, i32, u64, C.POINTER(vp)]),
        "hx_destroy": (i32, [vp]),
        "hx_dim": (i32, [vp]),
        "hx_row_bytes": (u64, [vp]),
        "hx_num_rows": (u64, [vp]),
        "hx_stream": (vp, [vp]),
        "hx_append_rows": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "hx_append_rows_device": (i32, [vp, vp, u64, C.POINTER(u64)]),
        "hx_pop_rows": (i32, [vp, u64]),
        "hx_read_rows": (i32, [vp, u64, u64, vp]),
        "hx_normalize_rows": (i32, [vp, u64, u64, vp]),
        "hx_set_queries": (i32, [vp, vp, u32, i32]),
        "hx_set_queries_device": (i32, [vp, vp, u32, i32]),
        "hx_distances": (i32, [vp, vp, vp, u32, vp]),
        "hx_distances_batch": (i32, [vp, u32, vp, vp, vp, vp]),
        "hx_pairwise": (i32, [vp, vp, u32, vp]),
        "hx_pairwise_many": (i32, [vp, u32, vp, vp, vp, vp, vp, vp]),
        "hx_pairwise_many_mfma": (i32, [vp, u32, vp, vp, vp, vp, vp, vp, vp]),
        "hx_rows_equal": (i32, [vp, u32, vp, vp, vp]),
        "hx_set_timing": (i32, [vp, i32]),
        "hx_last_kernel_ms": (i32, [vp, C.POINTER(C.c_float)]),
        "hx_kernel_stats": (i32, [vp, i32, C.POINTER(u64), C.POINTER(u64), C.POINTER(C.c_double), i32]),
        "hx_index_create": (i32, [vp, i32, i32, C.POINTER(vp)]),
        "hx_index_destroy": (i32, [vp]),
        "hx_index_last_error": (C.c_char_p, [vp]),
        "hx_index_set_threads": (i32, [vp, i32]),
        "hx_index_insert": (i32, [vp, u64, u32, vp, vp, u32, vp]),
        "hx_index_batch_begin": (i32, [vp, u64, u32, vp, vp]),
        "hx_index_batch_search": (i32, [vp, u32, u32]),
        "hx_index_batch_new_bytes": (u64, [vp, u32, u32]),
        "hx_index_batch_export_new": (i32, [vp, u32, u32, vp]),
        "hx_index_batch_import_new": (i32, [vp, u32, u32, vp]),
        "hx_index_batch_links": (i32, [vp, u32, u32]),
        "hx_index_batch_links_bytes": (u64, [vp]),
        "hx_index_batch_export_links": (i32, [vp, vp]),
        "hx_index_batch_import_links": (i32, [vp, vp, u64]),
        "hx_index_batch_end": (i32, [vp, vp]),
        "hx_index_insert_ondisk": (i32, [vp, u64, u32, vp, vp, u32, vp]),
        "hx_index_vacuum": (i32, [vp, vp, u64, u32, C.POINTER(u64), C.POINTER(u64)]),
        "hx_index_deleted": (i32, [vp, u32]),
        "hx_index_invalidate": (i32, [vp, u32, vp, vp, vp, C.POINTER(u32)]),
        "hx_index_dbatch_supported": (i32, [vp, vp, u32]),
        "hx_index_dbatch_record_bytes": (u64, [vp]),
        "hx_index_dbatch_list_record_bytes": (u64, [vp]),
        "hx_index_dbatch_begin": (i32, [vp, u64, u32, vp, vp]),
        "hx_index_dbatch_search": (i32, [vp, u32, u32, vp]),
        "hx_index_dbatch_links": (i32, [vp, u32, u32, vp, C.POINTER(u64)]),
        "hx_index_dbatch_export_links": (i32, [vp, vp]),
        "hx_index_dbatch_import_links": (i32, [vp, vp, u64]),
        "hx_index_dbatch_wtab_bytes": (u64, [vp]),
        "hx_index_dbatch_export_wtabs": (i32, [vp, u32, u32, vp]),
        "hx_index_dbatch_import_wtabs": (i32, [vp, u32, u32, vp]),
        "hx_index_dbatch_end": (i32, [vp, vp]),
        "hx_index_size": (u32, [vp]),
        "hx_index_entry": (i64, [vp]),
        "hx_index_level": (i32, [vp, u32]),
        "hx_index_neighbors": (i32, [vp, u32, i32, vp, vp]),
        "hx_index_heaptids": (i32, [vp, u32, vp]),
        "hx_index_export_levels": (i32, [vp, u32, u32, vp]),
        "hx_index_export_layer": (i32, [vp, i32, u32, u32, vp, vp, vp]),
        "hx_index_set_neighbors": (i32, [vp, u32, i32, u32, vp, vp]),
        "hx_index_counters": (i32, [vp, vp]),
        "hx_index_profile": (i32, [vp, vp, i32]),
        "hx_index_set_fused": (i32, [vp, i32]),
        "hx_index_set_mfma": (i32, [vp, i32]),
        "hx_index_mfma_stats": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "hx_index_fused_stats": (i32, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "hx_index_search": (i32, [vp, u32, u32, u32, vp, vp, vp, vp]),
        "hx_index_search_submit": (i32, [vp, u32, u32, u32, u32, u32]),
        "hx_index_search_wait": (i32, [vp, u32, vp, vp, vp, vp]),
        "hx_index_search_iterative": (i32, [vp, u32, u32, i32, i64, u32, vp, u64, vp, vp, vp]),
        "hx_index_search_null": (i32, [vp, u32, i32, i64, u32, vp, u64, vp, vp, vp]),
        "hx_index_serialize_pages": (i32, [vp, vp, u64, C.POINTER(u64), vp, vp]),
        "hx_index_load_pages": (i32, [vp, vp, u64, vp, vp, u64, C.POINTER(u64)]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("HX_LIB") and not hasattr(L, name):
            continue                # an older variant library loaded for an A/B timing run (tools/qsweep.py) may lack newer entry points
        fn = getattr(L, name)       # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


ABI_SYMBOLS = None  # filled lazily by tests from include/hnswrx.h


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


class Engine:
    """hx_engine: device row store + batched distance kernels (replaces graph::DistanceFn)."""

    def __init__(self, dtype, metric, dim, capacity, device=0):
        self.dtype, self.metric, self.dim = dtype, metric, dim
        h = C.c_void_p()
        rc = lib().hx_create(device, dtype, metric, dim, int(capacity), C.byref(h))
        if rc:
            raise HxError(rc, lib().hx_last_error(None).decode())
        self.h = h
        self.row_bytes = lib().hx_row_bytes(h)

    def close(self):
        if getattr(self, "h", None):
            lib().hx_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc:
            raise HxError(rc, lib().hx_last_error(self.h).decode())

    def _rows(self, a):
        a = np.ascontiguousarray(a, dtype=_NP[self.dtype])
        if a.nbytes % self.row_bytes:
            raise ValueError("row payload size mismatch")
        return a, a.nbytes // self.row_bytes

    @property
    def num_rows(self):
        return lib().hx_num_rows(self.h)

    def append(self, rows):
        a, n = self._rows(rows)
        first = C.c_uint64()
        self._ck(lib().hx_append_rows(self.h, _p(a), n, C.byref(first)))
        return first.value

    def append_device(self, dev_ptr, n):
        first = C.c_uint64()
        self._ck(lib().hx_append_rows_device(self.h, C.c_void_p(dev_ptr), n, C.byref(first)))
        return first.value

    def pop(self, n):
        self._ck(lib().hx_pop_rows(self.h, n))

    def read_rows(self, first, n):
        out = np.empty((n, self.row_bytes // np.dtype(_NP[self.dtype]).itemsize), _NP[self.dtype])
        self._ck(lib().hx_read_rows(self.h, first, n, _p(out)))
        return out

    def normalize_rows(self, first, n):
        norms = np.empty(n, np.float64)
        self._ck(lib().hx_normalize_rows(self.h, first, n, _p(norms)))
        return norms

    def set_queries(self, q, normalize=False):
        a, n = self._rows(q)
        self._ck(lib().hx_set_queries(self.h, _p(a), n, int(normalize)))
        return n

    def set_queries_device(self, dev_ptr, n, normalize=False):
        self._ck(lib().hx_set_queries_device(self.h, C.c_void_p(dev_ptr), n, int(normalize)))

    def distances(self, query, row_ids):
        q, _ = self._rows(query)
        ids = _u32(row_ids)
        out = np.empty(len(ids), np.float32)
        self._ck(lib().hx_distances(self.h, _p(q), _p(ids), len(ids), _p(out)))
        return out

    def distances_batch(self, group_query, group_offsets, row_ids):
        gq, go, ids = _u32(group_query), _u32(group_offsets), _u32(row_ids)
        out = np.empty(len(ids), np.float32)
        self._ck(lib().hx_distances_batch(self.h, len(gq), _p(gq), _p(go), _p(ids), _p(out)))
        return out

    def pairwise(self, ids):
        ids = _u32(ids)
        out = np.empty((len(ids), len(ids)), np.float32)
        self._ck(lib().hx_pairwise(self.h, _p(ids), len(ids), _p(out)))
        return out

    def pairwise_many(self, groups, mfma=False):
        """groups: list of (A_ids, B_ids or None).  Returns one array per group: packed lower triangle
        (B None) or an (na, nb) rectangle.  mfma: on the matrix cores (halfvec inner product; values in MFMA summation order);
        then also returns |row|^2 per id (concatenated A then B ids of every group)."""
        off, na, nb, ids, ooff, n_out = [0], [], [], [], [], 0
        for a, b in groups:
            b = [] if b is None else b
            ids.extend(a)
            ids.extend(b)
            na.append(len(a))
            nb.append(len(b))
            off.append(len(ids))
            ooff.append(n_out)
            n_out += len(a) * len(b) if len(b) else len(a) * (len(a) - 1) // 2
        off, ids = _u32(off), _u32(ids)
        na, nb = np.asarray(na, np.uint16), np.asarray(nb, np.uint16)
        ooff_a = np.asarray(ooff, np.uint64)
        out = np.empty(max(n_out, 1), np.float32)
        norm2 = np.empty(len(ids), np.float32) if mfma else None
        if mfma:
            self._ck(lib().hx_pairwise_many_mfma(self.h, len(groups), _p(off), _p(na), _p(nb), _p(ids), _p(ooff_a), _p(out), _p(norm2)))
        else:
            self._ck(lib().hx_pairwise_many(self.h, len(groups), _p(off), _p(na), _p(nb), _p(ids), _p(ooff_a), _p(out)))
        res = []
        for g, (a, b) in enumerate(groups):
            if b is None or len(b) == 0:
                res.append(out[ooff[g]:ooff[g] + len(a) * (len(a) - 1) // 2].copy())
            else:
                res.append(out[ooff[g]:ooff[g] + len(a) * len(b)].reshape(len(a), len(b)).copy())
        return (res, norm2) if mfma else res

    def rows_equal(self, a_ids, b_ids):
        a, b = _u32(a_ids), _u32(b_ids)
        out = np.empty(len(a), np.uint8)
        self._ck(lib().hx_rows_equal(self.h, len(a), _p(a), _p(b), _p(out)))
        return out.astype(bool)

    def set_timing(self, on=True):
        self._ck(lib().hx_set_timing(self.h, int(on)))

    def kernel_stats(self, kind, reset=False):
        l, u, ms = C.c_uint64(), C.c_uint64(), C.c_double()
        self._ck(lib().hx_kernel_stats(self.h, kind, C.byref(l), C.byref(u), C.byref(ms), int(reset)))
        return {"launches": l.value, "units": u.value, "ms": ms.value}


class Index:
    """hx_index: host-side HNSW graph (graph/mod.rs + build.rs + scan.rs control flow) over an Engine."""

    def __init__(self, engine, m=16, ef_construction=64):
        self.engine, self.m, self.efc = engine, m, ef_construction
        h = C.c_void_p()
        rc = lib().hx_index_create(engine.h, m, ef_construction, C.byref(h))
        if rc:
            raise HxError(rc, lib().hx_last_error(engine.h).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().hx_index_destroy(self.h)
            self.h = None

    __del__ = close

    def _ck(self, rc):
        if rc:
            raise HxError(rc, lib().hx_index_last_error(self.h).decode())

    def set_threads(self, n):
        self._ck(lib().hx_index_set_threads(self.h, n))

    def insert(self, first_row, levels, tids=None, batch=1):
        levels = np.ascontiguousarray(levels, np.int32)
        n = len(levels)
        tids = np.arange(first_row, first_row + n, dtype=np.int64) if tids is None else np.ascontiguousarray(tids, np.int64)
        out = np.empty(n, np.uint32)
        self._ck(lib().hx_index_insert(self.h, first_row, n, _p(levels), _p(tids), batch, _p(out)))
        return out

    def insert_ondisk(self, first_row, levels, tids=None, batch=1):
        """aminsert (insert.rs:1227-1480) for rows already appended to the engine."""
        levels = np.ascontiguousarray(levels, np.int32)
        n = len(levels)
        tids = np.arange(first_row, first_row + n, dtype=np.int64) if tids is None else np.ascontiguousarray(tids, np.int64)
        out = np.empty(n, np.uint32)
        self._ck(lib().hx_index_insert_ondisk(self.h, first_row, n, _p(levels), _p(tids), batch, _p(out)))
        return out

    def vacuum(self, dead_tids, batch=1):
        """ambulkdelete + amvacuumcleanup (vacuum.rs); returns (elements marked deleted, elements repaired)."""
        d = np.ascontiguousarray(dead_tids, np.int64)
        nd, nr = C.c_uint64(0), C.c_uint64(0)
        self._ck(lib().hx_index_vacuum(self.h, _p(d), len(d), batch, C.byref(nd), C.byref(nr)))
        return nd.value, nr.value

    def deleted(self, e):
        return lib().hx_index_deleted(self.h, e)

    def invalidate(self, blkno, offno, versions=None):
        """Drops loaded elements whose tuple version changed (scan.rs:262-265); returns how many were dropped."""
        b = np.ascontiguousarray(blkno, np.uint32)
        o = np.ascontiguousarray(offno, np.uint16)
        v = None if versions is None else np.ascontiguousarray(versions, np.uint8)
        nd = C.c_uint32(0)
        self._ck(lib().hx_index_invalidate(self.h, len(b), _p(b), _p(o), _p(v), C.byref(nd)))
        return nd.value

    # ---- staged batch (multi-GPU build; see dist_build.py) ----
    def batch_begin(self, first_row, levels, tids):
        levels = np.ascontiguousarray(levels, np.int32)
        tids = np.ascontiguousarray(tids, np.int64)
        self._ck(lib().hx_index_batch_begin(self.h, first_row, len(levels), _p(levels), _p(tids)))

    def batch_search(self, lo, hi):
        self._ck(lib().hx_index_batch_search(self.h, lo, hi))

    def batch_new_bytes(self, lo, hi):
        return lib().hx_index_batch_new_bytes(self.h, lo, hi)

    def batch_export_new(self, lo, hi):
        buf = np.empty(self.batch_new_bytes(lo, hi), np.uint8)
        self._ck(lib().hx_index_batch_export_new(self.h, lo, hi, _p(buf)))
        return buf

    def batch_import_new(self, lo, hi, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        assert buf.nbytes == self.batch_new_bytes(lo, hi)
        self._ck(lib().hx_index_batch_import_new(self.h, lo, hi, _p(buf)))

    def batch_links(self, rank, world):
        self._ck(lib().hx_index_batch_links(self.h, rank, world))

    def batch_links_bytes(self):
        return lib().hx_index_batch_links_bytes(self.h)

    def batch_export_links(self):
        buf = np.empty(self.batch_links_bytes(), np.uint8)
        self._ck(lib().hx_index_batch_export_links(self.h, _p(buf)))
        return buf

    def batch_import_links(self, buf):
        buf = np.ascontiguousarray(buf, np.uint8)
        self._ck(lib().hx_index_batch_import_links(self.h, _p(buf), buf.nbytes))

    def batch_end(self, n):
        out = np.empty(n, np.uint32)
        self._ck(lib().hx_index_batch_end(self.h, _p(out)))
        return out

    # ---- device-resident staged batch: the buffers are DEVICE pointers (ints), e.g. torch tensors' data_ptr() ----
    def dbatch_supported(self, levels):
        levels = np.ascontiguousarray(levels, np.int32)
        return bool(lib().hx_index_dbatch_supported(self.h, _p(levels), len(levels)))

    @property
    def dbatch_record_bytes(self):
        return lib().hx_index_dbatch_record_bytes(self.h)

    @property
    def dbatch_list_record_bytes(self):
        return lib().hx_index_dbatch_list_record_bytes(self.h)

    def dbatch_begin(self, first_row, levels, tids):
        levels = np.ascontiguousarray(levels, np.int32)
        tids = np.ascontiguousarray(tids, np.int64)
        self._ck(lib().hx_index_dbatch_begin(self.h, first_row, len(levels), _p(levels), _p(tids)))

    def dbatch_search(self, lo, hi, d_records):
        self._ck(lib().hx_index_dbatch_search(self.h, lo, hi, C.c_void_p(d_records)))

    def dbatch_links(self, rank, world, d_records):
        n = C.c_uint64(0)
        self._ck(lib().hx_index_dbatch_links(self.h, rank, world, C.c_void_p(d_records), C.byref(n)))
        return n.value

    def dbatch_export_links(self, d_out):
        self._ck(lib().hx_index_dbatch_export_links(self.h, C.c_void_p(d_out)))

    def dbatch_import_links(self, d_list_records, n):
        self._ck(lib().hx_index_dbatch_import_links(self.h, C.c_void_p(d_list_records), n))

    @property
    def dbatch_wtab_bytes(self):
        """Bytes per member of the W-table exchange of the open device batch (0: tables off)."""
        return lib().hx_index_dbatch_wtab_bytes(self.h)

    def dbatch_export_wtabs(self, lo, hi, d_out):
        self._ck(lib().hx_index_dbatch_export_wtabs(self.h, lo, hi, C.c_void_p(d_out)))

    def dbatch_import_wtabs(self, lo, hi, d_in):
        self._ck(lib().hx_index_dbatch_import_wtabs(self.h, lo, hi, C.c_void_p(d_in)))

    def dbatch_end(self, n):
        out = np.empty(n, np.uint32)
        self._ck(lib().hx_index_dbatch_end(self.h, _p(out)))
        return out

    @property
    def size(self):
        return lib().hx_index_size(self.h)

    @property
    def entry(self):
        return lib().hx_index_entry(self.h)

    def level(self, e):
        return lib().hx_index_level(self.h, e)

    def neighbors(self, e, layer):
        ids = np.empty(2 * self.m, np.uint32)
        d = np.empty(2 * self.m, np.float32)
        n = lib().hx_index_neighbors(self.h, e, layer, _p(ids), _p(d))
        if n < 0:
            return None, None
        return ids[:n].copy(), d[:n].copy()

    def heaptids(self, e):
        t = np.empty(10, np.int64)
        n = lib().hx_index_heaptids(self.h, e, _p(t))
        return t[:n].tolist()

    def export_levels(self, first=0, n=None):
        n = self.size - first if n is None else n
        out = np.empty(n, np.int32)
        self._ck(lib().hx_index_export_levels(self.h, first, n, _p(out)))
        return out

    def export_layer(self, layer, first=0, n=None, with_dist=True):
        n = self.size - first if n is None else n
        lm = 2 * self.m if layer == 0 else self.m
        ids = np.zeros((n, lm), np.uint32)
        d = np.zeros((n, lm), np.float32) if with_dist else None
        cnt = np.zeros(n, np.uint16)
        self._ck(lib().hx_index_export_layer(self.h, layer, first, n, _p(ids), _p(d), _p(cnt)))
        return ids, d, cnt

    def set_neighbors(self, e, layer, ids, dist):
        ids, dist = _u32(ids), np.ascontiguousarray(dist, np.float32)
        self._ck(lib().hx_index_set_neighbors(self.h, e, layer, len(ids), _p(ids), _p(dist)))

    def counters(self):
        c = np.zeros(8, np.uint64)
        self._ck(lib().hx_index_counters(self.h, _p(c)))
        return c

    def profile(self, reset=False):
        p = np.zeros(16, np.float64)
        self._ck(lib().hx_index_profile(self.h, _p(p), int(reset)))
        return {"advance_s": p[0], "compact_s": p[1], "fill_s": p[2], "round_s": p[3], "rounds": int(p[5]), "fused_s": p[6],
                "mirror_sync_s": p[7], "links_setup_s": p[8], "links_lockstep_s": p[9], "insert_total_s": p[10],
                "batch_search_s": p[11], "batch_begin_s": p[12], "links_max_chain": int(p[13]), "links_ops": int(p[14])}

    def serialize_pages(self):
        """hx_index_serialize_pages: (pages uint8 [n_pages, 8192], elem_blkno[n], elem_offno[n])."""
        n = self.size
        npg = C.c_uint64()
        self._ck(lib().hx_index_serialize_pages(self.h, None, 0, C.byref(npg), None, None))
        pages = np.zeros((npg.value, 8192), np.uint8)
        blk = np.zeros(max(n, 1), np.uint32)
        off = np.zeros(max(n, 1), np.uint16)
        self._ck(lib().hx_index_serialize_pages(self.h, _p(pages), npg.value, C.byref(npg), _p(blk), _p(off)))
        return pages, blk[:n], off[:n]

    def load_pages(self, pages):
        """hx_index_load_pages: fills the (empty) index and engine from a page image; returns (elem_blkno, elem_offno)."""
        pages = np.ascontiguousarray(pages, np.uint8).reshape(-1, 8192)
        cap = len(pages) * 400          # an 8 KB page holds fewer than 400 element tuples
        blk = np.zeros(max(cap, 1), np.uint32)
        off = np.zeros(max(cap, 1), np.uint16)
        n = C.c_uint64()
        self._ck(lib().hx_index_load_pages(self.h, _p(pages), len(pages), _p(blk), _p(off), cap, C.byref(n)))
        return blk[:n.value], off[:n.value]

    def set_fused(self, on):
        self._ck(lib().hx_index_set_fused(self.h, int(on)))

    def set_mfma(self, on):
        self._ck(lib().hx_index_set_mfma(self.h, int(on)))

    def mfma_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(lib().hx_index_mfma_stats(self.h, C.byref(a), C.byref(b)))
        return {"mfma_pairs": a.value, "exact_pairs": b.value}

    def fused_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(lib().hx_index_fused_stats(self.h, C.byref(a), C.byref(b)))
        return {"tasks": a.value, "redone": b.value & 0xFFFFFFFF, "max_candidate_heap": b.value >> 32}

    @staticmethod
    def _pad(cnt, k, *arrays_and_fills):
        """The ABI defines the first counts_out[q] entries of a row; the rest is set here (-1 / inf / 0), and only for the rows that are short --
        pre-filling 1.6 MB per 10 000-query call cost 3 % of the bench's step time."""
        short = np.nonzero(cnt < k)[0]
        if short.size:
            tail = np.arange(k)[None, :] >= cnt[short, None]
            for a, fill in arrays_and_fills:
                sub = a[short]
                sub[tail] = fill
                a[short] = sub

    def search(self, nq, ef_search, k):
        tids = np.empty((nq, k), np.int64)
        d = np.empty((nq, k), np.float32)
        el = np.empty((nq, k), np.uint32)
        cnt = np.zeros(nq, np.uint32)
        self._ck(lib().hx_index_search(self.h, nq, ef_search, k, _p(tids), _p(d), _p(el), _p(cnt)))
        self._pad(cnt, k, (tids, -1), (d, np.inf), (el, 0))
        return tids, d, el, cnt

    def search_submit(self, slot, first_query, nq, ef_search, k):
        """Pipelined scan: launches the scan of engine query slots [first_query, first_query + nq) on `slot` and returns at once."""
        self._ck(lib().hx_index_search_submit(self.h, slot, first_query, nq, ef_search, k))
        self._scan_shape = getattr(self, "_scan_shape", {})
        self._scan_shape[slot] = (nq, k)

    def search_wait(self, slot):
        """Results of the scan submitted on `slot`: as search()."""
        if slot not in getattr(self, "_scan_shape", {}):
            raise HxError(-6, "nothing submitted on this scan slot")
        nq, k = self._scan_shape.pop(slot)
        tids = np.empty((nq, k), np.int64)
        d = np.empty((nq, k), np.float32)
        el = np.empty((nq, k), np.uint32)
        cnt = np.zeros(nq, np.uint32)
        self._ck(lib().hx_index_search_wait(self.h, slot, _p(tids), _p(d), _p(el), _p(cnt)))
        self._pad(cnt, k, (tids, -1), (d, np.inf), (el, 0))
        return tids, d, el, cnt

    def search_null(self, ef_search, limit, mode=0, max_scan_tuples=20000, filter_pass=None):
        """ORDER BY val <-> NULL (scan.rs:186-187): (tids, elems) of the traversal with every distance 0.0."""
        tids = np.full(limit, -1, np.int64)
        el = np.zeros(limit, np.uint32)
        cnt = C.c_uint32()
        f = None if filter_pass is None else np.ascontiguousarray(filter_pass, np.uint8)
        self._ck(lib().hx_index_search_null(self.h, ef_search, mode, max_scan_tuples, limit, _p(f), 0 if f is None else len(f), _p(tids), _p(el), C.byref(cnt)))
        return tids[:cnt.value], el[:cnt.value]

    def search_iterative(self, nq, ef_search, mode, max_scan_tuples, limit, filter_pass=None):
        tids = np.empty((nq, limit), np.int64)
        d = np.empty((nq, limit), np.float32)
        cnt = np.zeros(nq, np.uint32)
        f = None if filter_pass is None else np.ascontiguousarray(filter_pass, np.uint8)
        self._ck(lib().hx_index_search_iterative(self.h, nq, ef_search, mode, max_scan_tuples, limit, _p(f),
                                                 0 if f is None else len(f), _p(tids), _p(d), _p(cnt)))
        self._pad(cnt, limit, (tids, -1), (d, np.inf))
        return tids, d, cnt
