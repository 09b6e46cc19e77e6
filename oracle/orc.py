"""ctypes binding of the CPU oracle (oracle/hnsw_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORC_LIB") or os.path.join(HERE, "_build", "liborc.so")   # ORC_LIB: e.g. an ASan/UBSan build of the oracle

F32, F16, BIT, SPARSE = 0, 1, 2, 3   # SPARSE rows are the engine's fixed-size sparsevec records (uint8; pgvector-rx_amd/binding.py: pack_sparse)
L2SQ, NEG_IP, L1, HAMMING, JACCARD = 0, 1, 2, 3, 4
SEQ, W64, VEC = 0, 1, 2     # VEC: reassociated + compiler-vectorised CPU variant (bench baseline only, not the reference's arithmetic)
ITER_OFF, ITER_RELAXED, ITER_STRICT = 0, 1, 2

_lib = None


def build(force=False):
    src = os.path.join(HERE, "hnsw_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, f64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_float
        sig = {
            "orc_half_to_f32": (f32, [C.c_uint16]),
            "orc_f32_to_half": (C.c_uint16, [f32]),
            "orc_hamming": (C.c_uint64, [vp, vp, i32]),
            "orc_jaccard": (f64, [vp, vp, i32]),
            "orc_distance": (f64, [i32, i32, i32, vp, vp, i32]),
            "orc_l2_distance": (f64, [i32, i32, vp, vp]),
            "orc_inner_product": (f64, [i32, i32, vp, vp]),
            "orc_cosine_distance": (f64, [i32, i32, vp, vp]),
            "orc_norm": (f64, [i32, i32, vp]),
            "orc_l2_normalize": (f64, [i32, i32, vp, vp]),
            "orc_max_level": (i32, [i32]),
            "orc_level_from_uniform": (i32, [f64, i32]),
            "orc_row_bytes": (C.c_size_t, [i32, i32]),
            "orc_index_new": (vp, [i32] * 6),
            "orc_index_free": (None, [vp]),
            "orc_index_insert": (i32, [vp, vp, i32, i64]),
            "orc_index_insert_batch": (None, [vp, vp, vp, vp, i32, vp]),
            "orc_index_size": (i32, [vp]),
            "orc_index_entry": (i32, [vp]),
            "orc_index_level": (i32, [vp, i32]),
            "orc_index_merged": (i32, [vp, i32]),
            "orc_index_ntids": (i32, [vp, i32]),
            "orc_index_tid": (i64, [vp, i32, i32]),
            "orc_index_counter": (C.c_uint64, [vp, i32]),
            "orc_index_reset_counters": (None, [vp]),
            "orc_index_neighbors": (i32, [vp, i32, i32, vp, vp]),
            "orc_index_load": (None, [vp, vp, i32, vp, i32]),
            "orc_index_set_layer": (None, [vp, i32, vp, vp, vp]),
            "orc_index_add_raw": (i32, [vp, vp, i32]),
            "orc_index_link_raw": (None, [vp, i32, i32, i32, f32]),
            "orc_search_layer_raw": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
            "orc_select_neighbors_raw": (i32, [vp, vp, vp, i32, i32, vp]),
            "orc_find_element_neighbors_raw": (None, [vp, i32, i32]),
            "orc_update_neighbor_connections_raw": (None, [vp, i32]),
            "orc_scan_begin": (vp, [vp, vp, i32, i32, i64]),
            "orc_scan_end": (None, [vp]),
            "orc_scan_next": (i32, [vp, vp, vp, vp]),
            "orc_search_topk": (i32, [vp, vp, i32, i32, vp, vp]),
            "orc_search_many": (None, [vp, vp, i32, i32, i32, i32, vp, vp]),
            "orc_index_write_pages": (C.c_uint64, [vp, vp, C.c_uint64, vp, vp]),
            "orc_bruteforce_topk": (i32, [vp, vp, i32, vp, vp]),
            "orc_distances_many": (None, [i32, i32, i32, vp, vp, vp, i32, i32, vp]),
            "orc_pairwise": (None, [i32, i32, i32, vp, vp, i32, i32, vp]),
            "orc_index_insert_on_disk": (i32, [vp, vp, i32, i64]),
            "orc_index_repair_element": (None, [vp, i32, vp]),
            "orc_index_mark_deleted": (None, [vp, i32, i32]),
            "orc_index_clear_tids": (None, [vp, i32]),
            "orc_index_vacuum": (None, [vp, vp, i32]),
            "orc_index_deleted": (i32, [vp, i32]),
            "orc_index_set_ondisk_tombstones": (None, [vp, i32]),
            "orc_acc_w64_plain": (f32, [i32, i32, i32, vp, vp]),
            "orc_acc_w64_fast": (f32, [i32, i32, i32, vp, vp]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


_NP = {F32: np.float32, F16: np.uint16, BIT: np.uint8, SPARSE: np.uint8}


def as_rows(dtype, a):
    a = np.ascontiguousarray(a, dtype=_NP[dtype])
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def distance(dtype, metric, dim, a, b, order=SEQ):
    a, b = as_rows(dtype, a), as_rows(dtype, b)
    return lib().orc_distance(dtype, metric, dim, _p(a), _p(b), order)


def distances_many(dtype, metric, dim, q, rows, ids=None, order=SEQ):
    q, rows = as_rows(dtype, q), as_rows(dtype, rows)
    n = len(ids) if ids is not None else rows.shape[0]
    out = np.empty(n, np.float64)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, np.int32)
        idp = _p(ids)
    lib().orc_distances_many(dtype, metric, dim, _p(q), _p(rows), idp, n, order, _p(out))
    return out


def pairwise(dtype, metric, dim, rows, ids, order=SEQ):
    rows = as_rows(dtype, rows)
    ids = np.ascontiguousarray(ids, np.int32)
    w = len(ids)
    out = np.empty((w, w), np.float64)
    lib().orc_pairwise(dtype, metric, dim, _p(rows), _p(ids), w, order, _p(out))
    return out


def l2_normalize(dtype, dim, a):
    a = as_rows(dtype, a)
    out = np.zeros_like(a)
    norm = lib().orc_l2_normalize(dtype, dim, _p(a), _p(out))
    return out, norm


def pack_sparse(dim, idx, val):
    """One sparsevec as the fixed-size record of ORC_SPARSE rows: {int32 nnz; int32 pad[3]; int32 index[cap]; float32 value[cap]}, cap = min(dim, 1000)."""
    cap = min(dim, 1000)
    rb = (16 + 8 * cap + 15) & ~15
    idx, val = np.asarray(idx, np.int32), np.asarray(val, np.float32)
    out = np.zeros(rb, np.uint8)
    out[0:4] = np.array([len(idx)], np.int32).view(np.uint8)
    out[16:16 + 4 * len(idx)] = idx.view(np.uint8)
    out[16 + 4 * cap:16 + 4 * cap + 4 * len(val)] = val.view(np.uint8)
    return out


def sparse_from_dense(v):
    """A dense literal -> the sparsevec the reference's vector::sparsevec cast stores (non-zero elements only)."""
    v = np.asarray(v, np.float32)
    nz = np.nonzero(v)[0]
    return pack_sparse(len(v), nz, v[nz])


def pack_bits(bitstring):
    """'10100000' -> PG VarBit payload bytes (MSB first, zero padded): bitvec.rs:28-37."""
    bits = np.array([c == "1" for c in bitstring], dtype=np.uint8)
    return np.packbits(bits, bitorder="big")


class Index:
    """Thin handle on orc_index (in-memory HNSW built with the reference's control flow)."""

    def __init__(self, dtype, metric, dim, m=16, ef_construction=64, order=SEQ):
        self.dtype, self.metric, self.dim, self.m, self.efc, self.order = dtype, metric, dim, m, ef_construction, order
        self.h = lib().orc_index_new(dtype, metric, dim, m, ef_construction, order)
        self.row_bytes = lib().orc_row_bytes(dtype, dim)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_free(self.h)
            self.h = None

    def insert(self, row, level, tid):
        row = as_rows(self.dtype, row)
        assert row.nbytes == self.row_bytes
        return lib().orc_index_insert(self.h, _p(row), int(level), int(tid))

    def set_ondisk_tombstones(self, on=True):
        lib().orc_index_set_ondisk_tombstones(self.h, int(on))

    def insert_on_disk(self, row, level, tid):
        """aminsert (insert.rs:1227-1480) on the in-memory mirror."""
        row = as_rows(self.dtype, row)
        assert row.nbytes == self.row_bytes
        return lib().orc_index_insert_on_disk(self.h, _p(row), int(level), int(tid))

    def repair_element(self, e, deleted=None):
        """repair_graph_element (vacuum.rs:288-407): deleted = uint8 flags per element (the vacuum's deleted set)."""
        d = None if deleted is None else np.ascontiguousarray(deleted, np.uint8)
        lib().orc_index_repair_element(self.h, int(e), None if d is None else _p(d))

    def vacuum(self, dead_tids):
        """ambulkdelete + amvacuumcleanup (vacuum.rs): dead_tids = heap TIDs reported dead."""
        d = np.ascontiguousarray(dead_tids, np.int64)
        lib().orc_index_vacuum(self.h, _p(d), len(d))

    def deleted(self, e):
        return lib().orc_index_deleted(self.h, int(e))

    def mark_deleted(self, e, flag=1):
        lib().orc_index_mark_deleted(self.h, int(e), int(flag))

    def clear_tids(self, e):
        lib().orc_index_clear_tids(self.h, int(e))

    def insert_batch(self, rows, levels, tids):
        rows = as_rows(self.dtype, rows)
        levels = np.ascontiguousarray(levels, np.int32)
        tids = np.ascontiguousarray(tids, np.int64)
        n = len(levels)
        assert rows.nbytes == n * self.row_bytes
        out = np.empty(n, np.int32)
        lib().orc_index_insert_batch(self.h, _p(rows), _p(levels), _p(tids), n, _p(out))
        return out

    def build(self, rows, levels, batch=1, tids=None):
        """batch==1: the reference's sequential schedule; otherwise fixed-size snapshot batches
        following `schedule` if batch is a list of batch sizes."""
        rows = as_rows(self.dtype, rows)
        n = len(levels)
        tids = np.arange(n, dtype=np.int64) if tids is None else np.asarray(tids, np.int64)
        if batch == 1:
            return np.array([self.insert(rows[i], levels[i], tids[i]) for i in range(n)], np.int32)
        sizes = batch if isinstance(batch, (list, tuple, np.ndarray)) else None
        out, i, k = [], 0, 0
        while i < n:
            b = int(sizes[k]) if sizes is not None else int(batch)
            k += 1
            out.append(self.insert_batch(rows[i:i + b], levels[i:i + b], tids[i:i + b]))
            i += b
        return np.concatenate(out)

    @property
    def size(self):
        return lib().orc_index_size(self.h)

    @property
    def entry(self):
        return lib().orc_index_entry(self.h)

    def level(self, i):
        return lib().orc_index_level(self.h, i)

    def merged(self, i):
        return lib().orc_index_merged(self.h, i)

    def tids(self, i):
        return [lib().orc_index_tid(self.h, i, k) for k in range(lib().orc_index_ntids(self.h, i))]

    def counters(self):
        return [lib().orc_index_counter(self.h, k) for k in range(5)]

    def neighbors(self, i, layer):
        cap = 2 * self.m
        ids = np.empty(cap, np.int32)
        d = np.empty(cap, np.float32)
        n = lib().orc_index_neighbors(self.h, i, layer, _p(ids), _p(d))
        if n < 0:
            return None, None
        return ids[:n].copy(), d[:n].copy()

    def graph(self):
        """[(level, [ (ids, dists) per layer ])] for every element."""
        g = []
        for i in range(self.size):
            lv = self.level(i)
            g.append((lv, [self.neighbors(i, l) for l in range(lv + 1)]))
        return g

    def load(self, rows, levels, entry, layers):
        """layers: list over layer of (ids [n][lm] uint32, dist [n][lm] float32 or None, cnt [n] uint16)."""
        rows = as_rows(self.dtype, rows)
        levels = np.ascontiguousarray(levels, np.int32)
        lib().orc_index_load(self.h, _p(rows), len(levels), _p(levels), int(entry))
        for layer, (ids, d, cnt) in enumerate(layers):
            ids = np.ascontiguousarray(ids, np.uint32)
            cnt = np.ascontiguousarray(cnt, np.uint16)
            dp = None if d is None else _p(np.ascontiguousarray(d, np.float32))
            lib().orc_index_set_layer(self.h, layer, _p(ids), dp, _p(cnt))

    # raw hooks mirroring the reference's pure-Rust unit tests
    def add_raw(self, row, level):
        row = as_rows(self.dtype, row)
        return lib().orc_index_add_raw(self.h, _p(row), level)

    def link_raw(self, i, layer, j, d):
        lib().orc_index_link_raw(self.h, i, layer, j, d)

    def search_layer_raw(self, q, ep, ef, layer):
        q = as_rows(self.dtype, q)
        ep = np.ascontiguousarray(ep, np.int32)
        ids = np.empty(ef + len(ep) + 1, np.int32)
        d = np.empty(ef + len(ep) + 1, np.float32)
        n = lib().orc_search_layer_raw(self.h, _p(q), _p(ep), len(ep), ef, layer, _p(ids), _p(d))
        return ids[:n].copy(), d[:n].copy()

    def select_neighbors_raw(self, cand_idx, cand_dist, maxn):
        ci = np.ascontiguousarray(cand_idx, np.int32)
        cd = np.ascontiguousarray(cand_dist, np.float32)
        out = np.empty(len(ci) + maxn, np.int32)
        n = lib().orc_select_neighbors_raw(self.h, _p(ci), _p(cd), len(ci), maxn, _p(out))
        return out[:n].copy()

    def find_element_neighbors_raw(self, new_idx, entry_idx):
        lib().orc_find_element_neighbors_raw(self.h, new_idx, entry_idx)

    def update_neighbor_connections_raw(self, new_idx):
        lib().orc_update_neighbor_connections_raw(self.h, new_idx)

    def scan(self, query, ef_search=40, iterative=ITER_OFF, max_scan_tuples=20000, limit=None):
        """Yield (tid, distance, element) like successive amgettuple calls."""
        qp = None
        if query is not None:
            q = as_rows(self.dtype, query)
            qp = _p(q)
        s = lib().orc_scan_begin(self.h, qp, ef_search, iterative, max_scan_tuples)
        out = []
        tid, d, e = C.c_int64(), C.c_double(), C.c_int()
        while (limit is None or len(out) < limit) and lib().orc_scan_next(s, C.byref(tid), C.byref(d), C.byref(e)):
            out.append((tid.value, d.value, e.value))
        lib().orc_scan_end(s)
        return out

    def search_topk(self, query, ef_search, k):
        q = as_rows(self.dtype, query)
        ids = np.empty(k, np.int32)
        d = np.empty(k, np.float64)
        n = lib().orc_search_topk(self.h, _p(q), ef_search, k, _p(ids), _p(d))
        return ids[:n].copy(), d[:n].copy()

    def search_many(self, queries, ef_search, k, n_threads=1):
        """The same scan for many queries on n_threads host threads (one query per thread at a time)."""
        q = as_rows(self.dtype, queries)
        nq = q.shape[0] if q.ndim > 1 else 1
        ids = np.full((nq, k), -1, np.int32)
        cnt = np.zeros(nq, np.int32)
        lib().orc_search_many(self.h, _p(q), nq, ef_search, k, int(n_threads), _p(ids), _p(cnt))
        return ids, cnt

    def write_pages(self, cap_pages=None):
        """The index as PostgreSQL HNSW pages (build.rs:545-821): (pages uint8 [n_pages, 8192], blkno[n], offno[n])."""
        n = self.size
        if cap_pages is None:
            cap_pages = 2 + 2 * n + 2
        pages = np.zeros((cap_pages, 8192), np.uint8)
        blk = np.zeros(max(n, 1), np.uint32)
        off = np.zeros(max(n, 1), np.uint16)
        got = lib().orc_index_write_pages(self.h, _p(pages), cap_pages, _p(blk), _p(off))
        if got == 0:
            raise RuntimeError("orc_index_write_pages failed (tuple too large / capacity / add failure)")
        return pages[:got].copy(), blk[:n], off[:n]

    def bruteforce_topk(self, query, k):
        q = as_rows(self.dtype, query)
        ids = np.empty(k, np.int32)
        d = np.empty(k, np.float64)
        n = lib().orc_bruteforce_topk(self.h, _p(q), k, _p(ids), _p(d))
        return ids[:n].copy(), d[:n].copy()


def levels_from_seed(n, m, seed):
    """Explicit level draws shared by oracle and device builds (the reference's rand::random is
    unseeded, build.rs:374): counter-based splitmix64 -> U(0,1) -> build.rs:373-377."""
    with np.errstate(over="ignore"):
        x = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    L = lib()
    return np.array([L.orc_level_from_uniform(float(v), m) for v in u], np.int32)
