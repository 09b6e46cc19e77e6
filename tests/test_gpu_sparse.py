"""sparsevec on the engine (include/hnswrx.h: HX_SPARSE): the merge-join distances of src/types/sparsevec.rs:873-950, 1038-1088
through the DistanceFn seam, sparsevec_l2_normalize_raw (:1123-1178) and an HNSW index on them (lock-step driver), all bit-identical to the oracle's
statement-by-statement restatement -- one lane walks one pair in the reference's own f32 accumulation order, so there is no tolerance to state."""
import importlib.util
import os

import numpy as np
import pytest

import pgvector_rx_amd as hx

_spec = importlib.util.spec_from_file_location("orc", os.path.join(os.path.dirname(__file__), "..", "oracle", "orc.py"))
orc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(orc)

METRICS = [hx.L2SQ, hx.NEG_IP, hx.L1]


def random_sparse(rng, n, dim, max_nnz):
    rows = []
    for _ in range(n):
        k = int(rng.integers(0, min(max_nnz, dim) + 1))
        idx = np.sort(rng.choice(dim, k, replace=False)).astype(np.int32)
        val = rng.standard_normal(k).astype(np.float32)
        val[val == 0] = 1.0
        rows.append((idx, val))
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("metric", METRICS)
@pytest.mark.parametrize("dim,max_nnz,n", [(3, 3, 60), (50, 12, 300), (100000, 40, 500), (1000000000, 1000, 70)])
def test_sparse_distances_bit_exact(metric, dim, max_nnz, n):
    rng = np.random.default_rng(dim % 1000 + max_nnz + metric)
    rows = random_sparse(rng, n, dim, max_nnz)
    rows[3] = (np.zeros(0, np.int32), np.zeros(0, np.float32))                    # '{}/dim'
    if max_nnz == 1000:
        rows[5] = (np.sort(rng.choice(dim, 1000, replace=False)).astype(np.int32), rng.standard_normal(1000).astype(np.float32))   # the index's maximum
    rec = hx.pack_sparse(dim, rows)
    assert rec.shape[1] == hx.sparse_record_bytes(dim) == orc.lib().orc_row_bytes(orc.SPARSE, dim)
    e = hx.Engine(hx.SPARSE, metric, dim, n)
    e.append(rec)
    assert np.array_equal(e.read_rows(0, n), rec)
    ids = rng.integers(0, n, 200).astype(np.uint32)
    q = hx.pack_sparse(dim, random_sparse(rng, 1, dim, max_nnz))
    got = e.distances(q[0], ids)
    want = orc.distances_many(orc.SPARSE, metric, dim, q[0], rec, ids.astype(np.int32)).astype(np.float32)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # query = a stored row, several groups in one launch
    gq = rng.integers(0, n, 9).astype(np.uint32)
    sizes = rng.integers(0, 70, 9)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    gids = rng.integers(0, n, int(off[-1])).astype(np.uint32)
    got = e.distances_batch(gq, off, gids)
    for g in range(9):
        w = orc.distances_many(orc.SPARSE, metric, dim, rec[gq[g]], rec, gids[off[g]:off[g + 1]].astype(np.int32)).astype(np.float32)
        assert (got[off[g]:off[g + 1]].view(np.uint32) == w.view(np.uint32)).all()
    # pair blocks: triangle and rectangle
    w = 41
    pid = rng.choice(n, w, replace=False).astype(np.uint32)
    P = e.pairwise(pid)
    O = orc.pairwise(orc.SPARSE, metric, dim, rec, pid.astype(np.int32)).astype(np.float32)
    iu = np.tril_indices(w, -1)
    assert (P[iu].view(np.uint32) == O[iu].view(np.uint32)).all()
    assert (P.T[iu].view(np.uint32) == O[iu].view(np.uint32)).all()               # symmetric bits: the merged index order does not depend on the side
    A, B = pid[:17], pid[17:40]
    tri, rect = e.pairwise_many([(A, None), (A, B)])
    assert (tri.view(np.uint32) == O[:17, :17][np.tril_indices(17, -1)].view(np.uint32)).all()
    assert (rect.view(np.uint32) == O[:17, 17:40].view(np.uint32)).all()
    e.close()


@pytest.mark.gpu
def test_sparse_normalize_bit_exact():
    rng = np.random.default_rng(4)
    dim, n = 2000, 120
    rows = random_sparse(rng, n, dim, 60)
    rows[0] = (np.zeros(0, np.int32), np.zeros(0, np.float32))
    rows[1] = (np.array([3, 9], np.int32), np.array([1e-30, 5.0], np.float32))      # 1e-30 / 5 underflows to a subnormal, not to zero
    rows[2] = (np.array([1, 2, 7], np.int32), np.array([1e-45, 3.0, 4.0], np.float32))   # the first value becomes 0.0 and is dropped (sparsevec.rs:1149-1175)
    rec = hx.pack_sparse(dim, rows)
    e = hx.Engine(hx.SPARSE, hx.NEG_IP, dim, n)
    e.append(rec)
    norms = e.normalize_rows(0, n)
    got = e.read_rows(0, n)
    for i in range(n):
        want, norm = orc.l2_normalize(orc.SPARSE, dim, rec[i])
        assert norms[i] == norm
        assert np.array_equal(got[i], want), i
    assert int(got[2, 0:4].view(np.int32)[0]) == 2
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [True, False], ids=["traversal-kernel", "lock-step"])
@pytest.mark.parametrize("metric,cosine", [(hx.L2SQ, False), (hx.NEG_IP, False), (hx.L1, False), (hx.NEG_IP, True)])
def test_sparse_index_equals_oracle(metric, cosine, fused):
    """Build + scan of an HNSW index on sparsevec rows (sparsevec_l2_ops / _ip_ops / _l1_ops / _cosine_ops, sparsevec.rs:1552-1582): graph and
    top-k identical to the oracle's; the reference's own 4-row orderings (tests/pg_regress/expected/hnsw_sparsevec.out) are pinned on the oracle
    in tests/test_oracle_golden.py.  Round 3: the searches of the build and the scans run in the traversal kernel (hx_fused_sparse.hip: one lane per
    row walks the merge join), select_neighbors and the back-links in k_select_w / k_list_ops (hx_biglist.hip); `lock-step` is the round-2 placement on the merge-join kernels."""
    rng = np.random.default_rng(11 + metric + cosine)
    dim, n, m, efc = 400, 600, 8, 40
    rows = random_sparse(rng, n, dim, 30)
    rows[17] = rows[5]                                                              # a duplicate: joins the first one's heap-TID list
    rec = hx.pack_sparse(dim, rows)
    e = hx.Engine(hx.SPARSE, metric, dim, n)
    o = orc.Index(orc.SPARSE, metric, dim, m=m, ef_construction=efc)
    keep = np.ones(n, bool)
    if cosine:                                                                      # build.rs:417-438: normalise, skip zero-norm rows
        nrec = np.zeros_like(rec)
        for i in range(n):
            nrec[i], norm = orc.l2_normalize(orc.SPARSE, dim, rec[i])
            keep[i] = norm != 0.0
        e.append(rec)
        norms = e.normalize_rows(0, n)
        assert np.array_equal(e.read_rows(0, n), nrec) and np.array_equal(norms != 0.0, keep)
        e.pop(n)
        rec = np.ascontiguousarray(nrec[keep])
    e.append(rec)
    nk = len(rec)
    levels = hx.draw_levels(nk, m, seed=3)
    tids = np.arange(nk, dtype=np.int64)
    ix = hx.Index(e, m, efc)
    ix.set_fused(fused)
    elem = ix.insert(0, levels, tids=tids, batch=1)
    oelem = np.concatenate([o.insert_batch(rec[i:i + 1], levels[i:i + 1], tids[i:i + 1]) for i in range(nk)])   # batch of one = the reference's schedule
    assert elem.tolist() == oelem.tolist()
    assert ix.size == o.size == nk and ix.entry == o.entry
    for i in range(nk):
        if o.merged(i):
            assert ix.level(i) < 0
            continue
        for layer in range(ix.level(i) + 1):
            gi, gd = ix.neighbors(i, layer)
            oi, od = o.neighbors(i, layer)
            assert gi.tolist() == oi.tolist(), (i, layer)
            assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (i, layer)
    qs = hx.pack_sparse(dim, random_sparse(rng, 12, dim, 30))
    if cosine:
        for i in range(12):
            qs[i], _ = orc.l2_normalize(orc.SPARSE, dim, qs[i])
    e.set_queries(qs)
    t, d, el, cnt = ix.search(12, 40, 10)
    for q in range(12):
        res = o.scan(qs[q], ef_search=40, limit=10)
        assert t[q, :cnt[q]].tolist() == [x for x, _, _ in res]
        assert (d[q, :cnt[q]].view(np.uint32) == np.array([y for _, y, _ in res], np.float32).view(np.uint32)).all()
    st = ix.fused_stats()
    assert (st["tasks"] >= nk and st["redone"] == 0) if fused else st["tasks"] == 0    # the searches were device tasks / the lock-step driver on the merge-join kernels
    if fused:
        assert ix.profile()["rounds"] == 0                                          # select_neighbors and the back-links too (hx_biglist.hip: k_select_w, k_list_ops)
    # iterative scans (relaxed and strict order) on the same placement
    for mode in (1, 2):
        ti, di, ci = ix.search_iterative(12, 10, mode, 200, 15)
        for q in range(12):
            want = o.scan(qs[q], ef_search=10, iterative=orc.ITER_RELAXED if mode == 1 else orc.ITER_STRICT, max_scan_tuples=200)[:15]
            assert ti[q, :ci[q]].tolist() == [x for x, _, _ in want], (mode, q)
    with pytest.raises(hx.HxError):
        ix.serialize_pages()
    ix.close()
    e.close()


@pytest.mark.gpu
def test_sparse_create_limits():
    with pytest.raises(hx.HxError):
        hx.Engine(hx.SPARSE, hx.HAMMING, 10, 4)
    with pytest.raises(hx.HxError):
        hx.Engine(hx.SPARSE, hx.L2SQ, 1000000001, 4)                               # SPARSEVEC_MAX_DIM
    e = hx.Engine(hx.SPARSE, hx.L2SQ, 1000000000, 4)
    assert e.row_bytes == 16 + 8 * 1000
    e.close()


@pytest.mark.gpu
def test_sparse_aminsert_and_vacuum_with_lists_of_more_than_64_slots():
    """sparsevec at m = 40 through aminsert (k_fused<OpSparse> MODE 3 + k_update_runs_big<OpSparse>), a VACUUM and more inserts: the oracle's graph after each step."""
    rng = np.random.default_rng(40)
    dim, n, m, efc = 300, 260, 40, 80
    rows = random_sparse(rng, n, dim, 20)
    rec = hx.pack_sparse(dim, rows)
    levels = hx.draw_levels(n, m, seed=4)
    e = hx.Engine(hx.SPARSE, hx.L2SQ, dim, n)
    e.append(rec)
    ix = hx.Index(e, m, efc)
    o = orc.Index(orc.SPARSE, orc.L2SQ, dim, m=m, ef_construction=efc)
    o.set_ondisk_tombstones(True)
    tids = np.arange(1, n + 1, dtype=np.int64)

    def same(size):
        assert ix.size == o.size == size and ix.entry == o.entry
        for i in range(size):
            if o.merged(i):
                assert ix.level(i) < 0
                continue
            assert ix.deleted(i) == o.deleted(i)
            for layer in range(ix.level(i) + 1):
                gi, gd = ix.neighbors(i, layer)
                oi, od = o.neighbors(i, layer)
                assert gi.tolist() == oi.tolist(), (i, layer)
                assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (i, layer)

    ix.insert_ondisk(0, levels[:200], tids=tids[:200], batch=1)
    for i in range(200):
        o.insert_on_disk(rec[i], int(levels[i]), int(tids[i]))
    same(200)
    assert max(len(ix.neighbors(i, 0)[0]) for i in range(200)) == 2 * m
    kill = tids[:200][::4].copy()
    ix.vacuum(kill, batch=1)
    o.vacuum(kill)
    same(200)
    ix.insert_ondisk(200, levels[200:], tids=tids[200:], batch=1)
    for i in range(200, n):
        o.insert_on_disk(rec[i], int(levels[i]), int(tids[i]))
    same(n)
    assert ix.profile()["rounds"] <= 2 * n                                          # one kernel per insert for the back-connections, no lock-step expansion rounds
    ix.close()
    e.close()
