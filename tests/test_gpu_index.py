"""GPU parity of the host graph driver (hx_index, lock-step over the device kernels) against the CPU oracle.

The oracle runs the reference's control flow with distances in the device's canonical order
(ORC_ORDER_W64), so graphs, result lists and tie orders must be IDENTICAL, not just close."""
import json
import os

import numpy as np
import pytest

import pgvector_rx_amd as hx
from oracle import orc

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))


def make_rows(dtype, n, dim, rng):
    if dtype == hx.F32:
        return rng.random((n, dim)).astype(np.float32)
    if dtype == hx.F16:
        return rng.random((n, dim)).astype(np.float16).view(np.uint16)
    return np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")


def build_both(dtype, metric, dim, rows, levels, m, efc, batch, fused=True):
    n = len(levels)
    e = hx.Engine(dtype, metric, dim, n)
    e.append(rows)
    ix = hx.Index(e, m, efc)
    ix.set_fused(fused)
    elem = ix.insert(0, levels, batch=batch)
    o = orc.Index(dtype, metric, dim, m=m, ef_construction=efc, order=orc.W64)
    tids = np.arange(n, dtype=np.int64)
    # the oracle's snapshot schedule with the same batch size (batch=1 == the reference's sequential
    # schedule, rows merged as duplicates staying behind as tombstones so ids line up)
    oelem = []
    i = 0
    for b in hx.batch_schedule(0, n, batch):
        oelem.append(o.insert_batch(rows[i:i + b], levels[i:i + b], tids[i:i + b]))
        i += b
    return e, ix, elem, o, np.concatenate(oelem)


def assert_same_graph(ix, o, n):
    assert ix.size == o.size == n
    assert ix.entry == o.entry
    for i in range(n):
        lv = ix.level(i)
        if o.merged(i):
            assert lv < 0
            continue
        assert lv == o.level(i), i
        assert ix.heaptids(i) == o.tids(i)
        for layer in range(lv + 1):
            gi, gd = ix.neighbors(i, layer)
            oi, od = o.neighbors(i, layer)
            assert gi.tolist() == oi.tolist(), (i, layer)
            assert (gd.view(np.uint32) == od.view(np.uint32)).all(), (i, layer)


CASES = [
    (hx.F32, hx.L2SQ, 16, 700, 8, 32),
    (hx.F32, hx.NEG_IP, 24, 500, 4, 16),
    (hx.F32, hx.L1, 3, 600, 16, 64),
    (hx.F16, hx.L2SQ, 10, 500, 16, 64),
    (hx.BIT, hx.HAMMING, 52, 600, 6, 24),      # integer distances: ties everywhere (tie-order test)
    (hx.BIT, hx.JACCARD, 40, 400, 16, 64),
]


# placements the default sizes never reach: m = 20 (fused searches, back-links on the lock-step path because k_links
# stops at m = 16), m = 40 (2m > 64: everything lock-step, select_neighbors in several blocks), ef_construction = 400,
# and the widest rows each type allows in an index (8 000-byte pitch: 8 chunks per row)
EDGE_CASES = [
    (hx.F32, hx.L2SQ, 8, 500, 20, 48, 37),
    (hx.F32, hx.NEG_IP, 12, 400, 40, 80, 29),
    (hx.F32, hx.L2SQ, 6, 900, 16, 400, 64),
    (hx.F32, hx.L1, 2000, 160, 4, 16, 16),
    (hx.F16, hx.NEG_IP, 4000, 150, 4, 16, 16),
    (hx.BIT, hx.HAMMING, 64000, 120, 4, 16, 16),
]


@pytest.mark.parametrize("dtype,metric,dim,n,m,efc,batch", EDGE_CASES)
def test_graph_identical_edge_placements(dtype, metric, dim, n, m, efc, batch):
    rng = np.random.default_rng(dim + m)
    rows = make_rows(dtype, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=6)
    e, ix, elem, o, oelem = build_both(dtype, metric, dim, rows, levels, m, efc, batch, True)
    assert elem.tolist() == oelem.tolist()
    assert_same_graph(ix, o, n)
    qs = make_rows(dtype, 10, dim, rng)
    e.set_queries(qs)
    tids, d, el, cnt = ix.search(10, 40, 10)
    for q in range(10):
        want = o.scan(qs[q], ef_search=40, limit=10)
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in want]
    ix.close()
    e.close()


@pytest.mark.parametrize("dtype,metric,dim,n,m,efc", CASES)
@pytest.mark.parametrize("batch", [1, 37])
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "lockstep"])
def test_graph_identical_to_oracle(dtype, metric, dim, n, m, efc, batch, fused):
    """fused = the device-resident traversal kernel (hx_fused.inc.h); lockstep = the host driver.  Both must
    reproduce the oracle's graph and result lists bit for bit."""
    rng = np.random.default_rng(dim * 7 + batch)
    rows = make_rows(dtype, n, dim, rng)
    rows[50] = rows[10]           # duplicates exercise build.rs:482-512
    rows[51] = rows[10]
    levels = hx.draw_levels(n, m, seed=4)
    e, ix, elem, o, oelem = build_both(dtype, metric, dim, rows, levels, m, efc, batch, fused)
    assert elem.tolist() == oelem.tolist()
    assert_same_graph(ix, o, n)
    st = ix.fused_stats()
    assert (st["tasks"] > 0) == fused and st["redone"] <= st["tasks"] // 10
    # search parity: same tids, same distances (bits), same order, incl. ties
    nq, efs, k = 40, 40, 10
    qs = make_rows(dtype, nq, dim, rng)
    e.set_queries(qs)
    tids, d, el, cnt = ix.search(nq, efs, k)
    for q in range(nq):
        want = o.scan(qs[q], ef_search=efs, limit=k)
        assert cnt[q] == len(want)
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in want]
        assert (d[q, :cnt[q]].view(np.uint32) == np.float32([x for _, x, _ in want]).view(np.uint32)).all()
    ix.close()
    e.close()


def test_sequential_schedule_equals_reference_schedule():
    """hx batch=1 (tombstones) and the oracle's pop-on-duplicate sequential insert (build.rs:507-509) give
    the same graph up to the renumbering caused by popped rows."""
    rng = np.random.default_rng(2)
    n, dim, m, efc = 300, 8, 8, 32
    rows = make_rows(hx.F32, n, dim, rng)
    rows[100] = rows[7]
    levels = hx.draw_levels(n, m, seed=1)
    e, ix, elem, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, 1)
    ref = orc.Index(hx.F32, hx.L2SQ, dim, m=m, ef_construction=efc, order=orc.W64)
    relem = ref.build(rows, levels, batch=1)
    # map device element ids -> reference element ids through the tids they hold
    dev2ref = {int(elem[t]): int(relem[t]) for t in range(n)}
    assert ref.size == n - 1
    for i in range(n):
        if ix.level(i) < 0:
            continue
        r = dev2ref[i]
        assert ix.level(i) == ref.level(r)
        for layer in range(ix.level(i) + 1):
            gi, _ = ix.neighbors(i, layer)
            oi, _ = ref.neighbors(r, layer)
            assert [dev2ref[int(x)] for x in gi] == oi.tolist()
    ix.close()
    e.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_iterative_scan_matches_oracle(mode):
    """hnsw.iterative_scan relaxed/strict (scan.rs:794-875) with a selective filter (tests/t/044)."""
    rng = np.random.default_rng(44)
    n, dim, m, efc = 3000, 3, 16, 64
    rows = make_rows(hx.F32, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=44)
    e, ix, _, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, 64)
    nq, efs, limit = 12, 40, 11
    qs = make_rows(hx.F32, nq, dim, rng)
    e.set_queries(qs)
    for c, max_tuples in [(1, 20000), (50, 20000), (50, 300), (500, 20000)]:
        passes = (np.arange(n) % c == 0).astype(np.uint8)
        tids, d, cnt = ix.search_iterative(nq, efs, mode, max_tuples, limit, passes)
        for q in range(nq):
            it = orc.ITER_RELAXED if mode == 1 else orc.ITER_STRICT
            want = [(t, x) for t, x, _ in o.scan(qs[q], ef_search=efs, iterative=it, max_scan_tuples=max_tuples) if passes[t]][:limit]
            assert tids[q, :cnt[q]].tolist() == [t for t, _ in want], (c, max_tuples, q)
    ix.close()
    e.close()


METRIC = {"l2": hx.L2SQ, "ip": hx.NEG_IP, "cosine": hx.NEG_IP, "l1": hx.L1, "hamming": hx.HAMMING, "jaccard": hx.JACCARD}
TYPE = {"vector": hx.F32, "halfvec": hx.F16, "bit": hx.BIT, "sparsevec": hx.SPARSE}


def enc(tname, v):
    if tname == "vector":
        return np.asarray(v, np.float32), len(v)
    if tname == "halfvec":
        return np.asarray(v, np.float32).astype(np.float16).view(np.uint16), len(v)
    if tname == "sparsevec":                      # the fixture writes '{1:3,2:4}/3' densely as [3, 4, 0]
        return orc.sparse_from_dense(v), len(v)
    return orc.pack_bits(v), len(v)


@pytest.mark.parametrize("case", G["regress_order"], ids=lambda c: c["ref"].split("/")[-1])
def test_regress_orderings_on_device(case):
    """The reference's pg_regress expected orderings, reproduced through the device path."""
    dt = TYPE[case["type"]]
    cosine = case["metric"] == "cosine"
    enc_rows = [enc(case["type"], r) for r in case["rows"]]
    dim = enc_rows[0][1]
    e = hx.Engine(dt, METRIC[case["metric"]], dim, 16)
    ix = hx.Index(e, 16, 64)
    for tid, (r, _) in enumerate(enc_rows):
        first = e.append(r[None, :] if r.ndim == 1 else r)
        if cosine:
            if e.normalize_rows(first, 1)[0] == 0.0:
                e.pop(1)                      # build.rs:433-435: zero-norm rows are not indexed
                continue
        if len(enc_rows) == 4 and tid == 3:
            ix.insert_ondisk(first, [0], tids=[tid], batch=1)   # the regress files INSERT their fourth row after CREATE INDEX: aminsert
        else:
            ix.insert(first, [0], tids=[tid], batch=1)
    q, _ = enc(case["type"], case["query"])
    e.set_queries(q[None, :], normalize=cosine)
    if case.get("iterative"):
        mode = 2 if case["iterative"] == "strict_order" else 1
        tids, _, cnt = ix.search_iterative(1, case.get("ef_search", 40), mode, 20000, 10)
    else:
        tids, _, _, cnt = ix.search(1, case.get("ef_search", 40), 10)
    got = [case["rows"][t] for t in tids[0, :cnt[0]]]
    assert got == case["expect"]
    ix.close()
    e.close()


@pytest.mark.parametrize("case", G["in_source_spi_tests"]["cases"], ids=lambda c: c["ref"].split("/")[-1])
def test_in_source_spi_cases_on_device(case):
    """The #[pg_test] cases of the reference's scan.rs / insert.rs / vacuum.rs that pin results of the path, through the C ABI: rows through hx_index_insert
    (build_callback), more rows through hx_index_insert_ondisk (aminsert), one scan."""
    dim, cosine = case["dim"], case["opclass"] == "cosine"
    n = len(case["build"]) + len(case["insert"])
    e = hx.Engine(hx.F32, METRIC[case["opclass"]], dim, max(n, 1))
    ix = hx.Index(e, 16, 64)
    rows = []
    for phase, src in (("build", case["build"]), ("insert", case["insert"])):
        for r in src:
            first = e.append(np.asarray(r, np.float32)[None, :])
            if cosine:
                assert e.normalize_rows(first, 1)[0] != 0.0
            if phase == "build":
                ix.insert(first, [0], tids=[len(rows)], batch=1)
            else:
                ix.insert_ondisk(first, [0], tids=[len(rows)], batch=1)
            rows.append(r)
    e.set_queries(np.asarray(case["query"], np.float32)[None, :], normalize=cosine)
    limit = case["limit"] if case["limit"] is not None else 1000
    if case.get("iterative"):
        tids, _, cnt = ix.search_iterative(1, case.get("ef_search", 40), 2 if case["iterative"] == "strict_order" else 1, 20000, limit)
    else:
        tids, _, _, cnt = ix.search(1, case.get("ef_search", 40), limit)
    if "expect_first" in case:
        assert [float(x) for x in rows[tids[0, 0]]] == [float(x) for x in case["expect_first"]]
    if "expect_count" in case:
        assert cnt[0] == case["expect_count"]
    ix.close()
    e.close()


@pytest.mark.parametrize("case", G["null_query_count"], ids=lambda c: c["ref"].split("/")[-1])
def test_null_and_zero_query_counts_on_device(case):
    """`ORDER BY val <-> (SELECT NULL::vector)` and the zero-vector cosine query of the reference's pg_regress files (scan.rs:186-187) on an index the
    device built: the expected counts, and -- with a NULL query every distance is 0.0, so the order is pure BinaryHeap tie behaviour -- the oracle's order."""
    dt = TYPE[case["type"]]
    cosine = case["metric"] == "cosine"
    enc_rows = [enc(case["type"], r) for r in case["rows"]]
    dim = enc_rows[0][1]
    e = hx.Engine(dt, METRIC[case["metric"]], dim, 16)
    ix = hx.Index(e, 16, 64)
    o = orc.Index(dt, METRIC[case["metric"]], dim, m=16, ef_construction=64, order=orc.W64)
    for tid, (r, _) in enumerate(enc_rows):
        first = e.append(r[None, :] if r.ndim == 1 else r)
        if cosine:
            if e.normalize_rows(first, 1)[0] == 0.0:
                e.pop(1)
                continue
        ix.insert(first, [0], tids=[tid], batch=1)
        o.insert(e.read_rows(first, 1)[0], 0, tid)
    if "query" in case:
        q, _ = enc(case["type"], case["query"])
        e.set_queries(q[None, :], normalize=cosine)
        tids, _, _, cnt = ix.search(1, 40, 10)
        got = tids[0, :cnt[0]].tolist()
        want = [t for t, _, _ in o.scan(orc.l2_normalize(dt, dim, q)[0] if cosine else q)]
    else:
        got = ix.search_null(40, 10)[0].tolist()
        want = [t for t, _, _ in o.scan(None)]
    assert len(got) == case["expect_count"]
    assert got == want
    ix.close()
    e.close()


def test_duplicates_20_identical_rows_on_device():
    """tests/t/015_hnsw_vector_duplicates.pl:24-37 through the device path."""
    e = hx.Engine(hx.F32, hx.L2SQ, 3, 32)
    e.append(np.ones((20, 3), np.float32))
    ix = hx.Index(e, 16, 64)
    ix.insert(0, np.zeros(20, np.int32), batch=1)
    live = [i for i in range(20) if ix.level(i) >= 0]
    assert len(live) == 2 and sorted(len(ix.heaptids(i)) for i in live) == [10, 10]
    e.set_queries(np.ones((1, 3), np.float32))
    tids, _, _, cnt = ix.search(1, 1, 40)
    assert cnt[0] == G["limits"]["duplicates_20_identical_rows_ef_search_1"]["expect_returned"]
    ix.close()
    e.close()


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "lockstep"])
@pytest.mark.parametrize("batch", [64, 500])
def test_duplicates_inside_one_batch(fused, batch):
    """Identical rows arriving in ONE batch cannot meet through the graph; they are merged all the same (device row hashes + byte
    comparison), in the order the sequential schedule would merge them: 20 identical rows -> 2 elements of 10 heap TIDs
    (tests/t/015_hnsw_vector_duplicates.pl:24-37), also when an older element with room exists (it is filled first)."""
    rng = np.random.default_rng(15)
    n, dim, m, efc = 700, 6, 8, 32
    rows = make_rows(hx.F32, n, dim, rng)
    rows[300:320] = rows[300]            # 20 identical rows, one batch
    rows[400:404] = rows[20]             # copies of an old row: merged through the graph
    rows[410] = rows[405]                # a pair inside the batch
    rows[650:662] = rows[300]            # 12 more copies of the 20, a later batch: the second element is full after 0, so a third appears
    levels = hx.draw_levels(n, m, seed=15)
    e, ix, elem, o, oelem = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, batch, fused)
    assert elem.tolist() == oelem.tolist()
    assert_same_graph(ix, o, n)
    live = sorted(set(int(x) for x in elem[300:320]))
    assert len(live) == 2 and sorted(len(ix.heaptids(i)) for i in live) == [10, 10]
    assert all(int(x) == 20 for x in elem[400:404]) and elem[410] == 405
    assert len(set(int(x) for x in elem[650:662])) == 2
    ix.close()
    e.close()


def test_option_limits():
    e = hx.Engine(hx.F32, hx.L2SQ, 3, 8)
    for m, efc in [(1, 64), (101, 1000), (16, 3), (16, 1001), (16, 31)]:      # options.rs:203-225, build.rs:865-867
        with pytest.raises(hx.HxError):
            hx.Index(e, m, efc)
    ix = hx.Index(e, 16, 64)
    tids, _, _, cnt = ix.search(0, 40, 10)                                      # nq = 0
    e.set_queries(np.zeros((1, 3), np.float32))
    tids, _, _, cnt = ix.search(1, 40, 10)                                      # empty index: scan.rs:469-472
    assert cnt[0] == 0
    with pytest.raises(hx.HxError):
        ix.search(1, 1001, 10)                                                  # options.rs:156-166
    ix.close()
    e.close()


def test_c1_recall_config():
    """BASELINE configs[0]: 10k x vector(128) L2, m=16, ef_construction=64, ef_search=40.  recall@10 of the
    batched device build vs exact brute force must equal, within sampling noise, the recall of the
    reference's own schedule and summation order (oracle: sequential inserts, ORC_ORDER_SEQ)."""
    rng = np.random.default_rng(1)
    n, dim, m, efc, efs, k = 10_000, 128, 16, 64, 40, 10
    rows = rng.random((n, dim), dtype=np.float32)
    qs = rng.random((1000, dim), dtype=np.float32)
    levels = hx.draw_levels(n, m, seed=1)
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append(rows)
    ix = hx.Index(e, m, efc)
    ix.insert(0, levels, batch=256)
    e.set_queries(qs)
    tids, d, _, cnt = ix.search(len(qs), efs, k)
    d2 = (qs.astype(np.float64) ** 2).sum(1)[:, None] + (rows.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * qs.astype(np.float64) @ rows.astype(np.float64).T
    exact = np.argsort(d2, axis=1)[:, :k]
    hits = np.array([len(set(tids[q, :cnt[q]].tolist()) & set(exact[q].tolist())) / k for q in range(len(qs))])
    o = orc.Index(orc.F32, orc.L2SQ, dim, m=m, ef_construction=efc, order=orc.SEQ)
    o.build(rows, levels, batch=1)
    ref_ids, ref_cnt = o.search_many(qs, efs, k, n_threads=8)
    ref_hits = np.array([len(set(ref_ids[q, :ref_cnt[q]].tolist()) & set(exact[q].tolist())) / k for q in range(len(qs))])
    recall, ref_recall = hits.mean(), ref_hits.mean()
    # uniform 128-d data is a hard case (the reference's own gates use 3-d data): the bar is equality with the reference's schedule and
    # summation order, query by query: the mean difference within 2 sigma of its sampling noise (+ 0.003)
    diff = hits - ref_hits
    sem = diff.std(ddof=1) / np.sqrt(len(diff))
    assert abs(diff.mean()) <= 2.0 * sem + 0.003, (recall, ref_recall, diff.mean(), sem)
    assert recall >= 0.5
    ix.close()
    e.close()


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "lockstep"])
def test_staged_batches_two_rank_simulation(fused):
    """The multi-GPU protocol of pgvector-rx_amd/dist_build.py, played by two engines on one GPU: each 'rank' searches
    its slice and prunes the lists it owns, the serialized lists are exchanged by hand, and both replicas must end up
    bit-identical to the single-rank build (and hence to the oracle)."""
    import importlib
    db = importlib.import_module("pgvector-rx_amd.dist_build")
    rng = np.random.default_rng(77)
    n, dim, m, efc, batch, world = 1500, 24, 8, 32, 96, 2
    rows = make_rows(hx.F32, n, dim, rng)
    rows[300] = rows[5]
    levels = hx.draw_levels(n, m, seed=3)
    tids = np.arange(n, dtype=np.int64)
    e0, ix0, _, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, batch, fused)
    ranks = []
    for r in range(world):
        e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
        e.append(rows)
        ix = hx.Index(e, m, efc)
        ix.set_fused(fused)
        ranks.append((e, ix))
    done = 0
    for b in hx.batch_schedule(0, n, batch):
        if ranks[0][1].entry < 0 or b < 16:
            for _, ix in ranks:
                ix.insert(done, levels[done:done + b], tids[done:done + b], batch=b)
            done += b
            continue
        lo, hi = db.slice_bounds(b, world)
        for r, (_, ix) in enumerate(ranks):
            ix.batch_begin(done, levels[done:done + b], tids[done:done + b])
            ix.batch_search(lo[r], hi[r])
        bufs = [ix.batch_export_new(lo[r], hi[r]) for r, (_, ix) in enumerate(ranks)]
        for r, (_, ix) in enumerate(ranks):
            for s in range(world):
                if s != r and hi[s] > lo[s]:
                    ix.batch_import_new(lo[s], hi[s], bufs[s])
        for r, (_, ix) in enumerate(ranks):
            ix.batch_links(r, world)
        bufs = [ix.batch_export_links() for _, ix in ranks]
        for r, (_, ix) in enumerate(ranks):
            for s in range(world):
                if s != r and len(bufs[s]):
                    ix.batch_import_links(bufs[s])
            ix.batch_end(b)
        done += b
    for _, ix in ranks:
        assert_same_graph(ix, o, n)
    # a search on a replica that imported half of its lists (mirror refreshed from imports)
    qs = make_rows(hx.F32, 16, dim, rng)
    for e, ix in ranks + [(e0, ix0)]:
        e.set_queries(qs)
    ref = ix0.search(16, 40, 10)
    for _, ix in ranks:
        got = ix.search(16, 40, 10)
        assert (got[0] == ref[0]).all() and (got[1].view(np.uint32) == ref[1].view(np.uint32)).all()
    for e, ix in ranks + [(e0, ix0)]:
        ix.close()
        e.close()


def test_device_batches_two_rank_simulation():
    """The device-resident exchange of pgvector-rx_amd/dist_build.py played by two engines on one GPU: member records and
    pruned-list records move between the 'ranks' as device buffers only; both replicas must equal the oracle's graph."""
    import torch
    rng = np.random.default_rng(78)
    n, dim, m, efc, batch, world = 2500, 24, 16, 64, 256, 2
    rows = make_rows(hx.F32, n, dim, rng)
    rows[300] = rows[5]
    rows[1203] = rows[1200]
    levels = hx.draw_levels(n, m, seed=3)
    tids = np.arange(n, dtype=np.int64)
    e0, ix0, _, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, batch)
    ranks = []
    for r in range(world):
        e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
        e.append(rows)
        ranks.append((e, hx.Index(e, m, efc)))
    rb, lb = ranks[0][1].dbatch_record_bytes, ranks[0][1].dbatch_list_record_bytes
    done, n_dev = 0, 0
    for b in hx.batch_schedule(0, n, batch):
        lv, td = levels[done:done + b], tids[done:done + b]
        if ranks[0][1].entry < 0 or b < 16 or not ranks[0][1].dbatch_supported(lv):
            for _, ix in ranks:
                ix.insert(done, lv, td, batch=b)
            done += b
            continue
        n_dev += 1
        per = -(-b // world)
        recs = torch.zeros(world * per * rb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()                                                 # the engines run on their own streams
        for r, (_, ix) in enumerate(ranks):
            ix.dbatch_begin(done, lv, td)
            lo, hi = min(b, r * per), min(b, r * per + per)
            ix.dbatch_search(lo, hi, recs.data_ptr() + lo * rb)                 # "all-gather": both ranks write into the one buffer
        counts = [ix.dbatch_links(r, world, recs.data_ptr()) for r, (_, ix) in enumerate(ranks)]
        bufs = []
        for r, (_, ix) in enumerate(ranks):
            t = torch.zeros(max(counts[r], 1) * lb, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ix.dbatch_export_links(t.data_ptr())
            bufs.append(t)
        for r, (_, ix) in enumerate(ranks):
            for s in range(world):
                if s != r and counts[s]:
                    ix.dbatch_import_links(bufs[s].data_ptr(), counts[s])
            ix.dbatch_end(b)
        done += b
    assert n_dev >= 5
    for _, ix in ranks:
        assert_same_graph(ix, o, n)
    for e, ix in ranks + [(e0, ix0)]:
        ix.close()
        e.close()


def test_fused_equals_lockstep_at_scale():
    """BASELINE-sized rows (d=768) and a graph large enough for multi-level descent: the device-resident traversal and
    the lock-step driver must return identical tids and distance bits, and both must agree with exact brute force to the
    usual HNSW recall (size-independent properties; the oracle would take minutes at this size)."""
    rng = np.random.default_rng(9)
    n, dim, m, efc, efs, k, nq = 60_000, 768, 16, 64, 64, 10, 500
    centres = rng.random((64, dim), dtype=np.float32)
    rows = (centres[rng.integers(0, 64, n)] + 0.1 * rng.standard_normal((n, dim), dtype=np.float32)).astype(np.float32)
    qs = (centres[rng.integers(0, 64, nq)] + 0.1 * rng.standard_normal((nq, dim), dtype=np.float32)).astype(np.float32)
    levels = hx.draw_levels(n, m, seed=2)
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append(rows)
    ix = hx.Index(e, m, efc)
    ix.insert(0, levels, batch=4096)
    assert ix.fused_stats()["redone"] == 0
    e.set_queries(qs)
    a = ix.search(nq, efs, k)
    ix.set_fused(False)
    b = ix.search(nq, efs, k)
    assert (a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() and (a[3] == b[3]).all()
    d2 = (qs.astype(np.float64) ** 2).sum(1)[:, None] + (rows.astype(np.float64) ** 2).sum(1)[None, :] - 2.0 * qs.astype(np.float64) @ rows.astype(np.float64).T
    exact = np.argsort(d2, axis=1)[:, :k]
    recall = np.mean([len(set(a[0][q, :a[3][q]].tolist()) & set(exact[q].tolist())) / k for q in range(nq)])
    assert recall >= 0.9, recall
    # sortedness and exactness of the returned distances
    assert (np.diff(a[1], axis=1) >= 0).all()
    chk = ((rows[a[0][:, 0]].astype(np.float64) - qs.astype(np.float64)) ** 2).sum(1)
    assert (np.abs(a[1][:, 0] - chk) <= 1e-5 * chk).all()
    ix.close()
    e.close()


@pytest.mark.gpu
def test_full_size_c2_structural_properties():
    """BASELINE configs[1] at its full size (1M x vector(768), m=16, ef_construction=200), where the oracle cannot follow:
    size-independent properties of the reference's graph and scan --
      * every list respects its bound (2m at layer 0, m above: hnsw_get_layer_m), holds valid, distinct, non-self ids of elements
        that reach that layer, and its stored distances are the exact distance bits of (owner, neighbour);
      * level histogram equals the drawn levels, the entry point is a top-level element (build.rs:516-525);
      * scan results come back sorted, distances within 1e-5 of an f64 recomputation, recall@10 vs exact brute force as bench.py
        reports it; scanning twice is idempotent; the lock-step placement returns the same tids and distance bits for a query sample."""
    import torch
    n, dim, m, efc, efs, k, nq = 1_000_000, 768, 16, 200, 100, 10, 2000
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    cen = torch.rand((1024, dim), generator=g, device="cuda")
    rows = torch.empty((n, dim), device="cuda")
    for i in range(0, n, 1 << 17):
        j = min(n, i + (1 << 17))
        rows[i:j] = cen[torch.randint(0, 1024, (j - i,), generator=g, device="cuda")] + 0.1 * torch.randn((j - i, dim), generator=g, device="cuda")
    qs = cen[torch.randint(0, 1024, (nq,), generator=g, device="cuda")] + 0.1 * torch.randn((nq, dim), generator=g, device="cuda")
    torch.cuda.synchronize()
    levels = hx.draw_levels(n, m, seed=11)
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append_device(rows.data_ptr(), n)
    ix = hx.Index(e, m, efc)
    ix.insert(0, levels, batch=32768)                                   # the bench's insert batch cap (bench.py --batch)
    assert ix.size == n and ix.fused_stats()["redone"] == 0
    lv = ix.export_levels()
    assert np.array_equal(lv, levels)                                   # no duplicates in this data: nothing tombstoned
    top = int(lv.max())
    assert lv[ix.entry] == top
    rng = np.random.default_rng(1)
    for layer in range(top + 1):
        ids, dist, cnt = ix.export_layer(layer)
        lm = 2 * m if layer == 0 else m
        owners = np.nonzero(lv >= layer)[0]
        c = cnt[owners].astype(np.int64)
        assert (c <= lm).all() and (cnt[lv < layer] == 0).all()
        if layer == 0:
            assert (c >= 1).all()                                        # every element is linked at layer 0
        valid = np.arange(lm)[None, :] < c[:, None]
        idv, dv = ids[owners], dist[owners]
        assert (idv[valid] < n).all() and (lv[idv[valid]] >= layer).all()
        assert not (idv == owners[:, None])[valid].any()                 # no self links
        srt = np.sort(np.where(valid, idv, np.uint32(0xFFFFFFFF)), axis=1)
        dup = (srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 0xFFFFFFFF)
        assert not dup.any()                                             # distinct neighbours
        # stored distances: exact bits of d(owner, neighbour) from the batched kernel, for a sample of lists
        pick = owners[rng.integers(0, len(owners), min(2000, len(owners)))]
        goff = np.concatenate([[0], np.cumsum(cnt[pick].astype(np.int64))]).astype(np.uint32)
        gids = np.concatenate([ids[p, :cnt[p]] for p in pick]).astype(np.uint32)
        fresh = e.distances_batch(pick.astype(np.uint32), goff, gids)
        stored = np.concatenate([dist[p, :cnt[p]] for p in pick])
        assert np.array_equal(fresh.view(np.uint32), stored.view(np.uint32))
    # scans
    e.set_queries_device(qs.data_ptr(), nq)
    a = ix.search(nq, efs, k)
    b = ix.search(nq, efs, k)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)                                      # idempotent
    assert (a[3] == k).all() and (np.diff(a[1], axis=1) >= 0).all()
    rn = (rows * rows).sum(1)
    gt = torch.cat([torch.topk(rn[None, :] - 2.0 * qs[i:i + 250] @ rows.T, k, dim=1, largest=False).indices for i in range(0, nq, 250)]).cpu().numpy()
    recall = np.mean([len(set(a[0][q].tolist()) & set(gt[q].tolist())) / k for q in range(nq)])
    assert recall >= 0.95, recall
    near = torch.from_numpy(a[0][:, 0].astype(np.int64)).cuda()
    chk = ((rows[near].double() - qs.double()) ** 2).sum(1).cpu().numpy()
    assert (np.abs(a[1][:, 0] - chk) <= 1e-5 * chk).all()
    ix.set_fused(False)
    e.set_queries_device(qs.data_ptr(), 200)
    c2 = ix.search(200, efs, k)
    assert np.array_equal(c2[0], a[0][:200]) and np.array_equal(c2[1].view(np.uint32), a[1][:200].view(np.uint32))
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dt,metric,dim,n", [(hx.F32, hx.L2SQ, 8, 6000), (hx.BIT, hx.HAMMING, 64, 5000), (hx.F16, hx.NEG_IP, 24, 4000)])
@pytest.mark.parametrize("mode", [1, 2])
def test_iterative_scan_on_device_equals_lockstep_and_oracle(dt, metric, dim, n, mode):
    """Iterative scans run inside k_fused (MODE 2: visited set and `discarded` heap kept across resumes on the device).  Same
    tids, same distance bits as the lock-step host driver, and the oracle's tids -- Hamming included, where equal distances
    make the order of pushes into `discarded` visible."""
    rng = np.random.default_rng(7)
    m, efc = 8, 32
    rows = make_rows(dt, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=7)
    e, ix, _, o, _ = build_both(dt, metric, dim, rows, levels, m, efc, 64)
    nq, efs, limit = 40, 20, 9
    qs = make_rows(dt, nq, dim, rng)
    e.set_queries(qs)
    for c, max_tuples in [(1, 20000), (40, 20000), (40, 150), (700, 20000), (3, 1), (900, 3_000_000)]:     # the last one sizes the tables for a huge max_scan_tuples
        passes = (np.arange(n) % c == 0).astype(np.uint8)
        before = ix.fused_stats()
        ix.set_fused(True)
        a = ix.search_iterative(nq, efs, mode, max_tuples, limit, passes)
        after = ix.fused_stats()
        assert after["tasks"] == before["tasks"] + nq and after["redone"] == before["redone"]     # served by the device kernel
        ix.set_fused(False)
        b = ix.search_iterative(nq, efs, mode, max_tuples, limit, passes)
        ix.set_fused(True)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), (c, max_tuples)
        it = orc.ITER_RELAXED if mode == 1 else orc.ITER_STRICT
        for q in range(0, nq, 5):
            want = [t for t, x, _ in o.scan(qs[q], ef_search=efs, iterative=it, max_scan_tuples=max_tuples) if passes[t]][:limit]
            assert a[0][q, :a[2][q]].tolist() == want, (c, max_tuples, q)
    ix.close()
    e.close()


@pytest.mark.gpu
def test_iterative_scan_that_outgrows_its_tables_is_retried_on_the_device():
    """An iterative scan whose visited table fills up reports FS_OVERFLOW; it is run again on the device with 8x the tables before the lock-step
    driver is considered (a deep scan there costs thousands of host round trips).  HX_ITER_VIS_SHIFT=-5 makes the first tables 32x too small."""
    import os
    rng = np.random.default_rng(19)
    n, dim, m, efc = 6000, 8, 8, 32
    rows = make_rows(hx.F32, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=19)
    e, ix, _, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, 64)
    nq, efs, limit = 30, 20, 9
    qs = make_rows(hx.F32, nq, dim, rng)
    e.set_queries(qs)
    passes = (np.arange(n) % 700 == 0).astype(np.uint8)
    ref = ix.search_iterative(nq, efs, 1, 20000, limit, passes)
    os.environ["HX_ITER_VIS_SHIFT"] = "-5"
    try:
        before = ix.fused_stats()
        got = ix.search_iterative(nq, efs, 1, 20000, limit, passes)
        after = ix.fused_stats()
    finally:
        del os.environ["HX_ITER_VIS_SHIFT"]
    assert after["redone"] == before["redone"]                       # nothing reached the lock-step driver
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for q in range(0, nq, 3):
        want = [t for t, x, _ in o.scan(qs[q], ef_search=efs, iterative=orc.ITER_RELAXED, max_scan_tuples=20000) if passes[t]][:limit]
        assert got[0][q, :got[2][q]].tolist() == want
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dt,dim", [(hx.F32, 24), (hx.F16, 40)])
def test_hub_lists_inner_product_long_back_link_chains(dt, dim):
    """Inner product on rows of very different norms: a few high-norm rows are everybody's neighbour, so single lists receive
    hundreds of back-links per batch (k_links_hub: ops evaluated speculatively by several waves, applied in order).  The graph
    must still equal the oracle's, list by list, distance bits included."""
    rng = np.random.default_rng(31)
    n, m, efc, batch = 5000, 16, 64, 2048
    base = rng.random((n, dim), dtype=np.float32)
    scale = np.exp(rng.normal(0.0, 1.2, n)).astype(np.float32)[:, None]         # log-normal norms: strong hubs
    rows = (base * scale).astype(np.float32)
    if dt == hx.F16:
        rows = rows.astype(np.float16).view(np.uint16)
    levels = hx.draw_levels(n, m, seed=31)
    e, ix, _, o, _ = build_both(dt, hx.NEG_IP, dim, rows, levels, m, efc, batch)
    prof = ix.profile()
    assert prof["links_max_chain"] >= 200, prof["links_max_chain"]                 # the hub path was exercised
    assert ix.fused_stats()["redone"] == 0
    assert_same_graph(ix, o, n)
    ix.close()
    e.close()


def _random_cases():
    rng = np.random.default_rng(2026)
    cases = []
    for i in range(28):
        dt = [hx.F32, hx.F16, hx.BIT][int(rng.integers(0, 3))]
        if dt == hx.BIT:
            metric = [hx.HAMMING, hx.JACCARD][int(rng.integers(0, 2))]
            dim = int(rng.choice([9, 52, 64, 100, 1024, 1500, 4100]))       # payloads on both sides of 128 B / 512 B
        else:
            metric = [hx.L2SQ, hx.NEG_IP, hx.L1][int(rng.integers(0, 3))]
            dim = int(rng.choice([1, 3, 17, 31, 32, 33, 64, 127, 129, 200, 257, 300]))
        m = int(rng.choice([2, 3, 5, 8, 12, 16]))                              # m in 17..32: test_m_above_16_stays_on_the_device
        efc = int(rng.choice([2 * m, 2 * m + 7, 40, 64]))
        efc = max(efc, 2 * m)
        n = int(rng.integers(250, 900))
        batch = int(rng.choice([1, 7, 64, 300]))
        dup = int(rng.choice([0, 0, 11]))
        cases.append((i, dt, metric, dim, m, efc, n, batch, dup))
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("case", _random_cases(), ids=lambda c: "r%d-t%d-m%d-d%d-M%d-e%d-n%d-b%d-dup%d" % c)
def test_randomized_shapes_match_oracle(case):
    """A seeded sweep over dtype x metric x dim x m x ef_construction x batch x duplicates: graph identity with the oracle,
    then the same top-k and the same relaxed iterative scan as the oracle, all through the device-resident kernels."""
    i, dt, metric, dim, m, efc, n, batch, dup = case
    rng = np.random.default_rng(1000 + i)
    rows = make_rows(dt, n, dim, rng)
    if dup:
        for j in range(dup, n, dup):
            rows[j] = rows[j - 1]
    levels = hx.draw_levels(n, m, seed=1000 + i)
    e, ix, _, o, _ = build_both(dt, metric, dim, rows, levels, m, efc, batch)
    assert ix.fused_stats()["redone"] <= int((levels >= 8).sum())      # only rows above the kernel's 8 layers go to the lock-step path
    assert_same_graph(ix, o, n)
    nq, efs, k = 8, max(10, m), 6
    qs = make_rows(dt, nq, dim, rng)
    e.set_queries(qs)
    tids, d, el, cnt = ix.search(nq, efs, k)
    passes = (np.arange(n) % 3 == 0).astype(np.uint8)
    it = ix.search_iterative(nq, efs, 1, 500, 5, passes)
    for q in range(nq):
        want = [t for t, _, _ in o.scan(qs[q], ef_search=efs, limit=k)]
        assert tids[q, :cnt[q]].tolist() == want
        wit = [t for t, _, _ in o.scan(qs[q], ef_search=efs, iterative=orc.ITER_RELAXED, max_scan_tuples=500) if passes[t]][:5]
        assert it[0][q, :it[2][q]].tolist() == wit
    ix.close()
    e.close()


@pytest.mark.gpu
def test_overflowed_tasks_are_retried_on_the_device_with_roomier_tables():
    """A task whose visited table or candidate heap overflows is re-run by k_fused with 8x the tables before the lock-step path
    is considered (it would need the host copy of every list).  HX_FORCE_OVERFLOW_MOD marks every 3rd task as overflowed."""
    import os
    rng = np.random.default_rng(77)
    n, dim, m, efc = 1500, 20, 8, 40
    rows = make_rows(hx.F32, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=77)
    os.environ["HX_FORCE_OVERFLOW_MOD"] = "3"
    try:
        e, ix, _, o, _ = build_both(hx.F32, hx.L2SQ, dim, rows, levels, m, efc, 128)
        assert ix.fused_stats()["redone"] == 0                       # nothing reached the lock-step path
        assert_same_graph(ix, o, n)
        qs = make_rows(hx.F32, 30, dim, rng)
        e.set_queries(qs)
        tids, d, el, cnt = ix.search(30, 24, 7)
        for q in range(30):
            assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=24, limit=7)]
        assert ix.fused_stats()["redone"] == 0
    finally:
        del os.environ["HX_FORCE_OVERFLOW_MOD"]
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,m", [(hx.F32, hx.L2SQ, 6, 40), (hx.BIT, hx.HAMMING, 64, 100)])
def test_scans_run_in_the_traversal_kernel_for_every_legal_m(dtype, metric, dim, m):
    """m in 33..100 (options.rs:203-225): plain and iterative scans run inside k_fused, which walks a layer-0 list of up to 200 ids 64 at a time -- same
    tids as the oracle."""
    rng = np.random.default_rng(m)
    n, efc = 1200, 2 * m
    rows = make_rows(dtype, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=m)
    e, ix, _, o, _ = build_both(dtype, metric, dim, rows, levels, m, efc, 64)
    assert_same_graph(ix, o, n)
    qs = make_rows(dtype, 16, dim, rng)
    e.set_queries(qs)
    before = ix.fused_stats()
    tids, d, el, cnt = ix.search(16, 60, 10)
    passes = (np.arange(n) % 7 == 0).astype(np.uint8)
    it = ix.search_iterative(16, 60, 1, 20000, 8, passes)
    after = ix.fused_stats()
    assert after["tasks"] == before["tasks"] + 32 and after["redone"] == before["redone"]
    for q in range(16):
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=60, limit=10)]
        want = [t for t, _, _ in o.scan(qs[q], ef_search=60, iterative=orc.ITER_RELAXED, max_scan_tuples=20000) if passes[t]][:8]
        assert it[0][q, :it[2][q]].tolist() == want
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,m,efc,batch", [
    (hx.F32, hx.L2SQ, 6, 40, 80, 64), (hx.F32, hx.NEG_IP, 24, 64, 128, 200), (hx.F16, hx.L2SQ, 40, 100, 200, 97),
    (hx.BIT, hx.HAMMING, 64, 100, 200, 64), (hx.F32, hx.L1, 300, 50, 120, 128), (hx.F16, hx.NEG_IP, 600, 33, 70, 1),
    (hx.F32, hx.L2SQ, 8, 40, 500, 128)])                                            # result sets of up to 500 candidates through k_select_w
def test_builds_for_every_legal_m_stay_on_the_device(dtype, metric, dim, m, efc, batch):
    """m in 33..100 (options.rs:203-225): the searches run in k_fused (MODE 3), select_neighbors in k_select_w and update_neighbor_connections in
    k_list_ops (hx_biglist.hip: lists of up to 200 slots walked 64 at a time, pair distances evaluated as check_element_closer asks for them).
    Graph identical to the oracle's (Hamming: ties everywhere), no lock-step round, no task redone."""
    rng = np.random.default_rng(m + dim)
    n = (1500 if m < 64 else 900 if m < 100 else 600) if dim < 100 else 600        # the oracle's share of this test grows with m^2 and with dim
    rows = make_rows(dtype, n, dim, rng)
    levels = hx.draw_levels(n, m, seed=m)
    e, ix, elem, o, oelem = build_both(dtype, metric, dim, rows, levels, m, efc, batch)
    assert elem.tolist() == oelem.tolist()
    assert_same_graph(ix, o, n)
    assert ix.profile()["rounds"] == 0
    assert ix.fused_stats()["redone"] == 0
    assert max(len(ix.neighbors(i, 0)[0]) for i in range(n)) == 2 * m          # the lists did fill: back-links were pruned
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("dtype,metric,dim", [(hx.F32, hx.L2SQ, 24), (hx.F16, hx.NEG_IP, 40), (hx.F32, hx.L1, 300)])
def test_sorted_array_search_equals_the_heap_search(dtype, metric, dim, ties):
    """HX_SORTED_ARRAY=1: queries search on one sorted array (f_search_layer_sa) and a query that meets two equal distances is redone by the
    heap kernel.  Small-integer coordinates make distances collide all the time (ties=True): both branches must give the oracle's answers."""
    import os
    rng = np.random.default_rng(dim + (7 if ties else 0))
    n, m, efc = 3000, 12, 48
    if ties:
        base = rng.integers(-2, 3, size=(n + 40, dim))
        base[100] = base[7]
        allr = base.astype(np.float16).view(np.uint16) if dtype == hx.F16 else base.astype(np.float32)
        rows, qs = np.ascontiguousarray(allr[:n]), np.ascontiguousarray(allr[n:])
    else:
        rows, qs = make_rows(dtype, n, dim, rng), make_rows(dtype, 40, dim, rng)
    levels = hx.draw_levels(n, m, seed=5)
    e, ix, _, o, _ = build_both(dtype, metric, dim, rows, levels, m, efc, 256)
    e.set_queries(qs)
    ref = ix.search(40, 64, 10)
    os.environ["HX_SORTED_ARRAY"] = "1"
    try:
        got = ix.search(40, 64, 10)
    finally:
        del os.environ["HX_SORTED_ARRAY"]
    assert ix.fused_stats()["redone"] == 0                           # ties are redone on the device, not by the lock-step driver
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    for q in range(40):
        assert got[0][q, :got[3][q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=64, limit=10)]
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,n,m,efc,batch", [(hx.F32, hx.L2SQ, 8, 1500, 20, 48, 64), (hx.F32, hx.NEG_IP, 12, 1200, 32, 64, 100), (hx.BIT, hx.HAMMING, 96, 1000, 24, 64, 37)])
def test_m_above_16_stays_on_the_device(dtype, metric, dim, n, m, efc, batch):
    """m in 17..32 (lists of up to 64): traversal, back-link kernels (64-slot build) and the batch pipeline all run on the device -- nothing falls
    to the lock-step host driver -- and the graph still equals the oracle's."""
    rng = np.random.default_rng(m * 31 + dim)
    rows = make_rows(dtype, n, dim, rng)
    rows[700] = rows[11]
    levels = hx.draw_levels(n, m, seed=17)
    e, ix, elem, o, oelem = build_both(dtype, metric, dim, rows, levels, m, efc, batch, True)
    assert elem.tolist() == oelem.tolist()
    assert ix.fused_stats()["redone"] == 0 and ix.profile()["rounds"] == 0
    assert_same_graph(ix, o, n)
    qs = make_rows(dtype, 10, dim, rng)
    e.set_queries(qs)
    tids, d, el, cnt = ix.search(10, 40, 10)
    for q in range(10):
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=40, limit=10)]
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim", [(hx.F32, hx.L2SQ, 200), (hx.F16, hx.NEG_IP, 40), (hx.BIT, hx.HAMMING, 96)])
def test_pipelined_scans_equal_the_plain_scan(dtype, metric, dim, monkeypatch):
    """hx_index_search_submit / _wait: several batches of queries in flight on slots with streams and tables of their own return, batch by
    batch, exactly what hx_index_search returns for the same query slots (and the oracle's answers); a busy slot, an empty slot and a
    mutation while a scan is in flight are refused.  The second pass forces overflow retries inside the waits."""
    rng = np.random.default_rng(dim)
    n, m, efc, nq, k, efs = 4000, 12, 48, 96, 10, 64
    rows, qs = make_rows(dtype, n, dim, rng), make_rows(dtype, 3 * nq, dim, rng)
    levels = hx.draw_levels(n, m, seed=21)
    e, ix, _, o, _ = build_both(dtype, metric, dim, rows, levels, m, efc, 256)
    e.set_queries(qs)
    ref = ix.search(3 * nq, efs, k)
    for force in (None, "5"):
        if force:
            monkeypatch.setenv("HX_FORCE_OVERFLOW_MOD", force)
        ix.search_submit(0, 0, nq, efs, k)
        ix.search_submit(1, nq, nq, efs, k)
        with pytest.raises(hx.HxError):
            ix.search_submit(1, 2 * nq, nq, efs, k)                     # slot busy
        with pytest.raises(hx.HxError):
            ix.insert(0, levels[:1], batch=1)                           # mutation while scans are in flight
        got0 = ix.search_wait(0)
        ix.search_submit(0, 2 * nq, nq, efs, k)                          # slot 0 again while slot 1 is still in flight
        got1 = ix.search_wait(1)
        got2 = ix.search_wait(0)
        with pytest.raises(hx.HxError):
            ix.search_wait(0)                                           # nothing submitted
        for b, got in enumerate((got0, got1, got2)):
            for a, r in zip(got, ref):
                assert np.array_equal(a, r[b * nq:(b + 1) * nq]), (force, b)
    for q in range(0, 3 * nq, 7):
        assert ref[0][q, :ref[3][q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=efs, limit=k)]
    ix.close()
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,n,m,efc,batch", [(hx.F32, hx.L2SQ, 24, 900, 40, 80, 64), (hx.BIT, hx.HAMMING, 128, 700, 100, 200, 50)])
def test_m_above_32_searches_in_the_traversal_kernel(dtype, metric, dim, n, m, efc, batch):
    """m in 33..100 (options.rs:203-225; lists of up to 200): the searches of the build run in the traversal kernel (MODE 3: lists longer than a
    wavefront are walked 64 ids at a time, every layer's W handed out); select_neighbors and the back-links follow in hx_biglist.hip's list kernels
    (test_builds_for_every_legal_m_stay_on_the_device).  Graph, duplicates and scans equal the oracle's."""
    rng = np.random.default_rng(m * 7 + dim)
    rows = make_rows(dtype, n, dim, rng)
    rows[n // 2] = rows[3]
    levels = hx.draw_levels(n, m, seed=19)
    e, ix, elem, o, oelem = build_both(dtype, metric, dim, rows, levels, m, efc, batch, True)
    assert elem.tolist() == oelem.tolist()
    st = ix.fused_stats()
    assert st["tasks"] > n // 2 and st["redone"] == 0                    # the members' searches were device tasks
    assert_same_graph(ix, o, n)
    qs = make_rows(dtype, 10, dim, rng)
    e.set_queries(qs)
    tids, d, el, cnt = ix.search(10, 40, 10)
    for q in range(10):
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in o.scan(qs[q], ef_search=40, limit=10)]
    ix.close()
    e.close()
