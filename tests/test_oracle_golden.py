"""Pins the CPU oracle (oracle/hnsw_oracle.c) to every known answer the reference's own tests hold
for the HNSW distance hot path (tests/golden/reference_known_answers.json, each with file:line)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import orc

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
TYPE = {"vector": orc.F32, "halfvec": orc.F16, "bit": orc.BIT, "sparsevec": orc.SPARSE}


def enc(tname, v):
    """Text literal -> the row payload the reference stores (vector.rs:43-48, halfvec.rs:41-46, bitvec.rs:28-37)."""
    if tname == "vector":
        return np.asarray(v, np.float32), len(v)
    if tname == "halfvec":
        return np.array([orc.lib().orc_f32_to_half(float(x)) for x in v], np.uint16), len(v)
    if tname == "sparsevec":                      # written densely in the fixture: '{1:3,2:4}/3' = [3, 4, 0]; sparsevec.rs:217-330 keeps the non-zeros
        return orc.sparse_from_dense(v), len(v)
    return orc.pack_bits(v), len(v)


@pytest.mark.parametrize("case", G["distance"], ids=lambda c: c["fn"] + "@" + c["ref"].split("/")[-1])
def test_distance_known_answers(case):
    L = orc.lib()
    dt = TYPE[case["type"]]
    a, dim = enc(case["type"], case["a"])
    pa = a.ctypes.data
    fn = case["fn"]
    if fn == "vector_norm":
        got = L.orc_norm(dt, dim, pa)
    else:
        b, _ = enc(case["type"], case["b"])
        pb = b.ctypes.data
        if fn == "l2_distance":
            got = L.orc_l2_distance(dt, dim, pa, pb)
            # the opclass proc (squared) must agree too
            assert math.isclose(math.sqrt(L.orc_distance(dt, orc.L2SQ, dim, pa, pb, orc.SEQ)), got, abs_tol=1e-12)
        elif fn == "inner_product":
            got = L.orc_inner_product(dt, dim, pa, pb)
        elif fn == "negative_inner_product":
            got = L.orc_distance(dt, orc.NEG_IP, dim, pa, pb, orc.SEQ)
        elif fn == "cosine_distance":
            got = L.orc_cosine_distance(dt, dim, pa, pb)
        elif fn == "l1_distance":
            got = L.orc_distance(dt, orc.L1, dim, pa, pb, orc.SEQ)
        elif fn == "hamming_distance":
            got = L.orc_distance(dt, orc.HAMMING, dim, pa, pb, orc.SEQ)
        elif fn == "jaccard_distance":
            got = L.orc_distance(dt, orc.JACCARD, dim, pa, pb, orc.SEQ)
        else:
            raise AssertionError(fn)
    assert abs(got - case["expect"]) < case["tol"], (case, got)


@pytest.mark.parametrize("case", G["half_roundtrip"], ids=lambda c: str(c["value"]))
def test_half_roundtrip(case):
    L = orc.lib()
    v = float(case["value"])
    r = L.orc_half_to_f32(L.orc_f32_to_half(v))
    if case["expect"] == "exact":
        assert r == v
    elif case["expect"] == "within":
        assert abs(r - v) < case["tol"]
    elif case["expect"] == "inf":
        assert math.isinf(r) and (r > 0) == (v > 0)
    else:
        assert math.isnan(r)


def test_half_conversion_matches_ieee():
    """Not a reference fixture: the restated bit-twiddling must equal IEEE binary16 (numpy) on all
    65536 halves and on an RNE sweep of floats."""
    L = orc.lib()
    h = np.arange(65536, dtype=np.uint16)
    want = h.view(np.float16).astype(np.float32)
    got = np.array([L.orc_half_to_f32(int(x)) for x in h], np.float32)
    ok = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert ok.all()
    rng = np.random.default_rng(5)
    f = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000).astype(np.float32),
                        np.array([65504.0, 65519.99, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 6.1e-5], np.float32)])
    with np.errstate(over="ignore"):
        want16 = f.astype(np.float16).view(np.uint16)
    got16 = np.array([L.orc_f32_to_half(float(x)) for x in f], np.uint16)
    # Reference quirk kept on purpose (halfvec.rs:111-113): every |x| < 2^-24 (biased exponent < 103)
    # becomes signed zero, whereas IEEE RNE rounds (2^-25, 2^-24) up to the smallest subnormal.
    bits = f.view(np.uint32)
    quirk = (((bits >> 23) & 0xFF) == 102) & ((bits & 0x7FFFFF) != 0)
    want16 = np.where(quirk, (bits >> 16) & 0x8000, want16).astype(np.uint16)
    assert quirk.any() and (got16 == want16).all()


# ---- graph/mod.rs pure-Rust unit tests -------------------------------------------------------
GU = G["graph_unit"]


def _raw_index(case, dim):
    idx = orc.Index(orc.F32, orc.L2SQ, dim, m=case["m"], ef_construction=case.get("ef_construction", 16))
    levels = case.get("levels", [0] * len(case["positions"]))
    for p, lv in zip(case["positions"], levels):
        idx.add_raw(np.asarray(p, np.float32), lv)
    return idx


def test_search_layer_basic():
    c = GU["search_layer_basic"]
    idx = _raw_index(c, 1)
    n = len(c["positions"])
    for i in range(n - 1):
        idx.link_raw(i, 0, i + 1, c["chain_link_distance"])
        idx.link_raw(i + 1, 0, i, c["chain_link_distance"])
    ids, d = idx.search_layer_raw(np.asarray(c["query"], np.float32), [c["entry"]], c["ef"], c["layer"])
    assert len(ids) == c["expect_len"]
    assert set(c["expect_contains"]) <= set(ids.tolist())


def test_select_neighbors_fits():
    c = GU["select_neighbors_fits"]
    idx = _raw_index(c, 1)
    sel = idx.select_neighbors_raw([x[0] for x in c["candidates"]], [x[1] for x in c["candidates"]], c["max_neighbors"])
    assert len(sel) == c["expect_len"]


def test_select_neighbors_prunes():
    c = GU["select_neighbors_prunes"]
    idx = _raw_index(c, 2)
    sel = idx.select_neighbors_raw([x[0] for x in c["candidates"]], [x[1] for x in c["candidates"]], c["max_neighbors"])
    assert len(sel) == c["expect_len"] and sel[0] == c["expect_first"]
    # Traced by hand from graph/mod.rs:284-305: idx 2 is rejected (d(2,1)=0.01 <= 1.21), idx 3 is rejected too
    # (d(3,1)=16 <= 25), and the back-fill from `discarded` takes idx 2 first.
    assert sel.tolist() == [1, 2]


def test_find_element_neighbors_two_elements():
    c = GU["find_element_neighbors_two_elements"]
    idx = _raw_index(c, 2)
    idx.find_element_neighbors_raw(1, 0)
    ids, _ = idx.neighbors(1, 0)
    assert ids.tolist() == c["expect_neighbors_of_1_layer0"]


def test_update_neighbor_connections():
    c = GU["update_neighbor_connections"]
    idx = _raw_index(c, 1)
    idx.find_element_neighbors_raw(1, 0)
    idx.update_neighbor_connections_raw(1)
    ids, _ = idx.neighbors(0, 0)
    assert len(ids) > 0 and ids[0] == c["after_insert_1"]["elem0_layer0_first"]
    idx.find_element_neighbors_raw(2, 0)
    idx.update_neighbor_connections_raw(2)
    ids, _ = idx.neighbors(1, 0)
    assert set(c["after_insert_2_entry0"]["elem1_layer0_contains"]) <= set(ids.tolist())


def test_multi_layer_search():
    c = GU["multi_layer_search"]
    idx = _raw_index(c, 1)
    idx.find_element_neighbors_raw(1, 0)
    idx.update_neighbor_connections_raw(1)
    idx.find_element_neighbors_raw(2, 0)
    idx.update_neighbor_connections_raw(2)
    for e, layer in c["expect_nonempty"]:
        ids, _ = idx.neighbors(e, layer)
        assert ids is not None and len(ids) > 0


# ---- pg_regress expected orderings -------------------------------------------------------------
METRIC = {"l2": orc.L2SQ, "ip": orc.NEG_IP, "cosine": orc.NEG_IP, "l1": orc.L1, "hamming": orc.HAMMING, "jaccard": orc.JACCARD}
ITER = {None: orc.ITER_OFF, "strict_order": orc.ITER_STRICT, "relaxed_order": orc.ITER_RELAXED}


def _build_small(case, order=orc.SEQ):
    dt = TYPE[case["type"]]
    cosine = case["metric"] == "cosine"
    rows = [enc(case["type"], r) for r in case["rows"]]
    dim = rows[0][1]
    idx = orc.Index(dt, METRIC[case["metric"]], dim, m=16, ef_construction=64, order=order)
    idx.set_ondisk_tombstones(True)
    for tid, (r, _) in enumerate(rows):
        if cosine:
            r, norm = orc.l2_normalize(dt, dim, r)
            if norm == 0.0:          # build.rs:433-435
                continue
        if len(rows) == 4 and tid == 3:
            idx.insert_on_disk(r, 0, tid)        # the regress files INSERT their fourth row after CREATE INDEX: aminsert, not build_callback
        else:
            idx.insert(r, 0, tid)
    return idx, dt, dim, cosine


@pytest.mark.parametrize("order", [orc.SEQ, orc.W64])
@pytest.mark.parametrize("case", G["regress_order"], ids=lambda c: c["ref"].split("/")[-1])
def test_regress_orderings(case, order):
    idx, dt, dim, cosine = _build_small(case, order)
    q, _ = enc(case["type"], case["query"])
    if cosine:
        q, _ = orc.l2_normalize(dt, dim, q)          # scan.rs:749-751
    res = idx.scan(q, ef_search=case.get("ef_search", 40), iterative=ITER[case.get("iterative")])
    got = [case["rows"][tid] for tid, _, _ in res]
    assert got == case["expect"]


@pytest.mark.parametrize("case", G["null_query_count"], ids=lambda c: c["ref"].split("/")[-1])
def test_null_and_zero_query_counts(case):
    idx, dt, dim, cosine = _build_small(case)
    if "query" in case:
        q, _ = enc(case["type"], case["query"])
        if cosine:
            q, _ = orc.l2_normalize(dt, dim, q)
    else:
        q = None                                      # scan.rs:186-187: NULL query => distance 0.0
    assert len(idx.scan(q)) == case["expect_count"]


def test_limits():
    L = orc.lib()
    assert L.orc_max_level(16) == G["limits"]["max_level_m16"]["expect"]


def test_duplicates_20_identical_rows():
    """tests/t/015_hnsw_vector_duplicates.pl:24-37: 20 identical rows, ef_search=1 -> exactly 10 rows
    (HNSW_HEAPTIDS TIDs merged into one element, the 11th identical row becomes a new node)."""
    idx = orc.Index(orc.F32, orc.L2SQ, 3, m=16, ef_construction=64)
    rng = np.random.default_rng(0)
    for tid in range(20):
        idx.insert(np.array([1, 1, 1], np.float32), int(rng.integers(0, 2)), tid)
    assert idx.size == 2 and sorted(len(idx.tids(i)) for i in range(2)) == [10, 10]
    res = idx.scan(np.array([1, 1, 1], np.float32), ef_search=1)
    assert len(res) == G["limits"]["duplicates_20_identical_rows_ef_search_1"]["expect_returned"]


# ---- statistical recall gates (tests/t/012, 020, 024) -------------------------------------------
def _exact_topk(tname, metric, rows, q, k, dim=None):
    L = orc.lib()
    dt = TYPE[tname]
    if dim is None:
        dim = rows.shape[1] if tname != "bit" else None
    d = []
    for r in rows:
        if metric == "cosine":
            d.append(L.orc_cosine_distance(dt, dim, q.ctypes.data, r.ctypes.data))
        elif metric == "l2":
            d.append(L.orc_l2_distance(dt, dim, q.ctypes.data, r.ctypes.data))
        else:
            d.append(L.orc_distance(dt, METRIC[metric], dim if dim else 52, q.ctypes.data, r.ctypes.data, orc.SEQ))
    d = np.asarray(d)
    return d, np.argsort(d, kind="stable")[:k]


@pytest.mark.parametrize("gate", G["recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_recall_gates(gate):
    rng = np.random.default_rng(12)
    n, dim, k = 4000, gate["dim"], gate["k"]      # 4000 rows keeps the CPU suite short; the gate itself is size-free
    tname = gate["type"]
    dt = TYPE[tname]
    if tname == "bit":
        bits = rng.integers(0, 2, (n, dim)).astype(np.uint8)
        rows = np.packbits(bits, axis=1, bitorder="big")
        qs = np.packbits(rng.integers(0, 2, (gate["queries"], dim)).astype(np.uint8), axis=1, bitorder="big")
    else:
        raw = (rng.random((n, dim)) * rng.random((n, dim))).astype(np.float32)   # random()*random(), 012:11
        qraw = rng.random((gate["queries"], dim)).astype(np.float32)
        if tname == "halfvec":
            rows = raw.astype(np.float16).view(np.uint16)
            qs = qraw.astype(np.float16).view(np.uint16)
        elif tname == "sparsevec":                # ARRAY[random() * random(), ...]::vector::sparsevec, 028:11,57
            rows = np.stack([orc.sparse_from_dense(r) for r in raw])
            qs = np.stack([orc.sparse_from_dense(r) for r in qraw])
        else:
            rows, qs = raw, qraw
    levels = orc.levels_from_seed(n, gate["m"], 7)
    for metric, min_recall in gate["min_recall"].items():
        idx = orc.Index(dt, METRIC[metric], dim, m=gate["m"], ef_construction=gate["ef_construction"])
        elem_of = {}
        for i in range(n):
            r = rows[i]
            if metric == "cosine":
                r, norm = orc.l2_normalize(dt, dim, r)
                if norm == 0.0:
                    continue
            idx.insert(r, levels[i], i)
        correct = total = 0
        for q in qs:
            dists, exact = _exact_topk(tname, metric, rows, q, k, dim if tname == "sparsevec" else None)
            qq = q
            if metric == "cosine":
                qq, _ = orc.l2_normalize(dt, dim, q)
            got = [t for t, _, _ in idx.scan(qq, ef_search=gate["ef_search"], limit=k)]
            kth = dists[exact[-1]]
            # 020:60-66 counts ties with the k-th distance as correct for bit types
            okset = set(np.nonzero(dists <= kth)[0].tolist()) if tname == "bit" else set(exact.tolist())
            correct += sum(1 for t in got if t in okset) if tname == "bit" else len(okset & set(got))
            total += k
        assert correct / total >= min_recall, (metric, correct / total)


def test_search_many_threads_equal_single_and_vec_order_within_tolerance():
    """orc_search_many (bench.py's all-cores CPU leg) returns what orc_search_topk returns per query, on any thread count;
    ORC_ORDER_VEC (the reassociated, vectorised CPU baseline) stays within 1e-5 relative of the reference's scalar order."""
    rng = np.random.default_rng(3)
    n, dim = 600, 48
    rows = rng.random((n, dim), dtype=np.float32)
    qs = rng.random((17, dim), dtype=np.float32)
    levels = orc.levels_from_seed(n, 8, 5)
    x = orc.Index(orc.F32, orc.L2SQ, dim, m=8, ef_construction=32, order=orc.SEQ)
    x.build(rows, levels, batch=1)
    ids1, cnt1 = x.search_many(qs, 24, 5, n_threads=1)
    ids4, cnt4 = x.search_many(qs, 24, 5, n_threads=4)
    assert np.array_equal(ids1, ids4) and np.array_equal(cnt1, cnt4)
    for q in range(len(qs)):
        ids, _ = x.search_topk(qs[q], 24, 5)
        assert ids.tolist() == ids1[q, :cnt1[q]].tolist()
    for metric in (orc.L2SQ, orc.NEG_IP, orc.L1):
        for d in (1, 15, 16, 17, 48, 131):
            a, b = rng.standard_normal(d).astype(np.float32), rng.standard_normal(d).astype(np.float32)
            s = orc.distance(orc.F32, metric, d, a, b, order=orc.SEQ)
            v = orc.distance(orc.F32, metric, d, a, b, order=orc.VEC)
            scale = float(np.abs(a.astype(np.float64) * b).sum()) if metric == orc.NEG_IP else abs(s)
            assert abs(s - v) <= 1e-5 * max(scale, 1e-30)


def test_w64_fast_order_equals_plain_restatement():
    """The lane-parallel arrangement of ORC_ORDER_W64 (what graph builds run through) gives the bits of its plain restatement."""
    L = orc.lib()
    rng = np.random.default_rng(64)
    for dt in (orc.F32, orc.F16):
        for dim in list(range(1, 20)) + [63, 64, 65, 255, 256, 257, 511, 512, 513, 768, 1536, 2000] + ([4000] if dt == orc.F16 else []):
            for kind in (0, 1, 2):
                if dt == orc.F32:
                    a = (rng.standard_normal(dim) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
                    b = rng.standard_normal(dim).astype(np.float32)
                else:
                    a = rng.integers(0, 0x7C00, dim).astype(np.uint16)
                    b = (rng.integers(0, 0x7C00, dim) | (rng.integers(0, 2, dim) << 15)).astype(np.uint16)
                x = np.float32(L.orc_acc_w64_plain(kind, dt, dim, a.ctypes.data, b.ctypes.data))
                y = np.float32(L.orc_acc_w64_fast(kind, dt, dim, a.ctypes.data, b.ctypes.data))
                assert x.view(np.uint32) == y.view(np.uint32), (dt, dim, kind)


def _iter_gate_expected(rows64, q64, metric, c, limit):
    if metric == "l2":
        d = np.sqrt(((rows64 - q64) ** 2).sum(1))
    else:
        d = 1.0 - (rows64 @ q64) / np.sqrt((rows64 ** 2).sum(1) * (q64 ** 2).sum())
    ids = np.arange(1, len(d) + 1)
    top = np.sort(d[ids % c == 0])[:limit]
    return set(ids[d <= top[-1]].tolist())


@pytest.mark.parametrize("gate", G["iterative_recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_iterative_recall_gate(gate):
    """tests/t/044_hnsw_iterative_scan_recall.pl: `WHERE i % c = 0 ORDER BY v <-> q LIMIT 20` under strict_order / relaxed_order
    must find >= 0.99 of the exact answer (the oracle runs the gate at a reduced row count to keep the CPU suite short)."""
    rng = np.random.default_rng(44)
    n, dim, limit = 12000, gate["dim"], gate["limit"]
    raw = rng.random((n, dim)).astype(np.float32)
    qs = rng.random((gate["queries"], dim)).astype(np.float32)
    levels = orc.levels_from_seed(n, gate["m"], 44)
    for metric in gate["metrics"]:
        idx = orc.Index(orc.F32, METRIC[metric], dim, m=gate["m"], ef_construction=gate["ef_construction"])
        for i in range(n):
            r = raw[i]
            if metric == "cosine":
                r, norm = orc.l2_normalize(orc.F32, dim, r)
                assert norm > 0
            idx.insert(r, levels[i], i + 1)                      # i = 1..n as in generate_series(1, n)
        for c in (50, 120):                                      # 500 of 50 000 rows -> the same 1-in-100ish selectivity at this size
            for mode in gate["modes"]:
                it = orc.ITER_STRICT if mode == "strict_order" else orc.ITER_RELAXED
                correct = total = 0
                for q in qs:
                    qq = orc.l2_normalize(orc.F32, dim, q)[0] if metric == "cosine" else q
                    got = [t for t, _, _ in idx.scan(qq, ef_search=gate["ef_search"], iterative=it, max_scan_tuples=20000) if t % c == 0][:limit]
                    ok = _iter_gate_expected(raw.astype(np.float64), q.astype(np.float64), metric, c, limit)
                    correct += sum(1 for t in got if t in ok)
                    total += limit
                assert correct / total >= gate["min_recall"], (metric, c, mode, correct / total)


def _live_recall(idx, qs, exact, ef_search, k, alive):
    """What `SELECT ... ORDER BY v <-> q LIMIT k` returns when the index still holds dead TIDs: the scan's ef_search tuples in order,
    the dead ones dropped by the heap visit, the first k kept.  exact[i]: the TIDs that count as correct for query i."""
    correct = 0
    for i, q in enumerate(qs):
        got = [t for t, _, _ in idx.scan(q, ef_search=ef_search) if alive(t)][:k]
        correct += len(set(got) & set(exact[i]))
    return correct / (k * len(qs))


def _vacuum_gate_data(gate, rng):
    """Rows / queries of the vacuum recall tests (random() per component: 014:60, 026:60, 030:60; random bits: 022:63), the oracle's encoding of them
    and, per query, the TIDs (1-based, among the rows that stay) that count as correct: the k nearest, plus ties with the k-th for bit (022:84-89)."""
    n, dim, k, keep, tname = gate["rows"], gate["dim"], gate["k"], gate["keep"], gate["type"]
    if tname == "bit":
        bits = rng.integers(0, 2, (n, dim)).astype(np.uint8); qbits = rng.integers(0, 2, (gate["queries"], dim)).astype(np.uint8)
        rows, qs = np.packbits(bits, axis=1, bitorder="big"), np.packbits(qbits, axis=1, bitorder="big")
        d = (qbits[:, None, :] != bits[None, :keep, :]).sum(2).astype(np.float64)
    else:
        raw = rng.random((n, dim)).astype(np.float32); qraw = rng.random((gate["queries"], dim)).astype(np.float32)
        if tname == "halfvec":
            raw, qraw = raw.astype(np.float16), qraw.astype(np.float16)
            rows, qs = raw.view(np.uint16), qraw.view(np.uint16)
        elif tname == "sparsevec":
            rows, qs = np.stack([orc.sparse_from_dense(r) for r in raw]), np.stack([orc.sparse_from_dense(r) for r in qraw])
        else:
            rows, qs = raw, qraw
        d = ((qraw[:, None, :].astype(np.float64) - raw[None, :keep, :].astype(np.float64)) ** 2).sum(2)
    exact = []
    for q in range(len(qs)):
        kth = np.sort(d[q], kind="stable")[k - 1]
        exact.append((np.nonzero(d[q] <= kth)[0] + 1).tolist() if tname == "bit" else (np.argsort(d[q], kind="stable")[:k] + 1).tolist())
    return rows, qs, exact


def _bit_population_queries(gate, rows, nq):
    """nq further random bit queries for a bit gate and their exact answers among the kept rows (ties with the k-th distance count)."""
    rng = np.random.default_rng(2200)
    dim, k, keep = gate["dim"], gate["k"], gate["keep"]
    qbits = rng.integers(0, 2, (nq, dim)).astype(np.uint8)
    bits = np.unpackbits(rows[:keep], axis=1, bitorder="big")[:, :dim]
    d = (qbits[:, None, :] != bits[None, :, :]).sum(2)
    exact = []
    for q in range(nq):
        kth = np.sort(d[q], kind="stable")[k - 1]
        exact.append((np.nonzero(d[q] <= kth)[0] + 1).tolist())
    return np.packbits(qbits, axis=1, bitorder="big"), exact


def _gate_data(gate, rng, n):
    """Rows and queries with the distributions of the reference's recall tests (012:11, 020:62, 024:12, 028:11), encoded for the oracle."""
    tname, dim = gate["type"], gate["dim"]
    if tname == "bit":
        rows = np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")
        qs = np.packbits(rng.integers(0, 2, (gate["queries"], dim)).astype(np.uint8), axis=1, bitorder="big")
        return rows, qs
    scale = 2.0 if tname == "halfvec" else 1.0
    raw = (scale * rng.random((n, dim)) * rng.random((n, dim))).astype(np.float32)
    qraw = rng.random((gate["queries"], dim)).astype(np.float32)
    if tname == "halfvec":
        return raw.astype(np.float16).view(np.uint16), qraw.astype(np.float16).view(np.uint16)
    if tname == "sparsevec":
        return np.stack([orc.sparse_from_dense(r) for r in raw]), np.stack([orc.sparse_from_dense(r) for r in qraw])
    return raw, qraw


@pytest.mark.parametrize("gate", G["insert_recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_insert_recall_gate(gate):
    """tests/t/013, 021, 025, 029: an index filled ONLY through aminsert (find_element_neighbors_on_disk + get_update_index) reaches the build path's
    recall, for every operator class of vector, bit, halfvec and sparsevec."""
    rng = np.random.default_rng(13)
    n, dim, k = 2000, gate["dim"], gate["k"]                  # the gate is size-free; 2000 rows keep the CPU suite short
    tname = gate["type"]
    dt = TYPE[tname]
    rows, qs = _gate_data(gate, rng, n)
    levels = orc.levels_from_seed(n, gate["m"], 13)
    for metric, min_recall in gate["min_recall"].items():
        idx = orc.Index(dt, METRIC[metric], dim, m=gate["m"], ef_construction=gate["ef_construction"])
        for i in range(n):
            r = rows[i]
            if metric == "cosine":
                r, norm = orc.l2_normalize(dt, dim, r)
                if norm == 0.0:
                    continue
            idx.insert_on_disk(r, levels[i], i)
        correct = 0
        for q in qs:
            dists, exact = _exact_topk(tname, metric, rows, q, k, dim if tname == "sparsevec" else None)
            qq = orc.l2_normalize(dt, dim, q)[0] if metric == "cosine" else q
            got = [t for t, _, _ in idx.scan(qq, ef_search=gate["ef_search"], limit=k)]
            ok = set(np.nonzero(dists <= dists[exact[-1]])[0].tolist())          # ties with the k-th distance count (021:60-66)
            correct += sum(1 for t in got if t in ok)
        assert correct / (k * len(qs)) >= min_recall, (metric, correct / (k * len(qs)))


@pytest.mark.parametrize("gate", G["vacuum_recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_vacuum_recall_gate(gate):
    """tests/t/014, 022, 026, 030 at their full size: 10 000 rows, m = 4, ef_construction = 8; rows 2501.. deleted; recall before and after VACUUM."""
    rng = np.random.default_rng(14)
    n, dim, k, keep = gate["rows"], gate["dim"], gate["k"], gate["keep"]
    rows, qs, exact = _vacuum_gate_data(gate, rng)
    levels = orc.levels_from_seed(n, gate["m"], 14)
    idx = orc.Index(TYPE[gate["type"]], METRIC[gate["metric"]], dim, m=gate["m"], ef_construction=gate["ef_construction"])
    idx.build(rows, levels, batch=1, tids=np.arange(1, n + 1))
    for g in gate["before_vacuum"]:
        r = _live_recall(idx, qs, exact, g["ef_search"], k, lambda t: t <= keep)
        assert r >= g["min_recall"] - gate.get("noise", 0.0), ("before", g, r)
    idx.vacuum(np.arange(keep + 1, n + 1))
    assert sum(idx.deleted(e) for e in range(n)) == n - keep
    for g in gate["after_vacuum"]:
        r = _live_recall(idx, qs, exact, g["ef_search"], k, lambda t: True)
        assert r >= g["min_recall"] - gate.get("noise", 0.0), ("after", g, r)
        if gate.get("population_queries"):      # the gate's stated threshold, no allowance, on a query sample large enough to measure the graph rather than the sample
            q2, ex2 = _bit_population_queries(gate, rows, gate["population_queries"])
            r2 = _live_recall(idx, q2, ex2, g["ef_search"], k, lambda t: True)
            assert r2 >= g["min_recall"], ("after, population", g, r2)
    # nothing dead is returned any more, and no live element links to a deleted one
    assert all(t <= keep for q in qs for t, _, _ in idx.scan(q, ef_search=100))
    for e in range(n):
        if not idx.deleted(e):
            for layer in range(idx.level(e) + 1):
                assert all(not idx.deleted(int(x)) for x in idx.neighbors(e, layer)[0])


# ---- duplicates for every type, through CREATE INDEX and through aminsert (tests/t/015, 023, 027, 031) ----------------------------------
def _dup_row(case):
    if case["type"] == "bit":
        return orc.pack_bits(case["row"])
    v = np.asarray(case["row"], np.float32)
    if case["type"] == "halfvec":
        return v.astype(np.float16).view(np.uint16)
    if case["type"] == "sparsevec":
        return orc.sparse_from_dense(v)
    return v


@pytest.mark.parametrize("case", G["limits"]["duplicates_20_identical_rows_all_types"]["cases"], ids=lambda c: c["ref"].split("/")[-1].split(".")[0])
@pytest.mark.parametrize("path", ["build", "aminsert"])
def test_duplicates_all_types_both_paths(case, path):
    """20 identical rows -> 2 elements of 10 heap TIDs -> a scan with ef_search = 1 returns exactly 10 rows: rows present at CREATE INDEX
    (build.rs:482-512; 0xx:34-37) and rows inserted one by one into the truncated index (insert.rs:1136-1214; 0xx:39-46)."""
    dt = TYPE[case["type"]]
    row = _dup_row(case)
    idx = orc.Index(dt, METRIC[case["metric"]], case["dim"], m=16, ef_construction=64)
    rng = np.random.default_rng(23)
    for tid in range(20):
        lv = int(orc.levels_from_seed(20, 16, 23)[tid])
        (idx.insert if path == "build" else idx.insert_on_disk)(row, lv, tid)
    live = [i for i in range(idx.size) if not idx.merged(i)]
    assert sorted(len(idx.tids(i)) for i in live) == [10, 10]
    res = idx.scan(row, ef_search=1)
    assert len(res) == G["limits"]["duplicates_20_identical_rows_all_types"]["expect_returned"]


# ---- the stop rule of iterative scans (tests/t/043) ---------------------------------------------------------------------------------
def _max_scan_tuples_data():
    g = G["max_scan_tuples_gate"]
    rng = np.random.default_rng(43)
    rows = rng.random((g["rows"], g["dim"])).astype(np.float32)
    return g, rows, orc.levels_from_seed(g["rows"], g["m"], 43)


def test_max_scan_tuples_gate():
    """tests/t/043_hnsw_iterative_scan.pl at its full size: with max_scan_tuples = 100000 all 10 rows with i % 10000 = 0 are found; with 30000 / 50000 /
    70000 the average number found over 20 queries is within 2 of max_scan_tuples / 10000 (scan.rs:827-841: resuming stops once that many tuples were
    returned, then `discarded` is drained one element at a time)."""
    g, rows, levels = _max_scan_tuples_data()
    idx = orc.Index(orc.F32, orc.L2SQ, g["dim"], m=g["m"], ef_construction=g["ef_construction"])
    idx.build(rows, levels, batch=1, tids=np.arange(1, g["rows"] + 1))            # i = 1..n (generate_series)

    def count(q, max_tuples):
        got = [t for t, _, _ in idx.scan(q, ef_search=g["ef_search"], iterative=orc.ITER_RELAXED, max_scan_tuples=max_tuples) if t % g["filter_mod"] == 0]
        return len(got[:g["limit"]])
    assert count(rows[0], g["full"]["max_scan_tuples"]) == g["full"]["expect_count"]
    for part in g["partial"]:
        avg = sum(count(rows[i], part["max_scan_tuples"]) for i in range(part["queries"])) / part["queries"]
        assert part["expect_avg"] - part["slack"] < avg < part["expect_avg"] + part["slack"], (part, avg)


# ---- sparsevec: aminsert into an empty index + DELETE + VACUUM, three rounds (tests/t/038) ----------------------------------------------
def sparse_038_rows(rng, n, dim, max_entries):
    """Rows as 038:19-33 makes them: int(rand() * 100) draws of an index in [1, dim - 1] (duplicates dropped), values rand(); possibly none."""
    out = []
    for _ in range(n):
        seen = {}
        for _ in range(int(rng.random() * (max_entries + 1))):
            ix = int(rng.random() * (dim - 1)) + 1
            if ix not in seen:
                seen[ix] = np.float32(rng.random())
        ks = sorted(seen)
        out.append((np.asarray([k - 1 for k in ks], np.int32), np.asarray([seen[k] for k in ks], np.float32)))     # text indices are 1-based
    return out


def test_sparsevec_vacuum_insert_rounds():
    """tests/t/038_hnsw_sparsevec_vacuum_insert.pl: nothing errors, and what the reference leaves implicit: after every VACUUM no dead TID is
    returned and no live element links to a deleted one; the inserts of the next round land in the vacuumed graph."""
    g = G["sparsevec_vacuum_insert"]
    rng = np.random.default_rng(38)
    dim = g["dim"]
    idx = orc.Index(orc.SPARSE, orc.L2SQ, dim, m=g["m"], ef_construction=g["ef_construction"])
    levels = orc.levels_from_seed(g["rounds"] * g["inserts_per_round"], g["m"], 38)
    serial, dead = 0, set()
    for _ in range(g["rounds"]):
        rows = sparse_038_rows(rng, g["inserts_per_round"], dim, g["max_entries"])
        for idxs, vals in rows:
            serial += 1
            idx.insert_on_disk(orc.pack_sparse(dim, idxs, vals), int(levels[serial - 1]), serial)
        kill = [i for i in range(1, serial + 1) if i % g["delete_mod"] == 0 and i not in dead]
        idx.vacuum(np.asarray(kill, np.int64))
        dead.update(kill)
        q = orc.pack_sparse(dim, *rows[0])
        got = [t for t, _, _ in idx.scan(q, ef_search=40)]
        assert got and not (set(got) & dead)
        for e in range(idx.size):
            if not idx.deleted(e) and not idx.merged(e):
                for layer in range(idx.level(e) + 1):
                    assert all(not idx.deleted(int(x)) for x in idx.neighbors(e, layer)[0])


@pytest.mark.parametrize("mods", [(97, 350, 911), (3, 5, 7)], ids=["mostly-distinct", "105-distinct-rows"])
def test_reference_011_all_but_one_comes_back(mods):
    """tests/t/011_hnsw_vacuum.pl:29-52 on the oracle: build over 10 000 rows ARRAY[i % a, i % b, i % c], delete all + vacuum, the same rows through aminsert,
    delete all but i = 123 + vacuum: `ORDER BY v <-> '[0,0,0]' LIMIT 10` returns 123 alone (the survivor is the entry point, which vacuum.rs:300-303 does not
    repair against itself: its lists still name deleted tuples, and load_element skips those, scan.rs:178-181)."""
    n, dim, m, efc = 10_000, 3, 16, 64
    i = np.arange(1, n + 1)
    rows = np.stack([i % mods[0], i % mods[1], i % mods[2]], axis=1).astype(np.float32)
    levels = orc.levels_from_seed(2 * n, m, 11).astype(np.int32)
    o = orc.Index(orc.F32, orc.L2SQ, dim, m=m, ef_construction=efc)
    o.set_ondisk_tombstones(True)
    tids = np.arange(1, n + 1, dtype=np.int64)
    for k in range(n):
        o.insert_batch(rows[k:k + 1], levels[k:k + 1], tids[k:k + 1])
    o.vacuum(tids)
    assert o.entry < 0
    tids2 = np.arange(n + 1, 2 * n + 1, dtype=np.int64)
    for k in range(n):
        o.insert_on_disk(rows[k], int(levels[n + k]), int(tids2[k]))
    q0 = np.zeros(dim, np.float32)
    assert len(o.scan(q0, ef_search=40, limit=10)) == 10
    keep = int(tids2[122])
    o.vacuum(tids2[tids2 != keep])
    assert [t for t, _, _ in o.scan(q0, ef_search=40, limit=10)] == [keep]


@pytest.mark.parametrize("case", G["in_source_spi_tests"]["cases"], ids=lambda c: c["ref"].split("/")[-1])
def test_in_source_spi_cases(case):
    """The #[pg_test] cases of the reference's scan.rs / insert.rs / vacuum.rs that pin results of the path: rows through build_callback, more rows through
    aminsert, one ORDER BY scan."""
    dim, cosine = case["dim"], case["opclass"] == "cosine"
    idx = orc.Index(orc.F32, METRIC[case["opclass"]], dim, m=16, ef_construction=64)
    idx.set_ondisk_tombstones(True)
    rows = []
    for phase, src in (("build", case["build"]), ("insert", case["insert"])):
        for r in src:
            v = np.asarray(r, np.float32)
            if cosine:
                v, norm = orc.l2_normalize(orc.F32, dim, v)
                assert norm != 0.0
            if phase == "build":
                idx.insert(v, 0, len(rows))
            else:
                idx.insert_on_disk(v, 0, len(rows))
            rows.append(r)
    q = np.asarray(case["query"], np.float32)
    if cosine:
        q, _ = orc.l2_normalize(orc.F32, dim, q)
    res = idx.scan(q, ef_search=case.get("ef_search", 40), iterative=ITER[case.get("iterative")], limit=case["limit"])
    if "expect_first" in case:
        assert [float(x) for x in rows[res[0][0]]] == [float(x) for x in case["expect_first"]]
    if "expect_count" in case:
        assert len(res) == case["expect_count"]
