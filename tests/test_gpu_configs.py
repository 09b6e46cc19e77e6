"""The BASELINE.json configurations as configurations (dtype, metric, dimension, m, ef_construction of each config, at a row count the
oracle can follow), and the reference's statistical recall gates at their full size, all through the device path.

C3 vector(1536) cosine  : rows and queries normalised with the f64 procedure (vector.rs:106-126), then negative inner product
                          (vector_cosine_ops FUNCTION 1, vector.rs:852-856); m 16, ef_construction 200
C4 halfvec(4000) IP     : 2*U*U rounded to f16 (tests/t/024:12); m 16, ef_construction 200
C5 bit(1024) Hamming    : m 16, ef_construction 64; iterative_scan = relaxed_order, max_scan_tuples 20000, a 1 % filter, ef_search 40
Graphs, top-k lists and iterative scans must equal the oracle's (ORC_ORDER_W64: bit for bit); the distances are also held to the
north-star tolerance against the reference's own summation order (ORC_ORDER_SEQ), and the worst relative error observed is reported."""
import json
import os

import numpy as np
import pytest

import pgvector_rx_amd as hx
from oracle import orc
from test_gpu_index import assert_same_graph, build_both

pytestmark = pytest.mark.gpu
G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_known_answers.json")))
METRIC = {"l2": hx.L2SQ, "ip": hx.NEG_IP, "cosine": hx.NEG_IP, "l1": hx.L1, "hamming": hx.HAMMING, "jaccard": hx.JACCARD}
TYPE = {"vector": hx.F32, "halfvec": hx.F16, "bit": hx.BIT, "sparsevec": hx.SPARSE}


def _seq_tolerance_report(name, dt, metric, dim, e, rows, qs, record_property):
    """Device distances (canonical order) against the reference's scalar order: |delta| <= 1e-5 * sum|a_i b_i| for inner product
    (a relative bound on a sum that cancels towards 0 has no meaning), <= 1e-5 * d for L2/L1.  Reports the worst |delta| / |d|."""
    rng = np.random.default_rng(5)
    ids = rng.integers(0, len(rows), 256).astype(np.uint32)
    worst_rel, worst_scaled = 0.0, 0.0
    for q in qs[:8]:
        got = e.distances(q, ids).astype(np.float64)
        ref = orc.distances_many(dt, metric, dim, q, rows, ids.astype(np.int32), order=orc.SEQ)
        if dt == hx.F32:
            qa, ra = q.astype(np.float64), rows[ids].astype(np.float64)
        else:
            qa, ra = q.view(np.float16).astype(np.float64), rows[ids].view(np.float16).astype(np.float64)
        scale = np.abs(ra * qa[None, :]).sum(1) if metric == hx.NEG_IP else np.abs(ref)
        delta = np.abs(got - ref)
        worst_scaled = max(worst_scaled, float((delta / np.maximum(scale, 1e-300)).max()))
        worst_rel = max(worst_rel, float((delta / np.maximum(np.abs(ref), 1e-300)).max()))
    record_property(name + "_max_rel_err_vs_reference_order", worst_rel)
    record_property(name + "_max_err_over_sum_abs_terms", worst_scaled)
    print("\n[%s] device vs reference summation order: max |delta|/|d| = %.3g, max |delta|/sum|a_i b_i| = %.3g" % (name, worst_rel, worst_scaled))
    assert worst_scaled <= 1e-5, (name, worst_scaled)
    return worst_rel


def _search_parity(e, ix, o, qs, efs, k, normalize=False):
    e.set_queries(qs, normalize=normalize)
    tids, d, _, cnt = ix.search(len(qs), efs, k)
    for q in range(len(qs)):
        qq = orc.l2_normalize(o.dtype, o.dim, qs[q])[0] if normalize else qs[q]
        want = o.scan(qq, ef_search=efs, limit=k)
        assert tids[q, :cnt[q]].tolist() == [t for t, _, _ in want], q
        assert (d[q, :cnt[q]].view(np.uint32) == np.float32([x for _, x, _ in want]).view(np.uint32)).all(), q


def test_config_c3_vector1536_cosine(record_property):
    rng = np.random.default_rng(21)
    n, dim, m, efc = 1500, 1536, 16, 200
    raw = rng.standard_normal((n, dim)).astype(np.float32)
    raw[77] = 0.0                                                # a zero vector: not indexed under cosine (build.rs:433-435)
    e = hx.Engine(hx.F32, hx.NEG_IP, dim, n)
    e.append(raw)
    norms = e.normalize_rows(0, n)
    keep = norms > 0
    assert (~keep).sum() == 1
    rows = e.read_rows(0, n)
    # the device's normalisation is the reference's (f64 norm, f64 divide, one rounding): bit-exact against the oracle
    for i in (0, 1, 500, n - 1):
        want, nrm = orc.l2_normalize(orc.F32, dim, raw[i])
        assert nrm == norms[i] and (want.view(np.uint32) == rows[i].view(np.uint32)).all()
    e.close()
    rows = rows[keep]
    n = len(rows)
    levels = hx.draw_levels(n, m, seed=21)
    e, ix, elem, o, oelem = build_both(hx.F32, hx.NEG_IP, dim, rows, levels, m, efc, 128)
    assert elem.tolist() == oelem.tolist() and ix.fused_stats()["redone"] == 0
    assert_same_graph(ix, o, n)
    qs = rng.standard_normal((24, dim)).astype(np.float32)
    _search_parity(e, ix, o, qs, 100, 10, normalize=True)
    qn = np.stack([orc.l2_normalize(orc.F32, dim, q)[0] for q in qs])
    _seq_tolerance_report("C3_vector1536_cosine", hx.F32, hx.NEG_IP, dim, e, rows, qn, record_property)
    ix.close()
    e.close()


def test_config_c4_halfvec4000_inner_product(record_property):
    rng = np.random.default_rng(31)
    n, dim, m, efc = 700, 4000, 16, 200
    rows = (2.0 * rng.random((n, dim)) * rng.random((n, dim))).astype(np.float16).view(np.uint16)
    levels = hx.draw_levels(n, m, seed=31)
    e, ix, elem, o, oelem = build_both(hx.F16, hx.NEG_IP, dim, rows, levels, m, efc, 96)
    assert elem.tolist() == oelem.tolist() and ix.fused_stats()["redone"] == 0
    assert_same_graph(ix, o, n)
    qs = (2.0 * rng.random((16, dim)) * rng.random((16, dim))).astype(np.float16).view(np.uint16)
    _search_parity(e, ix, o, qs, 100, 10)
    _seq_tolerance_report("C4_halfvec4000_ip", hx.F16, hx.NEG_IP, dim, e, rows, qs, record_property)
    ix.close()
    e.close()


@pytest.mark.parametrize("mode", [1, 2], ids=["relaxed_order", "strict_order"])
def test_config_c5_bit1024_hamming_iterative(mode):
    rng = np.random.default_rng(41)
    n, dim, m, efc, efs = 8000, 1024, 16, 64, 40
    rows = np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")
    rows[4000:4003] = rows[17]                                   # identical rows: Hamming distance 0, duplicate merge
    levels = hx.draw_levels(n, m, seed=41)
    e, ix, elem, o, oelem = build_both(hx.BIT, hx.HAMMING, dim, rows, levels, m, efc, 512)
    assert elem.tolist() == oelem.tolist() and ix.fused_stats()["redone"] == 0
    assert_same_graph(ix, o, n)
    qs = np.packbits(rng.integers(0, 2, (20, dim)).astype(np.uint8), axis=1, bitorder="big")
    _search_parity(e, ix, o, qs, efs, 10)
    e.set_queries(qs)
    passes = (np.arange(n) % 100 == 0).astype(np.uint8)          # the 1 % filter of configs[4]
    limit = 10
    before = ix.fused_stats()
    tids, d, cnt = ix.search_iterative(len(qs), efs, mode, 20000, limit, passes)
    after = ix.fused_stats()
    assert after["tasks"] == before["tasks"] + len(qs) and after["redone"] == before["redone"]     # served by the device kernel
    it = orc.ITER_RELAXED if mode == 1 else orc.ITER_STRICT
    for q in range(len(qs)):
        want = [(t, x) for t, x, _ in o.scan(qs[q], ef_search=efs, iterative=it, max_scan_tuples=20000) if passes[t]][:limit]
        assert tids[q, :cnt[q]].tolist() == [t for t, _ in want], q
        assert d[q, :cnt[q]].tolist() == [float(x) for _, x in want], q      # integer distances: exact
    ix.close()
    e.close()


def _gate_rows(gate, rng):
    n, dim = gate["rows"], gate["dim"]
    if gate["type"] == "bit":
        rows = np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")      # (random() * 2^52)::bigint::bit(52), 020:62
        qs = np.packbits(rng.integers(0, 2, (gate["queries"], dim)).astype(np.uint8), axis=1, bitorder="big")
        return rows, qs, None, None
    scale = 2.0 if gate["type"] == "halfvec" else 1.0                                                  # 2*random()*random() (024:12) / random()*random() (012:11)
    raw = (scale * rng.random((n, dim)) * rng.random((n, dim))).astype(np.float32)
    qraw = rng.random((gate["queries"], dim)).astype(np.float32)
    if gate["type"] == "halfvec":
        return raw.astype(np.float16).view(np.uint16), qraw.astype(np.float16).view(np.uint16), raw.astype(np.float16).astype(np.float64), qraw.astype(np.float16).astype(np.float64)
    if gate["type"] == "sparsevec":                                                                    # ARRAY[...]::vector::sparsevec, 028:11,57: the non-zero elements
        pack = lambda a: hx.pack_sparse(dim, [(np.nonzero(r)[0], r[np.nonzero(r)[0]]) for r in a])
        return pack(raw), pack(qraw), raw.astype(np.float64), qraw.astype(np.float64)
    return raw, qraw, raw.astype(np.float64), qraw.astype(np.float64)


def _exact(gate, metric, rows, q, r64, q64):
    if gate["type"] == "bit":
        a = np.unpackbits(rows, axis=1, bitorder="big")[:, :gate["dim"]].astype(np.int64)
        b = np.unpackbits(q, bitorder="big")[:gate["dim"]].astype(np.int64)
        if metric == "hamming":
            return (a != b).sum(1).astype(np.float64)
        ab = (a & b).sum(1)
        return np.where(ab == 0, 1.0, 1.0 - ab / np.maximum((a.sum(1) + b.sum() - ab), 1))
    if metric == "l2":
        return ((r64 - q64) ** 2).sum(1)
    if metric == "ip":
        return -(r64 @ q64)
    if metric == "l1":
        return np.abs(r64 - q64).sum(1)
    return 1.0 - (r64 @ q64) / np.sqrt((r64 ** 2).sum(1) * (q64 ** 2).sum())


@pytest.mark.parametrize("gate", G["recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_reference_recall_gates_on_device(gate):
    """tests/t/012:94, 024:97, 020:102, 028:102 at their full size (10 000 rows, k = 20) through the batched device build and device scan
    (sparsevec: the lock-step driver on the merge-join kernels)."""
    rng = np.random.default_rng(12)
    n, dim, k = gate["rows"], gate["dim"], gate["k"]
    dt = TYPE[gate["type"]]
    rows, qs, r64, q64 = _gate_rows(gate, rng)
    levels = hx.draw_levels(n, gate["m"], seed=12)
    for metric, min_recall in gate["min_recall"].items():
        cosine = metric == "cosine"
        e = hx.Engine(dt, METRIC[metric], dim, n)
        e.append(rows)
        if cosine:
            assert (e.normalize_rows(0, n) > 0).all()
        ix = hx.Index(e, gate["m"], gate["ef_construction"])
        ix.insert(0, levels, batch=128)
        e.set_queries(qs, normalize=cosine)
        tids, _, _, cnt = ix.search(len(qs), gate["ef_search"], k)
        correct = 0
        for q in range(len(qs)):
            dist = _exact(gate, metric, rows, qs[q], r64, None if q64 is None else q64[q])
            kth = np.sort(dist, kind="stable")[k - 1]
            ok = set(np.nonzero(dist <= kth)[0].tolist())                 # ties with the k-th distance count (020:60-66)
            correct += sum(1 for t in tids[q, :cnt[q]].tolist() if t in ok)
        assert correct / (k * len(qs)) >= min_recall, (gate["ref"], metric, correct / (k * len(qs)))
        ix.close()
        e.close()


@pytest.mark.parametrize("gate", G["iterative_recall_gates"], ids=lambda g: g["ref"].split("/")[-1])
def test_iterative_recall_gate_on_device(gate):
    """tests/t/044_hnsw_iterative_scan_recall.pl:111-112 at its full size: 50 000 x vector(3), `i % c = 0` for c = 50 and 500, LIMIT 20,
    strict_order and relaxed_order, L2 and cosine, recall >= 0.99; the scans run in the device kernel (k_fused MODE 2)."""
    rng = np.random.default_rng(44)
    n, dim, limit = gate["rows"], gate["dim"], gate["limit"]
    raw = rng.random((n, dim)).astype(np.float32)
    qs = rng.random((gate["queries"], dim)).astype(np.float32)
    r64 = raw.astype(np.float64)
    levels = hx.draw_levels(n, gate["m"], seed=44)
    tids_in = np.arange(1, n + 1, dtype=np.int64)                         # i = 1..n (generate_series, 044:66)
    for metric in gate["metrics"]:
        cosine = metric == "cosine"
        e = hx.Engine(hx.F32, METRIC[metric], dim, n)
        e.append(raw)
        if cosine:
            assert (e.normalize_rows(0, n) > 0).all()
        ix = hx.Index(e, gate["m"], gate["ef_construction"])
        ix.insert(0, levels, tids=tids_in, batch=512)
        e.set_queries(qs, normalize=cosine)
        for c in gate["filter_mod"]:
            passes = (np.arange(n + 1) % c == 0).astype(np.uint8)
            for mode in gate["modes"]:
                before = ix.fused_stats()
                tids, _, cnt = ix.search_iterative(len(qs), gate["ef_search"], 2 if mode == "strict_order" else 1, 20000, limit, passes)
                assert ix.fused_stats()["redone"] == before["redone"]
                correct = 0
                for q in range(len(qs)):
                    q64 = qs[q].astype(np.float64)
                    dist = np.sqrt(((r64 - q64) ** 2).sum(1)) if metric == "l2" else 1.0 - (r64 @ q64) / np.sqrt((r64 ** 2).sum(1) * (q64 ** 2).sum())
                    ids = np.arange(1, n + 1)
                    top = np.sort(dist[ids % c == 0])[:limit]
                    ok = set(ids[dist <= top[-1]].tolist())
                    got = tids[q, :cnt[q]].tolist()
                    assert all(t % c == 0 for t in got)
                    correct += sum(1 for t in got if t in ok)
                assert correct / (limit * len(qs)) >= gate["min_recall"], (metric, c, mode, correct / (limit * len(qs)))
        ix.close()
        e.close()
