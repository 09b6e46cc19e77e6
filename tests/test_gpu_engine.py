"""GPU parity of the distance kernels (through the C ABI) against the CPU oracle.

Bar: BIT-EXACT against the oracle evaluated in the device's canonical summation order (ORC_ORDER_W64),
and within the north-star tolerance against the reference's own order (ORC_ORDER_SEQ):
1e-5 relative for L2/L1, 1e-5 of sum|a_i*b_i| for inner product; integer metrics exact."""
import numpy as np
import pytest

import pgvector_rx_amd as hx
from oracle import orc

pytestmark = pytest.mark.gpu

FMETRICS = [hx.L2SQ, hx.NEG_IP, hx.L1]


def make_rows(dtype, n, dim, rng, scale=1.0):
    if dtype == hx.F32:
        return (rng.standard_normal((n, dim)) * scale).astype(np.float32)
    if dtype == hx.F16:
        return (rng.standard_normal((n, dim)) * scale).astype(np.float16).view(np.uint16)
    return np.packbits(rng.integers(0, 2, (n, dim)).astype(np.uint8), axis=1, bitorder="big")


def bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def check_tolerance(dtype, metric, dim, q, rows, ids, got):
    seq = orc.distances_many(dtype, metric, dim, q, rows, ids, orc.SEQ)
    if metric in (hx.HAMMING, hx.JACCARD):
        assert (np.float32(seq) == got).all()
        return
    if metric == hx.NEG_IP:
        qf = q.astype(np.float64) if dtype == hx.F32 else q.view(np.float16).astype(np.float64)
        rf = rows[ids].astype(np.float64) if dtype == hx.F32 else rows[ids].view(np.float16).astype(np.float64)
        scale = np.abs(rf * qf[None, :]).sum(1)
    else:
        scale = np.abs(seq)
    assert (np.abs(got.astype(np.float64) - seq) <= 1e-5 * scale + 1e-30).all()


@pytest.mark.parametrize("dtype,dims", [(hx.F32, [1, 3, 16, 100, 128, 129, 255, 256, 257, 768, 1536, 2000]),
                                        (hx.F16, [1, 10, 128, 513, 1000, 4000])])
@pytest.mark.parametrize("metric", FMETRICS)
def test_query_vs_rows_float(dtype, dims, metric):
    rng = np.random.default_rng(100 + metric)
    for dim in dims:
        n = 300
        rows = make_rows(dtype, n, dim, rng)
        q = make_rows(dtype, 1, dim, rng)[0]
        e = hx.Engine(dtype, metric, dim, n + 8)
        assert e.append(rows) == 0
        ids = rng.integers(0, n, 257).astype(np.uint32)
        got = e.distances(q, ids)
        want = np.float32(orc.distances_many(dtype, metric, dim, q, rows, ids, orc.W64))
        assert (bits(got) == bits(want)).all(), (dtype, metric, dim)
        check_tolerance(dtype, metric, dim, q, rows, ids, got)
        e.close()


@pytest.mark.parametrize("metric", [hx.HAMMING, hx.JACCARD])
def test_query_vs_rows_bit(metric):
    rng = np.random.default_rng(7)
    for dim in [3, 8, 52, 128, 1000, 1024, 4097, 64000]:
        n = 200
        rows = make_rows(hx.BIT, n, dim, rng)
        rows[5] = 0                       # all-zero row: Jaccard ab == 0 -> 1.0 (bitvec.rs:127-128)
        q = make_rows(hx.BIT, 1, dim, rng)[0]
        e = hx.Engine(hx.BIT, metric, dim, n)
        e.append(rows)
        ids = np.arange(n, dtype=np.uint32)
        got = e.distances(q, ids)
        want = np.float32(orc.distances_many(hx.BIT, metric, dim, q, rows, ids, orc.SEQ))
        assert (bits(got) == bits(want)).all(), (metric, dim)
        e.close()


def test_lockstep_groups_and_query_sources():
    """hx_distances_batch: ragged groups (empty, 1 row, > 2M rows), queries by row id and by slot."""
    rng = np.random.default_rng(3)
    dim, n = 96, 500
    rows = make_rows(hx.F32, n, dim, rng)
    qs = make_rows(hx.F32, 7, dim, rng)
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append(rows)
    e.set_queries(qs)
    sizes = [0, 1, 32, 5, 0, 77, 2, 31, 33, 64]
    gq, off, ids = [], [0], []
    for g, s in enumerate(sizes):
        gq.append((hx.QUERY_SLOT | (g % 7)) if g % 2 else int(rng.integers(0, n)))
        ids.extend(rng.integers(0, n, s).tolist())
        off.append(len(ids))
    got = e.distances_batch(gq, off, ids)
    for g, s in enumerate(sizes):
        q = qs[gq[g] & 0x7FFFFFFF] if gq[g] & hx.QUERY_SLOT else rows[gq[g]]
        want = np.float32(orc.distances_many(hx.F32, hx.L2SQ, dim, q, rows, ids[off[g]:off[g + 1]], orc.W64))
        assert (bits(got[off[g]:off[g + 1]]) == bits(want)).all()
    assert len(e.distances(qs[0], [])) == 0
    with pytest.raises(hx.HxError):
        e.distances(qs[0], [n])           # row id out of range
    e.close()


@pytest.mark.parametrize("dtype,metric,dim", [(hx.F32, hx.L2SQ, 768), (hx.F32, hx.NEG_IP, 100), (hx.F32, hx.L1, 1536),
                                             (hx.F16, hx.NEG_IP, 4000), (hx.F16, hx.L2SQ, 64),
                                             (hx.BIT, hx.HAMMING, 1024), (hx.BIT, hx.JACCARD, 52)])
def test_pairwise_matches_oracle_and_query_kernel(dtype, metric, dim):
    rng = np.random.default_rng(11)
    n = 200
    rows = make_rows(dtype, n, dim, rng)
    rows[17] = rows[3]                    # an exact duplicate pair: distance must be exactly 0 for L2/L1/Hamming
    e = hx.Engine(dtype, metric, dim, n)
    e.append(rows)
    order = orc.W64
    for w in [1, 2, 33, 64, 70, 130]:
        ids = rng.permutation(n)[:w].astype(np.uint32)
        if w >= 33:
            ids[0], ids[1] = 3, 17
        got = e.pairwise(ids)
        want = np.float32(orc.pairwise(dtype, metric, dim, rows, ids, order))
        assert (bits(got) == bits(want)).all(), (w,)
        # the same pair through the query-vs-rows kernel gives the same bits (one canonical order)
        k1 = e.distances_batch([int(ids[0])], [0, w], ids)
        assert (bits(k1) == bits(got[0])).all()
    if metric in (hx.L2SQ, hx.L1, hx.HAMMING):
        assert e.pairwise([3, 17])[0, 1] == 0.0
    e.close()


def test_pairwise_many_triangles_and_rectangles():
    rng = np.random.default_rng(21)
    dim, n = 300, 400
    rows = make_rows(hx.F32, n, dim, rng)
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append(rows)
    groups = []
    for na, nb in [(33, 0), (2, 0), (64, 0), (1, 5), (32, 32), (17, 0), (48, 0), (32, 31), (5, 59), (40, 24)]:
        ids = rng.permutation(n)[:na + nb]
        groups.append((ids[:na].tolist(), ids[na:].tolist() if nb else None))
    res = e.pairwise_many(groups)
    for (a, b), r in zip(groups, res):
        if b is None:
            full = np.float32(orc.pairwise(hx.F32, hx.L2SQ, dim, rows, a, orc.W64))
            want = np.array([full[i, j] for i in range(len(a)) for j in range(i)], np.float32)
        else:
            full = np.float32(orc.pairwise(hx.F32, hx.L2SQ, dim, rows, a + b, orc.W64))
            want = full[:len(a), len(a):].reshape(-1)
        assert (bits(r.reshape(-1)) == bits(want)).all()
    with pytest.raises(hx.HxError):
        e.pairwise_many([(list(range(65)), None)])     # > HX_PAIR_MAX_ROWS
    e.close()


@pytest.mark.parametrize("dtype,dim", [(hx.F32, 3), (hx.F32, 1536), (hx.F16, 10), (hx.F16, 4000)])
def test_normalize_rows_bit_exact(dtype, dim):
    """hx_normalize_rows == l2_normalize_raw (vector.rs:106-126 / halfvec.rs:204-233), f64 norm."""
    rng = np.random.default_rng(5)
    n = 70
    rows = make_rows(dtype, n, dim, rng, scale=3.0)
    rows[4] = 0                            # zero-norm row (build.rs:433-435 skips it)
    e = hx.Engine(dtype, hx.NEG_IP, dim, n)
    e.append(rows)
    norms = e.normalize_rows(0, n)
    got = e.read_rows(0, n)
    for i in range(n):
        want, wn = orc.l2_normalize(dtype, dim, rows[i])
        assert norms[i] == wn
        assert (got[i].view(np.uint8) == want.view(np.uint8)).all(), i
    assert norms[4] == 0.0
    # normalised query upload (scan.rs:749-751) gives the same bits
    e.set_queries(rows[:3], normalize=True)
    d = e.distances_batch([hx.QUERY_SLOT | 1], [0, 1], [1])
    w = np.float32(orc.distance(dtype, hx.NEG_IP, dim, got[1], got[1], orc.W64))
    assert bits(d)[0] == bits(w)
    e.close()


def test_rows_equal_append_pop_read():
    rng = np.random.default_rng(9)
    dim = 130
    rows = make_rows(hx.F32, 50, dim, rng)
    rows[10] = rows[20]
    rows[30] = rows[20]
    rows[30, 129] = np.nextafter(rows[30, 129], np.float32(9))    # differs in the last element only
    e = hx.Engine(hx.F32, hx.L2SQ, dim, 64)
    assert e.append(rows[:25]) == 0 and e.append(rows[25:]) == 25
    assert e.rows_equal([10, 10, 20, 0], [20, 11, 30, 0]).tolist() == [True, False, False, True]
    assert (e.read_rows(0, 50) == rows).all()
    e.pop(1)                                                      # build.rs:507-509
    assert e.num_rows == 49 and e.append(rows[:1]) == 49
    with pytest.raises(hx.HxError):
        e.append(rows[:20])                                       # capacity
    e.close()


def test_full_size_row_properties():
    """BASELINE sizes (d=768 / d=1536, 100k rows): size-independent properties instead of the oracle --
    d(x,x)=0, symmetry through both kernels, and agreement of every lane layout with a float64 sum."""
    rng = np.random.default_rng(1)
    for dim in (768, 1536):
        n = 100_000
        rows = rng.random((n, dim), dtype=np.float32)
        e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
        e.append(rows)
        ids = rng.integers(0, n, 4096).astype(np.uint32)
        qi = int(ids[0])
        d = e.distances_batch([qi], [0, len(ids)], ids)
        assert d[0] == 0.0
        ref = ((rows[ids].astype(np.float64) - rows[qi].astype(np.float64)) ** 2).sum(1)
        assert (np.abs(d - ref) <= 1e-5 * ref).all()
        back = e.distances_batch(ids[:64].tolist(), list(range(65)), [qi] * 64)   # d(b,a)
        assert (bits(back) == bits(d[:64])).all()
        e.close()
