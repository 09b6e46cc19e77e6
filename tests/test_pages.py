"""graph -> PostgreSQL HNSW pages.  The reference holds no page fixtures (its page tests assert sizes and
MAXALIGN only, types/hnsw.rs:353-432 -> byte layout 'parity unpinned'), so parity is anchored three ways: the sizes those tests
pin; a reader written from the reference's SCAN-side decoding (tests/pgpages.py) recovering exactly the graph that went in;
and, on the GPU, the engine's writer agreeing byte for byte with the oracle's step-by-step restatement of build.rs:545-821."""
import numpy as np
import pytest

from oracle import orc
from tests import pgpages


def heap_tids(n, first=0):
    """Heap TIDs as (block << 16) | offset with offsets from FirstOffsetNumber = 1 (offset 0 is PostgreSQL's invalid TID)."""
    i = np.arange(first, first + n, dtype=np.int64)
    return ((i // 64) << 16) | (i % 64 + 1)


def make_rows(dtype, dim, n, m, seed, dup_every=0):
    rng = np.random.default_rng(seed)
    if dtype == orc.BIT:
        rows = rng.integers(0, 256, (n, (dim + 7) // 8), dtype=np.uint8)
    elif dtype == orc.F16:
        rows = rng.random((n, dim), dtype=np.float32).astype(np.float16).view(np.uint16)
    else:
        rows = rng.random((n, dim), dtype=np.float32)
    if dup_every:
        for i in range(dup_every, n, dup_every):
            rows[i] = rows[i - 1]
    return rows, orc.levels_from_seed(n, m, seed)


def build_oracle(dtype, metric, dim, n, m, efc, seed, dup_every=0, batch=1):
    rng = np.random.default_rng(seed)
    if dtype == orc.BIT:
        rows = rng.integers(0, 256, (n, (dim + 7) // 8), dtype=np.uint8)
    elif dtype == orc.F16:
        rows = rng.random((n, dim), dtype=np.float32).astype(np.float16).view(np.uint16)   # raw binary16 bits
    else:
        rows = rng.random((n, dim), dtype=np.float32)
    if dup_every:
        for i in range(dup_every, n, dup_every):
            rows[i] = rows[i - 1]
    levels = orc.levels_from_seed(n, m, seed)
    x = orc.Index(dtype, metric, dim, m=m, ef_construction=efc, order=orc.W64)
    x.build(rows, levels, batch=batch)
    return x, rows, levels


def check_against_graph(x, rows, pages, blk, off, m, efc, dim, dtype):
    meta, elements, neigh, chain = pgpages.decode(pages, m)
    assert meta["magic"] == 0xA953A953 and meta["version"] == 1 and meta["dimensions"] == dim            # hnsw_constants.rs:23-27
    assert meta["m"] == m and meta["ef_construction"] == efc and meta["pd_lower"] == 24 + 28
    assert chain == list(range(1, len(pages))) and meta["insert_page"] == len(pages) - 1
    n = x.size
    live = [i for i in range(n) if not x.merged(i)]
    where = {i: (int(blk[i]), int(off[i])) for i in live}
    assert len(set(where.values())) == len(live) == len(elements)
    back = {v: k for k, v in where.items()}
    for i in range(n):
        if x.merged(i):
            assert blk[i] == 0xFFFFFFFF
    if x.entry >= 0:
        assert meta["entry"] == where[x.entry] and meta["entry_level"] == x.level(x.entry)
    else:
        assert meta["entry"] == (0xFFFFFFFF, 0) and meta["entry_level"] == -1
    raw = np.ascontiguousarray(rows)
    for i in live:
        e = elements[where[i]]
        lv = x.level(i)
        assert e["level"] == lv and e["deleted"] == 0 and e["version"] == 0 and e["unused"] == 0
        tids = x.tids(i)
        want = [((t >> 16) & 0xFFFFFFFF, t & 0xFFFF) for t in tids] + [(0xFFFFFFFF, 0)] * (10 - len(tids))
        assert e["heaptids"] == want
        # the value is the type's varlena: vector/halfvec {vl_len_, dim:i16, unused:i16}, bit {vl_len_, bit_len:i32}
        payload = raw[tids[0]].tobytes()      # an element's first heap TID is its own row (default tids = row numbers)
        v = e["value"]
        assert len(v) == 8 + len(payload) and v[8:] == payload
        if dtype == orc.BIT:
            assert int.from_bytes(v[4:8], "little") == dim
        else:
            assert int.from_bytes(v[4:6], "little") == dim and v[6:8] == b"\0\0"
        assert e["tuple_len"] == (72 + len(v) + 7) // 8 * 8                                                 # hnsw_element_tuple_size
        nt = neigh[e["neighbortid"]]
        assert nt["count"] == (lv + 2) * m and nt["version"] == 0 and nt["tuple_len"] == (4 + (lv + 2) * m * 6 + 7) // 8 * 8
        for layer in range(lv + 1):
            ids, _ = x.neighbors(i, layer)
            assert [back[t] for t in pgpages.neighbour_tids(nt, lv, layer, m)] == list(ids)
    return meta, elements, neigh


def test_sizes_pinned_by_the_reference_tests():
    # types/hnsw.rs:353-361 (maxalign), :404-416 (tuple sizes MAXALIGNed), :419-424 (4 KB < max size < BLCKSZ)
    x, rows, _ = build_oracle(orc.F32, orc.L2SQ, 128, 40, 16, 64, 1)
    pages, blk, off = x.write_pages()
    meta, elements, neigh, _ = pgpages.decode(pages, 16)
    assert all(e["tuple_len"] % 8 == 0 and e["tuple_len"] == (72 + 8 + 512 + 7) // 8 * 8 for e in elements.values())
    assert all(nt["tuple_len"] % 8 == 0 for nt in neigh.values())
    assert 8192 - 24 - 8 - 4 == 8156                                                                        # hnsw_max_size


@pytest.mark.parametrize("dtype,metric,dim,n,m,dup,batch", [
    (orc.F32, orc.L2SQ, 8, 400, 4, 0, 1),
    (orc.F32, orc.NEG_IP, 48, 300, 16, 7, 1),         # sequential duplicates: merged heap TIDs, popped rows
    (orc.F32, orc.L2SQ, 16, 300, 8, 5, 32),           # batched schedule: duplicates stay as tombstones without tuples
    (orc.F16, orc.L2SQ, 10, 250, 6, 0, 1),
    (orc.BIT, orc.HAMMING, 52, 300, 5, 0, 1),
])
def test_oracle_pages_decode_back_to_the_graph(dtype, metric, dim, n, m, dup, batch):
    x, rows, _ = build_oracle(dtype, metric, dim, n, m, 24, 3, dup_every=dup, batch=batch)
    pages, blk, off = x.write_pages()
    check_against_graph(x, rows, pages, blk, off, m, 24, dim, dtype)


def test_largest_legal_tuple_puts_its_neighbour_tuple_on_the_next_page():
    # halfvec(4000): 72 + 8008 = 8080 B element tuple; with the neighbour tuple it exceeds hnsw_max_size, so the neighbour
    # tuple is the first item of the following page (build.rs:657-661)
    x, rows, _ = build_oracle(orc.F16, orc.NEG_IP, 4000, 6, 16, 16, 2)
    pages, blk, off = x.write_pages()
    meta, elements, neigh = check_against_graph(x, rows, pages, blk, off, 16, 16, 4000, orc.F16)
    for (b, o), e in elements.items():
        assert o == 1 or o == 2
        assert e["neighbortid"] == (b + 1, 1)


def test_empty_index_has_meta_and_head_page():
    x = orc.Index(orc.F32, orc.L2SQ, 3, m=16, ef_construction=64)
    pages, _, _ = x.write_pages()
    meta, elements, neigh, chain = pgpages.decode(pages, 16)
    assert len(pages) == 2 and chain == [1] and not elements and meta["entry"] == (0xFFFFFFFF, 0) and meta["insert_page"] == 1   # build.rs:583-592


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,n,m,dup,batch", [
    (orc.F32, orc.L2SQ, 8, 500, 4, 0, 16),
    (orc.F32, orc.L2SQ, 16, 400, 8, 5, 32),
    (orc.F32, orc.NEG_IP, 200, 300, 16, 0, 8),
    (orc.F16, orc.L2SQ, 10, 300, 6, 0, 8),
    (orc.F16, orc.NEG_IP, 4000, 12, 16, 0, 1),
    (orc.BIT, orc.HAMMING, 52, 400, 5, 0, 16),
    (orc.BIT, orc.JACCARD, 1024, 200, 16, 3, 16),
])
def test_engine_pages_equal_oracle_pages(dtype, metric, dim, n, m, dup, batch):
    import pgvector_rx_amd as hx
    rows, levels = make_rows(dtype, dim, n, m, 9, dup)
    efc = max(24, 2 * m)                 # the engine enforces ef_construction >= 2 m like the reference (build.rs:856-861)
    # the oracle follows the device's batch schedule (ramp-up rule of hx_index_insert), duplicates staying as tombstones
    x = orc.Index(dtype, metric, dim, m=m, ef_construction=efc, order=orc.W64)
    i = 0
    for b in hx.batch_schedule(0, n, batch):
        x.insert_batch(rows[i:i + b], levels[i:i + b], np.arange(i, i + b, dtype=np.int64))
        i += b
    eng = hx.Engine(dtype, metric, dim, n)
    eng.append(rows)
    ix = hx.Index(eng, m, efc)
    ix.insert(0, levels, batch=batch)
    want, wblk, woff = x.write_pages()
    got, gblk, goff = ix.serialize_pages()
    assert got.shape == want.shape
    assert np.array_equal(gblk, wblk) and np.array_equal(goff, woff)
    assert np.array_equal(got, want)
    check_against_graph(x, rows, got, gblk, goff, m, efc, dim, dtype)


def _same_index(a, b, n):
    assert a.size == b.size == n and a.entry == b.entry
    for i in range(n):
        assert a.level(i) == b.level(i)
        assert list(a.heaptids(i)) == list(b.heaptids(i))
        for l in range(max(a.level(i), -1) + 1):
            ia, da = a.neighbors(i, l)
            ib, db = b.neighbors(i, l)
            assert np.array_equal(ia, ib) and np.array_equal(da.view(np.uint32), db.view(np.uint32))   # recomputed distances: same bits


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,metric,dim,n,m,batch", [
    (orc.F32, orc.L2SQ, 16, 600, 8, 32),
    (orc.F32, orc.NEG_IP, 200, 300, 16, 8),
    (orc.F16, orc.L2SQ, 10, 300, 6, 8),
    (orc.BIT, orc.HAMMING, 52, 400, 5, 16),
])
def test_load_pages_round_trip_then_keep_inserting(dtype, metric, dim, n, m, batch):
    """serialize -> load into a fresh engine gives the same index (ids, heap TIDs, distance bits), the same scans, the same
    pages again; and inserting more rows afterwards gives the same graph as inserting them into the original."""
    import pgvector_rx_amd as hx
    extra = 64
    rows, levels = make_rows(dtype, dim, n + extra, m, 13)
    efc = max(24, 2 * m)
    e1 = hx.Engine(dtype, metric, dim, n + extra); e1.append(rows[:n])
    a = hx.Index(e1, m, efc); a.insert(0, levels[:n], heap_tids(n), batch=batch)
    pages, blk, off = a.serialize_pages()
    e2 = hx.Engine(dtype, metric, dim, n + extra)
    b = hx.Index(e2, m, efc)
    lblk, loff = b.load_pages(pages)
    assert np.array_equal(lblk, blk) and np.array_equal(loff, off)          # no duplicates here: element i == row i on both sides
    _same_index(a, b, n)
    q = rows[n:n + 20]
    e1.set_queries(q); e2.set_queries(q)
    ra, rb = a.search(20, 16, 5), b.search(20, 16, 5)
    for u, v in zip(ra, rb):
        assert np.array_equal(u, v)
    again, _, _ = b.serialize_pages()
    assert np.array_equal(again, pages)
    e1.append(rows[n:]); e2.append(rows[n:])
    a.insert(n, levels[n:], heap_tids(extra, n), batch=16); b.insert(n, levels[n:], heap_tids(extra, n), batch=16)
    _same_index(a, b, n + extra)


@pytest.mark.gpu
def test_load_pages_written_by_the_oracle_with_popped_duplicates_and_a_deleted_element():
    """Pages as the CPU path writes them (sequential build, duplicates merged into heap TIDs and popped): the loaded index
    answers scans exactly as the oracle scans its own graph; an element flagged deleted is skipped like load_element does."""
    import pgvector_rx_amd as hx
    dim, n, m = 12, 300, 6
    rows, levels = make_rows(orc.F32, dim, n, m, 21, dup_every=9)
    x = orc.Index(orc.F32, orc.L2SQ, dim, m=m, ef_construction=24, order=orc.W64)
    ht = heap_tids(n)
    for i in range(n):
        x.insert(rows[i], int(levels[i]), int(ht[i]))
    pages, blk, off = x.write_pages()
    eng = hx.Engine(orc.F32, orc.L2SQ, dim, n)
    ix = hx.Index(eng, m, 24)
    lb, lo = ix.load_pages(pages)
    assert ix.size == x.size < n                                             # merged rows were popped
    rng = np.random.default_rng(5)
    qs = rng.random((15, dim), dtype=np.float32)
    eng.set_queries(qs)
    tids, d, el, cnt = ix.search(15, 24, 6)
    for q in range(15):
        ids, dist = x.search_topk(qs[q], 24, 6)
        # k counts heap tuples, and an element that absorbed duplicates returns several: compare element by element
        got_el = [int(v) for j, v in enumerate(el[q, :cnt[q]]) if j == 0 or v != el[q, j - 1]]
        want_el = [int(v) for j, v in enumerate(ids) if j == 0 or v != ids[j - 1]]      # the oracle's scan also yields one entry per heap tuple
        assert got_el == want_el and cnt[q] == len(ids)
        for j in range(cnt[q]):
            assert int(tids[q, j]) in x.tids(int(el[q, j]))
    # flag one non-entry element deleted in the image: it disappears from the loaded graph
    victim = next(i for i in range(x.size) if i != x.entry and x.level(i) == 0)
    pg = pages.copy()
    its = pgpages.items(bytes(pg[blk[victim]]))
    pg[blk[victim], its[off[victim] - 1][0] + 2] = 1
    eng2 = hx.Engine(orc.F32, orc.L2SQ, dim, n)
    ix2 = hx.Index(eng2, m, 24)
    ix2.load_pages(pg)
    assert ix2.size == x.size - 1


@pytest.mark.gpu
def test_load_pages_rejects_malformed_images():
    import pgvector_rx_amd as hx
    x, rows, _ = build_oracle(orc.F32, orc.L2SQ, 8, 50, 4, 24, 2)
    pages, _, _ = x.write_pages()
    def fresh(dim=8, m=4):
        e = hx.Engine(orc.F32, orc.L2SQ, dim, 64)
        return e, hx.Index(e, m, 24)
    bad = pages.copy(); bad[0, 24] ^= 0xFF                                   # magic
    with pytest.raises(hx.HxError):
        fresh()[1].load_pages(bad)
    with pytest.raises(hx.HxError):
        fresh(dim=9)[1].load_pages(pages)                                    # dimensions
    with pytest.raises(hx.HxError):
        fresh(m=5)[1].load_pages(pages)                                      # m
    with pytest.raises(hx.HxError):
        fresh()[1].load_pages(pages[:-1])                                    # chain leaves the image
    e, ix = fresh()
    ix.load_pages(pages)
    with pytest.raises(hx.HxError):
        ix.load_pages(pages)                                                 # not empty any more


@pytest.mark.gpu
def test_invalidate_drops_elements_whose_tuple_version_changed():
    """f2: a host that rewrites element tuples under a loaded mirror reports (blkno, offno, current version); tuples whose version no
    longer matches the loaded one (scan.rs:262-265, types/hnsw.rs:120) leave the mirror: never returned, never traversed, and the scan
    over the rest equals an exact scan restricted to the survivors to the usual recall."""
    import pgvector_rx_amd as hx
    dim, n, m, efc = 8, 1200, 8, 32
    rows, levels = make_rows(orc.F32, dim, n, m, 33)
    e1 = hx.Engine(orc.F32, orc.L2SQ, dim, n); e1.append(rows)
    a = hx.Index(e1, m, efc); a.insert(0, levels, heap_tids(n), batch=32)
    pages, blk, off = a.serialize_pages()
    e2 = hx.Engine(orc.F32, orc.L2SQ, dim, n)
    b = hx.Index(e2, m, efc)
    lblk, loff = b.load_pages(pages)
    # same versions (0, as built): nothing happens; unknown locations are ignored
    assert b.invalidate(lblk[:50], loff[:50], np.zeros(50, np.uint8)) == 0
    assert b.invalidate([999999], [1], [3]) == 0
    victims = np.unique(np.concatenate([[int(b.entry)], np.arange(5, n, 17)]))          # the entry point among them
    assert b.invalidate(lblk[victims], loff[victims], np.full(len(victims), 1, np.uint8)) == len(victims)
    assert b.invalidate(lblk[victims], loff[victims], np.full(len(victims), 1, np.uint8)) == 0     # already dropped
    vset = set(victims.tolist())
    assert b.entry >= 0 and int(b.entry) not in vset
    for i in range(n):
        if i in vset:
            assert b.deleted(i) == 1 and b.heaptids(i) == []
        for layer in range(max(b.level(i), 0) + 1):
            ids, _ = b.neighbors(i, layer)
            assert not (set(ids.tolist()) & vset) and (i not in vset or len(ids) == 0)
    rng = np.random.default_rng(3)
    qs = rng.random((40, dim), dtype=np.float32)
    e2.set_queries(qs)
    tids, d, el, cnt = b.search(40, 64, 10)
    alive = np.array([i for i in range(n) if i not in vset])
    d2 = ((qs[:, None, :].astype(np.float64) - rows[alive][None, :, :].astype(np.float64)) ** 2).sum(2)
    hit = 0
    for q in range(40):
        got = el[q, :cnt[q]].tolist()
        assert not (set(got) & vset)
        hit += len(set(got) & set(alive[np.argsort(d2[q])[:10]].tolist()))
    assert hit / 400 >= 0.9
    # an index that was built, not loaded, has nothing to invalidate against
    with pytest.raises(hx.HxError):
        a.invalidate([1], [1], [0])
