"""SURVEY 8 row g / BASELINE configs[3]: the batched-build distance GEMM on the matrix cores (hx_mfma.hip, v_mfma_f32_32x32x16_f16).

The pair blocks of select_neighbors / check_element_closer for halfvec inner product come out of MFMA tiles; products of halves are exact
in f32, so a value differs from the canonical-order value only through the order of the f32 additions, by at most
2 * dim * 2^-24 * |a| |b|.  Decisions inside that band are re-evaluated in the canonical order, so the graph stays the oracle's."""
import numpy as np
import pytest

import pgvector_rx_amd as hx
from oracle import orc
from test_gpu_index import assert_same_graph, build_both

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dim", [64, 1000, 4000, 3, 77])
def test_mfma_pair_blocks_within_the_stated_band_of_the_canonical_order(dim):
    rng = np.random.default_rng(dim)
    n = 400
    rows = ((2.0 * rng.random((n, dim)) * rng.random((n, dim))) * np.where(rng.random((n, dim)) < 0.3, -1.0, 1.0)).astype(np.float16).view(np.uint16)
    e = hx.Engine(hx.F16, hx.NEG_IP, dim, n)
    e.append(rows)
    groups = []
    for na, nb in [(64, 0), (33, 0), (2, 0), (17, 0), (48, 16), (32, 32), (1, 63), (40, 7), (5, 3)]:
        ids = rng.permutation(n)[:na + nb]
        groups.append((ids[:na].tolist(), ids[na:].tolist() if nb else None))
    exact = e.pairwise_many(groups)
    approx, norm2 = e.pairwise_many(groups, mfma=True)
    f = rows.view(np.float16).astype(np.float64)
    worst = 0.0
    pos = 0
    for (a, b), x, y in zip(groups, exact, approx):
        assert x.shape == y.shape
        ids = list(a) + list(b or [])
        nn = norm2[pos:pos + len(ids)]
        pos += len(ids)
        assert np.allclose(nn, (f[ids] ** 2).sum(1), rtol=1e-5)
        if b:
            bound = 2 * dim * 2.0 ** -24 * np.sqrt(np.outer(nn[:len(a)], nn[len(a):]))
            ref = -(f[a] @ f[b].T)
        else:
            bound = np.array([2 * dim * 2.0 ** -24 * np.sqrt(nn[i] * nn[j]) for i in range(len(a)) for j in range(i)])
            ref = np.array([-(f[a[i]] @ f[a[j]]) for i in range(len(a)) for j in range(i)])
        assert (np.abs(x.astype(np.float64) - y.astype(np.float64)) <= bound + 1e-30).all()
        assert np.allclose(y, ref, rtol=1e-4, atol=1e-4)
        worst = max(worst, float((np.abs(x.astype(np.float64) - y) / np.maximum(bound, 1e-30)).max()))
    print("\ndim %d: worst |mfma - canonical| / band = %.3f" % (dim, worst))
    e.close()


@pytest.mark.parametrize("dim,n,m,efc,batch", [(4000, 500, 16, 100, 64), (96, 1500, 8, 32, 37), (40, 900, 20, 48, 50)])
def test_graph_identical_to_oracle_with_the_mfma_path_on(dim, n, m, efc, batch):
    """Lock-step placement, halfvec inner product: select blocks on the matrix cores, in-band decisions re-evaluated exactly."""
    rng = np.random.default_rng(dim + n)
    rows = (2.0 * rng.random((n, dim)) * rng.random((n, dim))).astype(np.float16).view(np.uint16)
    levels = hx.draw_levels(n, m, seed=31)
    e = hx.Engine(hx.F16, hx.NEG_IP, dim, n)
    e.append(rows)
    ix = hx.Index(e, m, efc)
    ix.set_fused(False)
    ix.set_mfma(True)
    ix.insert(0, levels, batch=batch)
    st = ix.mfma_stats()
    assert st["mfma_pairs"] > 0 and st["exact_pairs"] < st["mfma_pairs"]
    print("\nMFMA pairs %d, re-evaluated exactly %d (%.2f %%)" % (st["mfma_pairs"], st["exact_pairs"], 100.0 * st["exact_pairs"] / st["mfma_pairs"]))
    o = orc.Index(orc.F16, orc.NEG_IP, dim, m=m, ef_construction=efc, order=orc.W64)
    i = 0
    for b in hx.batch_schedule(0, n, batch):
        o.insert_batch(rows[i:i + b], levels[i:i + b], np.arange(i, i + b))
        i += b
    assert_same_graph(ix, o, n)
    ix.close()
    e.close()


def test_mfma_is_refused_for_other_types():
    e = hx.Engine(hx.F32, hx.L2SQ, 8, 16)
    ix = hx.Index(e, 8, 32)
    with pytest.raises(hx.HxError):
        ix.set_mfma(True)
    ix.close()
    e.close()


@pytest.mark.parametrize("dim,n,m,efc,batch,signed", [(4000, 700, 16, 200, 128, False), (4000, 500, 16, 200, 64, True), (1000, 1000, 8, 64, 100, False),
                                                      (300, 1200, 32, 256, 200, True), (2000, 600, 16, 40, 50, False)])
def test_graph_identical_to_oracle_with_the_fused_placement_and_mfma_on(dim, n, m, efc, batch, signed):
    """The DEFAULT (device-resident) placement with hx_index_set_mfma(1): k_fused MODE 3 searches, k_wgemm_f16 computes each member's W x W Gram matrix
    on the matrix cores (C4 shape: halfvec(4000), m 16, ef_construction 200 -> 7 x 7 tiles), k_wselect replays select_neighbors on it and re-evaluates
    the in-band decisions in the canonical order.  Lists, distance bits, duplicates and entry point equal the oracle's; nothing falls to the
    lock-step driver.  Signed rows make inner products cancel (values near the thresholds: the band logic is exercised hard)."""
    rng = np.random.default_rng(dim + n)
    vals = 2.0 * rng.random((n, dim)) * rng.random((n, dim))
    if signed:
        vals *= np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0)
    rows = vals.astype(np.float16).view(np.uint16)
    rows[n // 2] = rows[5]
    levels = hx.draw_levels(n, m, seed=31)
    e = hx.Engine(hx.F16, hx.NEG_IP, dim, n)
    e.append(rows)
    e.set_timing(True)
    ix = hx.Index(e, m, efc)
    ix.set_mfma(True)                                   # fused placement stays on (the default)
    ix.insert(0, levels, batch=batch)
    st, ks = ix.mfma_stats(), e.kernel_stats(4)
    assert st["mfma_pairs"] > 0 and ks["launches"] > 0 and ks["units"] > 0        # the GEMM ran
    assert ix.fused_stats()["redone"] == 0 and ix.profile()["rounds"] == 0        # and nothing ran in the lock-step driver
    print("\ndecisions from the matrix %d, pairs re-evaluated in the canonical order %d (%.2f %%); GEMM pairs %d in %.2f ms" % (
        st["mfma_pairs"], st["exact_pairs"], 100.0 * st["exact_pairs"] / st["mfma_pairs"], ks["units"], ks["ms"]))
    o = orc.Index(orc.F16, orc.NEG_IP, dim, m=m, ef_construction=efc, order=orc.W64)
    i = 0
    for b in hx.batch_schedule(0, n, batch):
        o.insert_batch(rows[i:i + b], levels[i:i + b], np.arange(i, i + b))
        i += b
    assert_same_graph(ix, o, n)
    ix.close()
    e.close()
