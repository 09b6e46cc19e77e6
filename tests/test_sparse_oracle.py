"""CPU checks of the sparsevec restatement in the oracle (src/types/sparsevec.rs:873-950, 1038-1088, 1123-1178) beyond the reference's known answers
(tests/test_oracle_golden.py): on random rows the merge joins must give the bits of the dense sequential loops on the same vectors written densely
(the terms are the same and arrive in index order; an absent element contributes x - 0 = x exactly), they must not depend on the side, and
normalisation must equal the dense one on the stored elements.  Also the record packer shared by the engine's binding and the oracle's."""
import numpy as np
import pytest

import pgvector_rx_amd as hx
from oracle import orc


def rand_sparse(rng, dim, max_nnz):
    k = int(rng.integers(0, min(dim, max_nnz) + 1))
    idx = np.sort(rng.choice(dim, k, replace=False)).astype(np.int32)
    val = rng.standard_normal(k).astype(np.float32)
    val[val == 0] = 1.0
    return idx, val


@pytest.mark.parametrize("dim,max_nnz", [(3, 3), (40, 9), (5000, 60), (1000000, 1000)])
def test_sparse_distances_equal_dense_sequential_loops(dim, max_nnz):
    rng = np.random.default_rng(dim)
    rows = [rand_sparse(rng, dim, max_nnz) for _ in range(24)]
    rows[0] = (np.zeros(0, np.int32), np.zeros(0, np.float32))
    rec = hx.pack_sparse(dim, rows)
    assert rec.shape[1] == hx.sparse_record_bytes(dim) == orc.lib().orc_row_bytes(orc.SPARSE, dim)
    for i, (idx, val) in enumerate(rows):
        assert np.array_equal(rec[i], orc.pack_sparse(dim, idx, val))
    dense_ok = dim <= 5000
    if dense_ok:
        dense = np.zeros((len(rows), dim), np.float32)
        for i, (idx, val) in enumerate(rows):
            dense[i, idx] = val
    for metric in (orc.L2SQ, orc.NEG_IP, orc.L1):
        for a in range(len(rows)):
            for b in range(len(rows)):
                ds = orc.distance(orc.SPARSE, metric, dim, rec[a], rec[b])
                assert ds == orc.distance(orc.SPARSE, metric, dim, rec[b], rec[a])
                if dense_ok:
                    assert ds == orc.distance(orc.F32, metric, dim, dense[a], dense[b], orc.SEQ), (metric, a, b)


def test_sparse_normalize_equals_dense_on_the_stored_elements():
    rng = np.random.default_rng(5)
    dim = 300
    for _ in range(50):
        idx, val = rand_sparse(rng, dim, 40)
        rec = orc.pack_sparse(dim, idx, val)
        out, norm = orc.l2_normalize(orc.SPARSE, dim, rec)
        dense = np.zeros(dim, np.float32); dense[idx] = val
        dn, dnorm = orc.l2_normalize(orc.F32, dim, dense)
        assert norm == dnorm
        nz = np.nonzero(dn)[0]
        assert np.array_equal(out, orc.pack_sparse(dim, nz, dn[nz]))


def test_pack_sparse_rejects_bad_rows():
    with pytest.raises(AssertionError):
        hx.pack_sparse(10, [(np.array([3, 2]), np.array([1.0, 2.0]))])          # indices must ascend
    with pytest.raises(AssertionError):
        hx.pack_sparse(10, [(np.array([10]), np.array([1.0]))])                 # 0-based, < dim
    with pytest.raises(AssertionError):
        hx.pack_sparse(2000, [(np.arange(1001), np.ones(1001))])                # at most 1000 non-zero elements in an indexed sparsevec
