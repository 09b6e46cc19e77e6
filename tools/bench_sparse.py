"""sparsevec on the engine: build + scan timing on one MI355X (round 3: searches and scans in the traversal kernel, hx_fused_sparse.hip; select and back-links in the
list kernels of csrc/hx_biglist.hip; hx_index_set_fused(0): the lock-step driver over the merge-join kernels of csrc/hx_sparse.hip).
python tools/bench_sparse.py [rows] [dim] [max_nnz]"""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import pgvector_rx_amd as hx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rng = np.random.default_rng(1)
m, efc, efs, k, nq = 16, 64, 40, 10, 1000
# topic-like data: each row draws most of its indices from one of 256 "topics" (so that neighbours exist)
topics = [np.sort(rng.choice(dim, 4 * mx, replace=False)) for _ in range(256)]
def draw(cnt):
    out = []
    for _ in range(cnt):
        t = topics[rng.integers(0, 256)]
        kk = int(rng.integers(mx // 2, mx + 1))
        idx = np.unique(np.concatenate([rng.choice(t, kk - kk // 8, replace=False), rng.choice(dim, kk // 8, replace=False)])).astype(np.int32)
        out.append((idx, np.abs(rng.standard_normal(len(idx))).astype(np.float32) + 0.05))
    return out
rows, qs = draw(n), draw(nq)
rec, qrec = hx.pack_sparse(dim, rows), hx.pack_sparse(dim, qs)
e = hx.Engine(hx.SPARSE, hx.NEG_IP, dim, n)
e.append(rec)
e.normalize_rows(0, n)
ix = hx.Index(e, m, efc)
ix.profile(reset=True)
t0 = time.perf_counter(); ix.insert(0, hx.draw_levels(n, m, seed=1), batch=4096); build = time.perf_counter() - t0
pr = ix.profile(); cn = ix.counters()
e.set_queries(qrec, normalize=True)
ix.search(nq, efs, k)
t0 = time.perf_counter(); tids, d, el, cnt = ix.search(nq, efs, k); dt = time.perf_counter() - t0
# exact top-k through the engine's own distance kernel (brute force over all rows, 64 queries)
hits = 0
ids = np.arange(n, dtype=np.uint32)
for q in range(64):
    dq = e.distances_batch(np.array([hx.QUERY_SLOT | q], np.uint32), np.array([0, n], np.uint32), ids)
    top = set(np.argsort(dq, kind="stable")[:k].tolist())
    hits += len(top & set(tids[q, :cnt[q]].tolist()))
print(json.dumps({"rows": n, "dim": dim, "max_nnz": mx, "record_bytes": int(rec.shape[1]), "metric": "cosine (normalised, negative inner product)", "m": m, "ef_construction": efc,
                  "build_sec": round(build, 2), "build_profile": {k2: (round(v, 3) if isinstance(v, float) else v) for k2, v in pr.items()},
                  "build_distance_evals": {"search": int(cn[1]), "select": int(cn[2]), "backlink": int(cn[3])}, "qps": round(nq / dt, 1), "ef_search": efs, "recall_at_10": round(hits / (64 * k), 4), "path": "traversal kernel (k_fused<OpSparse>: scans, build searches); select + back-links: k_select_w / k_list_ops (hx_biglist.hip)" if ix.fused_stats()["tasks"] else "lock-step driver, k_sparse_groups / k_sparse_pairs"}))
