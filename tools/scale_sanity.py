"""Scale sanity of the round-3 list kernels: builds at m = 32 (k_list_ops on the device mirror) and m = 48 (k_select_w / k_list_ops, host-mastered lists), then
aminserts in batches (k_update_runs / k_update_runs_big), a VACUUM and more aminserts -- recall@10 against brute force after every stage, no lock-step round.
Recall after the VACUUM falls on clustered data at this size: repair_graph_element rewrites the lists of every element that had a deleted neighbour with the lm NEAREST
results of its search, without the diversity heuristic (insert.rs:1103-1117 via vacuum.rs:342) -- the reference's behaviour, reproduced bit for bit (tests/test_gpu_ondisk.py);
at 30 000 rows, where the nearest 64 still span clusters, the same VACUUM leaves recall at 0.996 in every placement and batch size.
python tools/scale_sanity.py [rows] [dim]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
extra, nq, k, efs = 60_000, 1000, 10, 100
n = n0 + extra
rows, _ = bench.synth(n, dim, "gmm", 1, "cuda")
qs, _ = bench.synth(nq, dim, "gmm", 2, "cuda")


def recall(ix, e, live):
    e.set_queries_device(qs.data_ptr(), nq)
    tids, d, el, cnt = ix.search(nq, efs, k)
    idx = torch.nonzero(live).squeeze(1)
    dist = torch.cdist(qs, rows[idx])
    gt = idx[torch.topk(dist, k, dim=1, largest=False).indices].cpu().numpy()
    hits = sum(len(set(gt[q].tolist()) & set(tids[q, :cnt[q]].tolist())) for q in range(nq))
    return round(hits / (nq * k), 4)


for m in (32, 48):
    efc = 128
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n); e.append_device(rows.data_ptr(), n)
    ix = hx.Index(e, m, efc)
    levels = hx.draw_levels(n, m, seed=1)
    live = torch.zeros(n, dtype=torch.bool, device="cuda")
    out = {"m": m, "ef_construction": efc, "dim": dim}
    t0 = time.perf_counter(); ix.insert(0, levels[:n0], batch=32768); out["build_rows"] = n0; out["build_s"] = round(time.perf_counter() - t0, 2)
    live[:n0] = True
    out["recall_after_build"] = recall(ix, e, live)
    a = n0
    t0 = time.perf_counter(); ix.insert_ondisk(a, levels[a:a + 40_000], tids=np.arange(a, a + 40_000), batch=2048); out["aminsert_rows_per_s"] = round(40_000 / (time.perf_counter() - t0), 1)
    a += 40_000; live[:a] = True
    out["recall_after_aminsert"] = recall(ix, e, live)
    dead = np.arange(0, a, 20, dtype=np.int64)
    t0 = time.perf_counter(); nd, nr = ix.vacuum(dead, batch=2048); out["vacuum_s"] = round(time.perf_counter() - t0, 2); out["deleted"] = int(nd); out["repaired"] = int(nr)
    live[torch.from_numpy(dead).cuda()] = False
    out["recall_after_vacuum"] = recall(ix, e, live)
    t0 = time.perf_counter(); ix.insert_ondisk(a, levels[a:n], tids=np.arange(a, n), batch=2048); out["aminsert_after_vacuum_rows_per_s"] = round((n - a) / (time.perf_counter() - t0), 1)
    live[a:n] = True
    out["recall_after_second_aminsert"] = recall(ix, e, live)
    out["update_kernel_launches_and_lock_step_rounds"] = ix.profile()["rounds"]; out["fused_redone"] = ix.fused_stats()["redone"]
    print(json.dumps(out), flush=True)
    ix.close(); e.close()
