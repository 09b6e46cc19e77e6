"""aminsert on the engine (hx_index_insert_ondisk): rows/s into an index of `base` rows, per batch (= concurrent backends).  Round 3: the neighbour search of
a batch runs in the traversal kernel (MODE 3, search_layer_disk semantics) and the batch's get_update_index calls run as waves (one lock-step round per
wave instead of one per member).  One untimed batch first (buffers, task pools), then the timed rows.
python tools/bench_ondisk.py [base_rows] [insert_rows] [dim]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

base = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
ins = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 768
m, efc = 16, 64
for b in (10, 256, 2048, 8192):
    n = base + ins + 2 * b
    rows, _ = bench.synth(n, dim, "gmm", 1, "cuda")
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n); e.append_device(rows.data_ptr(), n)
    ix = hx.Index(e, m, efc)
    levels = hx.draw_levels(n, m, seed=1)
    ix.insert(0, levels[:base], batch=32768)
    k = min(ins, max(b * 4, 512))
    ix.insert_ondisk(base, levels[base:base + b], tids=np.arange(base, base + b), batch=b)          # warm-up batch (untimed)
    ix.profile(reset=True)
    t0 = time.perf_counter(); ix.insert_ondisk(base + b, levels[base + b:base + b + k], tids=np.arange(base + b, base + b + k), batch=b); dt = time.perf_counter() - t0
    pr = ix.profile()
    print(json.dumps({"base_rows": base, "dim": dim, "m": m, "ef_construction": efc, "concurrent_inserts": b, "rows": k, "rows_per_s": round(k / dt, 1),
                      "lock_step_rounds": pr["rounds"], "fused_s": round(pr["fused_s"], 4), "round_s": round(pr["round_s"], 4), "advance_s": round(pr["advance_s"], 4),
                      "mirror_sync_s": round(pr["mirror_sync_s"], 4), "members_s": round(pr["links_setup_s"], 4), "updates_s": round(pr["links_lockstep_s"], 4),
                      "total_s": round(pr["insert_total_s"], 4)}), flush=True)
    if b == 2048:                                                                   # VACUUM of every 10th heap TID of the same index: the three passes, repairs `b` at a time
        size = ix.size
        dead = np.arange(0, size, 10, dtype=np.int64)
        f0 = ix.fused_stats(); ix.profile(reset=True)
        t0 = time.perf_counter(); nd, nr = ix.vacuum(dead, batch=b); dt = time.perf_counter() - t0
        f1 = ix.fused_stats(); pr = ix.profile()
        print(json.dumps({"vacuum_of_rows": int(len(dead)), "index_rows": int(size), "deleted": int(nd), "repaired": int(nr), "seconds": round(dt, 3), "repairs_per_s": round(nr / dt, 1),
                          "repair_batch": b, "device_repair_searches": int(f1["tasks"] - f0["tasks"]), "handed_to_lock_step": int(f1["redone"] - f0["redone"]),
                          "lock_step_rounds": pr["rounds"], "fused_s": round(pr["fused_s"], 4)}), flush=True)
        t0 = time.perf_counter(); ix.insert_ondisk(size, levels[size:size + b], tids=np.arange(size, size + b), batch=b); dt = time.perf_counter() - t0
        print(json.dumps({"after_vacuum": True, "concurrent_inserts": b, "rows": b, "rows_per_s": round(b / dt, 1)}), flush=True)
    ix.close(); e.close(); del rows
