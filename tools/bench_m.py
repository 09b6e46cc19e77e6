"""Build cost against m (options.rs:203-225 allows m <= 100): seconds, distance evaluations and ns per evaluation for the same rows at several m.
m <= 32 builds entirely in the device kernels; above that the neighbour searches run in the traversal kernel and select / back-links in the lock-step driver.
python tools/bench_m.py [rows] [dim] [ef_construction]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
efc = int(sys.argv[3]) if len(sys.argv) > 3 else 200
rows, _ = bench.synth(n, dim, "gmm", 1, "cuda")
for m in (16, 32, 40, 64, 100):
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n); e.append_device(rows.data_ptr(), n)
    ix = hx.Index(e, m, max(efc, 2 * m))
    levels = hx.draw_levels(n, m, seed=1)
    ix.profile(reset=True)
    t0 = time.perf_counter(); ix.insert(0, levels, batch=32768); dt = time.perf_counter() - t0
    c = ix.counters(); pr = ix.profile(); fs = ix.fused_stats()
    nd = int(c[1]) + int(c[2]) + int(c[3])
    print(json.dumps({"rows": n, "dim": dim, "m": m, "ef_construction": max(efc, 2 * m), "build_s": round(dt, 3), "search_distances": int(c[1]), "select_distances": int(c[2]),
                      "backlink_distances": int(c[3]), "ns_per_distance": round(dt / max(nd, 1) * 1e9, 3), "lock_step_rounds": pr["rounds"],
                      "fused_tasks": int(fs["tasks"]), "fused_redone": int(fs["redone"])}), flush=True)
    ix.close(); e.close()
