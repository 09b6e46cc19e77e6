"""Latency of small scans: hx_index_search for 1 ... 4096 queries per call on a built index (median / p99 over repeated calls, host clock around the C-ABI call,
queries already resident).  What one backend's `ORDER BY embedding <-> $1 LIMIT 10` pays, as opposed to bench.py's 10 000-query steps.
python tools/bench_latency.py [rows] [dim] [ef_search]"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
efs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rows, _ = bench.synth(n, dim, "gmm", 1, "cuda")
qs, _ = bench.synth(4096, dim, "gmm", 2, "cuda")
e = hx.Engine(hx.F32, hx.L2SQ, dim, n); e.append_device(rows.data_ptr(), n)
ix = hx.Index(e, 16, 200)
t0 = time.perf_counter(); ix.insert(0, hx.draw_levels(n, 16, seed=1), batch=32768); build = time.perf_counter() - t0
e.set_queries_device(qs.data_ptr(), 4096)
for nq in (1, 8, 64, 512, 4096):
    reps = 300 if nq <= 64 else 60
    for _ in range(10):
        ix.search(nq, efs, 10)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); ix.search(nq, efs, 10); ts.append(time.perf_counter() - t0)
    ts = np.sort(np.asarray(ts)) * 1e3
    print(json.dumps({"rows": n, "dim": dim, "ef_search": efs, "queries_per_call": nq, "median_ms": round(float(ts[len(ts) // 2]), 3), "p99_ms": round(float(ts[int(len(ts) * 0.99) - 1]), 3),
                      "min_ms": round(float(ts[0]), 3), "queries_per_s_at_median": round(nq / (ts[len(ts) // 2] * 1e-3), 1), "build_s": round(build, 2)}), flush=True)
ix.close(); e.close()
