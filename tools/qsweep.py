"""QPS of the fused scan kernel vs queries per launch (tail / residency quantisation check).  python tools/qsweep.py [rows]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
dim, m, efc, efs, k = 768, 16, 200, 100, 10
rows, cen = bench.synth(n, dim, "gmm", 1, "cuda")
qs, _ = bench.synth(40_000, dim, "gmm", 2, "cuda", cen)
eng = hx.Engine(hx.F32, hx.L2SQ, dim, n)
eng.append_device(rows.data_ptr(), n)
ix = hx.Index(eng, m, efc)
t0 = time.perf_counter(); ix.insert(0, hx.draw_levels(n, m, seed=1), batch=8192); print("build", round(time.perf_counter() - t0, 2), flush=True)
eng.set_timing(True)
for nq in [int(x) for x in os.environ.get('HX_QSWEEP', '256,1024,3840,7680,10000,20000,40000').split(',')]:
    eng.set_queries_device(qs.data_ptr(), nq)
    ix.search(nq, efs, k); eng.kernel_stats(2, reset=True)
    t0 = time.perf_counter()
    for _ in range(3): ix.search(nq, efs, k)
    dt = (time.perf_counter() - t0) / 3
    st = eng.kernel_stats(2, reset=True)
    print(json.dumps({"nq": nq, "qps": round(nq / dt), "kernel_ms": round(st["ms"] / 3, 3), "wall_ms": round(dt * 1e3, 3),
                      "GBps": round(st["units"] * 3072 / st["ms"] / 1e6, 1)}), flush=True)
