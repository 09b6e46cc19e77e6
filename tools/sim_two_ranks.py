"""Stage timing of the device-batch protocol with `world` engines in ONE process on one GPU (the simulation of tests/test_gpu_index.py at bench
sizes): how long do search / links / export / import take per rank when the ranks do not contend for the GPU?
    python tools/sim_two_ranks.py [rows] [world] [wtabs: 1 = the ranks exchange the members' W tables before the links stage (the default of dist_build), 0 = not]
Prints one JSON line at the end: per-stage seconds (max over ranks) and the single-engine stage seconds for the same schedule (world = 1 run first)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import pgvector_rx_amd as hx
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wtabs = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
dim, m, efc, cap = 768, 16, 200, 32768 * world
rows, _ = bench.synth(n, dim, "gmm", 1, "cuda")
levels = hx.draw_levels(n, m, seed=1)
tids = np.arange(n, dtype=np.int64)
ranks = []
for r in range(world):
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append_device(rows.data_ptr(), n)
    e.set_timing(True)
    ranks.append((e, hx.Index(e, m, efc)))
rb, lb = ranks[0][1].dbatch_record_bytes, ranks[0][1].dbatch_list_record_bytes
T = {k: [0.0] * world for k in ("begin", "search", "wtabs", "links", "export", "import", "end")}
done = 0
for b in hx.batch_schedule(0, n, cap):
    lv, td = levels[done:done + b], tids[done:done + b]
    if ranks[0][1].entry < 0 or b < 16 or not ranks[0][1].dbatch_supported(lv):
        for _, ix in ranks:
            ix.insert(done, lv, td, batch=b)
        done += b
        continue
    per = -(-b // world)
    recs = torch.zeros(world * per * rb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    def tm(key, r, f):
        t0 = time.perf_counter(); v = f(); torch.cuda.synchronize(); T[key][r] += time.perf_counter() - t0; return v
    for r, (_, ix) in enumerate(ranks):
        tm("begin", r, lambda: ix.dbatch_begin(done, lv, td))
        lo, hi = min(b, r * per), min(b, r * per + per)
        tm("search", r, lambda: ix.dbatch_search(lo, hi, recs.data_ptr() + lo * rb))
    wb = ranks[0][1].dbatch_wtab_bytes if (wtabs and world > 1) else 0
    if wb:
        wbuf = torch.zeros(world * per * wb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for r, (_, ix) in enumerate(ranks):
            lo, hi = min(b, r * per), min(b, r * per + per)
            tm("wtabs", r, lambda: ix.dbatch_export_wtabs(lo, hi, wbuf.data_ptr() + r * per * wb))
        for r, (_, ix) in enumerate(ranks):
            for s in range(world):
                lo, hi = min(b, s * per), min(b, s * per + per)
                if s != r and hi > lo:
                    tm("wtabs", r, lambda: ix.dbatch_import_wtabs(lo, hi, wbuf.data_ptr() + s * per * wb))
    counts = [tm("links", r, lambda: ix.dbatch_links(r, world, recs.data_ptr())) for r, (_, ix) in enumerate(ranks)]
    bufs = []
    for r, (_, ix) in enumerate(ranks):
        t = torch.zeros(max(counts[r], 1) * lb, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        tm("export", r, lambda: ix.dbatch_export_links(t.data_ptr()))
        bufs.append(t)
    for r, (_, ix) in enumerate(ranks):
        for s in range(world):
            if s != r and counts[s]:
                tm("import", r, lambda: ix.dbatch_import_links(bufs[s].data_ptr(), counts[s]))
        tm("end", r, lambda: ix.dbatch_end(b))
    done += b
for k, v in T.items():
    print(k, [round(x, 3) for x in v])
for r, (e, ix) in enumerate(ranks):
    print("rank", r, "fused", e.kernel_stats(2), "links", e.kernel_stats(3), {k: round(v, 3) for k, v in ix.profile().items() if v})
print(json.dumps({"rows": n, "world": world, "wtabs_exchanged": bool(wtabs and world > 1), "batch_cap": cap,
                  "stage_seconds_max_over_ranks": {k: round(max(v), 3) for k, v in T.items()},
                  "kernel_ms_rank0": {"fused_insert": round(ranks[0][0].kernel_stats(2)["ms"], 1), "links": round(ranks[0][0].kernel_stats(3)["ms"], 1)}}))
