#!/usr/bin/env python3
"""Generates tests/golden/recall_parity_100k.json: recall@10 of the REFERENCE schedule and arithmetic (oracle: strictly sequential
build_callback, ORC_ORDER_SEQ, one row at a time) on 100 000 x vector(768) L2, m=16, ef_construction=200 -- BASELINE configs[1]'s
shape and data distribution at a size one CPU core finishes in under an hour.  tests/test_gpu_recall_parity.py rebuilds the same rows
with the batched device build (batch cap 8192) and demands the same recall within sampling noise.

Run once on a CPU box:  python tools/make_recall_fixture.py [rows [centres]]     (100 000 rows: 10 min of one core, 300 000: ~40 min, 1 000 000 x 1024 centres: hours; commit the JSON it writes)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
import pgvector_rx_amd as hx  # noqa: E402

CFG = {"rows": 100_000, "dim": 768, "m": 16, "ef_construction": 200, "ef_search": [10, 20, 40, 100], "k": 10, "queries": 1000,
       "centres": 96, "sigma": 0.1, "seed_rows": 2026, "seed_queries": 2027, "seed_levels": 2026}


def make_data(cfg):
    """1024-centre Gaussian mixture, sigma 0.1, float32; numpy's default_rng is stable across platforms."""
    rng = np.random.default_rng(cfg["seed_rows"])
    cen = rng.random((cfg["centres"], cfg["dim"]), dtype=np.float32)
    a = rng.integers(0, cfg["centres"], cfg["rows"])
    rows = cen[a] + np.float32(cfg["sigma"]) * rng.standard_normal((cfg["rows"], cfg["dim"]), dtype=np.float32)
    rq = np.random.default_rng(cfg["seed_queries"])
    aq = rq.integers(0, cfg["centres"], cfg["queries"])
    qs = cen[aq] + np.float32(cfg["sigma"]) * rq.standard_normal((cfg["queries"], cfg["dim"]), dtype=np.float32)
    return rows.astype(np.float32), qs.astype(np.float32)


def exact_topk(rows, qs, k):
    r64 = rows.astype(np.float64)
    rn = (r64 * r64).sum(1)
    out = np.empty((len(qs), k), np.int64)
    for i in range(0, len(qs), 100):
        q = qs[i:i + 100].astype(np.float64)
        d = rn[None, :] - 2.0 * q @ r64.T
        out[i:i + 100] = np.argsort(d, axis=1, kind="stable")[:, :k]
    return out


def main():
    cfg = dict(CFG)
    if len(sys.argv) > 1:                                      # python tools/make_recall_fixture.py 300000 -> tests/golden/recall_parity_300k.json
        cfg["rows"] = int(sys.argv[1])
    if len(sys.argv) > 2:                                      # python tools/make_recall_fixture.py 1000000 1024 -> the bench's own size and mixture (1024 centres)
        cfg["centres"] = int(sys.argv[2])
    rows, qs = make_data(cfg)
    levels = hx.draw_levels(cfg["rows"], cfg["m"], seed=cfg["seed_levels"])
    o = orc.Index(orc.F32, orc.L2SQ, cfg["dim"], m=cfg["m"], ef_construction=cfg["ef_construction"], order=orc.SEQ)
    t0 = time.time()
    for i in range(cfg["rows"]):
        o.insert(rows[i], levels[i], i)
        if i % 5000 == 0:
            print("inserted %d rows, %.0f s" % (i, time.time() - t0), flush=True)
    build_s = time.time() - t0
    gt = exact_topk(rows, qs, cfg["k"])
    out = dict(cfg)
    out["oracle"] = "ORC_ORDER_SEQ, orc_index_insert one row at a time (the reference's schedule and summation order)"
    out["oracle_build_seconds_one_core"] = round(build_s, 1)
    out["recall_at_k"] = {}
    for efs in cfg["ef_search"]:
        ids, cnt = o.search_many(qs, efs, cfg["k"], n_threads=8)
        per_q = np.array([len(set(ids[q, :cnt[q]].tolist()) & set(gt[q].tolist())) / cfg["k"] for q in range(len(qs))])
        out["recall_at_k"][str(efs)] = {"mean": float(per_q.mean()), "std_of_mean": float(per_q.std(ddof=1) / np.sqrt(len(per_q))),
                                        "hits_per_query": "".join("%x" % int(round(v * cfg["k"])) for v in per_q)}   # one hex digit per query (k = 10): lets the device test compare query by query
        print("ef_search %d: recall@%d %.4f" % (efs, cfg["k"], per_q.mean()), flush=True)
    out["distance_evaluations"] = {"search": o.counters()[1], "select": o.counters()[2], "backlink": o.counters()[3]}
    path = os.path.join(ROOT, "tests", "golden", "recall_parity_%dk.json" % (cfg["rows"] // 1000))
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
