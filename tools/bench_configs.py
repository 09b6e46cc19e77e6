"""The other BASELINE.json configs at single-GPU sizes (bench.py is the contract bench for configs[1]).

    python tools/bench_configs.py c1|c3|c4|c5 [rows]

c1: 10k x vector(128) L2, m16 efc64 efs40                      (configs[0], also what the CPU oracle can run in full)
c3: N x vector(1536) cosine (N(0,1) rows, f64-normalised on the device), m16 efc200      (configs[2] shape, 1 GPU)
c4: N x halfvec(4000) inner product (2*U*U rounded to f16), m16 efc200                   (configs[3] shape, 1 GPU)
c5: N x bit(1024) Hamming, plain scan + iterative_scan=relaxed_order with a 1 % filter, max_scan_tuples=20000 (configs[4] shape)
Prints one JSON line: build seconds, QPS, recall@10 vs exact brute force (torch), kernel roofline figures.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import pgvector_rx_amd as hx  # noqa: E402

DEV = "cuda"


def recall(tids, cnt, gt, k):
    return sum(len(set(tids[q, :cnt[q]].tolist()) & set(gt[q].tolist())) for q in range(gt.shape[0])) / (gt.shape[0] * k)


def topk_chunks(score_fn, nq, n_rows, k, largest):
    out = []
    step = max(1, int(1.5e9 // (n_rows * 4)))
    for i in range(0, nq, step):
        s = score_fn(i, min(nq, i + step))
        out.append(torch.topk(s, k, dim=1, largest=largest).indices)
    return torch.cat(out).cpu().numpy()


def clustered(g, n, dim, centers, sigma):
    c = torch.randn((centers, dim), generator=g, device=DEV)
    out = torch.empty((n, dim), device=DEV)
    for i in range(0, n, 1 << 18):
        j = min(n, i + (1 << 18))
        out[i:j] = c[torch.randint(0, centers, (j - i,), generator=g, device=DEV)] + sigma * torch.randn((j - i, dim), generator=g, device=DEV)
    return out


def run(name, n, data="baseline"):
    g = torch.Generator(device=DEV)
    g.manual_seed(21)
    k, nq, batch = 10, 10_000, int(os.environ.get("HX_BATCH_CAP", "32768"))   # the bench's insert batch cap (a batch is also at most 1/8 of the graph)
    iterative = None
    if name == "c1":
        n = n or 10_000
        dim, m, efc, efs, dt, mt, nq = 128, 16, 64, 40, hx.F32, hx.L2SQ, 1000
        rows = torch.rand((n, dim), generator=g, device=DEV)
        qs = torch.rand((nq, dim), generator=g, device=DEV)
        gt_fn = lambda: topk_chunks(lambda a, b: (rows * rows).sum(1)[None, :] - 2.0 * qs[a:b] @ rows.T, nq, n, k, False)
        normalize = False
    elif name == "c3":
        n = n or 1_000_000
        dim, m, efc, efs, dt, mt = 1536, 16, 200, 100, hx.F32, hx.NEG_IP
        if data == "clustered":
            both = clustered(g, n + nq, dim, 2000, 0.6)
            rows, qs = both[:n].contiguous(), both[n:].contiguous()
        else:
            rows = torch.randn((n, dim), generator=g, device=DEV)
            qs = torch.randn((nq, dim), generator=g, device=DEV)
        normalize = True
        gt_fn = None       # set after normalisation (cosine == -IP on unit vectors, vector.rs:852-856)
    elif name == "c4":
        n = n or 500_000
        dim, m, efc, efs, dt, mt = 4000, 16, 200, 100, hx.F16, hx.NEG_IP
        rows = (2.0 * torch.rand((n, dim), generator=g, device=DEV) * torch.rand((n, dim), generator=g, device=DEV)).to(torch.float16)
        qs = (2.0 * torch.rand((nq, dim), generator=g, device=DEV) * torch.rand((nq, dim), generator=g, device=DEV)).to(torch.float16)
        normalize = False
        gt_fn = lambda: topk_chunks(lambda a, b: qs[a:b].float() @ rows.float().T if n <= 200_000 else (qs[a:b] @ rows.T).float(), nq, n, k, True)
    elif name == "c5":
        n = n or 2_000_000
        dim, m, efc, efs, dt, mt, nq = 1024, 16, 64, 40, hx.BIT, hx.HAMMING, int(os.environ.get("HX_C5_QUERIES", "2000"))
        rows = torch.randint(0, 256, (n, dim // 8), generator=g, device=DEV, dtype=torch.uint8)
        qs = torch.randint(0, 256, (nq, dim // 8), generator=g, device=DEV, dtype=torch.uint8)
        if data == "clustered":        # 4096 random centre patterns, every bit flipped with probability 1/8 (AND of three random bytes)
            cen = torch.randint(0, 256, (4096, dim // 8), generator=g, device=DEV, dtype=torch.uint8)
            def noisy(x):
                f = x.clone()
                for _ in range(2):
                    f &= torch.randint(0, 256, x.shape, generator=g, device=DEV, dtype=torch.uint8)
                return cen[torch.randint(0, 4096, (x.shape[0],), generator=g, device=DEV)] ^ f
            rows, qs = noisy(rows), noisy(qs)
        normalize = False
        iterative = {"mode": 1, "max_scan_tuples": 20000, "limit": 10, "filter_every": 100}

        def bits_pm1(x):
            sh = torch.arange(7, -1, -1, device=DEV, dtype=torch.uint8)
            return (((x[:, :, None] >> sh) & 1).reshape(x.shape[0], -1).to(torch.float16) * 2 - 1)
        rpm = bits_pm1(rows)
        gt_fn = lambda: topk_chunks(lambda a, b: (bits_pm1(qs[a:b]) @ rpm.T).float(), nq, n, k, True)   # larger dot = smaller Hamming
    else:
        raise SystemExit(__doc__)
    torch.cuda.synchronize()
    eng = hx.Engine(dt, mt, dim, n)
    eng.append_device(rows.data_ptr(), n)
    skipped = 0
    if normalize:
        norms = eng.normalize_rows(0, n)
        skipped = int((norms == 0).sum())
        if n <= 200_000:
            rown = torch.from_numpy(eng.read_rows(0, n)).to(DEV)
        else:                       # in place, chunk by chunk (at the full C3 size a second copy would not fit next to the engine's)
            for i in range(0, n, 1 << 18):
                j = min(n, i + (1 << 18))
                rows[i:j] = torch.nn.functional.normalize(rows[i:j].double(), dim=1).float()
            rown = rows
        gt_fn = lambda: topk_chunks(lambda a, b: qs[a:b] @ rown.T, nq, n, k, True)
    levels = hx.draw_levels(n, m, seed=21)
    ix = hx.Index(eng, m, efc)
    eng.set_timing(True)
    mfma_build = name == "c4" and os.environ.get("HX_MFMA", "1") != "0"     # configs[3]: select_neighbors of the build on the matrix cores (HX_MFMA=0: VALU select inside k_fused)
    if name == "c4":
        ix.set_mfma(mfma_build)                                             # (on by default for halfvec inner product)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix.insert(0, levels, batch=batch)
    build = time.perf_counter() - t0
    bstats = {"fused_insert": eng.kernel_stats(2, reset=True), "links": eng.kernel_stats(3, reset=True), "gemm": eng.kernel_stats(4, reset=True), "wsel": eng.kernel_stats(6, reset=True)}
    eng.set_queries_device(qs.data_ptr(), nq, normalize=normalize)
    ix.search(nq, efs, k)
    eng.kernel_stats(2, reset=True)
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        tids, d, _, cnt = ix.search(nq, efs, k)
    dt_s = time.perf_counter() - t0
    fq = eng.kernel_stats(2)
    gt = gt_fn()
    row_bytes = eng.row_bytes
    out = {"config": name, "data": data, "rows": n, "dim": dim, "dtype": {0: "f32", 1: "f16", 2: "bit"}[dt], "metric": {0: "l2", 1: "neg_ip", 3: "hamming"}[mt],
           "m": m, "ef_construction": efc, "ef_search": efs, "build_sec": round(build, 2), "rows_skipped_zero_norm": skipped,
           "qps": round(nq * steps / dt_s, 1), "queries_per_step": nq, "recall_at_10": round(recall(tids, cnt, gt, k), 4),
           "k_fused_query_GBps": round(fq["units"] * row_bytes / max(fq["ms"], 1e-9) / 1e6, 1),
           "k_fused_insert_GBps": round(bstats["fused_insert"]["units"] * row_bytes / max(bstats["fused_insert"]["ms"], 1e-9) / 1e6, 1),
           "k_fused_insert_ms": round(bstats["fused_insert"]["ms"], 1), "k_links_ms": round(bstats["links"]["ms"], 1),
           "fused": ix.fused_stats(), "host_profile": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ix.profile().items()}}
    if name == "c4":
        gm = bstats["gemm"]
        if mfma_build and gm["launches"]:
            # the build's select_neighbors ran as k_fused MODE 3 -> k_wgemm_f16 -> k_wselect: units = candidate pairs of the lower triangles, 2 * dim flops each
            tf = gm["units"] * 2.0 * dim / max(gm["ms"], 1e-9) / 1e9
            st = ix.mfma_stats()
            out["build_select_gemm"] = {"kernel": "k_wgemm_f16 (v_mfma_f32_32x32x16_f16, LDS-staged W x W per member)", "launches": gm["launches"], "pairs": gm["units"], "ms": round(gm["ms"], 1),
                                        "TFLOPs": round(tf, 1), "mfma_util_vs_2.5PF_dense_f16": round(tf / 2500.0, 4),
                                        "rows_streamed_GBps": round(gm["units"] / (efc * (efc - 1) / 2.0) * efc * row_bytes / max(gm["ms"], 1e-9) / 1e6, 1),
                                        "k_wselect_ms": round(bstats["wsel"]["ms"], 1), "decisions_from_matrix": st["mfma_pairs"], "pairs_re_evaluated_canonically": st["exact_pairs"]}
        out["build_select_on_matrix_cores"] = bool(mfma_build)
        # configs[3]: "fp16 MFMA batched-build distance GEMM".  The operands of select_neighbors are blocks of <= 64 rows that lie close together in the
        # graph: an element, its layer-0 neighbours and theirs.  k_pair_mfma_f16 (matrix cores) against k_pair_groups (exact VALU order) on such blocks
        # of THIS index; utilisation against the 2.5 PFLOP/s dense f16 peak (MI355X_MICROARCH.md).
        rng = np.random.default_rng(4)
        groups = []
        for x in rng.integers(0, n, 8192).tolist():
            ids = [x]
            seen = {x}
            nb = ix.neighbors(x, 0)[0].tolist()
            for y in nb + [z for y in nb[:4] for z in ix.neighbors(int(y), 0)[0].tolist()]:
                if y not in seen and len(ids) < 64:
                    seen.add(y); ids.append(int(y))
            if len(ids) >= 33:
                groups.append((ids, None))
        pairs = sum(len(g[0]) * (len(g[0]) - 1) // 2 for g in groups)
        tiles = sum(3 if len(g[0]) > 32 else 1 for g in groups)
        mf = {}
        for kind, on in ((4, True), (1, False)):
            eng.pairwise_many(groups, mfma=on)
            eng.kernel_stats(kind, reset=True)
            for _ in range(3):
                eng.pairwise_many(groups, mfma=on)
            st = eng.kernel_stats(kind, reset=True)
            mf["mfma" if on else "valu"] = st["ms"] / st["launches"]
        flops_issued = tiles * 2.0 * 32 * 32 * (64 * ((dim + 63) // 64))
        out["select_operand_blocks"] = {"blocks": len(groups), "rows_per_block": "33-64 (an element, its layer-0 neighbours and theirs)", "pairs": pairs,
                                        "k_pair_mfma_f16_ms": round(mf["mfma"], 3), "k_pair_groups_ms": round(mf["valu"], 3),
                                        "mfma_Gpairs_per_s": round(pairs / mf["mfma"] / 1e6, 2), "valu_Gpairs_per_s": round(pairs / mf["valu"] / 1e6, 2),
                                        "mfma_TFLOPs_issued": round(flops_issued / mf["mfma"] / 1e9, 1), "mfma_utilisation_vs_2.5PF_dense_f16": round(flops_issued / mf["mfma"] / 1e9 / 2500.0, 4),
                                        "note": "HBM/L2-bound at these block sizes (<= 64 flop per byte); the build itself selects inside k_fused<insert> (exact VALU order) -- hx_index_set_mfma serves the lock-step placement"}
    if iterative:
        passes = (np.arange(n) % iterative["filter_every"] == 0).astype(np.uint8)
        nqi = int(os.environ.get("HX_ITER_QUERIES", "500"))
        t0 = time.perf_counter()
        ix.search_iterative(nqi, efs, iterative["mode"], iterative["max_scan_tuples"], iterative["limit"], passes)   # first call: allocates the per-query tables (GBs)
        it_first_s = time.perf_counter() - t0
        ix.profile(reset=True)
        eng.kernel_stats(2, reset=True)
        t0 = time.perf_counter()
        it_tids, it_d, it_cnt = ix.search_iterative(nqi, efs, iterative["mode"], iterative["max_scan_tuples"], iterative["limit"], passes)
        it_s = time.perf_counter() - t0
        it_k = eng.kernel_stats(2, reset=True)
        it_prof = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in ix.profile().items() if k in ("advance_s", "compact_s", "fill_s", "round_s", "rounds")}
        # exact answer under the filter
        sub = torch.nonzero(torch.from_numpy(passes).to(DEV)).squeeze(1)
        sc = (bits_pm1(qs[:nqi]) @ rpm[sub].T).float()
        gti = sub[torch.topk(sc, k, dim=1).indices].cpu().numpy()
        # the oracle (scalar CPU restatement) on the same graph and filter, one host core, 60 queries
        from oracle import orc
        lvh = ix.export_levels()
        o = orc.Index(orc.BIT, orc.HAMMING, dim, m=m, ef_construction=efc, order=orc.SEQ)
        o.load(rows.cpu().numpy(), lvh, ix.entry, [ix.export_layer(l, with_dist=False) for l in range(int(max(lvh.max(), 0)) + 1)])
        hq = qs[:60].cpu().numpy()
        t0 = time.perf_counter()
        for q in range(len(hq)):
            import ctypes as C
            L = orc.lib()
            qrow = np.ascontiguousarray(hq[q])
            sc = L.orc_scan_begin(o.h, qrow.ctypes.data_as(C.c_void_p), efs, orc.ITER_RELAXED, iterative["max_scan_tuples"])
            tid, dd, ee, got = C.c_int64(), C.c_double(), C.c_int(), 0
            while got < iterative["limit"] and L.orc_scan_next(sc, C.byref(tid), C.byref(dd), C.byref(ee)):   # the executor stops pulling at LIMIT
                got += int(passes[tid.value])
            L.orc_scan_end(sc)
        cpu_it = len(hq) / (time.perf_counter() - t0)
        del o
        out["iterative_relaxed_cpu_oracle_1core_qps"] = round(cpu_it, 1)
        out["iterative_relaxed_first_call_qps"] = round(nqi / it_first_s, 1)
        out["iterative_relaxed"] = {"queries": nqi, "filter": "tid %% %d == 0" % iterative["filter_every"], "max_scan_tuples": iterative["max_scan_tuples"],
                                    "qps": round(nqi / it_s, 1), "recall_at_10": round(recall(it_tids, it_cnt, gti, k), 4),
                                    "mean_returned": float(it_cnt.mean()),
                                    "roofline": {"kernel": "k_fused<OpHamming, 2, 8> (iterative scan, MODE 2)", "bound": "hbm", "bytes_per_distance": row_bytes, "distances": it_k["units"],
                                                 "kernel_ms": round(it_k["ms"], 2), "achieved": round(it_k["units"] * row_bytes / max(it_k["ms"], 1e-9) / 1e6, 1), "peak": 8000.0, "unit": "GB/s",
                                                 "frac": round(it_k["units"] * row_bytes / max(it_k["ms"], 1e-9) / 1e6 / 8000.0, 4),
                                                 "note": "128-byte rows: the scan is bound by the hops of one search (list, visited buckets, heaps, `discarded` bookkeeping), not by bytes"},
                                    "path": "k_fused MODE 2 (device-resident iterative scan); lock-step host driver when set_fused(False)", "host_profile": it_prof}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0, sys.argv[3] if len(sys.argv) > 3 else "baseline")
