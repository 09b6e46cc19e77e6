#!/usr/bin/env python3
"""bench.py -- HNSW build seconds + QPS@recall@10 on 1M x vector(768) L2 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One "step" = one search pass of the hot path over one batch of --queries synthetic queries with its complete
results on the host (get_scan_items + amgettuple semantics, ef_search = --efs, k = 10).  Consecutive steps take
alternating query batches (two batches resident in HBM) and are pipelined through hx_index_search_submit / _wait:
step i+1 is submitted before step i is collected, so the first round of its launch fills the CUs that the last
round of step i's launch leaves idle (--no-pipeline: one launch at a time).  The index those steps run on is built
first, inside this script, by the batched device build (timed separately -> "build_sec").
`value` = queries/s over the K timed steps, whole job (N>1: every rank searches its own query batch on its
replica of the graph; the BUILD is what the ranks share: each lock-step batch of inserts is split over
the ranks and the results are all-gathered over RCCL, see pgvector-rx_amd/dist_build.py).

Prints ONE JSON line (rank 0).  Inputs are synthetic and resident in HBM before any timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pgvector_rx_amd as hx  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s measured-achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--rows", type=int, default=1_000_000)
    p.add_argument("--dim", type=int, default=768)
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--efc", type=int, default=200)
    p.add_argument("--efs", type=int, default=100)
    p.add_argument("--k", type=int, default=10)
    p.add_argument("--queries", type=int, default=10_000)
    p.add_argument("--batch", type=int, default=32768,
                   help="insert batch cap (a batch is also at most 1/8 of the rows already in the graph). Measured on 1M x 768, build s / recall@10: "
                        "8192: 2.04 / 0.9679, 16384: 1.83 / 0.9697, 32768: 1.73 / 0.9680, 65536: 1.69 / 0.9677 -- fewer launches, fewer last rounds")
    p.add_argument("--dist", default="gmm", choices=["gmm", "uniform"])
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--cpu-build-rows", type=int, default=600)
    p.add_argument("--cpu-queries", type=int, default=300)
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-k1-1536", action="store_true", help="skip the K1 micro-benchmark on a vector(1536) table")
    p.add_argument("--no-query-sweep", action="store_true", help="skip the 2x / 4x query-batch lines")
    p.add_argument("--no-pipeline", action="store_true", help="one scan launch at a time (hx_index_search) instead of submitting step i+1 before collecting step i")
    p.add_argument("--slots", type=int, default=2, choices=[2, 3, 4], help="scan slots (= query batches resident in HBM) the pipelined steps rotate over: step i+slots-1 is submitted before step i is collected")
    p.add_argument("--no-efs-sweep", action="store_true", help="skip the ef_search 40 / 200 lines (BASELINE.md C2 lists 40, 100, 200)")
    p.add_argument("--no-fused", action="store_true", help="run every traversal in the lock-step host driver")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo = rehearsal of the multi-rank build with CPU-side exchange (several ranks may share one GPU)")
    p.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    return p.parse_args()


def synth(n, dim, dist, seed, device, centres=None):
    """U[0,1) or a 1024-centre Gaussian mixture (sigma 0.1), f32, seeded."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if dist == "uniform":
        return torch.rand((n, dim), generator=g, device=device, dtype=torch.float32), None
    if centres is None:
        gc = torch.Generator(device=device)
        gc.manual_seed(11)
        centres = torch.rand((1024, dim), generator=gc, device=device, dtype=torch.float32)
    out = torch.empty((n, dim), device=device, dtype=torch.float32)
    step = 131072
    for i in range(0, n, step):
        b = min(step, n - i)
        a = torch.randint(0, centres.shape[0], (b,), generator=g, device=device)
        out[i:i + b] = centres[a] + 0.1 * torch.randn((b, dim), generator=g, device=device, dtype=torch.float32)
    return out, centres


def ground_truth(rows, queries, k):
    """Exact top-k by squared L2 with torch (plumbing for the recall figure, not the product path)."""
    rn = (rows * rows).sum(1)
    out = []
    step = max(1, int(2e9 // (rows.shape[0] * 4)))
    for i in range(0, queries.shape[0], step):
        q = queries[i:i + step]
        d = rn[None, :] - 2.0 * (q @ rows.T)
        out.append(torch.topk(d, k, dim=1, largest=False).indices)
    return torch.cat(out).cpu().numpy()


def recall_at_k(tids, cnt, gt, k):
    hit = 0
    for q in range(gt.shape[0]):
        hit += len(set(tids[q, :cnt[q]].tolist()) & set(gt[q].tolist()))
    return hit / (gt.shape[0] * k)


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves (fresh children, before this process makes any
    GPU call -- a process that touched the GPU must never be re-executed) and pass rank 0's JSON line through.  Watchdog: every child is
    polled; the first one that exits non-zero (or the overall deadline, HX_BENCH_RANK_TIMEOUT seconds) takes the others down with it and
    this process exits non-zero -- no rank is left waiting for a dead peer inside a collective."""
    import socket
    import subprocess
    import tempfile
    visible = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if not a.share_gpu and visible < a.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (a.gpus, visible), file=sys.stderr, flush=True)
        raise SystemExit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    deadline = time.monotonic() + float(os.environ.get("HX_BENCH_RANK_TIMEOUT", "3000"))
    out0 = tempfile.TemporaryFile()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    bad = None
    while True:
        rcs = [p.poll() for p in procs]
        failed = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if failed:
            bad = "rank %d exited with code %d" % failed[0]
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            bad = "deadline of HX_BENCH_RANK_TIMEOUT seconds passed"
            break
        time.sleep(0.2)
    if bad:
        for p in procs:                           # exactly the children started above, by handle
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        print("bench.py: %s; the other ranks were stopped" % bad, file=sys.stderr, flush=True)
        raise SystemExit(1)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    raise SystemExit(0)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)
    # stdout carries ONE line, the JSON: libraries that print to fd 1 (gloo's "[Gloo] Rank 0 is connected ..." banner) go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    try:
        run(a, json_fd)
    finally:
        os.close(json_fd)


def run(a, json_fd):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world), file=sys.stderr, flush=True)
        raise SystemExit(2)
    import torch.distributed as dist
    if a.share_gpu:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        print("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()), file=sys.stderr, flush=True)
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    comm = None
    xdev = torch.device("cpu")                   # where the data-path collectives' tensors live
    from importlib import import_module
    dbm = import_module("pgvector-rx_amd.dist_build")
    if world > 1:
        # gloo first (explicit timeout), then RCCL for the data path with the ranks agreeing over gloo at every step (dist_build.bring_up)
        comm, xdev = dbm.bring_up(rank, world, dev, want=a.dist_backend, log=lambda m: print("bench.py: " + m, file=sys.stderr, flush=True))
        a.dist_backend = comm.backend

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- synthetic inputs, resident in HBM (identical on every rank: counter-seeded) ----
    rows, centres = synth(a.rows, a.dim, a.dist, 11, dev)
    queries, _ = synth(a.queries, a.dim, a.dist, 12 + 1000 * rank, dev, centres)
    levels = hx.draw_levels(a.rows, a.m, seed=11)
    torch.cuda.synchronize()
    eng = hx.Engine(hx.F32, hx.L2SQ, a.dim, a.rows, device=local_rank)
    eng.append_device(rows.data_ptr(), a.rows)
    ix = hx.Index(eng, a.m, a.efc)
    if a.threads:
        ix.set_threads(a.threads)
    eng.set_timing(True)
    if a.no_fused:
        ix.set_fused(False)

    # ---- build (timed once; barrier + sync on both sides; max over ranks) ----
    # a batch is shared by the ranks, so its cap grows with them: every GPU keeps a full launch of searches per batch
    eff_batch = min(a.batch * world, 262144)
    dist_stages = None
    barrier()
    t0 = time.perf_counter()
    if world > 1:
        dbm.insert_sharded(ix, 0, levels, eff_batch, comm, xdev, gpu=dev)
        dist_stages = {k: round(v, 3) for k, v in dbm.STAGE_SECONDS.items()}
    else:
        ix.insert(0, levels, batch=eff_batch)
    barrier()
    build_sec = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([build_sec], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        build_sec = float(t.item())
    build_stats = {"dist": eng.kernel_stats(0, reset=True), "pair": eng.kernel_stats(1, reset=True), "fused": eng.kernel_stats(2, reset=True), "links": eng.kernel_stats(3, reset=True)}
    counters = ix.counters()
    build_prof = ix.profile(reset=True)

    # ---- search steps ----
    # a.slots query batches resident in HBM; step i scans batch i % slots
    qbatches = [queries] + [synth(a.queries, a.dim, a.dist, 14 + j + 1000 * rank, dev, centres)[0] for j in range(a.slots - 1)]
    queries_b = qbatches[1]
    qboth = torch.cat(qbatches)
    torch.cuda.synchronize()
    eng.set_queries_device(qboth.data_ptr(), a.slots * a.queries)
    pipelined = not a.no_pipeline and not a.no_fused
    S = a.slots

    def run_steps(n, efs):
        """n steps; returns the last results per query batch (each step's results are complete on the host when it ends)"""
        res = [None] * S
        if not pipelined:
            for i in range(n):
                eng_first = (i % S) * a.queries
                if eng_first:     # plain scans read query slots 0..nq-1: the other batches are scanned through a submit on slot 0 + wait (no overlap)
                    ix.search_submit(0, eng_first, a.queries, efs, a.k)
                    res[i % S] = ix.search_wait(0)
                else:
                    res[0] = ix.search(a.queries, efs, a.k)
            return res
        for j in range(min(S - 1, n)):
            ix.search_submit(j % S, (j % S) * a.queries, a.queries, efs, a.k)
        for i in range(n):
            j = i + S - 1
            if j < n:
                ix.search_submit(j % S, (j % S) * a.queries, a.queries, efs, a.k)
            res[i % S] = ix.search_wait(i % S)
        return res

    run_steps(a.warmup, a.efs)
    warm_stat = eng.kernel_stats(0, reset=True)
    warm_fused = eng.kernel_stats(2, reset=True)
    eng.kernel_stats(5, reset=True)
    barrier()
    t0 = time.perf_counter()
    results = run_steps(a.steps, a.efs)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    sstat = eng.kernel_stats(0)
    fstat = eng.kernel_stats(2)
    scan_stat = eng.kernel_stats(5)
    search_prof = ix.profile()
    qps = world * a.queries * a.steps / dt

    # the batched L2 kernel of the lock-step placement (K1 k_dist_groups: query parked in LDS, coalesced row streaming,
    # shuffle reduction) on the same resident table, at a lock-step-sized launch: 32768 expansions x 32 rows
    k1 = None
    if rank == 0:
        rng = np.random.default_rng(5)
        g_n, per = 32768, 2 * a.m
        gq = rng.integers(0, a.rows, g_n).astype(np.uint32)
        goff = (np.arange(g_n + 1) * per).astype(np.uint32)
        gids = rng.integers(0, a.rows, g_n * per).astype(np.uint32)
        eng.distances_batch(gq, goff, gids)
        eng.kernel_stats(0, reset=True)
        for _ in range(10):                                   # fresh ids per launch: no launch re-reads the rows of the one before
            eng.distances_batch(rng.integers(0, a.rows, g_n).astype(np.uint32), goff, rng.integers(0, a.rows, g_n * per).astype(np.uint32))
        ks = eng.kernel_stats(0, reset=True)
        k1_gbps = ks["units"] * a.dim * 4 / max(ks["ms"], 1e-9) / 1e6
        k1 = {"kernel": "k_dist_groups", "bound": "hbm", "achieved": round(k1_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
              "frac": round(k1_gbps / HBM_PEAK_GBPS, 4), "launches": ks["launches"], "avg_launch_ms": round(ks["ms"] / ks["launches"], 4),
              "distances_per_launch": g_n * per, "bytes_per_distance": a.dim * 4}

        # the same kernel at the north-star shape of its >= 50 % target: vector(1536) on a table of its own of 1M rows (6.1 GB: far beyond the 256 MiB
        # Infinity Cache), FRESH random ids for every launch (32768 expansions x 32 rows = 1M picks of 1M rows: a row is read ~once per launch and
        # never the same set twice), so the figure is an HBM rate, not a cache rate.  K1 is the lock-step placement's kernel: the default build and
        # scan paths run in k_fused (build_kernels.k_dist_groups.launches = 0 above).
        if not a.no_k1_1536:
            d2, n2 = 1536, 1_048_576
            g2 = torch.Generator(device=dev)
            g2.manual_seed(7)
            t2 = torch.rand((n2, d2), generator=g2, device=dev, dtype=torch.float32)
            torch.cuda.synchronize()
            e2 = hx.Engine(hx.F32, hx.L2SQ, d2, n2, device=local_rank)
            e2.append_device(t2.data_ptr(), n2)
            del t2
            e2.set_timing(True)
            e2.distances_batch(rng.integers(0, n2, g_n).astype(np.uint32), goff, rng.integers(0, n2, g_n * per).astype(np.uint32))
            e2.kernel_stats(0, reset=True)
            uniq = 0
            for _ in range(10):
                gids2 = rng.integers(0, n2, g_n * per).astype(np.uint32)
                uniq += len(np.unique(gids2))
                e2.distances_batch(rng.integers(0, n2, g_n).astype(np.uint32), goff, gids2)
            ks2 = e2.kernel_stats(0, reset=True)
            gb2 = ks2["units"] * d2 * 4 / max(ks2["ms"], 1e-9) / 1e6
            k1["d1536"] = {"kernel": "k_dist_groups", "bound": "hbm", "achieved": round(gb2, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gb2 / HBM_PEAK_GBPS, 4),
                           "launches": ks2["launches"], "avg_launch_ms": round(ks2["ms"] / ks2["launches"], 4), "distances_per_launch": g_n * per,
                           "bytes_per_distance": d2 * 4, "table_rows": n2, "ids": "fresh uniform random ids per launch",
                           "unique_rows_per_launch": round(uniq / 10), "unique_row_GBps": round(uniq * d2 * 4 / max(ks2["ms"], 1e-9) / 1e6, 1),
                           "on_default_path": False}
            e2.close()
        k1["on_default_path"] = False

    # the same scan at larger query batches: a launch costs a fixed ~2 ms (the drain of its last round of searches, one search long) on top of
    # ~0.7 us per query, so the rate of the kernel itself shows at batches that amortise it
    sweep = None
    if rank == 0 and not a.no_query_sweep:
        sweep = []
        big, _ = synth(4 * a.queries, a.dim, a.dist, 99, dev, centres)
        for nq in (2 * a.queries, 4 * a.queries):
            eng.set_queries_device(big.data_ptr(), nq)
            ix.search(nq, a.efs, a.k)
            eng.kernel_stats(2, reset=True)
            t1 = time.perf_counter()
            for _ in range(2):
                ix.search(nq, a.efs, a.k)
            dq = (time.perf_counter() - t1) / 2
            st = eng.kernel_stats(2, reset=True)
            gb = st["units"] * a.dim * 4 / max(st["ms"], 1e-9) / 1e6
            sweep.append({"queries_per_launch": nq, "qps": round(nq / dq, 1), "kernel_ms": round(st["ms"] / max(1, st["launches"]), 3),
                          "achieved": round(gb, 1), "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBPS, 4)})
        eng.set_queries_device(qboth.data_ptr(), a.slots * a.queries)
        del big

    gt = ground_truth(rows, queries, a.k)
    gt_b = ground_truth(rows, queries_b, a.k)
    hits_n, hits_d = 0.0, 0
    for r, g_ in ((results[0], gt), (results[1], gt_b)):              # (batches beyond the second are scanned and timed like these two; recall is taken on two)
        if r is not None:
            hits_n += recall_at_k(r[0], r[3], g_, a.k) * g_.shape[0]
            hits_d += g_.shape[0]
    recall = hits_n / max(1, hits_d)

    # BASELINE.md C2 lists ef_search 40 / 100 / 200: the same pipelined steps at the other two operating points
    efs_sweep = None
    if rank == 0 and not a.no_efs_sweep:
        efs_sweep = []
        for efs in (40, 200):
            if efs == a.efs:
                continue
            run_steps(1, efs)
            eng.kernel_stats(2, reset=True)
            eng.kernel_stats(5, reset=True)
            t1 = time.perf_counter()
            rr = run_steps(8, efs)
            dq = time.perf_counter() - t1
            st = eng.kernel_stats(2, reset=True)
            rc_ = (recall_at_k(rr[0][0], rr[0][3], gt, a.k) + recall_at_k(rr[1][0], rr[1][3], gt_b, a.k)) / 2
            gb = st["units"] * a.dim * 4 / dq / 1e9
            efs_sweep.append({"ef_search": efs, "qps": round(8 * a.queries / dq, 1), "recall_at_10": round(rc_, 4), "steps": 8,
                              "achieved": round(gb, 1), "unit": "GB/s", "frac": round(gb / HBM_PEAK_GBPS, 4), "distances_per_query": round(st["units"] / (8 * a.queries), 1)})
    if world > 1:
        t = torch.tensor([recall], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        recall = float(t.item()) / world

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel of the timed region (query-vs-rows, K1), HIP-event timed on its stream ----
    row_bytes = a.dim * 4
    k1_ms = sstat["ms"] / max(1, sstat["launches"])
    k1_gbps = sstat["units"] * row_bytes / max(sstat["ms"], 1e-9) / 1e6
    roofline = {"bound": "hbm", "kernel": "k_dist_groups", "achieved": round(k1_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(k1_gbps / HBM_PEAK_GBPS, 4), "traffic": None,
                "launches": sstat["launches"], "avg_launch_ms": round(k1_ms, 4),
                "distances_per_launch": round(sstat["units"] / max(1, sstat["launches"]), 1), "bytes_per_distance": row_bytes}
    if fstat["launches"]:
        # the timed region ran in the device-resident traversal kernel: it is the dominant kernel, and every distance it
        # evaluates streams one row from HBM exactly like K1 does
        f_ms = fstat["ms"] / fstat["launches"]
        f_gbps = fstat["units"] * row_bytes / max(fstat["ms"], 1e-9) / 1e6
        roofline = {"bound": "hbm", "kernel": "k_fused<query>", "achieved": round(f_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(f_gbps / HBM_PEAK_GBPS, 4), "traffic": None, "launches": fstat["launches"],
                    "avg_launch_ms": round(f_ms, 3), "distances_per_launch": round(fstat["units"] / fstat["launches"], 1),
                    "bytes_per_distance": row_bytes}
        if pipelined and scan_stat["launches"]:
            # consecutive launches overlap (two streams), so the sum of their durations exceeds the time the chip was busy: achieved = algorithmic
            # bytes of the timed steps / the timed WALL (host clock, barrier + synchronize on both sides; it includes every step's result copy and
            # the host-side expansion of the results).  Next to it: the same bytes / the union of the launches' [start, end] intervals (HIP events on
            # each launch's own stream), and each launch's own duration (what rocprofv3 --kernel-trace --stats averages).
            w_gbps = scan_stat["units"] * row_bytes / dt / 1e9
            u_gbps = scan_stat["units"] * row_bytes / max(scan_stat["ms"], 1e-9) / 1e6
            roofline.update({"achieved": round(w_gbps, 1), "frac": round(w_gbps / HBM_PEAK_GBPS, 4),
                             "achieved_basis": "algorithmic bytes of the timed steps / timed wall (launches of consecutive steps overlap)",
                             "timed_wall_ms": round(1000.0 * dt, 3),
                             "kernel_busy_union_ms": round(scan_stat["ms"], 3), "achieved_over_kernel_busy_union": round(u_gbps, 1),
                             "sum_of_launch_durations_ms": round(fstat["ms"], 3), "achieved_per_launch_duration": round(f_gbps, 1)})
        # every launch of this kernel in the process (warmup + timed): what `rocprofv3 --kernel-trace --stats` averages
        al, ams = warm_fused["launches"] + fstat["launches"], warm_fused["ms"] + fstat["ms"]
        roofline["whole_run"] = {"launches": al, "avg_launch_ms": round(ams / max(1, al), 3),
                                 "achieved": round((warm_fused["units"] + fstat["units"]) * row_bytes / max(ams, 1e-9) / 1e6, 1)}
        # HBM traffic per launch from a separate `rocprofv3 --pmc FETCH_SIZE` pass of this same command (profiles/), if recorded
        pmc = os.path.join(ROOT, "profiles", "pmc_k_fused_query.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                import glob
                import hashlib
                hh = hashlib.sha256()
                for f in sorted(glob.glob(os.path.join(ROOT, "pgvector-rx_amd", "csrc", "*"))):
                    hh.update(os.path.basename(f).encode())
                    hh.update(open(f, "rb").read())
                same_kernels = rec.get("csrc_sha16") == hh.hexdigest()[:16]
                if rec.get("rows") == a.rows and rec.get("dim") == a.dim and rec.get("queries") == a.queries and rec.get("efs") == a.efs and same_kernels:
                    roofline["traffic"] = rec["hbm_bytes_per_launch"]
                    roofline["traffic_source"] = rec.get("source")
                elif not same_kernels:
                    roofline["traffic_source"] = "none: the kernel sources changed since the recorded rocprofv3 --pmc FETCH_SIZE pass (profiles/pmc_k_fused_query.json)"
            except Exception:
                pass
    bd, bp = build_stats["dist"], build_stats["pair"]
    bf = build_stats["fused"]
    bl = build_stats["links"]
    # every K1 launch of this process (build + warmup + timed steps): the figure a `rocprofv3 --kernel-trace --stats`
    # run of this same command reports as the kernel's average duration
    if not fstat["launches"]:
        all_l = bd["launches"] + warm_stat["launches"] + sstat["launches"]
        all_ms = bd["ms"] + warm_stat["ms"] + sstat["ms"]
        roofline["whole_run"] = {"launches": all_l, "avg_launch_us": round(1000.0 * all_ms / max(1, all_l), 2),
                                 "achieved": round((bd["units"] + warm_stat["units"] + sstat["units"]) * row_bytes / max(all_ms, 1e-9) / 1e6, 1)}
    build_kernels = {
        "k_dist_groups": {"launches": bd["launches"], "distances": bd["units"], "ms": round(bd["ms"], 1),
                          "GBps": round(bd["units"] * row_bytes / max(bd["ms"], 1e-9) / 1e6, 1)},
        "k_fused<insert>": {"launches": bf["launches"], "distances": bf["units"], "ms": round(bf["ms"], 1),
                            "GBps": round(bf["units"] * row_bytes / max(bf["ms"], 1e-9) / 1e6, 1)},
        "k_links": {"launches": bl["launches"], "pairs": bl["units"], "ms": round(bl["ms"], 1),
                    "Gpairs_per_s": round(bl["units"] / max(bl["ms"], 1e-9) / 1e6, 2)},
        "k_pair_groups": {"launches": bp["launches"], "pairs": bp["units"], "ms": round(bp["ms"], 1),
                          "Gpairs_per_s": round(bp["units"] / max(bp["ms"], 1e-9) / 1e6, 2)},
    }

    # ---- CPU baseline: the oracle (scalar restatement of the reference) on this box's host cores, 1 thread ----
    cpu = None
    if not a.no_cpu and world == 1:
        from oracle import orc
        # search baseline: the scalar scan on the SAME graph the device built
        all_rows = rows.cpu().numpy()
        o = orc.Index(orc.F32, orc.L2SQ, a.dim, m=a.m, ef_construction=a.efc, order=orc.SEQ)
        lv = ix.export_levels()
        layers = [ix.export_layer(l, with_dist=True) for l in range(int(max(lv.max(), 0)) + 1)]
        o.load(all_rows, lv, ix.entry, layers)
        hq = queries[:a.cpu_queries].cpu().numpy()
        t0 = time.perf_counter()
        hits = 0
        for q in range(len(hq)):
            ids, _ = o.search_topk(hq[q], a.efs, a.k)
            hits += len(set(ids.tolist()) & set(gt[q].tolist()))
        cpu_search = time.perf_counter() - t0
        # the same scan on all host cores the box grants (one query per thread at a time = one backend per connection),
        # and a reassociated, compiler-vectorised variant (NOT the reference's arithmetic) so the GPU figure is also
        # quoted against a CPU that is not held back by the reference's dependent scalar add chain
        try:
            quota = open("/sys/fs/cgroup/cpu.max").read().split()
            host_threads = max(1, int(int(quota[0]) / int(quota[1]))) if quota[0] != "max" else (os.cpu_count() or 1)
        except Exception:
            host_threads = os.cpu_count() or 1
        host_threads = min(host_threads, os.cpu_count() or 1)
        nq_all = min(a.queries, max(a.cpu_queries, 200 * host_threads))
        hq_all = queries[:nq_all].cpu().numpy()
        t0 = time.perf_counter()
        ids_all, cnt_all = o.search_many(hq_all, a.efs, a.k, n_threads=host_threads)
        cpu_all = time.perf_counter() - t0
        hits_all = sum(len(set(ids_all[q, :cnt_all[q]].tolist()) & set(gt[q].tolist())) for q in range(nq_all))
        # build baseline: the reference's sequential build_callback (one row at a time, one core) inserting fresh rows of the same
        # distribution INTO the full-size graph the device built -- the per-row cost at this index size, not that of a tiny graph
        n_cb = a.cpu_build_rows
        extra, _ = synth(n_cb, a.dim, a.dist, 13, dev, centres)
        extra = extra.cpu().numpy()
        extra_levels = hx.draw_levels(n_cb, a.m, seed=13)
        t0 = time.perf_counter()
        for i in range(n_cb):
            o.insert(extra[i], extra_levels[i], a.rows + i)
        cpu_build = time.perf_counter() - t0
        del o
        ov = orc.Index(orc.F32, orc.L2SQ, a.dim, m=a.m, ef_construction=a.efc, order=orc.VEC)
        ov.load(all_rows, lv, ix.entry, layers)
        t0 = time.perf_counter()
        ids_v, cnt_v = ov.search_many(hq_all, a.efs, a.k, n_threads=host_threads)
        cpu_vec = time.perf_counter() - t0
        hits_v = sum(len(set(ids_v[q, :cnt_v[q]].tolist()) & set(gt[q].tolist())) for q in range(nq_all))
        del ov
        cpu_more = {
            "all_cores": {"value": round(nq_all / cpu_all, 1), "unit": "queries/s", "cores": host_threads, "queries": nq_all,
                          "recall_at_10": round(hits_all / (nq_all * a.k), 4), "arithmetic": "reference order (scalar f32 chain)"},
            "all_cores_vectorised": {"value": round(nq_all / cpu_vec, 1), "unit": "queries/s", "cores": host_threads, "queries": nq_all,
                                     "recall_at_10": round(hits_v / (nq_all * a.k), 4),
                                     "arithmetic": "reassociated, 16 partial sums, gcc -O3 avx512f/avx2 clones: not the reference's rounding"}}
        cpu = {"value": round(len(hq) / cpu_search, 1), "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": "%d of the same queries, ef_search=%d, scalar oracle (ORC_ORDER_SEQ) scanning the same %d-row graph; "
                         "omits fmgr/bufmgr/lock overhead so it is faster than the reference itself" % (len(hq), a.efs, a.rows),
               "recall_at_10": round(hits / (len(hq) * a.k), 4),
               "build_rows_per_s": round(n_cb / cpu_build, 2),
               "build_sample": "sequential oracle inserts (build_callback, one core) of %d fresh rows into the %d-row graph the device built: %.1f s" % (n_cb, a.rows, cpu_build),
               "host_cpus": os.cpu_count(), **cpu_more}

    out = {
        "metric": "HNSW build sec + QPS@recall@10, 1Mx768 f32 L2",
        "value": round(qps, 1), "unit": "queries/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(1000.0 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: %d x vector(%d) L2, m=%d ef_construction=%d, build + search on %d MI355X" % (a.rows, a.dim, a.m, a.efc, world),
                   "distribution": a.dist, "ef_search": a.efs, "k": a.k, "queries_per_step": a.queries,
                   "steps_pipelined": bool(pipelined), "query_batches_resident": a.slots,
                   "insert_batch_cap": eff_batch, "host_threads": a.threads or min(16, os.cpu_count() or 1)},
        "build_sec": round(build_sec, 2),
        "build_rows_per_s": round(a.rows / build_sec, 1),
        "recall_at_10": round(recall, 4),
        "roofline": roofline,
        "roofline_k1_batched_l2": k1,
        "query_batch_sweep": sweep,
        "ef_search_sweep": efs_sweep,
        "cpu_baseline": cpu,
        "build_kernels": build_kernels,
        "host_profile": {"build": {k: round(v, 2) for k, v in build_prof.items()}, "search_all_steps": {k: round(v, 3) for k, v in search_prof.items()}},
        "fused": ix.fused_stats(),
        "parity": "graphs, top-k lists and iterative scans bit-identical to the CPU oracle run in the device's canonical summation order (ORC_ORDER_W64: tests/, -m gpu); "
                  "distances within 1e-5 * d (L2, L1) / 1e-5 * sum|a_i b_i| (inner product) of the reference's sequential f32 order; the build uses snapshot batches "
                  "(insert_batch_cap), whose recall at this ef_search equals the reference's one-row-at-a-time schedule query by query at 100k, 300k and 1M x 768 (the 1M fixture is this bench's own size and mixture) "
                  "(tests/test_gpu_recall_parity.py; searches starved to ef_search <= 20 lose up to 1.4 % at this cap, none at caps <= 4096: DESIGN.md 4)",
        "dist_backend": (comm.backend if comm is not None else None),
        "dist_build_stage_seconds_rank0": dist_stages,
        "build_distance_evals": {"search": int(counters[1]), "select": int(counters[2]), "backlink": int(counters[3])},
    }
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
