"""Kernel micro-benchmark: achieved HBM GB/s of the batched query-vs-rows kernel (K1 k_dist_groups) and the pair rate of the
pair-block kernel (K2) at lock-step request shapes, HIP-event timed on the engine's stream.
Usage: python tools/kbench.py [dtype f32|f16|bit] [metric l2|ip|l1|hamming] [dim] [n_rows]   -> one JSON line per shape"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pgvector_rx_amd as hx  # noqa: E402

DT = {"f32": hx.F32, "f16": hx.F16, "bit": hx.BIT}
MT = {"l2": hx.L2SQ, "ip": hx.NEG_IP, "l1": hx.L1, "hamming": hx.HAMMING, "jaccard": hx.JACCARD}


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
    metric = sys.argv[2] if len(sys.argv) > 2 else "l2"
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 1536
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    torch.manual_seed(0)
    if dtype == "f32":
        rows = torch.rand((n, dim), device="cuda", dtype=torch.float32)
    elif dtype == "f16":
        rows = torch.rand((n, dim), device="cuda", dtype=torch.float16)
    else:
        rows = torch.randint(0, 256, (n, (dim + 7) // 8), device="cuda", dtype=torch.uint8)
    torch.cuda.synchronize()
    e = hx.Engine(DT[dtype], MT[metric], dim, n)
    e.append_device(rows.data_ptr(), n)
    row_bytes = e.row_bytes
    del rows
    e.set_timing(True)
    rng = np.random.default_rng(0)
    for groups, per in [(4096, 32), (8192, 16), (16384, 32), (65536, 32)]:
        gq = rng.integers(0, n, groups).astype(np.uint32)
        off = (np.arange(groups + 1) * per).astype(np.uint32)
        ids = rng.integers(0, n, groups * per).astype(np.uint32)
        for _ in range(3):
            e.distances_batch(gq, off, ids)
        e.kernel_stats(0, reset=True)
        for _ in range(10):
            e.distances_batch(gq, off, ids)
        s = e.kernel_stats(0)
        ms = s["ms"] / s["launches"]
        gbs = groups * per * row_bytes / ms / 1e6
        print(json.dumps({"kernel": "k_dist_groups", "dtype": dtype, "metric": metric, "dim": dim, "row_bytes": int(row_bytes), "rows_in_table": n,
                          "groups": groups, "rows_per_group": per, "avg_launch_ms": round(ms, 4), "GBps_rows_only": round(gbs, 1),
                          "frac_of_8TBps": round(gbs / 8000, 3)}), flush=True)
    for groups, w in [(16384, 33), (4096, 48)]:
        gl = [(rng.integers(0, n, w).tolist(), None) for _ in range(groups)]
        for _ in range(2):
            e.pairwise_many(gl)
        e.kernel_stats(1, reset=True)
        for _ in range(5):
            e.pairwise_many(gl)
        s = e.kernel_stats(1)
        ms = s["ms"] / s["launches"]
        pairs = groups * w * (w - 1) // 2
        print(json.dumps({"kernel": "k_pair_groups", "dtype": dtype, "metric": metric, "dim": dim, "groups": groups, "w": w, "avg_launch_ms": round(ms, 4),
                          "Gpairs_per_s": round(pairs / ms / 1e6, 2), "row_GBps": round(groups * w * row_bytes / ms / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
