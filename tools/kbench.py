"""Kernel micro-benchmark: achieved HBM GB/s of the query-vs-rows kernel (K1) and pair rate of the
pair-block kernel (K2) at lock-step request shapes.  Usage: python tools/kbench.py [dim] [n_rows]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pgvector_rx_amd as hx  # noqa: E402


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    torch.manual_seed(0)
    rows = torch.rand((n, dim), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    e = hx.Engine(hx.F32, hx.L2SQ, dim, n)
    e.append_device(rows.data_ptr(), n)
    del rows
    e.set_timing(True)
    rng = np.random.default_rng(0)
    out = []
    for groups, per in [(4096, 32), (8192, 16), (8192, 8), (16384, 32), (1024, 32)]:
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
        gbs = groups * per * dim * 4 / ms / 1e6
        out.append({"kernel": "dist_groups", "dim": dim, "groups": groups, "rows_per_group": per,
                    "ms": round(ms, 4), "GBps_rows_only": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000, 3)})
        print(json.dumps(out[-1]), flush=True)
    for groups, w in [(4096, 33), (16384, 33), (2048, 48)]:
        gl = [(rng.integers(0, n, w).tolist(), None) for _ in range(groups)]
        for _ in range(2):
            e.pairwise_many(gl)
        e.kernel_stats(1, reset=True)
        for _ in range(5):
            e.pairwise_many(gl)
        s = e.kernel_stats(1)
        ms = s["ms"] / s["launches"]
        pairs = groups * w * (w - 1) // 2
        out.append({"kernel": "pair_groups", "dim": dim, "groups": groups, "w": w, "ms": round(ms, 4),
                    "Gpairs_per_s": round(pairs / ms / 1e6, 2), "row_GBps": round(groups * w * dim * 4 / ms / 1e6, 1)})
        print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    main()
