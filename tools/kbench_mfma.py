"""MFMA utilisation of the batched-build distance GEMM (hx_mfma.hip, k_pair_mfma_f16) against the gfx950 dense f16 peak, next to the exact
VALU pair kernel (K2 k_pair_groups) on the same groups.  BASELINE configs[3] shape: halfvec(4000) inner product.
Usage: python tools/kbench_mfma.py [dim] [n_rows] [locality]   -> one JSON line per group shape
locality = rows of a group drawn from a window of that many consecutive rows (0: the whole table): candidates of one select_neighbors call
are rows the same search has just read, i.e. L2-resident; the whole-table case is the cold extreme."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import pgvector_rx_amd as hx  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
HBM_PEAK_GBPS = 8000.0


def main():
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
    window = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    torch.manual_seed(0)
    rows = (2.0 * torch.rand((n, dim), device="cuda") * torch.rand((n, dim), device="cuda")).to(torch.float16)
    torch.cuda.synchronize()
    e = hx.Engine(hx.F16, hx.NEG_IP, dim, n)
    e.append_device(rows.data_ptr(), n)
    row_bytes = e.row_bytes
    del rows
    e.set_timing(True)
    rng = np.random.default_rng(0)
    for groups, w in [(16384, 64), (16384, 33), (8192, 48)]:
        if window:
            base = rng.integers(0, n - window, groups)
            gl = [((base[g] + rng.permutation(window)[:w]).tolist(), None) for g in range(groups)]
        else:
            gl = [(rng.integers(0, n, w).tolist(), None) for _ in range(groups)]
        pairs = groups * w * (w - 1) // 2
        tiles = groups * (3 if w > 32 else 1)
        flops_issued = tiles * 2.0 * 32 * 32 * (64 * ((dim + 63) // 64))        # what the matrix cores execute (whole tiles, padded k)
        flops_useful = pairs * 2.0 * dim
        for kind, mf in ((4, True), (1, False)):
            for _ in range(2):
                e.pairwise_many(gl, mfma=mf)
            e.kernel_stats(kind, reset=True)
            for _ in range(5):
                e.pairwise_many(gl, mfma=mf)
            s = e.kernel_stats(kind)
            ms = s["ms"] / s["launches"]
            rec = {"kernel": "k_pair_mfma_f16" if mf else "k_pair_groups", "dtype": "f16", "metric": "ip", "dim": dim, "groups": groups, "w": w,
                   "row_window": window or n, "avg_launch_ms": round(ms, 4), "Gpairs_per_s": round(pairs / ms / 1e6, 2),
                   "row_GBps": round(groups * w * row_bytes / ms / 1e6, 1), "row_frac_of_hbm_peak": round(groups * w * row_bytes / ms / 1e6 / HBM_PEAK_GBPS, 3)}
            if mf:
                rec["mfma_TFLOPs_issued"] = round(flops_issued / ms / 1e9, 1)
                rec["mfma_utilisation_vs_2.5PF"] = round(flops_issued / ms / 1e9 / MFMA_PEAK_TFLOPS, 4)
                rec["useful_TFLOPs"] = round(flops_useful / ms / 1e9, 1)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
