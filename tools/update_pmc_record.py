"""Rewrites profiles/pmc_k_fused_query.json from a FETCH_SIZE summary produced by tools/profile_round.sh (tools/rocprof_summary.py output).
Usage: python tools/update_pmc_record.py profiles/r02_rocprofv3_pmc_fetch_size_1m.txt"""
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pgvector-rx_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def main():
    path = sys.argv[1]
    mean, total = {}, {}
    for line in open(path):
        m = re.match(r"(.+?)\s+FETCH_SIZE\s+dispatches=(\d+)\s+mean=([\d.]+) total=([\d.]+)", line)
        if m:
            k = m.group(1).strip()
            mean[k], total[k] = float(m.group(3)), float(m.group(4))
    q = next(k for k in mean if k.startswith("void k_fused<OpF32<0>, 0, 64, false>") or k.startswith("void k_fused<OpF32<0>, 0, 64>"))
    k1 = next((k for k in mean if "k_dist_groups" in k), None)
    rec_path = os.path.join(ROOT, "profiles", "pmc_k_fused_query.json")
    rec = json.load(open(rec_path))
    rec["fetch_size_kb_per_launch_raw"] = mean[q]
    rec["hbm_bytes_per_launch"] = int(mean[q] * 1024 * 2)
    rec["kernel"] = q.split("(")[0].replace("void ", "")
    rec["source"] = "%s: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2 (own pass)" % os.path.relpath(path, ROOT)
    rec["csrc_sha16"] = csrc_hash()
    keep = rec.get("build_kernels_fetch_size_kb_total_raw", {}).get("round1_k_links_cached")
    rec["build_kernels_fetch_size_kb_total_raw"] = {k.split("(")[0].replace("void ", ""): v for k, v in total.items() if "k_fused<OpF32<0>, 1" in k or "k_links" in k or "k_pm_fill" in k}
    if keep:
        rec["build_kernels_fetch_size_kb_total_raw"]["round1_k_links_cached"] = keep
    if k1:
        rec["k1_batched_l2"]["fetch_size_kb_per_launch_raw"] = mean[k1]
        rec["k1_batched_l2"]["hbm_bytes_per_launch"] = int(mean[k1] * 1024 * 2)
    json.dump(rec, open(rec_path, "w"), indent=1)
    print("updated", rec_path, rec["csrc_sha16"], rec["hbm_bytes_per_launch"])


if __name__ == "__main__":
    main()
