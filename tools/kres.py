#!/usr/bin/env python3
"""kres.py -- per-kernel register / spill / scratch table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/kres.py pgvector-rx_amd/csrc/hx_fused_f32.hip [substring ...]      (extra HX_CFLAGS honoured)

Compiles the translation unit for gfx950 with the library's own flags (nothing is linked or run) and prints one line per kernel:
SGPRs, VGPRs (arch), AGPRs, SGPR spills, VGPR spills, scratch bytes per lane, occupancy (waves/SIMD).  The numbers the
kernel-trace CSV of rocprofv3 does NOT show (it reports the allocation granule, and no scratch)."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-x", "hip"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def resources(src, extra=()):
    with tempfile.TemporaryDirectory() as td:
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + list(extra) + os.environ.get("HX_CFLAGS", "").split() + \
              ["-c", src, "-o", os.path.join(td, "x.o"), "-Rpass-analysis=kernel-resource-usage"]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark: [^:]*:\d+:\d+: +(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+) \[", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    return rows


def main():
    src = sys.argv[1]
    subs = sys.argv[2:]
    rows = resources(src)
    dm = demangle([r["name"] for r in rows])
    print("%-7s %-6s %-6s %-10s %-10s %-8s %-5s  kernel" % ("SGPRs", "VGPRs", "AGPRs", "SGPRspill", "VGPRspill", "scratch", "occ"))
    for r in rows:
        name = dm.get(r["name"], r["name"])
        if subs and not any(s in name for s in subs):
            continue
        print("%-7d %-6d %-6d %-10d %-10d %-8d %-5d  %s" % (r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("SGPRs Spill", 0), r.get("VGPRs Spill", 0),
                                                         r.get("ScratchSize [bytes/lane]", 0), r.get("Occupancy [waves/SIMD]", 0), name[:150]))


if __name__ == "__main__":
    main()
