"""Condenses a rocprofv3 output directory (kernel-trace/--stats CSVs, optional PMC CSVs) into a small text
summary for profiles/.  Usage: python tools/rocprof_summary.py <rocprof_out_dir> [label]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else d
    print("# rocprofv3 summary: %s" % label)
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        print("\n## kernel stats (%s)" % os.path.basename(f))
        rows = list(csv.DictReader(open(f)))
        print("%-60s %10s %14s %12s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
        for r in rows[:12]:
            name = r.get("Name", "")[:60]
            print("%-60s %10s %14s %12s %8s" % (name, r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0, 0, 0, 0, 0, 0, 0])

        def num(r, *names):
            for nm in names:
                v = r.get(nm)
                if v not in (None, ""):
                    return int(float(v))
            return 0
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            a = agg[k]
            a[0] += 1
            a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a[2] = max(a[2], num(r, "VGPR_Count", "Arch_VGPR_Count"))           # the ALLOCATION granule (multiples of 8), not the compiler's count: tools/kres.py prints that
            a[3] = max(a[3], num(r, "LDS_Block_Size"))
            a[4] = max(a[4], num(r, "Workgroup_Size", "Workgroup_Size_X"))
            a[5] = max(a[5], num(r, "Accum_VGPR_Count"))
            a[6] = max(a[6], num(r, "SGPR_Count"))
            a[7] = max(a[7], num(r, "Scratch_Size", "Private_Segment_Size", "Scratch_Memory_Size"))
        print("\n## kernel trace aggregate (%s)" % os.path.basename(f))
        print("%-60s %10s %14s %12s %9s %10s %5s %8s %8s %6s" % ("kernel", "calls", "total_ns", "avg_ns", "arch_vgpr", "accum_vgpr", "sgpr", "scratch", "lds", "wg"))
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
            print("%-60s %10d %14d %12.1f %9d %10d %5d %8d %8d %6d" % (k, a[0], a[1], a[1] / a[0], a[2], a[5], a[6], a[7], a[3], a[4]))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            a = agg[r["Kernel_Name"][:60]][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print("\n## PMC counters (%s): per-dispatch mean" % os.path.basename(f))
        for k, cs in agg.items():
            for c, a in cs.items():
                print("%-60s %-20s dispatches=%-8d mean=%.1f total=%.1f" % (k, c, a[0], a[1] / a[0], a[1]))


if __name__ == "__main__":
    main()
