"""Condense rocprofv3 CSV output (kernel trace / stats / counter collection) into small text
summaries that can be committed under profiles/.  Usage: prof_summary.py <dir> <out.txt>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    return name[-70:]


def main(d, out):
    lines = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
        lines.append("== kernel stats: %s" % os.path.basename(f))
        with open(f) as fh:
            rows = list(csv.DictReader(fh))
        for r in rows[:40]:
            lines.append("%-72s calls %6s total_ns %12s avg_ns %12s pct %6s" % (
                short(r.get("Name", "")), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))
    for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        lines.append("== kernel trace: %s" % os.path.basename(f))
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
            lines.append("%-72s calls %6d total_us %12.1f avg_us %10.2f" % (k, c, t / 1e3, t / 1e3 / c))
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        meta = {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = short(r["Kernel_Name"])
                a = agg[k][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
                meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                           r.get("Workgroup_Size"))
        lines.append("== counters: %s" % os.path.basename(f))
        for k in sorted(agg, key=lambda kk: -max(v[1] for v in agg[kk].values())):
            lines.append("%s  [vgpr %s agpr %s sgpr %s lds %s wg %s]" % ((k,) + meta[k]))
            for cn, (c, s) in sorted(agg[k].items()):
                lines.append("    %-28s dispatches %5d sum %16.0f mean %16.1f" % (cn, c, s, s / c))
    with open(out, "w") as fh:
        fh.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
