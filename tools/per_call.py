"""Per-dispatch durations of selected kernels from a rocprofv3 kernel trace: per_call.py <dir> <substr> [<substr> ...]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for pat in sys.argv[2:]:
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
        print(pat, " ".join("%.0f" % v for v in durs))
