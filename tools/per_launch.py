"""Per-LAUNCH durations of the kernels whose name contains <substr>, in launch order, from a rocprofv3 --kernel-trace directory.
usage: per_launch.py <dir> <substr> [max]"""
import csv, glob, os, sys
d, sub = sys.argv[1], sys.argv[2]
mx = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if sub in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
print(sub, "launches", len(rows), "us:", [round(t / 1e3, 1) for _, t, _ in rows[:mx]])
