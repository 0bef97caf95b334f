"""Timeline of the forest stage from a rocprofv3 kernel trace of tools/qbench.py: for the LAST build in the trace, the phases of the
forest (prep, sample gather, sample-forest levels, recording finishers, routing, cell finishers) with their wall time, summed kernel
time and launch counts -- where the stage's time is gaps between tiny launches rather than kernels.  usage: forest_timeline.py <trace dir>"""
import csv
import glob
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[-60:]


def main(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # the last build: from the last k_prep_rows* to the last k_finalize*
    i0 = max(i for i, r in enumerate(rows) if r[2].startswith("k_prep_rows") or r[2].startswith("k_colsum_partial"))
    while i0 > 0 and rows[i0 - 1][2].startswith("k_colsum"):
        i0 -= 1
    i1 = max(i for i, r in enumerate(rows) if r[2].startswith("k_finalize"))
    b = rows[i0:i1 + 1]
    t0 = b[0][0]
    first_leaf = next(i for i, r in enumerate(b) if r[2].startswith("k_leaf_join"))
    forest = b[:first_leaf]
    print("build: %d launches, %.3f ms wall, %.3f ms in kernels" % (len(b), (b[-1][1] - t0) / 1e6, sum(r[1] - r[0] for r in b) / 1e6))
    print("before the first leaf kernel: %d launches, %.3f ms wall, %.3f ms in kernels" % (
        len(forest), (forest[-1][1] - t0) / 1e6, sum(r[1] - r[0] for r in forest) / 1e6))
    # phases by marker kernels
    def idx(pred, start=0):
        for i in range(start, len(forest)):
            if pred(forest[i][2]):
                return i
        return None
    marks = [("prep", 0)]
    g = idx(lambda s: s.startswith("k_gather_sample"))
    if g is not None:
        marks.append(("sample gather + levels + recording finishers", g))
    rt = idx(lambda s: s.startswith("k_route_top"))
    if rt is not None:
        marks.append(("routing (top, bucket, place)", rt))
        fin = idx(lambda s: s.startswith("k_finish_subtrees"), rt)
        if fin is not None:
            marks.append(("cell finishers + leaf tables", fin))
    marks.append(("end", len(forest)))
    for (name, a), (_, e) in zip(marks[:-1], marks[1:]):
        seg = forest[a:e]
        if not seg:
            continue
        wall = ((forest[e][0] if e < len(forest) else seg[-1][1]) - seg[0][0]) / 1e6
        kt = sum(r[1] - r[0] for r in seg) / 1e6
        tiny = [r for r in seg if r[1] - r[0] < 20000]
        print("  %-48s %4d launches  wall %.3f ms  kernels %.3f ms  (gaps %.3f)  launches under 20 us: %d, %.3f ms" % (
            name, len(seg), wall, kt, wall - kt, len(tiny), sum(r[1] - r[0] for r in tiny) / 1e6))
    # the sample-forest levels: per level = from one k_hyperplane to the next
    if g is not None and rt is not None:
        lv = [i for i in range(g, rt) if forest[i][2].startswith("k_hyperplane")]
        for q, i in enumerate(lv):
            j = lv[q + 1] if q + 1 < len(lv) else rt
            seg = forest[i:j]
            print("    level %2d: %2d launches  wall %6.1f us  kernels %6.1f us  : %s" % (
                q, len(seg), (forest[j][0] - seg[0][0]) / 1e3, sum(r[1] - r[0] for r in seg) / 1e3,
                " ".join("%s=%.0f" % (r[2].split("<")[0].replace("k_", ""), (r[1] - r[0]) / 1e3) for r in seg)[:230]))


if __name__ == "__main__":
    main(sys.argv[1])
