"""HBM traffic per launch of the build's kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes, /opt/skills/guides/MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots").

    pmc_traffic.py <fetch_dir> <write_dir> <out.json>

Corrections applied as that guide prescribes for gfx950: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts
128-byte requests of wide (16 B/lane) coalesced loads at 64 B, so it is DOUBLED for kernels whose reads are
16-byte-per-lane row streams / gathers (all of ours).  WRITE_SIZE is used uncorrected (uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    a = agg[r["Kernel_Name"].split("(")[0]]
                    a[0] += 1
                    a[1] += float(r["Counter_Value"])
    return agg


def main(fetch_dir, write_dir, out):
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in fe:
        n, s = fe[k]
        wn, ws = wr.get(k, [0, 0.0])
        res[k] = {"launches": n, "fetch_bytes_per_launch": round(2.0 * s * 1024 / n),
                  "write_bytes_per_launch": round(ws * 1024 / wn) if wn else None,
                  "traffic_bytes_per_launch": round(2.0 * s * 1024 / n + (ws * 1024 / wn if wn else 0))}
    res["_method"] = "rocprofv3 --pmc FETCH_SIZE (pass 1), --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (pass 2); " \
                     "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 / launches (gfx950 halves wide-load fetches)"
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:4])
