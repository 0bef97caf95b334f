"""HBM traffic per launch of the build's kernels from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE passes, /opt/skills/guides/MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots").

    pmc_traffic.py <fetch_dir> <write_dir> <out.json> [calibration.json]

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-counts reads by a factor that depends on the access
pattern (the guide: exactly 1/2 for wide coalesced streaming reads; other patterns uncalibrated).  The factor applied
to a kernel is the one MEASURED for its pattern by tools/fetch_calib.hip (profiles/r02_fetch_calibration.json):
  gather : the local join's operand gather (random 512-byte rows, lane (r16, g) loads 16-byte chunks 4t + g)
  quad   : a quad (or a pair) of lanes per random 256-byte half-precision row (forest margins, finishers, routing passes)
  stream : everything else (16 B per lane, consecutive)
WRITE_SIZE is used uncorrected (uncalibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

PATTERN = (("k_local_join", "gather"), ("k_finalize", "gather"), ("k_leaf_join", "gather"), ("k_margin", "quad"),
           ("k_finish_subtrees", "quad"), ("k_hyperplane", "quad"), ("k_route", "quad"))


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


def main(fetch_dir, write_dir, out, calib=None):
    factors = {"stream": 2.0, "gather": 2.0, "quad": 2.0}
    source = "guide default x2 (no calibration file)"
    if calib and os.path.exists(calib):
        cj = json.load(open(calib))
        for key, name in (("k_calib_stream", "stream"), ("k_calib_gather", "gather"), ("k_calib_quad", "quad")):
            if key in cj:
                factors[name] = float(cj[key]["factor"])
        source = os.path.basename(calib)
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in fe:
        n, s = fe[k]
        wn, ws = wr.get(k, [0, 0.0])
        pat = next((p for sub, p in PATTERN if sub in k), "stream")
        f = factors[pat]
        res[k] = {"launches": n, "pattern": pat, "fetch_factor": f, "fetch_bytes_per_launch": round(f * s * 1024 / n),
                  "write_bytes_per_launch": round(ws * 1024 / wn) if wn else None,
                  "traffic_bytes_per_launch": round(f * s * 1024 / n + (ws * 1024 / wn if wn else 0))}
    res["_method"] = ("rocprofv3 --pmc FETCH_SIZE (pass 1), --pmc WRITE_SIZE (pass 2); bytes = (factor * FETCH_SIZE + WRITE_SIZE) KiB "
                      "* 1024 / launches; factors per access pattern from " + source + ": " + json.dumps(factors))
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main(*sys.argv[1:5])
