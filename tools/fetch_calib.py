"""Run on the GPU box: builds tools/fetch_calib.hip, profiles it with --pmc FETCH_SIZE and writes the calibration factors
(bytes actually read / FETCH_SIZE bytes) per access pattern to the given JSON file (commit it under profiles/)."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(out):
    exe = "/tmp/fetch_calib"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", os.path.join(ROOT, "tools", "fetch_calib.hip"), "-o", exe])
    d = "/tmp/calib_prof"
    subprocess.call(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", "FETCH_SIZE", "--output-format", "csv", "-d", d, "--", exe], cwd="/tmp", env=env)
    known = {"k_calib_stream": 4 * 1024 * 1024 * 512, "k_calib_gather": 4 * 1024 * 1024 * 512, "k_calib_quad": 4 * 1024 * 1024 * 256}
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                name = r["Kernel_Name"].split("(")[0]
                agg.setdefault(name, []).append(float(r["Counter_Value"]))
    res = {"_method": "tools/fetch_calib.hip: every byte of a 2 GiB table (1 GiB for the bf16 quad pattern) read exactly once; "
                      "factor = bytes read / (FETCH_SIZE KiB * 1024); rocprofv3 --pmc FETCH_SIZE, mean of the launches"}
    for name, vals in agg.items():
        for key, nbytes in known.items():
            if key in name:
                mean_kib = sum(vals) / len(vals)
                res[key] = {"launches": len(vals), "fetch_size_kib": mean_kib, "bytes_read": nbytes,
                            "factor": round(nbytes / (mean_kib * 1024.0), 4)}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1])
