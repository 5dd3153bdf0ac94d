"""HBM/fabric bytes of the LAST blind-rotation dispatch of a profiled run: usage
pmc_one_launch.py <fetch_dir> <write_dir>   (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, csv output)"""
import csv
import glob
import json
import os
import sys


def last(d, counter):
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(fs[-1])) if "blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    did = max(int(r["Dispatch_Id"]) for r in rows)
    return sum(float(r["Counter_Value"]) for r in rows if int(r["Dispatch_Id"]) == did), rows[-1]["Kernel_Name"].split("(")[0], rows[-1]["Grid_Size"]


f, k, g = last(sys.argv[1], "FETCH_SIZE")
w, _, _ = last(sys.argv[2], "WRITE_SIZE")
print(json.dumps({"kernel": k, "grid_threads": g, "fetch_kib_raw": f, "write_kib": w, "hbm_bytes": (2 * f + w) * 1024,
                  "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is; KiB -> bytes"}))
