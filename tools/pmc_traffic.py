"""Turn rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, separate runs) of bench.py into the
per-launch HBM traffic of the blind-rotation kernel, with the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read stream; both counters are
in KiB).  Usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import os
import sys


def per_kernel(d, counter):
    fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    tot = collections.defaultdict(float)
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] == counter:
            tot[r["Kernel_Name"]] += float(r["Counter_Value"])
            disp[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (v, len(disp[k])) for k, v in tot.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k in fetch:
        if "blind_rotate" not in k and "k_tail" not in k:
            continue
        f, n = fetch[k]
        w, _ = write.get(k, (0.0, n))
        out[k.split("(")[0].strip()] = {
            "launches": n,
            "fetch_kib_raw_per_launch": f / n,
            "write_kib_per_launch": w / n,
            "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0 / n,
            "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is; KiB -> bytes",
        }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
