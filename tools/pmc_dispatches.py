"""Per-dispatch totals of one rocprofv3 --pmc counter for the bootstrapping kernels of a profiled run.
Usage: pmc_dispatches.py <dir> <COUNTER>"""
import collections, csv, glob, json, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
tot = collections.OrderedDict()
for r in csv.DictReader(open(fs[-1])):
    if r["Counter_Name"] != sys.argv[2]:
        continue
    k = r["Kernel_Name"].split("(")[0]
    if "blind_rotate" not in k and "bootstrap_dag" not in k:
        continue
    key = (int(r["Dispatch_Id"]), k[-60:], r["Grid_Size"])
    tot[key] = tot.get(key, 0.0) + float(r["Counter_Value"])
for (d, k, g), v in tot.items():
    print(json.dumps({"dispatch": d, "kernel": k, "grid_threads": g, sys.argv[2]: v}))
