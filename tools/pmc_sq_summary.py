"""Sum rocprofv3 --pmc SQ_* counters over the dispatches of the largest blind-rotation launch of a run
(tools/quick_perf.py <batch>): usage pmc_sq_summary.py <pmc_dir> [<pmc_dir> ...] -> JSON on stdout."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    out = {}
    kernel = None
    for d in sys.argv[1:]:
        fs = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
        rows = [r for r in csv.DictReader(open(fs[-1])) if "blind_rotate" in r["Kernel_Name"]]
        # the timed launches of quick_perf are the last dispatches; take the LAST dispatch id
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                kernel = r["Kernel_Name"].split("(")[0]
                grid = r.get("Grid_Size"), r.get("Workgroup_Size")
    out["_kernel"] = kernel
    out["_grid_threads_workgroup_threads"] = grid
    out["_note"] = "one dispatch (the last blind-rotation launch of tools/quick_perf.py); SQ_* wave counters are in quad-cycles per the microarchitecture guide"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
