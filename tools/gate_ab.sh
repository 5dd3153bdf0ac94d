#!/bin/bash
# Development aid (GPU box): per-launch XCD start gate (BCE_XCD_GATE=1) against the plain launch: time and FETCH_SIZE of saturated
# multi-round STD128_OPT launches, alternating.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for rep in 1 2 3; do
  for G in 0 1; do
    echo "=== BCE_XCD_GATE=$G BCE_XCD_GATE_US=${BCE_XCD_GATE_US:-100}"
    cd "$R"; BCE_XCD_GATE=$G python tools/quick_perf.py 2048 6144 2>&1 | grep batch
  done
done
for G in 0 1; do
  cd /tmp; rm -rf /tmp/gate_ab
  BCE_XCD_GATE=$G rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/gate_ab -- python3 "$R/tools/quick_perf.py" 6144 > /tmp/gate_ab.out 2> /tmp/gate_ab.err
  python3 - "$G" <<'PY'
import csv, glob, os, sys
fs = sorted(glob.glob("/tmp/gate_ab/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(fs[-1])) if "blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
by = {}
for r in rows:
    by[int(r["Dispatch_Id"])] = by.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
print("gate %s: FETCH_SIZE x2 of the last launches: %s GB" % (sys.argv[1], ", ".join("%.1f" % (2 * v * 1024 / 1e9) for k, v in sorted(by.items())[-3:])))
PY
done
