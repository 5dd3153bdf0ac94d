#!/bin/bash
# Development aid (GPU box): FETCH_SIZE of one saturated 6,144-bootstrap STD128_OPT launch under alternating builds.
# usage: fetch_ab.sh "<flags A>" "<flags B>" ...      ("" = the shipped build)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for rep in 1 2; do
  for F in "$@"; do
    cd "$R"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    cd /tmp
    rm -rf /tmp/fetch_ab
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fetch_ab -- python3 "$R/tools/quick_perf.py" ${FETCH_AB_BATCH:-6144} > /tmp/fetch_ab.out 2> /tmp/fetch_ab.err
    python3 - "$F" <<'PY'
import csv, glob, os, sys
fs = sorted(glob.glob("/tmp/fetch_ab/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(fs[-1])) if "blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), 0.0)
    by[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
big = [v for k, v in sorted(by.items())][-3:]
print("flags %-22r FETCH_SIZE x2 of the last saturated launches: %s GB" % (sys.argv[1], ", ".join("%.1f" % (2 * v * 1024 / 1e9) for v in big)))
PY
    grep batch /tmp/fetch_ab.out | tail -1
  done
done
cd "$R"; BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
