#!/bin/bash
# gpurun_out/<tag>/ of tools/collect_evidence_cfg5.sh -> profiles/<prefix>_cfg5_*: both roofs of the config-5 kernel
#   tools/make_profiles_cfg5.sh <tag> <prefix>          (CPU container; hipcc -S only)
set -eo pipefail
TAG=$1; PRE=$2
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/$TAG
P=$R/profiles
cp "$SRC/kernel_stats.csv" "$P/${PRE}_cfg5_kernel_stats.csv"
cp "$SRC/pmc_sq.json" "$P/${PRE}_cfg5_pmc_sq.json"
cp "$SRC/pmc_traffic.json" "$P/${PRE}_cfg5_pmc_traffic.json"
cp "$SRC/quick_perf.log" "$P/${PRE}_cfg5_quick_perf.log"
cd /tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only -o /tmp/bce_kernels64.s \
    "$R/openfhe-boolean-circuit-evaluator_amd/csrc/kernels64.hip" 2>/dev/null
python3 "$R/tools/valu_model.py" /tmp/bce_kernels64.s k_blind_rotate64dILi11ELi3ELb1ELb1ELb1ELb1ELb1E "$P/${PRE}_valu_issue.jsonl" "$P/${PRE}_cfg5_pmc_sq.json" \
    "wd::k_blind_rotate64d<11,3,AP,SPLIT,W16,FOLD,FUSE> (STD192 / AP, 1,024 threads, folded key, tail in the epilogue)" 4 > "$P/${PRE}_cfg5_valu_model.json"
python3 - "$P" "$PRE" <<'PY'
import csv, json, sys
P, PRE = sys.argv[1:3]
m = json.load(open("%s/%s_cfg5_valu_model.json" % (P, PRE)))
t = json.load(open("%s/%s_cfg5_pmc_traffic.json" % (P, PRE)))
rows = [r for r in csv.DictReader(open("%s/%s_cfg5_kernel_stats.csv" % (P, PRE))) if "k_blind_rotate64d" in r["Name"]]
r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
# quick_perf_cfg runs one warm-up launch (cold key) and one timed launch of 256 bootstraps: the timed one is MinNs, and it
# is the dispatch the PMC passes read (pmc_sq_summary.py / pmc_one_launch.py take the LAST dispatch)
launch_ms = float(r["MinNs"]) / 1e6
boots = m["pmc_launch_bootstraps"]
floor_ms = m["valu_insts_per_bootstrap"] * boots * m["ns_per_wave_inst_per_simd"] / (4 * 256) / 1e6
# algorithmic bytes per bootstrap: the engine's own figure (bce_bytes_per_bootstrap, SURVEY 8(d) formula at this build's
# widths: n*dR steps x one RGSW key of 2*dG rows x 2 polynomials x N doubles, + KSK rows + ciphertext I/O)
import re
alg = int(re.search(r"bytes/bootstrap (\d+)", open("%s/%s_cfg5_quick_perf.log" % (P, PRE)).read()).group(1))
out = {"kernel": r["Name"].split("(")[0], "bootstraps_per_launch": boots, "launches_in_stats": int(r["Calls"]), "launch_ms": launch_ms, "launch_ms_is": "MinNs of the kernel-stats row = the timed (second) launch",
       "valu": {"insts_per_bootstrap": m["valu_insts_per_bootstrap"], "ns_per_wave_inst_per_simd": m["ns_per_wave_inst_per_simd"],
                "floor_ms_per_launch": floor_ms, "frac": floor_ms / launch_ms,
                "note": "fp64 FMA/mul/add/rndne at 4.2 cycles per wave-instruction (profiles/%s_valu_issue.jsonl); mix of the step loop in %s_cfg5_valu_model.json" % (PRE, PRE)},
       "hbm": {"algorithmic_bytes_per_bootstrap": alg, "algorithmic_bytes_per_launch": alg * boots, "traffic_bytes_per_launch": t["hbm_bytes"],
               "achieved_GBps_algorithmic": alg * boots / launch_ms / 1e6, "achieved_GBps_traffic": t["hbm_bytes"] / launch_ms / 1e6,
               "peak_GBps": 8000.0, "frac_algorithmic": alg * boots / launch_ms / 1e6 / 8000.0, "frac_traffic": t["hbm_bytes"] / launch_ms / 1e6 / 8000.0}}
json.dump(out, open("%s/%s_cfg5_roofline.json" % (P, PRE), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
