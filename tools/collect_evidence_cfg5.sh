#!/bin/bash
# Counters of ONE saturated launch of the BASELINE config-5 kernel (STD192, AP; CFG5_BATCH bootstraps, default 1024 = four rounds of one workgroup per CU):
#   tools/collect_evidence_cfg5.sh <tag>   -> gpurun_out/<tag>/  (run from the repo root on the GPU box)
# kernel time (rocprofv3 kernel stats), executed VALU wave-instructions (SQ pass), fabric traffic (FETCH_SIZE / WRITE_SIZE
# in separate passes).  Post-processing into profiles/: tools/make_profiles_cfg5.sh.
set -eo pipefail
TAG=${1:-cfg5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
QP="python3 $R/tools/quick_perf_cfg.py STD192 AP ${CFG5_BATCH:-1024}"
$QP > "$OUT/quick_perf.log" 2>&1
cat "$OUT/quick_perf.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $QP > "$OUT/stats.out" 2> "$OUT/stats.err"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq1" -- $QP > "$OUT/pmc_sq1.out" 2> "$OUT/pmc_sq1.err"
python3 "$R/tools/pmc_sq_summary.py" "$OUT/pmc_sq1" > "$OUT/pmc_sq.json"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $QP > "$OUT/pmc_fetch.out" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $QP > "$OUT/pmc_write.out" 2> "$OUT/pmc_write.err"
python3 "$R/tools/pmc_one_launch.py" "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/pmc_traffic.json"
cat "$OUT/pmc_sq.json" "$OUT/pmc_traffic.json"
rm -rf "$OUT/stats" "$OUT/pmc_sq1" "$OUT/pmc_fetch" "$OUT/pmc_write"
echo "config-5 evidence collected in $OUT"
