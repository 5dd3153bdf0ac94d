#!/bin/bash
# LDS counters of one saturated launch of the headline kernel with one phase's LDS traffic compiled out at a time
# (development: the results of the modified builds are WRONG, only their counters are read).  GPU box, repo root:
#   bash tools/lds_attribution.sh <tag>
set -eo pipefail
TAG=${1:-ldsattr}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for V in none SKIP_FWD SKIP_MAC; do
  cd "$R"
  if [ "$V" = none ]; then BCE_EXTRA_FLAGS="" python3 openfhe-boolean-circuit-evaluator_amd/build.py --force > "$OUT/build_$V.log" 2>&1
  else BCE_EXTRA_FLAGS="-DBCE_$V" python3 openfhe-boolean-circuit-evaluator_amd/build.py --force > "$OUT/build_$V.log" 2>&1; fi
  cd /tmp
  python3 $R/tools/quick_perf.py 6144 > "$OUT/qp_$V.log" 2>&1 || true
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/p_$V" -- python3 $R/tools/quick_perf.py 6144 > "$OUT/p_$V.out" 2> "$OUT/p_$V.err" || true
  python3 "$R/tools/pmc_sq_summary.py" "$OUT/p_$V" > "$OUT/lds_$V.json"
  rm -rf "$OUT/p_$V"
  echo "== $V"; tail -1 "$OUT/qp_$V.log"; cat "$OUT/lds_$V.json"
done
cd "$R" && BCE_EXTRA_FLAGS="" python3 openfhe-boolean-circuit-evaluator_amd/build.py --force > "$OUT/build_restore.log" 2>&1
