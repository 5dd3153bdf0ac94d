#!/bin/bash
# Round evidence in one GPU call (run from the repo root on the GPU box):
#   tools/collect_evidence.sh <tag> [profile-only]    -> everything under gpurun_out/<tag>/
# GPU tests, the default bench line, rocprofv3 kernel stats of the same bench command, HBM traffic
# from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), SQ/LDS counters of a saturated launch.
set -eo pipefail
TAG=${1:-evidence}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
if [ "$2" != "profile-only" ]; then
  python -m pytest tests -m gpu -x -q 2>&1 | tee "$OUT/gpu_tests.log" | tail -3
  python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  tail -c 600 "$OUT/bench.json"; echo
fi
export TMPDIR=/tmp
cd /tmp
# --fresh-inputs: the refresh bootstraps of SetInput (reference-shaped BOOTSTRAPPED default) run on the same kernel as the
# timed steps; without them the kernel statistics and the per-launch counters below hold the 416 timed launches only
BENCH1="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-block-latency --no-dataflow-leg --fresh-inputs"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH1 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv"
head -5 "$OUT/bench_kernel_stats.csv"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH1 > "$OUT/pmc_fetch.out" 2> "$OUT/pmc_fetch.err"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH1 > "$OUT/pmc_write.out" 2> "$OUT/pmc_write.err"
echo "write pass done"
python3 "$R/tools/pmc_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_raw.json" > /dev/null
QP="python3 $R/tools/quick_perf.py 6144"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_sq1" -- $QP > "$OUT/pmc_sq1.out" 2> "$OUT/pmc_sq1.err"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq2" -- $QP > "$OUT/pmc_sq2.out" 2> "$OUT/pmc_sq2.err"
python3 "$R/tools/pmc_sq_summary.py" "$OUT/pmc_sq1" "$OUT/pmc_sq2" > "$OUT/pmc_sq_lds.json"
cat "$OUT/pmc_sq_lds.json"
# the raw per-dispatch CSVs are large: keep the summaries only
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq1" "$OUT/pmc_sq2"
echo "evidence collected in $OUT"
