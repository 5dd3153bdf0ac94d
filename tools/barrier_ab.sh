#!/bin/bash
# Development aid (GPU box): round-1 barrier schedule of the split inverse transforms (-DBCE_STEP_BARRIERS: every
# exchange behind a workgroup barrier) vs wave-local exchanges, same box, alternating builds.
# usage: barrier_ab.sh [std128|std192]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
what=${1:-std128}
for rep in 1 2; do
  for F in "-DBCE_STEP_BARRIERS" ""; do
    echo "=== flags: '$F'"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    if [ "$what" = std128 ]; then
      python tools/quick_perf.py 1 128 256 512 6144 2>&1 | grep batch
    else
      python tools/quick_perf_cfg.py STD192 GINX 1 256 512 2>&1 | grep batch
      python tools/quick_perf_cfg.py STD192 AP 1 256 512 2>&1 | grep batch
    fi
  done
done
BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
