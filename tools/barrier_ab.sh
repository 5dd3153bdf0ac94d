#!/bin/bash
# Development aid (GPU box): round-1 barrier schedule of the split inverse transform (6 workgroup barriers per step,
# -DBCE_STEP_BARRIERS) vs the wave-local exchanges (3 per step), same box, alternating builds.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for F in "-DBCE_STEP_BARRIERS" ""; do
    echo "=== flags: '$F'"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    python tools/quick_perf.py 1 128 256 512 6144 2>&1 | grep batch
  done
done
BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
