#!/bin/bash
# Development aid (GPU box): MAC tail of the split-transform kernels in Montgomery form (default) against round 3's
# fold + Barrett form (-DBCE_BARRETT_TAIL), same box, alternating builds; saturated 6,144-bootstrap launches and lone ones.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for F in "-DBCE_BARRETT_TAIL" ""; do
    echo "=== flags: '$F'"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    python tools/quick_perf.py 1 256 512 6144 2>&1 | grep batch
  done
done
BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
