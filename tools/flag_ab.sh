#!/bin/bash
# Development aid (GPU box): same-box alternating A/B of compile-time switches on saturated and lone STD128_OPT launches.
# usage: flag_ab.sh "<flags A>" "<flags B>" [...]      ("" = the shipped build)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for F in "$@"; do
    echo "=== flags: '$F'"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    python tools/quick_perf.py 1 512 6144 2>&1 | grep batch
  done
done
BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
