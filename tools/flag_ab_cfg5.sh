#!/bin/bash
# Development aid (GPU box): same-box alternating A/B of compile-time switches on the config-5 kernel (STD192 / AP) and the
# STD192 / GINX one: 256- and 1,024-bootstrap launches.   usage: flag_ab_cfg5.sh "<flags A>" "<flags B>" [...]   ("" = shipped)
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for F in "$@"; do
    echo "=== flags: '$F'"
    BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    python tools/quick_perf_cfg.py STD192 AP 256 1024 2>&1 | grep batch
    python tools/quick_perf_cfg.py STD192 GINX 256 1024 2>&1 | grep batch
  done
done
BCE_EXTRA_FLAGS="" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
