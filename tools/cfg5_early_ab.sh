#!/bin/bash
# Development aid (GPU box): same-box A/B of the 16-wave STD192/AP kernel with and without the early key-row requests
# (BCE_W16_EARLY).  Rebuilds libbce_amd.so in the scratch copy per variant; the committed build is the source default.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for rep in 1 2; do
  for v in 0 1; do
    echo "=== BCE_W16_EARLY=$v (pass $rep)"
    BCE_EXTRA_FLAGS="-DBCE_W16_EARLY=$v" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
    python tools/quick_perf_cfg.py STD192 AP 256 1024 2>&1 | grep batch
  done
done
python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
