#!/bin/bash
# Development aid (GPU box): rows in flight per lane in the fused tail (kernels.hip fused_tail) vs saturated launch time.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
for U in 8 16 4; do
  echo "=== BCE_FUSED_U=$U"
  BCE_EXTRA_FLAGS="-DBCE_FUSED_U=$U" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
  python tools/quick_perf.py 6144 2>&1 | grep batch
  python tools/quick_perf.py 6144 2>&1 | grep batch
done
