#!/bin/bash
# Development aid (GPU box): time the 16-wave STD192 kernel against the 8-wave build for several key-row pipeline depths.
# Rebuilds libbce_amd.so in the scratch copy for every variant; the committed build is whatever kernels64.hip defaults to.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
run() {
  echo "=== $1 (flags: $2, env: $3)"
  BCE_EXTRA_FLAGS="$2" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
  env $3 python tools/quick_perf_cfg.py STD192 AP 256 2>&1 | grep batch
  env $3 python tools/quick_perf_cfg.py STD192 GINX 256 2>&1 | grep batch
}
run "8 waves (r01 kernel)" "" "BCE_VARIANT=2"
run "16 waves, AP 2/2, GINX 1/1" "-DBCE_W16_NBUF_AP=2 -DBCE_W16_NPRE_AP=2 -DBCE_W16_NBUF_GINX=1 -DBCE_W16_NPRE_GINX=1" "BCE_VARIANT=3"
run "16 waves, AP 1/0, GINX 1/0 (shipped for AP)" "" "BCE_VARIANT=3"
run "16 waves, AP 2/0, GINX 1/0" "-DBCE_W16_NBUF_AP=2 -DBCE_W16_NPRE_AP=0 -DBCE_W16_NBUF_GINX=1 -DBCE_W16_NPRE_GINX=0" "BCE_VARIANT=3"
run "16 waves, AP 1/1, GINX 1/1" "-DBCE_W16_NBUF_AP=1 -DBCE_W16_NPRE_AP=1 -DBCE_W16_NBUF_GINX=1 -DBCE_W16_NPRE_GINX=1" "BCE_VARIANT=3"
