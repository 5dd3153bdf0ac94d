#!/bin/bash
# Development aid (GPU box): the default bench workload (AES-expanded, K = 32) under both barrier schedules, alternating builds.
set -eo pipefail
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r02ab
for rep in 1 2; do
for F in "-DBCE_STEP_BARRIERS" ""; do
  BCE_EXTRA_FLAGS="$F" python openfhe-boolean-circuit-evaluator_amd/build.py --force > /dev/null
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02ab/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r02ab/b.json')); print('flags [$F]', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['single_block_latency_s'])"
done
done
