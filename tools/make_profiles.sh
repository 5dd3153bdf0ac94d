#!/bin/bash
# Turn one tools/collect_evidence.sh result (gpurun_out/<tag>/) into the committed summaries under profiles/:
#   tools/make_profiles.sh <tag> <prefix>      e.g.  tools/make_profiles.sh r02d r02
# Runs on the CPU container: the only compute is hipcc -S of the kernels (for the instruction mix of the VALU model).
set -eo pipefail
TAG=$1; PRE=$2
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/gpurun_out/$TAG
P=$R/profiles
cp "$SRC/bench.json" "$P/${PRE}_bench.json"
cp "$SRC/bench_kernel_stats.csv" "$P/${PRE}_bench_kernel_stats.csv"
cp "$SRC/bench_under_rocprof.json" "$P/${PRE}_bench_under_rocprof.json"
cp "$SRC/gpu_tests.log" "$P/${PRE}_gpu_tests.log"
cp "$SRC/pmc_sq_lds.json" "$P/${PRE}_pmc_sq_lds.json"
python3 - "$SRC/pmc_traffic_raw.json" "$P/${PRE}_pmc_traffic.json" <<'PY'
import json, sys
raw = json.load(open(sys.argv[1]))
br = next(k for k in raw if "blind_rotate" in k)
tails = {k: v for k, v in raw.items() if "k_tail" in k}
label = "k_blind_rotate_lat<4,4> (split transform, 2 workgroups/CU)" if br.startswith("void bce::k_blind_rotate_lat<4, 4, false") else br
out = {"kernel": br, "bench_kernel": label,
       "workload": "AES-expanded.txt STD128_OPT GINX instances_per_gpu=32, bootstrap-depth schedule (bench.py default; inputs encrypted FRESH so that the kernel statistics hold the timed launches only), 1 step",
       "instances_per_gpu": 32, "relevel": True,
       "hbm_bytes_per_launch": raw[br]["hbm_bytes_per_launch"], "launches": raw[br]["launches"],
       "fetch_kib_raw_per_launch": raw[br]["fetch_kib_raw_per_launch"], "write_kib_per_launch": raw[br]["write_kib_per_launch"],
       "tail_hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in tails.values()),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `python3 bench.py --steps 1 --warmup 0 "
                 "--no-cpu-baseline --no-block-latency --fresh-inputs` (tools/collect_evidence.sh); FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE as is; "
                 "KiB -> bytes; Infinity-Cache hits are included in these fabric-side counters",
       "tail_kernels": tails}
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
cd /tmp
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only -o /tmp/bce_kernels.s \
    "$R/openfhe-boolean-circuit-evaluator_amd/csrc/kernels.hip" 2>/dev/null
python3 "$R/tools/valu_model.py" /tmp/bce_kernels.s k_blind_rotate_latILi4ELi4ELb0ELb1ELb1E "$P/${PRE}_valu_issue.jsonl" "$P/${PRE}_pmc_sq_lds.json" \
    "k_blind_rotate_lat<4,4> (split transform, 2 workgroups/CU)" 4 > "$P/${PRE}_valu_model.json"
python3 - "$P/${PRE}_valu_model.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("valu model: %.0f insts/bootstrap, %.3f ns per wave-inst per SIMD, step loop %s" % (d["valu_insts_per_bootstrap"], d["ns_per_wave_inst_per_simd"], d["step_loop_instructions"]))
PY
