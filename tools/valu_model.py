"""Integer-VALU roof of a blind-rotation kernel, from three committed measurements:

  (1) the instruction MIX of the kernel's step loop, read from the gfx950 ISA the compiler emits for it
      (hipcc -S --cuda-device-only of csrc/kernels*.hip; the step loop is the largest loop of the kernel),
  (2) the measured issue cost of every opcode in that mix (tools/valu_issue.hip: ns per wave-instruction per SIMD at a
      pinned number of resident waves per SIMD; opcodes not measured take the cost of their encoding class),
  (3) SQ_INSTS_VALU of one saturated launch (rocprofv3 --pmc pass, tools/pmc_sq_summary.py) = executed VALU
      wave-instructions per bootstrap.

Output (JSON): weighted ns per wave-instruction per SIMD of this kernel's mix and the wave-instructions per bootstrap;
bench.py turns them into `roofline.valu`: floor = insts/bootstrap x bootstraps/launch x ns / SIMDs, frac = floor / measured.

usage: valu_model.py <kernels.s> <mangled-kernel-substring> <valu_issue.jsonl> <pmc_sq.json> <bench kernel label> [waves_per_simd]
"""
import collections
import json
import re
import sys


def kernel_body(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(key), l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    # the function may continue after the first s_endpgm (cold blocks); stop at .Lfunc_end
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def step_loop(body):
    """largest backward-branch span of the function"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = (0, 0, 0)
    for i, l in enumerate(body):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = i - labels[m.group(1)]
            if span > best[0]:
                best = (span, labels[m.group(1)], i)
    return body[best[1]:best[2] + 1]


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    asm, key, issue_path, pmc_path, label = sys.argv[1:6]
    wps = float(sys.argv[6]) if len(sys.argv) > 6 else 4.0
    loop = step_loop(kernel_body(asm, key))
    mix = collections.Counter()
    kinds = collections.Counter()
    for l in loop:
        m = re.match(r"^\s+([a-z_0-9]+)\b", l)
        if not m or l.lstrip().startswith((";", ".")):
            continue
        op = m.group(1)
        kinds[classify(op)] += 1
        if classify(op) == "valu":
            mix[op] += 1
    # measured issue costs at the requested residency
    cost = {}
    for l in open(issue_path):
        l = l.strip()
        if l.startswith("{"):
            d = json.loads(l)
            # v_cndmask is left to its encoding class: the micro-benchmark reads a VCC nothing writes, and its 19-cycle
            # figure is that artefact, not the cost of a v_cmp / v_cndmask pair in real code
            if d["waves_per_simd"] == wps and not d["op"].startswith("v_cndmask"):
                cost[d["op"]] = d["ns_per_wave_inst_per_simd"]
    def base(op):
        return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    by_base = {base(k): v for k, v in cost.items()}
    mul_class = by_base.get("v_mul_lo_u32")
    vop3_class = by_base.get("v_lshl_add_u32")
    vop2_class = by_base.get("v_add_u32")
    def price(op):
        b = base(op)
        if b in by_base:
            return by_base[b], "measured"
        if b.startswith(("v_mul_", "v_mad_u64", "v_mad_i64")):
            return mul_class, "class: 32-bit multiply"
        if op.endswith("_e64") or b.startswith(("v_add3", "v_lshl_", "v_add_lshl", "v_bfe", "v_alignbit", "v_perm", "v_mad_", "v_fma_", "v_lshl_or", "v_and_or", "v_or3", "v_xad", "v_min3", "v_max3", "v_med3", "v_cvt", "v_rcp", "v_readfirstlane", "v_readlane", "v_writelane", "v_cmp", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64")):
            return vop3_class, "class: VOP3"
        return vop2_class, "class: VOP2"
    total = sum(mix.values())
    weighted = sum(n * price(op)[0] for op, n in mix.items()) / total
    pmc = json.load(open(pmc_path))
    boots = int(pmc["_grid_threads_workgroup_threads"][0]) // int(pmc["_grid_threads_workgroup_threads"][1])
    out = {
        "bench_kernel": label,
        "isa_kernel": key,
        "waves_per_simd": wps,
        "step_loop_instructions": dict(kinds),
        "step_loop_valu_mix": {op: {"count": n, "ns": price(op)[0], "how": price(op)[1]} for op, n in mix.most_common()},
        "ns_per_wave_inst_per_simd": weighted,
        "valu_insts_per_bootstrap": pmc["SQ_INSTS_VALU"] / boots,
        "pmc_launch_bootstraps": boots,
        "cu_count": 256,
        "sources": {"isa": "hipcc -S --cuda-device-only csrc/kernels.hip", "issue_costs": issue_path, "pmc": pmc_path},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
