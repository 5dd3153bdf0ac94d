"""Static itemisation of a blind-rotation kernel's step loop: VALU / LDS / VMEM / SALU instructions per phase.

Build the ISA with the phase marks in (-DBCE_PHASE_PROF: every BCE_PROF_MARK is one s_memtime), cut the step loop of the
named kernel at the marks and count instructions per segment.  The step loop holds both forward-transform bodies (whole
row, waves 0..3; half rows, waves 4..7): segments are reported with their label-delimited basic blocks so that the
either/or blocks can be told apart.

usage: phase_itemize.py <kernels_prof.s> <mangled-kernel-substring>
"""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from valu_model import kernel_body, step_loop, classify  # noqa: E402


def main():
    asm, key = sys.argv[1:3]
    loop = step_loop(kernel_body(asm, key))
    segs, cur, blocks = [], collections.Counter(), []
    blk = collections.Counter()
    mix = collections.Counter()
    for l in loop:
        m = re.match(r"^\s+([a-z_0-9]+)", l)
        if re.match(r"^\.LBB", l):
            if sum(blk.values()):
                blocks.append(dict(blk))
            blk = collections.Counter()
            continue
        if not m:
            continue
        op = m.group(1)
        if op == "s_memtime":
            if sum(blk.values()):
                blocks.append(dict(blk))
            segs.append((dict(cur), blocks, dict(mix)))
            cur, blocks, blk, mix = collections.Counter(), [], collections.Counter(), collections.Counter()
            continue
        c = classify(op)
        cur[c] += 1
        blk[c] += 1
        if c == "valu":
            mix[re.sub(r"_e(32|64)$", "", op)] += 1
    segs.append((dict(cur), blocks + ([dict(blk)] if sum(blk.values()) else []), dict(mix)))
    tot = collections.Counter()
    for i, (c, blocks, mix) in enumerate(segs):
        tot.update(c)
        top = ", ".join("%s %d" % kv for kv in sorted(mix.items(), key=lambda kv: -kv[1])[:8])
        print("segment %2d: valu %4d lds %3d vmem %3d salu %3d | valu per basic block %s" % (
            i, c.get("valu", 0), c.get("lds", 0), c.get("vmem", 0), c.get("salu", 0), [b.get("valu", 0) for b in blocks]))
        print("            %s" % top)
    print("step loop total:", dict(tot))


if __name__ == "__main__":
    main()
