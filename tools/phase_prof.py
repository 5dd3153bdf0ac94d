"""Per-phase cycle split of a blind-rotation step (development aid, GPU box only).

Rebuilds libbce_amd.so IN THE SCRATCH COPY with -DBCE_PHASE_PROF (the committed build never defines
it), runs batched STD128_OPT/GINX NAND bootstraps and prints, for workgroup 0, the cycles its thread 0
spent in each barrier-delimited phase.  Usage: python tools/phase_prof.py [STD192|STD192_AP] [batch ...]
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, BCE_EXTRA_FLAGS=(os.environ.get("BCE_EXTRA_FLAGS", "") + " -DBCE_PHASE_PROF").strip())   # other development flags ride along
subprocess.check_call([sys.executable, os.path.join(ROOT, "openfhe-boolean-circuit-evaluator_amd", "build.py"), "--force"],
                      env=env, stdout=subprocess.DEVNULL)
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
lib = C.CDLL(bce.LIB_PATH)
NAMES = ["phase 1 (thread 0: loads issue, inverse NTT, digits)", "wait barrier 1", "forward NTT", "wait barrier 2", "RGSW MAC", "wait barrier 3"]


def main():
    args = sys.argv[1:]
    ps, method, getter = "STD128_OPT", "GINX", lib.bce_debug_phase_prof
    if args and args[0] in ("STD192", "STD192_AP"):   # 64-bit modulus kernel (kernels64.hip)
        ps, method, getter = "STD192", ("AP" if args[0].endswith("AP") else "GINX"), lib.bce_debug_phase_prof64
        args = args[1:]
    batches = [int(x) for x in args] or [1, 256]
    c = bce.BinFHEContext(getattr(bce, ps), getattr(bce, method))
    c.KeyGen(42)
    nmax = max(batches)
    c.pool_reserve(3 * nmax)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    out = (C.c_ulonglong * 256)()
    for nb in batches:
        descs = bce.make_descs([(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)])
        c.EvalGates(descs)
        c.synchronize()
        getter(out, 1)
        c.timing_reset()
        c.EvalGates(descs)
        c.synchronize()
        getter(out, 1)
        t = c.timing()
        print("batch %d: blind_rotate %.2f ms; workgroup 0, cycles per wave (rows) and phase (columns), wave 0 total %.0f" % (
            nb, t["blind_rotate_ms"], float(sum(out[:16]))))
        cols = [8, 9, 10, 11, 0, 1, 2, 3, 4, 5] if ps == "STD128_OPT" else [0, 1, 2, 3, 4, 5]
        names = {8: "head", 9: "inv p1", 10: "inv p2", 11: "p3+bar", 0: "phase1" if ps != "STD128_OPT" else "p4+digits", 1: "wait b1", 2: "forward", 3: "wait b2", 4: "MAC", 5: "wait b3"}
        print("   wave " + " ".join("%10s" % names[k] for k in cols) + "      total")
        for w in range(16):
            row = out[16 * w:16 * w + 16]
            if not any(row):
                continue
            print("   %4d " % w + " ".join("%10d" % row[k] for k in cols) + " %10d" % sum(row))



main()
