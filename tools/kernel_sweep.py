"""Launch-size sweep of the blind-rotation kernel variants (development aid, GPU only).

Runs batched STD128_OPT/GINX NAND bootstraps for each launch size with (a) the one-wave-per-inverse-transform
kernel (BCE_VARIANT=1), (b) the split-transform kernel at one workgroup per CU (2), (c) at two (3); prints blind-rotation milliseconds.
Each variant needs its own process (the knobs are read at context creation), so this script
re-invokes itself as a child per variant.
"""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [64, 128, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096, 5120, 6144]


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(42)
    nmax = max(SIZES)
    c.pool_reserve(3 * nmax)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    res = {}
    for nb in SIZES:
        descs = bce.make_descs([(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)])
        c.EvalGates(descs)
        c.synchronize()
        c.timing_reset()
        reps = 3
        for _ in range(reps):
            c.EvalGates(descs)
        c.synchronize()
        res[nb] = c.timing()["blind_rotate_ms"] / reps
    print("RESULT " + json.dumps(res))


def main():
    variants = [("1 wave/INTT", {"BCE_VARIANT": "1"}), ("split, x1/CU", {"BCE_VARIANT": "2"}), ("split, x2/CU", {"BCE_VARIANT": "3"})]
    table = {}
    for name, env in variants:
        e = dict(os.environ, **env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(out.stdout, out.stderr)
            raise SystemExit("variant %s failed" % name)
        table[name] = json.loads(line[0][7:])
    print("%8s %13s %13s %13s   (blind-rotation ms; best marked)" % ("boots", *[v[0] for v in variants]))
    for nb in SIZES:
        row = [table[v[0]][str(nb)] for v in variants]
        best = min(range(3), key=lambda i: row[i])
        print("%8d " % nb + " ".join(("%12.2f%s" % (x, "*" if i == best else " ")) for i, x in enumerate(row)))


if __name__ == "__main__":
    child() if "--child" in sys.argv else main()
