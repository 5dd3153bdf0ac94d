"""Wall time of one frontier call (blind rotation + tail, automatic kernel choice) over launch sizes: the curve a
host-side schedule is priced with (development aid, GPU only).  usage: launch_curve.py [sizes ...]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")

SIZES = [1, 16, 64, 128, 192, 256, 257, 288, 320, 384, 448, 512, 513, 544, 576, 640, 704, 768, 896, 1024, 1025, 1152,
         1280, 1536, 1792, 2048, 2560, 3072, 4096, 5120, 6144, 8192]


def main():
    sizes = [int(x) for x in sys.argv[1:]] or SIZES
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(42)
    nmax = max(sizes)
    c.pool_reserve(3 * nmax)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    out = {}
    for nb in sizes:
        descs = bce.make_descs([(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)])
        c.EvalGates(descs)
        c.synchronize()
        reps = 5 if nb <= 1024 else 3
        t0 = time.time()
        for _ in range(reps):
            c.EvalGates(descs)
        c.synchronize()
        out[nb] = (time.time() - t0) / reps * 1e3
        print("%6d bootstraps: %8.3f ms  -> %8.0f /s" % (nb, out[nb], nb / out[nb] * 1e3), flush=True)
    print("CURVE " + json.dumps(out))


if __name__ == "__main__":
    main()
