"""Device timing of batched bootstraps for any parameter set / method (development aid, GPU only).
usage: quick_perf_cfg.py STD192 AP 64 256"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")


def main():
    ps, method = sys.argv[1], sys.argv[2]
    batches = [int(x) for x in sys.argv[3:]] or [1, 256]
    t0 = time.time()
    c = bce.BinFHEContext(getattr(bce, ps), getattr(bce, method))
    c.KeyGen(42)
    print("%s %s ctx+keygen %.2fs bytes/bootstrap %d" % (ps, method, time.time() - t0, c.bytes_per_bootstrap()), flush=True)
    nmax = max(batches)
    c.pool_reserve(3 * nmax)
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    for nb in batches:
        descs = bce.make_descs([(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)])
        c.EvalGates(descs)
        c.synchronize()
        c.timing_reset()
        t0 = time.time()
        c.EvalGates(descs)
        c.synchronize()
        wall = time.time() - t0
        t = c.timing()
        out = c.Decrypt(np.arange(2 * nmax, 2 * nmax + nb))
        exp = 1 - (bits[0:2 * nb:2] & bits[1:2 * nb:2])
        print("batch %5d: wall %9.2f ms  blind_rotate %9.2f ms  tail %6.2f ms -> %8.0f bootstraps/s  %.1f GB/s algorithmic  correct %d/%d"
              % (nb, wall * 1e3, t["blind_rotate_ms"], t["tail_ms"], nb / wall,
                 c.bytes_per_bootstrap() * nb / (t["blind_rotate_ms"] * 1e-3) / 1e9, int((out == exp).sum()), nb), flush=True)


if __name__ == "__main__":
    main()
