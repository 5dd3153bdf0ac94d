"""Wall time of bce_keygen (secret keys on the host, all key rows sampled / transformed on the device) per parameter set.
usage: keygen_time.py   (GPU box)"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")

for ps, method in (("TOY", "GINX"), ("STD128_OPT", "GINX"), ("STD128_OPT", "AP"), ("STD192", "GINX"), ("STD192", "AP")):
    c = bce.BinFHEContext(getattr(bce, ps), getattr(bce, method))
    c.KeyGen(1)          # first call pays one-off costs (module load, allocations)
    t0 = time.time()
    c.KeyGen(2)
    dt = time.time() - t0
    words = c._L.bce_bsk_words(c.h)
    print("%-11s %-4s keygen %8.3f s   bootstrapping key %7.2f GB, key-switching key %6.1f MB" % (
        ps, method, dt, words * (8 if c.params["Q"] >= (1 << 28) else 4) / 1e9, c._L.bce_ksk_words(c.h) * 2 / 1e6), flush=True)
    c.close()
