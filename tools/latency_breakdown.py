"""Wall time vs device kernel time of ONE encrypted circuit evaluation (development aid, GPU only).
usage: latency_breakdown.py [circuit] [K]   (default AES-expanded.txt, K = 1, bootstrap-depth schedule)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
import kat  # noqa: E402


def main():
    fname = sys.argv[1] if len(sys.argv) > 1 else "AES-expanded.txt"
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cc = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    cc.KeyGen(0x0FE5EED)
    c = bce.Circuit(cc)
    c.ReadBristol(os.path.join(kat.CIRCUITS, fname), new_flag=False)
    info = c.info()
    c.setInstances(K)
    rng = np.random.default_rng(1)
    c.Reset(); c.setEncrypted(True); c.setRelevel(True)
    for k in range(K):
        c.SetInput([rng.integers(0, 2, w).tolist() for w in info["n_input_bits"] if w], instance=k)
    c.Clock()
    pred = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.predict")
    steps = c.relevel_steps()
    print("%d steps, staircase model %.1f ms; launch sizes: <=256: %d, 257..512: %d, >512: %d" % (
        len(steps), sum(pred.launch_ms(n * K) for n in steps), sum(1 for n in steps if n * K <= 256),
        sum(1 for n in steps if 256 < n * K <= 512), sum(1 for n in steps if n * K > 512)))
    for rep in range(2):
        c.Rearm()
        cc.synchronize(); cc.timing_reset()
        t0 = time.time()
        c.Clock()
        wall = time.time() - t0
        tm = cc.timing()
        dev = tm["blind_rotate_ms"] + tm["tail_ms"]
        print("%s K=%d: wall %.1f ms, blind rotation %.1f ms + tail %.1f ms = %.1f ms on the device (%.1f%%), %d launches -> %.3f ms gap per launch"
              % (fname, K, wall * 1e3, tm["blind_rotate_ms"], tm["tail_ms"], dev, 100 * dev / (wall * 1e3), tm["blind_rotate_launches"],
                 (wall * 1e3 - dev) / max(1, tm["blind_rotate_launches"])))
        print("   ", [(k["kernel"].split(" ")[0], k["launches"], round(k["ms"], 1)) for k in tm["by_kernel"] if k["launches"]], c.stats())


main()
