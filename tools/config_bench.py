"""Encrypted evaluation of the BASELINE.json parity configs on one GPU: per-evaluation latency and
gate-bootstraps/s for K lock-step instances (development / documentation aid; bench.py is the headline)."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
import kat  # noqa: E402

CONFIGS = [
    # name, file, reader, paramset, method, K list
    ("1 adder_2bit TOY GINX", "adder_2bit.out", "out", "TOY", "GINX", [1]),
    ("2 adder_64bit STD128_OPT GINX", "adder_64bit.txt", "old", "STD128_OPT", "GINX", [1, 64, 256]),
    ("3 AES-expanded STD128_OPT GINX", "AES-expanded.txt", "old", "STD128_OPT", "GINX", [1, 16]),
    ("2r adder_64bit STD128_OPT GINX bootstrap-depth schedule", "adder_64bit.txt", "old", "STD128_OPT", "GINX", [64, 256]),
    ("2u adder_64bit STD128_OPT GINX bootstrap-depth schedule, ASAP placement (balance off)", "adder_64bit.txt", "old", "STD128_OPT", "GINX", [64, 256]),
    ("3r AES-expanded STD128_OPT GINX bootstrap-depth schedule", "AES-expanded.txt", "old", "STD128_OPT", "GINX", [1, 2, 4, 8, 32]),
    ("3u AES-expanded STD128_OPT GINX bootstrap-depth schedule, ASAP placement (balance off)", "AES-expanded.txt", "old", "STD128_OPT", "GINX", [1, 2, 4, 8, 32]),
    ("3n AES-non-expanded (33,616 gates) STD128_OPT GINX bootstrap-depth schedule", "AES-non-expanded.txt", "old", "STD128_OPT", "GINX", [1, 16]),
    ("3m md5 STD128_OPT GINX bootstrap-depth schedule", "md5.txt", "old", "STD128_OPT", "GINX", [16]),
    ("4 sha256 (new format) STD128_OPT GINX", "sha256_new.txt", "new", "STD128_OPT", "GINX", [16]),
    ("4r sha256 (new format) STD128_OPT GINX bootstrap-depth schedule", "sha256_new.txt", "new", "STD128_OPT", "GINX", [1, 4, 8, 16, 32]),
    ("4u sha256 (new format) STD128_OPT GINX bootstrap-depth schedule, ASAP placement (balance off)", "sha256_new.txt", "new", "STD128_OPT", "GINX", [16]),
    ("5 adder_64bit STD192 AP", "adder_64bit.txt", "old", "STD192", "AP", [64]),
    ("5b AES-expanded STD192 AP", "AES-expanded.txt", "old", "STD192", "AP", [2, 8]),
    ("5r AES-expanded STD192 AP bootstrap-depth schedule", "AES-expanded.txt", "old", "STD192", "AP", [2, 8]),
]


def main():
    only = sys.argv[1:]
    ctxs = {}
    rows = []
    for name, fname, kind, ps, method, Ks in CONFIGS:
        if only and name.split()[0] not in only:
            continue
        key = (ps, method)
        if key not in ctxs:
            cc = bce.BinFHEContext(getattr(bce, ps), getattr(bce, method))
            cc.KeyGen(0x0FE5EED)
            ctxs[key] = cc
        cc = ctxs[key]
        for K in Ks:
            c = bce.Circuit(cc)
            path = os.path.join(kat.CIRCUITS, fname)
            if kind == "out":
                c.ReadFile(path)
            else:
                c.ReadBristol(path, new_flag=(kind == "new"))
            info = c.info()
            c.setInstances(K)
            if "balance off" in name:
                c.setBalance(False)
            rng = np.random.default_rng(1)
            ins = [[rng.integers(0, 2, w).tolist() for w in info["n_input_bits"] if w] for _ in range(K)]
            c.Reset(); c.setPlaintext(True)
            for k in range(K):
                c.SetInput(ins[k], instance=k)
            c.Clock()
            want = [c.Outputs(k)[0] for k in range(K)]
            c.Reset(); c.setEncrypted(True)
            c.setRelevel("bootstrap-depth" in name)
            sched_steps = None
            if "bootstrap-depth" in name:
                sched_steps = len(c.relevel_steps())
            for k in range(K):
                c.SetInput(ins[k], instance=k)
            c.Clock()                      # warm-up
            c.Rearm()
            t0 = time.time()
            c.Clock()
            dt = time.time() - t0
            ok = all(c.Outputs(k)[0] == want[k] for k in range(K))
            st = c.stats()
            row = {"config": name, "K": K, "bootstraps_per_eval": info["n_bootstraps"], "sublaunches": info["n_sublaunches"],
                   "seconds": round(dt, 3), "bootstraps_per_s": round(st["bootstraps"] / dt), "ms_per_sublaunch": round(dt / max(1, st["sublaunches"]) * 1e3, 3),
                   "correct": ok}
            if sched_steps is not None:
                row["sublaunches"] = st["sublaunches"]
            rows.append(row)
            print(json.dumps(row), flush=True)
            c.close()
    return rows


if __name__ == "__main__":
    main()
