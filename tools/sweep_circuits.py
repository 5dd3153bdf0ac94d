"""Every circuit under tests/golden/circuits, encrypted on the GPU with the slack-filled bootstrap-depth schedule, K = 3
random input sets each, against the plaintext evaluation of the same runtime -- step by step, then once more replayed as
one hipGraph (development aid / robustness sweep)."""
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

NEW = {"sha256_new.txt", "aes_128_new.txt", "FP-add.txt", "FP-eq.txt", "FP-f2i.txt", "FP-mul.txt", "adder64.txt", "mult64.txt",
       "mult2_64.txt", "neg64.txt", "sub64.txt", "zero_equal.txt"}


def main():
    K = 3
    cc = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    cc.KeyGen(None)
    bad = 0
    for fname in sorted(os.listdir(kat.CIRCUITS)):
        if not fname.endswith(".txt") or fname.endswith("-test.txt"):
            continue
        c = bce.Circuit(cc)
        try:
            c.ReadBristol(os.path.join(kat.CIRCUITS, fname), new_flag=fname in NEW)
        except bce.BceError as e:
            print(json.dumps({"circuit": fname, "skipped": str(e)[:80]}), flush=True)
            continue
        info = c.info()
        in_w, out_w = c.buses()
        c.setInstances(K)
        rng = np.random.default_rng(5)
        ins = [[rng.integers(0, 2, w).tolist() for w in in_w] for _ in range(K)]
        c.Reset(); c.setPlaintext(True)
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        c.Clock()
        want = [c.Outputs(k) for k in range(K)]
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        c.check_relevel()
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        t0 = time.time()
        c.Clock()
        dt = time.time() - t0
        ok = all(c.Outputs(k) == want[k] for k in range(K))
        # the same evaluation replayed as one hipGraph (Circuit.setGraph): outputs again
        c.Rearm(); c.setGraph(True)
        t0 = time.time()
        c.Clock()
        dtg = time.time() - t0
        okg = c.graphActive() and all(c.Outputs(k) == want[k] for k in range(K))
        bad += 0 if (ok and okg) else 1
        print(json.dumps({"circuit": fname, "bootstraps": info["n_bootstraps"], "steps": len(c.relevel_steps()), "K": K,
                          "seconds": round(dt, 3), "correct": ok, "graph_seconds": round(dtg, 3), "graph_correct": okg}), flush=True)
        c.close()
    print("SWEEP %s" % ("ok" if bad == 0 else "%d circuits WRONG" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
