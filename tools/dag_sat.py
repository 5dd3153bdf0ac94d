"""Saturated rate of the persistent dependency-driven kernel against the per-frontier launch on the SAME independent
bootstraps (no dependencies: only the kernels differ), plus a chain-structured DAG that keeps every workgroup busy while
letting them drift apart (development aid, GPU only).  Usage: python tools/dag_sat.py [n_boot ...]"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [512, 6144, 24576]
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(42)
    nmax = max(sizes)
    c.pool_reserve(3 * nmax + 64)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    for nb in sizes:
        tasks = [(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)]
        descs = bce.make_descs(tasks)
        c.EvalGates(descs); c.synchronize()
        t0 = time.time(); c.EvalGates(descs); c.synchronize(); t_launch = time.time() - t0
        row = {"bootstraps": nb, "per_frontier_launch_ms": round(t_launch * 1e3, 2), "per_frontier_k_per_s": round(nb / t_launch / 1e3, 1)}
        for wg in (2, 1):
            c.dag_set_limits(workgroups_per_cu=wg)
            dag = c.dag_create(tasks)
            c.dag_run(dag); c.synchronize()
            t0 = time.time(); c.dag_run(dag); c.synchronize(); dt = time.time() - t0
            row["dag_wg%d_ms" % wg] = round(dt * 1e3, 2)
            row["dag_wg%d_k_per_s" % wg] = round(nb / dt / 1e3, 1)
            c.dag_destroy(dag)
        # 512 independent chains of length nb / 512: every workgroup always has work, none waits for a frontier
        L = max(1, nb // 512)
        ch = []
        for j in range(512):
            prev = 2 * j
            for l in range(L):
                out = 2 * nmax + j * L + l
                ch.append((bce.NAND, prev, 2 * j + 1, out)); prev = out
        ch.sort(key=lambda t: (t[3] - 2 * nmax) % L)   # level-major order is still topological
        c.dag_set_limits(workgroups_per_cu=2)
        dag = c.dag_create(ch)
        c.dag_run(dag); c.synchronize()
        t0 = time.time(); c.dag_run(dag); c.synchronize(); dt = time.time() - t0
        row["dag_chains_ms"] = round(dt * 1e3, 2); row["dag_chains_k_per_s"] = round(len(ch) / dt / 1e3, 1)
        c.dag_destroy(dag)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
