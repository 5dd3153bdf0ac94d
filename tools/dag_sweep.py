"""Kernel time of the persistent dependency-driven kernel over launch sizes of INDEPENDENT bootstraps (development aid).
Usage: python tools/dag_sweep.py wg [sizes...]   (environment knobs BCE_DAG_PLACE, BCE_DAG_DEBUG apply)"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")

def main():
    wg = int(sys.argv[1])
    sizes = [int(x) for x in sys.argv[2:]] or [128, 256, 384, 512, 768, 1024, 2048]
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(42)
    nmax = max(sizes)
    c.pool_reserve(3 * nmax + 64)
    bits = np.random.default_rng(0).integers(0, 2, size=2 * nmax).astype(np.uint8)
    c.Encrypt(bits, np.arange(2 * nmax), enc_index_base=0)
    c.dag_set_limits(workgroups_per_cu=wg)
    for nb in sizes:
        tasks = [(bce.NAND, 2 * i, 2 * i + 1, 2 * nmax + i) for i in range(nb)]
        dag = c.dag_create(tasks)
        c.dag_run(dag); c.synchronize()
        c.timing_reset()
        reps = 3
        t0 = time.time()
        for _ in range(reps):
            c.dag_run(dag)
        c.synchronize()
        wall = (time.time() - t0) / reps
        t = c.timing()
        k = [x for x in t["by_kernel"] if x["launches"]]
        print(json.dumps({"wg": wg, "bootstraps": nb, "kernel_ms": round(t["blind_rotate_ms"] / reps, 3), "wall_ms": round(wall * 1e3, 3), "last": c.dag_last_run(), "kernels": [x["kernel"][:20] for x in k]}), flush=True)
        c.dag_destroy(dag)

if __name__ == "__main__":
    main()
