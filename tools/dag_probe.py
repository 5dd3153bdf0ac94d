"""Dependency-driven (dataflow) schedule against the step schedules on one GPU: ciphertext identity on every
bootstrapped register, then seconds / gate-bootstraps per second for the same circuit and K under each schedule
(development / documentation aid; bench.py is the headline).

    python tools/dag_probe.py parity            # random SSA DAG + adder_64 + AES registers, both workgroup counts
    python tools/dag_probe.py perf aes 1 2 4 8  # steps vs dataflow
"""
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

FILES = {"aes": ("AES-expanded.txt", False), "adder": ("adder_64bit.txt", False), "sha256": ("sha256_new.txt", True),
         "md5": ("md5.txt", False)}


def random_ssa_dag(rng, n_inputs, n_tasks):
    """tasks in topological order over slots [0, n_inputs + n_tasks): task i writes slot n_inputs + i"""
    tasks = []
    for i in range(n_tasks):
        hi = n_inputs + i
        lo = max(0, hi - 40)
        a, b = rng.integers(lo, hi, 2)
        op = int(rng.choice([bce.AND, bce.OR, bce.NAND, bce.NOR]))
        tasks.append((op, int(a), int(b), hi, int(rng.integers(0, 2)), int(rng.integers(0, 2))))
    return tasks


def levelise(tasks, n_inputs):
    lvl = {}
    out = []
    for t in tasks:
        l = 1 + max(lvl.get(t[1], 0), lvl.get(t[2], 0))
        lvl[t[3]] = l
        while len(out) < l:
            out.append([])
        out[l - 1].append(t)
    return out


def parity(cc):
    rng = np.random.default_rng(7)
    n_in, n_tasks, K = 24, 900, 3
    stride = n_in + n_tasks
    tasks = random_ssa_dag(rng, n_in, n_tasks)
    cc.pool_reserve(2 * K * stride)
    bits = rng.integers(0, 2, K * n_in).astype(np.uint8)
    slots_a = np.array([k * stride + i for k in range(K) for i in range(n_in)], dtype=np.uint32)
    cc.set_encrypt_seed(bytes(range(32)))
    cc.Encrypt(bits, slots_a, enc_index_base=100)
    base_b = K * stride
    cc.Encrypt(bits, slots_a + base_b, enc_index_base=100)       # identical ciphertexts for the second schedule
    for level in levelise(tasks, n_in):
        cc.EvalGates(level, instances=K, slot_stride=stride)
    want = cc.lwe_read(np.arange(0, K * stride, dtype=np.uint32))
    for wg in (1, 2):
        for placement in (1, 0):
            cc.dag_set_limits(workgroups_per_cu=wg, placement=placement)
            dag = cc.dag_create(tasks, prio=[int(x) for x in rng.integers(0, 4, n_tasks)])
            for rep in range(2):                                # the second run re-arms the same queues
                cc.lwe_write(np.arange(base_b + n_in, base_b + stride, dtype=np.uint32), np.zeros((n_tasks, want.shape[1]), dtype=np.uint64))
                cc.dag_run(dag, K, stride, base_b)
                cc.synchronize()
                got = cc.lwe_read(np.arange(base_b, base_b + K * stride, dtype=np.uint32))
                assert np.array_equal(got, want), "random DAG: dataflow registers differ (wg %d placement %d rep %d)" % (wg, placement, rep)
            print("random DAG ok", wg, placement, cc.dag_last_run(), flush=True)
            cc.dag_destroy(dag)
    cc.dag_set_limits()
    # the bounded spin: a task that never becomes ready
    cc.dag_set_limits(stall_ms=300)
    dag = cc.dag_create(tasks)
    cc.dag_debug_block_task(dag, n_tasks // 2)
    cc.dag_run(dag, 1, stride, base_b)
    t0 = time.time()
    try:
        cc.synchronize()
        raise SystemExit("blocked DAG did not report a stall")
    except bce.BceError as e:
        print("stall reported after %.2f s: %s" % (time.time() - t0, e), flush=True)
    cc.dag_destroy(dag)
    cc.dag_set_limits()
    # circuits: every bootstrapped register under the step schedule == dataflow
    for name, K in (("adder", 3), ("aes", 2)):
        fname, new = FILES[name]
        regs = {}
        for mode in ("steps", "dataflow"):
            c = bce.Circuit(cc)
            c.ReadBristol(os.path.join(kat.CIRCUITS, fname), new_flag=new)
            info = c.info()
            c.setInstances(K)
            c.Reset(); c.setEncrypted(True); c.setRelevel(True)
            if mode == "dataflow":
                c.setDataflow(True)
            r2 = np.random.default_rng(3)
            for k in range(K):
                c.SetInput([r2.integers(0, 2, w).tolist() for w in info["n_input_bits"] if w], instance=k)
            if mode == "dataflow":
                assert c.dataflowActive()
            out = c.Clock()
            st = c.stats()
            stride_c = c.info()["slot_stride"]
            W = info["n_wires"]
            lines = [l.split() for l in open(os.path.join(kat.CIRCUITS, fname)) if l.strip()]
            n_inw = int(lines[1][0]) + int(lines[1][1])
            boot = np.array([n_inw + gi for gi, t in enumerate(lines[2:]) if t[-1] in ("AND", "XOR")], dtype=np.uint32)
            regs[mode] = (np.concatenate([cc.lwe_read(boot + k * stride_c) for k in range(K)]), [c.Outputs(k)[0] for k in range(K)], st)
            c.close()
        assert regs["steps"][1] == regs["dataflow"][1]
        assert np.array_equal(regs["steps"][0], regs["dataflow"][0]), name + ": dataflow registers differ from the step schedule"
        print(name, "K", K, "registers identical:", regs["steps"][0].shape, regs["dataflow"][2], cc.dag_last_run(), flush=True)


def perf(cc, name, Ks, modes):
    fname, new = FILES[name]
    for K in Ks:
        for mode in modes:
            c = bce.Circuit(cc)
            c.ReadBristol(os.path.join(kat.CIRCUITS, fname), new_flag=new)
            info = c.info()
            c.setInstances(K)
            rng = np.random.default_rng(1)
            ins = [[rng.integers(0, 2, w).tolist() for w in info["n_input_bits"] if w] for _ in range(K)]
            c.Reset(); c.setPlaintext(True)
            for k in range(K):
                c.SetInput(ins[k], instance=k)
            c.Clock()
            want = [c.Outputs(k)[0] for k in range(K)]
            c.Reset(); c.setEncrypted(True); c.setRelevel(True)
            if mode.startswith("dataflow"):
                c.setDataflow(True)
            if mode.startswith("graph"):
                c.setGraph(True)
            cc.timing_set_events(not mode.endswith("noevents"))
            for k in range(K):
                c.SetInput(ins[k], instance=k)
            c.Clock()
            c.Rearm()
            t0 = time.time()
            c.Clock()
            dt = time.time() - t0
            ok = all(c.Outputs(k)[0] == want[k] for k in range(K))
            st = c.stats()
            row = {"circuit": name, "K": K, "schedule": mode, "seconds": round(dt, 4), "bootstraps_per_s": round(st["bootstraps"] / dt),
                   "launches": st["sublaunches"], "correct": ok}
            if mode.startswith("dataflow"):
                row["last_run"] = cc.dag_last_run()
            print(json.dumps(row), flush=True)
            c.close()


def main():
    # DAG_PROBE_PARAMS=STD192:AP -> BASELINE config 5 (k_bootstrap_dag64)
    ps, method = os.environ.get("DAG_PROBE_PARAMS", "STD128_OPT:GINX").split(":")
    cc = bce.BinFHEContext(getattr(bce, ps), getattr(bce, method))
    cc.KeyGen(0x0FE5EED)
    if sys.argv[1] == "parity":
        parity(cc)
    else:
        name = sys.argv[2]
        Ks = [int(x) for x in sys.argv[3:] if x.isdigit()] or [1]
        modes = [x for x in sys.argv[3:] if not x.isdigit()] or ["steps", "dataflow"]
        perf(cc, name, Ks, modes)


if __name__ == "__main__":
    main()
