"""Dependency-driven evaluation (bce_dag_*, Circuit.setDataflow): one persistent launch per evaluation in which a
finished bootstrap releases its consumers on the device -- the reference's ready-gate rule
(src/circuit.cpp:575-683) per gate instead of per frontier (src/circuit.cpp:698-710).

Parity bar: every register holds the ciphertext the frontier-by-frontier path leaves there (which the rest of the GPU
suite pins against the CPU oracle stage by stage), and gates replayed on the oracle from their input registers agree
bit for bit.  All through the C ABI."""
import os
import random
import time

import numpy as np
import pytest

import kat
from kat import CIRCUITS
from test_random_circuits import random_netlist

pytestmark = pytest.mark.gpu
SEED = 0x0FE5EED


@pytest.fixture(scope="module")
def std(bce, orc):
    o = orc.Oracle(orc.STD128_OPT, orc.GINX)
    o.keygen(SEED)
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(SEED)
    assert c.dag_supported()
    yield o, c
    o.close()
    c.close()


def _random_ssa_dag(bce, rng, n_inputs, n_tasks, window=40):
    tasks = []
    for i in range(n_tasks):
        hi = n_inputs + i
        a, b = rng.integers(max(0, hi - window), hi, 2)
        op = int(rng.choice([bce.AND, bce.OR, bce.NAND, bce.NOR]))
        tasks.append((op, int(a), int(b), hi, int(rng.integers(0, 2)), int(rng.integers(0, 2))))
    return tasks


def _levels(tasks):
    lvl, out = {}, []
    for t in tasks:
        l = 1 + max(lvl.get(t[1], 0), lvl.get(t[2], 0))
        lvl[t[3]] = l
        while len(out) < l:
            out.append([])
        out[l - 1].append(t)
    return out


def test_random_dag_every_register_equals_frontier_path_and_oracle(bce, orc, std):
    """600 dependent gates x 2 instances: both workgroup residencies, placement on and off, random priority classes, two
    runs of the same bce_dag (the second re-arms the queues over the first one's leftovers)"""
    o, cc = std
    rng = np.random.default_rng(11)
    n_in, n_tasks, K = 16, 600, 2
    stride = n_in + n_tasks
    tasks = _random_ssa_dag(bce, rng, n_in, n_tasks)
    cc.pool_reserve(2 * K * stride)
    bits = rng.integers(0, 2, K * n_in).astype(np.uint8)
    slots = np.array([k * stride + i for k in range(K) for i in range(n_in)], dtype=np.uint32)
    cc.set_encrypt_seed(SEED)
    base = K * stride
    cc.Encrypt(bits, slots, enc_index_base=500)
    cc.Encrypt(bits, slots + base, enc_index_base=500)
    cc.set_encrypt_seed(None)
    for level in _levels(tasks):
        cc.EvalGates(level, instances=K, slot_stride=stride)
    want = cc.lwe_read(np.arange(0, K * stride, dtype=np.uint32))
    assert np.array_equal(want[:n_in], cc.lwe_read(np.arange(base, base + n_in, dtype=np.uint32)))
    # the oracle agrees with the frontier path on a gate deep in the DAG (instance 1)
    op, a, b, out, n0, n1 = tasks[-1]
    ca, cb = want[stride + a], want[stride + b]
    ca = o.eval_not(ca) if n0 else ca
    cb = o.eval_not(cb) if n1 else cb
    assert np.array_equal(want[stride + out], o.eval_bingate({bce.AND: orc.AND, bce.OR: orc.OR, bce.NAND: orc.NAND, bce.NOR: orc.NOR}[op], ca, cb))
    zeros = np.zeros((n_tasks, want.shape[1]), dtype=np.uint64)
    for wg in (1, 2):
        for placement in (1, 0):
            cc.dag_set_limits(workgroups_per_cu=wg, placement=placement)
            dag = cc.dag_create(tasks, prio=[int(x) for x in rng.integers(0, 4, n_tasks)])
            for rep in range(2):
                for k in range(K):
                    cc.lwe_write(np.arange(base + k * stride + n_in, base + (k + 1) * stride, dtype=np.uint32), zeros)
                cc.dag_run(dag, K, stride, base)
                cc.synchronize()
                got = cc.lwe_read(np.arange(base, base + K * stride, dtype=np.uint32))
                assert np.array_equal(got, want), "dataflow registers differ (workgroups/CU %d, placement %d, run %d)" % (wg, placement, rep)
                last = cc.dag_last_run()
                assert last["done"] == K * n_tasks and last["abort"] == 0 and last["workgroups_per_cu"] == wg
            cc.dag_destroy(dag)
    cc.dag_set_limits()


def test_dag_create_rejects_what_is_not_a_dag_in_ssa_form(bce, std):
    _, cc = std
    cc.pool_reserve(64)
    with pytest.raises(bce.BceError):                      # slot 5 written twice
        cc.dag_create([(bce.AND, 0, 1, 5), (bce.OR, 2, 3, 5)])
    with pytest.raises(bce.BceError):                      # slot 4 read, then overwritten: order would matter
        cc.dag_create([(bce.AND, 4, 1, 5), (bce.OR, 2, 3, 4)])
    with pytest.raises(bce.BceError):                      # not a bootstrapped gate
        cc.dag_create([(bce.OP_NOT, 0, 0, 5)])
    with pytest.raises(bce.BceError):                      # priority class out of range
        cc.dag_create([(bce.AND, 0, 1, 5)], prio=[7])
    dag = cc.dag_create([(bce.AND, 0, 1, 5)])
    with pytest.raises(bce.BceError):                      # instance 1 would leave the pool
        cc.dag_run(dag, 2, 1 << 30, 0)
    cc.dag_destroy(dag)


def test_a_run_that_cannot_progress_ends_and_reports(bce, std):
    """bounded spins: a task whose producer count is one too high never becomes ready; every workgroup leaves after the
    stall limit, the next synchronising call fails, and the context keeps working"""
    _, cc = std
    rng = np.random.default_rng(5)
    n_in, n_tasks = 8, 120
    stride = n_in + n_tasks
    tasks = _random_ssa_dag(bce, rng, n_in, n_tasks, window=12)
    cc.pool_reserve(stride)
    cc.Encrypt(rng.integers(0, 2, n_in).astype(np.uint8), np.arange(n_in, dtype=np.uint32))
    cc.dag_set_limits(stall_ms=200)
    dag = cc.dag_create(tasks)
    cc.dag_debug_block_task(dag, 30)
    cc.dag_run(dag, 1, stride, 0)
    t0 = time.time()
    with pytest.raises(bce.BceError) as e:
        cc.synchronize()
    assert time.time() - t0 < 5.0
    assert "scheduler gave up" in str(e.value)
    last = cc.dag_last_run()
    assert last["abort"] != 0 and 30 <= last["done"] < n_tasks
    cc.dag_destroy(dag)
    cc.dag_set_limits()
    dag = cc.dag_create(tasks)                               # the same context, a healthy run
    cc.dag_run(dag, 1, stride, 0)
    cc.synchronize()
    assert cc.dag_last_run()["done"] == n_tasks
    cc.dag_destroy(dag)


def test_aes_expanded_dataflow_leaves_the_step_schedule_s_ciphertexts_in_all_25765_registers(bce, orc, std):
    o, cc = std
    path = os.path.join(CIRCUITS, "AES-expanded.txt")
    v = [x for x in kat.AES_VECTORS if x["circuit"] == "AES-expanded"][1]
    lines = [l.split() for l in open(path) if l.strip()]
    n_in = int(lines[1][0]) + int(lines[1][1])
    boot = np.array([n_in + gi for gi, t in enumerate(lines[2:]) if t[-1] in ("AND", "XOR")], dtype=np.uint32)
    assert boot.size == 20325 + 5440
    regs = {}
    cc.set_encrypt_seed(SEED)                                # both circuits draw the same input ciphertexts
    for mode in ("steps", "dataflow"):
        c = bce.Circuit(cc)
        c.ReadBristol(path)
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        if mode == "dataflow":
            c.setDataflow(True)
        c.SetInput(kat.aes_case(v)[0])
        assert c.dataflowActive() == (mode == "dataflow")
        assert c.Clock()[0] == kat.aes_case(v)[1]
        st = c.stats()
        assert st["bootstraps"] == 66415
        assert st["sublaunches"] == (416 if mode == "steps" else 1)
        regs[mode] = cc.lwe_read(boot)
        if mode == "dataflow":
            c.Rearm()                                        # a second evaluation re-arms the same DAG object
            assert c.Clock()[0] == kat.aes_case(v)[1]
            assert np.array_equal(cc.lwe_read(boot), regs[mode])
            # the last XOR of the netlist replayed on the oracle from its input registers (src/gate.cpp:198-202)
            gi = max(i for i, t in enumerate(lines[2:]) if t[-1] == "XOR")
            reg = {w: w for w in range(n_in)}
            reg.update({int(t[-2]): n_in + i for i, t in enumerate(lines[2:])})
            a, b = cc.lwe_read([reg[int(lines[2 + gi][2])], reg[int(lines[2 + gi][3])]])
            x = o.eval_bingate(orc.OR, o.eval_bingate(orc.AND, a, o.eval_not(b)), o.eval_bingate(orc.AND, o.eval_not(a), b))
            assert np.array_equal(cc.lwe_read([n_in + gi])[0], x), "dataflow XOR register differs from the oracle"
        c.close()
    cc.set_encrypt_seed(None)
    assert np.array_equal(regs["steps"], regs["dataflow"]), "dataflow AES registers differ from the bootstrap-depth schedule"


@pytest.mark.parametrize("seed", range(6))
def test_random_netlists_dataflow_equals_step_schedule_and_plaintext(bce, tmp_path, std, seed):
    """Bristol Fashion netlists with NOT chains, constants, MAND and wire copies; K = 3; outputs against the Python
    evaluator and every bootstrapped register against the step schedule"""
    _, cc = std
    rnd = random.Random(9100 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(20, 70))
    path = tmp_path / "rand.txt"
    path.write_text(text)
    K = 3
    ins = [[[rnd.randint(0, 1) for _ in range(w)] for w in in_w] for _ in range(K)]
    snap = {}
    cc.set_encrypt_seed(SEED + seed)
    for mode in ("steps", "dataflow"):
        c = bce.Circuit(cc)
        c.ReadBristol(str(path), new_flag=True)
        c.setInstances(K)
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        c.setDataflow(mode == "dataflow")
        info = c.info()
        W, stride = info["n_wires"], info["slot_stride"]
        # registers nobody writes (NOT / copy wires are folded into their consumers) stay as they are: start from zeros so
        # that the whole register file of the K instances can be compared
        cc.pool_reserve(K * stride)
        cc.lwe_write(np.arange(K * stride, dtype=np.uint32), np.zeros((K * stride, cc.n + 1), dtype=np.uint64))
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        c.Clock()
        for k in range(K):
            assert c.Outputs(k) == evaluate(ins[k]), "instance %d, %s" % (k, mode)
        snap[mode] = np.concatenate([cc.lwe_read(np.arange(k * stride, k * stride + W, dtype=np.uint32)) for k in range(K)])
        assert snap[mode].any(axis=1).sum() >= K * (sum(in_w) + 1)
        c.close()
    cc.set_encrypt_seed(None)
    assert np.array_equal(snap["steps"], snap["dataflow"])


def test_sha256_dataflow_four_reference_vectors(bce, std):
    """The largest DAG of the BASELINE configs through the persistent kernel: new-format sha256, 354,505 tasks x 4 instances
    (1.4 M bootstraps, 9,055 dependency levels) in ONE launch; the four vectors of sha-256-test.txt must come out."""
    _, cc = std
    c = bce.Circuit(cc)
    c.ReadBristol(os.path.join(CIRCUITS, "sha256_new.txt"), new_flag=True)
    vecs = kat.hash_vectors("sha-256-test.txt")
    c.setInstances(len(vecs))
    c.Reset(); c.setEncrypted(True); c.setRelevel(True); c.setDataflow(True)
    for k, (inhex, outhex) in enumerate(vecs):
        c.SetInput(kat.sha256_new_case(inhex, outhex)[0], instance=k)
    assert c.dataflowActive()
    t0 = cc.timing()
    c.Clock()
    t1 = cc.timing()
    for k, (inhex, outhex) in enumerate(vecs):
        assert c.Outputs(k)[0] == kat.sha256_new_case(inhex, outhex)[1], "sha-256 vector %d" % k
    assert t1["bootstraps"] - t0["bootstraps"] == 354505 * len(vecs)
    assert t1["blind_rotate_launches"] - t0["blind_rotate_launches"] == 1
    last = cc.dag_last_run()
    assert last["done"] == 354505 * len(vecs) and last["abort"] == 0
    c.close()


def test_a_long_urgent_chain_does_not_park_the_chip(bce, std):
    """claims were tickets (fetch-add) whenever a class showed ANY backlog, so at the start of a run
    every eager workgroup took a ticket of the most urgent class, and those beyond its first entries then waited for FUTURE
    entries of that class without looking at the others.  Two-class DAG that shows it: one chain of 100 dependent
    bootstraps in class 0 and 65,536 independent ones in class 3.  Under the old rule (BCE_DAG_TICKETS=1) 99 workgroups
    sit on tickets for chain links that arrive one bootstrap latency apart; with the hybrid claim (ticket only when the
    backlog covers every poller, compare-and-swap of the observed head otherwise) idle workgroups take the wide work.
    Checked on the scheduler's own counters: ticks spent looking for a bootstrap against ticks spent running one."""
    _, cc = std
    rng = np.random.default_rng(5)
    chain, wide = 100, 65536            # the wide work outlasts the chain: nobody idles for lack of work
    n_in = 2
    tasks, prio = [], []
    prev = 0
    for i in range(chain):
        tasks.append((bce.AND if i % 2 else bce.OR, prev, 1, n_in + i)); prio.append(0)
        prev = n_in + i
    for j in range(wide):
        tasks.append((int(rng.choice([bce.AND, bce.NAND, bce.OR])), 0, 1, n_in + chain + j, j & 1, 0)); prio.append(3)
    stride = n_in + len(tasks)
    cc.pool_reserve(2 * stride)
    cc.Encrypt(np.array([1, 1], dtype=np.uint8), np.array([0, 1], dtype=np.uint32))
    cc.Encrypt(np.array([1, 1], dtype=np.uint8), np.array([stride, stride + 1], dtype=np.uint32))
    cc.lwe_write([stride, stride + 1], cc.lwe_read([0, 1]))
    for level in _levels(tasks):
        cc.EvalGates(level)
    want = cc.lwe_read(np.arange(0, stride, dtype=np.uint32))
    ratios = {}
    for rule in ("hybrid", "tickets"):
        if rule == "tickets":
            os.environ["BCE_DAG_TICKETS"] = "1"
        try:
            cc.dag_set_limits(workgroups_per_cu=2)
            dag = cc.dag_create(tasks, prio=prio)
            t0 = time.time()
            cc.dag_run(dag, 1, stride, stride)
            cc.synchronize()
            dt = time.time() - t0
        finally:
            os.environ.pop("BCE_DAG_TICKETS", None)
        last = cc.dag_last_run()
        assert last["done"] == len(tasks) and last["abort"] == 0
        assert np.array_equal(cc.lwe_read(np.arange(stride, 2 * stride, dtype=np.uint32)), want), rule
        ratios[rule] = last["wait_ticks"] / max(1, last["busy_ticks"])
        print("claim rule %-8s: %.3f s, wait / busy ticks = %.4f" % (rule, dt, ratios[rule]))
        cc.dag_destroy(dag)
    cc.dag_set_limits()
    assert ratios["hybrid"] < 0.05, ratios
    assert ratios["tickets"] > 4 * ratios["hybrid"], ratios
