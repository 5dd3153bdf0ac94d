"""A whole step schedule resident on the device (bce_plan_*): `bce_plan_run_step` = one frontier call without the
descriptor upload, `bce_plan_run` = every step's launches replayed as ONE hipGraph (Circuit.setGraph).  The reference
finds the same ready gates in the same order every time a circuit is clocked (src/circuit.cpp:575-683); the plan keeps
that list on the device.

Parity bar: every register holds the ciphertext the frontier-by-frontier calls leave there (which the rest of the GPU
suite pins against the CPU oracle stage by stage), and a deep gate is replayed on the oracle.  All through the C ABI."""
import os
import random

import numpy as np
import pytest

import kat
from kat import CIRCUITS
from test_gpu_dataflow import _levels, _random_ssa_dag
from test_random_circuits import random_netlist

pytestmark = pytest.mark.gpu
SEED = 0x0FE5EED


@pytest.fixture(scope="module")
def std(bce, orc):
    o = orc.Oracle(orc.STD128_OPT, orc.GINX)
    o.keygen(SEED)
    c = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    c.KeyGen(SEED)
    yield o, c
    o.close()
    c.close()


def _graph_kind(t):
    k = [i for i, x in enumerate(t["by_kernel"]) if "hipGraph" in x["kernel"]]
    assert len(k) == 1
    return t["by_kernel"][k[0]]


def test_plan_steps_and_graph_leave_the_frontier_path_s_registers(bce, orc, std):
    """400 dependent gates x 3 instances on a shifted slot base: step-by-step from the resident descriptors, then the
    captured graph twice, then again after the pool was re-allocated under the graph (it must notice and re-capture)"""
    o, cc = std
    rng = np.random.default_rng(23)
    n_in, n_tasks, K = 16, 400, 3
    stride = n_in + n_tasks
    tasks = _random_ssa_dag(bce, rng, n_in, n_tasks)
    levels = _levels(tasks)
    cc.pool_reserve(2 * K * stride)
    bits = rng.integers(0, 2, K * n_in).astype(np.uint8)
    slots = np.array([k * stride + i for k in range(K) for i in range(n_in)], dtype=np.uint32)
    base = K * stride
    cc.set_encrypt_seed(SEED)
    cc.Encrypt(bits, slots, enc_index_base=900)
    cc.Encrypt(bits, slots + base, enc_index_base=900)
    cc.set_encrypt_seed(None)
    for level in levels:
        cc.EvalGates(level, instances=K, slot_stride=stride)
    want = cc.lwe_read(np.arange(0, K * stride, dtype=np.uint32))
    op, a, b, out, n0, n1 = tasks[-1]
    ca, cb = want[2 * stride + a], want[2 * stride + b]
    ca = o.eval_not(ca) if n0 else ca
    cb = o.eval_not(cb) if n1 else cb
    assert np.array_equal(want[2 * stride + out], o.eval_bingate({bce.AND: orc.AND, bce.OR: orc.OR, bce.NAND: orc.NAND, bce.NOR: orc.NOR}[op], ca, cb))

    zeros = np.zeros((n_tasks, want.shape[1]), dtype=np.uint64)

    def clear():
        for k in range(K):
            cc.lwe_write(np.arange(base + k * stride + n_in, base + (k + 1) * stride, dtype=np.uint32), zeros)

    def got():
        return cc.lwe_read(np.arange(base, base + K * stride, dtype=np.uint32))

    plan = cc.plan_create(levels, K, stride, base)
    clear()
    t0 = cc.timing()
    for s in range(len(levels)):
        cc.plan_run_step(plan, s)
    t1 = cc.timing()
    assert np.array_equal(got(), want), "plan steps differ from the frontier calls"
    assert t1["blind_rotate_launches"] - t0["blind_rotate_launches"] == len(levels)
    assert t1["bootstraps"] - t0["bootstraps"] == K * n_tasks
    for rep in range(2):
        clear()
        t0 = cc.timing()
        cc.plan_run(plan)
        t1 = cc.timing()
        assert np.array_equal(got(), want), "captured graph differs from the frontier calls (run %d)" % rep
        g0, g1 = _graph_kind(t0), _graph_kind(t1)
        assert g1["launches"] - g0["launches"] == len(levels) and g1["bootstraps"] - g0["bootstraps"] == K * n_tasks
        assert g1["ms"] > g0["ms"] and t1["bootstraps"] - t0["bootstraps"] == K * n_tasks
    # grow the pool: the device buffer moves, the captured kernel arguments are stale, the next run re-captures
    before = cc.lwe_read(np.arange(0, 2 * K * stride, dtype=np.uint32))
    cc.pool_reserve(64 * K * stride)
    assert np.array_equal(cc.lwe_read(np.arange(0, 2 * K * stride, dtype=np.uint32)), before)
    clear()
    cc.plan_run(plan)
    assert np.array_equal(got(), want), "graph after the pool moved"
    cc.plan_destroy(plan)


def test_plan_create_rejects_bad_schedules(bce, std):
    _, cc = std
    cc.pool_reserve(64)
    with pytest.raises(bce.BceError):
        cc.plan_create([[(bce.AND, 0, 1, 2)], []], 1, 0, 0)                      # empty step
    with pytest.raises(bce.BceError):
        cc.plan_create([[(bce.OP_NOT, 0, 0, 2)]], 1, 0, 0)                        # not a bootstrapped gate
    with pytest.raises(bce.BceError):
        cc.plan_create([[(bce.AND, 0, 1, 2)]], 2, 1 << 30, 0)                     # second instance outside the pool
    with pytest.raises(bce.BceError):
        cc.plan_create([], 1, 0, 0)
    p = cc.plan_create([[(bce.AND, 0, 1, 2)]], 1, 0, 0)
    with pytest.raises(bce.BceError):
        cc.plan_run_step(p, 1)
    cc.plan_destroy(p)


def test_aes_expanded_graph_schedule_leaves_the_step_schedule_s_ciphertexts(bce, std):
    """Circuit.setGraph on AES-expanded, K = 2: one graph launch per Clock(), all 25,765 bootstrapped registers of both
    instances identical to the step-by-step schedule, outputs = the reference's vectors (src/test_aes.cpp:186-228)"""
    _, cc = std
    path = os.path.join(CIRCUITS, "AES-expanded.txt")
    vecs = [v for v in kat.AES_VECTORS if v["circuit"] == "AES-expanded"]
    lines = [l.split() for l in open(path) if l.strip()]
    n_inw = int(lines[1][0]) + int(lines[1][1])
    boot = np.array([n_inw + gi for gi, t in enumerate(lines[2:]) if t[-1] in ("AND", "XOR")], dtype=np.uint32)
    regs = {}
    cc.set_encrypt_seed(SEED)
    for mode in ("steps", "graph"):
        c = bce.Circuit(cc)
        c.ReadBristol(path)
        c.setInstances(2)
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        c.setGraph(mode == "graph")
        assert c.graphActive() == (mode == "graph")
        for k, v in enumerate(vecs):
            c.SetInput(kat.aes_case(v)[0], instance=k)
        for rep in range(2):          # the second Clock() replays the instantiated graph
            if rep:
                c.Rearm()
            t0 = cc.timing()
            c.Clock()
            t1 = cc.timing()
            for k, v in enumerate(vecs):
                assert c.Outputs(k)[0] == kat.aes_case(v)[1], "AES vector %d, %s, run %d" % (k, mode, rep)
            assert t1["bootstraps"] - t0["bootstraps"] == 2 * 66415
            if mode == "graph":
                assert _graph_kind(t1)["launches"] - _graph_kind(t0)["launches"] == 416
        stride = c.info()["slot_stride"]
        regs[mode] = np.concatenate([cc.lwe_read(boot + k * stride) for k in range(2)])
        assert len(boot) == 25765
        c.close()
    cc.set_encrypt_seed(None)
    assert np.array_equal(regs["steps"], regs["graph"]), "graph schedule registers differ from the step schedule"


@pytest.mark.parametrize("seed", range(3))
def test_random_netlists_graph_equals_step_schedule_and_plaintext(bce, tmp_path, std, seed):
    _, cc = std
    rnd = random.Random(7700 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(20, 70))
    path = tmp_path / "rand.txt"
    path.write_text(text)
    K = 3
    ins = [[[rnd.randint(0, 1) for _ in range(w)] for w in in_w] for _ in range(K)]
    snap = {}
    cc.set_encrypt_seed(SEED + seed)
    for mode in ("steps", "graph"):
        c = bce.Circuit(cc)
        c.ReadBristol(str(path), new_flag=True)
        c.setInstances(K)
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        c.setGraph(mode == "graph")
        info = c.info()
        W, stride = info["n_wires"], info["slot_stride"]
        cc.pool_reserve(K * stride)
        cc.lwe_write(np.arange(K * stride, dtype=np.uint32), np.zeros((K * stride, cc.n + 1), dtype=np.uint64))
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        c.Clock()
        for k in range(K):
            assert c.Outputs(k) == evaluate(ins[k]), "instance %d, %s" % (k, mode)
        snap[mode] = np.concatenate([cc.lwe_read(np.arange(k * stride, k * stride + W, dtype=np.uint32)) for k in range(K)])
        c.close()
    cc.set_encrypt_seed(None)
    assert np.array_equal(snap["steps"], snap["graph"])


def test_events_off_counts_without_timing_and_leaves_the_same_ciphertexts(bce, std):
    """bce_timing_set_events(0): no HIP event between dependent kernels -- launches and bootstraps still counted, the
    millisecond fields stand still, results unchanged"""
    _, cc = std
    rng = np.random.default_rng(31)
    n_in, n_tasks = 8, 40
    tasks = _random_ssa_dag(bce, rng, n_in, n_tasks, window=10)
    levels = _levels(tasks)
    stride = n_in + n_tasks
    cc.pool_reserve(2 * stride)
    bits = rng.integers(0, 2, n_in).astype(np.uint8)
    cc.set_encrypt_seed(SEED)
    cc.Encrypt(bits, np.arange(n_in), enc_index_base=40)
    cc.Encrypt(bits, np.arange(n_in) + stride, enc_index_base=40)
    cc.set_encrypt_seed(None)
    for level in levels:
        cc.EvalGates(level)
    want = cc.lwe_read(np.arange(stride, dtype=np.uint32))
    cc.timing_set_events(False)
    try:
        t0 = cc.timing()
        for level in levels:
            cc.EvalGates([(op, a + stride, b + stride, o + stride, n0, n1) for (op, a, b, o, n0, n1) in level])
        t1 = cc.timing()
    finally:
        cc.timing_set_events(True)
    assert np.array_equal(cc.lwe_read(np.arange(stride, 2 * stride, dtype=np.uint32)), want)
    assert t1["bootstraps"] - t0["bootstraps"] == n_tasks and t1["blind_rotate_launches"] - t0["blind_rotate_launches"] == len(levels)
    assert t1["blind_rotate_ms"] == t0["blind_rotate_ms"] and t1["tail_ms"] == t0["tail_ms"]


def test_an_engine_may_be_destroyed_before_its_circuits(bce):
    """Every encrypted Clock() leaves a device-resident schedule (bce_plan; bce_dag under setDataflow) behind that the
    circuit destroys with itself.  A host that tears down in the other order -- cc.close() before circ.close() -- must
    not read freed memory: the context releases the schedules it still owns, the circuit's late destroy is a no-op."""
    path = os.path.join(CIRCUITS, "adder_2bit.out")
    for dataflow in (False, True):
        cc = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
        cc.KeyGen(3)
        circs = []
        for _ in range(2):
            c = bce.Circuit(cc)
            c.ReadFile(path)
            if dataflow:
                c.setDataflow(True)
            c.Reset(); c.setEncrypted(True)
            c.SetInput([[1, 1], [1, 0]])
            out = c.Clock()[0]
            assert out[0] + 2 * out[1] + 4 * out[2] == 4
            circs.append(c)
        circs[0].close()              # the usual order for one of them
        cc.close()
        circs[1].close()              # ... and the engine first for the other
    # the engine-level handles behave the same way
    cc = bce.BinFHEContext(bce.STD128_OPT, bce.GINX)
    cc.KeyGen(4)
    cc.pool_reserve(8)
    plan = cc.plan_create([[(bce.AND, 0, 1, 2)]])
    dag = cc.dag_create([(bce.AND, 0, 1, 2)])
    h = cc.h
    cc.close()
    cc.h = h                          # a stale handle, as a C caller would hold it
    cc.plan_destroy(plan)
    cc.dag_destroy(dag)
    cc.h = None
