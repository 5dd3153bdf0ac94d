"""BASELINE config 5 on the GPU: AES-expanded with the AP method and the STD192 parameter set
(n = 1024, N = 2048, 37-bit Q, qKS = 2^19; 12.9 GB bootstrapping key generated on the device).

One module-scoped engine + oracle pair (same key seed on both sides; keygen parity itself is established in
test_gpu_engine.py) carries: stage-level parity of single gates on the N = 2048 kernel, and the circuit of
the config run as K = 2 lock-step instances on the reference's two AES vectors (src/test_aes.cpp:186-228),
functionally AND at ciphertext level for gates of the first two frontiers replayed on the oracle.
"""
import os

import numpy as np
import pytest

import kat
from kat import CIRCUITS

pytestmark = pytest.mark.gpu
SEED = 2718


@pytest.fixture(scope="module")
def std192_ap(bce, orc):
    o = orc.Oracle(orc.STD192, orc.AP)
    o.keygen(SEED)
    c = bce.BinFHEContext(bce.STD192, bce.AP)
    c.KeyGen(SEED)
    assert o.params == c.params and o.params["N"] == 2048 and o.params["method"] == 1
    yield o, c
    o.close()
    c.close()


def _truth(gate, a, b):
    return [a | b, a & b, 1 - (a | b), 1 - (a & b), a ^ b, 1 - (a ^ b)][gate]


def test_std192_ap_gates_bit_exact_every_stage(std192_ap, bce):
    """accumulator after the blind rotation, after extract + ModSwitch, after KeySwitch and the final
    ciphertext, for AND / OR / NAND / NOR on mixed inputs (4 bootstraps of the config-5 kernel)"""
    o, c = std192_ap
    cases = [(bce.AND, 1, 0), (bce.OR, 1, 0), (bce.NAND, 1, 1), (bce.NOR, 0, 0)]
    nb = len(cases)
    cts = [(o.encrypt(a, 2 * i), o.encrypt(b, 2 * i + 1)) for i, (_, a, b) in enumerate(cases)]
    c.pool_reserve(3 * nb)
    c.lwe_write(np.arange(2 * nb, dtype=np.uint32), np.concatenate([np.stack(p) for p in cts]))
    descs = [(g, 2 * i, 2 * i + 1, 2 * nb + i) for i, (g, _, _) in enumerate(cases)]
    acc, lweN, ks = c.debug_eval_stages(descs)
    out = c.lwe_read(np.arange(2 * nb, 3 * nb, dtype=np.uint32))
    for i, (g, a, b) in enumerate(cases):
        ca, cb = cts[i]
        r_acc = o.blind_rotate(g, o.gate_prep(g, ca, cb))
        assert np.array_equal(acc[i], r_acc), "STD192/AP accumulator differs, case %d" % i
        r_lweN = o.extract_modswitch(r_acc)
        assert np.array_equal(lweN[i], r_lweN)
        r_ks = o.keyswitch(r_lweN)
        assert np.array_equal(ks[i], r_ks)
        assert np.array_equal(out[i], o.modswitch_final(r_ks))
        assert np.array_equal(out[i], o.eval_bingate(g, ca, cb))
        assert o.decrypt(out[i]) == _truth(g, a, b)
    assert list(c.Decrypt(np.arange(2 * nb, 3 * nb, dtype=np.uint32))) == [_truth(g, a, b) for g, a, b in cases]


def _bristol_gates(path):
    """(op, in wires, out wire) of an old-format Bristol netlist, file order; header: gates wires / n1 n2 nout"""
    lines = [l.split() for l in open(path) if l.strip()]
    n_in = int(lines[1][0]) + int(lines[1][1])
    gates = []
    for t in lines[2:]:
        nin = int(t[0])
        gates.append((t[-1], [int(x) for x in t[2:2 + nin]], int(t[2 + nin])))
    return n_in, gates


def test_config5_aes_expanded_std192_ap_two_vectors(std192_ap, bce, orc):
    """BASELINE config 5 as named: AES-expanded, AP, STD192 -- both reference vectors in lock-step, verify off.
    Functional: decrypted outputs == the reference's golden ciphertext bits.  Ciphertext level: four gates from
    the first to the last frontier are replayed on the oracle from their input registers read back from the device
    pool; the output registers the circuit run produced must be identical."""
    o, cc = std192_ap
    path = os.path.join(CIRCUITS, "AES-expanded.txt")
    c = bce.Circuit(cc)
    c.ReadBristol(path)
    vecs = [v for v in kat.AES_VECTORS if v["circuit"] == "AES-expanded"]
    assert len(vecs) == 2
    c.setInstances(2)
    c.Reset()
    c.setEncrypted(True)
    for k, v in enumerate(vecs):
        c.SetInput(kat.aes_case(v)[0], instance=k)
    c.Clock()
    for k, v in enumerate(vecs):
        assert c.Outputs(k)[0] == kat.aes_case(v)[1], "AES vector %d" % k
    st = c.stats()
    assert st["bootstraps"] == 66415 * 2 and st["verify_fixes"] == 0
    print("config 5 (AES-expanded STD192/AP, K=2): %.1f s, %.0f gate-bootstraps/s" % (st["total_ms"] / 1e3, st["bootstraps"] / st["total_ms"] * 1e3))

    # ---- ciphertext-level replay of a few gates on the oracle (instance 1 = the all-ones vector) ----
    n_in, gates = _bristol_gates(path)
    # registers: inputs 0..n_in-1 in wire order, then one register per gate in file order (ReadBristol);
    # the pool slot of register r of instance k is k * slot_stride + r
    reg_of_wire = {w: w for w in range(n_in)}
    for gi, (op, ins, out) in enumerate(gates):
        reg_of_wire[out] = n_in + gi
    stride = c.info()["slot_stride"]

    def slot(k, wire):
        return k * stride + reg_of_wire[wire]

    def oracle_gate(op, a, b):
        if op == "AND":
            return o.eval_bingate(orc.AND, a, b)
        assert op == "XOR"   # (a AND !b) OR (!a AND b), src/gate.cpp:198-202
        return o.eval_bingate(orc.OR, o.eval_bingate(orc.AND, a, o.eval_not(b)), o.eval_bingate(orc.AND, o.eval_not(a), b))

    # registers are SSA and never overwritten, so any gate can be replayed from its input registers: the first XOR
    # (first frontier), the first AND (S-box, a few frontiers in), the last AND and the last XOR of the netlist
    ands = [g for g in gates if g[0] == "AND"]
    xors = [g for g in gates if g[0] == "XOR"]
    picked = [xors[0], ands[0], ands[-1]]
    second = xors[-1]
    k = 1
    # the default (bootstrap-depth) schedule folds EvalNOT into its consumers: the register of an INV gate holds no
    # ciphertext, its value is EvalNOT of the register it negates
    inv_src = {out: ins[0] for op, ins, out in gates if op == "INV"}

    def read_wire(w):
        return o.eval_not(read_wire(inv_src[w])) if w in inv_src else cc.lwe_read([slot(k, w)])[0]

    for op, ins, out in picked + [second]:
        a, b = read_wire(ins[0]), read_wire(ins[1])
        got = cc.lwe_read([slot(k, out)])[0]
        assert np.array_equal(got, oracle_gate(op, a, b)), "config-5 circuit register of gate %s %s differs from the oracle" % (op, ins)


@pytest.mark.parametrize("seed", range(2))
def test_config5_dataflow_kernel_leaves_the_step_schedule_s_registers(std192_ap, bce, tmp_path, seed):
    """The dependency-driven schedule on the config-5 kernel (k_bootstrap_dag64: persistent 1,024-thread workgroups, AP,
    folded key, fused tail): random Bristol Fashion netlists, K = 3, outputs against the Python evaluator, every register of
    the K instances against the bootstrap-depth step schedule (whose kernel the stage test above pins on the oracle), and one
    bootstrapped gate of the dataflow run replayed on the oracle from its input registers."""
    import random
    from test_random_circuits import random_netlist
    o, cc = std192_ap
    assert cc.dag_supported()
    rnd = random.Random(5200 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(30, 60))
    path = tmp_path / "rand5.txt"
    path.write_text(text)
    K = 3
    ins = [[[rnd.randint(0, 1) for _ in range(w)] for w in in_w] for _ in range(K)]
    snap = {}
    cc.set_encrypt_seed(SEED + 77 + seed)
    for mode in ("steps", "dataflow"):
        c = bce.Circuit(cc)
        c.ReadBristol(str(path), new_flag=True)
        c.setInstances(K)
        c.Reset(); c.setEncrypted(True); c.setRelevel(True)
        c.setDataflow(mode == "dataflow")
        info = c.info()
        W, stride = info["n_wires"], info["slot_stride"]
        cc.pool_reserve(K * stride)
        cc.lwe_write(np.arange(K * stride, dtype=np.uint32), np.zeros((K * stride, cc.n + 1), dtype=np.uint64))
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        t0 = cc.timing()
        c.Clock()
        t1 = cc.timing()
        assert c.dataflowActive() == (mode == "dataflow")
        if mode == "dataflow":
            dag = [i for i, k in enumerate(t1["by_kernel"]) if "dag" in k["kernel"]]
            assert len(dag) == 1 and t1["by_kernel"][dag[0]]["launches"] - t0["by_kernel"][dag[0]]["launches"] == 1
            assert t1["blind_rotate_launches"] - t0["blind_rotate_launches"] == 1, "one persistent launch per evaluation"
        for k in range(K):
            assert c.Outputs(k) == evaluate(ins[k]), "instance %d, %s" % (k, mode)
        snap[mode] = np.concatenate([cc.lwe_read(np.arange(k * stride, k * stride + W, dtype=np.uint32)) for k in range(K)])
        c.close()
    cc.set_encrypt_seed(None)
    assert np.array_equal(snap["steps"], snap["dataflow"]), "config-5 dataflow registers differ from the step schedule"
