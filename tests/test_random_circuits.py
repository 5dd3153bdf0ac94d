"""Randomised netlists through the Bristol Fashion front end and the circuit runtime, against a 20-line Python
evaluator: the host side of the hot path (parser, levelisation, CSR fan-out, XOR expansion with its scratch slots, NOT
folding of the bootstrap-depth schedule, constants, wire copies, several input / output values) on DAG shapes none of
the reference's circuit files has.  CPU: plaintext mode.  GPU (TOY parameters): encrypted, both schedules, K = 3."""
import os
import random

import pytest


def random_netlist(rnd, n_gates):
    """returns (text, input widths, output widths, evaluator)"""
    in_w = [rnd.randint(1, 5) for _ in range(rnd.randint(1, 3))]
    n_in = sum(in_w)
    lines, defs = [], []          # defs: (op, ins, outs) in file order
    wires = list(range(n_in))     # wires that carry a value so far
    nxt = n_in
    for _ in range(n_gates):
        op = rnd.choices(["XOR", "AND", "INV", "EQ", "EQW", "MAND"], weights=[6, 6, 3, 1, 1, 1])[0]
        if op in ("XOR", "AND"):
            a, b = rnd.choice(wires), rnd.choice(wires)
            defs.append((op, [a, b], [nxt])); lines.append("2 1 %d %d %d %s" % (a, b, nxt, op)); wires.append(nxt); nxt += 1
        elif op in ("INV", "EQW"):
            a = rnd.choice(wires)
            defs.append((op, [a], [nxt])); lines.append("1 1 %d %d %s" % (a, nxt, op)); wires.append(nxt); nxt += 1
        elif op == "EQ":
            v = rnd.randint(0, 1)
            defs.append((op, [v], [nxt])); lines.append("1 1 %d %d EQ" % (v, nxt)); wires.append(nxt); nxt += 1
        else:
            m = rnd.randint(2, 3)
            a = [rnd.choice(wires) for _ in range(2 * m)]
            outs = list(range(nxt, nxt + m))
            defs.append((op, a, outs)); lines.append("%d %d %s %s MAND" % (2 * m, m, " ".join(map(str, a)), " ".join(map(str, outs))))
            wires.extend(outs); nxt += m
    # outputs are the LAST wires: close the netlist with wire copies of random earlier wires
    out_w = [rnd.randint(1, 4) for _ in range(rnd.randint(1, 2))]
    for _ in range(sum(out_w)):
        a = rnd.choice(wires)
        defs.append(("EQW", [a], [nxt])); lines.append("1 1 %d %d EQW" % (a, nxt)); nxt += 1
    header = "%d %d\n%d %s\n%d %s\n\n" % (len(lines), nxt, len(in_w), " ".join(map(str, in_w)), len(out_w), " ".join(map(str, out_w)))

    def evaluate(inputs):
        val = {}
        flat = [b for bus in inputs for b in bus]
        for i, b in enumerate(flat):
            val[i] = b
        for op, ins, outs in defs:
            if op == "XOR": val[outs[0]] = val[ins[0]] ^ val[ins[1]]
            elif op == "AND": val[outs[0]] = val[ins[0]] & val[ins[1]]
            elif op == "INV": val[outs[0]] = 1 - val[ins[0]]
            elif op == "EQW": val[outs[0]] = val[ins[0]]
            elif op == "EQ": val[outs[0]] = ins[0]
            else:
                m = len(outs)
                for k in range(m):
                    val[outs[k]] = val[ins[k]] & val[ins[m + k]]
        res, pos = [], nxt - sum(out_w)
        for w in out_w:
            res.append([val[pos + k] for k in range(w)]); pos += w
        return res

    return header + "\n".join(lines) + "\n", in_w, out_w, evaluate


@pytest.mark.parametrize("seed", range(40))
def test_random_netlists_plaintext(bce, tmp_path, seed):
    rnd = random.Random(1000 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(5, 120))
    path = tmp_path / "rand.txt"
    path.write_text(text)
    c = bce.Circuit()
    c.ReadBristol(str(path), new_flag=True)
    info = c.info()
    assert [w for w in info["n_input_bits"] if w] == in_w and info["output_buses"] == out_w
    K = 3
    c.setInstances(K)
    c.Reset()
    c.setPlaintext(True)
    ins = [[[rnd.randint(0, 1) for _ in range(w)] for w in in_w] for _ in range(K)]
    for k in range(K):
        c.SetInput(ins[k], instance=k)
    c.Clock()
    for k in range(K):
        assert c.Outputs(k) == evaluate(ins[k]), "instance %d" % k
    # the same netlist through the assembler text format
    out = str(tmp_path / "rand_FHE.out")
    bce.assemble_bristol(str(path), out, new_flag=True)
    d = bce.Circuit()
    d.ReadFile(out)
    d.Reset(); d.setPlaintext(True); d.SetInput(ins[0])
    assert d.Clock() == evaluate(ins[0])


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_random_netlists_encrypted_both_schedules(bce, tmp_path, seed):
    rnd = random.Random(7000 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(20, 90))
    path = tmp_path / "rand.txt"
    path.write_text(text)
    cc = bce.BinFHEContext(bce.TOY, bce.GINX)
    cc.KeyGen()
    K = 3
    ins = [[[rnd.randint(0, 1) for _ in range(w)] for w in in_w] for _ in range(K)]
    for relevel in (False, True):
        c = bce.Circuit(cc)
        c.ReadBristol(str(path), new_flag=True)
        c.setInstances(K)
        c.Reset()
        c.setEncrypted(True)
        c.setRelevel(relevel)
        for k in range(K):
            c.SetInput(ins[k], instance=k)
        c.Clock()
        for k in range(K):
            assert c.Outputs(k) == evaluate(ins[k]), "instance %d, relevel %s" % (k, relevel)
    cc.close()
