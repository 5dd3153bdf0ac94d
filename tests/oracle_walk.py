"""The reference's encrypted circuit evaluation restated on the CPU oracle ALONE (test infrastructure).

Nothing of the product is imported here: the netlists are read by the two small parsers below and every
encrypted operation is a call into oracle/ -- so a known-answer test that goes through this module pins the
ORACLE to the reference's own golden outputs, independently of the HIP path.

What is followed, line by line:
* `Circuit::SetInput` (src/circuit.cpp:455-530): one `cc.Encrypt(sk, bit)` per LOAD (:506) -- OpenFHE v1.0.x's
  default output mode BOOTSTRAPPED = fresh encryption + one refresh bootstrap;
* `Circuit::Clock` (src/circuit.cpp:532-573, manager :575-683, executor :685-817): a gate runs in the first round
  after its last input wire arrived, every ready gate of a round in one parallel region (:698-710);
* `Gate::Evaluate` (src/gate.cpp:105-203): NOT = `EvalNOT`, AND / OR = `EvalBinGate`, XOR = `EvalNOT`, `EvalNOT`,
  `EvalBinGate(AND, in0, !in1)`, `EvalBinGate(AND, !in0, in1)`, `EvalBinGate(OR, ., .)` (:198-202);
* OUTPUT gates decrypt (src/circuit.cpp:800).
"""
import re

import numpy as np


class Netlist:
    """gates: (op, dst, a, b) over register numbers, op in {"NOT", "AND", "OR", "XOR"}; SSA, topological order."""

    def __init__(self):
        self.loads = []      # (register, input value index 0-based, bit)
        self.gates = []
        self.stores = {}     # output bit -> register
        self.n_regs = 0


_LOAD = re.compile(r"R(\d+)\s*=\s*LOAD\(In(\d+),\s*(\d+)\)")
_GATE2 = re.compile(r"R(\d+)\s*=\s*(AND|OR|XOR)\(R(\d+),\s*R(\d+)\)")
_GATE1 = re.compile(r"R(\d+)\s*=\s*NOT\(R(\d+)\)")
_STORE = re.compile(r"Out(\d+)\s*=\s*STORE\(R(\d+)\)")


def read_assembler_text(path):
    """The assembler's `.out` text (reader rules of src/circuit.cpp:135-300)."""
    nl = Netlist()
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        m = _LOAD.match(line)
        if m:
            nl.loads.append((int(m.group(1)), int(m.group(2)) - 1, int(m.group(3))))
            continue
        m = _STORE.match(line)
        if m:
            nl.stores[int(m.group(1))] = int(m.group(2))
            continue
        m = _GATE1.match(line)
        if m:
            nl.gates.append(("NOT", int(m.group(1)), int(m.group(2)), -1))
            continue
        m = _GATE2.match(line)
        if m:
            nl.gates.append((m.group(2), int(m.group(1)), int(m.group(3)), int(m.group(4))))
            continue
        raise ValueError("unreadable line: " + line)
    nl.n_regs = 1 + max([r for r, _, _ in nl.loads] + [g[1] for g in nl.gates])
    return nl


def read_bristol_old(path):
    """Old Bristol format (src/analyze.cpp:114-121,165-179,223-283, src/assemble.cpp:152-153,187-193):
    `<gates> <wires>` / `<n_in1> <n_in2> <n_out>` / blank / `<n_in> <n_out> <in wires> <out wire> <XOR|AND|INV>`;
    input wires first (value 1 then value 2), the outputs are the LAST n_out wires."""
    toks = [ln.split() for ln in open(path) if ln.strip()]
    n_gates, n_wires = int(toks[0][0]), int(toks[0][1])
    n1, n2, n_out = (int(v) for v in toks[1][:3])
    nl = Netlist()
    nl.loads = [(w, 0, w) for w in range(n1)] + [(n1 + w, 1, w) for w in range(n2)]
    for t in toks[2:2 + n_gates]:
        op = t[-1]
        if op == "INV":
            nl.gates.append(("NOT", int(t[3]), int(t[2]), -1))
        elif op in ("AND", "XOR"):
            nl.gates.append((op, int(t[4]), int(t[2]), int(t[3])))
        else:
            raise ValueError("unknown Bristol op " + op)
    nl.stores = {o: n_wires - n_out + o for o in range(n_out)}
    nl.n_regs = n_wires
    return nl


def rounds(nl):
    """Ready-gate rounds of the manager (src/circuit.cpp:575-683): round r holds the gates whose last input was
    produced in round r - 1 (inputs are there before round 0)."""
    level = {r: -1 for r, _, _ in nl.loads}
    out = []
    for g in nl.gates:
        op, dst, a, b = g
        lv = 1 + max(level[a], level[b] if b >= 0 else -1)
        level[dst] = lv
        while len(out) <= lv:
            out.append([])
        out[lv].append(g)
    return out


def evaluate(O, o, nl, inputs, enc_index=0, nthreads=0, fresh=False):
    """Encrypted evaluation on oracle context `o` (module `O` = oracle.oracle); returns (output bits, bootstraps)."""
    W = o.n + 1
    ready = rounds(nl)
    widest = max(sum(1 for g in r if g[0] == "XOR") for r in ready) if ready else 0
    T = nl.n_regs                                           # temporaries of the XORs of one round: 4 each
    pool = np.zeros((T + 4 * widest, W), dtype=np.uint64)
    boots = 0
    for reg, value, bit in nl.loads:                        # SetInput: cc.Encrypt(sk, bit), BOOTSTRAPPED by default
        pool[reg] = o.encrypt(inputs[value][bit], enc_index)
        enc_index += 1
    if not fresh:
        boots += o.eval_gates(pool, [(O.OP_REFRESH, r, r, r, 0, 0) for r, _, _ in nl.loads], nthreads)
    for gates in ready:
        s0, sA, sB = [], [], []                             # EvalNOTs; first EvalBinGate of every gate; the XORs' OR
        x = 0
        for op, dst, a, b in gates:
            if op == "NOT":
                s0.append((O.OP_NOT, a, a, dst, 0, 0))
            elif op == "AND":
                sA.append((O.AND, a, b, dst, 0, 0))
            elif op == "OR":
                sA.append((O.OR, a, b, dst, 0, 0))
            else:                                           # src/gate.cpp:198-202
                na, nb, t1, t2 = (T + 4 * x + k for k in range(4))
                x += 1
                s0 += [(O.OP_NOT, a, a, na, 0, 0), (O.OP_NOT, b, b, nb, 0, 0)]
                sA += [(O.AND, a, nb, t1, 0, 0), (O.AND, na, b, t2, 0, 0)]
                sB.append((O.OR, t1, t2, dst, 0, 0))
        for stage in (s0, sA, sB):
            if stage:
                boots += o.eval_gates(pool, stage, nthreads)
    n_out = 1 + max(nl.stores)
    return [o.decrypt(pool[nl.stores[k]]) for k in range(n_out)], boots
