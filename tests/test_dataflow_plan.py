"""Host side of the dataflow schedule without a GPU: the task list Circuit::setDataflow hands to bce_dag_create
(the device-side ready-gate rule of src/circuit.cpp:575-683 needs it in SSA form and topological order)
evaluated in plaintext in RANDOM dependency-respecting orders must give the circuit's outputs -- i.e. whatever order the
device's workgroups pick, the result is the same.  Also pins the reference-shaped defaults of the driver API."""
import os
import random

import pytest

from conftest import CIRCUITS
from test_random_circuits import random_netlist

AND, OR = 1, 0


def _check_ssa_and_run(tasks, prio, inputs_by_slot, consts, rnd):
    """validates the list, then executes it in a random topological order; returns slot -> bit"""
    assert len(tasks) == len(prio) and all(0 <= p < 4 for p in prio)
    writer = {}
    for i, (op, a, b, out, n0, n1) in enumerate(tasks):
        assert op in (AND, OR)
        assert out not in writer, "slot written twice"
        assert out not in inputs_by_slot and out not in consts
        for x in (a, b):
            assert x in writer or x in inputs_by_slot or x in consts, "task %d reads slot %d nobody defines" % (i, x)
        writer[out] = i
    val = dict(inputs_by_slot)
    val.update(consts)
    deps = [{writer[x] for x in (t[1], t[2]) if x in writer} for t in tasks]
    for i, d in enumerate(deps):
        assert all(j < i for j in d), "not in topological order"
    cons = [[] for _ in tasks]
    for i, d in enumerate(deps):
        for j in d:
            cons[j].append(i)
    left = [len(d) for d in deps]
    ready = [i for i, n in enumerate(left) if n == 0]
    done = 0
    while ready:
        i = ready.pop(rnd.randrange(len(ready)))          # any ready task, like any idle workgroup
        op, a, b, out, n0, n1 = tasks[i]
        x, y = val[a] ^ n0, val[b] ^ n1
        val[out] = (x & y) if op == AND else (x | y)
        done += 1
        for c in cons[i]:
            left[c] -= 1
            if left[c] == 0:
                ready.append(c)
    assert done == len(tasks), "dependency cycle or unreachable task"
    return val


@pytest.mark.parametrize("seed", range(25))
def test_dataflow_task_list_is_a_valid_ssa_dag_and_order_independent(bce, tmp_path, seed):
    rnd = random.Random(4200 + seed)
    text, in_w, out_w, evaluate = random_netlist(rnd, rnd.randint(10, 150))
    path = tmp_path / "rand.txt"
    path.write_text(text)
    c = bce.Circuit()
    c.ReadBristol(str(path), new_flag=True)
    c.setDataflow(True)
    assert not c.dataflowActive()                         # no engine: the plan exists, the GPU path cannot run
    tasks, prio = c.dataflow_plan()
    info = c.info()
    assert len(tasks) == info["n_bootstraps"]
    assert max([t[3] for t in tasks] + [0]) < info["slot_stride"]
    ins = [[rnd.randint(0, 1) for _ in range(w)] for w in in_w]
    # plaintext run gives every wire's value: registers 0.. = inputs in bus order (ReadBristol's numbering)
    c.Reset(); c.setPlaintext(True); c.SetInput(ins)
    assert c.Clock() == evaluate(ins)
    flat = [b for bus in ins for b in bus]
    inputs_by_slot = {i: b for i, b in enumerate(flat)}
    # constants (EQ lines) are registers that are live from the start: find them as slots tasks read but nobody writes
    written = {t[3] for t in tasks}
    consts = {}
    const_vals = [int(l.split()[2]) for l in text.splitlines() if l.strip().endswith(" EQ")]
    unknown = sorted({x for t in tasks for x in (t[1], t[2])} - written - set(inputs_by_slot))
    # registers are numbered in file order (one per EQ / gate): the k-th unknown register is the k-th EQ that is read
    eq_regs = []
    reg = len(flat)
    for l in text.splitlines()[4:]:
        tk = l.split()
        if not tk:
            continue
        if tk[-1] == "EQ":
            eq_regs.append((reg, int(tk[2]))); reg += 1
        elif tk[-1] == "MAND":
            reg += int(tk[1])
        elif tk[-1] in ("XOR", "AND", "INV"):
            reg += 1
    consts = {r: v for r, v in eq_regs}
    assert set(unknown) <= set(consts), (unknown, consts)
    # reference value of every REGISTER, from the netlist text: wire -> (register, value); EQW aliases its input's register
    wire = {i: (i, b) for i, b in enumerate(flat)}
    reg = len(flat)
    regval = dict(inputs_by_slot)
    for l in text.splitlines()[4:]:
        tk = l.split()
        if not tk:
            continue
        op = tk[-1]
        if op in ("XOR", "AND"):
            a, b, o = int(tk[2]), int(tk[3]), int(tk[4])
            v = wire[a][1] ^ wire[b][1] if op == "XOR" else wire[a][1] & wire[b][1]
            wire[o] = (reg, v); regval[reg] = v; reg += 1
        elif op == "INV":
            a, o = int(tk[2]), int(tk[3])
            wire[o] = (reg, 1 - wire[a][1]); regval[reg] = 1 - wire[a][1]; reg += 1
        elif op == "EQ":
            wire[int(tk[3])] = (reg, int(tk[2])); regval[reg] = int(tk[2]); reg += 1
        elif op == "EQW":
            wire[int(tk[3])] = wire[int(tk[2])]
        else:
            m = int(tk[1])
            a = [int(x) for x in tk[2:2 + 2 * m]]
            for k in range(m):
                v = wire[a[k]][1] & wire[a[m + k]][1]
                wire[int(tk[2 + 2 * m + k])] = (reg, v); regval[reg] = v; reg += 1
    assert reg == info["n_wires"]
    for trial in range(3):
        val = _check_ssa_and_run(tasks, prio, inputs_by_slot, consts, random.Random(seed * 10 + trial))
        gate_regs = [t[3] for t in tasks if t[3] < info["n_wires"]]
        assert gate_regs, "no gate register"
        for r in gate_regs:
            assert val[r] == regval[r], "register %d: dataflow order gives %d, netlist says %d" % (r, val[r], regval[r])


def test_dataflow_plan_of_aes_expanded(bce):
    c = bce.Circuit()
    c.ReadBristol(os.path.join(CIRCUITS, "AES-expanded.txt"))
    base = c.info()["slot_stride"]
    c.setDataflow(True)
    tasks, prio = c.dataflow_plan()
    info = c.info()
    assert len(tasks) == 66415 and info["n_bootstraps"] == 66415
    assert info["slot_stride"] == info["n_wires"] + 2 * 20325 and info["slot_stride"] > base    # every XOR owns its two temporaries
    assert len({t[3] for t in tasks}) == len(tasks)
    assert 0 in prio and max(prio) <= 3
    # critical path: the tasks of class 0 contain a chain as long as the bootstrap depth (416 steps)
    level = {}
    for op, a, b, out, n0, n1 in tasks:
        level[out] = 1 + max(level.get(a, 0), level.get(b, 0))
    assert max(level.values()) == 416
    deepest = max(level.values())
    assert any(p == 0 and level[t[3]] == deepest for t, p in zip(tasks, prio))
    c.setDataflow(False)
    assert c.info()["slot_stride"] == base


def test_reference_shaped_defaults(bce):
    """cc.Encrypt(sk, bit) of OpenFHE v1.0.x bootstraps fresh ciphertexts by default (src/circuit.cpp:506 and the verify
    repairs src/gate.cpp:118,139,143,158,179,211 use that default): so does the driver; FRESH is the opt-in"""
    c = bce.Circuit()
    assert c.getEncryptMode() == bce.BOOTSTRAPPED
    c.setEncryptMode(bce.FRESH)
    assert c.getEncryptMode() == bce.FRESH
    with pytest.raises(bce.BceError):
        c.setEncryptMode(7)
    # the reference's call sequence (Circuit; ReadFile; Reset; setEncrypted; SetInput; Clock, src/test_aes.cpp:338-343) runs
    # the register-identical bootstrap-depth schedule unless the caller asks for the gate-level rounds of
    # src/circuit.cpp:532-573; Reset() clears the three mode flags like the reference's, not the schedule
    assert c.getRelevel()
    c.ReadFile(os.path.join(CIRCUITS, "adder_2bit.out"))
    c.Reset()
    assert c.getRelevel()
    c.setRelevel(False)
    assert not c.getRelevel()
    c.Reset()
    assert not c.getRelevel()


def test_predicted_gate_sharding_curve_is_consistent(bce):
    """host-side model behind `shard_gates.predicted` of the bench line (no GPU): more ranks never make the modelled
    evaluation slower, the one-rank row is the plain step schedule, the crossing outputs match the plans"""
    import importlib
    pred = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.predict")
    r = pred.predict_gate_sharding(os.path.join(CIRCUITS, "adder_64bit.txt"), False, 64, worlds=(1, 2, 4))
    rows = r["rows"]
    assert [x["gpus"] for x in rows] == [1, 2, 4]
    assert rows[0]["exchanges"] == 0 and rows[0]["speedup_vs_1"] == 1.0
    assert rows[0]["ms_per_evaluation"] >= rows[1]["ms_per_evaluation"] >= rows[2]["ms_per_evaluation"]
    assert all(0 < x["efficiency"] <= 1.0 for x in rows)
    assert rows[1]["crossing_outputs_per_instance"] > 0 and rows[2]["crossing_outputs_per_instance"] >= rows[1]["crossing_outputs_per_instance"]
    assert pred.launch_ms(256) < pred.launch_ms(257) < pred.launch_ms(512) < pred.launch_ms(513)
    # weak form (`shard_gates_weak.predicted`): K blocks PER GPU; each world's row carries K x world instances, the
    # throughput ratio is what `speedup_vs_1` holds and it cannot exceed the number of GPUs
    w = pred.predict_gate_sharding(os.path.join(CIRCUITS, "adder_64bit.txt"), False, 64, worlds=(1, 2, 4), weak=True)
    assert w["scaling"] == "weak" and [x["instances"] for x in w["rows"]] == [64, 128, 256]
    assert w["rows"][0]["speedup_vs_1"] == 1.0
    for x in w["rows"]:
        assert 0 < x["efficiency"] <= 1.0 and x["speedup_vs_1"] <= x["gpus"] + 1e-9
    assert w["rows"][2]["gate_bootstraps_per_s"] > w["rows"][1]["gate_bootstraps_per_s"] > w["rows"][0]["gate_bootstraps_per_s"]


def test_bench_cpu_baseline_walks_the_reference_s_rounds(orc):
    """bench.py's cpu_baseline leg: the oracle evaluates the circuit's real ready-gate rounds, one
    OpenMP task per gate like src/circuit.cpp:698-710, and checks decrypted outputs on the way (TOY keys here: seconds)"""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n_in, n_wires, rounds = bench.bristol_frontiers(os.path.join(CIRCUITS, "adder_32bit.txt"))
    assert n_in == 64 and sum(len(r) for r in rounds) == 127 + 61 + 187      # AND + XOR + INV gates of the netlist
    assert all(all(g[0] in ("AND", "XOR", "INV") for g in r) for r in rounds)
    r = bench.cpu_baseline(os.path.join(CIRCUITS, "adder_32bit.txt"), "TOY", "GINX", seconds_budget=2.0)
    assert r["kind"] == "port" and r["value"] > 0 and r["cores"] >= 1 and r["cpu_model"]
    assert "ready-gate rounds" in r["sample"] and "not OpenFHE" in r["sample"]
