"""N>1 path on CPU: two ranks over gloo evaluate one circuit with both sharding modes of the
host runtime (include/bce_circuit.h: set_exchange).  Plaintext mode exercises exactly the
partition / publish / gather logic the encrypted multi-GPU run uses (the payload is a bit
instead of an LWE ciphertext, the allgather is gloo instead of RCCL)."""
import importlib
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, shard_mode, circuit, K, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
        xmod = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.dist")
        import kat
        c = bce.Circuit()
        rand_eval = None
        if circuit.startswith("random:"):
            # a randomised Bristol Fashion netlist (constants, MAND, wire copies, several buses): same seed on every rank
            import random
            import tempfile
            from test_random_circuits import random_netlist
            rnd = random.Random(int(circuit.split(":")[1]))
            text, in_w, out_w, rand_eval = random_netlist(rnd, 150)
            with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
                f.write(text)
            c.ReadBristol(f.name, new_flag=True)
            os.unlink(f.name)
        else:
            c.ReadBristol(os.path.join(kat.CIRCUITS, circuit))
        c.setInstances(K)
        if os.environ.get("BCE_TEST_PLAN_MISMATCH") == "1":
            # ranks that disagree on the launch capacities (different devices / environment knobs) build different plans
            c.setRelevel(True)
            c.setBalance(True, 8 + 8 * rank, 16 + 16 * rank)
            try:
                xmod.Exchange(c, shard_mode, encrypted=False, device=None)
                q.put((rank, False, "plan mismatch went unnoticed", 0))
            except RuntimeError as e:
                q.put((rank, "different sharding plans" in str(e), 0, 0))
            dist.barrier()
            dist.destroy_process_group()
            return
        x = xmod.Exchange(c, shard_mode, encrypted=False, device=None)
        nbits = c.info()["n_input_bits"][0]
        if rand_eval is not None:
            rnd2 = random.Random(99)
            cases = []
            for _ in range(K):
                ins = [[rnd2.randint(0, 1) for _ in range(w)] for w in in_w]
                cases.append((ins, rand_eval(ins)))
        elif circuit.startswith("mult"):
            cases = [kat.multiplier_case(t % 10) for t in range(K)]
        else:
            cases = [kat.adder_case(t % 10, nbits) for t in range(K)]
        c.Reset()
        c.setPlaintext(True)
        for k, (ins, _) in enumerate(cases):
            c.SetInput(ins, instance=k)
        c.Clock()
        if rand_eval is not None:
            ok = all(c.Outputs(k) == want for k, (_, want) in enumerate(cases))
        else:
            ok = all(c.Outputs(k)[0] == want for k, (_, want) in enumerate(cases))
        st = c.stats()
        q.put((rank, ok, x.calls, st["exchanges"]))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface worker failures in the parent
        q.put((rank, False, repr(e), 0))


def _run(shard_mode, circuit, K, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + shard_mode * 7 + world * 13 + K
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard_mode, circuit, K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


def test_instance_sharding_two_ranks():
    """shard_mode 0: instances split over ranks, one allgather of the outputs at the end"""
    res = _run(0, "adder_32bit.txt", K=4)
    for rank, ok, calls, exchanges in res:
        assert ok is True, res
        assert calls == 1 and exchanges == 1


def test_gate_sharding_two_ranks():
    """shard_mode 1: every level's gates split over ranks, only boundary wires are published"""
    res = _run(1, "adder_32bit.txt", K=2)
    for rank, ok, calls, exchanges in res:
        assert ok is True, res
        assert 0 < exchanges == calls <= 128   # at most one allgather per level (127 levels + outputs)


def test_gate_sharding_single_instance_wide_circuit():
    res = _run(1, "mult_32x32.txt", K=1)
    for rank, ok, calls, exchanges in res:
        assert ok is True, res


@pytest.mark.parametrize("seed", [1, 2])
def test_gate_sharding_random_netlist(seed):
    """constants, MAND, wire copies and several input / output values under gate sharding"""
    res = _run(1, "random:%d" % seed, K=2)
    for rank, ok, calls, exchanges in res:
        assert ok is True, res


@pytest.mark.parametrize("world,circuit,K", [(4, "adder_32bit.txt", 2), (8, "adder_32bit.txt", 1), (4, "random:3", 2), (8, "random:4", 3)])
def test_gate_sharding_four_and_eight_ranks(world, circuit, K):
    """the same partition / publish / gather code with more ranks than two (the driver's node has eight GPUs): every
    rank's plan digest agrees (dist.Exchange checks it), every rank ends with every output"""
    res = _run(1, circuit, K=K, world=world)
    assert len(res) == world
    for rank, ok, calls, exchanges in res:
        assert ok is True, res
        assert exchanges == calls > 0


def test_instance_sharding_eight_ranks():
    res = _run(0, "adder_32bit.txt", K=8, world=8)
    for rank, ok, calls, exchanges in res:
        assert ok is True, res
        assert calls == 1 and exchanges == 1


def test_ranks_with_different_plans_are_told_before_the_first_exchange(monkeypatch):
    """every rank builds the plan itself from its device's launch capacity; a disagreement used to surface as
    all-gathers of different sizes.  dist.Exchange compares the plan digests (one MIN / MAX all-reduce) and raises."""
    monkeypatch.setenv("BCE_TEST_PLAN_MISMATCH", "1")
    res = _run(1, "adder_32bit.txt", K=5)
    for rank, ok, _, _ in res:
        assert ok is True, res
