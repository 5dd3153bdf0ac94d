"""Encrypted multi-rank evaluation with both sharding modes.  Two ranks share the one GPU of the
test box (gloo carries the allgather, staged through the host; on a multi-GPU node the same code
path uses RCCL all_gather_into_tensor on the registered device buffers)."""
import importlib
import os
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, shard_mode, K, q, relevel=False):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
        xmod = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.dist")
        import kat
        cc = bce.BinFHEContext(bce.TOY, bce.GINX, device=0)
        cc.KeyGen(0x0FE5EED)                       # same seed on every rank = replicated keys
        cc.set_encrypt_seed(0x0FE5EED)             # and identical input ciphertexts on every rank (gate sharding needs them)
        c = bce.Circuit(cc)
        c.ReadBristol(os.path.join(kat.CIRCUITS, "adder_32bit.txt"))
        c.setInstances(K)
        x = xmod.Exchange(c, shard_mode, encrypted=True, device=torch.device("cuda", 0))
        cases = [kat.adder_case(t, 32) for t in range(K)]
        c.Reset()
        c.setEncrypted(True)
        c.setRelevel(bool(relevel))
        if relevel:
            c.check_relevel()
        for k, (ins, _) in enumerate(cases):
            c.SetInput(ins, instance=k)
        c.Clock()
        ok = all(c.Outputs(k)[0] == want for k, (_, want) in enumerate(cases))
        st = c.stats()
        ident = None
        if shard_mode == 1:
            # ciphertext identity: every register this rank computed or received == the single-rank evaluation of the same
            # input ciphertexts (raises on the first difference)
            ident = xmod.check_against_single_rank(cc, c, os.path.join(kat.CIRCUITS, "adder_32bit.txt"), False, relevel, K)
        q.put((rank, ok, st["bootstraps"], st["exchanges"], st["exchanged_cts"], ident))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, False, repr(e), 0, 0, None))


def _run(shard_mode, K, world=2, relevel=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + shard_mode + (7 if relevel else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard_mode, K, q, relevel)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    return sorted(res)


def test_encrypted_instance_sharding():
    res = _run(0, K=4)
    total = 0
    for rank, ok, boots, exchanges, cts, _ in res:
        assert ok is True, res
        assert exchanges == 1 and cts == 0          # only the decrypted outputs are gathered
        total += boots
    assert total == 310 * 4                          # adder_32bit: 310 bootstraps per evaluation


def test_encrypted_gate_sharding_exchanges_boundary_ciphertexts():
    res = _run(1, K=1)
    total = 0
    for rank, ok, boots, exchanges, cts, ident in res:
        assert ok is True, res
        assert exchanges > 0 and cts > 0
        total += boots
        held, same, per_inst = ident
        assert held == same and per_inst == 127 + 61 and held >= per_inst // 2      # adder_32bit: 127 AND + 61 XOR registers
    assert total == 310                              # every bootstrap ran on exactly one rank
    assert sum(r[5][0] for r in res) >= 127 + 61     # together the ranks hold every bootstrapped register


def test_encrypted_gate_sharding_on_the_bootstrap_depth_schedule():
    """gate sharding on the slack-filled bootstrap-depth schedule: every step's units are split over the ranks, the
    outputs whose consumers sit on the other rank are exchanged after the step; same sums, every bootstrap on one rank"""
    res = _run(1, K=2, relevel=True)
    total = 0
    for rank, ok, boots, exchanges, cts, ident in res:
        assert ok is True, res
        assert exchanges > 0 and cts > 0
        total += boots
        held, same, per_inst = ident
        assert held == same and held >= per_inst      # K = 2: at least half of 2 x 188 registers on each rank
    assert total == 310 * 2
    assert sum(r[5][0] for r in res) >= 2 * (127 + 61)


def test_in_library_rccl_allgather_world_of_one():
    """The library's own RCCL all-gather (rccl_xchg.cpp: dlopen, ncclCommInitRank, ncclAllGather on the engine
    stream).  A one-GPU box can only form a world of one (RCCL refuses two ranks on one device), which still
    exercises the loader, the communicator, the stream plumbing and the ordering with the engine's kernels:
    pool rows packed by k_pool_pack -> all-gather -> read back, without any synchronisation in between."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
    cc = bce.BinFHEContext(bce.TOY, bce.GINX, device=0)
    cc.KeyGen(11)
    cc.pool_reserve(8)
    cc.Encrypt([1, 0, 1, 1], [0, 1, 2, 3])
    want = cc.lwe_read([0, 1, 2, 3]).astype(np.uint32)
    uid = cc.rccl_unique_id()
    assert len(uid) == 128
    cc.rccl_init(uid, 0, 1)
    assert cc.rccl_comm_info() == {"ranks": 1, "rank": 0, "device": 0}          # ncclCommCount / UserRank / CuDevice of that communicator
    W = cc.n + 1
    send = torch.zeros(4 * W, dtype=torch.int32, device="cuda:0")
    recv = torch.zeros(4 * W, dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    slots = np.arange(4, dtype=np.uint32)
    L = bce._bind_circuit()                                                           # sets the argtypes of bce_pool_gather
    assert L.bce_pool_gather(cc.h, slots.ctypes.data, 4, send.data_ptr()) == 0      # engine stream, asynchronous
    cc.rccl_allgather(send.data_ptr(), recv.data_ptr(), 4 * W * 4)                    # same stream, no sync before
    cc.synchronize()
    got = recv.cpu().numpy().view(np.uint32).reshape(4, W)
    assert np.array_equal(got, want)
    with pytest.raises(bce.BceError):
        bce.BinFHEContext(bce.TOY, bce.GINX, device=0).rccl_allgather(send.data_ptr(), recv.data_ptr(), 16)   # no communicator
