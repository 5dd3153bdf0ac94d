"""torch.distributed plumbing for multi-rank circuit evaluation (one process per GPU).

The C++ circuit runtime decides WHAT crosses ranks (include/bce_circuit.h: set_exchange); this
module only provides the allgather it calls back into: RCCL (backend "nccl") on device buffers,
or gloo on host buffers for CPU-only tests.  Buffers are torch tensors whose data_ptr() is
registered with the runtime.
"""
import torch
import torch.distributed as dist


class Exchange:
    """in_library=True (and backend nccl): device payloads go through the library's own RCCL all-gather on the engine
    stream (bce_rccl_init / bce_circuit_enable_rccl) -- no host synchronisation and no Python in the per-level loop;
    this class then only carries host payloads (plaintext bits, final outputs) and the rendezvous of the unique id.
    `self.in_library` tells which path is active (False with a reason in `self.why` when RCCL could not be set up:
    every rank then uses the callback path, by agreement)."""

    def __init__(self, circuit, shard_mode, encrypted, device=None, group=None, in_library=False):
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.group = group
        self.device = device
        cap = circuit.exchange_capacity(self.world, shard_mode, encrypted)
        cap = (max(cap, 64) + 63) // 64 * 64
        self.cap = cap
        self.host_send = torch.zeros(cap, dtype=torch.uint8)
        self.host_recv = torch.zeros(cap * self.world, dtype=torch.uint8)
        self.dev_send = self.dev_recv = None
        if device is not None:
            self.dev_send = torch.zeros(cap, dtype=torch.uint8, device=device)
            self.dev_recv = torch.zeros(cap * self.world, dtype=torch.uint8, device=device)
        self.calls = 0
        self.in_library, self.why = False, "not requested"
        circuit.set_exchange(self.rank, self.world, shard_mode, self._allgather,
                             self.host_send.data_ptr(), self.host_recv.data_ptr(),
                             self.dev_send.data_ptr() if device is not None else None,
                             self.dev_recv.data_ptr() if device is not None else None, cap)
        # the plan for this world exists now: it may publish more per step than the estimate above (slack filling widens
        # steps up to world x the device's capacity) -- ask again and grow the buffers if needed
        need = (max(circuit.exchange_capacity(self.world, shard_mode, encrypted), 64) + 63) // 64 * 64
        if need > cap:
            self.cap = cap = need
            self.host_send = torch.zeros(cap, dtype=torch.uint8)
            self.host_recv = torch.zeros(cap * self.world, dtype=torch.uint8)
            if device is not None:
                self.dev_send = torch.zeros(cap, dtype=torch.uint8, device=device)
                self.dev_recv = torch.zeros(cap * self.world, dtype=torch.uint8, device=device)
            circuit.set_exchange(self.rank, self.world, shard_mode, self._allgather,
                                 self.host_send.data_ptr(), self.host_recv.data_ptr(),
                                 self.dev_send.data_ptr() if device is not None else None,
                                 self.dev_recv.data_ptr() if device is not None else None, cap)
        self._check_plans_agree(circuit)
        if in_library and device is not None and encrypted:
            self._setup_in_library(circuit)

    def _check_plans_agree(self, circuit):
        """every rank builds the sharding plan itself (from its device's launch capacity and environment): different plans
        would mean all-gathers of different sizes -- a hang or silently wrong registers.  One MIN / MAX all-reduce of the
        plan digest settles it before the first evaluation."""
        h = circuit.plan_hash()
        dev = self.device if dist.get_backend(self.group) == "nccl" else "cpu"
        parts = torch.tensor([h & 0x7FFFFFFF, (h >> 31) & 0x7FFFFFFF, h >> 62], dtype=torch.int64, device=dev)
        lo, hi = parts.clone(), parts.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise RuntimeError("the ranks built different sharding plans (different devices or BCE_* environment knobs?): "
                               "rank %d has digest %016x" % (self.rank, h))

    def _setup_in_library(self, circuit):
        """Every step that could leave the ranks disagreeing is agreed on first (all-reduce of a flag), so that either
        ALL ranks enter ncclCommInitRank or none does."""
        on_gpu = dist.get_backend(self.group) == "nccl"
        dev = self.device if on_gpu else "cpu"
        self.why = ""      # from here on `why` holds a REASON (it starts as the truthy "not requested")

        def all_ok(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
            return int(flag.item()) == 1

        if not all_ok(circuit.cc.rccl_available()):
            self.why = "RCCL library not loadable on every rank"
            return
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        ok = True
        if self.rank == 0:
            try:
                uid.copy_(torch.frombuffer(bytearray(circuit.cc.rccl_unique_id()), dtype=torch.uint8))
            except Exception as e:
                ok, self.why = False, repr(e)
        if not all_ok(ok):
            self.why = self.why or "rank 0 could not obtain an RCCL unique id"
            return
        dist.broadcast(uid, 0, group=self.group)
        try:
            circuit.cc.rccl_init(bytes(uid.cpu().numpy().tobytes()), self.rank, self.world)   # collective: all ranks are here
        except Exception as e:
            ok, self.why = False, repr(e)
        if all_ok(ok):
            circuit.enable_rccl(True)
            self.in_library, self.why = True, ""
        else:
            if ok:   # this rank's communicator exists but another rank's does not: give it back, all ranks use the callback
                try:
                    circuit.cc.rccl_shutdown()
                except Exception:
                    pass
            self.why = self.why or "another rank could not initialise RCCL"

    def _allgather(self, nbytes, on_device):
        try:
            self.calls += 1
            if on_device and dist.get_backend(self.group) != "nccl":
                # no RCCL (e.g. several ranks sharing one GPU under gloo): stage through the host
                self.host_send[:nbytes].copy_(self.dev_send[:nbytes])
                dist.all_gather_into_tensor(self.host_recv[: nbytes * self.world], self.host_send[:nbytes], group=self.group)
                self.dev_recv[: nbytes * self.world].copy_(self.host_recv[: nbytes * self.world])
                torch.cuda.synchronize(self.device)
            elif on_device:
                # engine stream was synchronized by the runtime before this call
                dist.all_gather_into_tensor(self.dev_recv[: nbytes * self.world], self.dev_send[:nbytes], group=self.group)
                torch.cuda.synchronize(self.device)
            elif self.device is not None and dist.get_backend(self.group) == "nccl":
                # host payload (plaintext bits / final outputs) staged through the GPU for RCCL
                self.dev_send[:nbytes].copy_(self.host_send[:nbytes])
                dist.all_gather_into_tensor(self.dev_recv[: nbytes * self.world], self.dev_send[:nbytes], group=self.group)
                self.host_recv[: nbytes * self.world].copy_(self.dev_recv[: nbytes * self.world])
                torch.cuda.synchronize(self.device)
            else:
                dist.all_gather_into_tensor(self.host_recv[: nbytes * self.world], self.host_send[:nbytes], group=self.group)
            return 0
        except Exception as e:  # never raise through the C callback
            print("exchange failed:", repr(e), flush=True)
            return 1


def gate_registers(circuit_path, new_flag=False):
    """registers (slot numbers of instance 0) that hold the output of a bootstrapped gate, and the circuit's input registers"""
    import importlib
    bce = importlib.import_module(__package__)
    c = bce.Circuit()
    c.ReadBristol(circuit_path, new_flag=new_flag)
    c.setDataflow(True)
    n_wires = c.info()["n_wires"]
    tasks, _ = c.dataflow_plan()
    n_in = c.info()["n_input_gates"]
    c.close()
    return sorted(t[3] for t in tasks if t[3] < n_wires), list(range(n_in))


def check_against_single_rank(cc, circ, circuit_path, new_flag, relevel, instances):
    """After a gate-sharded Clock() of `circ` (shard mode 1): every register of a bootstrapped gate this rank computed or
    received must hold, bit for bit, what an un-sharded evaluation of the same input ciphertexts leaves there (replicated
    keys, deterministic bootstraps).  Runs that reference evaluation on THIS rank for the first `instances` instances
    (their input ciphertexts are copied from the sharded run's pool).  Returns (registers held, registers compared equal,
    bootstrapped registers per instance); raises AssertionError on the first difference."""
    import importlib
    import numpy as np
    bce = importlib.import_module(__package__)
    regs, ins = gate_registers(circuit_path, new_flag)
    regs = np.array(regs, dtype=np.uint32)
    ins = np.array(ins, dtype=np.uint32)
    stride = circ.info()["slot_stride"]
    sharded = [cc.lwe_read(regs + k * stride) for k in range(instances)]
    inputs = [cc.lwe_read(ins + k * stride) for k in range(instances)]
    ref = bce.Circuit(cc)
    ref.ReadBristol(circuit_path, new_flag=new_flag)
    ref.setInstances(instances)
    ref.Reset()
    ref.setEncrypted(True)
    ref.setEncryptMode(bce.FRESH)          # placeholders: the real input ciphertexts are copied in below
    ref.setRelevel(bool(relevel))
    widths = [w for w in ref.info()["n_input_bits"] if w]
    for k in range(instances):
        ref.SetInput([[0] * w for w in widths], instance=k)
    rstride = ref.info()["slot_stride"]
    for k in range(instances):
        cc.lwe_write(ins + k * rstride, inputs[k])
    ref.Clock()
    held = same = 0
    for k in range(instances):
        single = cc.lwe_read(regs + k * rstride)
        have = sharded[k].any(axis=1)          # a register this rank neither computed nor received is still all zero
        held += int(have.sum())
        eq = (sharded[k] == single).all(axis=1)
        bad = np.nonzero(have & ~eq)[0]
        assert bad.size == 0, "gate-sharded register %d of instance %d differs from the single-rank evaluation" % (int(regs[bad[0]]), k)
        same += int((have & eq).sum())
    ref.close()
    return held, same, int(regs.size)
