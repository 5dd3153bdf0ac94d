#!/usr/bin/env python3
"""Headline benchmark: encrypted gate-bootstraps/sec, AES-128 Bristol circuit, STD128_OPT GINX.

One "step" = one full encrypted evaluation (Circuit::Clock, verify off) of AES-expanded
(old Bristol, 27,692 gates = 66,415 gate bootstraps) on K independent input blocks evaluated in
lock-step per GPU (K = 32 by default); every dependent step of the schedule goes through bce_eval_gates_strided() to the
HIP blind-rotation + key-switch kernels (`--schedule dataflow`: the whole DAG through bce_dag_run, one persistent launch).
`--config N` selects another BASELINE.json config (2 adder_64bit, 4 sha256, 5 AES-expanded STD192 AP) with the same line.  Keys, parsing and input encryption are outside the timed
region (input ciphertexts are resident in HBM when timing starts).  Multi-GPU (`--gpus N`, one
rank per GPU under torch.distributed.run): keys replicated from the same seed, and -- north_star's partition --
EVERY STEP'S READY GATES SPLIT OVER THE RANKS by bootstrap weight, K x N input blocks in lock-step (the per-GPU load
of the one-GPU run: weak scaling), one RCCL all-gather per step of the outputs whose consumers sit on another rank.
That run is the line's `value` for N > 1; at N = 1 it is the plain run, so SCALE(N = 1) = BENCH by construction.
The collective-free form (every rank its own K blocks, nothing exchanged) rides along as `replicas`, the same
partition at K blocks in total (strong scaling) as `shard_gates`; `--shard instances|gates-strong` make either the headline.

`python3 bench.py --gpus N` (N > 1, no WORLD_SIZE in the environment) starts the N ranks itself: the parent
spawns N fresh child processes of this script BEFORE anything touches the GPU (it never does itself), forwards
rank 0's JSON line and exits non-zero if any child fails.  The line carries `rccl`: the ranks the collective spans as
counted BY the collective (all-reduce of ones; ncclCommCount of the library's own communicator for the in-library leg),
the RCCL version, one device per rank, per-rank step times.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "encrypted gate-bootstraps/sec (whole node), AES-128 Bristol ckt STD128 GINX"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def usable_cores():
    """host threads this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
        except Exception:
            pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def bristol_frontiers(path):
    """old-format Bristol netlist -> the reference's Clock() rounds (src/circuit.cpp:575-683: a gate runs in the round
    after its last input arrived; INV gates occupy rounds too), as lists of (op, in0, in1, out)."""
    toks = [l.split() for l in open(path) if l.strip()]
    n_in = int(toks[1][0]) + int(toks[1][1])
    level = {w: 0 for w in range(n_in)}
    rounds = []
    for t in toks[2:]:
        nin = int(t[0])
        ins = [int(x) for x in t[2:2 + nin]]
        out = int(t[2 + nin])
        l = 1 + max(level[w] for w in ins)
        level[out] = l
        while len(rounds) < l:
            rounds.append([])
        rounds[l - 1].append((t[-1], ins[0], ins[-1], out))
    return n_in, int(toks[0][1]), rounds


def cpu_baseline(circuit_path, paramset, method, seconds_budget=20.0):
    """the CPU restatement of the OpenFHE algorithm (oracle/, NOT OpenFHE itself) walks the REAL ready
    frontiers of the benchmark circuit the way the reference does -- one OpenMP task per gate of a frontier
    (src/circuit.cpp:698-710), XOR = NOT, NOT, AND, AND, OR (src/gate.cpp:198-202) -- on this box's host cores, for as
    many rounds as fit the time budget."""
    from oracle import oracle as O
    cores = usable_cores()
    o = O.Oracle(getattr(O, paramset), getattr(O, method))
    o.keygen(0x0FE5EED)
    n_in, n_wires, rounds = bristol_frontiers(circuit_path)
    max_x = max(sum(1 for g in r if g[0] == "XOR") for r in rounds)
    W = o.n + 1
    pool = np.zeros((n_wires + 2 * max_x, W), dtype=np.uint64)
    rng = np.random.default_rng(99)
    bits = {}
    for w in range(n_in):
        bits[w] = int(rng.integers(0, 2))
        pool[w] = o.encrypt(bits[w], w)
    boots, t_used, walked, checked = 0, 0.0, 0, 0
    for r in rounds:
        stage_a, stage_b = [], []
        x = 0
        for op, a, b, out in r:
            if op == "AND":
                stage_a.append((O.AND, a, b, out, 0, 0)); bits[out] = bits[a] & bits[b]
            elif op == "XOR":
                t1, t2 = n_wires + 2 * x, n_wires + 2 * x + 1
                x += 1
                stage_a += [(O.AND, a, b, t1, 0, 1), (O.AND, a, b, t2, 1, 0)]
                stage_b.append((O.OR, t1, t2, out, 0, 0)); bits[out] = bits[a] ^ bits[b]
            else:
                stage_a.append((O.OP_NOT, a, a, out, 0, 0)); bits[out] = 1 - bits[a]
        t0 = time.time()
        for st in (stage_a, stage_b):
            if st:
                boots += o.eval_gates(pool, st, nthreads=cores)
        t_used += time.time() - t0
        walked += 1
        for op, a, b, out in r[:: max(1, len(r) // 4)]:
            assert o.decrypt(pool[out]) == bits[out], "CPU baseline: gate output decrypts wrongly"
            checked += 1
        if t_used >= seconds_budget:
            break
    return {"value": boots / t_used, "unit": "gate-bootstraps/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
            "sample": "the first %d of %d ready-gate rounds of %s (%d gate-bootstraps, %s %s; XOR = 3 bootstraps), one OpenMP task per "
                      "gate of a round on %d threads as src/circuit.cpp:698-710 does, %.1f s, %d outputs decrypted and checked; "
                      "CPU restatement of the OpenFHE algorithm, not OpenFHE" % (
                          walked, len(rounds), os.path.basename(circuit_path), boots, paramset, method, cores, t_used, checked)}


def spawn_ranks(n, argv):
    """Parent of a self-launched multi-GPU run: N child processes of this script, one rank per GPU, rendezvous
    on 127.0.0.1.  The parent never initialises the GPU (no torch import, no HIP call) and never execs."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        sys.exit(1)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--instances", type=int, default=32, help="AES blocks evaluated in lock-step per GPU")
    ap.add_argument("--circuit", default="AES-expanded.txt")
    ap.add_argument("--paramset", default="STD128_OPT")
    ap.add_argument("--method", choices=["GINX", "AP"], default="GINX")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=None,
                    help="a BASELINE.json config by number: 2 adder_64bit (K = 256), 3 the headline (default), 4 sha256 (K = 16), "
                         "5 AES-expanded STD192 AP (K = 8); explicit --circuit / --paramset / --method / --instances win over it")
    ap.add_argument("--schedule", choices=["steps", "dataflow", "graph"], default="steps",
                    help="steps: one launch per dependent step of the bootstrap-depth schedule (default); dataflow: the whole "
                         "bootstrap DAG in one persistent launch with device-side ready queues (bce_dag_*); graph: the step "
                         "schedule's launches replayed as one hipGraph per evaluation (bce_plan_run; timed as one unit)")
    ap.add_argument("--shard", choices=["gates", "instances", "gates-strong"], default="gates",
                    help="N > 1 headline: gates = every step's gates split over the ranks, K x N blocks in lock-step (north_star's partition "
                         "at the one-GPU load, weak scaling; default); instances = independent replicas, no data-path collective; "
                         "gates-strong = the gate split at K blocks in total")
    ap.add_argument("--replica-steps", type=int, default=2, help="N > 1: timed steps of the collective-free replica leg (0 = skip)")
    ap.add_argument("--no-relevel", dest="relevel", action="store_false",
                    help="schedule by gate level exactly like the reference's Clock() rounds (496 launches for AES) instead of "
                         "by bootstrap depth (416 launches, identical ciphertexts)")
    ap.set_defaults(relevel=True)
    ap.add_argument("--xor-fast", action="store_true", help="opt-in native XOR (NOT the reference's XOR = 3 bootstraps)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fresh-inputs", action="store_true",
                    help="encrypt inputs FRESH instead of the reference-shaped BOOTSTRAPPED default (profiled runs: the refresh launches of "
                         "SetInput use the same kernel and would mix into its rocprofv3 statistics; they are setup, outside the timed region)")
    ap.add_argument("--no-dataflow-leg", action="store_true",
                    help="N = 1, step schedule: skip the secondary run of the same workload as ONE persistent launch (bce_dag_run)")
    ap.add_argument("--no-block-latency", action="store_true",
                    help="skip the K = 1 single-block leg (profiled runs: keeps the kernel statistics to the timed workload)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--gates-timeout", type=int, default=540,
                    help="N > 1: watchdog (s, from the moment the fallback line exists) over the gate-sharded runs and the teardown")
    ap.add_argument("--gates-steps", type=int, default=2, help="N > 1: timed steps of the strong-scaling gate-sharded leg (0 = skip)")
    args = ap.parse_args()
    if args.config is not None:
        preset = {2: ("adder_64bit.txt", "STD128_OPT", "GINX", 256), 3: ("AES-expanded.txt", "STD128_OPT", "GINX", 32),
                  4: ("sha256_new.txt", "STD128_OPT", "GINX", 16), 5: ("AES-expanded.txt", "STD192", "AP", 8)}[args.config]
        given = " ".join(sys.argv[1:])
        if "--circuit" not in given: args.circuit = preset[0]
        if "--paramset" not in given: args.paramset = preset[1]
        if "--method" not in given: args.method = preset[2]
        if "--instances" not in given: args.instances = preset[3]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])      # does not return

    import torch
    bce = importlib.import_module("openfhe-boolean-circuit-evaluator_amd")
    import kat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    # test-only knobs (a one-GPU box rehearsing the N > 1 path): all ranks on device 0, gloo instead of RCCL
    if os.environ.get("BCE_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("BCE_BENCH_BACKEND", "nccl")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend != "nccl":   # host-side rendezvous needs no GPU: do it first, so that start-up problems show as such
            dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
            sys.stderr.write("bench.py: rank %d/%d rendezvous ok (%s)\n" % (rank, world, backend))
    if not torch.cuda.is_available():
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 and backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        sys.stderr.write("bench.py: rank %d/%d rendezvous ok (nccl = RCCL)\n" % (rank, world))
    red_dev = "cuda" if backend == "nccl" else "cpu"  # where the few scalars of the result are reduced

    # ---- setup (untimed): context, keys (same seed on every rank = replicated), circuit, inputs
    t_setup = time.time()
    cc = bce.BinFHEContext(getattr(bce, args.paramset), getattr(bce, args.method), device=local_rank)
    t_kg = time.time()
    cc.KeyGen(0x0FE5EED)            # explicit seed: the SAME key set on every rank (synthetic benchmark keys)
    keygen_s = time.time() - t_kg
    path = os.path.join(ROOT, "tests", "golden", "circuits", args.circuit)
    K_total = args.instances
    first_run = {"t": None}

    def run_mode(shard_mode, steps, warmup, relevel, exchange="callback", K_run=None, schedule=None):
        """One timed run.  shard_mode 0 (instances): every rank evaluates ITS OWN K input blocks with its own
        circuit object -- independent units, no data-path collective (only the barrier / reductions of this
        script).  shard_mode 1 (gates): ONE set of K blocks, every level's gates split over the ranks by bootstrap
        weight, boundary ciphertexts exchanged per level (RCCL allgather)."""
        K_run = K_run or K_total
        t_begin = time.time()
        if first_run["t"] is None:
            first_run["t"] = t_begin
        circ = bce.Circuit(cc)
        circ.ReadBristol(path, new_flag=args.circuit.startswith("sha256_new"))
        if args.xor_fast:
            circ.setXorFast(True)
        gates = world > 1 and shard_mode == 1
        if gates:
            cc.set_encrypt_seed(0x0FE5EED)   # every rank must encrypt IDENTICAL input ciphertexts
        circ.setRelevel(bool(relevel))
        schedule = schedule or args.schedule
        if schedule == "dataflow" and not gates:
            circ.setDataflow(True)
        if schedule == "graph" and not gates:
            circ.setGraph(True)
        info = circ.info()
        circ.setInstances(K_run)
        xch = None
        if gates:
            # boundary ciphertexts: the library's own RCCL all-gather on the engine stream when RCCL is the backend
            # (falls back, on every rank together, to the torch.distributed callback if RCCL cannot be initialised)
            xch = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.dist").Exchange(
                circ, 1, encrypted=True, device=torch.device("cuda", local_rank),
                in_library=(backend == "nccl" and exchange == "rccl"))
        rng = np.random.default_rng(12345 + (0 if gates else rank))
        widths = info["n_input_bits"]
        inputs = []
        for k in range(K_run):
            if args.circuit == "AES-expanded.txt" and k < 2:  # the reference's two vectors first
                v = [x for x in kat.AES_VECTORS if x["circuit"] == "AES-expanded"][k]
                inputs.append(kat.aes_case(v)[0])
            else:
                inputs.append([rng.integers(0, 2, w).tolist() for w in widths if w])
        # plaintext pass = expected outputs (host logic only; under gate sharding it runs the same exchange plan on bits)
        circ.Reset()
        circ.setPlaintext(True)
        for k in range(K_run):
            circ.SetInput(inputs[k], instance=k)
        circ.Clock()
        expect = [circ.Outputs(k)[0] for k in range(K_run)]
        circ.Reset()
        circ.setEncrypted(True)
        if args.fresh_inputs:
            circ.setEncryptMode(bce.FRESH)
        for k in range(K_run):
            circ.SetInput(inputs[k], instance=k)
        cc.synchronize()
        t_ready = time.time()
        df_active = circ.dataflowActive()
        xmod = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.dist") if gates else None
        k_ident = min(2, K_run)
        if gates:
            # ciphertext identity below compares the registers this rank holds after the run with a single-rank evaluation:
            # start from zeros, so that "holds" is readable from the pool (the pool may carry an earlier run's registers)
            regs, _ = xmod.gate_registers(path, args.circuit.startswith("sha256_new"))
            stride0 = circ.info()["slot_stride"]
            zeros = np.zeros((len(regs), cc.n + 1), dtype=np.uint64)
            for k in range(k_ident):
                cc.lwe_write(np.array(regs, dtype=np.uint32) + k * stride0, zeros)

        def step():
            circ.Rearm()
            circ.Clock()

        for _ in range(warmup):
            step()
        cc.synchronize()
        cc.timing_reset()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(steps):
            step()
        cc.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.time() - t0
        tm = cc.timing()
        # correctness of the timed work: decrypted outputs of every instance == plaintext evaluation
        verified = [circ.Outputs(k)[0] for k in range(K_run)] == expect
        st = circ.stats()
        total_boot = float(tm["bootstraps"])
        per_rank_s = [elapsed]
        if dist is not None:
            t = torch.tensor([elapsed, total_boot, 0.0 if verified else 1.0, float(st["exchanged_cts"])], dtype=torch.float64, device=red_dev)
            every = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(every, t)
            per_rank_s = [float(e[0]) for e in every]
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            elapsed = float(tmax[0])
            total_boot = float(t[1])
            verified = float(tmax[2]) == 0.0      # every rank's outputs
            xcts = float(t[3])
        else:
            xcts = 0.0
        identity = None
        if gates:
            # every bootstrapped register this rank computed or received == the un-sharded evaluation of the same input
            # ciphertexts, bit for bit (first instances; replicated keys make the bootstraps deterministic)
            try:
                held, same, per_inst = xmod.check_against_single_rank(cc, circ, path, args.circuit.startswith("sha256_new"), relevel, k_ident)
                ident_ok = held == same
            except AssertionError as e:
                held, same, per_inst, ident_ok = -1, -1, -1, False
                sys.stderr.write("bench.py: rank %d: %s\n" % (rank, e))
            t = torch.tensor([float(held), 0.0 if ident_ok else 1.0], dtype=torch.float64, device=red_dev)
            tmx = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
            identity = {"instances_checked": k_ident, "bootstrapped_registers_per_instance": per_inst,
                        "registers_held_over_all_ranks": int(t[0]), "identical_to_single_rank_evaluation": bool(float(tmx[1]) == 0.0)}
            verified = verified and identity["identical_to_single_rank_evaluation"]
            cc.set_encrypt_seed(None)
        rccl_info = None
        if xch is not None and xch.in_library:
            try:
                rccl_info = cc.rccl_comm_info()      # ncclCommCount / rank / device of the communicator the exchange ran on
            except Exception as e:
                rccl_info = {"error": repr(e)}
        circ.close()
        return {"elapsed": elapsed, "total_boot": total_boot, "verified": verified, "tm": tm, "info": info, "relevel": relevel,
                "K_run": K_run, "shard_mode": shard_mode if world > 1 else 0, "per_rank_ms_per_step": [round(x / steps * 1e3, 3) for x in per_rank_s],
                "in_library_comm": rccl_info,
                "launches_per_step": st["sublaunches"], "dataflow": df_active, "identity": identity,
                "dag_last_run": cc.dag_last_run() if (schedule == "dataflow" and not gates) else None,
                "exchanges_per_step": st["exchanges"], "exchanged_cts_per_step": xcts, "steps": steps, "t_ready": t_ready, "t_begin": t_begin,
                "exchange_path": ("in-library ncclAllGather on the engine stream (no host sync)" if (xch is not None and xch.in_library) else
                                  ("torch.distributed all_gather_into_tensor callback after a stream sync" + (" [in-library RCCL unavailable: %s]" % xch.why if xch is not None and xch.why not in ("", "not requested") else "")) if xch is not None else "none")}

    # ---- which run is the headline.  N = 1: the plain run.  N > 1: north_star's partition -- every step's gates split over the
    # ranks, K x N blocks in lock-step (default); the collective-free replica run goes FIRST and short: it is the line that
    # is printed if the gate-sharded run fails or never returns from a collective (watchdog below).
    import threading
    emit_lock = threading.Lock()
    state = {"printed": False}
    out = None
    head_mode = {"gates": "gates", "instances": "instances", "gates-strong": "gates-strong"}[args.shard] if world > 1 else "single"
    exch = os.environ.get("BCE_EXCHANGE", "callback")

    def rccl_object():
        """what the collective itself says it spans (N > 1)"""
        ones = torch.ones(1, dtype=torch.float64, device=red_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # counted BY the collective: 1 from every rank it reaches
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "device": local_rank, "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                "pci_bus_id": getattr(props, "pci_bus_id", None)}
        every = [None] * world
        dist.all_gather_object(every, mine)
        ver = lib_ver = None
        try:
            ver = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
        except Exception:
            pass
        return {"backend": "nccl (= RCCL on ROCm)" if backend == "nccl" else backend + " (rehearsal: not RCCL)",
                "rccl_ranks": int(round(float(ones[0]))), "counted_by": "all-reduce(SUM) of 1 per rank over the process group the exchanges use",
                "rccl_version": ver, "rccl_version_seen_by_the_library": lib_ver,
                "devices": every, "distinct_devices": len({(d["device"], d["uuid"]) for d in every})}

    REP = REP_err = None
    block_latency_s = None
    wd = None
    if world > 1:
        rccl_obj = rccl_object()
        if head_mode != "instances" and args.replica_steps > 0:
            try:
                REP = run_mode(0, args.replica_steps, 1, args.relevel)
            except Exception as e:
                REP_err = repr(e)
    shard_mode = 0 if head_mode in ("single", "instances") else 1
    K_head = args.instances * world if head_mode == "gates" else args.instances

    def on_timeout():
        with emit_lock:
            if rank == 0 and not state["printed"]:
                msg = "the gate-sharded run / teardown did not finish within %d s: abandoned" % args.gates_timeout
                if out is not None:
                    out.setdefault("shard_gates", {"error": msg})
                    emit()
                elif REP is not None:
                    fb = make_line(REP, "instances", args.replica_steps, 1)
                    fb["error"] = "headline (gates split over the ranks) unavailable: " + msg + "; this line is the collective-free replica run"
                    fb["headline_fallback"] = "replicas"
                    fb["rccl"] = rccl_obj
                    state["printed"] = True
                    print(json.dumps(fb), flush=True)
            sys.stderr.write("bench.py: rank %d watchdog fired after %d s\n" % (rank, args.gates_timeout))
            sys.stderr.flush()
            # a process that has touched the GPU and hangs in a collective is a FAILURE of the run: the line is out,
            # the exit code says so (the parent prints "ranks failed"); no restart, no re-exec
            os._exit(3)

    if world > 1:
        wd = threading.Timer(args.gates_timeout, on_timeout)
        wd.daemon = True
        wd.start()
    R = R_err = None
    try:
        if os.environ.get("BCE_BENCH_TEST_HANG") == "head" and rank == world - 1:
            time.sleep(1e6)          # test hook (tests/test_bench_launch.py): one rank never reaches the headline's collectives
        R = run_mode(shard_mode, args.steps, args.warmup, args.relevel, exchange=exch, K_run=K_head)
    except Exception as e:
        if world == 1 or REP is None:
            raise
        R_err = repr(e)
        sys.stderr.write("bench.py: rank %d: headline run failed: %s\n" % (rank, R_err))
    if R is None:
        R, head_mode = REP, "instances"     # the fallback line (error field set below)
    # also the end-to-end latency of ONE input block (K = 1: every dependent launch is a single narrow frontier)
    def single_block_latency():
        c1 = bce.Circuit(cc)
        c1.ReadBristol(path, new_flag=args.circuit.startswith("sha256_new"))
        c1.setRelevel(bool(args.relevel))
        c1.Reset()
        c1.setEncrypted(True)
        w = [x for x in c1.info()["n_input_bits"] if x]
        c1.SetInput([np.random.default_rng(5).integers(0, 2, x).tolist() for x in w])
        c1.Clock()
        c1.Rearm()
        cc.synchronize()
        t0 = time.time()
        c1.Clock()
        dt = time.time() - t0
        c1.close()
        return dt
    block_latency_s = None if (args.no_block_latency or (world > 1 and R is REP)) else single_block_latency()

    def load_profile(name):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            return None

    def make_line(R, mode, steps, warmup):
        """the JSON line of one timed run (rank 0).  mode: single | gates (K x N blocks, gates split) | gates-strong | instances"""
        elapsed, total_boot, verified, tm, info = R["elapsed"], R["total_boot"], R["verified"], R["tm"], R["info"]
        shard_mode = R["shard_mode"]
        setup_s = first_run["t"] - t_setup + (R["t_ready"] - R["t_begin"])     # context + keygen, + this run's parsing, plaintext pass, input encryption
        default_cmd = (args.instances == 32 and args.circuit == "AES-expanded.txt" and args.paramset == "STD128_OPT" and args.method == "GINX"
                       and mode in ("single", "gates", "instances") and R["relevel"] and not args.xor_fast)
        # the three schedules run the same blind-rotation body (lat_bootstrap): one instruction model for all of them
        same_body = args.paramset in ("STD128_OPT", "STD128") and args.method == "GINX" and R["relevel"] and not args.xor_fast
        parts = cc.bytes_per_bootstrap_parts()         # {"bsk", "ksk", "ct"} at this build's widths (SURVEY 8(d) formula)
        bpb = parts["bsk"] + parts["ksk"] + parts["ct"]
        pr = cc.params
        ginx = args.method == "GINX"
        # the key read ONCE (what a launch must move if every bootstrap of it shared the key perfectly); AP bootstraps
        # walk digit-selected keys of their own, so nothing is shared by construction there
        bsk_once = (cc.bsk_word_bytes() * pr["n"] * 2 * (2 * pr["dG"]) * 2 * pr["N"]) if ginx else None
        # the DOMINANT blind-rotation kernel of this run (launch size and schedule pick between kernels)
        dom = max(tm["by_kernel"], key=lambda k: k["ms"])
        br_s = dom["ms"] / 1e3
        per_launch = dom["bootstraps"] / max(1, dom["launches"])
        avg_launch_ms = dom["ms"] / max(1, dom["launches"])
        # saturated launches of the split-transform kernel (and the persistent kernel) run the tail (KSK row gather) in their
        # epilogue: the KSK rows are then bytes of the blind-rotation kernel, and no tail kernel exists
        fused = tm["fused_tail_launches"] >= tm["blind_rotate_launches"] > 0
        br_bytes = parts["bsk"] + parts["ct"] + (parts["ksk"] if fused else 0)   # what the dominant kernel itself moves per bootstrap
        achieved = (br_bytes * dom["bootstraps"] / br_s) / 1e9 if br_s > 0 else 0.0
        tail_s = tm["tail_ms"] / 1e3
        tail_achieved = (parts["ksk"] * tm["bootstraps"] / tail_s) / 1e9 if tail_s > 0 else 0.0
        # committed PMC passes of this same command / of one saturated launch of this kernel (rocprofv3 cannot run inside
        # the timed region: REPLAYED constants, named with their files; the newest round's files first)
        def newest(names):
            for n in names:
                d = load_profile(n)
                if d is not None:
                    return d, n
            return None, None
        cfg5 = args.paramset == "STD192" and args.method == "AP"
        traffic, traffic_file = newest(["r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"]) if default_cmd else (None, None)
        valu, valu_file = newest(["r04_valu_model.json", "r03_valu_model.json", "r02_valu_model.json"]) if not cfg5 else newest(["r04_cfg5_roofline.json", "r03_cfg5_roofline.json", "r02_cfg5_roofline.json"])
        traffic_bytes = traffic["hbm_bytes_per_launch"] if traffic and traffic.get("bench_kernel") == dom["kernel"] else None
        compulsory = (bsk_once + (parts["ct"] + (parts["ksk"] if fused else 0)) * per_launch) if bsk_once is not None else br_bytes * per_launch
        hbm = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "what": "the task's convention: ALGORITHMIC bytes of this kernel per bootstrap (SURVEY 8(d) formula at this build's widths) x "
                    "bootstraps / its time by HIP events on the engine stream",
            "bytes_per_bootstrap": {"bsk_rows": parts["bsk"], "ct_io": parts["ct"], "ksk_rows": parts["ksk"], "total": bpb,
                                    "billed_to_this_kernel": br_bytes, "tail_fused_into_this_kernel": bool(fused)},
            "algorithmic_bytes_per_launch": br_bytes * per_launch,
            "compulsory_bytes_per_launch": compulsory,
            "compulsory_frac": (compulsory / (avg_launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if avg_launch_ms > 0 else None,
            "counter_traffic_bytes_per_launch": traffic_bytes,
            "counter_traffic_frac_of_peak": (traffic_bytes / (avg_launch_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if traffic_bytes else None,
            "wasted_traffic_ratio": (traffic_bytes / compulsory) if traffic_bytes else None,
            "traffic_source": ("profiles/%s: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this default command "
                               "(tools/collect_evidence.sh) -- a committed constant replayed here, NOT measured in this run" % traffic_file) if traffic_bytes else None,
        }
        roof = dict(hbm)
        roof.update({"kernel": dom["kernel"], "traffic": traffic_bytes, "bootstraps_per_launch": per_launch, "avg_launch_ms": avg_launch_ms,
                     "launches": dom["launches"], "share_of_blind_rotation_time": dom["ms"] / max(1e-9, tm["blind_rotate_ms"]),
                     "other_blind_rotation_kernels": [k for k in tm["by_kernel"] if k is not dom and k["launches"]],
                     "tail": ({"kernel": "none: extract + ModSwitch + KeySwitch + ModSwitch run in the epilogue of the blind-rotation kernel (fused_tail)",
                               "ms_total": tm["tail_ms"]} if fused else
                              {"kernel": "k_tail_gather + k_tail_finish", "bound": "hbm", "ms_total": tm["tail_ms"],
                               "avg_launch_ms": tm["tail_ms"] / max(1, tm["blind_rotate_launches"]),
                               "achieved": tail_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tail_achieved / HBM_PEAK_GBS,
                               "note": "KSK row gather: algorithmic = N*dKS rows of (n+1) words per bootstrap"})})
        # The roof that BINDS goes on top.  32-bit path (GINX, Q < 2^28): the key is served from L2 / Infinity Cache across the
        # batch, the HBM convention exceeds 1 and says nothing; the kernel is bound by integer-VALU issue.  64-bit AP path
        # (config 5): the digit-selected keys are hardly shared, HBM binds first by the convention and fp64 issue is next.
        vm = None
        if valu and not cfg5 and (valu.get("bench_kernel") == dom["kernel"] or (same_body and args.schedule in ("dataflow", "graph"))):
            simds = 4 * (valu.get("cu_count") or 256)
            vm = {"insts_per_bootstrap": valu["valu_insts_per_bootstrap"], "ns_per_wave_inst_per_simd": valu["ns_per_wave_inst_per_simd"], "simds": simds}
            vm_src = "profiles/%s (SQ_INSTS_VALU of one saturated launch + instruction mix of the ISA + measured issue cost per opcode; committed constants, the launch time is this run's)%s" % (
                valu_file, "" if valu.get("bench_kernel") == dom["kernel"] else "; counted on the per-step kernel %s, whose blind-rotation body this schedule's kernel shares" % valu.get("bench_kernel"))
        elif valu and cfg5 and "valu" in valu:
            vm = {"insts_per_bootstrap": valu["valu"]["insts_per_bootstrap"], "ns_per_wave_inst_per_simd": valu["valu"]["ns_per_wave_inst_per_simd"], "simds": 1024}
            vm_src = "profiles/%s (SQ_INSTS_VALU + fp64 issue cost of the step loop's mix; committed constants, the launch time is this run's)" % valu_file
        if vm:
            peak_ginst = vm["simds"] / vm["ns_per_wave_inst_per_simd"]                                  # G wave-instructions / s the chip can issue
            ach_ginst = vm["insts_per_bootstrap"] * dom["bootstraps"] / br_s / 1e9 if br_s > 0 else 0.0
            valu_obj = {"bound": "valu", "what": ("integer" if not cfg5 else "fp64") + " VALU issue: wave-instructions the kernel executes per second against what "
                        "the chip's %d SIMDs can issue for this kernel's instruction mix" % vm["simds"],
                        "achieved": ach_ginst, "peak": peak_ginst, "unit": "G wave-inst/s", "frac": ach_ginst / peak_ginst,
                        "insts_per_bootstrap": vm["insts_per_bootstrap"], "ns_per_wave_inst_per_simd": vm["ns_per_wave_inst_per_simd"],
                        "floor_ms_per_launch": vm["insts_per_bootstrap"] * per_launch * vm["ns_per_wave_inst_per_simd"] / vm["simds"] / 1e6,
                        "source": vm_src}
            # binding roof on top, the HBM figures below it
            for k in ("bound", "what", "achieved", "peak", "unit", "frac"):
                roof[k] = valu_obj[k]
            roof["valu"] = valu_obj
            roof["hbm"] = hbm
            for k in ("bytes_per_bootstrap", "algorithmic_bytes_per_launch", "compulsory_bytes_per_launch", "compulsory_frac", "counter_traffic_bytes_per_launch",
                      "counter_traffic_frac_of_peak", "wasted_traffic_ratio", "traffic_source"):
                roof.pop(k, None)
            if not cfg5:
                roof["note"] = ("bound = integer-VALU issue (what limits this kernel); `hbm` keeps the byte convention (frac > 1: the 62.8 MiB key is "
                                "served from L2 / Infinity Cache across the batch), the counter-based traffic and the wasted-traffic ratio")
            else:
                roof["note"] = ("bound = fp64 issue: with every bootstrap of a launch on the SAME keys (no HBM traffic to speak of) this kernel is only 8 % "
                                "faster (tools/ap_key_locality.py STD192, profiles/r03_cfg5_key_locality.log), so the bytes in `hbm` -- AP keys are "
                                "digit-selected per bootstrap and hardly shared -- are not what holds it; both fractions sit near 0.6 because the part runs "
                                "it at 95 % of its board power")
        if R["dataflow"]:
            roof["dataflow"] = R["dag_last_run"]
        out = {
            "metric": METRIC,
            "value": total_boot / elapsed,
            "unit": "gate-bootstraps/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if mode == "gates-strong" else "weak",
            "vs_baseline": None,
            "dtype": "u32" if not cc.is64() else ("f64 (exact integers < 2^53 in IEEE doubles)" if cc.fp64() else "u64"),
            "data": "synthetic",
            "config": {
                "workload": "%s (%d gates, %d gate-bootstraps/eval; %d kernel launches per evaluation as run) %s %s, "
                            "%s, verify off" % (
                                args.circuit, info["n_gates"] - info["n_output_bits"], info["n_bootstraps"],
                                R["launches_per_step"], args.paramset, args.method,
                                "%d input blocks in lock-step per GPU" % args.instances if mode in ("single", "instances") else
                                "%d input blocks in lock-step, every step's ready gates split over the %d GPUs (%s)" % (
                                    R["K_run"], world, "%d blocks per GPU: the one-GPU load" % args.instances if mode == "gates" else "strong scaling")),
                "baseline_config": args.config if args.config is not None else (3 if default_cmd else None),
                "instances_per_gpu": args.instances if mode != "gates-strong" else R["K_run"] / world, "instances_total": R["K_run"] * (world if mode == "instances" else 1),
                "sharding": {"single": "none (one GPU)", "instances": "instances: independent replicas, no data-path collective",
                             "gates": "gates: north_star's partition -- every step's ready gates split over the ranks by bootstrap weight, key replicated, "
                                      "crossing outputs all-gathered after the step; K x N blocks in lock-step",
                             "gates-strong": "gates, K blocks in total (strong scaling)"}[mode],
                "schedule": ("dataflow: the whole bootstrap DAG in one persistent launch, device-side ready queues (identical ciphertexts)" if R["dataflow"] else
                             ("bootstrap-depth levels (NOTs folded, steps filled by slack up to the launch staircase, identical ciphertexts)" +
                              (", every evaluation's launches replayed as one hipGraph" if args.schedule == "graph" else "")) if R["relevel"] else "gate levels (reference Clock rounds)"),
                "launches_per_step": R["launches_per_step"],
                "reference_clock_rounds": info["n_levels"], "reference_sub_launches": info["n_sublaunches"],
                "xor": "XOR_FAST (opt-in, 1 bootstrap)" if args.xor_fast else "NOT,NOT,AND,AND,OR (reference, 3 bootstraps)",
                "bootstraps_per_step": int(total_boot / steps),
                "forward_transforms_per_blind_rotation_step": cc.forward_transforms_per_step(),
                "gates_per_s": (info["n_gates"] - info["n_output_bits"]) * R["K_run"] * (world if mode == "instances" else 1) * steps / elapsed,
                "single_block_latency_s": None if block_latency_s is None else round(block_latency_s, 4),
                "outputs_verified": bool(verified), "setup_s": round(setup_s, 2), "keygen_s": round(keygen_s, 3),
                "input_encryption": ("FRESH (--fresh-inputs)" if args.fresh_inputs else
                                     "cc.Encrypt default of OpenFHE v1.0.x: BOOTSTRAPPED (one refresh bootstrap per input bit, %d per block, inside setup_s, outside the timed region)" % info["n_input_gates"]),
                "host_share_of_step": round(1.0 - (tm["blind_rotate_ms"] + tm["tail_ms"]) / (elapsed * 1e3), 4),
                "exchanges_per_step": R["exchanges_per_step"], "exchanged_cts_per_step": R["exchanged_cts_per_step"],
                "collective": "none in the timed region" if shard_mode == 0 else
                              "one all-gather of the crossing outputs per step (%s); %s" % (backend, R["exchange_path"]),
                "exchange_path": R["exchange_path"], "ciphertext_identity": R["identity"],
                "per_rank_ms_per_step": R["per_rank_ms_per_step"],
            },
            "roofline": roof,
        }

        return out

    # ---- N > 1: assemble.  The fallback (replica) line exists since before the watchdog was armed; from here on everything
    # runs under it, so that a collective that never returns costs the secondary objects, not the line.
    def emit():
        if rank == 0 and not state["printed"]:
            state["printed"] = True
            if not out.get("config", {}).get("outputs_verified", False):
                out.setdefault("error", "decrypted outputs differ from the plaintext evaluation")
            print(json.dumps(out), flush=True)

    verified = R["verified"]
    if rank == 0:
        out = make_line(R, head_mode, R["steps"], args.warmup if R is not REP else 1)
        if world > 1:
            out["rccl"] = dict(rccl_obj, per_rank_ms_per_step=R["per_rank_ms_per_step"], exchange_path=R["exchange_path"])
            if R is REP and args.shard != "instances":
                out["error"] = "headline (gates split over the ranks) failed: %s; this line is the collective-free replica run" % R_err
                out["headline_fallback"] = "replicas"
            elif REP is not None:
                out["replicas"] = {
                    "what": "every rank its own %d blocks, no data-path collective (linear by construction): the per-GPU rate the gate-sharded headline is to be read against" % args.instances,
                    "value": REP["total_boot"] / REP["elapsed"], "unit": "gate-bootstraps/s", "scaling": "weak", "steps": REP["steps"], "warmup": 1,
                    "ms_per_step": REP["elapsed"] / REP["steps"] * 1e3, "per_rank_ms_per_step": REP["per_rank_ms_per_step"],
                    "outputs_verified": bool(REP["verified"])}
                out["headline_over_replicas"] = (R["total_boot"] / R["elapsed"]) / (REP["total_boot"] / REP["elapsed"])
            elif REP_err is not None:
                out["replicas"] = {"error": REP_err}
    G = G_err = G2 = G2_err = None
    if world > 1 and R is not REP and head_mode == "gates":
        if args.gates_steps > 0:
            try:        # a failure of a secondary run must not cost the headline line
                if os.environ.get("BCE_BENCH_TEST_HANG") == "1" and rank == world - 1:
                    time.sleep(1e6)      # test hook (tests/test_bench_launch.py): one rank never reaches the collective
                G = run_mode(1, args.gates_steps, 1, args.relevel, exchange=exch)
            except Exception as e:
                G_err = repr(e)
        # the library's own RCCL all-gather on the engine stream has never run between two devices (no multi-GPU node was
        # available to the builder): the verified torch.distributed callback carries the legs above, and this short extra
        # leg puts the in-library path on record whenever a node is there (outputs verified like every other run)
        if backend == "nccl" and exch != "rccl" and G_err is None:
            try:
                G2 = run_mode(1, 1, 1, args.relevel, exchange="rccl")
            except Exception as e:
                G2_err = repr(e)
    if rank == 0:
        pred = None
        try:   # what the host-side model expects of this partition on 1, 2, 4, 8 GPUs (a prediction to check SCALE runs against)
            pred = importlib.import_module("openfhe-boolean-circuit-evaluator_amd.predict")
        except Exception:
            pass
        can_predict = pred is not None and args.paramset in ("STD128_OPT", "STD128") and R["relevel"]
        if world > 1 and head_mode == "gates" and R is not REP and can_predict:
            try:
                out["predicted"] = pred.predict_gate_sharding(path, args.circuit.startswith("sha256_new"), args.instances, weak=True)
            except Exception as e:
                out["predicted"] = {"error": repr(e)}
        if G_err is not None:
            out["shard_gates"] = {"error": G_err}
        if G is not None:
            out["shard_gates"] = {
                "what": "the same partition at %d blocks IN TOTAL (strong scaling): every step's gates split over the %d ranks by bootstrap weight; "
                        "crossing outputs exchanged with one all-gather per step (%s); %s" % (
                            args.instances, world, backend, "bootstrap-depth schedule, steps filled by slack up to the staircase of all ranks together"
                            if G["relevel"] else "reference gate-level schedule"),
                "value": G["total_boot"] / G["elapsed"], "unit": "gate-bootstraps/s", "scaling": "strong",
                "ms_per_step": G["elapsed"] / G["steps"] * 1e3, "steps": G["steps"], "warmup": 1,
                "exchanges_per_step": G["exchanges_per_step"], "exchanged_cts_per_step": G["exchanged_cts_per_step"],
                "exchange_path": G["exchange_path"], "per_rank_ms_per_step": G["per_rank_ms_per_step"],
                "outputs_verified": bool(G["verified"]),
                "ciphertext_identity": G["identity"],
            }
            if can_predict:
                try:
                    out["shard_gates"]["predicted"] = pred.predict_gate_sharding(path, args.circuit.startswith("sha256_new"), args.instances)
                except Exception as e:
                    out["shard_gates"]["predicted"] = {"error": repr(e)}
            if not G["verified"]:      # reported where it belongs; the headline run has its own flag (config.outputs_verified)
                out["shard_gates"]["error"] = "decrypted outputs differ from the plaintext evaluation"
        if G2 is not None or G2_err is not None:
            out["shard_gates_in_library_rccl"] = ({"error": G2_err} if G2 is None else {
                "what": "the strong-scaling leg once more with the exchange as ncclAllGather issued by the library on the engine stream (no host sync)",
                "value": G2["total_boot"] / G2["elapsed"], "unit": "gate-bootstraps/s", "ms_per_step": G2["elapsed"] / G2["steps"] * 1e3,
                "steps": G2["steps"], "exchange_path": G2["exchange_path"], "outputs_verified": bool(G2["verified"]),
                "rccl_ranks": (G2["in_library_comm"] or {}).get("ranks"), "communicator": G2["in_library_comm"],
                "counted_by": "ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the library's own communicator (rank 0's view)",
                "ciphertext_identity": G2["identity"]})
        if world > 1 and "rccl" in out:
            try:      # asked LAST: the engine dlopen()s the system's librccl for it, next to the one torch brought along
                out["rccl"]["rccl_version_seen_by_the_library"] = int(bce.lib().bce_rccl_version()) or None
            except Exception:
                pass
        if world == 1 and args.schedule == "steps" and R["relevel"] and not args.no_dataflow_leg and cc.dag_supported():
            # the same workload as ONE persistent launch with device-side ready queues (identical registers); secondary: the
            # headline keeps the per-step launches its per-launch roofline evidence is collected on
            try:
                D = run_mode(0, 1, 1, True, schedule="dataflow")
                out["dataflow"] = {"what": "the same K blocks, whole bootstrap DAG in one persistent launch per evaluation (bce_dag_run, Circuit.setDataflow)",
                                   "value": D["total_boot"] / D["elapsed"], "unit": "gate-bootstraps/s", "ms_per_step": D["elapsed"] / D["steps"] * 1e3,
                                   "steps": D["steps"], "warmup": 1, "launches_per_step": D["launches_per_step"], "active": bool(D["dataflow"]),
                                   "outputs_verified": bool(D["verified"]), "scheduler": D["dag_last_run"]}
            except Exception as e:
                out["dataflow"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(path, args.paramset, args.method, args.cpu_seconds)
        with emit_lock:
            emit()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if wd is not None:
        wd.cancel()
    if not verified:
        sys.exit(2)


if __name__ == "__main__":
    main()
