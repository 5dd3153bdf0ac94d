"""Host-side model of a gate-sharded run (one process per GPU, every step's gates split over the ranks, one all-gather
of the crossing outputs per step): PREDICTED time per evaluation for 1, 2, 4, 8 GPUs from the per-rank plans the circuit
runtime really builds and the launch-time staircase measured on one MI355X.  No GPU needed.

Nothing here has been checked against a multi-GPU run (none was available to the builder); the numbers exist so that the
first SCALE run has something to be compared with.  Model:

  * a rank's share of a step holding n bootstraps x K instances costs launch_ms(n K): one bootstrap is one workgroup, so a
    launch takes one bootstrap latency up to `lone` (= #CUs) bootstraps and one round per `full` (= resident workgroups at
    saturation) beyond -- profiles/r02_launch_curve.log, STD128_OPT / GINX;
  * a step ends when its slowest rank ends (the exchange is a barrier): max over ranks;
  * a step that publishes costs exch_us (pack kernel + collective launch + scatter kernels) plus the all-gather's bytes
    over one xGMI link direction: every rank contributes widest x K ciphertexts (the collective is padded to the widest
    rank), a ring all-gather moves (world - 1) / world of the total through each link.
"""
import importlib
import os

LAUNCH_CURVE_MS = {1: 1.924, 256: 2.034, 257: 3.026, 512: 3.155}    # profiles/r02_launch_curve.log


def launch_ms(n, lone=256, full=512):
    """measured staircase of the STD128_OPT kernels on one MI355X (same pricing as tests/test_circuit_plaintext.py)"""
    if n <= 0:
        return 0.0
    if n <= lone:
        return 1.93 + 0.10 * n / lone
    rounds, rem = divmod(n, full)
    t = 3.16 * rounds
    if rem == 0:
        return t
    if rounds == 0:
        return 3.03 + 0.13 * (rem - lone) / (full - lone)
    return t + (2.3 if rem <= lone else 3.16)


def predict_gate_sharding(circuit_path, new_flag, K, worlds=(1, 2, 4, 8), ct_bytes=2012, exch_us=60.0, link_GBps=150.0,
                          locality=True, weak=False):
    """weak=False: K input blocks in all, whatever the world (strong scaling).  weak=True: K blocks PER GPU, i.e. K x world in
    lock-step with every step's gates split over the ranks -- the per-GPU load of one GPU alone with K blocks."""
    bce = importlib.import_module(__package__)
    rows = []
    base = None
    K_per = K
    for world in worlds:
        K = K_per * world if weak else K_per
        plans, pubs = [], []
        for rank in range(world):
            c = bce.Circuit()
            c.ReadBristol(circuit_path, new_flag=new_flag)
            c.setInstances(K)
            c.setBalance(True, 256, 512)
            if world > 1:
                c.setShardLocality(locality)
                c.set_exchange(rank, world, 1, lambda nbytes, on_dev: 0, None, None, None, None, 0)
            plans.append(c.relevel_steps())
            pubs.append(c.relevel_publications() if world > 1 else [0] * len(plans[-1]))
            boots = c.info()["n_bootstraps"]
            c.close()
        steps = len(plans[0])
        t_compute = t_exchange = 0.0
        n_exchanges = 0
        crossing = 0
        for s in range(steps):
            t_compute += max(launch_ms(p[s] * K) for p in plans)
            widest = max(p[s] for p in pubs)
            crossing += sum(p[s] for p in pubs)
            if widest:
                n_exchanges += 1
                total_bytes = widest * K * ct_bytes * world
                t_exchange += exch_us / 1e3 + (world - 1) / world * total_bytes / (link_GBps * 1e9) * 1e3
        t = t_compute + t_exchange
        row = {"gpus": world, "ms_per_evaluation": round(t, 1), "gate_bootstraps_per_s": round(boots * K / t * 1e3),
               "compute_ms": round(t_compute, 1), "exchange_ms": round(t_exchange, 1), "exchanges": n_exchanges,
               "crossing_outputs_per_instance": crossing}
        if base is None:
            base = t
        if weak:
            row["instances"] = K
            row["speedup_vs_1"] = round(base / t * world, 2)     # throughput ratio: world x the work in t instead of base
            row["efficiency"] = round(base / t, 3)
        else:
            row["speedup_vs_1"] = round(base / t, 2)
            row["efficiency"] = round(base / t / world, 3)
        rows.append(row)
    return {"circuit": os.path.basename(circuit_path), "instances": ("%d per GPU" % K_per) if weak else K_per, "scaling": "weak" if weak else "strong", "schedule": "bootstrap-depth, steps filled by slack for all ranks together",
            "model": "max over ranks of the measured launch staircase per step + per-step all-gather (%.0f us + bytes over %.0f GB/s); "
                     "PREDICTION, not measured on more than one GPU" % (exch_us, link_GBps),
            "placement": "units follow their inputs' ranks" if locality else "contiguous in netlist order", "rows": rows}
