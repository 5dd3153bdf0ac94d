"""bench.py starts its own ranks: `python3 bench.py --gpus N` (how the driver invokes it) must spawn N child
processes, rendezvous them on 127.0.0.1 and fail loudly -- not hang, not silently fall back -- when something is
missing.  The CPU test runs without a GPU (the ranks meet over gloo, then every rank reports that the engine
needs a GPU and the parent exits non-zero); the GPU test rehearses the whole N = 2 path on the one GPU of the
test box (both ranks on device 0, gloo carrying the collectives), including the secondary gate-sharded run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(args, timeout, **extra):
    env = dict(os.environ, BCE_BENCH_SINGLE_DEVICE="1", BCE_BENCH_BACKEND="gloo", **extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus2_spawns_ranks_and_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked rehearsal below")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], timeout=300)
    assert p.returncode != 0
    assert "rank 0/2 rendezvous ok" in p.stderr and "rank 1/2 rendezvous ok" in p.stderr, p.stderr[-2000:]
    assert p.stderr.count("needs a GPU") == 2, p.stderr[-2000:]
    assert "ranks failed" in p.stderr and not any(l.startswith("{") for l in p.stdout.splitlines())


def test_bench_rejects_mismatched_world_size():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "does not match" in p.stderr


@pytest.mark.gpu
def test_bench_gpus2_rehearsal_on_one_gpu():
    """N > 1: the line's `value` is north_star's partition (every step's gates split over the ranks, K x N blocks in
    lock-step), with the ranks the collective spans counted by the collective, one device entry per rank, per-rank step
    times; the collective-free replica run and the strong-scaling leg ride along."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--circuit", "adder_32bit.txt", "--instances", "4",
              "--gates-steps", "1", "--replica-steps", "1"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["outputs_verified"] is True and "error" not in d
    assert d["config"]["sharding"].startswith("gates") and d["config"]["instances_total"] == 8 and d["config"]["instances_per_gpu"] == 4
    assert d["config"]["bootstraps_per_step"] == 8 * 310               # adder_32bit: 310 bootstraps, K x N = 8 blocks, summed over the ranks
    assert d["config"]["exchanges_per_step"] > 0 and d["config"]["exchanged_cts_per_step"] > 0
    assert d["config"]["ciphertext_identity"]["identical_to_single_rank_evaluation"] is True
    assert len(d["config"]["per_rank_ms_per_step"]) == 2 and all(x > 0 for x in d["config"]["per_rank_ms_per_step"])
    r = d["rccl"]
    assert r["rccl_ranks"] == 2 and len(r["devices"]) == 2 and [x["rank"] for x in r["devices"]] == [0, 1]
    assert "gloo" in r["backend"] and "callback" in r["exchange_path"]  # the rehearsal's transport, named as such
    rep = d["replicas"]
    assert rep["outputs_verified"] is True and rep["value"] > 0 and rep["scaling"] == "weak"
    g = d["shard_gates"]
    assert g["outputs_verified"] is True and g["scaling"] == "strong"
    assert g["exchanges_per_step"] > 0 and 0 < g["exchanged_cts_per_step"] < d["config"]["exchanged_cts_per_step"]
    assert g["ciphertext_identity"]["identical_to_single_rank_evaluation"] is True
    assert "roofline" in d and "cpu_baseline" not in d               # the CPU baseline is an N = 1 leg


@pytest.mark.gpu
def test_bench_replicas_can_still_be_the_headline():
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--circuit", "adder_32bit.txt", "--instances", "4",
              "--shard", "instances", "--gates-steps", "0"], timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert d["config"]["sharding"].startswith("instances") and d["config"]["bootstraps_per_step"] == 2 * 4 * 310
    assert d["config"]["exchanges_per_step"] == 0 and d["rccl"]["rccl_ranks"] == 2 and "replicas" not in d


@pytest.mark.gpu
def test_bench_headline_survives_a_hung_secondary_run():
    """A rank that never reaches the strong-scaling leg's collectives (simulated) must not cost the headline: the watchdog
    prints the one JSON line with it and ends every rank -- with a NON-ZERO exit code, because a process that touched the
    GPU and hung in a collective is a failed run (the parent reports `ranks failed`)."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--circuit", "adder_32bit.txt", "--instances", "4",
              "--gates-steps", "1", "--replica-steps", "1", "--gates-timeout", "40"], timeout=600, BCE_BENCH_TEST_HANG="1")
    assert p.returncode != 0, p.stderr[-3000:]
    assert "ranks failed" in p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["config"]["outputs_verified"] is True and d["value"] > 0
    assert d["config"]["sharding"].startswith("gates") and "did not finish" in d["shard_gates"]["error"]
    assert "watchdog fired" in p.stderr


@pytest.mark.gpu
def test_bench_falls_back_to_the_replica_line_when_the_gate_sharded_headline_hangs():
    """If the gate-sharded headline itself never returns, the line that is printed is the replica run measured before it,
    labelled as a fallback with an error field; the exit code is non-zero."""
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--circuit", "adder_32bit.txt", "--instances", "4",
              "--gates-steps", "1", "--replica-steps", "1", "--gates-timeout", "25"], timeout=600, BCE_BENCH_TEST_HANG="head")
    assert p.returncode != 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, p.stdout
    d = json.loads(line[0])
    # (the watchdog's message, or -- when the hung rank's own watchdog ended it first -- the collective's error on rank 0)
    assert d["headline_fallback"] == "replicas" and ("unavailable" in d["error"] or "failed" in d["error"]) and d["value"] > 0
    assert d["config"]["sharding"].startswith("instances") and d["config"]["bootstraps_per_step"] == 2 * 4 * 310
