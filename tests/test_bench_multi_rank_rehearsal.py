"""bench.py's --gpus N path, executed for real on a box with ONE GPU.

The driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` on a whole node at round
end; nothing in this repository can start that.  This test starts the very same command with N = 2 and 4 processes that
time-share the one GPU (CLSIMHIP_BENCH_REHEARSAL=1: every rank on device 0, torch.distributed over gloo) and with
tests/libfake_rccl.so in its process mode standing in for librccl (the real one refuses two ranks on one device).  What
it proves: the sharding into bunches per rank, clsimhip_comm_create / clsimhip_gather_hits between PROCESSES, the
fallback agreement, the max-over-ranks clock and the JSON line -- and, with --verify-gather, that rank 0 holds exactly the
records the ranks produced.  The rate it prints is not a measurement and the line says so (`value` null).
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "libfake_rccl.so")


def _run(world, workload, shard_steps, port, launcher="torchrun"):
    scratch = tempfile.mkdtemp(prefix="fake_rccl_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    env = dict(os.environ, CLSIMHIP_BENCH_REHEARSAL="1", CLSIMHIP_RCCL_LIBRARY=FAKE, FAKE_RCCL_DIR=scratch,
               MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    front = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
             "--master-port", str(port)] if launcher == "torchrun" else [sys.executable]     # "self": bench.py starts its own ranks
    if launcher == "self":
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
    cmd = front + [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "1",
                   "--workload", workload] + (["--shard-steps", str(shard_steps)] if shard_steps else [])      # (gather verification: default at N>1)
    try:
        p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)
    assert p.returncode == 0, "bench.py --gpus %d failed (%d):\n%s\n%s" % (world, p.returncode, p.stdout[-3000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world,workload,shard,port,launcher", [
    (2, "c2", 700000, 29711, "torchrun"), (4, "c2", 300000, 29713, "torchrun"), (2, "c5", 100000, 29715, "torchrun"),
    (2, "c2", None, 29717, "torchrun"),      # None: the driver's own command, C4's shard of 12.5M steps per rank
    (2, "c2", 700000, None, "self"),         # plain `python bench.py --gpus 2`: bench.py is its own launcher
    (4, "c5", 100000, None, "self")])
def test_bench_multi_rank_path_on_one_gpu(world, workload, shard, port, launcher):
    from tests.test_comm_fake_rccl import build_fake
    assert build_fake() == FAKE
    out = _run(world, workload, shard, port, launcher)
    assert "single_gpu_rate_same_shard" in out["config"]
    assert out["n_gpus"] == world and out["steps"] == 1
    assert out["value"] is None and "REHEARSAL" in out["rehearsal"]
    cfg = out["config"]
    assert cfg["hit_gather"].startswith("clsimhip_gather_hits"), cfg["hit_gather"]          # not the torch.distributed fallback
    assert cfg["gather_verified"] is True                    # without --verify-gather: it is the default at N > 1
    # who took part, as the library's communicators and devices report it
    assert cfg["rccl_ranks"] == world and cfg["control_group"]["world_size"] == world
    assert len(cfg["rank_devices"]) == world and all(d and d.count(":") == 2 for d in cfg["rank_devices"])
    assert cfg["distinct_devices"] == 1                      # the rehearsal's ranks time-share one GPU, and the line shows it
    assert cfg["n_ranks_on_n_devices"] is False and "1 distinct devices" in cfg["n_ranks_on_n_devices_why_not"]
    per = cfg["per_rank"]
    for key in ("kernel_ms", "gather_ms", "seconds"):
        assert per[key]["min"] > 0 and per[key]["max"] >= per[key]["min"] and 0 <= per[key]["argmax_rank"] < world, (key, per[key])
    assert per["gathers"]["min"] == per["gathers"]["max"] == cfg["bunches_per_pass"]
    assert len(set(per["pids"])) == world
    assert sum(per["hits_stored_timed_region"]) == cfg["hits_gathered_per_pass"] and min(per["hits_stored_timed_region"]) > 0
    assert per["records_sent"][0] == 0 and sum(per["records_sent"]) == per["records_received_by_root"] > 0
    assert "multi_gpu_evidence" in out
    rf = out["roofline"]
    for key in ("valu_useful_frac", "valu_issue_slot_frac", "valu_lane_utilisation", "valu_overhead_ratio", "traffic_ratio", "pmc_is_stored"):
        assert key in rf, key
    if shard is None:
        shard = 12500000
        assert cfg["bunches_per_pass"] == 3 and "C4 = BASELINE configs[3]" in cfg["workload"]
    assert cfg["steps_per_gpu"] == shard
    assert cfg["bunches_per_pass"] * cfg["steps_per_bunch"] >= shard
    assert cfg["overflowed_buffers"] == 0
    assert cfg["hits_gathered_per_pass"] > cfg["hits_last_pass_rank0"] > 0                  # more than rank 0's own photons arrived
    assert cfg["photons_per_pass_all_gpus"] == world * shard * cfg["photons_per_step"]


def _plain(argv, **env_changes):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_changes)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)


def test_gpus_and_world_size_must_agree():
    """`--gpus` is never ignored: a launcher's world that differs from it is refused before anything else happens."""
    p = _plain(["--gpus", "8"], RANK="0", WORLD_SIZE="1")
    assert p.returncode == 2 and "must agree" in p.stderr and "{" not in p.stdout
    p = _plain(["--gpus", "1"], RANK="0", WORLD_SIZE="4")
    assert p.returncode == 2 and "must agree" in p.stderr
    p = _plain(["--gpus", "2"], WORLD_SIZE="4")
    assert p.returncode == 2 and "refusing to guess" in p.stderr
    p = _plain(["--gpus", "0"])
    assert p.returncode != 0


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="the no-GPU behaviour of the launcher")
def test_plain_gpus_n_starts_n_ranks_of_its_own():
    """Without a GPU every rank stops at "bench.py needs a GPU": the launcher must have started --gpus ranks, each with its own
    RANK, and must hand back their failure instead of printing a one-GPU line."""
    p = _plain(["--gpus", "3", "--no-cpu-baseline"], CLSIMHIP_BENCH_ECHO_RANK="1")
    assert p.returncode != 0 and "{" not in p.stdout
    for r in range(3):
        assert "bench.py: rank %d of 3" % r in p.stderr, p.stderr[-2000:]
        assert "bench.py: rank %d exited with" % r in p.stderr
    assert p.stderr.count("needs a GPU") == 3


def test_an_n_gpu_line_needs_n_ranks_on_n_devices():
    """bench.py: check_world() -- what the ranks reported about themselves decides whether an N > 1 line is printed at all (exit code 4
    otherwise): N records, N distinct PCI bus ids, ranks 0..N-1, and every communicator counting N ranks with the process's own rank."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    def world(n, **change):
        recs = [dict(rank=r, pci_bus_id="0000:%02x:00.0" % (0x10 + r), rccl_ranks=n, rccl_rank=r) for r in range(n)]
        for r, fields in change.items():
            recs[int(r[1:])].update(fields)
        return recs
    assert bench.check_world(world(8), 8, True) == (True, "")
    assert bench.check_world(world(2), 2, False) == (True, "")
    ok, why = bench.check_world(world(4, r3=dict(pci_bus_id="0000:10:00.0")), 4, True)        # two ranks on one GPU
    assert not ok and "3 distinct devices" in why
    ok, why = bench.check_world(world(4, r2=dict(pci_bus_id=None)), 4, True)
    assert not ok and "could not name its device" in why
    ok, why = bench.check_world(world(4, r1=dict(rccl_ranks=1)), 4, True)                      # a communicator of its own
    assert not ok and "count" in why
    assert bench.check_world(world(4, r1=dict(rccl_ranks=1)), 4, False)[0]                     # (the torch.distributed fallback has no library communicator)
    ok, why = bench.check_world(world(4, r1=dict(rccl_rank=2), r2=dict(rccl_rank=1)), 4, True)
    assert not ok and "differs" in why
    ok, why = bench.check_world(world(4)[:3], 4, True)
    assert not ok and "3 ranks reported" in why
    ok, why = bench.check_world(world(2, r1=dict(rank=0)), 2, False)
    assert not ok and "not 0..1" in why
