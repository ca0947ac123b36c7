"""Child process of tests/test_comm_fake_rccl.py: runs `world` ranks as THREADS of this process on cuda:0 through
clsimhip_comm_create / clsimhip_gather_hits (clsim_amd/csrc/comm.cpp), with CLSIMHIP_RCCL_LIBRARY pointing at
tests/libfake_rccl.so (set by the parent BEFORE this process starts: the library caches its RCCL handle per process).

usage: comm_fake_rccl_child.py WORLD
Prints one line `case <name> ok` per case and `all ok` at the end; any failure raises."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from clsim_amd.distributed import HitGatherer  # noqa: E402


def records(rank, n, salt):
    """n 80-byte records that name their rank, their position and the case"""
    a = np.zeros((n, 80), dtype=np.uint8)
    a[:, 0] = rank
    a[:, 1] = salt
    a[:, 2:10] = np.arange(n, dtype=np.uint64).view(np.uint8).reshape(n, 8)
    a[:, 10:] = (np.arange(n)[:, None] * 7 + np.arange(70)[None, :] + rank * 13 + salt) & 0xff
    return a


def run_case(world, name, counters, capacity, root, gathered_capacity, salt, expect_error=False, rounds=1):
    assert "fake_rccl" in os.environ.get("CLSIMHIP_RCCL_LIBRARY", ""), "the parent must point CLSIMHIP_RCCL_LIBRARY at the fake library"
    dev = torch.device("cuda:0")
    uid = HitGatherer.unique_id()
    errors = [None] * world
    results = [None] * world
    stats = [None] * world
    barrier = threading.Barrier(world)

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            g = HitGatherer(0, rank, world, uid)
            info = g.info()
            assert info["rccl_ranks"] == world and info["rccl_rank"] == rank and info["device"] == 0, info
            assert len(info["pci_bus_id"]) >= 7 and info["pci_bus_id"].count(":") == 2, info          # "0000:75:00.0"
            stream = torch.cuda.Stream(device=dev)
            stored = min(counters[rank], capacity[rank])
            host = records(rank, capacity[rank], salt)
            with torch.cuda.stream(stream):
                photons = torch.from_numpy(host).to(dev)
                counter = torch.from_numpy(np.array([counters[rank]], dtype=np.uint32).view(np.int32)).to(dev)
                gathered = torch.zeros((max(gathered_capacity, 1), 80), dtype=torch.uint8, device=dev) if rank == root else None
            for _ in range(rounds):
                barrier.wait()
                try:
                    counts = g.gather(photons.data_ptr(), counter.data_ptr(), capacity[rank], root,
                                      gathered.data_ptr() if rank == root else 0, gathered_capacity if rank == root else 0, stream.cuda_stream)
                    raised = None
                except RuntimeError as e:
                    counts, raised = None, str(e)
                stream.synchronize()
                results[rank] = (counts, raised, gathered.cpu().numpy() if rank == root else None, host[:stored])
            st = g.statistics()
            assert st["gathers"] == rounds and st["gather_ms"] > 0.0, st
            stats[rank] = st
            assert g.statistics(reset=True)["gathers"] == rounds and g.statistics()["gathers"] == 0
            g.close()
        except BaseException as e:          # noqa: BLE001 -- reported by the main thread
            errors[rank] = e
            try:
                barrier.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive(), "a rank hangs in case " + name
    for r, e in enumerate(errors):
        if e is not None:
            raise RuntimeError("rank %d failed in case %s: %r" % (r, name, e))
    stored = [min(counters[r], capacity[r]) for r in range(world)]
    room = gathered_capacity
    travels = []
    for r in range(world):
        travels.append(min(stored[r], room))
        room -= travels[-1]
    for r in range(world):                  # the library's own account of what travelled
        assert stats[r]["records_sent"] == (0 if r == root else travels[r] * rounds), (name, r, stats[r], travels)
        assert stats[r]["records_received"] == ((sum(travels) - travels[root]) * rounds if r == root else 0), (name, r, stats[r], travels)
    expect = np.concatenate([results[r][3][:travels[r]] for r in range(world)]) if sum(travels) else np.zeros((0, 80), np.uint8)
    got = results[root][2]
    assert np.array_equal(got[:len(expect)], expect), "root buffer differs from the concatenation in case " + name
    assert not got[len(expect):].any(), "records beyond the gathered ones were touched in case " + name
    for r in range(world):
        counts, raised, _, _ = results[r]
        if expect_error:
            assert raised is not None and "gather buffer too small" in raised, (name, r, raised)
        else:
            assert raised is None, (name, r, raised)
            assert list(counts) == list(counters), (name, r, counts)
    print("case %s ok (world %d, %d records on the root)" % (name, world, len(expect)), flush=True)


def lie_case(world):
    """FAKE_RCCL_LIE_ABOUT_COUNT=1: every rank's communicator reports world - 1 ranks; clsimhip_comm_create must fail on all"""
    uid = HitGatherer.unique_id()
    seen = [None] * world

    def rank_main(rank):
        torch.cuda.set_device(0)
        try:
            HitGatherer(0, rank, world, uid)
            seen[rank] = "created"
        except RuntimeError as e:
            seen[rank] = str(e)
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
        assert not t.is_alive()
    for r in range(world):
        assert "reports rank %d of %d, asked for rank %d of %d" % (r, world - 1, r, world) in seen[r], seen
    print("refused ok", flush=True)


def main():
    world = int(sys.argv[1])
    if len(sys.argv) > 2 and sys.argv[2] == "lie":
        return lie_case(world)
    rng = np.random.default_rng(100 + world)
    lib = C.CDLL(os.environ["CLSIMHIP_RCCL_LIBRARY"])
    cap = [4096] * world
    ragged = [int(x) for x in rng.integers(1, 3000, size=world)]
    run_case(world, "ragged", ragged, cap, 0, sum(ragged), 1, rounds=3)
    run_case(world, "ragged_last_root", ragged, cap, world - 1, sum(ragged) + 17, 2)
    some_empty = list(ragged)
    some_empty[0] = 0
    some_empty[-1] = 0
    if world > 2:
        some_empty[world // 2] = 0
    run_case(world, "empty_ranks", some_empty, cap, 0, sum(some_empty), 3)
    run_case(world, "all_empty", [0] * world, cap, 0, 16, 4)
    # a hit counter that ran past the photon buffer (propagation_kernel.c.cl:329-334): the rank sends what it stored
    over = list(ragged)
    over[world // 2] = 4096 + 777
    over[0] = 5000
    run_case(world, "counter_beyond_capacity", over, cap, 0, sum(min(c, 4096) for c in over), 5)
    # ranks with different photon buffer sizes
    caps = [1024 + 512 * (r % 3) for r in range(world)]
    cnt = [int(x) for x in rng.integers(900, 2600, size=world)]
    run_case(world, "unequal_capacities", cnt, caps, 0, sum(min(c, k) for c, k in zip(cnt, caps)), 6)
    # the root's buffer is too small: every rank reports it, nobody hangs, the root holds the prefix that fits
    total = sum(ragged)
    run_case(world, "gather_buffer_too_small", ragged, cap, 0, total - ragged[-1] // 2 - 1, 7, expect_error=True)
    run_case(world, "gather_buffer_much_too_small", ragged, cap, world - 1, max(1, ragged[0] // 2), 8, expect_error=True)
    run_case(world, "again_after_the_error", ragged, cap, 0, total, 9)
    assert lib.fake_rccl_errors() == 0, "the fake library saw unmatched or mismatched transfers"
    print("all ok", flush=True)


if __name__ == "__main__":
    main()
