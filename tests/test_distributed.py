"""N>1 path on CPU: world_size-2 gloo processes exercise the shard plan and the
variable-size hit gather that bench.py uses over RCCL."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clsim_amd.distributed import gather_hits, shard_range


def test_shard_ranges_partition_the_bunch():
    for n in (0, 1, 7, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


def _fake_hits(rank, n):
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    a = rng.integers(0, 256, size=(n, 80), dtype=np.uint8)
    a[:, 0] = rank
    return a


def _worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for trial, cnt in enumerate(counts):
            n = cnt[rank]
            buf = torch.zeros((64, 80), dtype=torch.uint8)
            buf[:n] = torch.from_numpy(_fake_hits(rank + 10 * trial, n))
            got, c = gather_hits(buf, n, dst=0)
            assert c.tolist() == list(cnt)
            if rank == 0:
                exp = np.concatenate([_fake_hits(r + 10 * trial, cnt[r]) for r in range(world)], axis=0)
                assert got.shape == (sum(cnt), 80)
                assert np.array_equal(got.numpy(), exp)
            else:
                assert got is None
        # a counter that ran past the buffer's capacity: the stored rows travel, no hang
        buf = torch.from_numpy(_fake_hits(rank + 100, 64))
        got, c = gather_hits(buf, 64 + 1000 * (rank + 1), dst=0)
        assert c.tolist() == [64] * world
        if rank == 0:
            assert np.array_equal(got.numpy(), np.concatenate([_fake_hits(r + 100, 64) for r in range(world)], axis=0))
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_gather_hits_gloo(world):
    """world 4: the root posts three receives in one batch, like the 7 of an 8-GPU node."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    if world == 2:
        counts = [(5, 9), (0, 3), (7, 0), (0, 0), (64, 64)]      # ragged, empty shards, full buffers
    else:
        counts = [(5, 9, 1, 30), (0, 3, 0, 2), (7, 0, 0, 0), (0, 0, 0, 0), (64, 64, 64, 64)]
    procs = [ctx.Process(target=_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(r, "ok") for r in range(world)], res
