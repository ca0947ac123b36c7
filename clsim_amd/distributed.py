"""Multi-GPU plumbing: one process per GPU over torch.distributed (RCCL on ROCm).

The propagation path has no exchange step: steps are independent units and the
RNG streams belong to step slots, so a bunch is sharded into contiguous step
ranges, every rank propagates its own range with its own streams, and only the
detected photons travel: one variable-size gather to rank 0.  The reference's
counterpart is independent converters behind a ZeroMQ router
(private/clsim/I3CLSimServer.cxx:77-137) -- no collective at all.

xGMI is point-to-point (7 links per GPU), so the gather is an all_gather of the
counts followed by one direct send per peer to the root, posted as ONE batch
(ncclGroupStart/End under torch's batch_isend_irecv) so that the 7 transfers run
on their 7 links at once; a ring would only add hops.  The same code runs on CPU
tensors over gloo (tests/test_distributed.py).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous step range [lo, hi) of `rank`: sizes differ by at most one."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_hits(photons, count, dst=0, out=None, group=None):
    """Gathers the first `count` rows of each rank's `photons` ([capacity, 80] uint8)
    on rank `dst`.  Returns (gathered[:total], counts) on dst and (None, counts)
    elsewhere; `counts` is a CPU int64 tensor with one entry per rank.  `out` may
    provide a preallocated [>= total, 80] buffer on dst."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.as_tensor([int(count)], dtype=torch.int64, device=photons.device)
    counts = torch.zeros(world, dtype=torch.int64, device=photons.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    c = counts.cpu()
    total = int(c.sum())
    if rank == dst:
        if out is None or out.shape[0] < total:
            out = torch.empty((total, photons.shape[1]), dtype=photons.dtype, device=photons.device)
        off = 0
        ops = []
        for peer in range(world):
            k = int(c[peer])
            if peer == dst:
                if k:
                    out[off:off + k].copy_(photons[:k])
            elif k:
                ops.append(dist.P2POp(dist.irecv, out[off:off + k], peer, group))
            off += k
        for r in (dist.batch_isend_irecv(ops) if ops else []):
            r.wait()
        return out[:total], c
    if int(c[rank]):
        for r in dist.batch_isend_irecv([dist.P2POp(dist.isend, photons[:int(c[rank])].contiguous(), dst, group)]):
            r.wait()
    return None, c
