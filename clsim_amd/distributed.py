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

The product path is `HitGatherer`: the same gather behind the C ABI (clsimhip_comm_create / clsimhip_gather_hits in
include/clsimhip.h, RCCL called directly from C++), which is what a C++/IceTray host uses -- it has no torch.
`gather_hits` below is the same plan on torch tensors; it carries the N>1 logic tests over gloo on CPU.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


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
    # the kernel's hit counter keeps counting past the buffer's capacity (propagation_kernel.c.cl:329-334): a rank
    # sends what it stored, and announces that number, so the root's receives match the sends
    count = min(int(count), int(photons.shape[0]))
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


class HitGatherer:
    """RCCL gather of detected photons through the C ABI (one instance per rank / GPU)."""

    def __init__(self, device, rank, world, unique_id):
        self._lib = _lib.load()
        self.rank, self.world = int(rank), int(world)
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        rc = self._lib.clsimhip_comm_create(int(device), self.rank, self.world, C.cast(buf, C.c_void_p), C.byref(self._h))
        if rc != 0:
            raise RuntimeError((self._lib.clsimhip_last_error(None) or b"").decode())

    @staticmethod
    def unique_id():
        lib = _lib.load()
        buf = (C.c_uint8 * 128)()
        if lib.clsimhip_comm_get_unique_id(C.cast(buf, C.c_void_p)) != 0:
            raise RuntimeError((lib.clsimhip_last_error(None) or b"").decode())
        return bytes(buf)

    @classmethod
    def from_process_group(cls, device, group=None):
        """Rank 0 creates the RCCL unique id, torch.distributed carries it to the other ranks."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        return cls(device, rank, world, box[0])

    def gather(self, d_photons, d_hit_count, capacity, root=0, d_gathered=0, gathered_capacity=0, stream=0):
        """Device pointers (ints).  Returns the ranks' hit counters (numpy uint64); the transfers may still be running
        on `stream` when this returns."""
        counts = np.zeros(self.world, dtype=np.uint64)
        rc = self._lib.clsimhip_gather_hits(self._h, C.c_void_p(d_photons), C.c_void_p(d_hit_count), int(capacity), int(root),
                                            C.c_void_p(d_gathered), int(gathered_capacity), counts.ctypes.data_as(C.c_void_p), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError((self._lib.clsimhip_last_error(None) or b"").decode())
        return counts

    def info(self):
        """What the COMMUNICATOR reports (ncclCommCount / ncclCommUserRank), its HIP device and that device's PCI bus id."""
        ranks, rank, device = C.c_int(0), C.c_int(-1), C.c_int(-1)
        bus = C.create_string_buffer(32)
        rc = self._lib.clsimhip_comm_info(self._h, C.byref(ranks), C.byref(rank), C.byref(device), bus, 32)
        if rc != 0:
            raise RuntimeError((self._lib.clsimhip_last_error(None) or b"").decode())
        return {"rccl_ranks": ranks.value, "rccl_rank": rank.value, "device": device.value, "pci_bus_id": bus.value.decode()}

    def statistics(self, reset=False):
        """Gathers issued so far (waits for them): count, milliseconds on their stream, records sent / received."""
        n, sent, received, ms = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_double(0.0)
        rc = self._lib.clsimhip_comm_statistics(self._h, C.byref(n), C.byref(ms), C.byref(sent), C.byref(received), 1 if reset else 0)
        if rc != 0:
            raise RuntimeError((self._lib.clsimhip_last_error(None) or b"").decode())
        return {"gathers": n.value, "gather_ms": ms.value, "records_sent": sent.value, "records_received": received.value}

    def close(self):
        if self._h:
            self._lib.clsimhip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
