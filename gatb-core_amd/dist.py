"""Multi-GPU driver of the DSK hot path: one process per GPU, partitions owned by ranks, one exchange.

The exchange itself lives under the C-ABI (include/gkc.h "Multi-GPU", csrc/gkc_dist.hip): gkc_exchange routes every rank's
super-k-mer records to the owner of their partition (the device analogue of the reference's SuperKmerBinFiles disk shuffle,
tools/storage/impl/Storage.cpp:360-430), gkc_bloom_allreduce_or / gkc_mphf_build_solid_dist combine the per-rank Bloom filters and
MPHF levels. This module only provides what a Python launcher adds on top:

  * make_comm(): the communicator of this rank. Under torch.distributed with backend "nccl" (= RCCL on ROCm) rank 0 creates the
    ncclUniqueId and broadcasts it, and every rank opens its OWN RCCL communicator inside libgkc_hip.so (grouped ncclSend / ncclRecv
    over xGMI on the library's stream); with backend "gloo" — two ranks sharing one GPU in the tests, where RCCL refuses duplicate
    devices, or a box without xGMI — a host-staged transport (device -> pinned host -> gloo -> device) implements the two callbacks
    of gkc_transport;
  * DistributedCounter: per pass begin -> [push -> exchange]* -> finish, and the owner ranges afterwards;
  * owner_ranges / exchange_buckets: the host-side twins the CPU tests drive (gkc_exchange_plan is the library's planning function).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import gkc


class DevArray:
    """zero-copy view of library-owned device memory for torch (``torch.as_tensor(DevArray(ptr, n), device='cuda')``)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def owner_ranges(nb_partitions, world):
    """equal-count contiguous ranges [lo, hi) per rank (what gkc_balanced_owner_ranges gives for equal weights)"""
    first = gkc.balanced_owner_ranges(np.ones(nb_partitions, np.uint64), world)
    return [(int(first[r]), int(first[r + 1])) for r in range(world)]


class HostStagedTransport:
    """gkc_transport over a torch.distributed process group with CPU tensors (gloo): device buffers are staged through host memory.
    Used where RCCL cannot connect the ranks (two ranks on one GPU; no xGMI / no RCCL backend)."""

    def __init__(self, counter, group=None):
        self.c, self.group = counter, group
        self.world = dist.get_world_size(group); self.rank = dist.get_rank(group)

    def allgather_host(self, mine):
        t = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(outs, t, group=self.group)
        return [bytes(o.numpy().tobytes()) for o in outs]

    def sendrecv_device(self, sends, recvs):
        reqs, keep, landing = [], [], []
        for peer, ptr, n in recvs:                       # post the receives first, then the sends (all non-blocking), then wait
            t = torch.empty(int(n), dtype=torch.uint8)
            reqs.append(dist.irecv(t, src=peer, group=self.group)); landing.append((ptr, t))
        for peer, ptr, n in sends:
            t = torch.from_numpy(self.c.device_to_host(ptr, int(n)))
            keep.append(t); reqs.append(dist.isend(t, dst=peer, group=self.group))
        for r in reqs:
            r.wait()
        for ptr, t in landing:
            self.c.host_to_device(ptr, t.numpy())


LAST_COMM_KIND = None       # "rccl" | "host-staged" | "device-to-device IPC (...)" | "host-staged (fallback: <why>)": what make_comm gave this process last (bench.py reports it)


def _staged_comm(counter, group, world, rank, ipc):
    """transport communicator over a gloo group; ipc: its device messages go device to device through IPC memory handles (gkc_comm_enable_ipc) — checked by a small
    exchange over the real peers before it is trusted; if any rank fails, all of them go back to staging through the host. Returns (comm, used_ipc)."""
    comm = gkc.Comm.transport(counter, HostStagedTransport(counter, group), world, rank)
    if not ipc or world < 2:
        return comm, False
    why = None
    try:
        comm.enable_ipc(True)
        bad, _ = comm.selftest(1 << 20)
        if bad:
            why = "%d wrong words" % bad
    except Exception as e:      # noqa
        why = str(e)[:200]
    votes = [None] * world
    dist.all_gather_object(votes, why, group=group)
    if any(votes):
        comm.enable_ipc(False)
        return comm, False
    return comm, True


def make_comm(counter, group=None, try_rccl=None):
    """communicator of this rank for ``counter`` (see module docstring). Without an initialised process group: a one-rank RCCL communicator.

    Backend "nccl": every rank opens its RCCL communicator inside the library; the ranks then tell each other whether that worked, and if it failed on ANY of
    them (RCCL inside libgkc_hip.so has never met real peers before the first multi-GPU run: a refusal there must not cost the run) ALL of them switch to
    the host-staged transport over a gloo group created for the purpose — slower by the PCIe round trip, same results, and LAST_COMM_KIND says so.
    ``try_rccl=True`` makes a gloo-backed group attempt RCCL first as well (the dry run of this very fallback: two ranks on one device are refused by RCCL)."""
    global LAST_COMM_KIND
    if not dist.is_initialized():
        LAST_COMM_KIND = "rccl"
        return gkc.Comm.rccl(counter, gkc.Comm.unique_id(), 1, 0)
    world = dist.get_world_size(group); rank = dist.get_rank(group)
    is_nccl = dist.get_backend(group) == "nccl"
    if is_nccl or try_rccl:
        box = [gkc.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        comm, why = None, None
        try:
            comm = gkc.Comm.rccl(counter, box[0], world, rank)
        except Exception as e:      # noqa  (gkc.GkcError: the library's message names the RCCL call that failed)
            why = "rank %d: %s" % (rank, str(e)[:300])
        votes = [None] * world
        dist.all_gather_object(votes, why, group=group)
        bad = [v for v in votes if v]
        if not bad:
            LAST_COMM_KIND = "rccl"
            return comm
        # (a communicator that did open on this rank is left alone: destroying half of a broken clique may hang)
        staged_group = dist.new_group(backend="gloo") if is_nccl else group
        comm, used = _staged_comm(counter, staged_group, world, rank, ipc=os.environ.get("GKC_NO_IPC") is None)
        LAST_COMM_KIND = "%s (fallback: RCCL communicator refused on %d of %d ranks; %s)" % ("device-to-device IPC copies, host all-gathers over gloo" if used else "host-staged", len(bad), world, bad[0])
        return comm
    comm, used = _staged_comm(counter, group, world, rank, ipc=os.environ.get("GKC_IPC") == "1")
    LAST_COMM_KIND = "device-to-device IPC copies, host all-gathers over gloo" if used else "host-staged"
    return comm


class DistributedCounter:
    """Host-side driver of one rank. ``counter`` is a gkc.Counter configured with the SAME model / repartition table on every rank.
    Per pass: counter.begin_pass(p); [counter.push_reads*(...); self.exchange()]*; counter.finish_pass(). The number of pushes may differ
    between ranks as long as every rank calls exchange() the same number of times (a call without new pushes sends nothing)."""

    def __init__(self, counter, rank, world, nb_partitions, group=None, comm=None, owners=None, try_rccl=None):
        self.c, self.rank, self.world, self.P, self.group = counter, rank, world, nb_partitions, group
        self.comm = comm if comm is not None else make_comm(counter, group, try_rccl=try_rccl)
        if owners is not None:
            self.comm.set_owners(owners)

    def exchange(self):
        self.c.exchange(self.comm)

    def owners(self):
        return self.comm.owners(self.world)

    def owned(self):
        f = self.owners()
        return range(int(f[self.rank]), int(f[self.rank + 1]))

    def stats(self):
        return self.comm.stats()


def exchange_buckets(send, rec_counts, first, rank, world, record_bytes, group=None):
    """Host twin of gkc_exchange for ONE segment per rank, on CPU tensors (the gloo tests): all-gathers the per-partition record counts, asks
    the library's gkc_exchange_plan for the messages and moves the bytes. Returns (recv uint8 tensor, plan recvs, counts[world][P])."""
    P = len(rec_counts)
    mine = torch.as_tensor(np.asarray(rec_counts, dtype=np.int64))
    outs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(outs, mine, group=group)
    cnt = np.stack([o.numpy() for o in outs]).astype(np.uint64)                      # [world][P]
    counts = np.zeros((world, 1, 2, P), np.uint64); counts[:, 0, 0, :] = cnt
    sends, recvs, total = gkc.exchange_plan(world, rank, first, np.ones(world, np.uint64), counts)
    recv = torch.empty(int(total) * record_bytes, dtype=torch.uint8)
    reqs = []
    for peer, seg, beg, n in recvs:
        reqs.append(dist.irecv(recv[int(beg) * record_bytes:int(beg + n) * record_bytes], src=peer, group=group))
    for peer, seg, beg, n in sends:
        reqs.append(dist.isend(send[int(beg) * record_bytes:int(beg + n) * record_bytes].clone(), dst=peer, group=group))
    for r in reqs:
        r.wait()
    return recv, recvs, cnt
