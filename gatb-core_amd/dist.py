"""Multi-GPU sharding of the DSK hot path: one process per GPU, partitions owned by ranks, one exchange.

The reference has no distributed path: its inter-stage exchange is the disk shuffle of SuperKmerBinFiles (reference
tools/storage/impl/Storage.cpp:360-430: every thread appends super-k-mers to one file per partition, the counting stage
reads the files back). Here the same hand-over is ONE all-to-all of the device super-k-mer buckets over xGMI
(torch.distributed, backend "nccl" == RCCL on ROCm; "gloo" on CPU for the tests):

  * every rank scans its own slice of the reads (Stage A) into per-partition buckets (gkc_push_reads*);
  * partition p is owned by rank p // (P / world)  (contiguous ranges, so the bytes for one destination are ONE contiguous
    slice of the bucket arena — nothing is packed or copied before the send);
  * a small all-gather of the per-partition record / k-mer counts, then the all-to-all of the arena bytes (one batch of
    point-to-point messages below 2 GiB each — larger single transfers are corrupted by this RCCL/torch stack);
  * every rank imports the chunks it received as foreign segments (gkc_segment_import) and counts the partitions it
    owns (Stage B, gkc_finish_pass). Results stay sharded by partition; no further collective.

Volume: ~1.4 B per k-mer x (world-1)/world. xGMI is point-to-point (7 links x ~153 GB/s per GPU) and an all-to-all uses all
links at once, unlike a ring.
"""
import numpy as np
import torch
import torch.distributed as dist


class DevArray:
    """zero-copy view of library-owned device memory for torch (``torch.as_tensor(DevArray(...), device='cuda')``)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def owner_ranges(nb_partitions, world):
    """partition ranges [lo, hi) per rank; nb_partitions must be a multiple of world (ConfigurationAlgorithm.cpp:423-425
    rounds the partition count the same way for its parallel batches)"""
    if nb_partitions % world:
        raise ValueError("nb_partitions (%d) must be a multiple of the world size (%d)" % (nb_partitions, world))
    per = nb_partitions // world
    return [(r * per, (r + 1) * per) for r in range(world)]


CHUNK_BYTES = 1 << 30      # per (source, destination) message: RCCL / torch silently corrupt transfers of 2 GiB and more (measured on
                           # this stack: all_to_all_single of >= 2^31 bytes per peer returns wrong data), so every message stays below


def exchange_buckets(send, rec_off, kmers, record_bytes, rank, world, group=None, chunk_bytes=None):
    """Routes bucket bytes to the owners.

    send      uint8 tensor: this rank's bucket arena (partition-major), on the device of the process group's backend
    rec_off   int64[P+1] record offsets of the partitions inside ``send``;  kmers int64[P] k-mers per partition
    Returns (recv uint8 tensor, list over source ranks of (byte offset into recv, rec_off table int64[P+1] relative to that
    offset, kmers table int64[P])).
    """
    P = len(kmers)
    ranges = owner_ranges(P, world)
    dev = send.device
    # 1) everybody learns everybody's per-partition counts (2*P int64 per rank)
    mine = torch.cat([torch.as_tensor(np.diff(np.asarray(rec_off, dtype=np.int64))), torch.as_tensor(np.asarray(kmers, dtype=np.int64))]).to(dev)
    allc = torch.empty(world * 2 * P, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    allc = allc.cpu().numpy().reshape(world, 2, P)
    rec_cnt, km_cnt = allc[:, 0, :], allc[:, 1, :]
    # 2) the bytes: destination j gets my records of its partitions = one contiguous slice of the arena
    in_split = [int(rec_cnt[rank, lo:hi].sum()) * record_bytes for lo, hi in ranges]
    lo, hi = ranges[rank]
    out_split = [int(rec_cnt[s, lo:hi].sum()) * record_bytes for s in range(world)]
    recv = torch.empty(sum(out_split), dtype=torch.uint8, device=dev)
    assert sum(in_split) == send.numel(), (sum(in_split), send.numel())
    # all-to-all as one batch of point-to-point messages (what RCCL's all-to-all is underneath), chunked below 2 GiB;
    # the local slice is a plain device copy
    ch = int(chunk_bytes or CHUNK_BYTES) // record_bytes * record_bytes
    in_off = np.concatenate([[0], np.cumsum(in_split)]); out_off = np.concatenate([[0], np.cumsum(out_split)])
    ops = []
    for peer in range(world):
        s_sl = send[int(in_off[peer]):int(in_off[peer + 1])]
        r_sl = recv[int(out_off[peer]):int(out_off[peer + 1])]
        if peer == rank:
            r_sl.copy_(s_sl)
            continue
        for c0 in range(0, s_sl.numel(), ch):
            ops.append(dist.P2POp(dist.isend, s_sl[c0:c0 + ch], peer, group))
        for c0 in range(0, r_sl.numel(), ch):
            ops.append(dist.P2POp(dist.irecv, r_sl[c0:c0 + ch], peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    # 3) per-source tables for gkc_segment_import
    chunks = []
    pos = 0
    for s in range(world):
        ro = np.zeros(P + 1, dtype=np.int64)
        ro[lo + 1:hi + 1] = np.cumsum(rec_cnt[s, lo:hi])
        ro[hi + 1:] = ro[hi]
        km = np.zeros(P, dtype=np.int64)
        km[lo:hi] = km_cnt[s, lo:hi]
        chunks.append((pos, ro, km))
        pos += out_split[s]
    return recv, chunks


def allreduce_or(t, group=None, slice_bytes=256 << 20):
    """in-place bitwise-OR all-reduce of a uint8 tensor (Bloom bit arrays: every rank inserted the solid k-mers of the partitions it owns;
    the OR of the partial filters is the filter of the whole set, bit for bit — SURVEY.md §8e). torch's NCCL/RCCL backend has no bitwise
    reduce op, so every slice is all-gathered (world x slice_bytes of scratch) and ORed locally."""
    world = dist.get_world_size(group)
    flat = t.view(-1)
    if world == 1:
        return t
    step = max(1, int(slice_bytes) // max(1, t.element_size()))
    tmp = torch.empty((world, min(step, flat.numel())), dtype=flat.dtype, device=flat.device)
    for i in range(0, flat.numel(), step):
        part = flat[i:i + step]
        buf = tmp[:, :part.numel()].contiguous() if part.numel() != tmp.shape[1] else tmp
        dist.all_gather_into_tensor(buf.view(-1), part.contiguous(), group=group)
        acc = buf[0]
        for r in range(1, world):
            acc = acc | buf[r]
        part.copy_(acc)
    return t


def allreduce_or_bloom(bloom, group=None):
    """OR-reduce a gkc.Bloom across the ranks, in place on the device"""
    ptr, nbytes = bloom.device_array()
    t = torch.as_tensor(DevArray(ptr, nbytes), device="cuda")
    allreduce_or(t, group)
    torch.cuda.synchronize()
    return bloom


class DistributedCounter:
    """Host-side driver of one rank. ``counter`` is a gkc.Counter already configured with the SAME model / repartition
    table on every rank. Call ``exchange()`` between the pushes and ``finish_pass()`` of every pass."""

    def __init__(self, counter, rank, world, nb_partitions, group=None):
        self.c, self.rank, self.world, self.P, self.group = counter, rank, world, nb_partitions, group
        self.ranges = owner_ranges(nb_partitions, world)
        self._keep = []            # received buffers must outlive gkc_finish_pass

    def owned(self):
        return range(*self.ranges[self.rank])

    def exchange(self):
        c = self.c
        nseg = c.segment_count()
        self._keep = []
        recvs = []
        for s in range(nseg):
            ptr, rb, off, km = c.segment_export(s)
            nbytes = int(off[-1]) * rb
            send = torch.as_tensor(DevArray(ptr, nbytes), device="cuda") if nbytes else torch.empty(0, dtype=torch.uint8, device="cuda")
            recv, chunks = exchange_buckets(send, off.astype(np.int64), km.astype(np.int64), rb, self.rank, self.world, self.group)
            recvs.append((recv, chunks, rb))
        torch.cuda.synchronize()   # the sends read the context-owned arenas: finish before they are released
        c.segments_clear()
        for recv, chunks, rb in recvs:
            self._keep.append(recv)
            base = recv.data_ptr()
            for pos, ro, km in chunks:
                if ro[-1]:
                    c.segment_import(base + pos, ro.astype(np.uint64), km.astype(np.uint64))
