"""N>1 path on CPU: world_size 2, gloo backend. Exercises exactly the exchange code the GPU ranks run over RCCL
(gatb-core_amd/dist.py:exchange_buckets — all_gather of counts + all_to_all_single of bucket bytes + import tables)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _make_records(rank, P, rb, seed):
    """tagged fake records: every 16/32-byte record carries (source rank, partition, index) so routing can be checked"""
    rng = np.random.default_rng(seed + rank)
    counts = rng.integers(0, 40, P)
    counts[rng.integers(0, P)] = 0
    rec_off = np.zeros(P + 1, np.int64); rec_off[1:] = np.cumsum(counts)
    recs = np.zeros((int(rec_off[-1]), rb // 8), np.uint64)
    for p in range(P):
        for i in range(counts[p]):
            recs[rec_off[p] + i, 0] = (rank << 48) | (p << 24) | i
            recs[rec_off[p] + i, 1] = 0xABCD0000 + p
    kmers = counts * 7 + np.arange(P)
    kmers[counts == 0] = 0
    return recs, rec_off, kmers.astype(np.int64)


def _worker(rank, world, port, P, rb, q, chunk=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gdist = ge.load()
        from gatb_core_amd import dist as gd
        recs, rec_off, kmers = _make_records(rank, P, rb, 100)
        send = torch.from_numpy(recs.view(np.uint8).reshape(-1).copy())
        recv, chunks = gd.exchange_buckets(send, rec_off, kmers, rb, rank, world, chunk_bytes=chunk)
        lo, hi = gd.owner_ranges(P, world)[rank]
        got = recv.numpy().view(np.uint64).reshape(-1, rb // 8)
        ok = True
        total = 0
        for s, (pos, ro, km) in enumerate(chunks):
            srecs, soff, skm = _make_records(s, P, rb, 100)          # what source s holds (deterministic)
            assert pos % rb == 0
            base = pos // rb
            assert ro[lo] == 0 and (ro[:lo + 1] == 0).all() and (ro[hi:] == ro[hi]).all()
            for p in range(P):
                n = ro[p + 1] - ro[p]
                if lo <= p < hi:
                    exp = srecs[soff[p]:soff[p + 1]]
                    ok &= n == len(exp) and np.array_equal(got[base + ro[p]: base + ro[p + 1]], exp) and km[p] == skm[p]
                else:
                    ok &= n == 0 and km[p] == 0
            total += ro[-1]
        ok &= total == len(got)
        q.put((rank, bool(ok), int(total)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,rb,chunk", [(8, 16, None), (6, 32, None), (8, 16, 96)])
def test_bucket_exchange_world2_gloo(P, rb, chunk):
    """chunk=96 bytes forces several point-to-point messages per peer (the >= 2 GiB work-around path)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, rb, q, chunk)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    # every record sent arrives exactly once somewhere
    sent = sum(int(_make_records(r, P, rb, 100)[1][-1]) for r in range(world))
    assert sum(t for _, _, t in res) == sent


def test_owner_ranges():
    ge.load()
    from gatb_core_amd import dist as gd
    assert gd.owner_ranges(8, 2) == [(0, 4), (4, 8)]
    assert gd.owner_ranges(4096, 8)[7] == (3584, 4096)
    with pytest.raises(ValueError):
        gd.owner_ranges(10, 4)


def _or_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ge.load()
    from gatb_core_amd import dist as gdist
    rng = np.random.default_rng(100 + rank)
    mine = rng.integers(0, 256, size=100_003, dtype=np.uint8) & rng.integers(0, 256, size=100_003, dtype=np.uint8)
    t = torch.from_numpy(mine.copy())
    gdist.allreduce_or(t)
    q.put((rank, mine, t.numpy().copy()))
    dist.barrier(); dist.destroy_process_group()


def test_bloom_or_reduce_gloo():
    """partial Bloom filters of the ranks are combined with a bitwise-OR all-reduce (dist.allreduce_or)"""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=_or_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(timeout=60) for p in ps]
    want = got[0][1] | got[1][1]
    for _, _, red in got:
        assert np.array_equal(red, want)
