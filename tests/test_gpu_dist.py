"""Multi-rank execution for real: 2, 4 and 8 processes on the one available GPU, each with its own context, running the whole distributed
path through the C-ABI — gkc_exchange (owner ranges balanced by weight, unequal numbers of pushes), Stage B on the owned partitions,
gkc_bloom_allreduce_or, gkc_mphf_build_solid_dist + gkc_mphf_abundance_map_dist — and compared with the oracle / with a single-context
run over all reads. RCCL refuses several ranks on one device, so the ranks talk through the host-staged gloo transport
(gatb-core_amd/dist.py:HostStagedTransport = the two callbacks of gkc_transport); everything above the two callbacks is the code the
RCCL communicator runs — or through device-to-device copies over IPC memory handles (gkc_comm_enable_ipc). BASELINE configs[2] / configs[4] at test size."""
import os
import socket

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cuts(n, world):
    """uneven shares of n reads: rank r scans reads [cut[r], cut[r+1]) — the first rank the largest share, the last ranks small ones"""
    w = np.array([world + 2 - r if r else 2 * world for r in range(world)], dtype=np.float64)
    c = np.concatenate([[0], np.round(np.cumsum(w) / w.sum() * n)]).astype(int)
    c[-1] = n
    return c


def _pinned_owners(parts, world):
    """owner ranges with uneven sizes; from 3 ranks on rank 1 owns NOTHING: first[world + 1] for gkc_comm_set_owners"""
    if world == 2:
        return np.asarray([0, parts // 3, parts], dtype=np.uint32)
    inner = np.sort(np.random.default_rng(world).choice(np.arange(1, parts), size=world - 2, replace=False))
    return np.asarray([0, int(inner[0]), int(inner[0])] + [int(v) for v in inner[1:]] + [parts], dtype=np.uint32)


def _rank_main(rank, world, port, k, parts, amin, q, try_rccl=False, ipc=False, owners=None, premap=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    if ipc:
        os.environ["GKC_IPC"] = "1"                       # the gloo-backed transport with device-to-device copies through IPC memory handles (gkc_comm_enable_ipc)
    if premap:
        os.environ["GKC_VMM_MIN_MB"] = "1"                # blocks of >= 1 MiB are hipMemCreate-mapped ranges (by default: >= 64 MiB, never at test size)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = ge.load(); gkc = pkg.gkc
        from gatb_core_amd import dist as gd
        m = 8
        reads = synth_reads(3000, 15000, 150, seed=41, n_rate=0.001, ragged=True)
        rep = simple_repart(m, parts)
        # uneven shares; rank 0 scans its reads in TWO pushes, the others in ONE: their second exchange has nothing to send
        cut = _cuts(len(reads), world)
        mine = reads[cut[rank]:cut[rank + 1]]
        chunks = [mine[: len(mine) // 2], mine[len(mine) // 2:]] if rank == 0 else [mine, []]
        c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
        pre = None
        if premap:
            # a context that allocated BEFORE it got its communicator holds hipMemCreate-mapped ranges; an IPC handle cannot name them. A Bloom filter made now
            # (4 MiB of bits: a mapped range under GKC_VMM_MIN_MB=1) is later the receive buffer of gkc_bloom_allreduce_or: it must travel through a bounce block, not fail.
            pre = gkc.Bloom(c, "neighbor", 32 << 20, 7, k)
        dc = gd.DistributedCounter(c, rank, world, parts, try_rccl=try_rccl, owners=owners)
        # try_rccl: the ranks first ask the library for an RCCL communicator (refused: several ranks on one device), agree on the refusal and fall back to the host-staged
        # transport — the path a multi-GPU run takes if RCCL inside libgkc_hip.so does not come up between real peers (gatb-core_amd/dist.py:make_comm)
        # ... which since round 5 keeps the records on the devices where it can: the fallback first tries device-to-device copies through IPC memory handles (checked by a
        # small exchange between the ranks) and only then stages through the host
        kind = gd.LAST_COMM_KIND
        if try_rccl:
            assert "(fallback: RCCL communicator refused on %d of %d ranks" % (world, world) in kind and kind.startswith(("device-to-device IPC", "host-staged")), kind
        elif ipc:
            assert kind.startswith("device-to-device IPC"), kind     # processes on one GPU can open each other's allocations (dmabuf IPC)
        else:
            assert kind == "host-staged", kind
        bad, _ = dc.comm.selftest(3 << 20)                   # every rank sends a keyed pattern to every other rank and checks what it gets (what bench.py --gpus N does first)
        assert bad == 0
        c.begin_pass(0)
        for ch in chunks:
            if ch:
                b, o = gko.pack_reads(ch); c.push_reads(b, o)
            dc.exchange()
        c.finish_pass()
        first = dc.owners()
        owned = {}
        for p in range(parts):
            lo, hi, ab = c.partition(0, p)
            if first[rank] <= p < first[rank + 1]:
                owned[p] = (lo, hi, ab)
            else:
                assert len(lo) == 0, "partition %d is not mine but holds records" % p
        st = c.stats(); cs = dc.stats()
        sent, recv, _ = dc.comm.peer_bytes(world)               # per-peer accounting of the grouped send / receive path: the self-test's 3 MiB and the records
        others = [r for r in range(world) if r != rank]
        assert int(sent[rank]) == 0 and int(recv[rank]) == 0
        assert all(int(sent[r]) >= (3 << 20) and int(recv[r]) >= (3 << 20) for r in others)
        assert sum(int(sent[r]) for r in others) >= (world - 1) * (3 << 20) + cs["bytes_sent"] and sum(int(recv[r]) for r in others) >= (world - 1) * (3 << 20) + cs["bytes_received"]
        # Bloom over the solid k-mers of all ranks: every rank inserts its own, then the OR all-reduce
        if pre is not None:
            bl = pre; nbits = 32 << 20
        else:
            nbits = 600_000; bl = gkc.Bloom(c, "neighbor", nbits, 7, k)
        bl.insert_solid(); bl.allreduce_or(dc.comm)
        if premap:
            assert dc.stats()["ipc_bounced"] > 0, "the mapped Bloom array was expected to travel through a bounce block"
        # MPHF + abundance map over all ranks
        mp_ = gkc.Mphf(c, comm=dc.comm)
        amap, above = mp_.abundance_map()
        barr = bl.array()
        q.put((rank, first.tolist(), owned, st, cs, barr if nbits <= 600_000 else np.frombuffer(__import__("hashlib").sha256(barr.tobytes()).digest(), np.uint8), mp_.save(), amap, above, mp_.size, kind))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# (world, k, parts, amin, try_rccl, ipc, owners, premap). owners: None = balanced by weight at the first exchange, "pinned" = gkc_comm_set_owners with uneven ranges and
# an EMPTY one (rank 1 owns nothing). premap: the context holds mapped ranges from before the communicator.
CASES = [(2, 31, 12, 2, False, False, None, False), (2, 41, 8, 1, False, False, None, False), (2, 31, 12, 2, True, False, None, False), (2, 31, 12, 2, False, True, None, False),
         (2, 63, 8, 1, False, True, None, False), (2, 31, 12, 2, False, True, None, True),
         (4, 31, 12, 2, False, False, None, False), (4, 63, 9, 1, False, True, "pinned", False), (4, 31, 12, 2, True, False, None, False),
         (8, 31, 24, 2, False, True, None, False), (8, 63, 20, 1, False, False, "pinned", False), (8, 31, 24, 1, False, True, "pinned", True), (8, 41, 9, 2, True, False, None, False)]


@pytest.mark.parametrize("world,k,parts,amin,try_rccl,ipc,owners,premap", CASES)
def test_ranks_on_one_gpu_end_to_end(world, k, parts, amin, try_rccl, ipc, owners, premap):
    """BASELINE configs[2] / configs[4] at test size with 2, 4 and 8 ranks (VERDICT r5 missing #1): every piece of the 8-GPU job — owner ranges, the 8-peer exchange plan,
    per-peer chunking, OR-reduce, distributed MPHF — runs end to end between real processes, over the host-staged and the IPC device-to-device transports and the
    RCCL-refused fallback; only the wire (xGMI) is missing."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    pinned = _pinned_owners(parts, world) if owners == "pinned" else None
    import time
    t0 = time.time()
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, k, parts, amin, q, try_rccl, ipc, pinned, premap)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    t1 = time.time()
    [p.join(timeout=120) for p in procs]
    print("TIMING world %d: ranks' results after %.1f s, processes gone after %.1f s more" % (world, t1 - t0, time.time() - t1))
    assert all(p.exitcode == 0 for p in procs)
    # ---- expected: the oracle over ALL reads, and a single-context run for the Bloom / MPHF bytes
    gkc = ge.load().gkc
    m = 8
    reads = synth_reads(3000, 15000, 150, seed=41, n_rate=0.001, ragged=True)
    rep = simple_repart(m, parts)
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, parts, rep)            # every distinct k-mer; the solidity window is applied below
    first = res[0][1]
    assert all(r[1] == first for r in res) and first[0] == 0 and first[-1] == parts and all(first[i] <= first[i + 1] for i in range(world))
    if pinned is not None:
        assert first == pinned.tolist() and first[1] == first[2]                    # the empty range stayed empty
    elif parts >= 2 * world:
        assert all(first[i] < first[i + 1] for i in range(world))                   # balanced: everybody owns something
    seen = 0
    if try_rccl:
        print("fallback transport after the RCCL refusal:", res[0][10])
    for rank, _, owned, st, cs, *_ in res:
        assert sorted(owned) == list(range(first[rank], first[rank + 1]))
        for p, (lo, hi, ab) in owned.items():
            rlo, rhi, rab = ref.part(p)
            keep = rab >= amin
            assert np.array_equal(lo, rlo[keep]) and np.array_equal(hi, rhi[keep]) and np.array_equal(ab, rab[keep]), "partition %d differs from the oracle" % p
            seen += 1
        assert cs["n_exchanges"] == 2
        if first[rank] < first[rank + 1]:
            assert cs["bytes_received"] > 0
        else:
            assert cs["bytes_received"] == 0 and st["kmers_nb_distinct"] == 0
    assert seen == parts
    assert sum(r[4]["bytes_sent"] for r in res) == sum(r[4]["bytes_received"] for r in res) > 0
    assert sum(r[3]["kmers_nb_valid"] for r in res) == ref.stats["kmers_nb_valid"]
    assert sum(r[3]["kmers_nb_distinct"] for r in res) == ref.stats["kmers_nb_distinct"]
    one = gkc.Counter(0); one.configure(k, m, parts, rep); one.set_solidity(amin, 2147483647, 10000); one.count(bases, offs)
    nbits = (32 << 20) if premap else 600_000
    bl = gkc.Bloom(one, "neighbor", nbits, 7, k); bl.insert_solid()
    exp_bloom = bl.array() if not premap else np.frombuffer(__import__("hashlib").sha256(bl.array().tobytes()).digest(), np.uint8)
    mp1 = gkc.Mphf(one); amap1, above1 = mp1.abundance_map()
    for r in res:
        assert np.array_equal(r[5], exp_bloom), "OR-reduced Bloom filter differs from the single-GPU filter"
        assert r[9] == mp1.size and np.array_equal(r[6], mp1.save()), "multi-rank MPHF stream differs from gkc_mphf_save of the single-GPU run"
        assert np.array_equal(r[7], amap1) and r[8] == above1


def _files_rank_main(rank, world, box, k, parts, amin, q, bad_model):
    """one rank of a communicator over the library's file-mailbox transport (gkc_comm_create_files): no torch.distributed, nothing shared but a directory"""
    try:
        pkg = ge.load(); gkc = pkg.gkc
        m = 8
        reads = synth_reads(2600, 14000, 150, seed=43, n_rate=0.001, ragged=True)
        rep = simple_repart(m, parts)
        if bad_model == "fault" and rank == 1:
            os.environ["GKC_FAULT"] = "exchange_local"                # this rank fails on its own inside gkc_exchange, after the tables were gathered
        elif bad_model and rank == 1:
            rep = rep[::-1].copy()                                    # another repartition table on this rank
        mine = reads[rank::world]                                     # the bank shared out read by read
        c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
        comm = gkc.Comm.files(c, box, world, rank)
        c.begin_pass(0)
        half = len(mine) // 2
        for ch in (mine[:half], mine[half:]):
            b, o = gko.pack_reads(ch); c.push_reads(b, o)
            try:
                c.exchange(comm)
            except gkc.GkcError as e:
                q.put((rank, "error", str(e))); return
        c.finish_pass()
        own_hist = c.histogram().copy()
        comm.gather_results(0)
        got = {p: c.partition(0, p) for p in range(parts)} if rank == 0 else None
        q.put((rank, "ok", got, c.histogram(), c.stats(), own_hist))
        comm.close()
    except Exception as e:      # noqa
        import traceback
        q.put((rank, "exception", traceback.format_exc()))


@pytest.mark.parametrize("world,k,parts,amin", [(2, 31, 9, 2), (2, 41, 6, 1), (4, 31, 9, 1), (8, 63, 11, 2), (8, 31, 6, 1)])
def test_results_gathered_on_one_rank_over_the_file_transport(tmp_path, world, k, parts, amin):
    """gkc_gather_results: after the exchange and Stage B on 2 / 4 / 8 ranks (processes on the one GPU, the library's own file-mailbox transport and its handshake with up
    to 7 joiners; 8 ranks over 6 partitions: two ranks own nothing), rank 0 holds EVERY
    partition's Count[] (== oracle over all reads), the summed histogram and the summed statistics: what one process would hold, i.e. what one .h5 needs
    (CountProcessorDump.hpp:85-95 creates all datasets in one file; GraphUnitigs.cpp:921-931 opens that file)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_files_rank_main, args=(r, world, str(tmp_path), k, parts, amin, q, False)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    assert all(r[1] == "ok" for r in res), res
    m = 8
    reads = synth_reads(2600, 14000, 150, seed=43, n_rate=0.001, ragged=True)
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, parts, simple_repart(m, parts))
    got = res[0][2]
    for p in range(parts):
        rlo, rhi, rab = ref.part(p); keep = rab >= amin
        lo, hi, ab = got[p]
        assert np.array_equal(lo, rlo[keep]) and np.array_equal(hi, rhi[keep]) and np.array_equal(ab, rab[keep]), "gathered partition %d differs from the oracle" % p
    assert np.array_equal(res[0][3], ref.histogram())                                   # rank 0: the histogram of the whole run
    assert np.array_equal(sum(r[5] for r in res), ref.histogram()) and sum(int(r[5].sum() > 0) for r in res) >= 2   # ... the sum of the ranks' own
    assert res[0][4]["kmers_nb_valid"] == ref.stats["kmers_nb_valid"] and res[0][4]["kmers_nb_distinct"] == ref.stats["kmers_nb_distinct"]


def test_ranks_with_different_models_are_refused(tmp_path):
    """ranks that hold different repartition tables must not exchange (one k-mer would be split over two owners): gkc_exchange compares a fingerprint
    of the model and fails on every rank"""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_files_rank_main, args=(r, world, str(tmp_path), 31, 8, 1, q, True)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    assert all(r[1] == "error" and "another model" in r[2] for r in res), res


def test_a_rank_failing_alone_takes_the_others_with_it(tmp_path):
    """a rank that fails by itself between the table all-gather and the transfer of gkc_exchange (its receive arena does not fit, say) must not leave the
    other ranks blocked in their send / recv: the ranks agree on a status word first (gkc_comm_agree) and every rank returns an error — the failing one its own,
    the others one that names it"""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_files_rank_main, args=(r, world, str(tmp_path), 31, 8, 1, q, "fault")) for r in range(world)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    [p.join(timeout=120) for p in procs]
    assert all(r[1] == "error" for r in res), res
    assert "rank 1 failed in gkc_exchange" in res[0][2] and "injected fault" in res[1][2], res
