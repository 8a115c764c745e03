#!/usr/bin/env python
"""bench.py — one "step" = one full pass of the DSK hot path (Stage A scan+bucket, Stage B expand+sort+count) over one
batch of synthetic 150 bp reads that is already resident in HBM when the timed region starts. `value` follows SURVEY 8(d): the clock
of a step stops when every distinct k-mer's Count record (abundance-min 1) is in page-locked HOST memory (the batches cross PCIe
packed while later batches are counted); `value_device_resident` is the same K steps with the records left in HBM.

Contract: `python bench.py --gpus N --steps K --warmup W` (for N>1 launched by torch.distributed.run, one rank per GPU).
Rank 0 prints ONE JSON line. metric = BASELINE.json's "distinct k-mers/s at k=31".

N=1 workload = BASELINE configs[1] (k=31, 10^8 synthetic 150 bp reads, 30x coverage, 1 % substitutions, single pass,
no Bloom); a step takes < 1 s.
N>1: weak scaling on BASELINE configs[2]'s per-GPU share — every rank scans 1.25e8 reads of one global stream (N=8: the 10^9 reads of
configs[2]); super-k-mer buckets are routed to the partition's owner rank by gkc_exchange (grouped ncclSend/ncclRecv inside
libgkc_hip.so, RCCL over xGMI), each rank counts the partitions it owns. The N=1 line also carries `config.share_of_8`: that same
per-GPU share (1.25e8 reads, the partition count of the 8-GPU run, 4 pushes + exchanges through a one-rank RCCL communicator) timed
on this GPU, i.e. the per-GPU cost of configs[2] before any byte moves over xGMI.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); ~6.3 TB/s achievable


def repart_for_bench(m, parts):
    """a balanced minimizer->partition table: splitmix hash of the minimizer value (any table is a valid Repartitor)"""
    gkc = ge.load().gkc
    x = gkc.mix64_np(np.arange(4 ** m, dtype=np.uint64))
    return (x % np.uint64(parts)).astype(np.uint16)


def algorithmic_bytes_per_kmer(k, L, nbar, d):
    """SURVEY.md §8(d) fixed accounting model (LSD 8-bit passes), bytes per valid input k-mer"""
    W = 8 if k <= 31 else 16
    R = 16 if k <= 31 else 32
    P = (2 * k + 7) // 8
    b_in = L / (L - k + 1)
    b_sk = (1 + -(-(k + nbar - 1) // 4)) / nbar
    return b_in + 2 * b_sk + W + 2 * W * P + W + W + R * d


def cpu_baseline(c, k, m, parts, rep, n_reads=10_000_000):
    """The CPU port (oracle/gkc_oracle.c: the reference's algorithm restated in C) run as a REAL parallel DSK on all host cores of rank 0 — reads
    shared out over the threads for fillPartitions, partitions dealt to the threads for fillSolidKmers, like the reference's Dispatcher
    (gko_dsk_run_mt; SortingCountAlgorithm.cpp:1266-1275, 1456-1587) — on a bounded sample of the same synthetic stream (generated on the device,
    copied to the host before the clock starts). kind "port": the reference itself cannot be built under this project's build rules."""
    from oracle import gko
    L = 150
    cores = os.cpu_count() or 1
    d_b, d_o = c.synth_reads_device(1, n_reads, L, n_reads * 5, 10000)
    bases = c.device_to_host(d_b, n_reads * L); offs = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(L)
    c.device_free(d_b); c.device_free(d_o)
    n1 = min(n_reads, 500_000)
    t0 = time.time()
    d1 = gko.Dsk(bases[: n1 * L], offs[: n1 + 1], k, m, parts, rep)
    dt1 = time.time() - t0
    single = d1.stats["kmers_nb_distinct"] / dt1
    del d1
    t0 = time.time()
    d = gko.Dsk(bases, offs, k, m, parts, rep, threads=cores)
    dt = time.time() - t0
    return {"value": d.stats["kmers_nb_distinct"] / dt, "unit": "distinct k-mers/s", "cores": cores, "kind": "port",
            "valid_kmers_per_s": d.stats["kmers_nb_valid"] / dt,
            "sample": "oracle/gkc_oracle.c gko_dsk_run_mt, %d threads (all host cores), %d synthetic 150 bp reads (same generator, seed 1, 30x, %d partitions): %.1f s; "
                      "one thread on the first %d reads: %.1f s = %.3g distinct k-mers/s" % (cores, n_reads, parts, dt, n1, dt1, single),
            "single_thread": single}


def kernel_roofline(ktime, steps, n_bases, st, k, keys, distinct, keys_moved=None, dom_from=None):
    """algorithmic bytes of every timed kernel group (SURVEY §8d per-unit figures x the units of one step) -> the dominant one's roofline entry.
    ktime: {name: (ms, launches)} over `steps` steps; keys / distinct: this rank's share. dom_from: the timers the dominant kernel is CHOSEN from (the device-resident
    steps: there an event interval is kernel work; in the host-landed region the intervals of the groups that talk to the host — "compact" fetches its prefix
    tables — also hold the waits of their small copies behind the bulk transfers); its time is then taken from `ktime`."""
    rec_bytes = 16 if k <= 31 else 32; key_bytes = 8 if k <= 31 else 16
    alg = {
        "scan_count": n_bases * 1.0,
        "scan_emit": n_bases * 1.0 + st["nb_superkmers"] * (rec_bytes + 4),
        "expand_count": st["nb_superkmers"] * rec_bytes,
        "expand_scatter": st["nb_superkmers"] * rec_bytes + keys * key_bytes,
        "bucket_sort": keys * key_bytes + distinct * (key_bytes + 1),
        "compact": distinct * (key_bytes + 1 + rec_bytes),             # gather: key + abundance byte in, Count record out
    }
    if "dedupe_bin" in ktime and ktime["dedupe_bin"][0] > 0:
        alg["dedupe_bin"] = st["nb_superkmers"] * rec_bytes * 3.0            # records read twice (bin count, bin scatter), written once
        alg["dedupe_sort"] = st["nb_superkmers"] * rec_bytes * 2.0           # ... read once and rewritten (at most once) by the sort
    pick = dom_from if dom_from is not None else ktime
    dom = max(alg, key=lambda n_: pick[n_][0])
    dom_ms = ktime[dom][0] / max(1, steps)
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    launches = max(1, ktime[dom][1] // max(1, steps))
    return alg, dom, dom_ms, achieved, launches


M64 = (1 << 64) - 1


def input_checksum(c, chunks):
    """order-independent checksum of the canonical k-mer multiset of the device-resident reads (an independent one-thread-per-position kernel: no minimizers,
    no buckets): sum of mix(canonical) over the valid k-mers mod 2^64, and their number"""
    cs, nv = 0, 0
    for b_, o_, nr, nb_ in chunks:
        s_, n_ = c.kmer_checksum_device(b_, o_, nr, nb_)
        cs = (cs + int(s_)) & M64; nv += int(n_)
    return cs, nv


def verify_block(c, expect, amin=1):
    """OUTSIDE the clock: is what the timed block left on the device the count of its input? `expect` = input_checksum of the reads it counted.
    abundance-min 1 (every distinct k-mer is a record): sum abundance * mix(value) over the records == the input's checksum, sum abundance == valid k-mers.
    abundance-min > 1 (only solid records are kept): the records' sum of abundances == sum i * h[i] over the window of the device's histogram, which itself must
    account for every valid k-mer and every distinct one. Returns {"verified": bool, ...}."""
    cs_in, nv = expect
    cs_out, sa = c.result_checksum()
    st = c.stats()
    out = {"valid_kmers_in": nv, "sum_abundance_of_records": int(sa)}
    if amin <= 1:
        out["method"] = "sum abundance * mix(value) over the Count records == checksum of the input's canonical k-mers (independent kernel); sum abundance == valid k-mers"
        out["verified"] = bool(int(cs_out) == cs_in and int(sa) == nv and st["kmers_nb_valid"] == nv)
    else:
        h = c.histogram().astype(np.int64)
        idx = np.arange(len(h), dtype=np.int64)
        clamped = int(h[-1]) != 0                                  # abundances beyond histo_max share the last bin: the weighted sums are lower bounds then
        tot = int((idx * h).sum()); solid_w = int((idx[amin:] * h[amin:]).sum())
        out["method"] = "device histogram: sum i*h[i] == valid k-mers, sum h == distinct, sum h[i >= %d] == solid records, their sum of abundances == sum i*h[i >= %d]" % (amin, amin)
        out["verified"] = bool((tot == nv or clamped) and int(h[1:].sum()) == st["kmers_nb_distinct"] and int(h[amin:].sum()) == st["kmers_nb_solid"]
                               and (int(sa) == solid_w or clamped) and st["kmers_nb_valid"] == nv)
    return out


REF_DIR = os.path.join(ROOT, "integration", "_build", "ref")      # unpatched reference tools built by integration/build_reference.sh (git-ignored, shipped by gpurun)


def cpu_baseline_reference(c, k, n_reads=10_000_000):
    """The REFERENCE ITSELF as the CPU baseline (kind "reference"): the unpatched `dbgh5` of GATB-Core, built from /root/reference with its own cmake by
    integration/build_reference.sh, runs its multithreaded SortingCountAlgorithm (SortingCountAlgorithm.cpp:636-781) on all host cores of this box over a
    FASTA of the same synthetic stream (bounded sample, written to /dev/shm before the clock starts; temporary super-k-mer files and the .h5 also in /dev/shm:
    no disk in the measurement). `value` = distinct k-mers / the DSK step's own clock (`dsk.time` = fill_partitions + fill_solid_kmers, read back with the
    reference's `dbginfo`; the process wall, which adds the bank estimate, the Repartitor sample and start-up, is reported beside it). None if the tools are absent."""
    import shutil, subprocess, tempfile
    dbgh5, dbginfo = os.path.join(REF_DIR, "dbgh5"), os.path.join(REF_DIR, "dbginfo")
    if not (os.path.exists(dbgh5) and os.path.exists(dbginfo)):
        return None
    L = 150
    cores = os.cpu_count() or 1
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="gkc_refbase_", dir=shm)
    try:
        d_b, d_o = c.synth_reads_device(1, n_reads, L, n_reads * 5, 10000)
        bases = c.device_to_host(d_b, n_reads * L)
        c.device_free(d_b); c.device_free(d_o)
        rec = np.empty((n_reads, L + 4), dtype=np.uint8)                 # ">r\n" + bases + "\n"
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = 10; rec[:, 3:3 + L] = bases.reshape(n_reads, L); rec[:, 3 + L] = 10
        fa = os.path.join(work, "reads.fa")
        rec.tofile(fa)
        del rec, bases
        try:
            mem_mb = int(os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / (1 << 20) * 0.5)
        except (ValueError, OSError):
            mem_mb = 64000
        cmd = [dbgh5, "-in", fa, "-kmer-size", str(k), "-abundance-min", "1", "-nb-cores", str(cores), "-max-memory", str(mem_mb),
               "-bloom", "none", "-debloom", "none", "-branching-nodes", "none", "-no-mphf", "-out", os.path.join(work, "ref"), "-verbose", "0"]
        t0 = time.time()
        r = subprocess.run(cmd, cwd=work, capture_output=True, text=True, timeout=900)
        wall = time.time() - t0
        if r.returncode != 0:
            return {"error": "reference dbgh5 failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        info = subprocess.run([dbginfo, "-in", os.path.join(work, "ref.h5")], cwd=work, capture_output=True, text=True, timeout=300).stdout
        vals = {}
        for line in info.splitlines():                                   # "   key   : value"
            key, sep, val = line.partition(":")
            if sep and val.strip():
                vals.setdefault(key.strip(), val.strip())
        def f(name):
            try:
                return float(vals[name])
            except (KeyError, ValueError):
                return None
        distinct, valid, t_dsk = f("kmers_nb_distinct"), f("kmers_nb_valid"), f("time")
        if not distinct or not t_dsk:
            return {"error": "could not read kmers_nb_distinct / time from dbginfo", "wall_s": wall}
        return {"value": distinct / t_dsk, "unit": "distinct k-mers/s", "cores": cores, "kind": "reference",
                "valid_kmers_per_s": (valid or 0) / t_dsk, "dsk_time_s": t_dsk, "process_wall_s": wall,
                "fill_partitions_s": f("fill_partitions"), "fill_solid_kmers_s": f("fill_solid_kmers"),
                "fillsolid_read_sort_dump_s": [f("1.read"), f("2.sort"), f("3.dump")],
                "nb_partitions": f("nb_partitions"), "nb_passes": f("nb_passes"), "distinct_kmers": distinct,
                "sample": "GATB-Core's own dbgh5 (unpatched, built from the reference sources by integration/build_reference.sh): -kmer-size %d -abundance-min 1 -nb-cores %d "
                          "-max-memory %d -bloom none -debloom none -branching-nodes none -no-mphf, on %d synthetic 150 bp reads (same generator, seed 1, 30x) as FASTA in %s; "
                          "value = kmers_nb_distinct / dsk.time (SortingCountAlgorithm::execute: fill_partitions + fill_solid_kmers)" % (k, cores, mem_mb, n_reads, shm)}
    except Exception as e:      # noqa
        return {"error": "reference baseline: %r" % (e,)}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def fastq_parse_leg(c, n_reads=1_000_000, L=150):
    """input side, reported beside the metric (never inside it): a 4-line FASTQ text of n_reads reads already resident in HBM ->
    flat bases + offsets on the device (gkc_fastx_parse_device). Returns text GB/s and bases/s."""
    import torch
    rng = np.random.default_rng(1)
    hdr = 12
    rec = np.empty((n_reads, hdr + L + 3 + L + 1), dtype=np.uint8)
    ids = np.char.zfill(np.arange(n_reads).astype("U10"), 10)
    rec[:, 0] = ord("@"); rec[:, 1:11] = np.frombuffer("".join(ids).encode(), dtype=np.uint8).reshape(n_reads, 10); rec[:, 11] = 10
    rec[:, hdr:hdr + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n_reads, L))]
    rec[:, hdr + L:hdr + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    rec[:, hdr + L + 3:hdr + 2 * L + 3] = rng.integers(33, 74, size=(n_reads, L), dtype=np.uint8)
    rec[:, -1] = 10
    t = torch.from_numpy(rec.reshape(-1)).cuda()
    torch.cuda.synchronize()
    b = C.c_void_p(); o = C.c_void_p(); nr = C.c_uint64(0); nb = C.c_uint64(0); cons = C.c_uint64(0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        c._chk(c.L.gkc_fastx_parse_device(c.h, t.data_ptr(), t.numel(), 1, C.byref(b), C.byref(o), C.byref(nr), C.byref(nb), C.byref(cons)))
        dt = time.perf_counter() - t0                      # the call returns after its last kernel has finished
        c.device_free(b.value); c.device_free(o.value)
        best = dt if best is None else min(best, dt)
    assert nr.value == n_reads and nb.value == n_reads * L
    return {"sample": "%d reads of %d bp, 4-line FASTQ, %d bytes of text resident in HBM" % (n_reads, L, t.numel()),
            "text_GBps": t.numel() / best / 1e9, "gbases_per_s": n_reads * L / best / 1e9, "ms": best * 1e3}


def make_sink(c, gkc, distinct, world=1):
    """page-locked host sink for every Count record of a pass (+1 %). The sink and the library's staging buffer are page-locked: 1.6x the records; on a host that
    cannot spare twice that (per rank) it is not made (it must not take the box down). Returns (sink, None) or (None, {"skipped" | "error": why})."""
    RB = c.rec_bytes
    try:
        import psutil
        avail = psutil.virtual_memory().available / max(1, world)
        if avail < 2 * 1.6 * distinct * RB:
            return None, {"skipped": "page-locked sink + staging of %.0f GB on a host with %.0f GB available per rank" % (1.6 * distinct * RB / 1e9, avail / 1e9)}
    except ImportError:
        pass
    t_alloc = time.perf_counter()
    try:
        sink = gkc.HostBuffer(int(distinct * RB * 1.01) + (64 << 20))
    except Exception as e:      # noqa
        return None, {"error": "page-locked sink of %.1f GB: %s" % (distinct * RB / 1e9, e)}
    sink.alloc_s = time.perf_counter() - t_alloc
    return sink, None


def pcie_probe(c, sink):
    """what the box's PCIe link sustains device -> page-locked host (4 copies of 1 GB), GB/s"""
    import torch
    probe = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    n_probe = max(1, min(4 << 30, sink.nbytes) // (1 << 30))
    t0 = time.perf_counter()
    for i in range(n_probe):
        c._chk(c.L.gkc_device_to_host(c.h, C.c_void_p(sink._p.value + (i << 30)), probe.data_ptr(), min(1 << 30, sink.nbytes)))
    pcie = n_probe * min(1 << 30, sink.nbytes) / (time.perf_counter() - t0) / 1e9
    del probe
    return pcie


def host_landed_leg(c, gkc, step, sync, distinct, n_steps=5, n_steps_amin2=5, expect=None, parts=0, amins=(1, 2), host_to_host=True, sink=None):
    """SURVEY §8(d) wall: from the first push to the last partition's Count[] in (page-locked) host memory. Every Stage-B batch is copied into
    the host sink on a copy stream while the next batches are counted (gkc_set_host_sink); gkc_finish_pass returns when everything has landed.
    Reported beside `value` (which stops with the results in HBM), at abundance-min 1 (every distinct k-mer travels: PCIe-bound) and 2."""
    import torch
    out = {}
    RB = c.rec_bytes                                          # 16 (k <= 31) / 32 (k <= 63): Kmer<span>::Count in memory
    if sink is None:
        sink, why = make_sink(c, gkc, distinct)
        if sink is None:
            return why
    out["sink_GB"] = sink.nbytes / 1e9; out["sink_alloc_s"] = sink.alloc_s
    pcie = pcie_probe(c, sink)
    out["pcie_d2h_GBps"] = pcie
    c.set_host_sink(sink)
    for amin in amins:
        c.set_solidity(amin, 2147483647, 10000)
        step(); sync()                                        # the batch plan changes with the solidity window: one untimed step
        ns_ = n_steps_amin2 if amin == 2 else n_steps
        per = []
        t0 = time.perf_counter()
        for _ in range(ns_):
            t1 = time.perf_counter(); step(); per.append((time.perf_counter() - t1) * 1e3)      # (gkc_finish_pass returns when the last batch has landed)
        sync()
        dt = (time.perf_counter() - t0) / ns_
        st = c.stats()
        landed = st["kmers_nb_solid"] * RB
        # what crosses the link: the batches travel packed (csrc/gkc_sink.hip) unless GKC_SINK_PACKED=0 — 6.3 bytes per record where the partitions are dense and most
        # abundances are 1, 7 where dense, 8 where sparse; the library counts the bytes it queued (gkc_stats.reserved[1])
        packed = st.get("sink_wire_bytes", 0) > 0
        wire = st["sink_wire_bytes"] if packed else st["kmers_nb_solid"] * RB
        err = (c.L.gkc_last_error(c.h) or b"").decode()
        out["abundance_min_%d" % amin] = {"value": st["kmers_nb_distinct"] / dt, "unit": "distinct k-mers/s with every solid Count[] in page-locked host memory",
                                          "ms_per_step": dt * 1e3, "ms_per_step_median": float(np.median(per)), "ms_steps": [round(x, 1) for x in per], "steps": ns_,
                                          "solid_records": st["kmers_nb_solid"], "bytes_landed": landed,
                                          "bytes_over_the_link": wire, "packed_on_the_wire": packed,
                                          "landed_GBps_over_the_step": landed / dt / 1e9, "link_GBps_over_the_step": wire / dt / 1e9, "frac_of_pcie": wire / dt / 1e9 / pcie,
                                          "sink_overflow": "sink" in err}
        if expect is not None:                                 # outside the clock: the device's records are the count of the input, and what landed is those records
            v = verify_block(c, expect, amin)
            try:
                same = True
                for p_ in sorted({0, parts // 2, parts - 1}) if parts else []:
                    host, n_ = c.wait_partition(0, p_)
                    if n_:
                        same = same and host is not None and bool(np.array_equal(host, c.partition_records(0, p_)))
                v["sink_spot_check"] = "first / middle / last partition: bytes in the sink == bytes on the device: %s" % same
                v["verified"] = bool(v["verified"] and same)
            except Exception as e:      # noqa
                v["sink_spot_check"] = "skipped: %r" % (e,)
            out["abundance_min_%d" % amin]["verified"] = v["verified"]; out["abundance_min_%d" % amin]["verification"] = v
    # PCIe at BOTH ends (never `value`): bases in page-locked host memory in (gkc_push_reads: H2D of chunk j+1 under the scan of chunk j), every solid Count[]
    # into the page-locked sink out; 5e7 reads of the same generator (30x over their own genome)
    try:
        if not host_to_host:
            raise StopIteration
        n2, L = 50_000_000, 150
        db, do = c.synth_reads_device(2, n2, L, n2 * 5, 10000)
        pin = gkc.HostBuffer(n2 * L)
        c._chk(c.L.gkc_device_to_host(c.h, pin._p, db, n2 * L))
        c.device_free(db); c.device_free(do)
        ho = np.arange(n2 + 1, dtype=np.uint64) * np.uint64(L)
        h2h = {"reads": n2, "bases_in_GB": n2 * L / 1e9}
        def step2():
            c.begin_pass(0); c.push_reads(pin.a, ho); c.finish_pass()
        for amin in (2, 1):
            c.set_solidity(amin, 2147483647, 10000)
            step2(); sync()
            per2 = []
            t0 = time.perf_counter()
            for _ in range(5):
                t1 = time.perf_counter(); step2(); per2.append((time.perf_counter() - t1) * 1e3)
            sync(); dt = (time.perf_counter() - t0) / 5
            st = c.stats()
            h2h["abundance_min_%d" % amin] = {"value": st["kmers_nb_distinct"] / dt, "unit": "distinct k-mers/s, pinned host bases in -> solid Count[] in pinned host memory",
                                              "ms_per_step": dt * 1e3, "ms_per_step_median": float(np.median(per2)), "steps": 5, "count_bytes_out_GB": st["kmers_nb_solid"] * RB / 1e9}
        out["host_to_host"] = h2h
        del pin
    except StopIteration:
        pass
    except Exception as e:      # noqa
        out["host_to_host"] = {"error": repr(e)}
    c.set_host_sink(None)
    c.set_solidity(1, 2147483647, 10000)
    return out


def bind_to_gpu_numa_node(torch, local):
    """Several ranks on one host: this process (and with it the page-locked sink it allocates and the library's unpack threads, which follow the sink's node) goes to the
    NUMA node its GPU hangs on — 8 ranks left where the launcher put them tend to land on one socket and share ITS memory controllers for 8 x 100 GB per step.
    Returns a note for the line, or None when the topology cannot be read (nothing is changed then)."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return "rank bound to NUMA node %d of GPU %s (%d cores)" % (node, bdf, len(cpus))
    except Exception:      # noqa
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 10^8 = BASELINE configs[1] on one GPU; 1.25e8 = configs[2]'s per-GPU share on several)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--partitions", type=int, default=0, help="0 = auto (about 4M k-mers per partition)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bloom-mphf", action="store_true", help="skip the Bloom + MPHF block (BASELINE configs[4] on one GPU's share)")
    ap.add_argument("--no-freq-order", action="store_true", help="skip the block with minimizers in frequency order (the mode GraphUnitigs forces)")
    ap.add_argument("--no-skewed", action="store_true", help="skip the block on the repeat-rich genome with low-complexity reads")
    ap.add_argument("--no-k63", action="store_true", help="skip the second block (BASELINE configs[3]: k=63 at the same size)")
    ap.add_argument("--no-host-landed", action="store_true", help="skip the host-landed / host-to-host legs (results streamed into page-locked host memory)")
    ap.add_argument("--no-two-pass", action="store_true", help="skip the two-pass block (Stage A of pass 1 overlapped with Stage B of pass 0)")
    ap.add_argument("--no-share-of-8", action="store_true", help="skip the N=1 block that times BASELINE configs[2]'s per-GPU share")
    ap.add_argument("--pushes", type=int, default=4, help="multi-GPU: pushes (and exchanges) per pass and rank")
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: everything else that writes to fd 1 (RCCL prints its version banner there)
    # is routed to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("GKC_FORCE_DIST") == "1"       # GKC_FORCE_DIST=1: exercise the exchange path with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        # GKC_BENCH_BACKEND=gloo: the dry run of this file's several-rank path on a box with fewer GPUs than ranks (the ranks share the devices, the exchange goes
        # through the host-staged transport of gatb_core_amd.dist; RCCL refuses two ranks on one device). Never what the driver runs; the line says which one it was.
        backend = os.environ.get("GKC_BENCH_BACKEND", "nccl")
        local = local % max(1, torch.cuda.device_count())
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local)
    # The process (and with it the page-locked sink it allocates, and the library's expansion threads, which follow the sink's node) goes to the NUMA node its GPU hangs on:
    # a sink on the other socket costs the headline 8 % on the 2-socket host of these boxes (465 vs 502 ms per step, same wire bytes, round 6). One rank: the original
    # affinity comes back before the CPU baseline runs (it uses every core of the host).
    affinity0 = os.sched_getaffinity(0)
    numa_note = bind_to_gpu_numa_node(torch, local) if os.environ.get("GKC_BENCH_NO_NUMA_BIND") is None else None
    red_dev = "cuda" if (not use_dist or dist.get_backend() == "nccl") else "cpu"      # where the few scalars of the line are all-reduced
    gkc = ge.load().gkc
    if not os.path.exists(gkc.SO):
        raise SystemExit("libgkc_hip.so missing: run __graft_entry__.build()")

    k, m, L = args.k, args.m, 150
    SHARE_OF_8 = 125_000_000                                  # BASELINE configs[2]: 10^9 reads on 8 GPUs
    n_reads = args.reads or (100_000_000 if world == 1 else SHARE_OF_8)
    n_kmers = n_reads * (L - k + 1)
    # about 3e6 k-mers per partition (8192 sub-buckets of ~370 keys), a power of two: 4096 for configs[1], 32768 = 4096 per rank for configs[2] at 8 GPUs
    parts = args.partitions or int(min(32768, max(64 * world, 2 ** int(np.floor(np.log2(max(1, n_kmers * world / 3.0e6)) + 0.5)))))
    parts = (parts + world - 1) // world * world
    rep = repart_for_bench(m, parts)

    c = gkc.Counter(local)
    c.configure(k, m, parts, rep)
    genome = max(L, n_reads * world * L // 30)               # 30x coverage over the whole job
    # rank r draws reads [r*n, (r+1)*n) of ONE global read stream over the SAME genome (seed 2). Multi-GPU: the rank's reads are pushed in
    # n_push chunks, each followed by gkc_exchange — the exchange of chunk i (RCCL, the communicator's own stream) overlaps Stage A of chunk i+1
    # one rank: one push per 1.25e8 reads at most (the size Stage A's per-push buffers — 0.8 bytes of descriptors per base, the record arena and, above 4096 partitions,
    # its refined copy — are meant for; a single push of 2e8 reads works but spends seconds in hipMalloc on a GPU that also holds the previous step's results)
    n_push = args.pushes if use_dist else max(1, -(-n_reads // 125_000_000))
    per_push = (n_reads + n_push - 1) // n_push
    chunks = []
    for i in range(n_push):
        nr = min(per_push, n_reads - i * per_push)
        if nr > 0:
            b_, o_ = c.synth_reads_device(2, nr, L, genome, 10000, first_read=rank * n_reads + i * per_push)
            chunks.append((b_, o_, nr, nr * L))
    n_bases = n_reads * L

    if use_dist:
        from gatb_core_amd import dist as gdist          # noqa
        t_c0 = time.perf_counter()
        runner = gdist.DistributedCounter(c, rank, world, parts, try_rccl=os.environ.get("GKC_BENCH_TRY_RCCL") == "1")      # (RCCL refused on any rank: host-staged fallback, reported in `exchange.transport`)
        if rank == 0 or "fallback" in (gdist.LAST_COMM_KIND or ""):
            sys.stderr.write("[bench rank %d] communicator: %s\n" % (rank, gdist.LAST_COMM_KIND))
        comm_create_s = time.perf_counter() - t_c0
        # Start-up self-test over the REAL peers, before anything is timed (default on with several ranks; GKC_COMM_SELFTEST=0 skips it): every pair of GPUs exchanges
        # 64 MiB and 300 MiB (two chunks of the 256 MiB chunking) of a keyed pattern through the communicator's grouped ncclSend / ncclRecv path — the first thing
        # an 8-GPU run does with RCCL is a checked transfer, not the job. A mismatch or an RCCL error fails every rank loudly here.
        selftest = None
        if os.environ.get("GKC_COMM_SELFTEST", "1" if world > 1 else "0") != "0":
            selftest = {}
            for nb_ in (64 << 20, 300 << 20):
                bad_, ms_ = runner.comm.selftest(nb_)
                selftest["%d_MiB" % (nb_ >> 20)] = {"mismatching_words": int(bad_), "ms": ms_, "GBps_per_peer": (nb_ / (ms_ * 1e-3) / 1e9) if ms_ > 0 else None}
                sys.stderr.write("[bench rank %d] comm self-test %d MiB to each of %d peer(s): %d mismatching words, %.1f ms\n" % (rank, nb_ >> 20, max(world - 1, 1), bad_, ms_))
                if bad_:
                    raise SystemExit("rank %d: the communicator self-test received %d wrong words" % (rank, bad_))
    else:
        runner = None

    n_steps_run = [0]

    def step():
        n_steps_run[0] += 1
        c.begin_pass(0)
        for b_, o_, nr, nb_ in chunks:
            c.push_reads_device(b_, o_, nr, nb_)
            if runner is not None:
                runner.exchange()
        c.finish_pass()

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # per-kernel timers are HIP events on the context's own stream, accumulated inside the library
    names = ["scan_count", "scan_emit", "scan_refine", "dedupe_bin", "dedupe_sort", "expand_count", "expand_scatter", "bucket_sort", "bucket_sort_big", "bucket_sort_wg", "split_levels", "compact",
             "total_stage_a", "total_stage_b"]

    def timed_region(n_warm, n_steps):
        """n_warm untimed steps, then EXACTLY n_steps steps bracketed by barrier + synchronize on both sides; max over ranks. Returns (seconds, per-step ms of this rank, kernel timers)"""
        for _ in range(n_warm):
            step()
        sync()
        base = {nme: c.timing(nme) for nme in names}
        per = []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            t1 = time.perf_counter(); step(); per.append((time.perf_counter() - t1) * 1e3)
        sync()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], device=red_dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); dt_ = float(t.item())
        kt = {nme: ((c.timing(nme)[0] - base[nme][0]), (c.timing(nme)[1] - base[nme][1])) for nme in names}
        return dt_, per, kt

    # the wall of configs 2-4 runs from the first push to the last partition's Count[] in HOST memory at abundance-min 1. That is what `value` times
    # (VERDICT r4 #1a): every Stage-B batch crosses PCIe packed into a page-locked sink while the next batches are counted (gkc_set_host_sink); gkc_finish_pass returns
    # when the last record has landed. The same K steps with the records left in HBM are timed right after as `value_device_resident`.
    # GKC_BENCH_VALUE=device (or a host that cannot page-lock the sink: reported) makes the HBM-resident figure `value`.
    value_mode = os.environ.get("GKC_BENCH_VALUE", "host")
    sink = None; sink_why = None
    step(); sync()                                               # one untimed step: the allocator's blocks, and this rank's distinct k-mers (the size of the sink)
    if value_mode == "host":
        sink, sink_why = make_sink(c, gkc, c.stats()["kmers_nb_distinct"], world)
        ok_all = 1 if sink is not None else 0
        if world > 1:
            t = torch.tensor([ok_all], device=red_dev); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok_all = int(t.item())
        if not ok_all:
            sink = None; value_mode = "device"
            sys.stderr.write("[bench rank %d] no host sink (%s): `value` is the HBM-resident figure\n" % (rank, sink_why))
    pcie_gbs = None; host_verification = None; landed = None
    if value_mode == "host":
        pcie_gbs = pcie_probe(c, sink)
        c.set_host_sink(sink)
        # Several ranks share ONE host: packed batches cost the link 0.4x but the host 104 B of DRAM traffic per record (6.3 B DMA-written, read again by the expansion
        # threads, 16 B of non-temporal stores) against 16 B raw — N ranks' expansion threads meet at the same memory controllers (tools/hostmem_probe/unpack_ranks_probe:
        # DESIGN.md section 5). So with N > 1 (or GKC_BENCH_SINK_MODE=trial) every rank runs one untimed step each way, all ranks at once, and the job keeps the mode whose
        # slowest rank was faster; GKC_BENCH_SINK_MODE=packed|raw pins it. One rank: packed (the link is the bound), no trial.
        sink_mode = os.environ.get("GKC_BENCH_SINK_MODE", "trial" if world > 1 else "packed")
        sink_trial = None
        if sink_mode == "trial":
            sink_trial = {}
            for md in ("packed", "raw"):
                c.set_sink_mode(md)
                step(); sync()                                   # (the batch plan of the mode; staging buffers)
                t0_ = time.perf_counter(); step(); sync(); d_ = time.perf_counter() - t0_
                if world > 1:
                    t = torch.tensor([d_], device=red_dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); d_ = float(t.item())
                sink_trial[md + "_ms"] = d_ * 1e3
            sink_mode = "raw" if sink_trial["raw_ms"] < sink_trial["packed_ms"] else "packed"
        c.set_sink_mode(sink_mode)
        dt_host, per_host, ktime_host = timed_region(max(1, args.warmup), args.steps)        # (the batch plan changes with a sink: at least one untimed step)
        st_h = c.stats()
        wire = st_h["sink_wire_bytes"] if st_h.get("sink_wire_bytes", 0) > 0 else st_h["kmers_nb_solid"] * c.rec_bytes
        err_ = (c.L.gkc_last_error(c.h) or b"").decode()
        landed = {"ms_per_step": dt_host / args.steps * 1e3, "ms_per_step_median": float(np.median(per_host)), "ms_steps": [round(x, 1) for x in per_host], "steps": args.steps,
                  "solid_records": st_h["kmers_nb_solid"], "bytes_landed": st_h["kmers_nb_solid"] * c.rec_bytes, "bytes_over_the_link": wire,
                  "packed_on_the_wire": st_h.get("sink_wire_bytes", 0) > 0, "pcie_d2h_GBps": pcie_gbs,
                  "link_GBps_over_the_step": wire / (dt_host / args.steps) / 1e9, "frac_of_pcie": wire / (dt_host / args.steps) / 1e9 / pcie_gbs,
                  "sink_GB": sink.nbytes / 1e9, "sink_alloc_s": sink.alloc_s, "sink_overflow": "sink" in err_,
                  "sink_mode": sink_mode, "sink_mode_trial": sink_trial}
        if numa_note:
            landed["host_placement"] = numa_note
        # outside the clock: what landed is what the device holds (first / middle / last partition this rank owns)
        try:
            same = True
            own = list(runner.owned()) if runner is not None else list(range(parts))
            for p_ in sorted({own[0], own[len(own) // 2], own[-1]}):
                host_, n_ = c.wait_partition(0, p_)
                if n_:
                    same = same and host_ is not None and bool(np.array_equal(host_, c.partition_records(0, p_)))
            landed["sink_spot_check"] = "first / middle / last owned partition: bytes in the sink == bytes on the device: %s" % same
        except Exception as e:      # noqa
            same = False; landed["sink_spot_check"] = "failed: %r" % (e,)
        landed["sink_ok"] = bool(same and not landed["sink_overflow"])
        if world == 1:                                           # ... and the device's records are the count of the input (several ranks: checked over all ranks below)
            hv = verify_block(c, input_checksum(c, chunks), 1)
            landed["verification"] = hv; landed["sink_ok"] = bool(landed["sink_ok"] and hv["verified"])
        landed["verified"] = landed["sink_ok"]
        if world > 1:
            t = torch.tensor([1 if landed["sink_ok"] else 0], device=red_dev); dist.all_reduce(t, op=dist.ReduceOp.MIN); landed["sink_ok"] = bool(int(t.item()))
        c.set_host_sink(None); c.set_sink_mode("packed")
        dt, per_dev, ktime = timed_region(1, args.steps)             # the same K steps with the records left in HBM
        dt_value, ktime_value = dt_host, ktime_host
    else:
        dt, per_dev, ktime = timed_region(args.warmup, args.steps)
        dt_value, ktime_value = dt, ktime
    st = c.stats()
    distinct = st["kmers_nb_distinct"]; valid = st["kmers_nb_valid"]
    if world > 1:
        t = torch.tensor([distinct, valid], device=red_dev, dtype=torch.int64); dist.all_reduce(t); distinct, valid = int(t[0]), int(t[1])
    # OUTSIDE the clock: the records the last timed step left on the device are the count of the reads it was given (every rank: its reads in, the partitions it
    # owns out; over all ranks the two sides must meet)
    expect = input_checksum(c, chunks)
    if world > 1:
        cs_out, sa = c.result_checksum()
        mine_v = (expect[0], expect[1], int(cs_out), int(sa))
        all_v = [None] * world
        dist.all_gather_object(all_v, mine_v)
        verification = {"method": "over all ranks: sum abundance * mix(value) of the owned partitions' records == checksum of every rank's input k-mers; sum abundance == valid k-mers",
                        "valid_kmers_in": sum(v[1] for v in all_v), "sum_abundance_of_records": sum(v[3] for v in all_v)}
        verification["verified"] = bool(sum(v[0] for v in all_v) & M64 == sum(v[2] for v in all_v) & M64 and verification["valid_kmers_in"] == verification["sum_abundance_of_records"] == valid)
    else:
        verification = verify_block(c, expect, 1)
    all_verified = [verification["verified"]]
    exch = None
    if runner is not None:                                     # per-rank exchange figures over warmup + timed steps (gkc_comm_get_stats; every step run so far, timed or not)
        cs = runner.stats(); n_st = max(1, n_steps_run[0])
        ps_, pr_, init_ms = runner.comm.peer_bytes(world)
        mine_x = {"rank": rank, "owned_partitions": len(runner.owned()), "exchanges_per_step": cs["n_exchanges"] / n_st,
                  "ms_transfer_per_step": cs["ms_transfer"] / n_st, "ms_transfer_per_exchange": cs["ms_transfer"] / max(1, cs["n_exchanges"]),
                  "ms_host_per_step": cs["ms_host"] / n_st,
                  "bytes_sent_per_step": cs["bytes_sent"] / n_st, "bytes_received_per_step": cs["bytes_received"] / n_st,
                  "nccl_comm_init_rank_ms": init_ms, "communicator_create_s": comm_create_s,
                  "bytes_sent_to_peer": [int(x) for x in ps_], "bytes_received_from_peer": [int(x) for x in pr_],     # whole run incl. the self-test, grouped send / recv path
                  "selftest": selftest}
        sys.stderr.write("[bench rank %d] ncclCommInitRank %.0f ms; %d exchanges, %.2f ms of transfer each; sent per peer %s MB, received per peer %s MB\n" % (
            rank, init_ms, cs["n_exchanges"], mine_x["ms_transfer_per_exchange"], [int(x) >> 20 for x in ps_], [int(x) >> 20 for x in pr_]))
        if world > 1:
            exch = [None] * world
            dist.all_gather_object(exch, mine_x)
        else:
            exch = [mine_x]
    # Stage B runs two lanes (two streams): inside the timed region a kernel's event duration includes the time it shares the chip with the
    # other lane's kernels. One extra UNTIMED step with a single lane gives the same kernels' durations in isolation (reported beside, never
    # instead of, the timed-region figures).
    iso = None
    if world == 1 and os.environ.get("GKC_STAGEB_LANES", "2") != "1":
        prev = os.environ.get("GKC_STAGEB_LANES")
        os.environ["GKC_STAGEB_LANES"] = "1"
        b1 = {nme: c.timing(nme) for nme in names}
        step(); sync()
        iso = {nme: ((c.timing(nme)[0] - b1[nme][0]), (c.timing(nme)[1] - b1[nme][1])) for nme in names}
        if prev is None:
            os.environ.pop("GKC_STAGEB_LANES")
        else:
            os.environ["GKC_STAGEB_LANES"] = prev

    if rank == 0:
        ms_step = dt_value / args.steps * 1e3                   # the region `value` is quoted on (host-landed unless the line says otherwise)
        value = distinct / (dt_value / args.steps)
        ms_step_dev = dt / args.steps * 1e3                     # the same K steps with the records left in HBM
        value_dev = distinct / (dt / args.steps)
        ktime_dev = ktime; ktime = ktime_value
        nbar = (valid / world) / max(1, st["nb_superkmers"])
        d = distinct / max(1, valid)
        # dominant kernel = the one with the largest accumulated time; its algorithmic bytes per launch are stated in DESIGN.md §Kernels
        keys_per_rank = valid / world
        key_bytes = 8 if k <= 31 else 16
        alg, dom, dom_ms, achieved, launches_per_step = kernel_roofline(ktime, args.steps, n_bases, st, k, keys_per_rank, distinct / world, dom_from=ktime_dev)
        workload = ("k=%d, %d synthetic 150 bp reads per GPU, single-pass count (no Bloom), m=%d, %d partitions" % (k, n_reads, m, parts))
        if world > 1:
            workload = ("BASELINE configs[2] at %d GPU(s): k=%d, %d synthetic 150 bp reads per GPU (%d in all), minimizer-partition exchange (gkc_exchange: RCCL send/recv over xGMI), "
                        "m=%d, %d partitions" % (world, k, n_reads, n_reads * world, m, parts))
        traffic = None
        kname = {"scan_count": "k_scan_tile<false, 2, true, false>", "scan_emit": "k_emit_desc<2>", "expand_count": "k_expand_count<1, 2>",
                 "expand_scatter": "k_expand_scatter_pair", "bucket_sort": "k_wave_sort<1, true>", "compact": "k_gather_counts<1>", "dedupe_bin": "k_dedupe_bin<2>", "dedupe_sort": "k_dedupe_sort<2>"}
        try:   # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py), only if they were taken on this exact workload
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == workload and kname.get(dom) in pt["kernels"]:
                kt = pt["kernels"][kname[dom]]                     # per launch = per-step bytes / this run's launches per step (the batch split may differ)
                traffic = kt["hbm_bytes_per_step"] / max(1, ktime[dom][1] // max(1, args.steps)) if "hbm_bytes_per_step" in kt else kt["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        pipeline = None
        try:   # the whole step against the HBM roof: every byte the step's kernels moved at the memory (PMC, single-lane passes) over the timed step
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if pt.get("workload") == workload:
                tot_b = sum(v_["hbm_bytes_per_step"] for n_, v_ in pt["kernels"].items() if n_.startswith("k_") and not n_.startswith("k_synth") and "checksum" not in n_)      # (the step's kernels: not the input generator, not the verification)
                pipeline = {"counter_bytes_per_step": tot_b, "bytes_per_valid_kmer": tot_b / max(1, valid), "achieved": tot_b / (ms_step_dev * 1e-3) / 1e9, "unit": "GB/s",
                            "frac": tot_b / (ms_step_dev * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "source": "profiles/pmc_traffic.json: HBM bytes (FETCH_SIZE, x2 where tools/pmc_traffic.py says so, + WRITE_SIZE) summed over every kernel of one step "
                                      "(separate single-lane PMC passes) / this run's ms_per_step_device_resident (the PCIe-bound host-landed step moves the same bytes in more time)"}
        except Exception:
            pipeline = None
        out = {
            "metric": "distinct k-mers/s at k=%d" % k, "value": value, "unit": "distinct k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64" if k <= 31 else "u128",
            "value_definition": ("SURVEY 8(d) wall: reads resident in HBM -> every distinct k-mer's Count record (abundance-min 1) in page-locked HOST memory; the timed K steps are "
                                 "host-landed steps (PCIe-bound: see `landed`)" if value_mode == "host" else
                                 "records left in HBM (no host sink: %s)" % (sink_why if sink_why else "GKC_BENCH_VALUE=device")),
            "ms_per_step_median": float(np.median(per_host if value_mode == "host" else per_dev)),
            "value_device_resident": value_dev, "ms_per_step_device_resident": ms_step_dev, "ms_per_step_device_resident_median": float(np.median(per_dev)),
            "landed": landed,
            "verified": verification["verified"], "verification": verification,
            "data": "synthetic (device generator, seeded; 150 bp reads, 30x, 1% substitutions)",
            "config": {"workload": workload,
                       "reads_per_gpu": n_reads, "partitions": parts, "valid_kmers": valid, "distinct_kmers": distinct,
                       "valid_kmers_per_s": valid / (dt / args.steps), "gbases_per_s": n_bases * world / (dt / args.steps) / 1e9,
                       "mean_kmers_per_superkmer": nbar, "distinct_ratio": d,
                       "model_bytes_per_kmer": algorithmic_bytes_per_kmer(k, L, nbar, d),
                       "model_GBps": valid / world * algorithmic_bytes_per_kmer(k, L, nbar, d) / (dt / args.steps) / 1e9,
                       "model_note": "SURVEY 8(d)'s FIXED accounting (an 8-pass LSD radix sort: 128 of its 160 bytes per k-mer): this build sorts with one MSD scatter and a "
                                     "register network, so model_GBps over-charges the bytes about 3.5x and can exceed the HBM peak; it is not a bandwidth. The measured "
                                     "whole-pipeline figure is roofline.pipeline (PMC bytes per step / step time)",
                       "kernel_ms_per_step": {n_: round(ktime[n_][0] / args.steps, 3) for n_ in names},
                       "kernel_ms_per_step_device_resident": {n_: round(ktime_dev[n_][0] / args.steps, 3) for n_ in names}},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate single-lane passes; bytes per step / launches per step)" if traffic else None,
                         "launches_per_step": int(launches_per_step), "launch_ms": dom_ms / launches_per_step,
                         "algorithmic_bytes_per_launch": alg[dom] / launches_per_step,
                         "stage_b_lanes": int(os.environ.get("GKC_STAGEB_LANES", "2")), "pipeline": pipeline},
        }
        # the same kernel over the K steps that leave the records in HBM (no pack kernels, no copy stream beside it): `frac` above belongs to the region `value` is quoted on
        if value_mode == "host" and ktime_dev[dom][0] > 0:
            d_ms = ktime_dev[dom][0] / max(1, args.steps); d_l = max(1, ktime_dev[dom][1] // max(1, args.steps))
            out["roofline"]["device_resident_region"] = {"launch_ms": d_ms / d_l, "launches_per_step": int(d_l), "achieved": alg[dom] / (d_ms * 1e-3) / 1e9,
                                                         "frac": alg[dom] / (d_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                         "note": "same kernel, same algorithmic bytes, HIP events over the K steps of value_device_resident"}
        # Stage B merges identical super-k-mer records before the expansion: every k-mer is still counted (the algorithmic bytes above are per k-mer of the
        # input), but the kernels after it move fewer keys. Both figures are reported: `frac` on the algorithmic bytes, `frac_on_moved_bytes` on what
        # the dominant kernel really reads and writes
        dd_in, dd_out = st.get("dedupe_kmers_in", 0), st.get("dedupe_keys_out", 0)          # of the last step (the statistics are per pass)
        keys_written = max(0.0, keys_per_rank - (dd_in - dd_out) / world) if dd_in else keys_per_rank
        out["roofline"]["dedupe"] = {"kmers_in": dd_in, "weighted_keys_out": dd_out, "keys_per_step_after": keys_written, "ratio": keys_per_rank / keys_written if keys_written else None}
        if dom in ("expand_scatter", "bucket_sort") and dd_in:
            shrink = keys_written / keys_per_rank
            moved = {"expand_scatter": (st["nb_superkmers"] * 16 if k <= 31 else st["nb_superkmers"] * 32) * shrink + keys_written * key_bytes,
                     "bucket_sort": keys_written * key_bytes + (distinct / world) * (key_bytes + 1)}[dom]
            out["roofline"]["moved_bytes_per_launch"] = moved / launches_per_step
            out["roofline"]["frac_on_moved_bytes"] = moved / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_ms > 0 else None
        else:
            keys_written = keys_per_rank
        if dom == "expand_scatter" and k <= 31:
            keys_per_rank = keys_written                       # the 16-byte-store ceiling below is about the keys that are really written
            # what actually bounds this kernel: the rate of scattered 16-byte stores with 8192 open cursors per workgroup measured on this chip
            # (tools/scatter_bench -> profiles/r02_scatter_store_calibration.txt: 984 GB/s of useful bytes; requests, not bytes, are the limit)
            key_gbs = keys_per_rank * key_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            out["roofline"]["scattered_store_ceiling"] = {"GBps": 984.0, "store_bytes": 16, "achieved_key_GBps": key_gbs, "frac": key_gbs / 984.0,
                                                          "source": "profiles/r02_scatter_store_calibration.txt (tools/scatter_bench, WRITE_SIZE-checked)"}
        # the nominal 8 TB/s beside what a plain device copy reaches on this box (read + write bytes / time)
        try:
            a_ = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b_ = torch.empty_like(a_)
            b_.copy_(a_); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                b_.copy_(a_)
            e1.record(); torch.cuda.synchronize()
            copy_gbs = 5 * 2 * a_.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["roofline"]["device_copy_GBps"] = copy_gbs
            out["roofline"]["frac_of_device_copy"] = achieved / copy_gbs
            del a_, b_
        except Exception:
            pass
        if iso is not None and iso[dom][0] > 0:
            il = max(1, iso[dom][1])
            out["roofline"]["single_lane"] = {"note": "same kernel in one extra untimed step with one Stage-B lane (no other kernel on the chip)",
                                              "launch_ms": iso[dom][0] / il, "achieved": alg[dom] / (iso[dom][0] * 1e-3) / 1e9,
                                              "frac": alg[dom] / (iso[dom][0] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                              "kernel_ms_per_step": {n_: round(iso[n_][0], 3) for n_ in names}}
            if traffic:                                        # the PMC passes are single-lane passes too: THIS pair is a bandwidth
                sl = out["roofline"]["single_lane"]
                sl["traffic_per_launch"] = traffic
                sl["hbm_GBps_from_counters"] = traffic / (sl["launch_ms"] * 1e-3) / 1e9
                sl["frac_from_counters"] = sl["hbm_GBps_from_counters"] / HBM_PEAK_GBS
        if exch is not None:
            kind = gdist.LAST_COMM_KIND or "rccl"
            tot_sent = sum(x_["bytes_sent_per_step"] for x_ in exch)
            out["exchange"] = {"bytes_per_kmer": tot_sent / max(1, valid), "bytes_per_kmer_note": "bytes all ranks sent per step / valid k-mers of the job: 16-byte records, (world - 1) / world of them leave their rank",
                               "transport": ("RCCL grouped ncclSend/ncclRecv inside libgkc_hip.so (gkc_exchange), %d pushes per pass" % n_push) if kind == "rccl" and red_dev == "cuda" else
                               ("%s%s; %d pushes per pass" % ("DRY RUN (GKC_BENCH_BACKEND=%s, the ranks share %d device(s)): " % (dist.get_backend(), torch.cuda.device_count()) if red_dev != "cuda" else "", kind, n_push)),
                               "per_rank": exch}
            if numa_note:
                out["exchange"]["host_placement"] = numa_note + " (rank 0; every rank does the same for its own GPU)"
        if landed is not None:
            out["value_host_landed"] = value                   # (name kept from round 4: now the same number as `value`)
            all_verified.append(landed["sink_ok"])
        if world == 1 and not args.no_host_landed and sink is not None:
            # the other host-landed legs, on the sink `value` used: abundance-min 2 (only the solid records travel) and PCIe at both ends; 5 timed steps each, medians beside the means
            out["host_landed"] = host_landed_leg(c, gkc, step, sync, distinct, n_steps_amin2=5, expect=expect, parts=parts, amins=(2,), sink=sink)
            if "abundance_min_2" in out["host_landed"]:
                out["host_landed"]["abundance_min_2"]["vs_value_device_resident"] = out["host_landed"]["abundance_min_2"]["value"] / value_dev
                if "verified" in out["host_landed"]["abundance_min_2"]:
                    all_verified.append(out["host_landed"]["abundance_min_2"]["verified"])
        sink = None
        if world == 1 and k == 31 and not args.no_cpu_baseline:
            out["config"]["fastq_parse_on_device"] = fastq_parse_leg(c)
        if world == 1 and k == 31 and not args.no_bloom_mphf:
            # BASELINE configs[4] on ONE GPU's share (k=31, abundance-min 2, Bloom solid filter + MPHF + abundance map over the solid k-mers of this workload),
            # a clearly labelled block: count with the solidity window on the device, Bloom of the three kinds (11 bits per k-mer, 7 hashes like the
            # reference's defaults), BooPHF build, populate. The multi-rank combination of these (OR all-reduce, per-level reduce) is in tests/test_gpu_dist.py.
            c.set_solidity(2, 2147483647, 10000)
            step(); sync()
            t0 = time.perf_counter(); step(); sync(); dt_cnt = time.perf_counter() - t0
            ns = c.stats()["kmers_nb_solid"]
            blk = {"workload": "k=31, abundance-min 2, %d reads: count -> Bloom (11 bits / solid k-mer, 7 hashes) -> MPHF + abundance map" % n_reads,
                   "solid_kmers": ns, "count_ms": dt_cnt * 1e3}
            vb = verify_block(c, expect, 2)
            for kind in ("neighbor", "cache", "basic"):
                for rep_ in range(2):                             # the second build: scratch buffers come from the context's allocator, not from hipMalloc
                    bl = gkc.Bloom(c, kind, int(ns * 11.0), 7, k)
                    torch.cuda.synchronize(); t0 = time.perf_counter(); bl.insert_solid(); torch.cuda.synchronize()
                    blk["bloom_%s_ms" % kind] = (time.perf_counter() - t0) * 1e3
                    bl.close()
            # the query side (DebloomMinimizerAlgorithm.cpp:201: contains8 of every solid k-mer), on the device, neighbor kind
            bl = gkc.Bloom(c, "neighbor", int(ns * 11.0), 7, k); bl.insert_solid()
            for rep_ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter(); nq, npos = bl.query_solid(True); torch.cuda.synchronize()
                blk["bloom_contains8_ms"] = (time.perf_counter() - t0) * 1e3
            blk["bloom_contains8_kmers_per_s"] = nq / (blk["bloom_contains8_ms"] * 1e-3); blk["bloom_contains8_neighbours_per_s"] = 8 * blk["bloom_contains8_kmers_per_s"]
            blk["bloom_contains8_set_bits"] = npos
            torch.cuda.synchronize(); t0 = time.perf_counter(); nq1, npos1 = bl.query_solid(False); torch.cuda.synchronize()
            blk["bloom_contains_ms"] = (time.perf_counter() - t0) * 1e3; blk["bloom_contains_kmers_per_s"] = nq1 / (blk["bloom_contains_ms"] * 1e-3)
            assert npos1 == nq1 == ns                             # no false negatives
            bl.close()
            mp_ = gkc.Mphf(c); mp_.close()                      # first build warms the allocator
            torch.cuda.synchronize(); t0 = time.perf_counter(); mp_ = gkc.Mphf(c); torch.cuda.synchronize(); t1 = time.perf_counter()
            amap, above = mp_.abundance_map(); t2 = time.perf_counter()
            blk.update({"mphf_build_ms": (t1 - t0) * 1e3, "mphf_keys_per_s": mp_.size / (t1 - t0), "mphf_bits_per_key": mp_.L.gkc_mphf_save_size(mp_.h) * 8 / max(1, mp_.size),
                        "abundance_map_ms_incl_d2h": (t2 - t1) * 1e3})
            # ... and the structures built on the count: no false negative among the solid k-mers (asserted above), the MPHF is a bijection onto [0, n) — every cell of
            # the abundance map was written exactly once (an abundance index of a solid k-mer is >= 2 > 0), and no abundance lies beyond the table
            vb["bloom_mphf"] = "contains(every solid k-mer) == true: %s; abundance map: %d cells, all written: %s" % (npos1 == nq1 == ns, len(amap), bool((amap != 0).all()))
            vb["verified"] = bool(vb["verified"] and npos1 == nq1 == ns and len(amap) == ns and bool((amap != 0).all()))
            blk["verified"] = vb["verified"]; blk["verification"] = vb; all_verified.append(vb["verified"])
            blk["bloom_and_mphf_ready_ms"] = blk["count_ms"] + blk["bloom_neighbor_ms"] + blk["mphf_build_ms"]
            blk["distinct_kmers_per_s_to_bloom_and_mphf"] = c.stats()["kmers_nb_distinct"] / (blk["bloom_and_mphf_ready_ms"] * 1e-3)
            mp_.close(); del amap
            c.set_solidity(1, 2147483647, 10000)
            out["config"]["bloom_mphf"] = blk
        if world == 1 and k == 31 and not args.no_two_pass:
            # Two passes over the same reads (what a host does when one pass does not fit the HBM; the reference's loop: SortingCountAlgorithm.cpp:672-692), serial and
            # OVERLAPPED: gkc_finish_pass_async detaches pass 0, Stage A of pass 1 (issue-bound) runs beside Stage B of pass 0 (memory-bound). A clearly labelled block.
            c.close()                                           # (its allocator holds the HBM the headline blocks parked; the reads are plain device buffers and stay)
            c2 = gkc.Counter(local); c2.configure(k, m, parts, rep, nb_passes=2)
            def step_serial():
                for ps_ in range(2):
                    c2.begin_pass(ps_)
                    for b_, o_, nr, nb_ in chunks:
                        c2.push_reads_device(b_, o_, nr, nb_)
                    c2.finish_pass()
            def step_overlap():
                for ps_ in range(2):
                    c2.begin_pass(ps_)
                    for b_, o_, nr, nb_ in chunks:
                        c2.push_reads_device(b_, o_, nr, nb_)
                    c2.finish_pass_async()                      # (of pass 1: joins pass 0 first)
                c2.finish_pass_wait()
            blk2 = {"workload": "k=31, %d reads, nb_passes = 2 (pass p keeps the super-k-mers whose minimizer %% 2 == p), %d partitions per pass" % (n_reads, parts)}
            for name_, fn_ in (("serial", step_serial), ("overlapped", step_overlap)):
                fn_(); torch.cuda.synchronize()
                tb_ = {nme: c2.timing(nme) for nme in ("total_stage_a", "total_stage_b")}
                t0 = time.perf_counter()
                for _ in range(2):
                    fn_()
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t0) / 2
                v2 = verify_block(c2, expect, 1); all_verified.append(v2["verified"])
                blk2[name_] = {"ms_per_step": dt2 * 1e3, "value": c2.stats()["kmers_nb_distinct"] / dt2, "steps": 2, "warmup": 1, "verified": v2["verified"],
                               "sum_stage_a_ms": (c2.timing("total_stage_a")[0] - tb_["total_stage_a"][0]) / 2, "sum_stage_b_ms": (c2.timing("total_stage_b")[0] - tb_["total_stage_b"][0]) / 2}
            sa_, sb_ = blk2["serial"]["sum_stage_a_ms"], blk2["serial"]["sum_stage_b_ms"]
            blk2["overlapped_vs_max_of_the_sums"] = blk2["overlapped"]["ms_per_step"] / max(sa_, sb_)
            blk2["note"] = "serial = A0 B0 A1 B1; overlapped = A0 (B0 | A1) B1: the first Stage A and the last Stage B have nothing to hide behind"
            out["config"]["two_pass_overlap"] = blk2
            c = c2                                              # (the later blocks only free the reads through it)
        if world == 1 and k == 31 and not args.no_freq_order:
            # The mode GraphUnitigs forces (GraphUnitigs.cpp:861-870: -minimizer-type 1 -repartition-type 1; Model.hpp:957-973 order): minimizers by increasing frequency
            # of the canonical m-mers of a sample of the reads (RepartitionAlgorithm.cpp:311-380: the first reads; here 10^6 of them, counted on the device by
            # gkc_count_mmers; rank = position in the (count, value) order, unseen m-mers 4^m, the largest m-mer keeps 4^m - 1), same reads, same size. A clearly
            # labelled block; records left in HBM, 5 timed steps after one untimed.
            if c is not None:
                c.close()
            c = gkc.Counter(local)
            n_s = min(n_reads, 1_000_000)
            hb = c.device_to_host(chunks[0][0], n_s * L); ho = np.arange(n_s + 1, dtype=np.uint64) * np.uint64(L)
            t0 = time.perf_counter(); cnts = c.count_mmers(m, hb, ho); t_cnt = time.perf_counter() - t0
            idx = np.nonzero(cnts)[0]; order_ = idx[np.lexsort((idx, cnts[idx]))]
            freq = np.full(4 ** m, 4 ** m, dtype=np.uint32); freq[order_] = np.arange(len(order_), dtype=np.uint32); freq[-1] = 4 ** m - 1
            del hb
            c.configure(k, m, parts, rep, freq_order=freq)
            def step_f():
                c.begin_pass(0)
                for b_, o_, nr, nb_ in chunks:
                    c.push_reads_device(b_, o_, nr, nb_)
                c.finish_pass()
            step_f(); torch.cuda.synchronize()
            bft = {nme: c.timing(nme) for nme in names}
            perf_ = []
            t0 = time.perf_counter()
            for _ in range(5):
                t1 = time.perf_counter(); step_f(); perf_.append((time.perf_counter() - t1) * 1e3)
            torch.cuda.synchronize()
            dtf = (time.perf_counter() - t0) / 5
            sf = c.stats()
            vf = verify_block(c, expect, 1); all_verified.append(vf["verified"])
            ktf = {n_: round((c.timing(n_)[0] - bft[n_][0]) / 5, 3) for n_ in names}
            out["config"]["freq_order"] = {
                "workload": "k=%d, %d reads, minimizers in FREQUENCY order (-minimizer-type 1: what GraphUnitigs forces), m=%d, %d partitions, order table from the canonical m-mer counts of the first %d reads" % (k, n_reads, m, parts, n_s),
                "steps": 5, "warmup": 1, "ms_per_step": dtf * 1e3, "ms_per_step_median": float(np.median(perf_)), "value": sf["kmers_nb_distinct"] / dtf, "unit": "distinct k-mers/s (records left in HBM)",
                "vs_lexi_device_resident": (sf["kmers_nb_distinct"] / dtf) / value_dev, "verified": vf["verified"], "verification": vf,
                "distinct_kmers": sf["kmers_nb_distinct"], "nb_superkmers": sf["nb_superkmers"], "mean_kmers_per_superkmer": sf["kmers_nb_valid"] / max(1, sf["nb_superkmers"]),
                "kernel_ms_per_step": ktf, "scan_count_vs_lexi": ktf["scan_count"] / max(1e-9, ktime_dev["scan_count"][0] / args.steps),
                "stage_a_vs_lexi": ktf["total_stage_a"] / max(1e-9, ktime_dev["total_stage_a"][0] / args.steps),
                "count_mmers_of_the_sample_ms": t_cnt * 1e3}
        if world == 1 and k == 31 and not args.no_skewed:
            # A repeat-rich input at full size: the same generator over a genome with 50 repeat families (~9 % of the bases, ~300 copies each, 0.5 % diverged)
            # and 1 % low-complexity reads (GKC_SYNTH_SKEWED, include/gkc.h) — k-mers at 10^4 copies, sub-buckets of 10^8 identical keys, partitions of very different
            # sizes: what the reference answers with PartitionsCommand.cpp:505-545. Same k, reads, partitions as the headline; records left in HBM; 5 timed steps.
            if c is not None:
                for b_, o_, _, _ in chunks:
                    c.device_free(b_); c.device_free(o_)
                chunks.clear(); c.close()
            c = gkc.Counter(local); c.configure(k, m, parts, rep)
            bs_, os_ = c.synth_reads_device(2, n_reads, L, genome, 10000, profile=1)
            exp_s = input_checksum(c, [(bs_, os_, n_reads, n_bases)])
            def step_s():
                c.begin_pass(0); c.push_reads_device(bs_, os_, n_reads, n_bases); c.finish_pass()
            step_s(); torch.cuda.synchronize()
            bst = {nme: c.timing(nme) for nme in names}
            pers_ = []
            t0 = time.perf_counter()
            for _ in range(5):
                t1 = time.perf_counter(); step_s(); pers_.append((time.perf_counter() - t1) * 1e3)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / 5
            ss = c.stats()
            vs = verify_block(c, exp_s, 1); all_verified.append(vs["verified"])
            kts = {n_: round((c.timing(n_)[0] - bst[n_][0]) / 5, 3) for n_ in names}
            out["config"]["skewed"] = {
                "workload": "k=%d, %d reads over a genome with repeat families + 1 %% low-complexity reads (GKC_SYNTH_SKEWED), m=%d, %d partitions" % (k, n_reads, m, parts),
                "steps": 5, "warmup": 1, "ms_per_step": dts * 1e3, "ms_per_step_median": float(np.median(pers_)), "value": ss["kmers_nb_distinct"] / dts, "unit": "distinct k-mers/s (records left in HBM)",
                "valid_kmers_per_s": ss["kmers_nb_valid"] / dts, "ms_per_step_vs_uniform": dts * 1e3 / ms_step_dev, "verified": vs["verified"], "verification": vs,
                "distinct_kmers": ss["kmers_nb_distinct"], "nb_superkmers": ss["nb_superkmers"], "oversize_buckets": ss.get("oversize_buckets"), "kernel_ms_per_step": kts}
            c.device_free(bs_); c.device_free(os_)
        if world == 1 and k == 31 and not args.no_k63:
            # BASELINE configs[3] (k=63, LargeInt<2> 128-bit k-mer path, same reads per GPU) timed by the same driver run: a second, clearly labelled block —
            # `value` above stays the k=31 figure of configs[1]
            for b_, o_, _, _ in chunks:
                c.device_free(b_); c.device_free(o_)
            chunks.clear()
            c.close(); c = None
            c63 = gkc.Counter(local)
            # 16-byte keys: a wave's registers hold half as many, so the partitions are half the size (about 1.5e6 k-mers: 8192 sub-buckets of ~180 keys);
            # measured at 10^8 reads: 4096 partitions 410-434 ms, 8192: 403, 16384: 395 (Stage A +8 ms for the second level, the sort tiers -40)
            p63 = int(min(32768, max(64, 2 ** int(np.ceil(np.log2(max(1, n_reads * (L - 63 + 1) / 1.5e6)))))))
            c63.configure(63, m, p63, repart_for_bench(m, p63))
            b63, o63 = c63.synth_reads_device(2, n_reads, L, genome, 10000)
            def step63():
                c63.begin_pass(0); c63.push_reads_device(b63, o63, n_reads, n_bases); c63.finish_pass()
            step63(); torch.cuda.synchronize()
            b63t = {nme: c63.timing(nme) for nme in names}
            per63 = []
            t0 = time.perf_counter()
            for _ in range(5):
                t1 = time.perf_counter(); step63(); per63.append((time.perf_counter() - t1) * 1e3)
            torch.cuda.synchronize()
            dt63 = (time.perf_counter() - t0) / 5
            s63 = c63.stats()
            v63 = verify_block(c63, input_checksum(c63, [(b63, o63, n_reads, n_bases)]), 1); all_verified.append(v63["verified"])
            kt63 = {nme: ((c63.timing(nme)[0] - b63t[nme][0]), (c63.timing(nme)[1] - b63t[nme][1])) for nme in names}
            alg63, dom63, dom63_ms, ach63, ln63 = kernel_roofline(kt63, 5, n_bases, s63, 63, s63["kmers_nb_valid"], s63["kmers_nb_distinct"])
            out["config"]["k63"] = {"workload": "k=63, %d synthetic 150 bp reads, single-pass count, m=%d, %d partitions (BASELINE configs[3])" % (n_reads, m, p63),
                                    "steps": 5, "warmup": 1, "ms_per_step": dt63 * 1e3, "ms_per_step_median": float(np.median(per63)), "value": s63["kmers_nb_distinct"] / dt63, "unit": "distinct k-mers/s (records left in HBM; host-landed: value_host_landed)", "dtype": "u128",
                                    "verified": v63["verified"], "verification": v63,
                                    "valid_kmers": s63["kmers_nb_valid"], "distinct_kmers": s63["kmers_nb_distinct"], "valid_kmers_per_s": s63["kmers_nb_valid"] / dt63,
                                    "kernel_ms_per_step": {n_: round(kt63[n_][0] / 5, 3) for n_ in names},
                                    "roofline": {"bound": "hbm", "kernel": dom63, "achieved": ach63, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach63 / HBM_PEAK_GBS,
                                                 "traffic": None, "launches_per_step": int(ln63), "launch_ms": dom63_ms / ln63,
                                                 "algorithmic_bytes_per_launch": alg63[dom63] / ln63, "stage_b_lanes": int(os.environ.get("GKC_STAGEB_LANES", "2"))}}
            if not args.no_host_landed:
                # SURVEY 8(d)'s wall for configs[3]: every distinct 63-mer's 32-byte Count record in page-locked host memory (abundance-min 1): 147 GB of records, 16 bytes
                # per record on the link (csrc/gkc_sink.hip, 16-byte keys); one untimed + five timed steps (mean and median), verified like the others
                def sync63():
                    torch.cuda.synchronize()
                try:
                    hl63 = host_landed_leg(c63, gkc, step63, sync63, s63["kmers_nb_distinct"], n_steps=5, expect=input_checksum(c63, [(b63, o63, n_reads, n_bases)]), parts=p63,
                                           amins=(1,), host_to_host=False)
                except Exception as e:      # noqa
                    hl63 = {"error": repr(e)}
                out["config"]["k63"]["host_landed"] = hl63
                if "abundance_min_1" in hl63:
                    out["config"]["k63"]["value_host_landed"] = hl63["abundance_min_1"]["value"]
                    if "verified" in hl63["abundance_min_1"]:
                        all_verified.append(hl63["abundance_min_1"]["verified"])
            c63.device_free(b63); c63.device_free(o63); c63.close()
        if world == 1 and k == 31 and not args.no_share_of_8:
            # BASELINE configs[2]'s per-GPU share on THIS GPU: 1.25e8 reads of the 10^9-read stream, the partition count the 8-GPU run uses (-> two-level Stage A),
            # 4 pushes each followed by gkc_exchange through a one-rank RCCL communicator (the planning, narrowing and import code of the multi-GPU path; no peer,
            # so no byte crosses xGMI). All 32768 partitions are counted here (8 ranks would each count 4096 partitions of 8x the k-mers: the same number of k-mers);
            # the reads are drawn at 30x over a genome of their own, so that the k-mer multiplicities are those an owner rank sees.
            if c is not None:
                for b_, o_, _, _ in chunks:
                    c.device_free(b_); c.device_free(o_)
                chunks.clear(); c.close(); c = None
            from gatb_core_amd import dist as gdist          # noqa
            W8 = 8
            p8 = int(min(32768, 2 ** int(np.floor(np.log2(SHARE_OF_8 * (L - k + 1) * W8 / 3.0e6) + 0.5))))
            c8 = gkc.Counter(local); c8.configure(k, m, p8, repart_for_bench(m, p8))
            g8 = SHARE_OF_8 * L // 30                          # 30x over a genome of their own: Stage B sees the multiplicities an owner rank sees in the 8-GPU run
                                                             # (this share of the global stream alone would cover the 5e9-base genome 3.75x: nearly every k-mer distinct)
            np8 = args.pushes; per8 = (SHARE_OF_8 + np8 - 1) // np8
            ch8 = []
            for i in range(np8):
                nr = min(per8, SHARE_OF_8 - i * per8)
                if nr > 0:
                    b_, o_ = c8.synth_reads_device(2, nr, L, g8, 10000, first_read=i * per8)
                    ch8.append((b_, o_, nr, nr * L))
            r8 = gdist.DistributedCounter(c8, 0, 1, p8)
            def step8():
                c8.begin_pass(0)
                for b_, o_, nr, nb_ in ch8:
                    c8.push_reads_device(b_, o_, nr, nb_); r8.exchange()
                c8.finish_pass()
            step8(); torch.cuda.synchronize()
            b8t = {nme: c8.timing(nme) for nme in names}
            t0 = time.perf_counter()
            for _ in range(3):
                step8()
            torch.cuda.synchronize()
            dt8 = (time.perf_counter() - t0) / 3
            s8 = c8.stats(); cs8 = r8.stats()
            v8 = verify_block(c8, input_checksum(c8, ch8), 1); all_verified.append(v8["verified"])       # 32768 partitions, 4 pushes, two-level Stage A, the exchange's narrowing / import path
            out["config"]["share_of_8"] = {
                "verified": v8["verified"], "verification": v8,
                "workload": "BASELINE configs[2] per-GPU share on one GPU: k=31, %d of 10^9 reads, %d partitions (two-level Stage A), %d pushes + gkc_exchange (one-rank RCCL communicator)" % (SHARE_OF_8, p8, len(ch8)),
                "steps": 3, "warmup": 1, "ms_per_step": dt8 * 1e3, "value": s8["kmers_nb_distinct"] / dt8, "unit": "distinct k-mers/s (this GPU's share)",
                "x8_if_the_exchange_were_free": 8 * s8["kmers_nb_distinct"] / dt8,
                "valid_kmers": s8["kmers_nb_valid"], "distinct_kmers": s8["kmers_nb_distinct"],
                "kernel_ms_per_step": {n_: round((c8.timing(n_)[0] - b8t[n_][0]) / 3, 3) for n_ in names},
                "exchange": {"exchanges_per_step": cs8["n_exchanges"] / 4, "ms_host_per_step": cs8["ms_host"] / 4, "ms_transfer_per_step": cs8["ms_transfer"] / 4,
                             # what an owner-routed k-mer costs on xGMI: the exchange ships the device's fixed 16-byte records; the reference's wire format (Model.hpp:1386-1471:
                             # 1 byte + ceil((k + nbK - 1) / 4) per super-k-mer) would be 25 % fewer bytes for one more pass over the records
                             "bytes_per_kmer": s8["nb_superkmers"] * 16 / max(1, s8["kmers_nb_valid"]),
                             "bytes_per_kmer_reference_wire_format": (s8["nb_superkmers"] * 1.375 + (s8["kmers_nb_valid"] + s8["nb_superkmers"] * (k - 1)) / 4.0) / max(1, s8["kmers_nb_valid"]),
                             "sent_fraction_at_8_ranks": 7.0 / 8.0}}
            for b_, o_, _, _ in ch8:
                c8.device_free(b_); c8.device_free(o_)
            c8.close()
        if not args.no_cpu_baseline and world == 1:
            cb = c if c is not None else gkc.Counter(local)
            os.sched_setaffinity(0, affinity0)                    # (every core of the host for the CPU baseline)
            ref_b = cpu_baseline_reference(cb, k)
            cp = int(min(parts, 4096))
            port_b = cpu_baseline(cb, k, m, cp, repart_for_bench(m, cp))
            if ref_b is not None and "value" in ref_b:
                ref_b["port"] = port_b                         # the C restatement (oracle) beside it, clearly labelled
                out["cpu_baseline"] = ref_b
            else:
                if ref_b is not None:
                    port_b["reference_attempt"] = ref_b
                else:
                    port_b["reference_attempt"] = "integration/_build/ref/dbgh5 absent (integration/build_reference.sh builds it from /root/reference)"
                out["cpu_baseline"] = port_b
        elif world == 1:
            out["cpu_baseline"] = None
        out["all_blocks_verified"] = bool(all(all_verified))
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        if not all(all_verified):
            sys.stderr.write("bench.py: a timed block left results that are NOT the count of its input (see the `verified` fields)\n")
            bad_rc = 3
        else:
            bad_rc = 0
    else:
        bad_rc = 0
    if c is not None:
        for b_, o_, _, _ in chunks:
            c.device_free(b_); c.device_free(o_)
    if use_dist:
        t_rc = torch.tensor([bad_rc], device=red_dev); dist.all_reduce(t_rc, op=dist.ReduceOp.MAX); bad_rc = int(t_rc.item())      # a red `verified` fails every rank
        dist.barrier(); dist.destroy_process_group()
    if bad_rc:
        sys.exit(bad_rc)


if __name__ == "__main__":
    main()
