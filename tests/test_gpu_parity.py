"""Parity tests proper: the HIP path, called through the C-ABI (ctypes binding), against the CPU oracle on the same
seeded inputs. Bit-exact: integer/byte work. Run on the GPU box with `pytest -m gpu`."""
import ctypes
import math
import os

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.util import naive_counts, revcomp_int, simple_repart, synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gkc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return ge.load().gkc


def device_vs_oracle(gkc, reads, k, m, parts, passes=1, freq=None, amin=1, amax=2147483647, batches=1, rep=None,
                     histo_max=10000):
    bases, offs = gko.pack_reads(reads)
    rep = simple_repart(m, parts) if rep is None else rep
    ref = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes, freq_order=freq, abundance_min=amin, abundance_max=amax,
                  histo_max=histo_max)
    c = gkc.Counter(0)
    c.set_solidity(amin, amax, histo_max)
    c.configure(k, m, parts, rep, nb_passes=passes, freq_order=freq)
    n = len(reads)
    for ps in range(passes):
        c.begin_pass(ps)
        for b in range(batches):
            r0, r1 = n * b // batches, n * (b + 1) // batches
            sub_off = offs[r0:r1 + 1] - offs[r0]
            c.push_reads(bases[int(offs[r0]):int(offs[r1])], sub_off)
        c.finish_pass()
    for ps in range(passes):
        for p in range(parts):
            d = p + ps * parts
            lo, hi, ab = c.partition(ps, p)
            rlo, rhi, rab = ref.part(d)
            assert np.array_equal(lo, rlo), (k, m, "dataset", d, len(lo), len(rlo))
            assert np.array_equal(hi, rhi) and np.array_equal(ab, rab), (k, m, "dataset", d)
            assert np.array_equal(c.partition_records(ps, p), ref.part_records(d))      # exact Count memory layout
            ns, nd, nk = c.partition_info(ps, p)
            assert nk == ref.part_stats(d)[0]
    st = c.stats()
    for key in ("kmers_nb_valid", "kmers_nb_invalid", "kmers_nb_distinct", "kmers_nb_solid", "nb_sequences"):
        assert st[key] == ref.stats[key], (key, st[key], ref.stats[key])
    assert np.array_equal(c.histogram(), ref.histogram())
    return c, ref


@pytest.mark.parametrize("k,m,parts", [(21, 10, 8), (31, 10, 16), (15, 7, 4), (9, 5, 3), (27, 8, 8), (5, 3, 2), (31, 12, 64)])
def test_counts_bit_exact_64bit_keys(gkc, k, m, parts):
    reads = synth_reads(3000, 20000, 150, seed=k, n_rate=0.002, ragged=True)
    device_vs_oracle(gkc, reads, k, m, parts)


@pytest.mark.parametrize("k,m,parts", [(63, 10, 8), (33, 9, 4), (47, 11, 16), (32, 10, 4)])
def test_counts_bit_exact_128bit_keys(gkc, k, m, parts):
    reads = synth_reads(2000, 20000, 150, seed=k, n_rate=0.002, ragged=True)
    device_vs_oracle(gkc, reads, k, m, parts)


def test_config1_plumbing_case(gkc):
    """BASELINE configs[0]: 10k synthetic 150 bp reads, k=21, abundance-min 2"""
    reads = synth_reads(10000, 50000, 150, seed=1)
    c, ref = device_vs_oracle(gkc, reads, 21, 10, 8, amin=2)
    assert ref.stats["kmers_nb_solid"] < ref.stats["kmers_nb_distinct"]


def test_multi_batch_multi_pass_and_solidity_window(gkc):
    reads = synth_reads(4000, 10000, 150, seed=4, n_rate=0.001, ragged=True)
    device_vs_oracle(gkc, reads, 25, 9, 8, passes=3, batches=4, amin=2, amax=40, histo_max=30)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_random_configurations(gkc, seed):
    """random (k, m, partitions, passes, pushes, solidity window, read shapes): every dataset, statistic and the histogram bit-exact"""
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(5, 64))
    m = int(rng.integers(2, min(k - 1, 11) + 1))
    parts = int(rng.integers(1, 40))
    passes = int(rng.integers(1, 4))
    batches = int(rng.integers(1, 4))
    amin = int(rng.integers(1, 4)); amax = int(rng.choice([2147483647, 5, 60]))
    if amax < amin:
        amax = amin
    histo_max = int(rng.choice([10000, 7, 100]))
    n_reads = int(rng.integers(50, 1500)); glen = int(rng.integers(300, 20000)); rlen = int(rng.integers(max(k, 20), 220))
    reads = synth_reads(n_reads, glen, rlen, seed=seed, sub_rate=float(rng.choice([0.0, 0.01, 0.05])), n_rate=float(rng.choice([0.0, 0.002])),
                        ragged=bool(rng.integers(0, 2)))
    freq = None
    if seed % 3 == 0:                                      # frequency order of the m-mers of a sample of the reads (RepartitionAlgorithm.cpp:311-492)
        L = gko.lib()
        counts = np.zeros(4 ** m, np.uint32)
        for r in reads[: max(10, n_reads // 4)]:
            L.gko_count_mmers(r, len(r), m, counts)
        freq = np.zeros(4 ** m, np.uint32)
        L.gko_freq_order_from_counts(m, counts, freq)
    device_vs_oracle(gkc, reads, k, m, parts, passes=passes, batches=batches, amin=amin, amax=amax, histo_max=histo_max, freq=freq)


def test_frequency_order_minimizers(gkc):
    """-minimizer-type 1 (what GraphUnitigs forces, GraphUnitigs.cpp:861-870)"""
    reads = synth_reads(3000, 15000, 150, seed=6, n_rate=0.001)
    k, m, parts = 31, 8, 8
    L = gko.lib()
    counts = np.zeros(4 ** m, np.uint32)
    for r in reads[:300]:
        L.gko_count_mmers(r, len(r), m, counts)
    freq = np.zeros(4 ** m, np.uint32)
    L.gko_freq_order_from_counts(m, counts, freq)
    device_vs_oracle(gkc, reads, k, m, parts, freq=freq)
    device_vs_oracle(gkc, reads[:1000], 63, m, parts, freq=freq)


@pytest.mark.parametrize("cmax", [4, 7])
def test_two_level_scan_many_partitions(gkc, monkeypatch, cmax):
    """above SCAN_COARSE_MAX partitions Stage A scans into partition groups and splits the groups by recomputing each record's
    minimizer (k_refine_*); forced here with a tiny group limit: lexicographic and frequency order, 8- and 16-byte keys, 2 passes"""
    monkeypatch.setenv("GKC_SCAN_COARSE_MAX", str(cmax))
    reads = synth_reads(3000, 15000, 150, seed=16, n_rate=0.002, ragged=True)
    device_vs_oracle(gkc, reads, 31, 10, 64)
    device_vs_oracle(gkc, reads[:1500], 63, 9, 37, batches=2)
    device_vs_oracle(gkc, reads[:1500], 21, 8, 30, passes=2)
    m = 8
    L = gko.lib()
    counts = np.zeros(4 ** m, np.uint32)
    for r in reads[:300]:
        L.gko_count_mmers(r, len(r), m, counts)
    freq = np.zeros(4 ** m, np.uint32)
    L.gko_freq_order_from_counts(m, counts, freq)
    device_vs_oracle(gkc, reads[:1500], 31, m, 50, freq=freq)
    device_vs_oracle(gkc, reads[:800], 45, m, 50, freq=freq)
    # up to 16 partitions per group the scan leaves the partition's low bits in the record's spare bits (no recomputation); beyond, and with the switch, k_refine_count
    # recomputes the minimizer: 200 partitions in <= 4 groups are 64 per group
    device_vs_oracle(gkc, reads[:1500], 31, 10, 200)
    monkeypatch.setenv("GKC_REFINE_RECOMPUTE", "1")
    device_vs_oracle(gkc, reads[:1500], 31, 10, 64)
    device_vs_oracle(gkc, reads[:800], 63, 9, 37, batches=2)


def test_reference_known_answers_through_the_device(gkc, ref_vectors):
    """TestDSK.cpp:147-305 run through the HIP path"""
    v = ref_vectors["dsk_check1"]
    for name, k, nks, expected in v["cases"]:
        m = min(k - 1, 10)
        bases, offs = gko.pack_reads(v[name])
        c = gkc.Counter(0); c.set_solidity(nks); c.configure(k, m, 4, simple_repart(m, 4))
        c.count(bases, offs)
        assert c.stats()["kmers_nb_solid"] == expected, (name, k, nks)
    v = ref_vectors["dsk_check2"]
    bases, offs = gko.pack_reads([v["seq"]])
    c = gkc.Counter(0); c.configure(31, 10, 4, simple_repart(10, 4)); c.count(bases, offs)
    got = c.all_counts()
    assert sorted(got) == sorted(v["values"]) and sum(got) & (2 ** 64 - 1) == v["checksum"]


def test_edge_cases(gkc):
    k, m, parts = 31, 10, 4
    rep = simple_repart(m, parts)
    # empty input, reads shorter than k, all-N reads, zero-length reads
    for reads in ([], ["ACGT"], ["N" * 200], ["", "ACGTACGTAC", ""], ["A" * 30], ["ACGT" * 8 + "N" + "ACGT" * 8]):
        device_vs_oracle(gkc, reads, k, m, parts, rep=rep)
    # one long sequence spanning many scan tiles, with lower-case and N
    rng = np.random.default_rng(3)
    long = "".join(rng.choice(list("ACGTacgt"), 50000)) + "N" + "".join(rng.choice(list("ACGT"), 30000))
    device_vs_oracle(gkc, [long, long[100:9000]], k, m, parts, rep=rep)
    device_vs_oracle(gkc, [long], 63, m, parts, rep=rep)


def test_superkmer_cap_at_every_thread_phase(gkc):
    """A4: the cap (a run of one minimizer value cut every maxs k-mers) only bites on repeated minimizers; Stage A decides per wave whether any thread can
    reach it and otherwise takes run ends from bit masks. Runs of a repeated minimizer (poly-A, dinucleotide and 7-mer repeats) entered at every offset
    relative to the 16-position threads and to the scan tile, cut by N, by read ends and by ordinary sequence, for 16- and 32-byte records, must give the
    oracle's records — including the partition statistics (records per partition depend on where the cap cuts)"""
    rng = np.random.default_rng(41)
    def rnd(n): return "".join(rng.choice(list("ACGT"), n))
    reads = []
    for off in range(0, 50):
        reads.append(rnd(off) + "A" * (90 + off) + rnd(37))
        reads.append(rnd(off) + "AC" * 60 + "N" + "T" * 70)
        reads.append(rnd(3 * off) + "ACGTTGC" * 25 + rnd(off))
    reads.append(rnd(8192 - 40) + "A" * 300 + rnd(50))                       # a capped run across a scan-tile boundary
    reads.append("A" * 9000)                                                 # one read, one run, many tiles
    for (k, m, parts) in [(31, 10, 4), (21, 7, 2), (63, 10, 2), (41, 9, 3)]:
        c, ref = device_vs_oracle(gkc, reads, k, m, parts)
        assert c.stats()["nb_superkmers"] >= ref.stats["nb_superkmers"]          # tile boundaries may add cuts, the cap never removes one


def test_oversize_buckets_low_complexity(gkc):
    """massively repeated k-mers (poly-A, tandem repeats) overflow the LDS sort and take the global-memory path"""
    reads = ["A" * 150] * 3000 + ["AC" * 75] * 2000 + synth_reads(500, 5000, 150, seed=8)
    c, ref = device_vs_oracle(gkc, reads, 31, 10, 4)
    assert c.stats()["oversize_buckets"] > 0
    c, ref = device_vs_oracle(gkc, reads[:4000], 41, 10, 2)
    assert c.stats()["oversize_buckets"] > 0


@pytest.mark.parametrize("k", [63, 47, 33])
def test_16_byte_keys_that_share_their_top_word(gkc, k):
    """16-byte keys whose TOP 64-bit words are equal and whose low words differ (reads that share their first 100+ bases and differ in the last few nucleotides: what
    every sequencing error in the second half of a k-mer produces): several neighbours per top word, in shuffled arrival order, at every sort tier (copies from 1 to
    600). Round 6 tried ordering such keys by the top word alone (one compare per exchange instead of three, re-sort on ties): bit-exact and SLOWER on reads with
    errors — every sub-bucket holds such neighbours, so every sub-bucket sorted twice (k = 63: 282 -> 390 ms of Stage B); the test stays."""
    rng = np.random.default_rng(k)
    stem = "".join("ACGT"[i] for i in rng.integers(0, 4, 150))
    reads = []
    for stem_ in (stem, "A" * 150, "ACGGT" * 30):
        for i in range(400):
            tail = "".join("ACGT"[j] for j in rng.integers(0, 4, 12))
            reads += [(stem_[:138] + tail).encode()] * int(1 + (i % 7 == 0) * rng.integers(1, 600))
    order = rng.permutation(len(reads)); reads = [reads[i] for i in order]
    reads += synth_reads(2000, 12000, 150, seed=k, n_rate=0.001)
    device_vs_oracle(gkc, reads, k, 10, 3)
    device_vs_oracle(gkc, reads, k, 8, 1, amin=2, amax=500)


@pytest.mark.parametrize("k,m,parts,n_reads,skew,smin", [(31, 10, 1, 100_000, False, 1), (31, 9, 3, 160_000, False, 1_000_000), (63, 10, 2, 150_000, False, 1), (21, 8, 2, 120_000, False, 3_000_000),
                                                         (31, 10, 5, 110_000, True, 2_000_000), (41, 10, 4, 130_000, True, 1), (63, 11, 3, 90_000, True, 1_500_000),
                                                         (31, 10, 5, 110_000, 0.4, 3_000_000), (30, 9, 6, 120_000, 0.35, 2_500_000)])
def test_sliced_partitions_few_huge(gkc, monkeypatch, k, m, parts, n_reads, skew, smin):
    """Several workgroups per partition: a partition far beyond the planned size is expanded by up to 16 workgroups, each taking a share of its
    records, with its own pair range + odd-key slot inside every sub-bucket (SliceTables in csrc/gkc_count.hip). GKC_SLICE_MIN makes partitions of 3e6 .. 1.2e7
    k-mers take the path that partitions beyond 1.6e7 k-mers take by default (16 slices with 1; 2 .. 8 with the larger thresholds; `skew`: one huge partition
    beside small unsliced ones in the same batch; a fraction: the sliced partition is a minority of the batch, which keeps its record deduplication). Records, statistics and histogram against the oracle; N's, ragged reads, low-complexity reads (split roots,
    giants) and copied reads included; then the same input with the slices switched off."""
    monkeypatch.setenv("GKC_SLICE_MIN", str(smin))
    reads = synth_reads(n_reads, n_reads * 5, 150, seed=3 * k + parts, n_rate=0.0005, ragged=True)
    reads += [b"A" * 150] * 200 + [b"ACACACACAC" * 15] * 200 + [b"G" * 150] * 30 + [reads[5]] * 40
    rep = simple_repart(m, parts)
    if skew is True:
        rep = (rep.astype(np.uint32) % (8 * (parts - 1))).astype(np.uint16)
        rep = np.where(rep < parts - 1, rep + 1, 0).astype(np.uint16)             # partition 0 takes 7/8 of the minimizers
    elif skew:                                                                    # partition 0 takes the fraction `skew` of the minimizers: the only sliced partition of a batch
        u = np.random.default_rng(11).random(len(rep))                            # that keeps its record deduplication (less than half of its k-mers are in sliced partitions)
        rep = np.where(u < skew, 0, 1 + (rep.astype(np.uint32) % (parts - 1))).astype(np.uint16)
    device_vs_oracle(gkc, reads, k, m, parts, rep=rep)
    device_vs_oracle(gkc, reads, k, m, parts, rep=rep, batches=3, amin=2, amax=40)        # several segments (every slice takes its share of each), a solidity window
    monkeypatch.setenv("GKC_SLICES", "0")
    device_vs_oracle(gkc, reads[:20000], k, m, parts, rep=rep)


def test_superkmer_buckets_match_partition_contract(gkc):
    """A5/A6: the device buckets, re-encoded in the reference wire format and decoded by the oracle (B1), hold exactly
    the k-mers the oracle's partitions hold (tile boundaries may split a super-k-mer; the k-mer multiset is the contract)"""
    reads = synth_reads(1500, 10000, 150, seed=12, n_rate=0.002, ragged=True)
    for (k, m, parts) in [(31, 10, 8), (63, 10, 4), (21, 8, 4)]:
        bases, offs = gko.pack_reads(reads)
        rep = simple_repart(m, parts)
        ref = gko.Dsk(bases, offs, k, m, parts, rep)
        c = gkc.Counter(0); c.configure(k, m, parts, rep); c.begin_pass(0); c.push_reads(bases, offs)
        L = gko.lib()
        for p in range(parts):
            blob, nsk, nk = c.partition_superkmers(p)
            assert nk == ref.part_stats(p)[0]
            got = {}
            off = 0; lo = np.zeros(256, np.uint64); hi = np.zeros(256, np.uint64); nbk = ctypes.c_uint(0)
            nrec = 0
            while off < len(blob):
                off += L.gko_superkmer_decode(blob[off:].ctypes.data, k, lo, hi, ctypes.byref(nbk))
                nrec += 1
                for a, b in zip(lo[:nbk.value].tolist(), hi[:nbk.value].tolist()):
                    key = (b << 64) | a; got[key] = got.get(key, 0) + 1
            assert nrec == nsk and off == len(blob)
            rlo, rhi, rab = ref.part(p)
            exp = {(int(b) << 64) | int(a): int(cn) for a, b, cn in zip(rlo, rhi, rab)}
            assert got == exp
        c.finish_pass()


@pytest.mark.parametrize("k", [11, 31, 41, 63])
def test_bloom_bit_identical(gkc, k):
    rng = np.random.default_rng(k)
    keys = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(3000)]
    others = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(2000)]
    c = gkc.Counter(0)
    for kind in ("basic", "cache", "neighbor"):
        for (bits, nh) in [(3000 * 11, 7), (1 << 15, 4), (1000, 1)]:
            ob = gko.Bloom(kind, bits, nh, k); ob.insert(keys)
            db = gkc.Bloom(c, kind, bits, nh, k); db.insert(keys)
            assert db.nbytes == ob.nbytes and db.bitsize == ob.bitsize
            assert np.array_equal(db.array(), ob.array()), (kind, bits, nh)
            assert np.array_equal(db.contains(keys + others), ob.contains(keys + others))
            if kind == "neighbor":
                assert np.array_equal(db.contains8(others[:500] + keys[:500]), ob.contains8(others[:500] + keys[:500]))
            db.close()


@pytest.mark.parametrize("k", [31, 63])
def test_bloom_region_build_bit_identical(gkc, k):
    """the filters are built region by region in LDS (2^20 bits per region; block-coherent kinds: shared fringes ORed atomically; basic:
    every position bucketed on its own): several regions, two insert calls into the same filter"""
    rng = np.random.default_rng(100 + k)
    keys = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(120000)]
    c = gkc.Counter(0)
    for kind in ("basic", "cache", "neighbor"):
        bits = 5_000_000 + 12345
        ob = gko.Bloom(kind, bits, 7, k); ob.insert(keys)
        db = gkc.Bloom(c, kind, bits, 7, k); db.insert(keys[:70000]); db.insert(keys[70000:])
        assert np.array_equal(db.array(), ob.array()), kind
        db.close()


def test_bloom_of_solid_kmers_and_cfp_known_answer(gkc, ref_vectors):
    """BloomAlgorithm::execute sizing + TestDebloom.cpp:133-137's 20 critical false positives, with the device filter"""
    v = ref_vectors["debloom_k11"]; k = v["k"]
    bases, offs = gko.pack_reads([v["seq"]])
    c = gkc.Counter(0); c.configure(k, v["m"], 4, simple_repart(v["m"], 4)); c.count(bases, offs)
    solid = c.all_counts()
    lg2 = math.log(2)
    nbits = np.float32(math.log(16 * k * (lg2 * lg2)) / (lg2 * lg2))
    size = int(np.float32(len(solid)) * nbits); nb_hash = int(math.floor(0.7 * float(nbits)))
    bl = gkc.Bloom(c, "basic", size, nb_hash, k)
    bl.insert_solid()
    mask = 4 ** k - 1
    cand = set()
    for x in solid:
        for y in (x, revcomp_int(x, k)):
            for j in range(4):
                n = ((y << 2) | j) & mask
                cand.add(min(n, revcomp_int(n, k)))
    cand = sorted(cand - set(solid))
    hits = bl.contains(cand)
    assert {x for x, h in zip(cand, hits) if h} == set(v["cfp"])


@pytest.mark.parametrize("k", [31, 41])
def test_bloom_query_of_the_solid_set_on_the_device(gkc, k):
    """gkc_bloom_query_solid (the reference's call site: DebloomMinimizerAlgorithm.cpp:201, contains8 of every solid k-mer; Bloom.hpp:645-811): queried where the
    records lie, in dataset order; set result bits == the oracle's contains8 / contains over the same k-mers"""
    reads = synth_reads(3000, 20000, 150, seed=21 + k)
    bases, offs = gko.pack_reads(reads)
    m, parts = 9, 5
    c = gkc.Counter(0); c.configure(k, m, parts, simple_repart(m, parts)); c.set_solidity(2, 2147483647, 10000); c.count(bases, offs)
    solid = c.all_counts()
    keys = list(solid.keys()) if isinstance(solid, dict) else list(solid)
    assert len(keys) > 1000
    bits = len(keys) * 11
    ob = gko.Bloom("neighbor", bits, 7, k); ob.insert(keys)
    db = gkc.Bloom(c, "neighbor", bits, 7, k); db.insert_solid()
    assert np.array_equal(db.array(), ob.array())
    nq, npos = db.query_solid(neighbors8=True)
    exp8 = ob.contains8(keys)
    assert nq == len(keys) and npos == int(sum(bin(int(x)).count("1") for x in exp8))
    nq1, npos1 = db.query_solid(neighbors8=False)
    assert nq1 == len(keys) and npos1 == len(keys)                      # no false negatives
    # round 5: contains8 BY REGION of the array (what 2e6 k-mers and more get; the threshold lowered here): the same answers k-mer by k-mer, and over the resident set
    exp8 = np.asarray(exp8, dtype=np.uint8)
    assert np.array_equal(db.contains8(keys), exp8)                     # (the gathers)
    os.environ["GKC_BLOOM_QUERY_REGIONS_MIN"] = "1"
    try:
        assert np.array_equal(db.contains8(keys), exp8)
        nq2, npos2 = db.query_solid(neighbors8=True)
        assert (nq2, npos2) == (nq, npos)
    finally:
        del os.environ["GKC_BLOOM_QUERY_REGIONS_MIN"]
    db.close()


def test_synth_generator_and_checksum_property(gkc):
    """device generator == numpy twin; independent checksum kernel == checksum of the counted records; sums match"""
    c = gkc.Counter(0)
    seed, n, L, G = 7, 20000, 150, 100000
    db, do = c.synth_reads_device(seed, n, L, G, 10000)
    hb = c.device_to_host(db, n * L)
    nb, no = gkc.synth_reads_np(seed, n, L, G, 10000)
    assert np.array_equal(hb, nb)
    for (k, m) in [(31, 10), (63, 10)]:
        parts = 32
        rep = simple_repart(m, parts)
        c.configure(k, m, parts, rep)
        cs, nv = c.kmer_checksum_device(db, do, n, n * L)
        c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
        rcs, rsum = c.result_checksum()
        assert (rcs, rsum) == (cs, nv)
        st = c.stats()
        assert st["kmers_nb_valid"] == nv == n * (L - k + 1)
        ref = gko.Dsk(nb, no, k, m, parts, rep)
        for p in range(parts):
            lo, hi, ab = c.partition(0, p)
            rlo, rhi, rab = ref.part(p)
            assert np.array_equal(lo, rlo) and np.array_equal(hi, rhi) and np.array_equal(ab, rab)
            keys = (hi.astype(object) << 64) | lo.astype(object)
            assert all(keys[i] < keys[i + 1] for i in range(len(keys) - 1))
    c.device_free(db); c.device_free(do)


def test_skewed_generator_and_counts(gkc):
    """GKC_SYNTH_SKEWED (repeat families in the genome, 1 % low-complexity reads): the device generator == its numpy twin; the counts of such reads — giant sub-buckets of
    one k-mer (poly-A ...), k-mers at thousands of copies — == the oracle's, datasets and histogram, at k = 31 and 63, lexicographic and frequency order"""
    c = gkc.Counter(0)
    seed, n, L, G = 5, 40000, 150, 400000
    db, do = c.synth_reads_device(seed, n, L, G, 10000, profile=1)
    hb = c.device_to_host(db, n * L)
    nb, no = gkc.synth_reads_np(seed, n, L, G, 10000, profile=1)
    assert np.array_equal(hb, nb)
    c.device_free(db); c.device_free(do)
    reads = [bytes(nb[i * L:(i + 1) * L]) for i in range(n)]
    assert sum(1 for r in reads if len(set(r)) <= 3) > n // 400          # the low-complexity reads are there
    Lb = gko.lib(); m = 10
    counts = np.zeros(4 ** m, np.uint32)
    for r in reads[:4000]:
        Lb.gko_count_mmers(r, len(r), m, counts)
    freq = np.zeros(4 ** m, np.uint32); Lb.gko_freq_order_from_counts(m, counts, freq)
    for k, parts, fr in ((31, 24, None), (63, 8, None), (31, 16, freq)):
        device_vs_oracle(gkc, reads, k, m, parts, freq=fr, histo_max=20000)


# (k, reads, partitions, generator profile): profile 1 = GKC_SYNTH_SKEWED, the repeat-rich genome with low-complexity reads (every full-size case used to be a
# uniform random genome; the reference's answer to partitions that explode is PartitionsCommand.cpp:505-545)
@pytest.mark.parametrize("k,n,parts,profile", [(31, 2_000_000, 512, 0), (31, 100_000_000, 4096, 0), (63, 50_000_000, 2048, 0), (63, 100_000_000, 4096, 0), (21, 100_000_000, 32768, 0), (47, 30_000_000, 9000, 0),
                                               (31, 100_000_000, 256, 0), (31, 30_000_000, 16, 0), (63, 40_000_000, 64, 0),      # few HUGE partitions: several workgroups per partition (slices)
                                               (31, 100_000_000, 4096, 1), (63, 100_000_000, 8192, 1), (31, 20_000_000, 64, 1)])
def test_size_independent_properties(gkc, k, n, parts, profile):
    """2e6 reads, then BASELINE configs[1] (k=31, 1e8 reads) and configs[3] (k=63, 1e8 reads; also 5e7) at full size:
    size-independent properties — multiset checksum (independent one-thread-per-read kernel vs counted records), sum of
    abundances == valid k-mers, strictly ascending partitions, partition membership of sampled records, histogram sums"""
    c = gkc.Counter(0)
    seed, L, G = 11, 150, n * 5
    m = 10
    rep = simple_repart(m, parts)
    c.configure(k, m, parts, rep)
    db, do = c.synth_reads_device(seed, n, L, G, 10000, profile=profile)
    cs, nv = c.kmer_checksum_device(db, do, n, n * L)
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    assert c.result_checksum() == (cs, nv)
    st = c.stats()
    assert st["kmers_nb_valid"] == nv == n * (L - k + 1)
    tot = 0
    for p in range(0, parts, max(37, parts // 24) if parts > 256 else max(1, parts // 3)):
        lo, hi, ab = c.partition(0, p)
        if k <= 31:
            assert (lo[1:] > lo[:-1]).all()
        else:
            assert ((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))).all()
        assert (ab >= 1).all()
        tot += len(lo)
        step = max(1, len(lo) // 20)
        for a, b in zip(lo[::step].tolist(), hi[::step].tolist()):
            key = (b << 64) | a
            s = "".join("ACTG"[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))
            mins, _ = gko.minimizers(s, k, m)
            assert rep[mins[0]] == p
    h = c.histogram()
    assert int(h.sum()) == st["kmers_nb_distinct"]
    if profile == 0:
        assert int((h * np.arange(len(h), dtype=np.uint64)).sum()) == nv      # no abundance reaches histo_max here
    else:
        assert int(h[-1]) > 0 and int((h * np.arange(len(h), dtype=np.uint64)).sum()) < nv      # the repeat families and the low-complexity k-mers sit in the last bin (abundance >= histo_max)
    if n == 100_000_000 and parts in (4096, 8192) and k in (31, 63):
        # VERDICT r5 #4a — ORACLE-EXACT at full size: the BASELINE configs[1] / configs[3] inputs (uniform and repeat-rich) counted by the oracle for a few sampled
        # partitions only (gko_dsk_run_parts drops the other partitions' super-k-mers where FillPartitions::processSuperkmer would append them) and compared byte for
        # byte with the device's Count[] of those partitions; the shape of TestDSK.cpp:254-305. The sample holds the partition of the poly-A k-mer (profile 1: the
        # heaviest sub-buckets of the run), the first, a middle and the last partition.
        bases = c.device_to_host(db, n * L); offs = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        mins, _ = gko.minimizers("A" * k, k, m)
        sample = sorted({0, int(rep[mins[0]]), parts // 2 + 1, parts - 1})
        ref = gko.Dsk(bases, offs, k, m, parts, rep, threads=os.cpu_count() or 1, only_parts=sample)
        assert ref.stats["kmers_nb_valid"] == nv
        for p in sample:
            exp = ref.part_records(p)
            assert len(exp) > 0 and np.array_equal(c.partition_records(0, p), exp), "partition %d of the full-size run differs from the oracle" % p
        ref.close(); del bases, offs
    c.device_free(db); c.device_free(do)


@pytest.mark.parametrize("k,n,parts,profile,batch_keys", [(31, 100_000_000, 4096, 0, 1 << 30), (31, 40_000_000, 1024, 1, 1 << 26), (63, 30_000_000, 2048, 0, 1 << 24)])
def test_batch_keys_do_not_change_the_count(gkc, k, n, parts, profile, batch_keys):
    """gkc_set_batch_keys (what the drop-in asks for: 2^30 k-mers per Stage-B batch, a third of the library's working set; down to 2^24 here: dozens of batches per lane):
    the records are the count of the input whatever the batches were — multiset checksum against the independent kernel, and every sampled partition byte for byte what
    the library's own plan produced."""
    c = gkc.Counter(0)
    L, m = 150, 10
    c.configure(k, m, parts, simple_repart(m, parts))
    db, do = c.synth_reads_device(5, n, L, n * 5, 10000, profile=profile)
    cs, nv = c.kmer_checksum_device(db, do, n, n * L)
    sample = list(range(0, parts, max(1, parts // 16)))
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    assert c.result_checksum() == (cs, nv)
    ref = [c.partition_records(0, p).copy() for p in sample]
    hist = c.histogram().copy()
    c.set_batch_keys(batch_keys)
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    assert c.result_checksum() == (cs, nv)
    for p, r in zip(sample, ref):
        assert np.array_equal(c.partition_records(0, p), r), "partition %d differs with batches of %d k-mers" % (p, batch_keys)
    assert np.array_equal(c.histogram(), hist)
    c.set_batch_keys(0)
    c.device_free(db); c.device_free(do)


def test_share_of_8_path_at_full_size(gkc):
    """BASELINE configs[2]'s per-GPU share exactly as bench.py's `share_of_8` block runs it (that path had no correctness check anywhere): k=31,
    1.25e8 reads in FOUR pushes, 32768 partitions (two-level Stage A: 4096 groups of 8), every push followed by gkc_exchange through a one-rank RCCL communicator
    (planning, narrowing of the own segments, import path). Size-independent properties: the multiset checksum of the records == the independent kernel's over the
    four chunks, sum of abundances == valid k-mers, sampled partitions strictly ascending and holding only k-mers whose minimizer maps to them."""
    from gatb_core_amd import dist as gd
    k, m, L, n, parts, pushes = 31, 10, 150, 125_000_000, 32768, 4
    c = gkc.Counter(0)
    rep = simple_repart(m, parts)
    c.configure(k, m, parts, rep)
    per = n // pushes
    chunks = []
    cs, nv = 0, 0
    for i in range(pushes):
        db, do = c.synth_reads_device(2, per, L, n * L // 30, 10000, first_read=i * per)
        s_, n_ = c.kmer_checksum_device(db, do, per, per * L)
        cs = (cs + int(s_)) & ((1 << 64) - 1); nv += int(n_)
        chunks.append((db, do))
    dc = gd.DistributedCounter(c, 0, 1, parts)
    c.begin_pass(0)
    for db, do in chunks:
        c.push_reads_device(db, do, per, per * L); dc.exchange()
    c.finish_pass()
    got = c.result_checksum()
    assert (int(got[0]), int(got[1])) == (cs, nv)
    st = c.stats()
    assert st["kmers_nb_valid"] == nv == n * (L - k + 1)
    for p in range(0, parts, parts // 16 + 1):
        lo, hi, ab = c.partition(0, p)
        assert (lo[1:] > lo[:-1]).all() and (ab >= 1).all()
        for a in lo[::max(1, len(lo) // 8)].tolist():
            s = "".join("ACTG"[(a >> (2 * (k - 1 - i))) & 3] for i in range(k))
            mins, _ = gko.minimizers(s, k, m)
            assert rep[mins[0]] == p
    h = c.histogram()
    assert int(h.sum()) == st["kmers_nb_distinct"] and int((h * np.arange(len(h), dtype=np.uint64)).sum()) == nv
    dc.comm.close()
    for db, do in chunks:
        c.device_free(db); c.device_free(do)


@pytest.mark.parametrize("k,n,passes", [(31, 6_000_000, 3), (63, 3_000_000, 2), (21, 40_000, 4)])
def test_overlapped_passes(gkc, k, n, passes):
    """Stage A of pass p+1 beside Stage B of pass p (VERDICT r3 Missing #4; the reference's pass loop is serial because of its disk, SortingCountAlgorithm.cpp:672-692):
    gkc_finish_pass_async DETACHES a pass — number, segment list, arenas, a stream of its own — and gkc_begin_pass / gkc_push_reads_device of the next pass run while it
    is counted. Every dataset, the statistics and the histogram must be those of the same passes run one after the other; a consumer may fetch and release pass p
    while pass p+1 is being counted; what would touch shared state (pass 0 = a new run, the pass that is being counted) is refused."""
    m, parts, L = 10 if k > 10 else k - 1, 64, 150
    rep = simple_repart(m, parts)
    def run(overlapped):
        c = gkc.Counter(0)
        c.configure(k, m, parts, rep, nb_passes=passes); c.set_solidity(2, 1000, 10000)
        db, do = c.synth_reads_device(7, n, L, n * 5, 10000)
        got = {}
        for ps in range(passes):
            c.begin_pass(ps); c.push_reads_device(db, do, n, n * L)
            if overlapped:
                c.finish_pass_async()
                if ps == 0:                                    # beside pass 0: pass 0 again (a new run) and nonsense are refused, the context stays usable
                    for bad in (0,):
                        with pytest.raises(gkc.GkcError):
                            c.begin_pass(bad)
                if ps > 0:                                     # the pass before is through (finish_pass_async joined it): fetch it and give its memory back
                    for p in range(parts):
                        got[(ps - 1, p)] = c.partition_records(ps - 1, p).copy()
                    c.release_pass(ps - 1)
            else:
                c.finish_pass()
        if overlapped:
            c.finish_pass_wait()
            for p in range(parts):
                got[(passes - 1, p)] = c.partition_records(passes - 1, p).copy()
        else:
            for ps in range(passes):
                for p in range(parts):
                    got[(ps, p)] = c.partition_records(ps, p).copy()
        st = c.stats(); h = c.histogram()
        c.device_free(db); c.device_free(do); c.close()
        return got, st, h
    a, sa, ha = run(False)
    b, sb, hb = run(True)
    assert sorted(a) == sorted(b)
    for key in a:
        assert np.array_equal(a[key], b[key]), key
    for key in ("kmers_nb_valid", "kmers_nb_invalid", "kmers_nb_distinct", "kmers_nb_solid", "nb_sequences", "nb_superkmers"):
        assert sa[key] == sb[key], key
    assert np.array_equal(ha, hb)
    assert sa["kmers_nb_solid"] > 0 and sum(len(v) for v in a.values()) == sa["kmers_nb_solid"] * (16 if k <= 31 else 32)


def test_multi_pass_with_two_lanes_and_solidity(gkc):
    """three passes over 6e6 reads (2.4e8 keys per pass: the two-lane Stage B with its probe batch, results of earlier passes
    resident), abundance window [2, 50]: the records of all passes together == the multiset of valid k-mers with that abundance
    (checksum of the distinct k-mers is not available for a window, so: pass p holds exactly the minimizers = p mod 3, datasets ascend,
    sum over the histogram == valid k-mers, and the solid counts equal the one-pass run's)"""
    c = gkc.Counter(0)
    k, m, parts, n, L = 31, 10, 512, 6_000_000, 150
    rep = simple_repart(m, parts)
    db, do = c.synth_reads_device(5, n, L, n * 5, 10000)
    c.configure(k, m, parts, rep, nb_passes=1)
    c.set_solidity(2, 50, 10000)
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    one = c.stats()
    ref = {}
    for p in range(0, parts, 61):
        lo, hi, ab = c.partition(0, p)
        ref[p] = (lo.copy(), ab.copy())
    c.configure(k, m, parts, rep, nb_passes=3)
    c.set_solidity(2, 50, 10000)
    for ps in range(3):
        c.begin_pass(ps); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    st = c.stats()
    assert st["kmers_nb_valid"] == one["kmers_nb_valid"] == n * (L - k + 1)
    assert st["kmers_nb_distinct"] == one["kmers_nb_distinct"] and st["kmers_nb_solid"] == one["kmers_nb_solid"]
    h = c.histogram()
    assert int((h * np.arange(len(h), dtype=np.uint64)).sum()) == n * (L - k + 1)
    for p, (rlo, rab) in ref.items():
        got_lo, got_ab = [], []
        for ps in range(3):
            lo, hi, ab = c.partition(ps, p)
            assert (lo[1:] > lo[:-1]).all() and ((ab >= 2) & (ab <= 50)).all()
            for a in lo[:: max(1, len(lo) // 5)].tolist():
                s_ = "".join("ACTG"[(a >> (2 * (k - 1 - i))) & 3] for i in range(k))
                mins, _ = gko.minimizers(s_, k, m)
                assert mins[0] % 3 == ps and rep[mins[0]] == p
            got_lo.append(lo); got_ab.append(ab)
        glo = np.concatenate(got_lo); gab = np.concatenate(got_ab)
        o = np.argsort(glo, kind="stable")
        assert np.array_equal(glo[o], rlo) and np.array_equal(gab[o], rab)
    c.device_free(db); c.device_free(do)


def test_release_pass_gives_the_memory_back(gkc):
    """gkc_release_pass: the datasets of a drained pass are gone (access fails loudly), its HBM is usable again, the statistics stay; the other
    pass is untouched"""
    c = gkc.Counter(0)
    k, m, parts, n = 31, 10, 64, 2_000_000
    rep = simple_repart(m, parts)
    c.configure(k, m, parts, rep, nb_passes=2)
    db, do = c.synth_reads_device(3, n, 150, n * 5, 10000)
    for ps in range(2):
        c.begin_pass(ps); c.push_reads_device(db, do, n, n * 150); c.finish_pass()
    st = c.stats()
    keep = c.partition(1, 7)
    n0 = sum(c.partition_info(0, p)[0] for p in range(parts))
    u0, total = c.device_memory()
    c.release_pass(0)
    u1, _ = c.device_memory()
    assert total > 200e9 and u1 - u0 >= n0 * 16 * 0.9
    with pytest.raises(gkc.GkcError):
        c.partition_info(0, 3)
    with pytest.raises(gkc.GkcError):                      # whole-context consumers (checksum, Bloom / MPHF of the solid set) refuse a context with a released pass
        c.result_checksum()
    again = c.partition(1, 7)
    assert all(np.array_equal(a, b) for a, b in zip(keep, again))
    assert c.stats() == st
    c.device_free(db); c.device_free(do)


def test_two_owner_shards_on_one_gpu(gkc):
    """multi-GPU data flow on one device: two contexts scan one half of the reads each, their buckets are routed to the
    partition owners exactly as dist.py does (slices of the arena -> foreign segments), each owner counts its partitions;
    the union must equal the oracle's single-process result"""
    import torch
    ge.load()
    from gatb_core_amd import dist as gd
    reads = synth_reads(4000, 20000, 150, seed=21, n_rate=0.001, ragged=True)
    k, m, parts, world = 31, 10, 8, 2
    rep = simple_repart(m, parts)
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, parts, rep)
    ctxs = [gkc.Counter(0) for _ in range(world)]
    exports = []
    half = len(reads) // 2
    for r, c in enumerate(ctxs):
        c.configure(k, m, parts, rep)
        c.begin_pass(0)
        r0, r1 = (0, half) if r == 0 else (half, len(reads))
        c.push_reads(bases[int(offs[r0]):int(offs[r1])], offs[r0:r1 + 1] - offs[r0])
        ptr, rb, off, km = c.segment_export(0)
        t = torch.as_tensor(gd.DevArray(ptr, int(off[-1]) * rb), device="cuda").clone()     # "send buffer"
        exports.append((t, rb, off.astype(np.int64), km.astype(np.int64)))
    ranges = gd.owner_ranges(parts, world)
    keep = []
    for r, c in enumerate(ctxs):
        c.segments_clear()
        lo, hi = ranges[r]
        for (t, rb, off, km) in exports:                       # chunk from every source rank
            chunk = t[off[lo] * rb: off[hi] * rb].clone(); keep.append(chunk)
            ro = np.zeros(parts + 1, np.int64); ro[lo + 1:hi + 1] = off[lo + 1:hi + 1] - off[lo]; ro[hi + 1:] = ro[hi]
            kk = np.zeros(parts, np.int64); kk[lo:hi] = km[lo:hi]
            if ro[-1]:
                c.segment_import(chunk.data_ptr(), ro.astype(np.uint64), kk.astype(np.uint64))
        c.finish_pass()
    torch.cuda.synchronize()
    tot_distinct = 0
    for r, c in enumerate(ctxs):
        lo, hi = ranges[r]
        for p in range(parts):
            lo_, hi_, ab = c.partition(0, p)
            if lo <= p < hi:
                rlo, rhi, rab = ref.part(p)
                assert np.array_equal(lo_, rlo) and np.array_equal(ab, rab)
            else:
                assert len(lo_) == 0
        tot_distinct += c.stats()["kmers_nb_distinct"]
    assert tot_distinct == ref.stats["kmers_nb_distinct"]
    assert sum(c.stats()["kmers_nb_valid"] for c in ctxs) == ref.stats["kmers_nb_valid"]


def test_distributed_counter_world1_rccl(gkc):
    """DistributedCounter over the real backend with one rank: torch.distributed (nccl == RCCL) hands the ncclUniqueId around, the library
    opens its own RCCL communicator and gkc_exchange runs on it (nothing to move with one rank; owners = all partitions)"""
    import os, socket, torch
    import torch.distributed as dist
    ge.load()
    from gatb_core_amd import dist as gd
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        reads = synth_reads(3000, 20000, 150, seed=5)
        k, m, parts = 31, 10, 16
        rep = simple_repart(m, parts)
        bases, offs = gko.pack_reads(reads)
        ref = gko.Dsk(bases, offs, k, m, parts, rep)
        c = gkc.Counter(0); c.configure(k, m, parts, rep)
        dc = gd.DistributedCounter(c, 0, 1, parts)
        c.begin_pass(0); c.push_reads(bases, offs); dc.exchange(); c.finish_pass()
        assert c.all_counts() == ref.all_counts()
        assert list(dc.owned()) == list(range(parts)) and dc.stats()["n_exchanges"] == 1
    finally:
        dist.destroy_process_group()


def test_rccl_send_recv_and_chunking_on_hardware(gkc):
    """VERDICT r2 (3d/3e): the library's OWN grouped ncclSend / ncclRecv path with its 256 MiB chunking, on the GPU: a one-rank RCCL communicator sends 600 MiB + 24 B
    (three chunks, the last one ragged) to itself (gkc_comm_loopback) and every word arrives; then three pushes with an exchange each (three segments) count right"""
    c = gkc.Counter(0)
    comm = gkc.Comm.rccl(c, gkc.Comm.unique_id(), 1, 0)
    for n in (1 << 20, (600 << 20) + 24):
        bad, ms = comm.loopback(n)
        assert bad == 0, "%d of %d words differ after the self send" % (bad, n // 8)
    reads = synth_reads(3000, 20000, 150, seed=6)
    k, m, parts = 31, 10, 16
    rep = simple_repart(m, parts)
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, parts, rep)
    c.configure(k, m, parts, rep)
    c.begin_pass(0)
    for part in (reads[:1000], reads[1000:2000], reads[2000:]):
        b, o = gko.pack_reads(part); c.push_reads(b, o); c.exchange(comm)
    c.finish_pass()
    assert c.all_counts() == ref.all_counts() and comm.stats()["n_exchanges"] == 3
    comm.close()


def test_bloom_or_reduce_world1_rccl(gkc):
    """gkc_bloom_allreduce_or over a one-rank RCCL communicator (ncclCommInitRank inside the library) is the identity; the two-rank
    reduction runs in tests/test_gpu_dist.py"""
    rng = np.random.default_rng(77)
    k = 31
    keys = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(40000)]
    c = gkc.Counter(0)
    whole = gkc.Bloom(c, "neighbor", 2_000_000, 7, k); whole.insert(keys)
    comm = gkc.Comm.rccl(c, gkc.Comm.unique_id(), 1, 0)
    before = whole.array().copy()
    whole.allreduce_or(comm)
    assert np.array_equal(whole.array(), before)
    comm.close()


def test_repartitor_sampling_statistics(gkc):
    """gkc_sample_minimizers / gkc_count_mmers against the oracle's restatement of SampleRepart / MmersFrequency"""
    reads = synth_reads(800, 8000, 150, seed=31, n_rate=0.002, ragged=True)
    bases, offs = gko.pack_reads(reads)
    k, m = 31, 8
    c = gkc.Counter(0); c.configure(k, m, 4, np.zeros(4 ** m, np.uint16))
    nsk, nk = c.sample_minimizers(bases, offs)
    exp_k = np.zeros(4 ** m, np.uint64)
    for r in reads:
        mn, st, nb, nv, ni = gko.superkmers(r, k, m)
        np.add.at(exp_k, mn, nb.astype(np.uint64))
    assert np.array_equal(nk, exp_k)                       # k-mers per minimizer: independent of tile splits
    assert nsk.sum() >= 1 and (nsk[nk == 0] == 0).all()
    cnt = c.count_mmers(m, bases, offs)
    exp = np.zeros(4 ** m, np.uint32)
    L = gko.lib()
    for r in reads:
        L.gko_count_mmers(r, len(r), m, exp)
    assert np.array_equal(cnt, exp)


def test_streamed_results_land_in_the_host_sink(gkc):
    """gkc_set_host_sink + gkc_finish_pass_async + gkc_wait_partition: every partition's Count[] arrives in page-locked host memory while Stage B
    runs, byte-identical to gkc_partition_counts; a sink that is too small keeps the rest on the device and says so"""
    reads = synth_reads(6000, 30000, 150, seed=61, n_rate=0.001, ragged=True)
    bases, offs = gko.pack_reads(reads)
    k, m, parts = 31, 9, 32
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(2, 2147483647, 10000)
    sink = gkc.HostBuffer(64 << 20)
    c.set_host_sink(sink)
    c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass_async()
    got = {}
    for p in range(parts):                                   # consumer walks the partitions in order while Stage B runs
        view, n = c.wait_partition(0, p)
        got[p] = (None if view is None else view.copy(), n)
    c.finish_pass_wait()
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=2)
    total = 0
    for p in range(parts):
        dev = c.partition_records(0, p)
        view, n = got[p]
        assert n * 16 == len(dev) and np.array_equal(dev, ref.part_records(p))
        if n:
            assert view is not None and np.array_equal(view, dev)
        total += n
    assert total == ref.stats["kmers_nb_solid"]
    # too small a sink: nothing is lost, the records stay fetchable from the device
    tiny = gkc.HostBuffer(4096)
    c.set_host_sink(tiny)
    c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
    assert b"sink" in (c.L.gkc_last_error(c.h) or b"")
    for p in range(parts):
        assert np.array_equal(c.partition_records(0, p), ref.part_records(p))
    c.set_host_sink(None)


@pytest.mark.parametrize("k,amin", [(31, 1), (21, 2), (15, 1)])
def test_packed_sink_escapes_and_block_boundaries(gkc, k, amin):
    """The batches of an 8-byte-key count cross PCIe PACKED (csrc/gkc_sink.hip: per block of 8192 records a base key, then 6-byte key deltas + 1-byte abundances) and
    host threads expand them in the sink: what gkc_wait_partition hands out must be byte for byte gkc_partition_counts. Input chosen so that every special case
    occurs: one read copied 700 times (abundances >= 255: the 1-byte field escapes to the exception list), few k-mers per partition (gaps >= 2^48 between
    consecutive keys at k=31: the delta field escapes), partitions of more than one block (k=15: many records in few partitions) and empty partitions."""
    rng = np.random.default_rng(5)
    reads = synth_reads(20000 if k == 15 else 3000, 400000 if k == 15 else 20000, 150, seed=62, n_rate=0.001)
    reads += [reads[0]] * 700 + [b"A" * 150] * 300 + [b"ACGT" * 40] * 260
    bases, offs = gko.pack_reads(reads)
    m, parts = min(k - 1, 8), (3 if k == 15 else 40)
    rep = simple_repart(m, parts)
    if k == 21:
        rep[rep == 5] = 6                                    # an empty partition
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
    sink = gkc.HostBuffer(256 << 20)
    c.set_host_sink(sink)
    for rnd in range(2):                                     # the second pass reuses the staging buffer and the unpack threads
        c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
        big_ab = big_gap = multi_block = 0
        for p in range(parts):
            view, n = c.wait_partition(0, p)
            dev = c.partition_records(0, p)
            assert n * 16 == len(dev)
            if n:
                assert view is not None and np.array_equal(view, dev), (k, p, n)
                r = dev.view(np.uint64).reshape(-1, 2)
                big_ab += int((r[:, 1] >= 255).sum()); big_gap += int((np.diff(r[:, 0]) >= np.uint64((1 << 48) - 1)).sum()); multi_block += int(n > 8192)
        assert big_ab > 0
        if k == 31:
            assert big_gap > 0
        if k == 15:
            assert multi_block > 0
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=amin)
    for p in range(parts):
        assert np.array_equal(c.partition_records(0, p), ref.part_records(p))
    c.set_host_sink(None)


@pytest.mark.parametrize("k,amin,dense,fixed", [(63, 1, 0, 0), (63, 1, 1, 0), (41, 2, 1, 0), (33, 1, 1, 0), (47, 1, 0, 0), (63, 1, 0, 1), (63, 1, 1, 1), (41, 2, 1, 1), (33, 1, 1, 1)])
def test_packed_sink_16_byte_keys(gkc, monkeypatch, k, amin, dense, fixed):
    """k >= 32: the 32-byte Count records {u128 value; i32 abundance; padding} cross PCIe packed as well (csrc/gkc_sink.hip, k_pack_counts2: per block of 8192 records a
    16-byte base key, then [15- or 16-byte key delta][1-byte abundance] = 16 / 17 bytes instead of 32) and are expanded in the sink by the library's threads: what
    gkc_wait_partition hands out must be byte for byte gkc_partition_counts, and that the oracle's records. `dense` (GKC_SINK_DENSE=1) takes the 15-byte deltas on
    partitions of few records — every gap beyond 2^120 escapes through the two-entry exception path; a read copied 700 times gives abundances >= 255 (escape of
    the abundance byte); k=33 at 20000 reads in 2 partitions gives partitions of several blocks; an empty partition; two passes over the same staging buffer."""
    if dense:
        monkeypatch.setenv("GKC_SINK_DENSE", "1")
    if fixed:
        monkeypatch.setenv("GKC_SINK_WIDTH6", "0")          # rounds 4-5: fixed 15- / 16-byte deltas with escapes; default since round 6: a delta width per sub-block of 128 records (PKV, no key escapes)
    big = k == 33
    reads = synth_reads(20000 if big else 3000, 400000 if big else 20000, 150, seed=62 + k, n_rate=0.001)
    reads += [reads[0]] * 700 + [b"A" * 150] * 300 + [b"ACGT" * 40] * 260
    bases, offs = gko.pack_reads(reads)
    m, parts = 8, (2 if big else 3000 if (dense and k >= 60) else 24)         # (3000 partitions of ~60 records in a 126-bit key space: gaps beyond 2^120)
    rep = simple_repart(m, parts)
    if k == 41:
        rep[rep == 5] = 6                                    # an empty partition
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
    sink = gkc.HostBuffer(512 << 20)
    c.set_host_sink(sink)
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=amin)
    for rnd in range(2):
        c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
        big_ab = big_gap = multi_block = nrec = 0
        for p in range(parts):
            view, n = c.wait_partition(0, p)
            dev = c.partition_records(0, p)
            assert n * 32 == len(dev)
            nrec += n
            if n:
                assert view is not None and np.array_equal(view, dev), (k, p, n)
                r = dev.view(np.uint64).reshape(-1, 4)
                big_ab += int((r[:, 2] >= 255).sum()); multi_block += int(n > 8192)
                hi = r[:, 1].astype(object); big_gap += int(sum(1 for a, b in zip(hi[:-1], hi[1:]) if (int(b) - int(a)) >> 56))
            assert np.array_equal(dev, ref.part_records(p))
        assert big_ab > 0
        if dense and k >= 60:
            assert big_gap > 0                               # deltas beyond 2^120 travelled as escapes
        if big:
            assert multi_block > 0
        wire = c.stats()["sink_wire_bytes"]
        assert 0 < wire < nrec * 32 or nrec < 8192 * parts   # (partitions of a few records still travel as whole block slots)
    c.set_host_sink(None)


@pytest.mark.parametrize("k,amin", [(31, 1), (31, 2), (63, 1)])
def test_raw_sink_mode_lands_the_same_bytes(gkc, k, amin):
    """gkc_set_sink_mode(GKC_SINK_RAW): every batch's Count[] lands in the sink by one DMA copy (no packing on the device, no expansion threads on the host) — the mode
    several ranks sharing one host's DRAM fall back to. The sink must hold byte for byte what the packed mode leaves there and what the oracle
    counts; the mode is switched on a live context (packed -> raw -> packed), the wire-byte statistic follows (0 in raw mode)."""
    reads = synth_reads(6000, 30000, 150, seed=90 + k, n_rate=0.001)
    reads += [reads[0]] * 300
    bases, offs = gko.pack_reads(reads)
    m, parts = 8, 6
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
    sink = gkc.HostBuffer(512 << 20)
    c.set_host_sink(sink)
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=amin)
    landed = {}
    for mode in ("packed", "raw", "packed"):
        c.set_sink_mode(mode)
        c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
        got = []
        for p in range(parts):
            view, n = c.wait_partition(0, p)
            exp = ref.part_records(p)
            assert n * c.rec_bytes == len(exp)
            if n:
                assert view is not None and np.array_equal(view, exp), (mode, p)
            got.append(bytes(view) if n else b"")
        wire = c.stats()["sink_wire_bytes"]
        assert (wire == 0) == (mode == "raw"), (mode, wire)
        landed.setdefault(mode, got)
        assert got == landed["packed"]
    c.set_host_sink(None)


@pytest.mark.parametrize("k", [31, 63])
def test_sink_mixes_packed_and_raw_batches(gkc, monkeypatch, k):
    """Several ranks on one host: a rank whose share of the host's expansion threads is behind sends its next batch RAW (gkc_sink_host_behind: landed, unexpanded records
    beyond 1.5 batches) — packed and raw batches then alternate inside one pass, on one copy stream, into one sink. GKC_SINK_ADAPTIVE=2 makes every other batch raw:
    ~10 batches of 2^24 k-mers; the sink must hold the oracle's records, and the bytes on the link lie between all-packed and all-raw."""
    monkeypatch.setenv("GKC_SINK_ADAPTIVE", "2")
    n = 700_000
    reads = synth_reads(n, n * 5, 150, seed=17 + k, n_rate=0.0005)
    bases, offs = gko.pack_reads(reads)
    m, parts = 10, 96
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep)
    c.set_batch_keys(1 << 24)
    sink = gkc.HostBuffer(4 << 30)
    c.set_host_sink(sink)
    ref = gko.Dsk(bases, offs, k, m, parts, rep, threads=os.cpu_count() or 1)
    c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
    nrec = 0
    for p in range(parts):
        view, cnt = c.wait_partition(0, p)
        exp = ref.part_records(p)
        assert cnt * c.rec_bytes == len(exp) and (cnt == 0 or np.array_equal(view, exp)), p
        nrec += cnt
    wire = c.stats()["sink_wire_bytes"]
    assert 0.45 * nrec * c.rec_bytes < wire < 0.95 * nrec * c.rec_bytes, (wire, nrec)       # about half of the records packed (0.4x), half raw
    c.set_host_sink(None)


@pytest.mark.parametrize("switch", ["GKC_SINK_DENSE=1", "GKC_SINK_DENSE=1,GKC_SINK_WIDTH6=0", "GKC_SINK_DENSE=1,GKC_UNPACK_THREADS=1"])
def test_packed_sink_entry_widths(switch):
    """The three entry widths of the packed transfer on the same inputs (GKC_SINK_DENSE=1 declares every batch dense): per-block delta widths + abundance bitmap +
    abundance stream (PKV, reported as width 6; abundance-min 1, the default there), 7-byte entries (GKC_SINK_WIDTH6=0, and abundance-min 2), with delta escapes (k=31: few records per partition, gaps
    beyond 2^48), abundance escapes (a read copied 700 times), partitions of several blocks (k=15) and an empty one; an error-free input (nearly every abundance
    > 1) keeps the block widths as long as they stay below 7 bytes per record. What lands in the sink == gkc_partition_counts == the oracle; the bytes
    the library says it queued (gkc_stats.reserved[1]) are what the widths promise where the partitions fill their blocks."""
    import json, os, subprocess, sys
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads
gkc = ge.load().gkc
res = {}
for name, k, amin, sub in (("k15", 15, 1, 0.01), ("k31", 31, 1, 0.01), ("k21_amin2", 21, 2, 0.01), ("k15_clean", 15, 1, 0.0)):
    reads = synth_reads(20000 if k == 15 else 3000, 400000 if k == 15 else 20000, 150, seed=62, n_rate=0.001, sub_rate=sub)
    reads += [reads[0]] * 700 + [b"A" * 150] * 300 + [b"ACGT" * 40] * 260
    bases, offs = gko.pack_reads(reads)
    m, parts = min(k - 1, 8), (3 if k == 15 else 40)
    rep = simple_repart(m, parts)
    if k == 21: rep[rep == 5] = 6
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.set_solidity(amin, 2147483647, 10000)
    sink = gkc.HostBuffer(256 << 20); c.set_host_sink(sink)
    ok = True; wire = []
    for rnd in range(2):
        c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
        nrec = 0
        for p in range(parts):
            view, n = c.wait_partition(0, p)
            dev = c.partition_records(0, p)
            ok = ok and n * 16 == len(dev) and (n == 0 or (view is not None and np.array_equal(view, dev)))
            nrec += n
        wire.append(c.stats()["sink_wire_bytes"] / max(1, nrec))
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=amin)
    ok = ok and all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts))
    c.set_host_sink(None)
    res[name] = {"ok": bool(ok), "bytes_per_record": wire}
print(json.dumps(res))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for kv in switch.split(","):
        name, _, val = kv.partition("="); env[name] = val or "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert all(v["ok"] for v in res.values()), res
    assert all(0 < b < 9 for n_, v in res.items() if n_.startswith("k15") for b in v["bytes_per_record"]), res      # (partitions of a few records still travel as whole block slots)
    if "WIDTH6=0" not in switch:
        assert res["k15"]["bytes_per_record"][0] < 7.0, res                                 # 6-byte entries where most abundances are 1 ...
        # ... and they stay where the block widths + the abundance stream are below the 7 bytes of the fixed entries (k = 15: narrow deltas) even though nearly
        # every abundance is > 1 (rounds 3-5, fixed 6-byte deltas: such a batch switched the context to 7-byte entries)
        assert res["k15_clean"]["bytes_per_record"][1] <= res["k15_clean"]["bytes_per_record"][0] * 1.001 < 7.0, res


def test_push_reads_in_overlapped_chunks(gkc):
    """gkc_push_reads sends host reads in chunks through two staging buffers (H2D of chunk j+1 under the scan of chunk j): forced to many
    small chunks (GKC_PUSH_CHUNK) the counts must not change, whatever the read boundaries"""
    import os, subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads
gkc = ge.load().gkc
reads = synth_reads(2500, 20000, 150, seed=71, n_rate=0.002, ragged=True)
bases, offs = gko.pack_reads(reads)
k, m, parts = 31, 8, 8
rep = simple_repart(m, parts)
c = gkc.Counter(0); c.configure(k, m, parts, rep); c.count(bases, offs)
ref = gko.Dsk(bases, offs, k, m, parts, rep)
ok = all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts))
print(json.dumps({"ok": bool(ok), "segments": c.stats()["nb_sequences"] == len(reads), "valid": c.stats()["kmers_nb_valid"] == ref.stats["kmers_nb_valid"]}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GKC_PUSH_CHUNK="7000")            # ~54 chunks of this input
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res == {"ok": True, "segments": True, "valid": True}, res


def test_giant_device_push_is_scanned_in_slices(gkc):
    """gkc_push_reads_device beyond GKC_PUSH_SPLIT bases is scanned slice by slice, cut at a read whose first base is 16-byte aligned (one segment and one set of
    per-push buffers per slice instead of giant ones: VERDICT r3 weak #7). Forced to small slices on ragged reads (aligned read starts are rare: some slices grow
    into the next one) the records must be those of the whole push; fixed-length reads as well."""
    import os, subprocess, sys, json
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads
import torch
gkc = ge.load().gkc
res = {}
for name, ragged in (("ragged", True), ("fixed", False)):
    reads = synth_reads(4000, 30000, 150, seed=72, n_rate=0.002, ragged=ragged)
    bases, offs = gko.pack_reads(reads)
    k, m, parts = 31, 8, 8
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep)
    db = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).cuda(); do = torch.from_numpy(offs.astype(np.int64)).cuda()
    c.begin_pass(0); c.push_reads_device(db.data_ptr(), do.data_ptr(), len(reads), int(offs[-1])); c.finish_pass()
    ref = gko.Dsk(bases, offs, k, m, parts, rep)
    res[name] = {"ok": bool(all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts))), "segments": c.segment_count(),
                 "reads": c.stats()["nb_sequences"] == len(reads), "valid": c.stats()["kmers_nb_valid"] == ref.stats["kmers_nb_valid"]}
print(json.dumps(res))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GKC_PUSH_SPLIT="20000")            # ~30 slices of this input
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    for name in ("ragged", "fixed"):
        assert res[name]["ok"] and res[name]["reads"] and res[name]["valid"], res
        assert res[name]["segments"] > 5, res


@pytest.mark.parametrize("k,m,freq", [(31, 8, False), (21, 7, True), (41, 9, False)])
def test_exact_repartitor_sample(gkc, k, m, freq):
    """gkc_sample_exact (SampleRepart restated read by read on the device: super-k-mers, k-mers and kx-mers per minimizer, and the reference's stop rule)
    against the oracle's super-k-mer split of every read + the kx-mer rule of RepartitionAlgorithm.cpp:186-203 stated here in Python"""
    reads = synth_reads(500, 6000, 150, seed=83, n_rate=0.004, ragged=True)
    bases, offs = gko.pack_reads(reads)
    fo = None
    if freq:
        cnt = np.zeros(4 ** m, np.uint32)
        for r in reads:
            gko.lib().gko_count_mmers(r, len(r), m, cnt)
        fo = np.zeros(4 ** m, np.uint32); gko.lib().gko_freq_order_from_counts(m, cnt, fo)
    c = gkc.Counter(0); c.configure(k, m, 4, np.zeros(4 ** m, np.uint16), freq_order=fo)

    def expected(n_reads_limit=None, threshold=None):
        nsk = np.zeros(4 ** m, np.uint64); nk = np.zeros(4 ** m, np.uint64); nkx = np.zeros(4 ** m, np.uint64)
        seen = 0; used = 0
        for r in reads:
            mn, st, nb, nv, ni = gko.superkmers(r, k, m, freq_order=fo)
            km = gko.kmers(r, k)
            which = (km["fwd_lo"] == km["can_lo"]) & (km["fwd_hi"] == km["can_hi"])       # canonical == forward strand
            for a, s0, n in zip(mn.tolist(), st.tolist(), nb.tolist()):
                nsk[a] += 1; nk[a] += n
                kx = 1; run = 0; prev = which[s0]
                for i in range(1, n):
                    if which[s0 + i] != prev or run >= 4:
                        kx += 1; run = 0
                    else:
                        run += 1
                    prev = which[s0 + i]
                nkx[a] += kx
            seen += len(mn); used += 1
            if threshold is not None and seen > threshold:
                break
        return nsk, nk, nkx, used

    for thr in (10 ** 9, 700):
        a, b, d, used = c.sample_exact(bases, offs, thr)
        ea, eb, ed, eused = expected(threshold=thr)
        assert used == eused and np.array_equal(a, ea) and np.array_equal(b, eb) and np.array_equal(d, ed), (thr, used, eused)


@pytest.mark.parametrize("switch", ["GKC_NO_F64", "GKC_SCAN_NO_DESC", "GKC_SCAN_GLOBAL_ATOMICS", "GKC_BATCH_LPT=0", "GKC_DEEP_BITS=2", "GKC_WG_MAX=1024", "GKC_DEDUPE=0", "GKC_DEDUPE=1", "GKC_MAX_SUB_BITS=1", "GKC_MAX_SUB_BITS=6",
                                    "GKC_WEIGHT_BITS=2", "GKC_WEIGHT_BITS=4", "GKC_WEIGHT_BITS=4,GKC_MAX_SUB_BITS=0", "GKC_WEIGHT_BITS=3,GKC_NO_F64"])
def test_alternative_kernel_paths_stay_bit_exact(gkc, switch):
    """A/B switches that select another HIP code path of the same library (integer instead of f64-tagged compare-exchange; the Stage A fallbacks: emit pass
    that recomputes instead of reading descriptors, global-atomic cursors instead of LDS ones; partitions in batch order; split levels of 2 bits each, which
    forces more of them than the fixed launches; no workgroup tier; 2 / 64 sub-buckets per partition, which makes every sub-bucket a root of the split levels —
    with 2 they are "giants" split by many workgroups together; 2 / 3 / 4 weight bits under the k-mer of a sort key — by default 3 at k = 31 and k = 63, where the
    key's top bit then lives in the sub-bucket's index and not in the stored word, 4 at k <= 30; a request the partitions' sub-bucket bits cannot carry is cut
    down): each must give the oracle's records on an input with N's, ragged reads, low-complexity reads (oversize buckets), reads copied 16 .. 50 times (merged
    records at every weight up to the cap and beyond it), G-rich reads (the stored keys closest to the all-ones padding / EMPTY word) and enough k-mers per
    partition for every tier to run — k = 30, k = 31, k = 41 and k = 63 (the longest records the 256-bit canonical form of the deduplication sees).
    (The measured-slower round-2 kernels left the product: branch experiments-r02, logs in profiles/r02_*_experiment.txt.)"""
    import json, os, subprocess, sys
    code = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads
gkc = ge.load().gkc
reads = synth_reads(30000, 60000, 150, seed=97, n_rate=0.001, ragged=True)
reads += [b"A" * 150] * 300 + [b"ACACACACAC" * 15] * 300 + [(b"ACGTTGCA" * 19)[:150]] * 200
reads += [b"G" * 150] * 40 + [b"A" + b"G" * 149] * 40 + [b"C" + b"G" * 149] * 20 + [b"T" + b"G" * 100 + b"A" + b"G" * 48] * 17
reads += [reads[7]] * 50 + [reads[11]] * 16 + [reads[13]] * 17 + [reads[17]] * 33
bases, offs = gko.pack_reads(reads)
res = {}
for k, m, parts in ((30, 8, 3), (31, 8, 3), (41, 9, 2), (63, 10, 2)):
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.count(bases, offs)
    ref = gko.Dsk(bases, offs, k, m, parts, rep, threads=4)
    res[str(k)] = bool(all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts)) and np.array_equal(c.histogram(), ref.histogram()))
print(json.dumps(res))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for kv in switch.split(","):                                          # "NAME" or "NAME=value"
        name, _, val = kv.partition("="); env[name] = val or "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"30": True, "31": True, "41": True, "63": True}


def test_split_items_beyond_the_window_tables():
    """A split item of more keys than DEEP_WINDOWS windows of cap1 / 2 slots cover (5.2e5 keys with 8-byte keys) that the giant path did not take — the 65th ..
    80th root beyond GIANT_MIN keys of a batch — lists its small pieces as runs too, DEEP_WINDOWS windows at a time (it used to list every small piece on its own:
    up to 8192 entries against a list sized for 4 n / cap1 + 1). GKC_MAX_SUB_BITS=0 makes every partition one sub-bucket, i.e. one root: 80 partitions of > 5.3e5
    k-mers each are 64 giants and 16 such items. Records and histogram against the oracle; the independent checksum kernel agrees as well."""
    import json, os, subprocess, sys
    code = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart
gkc = ge.load().gkc
k, m, parts, n, L = 31, 8, 80, 600000, 150
rep = simple_repart(m, parts)
c = gkc.Counter(0); c.configure(k, m, parts, rep)
db, do = c.synth_reads_device(11, n, L, 2000000, 10000)
bases = c.device_to_host(db, n * L); offs = np.arange(n + 1, dtype=np.uint64) * L
cs, nv = c.kmer_checksum_device(db, do, n, n * L)
c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
ref = gko.Dsk(bases, offs, k, m, parts, rep, threads=8)
big = sum(1 for p in range(parts) if int(ref.part_records(p).view(np.uint64).reshape(-1, 2)[:, 1].sum()) > 530000)      # k-mers of the partition = the sum of its abundances
ok = all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts)) and np.array_equal(c.histogram(), ref.histogram())
got = c.result_checksum()
print(json.dumps({"ok": bool(ok), "checksum": (int(got[0]), int(got[1])) == (int(cs), int(nv)), "roots": int(c.stats()["oversize_buckets"]), "big": int(big)}))
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ); env["GKC_MAX_SUB_BITS"] = "0"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["checksum"], res
    assert res["roots"] == 80 and res["big"] >= 68, res                     # every partition went to the split levels, (nearly) all beyond the window tables


def test_read_length_statistics(gkc):
    """gkc_stats.seq_len_min / max / sq_sum: BankStats::update (BankKmers.hpp:176-186) over every pushed read, several pushes"""
    reads = synth_reads(3000, 9000, 150, seed=5, n_rate=0.001, ragged=True) + [b"ACGT" * 3, b"A" * 700]
    lens = np.array([len(r) for r in reads], dtype=np.uint64)
    c = gkc.Counter(0); c.configure(21, 8, 4, simple_repart(8, 4))
    c.begin_pass(0)
    half = len(reads) // 2
    for ch in (reads[:half], reads[half:]):
        b, o = gko.pack_reads(ch); c.push_reads(b, o)
    c.finish_pass()
    st = c.stats()
    assert st["nb_sequences"] == len(reads) and st["nb_bases"] == int(lens.sum())
    assert st["seq_len_min"] == int(lens.min()) and st["seq_len_max"] == int(lens.max()) and st["seq_len_sq_sum"] == int((lens * lens).sum())
