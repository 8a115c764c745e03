"""Pins oracle/gkc_oracle.c against every known-answer vector the reference's own unit tests hold for the path
(SURVEY.md §8c). The vectors are data extracted by tools/make_reference_test_vectors.py."""
import math

import numpy as np
import pytest

from oracle import gko
from tests.util import naive_counts, revcomp_int, simple_repart, str2int, synth_reads


def run_dsk(seqs, k, nks=1, nks_max=2147483647, m=None, parts=4, **kw):
    if m is None:
        m = min(k - 1, 10)           # ConfigurationAlgorithm.cpp:249-251 with default -minimizer-size 10
    bases, offs = gko.pack_reads(seqs)
    return gko.Dsk(bases, offs, k, m, parts, simple_repart(m, parts), abundance_min=nks, abundance_max=nks_max, **kw)


def test_math_constants(ref_vectors):
    L = gko.lib(); mv = ref_vectors["math"]
    x, k, exp = mv["revcomp"];        assert L.gko_revcomp64(x, k) == exp          # TestMath.cpp:94
    x, s, exp = mv["simplehash16"];   assert L.gko_simplehash16_ni64(x, s) == exp  # :96 (2-term NativeInt64 variant)
    x, s, exp = mv["hash1"];          assert L.gko_hash64(x, s) == exp             # :97
    x, exp = mv["oahash"];            assert L.gko_oahash64(x) == exp              # :98


def test_kmer_direct_and_canonical(ref_vectors):
    v = ref_vectors["kmer_k3"]                                                     # TestKmer.cpp:141-190
    r = gko.kmers(v["seq"], 3)
    assert r["fwd_lo"].tolist() == v["direct"]
    assert r["can_lo"].tolist() == v["canonical"]
    assert r["valid"].all()


def test_badchar_validity(ref_vectors):
    v = ref_vectors["badchar_k11"]                                                 # TestKmer.cpp:542-569
    r = gko.kmers(v["seq"], v["k"])
    assert len(v["table"]) == len(r["valid"])
    for i, (kmer, valid) in enumerate(v["table"]):
        assert bool(r["valid"][i]) == valid
        assert int(r["fwd_lo"][i]) == str2int(kmer.replace("N", "G"))              # "N replaced by G"


def test_minimizer_table(ref_vectors):
    v = ref_vectors["minimizer_k15_m7"]                                            # TestKmer.cpp:435-510
    mins, valid = gko.minimizers(v["seq"], v["k"], v["m"])
    km = gko.kmers(v["seq"], v["k"])
    assert len(mins) == len(v["table"])
    for i, (kmer, mini, pos, changed) in enumerate(v["table"]):
        assert int(km["can_lo"][i]) == str2int(kmer)
        assert int(mins[i]) == str2int(mini)


def test_minimizer_bruteforce_kmc2_property():
    """TestKmer.cpp:265-310: minimizer == min over window of allowed m-mers (here: canonical model)."""
    reads = synth_reads(20, 2000, 150, seed=3)
    for (k, m) in [(15, 7), (21, 8), (31, 10), (11, 5), (41, 10)]:
        mask = 4 ** m - 1
        for r in reads[:6]:
            mins, _ = gko.minimizers(r, k, m)
            km = gko.kmers(r, k)
            for i in range(len(mins)):
                f = (int(km["fwd_hi"][i]) << 64) | int(km["fwd_lo"][i])
                best = mask
                for j in range(k - m + 1):
                    x = (f >> (2 * j)) & mask
                    c = min(x, revcomp_int(x, m))
                    s = format(c, "0%db" % (2 * m))
                    nts = [s[2 * t:2 * t + 2] for t in range(m)]
                    if any(nts[t] == "00" and nts[t + 1] == "00" for t in range(1, m - 1)):
                        c = mask                                                     # AA anywhere but as prefix
                    best = min(best, c)
                assert best == int(mins[i])


def test_dsk_check1(ref_vectors):
    v = ref_vectors["dsk_check1"]                                                  # TestDSK.cpp:147-241
    for name, k, nks, expected in v["cases"]:
        d = run_dsk(v[name], k, nks)
        assert d.stats["kmers_nb_solid"] == expected, (name, k, nks)


def test_dsk_check2(ref_vectors):
    v = ref_vectors["dsk_check2"]                                                  # TestDSK.cpp:254-305
    d = run_dsk([v["seq"]], v["k"], 1)
    got = d.all_counts()
    assert sorted(got) == sorted(v["values"])
    assert sum(got) & (2 ** 64 - 1) == v["checksum"]
    # same through the 128-bit Count record layout used for span 64 (k=31 fits either span in the reference)
    assert set(got.values()) == {1}


def test_dsk_perbank_sum_rows(ref_vectors):
    for key in ("dsk_perbank1", "dsk_perbank2"):                                   # TestDSK.cpp:482-612 (sum solidity)
        v = ref_vectors[key]
        for case in v["sum_cases"]:
            if len(case) == 2:
                nmin, exp = case; nmax = 2 ** 30
            else:
                nmin, nmax, exp = case
            d = run_dsk(v["seqs"], v["k"], nmin, nmax)
            assert d.stats["kmers_nb_solid"] == exp, (key, case)


def test_dsk_all_kmers_bank():
    """TestDSK.cpp:615-678: a bank holding all 4^k k-mers -> 4^k/2 canonical k-mers of abundance 2 (odd k)."""
    k = 9
    seqs = []
    for x in range(4 ** k):
        seqs.append("".join("ACTG"[(x >> (2 * (k - 1 - i))) & 3] for i in range(k)))
    d = run_dsk(seqs, k, 1, m=8)
    assert d.stats["kmers_nb_distinct"] == 4 ** k // 2
    for nks, exp in [(1, 4 ** k // 2), (2, 4 ** k // 2), (3, 0)]:
        assert run_dsk(seqs, k, nks, m=8).stats["kmers_nb_solid"] == exp


def test_debloom_cfp_pins_basic_bloom(ref_vectors):
    """TestDebloom.cpp:84-170: 20 critical false positives = neighbours of solid k-mers that the BLOOM_BASIC filter
    (size = nbSolid * nbitsPerKmer, nbHash = floor(0.7 nbits), BloomAlgorithm.cpp:158-165; nbits from
    DebloomAlgorithm.cpp:628-650 DEBLOOM_ORIGINAL) wrongly contains. Pins hash1, the seeds and bit positions."""
    v = ref_vectors["debloom_k11"]; k = v["k"]
    d = run_dsk([v["seq"]], k, 1, m=v["m"])
    solid = d.all_counts()
    assert len(solid) == len(v["seq"]) - k + 1
    lg2 = math.log(2)
    nbits = np.float32(math.log(16 * k * (lg2 * lg2)) / (lg2 * lg2))
    size = int(np.float32(len(solid)) * nbits)            # (u_int64_t)(solidKmersNb * NBITS_PER_KMER), float arithmetic
    nb_hash = int(math.floor(np.float32(0.7) * nbits)) if False else int(math.floor(0.7 * float(nbits)))
    bl = gko.Bloom("basic", size, nb_hash, k)
    bl.insert(list(solid))
    mask = 4 ** k - 1
    cfp = set()
    for x in solid:
        for y in (x, revcomp_int(x, k)):
            for j in range(4):
                n = ((y << 2) | j) & mask
                c = min(n, revcomp_int(n, k))
                if c not in solid and bl.contains([c])[0]:
                    cfp.add(c)
    assert cfp == set(v["cfp"])


def test_superkmer_wire_roundtrip_and_partition_contract():
    """A6/B1: encode->decode regenerates exactly the canonical k-mers of the super-k-mer; every k-mer of partition p
    has repart[minimizer]==p; datasets ascending (PartitionsCommand.cpp:1760-1801)."""
    reads = synth_reads(300, 5000, 150, seed=5, n_rate=0.002, ragged=True)
    L = gko.lib()
    for (k, m) in [(21, 8), (31, 10), (63, 10), (33, 9)]:
        for r in reads[:40]:
            mn, st, nb, nv, ni = gko.superkmers(r, k, m)
            km = gko.kmers(r, k)
            if len(r) >= k:
                assert nv + ni == len(r) - k + 1
                assert nv == int(km["valid"].sum())
                assert int(nb.sum()) == nv
            for a, s, n in zip(mn.tolist(), st.tolist(), nb.tolist()):
                assert n <= (28 if k <= 31 else 60)
                buf = np.zeros(256, np.uint8)
                ln = L.gko_superkmer_encode(r[s:], k, n, buf)
                assert ln == 1 + (k + n - 1 + 3) // 4
                lo = np.zeros(256, np.uint64); hi = np.zeros(256, np.uint64)
                import ctypes
                nbk = ctypes.c_uint(0)
                used = L.gko_superkmer_decode(buf.ctypes.data, k, lo, hi, ctypes.byref(nbk))
                assert used == ln and nbk.value == n
                assert lo[:n].tolist() == km["can_lo"][s:s + n].tolist()
                assert hi[:n].tolist() == km["can_hi"][s:s + n].tolist()
        parts = 8
        rep = simple_repart(m, parts)
        bases, offs = gko.pack_reads(reads)
        d = gko.Dsk(bases, offs, k, m, parts, rep)
        ref = naive_counts(reads, k)
        assert d.all_counts() == ref
        assert d.stats["kmers_nb_valid"] == sum(ref.values())
        for p in range(parts):
            lo, hi, ab = d.part(p)
            keys = [(int(b) << 64) | int(a) for a, b in zip(lo, hi)]
            assert keys == sorted(keys) and len(set(keys)) == len(keys)
            for key in keys[:50]:
                s = "".join("ACTG"[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))
                mins, _ = gko.minimizers(s, k, m)
                assert rep[mins[0]] == p


def test_frequency_order_minimizers_same_counts():
    """-minimizer-type 1: a different minimizer order changes the partitioning, never the (k-mer,count) set."""
    reads = synth_reads(200, 4000, 150, seed=9)
    k, m, parts = 25, 8, 4
    L = gko.lib()
    counts = np.zeros(4 ** m, np.uint32)
    for r in reads[:50]:
        L.gko_count_mmers(r, len(r), m, counts)
    freq = np.zeros(4 ** m, np.uint32)
    L.gko_freq_order_from_counts(m, counts, freq)
    assert freq[4 ** m - 1] == 4 ** m - 1
    bases, offs = gko.pack_reads(reads)
    rep = simple_repart(m, parts)
    d1 = gko.Dsk(bases, offs, k, m, parts, rep, freq_order=freq)
    d0 = gko.Dsk(bases, offs, k, m, parts, rep)
    assert d1.all_counts() == d0.all_counts() == naive_counts(reads, k)
    # brute-force the frequency-order minimizer definition (SURVEY §8c')
    mask = 4 ** m - 1
    for r in reads[:5]:
        mins, _ = gko.minimizers(r, k, m, freq)
        km = gko.kmers(r, k)
        for i in range(len(mins)):
            f = int(km["fwd_lo"][i]); best = mask
            for j in range(k - m + 1):
                x = (f >> (2 * j)) & mask; c = min(x, revcomp_int(x, m))
                if (int(freq[c]), c) < (int(freq[best]), best):
                    best = c
            assert best == int(mins[i])


def test_multipass_datasets():
    reads = synth_reads(150, 3000, 150, seed=11)
    k, m, parts, passes = 21, 8, 4, 3
    bases, offs = gko.pack_reads(reads)
    rep = simple_repart(m, parts)
    d = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes)
    assert d.all_counts() == naive_counts(reads, k)
    for ds in range(parts * passes):
        lo, hi, ab = d.part(ds)
        for key in lo[:20].tolist():
            s = "".join("ACTG"[(key >> (2 * (k - 1 - i))) & 3] for i in range(k))
            mins, _ = gko.minimizers(s, k, m)
            assert rep[mins[0]] == ds % parts and int(mins[0]) % passes == ds // parts


def test_bloom_no_false_negative_and_neighbor_symmetry():
    """TestContainer.cpp:63-128 property (no false negatives) for the three kinds, 64- and 128-bit items; the
    'neighbor' kind is strand-symmetric; contains8 == 8 single queries."""
    rng = np.random.default_rng(2)
    for k in (11, 31, 41, 63):
        keys = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(500)]
        others = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(500)]
        for kind in ("basic", "cache", "neighbor"):
            b = gko.Bloom(kind, 500 * 12, 8, k)
            b.insert(keys)
            assert b.contains(keys).all()
            fp = b.contains(others).mean()
            assert fp < 0.2
            if kind == "neighbor":
                rc = [revcomp_int(x, k) for x in keys]
                assert b.contains(rc).all()
                c8 = b.contains8(others[:50])
                mask = 4 ** k - 1
                for x, bits in zip(others[:50], c8.tolist()):
                    exp = 0
                    for j in range(4):
                        exp |= int(b.contains([((x << 2) & mask) + j])[0]) << j
                        exp |= int(b.contains([(x >> 2) + (j << (2 * (k - 1)))])[0]) << (4 + j)
                    assert bits == exp


def test_bloom_seeds_documented_values():
    s = np.zeros(10, np.uint64)
    gko.lib().gko_bloom_seeds(0, s)
    assert int(s[0]) == 0xffaa54ffe6e6e6e7 and int(s[1]) == 0x1140aada557088a4      # SURVEY §8c' (oracle-verified)


def test_histogram_cutoff_properties():
    """Histogram::compute_threshold restated (Histogram.cpp:61-190): properties its code guarantees. The known answers (cutoff and
    nbsolidsforcutoff written by the reference itself) are in tests/test_reference_run.py."""
    x = np.arange(10001)
    h = np.zeros(10001, np.uint64)
    h[1:] = (1e6 * np.exp(-(x[1:] - 1) / 0.8) + 2e4 * np.exp(-0.5 * ((x[1:] - 30) / 5.5) ** 2)).astype(np.uint64)
    cut, nbs, peak = gko.histogram_cutoff(h, 3)
    assert peak == 30 and 3 <= cut < peak                          # valley between the error k-mers and the coverage peak
    assert nbs == int(h[cut:].sum())
    mono = np.zeros(10001, np.uint64); mono[1:200] = np.arange(199, 0, -1, dtype=np.uint64) * 1000   # never increases: default threshold
    assert gko.histogram_cutoff(mono, 3)[0] == 3
    assert gko.histogram_cutoff(mono, 7)[0] == 7


def test_mphf_check1_known_answers():
    """TestMPHF.cpp:95-161 MPHF_check1: k=11 on a 140-nt sequence -> 130 solid k-mers, the MPHF is a bijection onto [0,130)
    (TestMPHF.cpp:209-246 checks exactly that), the abundance map has 130 cells"""
    seq = ("CGCTACAGCAGCTAGTTCATCATTGTTTATCAATGATAAAATATAATAAGCTAAAAGGAAACTATAAATA"
           "ACCATGTATAATTATAAGTAGGTACCTATTTTTTTATTTTAAACTGAAATTCAATATTATATAGGCAAAG")
    k = 11
    bases, offs = gko.pack_reads([seq])
    d = gko.Dsk(bases, offs, k, 8, 4, simple_repart(8, 4))
    keys = sorted(d.all_counts().keys())
    assert len(keys) == len(seq) - k + 1 == 130
    m = gko.Mphf(keys, k)
    codes = m.lookup(keys)
    assert sorted(codes.tolist()) == list(range(130))
    # stream layout of mphf::save (BooPHF.h:933-958): gamma, nb_levels, lastbitsetrank, nelem
    s = m.save()
    assert s[:8].view(np.float64)[0] == 3.0 and s[8:12].view(np.int32)[0] == 25
    assert s[12:20].view(np.uint64)[0] == 130 and s[20:28].view(np.uint64)[0] == 130


def test_abundance_discretization():
    # MapMPHF.hpp:96-145: steps 1 (x70), 2 (x15), 10 (x40), 20 (x25), 100 (x40), 200 (x25), 1000 (x40); MPHFAlgorithm.cpp:253-266
    assert [gko.abundance_index(a) for a in (0, 1, 70, 71, 72, 100, 500, 501)] == [0, 1, 70, 70, 71, 85, 125, 125]
    assert gko.abundance_index(50000) == 255 and gko.abundance_index(49999) == 254


def test_parallel_oracle_equals_sequential_oracle():
    """gko_dsk_run_mt (the cpu_baseline of bench.py: fillPartitions over the reads, fillSolidKmers over the partitions, like the reference's
    Dispatcher) gives exactly the sequential restatement's datasets, statistics and histogram"""
    from tests.util import simple_repart, synth_reads
    reads = synth_reads(1500, 9000, 150, seed=9, n_rate=0.002, ragged=True)
    bases, offs = gko.pack_reads(reads)
    for k, m, parts, passes in ((31, 8, 7, 1), (41, 7, 5, 2)):
        rep = simple_repart(m, parts)
        a = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes, abundance_min=2)
        b = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes, abundance_min=2, threads=5)
        assert a.stats == b.stats and np.array_equal(a.histogram(), b.histogram())
        for d in range(parts * passes):
            assert np.array_equal(a.part_records(d), b.part_records(d))


def test_partition_restricted_oracle_equals_the_full_run():
    """gko_dsk_run_parts (the exact count of a few sampled partitions of a full-size input, tests/test_gpu_parity.py): on the kept partitions every dataset, k-mer and
    super-k-mer count is that of the unrestricted run; the other partitions are empty; the whole-input statistics stay"""
    for k, m, parts, passes, threads in ((31, 8, 12, 1, 1), (63, 9, 7, 2, 3), (21, 6, 5, 1, 2)):
        reads = synth_reads(800, 6000, 150, seed=70 + k, n_rate=0.002, ragged=True)
        bases, offs = gko.pack_reads(reads)
        rep = simple_repart(m, parts)
        full = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes)
        keep = [0, parts // 2, parts - 1]
        some = gko.Dsk(bases, offs, k, m, parts, rep, nb_passes=passes, threads=threads, only_parts=keep)
        for d in range(parts * passes):
            if d % parts in keep:
                assert np.array_equal(some.part_records(d), full.part_records(d)) and some.part_stats(d) == full.part_stats(d)
            else:
                assert len(some.part_records(d)) == 0 and some.part_stats(d) == (0, 0)
        assert some.stats["kmers_nb_valid"] == full.stats["kmers_nb_valid"] and some.stats["nb_sequences"] == full.stats["nb_sequences"]
        assert 0 < some.stats["kmers_nb_distinct"] < full.stats["kmers_nb_distinct"]
