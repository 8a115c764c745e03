"""Device MPHF (gkc_mphf_*) against the oracle's restatement of BooPHF as GATB instantiates it: same level bit arrays and rank samples
(the mphf::save byte stream is compared byte for byte), same codes for keys and non-keys, and the abundance map of MPHFAlgorithm::populate."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gkc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return ge.load().gkc


@pytest.mark.parametrize("k,n", [(31, 1), (31, 2), (31, 130), (31, 20000), (21, 3000), (63, 5000), (45, 64)])
def test_mphf_equals_boophf(gkc, k, n):
    rng = np.random.default_rng(k * 1000 + n)
    keys = sorted({int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(n)})
    others = [int.from_bytes(rng.bytes(16), "little") & (4 ** k - 1) for _ in range(500)]
    c = gkc.Counter(0)
    dm = gkc.Mphf(c, keys, k); om = gko.Mphf(keys, k)
    assert dm.size == len(keys)
    codes = dm.lookup(keys)
    assert sorted(codes.tolist()) == list(range(len(keys)))               # minimal + perfect (TestMPHF.cpp:225-246)
    assert np.array_equal(codes, om.lookup(keys))
    assert np.array_equal(dm.lookup(others), om.lookup(others))           # non-keys: same false positives, same misses
    assert np.array_equal(dm.save(), om.save())                           # byte stream of mphf::save
    dm.close()


@pytest.mark.parametrize("k,n,rmin", [(31, 200000, 1000), (63, 120000, 500), (21, 60000, 64), (31, 3_000_000, None)])
def test_mphf_region_build_equals_boophf(gkc, monkeypatch, k, n, rmin):
    """Levels of >= GKC_MPHF_REGIONS_MIN keys (default 2^21) are built region by region in LDS (k_mphf_regions / k_mphf_region_build): same level arrays and rank samples as
    the atomic path and as the oracle's BooPHF — the save() stream byte for byte —, same codes. The threshold is lowered so that several levels of a test-size key
    set take the path (and the last ones the atomic path); 3e6 keys take it by default; the ordered rebuild (keys left after 24 levels: never on real sets) is forced
    through GKC_MPHF_ORDERED and must give the same bytes."""
    if rmin is not None:
        monkeypatch.setenv("GKC_MPHF_REGIONS_MIN", str(rmin))
    rng = np.random.default_rng(k * 7 + n)
    raw = np.frombuffer(rng.bytes(16 * n), dtype=np.uint64).reshape(n, 2)
    keys = sorted({(int(a) | (int(b) << 64)) & (4 ** k - 1) for a, b in raw})
    c = gkc.Counter(0)
    dm = gkc.Mphf(c, keys, k)
    monkeypatch.setenv("GKC_MPHF_REGIONS", "0")
    d0 = gkc.Mphf(c, keys, k)                                             # the atomic / flag / stable-compaction path
    assert np.array_equal(dm.save(), d0.save())
    monkeypatch.delenv("GKC_MPHF_REGIONS")
    monkeypatch.setenv("GKC_MPHF_ORDERED", "1")
    d1 = gkc.Mphf(c, keys, k)
    assert np.array_equal(dm.save(), d1.save())
    monkeypatch.delenv("GKC_MPHF_ORDERED")
    sample = keys[:: max(1, len(keys) // 20000)]
    codes = dm.lookup(sample)
    assert np.array_equal(codes, d0.lookup(sample)) and len(set(codes.tolist())) == len(sample) and int(codes.max()) < len(keys)
    if n <= 200000:
        om = gko.Mphf(keys, k)
        assert np.array_equal(dm.save(), om.save()) and np.array_equal(codes, om.lookup(sample))
    dm.close(); d0.close(); d1.close()


@pytest.mark.parametrize("k,rmin", [(31, None), (31, 64), (63, 200), (21, 1000)])
def test_mphf_of_solid_kmers_and_abundance_map(gkc, monkeypatch, k, rmin):
    """rmin: GKC_MPHF_REGIONS_MIN lowered so that the build takes its by-region path on a test-size key set (k_mphf_region_build;
    and the map of populate() is checked on what that build gives"""
    if rmin is not None:
        monkeypatch.setenv("GKC_MPHF_REGIONS_MIN", str(rmin))
    m, parts = 10, 16
    reads = synth_reads(4000, 20000, 150, seed=21, n_rate=0.001)
    bases, offs = gko.pack_reads(reads)
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.set_solidity(2, 2147483647, 10000); c.configure(k, m, parts, rep)
    c.begin_pass(0); c.push_reads(bases, offs); c.finish_pass()
    ref = gko.Dsk(bases, offs, k, m, parts, rep, abundance_min=2)
    solid = ref.all_counts()                                              # {kmer: abundance}
    order = []                                                            # getSolidKmers() order: dataset by dataset, ascending
    for p in range(parts):
        lo, hi, ab = ref.part(p)
        order += [int(a) | (int(b) << 64) for a, b in zip(lo, hi)]
    dm = gkc.Mphf(c)
    om = gko.Mphf(order, k)
    assert dm.size == len(order) == ref.stats["kmers_nb_solid"]
    assert np.array_equal(dm.save(), om.save())
    amap, above = dm.abundance_map()
    codes = om.lookup(order)
    want = np.zeros(len(order), np.uint8)
    for x, cd in zip(order, codes):
        want[int(cd)] = gko.abundance_index(solid[x])
    assert np.array_equal(amap, want) and above == 0


def test_mphf_reference_check1(gkc):
    """TestMPHF.cpp:95-161: 130 k-mers of the 140-nt sequence at k=11, through the device counter and the device MPHF"""
    seq = ("CGCTACAGCAGCTAGTTCATCATTGTTTATCAATGATAAAATATAATAAGCTAAAAGGAAACTATAAATA"
           "ACCATGTATAATTATAAGTAGGTACCTATTTTTTTATTTTAAACTGAAATTCAATATTATATAGGCAAAG")
    bases, offs = gko.pack_reads([seq])
    c = gkc.Counter(0); c.configure(11, 8, 4, simple_repart(8, 4)); c.count(bases, offs)
    dm = gkc.Mphf(c)
    assert dm.size == 130
    amap, _ = dm.abundance_map()
    assert len(amap) == 130 and set(amap.tolist()) == {1}
    keys = sorted(c.all_counts().keys())
    assert sorted(dm.lookup(keys).tolist()) == list(range(130))


def test_bloom_and_mphf_at_share_size(gkc, monkeypatch):
    """BASELINE configs[4] at one GPU's share (Bloom + MPHF at 5.8e8 solid k-mers were checked only inside bench.py): k = 31, 1e8 reads, abundance-min 2.
    Bloom (neighbor kind, 11 bits per k-mer, 7 hashes): the region build == the atomic build, byte for byte; no false negative; contains8 by region == contains8 by gathers
    for every solid k-mer. MPHF: the region-build stream == the atomic-path stream; every cell of the abundance map written (the codes are a bijection onto [0, n)).
    Against the oracle on the first 1e6 solid k-mers: every bit the oracle's hash functions set for them is set in the device's array (the same functions at an array of
    6.4e9 bits), and the oracle's contains8 over the device's array gives the device's answers; their MPHF codes are distinct and below n."""
    import torch
    k, m, L, n, parts = 31, 10, 150, 100_000_000, 4096
    c = gkc.Counter(0); c.configure(k, m, parts, simple_repart(m, parts)); c.set_solidity(2, 2147483647, 10000)
    db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    c.device_free(db); c.device_free(do)
    ns = c.stats()["kmers_nb_solid"]
    assert ns >= 500_000_000
    bits = ns * 11
    bl = gkc.Bloom(c, "neighbor", bits, 7, k); bl.insert_solid()
    a_reg = bl.array()
    monkeypatch.setenv("GKC_BLOOM_ATOMIC", "1")
    bl0 = gkc.Bloom(c, "neighbor", bits, 7, k); bl0.insert_solid()
    assert np.array_equal(a_reg, bl0.array())
    bl0.close(); monkeypatch.delenv("GKC_BLOOM_ATOMIC")
    nq, npos = bl.query_solid(False)
    assert nq == npos == ns                                                # no false negative
    o_reg = torch.zeros(ns, dtype=torch.uint8, device="cuda"); o_gat = torch.zeros(ns, dtype=torch.uint8, device="cuda")
    r_reg = bl.query_solid(True, d_out=o_reg.data_ptr())
    monkeypatch.setenv("GKC_BLOOM_GATHER", "1")
    r_gat = bl.query_solid(True, d_out=o_gat.data_ptr())
    monkeypatch.delenv("GKC_BLOOM_GATHER")
    assert r_reg == r_gat and bool(torch.equal(o_reg, o_gat))
    keys = []; p = 0
    while len(keys) < 1_000_000:
        lo, hi, ab = c.partition(0, p); keys += lo.tolist(); p += 1
    ob = gko.Bloom("neighbor", bits, 7, k)
    oa = np.ctypeslib.as_array(gko.lib().gko_bloom_array(ob._h), shape=(ob.nbytes,))
    assert ob.nbytes == bl.nbytes
    ob.insert(keys)
    assert np.array_equal(oa & a_reg, oa)                                  # the oracle's bits of these k-mers are set in the device's array
    oa[:] = a_reg                                                         # ... and over the device's array the oracle answers what the device answered
    assert bool(ob.contains(keys).all())
    assert np.array_equal(ob.contains8(keys), o_reg[: len(keys)].cpu().numpy())
    del o_reg, o_gat, ob, oa
    bl.close()
    mp = gkc.Mphf(c)
    assert mp.size == ns
    s_reg = mp.save()
    amap, above = mp.abundance_map()
    assert len(amap) == ns and bool((amap != 0).all())                    # every cell written: n keys, n cells, none twice
    codes = mp.lookup(keys[:200_000])
    assert len(set(codes.tolist())) == 200_000 and int(codes.max()) < ns
    monkeypatch.setenv("GKC_MPHF_REGIONS", "0")
    mp0 = gkc.Mphf(c)
    assert np.array_equal(s_reg, mp0.save())
    assert np.array_equal(codes, mp0.lookup(keys[:200_000]))
    mp.close(); mp0.close()
