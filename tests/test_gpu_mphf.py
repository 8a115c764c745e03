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


def test_mphf_of_solid_kmers_and_abundance_map(gkc):
    k, m, parts = 31, 10, 16
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
        order += [int(x) for x in lo]
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
