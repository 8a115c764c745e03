"""The HIP path against outputs of the reference ITSELF (tests/golden/reference_run/*.npz: what the reference's own dbgh5 wrote for these
FASTA inputs): device FASTA parse -> count -> per-dataset records, histogram, Bloom arrays of the three kinds (gkc_bloom_insert_solid),
MPHF stream (gkc_mphf_build_solid + gkc_mphf_save); and the .h5 the C++ layer writes with the native HDF5 writer."""
import glob
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.h5mini import H5Mini
from tests.test_reference_run import FIX, load, freq_order_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gkc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return ge.load().gkc


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_device_equals_reference_run(gkc, path):
    z, k, m, nbpart, table, parts = load(path)
    freq = freq_order_of(z, m)                                     # the reference's own minimFrequency table where the fixture carries it
    if path.endswith("k21_freq.npz"):                              # one partition, no stored order: any valid frequency order gives the same dataset
        freq = np.arange(4 ** m, dtype=np.uint32)
    c = gkc.Counter(0); c.set_solidity(2, 2147483647, 10000); c.configure(k, m, nbpart, table, freq_order=freq)
    c.begin_pass(0)
    assert c.push_fastx(bytes(z["fasta"])) == len(z["fasta"])      # the text is parsed on the device
    c.finish_pass()
    if "_auto" in path:
        # -abundance-min auto (SortingCountAlgorithm.cpp:418-444, CountProcessorCutoff.hpp:88-99): the cut-off of the first count's histogram (Histogram::compute_threshold
        # with a floor of 3; the host-side rule, here the oracle's restatement of it applied to the DEVICE's histogram) is the abundance-min of the count that is written
        from oracle import gko
        amin = int(gko.histogram_cutoff(c.histogram(), 3)[0])
        c.set_solidity(amin, 2147483647, 10000)
        c.begin_pass(0); c.push_fastx(bytes(z["fasta"])); c.finish_pass()
    assert c.stats()["kmers_nb_solid"] == int(z["nb_solid_kmers"])
    for p in range(nbpart):
        lo, hi, ab = c.partition(0, p)
        got = [(int(a) | (int(b) << 64), int(x)) for a, b, x in zip(lo, hi, ab)]
        assert got == parts[p], p
    h = c.histogram()
    assert np.array_equal(h[1:len(z["histogram_abundance"]) + 1], z["histogram_abundance"])
    if "bloom" in z:
        kind = bytes(z["bloom_type"]).decode(); size = int(bytes(z["bloom_size"]).decode()); nh = int(bytes(z["bloom_nb_hash"]).decode())
        b = gkc.Bloom(c, kind, size, nh, k); b.insert_solid()
        assert np.array_equal(b.array(), z["bloom"]), kind
        b.close()
    if "mphf" in z:
        mp = gkc.Mphf(c)
        assert np.array_equal(mp.save(), z["mphf"])
        mp.close()


def test_h5_written_by_the_cpp_layer(gkc, tmp_path):
    """gkc_dsk -> out.h5 (native writer): datasets, attributes and values equal what the reference's dbgh5 stored for the same FASTA
    (k31_2parts_mphf fixture); the partition layout is this build's own Repartitor, so k-mers are compared as a set and per dataset against
    the stored minimRepart"""
    built = os.path.join(ge.ROOT, "gatb-core_amd", "host")
    subprocess.run(["make", "-C", built], check=True, capture_output=True)
    z, k, m, nbpart, table, parts = load(os.path.join(os.path.dirname(FIX[0]), "k31_2parts_mphf.npz"))
    fa = tmp_path / "in.fa"; fa.write_bytes(bytes(z["fasta"]))
    out = str(tmp_path / "ours")
    r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-nb-partitions", "2", "-out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    h = H5Mini(open(out + ".h5", "rb").read())
    assert h.listdir("/") == ["configuration", "dsk", "histogram", "minimizers"]
    a = h.attrs("/")
    assert a["kmer_size"] == "31" and a["nb_solid_kmers"] == str(int(z["nb_solid_kmers"])) and a["state"] == "7"
    assert h.attrs("/dsk/solid")["nb_partitions"] == "2" and h.listdir("/dsk/solid") == ["0", "1"]
    got = []
    for p in range(2):
        d = h.dataset("/dsk/solid/%d" % p)
        assert d.dtype.names == ("value", "abundance") and d.dtype.itemsize == 16
        assert np.all(np.diff(d["value"].astype(np.uint64)) > 0)                     # ascending inside a dataset
        got += list(zip(d["value"].tolist(), d["abundance"].tolist()))
    assert sorted(got) == sorted(x for p in parts for x in p)                         # the reference's solid set, counts included
    hh = h.dataset("/histogram/histogram")
    assert hh["index"].tolist() == z["histogram_index"].tolist() and np.array_equal(hh["abundance"], z["histogram_abundance"])
    assert int(h.dataset("/histogram/cutoff")[0]) == int(z["cutoff"]) and int(h.dataset("/histogram/nbsolidsforcutoff")[0]) == int(z["nbsolidsforcutoff"])
    rep = h.dataset("/minimizers/minimRepart")
    assert len(rep) == len(z["minimRepart"]) and int(rep[:2].view("<u2")[0]) == 2


@pytest.mark.parametrize("tag,extra", [("k21_freq_4parts", ["-minimizer-type", "1", "-repartition-type", "1"]),
                                       ("k21_lexi_grouped_parts", ["-repartition-type", "1"]),
                                       ("k21_default_parts", [])])
def test_cpp_repartitor_reproduces_the_reference_tables(gkc, tmp_path, tag, extra):
    """the Repartitor the C++ layer builds from the device statistics (gkc_count_mmers, gkc_sample_minimizers -> computeFrequencies /
    justGroup / justGroupLexi restated on the host) equals the reference's own /minimizers/minimRepart and /minimizers/minimFrequency BYTE FOR BYTE on the
    fixture FASTA — the modes GraphUnitigs / bcalm2 force (-minimizer-type 1 -repartition-type 1), the lexicographic grouping, and the default mode (computeDistrib on the
    kx-mers per minimizer of the exact sample, gkc_sample_exact); with identical tables
    every /dsk/solid/<p> dataset is the reference's dataset, record for record. (Scope of the identity: DESIGN.md section 12.)"""
    built = os.path.join(ge.ROOT, "gatb-core_amd", "host")
    subprocess.run(["make", "-C", built], check=True, capture_output=True)
    z, k, m, nbpart, table, parts = load(os.path.join(os.path.dirname(FIX[0]), tag + ".npz"))
    fa = tmp_path / "in.fa"; fa.write_bytes(bytes(z["fasta"]))
    out = str(tmp_path / "ours")
    r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-nb-partitions", str(nbpart), "-out", out] + extra,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    h = H5Mini(open(out + ".h5", "rb").read())
    rep = np.asarray(h.dataset("/minimizers/minimRepart"), dtype=np.uint8)
    assert np.array_equal(rep, z["minimRepart"]), "minimRepart differs from the reference's table"
    if "minimFrequency" in z:
        fr = np.asarray(h.dataset("/minimizers/minimFrequency"), dtype=np.uint8)
        assert np.array_equal(fr, z["minimFrequency"]), "minimFrequency differs from the reference's table"
    for p in range(nbpart):                                        # identical tables => identical datasets, in order
        d = h.dataset("/dsk/solid/%d" % p)
        assert list(zip(d["value"].tolist(), d["abundance"].tolist())) == parts[p], p
