"""The C++ host layer (gatb-core_amd/host/gatb_gkc.hpp: SortingCountAlgorithm<span>, ICountProcessor<span>, IBloom<T> mirrors)
run on the GPU: the reference's TestDSK known answers through the C++ classes, the processor call protocol, the CLI."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads

pytestmark = pytest.mark.gpu
HOST = os.path.join(ge.ROOT, "gatb-core_amd", "host")


@pytest.fixture(scope="module")
def built():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ge.build()
    return HOST


def test_reference_dsk_tests_through_cpp_classes(built, ref_vectors, tmp_path):
    lines = []
    v = ref_vectors["dsk_check1"]
    for name, k, nks, expected in v["cases"]:
        seqs = v[name]
        lines.append("check1 %d %d %d %d %s" % (k, nks, expected, len(seqs), " ".join(seqs)))
    v = ref_vectors["dsk_check2"]
    lines.append("check2 %d %x %d %s %s" % (v["k"], v["checksum"], len(v["values"]), " ".join("%x" % x for x in v["values"]), v["seq"]))
    f = tmp_path / "vectors.txt"; f.write_text("\n".join(lines) + "\n")
    r = subprocess.run([os.path.join(built, "test_host"), str(f)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "check1=33 check2=1 failures=0" in r.stdout


@pytest.mark.parametrize("k,mtype,fmt", [(31, 0, "fa"), (21, 1, "fq"), (45, 0, "fa.gz"), (31, 0, "fq-multiline")])
def test_cli_dump_matches_oracle(built, tmp_path, k, mtype, fmt):
    """file bank -> device text parser (gkc_push_fastx) -> count; "fq-multiline" is refused by the device parser and read by the host walk"""
    import gzip
    reads = synth_reads(3000, 20000, 150, seed=3, n_rate=0.001, ragged=True)
    fa = tmp_path / ("reads." + fmt)
    if fmt == "fq":
        text = "".join("@r%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)) for i, r in enumerate(reads))
    elif fmt == "fq-multiline":
        text = "".join("@r%d\n%s\n%s\n+\n%s\n" % (i, r.decode()[:70], r.decode()[70:], "I" * len(r)) if len(r) > 70 else
                       "@r%d\n%s\n+\n%s\n" % (i, r.decode(), "I" * len(r)) for i, r in enumerate(reads))
    else:
        text = "".join(">r%d\n%s\n" % (i, "\n".join(r.decode()[j:j + 60] for j in range(0, len(r), 60))) for i, r in enumerate(reads))
    if fmt.endswith(".gz"):
        with gzip.open(fa, "wb") as f:
            f.write(text.encode())
    else:
        fa.write_text(text)
    out = str(tmp_path / "out")
    r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-minimizer-type", str(mtype),
                        "-nb-partitions", "8", "-mphf", "1", "-out", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    info = dict(l.split("\t") for l in open(out + ".info").read().splitlines())
    # the repartition table the C++ layer built (device statistics -> host table) drives the oracle too
    raw = np.fromfile(out + ".minimRepart", dtype=np.uint8)
    nbpart = int(raw[:2].view(np.uint16)[0]); nmin = int(raw[2:10].view(np.uint64)[0])
    table = raw[12:12 + 2 * nmin].view(np.uint16).copy()
    m = int(np.log2(nmin) / 2)
    assert nbpart == 8 and m == min(k - 1, 10)
    freq = None
    if raw[12 + 2 * nmin]:
        freq = np.fromfile(out + ".minimRepart.minimFrequency", dtype=np.uint32)[:nmin].copy()
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, 8, table, freq_order=freq, abundance_min=2)
    rec = 16 if k <= 31 else 32
    for p in range(8):
        got = np.fromfile(out + ".solid.%d" % p, dtype=np.uint8)
        assert np.array_equal(got, ref.part_records(p)), p
        assert len(got) % rec == 0
    assert int(info["kmers_nb_solid"]) == ref.stats["kmers_nb_solid"]
    assert int(info["kmers_nb_distinct"]) == ref.stats["kmers_nb_distinct"]
    assert int(info["kmers_nb_valid"]) == ref.stats["kmers_nb_valid"]
    hist = {int(a): int(b) for a, b in (l.split("\t") for l in open(out + ".histo").read().splitlines())}
    rh = ref.histogram()
    assert hist == {i: int(c) for i, c in enumerate(rh) if c and i > 0}
    # MPHFAlgorithm through the C++ layer: saved hash = BooPHF's stream for the solid k-mers in getSolidKmers() order, abundance map
    order, ab = [], {}
    for p in range(8):
        lo, hi, a = ref.part(p)
        for x, y, z in zip(lo, hi, a):
            key = int(x) | (int(y) << 64); order.append(key); ab[key] = int(z)
    om = gko.Mphf(order, k)
    assert np.array_equal(np.fromfile(out + ".mphf", dtype=np.uint8), om.save())
    amap = np.fromfile(out + ".abundancemap", dtype=np.uint8)
    want = np.zeros(len(order), np.uint8)
    for key, cd in zip(order, om.lookup(order)):
        want[int(cd)] = gko.abundance_index(ab[key])
    assert np.array_equal(amap, want)
    # histogram/cutoff, nbsolidsforcutoff (Histogram::compute_threshold) computed by the C++ layer from the device histogram
    cut = [int(x) for x in open(out + ".cutoff").read().split()]
    assert tuple(cut) == gko.histogram_cutoff(rh, 2)


@pytest.mark.parametrize("k,span", [(31, 96), (47, 128), (31, 64)])
def test_cli_larger_spans_write_the_same_solid_sets(built, tmp_path, k, span):
    """SortingCountAlgorithm<96> / <128> (LargeInt<3>, LargeInt<4>: 32- / 40-byte Count) with k <= 63: the .h5 holds the same k-mers and
    abundances as the natural span's, with the value widened (integer of 64 N bits, LargeInt.hpp:655-660)"""
    from tests.h5mini import H5Mini
    reads = synth_reads(1500, 8000, 120, seed=8, n_rate=0.001)
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(">r%d\n%s\n" % (i, r.decode()) for i, r in enumerate(reads)))
    outs = {}
    for sp in (0, span):
        out = str(tmp_path / ("out%d" % sp))
        cmd = [os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-nb-partitions", "4", "-out", out]
        if sp:
            cmd += ["-span", str(sp)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[sp] = H5Mini(open(out + ".h5", "rb").read())
    nat = 8 if k <= 31 else 16
    wide = span // 32 * 8
    total = 0
    for p in range(4):
        a = outs[0].dataset("/dsk/solid/%d" % p); b = outs[span].dataset("/dsk/solid/%d" % p)
        assert len(a) == len(b) and b.dtype.itemsize == (32 if span <= 96 else 40)
        va = np.frombuffer(a["value"].tobytes(), dtype=np.uint8).reshape(len(a), nat)
        vb = np.frombuffer(b["value"].tobytes(), dtype=np.uint8).reshape(len(b), wide)
        assert np.array_equal(vb[:, :nat], va) and not vb[:, nat:].any()
        assert np.array_equal(a["abundance"], b["abundance"])
        total += len(a)
    assert total > 100


def test_cli_abundance_min_auto(built, tmp_path):
    """-abundance-min auto: a cutoff processor sees every distinct k-mer first (CountProcessorCutoff: Histogram::compute_threshold(3)), its
    threshold becomes the abundance min of the dsk chain (SortingCountAlgorithm.cpp:418-512) — solid sets equal the oracle's at that cutoff"""
    reads = synth_reads(6000, 30000, 150, seed=21, sub_rate=0.01)
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(">r%d\n%s\n" % (i, r.decode()) for i, r in enumerate(reads)))
    out = str(tmp_path / "out")
    k = 25
    r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "auto", "-nb-partitions", "4", "-out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    info = dict(l.split("\t") for l in open(out + ".info").read().splitlines())
    raw = np.fromfile(out + ".minimRepart", dtype=np.uint8)
    nmin = int(raw[2:10].view(np.uint64)[0]); table = raw[12:12 + 2 * nmin].view(np.uint16).copy(); m = int(np.log2(nmin) / 2)
    bases, offs = gko.pack_reads(reads)
    allk = gko.Dsk(bases, offs, k, m, 4, table, abundance_min=1)
    cutoff = gko.histogram_cutoff(allk.histogram(), 3)[0]
    assert cutoff >= 3 and int(info["cutoffs_auto.values"].split()[0]) == cutoff
    ref = gko.Dsk(bases, offs, k, m, 4, table, abundance_min=cutoff)
    for p in range(4):
        assert np.array_equal(np.fromfile(out + ".solid.%d" % p, dtype=np.uint8), ref.part_records(p)), p
    assert int(info["kmers_nb_solid"]) == ref.stats["kmers_nb_solid"] and int(info["kmers_nb_distinct"]) == ref.stats["kmers_nb_distinct"]


def test_cli_album_of_two_files_counts_like_their_concatenation(built, tmp_path):
    """-in a.fa,b.fq.gz (two banks, solidity kind sum): the same solid sets as the single concatenated file; the first file has no final newline"""
    import gzip
    reads = synth_reads(2400, 12000, 140, seed=33, n_rate=0.001)
    a, b = reads[:1000], reads[1000:]
    fa = tmp_path / "a.fa"; fb = tmp_path / "b.fa.gz"; fc = tmp_path / "c.fa"
    ta = "".join(">a%d\n%s\n" % (i, r.decode()) for i, r in enumerate(a)).rstrip("\n")
    tb = "".join(">b%d\n%s\n" % (i, r.decode()) for i, r in enumerate(b))
    fa.write_text(ta)
    with gzip.open(fb, "wb") as f:
        f.write(tb.encode())
    fc.write_text(ta + "\n" + tb)
    outs = []
    for name, uri in (("two", "%s,%s" % (fa, fb)), ("one", str(fc))):
        out = str(tmp_path / name)
        r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", uri, "-kmer-size", "27", "-abundance-min", "2", "-nb-partitions", "4", "-out", out],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(out)
    for p in range(4):
        x = np.fromfile(outs[0] + ".solid.%d" % p, dtype=np.uint8); y = np.fromfile(outs[1] + ".solid.%d" % p, dtype=np.uint8)
        assert len(x) > 0 and np.array_equal(x, y)
    i0 = dict(l.split("\t") for l in open(outs[0] + ".info").read().splitlines()); i1 = dict(l.split("\t") for l in open(outs[1] + ".info").read().splitlines())
    assert i0["kmers_nb_valid"] == i1["kmers_nb_valid"] and i0["seq_number"] == i1["seq_number"] == "2400"


def test_cli_three_passes_release_their_results(built, tmp_path):
    """-nb-passes 3: each pass is drained by the processors and then released on the device (gkc_release_pass); the datasets (part + pass * nb_partitions),
    statistics and the MPHF (built from the chain's solid counts) equal the oracle's"""
    reads = synth_reads(2500, 15000, 150, seed=5, n_rate=0.001)
    fa = tmp_path / "reads.fa"
    fa.write_text("".join(">r%d\n%s\n" % (i, r.decode()) for i, r in enumerate(reads)))
    out = str(tmp_path / "out"); k = 31
    r = subprocess.run([os.path.join(built, "gkc_dsk"), "-in", str(fa), "-kmer-size", str(k), "-abundance-min", "2", "-nb-partitions", "4", "-nb-passes", "3", "-mphf", "1", "-out", out],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    info = dict(l.split("\t") for l in open(out + ".info").read().splitlines())
    raw = np.fromfile(out + ".minimRepart", dtype=np.uint8)
    nmin = int(raw[2:10].view(np.uint64)[0]); table = raw[12:12 + 2 * nmin].view(np.uint16).copy(); m = int(np.log2(nmin) / 2)
    bases, offs = gko.pack_reads(reads)
    ref = gko.Dsk(bases, offs, k, m, 4, table, nb_passes=3, abundance_min=2)
    order = []
    for d in range(12):
        assert np.array_equal(np.fromfile(out + ".solid.%d" % d, dtype=np.uint8), ref.part_records(d)), d
        lo, hi, a = ref.part(d)
        order += [int(x) for x in lo]
    assert info["nb_passes"] == "3" and int(info["kmers_nb_solid"]) == ref.stats["kmers_nb_solid"] and int(info["kmers_nb_valid"]) == ref.stats["kmers_nb_valid"]
    assert np.array_equal(np.fromfile(out + ".mphf", dtype=np.uint8), gko.Mphf(order, k).save())
