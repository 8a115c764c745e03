"""The oracle against outputs of the reference ITSELF (tests/golden/reference_run/*.npz, written by tools/make_reference_run_vectors.py
from runs of the reference's own dbgh5): solid k-mer sets per dataset, histogram, cutoff, Bloom arrays of the three kinds, MPHF stream."""
import glob
import os

import numpy as np
import pytest

from oracle import gko

DIR = os.path.join(os.path.dirname(__file__), "golden", "reference_run")
FIX = sorted(glob.glob(os.path.join(DIR, "*.npz")))


def load(path):
    z = np.load(path)
    k = int(z["k"])
    vb = z["solid_value_bytes"]
    vals = [int.from_bytes(bytes(r), "little") for r in vb]
    sizes = z["solid_sizes"].tolist()
    parts, pos = [], 0
    for n in sizes:
        parts.append(list(zip(vals[pos:pos + n], z["solid_abundance"][pos:pos + n].tolist()))); pos += n
    rep = z["minimRepart"]
    nbpart = int(rep[:2].view("<u2")[0]); nmin = int(rep[2:10].view("<u8")[0])
    table = rep[12:12 + 2 * nmin].view("<u2").copy()
    m = int(round(np.log2(nmin) / 2))
    return z, k, m, nbpart, table, parts


def freq_order_of(z, m):
    """the reference's own /minimizers/minimFrequency (u32 freq_order[4^m] + magic) when the fixture carries it"""
    if "minimFrequency" not in z:
        return None
    raw = z["minimFrequency"]
    assert len(raw) == 4 * 4 ** m + 4 and int(raw[-4:].view("<u4")[0]) == 0x12345678
    return raw[:4 * 4 ** m].view("<u4").copy()


def oracle_run(z, k, m, nbpart, table, auto=False):
    bases, offs = gko.fastx_parse(bytes(z["fasta"]))
    # partition membership depends on the table and, in frequency mode, on the order (freq_order[c], c) of the reference's own minimFrequency;
    # the one-partition frequency fixture (k21_freq) has no stored order: every minimizer maps to partition 0 whatever the order
    amin = 2
    if auto:
        # -abundance-min auto (SortingCountAlgorithm.cpp:418-444, CountProcessorCutoff.hpp:88-99): a first count feeds the histogram, Histogram::compute_threshold with a
        # floor of 3 gives the cut-off, and that cut-off is the abundance-min of the count whose records are written
        h0 = gko.Dsk(bases, offs, k, m, nbpart, table, abundance_min=1, freq_order=freq_order_of(z, m)).histogram()
        amin = int(gko.histogram_cutoff(h0, 3)[0])
    return gko.Dsk(bases, offs, k, m, nbpart, table, abundance_min=amin, freq_order=freq_order_of(z, m))


@pytest.mark.parametrize("path", FIX, ids=[os.path.basename(p)[:-4] for p in FIX])
def test_oracle_equals_reference_run(path):
    z, k, m, nbpart, table, parts = load(path)
    if path.endswith("k21_freq.npz"):
        assert nbpart == 1
    if "4parts" in path:
        assert nbpart == 4 and freq_order_of(z, m) is not None and min(len(p) for p in parts) > 0     # frequency order really decides membership here
    d = oracle_run(z, k, m, nbpart, table, auto="_auto" in path)
    assert d.stats["kmers_nb_solid"] == int(z["nb_solid_kmers"]) == sum(len(p) for p in parts)
    for p in range(nbpart):
        lo, hi, ab = d.part(p)
        got = [(int(a) | (int(b) << 64), int(c)) for a, b, c in zip(lo, hi, ab)]
        assert got == parts[p], (p, len(got), len(parts[p]))                        # same k-mers, same counts, same (ascending) order
    h = d.histogram()
    assert z["histogram_index"].tolist() == list(range(1, len(z["histogram_index"]) + 1))
    assert np.array_equal(h[1:len(z["histogram_abundance"]) + 1], z["histogram_abundance"])
    cut, nbs, _ = gko.histogram_cutoff(h, 2)                                          # Histogram::compute_threshold, -abundance-min-threshold default 2
                                                                                      # (SortingCountAlgorithm.cpp:212)
    assert (cut, nbs) == (int(z["cutoff"]), int(z["nbsolidsforcutoff"]))
    order = [x for p in parts for x, _ in p]
    if "bloom" in z:
        kind = bytes(z["bloom_type"]).decode(); size = int(bytes(z["bloom_size"]).decode()); nh = int(bytes(z["bloom_nb_hash"]).decode())
        ob = gko.Bloom(kind, size, nh, k); ob.insert(order)
        assert np.array_equal(ob.array(), z["bloom"]), kind
    if "mphf" in z:
        assert np.array_equal(gko.Mphf(order, k).save(), z["mphf"])


def test_fixtures_are_what_the_reference_built_from_this_repository_writes():
    """tools/make_reference_run_vectors.py --check: the reference's own dbgh5 / gatb-h5dump (integration/_build/ref, built by integration/build_reference.sh from
    /root/reference with the reference's cmake) run again on the generated inputs must write exactly the committed fixtures — the pin is reproducible from the
    repository. Skipped where the reference tools are not built (the GPU box has them; /root/reference is not needed to RUN them)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "integration", "_build", "ref", "dbgh5")):
        pytest.skip("integration/_build/ref absent (integration/build_reference.sh)")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "make_reference_run_vectors.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, (r.stdout + r.stderr)[-2000:]
