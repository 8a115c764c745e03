"""Golden vectors from the reference ITSELF: runs the reference's dbgh5 / gatb-h5dump binaries — integration/_build/ref/, built from /root/reference by
integration/build_reference.sh (the reference's own cmake; reproducible from this repository) — on small generated inputs and stores what they wrote as
fixtures under tests/golden/reference_run/. The tests only read the committed fixtures.

    python tools/make_reference_run_vectors.py [directory with dbgh5 and gatb-h5dump] [--out DIR]
    python tools/make_reference_run_vectors.py --check        regenerates into a scratch directory and compares with the committed fixtures

What is kept per run: the input FASTA, the (k-mer, abundance) records of every /dsk/solid/<p> dataset in dataset order, the histogram
datasets, and — where the run produced them — the byte arrays /bloom/bloom (with its size / nb_hash / type attributes) and /dsk/mphf."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import synth_reads  # noqa: E402

_args = [a for a in sys.argv[1:] if not a.startswith("--")]
_default_bin = os.path.join(ROOT, "integration", "_build", "ref")
BIN = _args[0] if _args else (_default_bin if os.path.exists(os.path.join(_default_bin, "dbgh5")) else "/tmp/gatb_build/bin/Release")
COMMITTED = os.path.join(ROOT, "tests", "golden", "reference_run")
CHECK = "--check" in sys.argv
OUT = tempfile.mkdtemp(prefix="gkc_refrun_") if CHECK else (sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else COMMITTED)


def h5dump(args):
    return subprocess.run([os.path.join(BIN, "gatb-h5dump")] + args, capture_output=True, text=True).stdout


def dataset_bytes(h5, path, mode):
    with tempfile.NamedTemporaryFile() as t:
        subprocess.run([os.path.join(BIN, "gatb-h5dump"), "-d", path, "-b", mode, "-o", t.name, h5], capture_output=True)
        return np.fromfile(t.name, dtype=np.uint8)


def attr(h5, path):
    m = re.search(r'\(0\): "(.*?)"\s*\}', h5dump(["-a", path, h5]), re.S)
    return m.group(1) if m else None


def run(tag, reads, k, extra, want_bloom=False, want_mphf=False, cores=1, max_memory=2000, want_freq=False, want_graph=False, amin="2"):
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "in.fa")
        text = "".join(">r%d\n%s\n" % (i, r.decode()) for i, r in enumerate(reads))
        open(fa, "w").write(text)
        out = os.path.join(td, "ref")
        cmd = [os.path.join(BIN, "dbgh5"), "-in", fa, "-kmer-size", str(k), "-abundance-min", amin, "-out", out, "-out-tmp", td, "-nb-cores", str(cores),
               "-max-memory", str(max_memory), "-verbose", "0"] + extra
        subprocess.run(cmd, check=True, capture_output=True)
        h5 = out + ".h5"
        nparts = int(attr(h5, "/dsk/solid/nb_partitions"))
        rec = 12 if k <= 31 else 20                                # packed file records: value (8 / 16 bytes) + u32 abundance
        vals, abs_, sizes = [], [], []
        for p in range(nparts):
            raw = dataset_bytes(h5, "/dsk/solid/%d" % p, "FILE")
            n = len(raw) // rec
            raw = raw[:n * rec].reshape(n, rec)
            vals.append(raw[:, :rec - 4].copy()); abs_.append(raw[:, rec - 4:].copy().view("<u4")[:, 0]); sizes.append(n)
        hist = dataset_bytes(h5, "/histogram/histogram", "FILE")
        hist = hist[:len(hist) // 12 * 12].reshape(-1, 12)
        fx = {"fasta": np.frombuffer(text.encode(), dtype=np.uint8), "k": np.int64(k),
              "solid_value_bytes": np.concatenate(vals) if vals else np.zeros((0, rec - 4), np.uint8), "solid_abundance": np.concatenate(abs_),
              "solid_sizes": np.array(sizes, dtype=np.int64),
              "histogram_index": hist[:, :4].copy().view("<u4")[:, 0], "histogram_abundance": hist[:, 4:].copy().view("<u8")[:, 0],
              "cutoff": dataset_bytes(h5, "/histogram/cutoff", "LE").view("<u8")[0], "nbsolidsforcutoff": dataset_bytes(h5, "/histogram/nbsolidsforcutoff", "LE").view("<u8")[0],
              "minimRepart": dataset_bytes(h5, "/minimizers/minimRepart", "LE"), "nb_solid_kmers": np.int64(int(attr(h5, "/nb_solid_kmers")))}
        if want_bloom:
            fx["bloom"] = dataset_bytes(h5, "/bloom/bloom", "LE")
            for a in ("size", "nb_hash", "type", "kmer_size"):
                v = attr(h5, "/bloom/bloom/" + a)
                fx["bloom_" + a] = np.frombuffer((v or "").encode(), dtype=np.uint8)
        if want_mphf:
            fx["mphf"] = dataset_bytes(h5, "/dsk/mphf", "LE")
        if want_graph:                                             # what the steps behind the Bloom filter leave (dbgh5's DEFAULT flags: debloom cascading, branching nodes stored):
            for name in ("debloom/bloom2", "debloom/bloom3", "debloom/bloom4", "debloom/cfp", "branching/nodes"):      # they depend on every query the debloom step made
                fx[name.replace("/", "_")] = dataset_bytes(h5, "/" + name, "FILE")
        if want_freq:                                              # u32 freq_order[4^m] + u32 magic (RepartitionAlgorithm.cpp:352-380, PartiInfo.cpp:271-295)
            fx["minimFrequency"] = dataset_bytes(h5, "/minimizers/minimFrequency", "LE")
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **fx)
        print(tag, "partitions", nparts, "solid", int(fx["nb_solid_kmers"]), {k_: (v.shape if hasattr(v, "shape") else v) for k_, v in fx.items() if k_ in ("bloom", "mphf") or k_.startswith(("debloom", "branching"))})


def canonical_unitigs(fa):
    """(count, total length, sha256) of the unitigs of a .unitigs.fa as a SET of canonical sequences"""
    import hashlib
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    seqs = []
    for line in open(fa, "rb"):
        if not line.startswith(b">"):
            s_ = line.strip(); seqs.append(min(s_, s_.translate(comp)[::-1]))
    seqs.sort()
    return [len(seqs), sum(len(x) for x in seqs), hashlib.sha256(b"\n".join(seqs)).hexdigest()]


def run_unitigs(tag, reads, k):
    """the reference's GraphUnitigs (integration/unitigs_check.cpp linked against the UNPATCHED library: integration/_build/ref/unitigs_check) on the .h5 the reference's
    dbgh5 counted in the mode GraphUnitigs forces (-minimizer-type 1 -repartition-type 1, GraphUnitigs.cpp:861-870), and straight from the FASTA: the unitig set"""
    import json
    exe = os.path.join(BIN, "unitigs_check")
    if not os.path.exists(exe):
        print(tag, "skipped: no", exe); return
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "in.fa")
        open(fa, "w").write("".join(">r%d\n%s\n" % (i, r.decode()) for i, r in enumerate(reads)))
        subprocess.run([os.path.join(BIN, "dbgh5"), "-in", fa, "-kmer-size", str(k), "-abundance-min", "2", "-out", os.path.join(td, "ref"), "-out-tmp", td, "-nb-cores", "1",
                        "-max-memory", "1", "-verbose", "0", "-bloom", "none", "-debloom", "none", "-branching-nodes", "none", "-no-mphf", "-minimizer-type", "1", "-repartition-type", "1"],
                       check=True, capture_output=True)
        subprocess.run([exe, os.path.join(td, "ref.h5"), os.path.join(td, "from_h5"), "1"], check=True, capture_output=True, cwd=td)
        subprocess.run([exe, fa, os.path.join(td, "from_fa"), "1", "-kmer-size", str(k), "-abundance-min", "2", "-out-tmp", td], check=True, capture_output=True, cwd=td)
        a, b = canonical_unitigs(os.path.join(td, "from_h5.unitigs.fa")), canonical_unitigs(os.path.join(td, "from_fa.unitigs.fa"))
        assert a == b, (a, b)
        json.dump({"k": k, "abundance_min": 2, "unitigs": a[0], "total_length": a[1], "sha256_sorted_canonical": a[2]}, open(os.path.join(OUT, tag + ".json"), "w"))
        print(tag, a)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    reads = synth_reads(600, 3000, 150, seed=41, n_rate=0.002)
    count_only = ["-bloom", "none", "-debloom", "none", "-branching-nodes", "none", "-no-mphf"]
    run("k31_neighbor_mphf", reads, 31, ["-bloom", "neighbor", "-debloom", "none", "-branching-nodes", "none"], want_bloom=True, want_mphf=True)
    run("k31_basic", reads, 31, ["-bloom", "basic", "-debloom", "none", "-branching-nodes", "none", "-no-mphf"], want_bloom=True)
    run("k31_cache", reads, 31, ["-bloom", "cache", "-debloom", "none", "-branching-nodes", "none", "-no-mphf"], want_bloom=True)
    run("k21_freq", reads, 21, count_only + ["-minimizer-type", "1", "-repartition-type", "1"])
    # frequency-order minimizers with SEVERAL partitions (what GraphUnitigs forces: -minimizer-type 1 -repartition-type 1): -max-memory 1 makes the
    # reference cut this input into 4 partitions; minimFrequency and minimRepart are the reference's own tables
    run("k21_freq_4parts", synth_reads(6000, 30000, 150, seed=43, n_rate=0.002), 21, count_only + ["-minimizer-type", "1", "-repartition-type", "1"], max_memory=1, want_freq=True)
    # the same input with the default (lexicographic) minimizers and the bcalm-friendly lexicographic grouping (-repartition-type 1)
    run("k21_lexi_grouped_parts", synth_reads(6000, 30000, 150, seed=43, n_rate=0.002), 21, count_only + ["-repartition-type", "1"], max_memory=1)
    # ... and with everything default (lexicographic minimizers, computeDistrib balancing on kx-mers): pins the default Repartitor table
    run("k21_default_parts", synth_reads(6000, 30000, 150, seed=43, n_rate=0.002), 21, count_only, max_memory=1)
    run("k31_2parts_mphf", synth_reads(2000, 10000, 150, seed=3), 31, ["-bloom", "none", "-debloom", "none", "-branching-nodes", "none"], want_mphf=True, cores=2)
    run("k63_neighbor_mphf", reads[:300], 63, ["-bloom", "neighbor", "-debloom", "none", "-branching-nodes", "none"], want_bloom=True, want_mphf=True)
    # dbgh5 with its DEFAULT flags (MPHF, neighbor Bloom, cascading debloom, branching nodes): BASELINE configs[4]'s pipeline behind the counting step
    run("k31_defaults", reads, 31, [], want_bloom=True, want_mphf=True, want_graph=True)
    run("k63_defaults", reads[:300], 63, [], want_bloom=True, want_mphf=True, want_graph=True)
    run("k21_defaults_parts", synth_reads(6000, 30000, 150, seed=43, n_rate=0.002), 21, [], want_bloom=True, want_mphf=True, want_graph=True, max_memory=1)
    # -abundance-min auto (SortingCountAlgorithm.cpp:418-444: the cut-off processor + a proxy, TWO processors; the threshold is Histogram::compute_threshold's,
    # Histogram.cpp:61-190): 30x reads, several partitions
    run("k31_auto_parts", synth_reads(6000, 30000, 150, seed=45), 31, count_only, max_memory=1, amin="auto")
    run_unitigs("k21_freq_4parts_unitigs", synth_reads(6000, 30000, 150, seed=43, n_rate=0.002), 21)
    if CHECK:
        for f in sorted(os.listdir(OUT)):
            if f.endswith(".json") and open(os.path.join(OUT, f)).read() != (open(os.path.join(COMMITTED, f)).read() if os.path.exists(os.path.join(COMMITTED, f)) else None):
                print("%s differs from the committed one" % f); sys.exit(1)
        # every array of every regenerated fixture against the committed one
        bad = []
        for f in sorted(os.listdir(OUT)):
            if not f.endswith(".npz"):
                continue
            new = np.load(os.path.join(OUT, f), allow_pickle=True)
            if not os.path.exists(os.path.join(COMMITTED, f)):
                bad.append(f + ": not committed"); continue
            old = np.load(os.path.join(COMMITTED, f), allow_pickle=True)
            if sorted(new.files) != sorted(old.files):
                bad.append("%s: arrays %s != %s" % (f, sorted(new.files), sorted(old.files))); continue
            for name in new.files:
                a, b = new[name], old[name]
                if a.shape != b.shape or not np.array_equal(a, b):
                    bad.append("%s: %s differs" % (f, name))
        print("reference binaries: %s" % BIN)
        print("check of tests/golden/reference_run against a fresh run of the reference: %s" % ("IDENTICAL (%d fixtures)" % len([f for f in os.listdir(OUT) if f.endswith(".npz")]) if not bad else "\n".join(bad)))
        import shutil
        shutil.rmtree(OUT, ignore_errors=True)
        sys.exit(1 if bad else 0)
