"""(run on the GPU box, then `--compare` here)  The reference's OWN dbgh5, patched with integration/SortingCountAlgorithm.device.patch and linked against libgkc_hip.so
(integration/check_integration.sh --link -> integration/_build/dbgh5_device, a build-container artefact that travels with gpurun), run on the
GPU box on the inputs of the reference-run fixtures: the .h5 it writes must hold exactly the datasets the unpatched reference wrote
(tests/golden/reference_run/*.npz). This is the drop-in claim end to end: reference main(), reference Configuration / Repartitor / processors /
HDF5 storage, counting on the MI355X through PartitionsByDeviceCommand. Optional evidence (the artefact only exists where the reference was
built); log kept under profiles/.   python tools/run_patched_dbgh5.py [out dir]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.h5mini import H5Mini                       # noqa: E402
from tests.test_reference_run import load             # noqa: E402

EXE = os.path.join(ROOT, "integration", "_build", "dbgh5_device")
CASES = [("k21_freq_4parts", ["-minimizer-type", "1", "-repartition-type", "1"], "1"),
         ("k21_lexi_grouped_parts", ["-repartition-type", "1"], "1"),
         ("k21_default_parts", [], "1"),
         ("k31_2parts_mphf", [], "2000")]


def dump_dataset(bin_dir, h5, path, mode):
    with tempfile.NamedTemporaryFile() as t:
        subprocess.run([os.path.join(bin_dir, "gatb-h5dump"), "-d", path, "-b", mode, "-o", t.name, h5], capture_output=True)
        return np.fromfile(t.name, dtype=np.uint8)


def compare(outdir, bin_dir):
    """build container: the .h5 files the patched dbgh5 wrote on the GPU box, read with the reference's own gatb-h5dump, against the fixtures"""
    bad = 0
    for tag, extra, mem in CASES:
        z, k, m, nbpart, table, parts = load(os.path.join(ROOT, "tests", "golden", "reference_run", tag + ".npz"))
        h5 = os.path.join(outdir, tag + "_dev.h5")
        if not os.path.exists(h5):
            print(tag, "no .h5 from the GPU run"); bad += 1; continue
        rec = 12 if k <= 31 else 20
        ok = np.array_equal(dump_dataset(bin_dir, h5, "/minimizers/minimRepart", "LE"), z["minimRepart"])
        n_tot = 0
        for p in range(nbpart):
            raw = dump_dataset(bin_dir, h5, "/dsk/solid/%d" % p, "FILE")
            n = len(raw) // rec; raw = raw[:n * rec].reshape(n, rec)
            vals = [int.from_bytes(bytes(r), "little") for r in raw[:, :rec - 4]]
            ab = raw[:, rec - 4:].copy().view("<u4")[:, 0].tolist()
            ok &= list(zip(vals, ab)) == parts[p]; n_tot += n
        hist = dump_dataset(bin_dir, h5, "/histogram/histogram", "FILE"); hist = hist[:len(hist) // 12 * 12].reshape(-1, 12)
        ok &= np.array_equal(hist[:, 4:].copy().view("<u8")[:, 0], z["histogram_abundance"])
        print("%-24s %d partitions, %d solid k-mers : %s" % (tag, nbpart, n_tot, "IDENTICAL to the unpatched reference's .h5 (every /dsk/solid/<p> in order, histogram, minimRepart)" if ok else "DIFFERS"))
        bad += not ok
    return 1 if bad else 0


def main():
    if not os.path.exists(EXE):
        print("no integration/_build/dbgh5_device (run integration/check_integration.sh --link in the build container)"); return 2
    outdir = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp()
    if len(sys.argv) > 2 and sys.argv[2] == "--compare":
        return compare(outdir, sys.argv[3] if len(sys.argv) > 3 else "/tmp/gatb_build/bin/Release")
    os.makedirs(outdir, exist_ok=True)
    bad = 0
    for tag, extra, mem in CASES:
        z, k, m, nbpart, table, parts = load(os.path.join(ROOT, "tests", "golden", "reference_run", tag + ".npz"))
        fa = os.path.join(outdir, tag + ".fa"); open(fa, "wb").write(bytes(z["fasta"]))
        out = os.path.join(outdir, tag + "_dev")
        cmd = [EXE, "-in", fa, "-kmer-size", str(k), "-abundance-min", "2", "-out", out, "-out-tmp", outdir, "-nb-cores", "2" if tag.startswith("k31_2parts") else "1",
               "-max-memory", mem, "-verbose", "0", "-bloom", "none", "-debloom", "none", "-branching-nodes", "none", "-no-mphf"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print(tag, "FAILED rc", r.returncode, (r.stdout + r.stderr)[-1500:]); bad += 1; continue
        print("%-24s ran: rc 0, %d bytes of .h5 (compare in the build container: tools/run_patched_dbgh5.py <dir> --compare)" % (tag, os.path.getsize(out + ".h5")))
        for f in os.listdir(outdir):                               # keep only the .h5 files (gpurun_out is size-limited)
            if not f.endswith("_dev.h5"):
                try:
                    os.remove(os.path.join(outdir, f))
                except OSError:
                    pass
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
