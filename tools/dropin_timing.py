"""(GPU box) Wall of the DSK step inside the reference's own dbgh5, three ways, on the same FASTA of synthetic 150 bp reads in /dev/shm:
   unpatched reference (integration/_build/ref/dbgh5)  |  patched: the bank iterated, per-record hand-over (GATB_DEVICE_NO_TEXT=1 GATB_DEVICE_NO_BULK=1)  |
   patched: the bank iterated, bulk hand-over  |  patched: the FASTA text parsed on the device, bulk hand-over (default)
`dsk.time`, fill_partitions, fill_solid_kmers as the reference's own dbginfo prints them (SortingCountAlgorithm.cpp:770-781), plus the process wall.
    python tools/dropin_timing.py [n_reads=10000000] [abundance_min=2]        -> stdout (kept under profiles/)"""
import os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "integration", "_build", "ref")
DEV = os.path.join(ROOT, "integration", "_build", "dbgh5_device")


def info(h5):
    out = subprocess.run([os.path.join(REF, "dbginfo"), "-in", h5], capture_output=True, text=True).stdout
    vals = {}
    for line in out.splitlines():
        k, sep, v = line.partition(":")
        if sep and v.strip():
            vals.setdefault(k.strip(), v.strip())
    return vals


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    amin = sys.argv[2] if len(sys.argv) > 2 else "2"
    gkc = ge.load().gkc
    c = gkc.Counter(0)
    L = 150
    d_b, d_o = c.synth_reads_device(1, n, L, n * 5, 10000)
    bases = c.device_to_host(d_b, n * L); c.device_free(d_b); c.device_free(d_o); c.close()
    work = tempfile.mkdtemp(prefix="gkc_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        rec = np.empty((n, L + 4), dtype=np.uint8)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = 10; rec[:, 3:3 + L] = bases.reshape(n, L); rec[:, 3 + L] = 10
        fa = os.path.join(work, "reads.fa"); rec.tofile(fa); del rec, bases
        cores = os.cpu_count() or 1
        print("# %d synthetic 150 bp reads (30x, 1%% substitutions) as FASTA in %s, k=31, abundance-min %s, -nb-cores %d, -max-memory 200000 unless stated, -bloom none -debloom none -branching-nodes none -no-mphf" % (n, work, amin, cores))
        print("# %-78s %9s %9s %9s %9s %12s %12s   %s" % ("run", "wall s", "dsk s", "fill_part", "fill_solid", "distinct", "solid", "kind of partition commands"))
        ref_solid = None
        for name, exe, env in (("reference (unpatched dbgh5)", os.path.join(REF, "dbgh5"), {}),
                               ("patched, iterated bank, per-record hand-over", DEV, {"GATB_DEVICE_NO_BULK": "1", "GATB_DEVICE_NO_TEXT": "1"}),
                               ("patched, iterated bank, bulk hand-over", DEV, {"GATB_DEVICE_NO_TEXT": "1"}),
                               ("patched, text parsed on the device, bulk hand-over (default)", DEV, {}),
                               ("patched (default), -max-memory 5000 = dbgh5's own default: 2816 partitions", DEV, {"_maxmem": "5000"})):
            if os.environ.get("DROPIN_ONLY") and os.environ["DROPIN_ONLY"] not in name:
                continue
            if not os.path.exists(exe):
                print("# %s: %s absent" % (name, exe)); continue
            out = os.path.join(work, "out_%d" % abs(hash(name)))
            e = dict(os.environ); e.update({k_: v_ for k_, v_ in env.items() if not k_.startswith("_")})
            cmd = [exe, "-in", fa, "-kmer-size", "31", "-abundance-min", amin, "-nb-cores", str(cores), "-max-memory", env.get("_maxmem", "200000"), "-bloom", "none", "-debloom", "none",
                   "-branching-nodes", "none", "-no-mphf", "-out", out, "-verbose", "0"]
            t0 = time.time(); r = subprocess.run(cmd, cwd=work, env=e, capture_output=True, text=True); wall = time.time() - t0
            if os.environ.get("GATB_DEVICE_VERBOSE"):
                print("#   " + "\n#   ".join(l for l in (r.stdout + r.stderr).splitlines() if "device counting" in l or "[gkc]" in l)[:3000])
            if r.returncode != 0:
                print("# %s FAILED rc %d: %s" % (name, r.returncode, (r.stdout + r.stderr)[-400:])); continue
            v = info(out + ".h5")
            kinds = ", ".join("%s %s" % (k_, v[k_]) for k_ in ("vector", "hash", "device") if k_ in v)
            print("  %-78s %9.2f %9s %9s %9s %12s %12s   %s" % (name, wall, v.get("time", "?"), v.get("fill_partitions", "?"), v.get("fill_solid_kmers", "?"),
                                                             v.get("kmers_nb_distinct", "?"), v.get("kmers_nb_solid", "?"), kinds), flush=True)
            if ref_solid is None:
                ref_solid = (v.get("kmers_nb_distinct"), v.get("kmers_nb_solid"))
            else:
                print("#   same distinct / solid counts as the reference: %s" % ((v.get("kmers_nb_distinct"), v.get("kmers_nb_solid")) == ref_solid))
            os.remove(out + ".h5")
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
