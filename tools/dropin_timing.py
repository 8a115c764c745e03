"""(GPU box) Times inside the reference's own dbgh5 — unpatched, and patched with integration/gatb-core.device.patch — on the same FASTA of synthetic 150 bp reads in
/dev/shm, as the reference's own dbginfo prints them from the .h5 (getInfo() of every algorithm: SortingCountAlgorithm.cpp:770-781, BloomAlgorithm.cpp:188-193,
MPHFAlgorithm.cpp:268-275, DebloomAlgorithm), plus the process wall.

    python tools/dropin_timing.py [n_reads=10000000] [abundance_min=2] [count|pipeline|repart|all]        -> stdout (kept under profiles/)

  count      the DSK step alone (-bloom none -debloom none -branching-nodes none -no-mphf): the reference; the patched binary with the device-sized Configuration
             (default) at -max-memory 5000 and 200000; the patched binary with the REFERENCE's Configuration (GATB_DEVICE_REFERENCE_CONFIG=1: 2816 partitions at
             -max-memory 5000, 256 huge ones at 200000 — the partition-count cliff of round 3); the bank iterated instead of parsed on the device
  pipeline   dbgh5 with its default flags (MPHF, neighbor Bloom, cascading debloom, branching nodes: BASELINE configs[4]'s pipeline): the reference; patched with
             Bloom / MPHF / debloom queries on the device (default); patched with only the counting step on the device (GATB_DEVICE_NO_BLOOM=1 GATB_DEVICE_NO_MPHF=1)
DROPIN_SKIP_REF=1 leaves the unpatched reference out (80 s of DSK at 10^8 reads, minutes of debloom)."""
import os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "integration", "_build", "ref")
DEV = os.path.join(ROOT, "integration", "_build", "dbgh5_device")


def info(h5):
    """dbginfo's tree as {'section/key': value} (sections by indentation), first occurrence wins"""
    out = subprocess.run([os.path.join(REF, "dbginfo"), "-in", h5], capture_output=True, text=True).stdout
    vals, stack = {}, []
    for line in out.splitlines():
        if not line.strip():
            continue
        depth = (len(line) - len(line.lstrip(" "))) // 4
        k, sep, v = line.strip().partition(":")
        k = k.strip(); v = v.strip()
        stack = stack[:depth] + [k]
        path = "/".join(stack[1:])                       # without the root ("graph")
        if v:
            vals.setdefault(path, v)
    return vals


def g(v, path, default="?"):
    return v.get(path, default)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    amin = sys.argv[2] if len(sys.argv) > 2 else "2"
    mode = sys.argv[3] if len(sys.argv) > 3 else "all"
    gkc = ge.load().gkc
    c = gkc.Counter(0)
    L = 150
    d_b, d_o = c.synth_reads_device(1, n, L, n * 5, 10000)
    bases = c.device_to_host(d_b, n * L); c.device_free(d_b); c.device_free(d_o); c.close()
    work = tempfile.mkdtemp(prefix="gkc_dropin_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    skip_ref = os.environ.get("DROPIN_SKIP_REF") is not None
    try:
        rec = np.empty((n, L + 4), dtype=np.uint8)
        rec[:, 0] = ord(">"); rec[:, 1] = ord("r"); rec[:, 2] = 10; rec[:, 3:3 + L] = bases.reshape(n, L); rec[:, 3 + L] = 10
        fa = os.path.join(work, "reads.fa"); rec.tofile(fa); del rec, bases
        cores = os.cpu_count() or 1
        count_only = ["-bloom", "none", "-debloom", "none", "-branching-nodes", "none", "-no-mphf"]

        def run(name, exe, env, flags, maxmem):
            if not os.path.exists(exe):
                print("# %s: %s absent" % (name, exe)); return None
            out = os.path.join(work, "out_%d" % abs(hash(name)))
            e = dict(os.environ); e.update(env)
            cmd = [exe, "-in", fa, "-kmer-size", "31", "-abundance-min", amin, "-nb-cores", str(cores), "-max-memory", maxmem, "-out", out, "-verbose", "0"] + flags
            t0 = time.time(); r = subprocess.run(cmd, cwd=work, env=e, capture_output=True, text=True); wall = time.time() - t0
            if os.environ.get("GATB_DEVICE_VERBOSE"):
                print("#   " + "\n#   ".join(l for l in (r.stdout + r.stderr).splitlines() if l.startswith("[device") or "[gkc" in l)[:int(os.environ.get("DROPIN_LOG_CHARS", "4000"))])
            if r.returncode != 0:
                print("# %s FAILED rc %d: %s" % (name, r.returncode, (r.stdout + r.stderr)[-400:])); return None
            v = info(out + ".h5"); v["_wall"] = "%.2f" % wall
            os.remove(out + ".h5")
            return v

        # a freshly leased box pays ~26 ms per GB the FIRST time a region of its VRAM is allocated after boot (tools/alloc_probe: 128 GB of hipMalloc 3.4-3.9 s in the first
        # process, 2 ms in the next): every dbgh5 run below would carry 2-4 s of that inside its device_stage_b. One process touches the VRAM first, so the runs
        # measure the software (a machine that has been up for a while behaves like this all the time).
        probe = os.path.join(ROOT, "tools", "alloc_probe", "alloc_probe")
        if os.path.exists(probe) and not os.environ.get("DROPIN_NO_VRAM_TOUCH"):
            r = subprocess.run([probe, "malloc", "16", "16"], capture_output=True, text=True)
            print("# VRAM touched once before the runs (tools/alloc_probe malloc 16 16): " + " | ".join(r.stdout.strip().splitlines()[:2]))
        if mode in ("count", "all"):
            print("# DSK step alone: %d synthetic 150 bp reads (30x, 1%% substitutions) as FASTA in %s, k=31, abundance-min %s, -nb-cores %d, %s" % (n, work, amin, cores, " ".join(count_only)))
            print("# %-86s %8s %8s %9s %10s %6s %6s %9s %9s %9s %9s %12s %12s" % ("run", "wall s", "dsk s", "fill_part", "fill_solid", "parts", "passes", "dev A s", "dev B s", "dev wait", "hand-over", "distinct", "solid"))
            ref_counts = None
            runs = [] if skip_ref else [("reference (unpatched dbgh5), -max-memory 5000 (its default)", os.path.join(REF, "dbgh5"), {}, "5000")]
            runs += [("patched (default: Configuration from the HBM, text parsed on the device, bulk), -max-memory 5000", DEV, {}, "5000"),
                     ("patched, round 4's sink: BagHDF5Patch::insert under the storage lock (GATB_DEVICE_NO_DIRECT_SINK=1), -max-memory 5000", DEV, {"GATB_DEVICE_NO_DIRECT_SINK": "1"}, "5000"),
                     ("patched, direct sink without the page-locked ring: pageable fetch (GATB_DEVICE_NO_RING=1), -max-memory 5000", DEV, {"GATB_DEVICE_NO_RING": "1"}, "5000"),
                     ("patched (default), -max-memory 200000", DEV, {}, "200000"),
                     ("patched, the REFERENCE's Configuration, -max-memory 5000", DEV, {"GATB_DEVICE_REFERENCE_CONFIG": "1"}, "5000"),
                     ("patched, the REFERENCE's Configuration, -max-memory 200000 (few huge partitions)", DEV, {"GATB_DEVICE_REFERENCE_CONFIG": "1"}, "200000"),
                     ("patched (default), bank iterated by the reference's reader (GATB_DEVICE_NO_TEXT=1)", DEV, {"GATB_DEVICE_NO_TEXT": "1"}, "5000")]
            for name, exe, env, maxmem in runs:
                if os.environ.get("DROPIN_ONLY") and os.environ["DROPIN_ONLY"] not in name:
                    continue
                v = run(name, exe, env, count_only, maxmem)
                if v is None:
                    continue
                print("  %-86s %8s %8s %9s %10s %6s %6s %9s %9s %9s %9s %12s %12s" % (
                    name, v["_wall"], g(v, "dsk/time"), g(v, "dsk/time/fill_partitions"), g(v, "dsk/time/fill_solid_kmers"), g(v, "configuration/config/nb_partitions"),
                    g(v, "configuration/config/nb_passes"), g(v, "dsk/stats/fillsolid_time/device_stage_a", "-"), g(v, "dsk/stats/fillsolid_time/device_stage_b", "-"),
                    g(v, "dsk/stats/fillsolid_time/device_wait", "-"), g(v, "dsk/stats/fillsolid_time/device_hand_over", "-"),
                    g(v, "dsk/stats/kmers/kmers_nb_distinct"), g(v, "dsk/stats/kmers/kmers_nb_solid")), flush=True)
                counts = (g(v, "dsk/stats/kmers/kmers_nb_distinct"), g(v, "dsk/stats/kmers/kmers_nb_solid"))
                if ref_counts is None:
                    ref_counts = counts
                else:
                    print("#   same distinct / solid counts as the first run: %s" % (counts == ref_counts))
        if mode in ("repart", "all"):
            # RepartitorAlgorithm's serial sampling (RepartitionAlgorithm.cpp:348, :464) inside the patched dbgh5: functors of the reference vs the device (RepartitorDevice.hpp);
            # -minimizer-type 1 -repartition-type 1 is what GraphUnitigs forces (GraphUnitigs.cpp:861-870): there computeFrequencies walks 5 % of the bank on one core
            print("# the Repartitor's sampling inside the patched dbgh5 (count-only flags, -max-memory 5000): the reference's serial functors (GATB_DEVICE_NO_REPARTITOR=1) vs the device")
            print("# %-86s %8s %8s %9s %10s %12s" % ("run", "wall s", "dsk s", "fill_part", "fill_solid", "solid"))
            freq = ["-minimizer-type", "1", "-repartition-type", "1"]
            for name, env, flags in (("lexicographic minimizers (default), sampling on the device", {}, []),
                                     ("lexicographic minimizers (default), the reference's functors", {"GATB_DEVICE_NO_REPARTITOR": "1"}, []),
                                     ("frequency minimizers (GraphUnitigs' mode), sampling on the device", {}, freq),
                                     ("frequency minimizers (GraphUnitigs' mode), the reference's functors", {"GATB_DEVICE_NO_REPARTITOR": "1"}, freq)):
                v = run(name, DEV, env, count_only + flags, "5000")
                if v is not None:
                    print("  %-86s %8s %8s %9s %10s %12s" % (name, v["_wall"], g(v, "dsk/time"), g(v, "dsk/time/fill_partitions"), g(v, "dsk/time/fill_solid_kmers"), g(v, "dsk/stats/kmers/kmers_nb_solid")), flush=True)
        if mode in ("pipeline", "all"):
            print("# dbgh5 with its DEFAULT flags (MPHF, neighbor Bloom, cascading debloom, branching nodes), same input, -max-memory 5000")
            print("# (debloom = fill_debloom_file + finalize_debloom_file + cascading, the reference's own TimeInfo keys)")
            print("# %-86s %8s %8s %8s %8s %8s %8s %8s %8s %10s %12s %14s %10s" % ("run", "wall s", "dsk s", "mphf s", "bloom s", "debloom", "fill", "finalize", "cascad.", "branching", "solid", "bloom bits", "cfp nb"))
            runs = [] if skip_ref else [("reference (unpatched dbgh5)", os.path.join(REF, "dbgh5"), {})]
            runs += [("patched: counting, MPHF, Bloom, debloom queries on the device (default)", DEV, {}),
                     ("patched: only the counting step on the device (GATB_DEVICE_NO_BLOOM=1 GATB_DEVICE_NO_MPHF=1)", DEV, {"GATB_DEVICE_NO_BLOOM": "1", "GATB_DEVICE_NO_MPHF": "1"})]
            first = None
            for name, exe, env in runs:
                v = run(name, exe, env, [], "5000")
                if v is None:
                    continue
                print("  %-86s %8s %8s %8s %8s %8s %8s %8s %8s %10s %12s %14s %10s" % (name, v["_wall"], g(v, "dsk/time"), g(v, "mphf/time"), g(v, "bloom/time"), g(v, "debloom/time"),
                                                                     g(v, "debloom/time/fill_debloom_file"), g(v, "debloom/time/finalize_debloom_file"), g(v, "debloom/time/cascading"),
                                                                     g(v, "branching/time/build"), g(v, "dsk/stats/kmers/kmers_nb_solid"), g(v, "bloom/stats/bitsize"), g(v, "debloom/stats/cfp/nb")), flush=True)
                # (the number of critical false positives is left out of the comparison: in the UNPATCHED reference it moves by one with the partition layout —
                #  150000 reads, k=31: 291572 with 4 partitions, 291573 with 108 or 324 — and the device-sized Configuration is another layout)
                key = (g(v, "dsk/stats/kmers/kmers_nb_solid"), g(v, "bloom/stats/bitsize"), g(v, "branching/stats/nb_branching"), g(v, "branching/stats/checksum_branching"))
                if first is None:
                    first = key
                else:
                    print("#   same solid k-mers, Bloom size, number of branching nodes and branching-node checksum as the first run: %s  %s" % (key == first, key))
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
