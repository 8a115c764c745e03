"""Developer run (GPU box): an input beyond the single-pass capacity of one MI355X — 4e8 reads of 150 bp (4.8e10 k-mers at k=31), three passes (reads pushed in four batches per pass), every
pass released after it was drained (here: its partition statistics read). Checks: histogram sums == valid k-mers, every pass holds a third of the minimizers.
usage: python tools/multipass_demo.py [n_reads] [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k, m, L, parts = 31, 10, 150, 8192
c = gkc.Counter(0)
c.configure(k, m, parts, bench.repart_for_bench(m, parts), nb_passes=passes)
chunks = 4                                         # pushed in four batches: Stage A's scratch (descriptors, read-start mask) scales with the batch
nc = n // chunks
bufs = [c.synth_reads_device(2, nc, L, n * 5, 10000, first_read=i * nc) for i in range(chunks)]
print("reads resident: %.1f GB; usable HBM %.1f GB" % (n * (L + 8) / 1e9, c.device_memory()[0] / 1e9), flush=True)
t0 = time.perf_counter(); distinct = 0
for ps in range(passes):
    t1 = time.perf_counter()
    c.begin_pass(ps)
    for db, do in bufs:
        c.push_reads_device(db, do, nc, nc * L)
    c.finish_pass()
    d = sum(c.partition_info(ps, p)[1] for p in range(parts))
    distinct += d
    c.release_pass(ps)
    print("pass %d: %.2e distinct k-mers in %.0f ms" % (ps, d, (time.perf_counter() - t1) * 1e3), flush=True)
dt = time.perf_counter() - t0
st = c.stats(); h = c.histogram()
assert st["kmers_nb_valid"] == n * (L - k + 1) and st["kmers_nb_distinct"] == distinct == int(h.sum())
assert int((h * np.arange(len(h), dtype=np.uint64)).sum()) == st["kmers_nb_valid"]
print("total: %.2e valid, %.2e distinct k-mers in %.2f s = %.2e distinct k-mers/s (%d passes)" % (st["kmers_nb_valid"], distinct, dt, distinct / dt, passes))
