import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
k, m, parts = 31, 10, 4096
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rep = bench.repart_for_bench(m, parts)
c = gkc.Counter(0); c.set_solidity(2, 2147483647, 10000); c.configure(k, m, parts, rep)
d_b, d_o = c.synth_reads_device(2, n_reads, 150, n_reads * 5, 10000)
c.begin_pass(0); c.push_reads_device(d_b, d_o, n_reads, n_reads * 150); c.finish_pass()
ns = c.stats()["kmers_nb_solid"]
for kind in ("basic", "cache", "neighbor"):
    bits = int(ns * 11.0)                   # ~ NBITS_PER_KMER of the reference default
    b = gkc.Bloom(c, kind, bits, 7, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.insert_solid()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kind, "solid", ns, "bits", bits, "insert ms", round(dt * 1e3, 1), "G kmers/s", round(ns / dt / 1e9, 2))
    del b
