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
# MPHF of the same solid k-mers + abundance map (second build: the device pool is warm)
mp = gkc.Mphf(c); mp.close()
torch.cuda.synchronize(); t0 = time.perf_counter()
mp = gkc.Mphf(c)
torch.cuda.synchronize(); t1 = time.perf_counter()
amap, above = mp.abundance_map()
torch.cuda.synchronize(); t2 = time.perf_counter()
print("mphf keys", mp.size, "build ms", round((t1 - t0) * 1e3, 1), "G keys/s", round(mp.size / (t1 - t0) / 1e9, 2), "stream bytes", mp.L.gkc_mphf_save_size(mp.h),
      "bits/key", round(mp.L.gkc_mphf_save_size(mp.h) * 8 / mp.size, 2), "| abundance map ms (incl. D2H of %d MB)" % (mp.size >> 20), round((t2 - t1) * 1e3, 1), "above", above)
