"""Developer measurement (GPU box): the same count with the boundary handing over HOST buffers — reads pushed from host memory (H2D inside
gkc_push_reads) and every partition's Count[] fetched back to host memory (gkc_partition_counts) — next to the resident-in-HBM figure that
bench.py reports as `value`. usage: python tools/pcie_inclusive.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
k, m, L = 31, 10, 150
parts = 1024
c = gkc.Counter(0)
c.configure(k, m, parts, bench.repart_for_bench(m, parts))
db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
hb = c.device_to_host(db, n * L)
ho = np.arange(n + 1, dtype=np.uint64) * L
pin_in = gkc.HostBuffer(n * L); pin_in.a[:] = hb                   # page-locked copies (gkc_host_alloc)
pin_out = gkc.HostBuffer(1 << 30)
for it in range(3):
    t0 = time.perf_counter()
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    t1 = time.perf_counter()
    c.begin_pass(0); c.push_reads(hb, ho); c.finish_pass()
    t2 = time.perf_counter()
    tot = 0
    for p in range(parts):
        tot += len(c.partition_records(0, p))
    t3 = time.perf_counter()
    c.begin_pass(0); c.push_reads(pin_in.a, ho); c.finish_pass()
    t4 = time.perf_counter()
    tot2 = 0
    for p in range(parts):
        tot2 += len(c.partition_records(0, p, out=pin_out.a))
    t5 = time.perf_counter()
    d = c.stats()["kmers_nb_distinct"]
    print("page-locked buffers: reads in %.1f ms | records out %.1f ms (%.1f GB/s) | host-to-host %.2e distinct k-mers/s" % ((t4 - t3) * 1e3, (t5 - t4) * 1e3, tot2 / (t5 - t4) / 1e9, d / (t5 - t3)))
    print("resident %.1f ms (%.2e distinct/s) | host reads in %.1f ms | + records out (%.2f GB) %.1f ms | host-to-host %.2e distinct k-mers/s"
          % ((t1 - t0) * 1e3, d / (t1 - t0), (t2 - t1) * 1e3, tot / 1e9, (t3 - t2) * 1e3, d / (t3 - t1)), flush=True)
