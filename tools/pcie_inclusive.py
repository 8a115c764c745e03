"""Developer measurement (GPU box): the same count with the boundary handing over HOST buffers at both ends — reads pushed from page-locked host memory
(gkc_push_reads: H2D of chunk j+1 under the scan of chunk j) and every partition's Count[] streamed into a page-locked sink while Stage B runs
(gkc_set_host_sink) — next to the resident-in-HBM figure bench.py reports as `value`. usage: python tools/pcie_inclusive.py [n_reads] [abundance_min]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
amin = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k, m, L = 31, 10, 150
parts = 2048
c = gkc.Counter(0)
c.configure(k, m, parts, bench.repart_for_bench(m, parts)); c.set_solidity(amin, 2147483647, 10000)
db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
pin_in = gkc.HostBuffer(n * L); pin_in.a[:] = c.device_to_host(db, n * L)
ho = np.arange(n + 1, dtype=np.uint64) * L
for it in range(3):
    c.set_host_sink(None)
    t0 = time.perf_counter()
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
    t1 = time.perf_counter()
    st = c.stats()
    if it == 0:
        sink = gkc.HostBuffer(int(st["kmers_nb_solid"] * 16 * 1.02) + (64 << 20))
    c.begin_pass(0); c.push_reads(pin_in.a, ho); c.finish_pass()
    t2 = time.perf_counter()
    c.set_host_sink(sink)
    c.begin_pass(0); c.push_reads(pin_in.a, ho); c.finish_pass()
    t3 = time.perf_counter()
    d = st["kmers_nb_distinct"]
    print("abundance-min %d, %d reads: resident %.1f ms (%.2e distinct/s) | reads from host memory %.1f ms (%.1f GB in) | host to host (reads in, %.2f GB of Count[] out) %.1f ms = %.2e distinct k-mers/s"
          % (amin, n, (t1 - t0) * 1e3, d / (t1 - t0), (t2 - t1) * 1e3, n * L / 1e9, st["kmers_nb_solid"] * 16 / 1e9, (t3 - t2) * 1e3, d / (t3 - t2)), flush=True)
