"""(GPU box) two passes over 1e8 reads, serial and overlapped, several rounds; GKC_POOL_DEBUG=1 / GKC_VERBOSE=1 show the batch plan and the allocator per pass"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k, m, L, parts = 31, 10, 150, 4096
c = gkc.Counter(0)
c.configure(k, m, parts, bench.repart_for_bench(m, parts), nb_passes=2)
db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
def serial():
    for ps in range(2):
        t1 = time.perf_counter(); c.begin_pass(ps); t2 = time.perf_counter(); c.push_reads_device(db, do, n, n * L); t3 = time.perf_counter(); c.finish_pass()
        print("   pass %d: begin %.0f ms, push %.0f ms, finish %.0f ms" % (ps, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.perf_counter() - t3) * 1e3), flush=True)
def overlap():
    for ps in range(2):
        t1 = time.perf_counter(); c.begin_pass(ps); t2 = time.perf_counter(); c.push_reads_device(db, do, n, n * L); t3 = time.perf_counter(); c.finish_pass_async()
        print("   pass %d: begin %.0f ms, push %.0f ms, finish_async %.0f ms" % (ps, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.perf_counter() - t3) * 1e3), flush=True)
    t1 = time.perf_counter(); c.finish_pass_wait(); print("   wait %.0f ms" % ((time.perf_counter() - t1) * 1e3), flush=True)
for name, fn in (("serial", serial), ("serial", serial), ("serial", serial), ("overlapped", overlap), ("overlapped", overlap), ("overlapped", overlap)):
    t0 = time.perf_counter(); fn(); print("%s: %.0f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True)
