"""Developer probe: one bench-like step with every library timer printed (not part of the product)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
profile = int(sys.argv[4]) if len(sys.argv) > 4 else 0           # 1 = GKC_SYNTH_SKEWED
c = gkc.Counter(0)
rep = bench.repart_for_bench(10, parts)
c.configure(k, 10, parts, rep)
db, do = c.synth_reads_device(2, n, 150, n * 5, 10000, profile=profile)
names = ["scan_count", "scan_emit", "scan_refine", "dedupe_bin", "dedupe_sort", "expand_count", "expand_scatter", "bucket_sort", "bucket_sort_big", "bucket_sort_wg", "bucket_sort_deep", "split_levels", "compact",
         "total_stage_a", "total_stage_b"]
for it in range(3):
    base = {x: c.timing(x) for x in names}
    t0 = time.perf_counter()
    c.begin_pass(0); t1 = time.perf_counter()
    c.push_reads_device(db, do, n, n * 150); t2 = time.perf_counter()
    c.finish_pass(); t3 = time.perf_counter()
    print("iter", it, "begin %.1f ms push %.1f ms finish %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3),
          {x: (round(c.timing(x)[0] - base[x][0], 2), c.timing(x)[1] - base[x][1]) for x in names})
print(c.stats())
