"""(GPU box) ONE gkc_push_reads_device of 2e8 reads (3e10 bases, 8192 partitions): ms per step. Round 3: 2.6 s per step (giant per-push buffers); the push is now
scanned in slices of <= 1.6e10 bases (GKC_PUSH_SPLIT).   python tools/bigpush_probe.py [n_reads=200000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
k, m, L, parts = 31, 10, 150, 8192
c = gkc.Counter(0); c.configure(k, m, parts, bench.repart_for_bench(m, parts))
db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
cs = c.kmer_checksum_device(db, do, n, n * L)
for i in range(4):
    t0 = time.perf_counter(); c.begin_pass(0); c.push_reads_device(db, do, n, n * L); t1 = time.perf_counter(); c.finish_pass(); t2 = time.perf_counter()
    print("step %d: push %.0f ms, finish %.0f ms, %d segments, total %.0f ms" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, c.segment_count(), (t2 - t0) * 1e3), flush=True)
print("verified:", tuple(int(x) for x in c.result_checksum()) == tuple(int(x) for x in cs), "distinct %.3e" % c.stats()["kmers_nb_distinct"])
