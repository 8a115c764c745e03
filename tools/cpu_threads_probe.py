import sys, time, threading, os
sys.path.insert(0, '/root/repo')
import bench, __graft_entry__ as ge
from oracle import gko
gkc = ge.load().gkc
k, m, parts = 31, 10, 256
rep = bench.repart_for_bench(m, parts)
mb, mo = gkc.synth_reads_np(1, 50000, 150, 250000, 10000)
for T in (16, 32, 64, 96, 128):
    res = [0] * T
    def work(i):
        for _ in range(4):
            res[i] += gko.Dsk(mb, mo, k, m, parts, rep).stats["kmers_nb_distinct"]
    th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.time(); [t.start() for t in th]; [t.join() for t in th]; dt = time.time() - t0
    print(T, round(dt, 2), "%.3g" % (sum(res) / dt))
