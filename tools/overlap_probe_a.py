"""Probe: do two Stage-A pushes on two contexts (two streams) overlap usefully? sequential vs two host threads."""
import sys, time, threading
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import __graft_entry__ as ge
import bench
gkc = ge.load().gkc
k, m, parts, n = 31, 10, 4096, 50_000_000
rep = bench.repart_for_bench(m, parts)
ctxs = []
for i in range(2):
    c = gkc.Counter(0); c.configure(k, m, parts, rep)
    db, do = c.synth_reads_device(2 + i, n, 150, n * 5, 10000)
    ctxs.append((c, db, do))
def stage_a(c, db, do):
    c.begin_pass(0); c.push_reads_device(db, do, n, n * 150)
for mode in ("seq", "par", "seq", "par", "stagger"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode == "seq":
        for c, db, do in ctxs: stage_a(c, db, do)
    elif mode == "par":
        th = [threading.Thread(target=stage_a, args=x) for x in ctxs]
        [t.start() for t in th]; [t.join() for t in th]
    else:
        th = [threading.Thread(target=stage_a, args=x) for x in ctxs]
        th[0].start(); time.sleep(0.012); th[1].start(); [t.join() for t in th]
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(mode, "stage A of 2 x %d reads: %.1f ms" % (n, dt * 1e3))
