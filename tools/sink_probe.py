"""(GPU box) host-landed steps alone: 1e8 reads (or argv[1]), results streamed into a page-locked sink, abundance-min 1 and 2; GKC_SINK_DEBUG=1 prints per-batch
pack / copy / unpack times, GKC_SINK_PACKED=0 the plain 16-byte copies.   python tools/sink_probe.py [n_reads] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
gkc = ge.load().gkc
import bench
k, m, L = 31, 10, 150
parts = int(min(32768, max(64, 2 ** int(np.floor(np.log2(max(1, n * (L - k + 1) / 3.0e6)) + 0.5)))))
c = gkc.Counter(0); c.configure(k, m, parts, bench.repart_for_bench(m, parts))
db, do = c.synth_reads_device(2, n, L, n * 5, 10000)
def step():
    c.begin_pass(0); c.push_reads_device(db, do, n, n * L); c.finish_pass()
step()
distinct = c.stats()["kmers_nb_distinct"]
sink = gkc.HostBuffer(int(distinct * 16 * 1.01) + (64 << 20))
t0 = time.perf_counter(); c.set_host_sink(sink); print("set_host_sink: %.2f s" % (time.perf_counter() - t0), flush=True)
for amin in [int(x) for x in os.environ.get('PROBE_AMIN', '1,2').split(',')]:
    c.set_solidity(amin, 2147483647, 10000)
    step()
    t0 = time.perf_counter(); per = []
    for _ in range(steps):
        t1 = time.perf_counter(); step(); per.append("%.0f" % ((time.perf_counter() - t1) * 1e3))
    dt = (time.perf_counter() - t0) / steps
    print("   per step (ms):", " ".join(per), "| median %s" % sorted(per, key=float)[len(per) // 2], flush=True)
    st = c.stats()
    print("abundance-min %d: %.1f ms per step, %d solid records (%.1f GB of Count[]), %.2e distinct k-mers/s" % (amin, dt * 1e3, st["kmers_nb_solid"], st["kmers_nb_solid"] * 16 / 1e9, st["kmers_nb_distinct"] / dt), flush=True)
    for nme in ("total_stage_a", "total_stage_b", "compact"):
        print("   ", nme, c.timing(nme))
c.set_host_sink(None)
