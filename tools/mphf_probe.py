"""(GPU box) MPHF build over the solid k-mers of the bench input, several builds in a row (the first pays the allocator's first-touch hipMalloc): wall per build
with the region build and with the atomic path. usage: python tools/mphf_probe.py [reads=100000000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
from bench import repart_for_bench
gkc = ge.load().gkc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k, m, L, parts = 31, 10, 150, 4096
c = gkc.Counter(0); c.configure(k, m, parts, repart_for_bench(m, parts)); c.set_solidity(2, 2147483647, 10000)
b, o = c.synth_reads_device(2, n, L, n * 5, 10000)
c.begin_pass(0); c.push_reads_device(b, o, n, n * L); c.finish_pass()
print("solid", c.stats()["kmers_nb_solid"], flush=True)
ref = None
for tag, env in (("regions", None), ("regions", None), ("regions", None), ("atomic", "0"), ("atomic", "0"), ("atomic", "0")):
    if env is None: os.environ.pop("GKC_MPHF_REGIONS", None)
    else: os.environ["GKC_MPHF_REGIONS"] = env
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mp = gkc.Mphf(c)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s = mp.save(); h = hash(s.tobytes())
    if ref is None: ref = h
    torch.cuda.synchronize(); t1 = time.perf_counter()
    amap, above = mp.abundance_map()
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
    hm = hash(amap.tobytes())
    if tag + "_map" not in globals(): globals()[tag + "_map"] = hm
    print(tag, "build %.1f ms" % (dt * 1e3), "same bytes" if h == ref else "DIFFERENT", "| abundance map (incl. D2H of %d MB) %.1f ms" % (len(amap) >> 20, dt2 * 1e3), "map hash", hm & 0xFFFFFF, flush=True)
    mp.close()
