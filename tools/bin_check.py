import sys, json, numpy as np, os
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
from oracle import gko
from tests.util import simple_repart, synth_reads
gkc = ge.load().gkc
reads = synth_reads(30000, 60000, 150, seed=97, n_rate=0.001, ragged=True)
reads += [b"A" * 150] * 300 + [b"ACACACACAC" * 15] * 300 + [(b"ACGTTGCA" * 19)[:150]] * 200
bases, offs = gko.pack_reads(reads)
for k, m, parts in ((31, 8, 3), (21, 7, 2), (31, 8, 64)):
    rep = simple_repart(m, parts)
    c = gkc.Counter(0); c.configure(k, m, parts, rep); c.count(bases, offs)
    ref = gko.Dsk(bases, offs, k, m, parts, rep, threads=4)
    ok = all(np.array_equal(c.partition_records(0, p), ref.part_records(p)) for p in range(parts)) and np.array_equal(c.histogram(), ref.histogram())
    print(k, m, parts, "OK" if ok else "MISMATCH", c.stats()["kmers_nb_distinct"], ref.stats["kmers_nb_distinct"], flush=True)
