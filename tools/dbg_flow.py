import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
from oracle import gko
gkc = ge.load().gkc
v = json.load(open("/root/repo/tests/golden/reference_unit_vectors.json"))["dsk_check1"]
seqs = v["seqs4"]; k = 9; m = 8; P = 4
bases, offs = gko.pack_reads(seqs)
c = gkc.Counter(0)
c.configure(k, m, 1, np.zeros(4 ** m, np.uint16))
nsk, nk = c.sample_minimizers(bases, offs)
print("sample kmers", nk.sum(), "superk", nsk.sum())
# computeDistrib-like
order = np.argsort(-nk.astype(np.int64), kind="stable")
used = np.zeros(P, np.int64); table = np.zeros(4 ** m, np.uint16)
for i in order:
    j = int(np.argmin(used)); table[i] = j; used[j] += int(nk[i])
c.set_solidity(1, 2147483647, 10000)
c.configure(k, m, P, table)
c.count(bases, offs)
print(c.stats())
ref = gko.Dsk(bases, offs, k, m, P, table)
print(ref.stats)
