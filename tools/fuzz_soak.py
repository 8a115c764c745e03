"""Developer soak (GPU box): the fuzz of tests/test_gpu_parity.py::test_fuzz_random_configurations over many more seeds, plus low-complexity inserts
(repeated minimizers: the capped-run path of Stage A, oversize buckets of Stage B). usage: python tools/fuzz_soak.py [first_seed] [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
from oracle import gko
from tests.util import synth_reads
from tests.test_gpu_parity import device_vs_oracle
gkc = ge.load().gkc
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
bad = 0
for seed in range(first, first + n):
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.integers(5, 64)); m = int(rng.integers(2, min(k - 1, 11) + 1)); parts = int(rng.integers(1, 40))
    passes = int(rng.integers(1, 4)); batches = int(rng.integers(1, 4))
    amin = int(rng.integers(1, 4)); amax = max(amin, int(rng.choice([2147483647, 5, 60]))); histo_max = int(rng.choice([10000, 7, 100]))
    n_reads = int(rng.integers(50, 1500)); glen = int(rng.integers(300, 20000)); rlen = int(rng.integers(max(k, 20), 220))
    reads = synth_reads(n_reads, glen, rlen, seed=seed, sub_rate=float(rng.choice([0.0, 0.01, 0.05])), n_rate=float(rng.choice([0.0, 0.002])), ragged=bool(rng.integers(0, 2)))
    if seed % 2 == 0:                                             # low-complexity inserts at random places
        units = ["A", "AC", "ACG", "AAT", "ACGTTGC", "T"]
        for _ in range(int(rng.integers(1, 30))):
            u = units[int(rng.integers(0, len(units)))]; rep = (u * (int(rng.integers(20, 400)) // len(u) + 1))
            i = int(rng.integers(0, len(reads)))
            r = reads[i]; r = r.decode() if isinstance(r, bytes) else r
            cut = int(rng.integers(0, len(r) + 1))
            reads[i] = (r[:cut] + rep + r[cut:]).encode() if isinstance(reads[i], bytes) else r[:cut] + rep + r[cut:]
    freq = None
    if seed % 3 == 0:
        L = gko.lib(); counts = np.zeros(4 ** m, np.uint32)
        for r in reads[: max(10, n_reads // 4)]:
            rb = r if isinstance(r, bytes) else r.encode()
            L.gko_count_mmers(rb, len(rb), m, counts)
        freq = np.zeros(4 ** m, np.uint32); L.gko_freq_order_from_counts(m, counts, freq)
    try:
        device_vs_oracle(gkc, reads, k, m, parts, passes=passes, batches=batches, amin=amin, amax=amax, histo_max=histo_max, freq=freq)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, (k, m, parts, passes, batches), str(e)[:200], flush=True)
print("soak done:", n, "seeds,", bad, "mismatches", flush=True)
