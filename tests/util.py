"""Shared helpers for tests: seeded synthetic reads, naive dictionary counter (independent of the oracle)."""
import numpy as np

CODE = {65: 0, 67: 1, 84: 2, 71: 3, 97: 0, 99: 1, 116: 2, 103: 3}  # A C T G (GATB order)


def naive_counts(reads, k):
    """Pure-Python canonical k-mer counter (the simplest possible statement of the semantics): dict kmer->count"""
    out = {}
    mask = (1 << (2 * k)) - 1
    for r in reads:
        b = r.encode() if isinstance(r, str) else bytes(r)
        f = 0; rc = 0; good = 0
        for c in b:
            if c in CODE:
                code = CODE[c]
                f = ((f << 2) | code) & mask
                rc = (rc >> 2) | ((code ^ 2) << (2 * (k - 1)))
                good += 1
            else:
                good = 0; f = 0; rc = 0
            if good >= k:
                key = f if f < rc else rc
                out[key] = out.get(key, 0) + 1
    return out


def revcomp_int(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | ((x & 3) ^ 2)
        x >>= 2
    return r


def str2int(s):
    v = 0
    for ch in s:
        v = (v << 2) | CODE[ord(ch)]
    return v


def synth_reads(n_reads, genome_len, read_len=150, seed=1, sub_rate=0.01, n_rate=0.0, ragged=False):
    """Seeded synthetic reads: uniform genome, both strands, substitutions; optional N's and ragged lengths."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    alpha = np.frombuffer(b"ACTG", dtype=np.uint8)
    comp = np.array([2, 3, 0, 1], dtype=np.uint8)
    reads = []
    for i in range(n_reads):
        L = read_len
        if ragged:
            u = rng.random()
            if u < 0.05:
                L = int(rng.integers(1, 25))
            elif u < 0.10:
                L = int(rng.integers(read_len + 1, 2 * read_len + 1))
        L = min(L, genome_len)
        s = int(rng.integers(0, genome_len - L + 1))
        r = genome[s:s + L].copy()
        if rng.random() < 0.5:
            r = comp[r[::-1]]
        if sub_rate > 0:
            m = rng.random(L) < sub_rate
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3
        a = alpha[r].copy()
        if n_rate > 0:
            a[rng.random(L) < n_rate] = ord("N")
        reads.append(a.tobytes())
    return reads


def simple_repart(m, nb_partitions, seed=7):
    """A deterministic minimizer->partition table (any table is a valid Repartitor)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, nb_partitions, 4 ** m, dtype=np.uint16)
