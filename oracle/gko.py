"""ctypes loader for the CPU oracle (oracle/gkc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product path (gatb-core_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "libgkc_oracle.so")
    src = os.path.join(_HERE, "gkc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    vp = C.c_void_p
    L.gko_revcomp64.restype = C.c_uint64; L.gko_revcomp64.argtypes = [C.c_uint64, C.c_uint]
    L.gko_hash64.restype = C.c_uint64; L.gko_hash64.argtypes = [C.c_uint64, C.c_uint64]
    L.gko_oahash64.restype = C.c_uint64; L.gko_oahash64.argtypes = [C.c_uint64]
    L.gko_simplehash16_li1.restype = C.c_uint64; L.gko_simplehash16_li1.argtypes = [C.c_uint64, C.c_int]
    L.gko_simplehash16_ni64.restype = C.c_uint64; L.gko_simplehash16_ni64.argtypes = [C.c_uint64, C.c_int]
    L.gko_hash1_128.restype = C.c_uint64; L.gko_hash1_128.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    L.gko_revcomp128.restype = None
    L.gko_revcomp128.argtypes = [C.c_uint64, C.c_uint64, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gko_kmers.restype = C.c_int64
    L.gko_kmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, vp, vp, vp, vp, vp]
    L.gko_mmer_lut.restype = None; L.gko_mmer_lut.argtypes = [C.c_uint, C.c_int, u32p]
    L.gko_minimizers.restype = C.c_int64
    L.gko_minimizers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_uint, vp, u32p, u8p]
    L.gko_superkmers.restype = C.c_int64
    L.gko_superkmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, C.c_uint, vp, C.c_int, u32p, u32p, u32p,
                                 C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gko_superkmer_encode.restype = C.c_size_t; L.gko_superkmer_encode.argtypes = [C.c_char_p, C.c_uint, C.c_uint, u8p]
    L.gko_superkmer_decode.restype = C.c_size_t
    L.gko_superkmer_decode.argtypes = [vp, C.c_uint, u64p, u64p, C.POINTER(C.c_uint)]
    L.gko_freq_order_from_counts.restype = None; L.gko_freq_order_from_counts.argtypes = [C.c_uint, u32p, u32p]
    L.gko_count_mmers.restype = None; L.gko_count_mmers.argtypes = [C.c_char_p, C.c_uint64, C.c_uint, u32p]
    L.gko_repart_compute_distrib.restype = None; L.gko_repart_compute_distrib.argtypes = [C.c_uint, C.c_uint32, u64p, u16p]
    L.gko_repart_just_group_lexi.restype = None; L.gko_repart_just_group_lexi.argtypes = [C.c_uint, C.c_uint32, u64p, u16p]
    L.gko_repart_just_group.restype = None; L.gko_repart_just_group.argtypes = [C.c_uint, C.c_uint32, u64p, u32p, u16p]
    L.gko_dsk_run.restype = vp
    L.gko_dsk_run.argtypes = [vp, u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint32, C.c_uint32, u16p, vp,
                              C.c_int32, C.c_int32, C.c_uint32, C.c_int]
    L.gko_dsk_run_mt.restype = vp
    L.gko_dsk_run_mt.argtypes = [vp, u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint32, C.c_uint32, u16p, vp,
                                 C.c_int32, C.c_int32, C.c_uint32, C.c_int, C.c_uint32]
    L.gko_dsk_run_parts.restype = vp
    L.gko_dsk_run_parts.argtypes = [vp, u64p, C.c_uint64, C.c_uint, C.c_uint, C.c_uint32, C.c_uint32, u16p, vp,
                                    C.c_int32, C.c_int32, C.c_uint32, C.c_int, C.c_uint32, vp]
    L.gko_dsk_free.restype = None; L.gko_dsk_free.argtypes = [vp]
    L.gko_dsk_part_size.restype = C.c_uint64; L.gko_dsk_part_size.argtypes = [vp, C.c_uint32]
    L.gko_dsk_part_copy.restype = None; L.gko_dsk_part_copy.argtypes = [vp, C.c_uint32, u64p, u64p, i32p]
    L.gko_dsk_part_copy_records.restype = None; L.gko_dsk_part_copy_records.argtypes = [vp, C.c_uint32, u8p]
    L.gko_dsk_stats.restype = None; L.gko_dsk_stats.argtypes = [vp, u64p]
    L.gko_dsk_histogram.restype = None; L.gko_dsk_histogram.argtypes = [vp, u64p]
    L.gko_dsk_part_stats.restype = None
    L.gko_dsk_part_stats.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.gko_bloom_create.restype = vp; L.gko_bloom_create.argtypes = [C.c_int, C.c_uint64, C.c_uint, C.c_uint]
    L.gko_bloom_free.restype = None; L.gko_bloom_free.argtypes = [vp]
    L.gko_bloom_nbytes.restype = C.c_uint64; L.gko_bloom_nbytes.argtypes = [vp]
    L.gko_bloom_bitsize.restype = C.c_uint64; L.gko_bloom_bitsize.argtypes = [vp]
    L.gko_bloom_array.restype = C.POINTER(C.c_uint8); L.gko_bloom_array.argtypes = [vp]
    L.gko_bloom_seeds.restype = None; L.gko_bloom_seeds.argtypes = [C.c_uint64, u64p]
    L.gko_bloom_insert.restype = None; L.gko_bloom_insert.argtypes = [vp, u64p, vp, C.c_uint64]
    L.gko_bloom_contains.restype = None; L.gko_bloom_contains.argtypes = [vp, u64p, vp, C.c_uint64, u8p]
    L.gko_bloom_contains8.restype = None; L.gko_bloom_contains8.argtypes = [vp, u64p, vp, C.c_uint64, u8p]
    _LIB = L
    return L


# ---------------------------------------------------------------- convenience wrappers

def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fastx_parse(text):
    """FASTA/FASTQ text (bytes) -> (flat uint8 data, offsets uint64[n_seq+1]) with the reference reader's rules"""
    L = lib()
    buf = np.frombuffer(bytes(text), dtype=np.uint8)
    n = len(buf)
    data = np.zeros(max(n, 1), dtype=np.uint8)
    cap_seq = n // 2 + 2
    offs = np.zeros(cap_seq + 1, dtype=np.uint64)
    L.gko_fastx_parse.restype = C.c_int64
    L.gko_fastx_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
    r = L.gko_fastx_parse(_ptr(buf) if n else None, n, _ptr(data), len(data), _ptr(offs), cap_seq)
    if r < 0:
        raise RuntimeError("gko_fastx_parse: capacity")
    return data[:int(offs[r])].copy(), offs[:r + 1].copy()


def histogram_cutoff(histo, min_auto_threshold=3):
    """(cutoff, nb_solids, first_peak) of Histogram::compute_threshold on histo[0..length]"""
    L = lib()
    h = np.ascontiguousarray(histo, dtype=np.uint64)
    out = np.zeros(3, dtype=np.uint64)
    L.gko_histogram_cutoff.restype = None
    L.gko_histogram_cutoff.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    L.gko_histogram_cutoff(_ptr(h), len(h) - 1, min_auto_threshold, _ptr(out))
    return int(out[0]), int(out[1]), int(out[2])


class Mphf:
    """BooPHF restated (oracle/gkc_oracle.c gko_mphf_*)"""

    def __init__(self, keys, k):
        L = lib()
        L.gko_mphf_build.restype = C.c_void_p; L.gko_mphf_build.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int]
        L.gko_mphf_lookup.restype = C.c_uint64; L.gko_mphf_lookup.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.gko_mphf_save.restype = C.c_uint64; L.gko_mphf_save.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.gko_mphf_free.restype = None; L.gko_mphf_free.argtypes = [C.c_void_p]
        self.L = L; self.wide = k > 31
        a = np.zeros((len(keys), 2), dtype=np.uint64)
        for i, x in enumerate(keys):
            a[i, 0] = int(x) & 0xFFFFFFFFFFFFFFFF; a[i, 1] = int(x) >> 64
        self.n = len(keys)
        self.h = L.gko_mphf_build(_ptr(a), len(keys), 16, 1 if self.wide else 0)

    def lookup(self, keys):
        return np.array([self.L.gko_mphf_lookup(self.h, int(x) & 0xFFFFFFFFFFFFFFFF, int(x) >> 64) for x in keys], dtype=np.uint64)

    def save(self):
        n = self.L.gko_mphf_save(self.h, None, 0)
        out = np.zeros(n, np.uint8); self.L.gko_mphf_save(self.h, _ptr(out), n); return out

    def __del__(self):
        try:
            if self.h:
                self.L.gko_mphf_free(self.h); self.h = None
        except Exception:
            pass


def abundance_index(a):
    L = lib(); L.gko_abundance_index.restype = C.c_int; L.gko_abundance_index.argtypes = [C.c_int]
    return L.gko_abundance_index(int(a))


def pack_reads(reads):
    """list of bytes/str -> (flat uint8 array, offsets uint64[n+1])"""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    offs = np.zeros(len(bs) + 1, dtype=np.uint64)
    if bs:
        offs[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
    flat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy() if bs else np.zeros(0, np.uint8)
    return flat, offs


def kmers(seq, k):
    """-> dict of arrays fwd, can (python ints if k>31 else uint64), valid"""
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    n = max(0, len(s) - k + 1)
    flo = np.zeros(n, np.uint64); fhi = np.zeros(n, np.uint64)
    clo = np.zeros(n, np.uint64); chi = np.zeros(n, np.uint64); v = np.zeros(n, np.uint8)
    lib().gko_kmers(s, len(s), k, _ptr(flo), _ptr(fhi), _ptr(clo), _ptr(chi), _ptr(v))
    return dict(fwd_lo=flo, fwd_hi=fhi, can_lo=clo, can_hi=chi, valid=v)


def minimizers(seq, k, m, freq_order=None):
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    n = max(0, len(s) - k + 1)
    out = np.zeros(n, np.uint32); v = np.zeros(n, np.uint8)
    lib().gko_minimizers(s, len(s), k, m, _ptr(freq_order), out, v)
    return out, v


def superkmers(seq, k, m, freq_order=None, maxs=0):
    s = seq.encode() if isinstance(seq, str) else bytes(seq)
    cap = max(1, len(s))
    mn = np.zeros(cap, np.uint32); st = np.zeros(cap, np.uint32); nb = np.zeros(cap, np.uint32)
    nv = C.c_uint64(0); ni = C.c_uint64(0)
    n = lib().gko_superkmers(s, len(s), k, m, _ptr(freq_order), maxs, mn, st, nb, cap, C.byref(nv), C.byref(ni))
    return mn[:n].copy(), st[:n].copy(), nb[:n].copy(), nv.value, ni.value


def mmer_lut(m, has_freq=False):
    lut = np.zeros(4 ** m, np.uint32)
    lib().gko_mmer_lut(m, int(has_freq), lut)
    return lut


class Dsk:
    """Result of the oracle's SortingCountAlgorithm restatement."""

    def __init__(self, bases, offsets, k, m, nb_partitions, repart, nb_passes=1, freq_order=None,
                 abundance_min=1, abundance_max=2147483647, histo_max=10000, maxs=0, threads=1, only_parts=None):
        L = lib()
        self.k = k; self.nb_partitions = nb_partitions; self.nb_passes = nb_passes; self.histo_max = histo_max
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        repart = np.ascontiguousarray(repart, dtype=np.uint16)
        assert repart.size == 4 ** m
        # threads > 1: the same run parallelised like the reference (reads shared out for fillPartitions, partitions for fillSolidKmers)
        # only_parts: count nothing but these partitions (gko_dsk_run_parts): an exact count of sampled partitions of a full-size input
        keep = None
        if only_parts is not None:
            keep = np.zeros(nb_partitions, np.uint8); keep[np.asarray(list(only_parts), dtype=np.int64)] = 1
        self._h = L.gko_dsk_run_parts(_ptr(bases), offsets, len(offsets) - 1, k, m, nb_partitions, nb_passes, repart,
                                      _ptr(freq_order), abundance_min, abundance_max, histo_max, maxs, threads, _ptr(keep))
        s = np.zeros(8, np.uint64); L.gko_dsk_stats(self._h, s)
        self.stats = dict(kmers_nb_valid=int(s[0]), kmers_nb_invalid=int(s[1]), kmers_nb_distinct=int(s[2]),
                          kmers_nb_solid=int(s[3]), nb_superkmers=int(s[4]), nb_sequences=int(s[5]),
                          superkmer_bytes=int(s[6]), nb_short_sequences=int(s[7]))

    def part(self, d):
        """-> (lo uint64[], hi uint64[], abundance int32[]) of dataset d, ascending"""
        L = lib(); n = L.gko_dsk_part_size(self._h, d)
        lo = np.zeros(n, np.uint64); hi = np.zeros(n, np.uint64); ab = np.zeros(n, np.int32)
        L.gko_dsk_part_copy(self._h, d, lo, hi, ab)
        return lo, hi, ab

    def part_records(self, d):
        L = lib(); n = L.gko_dsk_part_size(self._h, d)
        rec = 16 if self.k <= 31 else 32
        out = np.zeros(max(1, n * rec), np.uint8)
        L.gko_dsk_part_copy_records(self._h, d, out)
        return out[: n * rec]

    def part_stats(self, d):
        a = C.c_uint64(0); b = C.c_uint64(0)
        lib().gko_dsk_part_stats(self._h, d, C.byref(a), C.byref(b))
        return a.value, b.value

    def histogram(self):
        h = np.zeros(self.histo_max + 1, np.uint64)
        lib().gko_dsk_histogram(self._h, h)
        return h

    def all_counts(self):
        """dict {kmer int: count} over all datasets"""
        out = {}
        for d in range(self.nb_partitions * self.nb_passes):
            lo, hi, ab = self.part(d)
            for a, b, c in zip(lo.tolist(), hi.tolist(), ab.tolist()):
                out[(b << 64) | a] = c
        return out

    def close(self):
        if self._h:
            lib().gko_dsk_free(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bloom:
    KINDS = {"basic": 0, "cache": 1, "neighbor": 2}

    def __init__(self, kind, tai_bits, nb_hash, k):
        self.kind = kind; self.k = k
        self._h = lib().gko_bloom_create(self.KINDS[kind], tai_bits, nb_hash, k)

    @staticmethod
    def _split(keys):
        ks = [int(x) for x in keys]
        lo = np.array([x & 0xFFFFFFFFFFFFFFFF for x in ks], dtype=np.uint64)
        hi = np.array([x >> 64 for x in ks], dtype=np.uint64)
        return lo, hi

    def insert(self, keys):
        lo, hi = self._split(keys)
        lib().gko_bloom_insert(self._h, lo, _ptr(hi), len(lo))

    def contains(self, keys):
        lo, hi = self._split(keys); out = np.zeros(len(lo), np.uint8)
        lib().gko_bloom_contains(self._h, lo, _ptr(hi), len(lo), out)
        return out

    def contains8(self, keys):
        lo, hi = self._split(keys); out = np.zeros(len(lo), np.uint8)
        lib().gko_bloom_contains8(self._h, lo, _ptr(hi), len(lo), out)
        return out

    @property
    def nbytes(self):
        return lib().gko_bloom_nbytes(self._h)

    @property
    def bitsize(self):
        return lib().gko_bloom_bitsize(self._h)

    def array(self):
        n = self.nbytes
        return np.ctypeslib.as_array(lib().gko_bloom_array(self._h), shape=(n,)).copy()

    def __del__(self):
        try:
            if self._h:
                lib().gko_bloom_free(self._h); self._h = None
        except Exception:
            pass
