"""ctypes binding over the C-ABI of libgkc_hip.so (include/gkc.h). No torch types cross the boundary.

The extension is REQUIRED: every entry point raises GkcError if the library is missing or a call fails;
there is no CPU fallback anywhere in this module (the CPU oracle lives in oracle/ and is test-only).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.environ.get("GKC_LIB") or os.path.join(CSRC, "libgkc_hip.so")      # GKC_LIB: an experiment build of the same library (tools/build_variant.sh)
_LIB = None

SYMBOLS = [
    "gkc_create", "gkc_destroy", "gkc_last_error", "gkc_version", "gkc_configure", "gkc_set_solidity",
    "gkc_set_max_superkmer", "gkc_set_batch_keys", "gkc_begin_pass", "gkc_push_reads", "gkc_push_reads_device", "gkc_finish_pass",
    "gkc_partition_info", "gkc_partition_counts", "gkc_partition_counts_range", "gkc_partition_counts_device", "gkc_histogram", "gkc_get_stats",
    "gkc_get_timing", "gkc_partition_superkmers", "gkc_segment_count", "gkc_segment_export", "gkc_segment_import",
    "gkc_segments_clear", "gkc_bloom_create", "gkc_bloom_destroy", "gkc_bloom_nbytes", "gkc_bloom_bitsize",
    "gkc_bloom_insert", "gkc_bloom_insert_device", "gkc_bloom_insert_solid", "gkc_bloom_query_solid", "gkc_bloom_contains",
    "gkc_bloom_contains8", "gkc_bloom_get_array", "gkc_bloom_set_array", "gkc_bloom_device_array", "gkc_synth_reads_device", "gkc_synth_reads_profile_device", "gkc_device_free", "gkc_host_alloc", "gkc_host_free", "gkc_release_pass", "gkc_device_memory",
    "gkc_fastx_parse_device", "gkc_push_fastx", "gkc_mphf_build", "gkc_mphf_build_solid", "gkc_mphf_destroy", "gkc_mphf_size",
    "gkc_mphf_lookup", "gkc_mphf_save_size", "gkc_mphf_save", "gkc_mphf_abundance_map",
    "gkc_device_to_host", "gkc_kmer_checksum_device", "gkc_result_checksum", "gkc_sample_minimizers", "gkc_count_mmers",
    "gkc_host_to_device", "gkc_comm_unique_id", "gkc_comm_create_rccl", "gkc_comm_create_transport", "gkc_comm_create_files", "gkc_comm_enable_ipc", "gkc_gather_results", "gkc_comm_loopback", "gkc_comm_selftest", "gkc_comm_peer_bytes", "gkc_comm_destroy", "gkc_comm_set_owners",
    "gkc_comm_get_owners", "gkc_balanced_owner_ranges", "gkc_exchange", "gkc_comm_get_stats", "gkc_bloom_allreduce_or",
    "gkc_mphf_build_solid_dist", "gkc_mphf_abundance_map_dist", "gkc_exchange_plan",
    "gkc_sample_exact", "gkc_set_host_sink", "gkc_set_sink_mode", "gkc_finish_pass_async", "gkc_wait_partition", "gkc_finish_pass_wait",
]


class GkcError(RuntimeError):
    pass


class Xfer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("reserved", C.c_int32), ("d_ptr", C.c_void_p), ("n_bytes", C.c_uint64)]


ALLGATHER_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)
SENDRECV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(Xfer), C.c_uint32, C.POINTER(Xfer), C.c_uint32)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather_host", ALLGATHER_CB), ("sendrecv_device", SENDRECV_CB)]


class PlanMsg(C.Structure):
    _fields_ = [("peer", C.c_int32), ("seg", C.c_uint32), ("rec_begin", C.c_uint64), ("n_recs", C.c_uint64)]


class CommStats(C.Structure):
    _fields_ = [("n_exchanges", C.c_uint64), ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("ms_transfer", C.c_double),
                ("ms_host", C.c_double), ("reserved", C.c_uint64 * 4)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "kmers_nb_valid", "kmers_nb_invalid", "kmers_nb_distinct", "kmers_nb_solid", "nb_superkmers", "nb_sequences",
        "nb_bases", "superkmer_bytes", "oversize_buckets", "dedupe_kmers_in", "dedupe_keys_out", "seq_len_min", "seq_len_max", "seq_len_sq_sum")] + [("reserved", C.c_uint64 * 2)]


def build(force=False):
    """Compile csrc/*.hip for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "gkc.h"))
    stale = force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO):
        raise GkcError("libgkc_hip.so is missing (%s): run __graft_entry__.build(); there is no CPU fallback" % SO)
    L = C.CDLL(SO)
    vp, u64, u32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32
    P = C.POINTER
    sig = {
        "gkc_create": (C.c_int, [C.c_int, P(vp)]),
        "gkc_destroy": (None, [vp]),
        "gkc_last_error": (C.c_char_p, [vp]),
        "gkc_version": (C.c_char_p, []),
        "gkc_configure": (C.c_int, [vp, u32, u32, u32, u32, C.c_int, vp, vp]),
        "gkc_set_solidity": (C.c_int, [vp, i32, i32, u32]),
        "gkc_set_max_superkmer": (C.c_int, [vp, u32]),
        "gkc_set_batch_keys": (C.c_int, [vp, u64]),
        "gkc_begin_pass": (C.c_int, [vp, u32]),
        "gkc_push_reads": (C.c_int, [vp, vp, vp, u64]),
        "gkc_push_reads_device": (C.c_int, [vp, vp, vp, u64, u64]),
        "gkc_finish_pass": (C.c_int, [vp]),
        "gkc_partition_info": (C.c_int, [vp, u32, u32, P(u64), P(u64), P(u64)]),
        "gkc_partition_counts": (C.c_int, [vp, u32, u32, vp, u64, P(u64)]),
        "gkc_partition_counts_range": (C.c_int, [vp, u32, u32, u64, u64, vp]),
        "gkc_partition_counts_device": (C.c_int, [vp, u32, u32, P(vp), P(u64)]),
        "gkc_histogram": (C.c_int, [vp, vp, u32]),
        "gkc_get_stats": (C.c_int, [vp, P(Stats)]),
        "gkc_get_timing": (C.c_int, [vp, C.c_char_p, P(C.c_double), P(u64)]),
        "gkc_partition_superkmers": (C.c_int, [vp, u32, vp, u64, P(u64), P(u64), P(u64)]),
        "gkc_segment_count": (C.c_int, [vp, P(u32)]),
        "gkc_segment_export": (C.c_int, [vp, u32, P(vp), P(u32), vp, vp]),
        "gkc_segment_import": (C.c_int, [vp, vp, vp, vp]),
        "gkc_segments_clear": (C.c_int, [vp]),
        "gkc_bloom_create": (C.c_int, [vp, C.c_int, u64, u32, u32, P(vp)]),
        "gkc_bloom_destroy": (None, [vp]),
        "gkc_bloom_nbytes": (u64, [vp]),
        "gkc_bloom_bitsize": (u64, [vp]),
        "gkc_bloom_insert": (C.c_int, [vp, vp, u64, u32]),
        "gkc_bloom_insert_device": (C.c_int, [vp, vp, u64, u32]),
        "gkc_bloom_insert_solid": (C.c_int, [vp, vp]),
        "gkc_bloom_query_solid": (C.c_int, [vp, vp, C.c_int, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "gkc_bloom_contains": (C.c_int, [vp, vp, u64, u32, vp]),
        "gkc_bloom_contains8": (C.c_int, [vp, vp, u64, u32, vp]),
        "gkc_bloom_get_array": (C.c_int, [vp, vp, u64]),
        "gkc_bloom_set_array": (C.c_int, [vp, vp, u64]),
        "gkc_bloom_device_array": (C.c_int, [vp, P(vp), P(u64)]),
        "gkc_synth_reads_device": (C.c_int, [vp, u64, u64, u64, u32, u64, u32, P(vp), P(vp)]),
        "gkc_synth_reads_profile_device": (C.c_int, [vp, u64, u64, u64, u32, u64, u32, u32, P(vp), P(vp)]),
        "gkc_device_free": (C.c_int, [vp, vp]),
        "gkc_host_alloc": (C.c_int, [P(vp), u64]),
        "gkc_host_free": (C.c_int, [vp]),
        "gkc_release_pass": (C.c_int, [vp, u32]),
        "gkc_device_memory": (C.c_int, [vp, P(u64), P(u64)]),
        "gkc_device_to_host": (C.c_int, [vp, vp, vp, u64]),
        "gkc_fastx_parse_device": (C.c_int, [vp, vp, u64, C.c_int, P(vp), P(vp), P(u64), P(u64), P(u64)]),
        "gkc_push_fastx": (C.c_int, [vp, vp, u64, C.c_int, P(u64)]),
        "gkc_mphf_build": (C.c_int, [vp, vp, u64, u32, u32, P(vp)]), "gkc_mphf_build_solid": (C.c_int, [vp, P(vp)]),
        "gkc_mphf_destroy": (None, [vp]), "gkc_mphf_size": (u64, [vp]), "gkc_mphf_lookup": (C.c_int, [vp, vp, u64, u32, vp]),
        "gkc_mphf_save_size": (u64, [vp]), "gkc_mphf_save": (C.c_int, [vp, vp, u64]),
        "gkc_mphf_abundance_map": (C.c_int, [vp, vp, vp, u64, P(u64)]),
        "gkc_kmer_checksum_device": (C.c_int, [vp, vp, vp, u64, u64, P(u64), P(u64)]),
        "gkc_result_checksum": (C.c_int, [vp, P(u64), P(u64)]),
        "gkc_sample_minimizers": (C.c_int, [vp, vp, vp, u64, vp, vp]),
        "gkc_count_mmers": (C.c_int, [vp, u32, vp, vp, u64, vp]),
        "gkc_host_to_device": (C.c_int, [vp, vp, vp, u64]),
        "gkc_comm_unique_id": (C.c_int, [vp]),
        "gkc_comm_create_rccl": (C.c_int, [vp, vp, C.c_int, C.c_int, P(vp)]),
        "gkc_comm_create_transport": (C.c_int, [vp, P(Transport), C.c_int, C.c_int, P(vp)]),
        "gkc_comm_create_files": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, P(vp)]),
        "gkc_comm_enable_ipc": (C.c_int, [vp, C.c_int]),
        "gkc_gather_results": (C.c_int, [vp, vp, C.c_int]),
        "gkc_comm_loopback": (C.c_int, [vp, vp, u64, P(u64), P(C.c_double)]),
        "gkc_comm_selftest": (C.c_int, [vp, vp, u64, P(u64), P(C.c_double)]),
        "gkc_comm_peer_bytes": (C.c_int, [vp, vp, vp, P(C.c_double)]),
        "gkc_comm_destroy": (None, [vp]),
        "gkc_comm_set_owners": (C.c_int, [vp, vp]),
        "gkc_comm_get_owners": (C.c_int, [vp, vp]),
        "gkc_balanced_owner_ranges": (C.c_int, [vp, u32, C.c_int, vp]),
        "gkc_exchange": (C.c_int, [vp, vp]),
        "gkc_comm_get_stats": (C.c_int, [vp, P(CommStats)]),
        "gkc_bloom_allreduce_or": (C.c_int, [vp, vp]),
        "gkc_mphf_build_solid_dist": (C.c_int, [vp, vp, P(vp)]),
        "gkc_mphf_abundance_map_dist": (C.c_int, [vp, vp, vp, vp, u64, P(u64)]),
        "gkc_sample_exact": (C.c_int, [vp, vp, vp, u64, u64, vp, vp, vp, P(u64)]),
        "gkc_set_host_sink": (C.c_int, [vp, vp, u64]),
        "gkc_set_sink_mode": (C.c_int, [vp, C.c_int]),
        "gkc_finish_pass_async": (C.c_int, [vp]),
        "gkc_wait_partition": (C.c_int, [vp, u32, u32, P(vp), P(u64)]),
        "gkc_finish_pass_wait": (C.c_int, [vp]),
        "gkc_exchange_plan": (C.c_int, [C.c_int, C.c_int, u32, vp, vp, u64, vp, P(PlanMsg), P(u32), P(PlanMsg), P(u32), P(u64)]),
    }
    for name in SYMBOLS:
        f = getattr(L, name)          # raises AttributeError if the symbol is not exported
        f.restype, f.argtypes = sig[name]
    _LIB = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---- host twins of device helpers (full-size parity properties) ----
M64 = (1 << 64) - 1


def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def mix64_np(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def synth_rnd_np(seed, stream, idx):
    with np.errstate(over="ignore"):
        base = mix64_np(np.uint64(seed) ^ (np.uint64(stream) << np.uint64(56)))
        return mix64_np(base + np.asarray(idx, dtype=np.uint64))


def synth_genome_np(seed, pos, profile=0):
    """numpy twin of synth_genome_code (csrc/gkc_api.hip): 2-bit codes of the genome at `pos`"""
    pos = np.asarray(pos, dtype=np.uint64)
    code = (synth_rnd_np(seed, 1, pos) & np.uint64(3)).astype(np.uint32)
    if profile == 1:
        slot = pos >> np.uint64(13); inn = pos & np.uint64(8191)
        h = synth_rnd_np(seed, 4, slot)
        f = (h >> np.uint64(8)) % np.uint64(50)
        ln = np.uint64(1000) + synth_rnd_np(seed, 5, f) % np.uint64(4001)
        rep = ((h & np.uint64(3)) == 0) & (inn < ln)
        fc = (synth_rnd_np(seed, 6, f * np.uint64(8192) + inn) & np.uint64(3)).astype(np.uint32)
        d = synth_rnd_np(seed, 7, pos)
        fc = np.where(d % np.uint64(1000) < np.uint64(5), (fc + 1 + ((d >> np.uint64(32)) % np.uint64(3)).astype(np.uint32)) & 3, fc)
        code = np.where(rep, fc, code)
    return code


def synth_reads_np(seed, n_reads, read_len, genome_len, sub_ppm, first_read=0, profile=0):
    """numpy twin of the k_synth_reads kernel (csrc/gkc_api.hip): -> (bases uint8[n*L], offsets uint64[n+1]); profile 1 = GKC_SYNTH_SKEWED"""
    i = np.repeat(np.arange(first_read, first_read + n_reads, dtype=np.uint64), read_len)
    j = np.tile(np.arange(read_len, dtype=np.uint64), n_reads)
    g = (i - np.uint64(first_read)) * np.uint64(read_len) + j
    u = synth_rnd_np(seed, 2, i)
    start = (u >> np.uint64(1)) % np.uint64(genome_len - read_len + 1)
    rev = (u & np.uint64(1)).astype(bool)
    pos = np.where(rev, start + np.uint64(read_len - 1) - j, start + j)
    code = synth_genome_np(seed, pos, profile)
    code = np.where(rev, code ^ 2, code)
    if profile == 1:
        hl = synth_rnd_np(seed, 8, i)
        ul = np.uint64(1) + (hl >> np.uint64(8)) % np.uint64(3)
        lc = ((hl >> (np.uint64(16) + np.uint64(2) * (j % ul))) & np.uint64(3)).astype(np.uint32)
        code = np.where(hl % np.uint64(100) == 0, lc, code)
    v = synth_rnd_np(seed, 3, np.uint64(first_read) * np.uint64(read_len) + g)
    sub = (v % np.uint64(1000000)) < np.uint64(sub_ppm)
    code = np.where(sub, (code + 1 + ((v >> np.uint64(32)) % np.uint64(3)).astype(np.uint32)) & 3, code)
    bases = np.frombuffer(b"ACTG", dtype=np.uint8)[code]
    return bases.copy(), np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)


class HostBuffer:
    """page-locked host memory (gkc_host_alloc) as a numpy uint8 array: ``buf.a``; freed with the object"""

    def __init__(self, nbytes):
        self._p = C.c_void_p()
        rc = lib().gkc_host_alloc(C.byref(self._p), int(nbytes))
        if rc != 0 or not self._p.value:
            raise GkcError("gkc_host_alloc(%d) failed (%d)" % (nbytes, rc))
        self.nbytes = int(nbytes)
        self.a = np.ctypeslib.as_array((C.c_uint8 * max(1, self.nbytes)).from_address(self._p.value))[: self.nbytes]

    def __del__(self):
        try:
            if self._p.value:
                lib().gkc_host_free(self._p); self._p = C.c_void_p()
        except Exception:
            pass


class Counter:
    """Host-side handle over one gkc_ctx (one GPU). Mirrors the call protocol of SortingCountAlgorithm::execute
    (reference kmer/impl/SortingCountAlgorithm.cpp:636-781): configure -> per pass: begin, push reads, finish -> fetch."""

    def __init__(self, device=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.gkc_create(device, C.byref(h))
        if rc != 0:
            raise GkcError("gkc_create failed (%d): %s" % (rc, (self.L.gkc_last_error(None) or b"").decode()))
        self.h = h
        self.k = self.m = self.nb_partitions = self.nb_passes = None
        self._keep = []

    def _chk(self, rc):
        if rc != 0:
            raise GkcError("gkc error %d: %s" % (rc, (self.L.gkc_last_error(self.h) or b"").decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.gkc_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def configure(self, k, m, nb_partitions, repart, nb_passes=1, freq_order=None):
        repart = np.ascontiguousarray(repart, dtype=np.uint16)
        if repart.size != 4 ** m:
            raise GkcError("repart must have 4^m entries")
        fo = None if freq_order is None else np.ascontiguousarray(freq_order, dtype=np.uint32)
        self._chk(self.L.gkc_configure(self.h, k, m, nb_partitions, nb_passes, 0 if fo is None else 1, _p(repart), _p(fo)))
        self.k, self.m, self.nb_partitions, self.nb_passes = k, m, nb_partitions, nb_passes
        self.rec_bytes = 16 if k <= 31 else 32

    def set_solidity(self, amin=1, amax=2147483647, histo_max=10000):
        self._chk(self.L.gkc_set_solidity(self.h, amin, amax, histo_max)); self.histo_max = histo_max

    def set_max_superkmer(self, maxs):
        self._chk(self.L.gkc_set_max_superkmer(self.h, maxs))

    def set_batch_keys(self, max_keys):
        """upper bound on the k-mers of one Stage-B batch (0: the library's plan)"""
        self._chk(self.L.gkc_set_batch_keys(self.h, max_keys))

    def begin_pass(self, p=0):
        self._chk(self.L.gkc_begin_pass(self.h, p))

    def push_reads(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8); offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._chk(self.L.gkc_push_reads(self.h, _p(bases), _p(offsets), len(offsets) - 1))

    def push_reads_device(self, d_bases, d_offsets, n_reads, n_bases):
        self._chk(self.L.gkc_push_reads_device(self.h, d_bases, d_offsets, n_reads, n_bases))

    def finish_pass(self):
        self._chk(self.L.gkc_finish_pass(self.h))

    def set_host_sink(self, host_buffer):
        """stream every Stage-B batch's records into ``host_buffer`` (a HostBuffer: page-locked) while Stage B runs; None switches it off"""
        self._sink = host_buffer
        self._chk(self.L.gkc_set_host_sink(self.h, None if host_buffer is None else host_buffer._p, 0 if host_buffer is None else host_buffer.nbytes))

    def set_sink_mode(self, mode):
        """"packed" (default: 0.4x of the bytes on the link, expanded by host threads) | "raw" (plain Count[] by DMA, no host core touches a byte)"""
        self._chk(self.L.gkc_set_sink_mode(self.h, {"packed": 0, "raw": 1}[mode]))

    def finish_pass_async(self):
        self._chk(self.L.gkc_finish_pass_async(self.h))

    def finish_pass_wait(self):
        self._chk(self.L.gkc_finish_pass_wait(self.h))

    def wait_partition(self, pass_, part):
        """-> (uint8 view of the partition's Count records inside the host sink or None, n_solid) once they have landed"""
        p = C.c_void_p(); n = C.c_uint64()
        self._chk(self.L.gkc_wait_partition(self.h, pass_, part, C.byref(p), C.byref(n)))
        if not p.value or not n.value:
            return None, n.value
        off = p.value - self._sink._p.value
        return self._sink.a[off: off + n.value * self.rec_bytes], n.value

    def release_pass(self, p):
        """gives the result buffers of a finished (and drained) pass back"""
        self._chk(self.L.gkc_release_pass(self.h, p))

    def device_memory(self):
        """(usable, total) bytes of HBM"""
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.L.gkc_device_memory(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def count(self, bases, offsets):
        """all passes over one host batch"""
        for p in range(self.nb_passes):
            self.begin_pass(p); self.push_reads(bases, offsets); self.finish_pass()

    def partition_info(self, pass_, part):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.gkc_partition_info(self.h, pass_, part, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def partition_records(self, pass_, part, out=None):
        """raw Count records (uint8 view); ``out``: a uint8 array to receive them (e.g. HostBuffer.a: page-locked, DMA at PCIe rate)"""
        ns, _, _ = self.partition_info(pass_, part)
        if out is None:
            out = np.zeros(max(1, ns * self.rec_bytes), np.uint8)
        elif out.nbytes < ns * self.rec_bytes:
            raise GkcError("output buffer too small")
        n = C.c_uint64()
        self._chk(self.L.gkc_partition_counts(self.h, pass_, part, _p(out), ns, C.byref(n)))
        return out[: ns * self.rec_bytes]

    def partition_records_range(self, pass_, part, first, n):
        """records [first, first + n) of the dataset (gkc_partition_counts_range)"""
        out = np.zeros(max(1, n * self.rec_bytes), np.uint8)
        self._chk(self.L.gkc_partition_counts_range(self.h, pass_, part, first, n, _p(out)))
        return out[: n * self.rec_bytes]

    def partition(self, pass_, part):
        """-> (lo uint64[], hi uint64[], abundance int32[])"""
        raw = self.partition_records(pass_, part)
        if self.rec_bytes == 16:
            r = raw.view(np.uint64).reshape(-1, 2)
            return r[:, 0].copy(), np.zeros(len(r), np.uint64), (r[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int32)
        r = raw.view(np.uint64).reshape(-1, 4)
        return r[:, 0].copy(), r[:, 1].copy(), (r[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int32)

    def partition_device(self, pass_, part):
        p = C.c_void_p(); n = C.c_uint64()
        self._chk(self.L.gkc_partition_counts_device(self.h, pass_, part, C.byref(p), C.byref(n)))
        return p.value, n.value

    def all_counts(self):
        out = {}
        for ps in range(self.nb_passes):
            for pt in range(self.nb_partitions):
                lo, hi, ab = self.partition(ps, pt)
                for a, b, c in zip(lo.tolist(), hi.tolist(), ab.tolist()):
                    out[(b << 64) | a] = c
        return out

    def histogram(self):
        hm = getattr(self, "histo_max", 10000)
        h = np.zeros(hm + 1, np.uint64)
        self._chk(self.L.gkc_histogram(self.h, _p(h), hm + 1))
        return h

    def stats(self):
        s = Stats(); self._chk(self.L.gkc_get_stats(self.h, C.byref(s)))
        d = {n: getattr(s, n) for n, _ in Stats._fields_ if n != "reserved"}
        d["pass_nb_sequences"] = int(s.reserved[0]); d["sink_wire_bytes"] = int(s.reserved[1])
        return d

    def timing(self, name):
        ms = C.c_double(); n = C.c_uint64()
        self._chk(self.L.gkc_get_timing(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def partition_superkmers(self, part, cap_bytes=1 << 26):
        out = np.zeros(cap_bytes, np.uint8)
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.L.gkc_partition_superkmers(self.h, part, _p(out), cap_bytes, C.byref(a), C.byref(b), C.byref(c)))
        return out[: a.value].copy(), b.value, c.value

    # ---- Repartitor sampling
    def sample_minimizers(self, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8); offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nsk = np.zeros(4 ** self.m, np.uint64); nk = np.zeros(4 ** self.m, np.uint64)
        self._chk(self.L.gkc_sample_minimizers(self.h, _p(bases), _p(offsets), len(offsets) - 1, _p(nsk), _p(nk)))
        return nsk, nk

    def sample_exact(self, bases, offsets, max_superkmers):
        """-> (superkmers, kmers, kxmers per minimizer, reads used): SampleRepart restated read by read on the device"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8); offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        a = np.zeros(4 ** self.m, np.uint64); b = np.zeros(4 ** self.m, np.uint64); d = np.zeros(4 ** self.m, np.uint64); used = C.c_uint64()
        self._chk(self.L.gkc_sample_exact(self.h, _p(bases), _p(offsets), len(offsets) - 1, max_superkmers, _p(a), _p(b), _p(d), C.byref(used)))
        return a, b, d, used.value

    def count_mmers(self, m, bases, offsets):
        bases = np.ascontiguousarray(bases, dtype=np.uint8); offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        cnt = np.zeros(4 ** m, np.uint32)
        self._chk(self.L.gkc_count_mmers(self.h, m, _p(bases), _p(offsets), len(offsets) - 1, _p(cnt)))
        return cnt

    # ---- segments (multi-GPU exchange surface)
    def segment_count(self):
        n = C.c_uint32(); self._chk(self.L.gkc_segment_count(self.h, C.byref(n))); return n.value

    def segment_export(self, seg):
        p = C.c_void_p(); rb = C.c_uint32()
        off = np.zeros(self.nb_partitions + 1, np.uint64); km = np.zeros(self.nb_partitions, np.uint64)
        self._chk(self.L.gkc_segment_export(self.h, seg, C.byref(p), C.byref(rb), _p(off), _p(km)))
        return p.value, rb.value, off, km

    def segment_import(self, d_records, rec_offsets, kmers):
        ro = np.ascontiguousarray(rec_offsets, dtype=np.uint64); km = np.ascontiguousarray(kmers, dtype=np.uint64)
        self._chk(self.L.gkc_segment_import(self.h, d_records, _p(ro), _p(km)))

    def segments_clear(self):
        self._chk(self.L.gkc_segments_clear(self.h))

    # ---- synthetic input + checksums
    def push_fastx(self, text, final=True):
        """FASTA / FASTQ text (bytes) parsed on the device and pushed; returns the number of bytes consumed"""
        buf = np.frombuffer(bytes(text), dtype=np.uint8)
        cons = C.c_uint64(0)
        self._chk(self.L.gkc_push_fastx(self.h, _p(buf) if len(buf) else None, len(buf), 1 if final else 0, C.byref(cons)))
        return cons.value

    def fastx_parse(self, text, final=True):
        """device parse of FASTA / FASTQ text -> (bases uint8[], offsets uint64[n+1], consumed); copies text in and results out (tests)"""
        import torch
        buf = np.frombuffer(bytes(text), dtype=np.uint8)
        t = torch.from_numpy(buf.copy()).cuda() if len(buf) else torch.empty(0, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        b = C.c_void_p(); o = C.c_void_p(); nr = C.c_uint64(0); nb = C.c_uint64(0); cons = C.c_uint64(0)
        self._chk(self.L.gkc_fastx_parse_device(self.h, t.data_ptr() if len(buf) else None, len(buf), 1 if final else 0,
                                                C.byref(b), C.byref(o), C.byref(nr), C.byref(nb), C.byref(cons)))
        bases = self.device_to_host(b.value, nb.value) if nb.value else np.zeros(0, np.uint8)
        offs = self.device_to_host(o.value, (nr.value + 1) * 8).view(np.uint64)
        self.device_free(b.value); self.device_free(o.value)
        return bases, offs, cons.value

    def synth_reads_device(self, seed, n_reads, read_len, genome_len, sub_ppm, first_read=0, profile=0):
        """profile 1 = GKC_SYNTH_SKEWED: repeat families in the genome + 1 % low-complexity reads (include/gkc.h)"""
        b = C.c_void_p(); o = C.c_void_p()
        self._chk(self.L.gkc_synth_reads_profile_device(self.h, seed, first_read, n_reads, read_len, genome_len, sub_ppm, profile, C.byref(b), C.byref(o)))
        return b.value, o.value

    def device_free(self, p):
        self._chk(self.L.gkc_device_free(self.h, p))

    def device_to_host(self, d_ptr, nbytes):
        out = np.zeros(nbytes, np.uint8)
        self._chk(self.L.gkc_device_to_host(self.h, _p(out), d_ptr, nbytes))
        return out

    def host_to_device(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.gkc_host_to_device(self.h, d_ptr, _p(arr), arr.nbytes))

    def exchange(self, comm):
        """multi-GPU: route the records pushed since the last exchange of this pass to the owners of their partitions (collective)"""
        self._chk(self.L.gkc_exchange(self.h, comm.h))

    def kmer_checksum_device(self, d_bases, d_offsets, n_reads, n_bases):
        a = C.c_uint64(); b = C.c_uint64()
        self._chk(self.L.gkc_kmer_checksum_device(self.h, d_bases, d_offsets, n_reads, n_bases, C.byref(a), C.byref(b)))
        return a.value, b.value

    def result_checksum(self):
        a = C.c_uint64(); b = C.c_uint64()
        self._chk(self.L.gkc_result_checksum(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value


def balanced_owner_ranges(weights, world):
    """contiguous partition ranges per rank balanced by weight (pure host function of the library): first[world+1]"""
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    first = np.zeros(world + 1, np.uint32)
    rc = lib().gkc_balanced_owner_ranges(_p(w), len(w), world, _p(first))
    if rc != 0:
        raise GkcError("gkc_balanced_owner_ranges failed (%d)" % rc)
    return first


def exchange_plan(world, rank, first, n_segs, counts):
    """gkc_exchange_plan (pure host): counts uint64[world][l_max][2][P] -> (sends, recvs, recv_total) with (peer, seg, rec_begin, n_recs) tuples"""
    counts = np.ascontiguousarray(counts, dtype=np.uint64)
    l_max, P = counts.shape[1], counts.shape[3]
    first = np.ascontiguousarray(first, dtype=np.uint32); n_segs = np.ascontiguousarray(n_segs, dtype=np.uint64)
    cap = world * max(1, l_max) + 1
    ps = (PlanMsg * cap)(); pr = (PlanMsg * cap)(); a = C.c_uint32(); b = C.c_uint32(); tot = C.c_uint64()
    rc = lib().gkc_exchange_plan(world, rank, P, _p(first), _p(n_segs), l_max, _p(counts), ps, C.byref(a), pr, C.byref(b), C.byref(tot))
    if rc != 0:
        raise GkcError("gkc_exchange_plan failed (%d)" % rc)
    f = lambda arr, n: [(arr[i].peer, arr[i].seg, arr[i].rec_begin, arr[i].n_recs) for i in range(n)]
    return f(ps, a.value), f(pr, b.value), tot.value


class Comm:
    """gkc_comm: one rank of a multi-GPU run. ``Comm.rccl(counter, id_bytes, world, rank)`` or ``Comm.transport(counter, t, world, rank)``
    where ``t`` offers allgather_host(bytes) -> list of bytes and sendrecv_device(sends, recvs) with (peer, device pointer, nbytes) tuples."""

    def __init__(self, counter, h, keep=None):
        self.c = counter; self.L = counter.L; self.h = h; self._keep = keep

    @staticmethod
    def unique_id():
        buf = np.zeros(128, np.uint8)
        if lib().gkc_comm_unique_id(_p(buf)) != 0:
            raise GkcError("ncclGetUniqueId failed")
        return buf.tobytes()

    @classmethod
    def rccl(cls, counter, id_bytes, world, rank):
        h = C.c_void_p(); buf = np.frombuffer(bytes(id_bytes), dtype=np.uint8).copy()
        counter._chk(counter.L.gkc_comm_create_rccl(counter.h, _p(buf), world, rank, C.byref(h)))
        return cls(counter, h)

    @classmethod
    def transport(cls, counter, t, world, rank):
        def _ag(user, mine, n, out):
            try:
                parts = t.allgather_host(C.string_at(mine, n))
                C.memmove(out, b"".join(parts), n * world)
                return 0
            except Exception as e:      # noqa
                import traceback; traceback.print_exc()
                return 1

        def _sr(user, sends, ns, recvs, nr):
            try:
                t.sendrecv_device([(sends[i].peer, sends[i].d_ptr, sends[i].n_bytes) for i in range(ns)],
                                  [(recvs[i].peer, recvs[i].d_ptr, recvs[i].n_bytes) for i in range(nr)])
                return 0
            except Exception as e:      # noqa
                import traceback; traceback.print_exc()
                return 1

        cb1, cb2 = ALLGATHER_CB(_ag), SENDRECV_CB(_sr)
        st = Transport(None, cb1, cb2)
        h = C.c_void_p()
        counter._chk(counter.L.gkc_comm_create_transport(counter.h, C.byref(st), world, rank, C.byref(h)))
        return cls(counter, h, keep=(cb1, cb2, st, t))

    @classmethod
    def files(cls, counter, directory, world, rank):
        """file-mailbox transport inside the library (gkc_comm_create_files): ranks that share nothing but a directory"""
        h = C.c_void_p()
        counter._chk(counter.L.gkc_comm_create_files(counter.h, str(directory).encode(), world, rank, C.byref(h)))
        return cls(counter, h)

    def enable_ipc(self, on=True):
        """device messages of a transport communicator go device to device through IPC memory handles (gkc_comm_enable_ipc)"""
        self.c._chk(self.L.gkc_comm_enable_ipc(self.h, 1 if on else 0))

    def loopback(self, n_bytes):
        """this rank sends n_bytes to itself through the communicator's send / receive path (chunked like gkc_exchange) -> (mismatching words, ms)"""
        bad = C.c_uint64(0); ms = C.c_double(0)
        self.c._chk(self.L.gkc_comm_loopback(self.c.h, self.h, int(n_bytes), C.byref(bad), C.byref(ms)))
        return bad.value, ms.value

    def selftest(self, n_bytes):
        """collective: every rank sends n_bytes of a keyed pattern to every other rank in one grouped exchange and checks what arrives (gkc_comm_selftest);
        raises on every rank when one of them saw garbage -> (mismatching words on this rank, ms)"""
        bad = C.c_uint64(0); ms = C.c_double(0)
        self.c._chk(self.L.gkc_comm_selftest(self.c.h, self.h, int(n_bytes), C.byref(bad), C.byref(ms)))
        return bad.value, ms.value

    def peer_bytes(self, world):
        """-> (bytes sent to every peer, bytes received from every peer, ms ncclCommInitRank took)"""
        a = np.zeros(world, np.uint64); b = np.zeros(world, np.uint64); ms = C.c_double(0)
        self.c._chk(self.L.gkc_comm_peer_bytes(self.h, _p(a), _p(b), C.byref(ms)))
        return a, b, ms.value

    def gather_results(self, root=0):
        """collective after finish_pass: every partition's Count[], the histogram and the statistics on `root` (gkc_gather_results)"""
        self.c._chk(self.L.gkc_gather_results(self.c.h, self.h, root))

    def set_owners(self, first):
        a = None if first is None else np.ascontiguousarray(first, dtype=np.uint32)
        self.c._chk(self.L.gkc_comm_set_owners(self.h, _p(a)))

    def owners(self, world):
        a = np.zeros(world + 1, np.uint32)
        self.c._chk(self.L.gkc_comm_get_owners(self.h, _p(a))); return a

    def stats(self):
        s = CommStats(); self.c._chk(self.L.gkc_comm_get_stats(self.h, C.byref(s)))
        d = {n: getattr(s, n) for n, _ in CommStats._fields_ if n != "reserved"}
        d["ipc_bounced"] = int(s.reserved[0])      # receive buffers of the IPC transport that could not be exported (mapped ranges) and went through a hipMalloc bounce block
        return d

    def close(self):
        if getattr(self, "h", None):
            self.L.gkc_comm_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mphf:
    """BooPHF minimal perfect hash built on the device (include/gkc.h gkc_mphf_*)"""

    def __init__(self, counter, keys=None, k=None, comm=None):
        """keys=None: the solid k-mers of the counter (comm: of all ranks' counters — collective); else a list of ints / uint64 array
        (k <= 31) or list of ints (k <= 63)"""
        self.c = counter; self.L = counter.L; self.comm = comm
        h = C.c_void_p()
        if keys is None and comm is not None:
            counter._chk(self.L.gkc_mphf_build_solid_dist(counter.h, comm.h, C.byref(h))); self.k = counter.k
        elif keys is None:
            counter._chk(self.L.gkc_mphf_build_solid(counter.h, C.byref(h))); self.k = counter.k
        else:
            self.k = k
            a = self._keys(keys)
            counter._chk(self.L.gkc_mphf_build(counter.h, _p(a), len(a), a.strides[0], k, C.byref(h)))
        self.h = h

    def _keys(self, keys):
        if self.k <= 31:
            return np.ascontiguousarray(np.array([int(x) for x in keys], dtype=np.uint64))
        a = np.zeros((len(keys), 2), dtype=np.uint64)
        for i, x in enumerate(keys):
            a[i, 0] = int(x) & 0xFFFFFFFFFFFFFFFF; a[i, 1] = int(x) >> 64
        return a

    @property
    def size(self):
        return self.L.gkc_mphf_size(self.h)

    def lookup(self, keys):
        a = self._keys(keys); out = np.zeros(len(a), np.uint64)
        self.c._chk(self.L.gkc_mphf_lookup(self.h, _p(a), len(a), a.strides[0], _p(out))); return out

    def save(self):
        n = self.L.gkc_mphf_save_size(self.h); out = np.zeros(n, np.uint8)
        self.c._chk(self.L.gkc_mphf_save(self.h, _p(out), n)); return out

    def abundance_map(self):
        out = np.zeros(self.size, np.uint8); above = C.c_uint64(0)
        if self.comm is not None:
            self.c._chk(self.L.gkc_mphf_abundance_map_dist(self.h, self.c.h, self.comm.h, _p(out), len(out), C.byref(above))); return out, above.value
        self.c._chk(self.L.gkc_mphf_abundance_map(self.h, self.c.h, _p(out), len(out), C.byref(above))); return out, above.value

    def close(self):
        if self.h:
            self.L.gkc_mphf_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Bloom:
    KINDS = {"basic": 0, "cache": 1, "neighbor": 2}

    def __init__(self, counter, kind, tai_bits, nb_hash, k):
        self.c = counter; self.L = counter.L; self.k = k
        h = C.c_void_p()
        counter._chk(self.L.gkc_bloom_create(counter.h, self.KINDS[kind], tai_bits, nb_hash, k, C.byref(h)))
        self.h = h; self.stride = 8 if k <= 31 else 16

    def _keys(self, keys):
        ks = [int(x) for x in keys]
        if self.stride == 8:
            return np.array(ks, dtype=np.uint64)
        a = np.zeros((len(ks), 2), np.uint64)
        a[:, 0] = [x & M64 for x in ks]; a[:, 1] = [x >> 64 for x in ks]
        return a

    def insert(self, keys):
        a = self._keys(keys); self.c._chk(self.L.gkc_bloom_insert(self.h, _p(a), len(a), self.stride))

    def insert_solid(self):
        self.c._chk(self.L.gkc_bloom_insert_solid(self.h, self.c.h))

    def query_solid(self, neighbors8=True, d_out=None):
        """contains8 (or contains) of every solid k-mer of the counter, on the device (gkc_bloom_query_solid) -> (k-mers queried, set result bits)"""
        nq = C.c_uint64(0); npos = C.c_uint64(0)
        self.c._chk(self.L.gkc_bloom_query_solid(self.h, self.c.h, 1 if neighbors8 else 0, d_out, C.byref(nq), C.byref(npos)))
        return nq.value, npos.value

    def contains(self, keys):
        a = self._keys(keys); out = np.zeros(len(a), np.uint8)
        self.c._chk(self.L.gkc_bloom_contains(self.h, _p(a), len(a), self.stride, _p(out))); return out

    def contains8(self, keys):
        a = self._keys(keys); out = np.zeros(len(a), np.uint8)
        self.c._chk(self.L.gkc_bloom_contains8(self.h, _p(a), len(a), self.stride, _p(out))); return out

    @property
    def nbytes(self):
        return self.L.gkc_bloom_nbytes(self.h)

    @property
    def bitsize(self):
        return self.L.gkc_bloom_bitsize(self.h)

    def allreduce_or(self, comm):
        """OR the partial filters of all ranks in place (collective): gkc_bloom_allreduce_or"""
        self.c._chk(self.L.gkc_bloom_allreduce_or(self.h, comm.h))

    def device_array(self):
        """(device pointer, bytes) of the bit array"""
        p = C.c_void_p(); n = C.c_uint64(0)
        self.c._chk(self.L.gkc_bloom_device_array(self.h, C.byref(p), C.byref(n))); return p.value, n.value

    def array(self):
        out = np.zeros(self.nbytes, np.uint8)
        self.c._chk(self.L.gkc_bloom_get_array(self.h, _p(out), len(out))); return out

    def close(self):
        if getattr(self, "h", None):
            self.L.gkc_bloom_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
