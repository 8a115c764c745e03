/* gkc.h — C-ABI of libgkc_hip.so: the MI355X (gfx950) implementation of GATB-Core's DSK k-mer-counting hot path.
 *
 * This is the drop-in boundary. Everything crossing it is plain C: pointers, sizes, integers.
 * No C++ exceptions, STL or torch types cross the ABI. Every function returns 0 on success, non-zero on error
 * (then gkc_last_error() describes it). The host-side C++ classes in gatb-core_amd/host/ (mirrors of
 * gatb::core::kmer::impl::SortingCountAlgorithm<span>, ICountProcessor<span>, IBloom<Item>) and the Python binding
 * gatb-core_amd/gkc.py call nothing else.
 *
 * Reference citations are file:line under /root/reference/gatb-core/src/gatb/.
 *
 * Data conventions (identical to the reference):
 *   nucleotide code  (c>>1)&3 : A=0 C=1 T=2 G=3, anything outside ACGTacgt is invalid   (tools/misc/api/Data.hpp:185)
 *   k-mer integer    first nucleotide in the most significant used bits                  (kmer/impl/Model.hpp:637-657)
 *   canonical        min(forward, reverse complement)                                    (kmer/impl/Model.hpp:294)
 *   Count record     k<=31: 16 B {u64 value; i32 abundance; 4 B pad}
 *                    k<=63: 32 B {u128 value (little endian); i32 abundance; 12 B pad}   (tools/misc/api/Abundance.hpp:68-129)
 */
#ifndef GKC_H
#define GKC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is compiled with -fvisibility=hidden and linked with a version script (csrc/gkc.map): the declarations below are its whole dynamic symbol table. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GKC_OK            0
#define GKC_ERR_ARG       1   /* bad argument / bad state                        */
#define GKC_ERR_HIP       2   /* a HIP runtime call failed (message has details) */
#define GKC_ERR_NOMEM     3   /* device or host allocation failed                */
#define GKC_ERR_CAPACITY  4   /* caller buffer too small                         */
#define GKC_ERR_NODEVICE  5   /* no usable gfx950 device                         */
#define GKC_ERR_FORMAT    6   /* input text the device parser refuses (see gkc_fastx_parse_device) */

#define GKC_MINIMIZER_LEXI 0  /* -minimizer-type 0: lexicographic + KMC2 "no inner AA" rule (Model.hpp:1220-1251) */
#define GKC_MINIMIZER_FREQ 1  /* -minimizer-type 1: (freq_order[c], c) order (Model.hpp:957-973)                  */

typedef struct gkc_ctx gkc_ctx;

/* ---------------------------------------------------------------------------------------------------------------
 * Context. Replaces the per-run state of SortingCountAlgorithm<span> (kmer/impl/SortingCountAlgorithm.hpp:65-263):
 * owns the HIP stream, every device buffer and the per-pass results.
 * ------------------------------------------------------------------------------------------------------------- */
int         gkc_create(int device, gkc_ctx** out);
/* Objects built from a context (gkc_bloom, gkc_mphf, gkc_comm) hold device memory of its allocator and keep it alive: gkc_destroy may be
 * called first, the context is then released when the last of them is destroyed. */
void        gkc_destroy(gkc_ctx* ctx);
const char* gkc_last_error(const gkc_ctx* ctx);      /* ctx may be NULL: last error of a failed gkc_create */
const char* gkc_version(void);

/* Model + Repartitor. Replaces `Model model(k, m, cmp, freq_order)` (SortingCountAlgorithm.cpp:1251-1256, LUT built at
 * Model.hpp:1032-1064) and `Repartitor::operator()` (kmer/impl/PartiInfo.hpp:323).
 *   repart      u16[4^m]  minimizer value -> partition  (the table Repartitor::load reads, PartiInfo.cpp:223-262)
 *   freq_order  u32[4^m]  or NULL; required iff minimizer_type==GKC_MINIMIZER_FREQ (RepartitionAlgorithm.cpp:311-384)
 * k in [3,63] (k<=2 refused like SortingCountAlgorithm.cpp:662-666), m in [2,min(k-1,14)], nb_partitions in [1,65535].
 * Sizing: Stage B is planned for partitions of 2e6 .. 4e6 k-mers (what gkc_device_memory-driven callers and the drop-in's DeviceConfiguration choose: ~4000
 * partitions for 1e8 reads of 150 bp). Any count is correct; partitions beyond 8e6 k-mers are expanded by several workgroups each but skip the record
 * deduplication and send their sub-buckets through one split level (256 partitions for 1e8 reads: 375 ms per pass against 203 ms with 4096). */
int gkc_configure(gkc_ctx* ctx, uint32_t k, uint32_t m, uint32_t nb_partitions, uint32_t nb_passes,
                  int minimizer_type, const uint16_t* repart, const uint32_t* freq_order);

/* Solidity window and histogram length. Replaces CountProcessorSoliditySum::check (CountProcessorSolidity.hpp:186-189,
 * closed interval on the sum) and Histogram::inc clamping (tools/misc/impl/Histogram.hpp:92). Defaults: [1, INT32_MAX],
 * histo_max 10000, i.e. every distinct k-mer is returned and the host's own processor chain can do the filtering. */
int gkc_set_solidity(gkc_ctx* ctx, int32_t abundance_min, int32_t abundance_max, uint32_t histo_max);

/* Super-k-mer length cap (Sequence2SuperKmer.hpp:147). 0 = reference default 28 (k<=31) / 60 (k<=63). */
int gkc_set_max_superkmer(gkc_ctx* ctx, uint32_t maxs);

/* Upper bound on the k-mers of one Stage-B batch (0 = the library's plan). Stage B counts a pass in batches of whole partitions; their size sets the working set
 * in HBM (about 22 bytes per k-mer of a batch, two batches in flight) and how often the chip drains between batches. The library's own plan is made for a host
 * that counts pass after pass in one process: few, large batches (3.2e9 k-mers with 8-byte keys: 10^8 reads = 4 batches, 130 GB of working set, every block reused
 * by the next pass). A host that counts ONE pass per process — dbgh5 — asks for less: 2^30 k-mers per batch cost 8 % of Stage B (157 vs 145 ms at 10^8 reads) and
 * take 46 GB, and a machine whose memory is handed out slowly the first time (28 ms per GB beyond ~120 GB on a freshly booted MI355X box: profiles/
 * r05_cold_pass.txt) does not charge the difference to the one pass there is. The reference's counterpart is the memory the partitions of a pass are sized for
 * (-max-memory, ConfigurationAlgorithm.cpp:398-425). Results do not depend on it. */
int gkc_set_batch_keys(gkc_ctx* ctx, uint64_t max_keys);

/* ---------------------------------------------------------------------------------------------------------------
 * Repartitor sampling — device side of RepartitorAlgorithm (kmer/impl/RepartitionAlgorithm.cpp:287-492). The tables
 * themselves (computeDistrib / justGroup / justGroupLexi, kmer/impl/PartiInfo.cpp:48-218, and the frequency ranking,
 * RepartitionAlgorithm.cpp:352-380) are built by the host from these statistics, like the reference does.
 *   gkc_sample_minimizers : per minimizer value, number of super-k-mers and of k-mers of the sample under the CURRENT model
 *                           (PartiInfo::incSuperKmer_per_minimBin); arrays of 4^m u64, ACCUMULATED into.
 *   gkc_count_mmers       : occurrences of every canonical m-mer at valid positions (MmersFrequency functor, :88-120);
 *                           u32[4^m], ACCUMULATED into.
 * ------------------------------------------------------------------------------------------------------------- */
int gkc_sample_minimizers(gkc_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_reads,
                          uint64_t* superkmers_per_minim, uint64_t* kmers_per_minim);
int gkc_count_mmers(gkc_ctx* ctx, uint32_t m, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint32_t* counts);
/* The EXACT sample of RepartitorAlgorithm::computeRepartition (RepartitionAlgorithm.cpp:395-475): the reads are walked one by one like
 * Sequence2SuperKmer does (no tiles), the sample stops with the read in which the running number of super-k-mers of pass 0 (of the context's
 * nb_passes: the reference's SampleRepart is a ONE-pass Sequence2SuperKmer, :225, so its callers configure 1 pass for the sample) first exceeds
 * max_superkmers (SampleRepart::processSuperkmer :205-212 sets the cancel flag, checked between sequences), and per minimizer value the
 * super-k-mers, k-mers AND kx-mers are counted (:186-203, _kx = 4) — the last being what Repartitor::computeDistrib balances on
 * (PartiInfo.cpp:48-106). Arrays of 4^m u64 (any may be NULL), ACCUMULATED into; *reads_used = reads of the sample. */
int gkc_sample_exact(gkc_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_reads, uint64_t max_superkmers,
                     uint64_t* superkmers_per_minim, uint64_t* kmers_per_minim, uint64_t* kxmers_per_minim, uint64_t* reads_used);

/* ---------------------------------------------------------------------------------------------------------------
 * Stage A — replaces SortingCountAlgorithm::fillPartitions (SortingCountAlgorithm.cpp:1211-1344): Sequence2SuperKmer
 * (Sequence2SuperKmer.hpp:81-159) + FillPartitions::processSuperkmer (:1081-1151) + the SuperKmerBinFiles disk shuffle
 * (tools/storage/impl/Storage.cpp:360-430). Reads arrive as a flat ASCII buffer + CSR offsets (offsets[n_reads] ==
 * number of bases); any number of pushes per pass. Super-k-mers whose minimizer % nb_passes != pass are dropped (:1083).
 * ------------------------------------------------------------------------------------------------------------- */
int gkc_begin_pass(gkc_ctx* ctx, uint32_t pass);
int gkc_push_reads(gkc_ctx* ctx, const char* bases, const uint64_t* offsets, uint64_t n_reads);
/* same, inputs already resident in HBM (d_bases 16-byte aligned). The buffers may be released after the call returns.
 * A push is one segment of super-k-mer records with its own per-push buffers (about 0.8 bytes of descriptors per base besides the records): pushes of up to ~10^8 reads
 * (1.5*10^10 bases) are what those are sized for; a larger input is pushed in several calls (any number per pass). */
int gkc_push_reads_device(gkc_ctx* ctx, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads, uint64_t n_bases);

/* Stage B — replaces SortingCountAlgorithm::fillSolidKmers (:1384-1602) = PartitionsByVectorCommand read/sort/dump
 * (kmer/impl/PartitionsCommand.cpp:1206-1805) for every partition of the pass, plus the default processor chain
 * histogram -> solidity -> dump (CountProcessorChain.hpp:128-135). Results stay in HBM until fetched. */
int gkc_finish_pass(gkc_ctx* ctx);

/* Streamed results (the wall ends when the last partition's Count[] is in host memory). With a host sink set, every Stage-B
 * batch is copied into it on a copy stream as soon as it is compacted, while the next batches are counted: the reference's sink it stands
 * in for is CountProcessorDump -> BagCache -> CollectionHDF5Patch (CountProcessorDump.hpp:148-152, CollectionHDF5Patch.hpp:262-308), fed per
 * partition by the dump threads. The sink must be page-locked (gkc_host_alloc) and holds ONE pass (gkc_begin_pass rewinds it); what does
 * not fit stays on the device and gkc_finish_pass reports GKC_ERR_CAPACITY in gkc_last_error while still returning 0 records lost.
 * On the link the batches travel PACKED (base key per block of 8192 records + key deltas bit-packed at the width of their sub-block of 128 records + a bitmap and a
 * stream of the abundances that are not 1: 5.8 of the 16 bytes of a record at k = 31, 14.5 of the 32 at k = 63, 10^8 reads; 7 - 8 / 16 - 17 with fixed-width entries
 * where the partitions are sparse) and are expanded into the exact Count[] layout inside the sink by threads of the library; what gkc_wait_partition hands out
 * is byte for byte what gkc_partition_counts gives.
 *   gkc_finish_pass_async : Stage B on a worker thread of the library; returns at once.
 *   gkc_wait_partition    : blocks until dataset (pass, part) is counted and, with a sink, has landed; *host_records points into the sink
 *                           (NULL without a sink / for what did not fit). Partitions complete in ascending order inside a batch, batches in
 *                           ascending order per lane: a consumer walking the partitions in order overlaps its work with Stage B.
 *   gkc_finish_pass_wait  : joins the worker; the pass is finished like after gkc_finish_pass. If Stage B FAILED on the worker (e.g. GKC_ERR_NOMEM) and the
 *                           context has not begun another pass meanwhile, the pass is in progress again — as after a failed gkc_finish_pass — and
 *                           gkc_finish_pass / gkc_finish_pass_async may be called again (the retry of gkc_count_pass). */
int gkc_set_host_sink(gkc_ctx* ctx, void* pinned_host, uint64_t cap_bytes);     /* NULL: no sink */
/* How the batches cross the link, chosen before gkc_set_host_sink (default GKC_SINK_PACKED):
 *   GKC_SINK_PACKED  as described above: 0.4x of the bytes on the link, and per record 6.3 B read + 16 B of non-temporal stores by CPU threads of this host;
 *   GKC_SINK_RAW     the Count[] of every batch lands at its place in the sink by ONE DMA copy: 16 / 32 B per record on the link, NO host core touches a byte.
 * One rank per host: packed (the link is the bound). Several ranks sharing ONE host's DRAM (8 GPUs of a node, each with its own PCIe link): the expansion threads of all
 * ranks meet at the host's memory controllers (104 B of host traffic per record packed against 16 raw) — the launcher measures one step each way and keeps the faster
 * (bench.py does; DESIGN.md section 5 has the numbers). Both modes leave byte for byte the same sink. */
#define GKC_SINK_PACKED 0
#define GKC_SINK_RAW    1
int gkc_set_sink_mode(gkc_ctx* ctx, int mode);
int gkc_finish_pass_async(gkc_ctx* ctx);
int gkc_wait_partition(gkc_ctx* ctx, uint32_t pass, uint32_t part, const void** host_records, uint64_t* n_solid);
int gkc_finish_pass_wait(gkc_ctx* ctx);

/* Result of dataset (part + pass*nb_partitions) (CountProcessorDump.hpp:131): ascending Count records of the SOLID
 * k-mers, in the exact in-memory layout of Kmer<span>::Count so the host can pass whole arrays to
 * Bag<Count>::insert(const Item*, len) (tools/storage/impl/CollectionHDF5Patch.hpp:262). */
int gkc_partition_info(gkc_ctx* ctx, uint32_t pass, uint32_t part,
                       uint64_t* n_solid, uint64_t* n_distinct, uint64_t* n_kmers);
int gkc_partition_counts(gkc_ctx* ctx, uint32_t pass, uint32_t part, void* out_counts, uint64_t cap_records,
                         uint64_t* n_solid);
/* records [first, first + n) of the dataset into `out` — for a consumer that moves a partition through page-locked buffers smaller than the partition
 * (integration/gatb_device/DeviceCounting.hpp: a ring of 4 MiB slots between the link and the .h5 file). Same stream and ordering as gkc_partition_counts. */
int gkc_partition_counts_range(gkc_ctx* ctx, uint32_t pass, uint32_t part, uint64_t first, uint64_t n, void* out_counts);
/* device pointer to the same records (valid until gkc_begin_pass of the same pass index or gkc_destroy) */
int gkc_partition_counts_device(gkc_ctx* ctx, uint32_t pass, uint32_t part, const void** d_counts, uint64_t* n_solid);

/* Histogram of abundances over ALL distinct k-mers (CountProcessorHistogram.hpp:173-184): histo_max+1 bins. */
int gkc_histogram(gkc_ctx* ctx, uint64_t* out, uint32_t n_bins);

/* Counters the reference reports through getInfo() (SortingCountAlgorithm.cpp:728-780, Sequence2SuperKmer.hpp:103,108) */
typedef struct gkc_stats {
    uint64_t kmers_nb_valid;      /* pass 0 only, like the reference                      */
    uint64_t kmers_nb_invalid;
    uint64_t kmers_nb_distinct;   /* summed over finished passes                          */
    uint64_t kmers_nb_solid;
    uint64_t nb_superkmers;       /* device records (tile boundaries may add a few splits) */
    uint64_t nb_sequences;        /* pass 0 only                                          */
    uint64_t nb_bases;            /* pass 0 only                                          */
    uint64_t superkmer_bytes;     /* bytes of the device record buckets                   */
    uint64_t oversize_buckets;    /* sub-buckets that took the global-memory sort path    */
    uint64_t dedupe_kmers_in;     /* k-mers of the record bins that were deduplicated before the expansion (Stage B, k <= 31) ... */
    uint64_t dedupe_keys_out;     /* ... and the weighted keys they became (0 / 0: the step did not run)                          */
    uint64_t seq_len_min;         /* pass 0 only: shortest / longest read and the sum of the squared read lengths — what BankStats::update keeps   */
    uint64_t seq_len_max;         /* (BankKmers.hpp:164-200: seq_size_min / max / deviation of getInfo(), SortingCountAlgorithm.cpp:735-739);       */
    uint64_t seq_len_sq_sum;      /* seq_len_min is 0 when no read was pushed                                                                       */
    uint64_t reserved[2];         /* [0] reads pushed in the CURRENT pass (nb_sequences is pass 0's); [1] bytes the packed result batches of the last counted pass took on
                                     the link to the host sink (gkc_set_host_sink; 0: nothing travelled packed)                                                       */
} gkc_stats;
int gkc_get_stats(gkc_ctx* ctx, gkc_stats* out);

/* Kernel timing of the last gkc_finish_pass / pushes (HIP events on the context's stream), milliseconds.
 * names: "scan_count", "scan_emit", "expand_count", "expand_scatter", "bucket_sort", "compact", "total_stage_a", "total_stage_b" */
int gkc_get_timing(gkc_ctx* ctx, const char* name, double* ms, uint64_t* launches);

/* ---------------------------------------------------------------------------------------------------------------
 * Parity / exchange surface for the super-k-mer buckets (the device analogue of SuperKmerBinFiles).
 * ------------------------------------------------------------------------------------------------------------- */
/* Partition `part` of the current pass re-encoded in the REFERENCE wire format: concatenated
 * [u8 nbK][ceil((k+nbK-1)/4) bytes] records (Model.hpp:1386-1471, Storage.cpp:567-580). Order of records is unspecified. */
int gkc_partition_superkmers(gkc_ctx* ctx, uint32_t part, uint8_t* out, uint64_t cap_bytes,
                             uint64_t* n_bytes, uint64_t* n_superkmers, uint64_t* n_kmers);

/* Segment surface (parity tests, custom exchanges): records of partition p occupy rec_index in [rec_offsets[p], rec_offsets[p+1]) of the
 * segment's arena; record_bytes is 16 (k<=31) or 32.
 * gkc_segment_export: segment `seg` (one per push) of the current pass. d_records stays owned by the context.
 * gkc_segment_import: adds a foreign segment; the memory stays owned by the caller and must outlive gkc_finish_pass.
 * gkc_segments_clear: forget all segments of the current pass (frees owned arenas). The multi-GPU exchange itself is gkc_exchange below. */
int gkc_segment_count(gkc_ctx* ctx, uint32_t* n_segments);
int gkc_segment_export(gkc_ctx* ctx, uint32_t seg, const void** d_records, uint32_t* record_bytes,
                       uint64_t* rec_offsets /* [nb_partitions+1] */, uint64_t* kmers_per_partition /* [nb_partitions] */);
int gkc_segment_import(gkc_ctx* ctx, const void* d_records, const uint64_t* rec_offsets /* [nb_partitions+1] */,
                       const uint64_t* kmers_per_partition /* [nb_partitions] */);
int gkc_segments_clear(gkc_ctx* ctx);

/* ---------------------------------------------------------------------------------------------------------------
 * Multi-GPU: one process = one context = one GPU; partitions are OWNED by ranks (contiguous ranges), every rank scans
 * its own reads, and one exchange routes each rank's super-k-mer records to the owner of their partition — the device analogue of the
 * reference's disk shuffle (SuperKmerBinFiles, tools/storage/impl/Storage.cpp:360-430; fillPartitions writes, fillSolidKmers reads:
 * SortingCountAlgorithm.cpp:1211-1344, 1384-1602). Same canonical k-mer => same minimizer => same partition, so after the exchange every
 * rank counts the partitions it owns with no further communication.
 *
 * A communicator is either RCCL (grouped ncclSend / ncclRecv over xGMI, librccl linked directly, asynchronous on its own HIP stream so
 * that the exchange of push i overlaps Stage A of push i+1) or a caller-supplied transport (two callbacks; used where RCCL cannot run,
 * e.g. two ranks sharing one GPU in the tests: gloo with host staging, gatb-core_amd/dist.py).
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct gkc_comm gkc_comm;
#define GKC_COMM_ID_BYTES 128                      /* sizeof(ncclUniqueId) */
typedef struct gkc_xfer { int32_t peer; int32_t reserved; void* d_ptr; uint64_t n_bytes; } gkc_xfer;   /* d_ptr: DEVICE memory */
typedef struct gkc_transport {
    void* user;
    /* every rank contributes n_bytes of HOST memory; all[r * n_bytes ...] = rank r's contribution. Blocking. */
    int (*allgather_host)(void* user, const void* mine, uint64_t n_bytes, void* all);
    /* one grouped point-to-point exchange (ncclGroupStart ... ncclGroupEnd semantics): every send and receive of the call is posted
     * before any is waited for; messages between one pair of ranks match in list order; returns when all have completed. */
    int (*sendrecv_device)(void* user, const gkc_xfer* sends, uint32_t n_sends, const gkc_xfer* recvs, uint32_t n_recvs);
} gkc_transport;
/* rank 0 makes the id (ncclGetUniqueId) and the launcher hands it to every rank (torch.distributed store, MPI, a file) */
int  gkc_comm_unique_id(uint8_t id[GKC_COMM_ID_BYTES]);
int  gkc_comm_create_rccl(gkc_ctx* ctx, const uint8_t id[GKC_COMM_ID_BYTES], int world, int rank, gkc_comm** out);
int  gkc_comm_create_transport(gkc_ctx* ctx, const gkc_transport* t, int world, int rank, gkc_comm** out);
/* A transport that needs nothing but a directory all ranks can see (a mailbox of files: host all-gather = one file per rank, device send/recv = staged
 * through host memory, one file per message). For ranks RCCL cannot connect — two processes sharing one GPU in the tests (RCCL refuses a duplicate device),
 * hosts without xGMI — and for launchers that have nothing but a shared file system. Development / test transport: bandwidth is that of the file system. */
int  gkc_comm_create_files(gkc_ctx* ctx, const char* directory, int world, int rank, gkc_comm** out);
/* A transport communicator (gkc_comm_create_transport / _files) whose DEVICE messages stay on the devices: every receive buffer is published as an IPC memory handle
 * through the transport's host all-gather and the sender copies device to device (xGMI peer copy between the GPUs of a node). What a launcher falls back to when RCCL
 * refuses the communicator: the host side of the transport is then only used for a few hundred bytes per exchange. Call before the first push on every rank. */
int  gkc_comm_enable_ipc(gkc_comm* comm, int on);
void gkc_comm_destroy(gkc_comm* comm);
/* Owner ranges: rank r owns partitions [first[r], first[r+1]); first[0] = 0, first[world] = nb_partitions. By default the first
 * exchange of a pass balances them by the k-mers per partition all ranks report ("balanced by weight"); gkc_comm_set_owners
 * pins them (e.g. from the Repartitor's sample, PartiInfo.cpp:48-106), NULL goes back to balancing. gkc_balanced_owner_ranges is the
 * pure host function behind it (no context, no GPU). */
int  gkc_comm_set_owners(gkc_comm* comm, const uint32_t* first /* [world+1] or NULL */);
int  gkc_comm_get_owners(gkc_comm* comm, uint32_t* first /* [world+1] */);
int  gkc_balanced_owner_ranges(const uint64_t* weights, uint32_t nb_partitions, int world, uint32_t* first /* [world+1] */);
/* Routes the records of every push since the last gkc_exchange of this pass to their owners and imports what arrives; the rank's own
 * segments keep only the partitions it owns. Collective: every rank calls it the same number of times per pass (a rank without new
 * pushes takes part with nothing to send — pushes per rank may differ). With RCCL the transfers run on the communicator's stream and
 * gkc_finish_pass waits for them. */
int  gkc_exchange(gkc_ctx* ctx, gkc_comm* comm);
/* The message plan of one exchange for one rank — the pure host function gkc_exchange runs after the all-gather of the count tables (no
 * context, no GPU: the CPU tests drive it with gloo). counts: [world][l_max][2][P] = records, k-mers per partition of segment j of rank r
 * (zero rows beyond n_segs[r]); first: owner ranges. sends[i]: n_recs records starting at record rec_begin of this rank's own segment
 * `seg` go to `peer`; recvs[i]: n_recs records of segment `seg` of `peer` arrive at record rec_begin of the receive arena. Messages between
 * two ranks are listed in the same order on both sides. Capacity of sends / recvs: world * l_max entries each. */
typedef struct gkc_plan_msg { int32_t peer; uint32_t seg; uint64_t rec_begin; uint64_t n_recs; } gkc_plan_msg;
int  gkc_exchange_plan(int world, int rank, uint32_t nb_partitions, const uint32_t* first, const uint64_t* n_segs, uint64_t l_max, const uint64_t* counts,
                       gkc_plan_msg* sends, uint32_t* n_sends, gkc_plan_msg* recvs, uint32_t* n_recvs, uint64_t* recv_total_recs);
typedef struct gkc_comm_stats {
    uint64_t n_exchanges, bytes_sent, bytes_received;   /* record bytes that left / reached this rank (not counting what stayed) */
    double   ms_transfer;                                 /* HIP-event time of the grouped send/recv on the communicator's stream (RCCL) or wall time of the callback */
    double   ms_host;                                     /* wall time inside gkc_exchange (tables, allocation, enqueue) */
    uint64_t reserved[4];                                 /* [0]: receive buffers the IPC transport took through a bounce block (their allocation could not be exported) */
} gkc_comm_stats;
int  gkc_comm_get_stats(gkc_comm* comm, gkc_comm_stats* out);
/* Collective, after gkc_finish_pass of the current pass on every rank: the Count[] arrays of all partitions are gathered on `root` (each owner sends the arrays of
 * its partitions, whole Stage-B batches at a time), and so are the abundance histogram and the pass statistics (summed). Afterwards gkc_wait_partition /
 * gkc_partition_counts* / gkc_histogram / gkc_get_stats on `root` serve EVERY partition of the pass — what a single process would hold — so that one process
 * writes the one result file the reference's consumers open (CountProcessorDump.hpp:85-95: all nb_partitions x nb_passes datasets live in one file;
 * GraphUnitigs.cpp:921-931 opens that file). The other ranks keep what they had. The results must fit root's HBM (solid k-mers only travel). */
int  gkc_gather_results(gkc_ctx* ctx, gkc_comm* comm, int root);
/* Diagnostic: this rank sends n_bytes of a pattern to ITSELF through the communicator's own grouped send / receive path (the message is cut into the same
 * 256 MiB chunks as the messages of gkc_exchange) and compares what arrived. With one rank this is the only way to run ncclSend / ncclRecv and the chunking on
 * hardware. mismatches: 8-byte words that differ; ms: wall time of the transfer. */
int  gkc_comm_loopback(gkc_ctx* ctx, gkc_comm* comm, uint64_t n_bytes, uint64_t* mismatches, double* ms);
/* Start-up self-test over the real peers, collective: in ONE grouped exchange every rank sends n_bytes of a pattern keyed by (source, destination) to every other
 * rank and checks what arrives from each — the send / receive path of gkc_exchange (RCCL: grouped ncclSend / ncclRecv over xGMI, cut into the same 256 MiB chunks)
 * between every pair of GPUs before any record travels. A rank that receives anything but the pattern makes the call fail on EVERY rank (gkc_comm_agree).
 * mismatches: 8-byte words that differ on this rank; ms: wall time of the exchange on this rank. One rank: the loopback above. */
int  gkc_comm_selftest(gkc_ctx* ctx, gkc_comm* comm, uint64_t n_bytes, uint64_t* mismatches, double* ms);
/* What went where: bytes this rank has sent to / received from every peer through the communicator's grouped send / receive path since it was created ([world] each,
 * either may be NULL), and the wall time ncclCommInitRank took when the communicator was made (0 for the other transports). */
int  gkc_comm_peer_bytes(gkc_comm* comm, uint64_t* sent /* [world] */, uint64_t* received /* [world] */, double* init_ms);

/* ---------------------------------------------------------------------------------------------------------------
 * Bloom filter of solid k-mers — replaces BloomBuilder::build / IBloom::insert (kmer/impl/BloomBuilder.hpp:102-128,
 * 174-182) and the query side IBloom::contains / contains8 (tools/collections/impl/Bloom.hpp:211-234, 437-490,
 * 555-828). kind: 0 "basic" (BloomSynchronized), 1 "cache" (BloomCacheCoherent), 2 "neighbor" (BloomNeighborCoherent) —
 * the three kinds BloomFactory::createBloom can return for a k-mer item (Bloom.hpp:1254-1266).
 * The bit array is byte-identical to the reference's getArray() for the same inserted set.
 * ------------------------------------------------------------------------------------------------------------- */
typedef struct gkc_bloom gkc_bloom;
int      gkc_bloom_create(gkc_ctx* ctx, int kind, uint64_t tai_bits, uint32_t nb_hash, uint32_t k, gkc_bloom** out);
void     gkc_bloom_destroy(gkc_bloom* b);
uint64_t gkc_bloom_nbytes(const gkc_bloom* b);     /* IBloom::getSize()    */
uint64_t gkc_bloom_bitsize(const gkc_bloom* b);    /* IBloom::getBitSize() */
/* keys: n items, `stride` bytes apart, each starting with the k-mer value (8 B for k<=31, 16 B for k<=63): pass Count
 * arrays with stride 16/32 or raw key arrays with stride 8/16. *_device variants take HBM-resident keys. */
int gkc_bloom_insert(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride);
int gkc_bloom_insert_device(gkc_bloom* b, const void* d_keys, uint64_t n, uint32_t stride);
/* inserts every solid k-mer of every finished dataset of the context (what BloomAlgorithm::execute does, BloomAlgorithm.cpp:155-199) */
int gkc_bloom_insert_solid(gkc_bloom* b, gkc_ctx* ctx);
int gkc_bloom_contains(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride, uint8_t* out);
int gkc_bloom_contains8(gkc_bloom* b, const void* keys, uint64_t n, uint32_t stride, uint8_t* out);  /* neighbor kind */
/* The query side at the reference's call site (DebloomMinimizerAlgorithm.cpp:201: contains8 of every solid k-mer; Bloom.hpp:645-811), on the device: every solid
 * k-mer of every finished dataset of the context is queried where it lies (HBM), in dataset order. neighbors8 != 0: contains8 (neighbor kind), one result byte per
 * k-mer (bit j: neighbour j is in the filter); else contains, 0/1 per k-mer. d_out: device buffer of *n_queried bytes or NULL (results discarded).
 * n_positive: number of set result bits. */
int gkc_bloom_query_solid(gkc_bloom* b, gkc_ctx* ctx, int neighbors8, uint8_t* d_out, uint64_t* n_queried, uint64_t* n_positive);
int gkc_bloom_get_array(gkc_bloom* b, uint8_t* out, uint64_t cap_bytes);                               /* IBloom::getArray() */
int gkc_bloom_set_array(gkc_bloom* b, const uint8_t* in, uint64_t n_bytes);                            /* StorageTools::loadBloom */
/* the bit array where it lives (device memory, n_bytes = gkc_bloom_nbytes rounded up to 4): multi-GPU runs insert their own
 * partitions' solid k-mers and OR-reduce the arrays in place (all-reduce with bitwise OR over RCCL) */
int gkc_bloom_device_array(gkc_bloom* b, void** d_bits, uint64_t* n_bytes);
/* in-place bitwise-OR all-reduce of the bit array over the communicator: reduce-scatter (every rank ORs one slice of everybody's array)
 * + all-gather, 2 (world-1)/world of the array per rank instead of (world-1) arrays */
int gkc_bloom_allreduce_or(gkc_bloom* b, gkc_comm* comm);

/* ---------------------------------------------------------------------------------------------------------------
 * Synthetic reads generated in HBM (bench / parity at full size; SURVEY §8d generator): genome of genome_len uniform
 * bases from a counter-based hash of (seed, position); read i (global index first_read+i, so ranks can draw disjoint
 * slices of one read stream over the same genome) starts at hash(seed,index) % (genome_len-read_len+1), is
 * reverse-complemented with probability 1/2 and each base is substituted with probability sub_rate_ppm/1e6.
 * Bit-identical to gatb-core_amd/gkc.py:synth_reads_np (numpy). Outputs are device pointers owned by the caller of
 * gkc_device_free.
 * ------------------------------------------------------------------------------------------------------------- */
int gkc_synth_reads_device(gkc_ctx* ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                           uint64_t genome_len, uint32_t sub_rate_ppm, char** d_bases, uint64_t** d_offsets);
/* The same generator over a REPEAT-RICH genome with low-complexity reads (full-size parity and bench of the paths the reference answers skew with:
 * PartitionsCommand.cpp:505-545, TempCountFileMerger :217-360). GKC_SYNTH_SKEWED: the genome is cut into slots of 8192 bases; a quarter of them start with a copy
 * of one of 50 family sequences of 1000..5000 bases (about genome_len / 1.6e6 copies per family, ~9 % of the bases; every copy 0.5 % diverged), the rest is
 * uniform; 1 % of the reads are a unit of 1..3 bases repeated (poly-A, (AC)n, (ACG)n ...) before the substitutions. Twin: gkc.py synth_reads_np(profile=1). */
#define GKC_SYNTH_UNIFORM 0u
#define GKC_SYNTH_SKEWED  1u
int gkc_synth_reads_profile_device(gkc_ctx* ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads, uint32_t read_len,
                                   uint64_t genome_len, uint32_t sub_rate_ppm, uint32_t profile, char** d_bases, uint64_t** d_offsets);
/* ---- MPHF + abundance map ------------------------------------------------------------------------------
 * Replaces MPHFAlgorithm::execute / populate (kmer/impl/MPHFAlgorithm.cpp:150-275) and the BooPHF build behind it
 * (thirdparty/BooPHF/BooPHF.h:734-1108 as instantiated by tools/collections/impl/BooPHF.hpp:236-300: jenkins64 hasher seeded by
 * std::mt19937_64(37), gamma 3, 25 levels). The level bit arrays and rank samples are the ones BooPHF builds for the same keys, so
 * gkc_mphf_save writes the byte stream of boomphf::mphf::save (what MPHF<>::save puts into the "dsk/mphf" collection, loadable by
 * BooPHF::load) and gkc_mphf_lookup returns BooPHF::lookup's codes. Keys: `stride` bytes per item, the k-mer in the first 8 (k <= 31)
 * or 16 (k <= 63) bytes — a Count array works as it is. */
typedef struct gkc_mphf gkc_mphf;
int      gkc_mphf_build(gkc_ctx* ctx, const void* keys /* host */, uint64_t n, uint32_t stride, uint32_t k, gkc_mphf** out);
int      gkc_mphf_build_solid(gkc_ctx* ctx, gkc_mphf** out);          /* keys = the solid k-mers of every finished dataset (getSolidKmers() order) */
/* Multi-GPU build over the solid k-mers of ALL ranks (each rank holds the partitions it owns): BooPHF's level arrays do not depend on the
 * order keys are processed in, so every level is built from per-rank partial arrays combined with seen = OR, collided = OR | (seen by two
 * ranks) — reduce-scatter + all-gather like the Bloom. Every rank ends up with the complete function: gkc_mphf_save / lookup give the
 * bytes / codes of a single-GPU build over the same key set (rank order = partition order = getSolidKmers() order). Collective. */
int      gkc_mphf_build_solid_dist(gkc_ctx* ctx, gkc_comm* comm, gkc_mphf** out);
/* populate() across ranks: every rank fills the cells of its own solid k-mers, the maps are OR-combined; out (host, gkc_mphf_size bytes) is
 * complete on every rank */
int      gkc_mphf_abundance_map_dist(gkc_mphf* m, gkc_ctx* ctx, gkc_comm* comm, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision);
void     gkc_mphf_destroy(gkc_mphf* m);
uint64_t gkc_mphf_size(const gkc_mphf* m);                             /* number of keys (BooPHF::size) */
int      gkc_mphf_lookup(gkc_mphf* m, const void* keys /* host */, uint64_t n, uint32_t stride, uint64_t* codes);   /* ~0: not a key (final level miss) */
uint64_t gkc_mphf_save_size(const gkc_mphf* m);
int      gkc_mphf_save(gkc_mphf* m, uint8_t* out, uint64_t cap);
/* MPHFAlgorithm::populate: out[code(kmer)] = index of the k-mer's abundance in MapMPHF's discretization table (MapMPHF.hpp:96-145),
 * for every solid k-mer of the context; *nb_above_precision = abundances beyond the table (MPHFAlgorithm.cpp:254-258) */
int      gkc_mphf_abundance_map(gkc_mphf* m, gkc_ctx* ctx, uint8_t* out, uint64_t cap, uint64_t* nb_above_precision);

/* ---- input: FASTA / FASTQ text -> flat bases + offsets ON THE DEVICE ------------------------------------
 * Replaces BankFasta::Iterator::get_next_seq_from_file (bank/impl/BankFasta.cpp:488-571, buffered_gets :425-483) and the
 * per-sequence copy into the flat buffer that gkc_push_reads takes. Same result as the reference reader for well-formed text:
 *   FASTA : a line starting with '>' or '@' is a header, all other lines are sequence data (multi-line, CRLF: one trailing '\r' per
 *           line dropped exactly like BankFasta.cpp:479; empty lines ignored);
 *   FASTQ : four lines per record, quality at least as long as the sequence.
 * Anything whose result under the reference's character state machine is NOT expressible per line (multi-line FASTQ, quality shorter
 * than the sequence, a sequence line starting with '+', '>' / '@' before the first header line) returns GKC_ERR_FORMAT — there is no
 * host fallback inside the library.
 * d_text: n_bytes of text in device memory. final_chunk = 0: only complete records are parsed and *consumed tells how many bytes
 * they covered (feed the rest again in front of the next chunk); 1: the text ends here. Outputs are allocated by the library
 * (gkc_device_free): d_bases[n_bases], d_offsets[n_reads + 1] — exactly the arguments of gkc_push_reads_device. */
int gkc_fastx_parse_device(gkc_ctx* ctx, const char* d_text, uint64_t n_bytes, int final_chunk,
                           char** d_bases, uint64_t** d_offsets, uint64_t* n_reads, uint64_t* n_bases, uint64_t* consumed);
/* host text -> H2D -> gkc_fastx_parse_device -> gkc_push_reads_device, inside a pass */
int gkc_push_fastx(gkc_ctx* ctx, const char* text, uint64_t n_bytes, int final_chunk, uint64_t* consumed);

int gkc_device_free(gkc_ctx* ctx, void* d_ptr);
/* Gives the result buffers of a finished pass back (its datasets are gone afterwards: gkc_partition_* on them fail). A host that has drained
 * a pass — the processors have seen every partition — calls it before the next pass, so that a multi-pass count keeps only ONE pass of results
 * in HBM: the point of nb_passes in the reference, which bounds the footprint of a pass and removes the partition files of a pass before the
 * next one (SortingCountAlgorithm.cpp:1219-1241, :705-720; pass count from the available memory / disk: ConfigurationAlgorithm.cpp:398-425). */
int gkc_release_pass(gkc_ctx* ctx, uint32_t pass);
/* HBM the context can use right now (free on the device + blocks parked in its own allocator) and the device total, in bytes: what a host-side
 * configuration step sizes nb_passes / nb_partitions from (the reference reads System::info().getMemoryPhysicalTotal / -max-memory there). */
int gkc_device_memory(gkc_ctx* ctx, uint64_t* usable_bytes, uint64_t* total_bytes);
/* Page-locked host memory for the buffers the host side hands to gkc_push_reads / gkc_push_fastx and receives Count[] records in
 * (gkc_partition_counts): the DMA engines then move them at PCIe rate, pageable memory is staged by the driver at about half of it
 * (measured: DESIGN.md section 6). This is the role of the reference's host-side buffer provider for partition data
 * (tools/misc/impl/Pool.hpp:343-418, MemAllocator, and the BagCache buffers of CountProcessorDump.hpp:134): memory the counting
 * back-end fills and the processors read. No context needed; *p = NULL and GKC_ERR_NOMEM when the allocation fails. */
int gkc_host_alloc(void** p, uint64_t n_bytes);
int gkc_host_free(void* p);
int gkc_device_to_host(gkc_ctx* ctx, void* dst, const void* d_src, uint64_t n_bytes);
int gkc_host_to_device(gkc_ctx* ctx, void* d_dst, const void* src, uint64_t n_bytes);
/* order-independent checksum of the canonical k-mer multiset of device-resident reads, computed by an independent
 * one-thread-per-position kernel (no minimizers, no buckets): sum over valid k-mers of mix(canonical) mod 2^64, and the
 * number of valid k-mers. Used by the full-size parity property "sum_records count*mix(value) == this". */
int gkc_kmer_checksum_device(gkc_ctx* ctx, const char* d_bases, const uint64_t* d_offsets, uint64_t n_reads,
                             uint64_t n_bases, uint64_t* checksum, uint64_t* n_valid);
/* the same checksum over the finished datasets of the context: sum abundance*mix(value), sum abundance */
int gkc_result_checksum(gkc_ctx* ctx, uint64_t* checksum, uint64_t* sum_abundance);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GKC_H */
