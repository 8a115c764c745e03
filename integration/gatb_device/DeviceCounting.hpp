/* DeviceCounting.hpp — the reference-side binding of libgkc_hip.so: what a GATB-Core maintainer adds to
 * src/gatb/kmer/impl/ so that SortingCountAlgorithm<span> counts on an MI355X. Compiled against the reference's own headers by
 * integration/check_integration.sh (it is NOT part of libgkc_hip.so and includes nothing of this repository but include/gkc.h).
 *
 *   DeviceSession               one gkc_ctx per process (one GPU per process), configured from the reference's Configuration + Repartitor;
 *                               optionally one rank of a communicator (environment, see enableMultiGpuFromEnvironment)
 *   FillPartitionsDevice<span>  the functor fillPartitions() iterates the bank with (SortingCountAlgorithm.cpp:1081-1151 FillPartitions,
 *                               dispatched at :1266-1275): every worker thread packs its sequences into its OWN flat ASCII buffer + CSR offsets
 *                               (no lock per sequence) and hands a full buffer to gkc_push_reads (Stage A on the device)
 *   PartitionsByDeviceCommand   the ICommand fillSolidKmers_aux() dispatches per partition (SortingCountAlgorithm.cpp:1456-1587), beside
 *                               PartitionsByVectorCommand / PartitionsByHashCommand (PartitionsCommand.hpp:100-160): waits for the device's
 *                               Count[] of its partition and hands it to the count processor — as ONE block when the processor is the default
 *                               chain (histogram -> solidity -> dump, SortingCountAlgorithm.cpp:376-400), record by record otherwise
 */
#ifndef _GATB_CORE_KMER_IMPL_DEVICE_COUNTING_HPP_
#define _GATB_CORE_KMER_IMPL_DEVICE_COUNTING_HPP_

#include <gatb/kmer/impl/PartitionsCommand.hpp>
#include <gatb/kmer/impl/PartiInfo.hpp>
#include <gatb/kmer/impl/BankKmers.hpp>
#include <gatb/kmer/impl/Configuration.hpp>
#include <gatb/kmer/impl/CountProcessor.hpp>
#include <gatb/bank/api/Sequence.hpp>
#include <gatb/system/api/Exception.hpp>
#include <gatb/system/impl/System.hpp>

#include <gkc.h>
#include <gatb_device/DeviceContext.hpp>
#include <gatb_device/SolidSinkDirect.hpp>

#include <vector>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <deque>
#include <list>
#include <atomic>
#include <thread>
#include <memory>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>
#include <gatb/bank/impl/BankFasta.hpp>
#include <gatb/bank/impl/BankComposite.hpp>
#include <string>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

namespace gatb { namespace core { namespace kmer { namespace impl {

/********************************************************************************/
/** How the Count records of a partition reach the count processor. */
struct DeviceBulkPlan
{
    bool     on;           /**< the processor is the default chain with ONE abundance range: the device applies the solidity window and the histogram,
                                the dump takes each partition's records as one block (BagHDF5Patch::insert(const Item*, size_t), CollectionHDF5Patch.hpp:262-308) */
    int32_t  abundanceMin, abundanceMax;
    uint32_t histoMax;
    DeviceBulkPlan () : on(false), abundanceMin(1), abundanceMax(2147483647), histoMax(10000) {}
};

/********************************************************************************/
/** One device context per process. */
class DeviceSession
{
public:
    static DeviceSession& singleton ()  { static DeviceSession s; return s; }

    gkc_ctx* ctx ()  {  open();  return _ctx;  }

    /** Configuration + Repartitor of the run -> gkc_configure (replaces `Model model(...)`, SortingCountAlgorithm.cpp:1251-1256, and
     *  Repartitor::operator(), PartiInfo.hpp:323). Without a bulk plan the solidity window stays open and the reference's processor chain filters. */
    void configure (const Configuration& config, Repartitor& repartitor, const DeviceBulkPlan& plan, size_t pass)
    {
        open();
        if (pass > 0)  { return; }                                   /* one configuration per run: the passes share the datasets and the histogram */
        enableMultiGpuFromEnvironment();
        device::DeviceContext::singleton().setResident (device::DeviceContext::Resident());      /* a new count: what an earlier one left in HBM is going away */
        _plan = plan;  _nbPartitions = config._nb_partitions;  _kmerSize = config._kmerSize;
        _waitS = _handOverS = 0;
        if (plan.on  &&  _prepareFn)  { prepareRing (config._nbCores); }     /* (page-locked once, before anything runs on the device; only where the direct sink can take the records: bulkPlan set a preparer) */
        const u_int64_t nbMinims = (u_int64_t)1 << (2 * config._minim_size);
        std::vector<uint16_t> table (nbMinims);
        for (u_int64_t m = 0; m < nbMinims; m++)  { table[m] = (uint16_t) repartitor (m); }
        /* one pass per process is what a dbgh5 run is: batches of 2^30 k-mers instead of the library's 3.2e9 (a host that counts pass after pass keeps those) —
         * a third of Stage B's working set for 8 % of its time; on a freshly booted box, where HBM beyond ~120 GB costs 28 ms per GB the first time it is handed out,
         * that is 2 s of the one pass there is (profiles/r05_cold_pass.txt). GATB_DEVICE_BATCH_KEYS=<n> sets another bound, 0 the library's plan. */
        {
            const char* keys = getenv ("GATB_DEVICE_BATCH_KEYS");
            check (gkc_set_batch_keys (_ctx, keys != 0 ? (uint64_t) strtoull (keys, 0, 10) : ((uint64_t)1 << 30)));
        }
        if (plan.on)  { check (gkc_set_solidity (_ctx, plan.abundanceMin, plan.abundanceMax, plan.histoMax)); }
        else          { check (gkc_set_solidity (_ctx, 1, 2147483647, 10000)); }      /* the histogram and the filter are the chain's job */
        check (gkc_configure (_ctx, (uint32_t) config._kmerSize, (uint32_t) config._minim_size, (uint32_t) config._nb_partitions, (uint32_t) config._nb_passes,
                              config._minimizerType == 1 ? GKC_MINIMIZER_FREQ : GKC_MINIMIZER_LEXI, table.data(),
                              config._minimizerType == 1 ? repartitor.getMinimizerFrequencies() : 0));
        /* multi-rank: a fixed number of exchanges per pass, the same on every rank (derived from the bank estimate every rank computes from the same bank),
         * so that the i-th gkc_exchange of one rank always meets the i-th of the others whatever the threads' timing */
        _nbExchanges = 1;
        if (_comm != 0)  { _nbExchanges = (uint32_t) std::max<u_int64_t> (1, std::min<u_int64_t> (64, config._estimateSeqNb / 4000000)); }
        _readsPerExchange = std::max<u_int64_t> (1, config._estimateSeqNb / (u_int64_t)_ranks / _nbExchanges);
    }

    const DeviceBulkPlan& plan () const  { return _plan; }
    size_t nbPartitions () const  { return _nbPartitions; }

    void check (int rc)
    {
        if (rc != GKC_OK)  { throw system::Exception ("device counting: error %d: %s", rc, gkc_last_error(_ctx)); }
    }

    /** Multi-GPU (one process per GPU). In the environment of each process:
     *      GATB_DEVICE_RANKS, GATB_DEVICE_RANK          world size and rank
     *      GATB_DEVICE_COMM_ID = <file>                 RCCL over xGMI: rank 0 writes the 128-byte ncclUniqueId there, the others wait for it
     *   or GATB_DEVICE_TRANSPORT_DIR = <directory>      the library's file-mailbox transport (ranks RCCL cannot connect: two processes on one GPU, no xGMI)
     *  Every process opens the SAME bank with the same options (so that Configuration and Repartitor are the same everywhere: gkc_exchange checks it) and
     *  counts the sequences whose index is its rank modulo the world size; after every few million reads the super-k-mers are routed to the rank owning
     *  their partition (gkc_exchange, overlapping the scan of the next reads); at the end of the pass the Count[] of all partitions, the histogram and the
     *  statistics are gathered on rank 0 (gkc_gather_results), whose .h5 is THE result: every dataset of the single-process file, in one file. */
    void enableMultiGpuFromEnvironment ()
    {
        const char* ranks = getenv ("GATB_DEVICE_RANKS");  const char* rank = getenv ("GATB_DEVICE_RANK");
        const char* idFile = getenv ("GATB_DEVICE_COMM_ID");  const char* boxDir = getenv ("GATB_DEVICE_TRANSPORT_DIR");
        if (_comm != 0  ||  ranks == 0  ||  rank == 0  ||  (idFile == 0 && boxDir == 0)  ||  atoi(ranks) < 2)  { return; }
        open();
        _ranks = atoi(ranks);  _rank = atoi(rank);
        if (boxDir != 0)  { check (gkc_comm_create_files (_ctx, boxDir, _ranks, _rank, &_comm));  return; }
        uint8_t id [GKC_COMM_ID_BYTES];
        if (_rank == 0)
        {
            check (gkc_comm_unique_id (id));
            std::string tmp = std::string(idFile) + ".tmp";
            FILE* f = fopen (tmp.c_str(), "wb");  if (f == 0) { throw system::Exception ("device counting: cannot write %s", tmp.c_str()); }
            fwrite (id, 1, sizeof(id), f);  fclose (f);  rename (tmp.c_str(), idFile);
        }
        else
        {
            FILE* f = 0;
            for (int tries = 0; tries < 6000 && (f = fopen (idFile, "rb")) == 0; tries++)  { usleep (10000); }
            if (f == 0  ||  fread (id, 1, sizeof(id), f) != sizeof(id))  { throw system::Exception ("device counting: no communicator id in %s", idFile); }
            fclose (f);
        }
        check (gkc_comm_create_rccl (_ctx, id, _ranks, _rank, &_comm));
    }
    gkc_comm* comm  ()  { return _comm; }
    int       ranks () const { return _ranks; }
    int       rank  () const { return _rank;  }

    void beginPass (size_t pass)
    {
        /* Several passes: the device results of the pass before have been handed to every processor (fillSolidKmers of that pass is over) and go back to the allocator —
         * DeviceConfiguration sizes a pass for ITS records + ITS Count records, not for the Count records of all passes so far. What BloomAlgorithm /
         * MPHFAlgorithm need later they then read from the storage (DeviceContext::Resident is only set by a one-pass count). */
        if (pass > 0  &&  pass != _releasedBelow)  { for (size_t q = _releasedBelow; q < pass; q++) { check (gkc_release_pass (_ctx, (uint32_t) q)); }  _releasedBelow = pass; }
        if (pass == 0)  { _releasedBelow = 0; }
        check (gkc_begin_pass (_ctx, (uint32_t)pass));  _pushedReads = 0;  _exchangesDone = 0;  _progressReported = 0;      /* (a debt of a refused text pass stays) */
    }

    /** one block of reads to Stage A (the caller holds the packers' lock: one thread drives the context at a time); multi-rank: the exchanges that are due */
    void push (const char* bases, const uint64_t* offsets, uint64_t nbReads)
    {
        check (gkc_push_reads (_ctx, bases, offsets, nbReads));
        _pushedReads += nbReads;
        while (_comm != 0  &&  _exchangesDone + 1 < _nbExchanges  &&  _pushedReads >= (u_int64_t)(_exchangesDone + 1) * _readsPerExchange)
        {
            check (gkc_exchange (_ctx, _comm));  _exchangesDone++;
        }
    }
    /** after the last read of the pass: the remaining exchanges (every rank makes exactly _nbExchanges of them) */
    void endOfReads ()
    {
        while (_comm != 0  &&  _exchangesDone < _nbExchanges)  { check (gkc_exchange (_ctx, _comm));  _exchangesDone++; }
    }
    /** The plain-text FASTA / FASTQ files behind a bank, in iteration order (a BankFasta holds one file, BankFasta.cpp:109-116; an album / a list of banks is a
     *  BankComposite of them, BankComposite.hpp:56-160). False when the bank is anything else, or a file is gzipped or cannot be opened: the bank is iterated then. */
    static bool plainTextFiles (bank::IBank* bank, std::vector<std::string>& files)
    {
        if (bank == 0)  { return false; }
        if (bank::impl::BankFasta* fasta = dynamic_cast<bank::impl::BankFasta*> (bank))
        {
            const std::string name = fasta->getId();
            FILE* f = fopen (name.c_str(), "rb");  if (f == 0)  { return false; }
            unsigned char magic[2] = {0, 0};  const size_t got = fread (magic, 1, 2, f);  fclose (f);
            /* gzip (BankFasta.cpp:425-483 inflates it inside the locked reader): one rank inflates it on a thread of its own into the same text path (pushTextFiles);
             * several ranks cannot cut a deflate stream into byte ranges: iterated */
            if (got == 2  &&  magic[0] == 0x1f  &&  magic[1] == 0x8b  &&  (singleton().ranks() > 1  ||  getenv ("GATB_DEVICE_NO_GZ") != 0))  { return false; }
            files.push_back (name);
            return true;
        }
        if (bank::impl::BankComposite* composite = dynamic_cast<bank::impl::BankComposite*> (bank))
        {
            const std::vector<bank::IBank*> subs = composite->getBanks();
            if (subs.empty())  { return false; }
            for (size_t i = 0; i < subs.size(); i++)  { if (subs[i] == bank  ||  !plainTextFiles (subs[i], files))  { return false; } }
            return true;
        }
        return false;
    }

    /** First byte of a record at or behind `pos` (a FASTA record starts with a '>' line; a FASTQ record with an '@' line whose second next line starts with '+':
     *  a quality line may start with '@' too, but the line two below it is then a sequence). `size` when there is none. */
    static uint64_t recordStart (int fd, uint64_t pos, uint64_t size, bool fastq)
    {
        if (pos == 0  ||  pos >= size)  { return std::min (pos, size); }
        std::vector<char> win;
        for (uint64_t len = 1 << 20;  ;  len *= 4)
        {
            const uint64_t from = pos - 1, n = std::min<uint64_t> (len, size - from);       /* from the byte before: is `pos` itself a line start? */
            win.resize (n);
            uint64_t done = 0;
            while (done < n)
            {
                const ssize_t g = pread (fd, win.data() + done, n - done, (off_t)(from + done));
                if (g <= 0)  { throw system::Exception ("device counting: read error while looking for a record start (offset %llu)", (unsigned long long)(from + done)); }   /* never a silently shortened byte range */
                done += (uint64_t)g;
            }
            std::vector<uint64_t> lines;                                                    /* line starts inside the window (window offsets) */
            for (uint64_t i = 1; i < n; i++)  { if (win[i-1] == '\n')  { lines.push_back (i); } }
            for (size_t l = 0; l < lines.size(); l++)
            {
                const char c0 = win[lines[l]];
                if (!fastq)  { if (c0 == '>')  { return from + lines[l]; }  continue; }
                if (c0 == '@'  &&  l + 2 < lines.size()  &&  win[lines[l+2]] == '+')  { return from + lines[l]; }
            }
            if (from + n >= size)  { return size; }
        }
    }

    /** The whole text of the files to the device: parsed there (gkc_push_fastx: gkc_fastx_parse_device + Stage A) instead of sequence by sequence through
     *  BankFasta::Iterator (BankFasta.cpp:488-571) and the locked group reader of Dispatcher::iterate (ICommand.hpp:291-335). Chunks of CHUNK bytes are read
     *  by several threads (pread) — the next one while this one is parsed and scanned — and the bytes behind the last complete record of
     *  a chunk open the next one. Several ranks: every rank
     *  takes its own byte range of every file, cut at record starts (which rank scans which read does not matter: the super-k-mers go to the owner of their
     *  partition), and makes the exchanges that are due as its reads go by.
     *  Returns false when the text is not what the device parser takes (GKC_ERR_FORMAT: e.g. a multi-line FASTQ) — one rank: the caller starts the pass again
     *  and iterates the bank; several ranks: an error (the other ranks cannot be called back). */
    bool pushTextFiles (const std::vector<std::string>& files, gatb::core::tools::dp::IteratorListener* progress)
    {
        enum { CHUNK = 1 << 28, PAD = 1 << 24, READERS = 16 };      /* (large chunks: every chunk is one segment of super-k-mer records, and Stage B walks a partition segment by segment) */
        /* (ordinary memory: page-locking 2 x 272 MB costs more than the staged copy of 1.5 GB loses — measured 0.31 s against 0.26 s of fill_partitions at 10^7 reads;
         *  at 10^8 reads, 15.4 GB of text, with 32 readers: 1.63 s against 1.24-1.33 s — pread into page-locked pages is the slower side) */
        for (int i = 0; i < 2; i++)  { if (_text[i] == 0)  { _text[i] = (char*) malloc ((size_t)CHUNK + PAD + 64);  if (_text[i] == 0) { throw system::Exception ("device counting: out of host memory"); } } }
        uint64_t seenReads = 0;
        for (size_t fi = 0; fi < files.size(); fi++)
        {
            const int fd = ::open (files[fi].c_str(), O_RDONLY);
            if (fd < 0)  { throw system::Exception ("device counting: cannot open %s", files[fi].c_str()); }
            struct Handles  { int fd;  gzFile zf;  ~Handles ()  { if (zf != 0) { gzclose (zf); }  if (fd >= 0) { ::close (fd); } } }  handles = { fd, 0 };      /* closed on every way out, exceptions included */
            struct stat sb;  if (fstat (fd, &sb) != 0)  { throw system::Exception ("device counting: cannot stat %s", files[fi].c_str()); }
            const uint64_t fileSize = (uint64_t) sb.st_size;
            uint64_t off = 0, size = fileSize;
            if (_ranks > 1  &&  fileSize > 0)
            {
                char first = 0;  if (pread (fd, &first, 1, 0) != 1)  { throw system::Exception ("device counting: read error in %s", files[fi].c_str()); }
                const bool fastq = first == '@';
                off  = recordStart (fd, fileSize / (uint64_t)_ranks * (uint64_t)_rank, fileSize, fastq);
                size = _rank + 1 == _ranks ? fileSize : recordStart (fd, fileSize / (uint64_t)_ranks * (uint64_t)(_rank + 1), fileSize, fastq);
            }
            /* n bytes of the file from `from` to dst, by READERS threads; false on a read error */
            auto fill = [fd] (char* dst, uint64_t from, uint64_t n) -> bool
            {
                std::vector<std::thread> readers;  std::atomic<bool> failed (false);
                const uint64_t share = (n + READERS - 1) / READERS;
                for (int r = 0; r < READERS; r++)
                {
                    const uint64_t b = std::min<uint64_t> (n, (uint64_t)r * share), e = std::min<uint64_t> (n, b + share);
                    if (b < e)  { readers.emplace_back ([=, &failed] { uint64_t done = b;  while (done < e) { const ssize_t g = pread (fd, dst + done, e - done, (off_t)(from + done));  if (g <= 0) { failed = true; break; }  done += (uint64_t)g; } }); }
                }
                for (size_t r = 0; r < readers.size(); r++)  { readers[r].join(); }
                return !failed.load();
            };
            /* a gzipped file (one rank): the text comes out of zlib on the prefetch thread — CHUNK bytes of text at a time, while the chunk before is parsed and scanned
             * on the device; where the text ends is only known when gzread returns less than asked, so the chunks go to the parser as "not the last" and what the last
             * one leaves (a record without its newline) is pushed as the final piece */
            unsigned char magic[2] = {0, 0};
            const bool gz = pread (fd, magic, 2, 0) == 2  &&  magic[0] == 0x1f  &&  magic[1] == 0x8b;
            gzFile zf = 0;
            if (gz)
            {
                zf = gzopen (files[fi].c_str(), "rb");
                if (zf == 0)  { throw system::Exception ("device counting: cannot open %s through zlib", files[fi].c_str()); }
                handles.zf = zf;
                gzbuffer (zf, 1 << 20);
                if (getenv ("GATB_DEVICE_VERBOSE") != 0)  { fprintf (stderr, "[device counting] %s: gzipped text, inflated on a host thread into the device parser\n", files[fi].c_str()); }
                size = ~(uint64_t)0;  off = 0;
            }
            auto inflate = [zf] (char* dst, uint64_t n, uint64_t& got) -> bool      /* up to n bytes of text; got < n: the end */
            {
                got = 0;
                while (got < n)
                {
                    const int g = gzread (zf, dst + got, (unsigned) std::min<uint64_t> (n - got, (uint64_t)1 << 30));
                    if (g < 0)  { return false; }
                    if (g == 0)  { break; }
                    got += (uint64_t) g;
                }
                return true;
            };
            /* two buffers: the next chunk is read (behind PAD bytes of room for what the parser leaves of this one) while this one is parsed and scanned */
            int cur = 0;
            uint64_t have = gz ? 0 : std::min<uint64_t> ((uint64_t)CHUNK, size - off);
            char* ptr = _text[cur] + PAD;
            bool ok = true, readError = gz ? !inflate (ptr, (uint64_t)CHUNK, have) : (have > 0 && !fill (ptr, off, have));
            bool gzEnd = gz && have < (uint64_t)CHUNK;
            if (!gz)  { off += have; }
            while (ok  &&  !readError  &&  have > 0)
            {
                uint64_t next = gz ? (gzEnd ? 0 : (uint64_t)CHUNK) : std::min<uint64_t> ((uint64_t)CHUNK, size - off);
                std::atomic<bool> nextFailed (false);
                std::atomic<uint64_t> nextGot (0);
                std::thread prefetch;
                if (next > 0)
                {
                    char* dst = _text[cur ^ 1] + PAD;  const uint64_t from = off;
                    if (gz)  { prefetch = std::thread ([&inflate, &nextFailed, &nextGot, dst, next] { uint64_t g = 0;  nextFailed = !inflate (dst, next, g);  nextGot = g; }); }
                    else     { prefetch = std::thread ([&fill, &nextFailed, dst, from, next] { nextFailed = !fill (dst, from, next); }); }
                }
                const int final = next == 0 ? 1 : 0;
                uint64_t consumed = 0;
                const int rc = gkc_push_fastx (_ctx, ptr, have, final, &consumed);
                if (prefetch.joinable())  { prefetch.join(); }
                readError = nextFailed.load();
                if (gz  &&  next > 0)  { next = nextGot.load();  gzEnd = next < (uint64_t)CHUNK; }      /* (next == 0: the text ended exactly at the chunk: what is left goes out as the final piece below) */
                const uint64_t left = final ? 0 : have - consumed;
                if (rc == GKC_ERR_FORMAT  ||  (rc == GKC_OK  &&  left > (uint64_t)PAD))  { ok = false;  break; }      /* (or a record larger than the room: a genome, not reads) */
                if (rc != GKC_OK)  { check (rc); }
                /* reads of THIS pass so far (gkc_stats.nb_sequences is pass 0's only: on later passes it would add the whole bank at the first chunk and
                 * nothing afterwards — progress, and with several ranks the pacing of the exchanges, need the pass's own count) */
                gkc_stats st;  check (gkc_get_stats (_ctx, &st));
                const uint64_t passReads = st.reserved[0];
                if (passReads > seenReads)
                {
                    if (progress != 0)  { progress->inc (passReads - seenReads);  _progressReported += passReads - seenReads; }
                    _pushedReads += passReads - seenReads;  seenReads = passReads;
                }
                while (_comm != 0  &&  _exchangesDone + 1 < _nbExchanges  &&  _pushedReads >= (u_int64_t)(_exchangesDone + 1) * _readsPerExchange)
                {
                    check (gkc_exchange (_ctx, _comm));  _exchangesDone++;
                }
                if (next == 0)
                {
                    if (!final  &&  left > 0)                              /* gz: the text ended with the chunk just pushed as "not the last": its tail is the final piece */
                    {
                        uint64_t c2 = 0;
                        const int rc2 = gkc_push_fastx (_ctx, ptr + consumed, left, 1, &c2);
                        if (rc2 == GKC_ERR_FORMAT)  { ok = false; }  else if (rc2 != GKC_OK)  { check (rc2); }
                    }
                    break;
                }
                char* nptr = _text[cur ^ 1] + PAD - left;
                if (left > 0)  { memcpy (nptr, ptr + consumed, left); }
                ptr = nptr;  have = left + next;  if (!gz) { off += next; }  cur ^= 1;
            }
            if (readError)  { throw system::Exception ("device counting: read error in %s", files[fi].c_str()); }
            if (!ok)
            {
                if (_ranks > 1)  { throw system::Exception ("device counting: %s is not FASTA / FASTQ text the device parser takes; with several ranks set GATB_DEVICE_NO_TEXT=1", files[fi].c_str()); }
                /* the pass starts again, iterated: what has been reported already is owed — the iterating functors report that much less (reportable()); a progress
                 * listener cannot be wound back (Progress::inc loops over the steps it is given, Progress.cpp:119-129) */
                {  std::lock_guard<std::mutex> guard (_timesLock);  _progressDebt += _progressReported;  _progressReported = 0;  }
                return false;
            }
        }
        return true;
    }

    /** Stage B. One rank: started in the background, the partition commands wait for their partition. Several ranks: counted, then gathered on rank 0. */
    void finishPass (size_t pass)
    {
        _finishWall = wallNow();
        if (_comm == 0)  { check (gkc_finish_pass_async (_ctx));  _stageBPending = true;  startPreparer (pass);  return; }
        check (gkc_finish_pass (_ctx));
        check (gkc_gather_results (_ctx, _comm, 0));
    }
    /** after the partition commands of the pass; after the last pass of a bulk-mode count the device's datasets ARE /dsk/solid (the window was applied on the
     *  device) and they stay in HBM: BloomAlgorithm / MPHFAlgorithm find them there (DeviceContext::residentMatches) */
    void joinPass (size_t pass, size_t nbPasses)
    {
        /* (fillSolidKmers_aux runs once per processor of the run — two with -abundance-min auto, SortingCountAlgorithm.cpp:1388-1393 — over the same device results:
         *  Stage B is joined by the first) */
        if (_comm == 0  &&  _stageBPending)  { _stageBPending = false;  check (gkc_finish_pass_wait (_ctx)); }
        joinPreparer();
        if (hasRing())  { drainWriter(); }                                /* every Count[] of the pass is in the file */
        if (pass + 1 == nbPasses)  { freeRing(); }
        if (pass + 1 == nbPasses  &&  nbPasses == 1  &&  _plan.on  &&  (_comm == 0  ||  _rank == 0))
        {
            gkc_stats st;  check (gkc_get_stats (_ctx, &st));
            device::DeviceContext::Resident r;
            r.on = true;  r.nbSolid = st.kmers_nb_solid;  r.kmerSize = (uint32_t) _kmerSize;  r.keyBytes = _kmerSize <= 31 ? 8 : 16;
            device::DeviceContext::singleton().setResident (r);
        }
        if (getenv ("GATB_DEVICE_VERBOSE") != 0  &&  _writerBytes > 0)
            fprintf (stderr, "[device counting] the file's writer thread: %.2f GB in %.2f s of pwrite (%.2f GB/s while writing)\n", _writerBytes / 1e9, _writerBusyS, _writerBytes / 1e9 / std::max (1e-9, _writerBusyS));
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
        {
            double a = 0, b = 0;  uint64_t n = 0;
            gkc_get_timing (_ctx, "total_stage_a", &a, &n);  gkc_get_timing (_ctx, "total_stage_b", &b, &n);
            fprintf (stderr, "[device counting] Stage A %.0f ms, Stage B %.0f ms on the device (summed over the passes so far); %.2f s of wall between the start of Stage B and the last partition handed over\n",
                     a, b, wallNow() - _finishWall);
        }
    }
    static double wallNow ()  { struct timespec ts;  clock_gettime (CLOCK_MONOTONIC, &ts);  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

    /** The device's figures where the reference's commands put "1.read / 2.sort / 3.dump" (SortingCountAlgorithm.cpp:777-780, PartitionsCommand.cpp TIME_INFO):
     *  device_stage_a / device_stage_b = HIP-event time of the two stages on the MI355X summed over the passes, device_wait / device_hand_over = host wall the
     *  partition commands spent waiting for Stage B and handing Count[] to the processors, divided by the dispatcher's units like the
     *  reference's entries (:777). TimeInfo only counts between start() and stop() of its clock (TimeInfo.hpp:86-92): a clock set by hand carries the values in. */
    void addTimes (tools::misc::impl::TimeInfo& into, size_t nbUnits)
    {
        struct Clock : public system::ITime
        {
            u_int32_t now;  Clock () : now(0) {}
            Value getTimeStamp ()  { return now; }
            Unit getUnit ()  { return MSEC; }
            std::string getDateString ()  { return ""; }
        } clock;
        tools::misc::impl::TimeInfo mine (clock);
        double a = 0, b = 0;  uint64_t n = 0;
        if (_ctx != 0)  { gkc_get_timing (_ctx, "total_stage_a", &a, &n);  gkc_get_timing (_ctx, "total_stage_b", &b, &n); }
        const char* names[4] = { "device_stage_a", "device_stage_b", "device_wait", "device_hand_over" };
        const double ms[4] = { a, b, 1000.0 * _waitS / (double) std::max<size_t> (1, nbUnits), 1000.0 * _handOverS / (double) std::max<size_t> (1, nbUnits) };
        for (int i = 0; i < 4; i++)  { clock.now = 0;  mine.start (names[i]);  clock.now = (u_int32_t) (ms[i] + 0.5);  mine.stop (names[i]); }
        into += mine;
    }
    /** how much of n sequences an iterating functor may report to the progress listener: what the refused text path had reported for this pass comes off first */
    u_int64_t reportable (u_int64_t n)
    {
        std::lock_guard<std::mutex> guard (_timesLock);
        const u_int64_t owed = std::min (n, _progressDebt);
        _progressDebt -= owed;
        return n - owed;
    }
    /** The ring between the link and the result file (bulk mode): `slots` page-locked buffers of SLOT_BYTES in ONE allocation made before the reads are scanned
     *  (page-locking memory while Stage B runs stalls its launches: measured, 0.2 -> 2-4 s), and ONE writer thread. A partition command makes its dataset
     *  (SolidSinkDirect::prepare), then moves its Count[] piece by piece: take a slot, gkc_partition_counts_range into it (DMA at the link's rate; pageable memory is
     *  staged by the runtime at ~3.4 GB/s with 256 threads asking at once), queue {slot, file offset}; the writer pwrite()s the pieces in the order they were queued
     *  and gives the slots back. One writer because that is what one file takes fastest (tools/filewrite_probe: 9.1 GB/s against 3-4 GB/s with several). */
    enum { SLOT_BYTES = 4 << 20 };
    void prepareRing (size_t nbCores)
    {
        if (_ring != 0  ||  getenv ("GATB_DEVICE_NO_RING") != 0)  { return; }
        size_t slots = 256;                                              /* 1 GiB: a whole group of nb-cores partition commands queues its records and returns while the writer is still at it */
        (void) nbCores;
        if (getenv ("GATB_DEVICE_RING_SLOTS") != 0)  { slots = std::max (1, atoi (getenv ("GATB_DEVICE_RING_SLOTS"))); }
        void* p = 0;
        if (gkc_host_alloc (&p, (uint64_t) slots * SLOT_BYTES) != GKC_OK)  { return; }      /* no page-locked memory: the commands fetch into pageable memory */
        std::lock_guard<std::mutex> guard (_ringLock);
        _ring = p;  _ringFree.clear();
        for (size_t i = 0; i < slots; i++)  { _ringFree.push_back ((char*) p + i * (size_t) SLOT_BYTES); }
        _writerStop = false;  _writerError.clear();  _jobsPending = 0;
        _writer = std::thread ([this] { writerLoop(); });
    }
    bool  hasRing ()  { return _ring != 0; }
    void* takeSlot ()  { std::unique_lock<std::mutex> lk (_ringLock);  _ringCv.wait (lk, [this] { return !_ringFree.empty(); });  void* p = _ringFree.back();  _ringFree.pop_back();  return p; }
    void  giveSlot (void* p)  { { std::lock_guard<std::mutex> guard (_ringLock);  _ringFree.push_back (p); }  _ringCv.notify_one(); }
    /** `bytes` of a slot to `offset` of the file `path`; the writer gives the slot back */
    void  queueWrite (const std::string& path, uint64_t offset, void* slot, size_t bytes)
    {
        { std::lock_guard<std::mutex> guard (_ringLock);  WriteJob j;  j.path = &pathOf (path);  j.offset = offset;  j.slot = slot;  j.bytes = bytes;  _jobs.push_back (j);  _jobsPending++; }
        _jobCv.notify_one();
    }
    /** everything queued is in the file (throws what the writer met) */
    void  drainWriter ()
    {
        std::unique_lock<std::mutex> lk (_ringLock);
        _idleCv.wait (lk, [this] { return _jobsPending == 0; });
        if (!_writerError.empty())  { const std::string e = _writerError;  _writerError.clear();  throw system::Exception ("%s", e.c_str()); }
    }
    void  freeRing ()
    {
        if (_writer.joinable())
        {
            { std::lock_guard<std::mutex> guard (_ringLock);  _writerStop = true; }
            _jobCv.notify_all();  _writer.join();
        }
        std::lock_guard<std::mutex> guard (_ringLock);
        if (_ring)  { gkc_host_free (_ring); }
        _ring = 0;  _ringFree.clear();  _paths.clear();
    }

    /** The datasets of a pass made AHEAD of the partition commands, by one thread. Every command used to replace its partition's dataset itself
     *  (SolidSinkDirect::prepare: H5Ldelete + H5Dcreate2 + H5Dget_offset under the storage's lock): 26 us alone, but with the 256 commands of a group
     *  arriving at the lock together ~0.4-0.6 ms each — at 3884 partitions more than the 1.5 s the writer needs for the records, and the writer starved
     *  (fill_solid_kmers 2.3-2.5 s against 1.84 s with 2816 partitions and the same bytes: profiles/r05_dropin_timing_1e8reads.txt). Now one thread follows Stage B
     *  partition by partition (gkc_wait_partition), makes the dataset, publishes {address, path}; a command only waits for its entry.
     *  `fn` comes from PartitionsByDeviceCommand<span>::bulkPlan (it knows the Count type and the Partition). GATB_DEVICE_NO_PREPARER=1: the commands prepare. */
    typedef std::function<bool (size_t part, size_t pass, uint64_t nbItems, uint64_t& address, std::string& path)> PrepareFn;
    void setPreparer (const PrepareFn& fn)  { joinPreparer();  _prepareFn = fn; }
    void startPreparer (size_t pass)
    {
        joinPreparer();
        if (!_plan.on  ||  !hasRing()  ||  !_prepareFn  ||  getenv ("GATB_DEVICE_NO_PREPARER") != 0)  { return; }
        { std::lock_guard<std::mutex> guard (_prepLock);  _prepared.assign (_nbPartitions, Prepared());  _prepError.clear();  _prepActive = true; }
        _preparer = std::thread ([this, pass] { preparerLoop (pass); });
    }
    void joinPreparer ()
    {
        if (_preparer.joinable())  { _preparer.join(); }
        std::lock_guard<std::mutex> guard (_prepLock);  _prepActive = false;
    }
    bool preparerActive ()  { std::lock_guard<std::mutex> guard (_prepLock);  return _prepActive; }
    /** the entry of a partition: true = its dataset exists at its final size, the records go to `address` of `path`; false = not a direct-sink collection */
    bool waitPrepared (size_t part, uint64_t& address, std::string& path)
    {
        std::unique_lock<std::mutex> lk (_prepLock);
        _prepCv.wait (lk, [this, part] { return !_prepError.empty()  ||  _prepared[part].state != 0; });
        if (_prepared[part].state == 0)  { throw system::Exception ("%s", _prepError.c_str()); }
        address = _prepared[part].address;  path = _prepared[part].path;
        return _prepared[part].state == 1;
    }

    /** called by every partition command: seconds it waited for Stage B / spent handing its records over */
    void addCommandTimes (double waitS, double handOverS)  { std::lock_guard<std::mutex> guard (_timesLock);  _waitS += waitS;  _handOverS += handOverS; }

    ~DeviceSession ()  { if (_preparer.joinable()) { _preparer.join(); }  freeRing();  for (int i = 0; i < 2; i++) { if (_text[i]) { free (_text[i]); } }  if (_comm) { gkc_comm_destroy (_comm); } }      /* (the context is the process's: DeviceContext) */

private:
    DeviceSession () : _ctx(0), _comm(0), _ranks(1), _rank(0), _nbExchanges(1), _exchangesDone(0), _readsPerExchange(1), _pushedReads(0), _nbPartitions(0) {}
    void open ()
    {
        if (_ctx == 0)
        {
            _ctx = device::DeviceContext::singleton().ctx();
            if (_ctx == 0)  { throw system::Exception ("device counting: %s", device::DeviceContext::singleton().error().c_str()); }
        }
    }
    char*     _text[2] = {0, 0};        /**< text buffers of pushTextFiles */
    gkc_ctx*  _ctx;
    gkc_comm* _comm;
    int       _ranks, _rank;
    uint32_t  _nbExchanges, _exchangesDone;
    u_int64_t _readsPerExchange, _pushedReads;
    DeviceBulkPlan _plan;
    size_t    _nbPartitions;
    size_t    _kmerSize = 0;
    double    _finishWall = 0;
    bool      _stageBPending = false;
    size_t    _releasedBelow = 0;          /**< passes [0, _releasedBelow) have been released on the device */
    u_int64_t _progressReported = 0, _progressDebt = 0;
    std::mutex _timesLock;
    double    _waitS = 0, _handOverS = 0;
    struct WriteJob  { const std::string* path;  uint64_t offset;  void* slot;  size_t bytes; };
    const std::string& pathOf (const std::string& p)  { for (std::list<std::string>::iterator it = _paths.begin(); it != _paths.end(); ++it) { if (*it == p) { return *it; } }  _paths.push_back (p);  return _paths.back(); }      /* (under _ringLock) */
    void writerLoop ()
    {
        int fd = -1;  const std::string* open_ = 0;
        for (;;)
        {
            WriteJob j;
            {
                std::unique_lock<std::mutex> lk (_ringLock);
                _jobCv.wait (lk, [this] { return _writerStop  ||  !_jobs.empty(); });
                if (_jobs.empty())  { break; }
                j = _jobs.front();  _jobs.pop_front();
            }
            std::string error;
            try
            {
                if (j.path != open_)  { if (fd >= 0) { ::close (fd); }  fd = ::open (j.path->c_str(), O_WRONLY);  open_ = j.path;
                                         if (fd < 0)  { open_ = 0;  throw system::Exception ("device sink: open (%s): %s", j.path->c_str(), strerror (errno)); } }
                const double w0 = wallNow();
                SolidSinkDirect<char>::writeAt (fd, j.offset, j.slot, j.bytes);
                _writerBusyS += wallNow() - w0;  _writerBytes += j.bytes;
            }
            catch (system::Exception& e)  { error = e.getMessage(); }
            catch (std::exception& e)     { error = e.what(); }           /* (the commands wait in takeSlot / drainWriter: the writer must outlive whatever it meets) */
            giveSlot (j.slot);
            { std::lock_guard<std::mutex> guard (_ringLock);  if (!error.empty()  &&  _writerError.empty()) { _writerError = error; }  _jobsPending--; }
            _idleCv.notify_all();
        }
        if (fd >= 0)  { ::close (fd); }
    }
    struct Prepared  { int state;  uint64_t address;  std::string path;  Prepared () : state(0), address(0) {} };      /* state 0: not yet, 1: direct, 2: the collection's own insert */
    void preparerLoop (size_t pass)
    {
        std::string error;
        try
        {
            for (size_t p = 0; p < _nbPartitions; p++)
            {
                const void* landed = 0;  uint64_t nbSolid = 0;
                check (gkc_wait_partition (_ctx, (uint32_t) pass, (uint32_t) p, &landed, &nbSolid));
                Prepared e;  e.state = 2;
                if (landed == 0  &&  nbSolid > 0  &&  _prepareFn (p, pass, nbSolid, e.address, e.path))  { e.state = 1; }
                { std::lock_guard<std::mutex> guard (_prepLock);  _prepared[p] = e; }
                _prepCv.notify_all();
            }
        }
        catch (system::Exception& e)  { error = e.getMessage(); }
        catch (std::exception& e)     { error = e.what(); }
        if (!error.empty())  { { std::lock_guard<std::mutex> guard (_prepLock);  _prepError = "device sink (dataset preparation): " + error; }  _prepCv.notify_all(); }
    }
    std::mutex _prepLock;  std::condition_variable _prepCv;
    std::vector<Prepared> _prepared;  std::string _prepError;  bool _prepActive = false;
    PrepareFn  _prepareFn;
    std::thread _preparer;
    std::mutex _ringLock;  std::condition_variable _ringCv, _jobCv, _idleCv;
    void*     _ring = 0;
    std::vector<void*> _ringFree;
    std::deque<WriteJob> _jobs;  size_t _jobsPending = 0;  bool _writerStop = false;  std::string _writerError;
    std::list<std::string> _paths;
    double    _writerBusyS = 0;  u_int64_t _writerBytes = 0;       /* (written by the writer thread, read after drainWriter) */
    std::thread _writer;
};

/********************************************************************************/
/** Functor for Dispatcher::iterate over the sequences of the bank: one COPY per worker thread (ICommand.hpp:291-335 news a copy per thread and deletes it
 *  when the thread is done), each with its own packing buffers; only a full buffer takes the lock (gkc_push_reads: one thread drives the context at a time). */
template<size_t span>
class FillPartitionsDevice
{
public:
    struct Shared
    {
        std::mutex  lock;
        BankStats   stats;
        std::string error;          /**< first failure of a push (a destructor must not throw) */
    };
    enum { BLOCK_BYTES = 1 << 26 };

    FillPartitionsDevice (Shared& shared, gatb::core::tools::dp::IteratorListener* progress, size_t kmerSize)
        : _shared(shared), _progress(progress), _kmerSize(kmerSize), _nbWritten(0),
          _ranks (DeviceSession::singleton().ranks()), _rank (DeviceSession::singleton().rank())  { _offsets.push_back (0); }

    FillPartitionsDevice (const FillPartitionsDevice& o)
        : _shared(o._shared), _progress(o._progress), _kmerSize(o._kmerSize), _nbWritten(0), _ranks(o._ranks), _rank(o._rank)  { _offsets.push_back (0); }

    ~FillPartitionsDevice ()  { flush(); }

    void operator() (bank::Sequence& sequence)
    {
        if (_ranks > 1  &&  (int)(sequence.getIndex() % (size_t)_ranks) != _rank)  { return; }      /* another rank's read */
        const size_t len = sequence.getDataSize();
        _stats.update (sequence);
        _bases.insert (_bases.end(), sequence.getDataBuffer(), sequence.getDataBuffer() + len);
        _offsets.push_back (_bases.size());
        if (_bases.size() >= (size_t)BLOCK_BYTES)  { flush(); }
        if (_nbWritten++ > 500000)  { const u_int64_t n = DeviceSession::singleton().reportable (_nbWritten);  if (n > 0) { _progress->inc (n); }  _nbWritten = 0; }
    }

private:
    /** hands what this thread has packed to Stage A */
    void flush ()
    {
        if (_offsets.size() <= 1)  { return; }
        std::lock_guard<std::mutex> guard (_shared.lock);
        _shared.stats += _stats;  _stats = BankStats();
        try  {  if (_shared.error.empty())  { DeviceSession::singleton().push (_bases.data(), _offsets.data(), _offsets.size()-1); }  }
        catch (system::Exception& e)  { _shared.error = e.getMessage(); }
        _bases.clear();  _offsets.assign (1, 0);
    }

    Shared& _shared;
    gatb::core::tools::dp::IteratorListener* _progress;
    size_t _kmerSize;
    size_t _nbWritten;
    int    _ranks, _rank;
    std::vector<char> _bases;  std::vector<uint64_t> _offsets;  BankStats _stats;
};

/********************************************************************************/
/** Counting of one partition on the device (the third sibling of PartitionsByHashCommand / PartitionsByVectorCommand). */
template<size_t span>
class PartitionsByDeviceCommand : public PartitionsCommand<span>
{
public:
    typedef typename Kmer<span>::Type           Type;
    typedef typename Kmer<span>::Count          Count;
    typedef ICountProcessor<span>               CountProcessor;

    PartitionsByDeviceCommand (
        CountProcessor*                                 processor,
        size_t                                          cacheSize,
        gatb::core::tools::dp::IteratorListener*        progress,
        tools::misc::impl::TimeInfo&                    timeInfo,
        PartiInfo<5>&                                   pInfo,
        int                                             passi,
        int                                             parti,
        size_t                                          nbCores,
        size_t                                          kmerSize,
        gatb::core::tools::misc::impl::MemAllocator&    pool,
        tools::storage::impl::SuperKmerBinFiles*        superKstorage
    )
        : PartitionsCommand<span> (processor, cacheSize, progress, timeInfo, pInfo, passi, parti, nbCores, kmerSize, pool, superKstorage)  {}

    const char* getName() const { return "device"; }

    /** The default chain (SortingCountAlgorithm.cpp:376-400) with one abundance range (one bank, sum solidity, no auto cut-off) is what the device itself
     *  computes — histogram of every distinct k-mer, solidity window, ascending Count records — so the per-k-mer virtual process() calls
     *  (CountProcessorChain.hpp:128-135 -> CountProcessorDump.hpp:148-152 -> BagCache) can be replaced by one block insert per partition. */
    static DeviceBulkPlan bulkPlan (CountProcessor* processor, size_t nbProcessors, const Configuration& config)
    {
        DeviceBulkPlan plan;
        if (processor == 0  ||  nbProcessors != 1  ||  config._abundance.size() != 1  ||  config._abundance[0].getBegin() < 1  ||  getenv ("GATB_DEVICE_NO_BULK") != 0)  { return plan; }
        std::vector<CountProcessor*> items = processor->getInstances();
        if (items.size() != 3)  { return plan; }
        CountProcessorHistogram<span>*   histo = dynamic_cast<CountProcessorHistogram<span>*>   (items[0]);
        CountProcessorSoliditySum<span>* solid = dynamic_cast<CountProcessorSoliditySum<span>*> (items[1]);
        CountProcessorDump<span>*        dump  = dynamic_cast<CountProcessorDump<span>*>        (items[2]);
        if (histo == 0  ||  solid == 0  ||  dump == 0  ||  histo->getHistogram() == 0)  { return plan; }
        plan.on = true;
        if (dump->getSolidCounts() != 0)  { SolidSinkDirect<Count>::closeHandles (*dump->getSolidCounts()); }      /* (what the direct sink needs of the partition before the commands run) */
        {
            tools::storage::impl::Partition<Count>* solids = dump->getSolidCounts();
            const size_t nbPartitions = config._nb_partitions;  const size_t recBytes = config._kmerSize <= 31 ? 16 : 32;
            if (solids != 0  &&  sizeof(Count) == recBytes)
                DeviceSession::singleton().setPreparer ([solids, nbPartitions] (size_t part, size_t pass, uint64_t n, uint64_t& address, std::string& path)
                    { return SolidSinkDirect<Count>::prepare ((*solids) [part + pass * nbPartitions], (size_t) n, address, path); });
            else
                DeviceSession::singleton().setPreparer (DeviceSession::PrepareFn());
        }
        plan.abundanceMin = (int32_t) config._abundance[0].getBegin();
        plan.abundanceMax = (int32_t) config._abundance[0].getEnd();
        plan.histoMax     = (uint32_t) histo->getHistogram()->getLength();
        return plan;
    }

    /** bulk mode, main thread, after the last pass: the device's abundance histogram into the prototype's (what the clones' HistogramCache instances
     *  would have merged, Histogram.hpp:221-238: bins 1 .. length-1) */
    static void mergeDeviceHistogram (CountProcessor* processor)
    {
        DeviceSession& dev = DeviceSession::singleton();
        if (!dev.plan().on)  { return; }
        std::vector<CountProcessor*> items = processor->getInstances();
        CountProcessorHistogram<span>* histo = items.empty() ? 0 : dynamic_cast<CountProcessorHistogram<span>*> (items[0]);
        if (histo == 0)  { return; }
        tools::misc::IHistogram* h = histo->getHistogram();
        std::vector<uint64_t> dh (dev.plan().histoMax + 1);
        dev.check (gkc_histogram (dev.ctx(), dh.data(), (uint32_t) dh.size()));
        for (size_t cc = 1; cc < h->getLength(); cc++)  { h->get (cc) += dh[cc]; }
        /* ... and the solidity processor's counters (kmers_nb_distinct / kmers_nb_solid of the run's properties): what its clones would have counted one
         * process() at a time comes from the device's statistics, through the same finishClones() the clones go through (CountProcessorSolidity.hpp:118-131) */
        CountProcessorSoliditySum<span>* solid = items.size() > 1 ? dynamic_cast<CountProcessorSoliditySum<span>*> (items[1]) : 0;
        if (solid != 0)
        {
            struct Totals : public CountProcessorSoliditySum<span>
            {
                Totals (u_int64_t total, u_int64_t ok)  { this->_total = total;  this->_ok = ok; }
            };
            gkc_stats st;  dev.check (gkc_get_stats (dev.ctx(), &st));
            Totals totals (st.kmers_nb_distinct, st.kmers_nb_solid);
            std::vector<ICountProcessor<span>*> one (1, &totals);
            solid->finishClones (one);
        }
    }

    void execute ()
    {
        DeviceSession& dev = DeviceSession::singleton();

        this->_processor->beginPart (this->_pass_num, this->_parti_num, this->_cacheSize, this->getName());

        /* the device's Count records of this partition (ascending), as soon as Stage B has produced them */
        const void* landed = 0;  uint64_t nbSolid = 0;
        const double t0 = now();
        dev.check (gkc_wait_partition (dev.ctx(), this->_pass_num, this->_parti_num, &landed, &nbSolid));
        const double t1 = now();

        CountProcessorDump<span>* dump = 0;
        if (dev.plan().on)
        {
            std::vector<CountProcessor*> items = this->_processor->getInstances();
            if (items.size() == 3)  { dump = dynamic_cast<CountProcessorDump<span>*> (items[2]); }
        }

        /* the device record width follows k (16 bytes for k <= 31, 32 bytes above) */
        const size_t recBytes = this->_kmerSize <= 31 ? 16 : 32;
        const size_t actualPartId = this->_parti_num + this->_pass_num * dev.nbPartitions();      /* CountProcessorDump.hpp:131 */
        const bool   bulk = dump != 0  &&  dump->getSolidCounts() != 0;
        double fetchS = 0;
        bool   written = false;

        /* Bulk mode, HDF5 storage, the device layout IS Abundance<Type,int32> (Abundance.hpp:68-129): the partition's dataset is made at its final size and the
         * records go from the device through the page-locked slots of the session's ring to the file's writer thread (SolidSinkDirect.hpp, DeviceSession::prepareRing) —
         * no block of the partition's size on the host, no H5Dwrite under the storage's lock */
        if (bulk  &&  landed == 0  &&  nbSolid > 0  &&  sizeof(Count) == recBytes  &&  dev.hasRing())
        {
            uint64_t address = 0;  std::string path;
            /* (the dataset: made ahead by the session's preparer thread, else here) */
            const bool direct = dev.preparerActive() ? dev.waitPrepared (this->_parti_num, address, path)
                                                     : SolidSinkDirect<Count>::prepare ((*dump->getSolidCounts()) [actualPartId], (size_t) nbSolid, address, path);
            if (direct)
            {
                const uint64_t per = (uint64_t) DeviceSession::SLOT_BYTES / recBytes;
                for (uint64_t first = 0; first < nbSolid; first += per)
                {
                    const uint64_t n = std::min<uint64_t> (per, nbSolid - first);
                    struct Slot  { DeviceSession& dev;  void* p;  Slot (DeviceSession& d) : dev(d), p(d.takeSlot()) {}  ~Slot ()  { if (p) { dev.giveSlot (p); } } }  slot (dev);
                    const double f0 = now();
                    dev.check (gkc_partition_counts_range (dev.ctx(), this->_pass_num, this->_parti_num, first, n, slot.p));
                    fetchS += now() - f0;
                    dev.queueWrite (path, address + first * recBytes, slot.p, (size_t) (n * recBytes));
                    slot.p = 0;                                          /* (the writer gives it back) */
                }
                written = true;
            }
        }

        std::unique_ptr<unsigned char[]> fetched;                 /* (not a vector: no zero-fill of a block that is overwritten at once) */
        const unsigned char* recs = (const unsigned char*) landed;
        if (!written  &&  recs == 0  &&  nbSolid > 0)
        {
            fetched.reset (new unsigned char [nbSolid * recBytes]);
            uint64_t got = 0;
            const double f0 = now();
            dev.check (gkc_partition_counts (dev.ctx(), this->_pass_num, this->_parti_num, fetched.get(), nbSolid, &got));
            fetchS += now() - f0;
            recs = fetched.get();
        }
        const double t2 = t1 + fetchS;

        if (bulk)
        {
            /* one block per partition into the partition's collection: the direct sink when the records are in host memory already (a host sink is set) or the ring
             * is absent; otherwise the collection's own insert (HDF5: one H5Dwrite under the storage's lock; file storage: BagFile) */
            tools::collections::Collection<Count>& coll = (*dump->getSolidCounts()) [actualPartId];
            if (!written  &&  nbSolid > 0)
            {
                if (sizeof(Count) == recBytes)  { if (!SolidSinkDirect<Count>::insert (coll, (const Count*) recs, (size_t) nbSolid))  { coll.insert ((const Count*) recs, (size_t) nbSolid); } }
                else
                {
                    std::vector<Count> block (nbSolid);
                    for (uint64_t i = 0; i < nbSolid; i++)  { decode (recs + i * recBytes, recBytes, block[i].value, block[i].abundance); }
                    coll.insert (block.data(), block.size());
                }
            }
        }
        else
        {
            CounterBuilder solidCounter;
            for (uint64_t i = 0; i < nbSolid; i++)
            {
                Type kmer;  CountNumber ab;
                decode (recs + i * recBytes, recBytes, kmer, ab);
                solidCounter.set (ab);
                this->insert (kmer, solidCounter);
            }
        }

        this->_progress->inc (this->_pInfo.getNbKmer (this->_parti_num));
        this->_processor->endPart (this->_pass_num, this->_parti_num);
        dev.addCommandTimes (t1 - t0, now() - t1);
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
        {
            static std::mutex mu;  static double waited = 0, fetchedS = 0, handed = 0;  static size_t done = 0;
            std::lock_guard<std::mutex> guard (mu);
            waited += t1 - t0;  fetchedS += t2 - t1;  handed += now() - t2;  done++;
            if (done % dev.nbPartitions() == 0)
            {
                fprintf (stderr, "[device counting] pass %d, summed over the partition commands: waiting for Stage B %.2f s, Count[] to the host %.2f s, hand-over to the processors %.2f s\n",
                         (int)this->_pass_num, waited, fetchedS, handed);
                waited = fetchedS = handed = 0;
            }
        }
    }

private:
    static double now ()  { struct timespec ts;  clock_gettime (CLOCK_MONOTONIC, &ts);  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    static void decode (const unsigned char* r, size_t recBytes, Type& kmer, CountNumber& abundance)
    {
        if (recBytes == 16)  {  kmer.setVal (*(const u_int64_t*) r);  }
        else                 {  setWide (kmer, ((const u_int64_t*) r)[0], ((const u_int64_t*) r)[1]);  }
        abundance = *(const int32_t*) (r + (recBytes == 16 ? 8 : 16));
    }

    /* 128-bit value into a Type of 2+ words (LargeInt<2..4>); spans of one word never get here (k <= 31) */
    template<typename T> static void setWide (T& kmer, u_int64_t lo, u_int64_t hi)
    {
        kmer.setVal (hi);  kmer <<= 32;  kmer <<= 32;  T low;  low.setVal (lo);  kmer += low;
    }
};

/********************************************************************************/
} } } } /* end of namespaces. */
/********************************************************************************/

#endif /* _GATB_CORE_KMER_IMPL_DEVICE_COUNTING_HPP_ */
