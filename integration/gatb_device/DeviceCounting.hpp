/* DeviceCounting.hpp — the reference-side binding of libgkc_hip.so: what a GATB-Core maintainer adds to
 * src/gatb/kmer/impl/ so that SortingCountAlgorithm<span> counts on an MI355X. Compiled against the reference's own headers by
 * integration/check_integration.sh (it is NOT part of libgkc_hip.so and includes nothing of this repository but include/gkc.h).
 *
 *   DeviceSession               one gkc_ctx per process (one GPU per process), configured from the reference's Configuration + Repartitor
 *   FillPartitionsDevice<span>  the functor fillPartitions() iterates the bank with (SortingCountAlgorithm.cpp:1081-1151 FillPartitions,
 *                               dispatched at :1266-1275): packs sequences into a flat ASCII buffer + CSR offsets and hands them to
 *                               gkc_push_reads (Stage A on the device) instead of cutting super-k-mers on the CPU
 *   PartitionsByDeviceCommand   the ICommand fillSolidKmers_aux() dispatches per partition (SortingCountAlgorithm.cpp:1456-1587), beside
 *                               PartitionsByVectorCommand / PartitionsByHashCommand (PartitionsCommand.hpp:100-160): waits for the device's
 *                               Count[] of its partition and feeds the CountProcessor clone in ascending k-mer order
 */
#ifndef _GATB_CORE_KMER_IMPL_DEVICE_COUNTING_HPP_
#define _GATB_CORE_KMER_IMPL_DEVICE_COUNTING_HPP_

#include <gatb/kmer/impl/PartitionsCommand.hpp>
#include <gatb/kmer/impl/PartiInfo.hpp>
#include <gatb/kmer/impl/BankKmers.hpp>
#include <gatb/kmer/impl/Configuration.hpp>
#include <gatb/bank/api/Sequence.hpp>
#include <gatb/system/api/Exception.hpp>
#include <gatb/system/impl/System.hpp>

#include <gkc.h>

#include <vector>
#include <mutex>
#include <string>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

namespace gatb { namespace core { namespace kmer { namespace impl {

/********************************************************************************/
/** One device context per process. */
class DeviceSession
{
public:
    static DeviceSession& singleton ()  { static DeviceSession s; return s; }

    gkc_ctx* ctx ()  {  open();  return _ctx;  }

    /** Configuration + Repartitor of the run -> gkc_configure (replaces `Model model(...)`, SortingCountAlgorithm.cpp:1251-1256, and
     *  Repartitor::operator(), PartiInfo.hpp:323). The solidity window stays open: the reference's processor chain filters. */
    void configure (const Configuration& config, Repartitor& repartitor)
    {
        open();
        enableMultiGpuFromEnvironment();
        const u_int64_t nbMinims = (u_int64_t)1 << (2 * config._minim_size);
        std::vector<uint16_t> table (nbMinims);
        for (u_int64_t m = 0; m < nbMinims; m++)  { table[m] = (uint16_t) repartitor (m); }
        check (gkc_set_solidity (_ctx, 1, 2147483647, 10000));      /* the histogram is the CountProcessorHistogram's job */
        check (gkc_configure (_ctx, (uint32_t) config._kmerSize, (uint32_t) config._minim_size, (uint32_t) config._nb_partitions, (uint32_t) config._nb_passes,
                              config._minimizerType == 1 ? GKC_MINIMIZER_FREQ : GKC_MINIMIZER_LEXI, table.data(),
                              config._minimizerType == 1 ? repartitor.getMinimizerFrequencies() : 0));
    }

    void check (int rc)
    {
        if (rc != GKC_OK)  { throw system::Exception ("device counting: error %d: %s", rc, gkc_last_error(_ctx)); }
    }

    /** Multi-GPU (one process per GPU): GATB_DEVICE_RANKS / GATB_DEVICE_RANK / GATB_DEVICE_COMM_ID (a file: rank 0 writes the 128-byte
     *  ncclUniqueId, the other ranks wait for it) in the environment of each process turn the session into one rank of an RCCL communicator. Each
     *  process reads ITS share of the reads; after every block pushed to Stage A the super-k-mers are routed to the rank owning their partition
     *  (gkc_exchange), and each process counts — and writes into its own .h5 — the partitions it owns (the other datasets stay empty). */
    void enableMultiGpuFromEnvironment ()
    {
        const char* ranks = getenv ("GATB_DEVICE_RANKS");  const char* rank = getenv ("GATB_DEVICE_RANK");  const char* idFile = getenv ("GATB_DEVICE_COMM_ID");
        if (_comm != 0  ||  ranks == 0  ||  rank == 0  ||  idFile == 0  ||  atoi(ranks) < 2)  { return; }
        open();
        uint8_t id [GKC_COMM_ID_BYTES];
        if (atoi(rank) == 0)
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
        check (gkc_comm_create_rccl (_ctx, id, atoi(ranks), atoi(rank), &_comm));
    }
    gkc_comm* comm ()  { return _comm; }
    /** collective, ONCE per pass on every rank, after the rank's last push: every block pushed so far goes to the owners of its partitions */
    void exchange ()  { if (_comm != 0) { check (gkc_exchange (_ctx, _comm)); } }

    ~DeviceSession ()  { if (_comm) { gkc_comm_destroy (_comm); }  if (_ctx) { gkc_destroy (_ctx); } }

private:
    DeviceSession () : _ctx(0), _comm(0) {}
    void open ()
    {
        if (_ctx == 0  &&  gkc_create (0, &_ctx) != GKC_OK)  { throw system::Exception ("device counting: %s", gkc_last_error(0)); }
    }
    gkc_ctx* _ctx;
    gkc_comm* _comm;
};

/********************************************************************************/
/** Functor for Dispatcher::iterate over the sequences of the bank (one instance per thread, copies share the packer). */
template<size_t span>
class FillPartitionsDevice
{
public:
    struct Packer
    {
        std::vector<char> bases;  std::vector<uint64_t> offsets;  std::mutex lock;
        BankStats stats;
        Packer ()  { offsets.push_back (0); }
        /** hands what has been packed to Stage A; called with the lock held, or at the end */
        void flush ()
        {
            if (offsets.size() > 1)
            {
                DeviceSession::singleton().check (gkc_push_reads (DeviceSession::singleton().ctx(), bases.data(), offsets.data(), offsets.size()-1));
                bases.clear();  offsets.assign (1, 0);
            }

        }
    };

    FillPartitionsDevice (Packer& packer, gatb::core::tools::dp::IteratorListener* progress, size_t kmerSize)
        : _packer(packer), _progress(progress), _kmerSize(kmerSize), _nbWritten(0)  {}

    void operator() (bank::Sequence& sequence)
    {
        const size_t len = sequence.getDataSize();
        std::lock_guard<std::mutex> guard (_packer.lock);
        _packer.stats.update (sequence);
        _packer.bases.insert (_packer.bases.end(), sequence.getDataBuffer(), sequence.getDataBuffer() + len);
        _packer.offsets.push_back (_packer.bases.size());
        if (_packer.bases.size() >= ((size_t)1 << 28))  { _packer.flush(); }
        if (_nbWritten++ > 500000)  { _progress->inc (_nbWritten);  _nbWritten = 0; }
    }

private:
    Packer& _packer;
    gatb::core::tools::dp::IteratorListener* _progress;
    size_t _kmerSize;
    size_t _nbWritten;
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

    void execute ()
    {
        DeviceSession& dev = DeviceSession::singleton();

        this->_processor->beginPart (this->_pass_num, this->_parti_num, this->_cacheSize, this->getName());

        /* the device's Count records of this partition (ascending), as soon as Stage B has produced them */
        const void* landed = 0;  uint64_t nbSolid = 0;
        dev.check (gkc_wait_partition (dev.ctx(), this->_pass_num, this->_parti_num, &landed, &nbSolid));

        /* the device record width follows k (16 bytes for k <= 31, 32 bytes above) */
        const size_t recBytes = this->_kmerSize <= 31 ? 16 : 32;
        std::vector<unsigned char> fetched;
        const unsigned char* recs = (const unsigned char*) landed;
        if (recs == 0  &&  nbSolid > 0)
        {
            fetched.resize (nbSolid * recBytes);
            uint64_t got = 0;
            dev.check (gkc_partition_counts (dev.ctx(), this->_pass_num, this->_parti_num, fetched.data(), nbSolid, &got));
            recs = fetched.data();
        }

        CounterBuilder solidCounter;
        for (uint64_t i = 0; i < nbSolid; i++)
        {
            const unsigned char* r = recs + i * recBytes;
            Type kmer;
            if (recBytes == 16)  {  kmer.setVal (*(const u_int64_t*) r);  }
            else                 {  setWide (kmer, ((const u_int64_t*) r)[0], ((const u_int64_t*) r)[1]);  }
            solidCounter.set (*(const int32_t*) (r + (recBytes == 16 ? 8 : 16)));
            this->insert (kmer, solidCounter);
        }

        this->_progress->inc (this->_pInfo.getNbKmer (this->_parti_num));
        this->_processor->endPart (this->_pass_num, this->_parti_num);
    }

private:
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
