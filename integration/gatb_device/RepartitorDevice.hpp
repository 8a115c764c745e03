/* RepartitorDevice.hpp — SURVEY 8(f)2 inside the reference: the two sampling iterations of RepartitorAlgorithm run on the device.
 *
 * What the reference does (kmer/impl/RepartitionAlgorithm.cpp): before the first pass, ONE thread (SerialDispatcher, :348 and :458) walks the head of the bank
 *   computeFrequencies  (:311-384, only with -minimizer-type 1 — what GraphUnitigs forces, GraphUnitigs.cpp:861-870): the canonical m-mers of 5 % of the sequences
 *                       (at most 5e7, :322) are counted by the MmersFrequency functor (:88-150); the ranking of the counts is the minimizer order of the run;
 *   computeRepartition  (:395-475): SampleRepart (:156-229) splits sequences into super-k-mers until more than max(5 % of the sequences, 1e6) of them have been
 *                       seen and counts super-k-mers, k-mers and kx-mers per minimizer value; justGroup / computeDistrib build the partition table from that.
 * At 10^8 reads of 150 bp that is 7e8 m-mers (4.9e6 sequences) and 4.9e6 super-k-mers (4.6e5 sequences) through scalar code on one core, beside a counting step of
 * 3-4 s for the whole bank: profiles/r05_repartitor_1e8reads.txt — dbgh5's wall outside the DSK step 3.4 s in frequency mode with the reference's functors, 1.4 s
 * with the device (0.49 + 0.23 s in the two calls below, most of it the bank iterator); 1.5 -> 0.9 s in the default (lexicographic) mode.
 *
 * Here the sequences of the sample are still read by the reference's own bank iterator (any bank: FASTA, FASTQ, gz, album), packed into flat blocks, and the
 * counting itself is the device's: gkc_count_mmers (the MmersFrequency functor) and gkc_sample_exact (SampleRepart, with the reference's stop rule: the cancel
 * flag is looked at between sequences, so the sequence in which the running number of super-k-mers first EXCEEDS the threshold is still counted). The tables are
 * built from the statistics by the reference's own code, unchanged: /minimizers/minimRepart and minimFrequency come out byte for byte (tests/test_gpu_dropin.py).
 * Both return false when the device is not used (no device, GATB_DEVICE_NO_REPARTITOR=1, k or m outside the device's range): the reference's iteration runs.
 */
#ifndef _GATB_CORE_KMER_IMPL_DEVICE_REPARTITOR_HPP_
#define _GATB_CORE_KMER_IMPL_DEVICE_REPARTITOR_HPP_

#include <gatb/bank/api/IBank.hpp>
#include <gatb/bank/api/Sequence.hpp>
#include <gatb/kmer/impl/Configuration.hpp>
#include <gatb/kmer/impl/PartiInfo.hpp>
#include <gatb/system/api/Exception.hpp>
#include <gatb/system/api/ISmartPointer.hpp>
#include <gatb/tools/designpattern/api/Iterator.hpp>

#include <gatb_device/DeviceContext.hpp>

#include <algorithm>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

namespace gatb { namespace core { namespace kmer { namespace impl {

class RepartitorDevice
{
public:
    /** MmersFrequency over the head of the bank (RepartitionAlgorithm.cpp:341-350): counts[v] = occurrences of the canonical m-mer v at valid positions of the
     *  first nbSeqsToSee + 1 sequences (the functor raises the cancel flag once it has seen MORE than nbSeqsToSee, :113-117). counts: 4^m entries, overwritten. */
    static bool countMmers (bank::IBank* bank, const Configuration& config, u_int64_t nbSeqsToSee, uint32_t* counts)
    {
        gkc_ctx* ctx = context (config);
        if (ctx == 0)  { return false; }
        const double t0 = now();
        const u_int64_t nbMmers = (u_int64_t)1 << (2 * config._minim_size);
        std::fill (counts, counts + nbMmers, (uint32_t)0);
        Block block;  u_int64_t nbSeen = 0;
        tools::dp::Iterator<bank::Sequence>* it = bank->iterator();  LOCAL (it);
        for (it->first();  !it->isDone()  &&  nbSeen <= nbSeqsToSee;  it->next())
        {
            block.add (it->item());  nbSeen++;
            if (block.full())  { check (ctx, gkc_count_mmers (ctx, (uint32_t) config._minim_size, block.bases.data(), block.offsets.data(), block.size(), counts));  block.clear(); }
        }
        if (block.size() > 0)  { check (ctx, gkc_count_mmers (ctx, (uint32_t) config._minim_size, block.bases.data(), block.offsets.data(), block.size(), counts)); }
        it->finalize();
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
            fprintf (stderr, "[device repartitor] m-mer frequencies of the first %llu sequences counted on the device (gkc_count_mmers): %.3f s\n", (unsigned long long) nbSeen, now() - t0);
        return true;
    }

    /** SampleRepart over the head of ONE bank (RepartitionAlgorithm.cpp:441-468): per minimizer value the super-k-mers, k-mers and kx-mers of the sequences up to
     *  and including the one in which the number of super-k-mers seen first exceeds nbSeqsToSee (:205-212), under the minimizer order of the run
     *  (freqOrder == 0: the lexicographic / KMC2 order). SampleRepart is a Sequence2SuperKmer of ONE pass (:225): every super-k-mer counts, whatever
     *  config._nb_passes says. Added to `info` as PartiInfo::incSuperKmer_per_minimBin / incKxmer_per_minimBin would have. */
    static bool sample (bank::IBank* bank, const Configuration& config, const uint32_t* freqOrder, u_int64_t nbSeqsToSee, PartiInfo<5>& info)
    {
        gkc_ctx* ctx = context (config);
        if (ctx == 0)  { return false; }
        const double t0 = now();
        const u_int64_t nbMinims = (u_int64_t)1 << (2 * config._minim_size);
        std::vector<uint16_t> onePartition (nbMinims, 0);
        check (ctx, gkc_configure (ctx, (uint32_t) config._kmerSize, (uint32_t) config._minim_size, 1, 1, freqOrder != 0 ? GKC_MINIMIZER_FREQ : GKC_MINIMIZER_LEXI,
                                   onePartition.data(), freqOrder));
        std::vector<uint64_t> nbSuperKmers (nbMinims, 0), nbKmers (nbMinims, 0), nbKxmers (nbMinims, 0);
        Block block;  u_int64_t seen = 0, nbSequences = 0;  bool stop = false;
        tools::dp::Iterator<bank::Sequence>* it = bank->iterator();  LOCAL (it);
        it->first();
        while (!stop)
        {
            for ( ;  !it->isDone()  &&  !block.full();  it->next())  { block.add (it->item()); }
            if (block.size() == 0)  { break; }
            /* the threshold left for this block: the sample stops in the sequence where seen + (super-k-mers of the block so far) > nbSeqsToSee */
            uint64_t used = 0;
            check (ctx, gkc_sample_exact (ctx, block.bases.data(), block.offsets.data(), block.size(), nbSeqsToSee - seen,
                                          nbSuperKmers.data(), nbKmers.data(), nbKxmers.data(), &used));
            nbSequences += used;
            stop = used < block.size();
            if (!stop)
            {
                seen = 0;  for (u_int64_t i = 0; i < nbMinims; i++)  { seen += nbSuperKmers[i]; }
                stop = seen > nbSeqsToSee;          /* (crossed in the last sequence of the block) */
            }
            block.clear();
        }
        it->finalize();
        for (u_int64_t i = 0; i < nbMinims; i++)
        {
            if (nbSuperKmers[i] == 0)  { continue; }
            if (nbKmers[i] > (uint64_t) 0x7FFFFFFF)  { throw system::Exception ("device repartitor: %llu k-mers under one minimizer in the sample", (unsigned long long) nbKmers[i]); }
            info.incSuperKmer_per_minimBin ((int) i, (int) nbKmers[i], 1);           /* one super-k-mer carrying the bin's k-mers ... */
            if (nbSuperKmers[i] > 1)  { info.incSuperKmer_per_minimBin ((int) i, 0, nbSuperKmers[i] - 1); }      /* ... and the others */
            info.incKxmer_per_minimBin ((int) i, nbKxmers[i]);
        }
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
        {
            u_int64_t total = 0;  for (u_int64_t i = 0; i < nbMinims; i++)  { total += nbSuperKmers[i]; }
            fprintf (stderr, "[device repartitor] %llu super-k-mers of the first %llu sequences sampled on the device (gkc_sample_exact): %.3f s\n",
                     (unsigned long long) total, (unsigned long long) nbSequences, now() - t0);
        }
        return true;
    }

private:
    /** sequences packed the way gkc_push_reads takes them: flat bases + offsets */
    struct Block
    {
        enum { MAX_BYTES = 1 << 26, MAX_SEQUENCES = 1 << 18 };
        std::vector<char> bases;  std::vector<uint64_t> offsets;
        Block ()  { offsets.push_back (0); }
        void   add   (bank::Sequence& s)  { bases.insert (bases.end(), s.getDataBuffer(), s.getDataBuffer() + s.getDataSize());  offsets.push_back (bases.size()); }
        size_t size  () const  { return offsets.size() - 1; }
        bool   full  () const  { return bases.size() >= (size_t) MAX_BYTES  ||  size() >= (size_t) MAX_SEQUENCES; }
        void   clear ()  { bases.clear();  offsets.assign (1, 0); }
    };

    static gkc_ctx* context (const Configuration& config)
    {
        if (getenv ("GATB_DEVICE_NO_REPARTITOR") != 0)  { return 0; }
        if (config._kmerSize > 63  ||  config._minim_size < 2  ||  config._minim_size > 14  ||  config._minim_size >= config._kmerSize)  { return 0; }
        return device::DeviceContext::singleton().ctx();
    }
    static void check (gkc_ctx* ctx, int rc)
    {
        if (rc != GKC_OK)  { throw system::Exception ("device repartitor: error %d: %s", rc, gkc_last_error (ctx)); }
    }
    static double now ()  { struct timespec ts;  clock_gettime (CLOCK_MONOTONIC, &ts);  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
};

} } } } /* end of namespaces. */

#endif /* _GATB_CORE_KMER_IMPL_DEVICE_REPARTITOR_HPP_ */
