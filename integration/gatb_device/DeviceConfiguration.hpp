/* DeviceConfiguration.hpp — the GPU-aware part of ConfigurationAlgorithm<span>::execute (kmer/impl/ConfigurationAlgorithm.cpp:398-425), reference-side binding.
 *
 * The reference sizes the partitions so that `nb_partitions_in_parallel` of them fit `-max-memory` of HOST RAM at a time, and the passes so that the partition
 * files fit the disk (:398-425). With device counting neither is what bounds a run: the super-k-mers and the counts live in HBM and nothing is written to disk.
 *   nb_passes      from the device's memory: a pass keeps its super-k-mer records (~1.5 bytes per k-mer) and leaves its Count records (16 / 32 bytes per distinct
 *                  k-mer, ~0.6 per k-mer assumed) in HBM; 75 % of the device total is planned with (the total, not what happens to be free, so that the
 *                  ranks of a multi-GPU run derive the same Configuration: gkc_exchange compares it);
 *   nb_partitions  about 3e6 k-mers each (1.5e6 with 16-byte keys): what a workgroup's LDS tables (8192 sub-buckets) and a wave's registers (~350 keys per
 *                  sub-bucket) are cut for — a partition ten times that size puts most of its sub-buckets on the device's split path and leaves the
 *                  one-workgroup-per-partition kernels with a quarter of the chip (256 partitions of 3.4e7 k-mers: Stage B 4x slower);
 *   every core hands over a partition at a time (`nb_partitions_in_parallel` = cores): a PartitionsByDeviceCommand only waits for Stage B and gives its block of
 *                  records to the processors, host memory per command is one block.
 * Like in the reference, the layout of the .h5 (number of /dsk/solid/<p> datasets, the Repartitor table) follows from the Configuration; the k-mer SET and the
 * counts do not depend on it. GATB_DEVICE_REFERENCE_CONFIG=1 keeps what the reference derived from host RAM and disk (the tests that compare datasets with
 * runs of the unpatched reference use it, with the -max-memory / -nb-cores of those runs). */
#ifndef _GATB_CORE_KMER_IMPL_DEVICE_CONFIGURATION_HPP_
#define _GATB_CORE_KMER_IMPL_DEVICE_CONFIGURATION_HPP_

#include <gatb/kmer/impl/Configuration.hpp>

#include <gatb_device/DeviceContext.hpp>

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>

namespace gatb { namespace core { namespace kmer { namespace impl {

struct DeviceConfiguration
{
    /** \param[in] typeBytes : sizeof(Kmer<span>::Type).  \return true when the Configuration was re-derived for the device */
    static bool apply (Configuration& config, size_t typeBytes)
    {
        if (getenv ("GATB_DEVICE_REFERENCE_CONFIG") != 0)  { return false; }
        gkc_ctx* ctx = device::DeviceContext::singleton().ctx();
        if (ctx == 0)  { return false; }
        uint64_t usable = 0, total = 0;
        if (gkc_device_memory (ctx, &usable, &total) != GKC_OK  ||  total == 0)  { return false; }

        const bool   wide       = config._kmerSize > 31;
        const double countBytes = wide ? 32.0 : 16.0;
        const double perKmer    = 1.5 + 0.6 * countBytes;                       /* HBM a pass holds per k-mer: records + the Count records it leaves */
        const char*  ranksEnv   = getenv ("GATB_DEVICE_RANKS");
        const double ranks      = ranksEnv != 0 && atoi (ranksEnv) > 1 ? (double) atoi (ranksEnv) : 1.0;
        const double kmers      = (double) config._kmersNb;
        const double budget     = 0.75 * (double) total;

        size_t passes = (size_t) (kmers / ranks * perKmer / budget) + 1;
        const double target = wide ? 1.5e6 : 3.0e6;                             /* k-mers per partition */
        double parts = kmers / (double) passes / target;
        /* up to 4096 partitions Stage A buckets in one level; beyond, in two (groups of consecutive partitions first): whole powers of two of groups */
        size_t nbPartitions = parts < 1.0 ? 1 : (size_t) (parts + 0.5);
        while (nbPartitions > 32768)  { passes++;  nbPartitions = (size_t) (kmers / (double) passes / target + 0.5); }      /* (Repartitor::Value is 16 bits: PartiInfo.hpp:297) */
        if (nbPartitions < 1)  { nbPartitions = 1; }

        config._nb_passes                 = passes;
        config._nb_partitions             = nbPartitions;
        /* the partition commands of a group (one thread each, Command.cpp:130-167) only wait for the device and move records into the ring of the file's writer:
         * 32 keep it fed; 256 compete with it for the cores (fill_solid_kmers at 10^8 reads, 3884 partitions: 2.05-2.30 s against 2.36-2.56 s; 1024: 3.0 s) */
        config._nb_partitions_in_parallel = std::max<size_t> (1, std::min<size_t> (config._nbCores, 32));
        if (getenv ("GATB_DEVICE_PARTITIONS_IN_PARALLEL") != 0)  { config._nb_partitions_in_parallel = std::max (1, atoi (getenv ("GATB_DEVICE_PARTITIONS_IN_PARALLEL"))); }
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
        {
            fprintf (stderr, "[device configuration] %.3e k-mers, %.0f GB of HBM: %zu pass(es), %zu partitions of ~%.2e k-mers (the reference's host-memory rule is not applied)\n",
                     kmers, (double) total / 1e9, passes, nbPartitions, kmers / (double) passes / (double) nbPartitions);
        }
        return true;
    }
};

} } } } /* end of namespaces. */

#endif /* _GATB_CORE_KMER_IMPL_DEVICE_CONFIGURATION_HPP_ */
