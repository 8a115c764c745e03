#!/usr/bin/env python
"""Applies the device patch to scratch COPIES of the reference's sources (the reference tree is never written) and emits the unified diff a maintainer
would apply (integration/gatb-core.device.patch).

    python integration/make_patched_sources.py <reference gatb-core dir> <scratch include dir> [--write-patch]

Seven files are touched (file:line cited per hunk below); every edit is anchored on text of the reference file that must be found exactly once, and everything
new is guarded by GATB_WITH_DEVICE_COUNTING, so the patched files still build the CPU path without the macro:
  kmer/impl/SortingCountAlgorithm.cpp       fillPartitions -> Stage A on the device, the partition command, the join, device time keys in getInfo()
  kmer/impl/ConfigurationAlgorithm.cpp      partitions / passes sized from the HBM of the device instead of host RAM and disk
  tools/collections/impl/Bloom.hpp          BloomFactory::createBloom returns BloomDevice<T> for the item types / kinds the device knows
  kmer/impl/BloomAlgorithm.cpp              execute(): the solid k-mers are inserted where the counting step left them in HBM
  kmer/impl/MPHFAlgorithm.cpp               execute(): BooPHF built on the device, stored, loaded by the reference's own MapMPHF::load; populate() on the device
  kmer/impl/DebloomMinimizerAlgorithm.cpp   contains8 of a partition's solid k-mers as ONE batched device query instead of one call per k-mer
  kmer/impl/RepartitionAlgorithm.cpp        the two serial sampling iterations (m-mer frequencies, super-k-mer statistics) counted on the device
The scratch copies land under <scratch include dir>/gatb/..., which check_integration.sh puts in front of the reference's src/ on the include path."""
import difflib
import os
import sys

# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/SortingCountAlgorithm.cpp
# ------------------------------------------------------------------------------------------------------------------------------------------
REL = "src/gatb/kmer/impl/SortingCountAlgorithm.cpp"

SCA_INCLUDE_ANCHOR = "#include <gatb/kmer/impl/SortingCountAlgorithm.hpp>"
SCA_INCLUDE = "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/DeviceCounting.hpp>   /* MI355X back-end: libgkc_hip.so */\n#endif\n"

# fillPartitions (SortingCountAlgorithm.cpp:1266-1282): the CPU functor is replaced by the packer + Stage A on the device
FILL_BEGIN = "\t\t\tgetDispatcher()->iterate(\n\t\t\t\titSeq,\n\t\t\t\tFillPartitions<span, true>("
FILL_END = "\t\t\t\tgroupSize, deleteSynchro);\n\n\t\t\t// GR: close the input bank here with call to finalize\n\t\t\titSeq->finalize();\n"
FILL_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
\t\t\t/** Stage A on the device: every worker thread packs its sequences and pushes full blocks (gkc_push_reads); Stage B starts right away
\t\t\t *  (gkc_finish_pass_async) and the PartitionsByDeviceCommand instances of fillSolidKmers wait for their partition. Several ranks
\t\t\t *  (GATB_DEVICE_RANKS ...): the super-k-mers are routed to the rank owning their partition as the reads go by, and the results are
\t\t\t *  gathered on rank 0. */
\t\t\tDeviceSession& device = DeviceSession::singleton();
\t\t\tdevice.configure (_config, *_repartitor, PartitionsByDeviceCommand<span>::bulkPlan (_processors.empty() ? 0 : _processors[0], _processors.size(), _config), pass);
\t\t\tdevice.beginPass (pass);
\t\t\ttypename FillPartitionsDevice<span>::Shared packed;
\t\t\t/* a bank of plain FASTA / FASTQ files: the text itself goes to the device and is parsed there (several ranks: a byte range of every file each); anything else is iterated */
\t\t\tstd::vector<std::string> textFiles;
\t\t\tbool direct = getenv ("GATB_DEVICE_NO_TEXT") == 0  &&  DeviceSession::plainTextFiles (_bank, textFiles);
\t\t\tif (direct  &&  !device.pushTextFiles (textFiles, _progress))  { direct = false;  device.beginPass (pass); }      /* not the device parser's subset: the pass again, iterated */
\t\t\tif (!direct)
\t\t\t{
\t\t\t\tgetDispatcher()->iterate (itSeq, FillPartitionsDevice<span> (packed, _progress, _config._kmerSize), groupSize, deleteSynchro);
\t\t\t\tif (!packed.error.empty())  { throw system::Exception ("%s", packed.error.c_str()); }
\t\t\t}
\t\t\tdevice.endOfReads();        /* multi-rank: the exchanges that are still due */
\t\t\titSeq->finalize();
\t\t\t/* per-partition sizes for fillSolidKmers (progress, getNbCoresList) */
\t\t\t{
\t\t\t\tuint32_t nbSegments = 0;  device.check (gkc_segment_count (device.ctx(), &nbSegments));
\t\t\t\tstd::vector<uint64_t> recOff (_config._nb_partitions + 1), nbKmers (_config._nb_partitions);
\t\t\t\tfor (uint32_t s=0; s<nbSegments; s++)
\t\t\t\t{
\t\t\t\t\tdevice.check (gkc_segment_export (device.ctx(), s, 0, 0, recOff.data(), nbKmers.data()));
\t\t\t\t\tfor (size_t p=0; p<_config._nb_partitions; p++)  { pInfo.incKmer (p, nbKmers[p]);  pInfo.incKxmer (p, recOff[p+1]-recOff[p]); }
\t\t\t\t}
\t\t\t}
\t\t\tdevice.finishPass (pass);
\t\t\t/* the bank statistics of pass 0, read AFTER finishPass: with several ranks that is after gkc_gather_results, which sums / combines the ranks' counters on
\t\t\t * rank 0 (every rank scanned only its share of the reads) — rank 0's .h5 then reports what a single process reports */
\t\t\tif (pass == 0)
\t\t\t{
\t\t\t\tgkc_stats st;  device.check (gkc_get_stats (device.ctx(), &st));
\t\t\t\tif (direct  ||  device.ranks() > 1)      /* BankStats::update (BankKmers.hpp:176-186) from the device's counters */
\t\t\t\t{
\t\t\t\t\tpacked.stats = BankStats();
\t\t\t\t\tpacked.stats.sequencesNb = st.nb_sequences;  packed.stats.sequencesTotalLength = st.nb_bases;  packed.stats.sequencesTotalLengthSquare = st.seq_len_sq_sum;
\t\t\t\t\tpacked.stats.sequencesMinLength = st.seq_len_min;  packed.stats.sequencesMaxLength = st.seq_len_max;
\t\t\t\t}
\t\t\t\tpacked.stats.kmersNbValid = st.kmers_nb_valid;  packed.stats.kmersNbInvalid = st.kmers_nb_invalid;
\t\t\t\t_bankStats += packed.stats;
\t\t\t}
#else
"""

# fillSolidKmers_aux (SortingCountAlgorithm.cpp:1487-1575): the command of a partition
CMD_BEGIN = "            ICommand* cmd = 0;\n"
CMD_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
            cmd = new PartitionsByDeviceCommand<span> (
                processorClone, cacheSize, _progress, _fillTimeInfo,
                pInfo, pass, p, _config._nbCores_per_partition, _config._kmerSize, pool, _superKstorage
            );
            (void) forceVector;  (void) memoryPartition;
#else
"""
CMD_END = "            cmds.push_back (cmd);\n"

# end of fillSolidKmers_aux (SortingCountAlgorithm.cpp:1596-1600): join Stage B
TAIL = "\tif(_config._solidityKind == KMER_SOLIDITY_SUM)\n\t\t_superKstorage->closeFiles();\n"
TAIL_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
\tDeviceSession::singleton().joinPass (pass, _config._nb_passes);
\tif (pass + 1 == _config._nb_passes)  { PartitionsByDeviceCommand<span>::mergeDeviceHistogram (processor); }      /* bulk mode: the device counted the histogram */
#endif
"""

# getInfo() (SortingCountAlgorithm.cpp:777-778): the reference's commands time "1.read / 2.sort / 3.dump" into _fillTimeInfo; the device command has none of these
# phases, so the device's own figures go where consumers look for them
TIME_ANCHOR = "    _fillTimeInfo /= getDispatcher()->getExecutionUnitsNumber();\n"
TIME_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
    DeviceSession::singleton().addTimes (_fillTimeInfo, getDispatcher()->getExecutionUnitsNumber());      /* device_stage_a / device_stage_b (HIP-event time on the MI355X), device_wait / device_hand_over (host wall) */
#endif
"""


def once(text, anchor):
    n = text.count(anchor)
    if n != 1:
        raise SystemExit("anchor found %d times (expected once): %r" % (n, anchor[:70]))
    return text.index(anchor)


def insert_after_line(text, anchor, new):
    i = once(text, anchor) + len(anchor)
    i = text.index("\n", i) + 1 if not anchor.endswith("\n") else i
    return text[:i] + new + text[i:]


def insert_before(text, anchor, new):
    i = once(text, anchor)
    return text[:i] + new + text[i:]


# fillPartitions (SortingCountAlgorithm.cpp:1244): the partition files of the disk shuffle. The device path writes none of them: ONE (empty) file keeps every
# later call on _superKstorage valid (flushFiles, closeFiles, getFilesStats, the commands' constructor argument) without nb_partitions open files
SUPERK = "\t\t\t_superKstorage = new SuperKmerBinFiles(_tmpStorageName_superK,\"superKparts\", _config._nb_partitions) ;\n"
SUPERK_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
\t\t\t_superKstorage = new SuperKmerBinFiles(_tmpStorageName_superK,"superKparts", 1) ;      /* the super-k-mers stay in HBM: no partition file is written */
#else
"""


def patch(src):
    """kmer/impl/SortingCountAlgorithm.cpp"""
    out = insert_after_line(src, SCA_INCLUDE_ANCHOR, SCA_INCLUDE)
    a = once(out, SUPERK)
    out = out[:a] + SUPERK_DEVICE + SUPERK + "#endif\n" + out[a + len(SUPERK):]
    a = once(out, FILL_BEGIN); b = out.index(FILL_END, a) + len(FILL_END)
    out = out[:a] + FILL_DEVICE + out[a:b] + "#endif\n" + out[b:]
    a = once(out, CMD_BEGIN) + len(CMD_BEGIN); b = once(out, CMD_END)
    out = out[:a] + CMD_DEVICE + out[a:b] + "#endif\n" + out[b:]
    out = insert_before(out, TAIL, TAIL_DEVICE)
    out = insert_after_line(out, TIME_ANCHOR, TIME_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# tools/collections/impl/Bloom.hpp — BloomFactory::createBloom (Bloom.hpp:1254-1266)
# ------------------------------------------------------------------------------------------------------------------------------------------
BLOOM_HPP = "src/gatb/tools/collections/impl/Bloom.hpp"
FACTORY_ANCHOR = "/** \\brief Factory that creates IBloom instances\n"
FACTORY_INCLUDE = """#ifdef GATB_WITH_DEVICE_COUNTING
template <typename Item> class BloomDevice;
} } } } }
#include <gatb_device/BloomDevice.hpp>   /* BloomDevice<Item>: the CPU class of the kind + a device copy of its array (libgkc_hip.so) */
namespace gatb { namespace core { namespace tools { namespace collections { namespace impl {
#endif

"""
CREATE_ANCHOR = "        switch (kind)\n        {\n            case tools::misc::BLOOM_NONE:      return new BloomNull<T>             ();\n"
CREATE_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
        /* a k-mer type the device knows, a kind it implements and an MI355X in the process: same bit array, bulk inserts and batched queries on the device */
        if (BloomDevice<T>::usable (kind, kmersize))  { return new BloomDevice<T> (kind, tai_bloom, nbHash, kmersize); }
#endif
"""


def patch_bloom_hpp(src):
    out = insert_before(src, FACTORY_ANCHOR, FACTORY_INCLUDE)
    out = insert_before(out, CREATE_ANCHOR, CREATE_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/BloomAlgorithm.cpp — execute() (BloomAlgorithm.cpp:155-199)
# ------------------------------------------------------------------------------------------------------------------------------------------
BLOOM_ALGO = "src/gatb/kmer/impl/BloomAlgorithm.cpp"
BA_INCLUDE_ANCHOR = "#include <gatb/kmer/impl/BloomAlgorithm.hpp>"
BA_INCLUDE = "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/DeviceContext.hpp>\n#endif\n"
BA_BUILD = "        bloom = builder.build (itKmers);\n"
BA_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
        /* the solid k-mers of the counting step are still in HBM (DeviceSession, bulk mode) and they are the Iterable we were given: they are inserted where
         * they lie (gkc_bloom_insert_solid) instead of being read back from /dsk/solid and inserted one by one (BloomBuilder.hpp:117, :178) */
        if (gatb::core::device::DeviceContext::singleton().residentMatches (solidKmersNb, sizeof(Type)))
        {
            IBloom<Type>* candidate = BloomFactory::singleton().createBloom<Type> (_bloomKind, estimatedBloomSize, nbHash, _kmerSize);
            BloomDevice<Type>* onDevice = dynamic_cast<BloomDevice<Type>*> (candidate);
            if (onDevice != 0)  { onDevice->insertSolid ();  bloom = candidate; }
            else                { candidate->use();  candidate->forget(); }
        }
        if (bloom == 0)
#endif
"""


def patch_bloom_algo(src):
    out = insert_after_line(src, BA_INCLUDE_ANCHOR, BA_INCLUDE)
    out = insert_before(out, BA_BUILD, BA_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/MPHFAlgorithm.cpp — execute() (:150-187), populate() (:216-275)
# ------------------------------------------------------------------------------------------------------------------------------------------
MPHF_ALGO = "src/gatb/kmer/impl/MPHFAlgorithm.cpp"
MA_INCLUDE_ANCHOR = "#include <gatb/kmer/impl/MPHFAlgorithm.hpp>"
MA_INCLUDE = "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/MphfDevice.hpp>   /* BooPHF build + abundance map on the MI355X (libgkc_hip.so) */\n#endif\n"
MA_BUILD_BEGIN = "        /** We build the hash. */\n        {   TIME_INFO (getTimeInfo(), \"build\");\n"
MA_BUILD_END = "            _dataSize = _abundanceMap->save (_group, _name);\n        }\n"
MA_BUILD_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
        /* BooPHF on the device: built (from the solid k-mers in HBM when they are still there), its stream stored as <group>/<name> — what save() writes — and
         * LOADED by the reference's own MapMPHF::load, so the object the consumers get is the reference's BooPHF read from those bytes */
        bool builtOnDevice = false;
        if (sizeof(Abundance_t) == 1)
        {
            {   TIME_INFO (getTimeInfo(), "build");
                builtOnDevice = MphfDevice::build<Type> (this, _group, _name, _solidKmers, _dataSize);
            }
            if (builtOnDevice)
            {   TIME_INFO (getTimeInfo(), "save");       /* (the stream is in the storage already: this is the read-back) */
                _abundanceMap->load (_group, _name);
            }
        }
        if (!builtOnDevice)
        {
#endif
"""
MA_BUILD_TAIL = """#ifdef GATB_WITH_DEVICE_COUNTING
        }
#endif
"""
MA_POP_ANCHOR = "    // set counts and at the same time, test the mphf\n"
MA_POP_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
    /* the same cells from the device: cell[code(kmer)] = abundance index, for the solid k-mers in HBM (gkc_mphf_abundance_map) */
    bool populatedOnDevice = false;
    if (sizeof(Abundance_t) == 1  &&  n > 0)
    {
        size_t above = 0;
        populatedOnDevice = MphfDevice::populate (this, (u_int8_t*) & _abundanceMap->at ((typename AbundanceMap::Hash::Code) 0), n, above);
        if (populatedOnDevice)  { _nb_abundances_above_precision = above;  nb_iterated = n; }
    }
    MphfDevice::release (this);
    if (!populatedOnDevice)
#endif
"""


def patch_mphf_algo(src):
    out = insert_after_line(src, MA_INCLUDE_ANCHOR, MA_INCLUDE)
    a = once(out, MA_BUILD_BEGIN); b = out.index(MA_BUILD_END, a) + len(MA_BUILD_END)
    out = out[:a] + MA_BUILD_DEVICE + out[a:b] + MA_BUILD_TAIL + out[b:]
    out = insert_after_line(out, MA_POP_ANCHOR, MA_POP_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/DebloomMinimizerAlgorithm.cpp — the contains8 call site (:201) and the partition loop (:352-372)
# ------------------------------------------------------------------------------------------------------------------------------------------
DEBLOOM_ALGO = "src/gatb/kmer/impl/DebloomMinimizerAlgorithm.cpp"
DB_MEMBER_ANCHOR = "    Model&        model;\n    IBloom<Type>* bloom;\n"
DB_MEMBER = """#ifdef GATB_WITH_DEVICE_COUNTING
    /** the same for the i-th solid k-mer of the partition when its 8 neighbour tests came out of ONE batched device query (masks[i], in the order of `solids`) */
    struct ByIndex
    {
        const FunctorKmersExtensionMinimizer& outer;  const std::vector<Type>& solids;  const u_int8_t* masks;
        FunctorNeighbors neighbors;       /* own copy per thread (Dispatcher::iterate copies the functor): it caches the thread's index */
        ByIndex (const FunctorKmersExtensionMinimizer& o, const std::vector<Type>& s, const u_int8_t* m) : outer(o), solids(s), masks(m), neighbors(o.functorNeighbors) {}
        void operator() (const u_int64_t& i)  { outer.model.iterateNeighbors (solids[i], neighbors, bitset<8> (masks[i])); }
    };
#endif
"""
DB_LOOP_ANCHOR = "            /** We iterate the solid kmers. */\n            this->getDispatcher()->iterate (itKmers, functorKmers);\n"
DB_LOOP_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
            /* the filter is a BloomDevice: the 8 neighbour tests of every solid k-mer of the partition in one device query (gkc_bloom_contains8), then the
             * neighbours are iterated by index over the (sorted) solids vector — same neighbours, same insertions as one contains8 call per k-mer */
            BloomDevice<Type>* onDevice = dynamic_cast<BloomDevice<Type>*> (bloom);
            if (onDevice != 0  &&  !solids.empty()  &&  getenv ("GATB_DEVICE_NO_BATCHED_QUERIES") == 0)
            {
                std::vector<u_int8_t> deviceMasks (solids.size());
                onDevice->contains8Batch (solids.data(), solids.size(), deviceMasks.data());
                typename FunctorKmersExtensionMinimizer<Model,ModelMini,Count,Type>::ByIndex byIndex (functorKmers, solids, deviceMasks.data());
                this->getDispatcher()->iterate (new typename Range<u_int64_t>::Iterator (0, solids.size() - 1), byIndex);
            }
            else
#endif
"""


def patch_debloom_algo(src):
    out = insert_after_line(src, DB_MEMBER_ANCHOR, DB_MEMBER)
    out = insert_before(out, DB_LOOP_ANCHOR, DB_LOOP_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/ConfigurationAlgorithm.cpp — partitions / passes sized from the device's memory (:398-431)
# ------------------------------------------------------------------------------------------------------------------------------------------
CONFIG_ALGO = "src/gatb/kmer/impl/ConfigurationAlgorithm.cpp"
CA_INCLUDE_ANCHOR = "#include <gatb/kmer/impl/ConfigurationAlgorithm.hpp>"
CA_INCLUDE = "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/DeviceConfiguration.hpp>   /* partitions and passes from the HBM of the MI355X, not from host RAM / disk */\n#endif\n"
CA_ANCHOR = "    //_nb_partitions_in_parallel = 1 ;\n"
CA_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
    /* counting happens in HBM: partitions of the size the device kernels are cut for, passes from the device's memory (the host-RAM / disk / open-files rules above
     * describe a machine the super-k-mers never touch). GATB_DEVICE_REFERENCE_CONFIG=1 keeps the reference's values. */
    DeviceConfiguration::apply (_config, sizeof(Type));
#endif
"""


def patch_config_algo(src):
    out = insert_after_line(src, CA_INCLUDE_ANCHOR, CA_INCLUDE)
    out = insert_before(out, CA_ANCHOR, CA_DEVICE)
    return out


# ------------------------------------------------------------------------------------------------------------------------------------------
# kmer/impl/RepartitionAlgorithm.cpp — the two serial sampling iterations (:348-353 MmersFrequency, :464-474 SampleRepart of a single bank)
# ------------------------------------------------------------------------------------------------------------------------------------------
REPART_ALGO = "src/gatb/kmer/impl/RepartitionAlgorithm.cpp"
RA_INCLUDE_ANCHOR = "#include <gatb/kmer/impl/RepartitionAlgorithm.hpp>"
RA_INCLUDE = "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/RepartitorDevice.hpp>   /* the sampling functors' work on the MI355X (libgkc_hip.so) */\n#endif\n"
RA_FREQ_ANCHOR = "    serialDispatcher.iterate (it_all_reads,  MmersFrequency<span> (\n"
RA_FREQ_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
    /* the m-mers of the sample are counted on the device (gkc_count_mmers); false = no device / outside its range: the functor below runs */
    if (RepartitorDevice::countMmers (_bank, _config, nbseq_sample, m_mer_counts) == false)
#endif
"""
RA_SAMPLE_ANCHOR = "\t\tserialDispatcher.iterate (it_all_reads, SampleRepart<span> (\n"
RA_SAMPLE_DEVICE = """#ifdef GATB_WITH_DEVICE_COUNTING
\t\t/* super-k-mers, k-mers and kx-mers per minimizer of the sample on the device (gkc_sample_exact, the reference's stop rule); false: the functor below runs */
\t\tif (RepartitorDevice::sample (_bank, _config, _freq_order, nbseq_sample, sample_info) == false)
#endif
"""


def patch_repart_algo(src):
    out = insert_after_line(src, RA_INCLUDE_ANCHOR, RA_INCLUDE)
    out = insert_before(out, RA_FREQ_ANCHOR, RA_FREQ_DEVICE)
    out = insert_before(out, RA_SAMPLE_ANCHOR, RA_SAMPLE_DEVICE)
    return out


FILES = [(REL, patch), (CONFIG_ALGO, patch_config_algo), (BLOOM_HPP, patch_bloom_hpp), (BLOOM_ALGO, patch_bloom_algo), (MPHF_ALGO, patch_mphf_algo), (DEBLOOM_ALGO, patch_debloom_algo), (REPART_ALGO, patch_repart_algo)]
PATCH_NAME = "gatb-core.device.patch"


def make_diff(ref):
    """the unified diff of all patched files against the reference tree at `ref`"""
    parts = []
    for rel, fn in FILES:
        src = open(os.path.join(ref, rel)).read()
        new = fn(src)
        parts.append("".join(difflib.unified_diff(src.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=1)))
    return "".join(parts)


def main():
    ref, scratch = sys.argv[1], sys.argv[2]
    for rel, fn in FILES:
        src = open(os.path.join(ref, rel)).read()
        dst = os.path.join(scratch, rel[len("src/"):])
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        new = fn(src)
        if not (os.path.exists(dst) and open(dst).read() == new):      # (untouched copies keep their time stamp: make-style rebuilds stay cheap)
            open(dst, "w").write(new)
        print(dst)
    if "--write-patch" in sys.argv:
        here = os.path.dirname(os.path.abspath(__file__))
        open(os.path.join(here, PATCH_NAME), "w").write(make_diff(ref))


if __name__ == "__main__":
    main()
