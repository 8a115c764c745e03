#!/usr/bin/env python
"""Applies the device-counting patch to a scratch COPY of the reference's SortingCountAlgorithm.cpp (the reference tree is never written)
and emits the unified diff a maintainer would apply (integration/SortingCountAlgorithm.device.patch).

    python integration/make_patched_sources.py <reference gatb-core dir> <scratch include dir> [--write-patch]

The edits are anchored on lines of the reference file (file:line cited per hunk); every anchor must be found exactly once. Everything new is
guarded by GATB_WITH_DEVICE_COUNTING, so the patched file still builds the CPU path without the macro."""
import difflib
import os
import sys

REL = "src/gatb/kmer/impl/SortingCountAlgorithm.cpp"

HUNKS = [
    # (anchor line, mode, text)   mode: "after" inserts after the anchor line, "wrap" guards [anchor .. end anchor] with #ifndef/#else
    # --- include (after the last include of the file head, SortingCountAlgorithm.cpp:20-40)
    ("#include <gatb/kmer/impl/SortingCountAlgorithm.hpp>", "after",
     "#ifdef GATB_WITH_DEVICE_COUNTING\n#include <gatb_device/DeviceCounting.hpp>   /* MI355X back-end: libgkc_hip.so */\n#endif\n"),
]

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
\t\t\tif (pass == 0)
\t\t\t{
\t\t\t\tgkc_stats st;  device.check (gkc_get_stats (device.ctx(), &st));
\t\t\t\tif (direct)      /* BankStats::update (BankKmers.hpp:176-186) from the device's counters */
\t\t\t\t{
\t\t\t\t\tpacked.stats.sequencesNb = st.nb_sequences;  packed.stats.sequencesTotalLength = st.nb_bases;  packed.stats.sequencesTotalLengthSquare = st.seq_len_sq_sum;
\t\t\t\t\tpacked.stats.sequencesMinLength = st.seq_len_min;  packed.stats.sequencesMaxLength = st.seq_len_max;
\t\t\t\t}
\t\t\t\tpacked.stats.kmersNbValid = st.kmers_nb_valid;  packed.stats.kmersNbInvalid = st.kmers_nb_invalid;
\t\t\t\t_bankStats += packed.stats;
\t\t\t}
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
\t\t\tdevice.finishPass();
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
\tDeviceSession::singleton().joinPass();
\tif (pass + 1 == _config._nb_passes)  { PartitionsByDeviceCommand<span>::mergeDeviceHistogram (processor); }      /* bulk mode: the device counted the histogram */
#endif
"""


def once(text, anchor):
    n = text.count(anchor)
    if n != 1:
        raise SystemExit("anchor found %d times (expected once): %r" % (n, anchor[:60]))
    return text.index(anchor)


def patch(src):
    out = src
    for anchor, mode, text in HUNKS:
        i = once(out, anchor) + len(anchor)
        i = out.index("\n", i) + 1
        out = out[:i] + text + out[i:]
    a = once(out, FILL_BEGIN); b = out.index(FILL_END, a) + len(FILL_END)
    out = out[:a] + FILL_DEVICE + out[a:b] + "#endif\n" + out[b:]
    a = once(out, CMD_BEGIN) + len(CMD_BEGIN); b = once(out, CMD_END)
    out = out[:a] + CMD_DEVICE + out[a:b] + "#endif\n" + out[b:]
    a = once(out, TAIL)
    out = out[:a] + TAIL_DEVICE + out[a:]
    return out


def main():
    ref, scratch = sys.argv[1], sys.argv[2]
    src = open(os.path.join(ref, REL)).read()
    new = patch(src)
    dst = os.path.join(scratch, "gatb/kmer/impl/SortingCountAlgorithm.cpp")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst, "w").write(new)
    if "--write-patch" in sys.argv:
        diff = difflib.unified_diff(src.splitlines(True), new.splitlines(True), "a/" + REL, "b/" + REL, n=1)
        here = os.path.dirname(os.path.abspath(__file__))
        open(os.path.join(here, "SortingCountAlgorithm.device.patch"), "w").write("".join(diff))
    print(dst)


if __name__ == "__main__":
    main()
