// unitigs_check — GraphUnitigs (the bcalm2 path) consuming a k-mer counted .h5: builds the compacted de Bruijn graph from
//     unitigs_check <file.h5 | reads.fa> <out prefix> [nb-cores] [more GraphUnitigs options, e.g. -kmer-size 21 -abundance-min 2]
// exactly as GraphUnitigsTemplate<span>::create does for an .h5 input (debruijn/impl/GraphUnitigs.cpp:907-944: configure_visitor loads
// /dsk/solid/*, /minimizers/minimRepart + minimFrequency; build_unitigs_postsolid -> bcalm_algo.cpp:291-330 reads the Repartitor and the
// solid partitions). Compiled against the reference library by integration/check_graphunitigs.sh (build container only); used to show that
// the .h5 written by this repository's HDF5 writer drops into GraphUnitigs unchanged: the unitigs must equal those built from the
// reference's own .h5 of the same input.
#include <gatb/gatb_core.hpp>
#include <gatb/debruijn/impl/GraphUnitigs.hpp>
#include <iostream>

int main (int argc, char* argv[])
{
    if (argc < 3)  { std::cerr << "usage: unitigs_check file.h5|reads.fa out_prefix [nb-cores] [options]" << std::endl; return 2; }
    try
    {
        const char* cores = argc > 3 ? argv[3] : "1";
        typedef gatb::core::debruijn::impl::GraphUnitigsTemplate<32> GraphUnitigs;      // span 32: k <= 31 (test/unit/src/debruijn/TestDebruijnUnitigs.cpp:88)
        // a FASTA / FASTQ input makes GraphUnitigs count first (GraphUnitigs.cpp:222: its own SortingCountAlgorithm — on the MI355X when this program is linked
        // with the patched units), an .h5 input takes the restart path (:907-944)
        std::string extra;  for (int i = 4; i < argc; i++)  { extra += " ";  extra += argv[i]; }
        GraphUnitigs graph = GraphUnitigs::create ("-in %s -out %s -nb-cores %s -verbose 0%s", argv[1], argv[2], cores, extra.c_str());
        std::cout << graph.getInfo();
        size_t nb = 0;
        gatb::core::debruijn::impl::GraphIterator<gatb::core::debruijn::impl::NodeGU> it = graph.iterator ();
        for (it.first(); !it.isDone(); it.next())  { nb++; }
        std::cout << "unitig_extremity_nodes " << nb << std::endl;
    }
    catch (Exception& e)  { std::cerr << "EXCEPTION: " << e.getMessage() << std::endl; return 1; }
    return 0;
}
