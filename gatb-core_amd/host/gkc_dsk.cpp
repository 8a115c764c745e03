// gkc_dsk — minimal command line over SortingCountAlgorithm<span> (the role tools/dbgh5.cpp plays for the count-only case):
//   gkc_dsk -in reads.fa|fq -kmer-size 31 [-abundance-min 2] [-minimizer-size 10] [-minimizer-type 0|1] [-nb-partitions N] [-span 64|96|128] -out prefix
// writes prefix.solid.<dataset> (raw Count[] records, ascending), prefix.minimRepart, prefix.histo, prefix.info
#include "gatb_gkc.hpp"
#include <iostream>
using namespace gatb::core;
using namespace gatb::core::kmer::impl;

template <size_t span> static int run(tools::misc::IProperties* params, const std::string& in, const std::string& out)
{
    SortingCountAlgorithm<span> dsk(new bank::BankFasta(in), params);
    dsk.execute();
    for (auto& kv : dsk.getInfo()->map()) std::cout << kv.first << " : " << kv.second << std::endl;
    if (!out.empty()) {
        auto* chain = dsk.getDskChain();
        if (chain) {
            chain->template get<CountProcessorDump<span>>()->saveRaw(out);
            auto* hp = chain->template get<CountProcessorHistogram<span>>();
            std::ofstream h(out + ".histo"); const auto& hv = hp->getHistogram();
            for (size_t i = 1; i < hv.size(); i++) if (hv[i]) h << i << "\t" << hv[i] << "\n";
            hp->compute_threshold(2);                                              // histogram/cutoff, nbsolidsforcutoff (SortingCountAlgorithm.cpp:700-726)
            std::ofstream cu(out + ".cutoff"); cu << hp->get_solid_cutoff() << "\t" << hp->get_nbsolids_auto() << "\t" << hp->get_first_peak() << "\n";
        }
        if (chain && params->has("-mphf") && params->getInt("-mphf")) {                         // dbgh5's mphf step on the counted k-mers (-mphf 1)
            MPHFAlgorithm<span> mphf(dsk.context(), (size_t)params->getInt(STR_KMER_SIZE), &chain->template get<CountProcessorDump<span>>()->getSolidCounts());
            mphf.execute();
            const auto h = mphf.savedHash(); std::ofstream mo(out + ".mphf", std::ios::binary); mo.write((const char*)h.data(), (std::streamsize)h.size());
            auto& d = mphf.getAbundanceMap()->data(); std::ofstream ao(out + ".abundancemap", std::ios::binary); ao.write((const char*)d.data(), (std::streamsize)d.size());
        }
        dsk.saveH5(out + ".h5");
        dsk.getRepartitor()->save(out + ".minimRepart");
        std::ofstream info(out + ".info"); for (auto& kv : dsk.getInfo()->map()) info << kv.first << "\t" << kv.second << "\n";
    }
    return 0;
}
int main(int argc, char** argv)
{
    // the step's own options (SortingCountAlgorithm::getOptionsParser) + the ones of this tool
    tools::misc::IOptionsParser* parser = SortingCountAlgorithm<32>::getOptionsParser(true);
    typedef tools::misc::IOptionsParser::Option O;
    parser->push_back(O{ "-mphf", "also build the MPHF + abundance map of the solid k-mers (1)", false, true, "0", 1 });
    parser->push_back(O{ "-span", "instantiate the classes with this span (64, 96, 128) instead of the smallest one holding k", false, true, "0", 1 });
    tools::misc::IProperties* params = nullptr;
    try { params = parser->parse(argc, argv); }
    catch (system::Exception& e) { std::cerr << e.getMessage() << std::endl << parser->help(); delete parser; return 2; }
    delete parser;
    const std::string in = params->getStr(STR_URI_INPUT), out = params->has(STR_URI_OUTPUT) ? params->getStr(STR_URI_OUTPUT) : "";
    try {
        const size_t k = (size_t)params->getInt(STR_KMER_SIZE);
        const size_t span = (size_t)params->getInt("-span");       // 0: the smallest span holding k, as Integer::apply / setVariant picks it (tools/math/Integer.hpp:58-90)
        if (span == 96) return run<96>(params, in, out);            // forced larger spans (what a build with KSIZE_LIST "32 64 96 128" instantiates): k <= 63 on this device
        if (span == 128) return run<128>(params, in, out);
        if (span == 64) return run<64>(params, in, out);
        return k <= 31 ? run<32>(params, in, out) : run<64>(params, in, out);
    } catch (system::Exception& e) { std::cerr << "EXCEPTION: " << e.getMessage() << std::endl; return 1; }
}
