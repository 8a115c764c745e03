#include "../gatb-core_amd/host/gatb_gkc.hpp"
#include <iostream>
using namespace gatb::core; using namespace gatb::core::kmer::impl;
int main(int argc, char** argv) {
    std::vector<std::string> seqs; for (int i = 2; i < argc; i++) seqs.push_back(argv[i]);
    auto* params = SortingCountAlgorithm<32>::getDefaultProperties();
    params->setInt(STR_KMER_SIZE, atoi(argv[1])); params->setInt(STR_KMER_ABUNDANCE_MIN, 1);
    SortingCountAlgorithm<32> dsk(new bank::BankStrings(seqs), params);
    dsk.execute();
    for (auto& kv : dsk.getInfo()->map()) std::cout << kv.first << " : " << kv.second << std::endl;
    auto& c = dsk.getConfig(); std::cout << "P=" << c._nb_partitions << " m=" << c._minim_size << " passes=" << c._nb_passes << " kmersNb=" << c._kmersNb << std::endl;
    size_t tot = 0; for (auto& d : dsk.getSolidCounts()) { std::cout << d.size() << " "; tot += d.size(); } std::cout << " total " << tot << std::endl;
    gkc_stats st; gkc_get_stats(dsk.context(), &st); std::cout << "dev distinct " << st.kmers_nb_distinct << " valid " << st.kmers_nb_valid << " sk " << st.nb_superkmers << std::endl;
    return 0;
}
