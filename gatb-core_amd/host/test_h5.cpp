// writes a file with the native HDF5 writer (gkc_h5.hpp) for tests/test_h5_writer.py:  test_h5 out.h5 n_datasets
#include "gkc_h5.hpp"
#include <cstdio>
#include <cstdlib>
#include <fstream>
struct Count16 { uint64_t value; uint32_t abundance; uint32_t pad; };
struct Count32 { unsigned __int128 value; uint32_t abundance; uint32_t pad[3]; };
int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const int nd = atoi(argv[2]);
    gkc_h5::File f((size_t)nd);
    f.set_attribute("/", "kmer_size", "31"); f.set_attribute("/", "xml", "\n<a>\n   <b>some longer text & stuff</b>\n</a>");
    f.set_attribute("/dsk/solid", "nb_partitions", std::to_string(nd));
    const gkc_h5::Type c16 = gkc_h5::Type::compound(16, { {"value", {0, gkc_h5::Type::integer(8, false)}}, {"abundance", {8, gkc_h5::Type::integer(4, false)}} });
    const gkc_h5::Type c32 = gkc_h5::Type::compound(32, { {"value", {0, gkc_h5::Type::integer(16, true, 128)}}, {"abundance", {16, gkc_h5::Type::integer(4, false)}} });
    for (int i = 0; i < nd; i++) {
        std::vector<Count16> v((size_t)(i % 7));                    // some datasets are empty
        for (size_t j = 0; j < v.size(); j++) { v[j].value = 1000003ULL * (uint64_t)i + j; v[j].abundance = (uint32_t)(i + j); v[j].pad = 0; }
        f.add_dataset("/dsk/solid/" + std::to_string(i), c16, v.data(), v.size());
    }
    std::vector<Count32> w(3);
    for (size_t j = 0; j < w.size(); j++) { memset(&w[j], 0, sizeof(Count32)); w[j].value = ((unsigned __int128)(j + 1) << 100) | (j + 5); w[j].abundance = (uint32_t)(j + 9); }
    f.add_dataset("/wide/0", c32, w.data(), w.size());
    std::vector<uint8_t> bytes(100000); for (size_t j = 0; j < bytes.size(); j++) bytes[j] = (uint8_t)(j * 7);
    f.add_dataset("/minimizers/minimRepart", gkc_h5::Type::integer(1, false), bytes.data(), bytes.size());
    const auto& img = f.finish();
    std::ofstream o(argv[1], std::ios::binary); o.write((const char*)img.data(), (std::streamsize)img.size());
    return 0;
}
