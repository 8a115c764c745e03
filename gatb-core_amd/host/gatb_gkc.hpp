// gatb_gkc.hpp — C++ host side of the MI355X DSK hot path, mirroring the reference's plug-in API for this path.
//
// Same namespaces, class names, method names, argument meaning and error behaviour as the reference classes the path sits
// behind, so a test written against gatb::core::kmer::impl reads the same here:
//   Kmer<span>::{Type,Count}        kmer/impl/Model.hpp:89-1592, tools/misc/api/Abundance.hpp:68-129
//   ICountProcessor<span>           kmer/api/ICountProcessor.hpp:92-185          (full call protocol preserved)
//   CountProcessor{Histogram,SoliditySum,Dump,Chain}   kmer/impl/CountProcessor*.hpp
//   Repartitor                      kmer/impl/PartiInfo.hpp:292-387, PartiInfo.cpp:48-295
//   Configuration                   kmer/impl/Configuration.hpp:38-117
//   SortingCountAlgorithm<span>     kmer/impl/SortingCountAlgorithm.hpp:65-263, .cpp:525-781
//   IBloom<Item> / BloomFactory     tools/collections/impl/Bloom.hpp:113-168, 1240-1282
//   system::Exception               system/api/Exception.hpp:59-100
// Everything below the API is the C-ABI of libgkc_hip.so (include/gkc.h); no algorithmic work is done on the host:
// the host only builds the repartition table from device-computed statistics (as the reference does on its host) and
// forwards the counted records to the processors in ascending order.
//
// Not mirrored (out of scope): bank parsers (a minimal in-memory / FASTA-FASTQ reader is provided so the
// examples run), the Storage/HDF5 layer (CountProcessorDump keeps the partitions in memory and can write raw
// `Count` arrays + the `minimRepart` byte stream), OptionsParser/Properties trees (a flat string map is used).
#pragma once
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>
#include <new>
#include <algorithm>
#include <cmath>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include <queue>
#include "../../include/gkc.h"
#include "gkc_h5.hpp"

namespace gatb { namespace core {

namespace system {
/** printf-style exception (system/api/Exception.hpp:59-100) */
class Exception {
public:
    Exception() {}
    Exception(const char* fmt, ...) { char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); _message = buf; }
    const char* getMessage() const { return _message.c_str(); }
protected:
    std::string _message;
};
/** intrusive reference counting (system/api/ISmartPointer.hpp:97-115) */
class SmartPointer {
public:
    SmartPointer() : _ref(0) {}
    virtual ~SmartPointer() {}
    void use() { _ref++; }
    void forget() { if (--_ref <= 0) delete this; }
private:
    int _ref;
};
}  // namespace system

typedef int32_t CountNumber;                       // system/api/types.hpp:49
typedef std::vector<CountNumber> CountVector;

namespace tools { namespace misc {
/** flat property set standing in for IProperties (keys are the reference's STR_* strings, StringsRepository.hpp:115-180) */
class Properties {
public:
    void setInt(const std::string& k, int64_t v) { _m[k] = std::to_string(v); }
    void setStr(const std::string& k, const std::string& v) { _m[k] = v; }
    int64_t getInt(const std::string& k) const { auto it = _m.find(k); if (it == _m.end()) throw system::Exception("Empty property '%s'", k.c_str()); return atoll(it->second.c_str()); }
    std::string getStr(const std::string& k) const { auto it = _m.find(k); if (it == _m.end()) throw system::Exception("Empty property '%s'", k.c_str()); return it->second; }
    bool has(const std::string& k) const { return _m.count(k) != 0; }
    void add(const std::string& k, const char* fmt, ...) { char buf[256]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap); _m[k] = buf; }
    const std::map<std::string, std::string>& map() const { return _m; }
private:
    std::map<std::string, std::string> _m;
};
typedef Properties IProperties;

/** Command-line options of the counting step, the role of IOptionsParser / OptionOneParam for this path (tools/misc/impl/OptionsParser.hpp;
 *  option list: SortingCountAlgorithm.cpp:202-236). parse() yields the property set the algorithm constructors take. */
class IOptionsParser {
public:
    struct Option { std::string name, help; bool mandatory; bool hasDefault; std::string defaultValue; int nbArgs; };
    explicit IOptionsParser(const std::string& name) : _name(name) {}
    void push_back(const Option& o) { _options.push_back(o); }
    const std::vector<Option>& getOptions() const { return _options; }
    const std::string& getName() const { return _name; }
    const Option* find(const std::string& name) const { for (auto& o : _options) if (o.name == name) return &o; return nullptr; }
    /** argv[1..]: "-name value" pairs (or a bare "-name" for options without argument); unknown options and missing mandatory ones throw */
    IProperties* parse(int argc, char** argv) const {
        IProperties* props = new IProperties();
        for (auto& o : _options) if (o.hasDefault) props->setStr(o.name, o.defaultValue);
        for (int i = 1; i < argc; i++) {
            const Option* o = find(argv[i]);
            if (!o) { delete props; throw system::Exception("Unknown parameter '%s'", argv[i]); }
            if (o->nbArgs == 0) { props->setStr(o->name, ""); continue; }
            if (i + 1 >= argc) { delete props; throw system::Exception("Too few arguments for the %s option...", o->name.c_str()); }
            props->setStr(o->name, argv[++i]);
        }
        for (auto& o : _options) if (o.mandatory && !props->has(o.name)) { const std::string n = o.name; delete props; throw system::Exception("Option '%s' is mandatory", n.c_str()); }
        return props;
    }
    std::string help() const {
        std::string h = "[" + _name + " options]\n";
        for (auto& o : _options) { char line[512]; snprintf(line, sizeof(line), "   %-22s (%d arg) :    %s%s%s\n", o.name.c_str(), o.nbArgs, o.help.c_str(), o.hasDefault ? "  [default '" : "", o.hasDefault ? (o.defaultValue + "']").c_str() : ""); h += line; }
        return h;
    }
private:
    std::string _name; std::vector<Option> _options;
};
}}  // namespace tools::misc

#define STR_KMER_SIZE            "-kmer-size"
#define STR_KMER_ABUNDANCE_MIN   "-abundance-min"
#define STR_KMER_ABUNDANCE_MAX   "-abundance-max"
#define STR_MINIMIZER_SIZE       "-minimizer-size"
#define STR_MINIMIZER_TYPE       "-minimizer-type"
#define STR_REPARTITION_TYPE     "-repartition-type"
#define STR_MAX_MEMORY           "-max-memory"
#define STR_HISTOGRAM_MAX        "-histo-max"
#define STR_URI_INPUT            "-in"
#define STR_URI_OUTPUT           "-out"
#define STR_NB_PARTITIONS        "-nb-partitions"      /* extension: force the partition count (0 = derive) */
#define STR_NB_PASSES            "-nb-passes"          /* extension: force the pass count (0 = derive)      */
#define STR_GPU_DEVICE           "-gpu"                /* extension: HIP device index                       */

// ------------------------------------------------------------------------------------------------ bank (minimal)
namespace bank {
struct Sequence { const char* data; size_t size; const char* getDataBuffer() const { return data; } size_t getDataSize() const { return size; } };
/** IBank restricted to what the hot path needs (bank/api/IBank.hpp:78-158): a flat base buffer + offsets */
class IBank : public system::SmartPointer {
public:
    virtual ~IBank() {}
    virtual const std::vector<char>& bases() const = 0;
    virtual const std::vector<uint64_t>& offsets() const = 0;     // n+1 entries
    /** number of files behind the bank (an album of several files is counted as their concatenation: the reference sums the per-bank counts
     *  of a k-mer when the solidity kind is "sum", its default — CounterBuilder, PartitionsCommand.hpp:57-97) */
    virtual size_t getNbBanks() const { return 1; }
    int64_t getNbItems() const { return (int64_t)offsets().size() - 1; }
    uint64_t getSize() const { return offsets().back(); }
    void estimate(uint64_t& number, uint64_t& totalSize, uint64_t& maxSize) const {
        number = (uint64_t)getNbItems(); totalSize = getSize(); maxSize = 0;
        for (size_t i = 0; i + 1 < offsets().size(); i++) maxSize = std::max<uint64_t>(maxSize, offsets()[i + 1] - offsets()[i]);
    }
};
/** BankStrings (bank/impl/BankStrings.hpp): sequences given as C strings */
class BankStrings : public IBank {
public:
    BankStrings() { _off.push_back(0); }
    BankStrings(const char* s, ...) { _off.push_back(0); va_list ap; va_start(ap, s); for (const char* p = s; p; p = va_arg(ap, const char*)) add(p); va_end(ap); }
    BankStrings(const char* seqs[], size_t n) { _off.push_back(0); for (size_t i = 0; i < n; i++) add(seqs[i]); }
    explicit BankStrings(const std::vector<std::string>& v) { _off.push_back(0); for (auto& s : v) add(s.c_str()); }
    void add(const char* s) { size_t n = strlen(s); _bases.insert(_bases.end(), s, s + n); _off.push_back(_bases.size()); }
    const std::vector<char>& bases() const { return _bases; }
    const std::vector<uint64_t>& offsets() const { return _off; }
protected:
    std::vector<char> _bases; std::vector<uint64_t> _off;
};
/** BankFasta (bank/impl/BankFasta.hpp): FASTA / FASTQ file, plain or gzip (gzread reads both, BankFasta.cpp:396).
 *  The file text is kept as read; the hot path hands it to the device parser (gkc_push_fastx) in chunks. bases()/offsets() — what
 *  the Repartitor sampling and estimate() use — come from a host walk of the text with the reader's rules
 *  (BankFasta::Iterator::get_next_seq_from_file, BankFasta.cpp:488-571): header = rest of the line after '>' / '@'; sequence lines
 *  appended up to '\n', one trailing '\r' dropped when the read is longer than 1; '+' starts a quality consumed by length. */
class BankFasta : public IBank {
public:
    /** one file, or several separated by commas (the "-in a.fa,b.fa" album of the reference's Bank::open, bank/impl/Bank.cpp / BankAlbum.hpp) */
    explicit BankFasta(const std::string& uri) : _nbFiles(0), _parsed(false) {
        size_t from = 0;
        std::vector<char> buf(1 << 22);
        while (from <= uri.size()) {
            size_t to = uri.find(',', from); if (to == std::string::npos) to = uri.size();
            const std::string path = uri.substr(from, to - from);
            from = to + 1;
            if (path.empty()) continue;
            gzFile f = gzopen(path.c_str(), "rb");
            if (!f) throw system::Exception("Unable to open file '%s'", path.c_str());
            if (!_text.empty() && _text.back() != '\n') _text.push_back('\n');          // a file without a final newline must not run into the next header
            for (;;) {
                const int n = gzread(f, buf.data(), (unsigned)buf.size());
                if (n < 0) { gzclose(f); throw system::Exception("read error in '%s'", path.c_str()); }
                if (n == 0) break;
                _text.append(buf.data(), (size_t)n);
            }
            gzclose(f);
            _nbFiles++;
        }
        if (_nbFiles == 0) throw system::Exception("Unable to open file '%s'", uri.c_str());
    }
    size_t getNbBanks() const { return _nbFiles; }
    const std::string& text() const { return _text; }
    const std::vector<char>& bases() const { parse(); return _bases; }
    const std::vector<uint64_t>& offsets() const { parse(); return _off; }
private:
    void parse() const {
        if (_parsed) return;
        _parsed = true; _off.assign(1, 0);
        const char* p = _text.data(); const size_t n = _text.size(); size_t pos = 0; int last = 0;
        auto getc = [&]() -> int { return pos < n ? (int)(signed char)p[pos++] : -1; };
        auto line = [&](std::vector<char>* dst, size_t start) -> bool {        // rest of the current line; false at end of text
            if (pos >= n) return false;
            size_t i = pos; while (i < n && p[i] != '\n') i++;
            if (dst) dst->insert(dst->end(), p + pos, p + i);
            pos = i < n ? i + 1 : i;
            if (dst && dst->size() - start > 1 && dst->back() == '\r') dst->pop_back();
            return true;
        };
        std::vector<char> qual;
        for (;;) {
            int c;
            if (last == 0) { while ((c = getc()) != -1 && c != '>' && c != '@') ; if (c == -1) break; last = c; }
            if (pos >= n) break;
            { size_t i = pos; while (i < n && !isspace((unsigned char)p[i])) i++; const int d = i < n ? p[i] : 0; pos = i < n ? i + 1 : i; if (d != '\n') line(nullptr, 0); }
            const size_t start = _bases.size();
            while ((c = getc()) != -1 && c != '>' && c != '+' && c != '@') { if (c == '\n') continue; _bases.push_back((char)c); line(&_bases, start); }
            if (c == '>' || c == '@') last = c;
            if (c == '+') {
                while ((c = getc()) != -1 && c != '\n') ;
                qual.clear();
                while (line(&qual, 0) && qual.size() < _bases.size() - start) ;
                last = 0;
            }
            _off.push_back(_bases.size());
        }
    }
    std::string _text; size_t _nbFiles;
    mutable bool _parsed; mutable std::vector<char> _bases; mutable std::vector<uint64_t> _off;
};
}  // namespace bank

namespace kmer { namespace impl {

// ------------------------------------------------------------------------------------------------ k-mer integer + Count
/** LargeInt<1> / LargeInt<2> as used by the path (tools/math/LargeInt1.pri, LargeInt2.pri): value semantics + toString */
/** LargeInt<3>, LargeInt<4> (tools/math/LargeInt.hpp:60-830: `u_int64_t value[precision]`, word 0 least significant): the spans 96 and 128
 *  of the reference's default build. The device counts k <= 63; these types let code instantiated with the larger spans (the reference's
 *  unit tests run DSK_check2 with span 96 and k = 31, TestDSK.cpp:254-305) use the same classes. */
template <int W> struct LargeInt {
    uint64_t value[W];
    LargeInt() { for (int i = 0; i < W; i++) value[i] = 0; }
    LargeInt(uint64_t v) { value[0] = v; for (int i = 1; i < W; i++) value[i] = 0; }
    uint64_t getVal() const { return value[0]; }
    void setVal(uint64_t v) { value[0] = v; for (int i = 1; i < W; i++) value[i] = 0; }
    void set128(unsigned __int128 v) { value[0] = (uint64_t)v; value[1] = (uint64_t)(v >> 64); for (int i = 2; i < W; i++) value[i] = 0; }
    bool operator<(const LargeInt& o) const { for (int i = W - 1; i >= 0; i--) if (value[i] != o.value[i]) return value[i] < o.value[i]; return false; }
    bool operator==(const LargeInt& o) const { for (int i = 0; i < W; i++) if (value[i] != o.value[i]) return false; return true; }
    bool operator!=(const LargeInt& o) const { return !(*this == o); }
    LargeInt& operator+=(const LargeInt& o) {
        unsigned carry = 0;
        for (int i = 0; i < W; i++) { const unsigned __int128 t = (unsigned __int128)value[i] + o.value[i] + carry; value[i] = (uint64_t)t; carry = (unsigned)(t >> 64); }
        return *this;
    }
    static size_t getSize() { return 64 * W; }
    static const char* getName() { return W == 3 ? "LargeInt<3>" : "LargeInt<4>"; }
    std::string toString(size_t k) const {
        std::string s(k, 'A');
        for (size_t i = 0; i < k; i++) { const size_t bit = 2 * i; s[k - 1 - i] = "ACTG"[(value[bit >> 6] >> (bit & 63)) & 3]; }
        return s;
    }
};

template <> struct LargeInt<1> {
    uint64_t value;
    LargeInt() : value(0) {}
    LargeInt(uint64_t v) : value(v) {}
    uint64_t getVal() const { return value; }
    void setVal(uint64_t v) { value = v; }
    void set128(unsigned __int128 v) { value = (uint64_t)v; }
    bool operator<(const LargeInt& o) const { return value < o.value; }
    bool operator==(const LargeInt& o) const { return value == o.value; }
    bool operator!=(const LargeInt& o) const { return value != o.value; }
    LargeInt& operator+=(const LargeInt& o) { value += o.value; return *this; }
    static size_t getSize() { return 64; }
    static const char* getName() { return "LargeInt<1>"; }
    std::string toString(size_t k) const { std::string s(k, 'A'); uint64_t t = value; for (size_t i = k; i-- > 0;) { s[i] = "ACTG"[t & 3]; t >>= 2; } return s; }
};
template <> struct LargeInt<2> {
    unsigned __int128 value;
    LargeInt() : value(0) {}
    LargeInt(uint64_t v) : value(v) {}
    uint64_t getVal() const { return (uint64_t)value; }
    void setVal(uint64_t v) { value = v; }
    void set128(unsigned __int128 v) { value = v; }
    bool operator<(const LargeInt& o) const { return value < o.value; }
    bool operator==(const LargeInt& o) const { return value == o.value; }
    bool operator!=(const LargeInt& o) const { return value != o.value; }
    LargeInt& operator+=(const LargeInt& o) { value += o.value; return *this; }
    static size_t getSize() { return 128; }
    static const char* getName() { return "LargeInt<2>"; }
    std::string toString(size_t k) const { std::string s(k, 'A'); unsigned __int128 t = value; for (size_t i = k; i-- > 0;) { s[i] = "ACTG"[(unsigned)(t & 3)]; t >>= 2; } return s; }
};

#define KMER_DEFAULT_SPAN 32
template <size_t span = KMER_DEFAULT_SPAN> struct Kmer {
    typedef LargeInt<(span + 31) / 32> Type;                       // Model.hpp:100
    /** {value, abundance}: 16 bytes (span 32) / 32 bytes (span 64) — the layout libgkc_hip.so returns (Abundance.hpp:68-129) */
    struct Count {
        Type value; CountNumber abundance;
        uint32_t _pad[sizeof(Type) == 16 ? 3 : 1];                 // explicit, zeroed: records are byte-reproducible on disk (16-byte alignment only for the 128-bit integer)
        Count() : abundance(0) { for (auto& x : _pad) x = 0; }
        Count(const Type& v, CountNumber a) : value(v), abundance(a) { for (auto& x : _pad) x = 0; }
        const Type& getValue() const { return value; }
        CountNumber getAbundance() const { return abundance; }
        bool operator<(const Count& o) const { return value < o.value; }
    };
};
static_assert(sizeof(Kmer<32>::Count) == 16, "Count layout must match the device records (k<=31)");
static_assert(sizeof(Kmer<64>::Count) == 32, "Count layout must match the device records (k<=63)");
static_assert(sizeof(Kmer<96>::Count) == 32 && sizeof(Kmer<128>::Count) == 40, "Count layout of the larger spans (Abundance.hpp:68-129: value, then a 4-byte abundance, 8-byte alignment)");

// ------------------------------------------------------------------------------------------------ Configuration
/** kmer/impl/Configuration.hpp:38-117 (fields the hot path reads) */
struct Configuration {
    size_t _kmerSize = 0, _minim_size = 0, _repartitionType = 0, _minimizerType = 0;
    uint64_t _max_memory = 0, _estimateSeqNb = 0, _estimateSeqTotalSize = 0, _estimateSeqMaxSize = 0, _kmersNb = 0, _volume = 0;
    uint32_t _nb_passes = 1, _nb_partitions = 0;
    CountNumber _abundance_min = 1, _abundance_max = 2147483647; uint32_t _histo_max = 10000;
    size_t _nb_banks = 1; bool _isComputed = false;
};

// ------------------------------------------------------------------------------------------------ Repartitor
/** minimizer -> partition table (PartiInfo.hpp:292-387). The builders restate PartiInfo.cpp:48-218 on statistics that were
 *  computed ON THE DEVICE (gkc_sample_minimizers); `save`/`load` use the byte layout of Repartitor::save (PartiInfo.cpp:271-295)
 *  in a plain file instead of an HDF5 stream. */
class Repartitor : public system::SmartPointer {
public:
    typedef uint16_t Value;
    Repartitor(int nbpart = 0, int minimsize = 0, int nbPass = 1) : _nbpart((uint16_t)nbpart), _mm(minimsize), _nb_minims(1ULL << (2 * minimsize)), _nbPass((uint16_t)nbPass) {}
    Value operator()(uint64_t minimizer_value) const { return _repart_table[minimizer_value]; }
    uint16_t getNbPasses() const { return _nbPass; }
    uint16_t getNbPartitions() const { return _nbpart; }
    uint64_t getNbMinimizers() const { return _nb_minims; }
    const std::vector<Value>& getTable() const { return _repart_table; }
    const uint32_t* getMinimizerFrequencies() const { return _freq_order.empty() ? nullptr : _freq_order.data(); }
    void setMinimizerFrequencies(const std::vector<uint32_t>& f) { _freq_order = f; }

    /** largest bin into the emptiest partition (computeDistrib, PartiInfo.cpp:48-106). Weights = kx-mers per minimizer of the sample
     *  (getNbKxmer_per_minim). The same standard-library calls with the same comparators as the reference (std::sort with compBin: size only;
     *  std::priority_queue with compSpaceTriple: space used only — PartiInfo.hpp:347-370): ties then fall the way they fall in the reference and
     *  the table is the reference's byte for byte (most of the 4^m bins are empty, so ties decide most of the table). */
    void computeDistrib(const std::vector<uint64_t>& weight) {
        typedef std::pair<uint64_t, uint64_t> ipair;                            // (bin size, bin number)
        struct itriple { uint64_t first, second, third; };
        struct compBin { bool operator()(ipair l, ipair r) { return l.first > r.first; } } comp_bins;
        struct compSpaceTriple { bool operator()(itriple l, itriple r) { return l.second > r.second; } };
        _repart_table.assign(_nb_minims, 0);
        std::vector<ipair> bin_size_vec;
        std::priority_queue<itriple, std::vector<itriple>, compSpaceTriple> pq;
        for (uint64_t ii = 0; ii < _nb_minims; ii++) bin_size_vec.push_back(ipair(weight[ii], ii));
        for (int jj = 0; jj < (int)_nbpart; jj++) pq.push(itriple{ (uint64_t)jj, 0, 0 });
        std::sort(bin_size_vec.begin(), bin_size_vec.end(), comp_bins);
        for (uint64_t cur = 0; cur < _nb_minims; cur++) {
            itriple smallest = pq.top(); pq.pop();
            _repart_table[bin_size_vec[cur].second] = (Value)smallest.first;
            smallest.second += bin_size_vec[cur].first; smallest.third++;
            pq.push(smallest);
        }
    }
    /** keeps minimizers in lexicographic order (justGroupLexi, PartiInfo.cpp:185-218; what bcalm2 needs) */
    void justGroupLexi(const std::vector<uint64_t>& nbKmers) {
        _repart_table.assign(_nb_minims, (Value)(_nbpart - 1));
        uint64_t sum = 0; for (uint64_t v : nbKmers) sum += v;
        const uint64_t mean = sum / _nbpart; uint64_t acc = 0, j = 0;
        for (uint64_t i = 0; i < _nb_minims; i++) {
            _repart_table[i] = (Value)std::min<uint64_t>(j, _nbpart - 1);        // the reference may write j == nbpart here; clamped
            acc += nbKmers[i];
            if (acc > mean) { acc = 0; if (j < _nbpart) j++; }
        }
    }
    /** frequency order grouping (justGroup, PartiInfo.cpp:131-183): counts = sorted (count, m-mer) pairs */
    void justGroup(const std::vector<uint64_t>& nbKmers, const std::vector<std::pair<uint32_t, uint32_t>>& counts) {
        _repart_table.assign(_nb_minims, (Value)(_nbpart - 1));
        uint64_t sum = 0; for (uint64_t v : nbKmers) sum += v;
        const uint64_t mean = sum / _nbpart; uint64_t acc = 0, j = 0;
        for (auto& c : counts) {
            _repart_table[c.second] = (Value)std::min<uint64_t>(j, _nbpart - 1);
            acc += nbKmers[c.second];
            if (acc > mean) { acc = 0; if (j < _nbpart) j++; }
        }
    }
    /** byte stream of Repartitor::save: u16 nbpart; u64 nb_minims; u16 nbPass; u16 table[]; u8 hasFreq; u32 magic (+ freq file) */
    /** the stream the reference stores as the u8 dataset minimizers/minimRepart (PartiInfo.cpp:271-295) */
    std::vector<uint8_t> stream() const {
        std::vector<uint8_t> v; const uint32_t magic = 0x12345678; const uint8_t hasf = !_freq_order.empty();
        auto put = [&](const void* p, size_t n) { v.insert(v.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
        put(&_nbpart, 2); put(&_nb_minims, 8); put(&_nbPass, 2); put(_repart_table.data(), 2 * _nb_minims); put(&hasf, 1); put(&magic, 4);
        return v;
    }
    std::vector<uint8_t> frequencyStream() const {
        std::vector<uint8_t> v; const uint32_t magic = 0x12345678;
        if (_freq_order.empty()) return v;
        v.insert(v.end(), (const uint8_t*)_freq_order.data(), (const uint8_t*)_freq_order.data() + 4 * _nb_minims);
        v.insert(v.end(), (const uint8_t*)&magic, (const uint8_t*)&magic + 4); return v;
    }
    void save(const std::string& path) const {
        std::ofstream os(path, std::ios::binary);
        const uint32_t magic = 0x12345678; const uint8_t hasf = !_freq_order.empty();
        os.write((const char*)&_nbpart, 2); os.write((const char*)&_nb_minims, 8); os.write((const char*)&_nbPass, 2);
        os.write((const char*)_repart_table.data(), 2 * _nb_minims); os.write((const char*)&hasf, 1); os.write((const char*)&magic, 4);
        if (hasf) { std::ofstream of(path + ".minimFrequency", std::ios::binary); of.write((const char*)_freq_order.data(), 4 * _nb_minims); of.write((const char*)&magic, 4); }
    }
    void load(const std::string& path) {
        std::ifstream is(path, std::ios::binary);
        if (!is) throw system::Exception("Unable to load Repartitor (minimRepart) '%s'", path.c_str());
        uint8_t hasf = 0; uint32_t magic = 0;
        is.read((char*)&_nbpart, 2); is.read((char*)&_nb_minims, 8); is.read((char*)&_nbPass, 2);
        _repart_table.resize(_nb_minims); is.read((char*)_repart_table.data(), 2 * _nb_minims);
        is.read((char*)&hasf, 1); is.read((char*)&magic, 4);
        if (magic != 0x12345678) throw system::Exception("Unable to load Repartitor (minimRepart), possibly due to bad format.");
        _mm = 0; while ((1ULL << (2 * _mm)) < _nb_minims) _mm++;
        if (hasf) { std::ifstream f(path + ".minimFrequency", std::ios::binary); _freq_order.resize(_nb_minims); f.read((char*)_freq_order.data(), 4 * _nb_minims); }
    }
    void setTable(const std::vector<Value>& t) { _repart_table = t; }
private:
    uint16_t _nbpart; int _mm; uint64_t _nb_minims; uint16_t _nbPass;
    std::vector<Value> _repart_table; std::vector<uint32_t> _freq_order;
};

// ------------------------------------------------------------------------------------------------ ICountProcessor + defaults
/** plug-in protocol of the counting stage (kmer/api/ICountProcessor.hpp:92-185): identical calls in identical order */
template <size_t span = KMER_DEFAULT_SPAN>
class ICountProcessor : public system::SmartPointer {
public:
    typedef typename Kmer<span>::Type Type;
    virtual void begin(const Configuration& config) = 0;
    virtual void end() = 0;
    virtual void beginPass(size_t passId) = 0;
    virtual void endPass(size_t passId) = 0;
    virtual ICountProcessor* clone() = 0;
    virtual void finishClones(std::vector<ICountProcessor<span>*>& clones) = 0;
    virtual void beginPart(size_t passId, size_t partId, size_t cacheSize, const char* name) = 0;
    virtual void endPart(size_t passId, size_t partId) = 0;
    virtual bool process(size_t partId, const Type& kmer, const CountVector& count, CountNumber sum = 0) = 0;
    virtual std::string getName() const = 0;
    virtual tools::misc::Properties getProperties() const { return tools::misc::Properties(); }
    /** EXTENSION (bulk sink): processors that can take a whole ascending Count[] block at once override
     *  this; the default forwards record by record to process(), so custom processors keep working unchanged. */
    virtual void processBulk(size_t partId, const typename Kmer<span>::Count* counts, size_t n) {
        CountVector v(1);
        for (size_t i = 0; i < n; i++) { v[0] = counts[i].abundance; process(partId, counts[i].value, v, counts[i].abundance); }
    }
};

template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorAbstract : public ICountProcessor<span> {
public:
    void begin(const Configuration&) {}
    void end() {}
    void beginPass(size_t) {}
    void endPass(size_t) {}
    void finishClones(std::vector<ICountProcessor<span>*>&) {}
    void beginPart(size_t, size_t, size_t, const char*) {}
    void endPart(size_t, size_t) {}
};

/** CountProcessorHistogram (kmer/impl/CountProcessorHistogram.hpp:173-184; Histogram::inc, Histogram.hpp:92) */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorHistogram : public CountProcessorAbstract<span> {
public:
    typedef typename Kmer<span>::Type Type;
    explicit CountProcessorHistogram(size_t histoMax = 10000, std::vector<uint64_t>* shared = nullptr)
        : _length(histoMax), _own(histoMax + 1, 0), _shared(shared ? shared : &_own) {}
    ICountProcessor<span>* clone() { return new CountProcessorHistogram(_length, _shared); }
    bool process(size_t, const Type&, const CountVector&, CountNumber sum) { (*_shared)[(size_t)sum >= _length ? _length : (size_t)sum]++; return true; }
    std::string getName() const { return "histogram"; }
    const std::vector<uint64_t>& getHistogram() const { return *_shared; }
    /** EXTENSION (device-assisted chain): the abundance histogram the device accumulated while counting (gkc_histogram) */
    void addCounts(const uint64_t* h, size_t nBins) { for (size_t i = 0; i < nBins && i <= _length; i++) (*_shared)[i] += h[i]; }

    /** Histogram::compute_threshold (tools/misc/impl/Histogram.cpp:61-190): smoothed histogram, first increase, first peak after it, cutoff =
     *  the minimum between them, capped where 25 % of the k-mer volume would be eliminated, floored by min_auto_threshold */
    void compute_threshold(int min_auto_threshold = 2 /* -abundance-min-threshold default, SortingCountAlgorithm.cpp:212 */) {
        const std::vector<uint64_t>& h = *_shared; const size_t L = _length;
        std::vector<uint64_t> sm(L + 1, 0);
        uint64_t sum_allk = 0;
        _cutoff = 0; _nbsolids = 0; _firstPeak = 0; _ratio_weak_volume = 0;
        if (L >= 2) { sm[1] = (uint64_t)(0.6 * (double)h[1] + 0.4 * (double)h[2]); sum_allk += h[1] * 1; }
        int first_inc = -1, idx_max = -1; uint64_t max_val = 0;
        for (size_t i = 2; i < L; i++) {
            sum_allk += h[i] * i;
            sm[i] = (uint64_t)(0.2 * (double)h[i - 1] + 0.6 * (double)h[i] + 0.2 * (double)h[i + 1]);
            if (first_inc == -1 && sm[i - 1] < sm[i]) first_inc = (int)i - 1;
            if (first_inc > 0 && sm[i] > max_val) { max_val = sm[i]; idx_max = (int)i; }
        }
        sum_allk += h[L] * L;
        if (first_inc == -1) { _cutoff = (size_t)min_auto_threshold; return; }
        _firstPeak = (size_t)idx_max;
        uint64_t min_val = 10000000000ULL; int idx_min = -1;
        for (int i = first_inc; i <= idx_max; i++) if (sm[i] < min_val) { min_val = sm[i]; idx_min = i; }
        if (idx_min != -1) _cutoff = (size_t)idx_min;
        uint64_t sum_elim = 0; size_t max_cutoff = 0;
        for (size_t i = 0; i < L + 1; i++) { sum_elim += h[i] * i; if ((double)sum_elim / sum_allk >= 0.25) { max_cutoff = i + 1; break; } }
        if (_cutoff > max_cutoff) _cutoff = max_cutoff;
        if (_cutoff < (size_t)min_auto_threshold) _cutoff = (size_t)min_auto_threshold;
        for (size_t i = _cutoff; i < L + 1; i++) _nbsolids += h[i];
        uint64_t vol_weak = 0, vol_total = 0;
        for (size_t i = 0; i < _cutoff; i++) vol_weak += h[i] * i;
        for (size_t i = 0; i < L + 1; i++) vol_total += h[i] * i;
        _ratio_weak_volume = (float)vol_weak / (float)vol_total;
    }
    size_t get_solid_cutoff() const { return _cutoff; }
    uint64_t get_nbsolids_auto() const { return _nbsolids; }
    size_t get_first_peak() const { return _firstPeak; }
    float get_ratio_weak() const { return _ratio_weak_volume; }
private:
    size_t _length; std::vector<uint64_t> _own; std::vector<uint64_t>* _shared;
    size_t _cutoff = 0; uint64_t _nbsolids = 0; size_t _firstPeak = 0; float _ratio_weak_volume = 0;
};

/** CountProcessorSoliditySum (kmer/impl/CountProcessorSolidity.hpp:60-190): closed interval test on the sum */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorSoliditySum : public CountProcessorAbstract<span> {
public:
    typedef typename Kmer<span>::Type Type;
    struct Totals { uint64_t total = 0, ok = 0; };
    CountProcessorSoliditySum(CountNumber amin, CountNumber amax, Totals* shared = nullptr) : _min(amin), _max(amax), _shared(shared ? shared : &_own) {}
    ICountProcessor<span>* clone() { return new CountProcessorSoliditySum(_min, _max, _shared); }
    bool process(size_t, const Type&, const CountVector&, CountNumber sum) { _shared->total++; const bool ok = sum >= _min && sum <= _max; if (ok) _shared->ok++; return ok; }
    std::string getName() const { return "sum"; }
    tools::misc::Properties getProperties() const {
        tools::misc::Properties p; p.add("kmers_nb_distinct", "%llu", (unsigned long long)_shared->total); p.add("kmers_nb_solid", "%llu", (unsigned long long)_shared->ok);
        p.add("kmers_nb_weak", "%llu", (unsigned long long)(_shared->total - _shared->ok)); return p; }
    const Totals& totals() const { return *_shared; }
    /** EXTENSION (device-assisted chain): totals of a pass whose solidity window was applied on the device */
    void addTotals(uint64_t total, uint64_t ok) { _shared->total += total; _shared->ok += ok; }
    CountNumber getAbundanceMax() const { return _max; }
    /** CountProcessorSolidityInfo::setAbundanceMin (CountProcessorSolidity.hpp:60-75): the automatic cutoff arrives after the cutoff processor's pass */
    void setAbundanceMin(CountNumber m) { _min = m; }
    CountNumber getAbundanceMin() const { return _min; }
private:
    CountNumber _min, _max; Totals _own; Totals* _shared;
};

/** CountProcessorDump (kmer/impl/CountProcessorDump.hpp:60-180): dataset index = part + pass*nb_partitions (:131);
 *  nb_partitions*nb_passes datasets exist even if empty (:85-95). Datasets are kept in memory (the reference appends them
 *  to dsk/solid/<i> through BagCache -> HDF5); saveRaw() writes each one as a raw Count[] file. */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorDump : public CountProcessorAbstract<span> {
public:
    typedef typename Kmer<span>::Type Type; typedef typename Kmer<span>::Count Count;
    typedef std::vector<std::vector<Count>> Store;
    explicit CountProcessorDump(size_t kmerSize, Store* shared = nullptr, size_t nbParts = 0) : _kmerSize(kmerSize), _shared(shared ? shared : &_own), _nbPartsPerPass(nbParts), _cur(0) {}
    void begin(const Configuration& config) { _nbPartsPerPass = config._nb_partitions; _shared->assign((size_t)config._nb_partitions * config._nb_passes, std::vector<Count>()); }
    ICountProcessor<span>* clone() { return new CountProcessorDump(_kmerSize, _shared, _nbPartsPerPass); }
    void beginPart(size_t passId, size_t partId, size_t, const char*) { _cur = partId + passId * _nbPartsPerPass; }
    bool process(size_t, const Type& kmer, const CountVector&, CountNumber sum) { (*_shared)[_cur].push_back(Count(kmer, sum)); return true; }
    void processBulk(size_t, const Count* counts, size_t n) { auto& d = (*_shared)[_cur]; d.insert(d.end(), counts, counts + n); }
    std::string getName() const { return "dump"; }
    const Store& getSolidCounts() const { return *_shared; }
    uint64_t getNbItems() const { uint64_t n = 0; for (auto& d : *_shared) n += d.size(); return n; }
    void saveRaw(const std::string& prefix) const {
        for (size_t i = 0; i < _shared->size(); i++) { std::ofstream os(prefix + ".solid." + std::to_string(i), std::ios::binary); os.write((const char*)(*_shared)[i].data(), (*_shared)[i].size() * sizeof(Count)); }
    }
private:
    size_t _kmerSize; Store _own; Store* _shared; size_t _nbPartsPerPass; size_t _cur;
};

/** CountProcessorChain (kmer/impl/CountProcessorChain.hpp:60-160): a k-mer goes down the chain while processors return true */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorChain : public ICountProcessor<span> {
public:
    typedef typename Kmer<span>::Type Type;
    CountProcessorChain(const std::vector<ICountProcessor<span>*>& items) : _items(items) { for (auto* i : _items) i->use(); }
    ~CountProcessorChain() { for (auto* i : _items) i->forget(); }
    void begin(const Configuration& c) { for (auto* i : _items) i->begin(c); }
    void end() { for (auto* i : _items) i->end(); }
    void beginPass(size_t p) { for (auto* i : _items) i->beginPass(p); }
    void endPass(size_t p) { for (auto* i : _items) i->endPass(p); }
    ICountProcessor<span>* clone() { std::vector<ICountProcessor<span>*> c; for (auto* i : _items) c.push_back(i->clone()); return new CountProcessorChain(c); }
    void finishClones(std::vector<ICountProcessor<span>*>& clones) {
        for (size_t i = 0; i < _items.size(); i++) { std::vector<ICountProcessor<span>*> sub; for (auto* c : clones) if (auto* ch = dynamic_cast<CountProcessorChain*>(c)) sub.push_back(ch->_items[i]); _items[i]->finishClones(sub); }
    }
    void beginPart(size_t a, size_t b, size_t c, const char* n) { for (auto* i : _items) i->beginPart(a, b, c, n); }
    void endPart(size_t a, size_t b) { for (auto* i : _items) i->endPart(a, b); }
    bool process(size_t partId, const Type& kmer, const CountVector& count, CountNumber sum = 0) {
        if (sum == 0) for (CountNumber c : count) sum += c;
        bool res = true;
        for (size_t i = 0; res && i < _items.size(); i++) res = _items[i]->process(partId, kmer, count, sum);
        return res;
    }
    std::string getName() const { return "chain"; }
    const std::vector<ICountProcessor<span>*>& items() const { return _items; }
    template <class T> T* get() const { for (auto* i : _items) if (T* t = dynamic_cast<T*>(i)) return t; return nullptr; }
private:
    std::vector<ICountProcessor<span>*> _items;
};

/** CountProcessorCutoff for one bank (kmer/impl/CountProcessorCutoff.hpp:38-160): histogram of every distinct k-mer; endPass computes the
 *  automatic abundance threshold with Histogram::compute_threshold(3) (:92-103). Used for "-abundance-min auto". */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorCutoff : public CountProcessorAbstract<span> {
public:
    typedef typename Kmer<span>::Type Type;
    explicit CountProcessorCutoff(size_t histoMax = 10000) : _histo(new CountProcessorHistogram<span>(histoMax)) { _histo->use(); }
    explicit CountProcessorCutoff(CountProcessorHistogram<span>* h) : _histo(h) { _histo->use(); }
    ~CountProcessorCutoff() { _histo->forget(); }
    ICountProcessor<span>* clone() { return new CountProcessorCutoff(static_cast<CountProcessorHistogram<span>*>(_histo->clone())); }
    bool process(size_t partId, const Type& kmer, const CountVector& count, CountNumber) {
        CountNumber sum = 0; for (CountNumber c : count) sum += c;
        return _histo->process(partId, kmer, count, sum);
    }
    void endPass(size_t) { _histo->compute_threshold(3); _cutoffs.assign(1, (CountNumber)_histo->get_solid_cutoff()); }
    std::string getName() const { return "cutoff"; }
    tools::misc::Properties getProperties() const { tools::misc::Properties p; std::string v; for (CountNumber c : _cutoffs) v += std::to_string(c) + " "; p.setStr("values", v); return p; }
    std::vector<CountNumber> getCutoffs() const { return _cutoffs; }
private:
    CountProcessorHistogram<span>* _histo; std::vector<CountNumber> _cutoffs;
};
/** the proxy that links the cutoff processor to the dsk processor (CountProcessorCustomProxy, SortingCountAlgorithm.cpp:418-444): every call goes to the
 *  cutoff processor; after its pass the cutoff becomes the abundance min of the dsk chain's solidity processor, which runs next */
template <size_t span = KMER_DEFAULT_SPAN>
class CountProcessorCutoffProxy : public ICountProcessor<span> {
public:
    typedef typename Kmer<span>::Type Type;
    CountProcessorCutoffProxy(CountProcessorCutoff<span>* cutoff, ICountProcessor<span>* dsk) : _cutoff(cutoff), _dsk(dsk) { _cutoff->use(); }
    ~CountProcessorCutoffProxy() { _cutoff->forget(); }
    void begin(const Configuration& c) { _cutoff->begin(c); }
    void end() { _cutoff->end(); }
    void beginPass(size_t p) { _cutoff->beginPass(p); }
    void endPass(size_t p) {
        _cutoff->endPass(p);
        if (auto* ch = dynamic_cast<CountProcessorChain<span>*>(_dsk)) if (auto* sol = ch->template get<CountProcessorSoliditySum<span>>())
            if (!_cutoff->getCutoffs().empty()) sol->setAbundanceMin(_cutoff->getCutoffs()[0]);
    }
    ICountProcessor<span>* clone() { return _cutoff->clone(); }
    void finishClones(std::vector<ICountProcessor<span>*>& clones) { _cutoff->finishClones(clones); }
    void beginPart(size_t a, size_t b, size_t c, const char* n) { _cutoff->beginPart(a, b, c, n); }
    void endPart(size_t a, size_t b) { _cutoff->endPart(a, b); }
    bool process(size_t partId, const Type& kmer, const CountVector& count, CountNumber sum = 0) { return _cutoff->process(partId, kmer, count, sum); }
    std::string getName() const { return "cutoffs_auto"; }
    tools::misc::Properties getProperties() const { return _cutoff->getProperties(); }
    CountProcessorCutoff<span>* cutoff() { return _cutoff; }
private:
    CountProcessorCutoff<span>* _cutoff; ICountProcessor<span>* _dsk;
};

// ------------------------------------------------------------------------------------------------ SortingCountAlgorithm
/** DSK driver on the GPU. Same constructors / execute() / accessors as kmer/impl/SortingCountAlgorithm.hpp:85-192.
 *  execute() = configure (Configuration + Repartitor if not injected, .cpp:525-625) -> processors begin -> per pass:
 *  fillPartitions (Stage A on the device) -> fillSolidKmers (Stage B on the device, then one clone per partition fed in
 *  ascending k-mer order: beginPart / process... / endPart, finishClones) -> processors end; getInfo() keys as :728-780. */
template <size_t span = KMER_DEFAULT_SPAN>
class SortingCountAlgorithm {
public:
    typedef typename Kmer<span>::Type Type; typedef typename Kmer<span>::Count Count;
    typedef ICountProcessor<span> CountProcessor;

    static tools::misc::IProperties* getDefaultProperties() {     // SortingCountAlgorithm.cpp:202-236 defaults
        auto* p = new tools::misc::IProperties();
        p->setInt(STR_KMER_SIZE, 31); p->setInt(STR_KMER_ABUNDANCE_MIN, 2); p->setInt(STR_KMER_ABUNDANCE_MAX, 2147483647);
        p->setInt(STR_MINIMIZER_SIZE, 10); p->setInt(STR_MINIMIZER_TYPE, 0); p->setInt(STR_REPARTITION_TYPE, 0);
        p->setInt(STR_MAX_MEMORY, 5000); p->setInt(STR_HISTOGRAM_MAX, 10000); p->setStr(STR_URI_OUTPUT, "");
        p->setInt(STR_NB_PARTITIONS, 0); p->setInt(STR_NB_PASSES, 0); p->setInt(STR_GPU_DEVICE, 0);
        return p;
    }
    /** options of the step (SortingCountAlgorithm.hpp:126): the reference's names for what this path reads, plus the three extensions above */
    static tools::misc::IOptionsParser* getOptionsParser(bool mandatory = true) {
        typedef tools::misc::IOptionsParser::Option O;
        auto* p = new tools::misc::IOptionsParser("kmer count");
        p->push_back(O{ STR_URI_INPUT, "reads file (FASTA / FASTQ, optionally gzipped)", mandatory, false, "", 1 });
        p->push_back(O{ STR_KMER_SIZE, "size of a k-mer", false, true, "31", 1 });
        p->push_back(O{ STR_KMER_ABUNDANCE_MIN, "smallest abundance of a solid k-mer", false, true, "2", 1 });
        p->push_back(O{ STR_KMER_ABUNDANCE_MAX, "largest abundance of a solid k-mer", false, true, "2147483647", 1 });
        p->push_back(O{ STR_HISTOGRAM_MAX, "number of abundance values the histogram keeps", false, true, "10000", 1 });
        p->push_back(O{ STR_MAX_MEMORY, "host memory budget in MBytes (kept for compatibility; batches are sized from the HBM)", false, true, "5000", 1 });
        p->push_back(O{ STR_URI_OUTPUT, "output prefix", false, false, "", 1 });
        p->push_back(O{ STR_MINIMIZER_TYPE, "minimizer order (0 = lexicographic / KMC2, 1 = by frequency)", false, true, "0", 1 });
        p->push_back(O{ STR_MINIMIZER_SIZE, "size of a minimizer", false, true, "10", 1 });
        p->push_back(O{ STR_REPARTITION_TYPE, "minimizer repartition (0 = unordered, 1 = ordered)", false, true, "0", 1 });
        p->push_back(O{ STR_NB_PARTITIONS, "force the number of partitions (0 = derive)", false, true, "0", 1 });
        p->push_back(O{ STR_NB_PASSES, "number of passes (0 = derive from the HBM)", false, true, "0", 1 });
        p->push_back(O{ STR_GPU_DEVICE, "HIP device index", false, true, "0", 1 });
        return p;
    }
    /** (params) only: the reference's first constructor (SortingCountAlgorithm.hpp:96); without a bank execute() refuses */
    explicit SortingCountAlgorithm(tools::misc::IProperties* params = 0)
        : _bank(nullptr), _params(params ? *params : tools::misc::IProperties()), _repartitor(nullptr), _ctx(nullptr), _textRefused(false) {}
    SortingCountAlgorithm(bank::IBank* bank, tools::misc::IProperties* params)
        : _bank(bank), _params(*params), _repartitor(nullptr), _ctx(nullptr), _textRefused(false) { _bank->use(); }
    SortingCountAlgorithm(bank::IBank* bank, const Configuration& config, Repartitor* repartitor, std::vector<CountProcessor*> processors, tools::misc::IProperties* params)
        : _bank(bank), _params(*params), _config(config), _repartitor(repartitor), _processors(processors), _ctx(nullptr), _textRefused(false) {
        _bank->use(); if (_repartitor) _repartitor->use(); for (auto* p : _processors) p->use();
    }
    ~SortingCountAlgorithm() {
        if (_ctx) gkc_destroy(_ctx);
        if (_bank) _bank->forget();
        if (_repartitor) _repartitor->forget();
        for (auto* p : _processors) p->forget();
    }
    void addProcessor(CountProcessor* p) { p->use(); _processors.push_back(p); }
    size_t getProcessorNumber() const { return _processors.size(); }
    CountProcessor* getProcessor(size_t i) { return _processors[i]; }
    const Configuration& getConfig() const { return _config; }
    Repartitor* getRepartitor() { return _repartitor; }
    const tools::misc::Properties* getInfo() const { return &_info; }
    /** solid counts of the default dump processor (getSolidCounts(), .hpp:168) */
    const typename CountProcessorDump<span>::Store& getSolidCounts() {
        for (auto* p : _processors) { if (auto* ch = dynamic_cast<CountProcessorChain<span>*>(p)) if (auto* d = ch->template get<CountProcessorDump<span>>()) return d->getSolidCounts();
                                      if (auto* d = dynamic_cast<CountProcessorDump<span>*>(p)) return d->getSolidCounts(); }
        throw system::Exception("no dump processor");
    }
    /** the solid k-mers alone, dataset after dataset, ascending inside each (getSolidKmers(), .hpp:171: the Iterable's content) */
    std::vector<Type> getSolidKmers() {
        std::vector<Type> v;
        for (auto& part : getSolidCounts()) for (auto& c : part) v.push_back(c.value);
        return v;
    }
    gkc_ctx* context() { return _ctx; }
    /** the default histogram -> solidity -> dump chain among the processors (it follows the cutoff proxy in "auto" mode), or null */
    CountProcessorChain<span>* getDskChain() { for (auto* p : _processors) if (auto* ch = dynamic_cast<CountProcessorChain<span>*>(p)) return ch; return nullptr; }

    /** the .h5 the DSK step of dbgh5 leaves (tools/storage/impl/StorageHDF5.hpp, CountProcessorDump.hpp:85-131, Histogram::save, Repartitor::save):
     *  /dsk/solid/<dataset> Count datasets, /histogram/{histogram,cutoff,nbsolidsforcutoff}, /minimizers/minimRepart(+minimFrequency),
     *  /configuration, string attributes. Written by the native writer gkc_h5.hpp (contiguous datasets; same names, types and values). */
    void saveH5(const std::string& path) {
        auto* chain = getDskChain();
        auto* dump = chain ? chain->template get<CountProcessorDump<span>>() : nullptr;
        auto* histo = chain ? chain->template get<CountProcessorHistogram<span>>() : nullptr;
        if (!dump || !histo) throw system::Exception("saveH5 needs the default processor chain");
        const auto& store = dump->getSolidCounts();
        gkc_h5::File f(std::max<size_t>(store.size(), 16));
        auto xml = [](const std::string& root, const std::map<std::string, std::string>& kv) {
            std::string x = "\n<" + root + ">\n"; for (auto& p : kv) x += "   <" + p.first + ">" + p.second + "</" + p.first + ">\n"; return x + "</" + root + ">"; };
        uint64_t nb_solid = 0; for (auto& d : store) nb_solid += d.size();
        f.set_attribute("/", "kmer_size", std::to_string(_config._kmerSize));
        f.set_attribute("/", "nb_solid_kmers", std::to_string(nb_solid));
        f.set_attribute("/", "state", "7");                                   // Graph.cpp: init | bank converter | sorting count done
        f.set_attribute("/", "xml", xml("gatb-core-library", { {"version", "gkc-" + std::string(gkc_version())}, {"build_kmer_size", "32 64"} }));
        f.set_attribute("/configuration", "xml", xml("configuration", {
            {"kmer_size", std::to_string(_config._kmerSize)}, {"mini_size", std::to_string(_config._minim_size)}, {"solidity_kind", "sum"},
            {"abundance_min", std::to_string(_config._abundance_min)}, {"abundance_max", std::to_string(_config._abundance_max)},
            {"estimated_sequence_number", std::to_string(_config._estimateSeqNb)}, {"estimated_kmers_number", std::to_string(_config._kmersNb)},
            {"max_memory", std::to_string(_config._max_memory)}, {"nb_passes", std::to_string(_config._nb_passes)}, {"nb_partitions", std::to_string(_config._nb_partitions)},
            {"nb_bits_per_kmer", std::to_string(span <= 32 ? 64 : 128)}, {"minimizer_type", std::to_string(_config._minimizerType)},
            {"repartition_type", std::to_string(_config._repartitionType)}, {"nb_banks", std::to_string(_config._nb_banks)} }));
        f.set_attribute("/dsk", "kmer_size", std::to_string(_config._kmerSize));
        f.set_attribute("/dsk", "xml", xml("dsk", _info.map()));
        f.set_attribute("/dsk/solid", "nb_partitions", std::to_string(store.size()));
        // Count = {value, abundance} (Abundance.hpp:108-125); value: u64, or the 128-bit integer of LargeInt<2>::hdf5 (LargeInt2.pri:137-142)
        const gkc_h5::Type vt = span <= 32 ? gkc_h5::Type::integer(8, false)                               // LargeInt<1>: H5T_NATIVE_UINT64
                                           : gkc_h5::Type::integer((uint32_t)sizeof(Type), true, 8 * (uint32_t)sizeof(Type));   // LargeInt<N>: H5T_NATIVE_INT with precision 64 N (LargeInt2.pri:137-142, LargeInt.hpp:655-660)
        const gkc_h5::Type ct = gkc_h5::Type::compound((uint32_t)sizeof(Count), { {"value", {0, vt}}, {"abundance", {(uint32_t)sizeof(Type), gkc_h5::Type::integer(4, false)}} });
        for (size_t i = 0; i < store.size(); i++) f.add_dataset("/dsk/solid/" + std::to_string(i), ct, store[i].data(), store[i].size());
        // Histogram::save: entries 1..length as {u16 index (stored on 4 bytes), u64 abundance}, then cutoff and nbsolids (Histogram.cpp:192-240)
        struct HEntry { uint32_t index; uint32_t pad; uint64_t abundance; };
        const auto& hv = histo->getHistogram();
        std::vector<HEntry> he; for (size_t i = 1; i < hv.size(); i++) he.push_back(HEntry{ (uint32_t)i, 0, hv[i] });
        const gkc_h5::Type ht = gkc_h5::Type::compound(16, { {"index", {0, gkc_h5::Type::integer(4, false)}}, {"abundance", {8, gkc_h5::Type::integer(8, false)}} });
        f.add_dataset("/histogram/histogram", ht, he.data(), he.size());
        histo->compute_threshold(_params.has("-abundance-min-threshold") ? (int)_params.getInt("-abundance-min-threshold") : 2);
        const uint64_t cutoff = histo->get_solid_cutoff(), nbs = histo->get_nbsolids_auto();
        f.add_dataset("/histogram/cutoff", gkc_h5::Type::integer(8, false), &cutoff, 1);
        f.add_dataset("/histogram/nbsolidsforcutoff", gkc_h5::Type::integer(8, false), &nbs, 1);
        const std::vector<uint8_t> rs = _repartitor->stream(), fs = _repartitor->frequencyStream();
        f.add_dataset("/minimizers/minimRepart", gkc_h5::Type::integer(1, false), rs.data(), rs.size());
        if (!fs.empty()) f.add_dataset("/minimizers/minimFrequency", gkc_h5::Type::integer(1, false), fs.data(), fs.size());
        const auto& img = f.finish();
        std::ofstream os(path, std::ios::binary);
        if (!os) throw system::Exception("Unable to create '%s'", path.c_str());
        os.write((const char*)img.data(), (std::streamsize)img.size());
    }

    /** file banks: the text goes to the device parser in chunks (gkc_push_fastx). Text the device parser refuses (GKC_ERR_FORMAT:
     *  e.g. multi-line FASTQ) is read by the host walk instead, like every bank of the reference; returns false in that case. */
    bool pushText(uint32_t pass) {
        auto* bf = dynamic_cast<bank::BankFasta*>(_bank);
        if (!bf || _textRefused) return false;
        const std::string& t = bf->text();
        const size_t CHUNK = (size_t)1 << 30;
        size_t pos = 0;
        while (pos < t.size()) {
            const size_t len = std::min(CHUNK, t.size() - pos); const int final_chunk = pos + len >= t.size();
            uint64_t consumed = 0;
            const int rc = gkc_push_fastx(_ctx, t.data() + pos, len, final_chunk, &consumed);
            if (rc == GKC_ERR_FORMAT && pos == 0) {                // nothing pushed yet: restart the pass with the host reader
                _textRefused = true; check(gkc_begin_pass(_ctx, pass)); return false;
            }
            check(rc);
            if (!final_chunk && consumed == 0) throw system::Exception("a record of the bank is larger than %zu bytes", CHUNK);
            pos += final_chunk ? len : (size_t)consumed;
        }
        return true;
    }

    void execute() {
        if (!_bank) throw system::Exception("SortingCountAlgorithm: no bank to count");
        configure();
        const uint32_t P = _config._nb_partitions;
        for (auto* p : _processors) p->begin(_config);
        auto bases = [&]() -> const std::vector<char>& { return _bank->bases(); };
        auto offs = [&]() -> const std::vector<uint64_t>& { return _bank->offsets(); };
        // the partition buffers live in page-locked memory (gkc_host_alloc): the Count[] arrays come over at PCIe rate
        struct Pinned {
            void* p = nullptr; size_t bytes = 0;
            ~Pinned() { if (p) gkc_host_free(p); }
            void* need(size_t b) {
                if (b > bytes) {
                    if (p) gkc_host_free(p);
                    p = nullptr; bytes = 0;
                    const size_t nb = b + b / 4 + 4096;
                    if (gkc_host_alloc(&p, nb) != GKC_OK) throw system::Exception("unable to allocate %zu bytes of page-locked memory", nb);
                    bytes = nb;
                }
                return p;
            }
        } pin, pin_narrow;
        // Device-assisted default chain: when the processors are exactly the default histogram -> solidity(sum) -> dump chain, the
        // solidity window and the histogram are applied on the device (gkc_set_solidity / gkc_histogram) and only SOLID records cross PCIe, streamed
        // into a page-locked sink while Stage B is still counting (gkc_set_host_sink, gkc_finish_pass_async, gkc_wait_partition): the dump clones
        // get whole ascending Count[] blocks in partition order. Any other processor set sees every distinct k-mer through the generic path below.
        CountProcessorHistogram<span>* devHisto = nullptr; CountProcessorSoliditySum<span>* devSol = nullptr;
        if (_deviceChain) { auto* ch = static_cast<CountProcessorChain<span>*>(_processors[0]); devHisto = ch->template get<CountProcessorHistogram<span>>(); devSol = ch->template get<CountProcessorSoliditySum<span>>(); }
        Pinned sinkMem;
        if (_deviceChain) {
            const uint64_t capMB = _params.has("-host-sink-mb") ? (uint64_t)_params.getInt("-host-sink-mb") : 16384;
            const uint64_t bound = (_config._kmersNb / std::max<uint32_t>(_config._nb_passes, 1) + 4096) * sizeof(Count);       // every k-mer distinct and solid
            const uint64_t want = std::max<uint64_t>(std::min<uint64_t>(bound, capMB << 20), 1 << 20);
            void* sp = nullptr;
            for (uint64_t b = want; !sp && b >= (1 << 20); b /= 2) if (gkc_host_alloc(&sp, b) == GKC_OK) { sinkMem.p = sp; sinkMem.bytes = (size_t)b; }
            if (sinkMem.p) check(gkc_set_host_sink(_ctx, sinkMem.p, sinkMem.bytes));
        }
        for (uint32_t pass = 0; pass < _config._nb_passes; pass++) {
            check(gkc_begin_pass(_ctx, pass));
            if (!pushText(pass)) check(gkc_push_reads(_ctx, bases().data(), offs().data(), offs().size() - 1));   // fillPartitions
            if (_deviceChain) {
                check(gkc_finish_pass_async(_ctx));                                                   // fillSolidKmers: Stage B runs while the clones below are fed
                CountProcessor* proc = _processors[0];
                proc->beginPass(pass);
                std::vector<CountProcessor*> clones;
                try {
                    for (uint32_t p = 0; p < P; p++) {
                        const void* hp = nullptr; uint64_t ns = 0;
                        check(gkc_wait_partition(_ctx, pass, p, &hp, &ns));
                        const Count* buf = static_cast<const Count*>(hp);
                        if (!buf && ns) {                                                             // did not fit the sink: plain fetch
                            Count* b2 = static_cast<Count*>(pin.need((size_t)ns * sizeof(Count))); uint64_t got = 0;
                            check(gkc_partition_counts(_ctx, pass, p, b2, ns, &got)); buf = b2;
                        }
                        CountProcessor* clone = proc->clone(); clone->use(); clones.push_back(clone);
                        clone->beginPart(pass, p, 4096, "vector");
                        if (ns) static_cast<CountProcessorChain<span>*>(clone)->template get<CountProcessorDump<span>>()->processBulk(p, buf, (size_t)ns);
                        clone->endPart(pass, p);
                    }
                } catch (...) { (void)gkc_finish_pass_wait(_ctx); for (auto* c : clones) c->forget(); throw; }
                check(gkc_finish_pass_wait(_ctx));
                proc->finishClones(clones);
                for (auto* c : clones) c->forget();
                proc->endPass(pass);
                if (_releasePasses) check(gkc_release_pass(_ctx, pass));
                continue;
            }
            check(gkc_finish_pass(_ctx));                                                             // fillSolidKmers (device part)
            for (auto* proc : _processors) {
                proc->beginPass(pass);
                std::vector<CountProcessor*> clones;
                for (uint32_t p = 0; p < P; p++) {
                    uint64_t ns = 0, nd = 0, nk = 0;
                    check(gkc_partition_info(_ctx, pass, p, &ns, &nd, &nk));
                    Count* buf = static_cast<Count*>(pin.need((size_t)std::max<uint64_t>(ns, 1) * sizeof(Count)));
                    uint64_t got = 0;
                    // the device's key width follows k (8-byte keys for k <= 31, 16-byte keys above: like the reference's run-time Integer
                    // dispatch); a span instantiated with a smaller k (the reference's unit tests do it), or the spans 96 / 128, get
                    // their records widened here
                    const bool dev16 = _config._kmerSize <= 31;
                    const bool same = dev16 ? sizeof(Count) == 16 : (sizeof(Count) == 32 && sizeof(Type) == 16);
                    if (same) check(gkc_partition_counts(_ctx, pass, p, buf, ns, &got));
                    else if (dev16) {
                        typedef typename Kmer<32>::Count DevCount;
                        DevCount* dev = static_cast<DevCount*>(pin_narrow.need((size_t)std::max<uint64_t>(ns, 1) * sizeof(DevCount)));
                        check(gkc_partition_counts(_ctx, pass, p, dev, ns, &got));
                        for (uint64_t i = 0; i < got; i++) { new (&buf[i]) Count(); buf[i].value.setVal(dev[i].value.getVal()); buf[i].abundance = dev[i].abundance; }
                    } else {
                        typedef typename Kmer<64>::Count DevCount;
                        DevCount* dev = static_cast<DevCount*>(pin_narrow.need((size_t)std::max<uint64_t>(ns, 1) * sizeof(DevCount)));
                        check(gkc_partition_counts(_ctx, pass, p, dev, ns, &got));
                        for (uint64_t i = 0; i < got; i++) { new (&buf[i]) Count(); buf[i].value.set128(dev[i].value.value); buf[i].abundance = dev[i].abundance; }
                    }
                    CountProcessor* clone = proc->clone(); clone->use(); clones.push_back(clone);
                    clone->beginPart(pass, p, 4096, "vector");
                    clone->processBulk(p, buf, (size_t)got);                                          // ascending k-mer order
                    clone->endPart(pass, p);
                }
                proc->finishClones(clones);
                for (auto* c : clones) c->forget();
                proc->endPass(pass);
            }
            if (_releasePasses) check(gkc_release_pass(_ctx, pass));                                 // every processor has seen the pass: its results leave the HBM
        }
        if (_deviceChain) {
            check(gkc_set_host_sink(_ctx, nullptr, 0));
            std::vector<uint64_t> h((size_t)_config._histo_max + 1, 0);
            check(gkc_histogram(_ctx, h.data(), (uint32_t)h.size()));
            devHisto->addCounts(h.data(), h.size());
            gkc_stats st0; check(gkc_get_stats(_ctx, &st0));
            devSol->addTotals(st0.kmers_nb_distinct, st0.kmers_nb_solid);
        }
        for (auto* p : _processors) p->end();
        gkc_stats st; check(gkc_get_stats(_ctx, &st));
        _info.add("kmers_nb_valid", "%llu", (unsigned long long)st.kmers_nb_valid);
        _info.add("kmers_nb_invalid", "%llu", (unsigned long long)st.kmers_nb_invalid);
        _info.add("nb_partitions", "%u", P); _info.add("nb_passes", "%u", _config._nb_passes);
        _info.add("nb_superkmers", "%llu", (unsigned long long)st.nb_superkmers);
        _info.add("seq_number", "%llu", (unsigned long long)st.nb_sequences);
        {   // the time keys consumers read (SortingCountAlgorithm.cpp:777-780: getTimeInfo() "fill_partitions" / "fill_solid_kmers", _fillTimeInfo
            // "1.read" / "2.sort" / "3.dump"), in seconds, from the device's own event timers (gkc_get_timing)
            auto ms = [&](const char* n) { double v = 0; uint64_t l = 0; (void)gkc_get_timing(_ctx, n, &v, &l); return v; };
            _info.add("fill_partitions", "%.3f", ms("total_stage_a") / 1e3);
            _info.add("fill_solid_kmers", "%.3f", ms("total_stage_b") / 1e3);
            _info.add("1.read", "%.3f", (ms("expand_count") + ms("expand_scatter")) / 1e3);
            _info.add("2.sort", "%.3f", (ms("bucket_sort") + ms("bucket_sort_big") + ms("bucket_sort_wg") + ms("bucket_sort_deep") + ms("split_levels")) / 1e3);
            _info.add("3.dump", "%.3f", ms("compact") / 1e3);
        }
        // the solidity processor sees every distinct k-mer (the device returns all of them; filtering is the chain's job)
        for (auto* p : _processors) if (auto* ch = dynamic_cast<CountProcessorChain<span>*>(p)) if (auto* s = ch->template get<CountProcessorSoliditySum<span>>()) {
            _info.add("kmers_nb_distinct", "%llu", (unsigned long long)s->totals().total); _info.add("kmers_nb_solid", "%llu", (unsigned long long)s->totals().ok); }
        for (auto* p : _processors) if (auto* px = dynamic_cast<CountProcessorCutoffProxy<span>*>(p)) _info.add("cutoffs_auto.values", "%s", px->getProperties().getStr("values").c_str());
        if (!_info.has("kmers_nb_distinct")) { _info.add("kmers_nb_distinct", "%llu", (unsigned long long)st.kmers_nb_distinct); _info.add("kmers_nb_solid", "%llu", (unsigned long long)st.kmers_nb_solid); }
    }

    /** default chain histogram -> solidity -> dump (getDefaultProcessorVector, SortingCountAlgorithm.cpp:376-400) */
    static CountProcessor* getDefaultProcessor(const Configuration& c) {
        std::vector<CountProcessor*> items;
        items.push_back(new CountProcessorHistogram<span>(c._histo_max));
        items.push_back(new CountProcessorSoliditySum<span>(c._abundance_min, c._abundance_max));
        items.push_back(new CountProcessorDump<span>(c._kmerSize));
        return new CountProcessorChain<span>(items);
    }

    /** getDefaultProcessorVector (SortingCountAlgorithm.cpp:456-512): with "-abundance-min auto" (abundance min -1) a cutoff processor runs over all
     *  partitions first and hands its threshold to the dsk chain, which then runs over them */
    static std::vector<CountProcessor*> getDefaultProcessorVector(const Configuration& c) {
        std::vector<CountProcessor*> v;
        CountProcessor* dsk = getDefaultProcessor(c);
        if (c._abundance_min == -1) v.push_back(new CountProcessorCutoffProxy<span>(new CountProcessorCutoff<span>(c._histo_max), dsk));
        v.push_back(dsk);
        return v;
    }

private:
    void check(int rc) { if (rc != GKC_OK) throw system::Exception("gkc error %d: %s", rc, gkc_last_error(_ctx)); }

    /** configure(), SortingCountAlgorithm.cpp:525-625 + ConfigurationAlgorithm.cpp:245-467 (GPU-aware partition count) */
    void configure() {
        if (_ctx) { gkc_destroy(_ctx); _ctx = nullptr; }                         // execute() called again: the previous run's context goes (objects built from it keep it alive)
        int rc = gkc_create((int)(_params.has(STR_GPU_DEVICE) ? _params.getInt(STR_GPU_DEVICE) : 0), &_ctx);
        if (rc != GKC_OK) throw system::Exception("gkc_create failed (%d): %s", rc, gkc_last_error(nullptr));
        if (!_config._isComputed) {
            _config._kmerSize = (size_t)_params.getInt(STR_KMER_SIZE);
            if (_config._kmerSize <= 2) throw system::Exception("Error: kmer size should be > 2");          // .cpp:662-666 (exit(1) there)
            if (_config._kmerSize >= span) throw system::Exception("Type '%s' has too low precision (%d bits) for the required %d kmer size",
                                                                   Type::getName(), (int)Type::getSize(), (int)_config._kmerSize);       // Model.hpp:398-404
            size_t m = (size_t)_params.getInt(STR_MINIMIZER_SIZE);
            if (m == 0) m = 8;                                                                              // ConfigurationAlgorithm.cpp:249-251
            _config._minim_size = std::min(_config._kmerSize - 1, m);
            _config._minimizerType = (size_t)_params.getInt(STR_MINIMIZER_TYPE); _config._repartitionType = (size_t)_params.getInt(STR_REPARTITION_TYPE);
            _config._abundance_min = _params.getStr(STR_KMER_ABUNDANCE_MIN) == "auto" ? -1 : (CountNumber)_params.getInt(STR_KMER_ABUNDANCE_MIN);   // ConfigurationAlgorithm.cpp: "auto" -> -1
            _config._abundance_max = (CountNumber)_params.getInt(STR_KMER_ABUNDANCE_MAX);
            _config._histo_max = (uint32_t)_params.getInt(STR_HISTOGRAM_MAX); _config._max_memory = (uint64_t)_params.getInt(STR_MAX_MEMORY);
            _config._nb_banks = _bank->getNbBanks();
            _bank->estimate(_config._estimateSeqNb, _config._estimateSeqTotalSize, _config._estimateSeqMaxSize);
            const uint64_t total = _config._estimateSeqTotalSize, nseq = _config._estimateSeqNb, k = _config._kmerSize;
            _config._kmersNb = total > nseq * (k - 1) ? total - nseq * (k - 1) : 0;                       // ConfigurationAlgorithm.cpp:308-319
            // Passes (ConfigurationAlgorithm.cpp:398-425 derives them from -max-memory / -max-disk): here from the HBM. During a pass the device holds the
            // pass's super-k-mer records (~1.5 B per k-mer), its results (one Count per solid k-mer; the ratio is not known yet: 0.6 assumed) and the
            // Stage-B working set; a pass is released once the processors have drained it (gkc_release_pass). 0 / absent = derive.
            const int64_t forcedPasses = _params.has(STR_NB_PASSES) ? _params.getInt(STR_NB_PASSES) : 0;
            if (forcedPasses > 0) _config._nb_passes = (uint32_t)forcedPasses;
            else {
                uint64_t usable = 0, totalMem = 0;
                check(gkc_device_memory(_ctx, &usable, &totalMem));
                const double perKmer = 1.5 + 0.6 * (double)sizeof(Count);
                const double cap = 0.75 * (double)usable / perKmer;
                _config._nb_passes = (uint32_t)std::max<double>(1.0, std::ceil((double)_config._kmersNb / std::max(cap, 1.0)));
            }
            uint32_t forced = _params.has(STR_NB_PARTITIONS) ? (uint32_t)_params.getInt(STR_NB_PARTITIONS) : 0;
            // GPU-aware sizing: a partition should hold ~3M k-mers (fits one Stage-B workgroup's sub-bucket fan-out); the scan keeps
            // its partition cursors in LDS up to 8192 partitions
            uint64_t want = forced ? forced : std::max<uint64_t>(4, (_config._kmersNb / _config._nb_passes + 2999999) / 3000000);
            if (!forced) { uint64_t p2 = 4; while (p2 < want) p2 <<= 1; want = std::min<uint64_t>(p2, 8192); }
            _config._nb_partitions = (uint32_t)std::min<uint64_t>(want, 65535);
            _config._isComputed = true;
        }
        _releasePasses = _config._nb_passes > 1 && !(_params.has("-keep-passes") && _params.getInt("-keep-passes"));   // (-keep-passes 1: results of all passes stay on the device)
        const size_t m = _config._minim_size; const uint32_t P = _config._nb_partitions;
        if (!_repartitor) { _repartitor = buildRepartitor(m, P); _repartitor->use(); }
        if (_processors.empty()) for (CountProcessor* p : getDefaultProcessorVector(_config)) { p->use(); _processors.push_back(p); }
        // default chain, fixed abundance window, Count records as wide as the device's: the window and the histogram are applied on the device and only
        // solid records travel (execute()); otherwise the device returns every distinct k-mer and the processors filter (drop-in behaviour for any chain)
        _deviceChain = false;
        {   const bool dev16 = _config._kmerSize <= 31;
            const bool same = dev16 ? sizeof(Count) == 16 : (sizeof(Count) == 32 && sizeof(Type) == 16);
            if (same && _processors.size() == 1 && _config._abundance_min >= 1 && !(_params.has("-host-chain") && _params.getInt("-host-chain")))
                if (auto* ch = dynamic_cast<CountProcessorChain<span>*>(_processors[0])) {
                    const auto& it = ch->items();
                    if (it.size() == 3 && dynamic_cast<CountProcessorHistogram<span>*>(it[0]) && dynamic_cast<CountProcessorSoliditySum<span>*>(it[1]) && dynamic_cast<CountProcessorDump<span>*>(it[2]))
                        _deviceChain = true;
                }
        }
        if (_deviceChain) {
            auto* sol = static_cast<CountProcessorChain<span>*>(_processors[0])->template get<CountProcessorSoliditySum<span>>();
            check(gkc_set_solidity(_ctx, sol->getAbundanceMin(), sol->getAbundanceMax(), _config._histo_max));
        } else check(gkc_set_solidity(_ctx, 1, 2147483647, _config._histo_max));
        check(gkc_configure(_ctx, (uint32_t)_config._kmerSize, (uint32_t)m, P, _config._nb_passes, _repartitor->getMinimizerFrequencies() ? GKC_MINIMIZER_FREQ : GKC_MINIMIZER_LEXI,
                            _repartitor->getTable().data(), _repartitor->getMinimizerFrequencies()));
    }

    /** RepartitorAlgorithm::execute (RepartitionAlgorithm.cpp:287-492): sample statistics on the device, tables on the host */
    Repartitor* buildRepartitor(size_t m, uint32_t P) {
        const auto& bases = _bank->bases(); const auto& offs = _bank->offsets();
        const uint64_t nseq = offs.size() - 1, nm = 1ULL << (2 * m);
        Repartitor* rep = new Repartitor((int)P, (int)m, (int)_config._nb_passes);
        std::vector<uint16_t> dummy(nm, 0);
        std::vector<std::pair<uint32_t, uint32_t>> counts;
        std::vector<uint32_t> freq;
        if (_config._minimizerType == 1) {                                                                  // computeFrequencies (:311-384)
            uint64_t ns = std::min<uint64_t>((uint64_t)(nseq * 0.05), 50000000ULL); if (ns == 0) ns = 1; ns = std::min<uint64_t>(ns + 1, nseq);
            std::vector<uint32_t> mc(nm, 0);
            check(gkc_configure(_ctx, (uint32_t)_config._kmerSize, (uint32_t)m, 1, 1, GKC_MINIMIZER_LEXI, dummy.data(), nullptr));
            check(gkc_count_mmers(_ctx, (uint32_t)m, bases.data(), offs.data(), ns, mc.data()));
            for (uint64_t i = 0; i < nm; i++) if (mc[i] > 0) counts.push_back({ mc[i], (uint32_t)i });
            std::sort(counts.begin(), counts.end());
            freq.assign(nm, (uint32_t)nm);
            for (size_t i = 0; i < counts.size(); i++) freq[counts[i].second] = (uint32_t)i;
            freq[nm - 1] = (uint32_t)(nm - 1);
            rep->setMinimizerFrequencies(freq);
        }
        // computeRepartition (:395-475): SampleRepart over the first reads of the bank, stopped like the reference: with the read in which the running
        // number of super-k-mers (of pass 0) first exceeds max(5 % of the sequences, 10^6) (:451, :205-212). gkc_sample_exact walks the reads one by one on
        // the device and also counts the kx-mers computeDistrib balances on.
        const uint64_t nbSeqSample = std::max<uint64_t>((uint64_t)(nseq * 0.05), 1000000ULL);
        std::vector<uint64_t> nsk(nm, 0), nk(nm, 0), nkx(nm, 0); uint64_t used = 0;
        // (ONE pass here whatever the run's number: SampleRepart is a Sequence2SuperKmer of 1 pass, :225 — every super-k-mer counts; the passes enter in computeDistrib)
        check(gkc_configure(_ctx, (uint32_t)_config._kmerSize, (uint32_t)m, 1, 1, freq.empty() ? GKC_MINIMIZER_LEXI : GKC_MINIMIZER_FREQ, dummy.data(), freq.empty() ? nullptr : freq.data()));
        check(gkc_sample_exact(_ctx, bases.data(), offs.data(), nseq, nbSeqSample, nsk.data(), nk.data(), nkx.data(), &used));
        if (_config._minimizerType == 1) rep->justGroup(nk, counts);
        else { rep->computeDistrib(nkx); if (_config._repartitionType == 1) rep->justGroupLexi(nk); }
        return rep;
    }

    bank::IBank* _bank; tools::misc::Properties _params; Configuration _config; Repartitor* _repartitor;
    std::vector<CountProcessor*> _processors; tools::misc::Properties _info; gkc_ctx* _ctx; bool _textRefused; bool _releasePasses = false; bool _deviceChain = false;
};

}}  // namespace kmer::impl

// ------------------------------------------------------------------------------------------------ Bloom
namespace tools { namespace collections { namespace impl {
enum BloomKind { BLOOM_NONE, BLOOM_BASIC, BLOOM_CACHE, BLOOM_NEIGHBOR, BLOOM_DEFAULT };
/** IBloom<Item> (tools/collections/impl/Bloom.hpp:113-168) backed by the device filter. insert()/contains() take one item to keep
 *  the interface; the bulk overloads are what a GPU wants and what BloomBuilder::build (BloomBuilder.hpp:102-128) maps to. */
template <typename Item> class IBloom : public system::SmartPointer {
public:
    virtual ~IBloom() {}
    virtual void insert(const Item& item) = 0;
    virtual bool contains(const Item& item) = 0;
    virtual uint8_t contains8(const Item& item) = 0;          // std::bitset<8> in the reference: bit j = neighbour j
    virtual uint8_t contains4(const Item& item, bool right) = 0;
    virtual std::vector<uint8_t> getArray() = 0;
    virtual uint64_t getSize() = 0;
    virtual uint64_t getBitSize() = 0;
    virtual size_t getNbHash() const = 0;
    virtual std::string getName() const = 0;
};
template <typename Item> class BloomDevice : public IBloom<Item> {
public:
    BloomDevice(gkc_ctx* ctx, BloomKind kind, uint64_t tai_bloom, size_t nbHash, size_t kmerSize) : _ctx(ctx), _kind(kind), _nbHash(nbHash), _b(nullptr) {
        const int kd = kind == BLOOM_BASIC ? 0 : (kind == BLOOM_CACHE || kind == BLOOM_DEFAULT) ? 1 : 2;
        if (kind == BLOOM_NONE) throw system::Exception("unknown Bloom kind");
        if (gkc_bloom_create(ctx, kd, tai_bloom, (uint32_t)nbHash, (uint32_t)kmerSize, &_b) != GKC_OK) throw system::Exception("%s", gkc_last_error(ctx));
    }
    ~BloomDevice() { gkc_bloom_destroy(_b); }
    void insert(const Item& item) { chk(gkc_bloom_insert(_b, &item, 1, sizeof(Item))); }
    void insert(const Item* items, size_t n) { chk(gkc_bloom_insert(_b, items, n, sizeof(Item))); }
    /** keys at the start of larger records (e.g. Count[]: value first) */
    void insertStrided(const void* first, size_t n, size_t strideBytes) { if (n) chk(gkc_bloom_insert(_b, first, n, (uint32_t)strideBytes)); }
    /** every solid k-mer of a counted context (BloomAlgorithm::execute, BloomAlgorithm.cpp:155-199) */
    void insertSolid(gkc_ctx* counted) { chk(gkc_bloom_insert_solid(_b, counted)); }
    bool contains(const Item& item) { uint8_t r = 0; chk(gkc_bloom_contains(_b, &item, 1, sizeof(Item), &r)); return r != 0; }
    void contains(const Item* items, size_t n, uint8_t* out) { chk(gkc_bloom_contains(_b, items, n, sizeof(Item), out)); }
    uint8_t contains8(const Item& item) { uint8_t r = 0; chk(gkc_bloom_contains8(_b, &item, 1, sizeof(Item), &r)); return r; }
    void contains8(const Item* items, size_t n, uint8_t* out) { chk(gkc_bloom_contains8(_b, items, n, sizeof(Item), out)); }
    uint8_t contains4(const Item& item, bool right) { uint8_t r = contains8(item); return right ? (r & 15) : (r >> 4); }
    std::vector<uint8_t> getArray() { std::vector<uint8_t> a(getSize()); chk(gkc_bloom_get_array(_b, a.data(), a.size())); return a; }
    uint64_t getSize() { return gkc_bloom_nbytes(_b); }
    uint64_t getBitSize() { return gkc_bloom_bitsize(_b); }
    size_t getNbHash() const { return _nbHash; }
    std::string getName() const { return _kind == BLOOM_BASIC ? "basic" : (_kind == BLOOM_NEIGHBOR ? "neighbor" : "cache"); }
private:
    void chk(int rc) { if (rc != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx)); }
    gkc_ctx* _ctx; BloomKind _kind; size_t _nbHash; gkc_bloom* _b;
};
/** BloomFactory::createBloom (Bloom.hpp:1254-1266) */
struct BloomFactory {
    template <typename Item> static IBloom<Item>* createBloom(gkc_ctx* ctx, BloomKind kind, uint64_t tai_bloom, size_t nbHash, size_t kmerSize) {
        return new BloomDevice<Item>(ctx, kind, tai_bloom, nbHash, kmerSize);
    }
};
}}}  // namespace tools::collections::impl

// ------------------------------------------------------------------------------------------------ BloomAlgorithm (C5)
namespace kmer { namespace impl {
enum DebloomKind { DEBLOOM_NONE, DEBLOOM_ORIGINAL, DEBLOOM_CASCADING, DEBLOOM_DEFAULT };
static const double gkc_debloom_rvalues[129][2] = {
#include "../../include/gkc_debloom_rvalues.inc"
};
/** DebloomAlgorithm<span>::getNbBitsPerKmer (kmer/impl/DebloomAlgorithm.cpp:628-660) */
inline float getNbBitsPerKmer(size_t kmerSize, DebloomKind debloomKind) {
    static double lg2 = log(2);
    float nbitsPerKmer = 0;
    if (kmerSize > 128 && debloomKind == DEBLOOM_CASCADING) throw system::Exception("kmer size %d too big for cascading bloom filters", (int)kmerSize);
    switch (debloomKind) {
        case DEBLOOM_CASCADING: nbitsPerKmer = (float)gkc_debloom_rvalues[kmerSize][1]; break;
        default: nbitsPerKmer = (float)(log(16 * kmerSize * (lg2 * lg2)) / (lg2 * lg2)); break;
    }
    if (nbitsPerKmer == 0) nbitsPerKmer = 1;
    return nbitsPerKmer;
}
/** BloomAlgorithm<span>::execute (kmer/impl/BloomAlgorithm.cpp:155-199): Bloom filter of the solid k-mers of a counted context.
 *  size = (u64)(nbSolid * nbitsPerKmer) in float arithmetic, nbHash = floor(0.7 * nbits), 1000 bits if empty; every solid
 *  k-mer is inserted ON THE DEVICE straight from the result buffers (BloomBuilder::build, BloomBuilder.hpp:102-128). */
template <size_t span = KMER_DEFAULT_SPAN>
class BloomAlgorithm {
public:
    typedef typename Kmer<span>::Type Type;
    typedef std::vector<std::vector<typename Kmer<span>::Count>> Store;
    /** solidCounts = the dump processor's datasets (getSolidCounts()): the SOLID k-mers, as the reference's BloomAlgorithm takes them (its `solidIterable`).
     *  Without it the k-mers come straight from the counting context's result buffers — every k-mer the device kept, i.e. the solid ones only when
     *  the context itself filtered with the wanted abundance window (gkc_set_solidity); SortingCountAlgorithm leaves the filtering to its chain. */
    BloomAlgorithm(gkc_ctx* countedCtx, size_t kmerSize, float nbitsPerKmer, tools::collections::impl::BloomKind kind = tools::collections::impl::BLOOM_DEFAULT,
                   const Store* solidCounts = nullptr)
        : _ctx(countedCtx), _kmerSize(kmerSize), _nbitsPerKmer(nbitsPerKmer), _kind(kind), _store(solidCounts), _bloom(nullptr) {}
    ~BloomAlgorithm() { if (_bloom) _bloom->forget(); }
    void execute() {
        gkc_stats st; if (gkc_get_stats(_ctx, &st) != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx));
        uint64_t solidKmersNb = st.kmers_nb_solid;
        if (_store) { solidKmersNb = 0; for (auto& d : *_store) solidKmersNb += d.size(); }
        const float NBITS_PER_KMER = _nbitsPerKmer;
        uint64_t estimatedBloomSize = (uint64_t)(solidKmersNb * NBITS_PER_KMER);
        const size_t nbHash = (size_t)(int)floorf(0.7 * NBITS_PER_KMER);
        if (estimatedBloomSize == 0) estimatedBloomSize = 1000;
        auto* b = new tools::collections::impl::BloomDevice<Type>(_ctx, _kind, estimatedBloomSize, nbHash, _kmerSize);
        b->use(); _bloom = b;
        if (_store) { for (auto& d : *_store) b->insertStrided(d.data(), d.size(), sizeof(typename Kmer<span>::Count)); }
        else b->insertSolid(_ctx);
        _info.add("kind", "%s", b->getName().c_str()); _info.add("bitsize", "%llu", (unsigned long long)b->getBitSize());
        _info.add("nb_hash", "%d", (int)nbHash); _info.add("nbits_per_kmer", "%f", _nbitsPerKmer);
    }
    tools::collections::impl::IBloom<Type>* getBloom() { return _bloom; }
    const tools::misc::Properties* getInfo() const { return &_info; }
private:
    gkc_ctx* _ctx; size_t _kmerSize; float _nbitsPerKmer; tools::collections::impl::BloomKind _kind; const Store* _store;
    tools::collections::impl::IBloom<Type>* _bloom; tools::misc::Properties _info;
};

/** MPHFAlgorithm (kmer/impl/MPHFAlgorithm.hpp:79-170, .cpp:150-275): BooPHF of the solid k-mers + abundance map, both built on the
 *  device from the counting context's result buffers. AbundanceMap = MapMPHF<Type, uint8_t>: size(), at(kmer), getCode(kmer)
 *  (tools/collections/impl/MapMPHF.hpp:60-260); the saved hash is the byte stream MPHF<>::save writes into "dsk/mphf". */
template <size_t span = KMER_DEFAULT_SPAN>
class MPHFAlgorithm {
public:
    typedef typename Kmer<span>::Type Type;
    class AbundanceMap {
    public:
        AbundanceMap(gkc_mphf* h, size_t kmerBytes) : _h(h), _kb(kmerBytes), _data(gkc_mphf_size(h), 0) {}
        size_t size() const { return _data.size(); }
        uint64_t getCode(const Type& kmer) const {
            uint64_t code = 0, raw[2] = { 0, 0 };
            memcpy(raw, &kmer, _kb);                                   // LargeInt<1|2> is its little-endian value (8 / 16 bytes)
            if (gkc_mphf_lookup(_h, raw, 1, (uint32_t)_kb, &code) != GKC_OK) throw system::Exception("MPHF lookup failed");
            return code;
        }
        uint8_t& at(const Type& kmer) { const uint64_t c = getCode(kmer); if (c >= _data.size()) throw system::Exception("MPHF check: value out of bounds"); return _data[c]; }
        uint8_t& at(uint64_t code) { return _data[code]; }
        std::vector<uint8_t>& data() { return _data; }
    private:
        gkc_mphf* _h; size_t _kb; std::vector<uint8_t> _data;
    };
    typedef std::vector<std::vector<typename Kmer<span>::Count>> Store;
    /** solidCounts = the dump processor's datasets (getSolidCounts()): keys and abundances come from there, as in the reference.
     *  Without it the device result buffers are used directly (valid when the device applied the solidity window itself). */
    explicit MPHFAlgorithm(gkc_ctx* countedCtx, size_t kmerSize, const Store* solidCounts = nullptr)
        : _ctx(countedCtx), _k(kmerSize), _store(solidCounts), _h(nullptr), _map(nullptr), _dataSize(0) {}
    ~MPHFAlgorithm() { delete _map; if (_h) gkc_mphf_destroy(_h); }
    /** index of an abundance in MapMPHF's discretization table (MapMPHF.hpp:96-145, MPHFAlgorithm.cpp:253-266) */
    static int abundanceIndex(int abundance) {
        static std::vector<int> disc;
        if (disc.empty()) { disc.resize(257); int total = 0, idx = 1; disc[0] = 0;
            auto run = [&](int cnt, int step) { for (int i = 1; i <= cnt; i++, idx++) { total += step; disc[idx] = total; } };
            run(70, 1); run(15, 2); run(40, 10); run(25, 20); run(40, 100); run(25, 200); run(40, 1000); disc[256] = total; }
        if (abundance >= disc[disc.size() - 2]) return (int)disc.size() - 2;
        return (int)(std::upper_bound(disc.begin(), disc.end(), abundance) - disc.begin()) - 1;
    }
    void execute() {
        const size_t kb = _k <= 31 ? 8 : 16;
        uint64_t above = 0;
        if (_store) {
            typedef typename Kmer<span>::Count Count;
            std::vector<Count> all; for (auto& d : *_store) all.insert(all.end(), d.begin(), d.end());
            if (gkc_mphf_build(_ctx, all.data(), all.size(), (uint32_t)sizeof(Count), (uint32_t)_k, &_h) != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx));   // build
            _map = new AbundanceMap(_h, kb);
            std::vector<uint64_t> codes(all.size());
            if (gkc_mphf_lookup(_h, all.data(), all.size(), (uint32_t)sizeof(Count), codes.data()) != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx));
            for (size_t i = 0; i < all.size(); i++) {                                                                  // populate
                if (codes[i] >= all.size()) throw system::Exception("MPHF check: value out of bounds");
                const int idx = abundanceIndex((int)all[i].abundance); if (idx == 255) above++;
                _map->at(codes[i]) = (uint8_t)idx;
            }
        } else {
            if (gkc_mphf_build_solid(_ctx, &_h) != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx));
            _map = new AbundanceMap(_h, kb);
            if (gkc_mphf_abundance_map(_h, _ctx, _map->data().data(), _map->data().size(), &above) != GKC_OK) throw system::Exception("%s", gkc_last_error(_ctx));
        }
        _dataSize = gkc_mphf_save_size(_h);                                                                           // save
        _info.add("nb_keys", "%llu", (unsigned long long)_map->size()); _info.add("data_size", "%llu", (unsigned long long)_dataSize);
        _info.add("bits_per_key", "%.3f", (float)(_dataSize * 8) / (float)_map->size()); _info.add("nb_abund_above_prec", "%llu", (unsigned long long)above);
    }
    AbundanceMap* getAbundanceMap() { return _map; }
    /** the stream MPHF<>::save puts into the storage group (tools/collections/impl/BooPHF.hpp:313-322) */
    std::vector<uint8_t> savedHash() const { std::vector<uint8_t> v(_dataSize); if (gkc_mphf_save(_h, v.data(), v.size()) != GKC_OK) throw system::Exception("MPHF save failed"); return v; }
    const tools::misc::Properties* getInfo() const { return &_info; }
private:
    gkc_ctx* _ctx; size_t _k; const Store* _store; gkc_mphf* _h; AbundanceMap* _map; uint64_t _dataSize; tools::misc::Properties _info;
};
}}  // namespace kmer::impl

}}  // namespace gatb::core
