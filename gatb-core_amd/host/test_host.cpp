// test_host.cpp — the reference's own DSK unit tests (test/unit/src/kmer/TestDSK.cpp:117-305, TestDebloom.cpp:84-170,
// tools/collections/TestContainer.cpp:63-128) re-expressed against the GPU-backed classes of gatb_gkc.hpp.
// Input vectors come from tests/golden/reference_unit_vectors.json flattened to a text file by the pytest wrapper:
//   line := "check1 <k> <nks> <expected> <nseq> <seq>..." | "check2 <k> <checksum_hex> <nvalues> <hex>... <seq>"
#include "gatb_gkc.hpp"
#include <iostream>
#include <set>
#include <sstream>

using namespace gatb::core;
using namespace gatb::core::kmer::impl;
using namespace gatb::core::tools::misc;
using namespace gatb::core::tools::collections::impl;

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { failures++; std::cerr << "FAILED " << #cond << " at line " << __LINE__ << std::endl; } } while (0)

template <size_t span>
static void DSK_check1_aux(const std::vector<std::string>& seqs, size_t kmerSize, size_t nks, size_t checkNbSolids)
{
    IProperties* params = SortingCountAlgorithm<span>::getDefaultProperties();
    params->setInt(STR_KMER_SIZE, kmerSize);
    params->setInt(STR_MAX_MEMORY, 1000);
    params->setInt(STR_KMER_ABUNDANCE_MIN, nks);
    params->setStr(STR_URI_OUTPUT, "foo");
    SortingCountAlgorithm<span> dsk(new bank::BankStrings(seqs), params);
    dsk.execute();
    if ((int64_t)checkNbSolids != dsk.getInfo()->getInt("kmers_nb_solid"))
        std::cout << "problem with kmersize " << kmerSize << " nks " << nks << " expected " << checkNbSolids << " solids, had " << dsk.getInfo()->getInt("kmers_nb_solid") << std::endl;
    CHECK((int64_t)checkNbSolids == dsk.getInfo()->getInt("kmers_nb_solid"));
    delete params;
}

template <size_t span>
static void DSK_check2_aux(const std::string& s1, size_t kmerSize, const std::vector<uint64_t>& ok, uint64_t checksumExpected)
{
    typedef typename Kmer<span>::Type Type;
    IProperties* params = SortingCountAlgorithm<span>::getDefaultProperties();
    params->setInt(STR_KMER_SIZE, kmerSize);
    params->setInt(STR_KMER_ABUNDANCE_MIN, 1);
    SortingCountAlgorithm<span> sortingCount(new bank::BankStrings(s1.c_str(), NULL), params);
    sortingCount.execute();
    std::set<uint64_t> okValues(ok.begin(), ok.end()), checkValues;
    Type checksum; checksum.setVal(0);
    Type prev; bool first = true;
    for (auto& part : sortingCount.getSolidCounts()) {
        first = true;
        for (auto& item : part) {
            CHECK(okValues.count(item.value.getVal()) == 1);
            checkValues.insert(item.value.getVal());
            checksum += item.value;
            if (!first) CHECK(prev < item.value);                  // ascending inside a partition
            prev = item.value; first = false;
        }
    }
    CHECK(checksum.getVal() == checksumExpected);
    CHECK(checkValues.size() == okValues.size());
    delete params;
}

// custom processor: the plug-in protocol is honoured call by call (begin, beginPass, clone, beginPart, process ascending, endPart,
// finishClones, endPass, end) — examples/kmer/kmer12.cpp style
template <size_t span>
struct ProtocolProbe : public CountProcessorAbstract<span> {
    typedef typename Kmer<span>::Type Type;
    struct Log { int begin = 0, end = 0, beginPass = 0, endPass = 0, clones = 0, beginPart = 0, endPart = 0, finish = 0; uint64_t processed = 0; bool ordered = true; uint64_t sum = 0; };
    Log* log; Type last; bool has;
    explicit ProtocolProbe(Log* l) : log(l), has(false) {}
    void begin(const Configuration&) { log->begin++; }
    void end() { log->end++; }
    void beginPass(size_t) { log->beginPass++; }
    void endPass(size_t) { log->endPass++; }
    ICountProcessor<span>* clone() { log->clones++; return new ProtocolProbe(log); }
    void finishClones(std::vector<ICountProcessor<span>*>& c) { log->finish += (int)c.size(); }
    void beginPart(size_t, size_t, size_t, const char* name) { log->beginPart++; has = false; CHECK(std::string(name) == "vector"); }
    void endPart(size_t, size_t) { log->endPart++; }
    bool process(size_t, const Type& kmer, const CountVector& count, CountNumber sum) {
        log->processed++; log->sum += (uint64_t)sum; CHECK(count.size() == 1 && count[0] == sum);
        if (has && !(last < kmer)) log->ordered = false;
        last = kmer; has = true; return true;
    }
    std::string getName() const { return "probe"; }
};

static void protocol_test()
{
    std::vector<std::string> seqs;
    uint64_t x = 88172645463325252ULL;
    for (int r = 0; r < 400; r++) { std::string s; for (int i = 0; i < 120; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; s += "ACGT"[x & 3]; } seqs.push_back(s); }
    IProperties* params = SortingCountAlgorithm<32>::getDefaultProperties();
    params->setInt(STR_KMER_SIZE, 21); params->setInt(STR_NB_PARTITIONS, 6); params->setInt(STR_NB_PASSES, 2);
    ProtocolProbe<32>::Log log;
    SortingCountAlgorithm<32> dsk(new bank::BankStrings(seqs), params);
    dsk.addProcessor(new ProtocolProbe<32>(&log));
    dsk.execute();
    CHECK(log.begin == 1 && log.end == 1 && log.beginPass == 2 && log.endPass == 2);
    CHECK(log.clones == 12 && log.beginPart == 12 && log.endPart == 12 && log.finish == 12);
    CHECK(log.ordered);
    CHECK(log.sum == (uint64_t)dsk.getInfo()->getInt("kmers_nb_valid"));
    CHECK(log.processed == (uint64_t)dsk.getInfo()->getInt("kmers_nb_distinct"));
    CHECK(dsk.getInfo()->getInt("kmers_nb_valid") == 400 * 100);
    delete params;
}

// TestContainer.cpp:63-128 property for the three kinds + BloomFactory
template <size_t span> static void bloom_test(size_t k)
{
    typedef typename Kmer<span>::Type Type;
    gkc_ctx* ctx = nullptr; if (gkc_create(0, &ctx) != GKC_OK) { failures++; return; }
    std::vector<Type> in, out; uint64_t x = 1234567;
    for (int i = 0; i < 20000; i++) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; Type t; t.setVal((x >> 3) & ((k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1))); (i & 1 ? in : out).push_back(t); }
    for (BloomKind kind : { BLOOM_BASIC, BLOOM_CACHE, BLOOM_NEIGHBOR }) {
        IBloom<Type>* bloom = BloomFactory::createBloom<Type>(ctx, kind, in.size() * 12, 8, k);
        bloom->use();
        static_cast<BloomDevice<Type>*>(bloom)->insert(in.data(), in.size());
        std::vector<uint8_t> r(in.size()); static_cast<BloomDevice<Type>*>(bloom)->contains(in.data(), in.size(), r.data());
        size_t miss = 0; for (uint8_t v : r) miss += !v;
        CHECK(miss == 0);                                           // no false negative
        CHECK(bloom->contains(in[0]));
        CHECK(bloom->getSize() == 1 + (in.size() * 12 + (kind == BLOOM_BASIC ? 0 : 8192)) / 8);
        bloom->forget();
    }
    gkc_destroy(ctx);
}

// TestDebloom.cpp:84-137 sizing through BloomAlgorithm: k=11, DEBLOOM_ORIGINAL, BLOOM_BASIC -> 130 solid k-mers,
// size = (u64)(130 * nbits) = 1200 bits, 6 hash functions; plus the cascading table entry used by dbgh5's default.
static void bloom_algorithm_test()
{
    const char* seq = "CGCTACAGCAGCTAGTTCATCATTGTTTATCAATGATAAAATATAATAAGCTAAAAGGAAACTATAAATA"
                      "ACCATGTATAATTATAAGTAGGTACCTATTTTTTTATTTTAAACTGAAATTCAATATTATATAGGCAAAG";
    IProperties* params = SortingCountAlgorithm<32>::getDefaultProperties();
    params->setInt(STR_KMER_SIZE, 11); params->setInt(STR_MINIMIZER_SIZE, 8); params->setInt(STR_KMER_ABUNDANCE_MIN, 1);
    SortingCountAlgorithm<32> sortingCount(new bank::BankStrings(seq, NULL), params);
    sortingCount.execute();
    CHECK(sortingCount.getInfo()->getInt("kmers_nb_solid") == 130);
    float nbits = getNbBitsPerKmer(11, DEBLOOM_ORIGINAL);
    BloomAlgorithm<32> bloom(sortingCount.context(), 11, nbits, BLOOM_BASIC);
    bloom.execute();
    CHECK(bloom.getInfo()->getInt("nb_hash") == 6);
    CHECK(bloom.getBloom()->getBitSize() == 1200);
    CHECK(bloom.getBloom()->getSize() == 1 + 1200 / 8);
    size_t miss = 0;
    for (auto& part : sortingCount.getSolidCounts()) for (auto& c : part) miss += !bloom.getBloom()->contains(c.value);
    CHECK(miss == 0);
    CHECK(getNbBitsPerKmer(31, DEBLOOM_CASCADING) > 5 && getNbBitsPerKmer(31, DEBLOOM_CASCADING) < 20);
    delete params;
    // abundance min 2: the Bloom filter holds the SOLID k-mers (the chain filtered them; the device kept every distinct one)
    {
        IProperties* p2 = SortingCountAlgorithm<32>::getDefaultProperties();
        p2->setInt(STR_KMER_SIZE, 9); p2->setInt(STR_MINIMIZER_SIZE, 6); p2->setInt(STR_KMER_ABUNDANCE_MIN, 2);
        std::string twice = std::string(seq) + "ACGT" + std::string(seq).substr(20, 60);      // 52 k-mers occur twice
        SortingCountAlgorithm<32> sc(new bank::BankStrings(twice.c_str(), NULL), p2);
        sc.execute();
        size_t nsolid = 0; for (auto& part : sc.getSolidCounts()) nsolid += part.size();
        CHECK(nsolid > 0 && (int64_t)nsolid == sc.getInfo()->getInt("kmers_nb_solid") && sc.getInfo()->getInt("kmers_nb_distinct") > (int64_t)nsolid);
        BloomAlgorithm<32> bl(sc.context(), 9, 12.0f, BLOOM_CACHE, &sc.getSolidCounts());
        bl.execute();
        CHECK(bl.getBloom()->getBitSize() == (uint64_t)(nsolid * 12.0f));
        IBloom<Kmer<32>::Type>* ref = BloomFactory::createBloom<Kmer<32>::Type>(sc.context(), BLOOM_CACHE, (uint64_t)(nsolid * 12.0f), (size_t)(int)floorf(0.7f * 12.0f), 9);
        ref->use();
        for (auto& part : sc.getSolidCounts()) for (auto& c : part) ref->insert(c.value);
        CHECK(ref->getArray() == bl.getBloom()->getArray());                                      // exactly the solid set, nothing else
        ref->forget();
        delete p2;
    }
}

int main(int argc, char** argv)
{
    if (argc < 2) { std::cerr << "usage: test_host vectors.txt" << std::endl; return 2; }
    std::ifstream in(argv[1]); std::string line; int n1 = 0, n2 = 0;
    try {
        while (std::getline(in, line)) {
            std::istringstream ss(line); std::string tag; ss >> tag;
            if (tag == "check1") {
                size_t k, nks, exp, nseq; ss >> k >> nks >> exp >> nseq; std::vector<std::string> seqs(nseq); for (auto& s : seqs) ss >> s;
                DSK_check1_aux<32>(seqs, k, nks, exp); DSK_check1_aux<64>(seqs, k, nks, exp); DSK_check1_aux<96>(seqs, k, nks, exp); n1++;
            } else if (tag == "check2") {
                size_t k, nv; std::string cs; ss >> k >> cs >> nv; std::vector<uint64_t> vals(nv); for (auto& v : vals) { std::string h; ss >> h; v = strtoull(h.c_str(), 0, 16); }
                std::string seq; ss >> seq;
                DSK_check2_aux<32>(seq, k, vals, strtoull(cs.c_str(), 0, 16)); DSK_check2_aux<64>(seq, k, vals, strtoull(cs.c_str(), 0, 16));
                DSK_check2_aux<96>(seq, k, vals, strtoull(cs.c_str(), 0, 16)); DSK_check2_aux<128>(seq, k, vals, strtoull(cs.c_str(), 0, 16)); n2++;   // spans 32 / 64 / 96 as TestDSK.cpp:300-304 (+128)
            }
        }
        protocol_test();
        bloom_algorithm_test();
        bloom_test<32>(31); bloom_test<64>(47); bloom_test<96>(47); bloom_test<128>(21);
        // error behaviour: k too small is refused, k >= span is refused (Model.hpp:398-404)
        bool threw = false;
        try { IProperties* p = SortingCountAlgorithm<32>::getDefaultProperties(); p->setInt(STR_KMER_SIZE, 2); SortingCountAlgorithm<32> d(new bank::BankStrings("ACGTACGT", NULL), p); d.execute(); }
        catch (system::Exception& e) { threw = true; }
        CHECK(threw);
        threw = false;
        try { IProperties* p = SortingCountAlgorithm<32>::getDefaultProperties(); p->setInt(STR_KMER_SIZE, 33); SortingCountAlgorithm<32> d(new bank::BankStrings("ACGTACGT", NULL), p); d.execute(); }
        catch (system::Exception& e) { threw = true; }
        CHECK(threw);
        // API surface: options parser (defaults = getDefaultProperties for the shared names), (params)-only constructor, getSolidKmers
        {
            tools::misc::IOptionsParser* parser = SortingCountAlgorithm<32>::getOptionsParser(true);
            const char* av[] = { "prog", "-in", "x.fa", "-kmer-size", "21", "-abundance-min", "3" };
            IProperties* pp = parser->parse(7, (char**)av);
            IProperties* dd = SortingCountAlgorithm<32>::getDefaultProperties();
            CHECK(pp->getInt(STR_KMER_SIZE) == 21 && pp->getInt(STR_KMER_ABUNDANCE_MIN) == 3 && pp->getStr(STR_URI_INPUT) == "x.fa");
            for (const char* name : { STR_KMER_ABUNDANCE_MAX, STR_MINIMIZER_SIZE, STR_MINIMIZER_TYPE, STR_REPARTITION_TYPE, STR_MAX_MEMORY, STR_HISTOGRAM_MAX })
                CHECK(pp->getStr(name) == dd->getStr(name));
            bool t1 = false, t2 = false;
            const char* bad[] = { "prog", "-in", "x.fa", "-no-such-option", "1" };
            try { parser->parse(5, (char**)bad); } catch (system::Exception&) { t1 = true; }
            const char* miss[] = { "prog", "-kmer-size", "21" };
            try { parser->parse(3, (char**)miss); } catch (system::Exception&) { t2 = true; }
            CHECK(t1 && t2);
            CHECK(parser->help().find(STR_KMER_SIZE) != std::string::npos);
            delete pp; delete dd; delete parser;
            bool t3 = false;
            try { SortingCountAlgorithm<32> none; none.execute(); } catch (system::Exception&) { t3 = true; }
            CHECK(t3);
            IProperties* p = SortingCountAlgorithm<32>::getDefaultProperties(); p->setInt(STR_KMER_SIZE, 11); p->setInt(STR_KMER_ABUNDANCE_MIN, 1);
            SortingCountAlgorithm<32> d(new bank::BankStrings("ACGTACGTTTGACCAGTAGGCATTACG", NULL), p);
            d.execute();
            size_t nrec = 0; for (auto& part : d.getSolidCounts()) nrec += part.size();
            CHECK(d.getSolidKmers().size() == nrec && nrec == 17);
            delete p;
        }
    } catch (system::Exception& e) { std::cerr << "EXCEPTION: " << e.getMessage() << std::endl; return 3; }
    std::cout << "host tests: check1=" << n1 << " check2=" << n2 << " failures=" << failures << std::endl;
    return failures ? 1 : 0;
}
