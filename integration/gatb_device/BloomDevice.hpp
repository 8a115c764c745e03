/* BloomDevice.hpp — IBloom<Item> over libgkc_hip.so: what BloomFactory::createBloom (tools/collections/impl/Bloom.hpp:1254-1266) returns in a build with
 * GATB_WITH_DEVICE_COUNTING for the k-mer item types the device knows (LargeInt<1> with k <= 31, LargeInt<2> with 32 <= k <= 63; anything else, or a
 * process without a usable MI355X, keeps the CPU classes). Included from Bloom.hpp itself, just before BloomFactory (the patch), so it sees IBloom and the
 * CPU classes and nothing of kmer/.
 *
 * A BloomDevice IS the CPU filter of its kind plus a device copy of the same bit array:
 *   * the "host twin" is the reference's own class for the kind (BloomSynchronized / BloomCacheCoherent / BloomNeighborCoherent): getArray(), getSize(),
 *     getBitSize(), getName() (the "type" attribute StorageTools::saveBloom writes and loadBloom parses, StorageTools.hpp:95-160), weight() and the
 *     SINGLE-ITEM virtuals contains / contains4 / contains8 are served by it — no kernel launch per item anywhere;
 *   * insert() only buffers (per-thread stripes: BloomBuilder inserts from a Dispatcher's threads, BloomBuilder.hpp:117) and whole blocks go to the device
 *     (gkc_bloom_insert); insertSolid() inserts the solid k-mers where the counting step left them (gkc_bloom_insert_solid, BloomAlgorithm.cpp:155-199);
 *   * the BATCHED queries containsBatch / contains8Batch run on the device (gkc_bloom_contains / gkc_bloom_contains8; the reference's call site is
 *     DebloomMinimizerAlgorithm.cpp:201, one contains8 per solid k-mer).
 * The two copies are brought together lazily: the array is fetched when the host side is first asked after an insert, and sent when the device is first
 * asked after getArray() handed the host array out (StorageTools::loadBloom reads a stored filter straight into it). Same bit layout as the CPU class for the
 * same inserted set: tests/golden/reference_run/*.npz hold /bloom/bloom of the unpatched reference, tests/test_gpu_dropin.py compares. */
#ifndef _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_
#define _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_

#include <gatb/tools/collections/impl/Bloom.hpp>
#include <gatb/tools/math/LargeInt.hpp>
#include <gatb/system/api/Exception.hpp>

#include <gatb_device/DeviceContext.hpp>

#include <vector>
#include <bitset>
#include <string>
#include <mutex>
#include <atomic>
#include <thread>
#include <functional>
#include <string.h>
#include <stdio.h>

namespace gatb { namespace core { namespace tools { namespace collections { namespace impl {

/** Which item types the device filter takes: the raw bytes of the item are the device's key (8 bytes little endian for k <= 31, 16 for k <= 63), and the hash
 *  family follows the TYPE in the reference (LargeInt<1>: hash1 + 3-term simplehash16, LargeInt1.pri:157-211; LargeInt<2>: XOR of the halves + 2-term
 *  variant, LargeInt2.pri:200-251) while the device selects it by k — so a type is only taken with the k range it is the natural type of. */
template <typename Item> struct BloomDeviceTraits              { static bool accepts (size_t)   { return false; } };
template <> struct BloomDeviceTraits<tools::math::LargeInt<1> > { static bool accepts (size_t k)  { return k >= 1  &&  k <= 31; } };
#if INT128_FOUND == 1
template <> struct BloomDeviceTraits<tools::math::LargeInt<2> > { static bool accepts (size_t k)  { return k >= 32  &&  k <= 63; } };
#endif

template <typename Item> class BloomDevice : public IBloom<Item>
{
public:
    /** True when createBloom should hand out a BloomDevice for these arguments. kmerSize 0 (callers that do not know k and ask for a kind that does not use it)
     *  stands for the type's natural range. */
    static bool usable (tools::misc::BloomKind kind, size_t kmerSize)
    {
        if (kind != tools::misc::BLOOM_BASIC  &&  kind != tools::misc::BLOOM_CACHE  &&  kind != tools::misc::BLOOM_NEIGHBOR  &&  kind != tools::misc::BLOOM_DEFAULT)  { return false; }
        if (getenv ("GATB_DEVICE_NO_BLOOM") != 0)  { return false; }
        if (kmerSize == 0  &&  kind != tools::misc::BLOOM_NEIGHBOR)  { kmerSize = sizeof(Item) == 8 ? 31 : 63; }
        if (!BloomDeviceTraits<Item>::accepts (kmerSize))  { return false; }
        return device::DeviceContext::singleton().available();
    }

    BloomDevice (tools::misc::BloomKind kind, u_int64_t tai_bloom, size_t nbHash, size_t kmerSize)
        : _ctx (device::DeviceContext::singleton().ctx()), _bloom(0), _twin(0), _hostStale(false), _deviceStale(false), _hostReady(true), _queries(0), _inserted(0)
    {
        if (_ctx == 0)  { throw system::Exception ("BloomDevice: %s", device::DeviceContext::singleton().error().c_str()); }
        if (kmerSize == 0)  { kmerSize = sizeof(Item) == 8 ? 31 : 63; }
        switch (kind)
        {
            case tools::misc::BLOOM_BASIC:     _twin = new BloomSynchronized<Item>     (tai_bloom, nbHash);            break;
            case tools::misc::BLOOM_NEIGHBOR:  _twin = new BloomNeighborCoherent<Item> (tai_bloom, kmerSize, nbHash);  break;
            default:                           _twin = new BloomCacheCoherent<Item>    (tai_bloom, nbHash);            break;      /* BLOOM_CACHE, BLOOM_DEFAULT */
        }
        _twin->use();
        const int k = kind == tools::misc::BLOOM_BASIC ? 0 : (kind == tools::misc::BLOOM_NEIGHBOR ? 2 : 1);
        if (gkc_bloom_create (_ctx, k, tai_bloom, (uint32_t) nbHash, (uint32_t) kmerSize, &_bloom) != GKC_OK)
        {
            const std::string msg = gkc_last_error (_ctx);  _twin->forget();
            throw system::Exception ("BloomDevice: %s", msg.c_str());
        }
        if (gkc_bloom_nbytes (_bloom) != _twin->getSize()  ||  gkc_bloom_bitsize (_bloom) != _twin->getBitSize())
        {
            gkc_bloom_destroy (_bloom);  _twin->forget();
            throw system::Exception ("BloomDevice: the device filter (%llu bytes) is not sized like the reference class (%llu bytes)",
                                     (unsigned long long) gkc_bloom_nbytes (_bloom), (unsigned long long) _twin->getSize());
        }
    }
    ~BloomDevice ()
    {
        if (getenv ("GATB_DEVICE_VERBOSE") != 0  &&  (_inserted != 0  ||  _queries != 0))
        {
            fprintf (stderr, "[device bloom] %s, %llu bits: %llu items inserted in blocks (gkc_bloom_insert), %llu items queried in batches (gkc_bloom_contains / contains8)\n",
                     getName().c_str(), (unsigned long long) getBitSize(), (unsigned long long) _inserted, (unsigned long long) _queries);
        }
        gkc_bloom_destroy (_bloom);  _twin->forget();
    }

    /** Bag: buffered per thread stripe; a full stripe goes to the device as one block */
    void insert (const Item& item)
    {
        Stripe& s = _stripes [stripeOfThread()];
        std::lock_guard<std::mutex> guard (s.lock);
        /* (read before written: a store per insert from the 256 threads of a Dispatcher keeps one cache line travelling between all of them — the cascading
         *  step of the debloom, DebloomAlgorithm.cpp:520-548, took 3.8 s instead of the CPU classes' 1.2 s at 10^7 reads) */
        if (_hostReady.load (std::memory_order_relaxed))  { _hostReady.store (false, std::memory_order_release); }
        s.items.push_back (item);
        if (s.items.size() >= (size_t) BLOCK_ITEMS)  { sendStripe (s); }
    }
    void flush ()
    {
        std::vector<Item> pending;       /* what the stripes still hold, as ONE block */
        for (size_t i = 0; i < (size_t) NB_STRIPES; i++)
        {
            std::lock_guard<std::mutex> guard (_stripes[i].lock);
            pending.insert (pending.end(), _stripes[i].items.begin(), _stripes[i].items.end());  _stripes[i].items.clear();
        }
        if (pending.empty())  { return; }
        std::lock_guard<std::mutex> guard (_sync);
        toDevice();
        check (gkc_bloom_insert (_bloom, pending.data(), pending.size(), (uint32_t) sizeof(Item)));
        _inserted += pending.size();  _hostStale = true;
    }
    /** a whole array of items (stride sizeof(Item)) at once */
    void insertBatch (const Item* items, size_t n)
    {
        if (n == 0)  { return; }
        std::lock_guard<std::mutex> guard (_sync);
        toDevice();
        _hostReady.store (false, std::memory_order_release);
        check (gkc_bloom_insert (_bloom, items, n, (uint32_t) sizeof(Item)));  _hostStale = true;  _inserted += n;
    }
    /** every solid k-mer of the counting step, where Stage B left it in HBM (BloomAlgorithm::execute, BloomAlgorithm.cpp:155-199) */
    void insertSolid ()
    {
        std::lock_guard<std::mutex> guard (_sync);
        toDevice();
        _hostReady.store (false, std::memory_order_release);
        check (gkc_bloom_insert_solid (_bloom, _ctx));  _hostStale = true;
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)  { fprintf (stderr, "[device bloom] %s, %llu bits: the solid k-mers inserted where Stage B left them (gkc_bloom_insert_solid)\n", getName().c_str(), (unsigned long long) getBitSize()); }
    }

    /** Container: single items are answered by the host twin (the reference's own code on the host copy of the array) */
    bool           contains  (const Item& item)              { return twin()->contains  (item); }
    std::bitset<4> contains4 (const Item& item, bool right)  { return twin()->contains4 (item, right); }
    std::bitset<8> contains8 (const Item& item)              { return twin()->contains8 (item); }

    /** whole arrays are answered by the device: out[i] = 0/1, resp. the 8 neighbour bits of items[i] (bits 0-3 successors, 4-7 predecessors, Bloom.hpp:801-811) */
    void containsBatch  (const Item* items, size_t n, u_int8_t* out)  { query (items, n, out, false); }
    void contains8Batch (const Item* items, size_t n, u_int8_t* out)  { query (items, n, out, true);  }
    u_int64_t nbDeviceQueries () const  { return _queries; }

    /** IBloom */
    u_int8_t*& getArray ()
    {
        IBloom<Item>* t = twin();
        { std::lock_guard<std::mutex> guard (_sync);  _deviceStale = true; }       /* the caller may write it (StorageTools::loadBloom does) */
        return t->getArray();
    }
    u_int64_t     getSize    ()        { return _twin->getSize(); }
    u_int64_t     getBitSize ()        { return _twin->getBitSize(); }
    size_t        getNbHash  () const  { return _twin->getNbHash(); }
    std::string   getName    () const  { return _twin->getName(); }                /* "basic" / "cache" / "neighbor": the stored "type" of the filter */
    unsigned long weight     ()        { return twin()->weight(); }

private:
    enum { NB_STRIPES = 512, BLOCK_ITEMS = 1 << 16 };
    struct Stripe  { std::mutex lock;  std::vector<Item> items;  char apart[64]; };      /* (no two stripes' lock and vector in one cache line) */
    /* threads take stripes in the order they first insert (a Dispatcher starts new threads for every iterate: Command.cpp:128-175), so the threads of one
     * iteration — up to -nb-cores of them — get stripes of their own */
    static size_t stripeOfThread ()
    {
        static std::atomic<size_t> next (0);
        thread_local size_t mine = next.fetch_add (1, std::memory_order_relaxed);
        return mine % NB_STRIPES;
    }

    void check (int rc)  { if (rc != GKC_OK) { throw system::Exception ("BloomDevice: error %d: %s", rc, gkc_last_error(_ctx)); } }

    /* caller holds the stripe's lock */
    void sendStripe (Stripe& s)
    {
        if (s.items.empty())  { return; }
        std::lock_guard<std::mutex> guard (_sync);
        toDevice();
        check (gkc_bloom_insert (_bloom, s.items.data(), s.items.size(), (uint32_t) sizeof(Item)));
        _inserted += s.items.size();  s.items.clear();  _hostStale = true;
    }
    /* caller holds _sync: the host array was handed out since the device last saw it */
    void toDevice ()
    {
        if (_deviceStale)  { check (gkc_bloom_set_array (_bloom, _twin->getArray(), _twin->getSize()));  _deviceStale = false; }
    }
    /* the host twin with every insert so far in its array (nothing pending: no lock is taken — the debloom threads ask one k-mer at a time) */
    IBloom<Item>* twin ()
    {
        if (_hostReady.load (std::memory_order_acquire))  { return _twin; }
        flush();
        std::lock_guard<std::mutex> guard (_sync);
        if (_hostStale)  { check (gkc_bloom_get_array (_bloom, _twin->getArray(), _twin->getSize()));  _hostStale = false; }
        _hostReady.store (true, std::memory_order_release);
        return _twin;
    }
    void query (const Item* items, size_t n, u_int8_t* out, bool eight)
    {
        if (n == 0)  { return; }
        flush();
        std::lock_guard<std::mutex> guard (_sync);
        toDevice();
        check (eight ? gkc_bloom_contains8 (_bloom, items, n, (uint32_t) sizeof(Item), out) : gkc_bloom_contains (_bloom, items, n, (uint32_t) sizeof(Item), out));
        _queries += n;
    }

    gkc_ctx*      _ctx;
    gkc_bloom*    _bloom;
    IBloom<Item>* _twin;
    std::mutex    _sync;            /**< one thread drives the device filter at a time */
    bool          _hostStale;       /**< the device array has inserts the host twin has not seen */
    bool          _deviceStale;     /**< the host array was handed out (getArray) and may have been written */
    std::atomic<bool> _hostReady;   /**< no insert is pending and the host twin's array is current */
    u_int64_t     _queries, _inserted;
    Stripe        _stripes [NB_STRIPES];
};

} } } } } /* end of namespaces. */

#endif /* _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_ */
