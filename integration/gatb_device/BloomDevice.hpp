/* BloomDevice.hpp — IBloom<Item> over libgkc_hip.so (reference-side binding, compiled against the reference's headers by
 * integration/check_integration.sh). Slots into BloomFactory::createBloom (tools/collections/impl/Bloom.hpp:1254-1266) beside
 * BloomSynchronized / BloomCacheCoherent / BloomNeighborCoherent: same bit layout (the array is byte-identical to the CPU class's for the
 * same inserted set), so consumers that read getArray() to save / load the filter (BloomBuilder.hpp:134-148, StorageTools::saveBloom)
 * keep working. insert() batches items and flushes them to the device; contains / contains4 / contains8 query it. */
#ifndef _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_
#define _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_

#include <gatb/tools/collections/impl/Bloom.hpp>
#include <gatb/system/api/Exception.hpp>

#include <gkc.h>

#include <vector>
#include <bitset>
#include <string>

namespace gatb { namespace core { namespace tools { namespace collections { namespace impl {

/** Item: a k-mer integer type (LargeInt<1>, LargeInt<2>, ...) holding k <= 63 nucleotides. */
template <typename Item> class BloomDevice : public IBloom<Item>
{
public:
    /** \param[in] kind : tools::misc::BLOOM_BASIC / BLOOM_CACHE (and BLOOM_DEFAULT) / BLOOM_NEIGHBOR */
    BloomDevice (gkc_ctx* ctx, tools::misc::BloomKind kind, u_int64_t tai_bloom, size_t nbHash, size_t kmerSize)
        : _ctx(ctx), _bloom(0), _kind(kind), _nbHash(nbHash), _kmerSize(kmerSize), _host(0), _hostValid(false)
    {
        const int k = kind == tools::misc::BLOOM_BASIC ? 0 : (kind == tools::misc::BLOOM_NEIGHBOR ? 2 : 1);
        if (gkc_bloom_create (_ctx, k, tai_bloom, (uint32_t) nbHash, (uint32_t) kmerSize, &_bloom) != GKC_OK)
            throw system::Exception ("BloomDevice: %s", gkc_last_error(_ctx));
        _words = kmerSize <= 31 ? 1 : 2;
    }
    ~BloomDevice ()  { gkc_bloom_destroy (_bloom);  delete[] _host; }

    /** Bag */
    void insert (const Item& item)  { push (item);  if (_pending.size() >= _words * (size_t)(1 << 20))  { flush(); } }
    void flush ()
    {
        if (!_pending.empty())
        {
            check (gkc_bloom_insert (_bloom, _pending.data(), _pending.size() / _words, (uint32_t)(8 * _words)));
            _pending.clear();  _hostValid = false;
        }
    }

    /** Container */
    bool contains (const Item& item)
    {
        flush();
        u_int64_t key[2];  unsigned char out = 0;  store (item, key);
        check (gkc_bloom_contains (_bloom, key, 1, (uint32_t)(8 * _words), &out));
        return out != 0;
    }
    std::bitset<8> contains8 (const Item& item)
    {
        flush();
        u_int64_t key[2];  unsigned char out = 0;  store (item, key);
        check (gkc_bloom_contains8 (_bloom, key, 1, (uint32_t)(8 * _words), &out));
        return std::bitset<8> (out);
    }
    std::bitset<4> contains4 (const Item& item, bool right)
    {
        std::bitset<8> all = contains8 (item);  std::bitset<4> res;           // bits 0-3: successors, 4-7: predecessors (Bloom.hpp:725-828)
        for (int i = 0; i < 4; i++)  { res[i] = all[right ? i : 4 + i]; }
        return res;
    }

    /** IBloom */
    u_int8_t*& getArray ()
    {
        flush();
        if (!_hostValid)
        {
            if (_host == 0)  { _host = new u_int8_t [gkc_bloom_nbytes (_bloom)]; }
            check (gkc_bloom_get_array (_bloom, _host, gkc_bloom_nbytes (_bloom)));
            _hostValid = true;
        }
        return _host;
    }
    u_int64_t getSize    ()        { return gkc_bloom_nbytes  (_bloom); }
    u_int64_t getBitSize ()        { return gkc_bloom_bitsize (_bloom); }
    size_t    getNbHash  () const  { return _nbHash; }
    std::string getName  () const  { return "device"; }
    unsigned long weight ()
    {
        u_int8_t* a = getArray();  unsigned long w = 0;
        for (u_int64_t i = 0; i < getSize(); i++)  { w += __builtin_popcount (a[i]); }
        return w;
    }

    /** loading a saved filter (StorageTools::loadBloom fills getArray() in place): push the host copy back to the device */
    void commitArray ()  { if (_host) { check (gkc_bloom_set_array (_bloom, _host, gkc_bloom_nbytes (_bloom))); } }

private:
    void check (int rc)  { if (rc != GKC_OK) { throw system::Exception ("BloomDevice: error %d: %s", rc, gkc_last_error(_ctx)); } }
    void store (const Item& item, u_int64_t* key)
    {
        key[0] = item.getVal();                                   // low 64 bits
        if (_words == 2)  { Item hi;  hi.setVal (item);  hi >>= 32;  hi >>= 32;  key[1] = hi.getVal(); }
    }
    void push (const Item& item)  { u_int64_t key[2];  store (item, key);  _pending.insert (_pending.end(), key, key + _words); }

    gkc_ctx* _ctx;  gkc_bloom* _bloom;  tools::misc::BloomKind _kind;  size_t _nbHash, _kmerSize, _words;
    std::vector<u_int64_t> _pending;
    u_int8_t* _host;  bool _hostValid;
};

} } } } } /* end of namespaces. */

#endif /* _GATB_CORE_TOOLS_COLLECTIONS_BLOOM_DEVICE_HPP_ */
