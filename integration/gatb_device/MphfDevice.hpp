/* MphfDevice.hpp — the BooPHF build and the abundance map of MPHFAlgorithm<span> (kmer/impl/MPHFAlgorithm.cpp:150-275) on the MI355X, reference-side binding
 * (compiled against the reference's headers by integration/check_integration.sh; used by the hunks of integration/gatb-core.device.patch in
 * MPHFAlgorithm.cpp).
 *
 *   build     the function is built on the device — from the solid k-mers where the counting step left them in HBM (gkc_mphf_build_solid), or from the keys
 *             of the Iterable read into a host array (gkc_mphf_build) — and its byte stream (gkc_mphf_save: the stream of boomphf::mphf::save, what
 *             BooPHF::save writes, tools/collections/impl/BooPHF.hpp:318-327) goes into the collection <group>/<name> with the "nb_keys" property, exactly
 *             what `_abundanceMap->save (_group, _name)` leaves (MPHFAlgorithm.cpp:175-177). The caller then LOADS it with the reference's own
 *             MapMPHF::load (MapMPHF.hpp:192-201 -> BooPHF::load): the object every consumer holds is the reference's BooPHF, read from the bytes the
 *             device wrote — no class of the reference changes its layout.
 *   populate  gkc_mphf_abundance_map: cell[code(kmer)] = index of the k-mer's abundance in MapMPHF's discretization table (MPHFAlgorithm.cpp:236-266), for
 *             the solid k-mers in HBM; copied into the map's value array.
 * Applies to key types whose bytes are the device's key (LargeInt<1>: 8 bytes, LargeInt<2>: 16 bytes — BooPHF hashes the raw bytes of the key,
 * BooPHF.hpp:53-61) and one-byte abundance cells; everything else keeps the CPU path (build() returns false). */
#ifndef _GATB_CORE_KMER_IMPL_MPHF_DEVICE_HPP_
#define _GATB_CORE_KMER_IMPL_MPHF_DEVICE_HPP_

#include <gatb/tools/storage/impl/Storage.hpp>
#include <gatb/tools/collections/api/Iterable.hpp>
#include <gatb/tools/misc/impl/Stringify.hpp>
#include <gatb/system/api/Exception.hpp>

#include <gatb_device/DeviceContext.hpp>

#include <map>
#include <mutex>
#include <vector>
#include <string>
#include <stdio.h>
#include <stdlib.h>

namespace gatb { namespace core { namespace kmer { namespace impl {

class MphfDevice
{
public:
    /** Builds the function of `keys` on the device and stores its stream as <group>/<name>. `owner` identifies the algorithm instance: the device function is
     *  kept under it until release(owner), so that populate() can use it. False (nothing written): not a case for the device — the caller builds on the CPU. */
    template<typename Type>
    static bool build (const void* owner, tools::storage::impl::Group& group, const std::string& name, tools::collections::Iterable<Type>* keys, size_t& dataSize)
    {
        if (getenv ("GATB_DEVICE_NO_MPHF") != 0  ||  (sizeof(Type) != 8  &&  sizeof(Type) != 16)  ||  keys == 0)  { return false; }
        device::DeviceContext& dc = device::DeviceContext::singleton();
        gkc_ctx* ctx = dc.ctx();
        if (ctx == 0)  { return false; }
        const u_int64_t n = (u_int64_t) keys->getNbItems();
        gkc_mphf* m = 0;
        bool resident = dc.residentMatches (n, sizeof(Type));
        if (resident)  { check (ctx, gkc_mphf_build_solid (ctx, &m)); }
        else
        {
            /* the keys of the Iterable in iteration order (getSolidKmers() order); their raw bytes are the device keys */
            std::vector<Type> host;  host.reserve (n);
            tools::dp::Iterator<Type>* it = keys->iterator();  LOCAL (it);
            for (it->first(); !it->isDone(); it->next())  { host.push_back (it->item()); }
            check (ctx, gkc_mphf_build (ctx, host.data(), host.size(), (uint32_t) sizeof(Type), sizeof(Type) == 8 ? 31 : 63, &m));
        }
        if (gkc_mphf_size (m) != n)
        {
            const unsigned long long got = gkc_mphf_size (m);  gkc_mphf_destroy (m);
            throw system::Exception ("device MPHF: built over %llu keys, expected %llu", got, (unsigned long long) n);
        }
        std::vector<uint8_t> bytes (gkc_mphf_save_size (m));
        check (ctx, gkc_mphf_save (m, bytes.data(), bytes.size()));
        {
            tools::storage::impl::Storage::ostream os (group, name);
            os.write (reinterpret_cast<const char*> (bytes.data()), bytes.size());
            os.flush();
        }
        group.addProperty ("nb_keys", tools::misc::impl::Stringify().format ("%d", (int) n));      /* as BooPHF::save does (BooPHF.hpp:325, same "%d") */
        dataSize = bytes.size();
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)
        {
            fprintf (stderr, "[device mphf] %llu keys (%s), %llu bytes stored as %s\n", (unsigned long long) n,
                     resident ? "the solid k-mers where Stage B left them: gkc_mphf_build_solid" : "read from the Iterable: gkc_mphf_build", (unsigned long long) bytes.size(), name.c_str());
        }
        std::lock_guard<std::mutex> guard (lock());
        Entry& e = registry() [owner];  e.mphf = m;  e.resident = resident;
        return true;
    }

    /** MPHFAlgorithm::populate on the device: cells[0 .. n) = the abundance index of every solid k-mer at its hash code. False: no device function under
     *  `owner`, or the solid k-mers are not in HBM (the function was built from the Iterable) — the caller's CPU loop fills the map. */
    static bool populate (const void* owner, u_int8_t* cells, size_t n, size_t& nbAbovePrecision)
    {
        gkc_mphf* m = 0;
        {
            std::lock_guard<std::mutex> guard (lock());
            std::map<const void*, Entry>::iterator it = registry().find (owner);
            if (it == registry().end()  ||  !it->second.resident)  { return false; }
            m = it->second.mphf;
        }
        gkc_ctx* ctx = device::DeviceContext::singleton().ctx();
        if (gkc_mphf_size (m) != n)  { return false; }
        uint64_t above = 0;
        check (ctx, gkc_mphf_abundance_map (m, ctx, cells, n, &above));
        nbAbovePrecision = (size_t) above;
        if (getenv ("GATB_DEVICE_VERBOSE") != 0)  { fprintf (stderr, "[device mphf] abundance map of %llu cells filled on the device (gkc_mphf_abundance_map)\n", (unsigned long long) n); }
        return true;
    }

    static void release (const void* owner)
    {
        std::lock_guard<std::mutex> guard (lock());
        std::map<const void*, Entry>::iterator it = registry().find (owner);
        if (it != registry().end())  { gkc_mphf_destroy (it->second.mphf);  registry().erase (it); }
    }

private:
    struct Entry  { gkc_mphf* mphf;  bool resident;  Entry () : mphf(0), resident(false) {} };
    static std::map<const void*, Entry>& registry ()  { static std::map<const void*, Entry> r;  return r; }
    static std::mutex& lock ()  { static std::mutex m;  return m; }
    static void check (gkc_ctx* ctx, int rc)  { if (rc != GKC_OK) { throw system::Exception ("device MPHF: error %d: %s", rc, gkc_last_error (ctx)); } }
};

} } } } /* end of namespaces. */

#endif /* _GATB_CORE_KMER_IMPL_MPHF_DEVICE_HPP_ */
