/* DeviceContext.hpp — the ONE gkc_ctx of the process (one process = one GPU), shared by everything of the reference-side binding that talks to
 * libgkc_hip.so: DeviceSession (counting, DeviceCounting.hpp), BloomDevice<Item> (BloomDevice.hpp) and MphfDevice (MphfDevice.hpp). It also keeps what the
 * later steps of the reference's pipeline need to know about the counting step: whether the solid k-mers it produced are still in HBM, so that
 * BloomAlgorithm / MPHFAlgorithm can work on them where they lie instead of reading /dsk/solid back from the storage.
 * Includes nothing of the reference: tools/collections must not depend on kmer/ (BloomDevice.hpp is included from Bloom.hpp). */
#ifndef _GATB_CORE_DEVICE_CONTEXT_HPP_
#define _GATB_CORE_DEVICE_CONTEXT_HPP_

#include <gkc.h>

#include <mutex>
#include <string>
#include <stdint.h>
#include <stdlib.h>

namespace gatb { namespace core { namespace device {

class DeviceContext
{
public:
    static DeviceContext& singleton ()  { static DeviceContext s;  return s; }

    /** the context, created on first use; 0 when there is no usable device (the message is kept for the caller's exception) */
    gkc_ctx* ctx ()
    {
        std::lock_guard<std::mutex> guard (_lock);
        if (_ctx == 0  &&  !_failed)
        {
            const char* dev = getenv ("GATB_DEVICE_ORDINAL");
            if (gkc_create (dev ? atoi (dev) : 0, &_ctx) != GKC_OK)  { _failed = true;  _error = gkc_last_error (0);  _ctx = 0; }
        }
        return _ctx;
    }
    bool available ()  { return ctx() != 0; }
    const std::string& error () const  { return _error; }

    /** What the counting step left in HBM. Set by DeviceSession when the last pass has been counted in bulk mode (the solidity window was applied on the
     *  device, so the device's datasets ARE /dsk/solid), cleared when a new count starts or a pass is released. */
    struct Resident
    {
        bool     on;            /**< every dataset of the run is in HBM, in dataset order, holding exactly the solid k-mers */
        uint64_t nbSolid;       /**< their number: a consumer checks it against the Iterable it was given */
        uint32_t kmerSize;
        uint32_t keyBytes;      /**< 8 (k <= 31) or 16: the consumer's Type must have this size */
        Resident () : on(false), nbSolid(0), kmerSize(0), keyBytes(0) {}
    };
    void setResident (const Resident& r)  { std::lock_guard<std::mutex> guard (_lock);  _resident = r; }
    Resident resident ()                  { std::lock_guard<std::mutex> guard (_lock);  return _resident; }
    /** the solid set a consumer holds (nbItems keys of typeBytes bytes) is the one in HBM */
    bool residentMatches (uint64_t nbItems, size_t typeBytes)
    {
        const Resident r = resident();
        return r.on  &&  r.nbSolid == nbItems  &&  r.keyBytes == typeBytes  &&  getenv ("GATB_DEVICE_NO_RESIDENT") == 0;
    }

    ~DeviceContext ()  { if (_ctx)  { gkc_destroy (_ctx); } }

private:
    DeviceContext () : _ctx(0), _failed(false) {}
    std::mutex  _lock;
    gkc_ctx*    _ctx;
    bool        _failed;
    std::string _error;
    Resident    _resident;
};

} } } /* end of namespaces. */

#endif /* _GATB_CORE_DEVICE_CONTEXT_HPP_ */
