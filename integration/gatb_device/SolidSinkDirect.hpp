/* SolidSinkDirect.hpp — SURVEY 8(f)1 for real: a partition's Count[] goes into its /dsk/solid/<p> dataset WITHOUT H5Dwrite under the storage's global lock.
 *
 * What the reference does (and what the binding did up to round 4): BagHDF5Patch::insert (CollectionHDF5Patch.hpp:262-308) takes the storage's synchronizer
 * (one lock for the whole file, StorageHDF5.hpp:143-145), extends the chunked dataset, selects a hyperslab and calls H5Dwrite — chunk allocation, B-tree inserts
 * and the copy of every 64 KB chunk into the file happen one partition at a time. At 10^8 reads that was 3.3 s for 9.3 GB beside 0.33 s on the device
 * (profiles/r04_dropin_timing_1e8reads.txt).
 *
 * Here the final size of a partition is known when its records arrive (the device counted it), so its dataset is made once, at that size:
 *   under the lock   (metadata only) the empty chunked dataset the reference's Partition constructor created (Storage.tpp:186-203 -> CollectionDataHDF5Patch's
 *                    constructor, CollectionHDF5Patch.hpp:66-80 -> retrieveDatasetId :158-205) is replaced by a CONTIGUOUS one of n items of the same type,
 *                    space allocated early, never filled; H5Dget_offset gives the address of its raw data in the file;
 *   outside the lock the records are written there with pwrite() on a descriptor of the binding's own — by ONE writer thread for the whole file
 *                    (DeviceCounting.hpp, DeviceSession::writer), fed through a ring of page-locked slots the partition commands fetch their Count[] into.
 *                    One, because a file takes buffered writes through its inode's lock: tools/filewrite_probe on the MI355X box's tmpfs — one thread with
 *                    pwrite 9.1 GB/s, 8-128 threads on disjoint ranges 3.2-4.0 GB/s, a shared mapping written by 8 / 32 / 256 threads 3.5 / 2.2 / 0.8 GB/s
 *                    (both were built here first and measured inside dbgh5: no faster than H5Dwrite under the lock).
 * The item type of the dataset is H5Tcopy of the collection's own type (Abundance::hdf5, Abundance.hpp:108-125), i.e. the in-memory layout of Count: what H5Dwrite
 * stores for identical memory and file types is the bytes themselves. Readers go through H5Dread (IterableHDF5Patch::retrieveCache :366-401, HDF5IteratorPatch,
 * gatb-h5dump), which does not care about the layout: the unpatched reference reads the file (tests/test_gpu_dropin.py: every dataset compared through the
 * reference's gatb-h5dump; GraphUnitigs on the device-written .h5). A compressed storage (-storage compress level > 0) keeps the reference's chunked path.
 * GATB_DEVICE_NO_DIRECT_SINK=1 keeps BagHDF5Patch::insert.
 */
#ifndef _GATB_CORE_KMER_IMPL_DEVICE_SOLID_SINK_DIRECT_HPP_
#define _GATB_CORE_KMER_IMPL_DEVICE_SOLID_SINK_DIRECT_HPP_

#include <gatb/tools/storage/impl/Storage.hpp>
#include <gatb/tools/storage/impl/CollectionHDF5Patch.hpp>
#include <gatb/system/api/Exception.hpp>

#include <hdf5/hdf5.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/stat.h>
#include <string>

namespace gatb { namespace core { namespace kmer { namespace impl {

template<class Item>
struct SolidSinkDirect
{
    /** the HDF5 bag behind a collection of the storage, or 0 (file storage, compressed storage) */
    static tools::storage::impl::BagHDF5Patch<Item>* hdf5Bag (tools::collections::Collection<Item>& coll)
    {
        tools::collections::Collection<Item>* ref = &coll;
        tools::storage::impl::CollectionNode<Item>* node = dynamic_cast<tools::storage::impl::CollectionNode<Item>*> (ref);
        if (node != 0)  { ref = node->getRef(); }
        tools::collections::impl::CollectionAbstract<Item>* abstract = dynamic_cast<tools::collections::impl::CollectionAbstract<Item>*> (ref);
        if (abstract == 0)  { return 0; }
        tools::storage::impl::BagHDF5Patch<Item>* bag = dynamic_cast<tools::storage::impl::BagHDF5Patch<Item>*> (abstract->bag());
        if (bag == 0  ||  bag->_common == 0  ||  bag->_common->_compress > 0  ||  bag->_common->_nbItems != 0)  { return 0; }
        return bag;
    }

    /** Main thread, before the partition commands of a pass: the dataset handles the collections of the partition keep open (CollectionDataHDF5Patch::_datasetId,
     *  opened by their constructor) are closed — see insert() for what an open handle costs every H5Ldelete; getDatasetId() reopens on demand. */
    template<class PartitionT> static void closeHandles (PartitionT& partition)
    {
        if (getenv ("GATB_DEVICE_NO_DIRECT_SINK") != 0)  { return; }
        for (size_t i = 0; i < partition.size(); i++)
        {
            tools::storage::impl::BagHDF5Patch<Item>* bag = hdf5Bag (partition[i]);
            if (bag != 0)  { bag->_common->clean(); }
        }
    }

    /** n items (n > 0) will be the whole content of the collection: its dataset is made (see the head of this file); `address` = where its raw data starts in the
     *  file `path`. false: not an (empty, uncompressed) HDF5 collection — nothing was touched, the caller inserts the reference's way. */
    static bool prepare (tools::collections::Collection<Item>& coll, size_t n, uint64_t& address, std::string& path)
    {
        static const bool off = getenv ("GATB_DEVICE_NO_DIRECT_SINK") != 0;
        if (off  ||  n == 0)  { return false; }
        tools::storage::impl::BagHDF5Patch<Item>* bag = hdf5Bag (coll);
        if (bag == 0)  { return false; }
        tools::storage::impl::CollectionDataHDF5Patch<Item>* common = bag->_common;

        system::LocalSynchronizer lock (common->_synchro);      /* HDF5 calls: one thread at a time (the library is not thread-safe) */
        char name[4096];
        if (H5Fget_name (common->_fileId, name, sizeof(name)) <= 0)  { return false; }
        path = name;
        if (common->_datasetId != 0)  { H5Dclose (common->_datasetId);  common->_datasetId = 0; }
        if (H5Ldelete (common->_fileId, common->_name.c_str(), H5P_DEFAULT) < 0)
            throw system::Exception ("device sink: H5Ldelete of the empty dataset %s failed", common->_name.c_str());
        hsize_t dims    = n;
        hid_t   spaceId = H5Screate_simple (1, &dims, NULL);
        hid_t   propId  = H5Pcreate (H5P_DATASET_CREATE);
        H5Pset_layout     (propId, H5D_CONTIGUOUS);
        H5Pset_alloc_time (propId, H5D_ALLOC_TIME_EARLY);
        H5Pset_fill_time  (propId, H5D_FILL_TIME_NEVER);
        hid_t   typeId  = H5Tcopy (common->_typeId);
        hid_t   dataset = H5Dcreate2 (common->_fileId, common->_name.c_str(), typeId, spaceId, H5P_DEFAULT, propId, H5P_DEFAULT);
        H5Tclose (typeId);  H5Pclose (propId);  H5Sclose (spaceId);
        if (dataset < 0)  { throw system::Exception ("device sink: H5Dcreate2 (contiguous, %llu items) of %s failed", (unsigned long long) n, common->_name.c_str()); }
        const haddr_t at = H5Dget_offset (dataset);
        common->_nbItems = n;
        /* closed again at once (getDatasetId() reopens it for whoever reads): H5Ldelete walks EVERY open object id of the file to keep their path names
         * right (H5G_name_replace -> H5I_iterate) — with the 3884 datasets of a partition held open, 0.76 ms per delete, 3 s in all; with none, 6 us */
        H5Dclose (dataset);
        if (at == HADDR_UNDEF)  { throw system::Exception ("device sink: the file driver of %s gives no raw-data address (GATB_DEVICE_NO_DIRECT_SINK=1 keeps the reference's insert)", name); }
        address = (uint64_t) at;
        return true;
    }

    /** bytes at an offset of the file, the plain way */
    static void writeAt (int fd, uint64_t offset, const void* src, size_t n)
    {
        const char* p = (const char*) src;  off_t at = (off_t) offset;
        while (n > 0)
        {
            const ssize_t w = pwrite (fd, p, n, at);
            if (w < 0  &&  errno == EINTR)  { continue; }
            if (w <= 0)  { throw system::Exception ("device sink: pwrite: %s", strerror (errno)); }
            p += w;  at += w;  n -= (size_t) w;
        }
    }

    /** n items in host memory as the whole content of the collection, written by the calling thread. */
    static bool insert (tools::collections::Collection<Item>& coll, const Item* items, size_t n)
    {
        uint64_t address = 0;  std::string path;
        if (!prepare (coll, n, address, path))  { return false; }
        const int fd = ::open (path.c_str(), O_WRONLY);
        if (fd < 0)  { throw system::Exception ("device sink: open (%s): %s", path.c_str(), strerror (errno)); }
        try  { writeAt (fd, address, items, n * sizeof(Item)); }  catch (...)  { ::close (fd);  throw; }
        ::close (fd);
        return true;
    }
};

} } } }

#endif
