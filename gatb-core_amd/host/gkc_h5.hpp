// gkc_h5.hpp — a small native HDF5 writer for the datasets the DSK step leaves in its .h5.
//
// The reference writes its results through the vendored HDF5 library (tools/storage/impl/StorageHDF5.hpp:205-330, CollectionHDF5Patch.hpp:
// 160-310): groups, string attributes and one-dimensional datasets of integers or of the compound Count type. The library is not buildable
// here (cmake), and only a thin slice of the file format is needed, so this header emits that slice directly, following the published
// HDF5 File Format Specification 2.0 (old-style groups = B-tree v1 + symbol table node + local heap, version-1 object headers):
//   * superblock v0; one B-tree leaf + ONE symbol table node per group (the group node K of the superblock is sized for the largest
//     group, e.g. dsk/solid with one dataset per partition);
//   * datasets with contiguous layout (the reference reads with H5Dread hyperslabs: any layout is fine for a reader);
//   * attributes as variable-length strings in a global heap collection, exactly the type the reference's getProperty reads
//     (H5T_C_S1, H5T_VARIABLE; StorageHDF5.hpp:296-330).
// Readers: the HDF5 library (h5dump, Storage::load of the reference). tests/h5mini.py is an independent minimal reader for the tests.
#pragma once
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gkc_h5 {

static const uint64_t UNDEF = ~0ULL;

/** datatype descriptions (HDF5 datatype message, spec IV.A.2.d) */
struct Type {
    std::vector<uint8_t> msg;      // encoded datatype message
    uint32_t size;                 // element size in bytes
    static void put(std::vector<uint8_t>& v, uint64_t x, int n) { for (int i = 0; i < n; i++) v.push_back((uint8_t)(x >> (8 * i))); }
    /** fixed-point, little endian */
    static Type integer(uint32_t bytes, bool is_signed, uint32_t precision_bits = 0) {
        Type t; t.size = bytes;
        t.msg.push_back(0x10 | 0);                                   // version 1, class 0
        t.msg.push_back(is_signed ? 0x08 : 0x00); t.msg.push_back(0); t.msg.push_back(0);
        put(t.msg, bytes, 4);
        put(t.msg, 0, 2); put(t.msg, precision_bits ? precision_bits : bytes * 8, 2);     // bit offset, precision
        return t;
    }
    /** compound, version 1 member encoding (name padded to 8, offset, 28 bytes of array fields, member type) */
    static Type compound(uint32_t bytes, const std::vector<std::pair<std::string, std::pair<uint32_t, Type>>>& members) {
        Type t; t.size = bytes;
        t.msg.push_back(0x10 | 6);
        t.msg.push_back((uint8_t)members.size()); t.msg.push_back((uint8_t)(members.size() >> 8)); t.msg.push_back(0);
        put(t.msg, bytes, 4);
        for (auto& m : members) {
            const std::string& name = m.first;
            const size_t start = t.msg.size();
            for (char ch : name) t.msg.push_back((uint8_t)ch);
            t.msg.push_back(0);
            while ((t.msg.size() - start) % 8) t.msg.push_back(0);     // the name field occupies a multiple of 8 bytes
            put(t.msg, m.second.first, 4);                           // byte offset of the member
            t.msg.push_back(0); t.msg.push_back(0); t.msg.push_back(0); t.msg.push_back(0);   // dimensionality 0 + 3 reserved
            put(t.msg, 0, 4); put(t.msg, 0, 4);                      // dimension permutation, reserved
            put(t.msg, 0, 4); put(t.msg, 0, 4); put(t.msg, 0, 4); put(t.msg, 0, 4);           // 4 dimension sizes
            t.msg.insert(t.msg.end(), m.second.second.msg.begin(), m.second.second.msg.end());
        }
        return t;
    }
    /** variable-length string, null terminated, ASCII: 16 bytes per element in the file (length, heap address, index) */
    static Type vlen_string() {
        Type t; t.size = 16;
        t.msg.push_back(0x10 | 9);
        t.msg.push_back(0x01); t.msg.push_back(0x00); t.msg.push_back(0);              // type = string, padding null-terminate, charset ASCII
        put(t.msg, 16, 4);
        // base type as the library encodes H5T_C_S1's character: a one-byte unsigned integer
        const Type ch = integer(1, false);
        t.msg.insert(t.msg.end(), ch.msg.begin(), ch.msg.end());
        return t;
    }
};

class File {
public:
    /** group_capacity >= the number of links of the largest group (the symbol table node of every group is sized for it) */
    explicit File(size_t group_capacity = 64) {
        _leafK = (uint16_t)std::min<size_t>(32767, std::max<size_t>(4, (group_capacity + 1) / 2));
        _buf.assign(96, 0);                                            // superblock, filled by finish()
        _groups["/"] = Group();
        _gcol_addr = UNDEF;
    }
    void add_group(const std::string& path) { ensure_group(path); }
    void set_attribute(const std::string& group_path, const std::string& key, const std::string& value) { ensure_group(group_path).attrs[key] = value; }
    /** one-dimensional dataset with contiguous layout; maxdims unlimited is what the reference's collections declare, but a contiguous
     *  dataset must have fixed dimensions: the current size is also the maximum */
    void add_dataset(const std::string& path, const Type& type, const void* data, uint64_t n_items) {
        const size_t slash = path.rfind('/');
        const std::string parent = slash == 0 ? "/" : path.substr(0, slash), name = path.substr(slash + 1);
        Group& g = ensure_group(parent);
        const uint64_t nbytes = n_items * type.size;
        const uint64_t data_addr = nbytes ? append(data, nbytes) : UNDEF;
        // object header: dataspace, datatype, fill value, layout
        std::vector<std::vector<uint8_t>> msgs; std::vector<uint16_t> types;
        { std::vector<uint8_t> m = { 1, 1, 0, 0, 0, 0, 0, 0 }; Type::put(m, n_items, 8); msgs.push_back(m); types.push_back(0x0001); }
        msgs.push_back(type.msg); types.push_back(0x0003);
        { std::vector<uint8_t> m = { 2, 2, 2, 0 }; msgs.push_back(m); types.push_back(0x0005); }          // fill value v2: late alloc, write if set, undefined
        { std::vector<uint8_t> m = { 3, 1 }; Type::put(m, data_addr, 8); Type::put(m, nbytes, 8); msgs.push_back(m); types.push_back(0x0008); }
        g.links[name] = std::make_pair(write_object_header(msgs, types), false);
    }
    /** lays out groups, heaps and the superblock; returns the file image */
    const std::vector<uint8_t>& finish() {
        // global heap with every attribute string
        std::vector<std::string> strings; std::map<std::string, uint32_t> sidx;
        for (auto& kv : _groups) for (auto& a : kv.second.attrs) if (!sidx.count(a.second)) { strings.push_back(a.second); sidx[a.second] = (uint32_t)strings.size(); }
        if (!strings.empty()) {
            std::vector<uint8_t> gc; for (char ch : std::string("GCOL")) gc.push_back((uint8_t)ch);
            gc.push_back(1); gc.push_back(0); gc.push_back(0); gc.push_back(0);
            Type::put(gc, 0, 8);                                       // collection size, patched below
            for (size_t i = 0; i < strings.size(); i++) {
                const std::string& sv = strings[i];
                Type::put(gc, i + 1, 2); Type::put(gc, 1, 2); Type::put(gc, 0, 4); Type::put(gc, sv.size() + 1, 8);
                for (char ch : sv) gc.push_back((uint8_t)ch);
                gc.push_back(0);
                while (gc.size() % 8) gc.push_back(0);
            }
            uint64_t total = std::max<uint64_t>(4096, (gc.size() + 16 + 7) / 8 * 8);
            // object 0: the free space that closes the collection
            const uint64_t free_sz = total - gc.size();
            Type::put(gc, 0, 2); Type::put(gc, 0, 2); Type::put(gc, 0, 4); Type::put(gc, free_sz, 8);
            gc.resize((size_t)total, 0);
            for (int i = 0; i < 8; i++) gc[8 + i] = (uint8_t)(total >> (8 * i));
            _gcol_addr = append(gc.data(), gc.size());
        }
        // groups, children before parents (longest path first)
        std::vector<std::string> order; for (auto& kv : _groups) order.push_back(kv.first);
        std::sort(order.begin(), order.end(), [](const std::string& a, const std::string& b) { return a.size() != b.size() ? a.size() > b.size() : a < b; });
        uint64_t root_oh = 0, root_bt = 0, root_heap = 0;
        for (const std::string& path : order) {
            Group& g = _groups[path];
            uint64_t bt, heap; const uint64_t oh = write_group(g, sidx, bt, heap);
            if (path == "/") { root_oh = oh; root_bt = bt; root_heap = heap; continue; }
            const size_t slash = path.rfind('/');
            const std::string parent = slash == 0 ? "/" : path.substr(0, slash), name = path.substr(slash + 1);
            _groups[parent].links[name] = std::make_pair(oh, true);
            _groups[parent].child_bt[name] = bt; _groups[parent].child_heap[name] = heap;
        }
        // superblock v0
        uint8_t* s = _buf.data();
        const uint8_t sig[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
        memcpy(s, sig, 8);
        s[8] = 0; s[9] = 0; s[10] = 0; s[11] = 0; s[12] = 0; s[13] = 8; s[14] = 8; s[15] = 0;
        s[16] = (uint8_t)_leafK; s[17] = (uint8_t)(_leafK >> 8); s[18] = 16; s[19] = 0;                 // group leaf node K, internal node K
        memset(s + 20, 0, 4);
        put64(s + 24, 0); put64(s + 32, UNDEF); put64(s + 40, _buf.size()); put64(s + 48, UNDEF);
        put64(s + 56, 0); put64(s + 64, root_oh); s[72] = 1; s[73] = s[74] = s[75] = 0; memset(s + 76, 0, 4);
        put64(s + 80, root_bt); put64(s + 88, root_heap);
        return _buf;
    }
private:
    struct Group {
        std::map<std::string, std::string> attrs;
        std::map<std::string, std::pair<uint64_t, bool>> links;       // name -> (object header address, is a group)
        std::map<std::string, uint64_t> child_bt, child_heap;
    };
    static void put64(uint8_t* p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
    Group& ensure_group(const std::string& path) {
        if (path.empty() || path[0] != '/') throw std::runtime_error("HDF5 path must be absolute: " + path);
        if (path != "/") { const size_t slash = path.rfind('/'); ensure_group(slash == 0 ? "/" : path.substr(0, slash)); }
        return _groups[path];
    }
    uint64_t append(const void* p, uint64_t n) {
        while (_buf.size() % 8) _buf.push_back(0);
        const uint64_t at = _buf.size();
        _buf.insert(_buf.end(), (const uint8_t*)p, (const uint8_t*)p + n);
        return at;
    }
    /** version-1 object header with the given messages in one chunk */
    uint64_t write_object_header(const std::vector<std::vector<uint8_t>>& msgs, const std::vector<uint16_t>& types) {
        std::vector<uint8_t> body;
        for (size_t i = 0; i < msgs.size(); i++) {
            const size_t dsz = (msgs[i].size() + 7) / 8 * 8;
            Type::put(body, types[i], 2); Type::put(body, dsz, 2); body.push_back(types[i] == 0x0003 ? 1 : 0); body.push_back(0); body.push_back(0); body.push_back(0);
            body.insert(body.end(), msgs[i].begin(), msgs[i].end());
            body.resize(body.size() + (dsz - msgs[i].size()), 0);
        }
        std::vector<uint8_t> oh = { 1, 0 };
        Type::put(oh, msgs.size(), 2); Type::put(oh, 1, 4); Type::put(oh, body.size(), 4); Type::put(oh, 0, 4);       // 16-byte prefix (4 bytes of alignment)
        oh.insert(oh.end(), body.begin(), body.end());
        return append(oh.data(), oh.size());
    }
    /** attribute message v1: a scalar-like (1-element simple dataspace) variable-length string whose characters sit in the global heap */
    std::vector<uint8_t> attribute_message(const std::string& key, uint32_t heap_index, uint32_t length) {
        const Type vt = Type::vlen_string();
        const std::vector<uint8_t> space = { 1, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0 };   // rank 1, dims {1}, max dims {1}
        std::vector<uint8_t> m = { 1, 0 };
        Type::put(m, key.size() + 1, 2); Type::put(m, vt.msg.size(), 2); Type::put(m, space.size(), 2);
        for (char ch : key) m.push_back((uint8_t)ch);
        m.push_back(0); while (m.size() % 8) m.push_back(0);
        m.insert(m.end(), vt.msg.begin(), vt.msg.end()); while (m.size() % 8) m.push_back(0);
        m.insert(m.end(), space.begin(), space.end()); while (m.size() % 8) m.push_back(0);
        Type::put(m, length, 4); Type::put(m, _gcol_addr, 8); Type::put(m, heap_index, 4);                     // the variable-length element
        return m;
    }
    uint64_t write_group(Group& g, const std::map<std::string, uint32_t>& sidx, uint64_t& bt_addr, uint64_t& heap_addr) {
        // local heap: "" at offset 0, then the link names (8-byte aligned)
        std::vector<uint8_t> hd(8, 0); std::map<std::string, uint64_t> off;
        for (auto& l : g.links) { off[l.first] = hd.size(); for (char ch : l.first) hd.push_back((uint8_t)ch); hd.push_back(0); while (hd.size() % 8) hd.push_back(0); }
        const uint64_t free_off = hd.size(); hd.resize(hd.size() + 16, 0);                                   // one free block closes the segment
        put64(hd.data() + free_off, 1); put64(hd.data() + free_off + 8, 16);                                   // next = H5HL_FREE_NULL, size 16
        const uint64_t data_addr = append(hd.data(), hd.size());
        std::vector<uint8_t> hp; for (char ch : std::string("HEAP")) hp.push_back((uint8_t)ch);
        hp.push_back(0); hp.push_back(0); hp.push_back(0); hp.push_back(0);
        Type::put(hp, hd.size(), 8); Type::put(hp, free_off, 8); Type::put(hp, data_addr, 8);
        heap_addr = append(hp.data(), hp.size());
        // symbol table node (all links, sorted by name = std::map order = strcmp order for these ASCII names)
        if (g.links.size() > (size_t)2 * _leafK) throw std::runtime_error("HDF5 writer: group holds more links than the file was sized for");
        std::vector<uint8_t> sn; for (char ch : std::string("SNOD")) sn.push_back((uint8_t)ch);
        sn.push_back(1); sn.push_back(0); Type::put(sn, g.links.size(), 2);
        uint64_t last_off = 0;
        for (auto& l : g.links) {
            Type::put(sn, off[l.first], 8); Type::put(sn, l.second.first, 8);
            if (l.second.second) { Type::put(sn, 1, 4); Type::put(sn, 0, 4); Type::put(sn, g.child_bt[l.first], 8); Type::put(sn, g.child_heap[l.first], 8); }
            else { Type::put(sn, 0, 4); Type::put(sn, 0, 4); Type::put(sn, 0, 8); Type::put(sn, 0, 8); }
            last_off = off[l.first];
        }
        sn.resize(8 + (size_t)2 * _leafK * 40, 0);
        const uint64_t snod_addr = append(sn.data(), sn.size());
        // B-tree v1 leaf (node type 0 = group nodes, level 0) with one child; internal K = 16 -> 2K+1 keys, 2K children
        std::vector<uint8_t> bt; for (char ch : std::string("TREE")) bt.push_back((uint8_t)ch);
        bt.push_back(0); bt.push_back(0); Type::put(bt, g.links.empty() ? 0 : 1, 2); Type::put(bt, UNDEF, 8); Type::put(bt, UNDEF, 8);
        Type::put(bt, 0, 8); Type::put(bt, snod_addr, 8); Type::put(bt, last_off, 8);
        bt.resize(24 + (2 * 16 + 1) * 8 + 2 * 16 * 8, 0);
        bt_addr = append(bt.data(), bt.size());
        // object header: symbol table message + attributes
        std::vector<std::vector<uint8_t>> msgs; std::vector<uint16_t> types;
        { std::vector<uint8_t> m; Type::put(m, bt_addr, 8); Type::put(m, heap_addr, 8); msgs.push_back(m); types.push_back(0x0011); }
        for (auto& a : g.attrs) { msgs.push_back(attribute_message(a.first, sidx.at(a.second), (uint32_t)a.second.size() + 1)); types.push_back(0x000C); }
        return write_object_header(msgs, types);
    }
    std::vector<uint8_t> _buf; std::map<std::string, Group> _groups; uint16_t _leafK; uint64_t _gcol_addr;
};

}  // namespace gkc_h5
