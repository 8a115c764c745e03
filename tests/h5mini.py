"""Minimal HDF5 reader for the subset gatb-core_amd/host/gkc_h5.hpp writes (superblock v0, old-style groups, version-1 object headers,
contiguous datasets, variable-length string attributes in a global heap). Written from the HDF5 File Format Specification, independent
of the writer's code, so that the tests read the files back through a second implementation of the format."""
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Mini:
    def __init__(self, data):
        self.d = bytes(data)
        d = self.d
        assert d[:8] == b"\x89HDF\r\n\x1a\n", "not an HDF5 file"
        assert d[8] == 0 and d[13] == 8 and d[14] == 8, "superblock v0 with 8-byte offsets/lengths expected"
        self.leaf_k, self.int_k = struct.unpack_from("<HH", d, 16)
        base, free, eof, drv = struct.unpack_from("<QQQQ", d, 24)
        assert base == 0 and eof == len(d), (base, eof, len(d))
        name_off, self.root_oh, cache = struct.unpack_from("<QQI", d, 56)
        assert cache == 1
        self.root_bt, self.root_heap = struct.unpack_from("<QQ", d, 80)

    # ---- object headers
    def messages(self, addr):
        d = self.d
        ver, _, nmsg, refc, size = struct.unpack_from("<BBHII", d, addr)
        assert ver == 1
        p = addr + 16; end = p + size; out = []
        for _ in range(nmsg):
            t, sz, flags = struct.unpack_from("<HHB", d, p)
            out.append((t, d[p + 8:p + 8 + sz])); p += 8 + sz
        assert p == end
        return out

    # ---- groups
    def links(self, oh):
        """name -> (object header address, is_group)"""
        d = self.d
        st = [m for t, m in self.messages(oh) if t == 0x0011]
        assert len(st) == 1
        bt, heap = struct.unpack("<QQ", st[0][:16])
        assert d[heap:heap + 4] == b"HEAP"
        seg_size, free_off, seg = struct.unpack_from("<QQQ", d, heap + 8)
        assert d[bt:bt + 4] == b"TREE" and d[bt + 4] == 0 and d[bt + 5] == 0
        used = struct.unpack_from("<H", d, bt + 6)[0]
        out = {}
        for i in range(used):
            child = struct.unpack_from("<Q", d, bt + 24 + 8 + 16 * i)[0]
            assert d[child:child + 4] == b"SNOD"
            n = struct.unpack_from("<H", d, child + 6)[0]
            prev = None
            for j in range(n):
                no, ohaddr, cache = struct.unpack_from("<QQI", d, child + 8 + 40 * j)
                name = d[seg + no:d.index(b"\0", seg + no)].decode()
                assert prev is None or prev < name, "symbol table entries must be sorted"
                prev = name
                out[name] = (ohaddr, cache == 1)
        return out

    def resolve(self, path):
        oh = self.root_oh
        for part in [p for p in path.split("/") if p]:
            oh = self.links(oh)[part][0]
        return oh

    def listdir(self, path):
        return sorted(self.links(self.resolve(path)))

    # ---- attributes (variable-length strings)
    def attrs(self, path):
        d = self.d; out = {}
        for t, m in self.messages(self.resolve(path)):
            if t != 0x000C:
                continue
            ver, _, nsz, tsz, ssz = struct.unpack_from("<BBHHH", m, 0)
            assert ver == 1
            pad = lambda x: (x + 7) // 8 * 8
            p = 8
            name = m[p:p + nsz - 1].decode(); p += pad(nsz)
            dt = m[p:p + tsz]; p += pad(tsz)
            assert dt[0] == 0x19 and dt[1] & 0x0F == 1, "variable-length string expected"
            p += pad(ssz)
            length, gaddr, idx = struct.unpack_from("<IQI", m, p)
            out[name] = self._gheap(gaddr, idx)[:length - 1].decode()
        return out

    def _gheap(self, addr, idx):
        d = self.d
        assert d[addr:addr + 4] == b"GCOL" and d[addr + 4] == 1
        total = struct.unpack_from("<Q", d, addr + 8)[0]
        p = addr + 16
        while p < addr + total:
            oi, ref, _, osz = struct.unpack_from("<HHIQ", d, p)
            if oi == 0:
                break
            if oi == idx:
                return d[p + 16:p + 16 + osz]
            p += 16 + (osz + 7) // 8 * 8
        raise KeyError(idx)

    # ---- datasets
    def _dtype(self, dt, off=0):
        """(numpy dtype, bytes consumed) of a datatype message"""
        cls, ver = dt[off] & 0x0F, dt[off] >> 4
        size = struct.unpack_from("<I", dt, off + 4)[0]
        if cls == 0:
            signed = bool(dt[off + 1] & 0x08)
            if size in (1, 2, 4, 8):
                return np.dtype(("<i" if signed else "<u") + str(size)), 12
            return np.dtype((np.void, size)), 12
        if cls == 6:
            nmemb = struct.unpack_from("<H", dt, off + 1)[0]
            p = off + 8; names, formats, offsets = [], [], []
            for _ in range(nmemb):
                e = dt.index(b"\0", p); name = dt[p:e].decode(); p += (e - p + 1 + 7) // 8 * 8
                moff = struct.unpack_from("<I", dt, p)[0]; p += 4 + 4 + 4 + 4 + 16
                mt, used = self._dtype(dt, p); p += used
                names.append(name); formats.append(mt); offsets.append(moff)
            return np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": size}), p - off
        raise NotImplementedError(cls)

    def dataset(self, path):
        d = self.d
        msgs = dict()
        for t, m in self.messages(self.resolve(path)):
            msgs[t] = m
        sp = msgs[0x0001]; assert sp[0] == 1 and sp[1] == 1
        n = struct.unpack_from("<Q", sp, 8)[0]
        dtype, _ = self._dtype(msgs[0x0003])
        lay = msgs[0x0008]; assert lay[0] == 3 and lay[1] == 1, "contiguous layout expected"
        addr, size = struct.unpack_from("<QQ", lay, 2)
        assert size == n * dtype.itemsize
        if n == 0:
            return np.zeros(0, dtype=dtype)
        return np.frombuffer(d, dtype=dtype, count=n, offset=addr)
