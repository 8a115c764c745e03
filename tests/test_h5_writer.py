"""The native HDF5 writer of the C++ host layer (gatb-core_amd/host/gkc_h5.hpp), read back through tests/h5mini.py — a second,
independent implementation of the same slice of the HDF5 file format. No GPU involved."""
import os
import subprocess

import numpy as np
import pytest

import __graft_entry__ as ge
from tests.h5mini import H5Mini

HOST = os.path.join(ge.ROOT, "gatb-core_amd", "host")


@pytest.fixture(scope="module")
def writer(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("h5") / "test_h5")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(HOST, "test_h5.cpp")], check=True)
    return exe


@pytest.mark.parametrize("nd", [1, 9, 5000])
def test_groups_datasets_attributes(writer, tmp_path, nd):
    out = str(tmp_path / "t.h5")
    subprocess.run([writer, out, str(nd)], check=True)
    h = H5Mini(open(out, "rb").read())
    assert h.listdir("/") == sorted(["dsk", "minimizers", "wide"])
    assert h.attrs("/") == {"kmer_size": "31", "xml": "\n<a>\n   <b>some longer text & stuff</b>\n</a>"}
    assert h.attrs("/dsk/solid") == {"nb_partitions": str(nd)}
    assert h.listdir("/dsk/solid") == sorted(str(i) for i in range(nd))
    for i in ([0, nd - 1, nd // 2] if nd > 3 else range(nd)):
        a = h.dataset("/dsk/solid/%d" % i)
        assert len(a) == i % 7 and a.dtype.itemsize == 16 and a.dtype.names == ("value", "abundance")
        assert a["value"].tolist() == [1000003 * i + j for j in range(i % 7)] and a["abundance"].tolist() == [i + j for j in range(i % 7)]
    w = h.dataset("/wide/0")
    assert w.dtype.itemsize == 32 and w.dtype.fields["abundance"][1] == 16
    assert [int.from_bytes(bytes(x), "little") for x in w["value"]] == [((j + 1) << 100) | (j + 5) for j in range(3)]
    assert w["abundance"].tolist() == [9, 10, 11]
    r = h.dataset("/minimizers/minimRepart")
    assert r.dtype == np.uint8 and np.array_equal(r, (np.arange(100000) * 7).astype(np.uint8))
