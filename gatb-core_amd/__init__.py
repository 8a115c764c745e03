"""gatb-core_amd — MI355X-native implementation of GATB-Core's DSK k-mer counting hot path.

Layout:
  csrc/   hand-written HIP kernels (gfx950) + the C-ABI implementation  -> csrc/libgkc_hip.so   (include/gkc.h)
  host/   C++ host layer mirroring gatb::core::kmer::impl (SortingCountAlgorithm<span>, ICountProcessor<span>, IBloom<T>)
  gkc.py  thin ctypes binding over the C-ABI (tests, bench, Python host)

The directory name contains a hyphen (it is the name the build contract asks for), so import it with
`tests/conftest.py`'s loader or `importlib` — see `load()` in the repo-root `__graft_entry__.py`.
"""
from . import gkc  # noqa: F401
