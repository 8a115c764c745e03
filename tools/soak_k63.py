"""Developer soak (GPU box): the 16-byte-key path (k = 63, 47, 33: pair scatter through the 128-bit LDS exchange) at full size, eight rounds of the
size-independent property test (multiset checksum, ascending datasets, histogram sums) = about 1.7e11 keys through ds_wrxchg2_rtn_b64. Last run: all ok."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import __graft_entry__ as ge
spec = importlib.util.spec_from_file_location("tgp", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "test_gpu_parity.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
gkc = ge.load().gkc
for it in range(8):
    for (k, n, parts) in [(63, 100_000_000, 4096), (47, 60_000_000, 2048), (33, 60_000_000, 2048)]:
        m.test_size_independent_properties(gkc, k, n, parts, it % 2)          # odd rounds: the repeat-rich generator (GKC_SYNTH_SKEWED)
    print("round", it, "ok", flush=True)
