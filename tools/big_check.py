"""Developer check (GPU box): the size-independent property test of tests/test_gpu_parity.py at capacity-regime sizes of one MI355X
(2.5e8 reads at k=31 = 3e10 k-mers, 1.5e8 reads at k=63), where the Stage-B planner has to shrink its batches. Not part of the test suite (minutes of GPU time)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util, numpy as np
import __graft_entry__ as ge
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
gkc = ge.load().gkc
for (k, n, parts) in [(31, 250_000_000, 8192), (63, 150_000_000, 8192), (31, 200_000_000, 4096)]:
    m.test_size_independent_properties.__wrapped__(gkc, k, n, parts) if hasattr(m.test_size_independent_properties, "__wrapped__") else m.test_size_independent_properties(gkc, k, n, parts)
    print("ok", k, n, parts, flush=True)
