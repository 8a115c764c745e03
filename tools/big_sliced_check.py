"""Developer check (GPU box): the size-independent property test at large sizes with FEW HUGE partitions (sliced batches: several workgroups per partition at the
sizes where the Stage-B planner shrinks its batches). Not part of the test suite."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import importlib.util, numpy as np
import __graft_entry__ as ge
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py")); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
gkc = ge.load().gkc
for (k, n, parts) in [(31, 200_000_000, 512), (63, 100_000_000, 256), (31, 150_000_000, 64)]:
    m.test_size_independent_properties(gkc, k, n, parts)
    print("ok", k, n, parts, flush=True)
