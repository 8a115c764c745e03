"""Developer check (GPU box): counting 2e7 reads (2.4e9 k-mers; ~12 GB of results) with most of the HBM taken by somebody else. The
Stage-B planner has to fit its batches into what is left, or the call must fail loudly (GkcError) and leave the process usable."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge, bench
gkc = ge.load().gkc
n, k, m, parts = 20_000_000, 31, 10, 1024
for leave_gb in (60, 30, 22, 18, 12, 6):
    free, _ = torch.cuda.mem_get_info()
    hog = torch.empty(int(free - leave_gb * 1e9), dtype=torch.uint8, device="cuda")
    try:
        c = gkc.Counter(0)
        c.configure(k, m, parts, bench.repart_for_bench(m, parts))
        db, do = c.synth_reads_device(2, n, 150, n * 5, 10000)
        cs, nv = c.kmer_checksum_device(db, do, n, n * 150)
        c.begin_pass(0); c.push_reads_device(db, do, n, n * 150); c.finish_pass()
        print("left %d GB: counted, checksum %s" % (leave_gb, "ok" if c.result_checksum() == (cs, nv) else "MISMATCH"), flush=True)
    except gkc.GkcError as e:
        print("left %d GB: refused loudly: %s" % (leave_gb, str(e)[:150]), flush=True)
    c = None; db = do = None
    del hog; torch.cuda.empty_cache()
