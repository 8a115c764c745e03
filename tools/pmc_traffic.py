"""profiles/pmc_traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the guide prescribes).
usage: pmc_traffic.py fetch.db write.db '<workload string>' out.json [steps the profiled command ran: default 1]
Units: FETCH_SIZE/WRITE_SIZE are KiB per dispatch. gfx950 correction (MI355X_MICROARCH.md §HBM, calibrated for this library's access patterns in
profiles/r05_fetch_size_calibration.txt): FETCH_SIZE reports half of the bytes of coalesced reads (16 / 8 / 4 bytes per lane, wave-contiguous runs) -> read bytes = 2 * FETCH_SIZE * 1024
for the streaming kernels; per-lane gathers are counted one 64-byte line per read at face value -> 1 * FETCH_SIZE * 1024 for the kernels listed in GATHER; byte-per-lane loads are not
counted at all (k_kmer_checksum: left out). WRITE_SIZE is taken as reported (calibration: profiles/r02_scatter_store_calibration.txt)."""
import json, sqlite3, sys


def per_kernel(dbfile, counter):
    db = sqlite3.connect(dbfile)
    out = {}
    for k, v, n in db.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name", (counter,)):
        out[k.split("(")[0].replace("void ", "").strip()] = (v, n)
    return out


GATHER = ("k_gather_counts", "k_root_write", "k_root_count", "k_bloom_contains", "k_gather_u64")      # reads dominated by per-lane gathers: FETCH_SIZE at face value
UNCOUNTED = ("k_kmer_checksum",)                                                                     # byte-per-lane loads: FETCH_SIZE sees nothing of them


def read_factor(name):
    return 1 if name.startswith(GATHER) else 2


STEPS = int(sys.argv[5]) if len(sys.argv) > 5 else 1
f = per_kernel(sys.argv[1], "FETCH_SIZE"); w = per_kernel(sys.argv[2], "WRITE_SIZE")
for k_ in list(f):
    if k_.startswith(UNCOUNTED):
        f.pop(k_); w.pop(k_, None)
res = {"workload": sys.argv[3], "note": "read_bytes = 2*FETCH_SIZE*1024 for coalesced readers, 1*FETCH_SIZE*1024 for the gather kernels (profiles/r05_fetch_size_calibration.txt), write_bytes = WRITE_SIZE*1024; per launch = sum / dispatches; collected with GKC_STAGEB_LANES=1 (with two lanes the device-wide counters of a dispatch include the other lane's kernels)",
       "kernels": {}}
for k in sorted(set(f) | set(w)):
    fv, fn = f.get(k, (0, 1)); wv, wn = w.get(k, (0, 1))
    rf = read_factor(k)
    res["kernels"][k] = {"dispatches": int(max(fn, wn)), "read_factor": rf, "read_bytes_total": rf * fv * 1024, "write_bytes_total": wv * 1024,
                         "hbm_bytes_per_launch": (rf * fv * 1024 + wv * 1024) / max(fn, wn),
                         "hbm_bytes_per_step": (rf * fv * 1024 + wv * 1024) / STEPS}
json.dump(res, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e9, 2) for k, v in res["kernels"].items()}, indent=0))
