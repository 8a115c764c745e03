"""Turns a rocprofv3 rocpd sqlite database into the text summaries committed under profiles/ (kernel stats, PMC per kernel)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
print("# rocprofv3 kernel stats (durations in us)  source:", sys.argv[1])
print("%-70s %8s %16s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-70s %8d %16.0f %14.0f %7.2f" % (name.split("(")[0][:70], calls, tot, avg, pct))
try:
    rows = list(db.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"))
    if rows:
        print("\n# PMC counters summed over dispatches")
        print("%-60s %-24s %20s %8s" % ("kernel", "counter", "sum", "dispatches"))
        for k, cn, v, n in rows:
            print("%-60s %-24s %20.0f %8d" % (k.split("(")[0][:60], cn, v, n))
except Exception as e:
    print("# no PMC data:", e)

# optional: --seq N  prints the last N dispatches in launch order (name, duration, grid, registers): where a step's time goes launch by launch
if "--seq" in sys.argv:
    n = int(sys.argv[sys.argv.index("--seq") + 1])
    rows = list(db.execute("select name, duration, grid_x, grid_y, workgroup_x, vgpr_count, accum_vgpr_count, scratch_size, lds_size from kernels order by start desc limit %d" % n))[::-1]
    print("\n# last %d dispatches in launch order: kernel, us, grid (threads), workgroup, vgpr+agpr, scratch, lds" % n)
    for nm, dur, gx, gy, wx, vg, ag, sc, lds in rows:
        print("%-50s %9.1f  %9d x %-4d %5d  %3d+%-3d %5d %6d" % (nm.split("(")[0][:50], dur / 1000.0, gx, gy, wx, vg, ag, sc, lds))
