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
