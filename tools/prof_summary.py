"""Turn a rocprofv3 rocpd database (kernel trace) into the per-kernel summary kept under profiles/."""
import glob, sqlite3, sys
src, dst = sys.argv[1], sys.argv[2]
dbs = glob.glob(src + "/**/*.db", recursive=True)
assert dbs, "no rocpd database under " + src
rows = []
for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows += list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(dst, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
    f.write("%-12s %8s %14s %12s %7s  %s\n" % ("", "calls", "total_us", "avg_us", "pct", "kernel"))
    for name, calls, tot, avg, pct in sorted(rows, key=lambda r: -r[2]):
        f.write("%-12s %8d %14.1f %12.1f %6.2f%%  %s\n" % ("", calls, tot, avg, pct, name))
print(open(dst).read())
