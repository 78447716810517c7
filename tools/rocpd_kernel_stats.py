"""Per-kernel totals of a rocprofv3 run kept as a rocpd sqlite file (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`
writes DIR/NAME_results.db on this image): name, calls, total ms, mean us - the table DESIGN quotes for plan builds.
usage: python tools/rocpd_kernel_stats.py <file.db> [calls-divisor]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = db.execute("select name, count(*), sum(end - start) from kernels group by name order by 3 desc").fetchall()
tot = 0.0
for name, calls, ns in rows:
    tot += ns
    print("%-90s calls %5d  per-build %8.3f ms  mean %8.1f us" % (name[:90], calls, ns / 1e6 / div, ns / 1e3 / calls))
print("total per-build %.3f ms" % (tot / 1e6 / div))
