#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run (rocpd .db)."""
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
print("%-80s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in cur:
    print("%-80s %8d %14.2f %12.2f %7.2f" % (name[:80], calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
