#!/usr/bin/env python3
"""Per-query GPU timeline (kernel durations and the gaps between them) from a
rocprofv3 --kernel-trace rocpd database."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if "scan_kernel" in r[0]]
if len(idx) < 4:
    sys.exit("not enough scan launches")
i0, i1 = idx[-3], idx[-1]
t0 = rows[i0][1]
prev = None
for name, s, e in rows[i0:i1 + 1]:
    m = re.search(r"(\w+_kernel)", name)
    short = (m.group(1) if m else name)[:28]
    print("%-28s start %9.1f us  dur %8.1f us  gap_before %6.1f us" % (short, (s - t0) / 1e3, (e - s) / 1e3,
                                                                    (s - prev) / 1e3 if prev else 0.0))
    prev = e
