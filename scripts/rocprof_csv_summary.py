#!/usr/bin/env python3
"""Summaries of rocprofv3 --output-format csv runs.
    rocprof_csv_summary.py stats <dir>     per-kernel calls / total / average from *kernel_stats.csv (or the trace)
    rocprof_csv_summary.py pmc <dir>       per-kernel average of every counter from *counter_collection.csv
    rocprof_csv_summary.py timeline <dir> <kernel-substring> [n]   start / duration / gap of the launches around the last n matches
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return hits


def short(name):
    name = re.sub(r"gsim::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def stats(d):
    files = find(d, "*kernel_stats.csv")
    if files:
        rows = list(csv.DictReader(open(files[0])))
        print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for r in rows:
            print("%-70s %8d %14.2f %12.2f %7.2f" % (short(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                                                    float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        return
    files = find(d, "*kernel_trace.csv")
    acc = defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in acc.values())
    print("%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print("%-70s %8d %14.2f %12.2f %7.2f" % (k, len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot))


def pmc(d):
    files = find(d, "*counter_collection.csv")
    acc = defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("%-70s %-22s %8s %18s %18s %18s" % ("kernel", "counter", "launches", "avg", "min", "max"))
    for (k, c), v in sorted(acc.items()):
        print("%-70s %-22s %8d %18.1f %18.1f %18.1f" % (k, c, len(v), sum(v) / len(v), min(v), max(v)))


def timeline(d, sub, n=3):
    files = find(d, "*kernel_trace.csv")
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", r.get("Stream_Id", "")))
                   for r in csv.DictReader(open(files[0]))))
    idx = [i for i, r in enumerate(rows) if sub in r[2]]
    i0, i1 = idx[-n], idx[-1]
    t0, prev = rows[i0][0], None
    for s, e, name, q in rows[i0:i1 + 1]:
        # (gap_before < 0: the launch started while the previous one -- on another queue -- was still running)
        print("%-50s queue %-4s start %9.1f us  end %9.1f us  dur %8.1f us  gap_before %6.1f us" % (name[:50], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
        prev = e


if __name__ == "__main__":
    mode, d = sys.argv[1], sys.argv[2]
    if mode == "stats":
        stats(d)
    elif mode == "pmc":
        pmc(d)
    else:
        timeline(d, sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 3)
