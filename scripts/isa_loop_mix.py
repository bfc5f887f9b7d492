#!/usr/bin/env python3
"""Instruction mix of the innermost loop that holds MFMAs, per kernel of an ISA listing (hipcc -S --cuda-device-only).
usage: isa_loop_mix.py listing.s [substring of the mangled kernel name]"""
import collections
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
kern = None
blocks = {}  # kernel -> list of (label, [instructions])
for ln in text:
    m = re.match(r"^(_Z\S+):\s*(;.*)?$", ln)
    if m:
        kern = m.group(1)
        blocks[kern] = [("entry", [])]
        continue
    if kern is None:
        continue
    m = re.match(r"^(\.LBB\S+):", ln)
    if m:
        blocks[kern].append((m.group(1), []))
        continue
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    blocks[kern][-1][1].append(s.split()[0])
    if s.startswith("s_endpgm"):
        kern = None
for k, bl in blocks.items():
    if want not in k:
        continue
    best = max(bl, key=lambda b: sum(1 for i in b[1] if i.startswith("v_mfma")))
    n = sum(1 for i in best[1] if i.startswith("v_mfma"))
    if n == 0:
        continue
    c = collections.Counter(best[1])
    valu = sum(v for i, v in c.items() if i.startswith("v_") and not i.startswith("v_mfma"))
    ds = sum(v for i, v in c.items() if i.startswith("ds_"))
    sal = sum(v for i, v in c.items() if i.startswith("s_"))
    print("%s\n  block %s: %d MFMA, %d VALU (%.2f per MFMA), %d DS, %d scalar/wait, %d scratch" % (
        k, best[0], n, valu, valu / n, ds, sal, sum(v for i, v in c.items() if i.startswith("scratch"))))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(14)))
