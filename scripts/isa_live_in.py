#!/usr/bin/env python3
"""VGPRs that are live into the basic block holding the most MFMAs of a kernel (read there before being written), and how
many of them the block only reads -- i.e. what the surrounding code keeps in registers across the contraction.
usage: isa_live_in.py listing.s substring-of-mangled-kernel-name"""
import re
import sys

text = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
kern = None
blocks = []
for ln in text:
    m = re.match(r"^(_Z\S+):\s*(;.*)?$", ln)
    if m:
        kern = m.group(1) if want in m.group(1) else None
        if kern:
            blocks = [[]]
        continue
    if not kern:
        continue
    if re.match(r"^\.LBB\S+:", ln):
        blocks.append([])
        continue
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    blocks[-1].append(s)
    if s.startswith("s_endpgm"):
        break
best = max(blocks, key=lambda b: sum(1 for i in b if i.startswith("v_mfma")))

def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out

written, live_in, readonly = set(), set(), set()
for ins in best:
    ins = ins.split(";")[0]
    parts = ins.split(None, 1)
    if len(parts) < 2:
        continue
    ops = parts[1].split(",")
    op = parts[0]
    stores = op.startswith(("ds_write", "global_store", "scratch_store", "buffer_store", "v_cmp", "global_load_lds", "s_"))
    dst = [] if stores else regs(ops[0])
    src = regs(",".join(ops if stores else ops[1:]))
    if op.startswith("v_mfma"):
        src = regs(",".join(ops[1:]))
    for r in src:
        if r not in written:
            live_in.add(r)
    for r in dst:
        written.add(r)
ro = sorted(r for r in live_in if r not in written)
print("block of %d instructions, %d MFMAs" % (len(best), sum(1 for i in best if i.startswith("v_mfma"))))
print("live-in VGPRs: %d; of those never written in the block: %d" % (len(live_in), len(ro)))
print("read-only:", ro)
print("live-in and later overwritten:", sorted(live_in & written))
print("highest VGPR touched:", max(written | live_in))
