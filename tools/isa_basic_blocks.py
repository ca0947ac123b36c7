#!/usr/bin/env python3
"""Basic blocks of one kernel with their instruction counts and the source lines most of their instructions come from.
ANALYSIS TOOL (no GPU needed).

  cd clsim_amd/csrc && hipcc <the Makefile's flags for the file> --cuda-device-only -gline-tables-only -S -o /tmp/pool_g.s prop_pool_kernel.hip
  tools/isa_basic_blocks.py /tmp/pool_g.s '_ZN8clsimhip16prop_pool_kernelILi1ELb1ELb0ELb0ELb1EEEvNS_7KParamsE'

Columns: label, vector ALU, scalar (incl. branches), other (LDS / memory / waitcnt), branch targets, the three source lines
with most instructions in the block.  Multiply by how often a block runs per loop trip (tools/exp_pool_census.py) for the
dynamic picture (divergence budget)."""
import collections
import re
import sys

asm = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2]
start = [i for i, l in enumerate(asm) if l.startswith(sym + ":")][0]
end = [i for i, l in enumerate(asm) if i > start and l.startswith(".Lfunc_end")][0]
files = {}
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
blocks = []
cur = None
loc = ("?", 0)
for l in asm[start:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = {"label": m.group(1), "valu": 0, "salu": 0, "other": 0, "lines": collections.Counter(), "branch": []}
        blocks.append(cur)
        continue
    if cur is None or not s or s.startswith((".", ";", "//")):
        continue
    op = s.split()[0]
    if op.startswith("v_"):
        cur["valu"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch":
        cur["salu"] += 1
        cur["branch"].append(s.split()[-1])
    elif op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop")):
        cur["salu"] += 1
    else:
        cur["other"] += 1
    cur["lines"][loc] += 1
print("%-11s %5s %5s %5s  %-34s %s" % ("block", "valu", "salu", "other", "branches to", "source lines (instructions)"))
for b in blocks:
    top = ", ".join("%s:%d (%d)" % (f, ln, c) for (f, ln), c in b["lines"].most_common(3))
    print("%-11s %5d %5d %5d  %-34s %s" % (b["label"], b["valu"], b["salu"], b["other"], ",".join(b["branch"])[:34], top))
print("total: %d vector, %d scalar, %d other in %d blocks" % (sum(b["valu"] for b in blocks), sum(b["salu"] for b in blocks), sum(b["other"] for b in blocks), len(blocks)))
