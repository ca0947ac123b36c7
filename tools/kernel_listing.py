#!/usr/bin/env python3
"""Compact listing of one kernel from an annotated assembly file (hipcc ... --cuda-device-only -gline-tables-only -S): one line per
instruction with the source line it comes from.  BUILD CONTAINER TOOL.   tools/kernel_listing.py pool_g.s [SYMBOL] > kernel.s"""
import re, sys
asm = open(sys.argv[1]).read().split("\n")
sym = sys.argv[2] if len(sys.argv) > 2 else "_ZN8clsimhip16prop_pool_kernelILi1ELb1ELb0ELb0ELb1ELb0EEEvNS_7KParamsE"
start = [i for i, l in enumerate(asm) if l.startswith(sym + ":")][0]
end = [i for i, l in enumerate(asm) if i > start and l.startswith(".Lfunc_end")][0]
files = {}
for l in asm:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
loc = ""
for l in asm[start:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        loc = "%s:%s" % (files.get(int(m.group(1)), "?").replace("prop_", "").replace(".hip", "").replace(".h", ""), m.group(2))
        continue
    if (s.startswith((".", ";", "//")) and not re.match(r"^\.LBB", l)) or not s:
        continue
    print("%-90s ; %s" % (l.rstrip()[:90], loc))
