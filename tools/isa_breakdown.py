#!/usr/bin/env python3
"""Attributes the static instructions of one kernel variant to source functions
(via .loc line info).  ANALYSIS TOOL.  usage: isa_breakdown.py file.s mangled-substring src.hip math.h"""
import collections
import re
import sys


def func_map(lines):
    out, cur = {}, "?"
    for i, l in enumerate(lines, 1):
        if l.startswith("DM ") or l.startswith("__global__") or (l.startswith("template") is False and re.match(r"^(static )?hipError_t|^size_t|^int ", l)):
            m = re.search(r"(\w+)\s*\(", re.sub(r"__launch_bounds__\([^)]*\)", "", l))
            if m:
                cur = m.group(1)
        out[i] = cur
    return out


def main():
    asm, key, src, math = sys.argv[1:5]
    txt = open(asm).read()
    m = re.search(r"(_ZN8clsimhip11prop_kernelI%s[^:\n]*):(.*?)\.Lfunc_end" % key, txt, re.S)
    body = m.group(2).split("\n")
    files = {int(a): (c or b) for a, b, c in re.findall(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt)}
    fm = {src.split("/")[-1]: func_map(open(src).read().split("\n")), math.split("/")[-1]: func_map(open(math).read().split("\n"))}
    cur = ("?", 0)
    cnt, vcnt = collections.Counter(), collections.Counter()
    for l in body:
        s = l.strip()
        mm = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if mm:
            cur = (files.get(int(mm.group(1)), "?").split("/")[-1], int(mm.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.split()[0].endswith(":"):
            continue
        fn = fm.get(cur[0], {}).get(cur[1], cur[0])
        cnt[fn] += 1
        if s.startswith("v_"):
            vcnt[fn] += 1
    print("total", sum(cnt.values()), "valu", sum(vcnt.values()))
    for k, v in cnt.most_common(45):
        print("%-28s all=%5d valu=%5d" % (k, v, vcnt[k]))


if __name__ == "__main__":
    main()
