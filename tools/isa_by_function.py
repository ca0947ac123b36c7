#!/usr/bin/env python3
"""Static instruction counts of one kernel, by the source function each instruction was inlined from.

  hipcc ... --cuda-device-only -gline-tables-only -S -o pool_g.s prop_pool_kernel.hip
  tools/isa_by_function.py pool_g.s '_ZN8clsimhip16prop_pool_kernelILi1ELb1ELb0ELb0ELb1EEEvNS_7KParamsE'

Columns: vector ALU, scalar ALU (incl. exec-mask work), branches, s_waitcnt, s_nop, scalar loads, LDS, vector memory.
Multiply by how often a region runs per loop trip (census) for the dynamic picture."""
import collections
import os
import re
import sys

asm, symbol = sys.argv[1], sys.argv[2]
src_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "clsim_amd", "csrc")
files = {}
rows = collections.defaultdict(lambda: collections.Counter())
inside = False
cur = ("?", 0)


def functions_of(path):
    """(first line, name) of every function-like definition, by a crude scan."""
    out = []
    pat = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:DM|DEV|static|inline|__device__|__global__|__host__|constexpr|\s)*[\w:<>\*&\s]+?\b(\w+)\s*\([^;]*$")
    for no, line in enumerate(open(path, errors="replace"), 1):
        if line[:1] in " \t#/}" or "(" not in line:
            continue
        m = pat.match(line.rstrip())
        if m:
            out.append((no, m.group(1)))
    return out


fn_cache = {}


def where(file_name, line):
    base = os.path.basename(file_name)
    path = os.path.join(src_dir, base)
    if not os.path.exists(path):
        return base
    if base not in fn_cache:
        fn_cache[base] = functions_of(path)
    name = "?"
    for no, fn in fn_cache[base]:
        if no <= line:
            name = fn
        else:
            break
    return "%s:%s" % (base.split(".")[0], name)


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


for line in open(asm, errors="replace"):
    s = line.strip()
    if s.startswith(".file"):
        m = re.match(r'\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', s)
        if m:
            files[int(m.group(1))] = m.group(2)
        continue
    if not inside:
        inside = s.startswith(symbol + ":")
        continue
    if s.startswith(".loc"):
        p = s.split()
        cur = (files.get(int(p[1]), "?"), int(p[2]))
        continue
    if s.startswith("s_endpgm"):
        break
    op = s.split()[0] if s else ""
    k = kind(op)
    if k:
        detail = "--lines" in sys.argv
        key = where(*cur) + ((":%d" % cur[1]) if detail else "")
        rows[key][k] += 1

cols = ["valu", "salu", "branch", "wait", "nop", "smem", "lds", "vmem"]
print("%-44s" % "function" + "".join("%8s" % c for c in cols))
tot = collections.Counter()
for key, c in sorted(rows.items(), key=lambda kv: -sum(kv[1].values())):
    print("%-44s" % key + "".join("%8d" % c[k] for k in cols))
    tot.update(c)
print("%-44s" % "total" + "".join("%8d" % tot[k] for k in cols))
