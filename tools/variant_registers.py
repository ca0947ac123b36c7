#!/usr/bin/env python3
"""tools/variant_registers.py LIB.so [substring]: VGPRs, scratch bytes and spilled registers of every propagation kernel in a
build (the same notes tests/test_codegen.py reads), to see what a build variant did to the register allocation."""
import os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
d = tempfile.mkdtemp()
lib = shutil.copy(sys.argv[1], d)
subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True)
want = sys.argv[2] if len(sys.argv) > 2 else "prop_pool_kernel"
rows = []
for f in sorted(os.listdir(d)):
    if not f.endswith("gfx950"):
        continue
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, f)], check=True, capture_output=True, text=True).stdout
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name or want not in name.group(1):
            continue
        g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, block).group(1)) if re.search(r"\.%s:\s+(\d+)" % key, block) else 0
        rows.append((name.group(1), g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
shutil.rmtree(d)
print("%-70s %5s %5s %8s %7s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "vspill", "sspill"))
for r in rows:
    print("%-70s %5d %5d %8d %7d %7d" % r)
print("%d kernels; max vgpr %d, with scratch %d" % (len(rows), max(r[1] for r in rows), sum(1 for r in rows if r[3])))
