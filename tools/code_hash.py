#!/usr/bin/env python3
"""sha256 of the .text section of every gfx950 code object in a library (default clsim_amd/libclsimhip.so): two builds with equal
hashes run the same machine code.  Used to check that a source clean-up (moving experiment switches out of the kernels) changed
no instruction of the shipped kernels.  BUILD CONTAINER TOOL."""
import hashlib, os, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clsim_amd", "libclsimhip.so")
with tempfile.TemporaryDirectory() as d:
    copy = shutil.copy(lib, d)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
    for f in sorted(os.listdir(d)):
        if not f.endswith("gfx950"):
            continue
        text = os.path.join(d, f + ".text")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", os.path.join(d, f), text], check=True)
        data = open(text, "rb").read()
        print(hashlib.sha256(data).hexdigest()[:16], len(data), f.split(".so.")[-1])
