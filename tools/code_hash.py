#!/usr/bin/env python3
"""sha256 of the .text section of every gfx950 code object in a library (default clsim_amd/libclsimhip.so): two builds with equal
hashes run the same machine code.  Used to check that a source clean-up (moving experiment switches out of the kernels) changed
no instruction of the shipped kernels.  BUILD CONTAINER TOOL.
   tools/code_hash.py [LIB]              one line per code object
   tools/code_hash.py [LIB] --kernels    one line per kernel symbol: sha256 of its instructions as the disassembler prints them, addresses
                                         branch targets and pc-relative literals taken out (a kernel keeps its hash when only its neighbours in the code
                                         object change)"""
import hashlib, os, re, shutil, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
per_kernel = "--kernels" in sys.argv
args = [a for a in sys.argv[1:] if a != "--kernels"]
lib = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "clsim_amd", "libclsimhip.so")
with tempfile.TemporaryDirectory() as d:
    copy = shutil.copy(lib, d)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], check=True, capture_output=True)
    for f in sorted(os.listdir(d)):
        if not f.endswith("gfx950"):
            continue
        if per_kernel:
            out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", os.path.join(d, f)],
                                 check=True, capture_output=True, text=True).stdout
            name, body = None, []
            def flush():
                if name is not None:
                    print(hashlib.sha256("\n".join(body).encode()).hexdigest()[:16], "%6d" % len(body), name)
            for line in out.split("\n"):
                m = re.match(r"^[0-9a-f]* ?<([^>]+)>:$", line)
                if m:
                    flush()
                    name, body = m.group(1), []
                elif name is not None and line.strip():
                    # (branch targets are printed as absolute addresses / symbol+offset: keep the mnemonic only)
                    t = line.split("//")[0].strip()
                    t = re.sub(r"(s_c?branch\S*|s_call\S*)\s.*", r"\1", t)
                    # (and the pc-relative literal behind s_getpc_b64: the distance to the code object's constant data)
                    if body and body[-1].startswith("s_getpc_b64") and t.startswith("s_add_u32"):
                        t = re.sub(r"0x[0-9a-f]+$", "PCREL", t)
                    body.append(t)
            flush()
            continue
        text = os.path.join(d, f + ".text")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.text", os.path.join(d, f), text], check=True)
        data = open(text, "rb").read()
        print(hashlib.sha256(data).hexdigest()[:16], len(data), f.split(".so.")[-1])
