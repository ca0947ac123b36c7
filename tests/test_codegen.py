"""What the build promises about the propagation kernels' machine code, checked on the shipped library without a GPU:
no packed single-precision arithmetic (the SLP vectoriser's v_pk_*_f32 cost 8-12 % on MI355X), nothing
spilled to scratch memory, and vector-register counts that allow the occupancy the launchers assume (pooled kernel: 6 waves
per SIMD = 80 registers; classic kernel: 7 = 72)."""
import os
import re
import shutil
import subprocess

import pytest

from clsim_amd import _lib

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    d = tmp_path_factory.mktemp("codegen")
    lib = shutil.copy(_lib.LIB_PATH, d)                      # the tool writes the bundles next to its input
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", lib], check=True, capture_output=True)
    objs = [os.path.join(d, f) for f in sorted(os.listdir(d)) if f.endswith("gfx950")]
    assert objs, "the library holds no gfx950 code object"
    return objs


def kernel_metadata(obj):
    """{kernel symbol: {'vgpr': n, 'scratch': bytes, 'vgpr_spills': n}} from the code object's notes."""
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", obj], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
        scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
        spills = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
        if name and vgpr and scratch:
            out[name.group(1)] = {"vgpr": int(vgpr.group(1)), "scratch": int(scratch.group(1)), "vgpr_spills": int(spills.group(1)) if spills else 0}
    return out


def test_propagation_kernels_keep_their_register_budget_and_spill_nothing(code_objects):
    seen = {"pool": 0, "classic": 0}
    for obj in code_objects:
        for name, k in kernel_metadata(obj).items():
            if "prop_pool_kernel" in name:
                seen["pool"] += 1
                assert k["vgpr"] <= 80 and k["scratch"] == 0 and k["vgpr_spills"] == 0, (name, k)
            elif "prop_kernel" in name:
                seen["classic"] += 1
                # TAB != 0 (template argument 5: the table maker, and 3 = without STOP_PHOTONS_ON_DETECTION) is built for 4 waves per SIMD: 128 registers
                tab = re.search(r"prop_kernelILi\d+ELb[01]ELb[01]ELb[01]ELi([0123])E", name)
                limit = 128 if (tab and tab.group(1) != "0") else 72
                assert k["vgpr"] <= limit and k["scratch"] == 0 and k["vgpr_spills"] == 0, (name, k)
    assert seen["pool"] == 96 and seen["classic"] >= 48, seen     # pooled: 48 with and 48 without STOP_PHOTONS_ON_DETECTION


def test_no_packed_single_precision_arithmetic_in_the_propagation_kernels(code_objects):
    packed = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")
    checked = 0
    for obj in code_objects:
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", obj], check=True, capture_output=True, text=True).stdout
        if "prop_pool_kernel" not in syms and "prop_kernelILi" not in syms:
            continue
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", obj], check=True, capture_output=True, text=True).stdout
        current = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                current = m.group(1)
                continue
            if current and ("prop_pool_kernel" in current or "prop_kernelILi" in current):
                # the table maker (TAB != 0) keeps whatever the compiler likes: its limit is the fp64 atomics
                tab = re.search(r"prop_kernelILi\d+ELb[01]ELb[01]ELb[01]ELi([012])E", current)
                if tab and tab.group(1) != "0":
                    continue
                assert not packed.search(line), (current, line.strip())
                checked += 1
    assert checked > 100000           # the kernels were really walked through
