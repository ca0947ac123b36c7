"""The C++ adapter class (same virtual interface as the reference's converter)
over the C ABI: compiled with g++ against include/clsimhip.h and linked to
libclsimhip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = os.path.join(ROOT, "clsim_amd", "cxx")


def build(tmp_path):
    exe = str(tmp_path / "adapter_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(CXX, "adapter_test.cxx"),
                           "-L" + os.path.join(ROOT, "clsim_amd"), "-lclsimhip", "-Wl,-rpath," + os.path.join(ROOT, "clsim_amd")])
    return exe


def test_adapter_configures_and_compiles(tmp_path):
    out = subprocess.check_output([build(tmp_path)], text=True)
    assert "adapter ok" in out and "workgroup 256" in out


@pytest.mark.gpu
def test_adapter_propagates_a_bunch(tmp_path):
    out = subprocess.check_output([build(tmp_path), "run"], text=True)
    assert "identifier 42" in out and "generated 200000" in out
    # the in-place view (no I3CLSimPhotonSeries, the records where the library left them), twice: the buffer returns to the pool
    assert "view 0: identifier 43" in out and "view 1: identifier 44" in out
