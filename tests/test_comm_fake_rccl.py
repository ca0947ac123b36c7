"""The multi-rank branch of clsimhip_gather_hits (clsim_amd/csrc/comm.cpp: count all-gather, grouped ncclSend / ncclRecv,
overflow decision) executed on ONE GPU: tests/fake_rccl.cpp stands in for librccl.so with ranks = threads (the real
library wants one GPU per rank).  The library caches its RCCL handle per process, so the body runs in a fresh child
started with CLSIMHIP_RCCL_LIBRARY set -- a child process, never a re-exec of this one.

Reference counterpart: none (independent converters behind I3CLSimServer.cxx:77-137); this is north_star's addition."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests")
FAKE = os.path.join(HERE, "libfake_rccl.so")


def build_fake():
    src = os.path.join(HERE, "fake_rccl.cpp")
    if not os.path.exists(FAKE) or os.path.getmtime(FAKE) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", FAKE, src,
                               "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"])
    return FAKE


def test_fake_rccl_exports_what_comm_cpp_binds():
    """CPU: the stand-in offers exactly the entry points comm.cpp looks up with dlsym"""
    import ctypes
    import re
    lib = ctypes.CDLL(build_fake())
    src = open(os.path.join(ROOT, "clsim_amd", "csrc", "comm.cpp")).read()
    wanted = re.findall(r'sym\("(nccl\w+)"\)', src)
    assert len(wanted) == 11 and "ncclCommCount" in wanted and "ncclCommUserRank" in wanted
    for name in wanted:
        assert hasattr(lib, name), name


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gather_hits_multi_rank_through_the_c_abi(world):
    env = dict(os.environ, CLSIMHIP_RCCL_LIBRARY=build_fake())
    p = subprocess.run([sys.executable, os.path.join(HERE, "comm_fake_rccl_child.py"), str(world)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    assert "all ok" in p.stdout
    assert p.stdout.count(" ok (world %d" % world) == 9


@pytest.mark.gpu
def test_a_communicator_that_counts_another_world_is_refused():
    """clsimhip_comm_create asks the communicator itself (ncclCommCount / ncclCommUserRank): one that reports a world other
    than the caller's never becomes a clsimhip_comm, so an N-rank record cannot be printed over it."""
    env = dict(os.environ, CLSIMHIP_RCCL_LIBRARY=build_fake(), FAKE_RCCL_LIE_ABOUT_COUNT="1")
    p = subprocess.run([sys.executable, os.path.join(HERE, "comm_fake_rccl_child.py"), "2", "lie"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    assert "refused ok" in p.stdout
