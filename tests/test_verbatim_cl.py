"""Hit records of the reference's OpenCL kernel text itself.

tests/golden/verbatim_cl_<config>.npz were written by tools/verbatim_cl_check.py in the build container: the files
resources/kernels/{mwcrng_kernel, propagation_kernel.h, sparse_collision_kernel.h, sparse_collision_kernel.c,
propagation_kernel.c}.cl of the reference compiled VERBATIM for x86-64 (ROCm clang, OpenCL C 1.2) behind a generated section
emitted from oracle/builders.py and OpenCL builtins on oracle/oracle_math.h, run work item by work item over the seeded step
bunches of tests/common.py.  They hold outputs only (sorted 80-byte hit records, final RNG states); the inputs are
regenerated here from the same seeds.

What they pin: the transcription of the two static kernel files by oracle/clsim_oracle.c (CPU test below) and by the HIP
kernels (GPU test) -- a guard against one author misreading 1 500 lines of OpenCL C twice in the same way.  They do not pin
the generated section nor the math library, which are this repository's on every side."""
import os
import subprocess
import sys

import numpy as np
import pytest

from clsim_amd.synthetic import PHOTON_DTYPE
from oracle import capi
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = ["c1", "mie", "lea", "flasher"]


def fixture(name):
    f = np.load(os.path.join(ROOT, "tests", "golden", "verbatim_cl_%s.npz" % name))
    cfg = common.config(name)
    steps = common.steps_for(cfg, int(f["n_steps"]), seed=int(f["seed"]))
    x, a = common.streams(len(steps))
    hits = np.frombuffer(f["hits"].tobytes(), dtype=PHOTON_DTYPE)
    return cfg, steps, x, a, hits, f["rng_x"]


@pytest.mark.parametrize("name", CONFIGS)
def test_oracle_equals_the_verbatim_kernel(name):
    cfg, steps, x, a, hits, rng_x = fixture(name)
    T = common.oracle_tables(cfg)
    ph_o, cnt_o, x_o, _ = capi.propagate(T, steps, x, a, threads=8)
    assert cnt_o == len(hits) and cnt_o > 100
    # the kernel emits DOM / string indices; the fixture holds what the kernel wrote
    assert common.sort_photons(ph_o).tobytes() == common.sort_photons(hits).tobytes()
    assert np.array_equal(x_o, rng_x)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CONFIGS)
def test_hip_path_equals_the_verbatim_kernel(name):
    cfg, steps, x, a, hits, rng_x = fixture(name)
    T = common.oracle_tables(cfg)
    conv = common.product_converter(cfg, len(steps))
    conv.EnqueueSteps(steps, 7)
    ident, ph_p = conv.GetConversionResult()
    assert ident == 7 and len(ph_p) == len(hits)
    # the converter hands out string / DOM IDs (OpenCL.cxx:1565-1619): the same translation applied to the kernel's indices
    expect = capi.replace_indices_with_ids(hits.copy(), T.geo)
    assert common.sort_photons(ph_p).tobytes() == common.sort_photons(expect).tobytes()
    assert np.array_equal(conv.GetRNGState(len(steps)), rng_x)


@pytest.mark.skipif(not os.path.isdir("/root/reference/resources/kernels"), reason="the reference tree is not on this machine")
def test_fixture_is_what_the_reference_kernel_text_yields_today():
    """build container only: recompile the reference's .cl files and run the smallest configuration again (the tool exits
    non-zero unless the verbatim kernel and the oracle agree bit for bit); `tools/verbatim_cl_check.py` runs all four"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "verbatim_cl_check.py"), "--configs", "c1"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "hit records IDENTICAL | final RNG states IDENTICAL" in p.stdout
