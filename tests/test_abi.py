"""The C ABI: the shared library loads, exports every symbol include/clsimhip.h
declares, keeps the reference's record layouts, and fails loudly (no fallback)
where a GPU is required.  No compute calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from clsim_amd import _lib
from clsim_amd import converter as CV
from clsim_amd.synthetic import PHOTON_DTYPE, STEP_DTYPE
from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "clsimhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(clsimhip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 40
    assert sorted(_lib.SYMBOLS) == declared          # the binding knows exactly the header's functions
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.clsimhip_version()


def test_record_layouts_are_the_reference_wire_structs():
    # public/clsim/I3CLSimStep.h:141-155 (48 B), I3CLSimPhoton.h:194-213 (80 B)
    assert STEP_DTYPE.itemsize == 48 and PHOTON_DTYPE.itemsize == 80
    assert STEP_DTYPE.fields["num"][1] == 32 and STEP_DTYPE.fields["id"][1] == 40 and STEP_DTYPE.fields["sourceType"][1] == 44
    assert PHOTON_DTYPE.fields["numScatters"][1] == 32 and PHOTON_DTYPE.fields["stringID"][1] == 44
    assert PHOTON_DTYPE.fields["omID"][1] == 46 and PHOTON_DTYPE.fields["sx"][1] == 48 and PHOTON_DTYPE.fields["distInAbsLens"][1] == 76


def test_use_before_initialize_raises_like_the_reference():
    """OpenCL.cxx:1525-1544, 1604-1607: '... is not initialized!'"""
    conv = common.product_converter(common.config("c1"), 512, initialize=False)
    assert not conv.IsInitialized()
    steps = np.zeros(512, dtype=STEP_DTYPE)
    for call in (lambda: conv.EnqueueSteps(steps, 0), conv.GetConversionResult, conv.QueueSize, conv.MorePhotonsAvailable):
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="not initialized"):
            call()


def test_unsupported_modes_are_refused_at_compile():
    cfg = common.config("c1")
    for setter, value in (("SetDoublePrecision", True), ("SetSaveAllPhotons", True), ("SetPhotonHistoryEntries", 5000)):
        conv = common.product_converter(cfg, 512, initialize=False)
        getattr(conv, setter)(value)
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception) as e:
            conv.Compile()
        assert e.value.code == _lib.ERR_CONFIG
    # restated modes compile (they only need the GPU from Initialize on)
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.SetPhotonHistoryEntries(4)
    conv.SetFixedNumberOfAbsorptionLengths(46.0)
    conv.SetStopDetectedPhotons(False)
    conv.Compile()


def test_degenerate_medium_values_are_refused():
    """Zero / infinite / NaN lengths would turn absorption budgets into NaN (an endless photon loop in the reference)."""
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="finite and positive"):
            CV.MakeHomogeneousMediumProperties(absLen=bad)
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="finite and positive"):
            CV.MakeHomogeneousMediumProperties(scaLen=bad)


def test_group_velocity_from_dispersion_needs_a_phase_index_with_a_derivative():
    """CLSIMHIP_REFINDEX_DISPERSION (no group refractive index override, MediumPropertiesSource.cxx:274-300): accepted with the
    RefIndexIceCube phase index, refused with a tabulated one (FromTable has no derivative, I3CLSimFunctionFromTable.h:67); a
    converter compiles with it on the generic kernels (no FAST instantiation: tables.cpp)."""
    import ctypes as C
    cfg = common.config("mie_dispersion")
    assert cfg["med_p"].describe()["group_index_kind"] == _lib.REFINDEX_DISPERSION == 2
    conv = common.product_converter(cfg, 512, initialize=False)
    conv.Compile()
    assert conv.GetTable("fast_variant")[0] == 0.0
    tab = CV.MakeIceCubeMediumPropertiesPhotonics(common.PHOTONICS["photonics_mie"])
    d = _lib.MediumDesc()
    assert _lib.load().clsimhip_medium_describe(tab._h, C.byref(d)) == 0
    d.group_index_kind = _lib.REFINDEX_DISPERSION
    h = C.c_void_p()
    assert _lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)) == _lib.ERR_ARGUMENT
    assert b"derivative" in _lib.load().clsimhip_last_error(None)
    d.group_index_kind = 7
    assert _lib.load().clsimhip_medium_create(C.byref(d), C.byref(h)) == _lib.ERR_ARGUMENT


def test_degenerate_wavelength_generators_are_refused():
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    for gen in (CV.I3CLSimRandomValueConstant(0.0), CV.I3CLSimRandomValueConstant(float("nan")),
                CV.I3CLSimRandomValueInterpolatedDistribution(3e-7, 1e-8, [0.0, 0.0, 0.0]),
                CV.I3CLSimRandomValueInterpolatedDistribution(3e-7, 0.0, [1.0, 2.0]),
                CV.I3CLSimRandomValueWlenCherenkovNoDispersion(7e-7, 3e-7),             # WlenCherenkovNoDispersion.cxx:47-51
                CV.I3CLSimRandomValueWlenCherenkovNoDispersion(float("nan"), 3e-7),
                CV.I3CLSimRandomValueWlenCherenkovNoDispersion(0.0, 3e-7)):
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
            conv.SetWlenGenerators([gen])
    conv.SetWlenGenerators([CV.I3CLSimRandomValueWlenCherenkovNoDispersion(265e-9, 675e-9)])


def test_incomplete_configuration_is_refused():
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="WlenGenerators"):
        conv.Compile()
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
        conv.SetWorkgroupSize(100000)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
        conv.SetMaxNumWorkitems(0)


def test_bad_arguments():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.clsimhip_medium_create_from_ppc(b"/nonexistent/dir", 1948.07, 1, C.byref(h)) == _lib.ERR_IO
    assert b"cannot open" in lib.clsimhip_last_error(None)
    assert lib.clsimhip_create(0, None) == _lib.ERR_ARGUMENT
    assert lib.clsimhip_mwc_multipliers(None, 4) == _lib.ERR_ARGUMENT
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    g = CV.I3CLSimSimpleGeometry([1, 1], [1, 2], [0.0, 0.0], [0.0, 0.0], [0.0, -17.0], ["a", "a"], -1.0)
    conv.SetGeometry(g)          # stored; rejected when compiled (GeometrySource.cxx:735)
    conv.SetWlenGenerators([CV.I3CLSimRandomValueConstant(4e-7)])
    conv.SetWlenBias(CV.I3CLSimFunctionConstant(1.0))
    conv.SetMediumProperties(CV.MakeHomogeneousMediumProperties())
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="OM radius"):
        conv.Compile()


def test_initialize_without_gpu_fails_loudly():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    conv = common.product_converter(common.config("c1"), 512, initialize=False)
    x, a = common.streams(512)
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception, match="no HIP device") as e:
        conv.InitializeWithStreams(x, a)
    assert e.value.code == _lib.ERR_DEVICE
    assert not conv.IsInitialized()


def test_safeprimes_file_formats(tmp_path):
    """mwcrng_init.h:62-103: binary ("safeprimes_base32" + int64 LE) and plain-text multiplier files."""
    import struct
    lib = _lib.load()
    ref = np.load(os.path.join(ROOT, "tests", "golden", "mwc_multipliers.npy"))[:500]
    binary = tmp_path / "safeprimes_base32.bin"
    binary.write_bytes(b"safeprimes_base32" + b"".join(struct.pack("<q", int(v)) for v in ref))
    text = tmp_path / "safeprimes_base32.txt"
    text.write_text("".join("%d %d %d\n" % (v, int(v) * 2 ** 32 - 1, (int(v) * 2 ** 32 - 2) // 2) for v in ref))
    for path in (binary, text):
        out = np.zeros(500, dtype=np.uint32)
        assert lib.clsimhip_mwc_multipliers_from_file(str(path).encode(), out.ctypes.data_as(C.c_void_p), 500) == 0
        assert np.array_equal(out, ref)
    out = np.zeros(501, dtype=np.uint32)
    assert lib.clsimhip_mwc_multipliers_from_file(str(binary).encode(), out.ctypes.data_as(C.c_void_p), 501) == _lib.ERR_IO
    assert lib.clsimhip_mwc_multipliers_from_file(b"/nonexistent", out.ctypes.data_as(C.c_void_p), 1) == _lib.ERR_IO


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/clsimhip.h compiles as C99 with -pedantic and the record sizes hold."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "clsimhip.h"\n'
                   'int main(void) { return (sizeof(clsimhip_step) == 48 && sizeof(clsimhip_photon) == 80) ? 0 : 1; }\n')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_option_getters_and_the_class_defaults():
    """GetEnableDoubleBuffering ... GetDOMPancakeFactor (OpenCL.h:138-258); a fresh converter holds the defaults of the reference's
    constructor (OpenCL.cxx:83-92), initializeHIP what its caller passed (I3CLSimModuleHelper.cxx:319-369)"""
    import math
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    assert conv.GetEnableDoubleBuffering() is False and conv.GetDoublePrecision() is False
    assert conv.GetStopDetectedPhotons() is False and conv.GetSaveAllPhotons() is False
    assert math.isnan(conv.GetFixedNumberOfAbsorptionLengths()) and conv.GetDOMPancakeFactor() == 1.0
    assert conv.GetPhotonHistoryEntries() == 0
    assert conv.GetSaveAllPhotonsPrescale() == 0.001                 # OpenCL.cxx:88
    conv.SetEnableDoubleBuffering(True); conv.SetStopDetectedPhotons(True); conv.SetFixedNumberOfAbsorptionLengths(46.0)
    conv.SetDOMPancakeFactor(5.0); conv.SetPhotonHistoryEntries(3); conv.SetSaveAllPhotonsPrescale(0.25)
    assert conv.GetEnableDoubleBuffering() is True and conv.GetStopDetectedPhotons() is True
    assert conv.GetFixedNumberOfAbsorptionLengths() == 46.0 and conv.GetDOMPancakeFactor() == 5.0
    assert conv.GetPhotonHistoryEntries() == 3 and conv.GetSaveAllPhotonsPrescale() == 0.25
    assert conv.GetNumKernelCalls() == 0 and conv.GetTotalNumPhotonsGenerated() == 0 and conv.GetTotalNumPhotonsAtDOMs() == 0


def test_tuning_goes_through_the_c_abi_and_not_through_the_environment():
    """launcher parameters are set with clsimhip_set_tuning (keys and ranges: include/clsimhip.h); the default
    build of the library reads NO tuning from the environment -- the only variable names in the shared object are the two the
    header documents -- and the sources call getenv outside `#ifdef CLSIMHIP_DEVELOPER` three times at most."""
    import re
    import subprocess
    names = set(re.findall(rb"CLSIMHIP_[A-Z0-9_]{3,}", open(os.path.join(common.ROOT, "clsim_amd", "libclsimhip.so"), "rb").read()))
    env_like = {n.decode() for n in names if not n.startswith((b"CLSIMHIP_ERR", b"CLSIMHIP_LENGTHS", b"CLSIMHIP_FUNCTION", b"CLSIMHIP_AX"))}
    assert env_like <= {"CLSIMHIP_SAFEPRIMES_FILE", "CLSIMHIP_RCCL_LIBRARY"}, env_like
    src = os.path.join(common.ROOT, "clsim_amd", "csrc")
    calls = 0
    for name in sorted(os.listdir(src)):
        if not name.endswith((".cpp", ".hip", ".h")):
            continue
        text = subprocess.run(["g++", "-fpreprocessed", "-dD", "-E", "-P", "-x", "c++", os.path.join(src, name)], capture_output=True, text=True).stdout
        depth_dev, stack = 0, []
        for line in text.splitlines():
            t = line.strip()
            if t.startswith(("#if", "#ifdef", "#ifndef")):
                stack.append("CLSIMHIP_DEVELOPER" in t and t.startswith("#ifdef"))
            elif t.startswith("#else") and stack:
                stack[-1] = False
            elif t.startswith("#endif") and stack:
                stack.pop()
            elif "getenv(" in t and not any(stack):
                calls += 1
    assert 1 <= calls <= 3, calls
    # the API itself (host side: no GPU needed before Initialize)
    conv = CV.I3CLSimStepToPhotonConverterHIP(0)
    assert conv.GetTuning("kernel") == 0 and conv.GetTuning("k_wait") == -1 and conv.GetTuning("string_map_cells") == 512
    conv.SetTuning("kernel", "pool"); conv.SetTuning("slices", 5); conv.SetTuning("named_search", 0)
    assert conv.GetTuning("kernel") == 1 and conv.GetTuning("slices") == 5 and conv.GetTuning("named_search") == 0
    for key, value in (("no_such_key", 1), ("kernel", 3), ("k_pop", 65), ("string_map_cells", 4), ("slices", -1)):
        with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception) as err:
            conv.SetTuning(key, value)
        assert err.value.code == _lib.ERR_ARGUMENT
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception):
        conv.GetTuning("no_such_key")
    # the three table keys shape Compile()'s tables: afterwards they are refused, the launcher's keys are not
    conv = common.product_converter(common.config("c1"), 512, initialize=False)
    conv.SetTuning("dom_map_cells", 64)
    conv.Compile()
    with pytest.raises(CV.I3CLSimStepToPhotonConverter_exception) as err:
        conv.SetTuning("dom_map_cells", 128)
    assert err.value.code == _lib.ERR_STATE
    conv.SetTuning("k_new", 3)
    assert conv.GetTuning("k_new") == 3
