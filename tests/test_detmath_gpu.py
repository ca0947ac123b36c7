"""GPU: the device math library (clsim_amd/csrc/detmath.hip.h) against the
oracle's math spec (oracle/oracle_math.h), BIT FOR BIT, on dense samples of the
ranges the kernel uses and on wider ones.  This is what makes bit-exact hit
parity between an x86 build and a gfx950 build possible at all."""
import ctypes as C

import numpy as np
import pytest

from clsim_amd import _lib
from oracle import capi

pytestmark = pytest.mark.gpu


def device_eval(what, x, y=None):
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    yp = None
    if y is not None:
        y = np.ascontiguousarray(y, dtype=np.float32)
        yp = y.ctypes.data_as(C.c_void_p)
    rc = lib.clsimhip_eval_math(0, what, x.ctypes.data_as(C.c_void_p), yp, len(x), out.ctypes.data_as(C.c_void_p))
    assert rc == 0, lib.clsimhip_last_error(None)
    return out


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


N = 1 << 21


@pytest.mark.parametrize("what,lo,hi", [
    (0, 5.9604645e-8, 1.0), (0, 1e-30, 1e30), (1, -30.0, 0.0), (1, -90.0, 90.0), (2, 0.0, 6.2831855), (3, 0.0, 6.2831855),
    (2, -1000.0, 1000.0), (3, -1000.0, 1000.0), (5, -1.0, 1.0), (7, 1e-8, 4.0), (8, 0.0, 1e6), (10, -1.0, 1.0), (10, -1.0000005, 1.0000005),
    (15, 0.0, 600.0), (15, -7e3, 7e3)])
def test_unary_functions_bit_exact(oracle_lib, what, lo, hi):
    rng = np.random.Generator(np.random.PCG64(1000 + what))
    if lo > 0 and hi / lo > 1e6:
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), N)).astype(np.float32)
    else:
        x = rng.uniform(lo, hi, N).astype(np.float32)
    x[:4] = [lo, hi, np.float32(lo) + np.float32(0), np.nextafter(np.float32(hi), np.float32(lo))]
    assert np.array_equal(bits(device_eval(what, x)), bits(capi.eval_math(what, x)))


@pytest.mark.parametrize("lo,hi,ys", [
    (265.0, 675.0, (-1.084106802940,)), (0.6, 1.7, (-0.898608505726,)), (0.0, 1.0, (0.0526315793, 0.0526315789)),
    (1e-6, 1e6, (-3.0, -0.5, 0.5, 2.5))])
def test_powr_bit_exact(oracle_lib, lo, hi, ys):
    rng = np.random.Generator(np.random.PCG64(77))
    x = rng.uniform(lo, hi, N).astype(np.float32)
    x[0] = lo
    for yv in ys:
        y = np.full(N, yv, dtype=np.float32)
        assert np.array_equal(bits(device_eval(4, x, y)), bits(capi.eval_math(4, x, y)))


def test_powr_unit_bit_exact(oracle_lib):
    """the scattering-angle power u^beta (single-word logarithm form), u a uniform in [0, 1) incl. 0 and the smallest draw"""
    rng = np.random.Generator(np.random.PCG64(78))
    x = rng.uniform(0.0, 1.0, N).astype(np.float32)
    x[:5] = [0.0, 2.3283064e-10, 1.0, 0.99999994, 5.9604645e-8]
    x[5:N // 2] = np.exp(rng.uniform(np.log(2.3283064e-10), 0.0, N // 2 - 5)).astype(np.float32)
    for yv in (0.0526315793, 0.0526315789, 0.09, 0.01):
        y = np.full(N, yv, dtype=np.float32)
        assert np.array_equal(bits(device_eval(14, x, y)), bits(capi.eval_math(14, x, y)))


def test_binary_functions_bit_exact(oracle_lib):
    rng = np.random.Generator(np.random.PCG64(5))
    x = rng.uniform(-5.0, 5.0, N).astype(np.float32)
    y = rng.uniform(-5.0, 5.0, N).astype(np.float32)
    x[:6] = [0.0, 0.0, 1.0, -1.0, 0.0, -0.0]
    y[:6] = [0.0, 1.0, 0.0, 0.0, -1.0, -1.0]
    assert np.array_equal(bits(device_eval(6, x, y)), bits(capi.eval_math(6, x, y)))        # atan2
    d = rng.uniform(1e-3, 1e3, N).astype(np.float32) * np.where(rng.random(N) < 0.5, -1, 1).astype(np.float32)
    assert np.array_equal(bits(device_eval(9, x, d)), bits(capi.eval_math(9, x, d)))        # IEEE divide


def check_exhaustive(what, exp_lo, exp_hi, cap=64):
    lib = _lib.load()
    res = np.zeros(cap, dtype=np.uint32)
    rc = lib.clsimhip_check_math_exhaustive(0, what, exp_lo, exp_hi, res.ctypes.data_as(C.c_void_p), cap)
    assert rc == 0, lib.clsimhip_last_error(None)
    return int(res[0]), res[1:1 + min(int(res[0]), cap - 1)].view(np.float32)


def test_range_restricted_divide_equals_the_ieee_divide():
    """dm::div_near_ (detmath.hip.h; Markstein's scheme on the exact reciprocal) against the IEEE divide on the device: all
    2^23 divisor significands x divisor exponents -50 ... 50 (every fifth and both ends) x both divisor signs x 40
    numerators each over exponents -40 ... 60 and both signs -- 32 pseudo-random, 8 built from the divisor (exact and
    nearly exact quotients, all-ones, powers of two)."""
    total = 0
    for e in sorted(set(list(range(-50, 51, 5)) + [-49, -1, 0, 1, 49])):
        bad, res = check_exhaustive(16, e, e)
        assert bad == 0, "divisor exponent %d: %d mismatches, first (numerator, divisor) %s" % (e, bad, [float.hex(float(v)) for v in res[:4]])
        total += (1 << 23) * 80
    assert total > 1.5e10


def test_range_restricted_divide_through_the_evaluation_entry_point():
    """dm::div_near_ on ordinary operands (numerators of magnitude >= 2^-40, divisors in [2^-50, 2^50]) equals x / y."""
    rng = np.random.Generator(np.random.PCG64(5))
    y = np.exp(rng.uniform(np.log(2.0 ** -50), np.log(2.0 ** 50), 1 << 16)).astype(np.float32)
    x = (rng.standard_normal(1 << 16) * np.exp(rng.uniform(-20, 20, 1 << 16))).astype(np.float32)
    x = np.where(np.abs(x) < 2.0 ** -40, np.float32(1.0), x)
    assert np.array_equal(device_eval(16, x, y).view(np.uint32), (x / y).view(np.uint32))


@pytest.mark.parametrize("what,exp_lo,exp_hi", [(11, -100, 100), (12, -96, 100), (13, -96, 100)])
def test_range_restricted_operations_equal_the_ieee_ones_on_every_input(what, exp_lo, exp_hi):
    """dm::rcp_ / dm::sqrt_near_ / dm::rsqrt_near_ (detmath.hip.h) against the IEEE divide and sqrt
    on the device: ALL 2^23 significands x every exponent of the admitted range (both signs for the reciprocal)."""
    bad, first = check_exhaustive(what, exp_lo, exp_hi)
    assert bad == 0, "%d mismatches, first arguments %s" % (bad, [float.hex(float(v)) for v in first[:8]])


def test_axis_bin_conversion_equals_the_spelled_out_one_on_every_bit_pattern():
    """The table maker's axis_bin_ (prop_kernel.hip: v_cvt_flr_i32_f32 + v_med3_i32) against the generic saturating floor
    conversion and clamp of Axis::GetIndexCode (Axis.cxx:45-60): all 2^32 bit patterns -- NaNs, infinities, denormals,
    both signs -- x five bin counts."""
    bad, first = check_exhaustive(19, 0, 0)
    assert bad == 0, "%d mismatches, first arguments %s" % (bad, [float.hex(float(v)) for v in first[:8]])


def test_square_root_of_zero_stays_zero():
    x = np.array([0.0, 1.0, 4.0, 2.0, 5.9604645e-8], dtype=np.float32)
    assert np.array_equal(bits(device_eval(12, x)), bits(np.sqrt(x)))


def test_reciprocal_of_a_reciprocal_from_its_argument():
    """dm::rcp_of_rcp_(b, x) with b = RN(1/x): one Newton step from the seed x is RN(1/b) -- for all 2^23 significands of x, every
    exponent -100 ... 100 and both signs, against the IEEE divides on the device (round 4: the layer walk's 1 / length)"""
    bad, args = check_exhaustive(17, -100, 100)
    assert bad == 0, "%d mismatches, first arguments %s" % (bad, args[:8])
    # and the harness bites: beyond the admitted range the reciprocals leave the normal numbers
    bad, _ = check_exhaustive(17, 126, 127)
    assert bad > 0


def test_reciprocal_root_next_to_one_in_integer_arithmetic():
    """dm::rsqrt_unit_: RN(1 / RN(sqrt x)) for the 2047 floats within 1023 ulps of one, from the bit pattern alone (the
    renormalisation after a rotation), against the IEEE operations on the device; its range test admits exactly that window"""
    bad, args = check_exhaustive(18, 0, 0)
    assert bad == 0, "%d mismatches, first arguments %s" % (bad, args[:8])
