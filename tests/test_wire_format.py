"""Wire format of the step and photon series (clsimhip_encode/decode_*_series): byte-level tests built
from the format description -- I3Vector<I3CLSimStep>::serialize(portable_binary_oarchive) writes, after the archive's
own records, the class version, the number of records and the records as one little-endian blob
(/root/reference/private/clsim/I3CLSimStep.cxx:139-147, I3CLSimPhoton.cxx:159-168); the archive encodes an unsigned
integer as one byte holding the number of significant bytes followed by those bytes, least significant first."""
import ctypes as C

import numpy as np
import pytest

from clsim_amd import _lib
from clsim_amd.synthetic import PHOTON_DTYPE, STEP_DTYPE, cascade_steps


def portable_uint(v):
    body = b""
    while v:
        body += bytes([v & 0xff])
        v >>= 8
    return bytes([len(body)]) + body


def encode(kind, records):
    lib = _lib.load()
    n = len(records)
    size = C.c_size_t()
    assert getattr(lib, "clsimhip_%s_series_blob_size" % kind)(n, C.byref(size)) == 0
    out = np.zeros(size.value, dtype=np.uint8)
    written = C.c_size_t()
    assert getattr(lib, "clsimhip_encode_%s_series" % kind)(records.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), out.size,
                                                            C.byref(written)) == 0
    assert written.value == size.value
    return out.tobytes()


def decode(kind, blob, dtype, capacity=None):
    lib = _lib.load()
    buf = np.frombuffer(blob, dtype=np.uint8).copy()
    n, used = C.c_size_t(), C.c_size_t()
    f = getattr(lib, "clsimhip_decode_%s_series" % kind)
    rc = f(buf.ctypes.data_as(C.c_void_p), buf.size, None, 0, C.byref(n), C.byref(used))
    if rc != 0:
        return rc, (lib.clsimhip_last_error(None) or b"").decode()
    out = np.zeros(n.value if capacity is None else capacity, dtype=dtype)
    rc = f(buf.ctypes.data_as(C.c_void_p), buf.size, out.ctypes.data_as(C.c_void_p), len(out), C.byref(n), C.byref(used))
    if rc != 0:
        return rc, (lib.clsimhip_last_error(None) or b"").decode()
    return out[:n.value], used.value


@pytest.mark.parametrize("value,expected", [(0, b"\x00"), (1, b"\x01\x01"), (255, b"\x01\xff"), (256, b"\x02\x00\x01"), (300, b"\x02\x2c\x01"),
                                            (0x01020304, b"\x04\x04\x03\x02\x01"), (2 ** 64 - 1, b"\x08" + b"\xff" * 8)])
def test_portable_unsigned_integer_encoding(value, expected):
    lib = _lib.load()
    out = (C.c_uint8 * 9)()
    n = C.c_size_t()
    assert lib.clsimhip_encode_portable_uint(value, C.cast(out, C.c_void_p), C.byref(n)) == 0
    assert bytes(out[:n.value]) == expected == portable_uint(value)


@pytest.mark.parametrize("kind,dtype,record", [("step", STEP_DTYPE, 48), ("photon", PHOTON_DTYPE, 80)])
@pytest.mark.parametrize("n", [0, 1, 3, 255, 256, 70000])
def test_series_blob_is_version_count_and_raw_records(kind, dtype, record, n):
    rng = np.random.default_rng(n + record)
    records = np.frombuffer(rng.integers(0, 256, n * record, dtype=np.uint8).tobytes(), dtype=dtype).copy()
    blob = encode(kind, records)
    assert blob == b"\x00" + portable_uint(n) + records.tobytes()         # class version 0, num, one blob
    back, used = decode(kind, blob + b"trailing bytes of the next archive item", dtype)
    assert used == len(blob) and back.tobytes() == records.tobytes()


def test_step_fields_sit_where_the_reference_struct_has_them():
    """x, y, z, time, theta, phi, length, beta, num, weight, id, sourceType, dummy1, dummy2 (I3CLSimStep.h:141-148),
    little-endian."""
    steps = cascade_steps(2, seed=4)
    steps["id"] = [0x11223344, 7]
    steps["num"] = [200, 0x01020304]
    blob = encode("step", steps)
    body = blob[2:]
    assert blob[:3] == b"\x00\x01\x02"                                    # class version 0, two records
    first = np.frombuffer(blob[3:3 + 48], dtype="<f4")
    assert first[0] == steps["x"][0] and first[7] == steps["beta"][0]
    assert blob[3 + 32:3 + 36] == (200).to_bytes(4, "little") and blob[3 + 40:3 + 44] == bytes([0x44, 0x33, 0x22, 0x11])
    assert len(body) == 1 + 2 * 48


def test_decode_refuses_other_versions_and_truncated_input():
    steps = cascade_steps(5, seed=1)
    blob = encode("step", steps)
    rc, msg = decode("step", b"\x01\x01" + blob[1:], STEP_DTYPE)              # class version 1
    assert rc != 0 and "can only read I3Vector<I3CLSimStep> version 0, but 1 was provided" in msg
    for cut in (0, 1, 2, len(blob) - 1):
        rc, msg = decode("step", blob[:cut], STEP_DTYPE)
        assert rc != 0 and "truncated" in msg
    rc, msg = decode("step", blob, STEP_DTYPE, capacity=2)
    assert rc != 0 and "too small" in msg
    rc, msg = decode("photon", b"\x00\x09" + b"\x01" * 9, PHOTON_DTYPE)      # a 9-byte integer is not a count
    assert rc != 0
