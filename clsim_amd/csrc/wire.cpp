// Wire format of the step and photon series: what travels between clsim's client modules and
// I3CLSimServer are portable-binary-archive images of I3Vector<I3CLSimStep> / I3Vector<I3CLSimPhoton>
// (private/clsim/I3CLSimServer.cxx:46-74, 320, 339, 386, 408).  The reference's own code defines their payload
//   private/clsim/I3CLSimStep.cxx:111-147     I3Vector<I3CLSimStep>::serialize(portable_binary_[io]archive)
//   private/clsim/I3CLSimPhoton.cxx:141-168   I3Vector<I3CLSimPhoton>::serialize(...)
// as  [I3FrameObject base record] [class version: unsigned] [num: uint64] [num records as ONE little-endian blob of
// 48 / 80 bytes each] -- the records are the structs of include/clsimhip.h, so a series is read and written without
// touching its records.
//
// Built here: that payload from the class version on.  The archive's own framing around it -- stream header, the
// class-id / tracking / object-id records of the shared_ptr and of the I3FrameObject base -- is written by
// icecube::serialization / icecube::archive (icetray projects `serialization` and `archive`, absent from the reference
// tree, version unpinned); the integers below use that archive's published encoding (boost's portable_binary_archive:
// one byte holding the number of significant bytes, then those bytes, least significant first; 0 is the single byte 0).
// PARITY UNPINNED: no reference file or test holds a byte image of such a message.
#include <cstring>

#include "host_model.h"

namespace clsimhip {

size_t portable_uint_encode(uint64_t v, uint8_t out[9])
{
    size_t size = 0;
    for (uint64_t t = v; t != 0; t >>= 8) ++size;
    out[0] = static_cast<uint8_t>(size);
    for (size_t i = 0; i < size; ++i) out[1 + i] = static_cast<uint8_t>(v >> (8 * i));
    return 1 + size;
}

// returns the bytes consumed, 0 when the input is truncated or not an unsigned value of at most 8 bytes
size_t portable_uint_decode(const uint8_t *in, size_t bytes, uint64_t *v)
{
    if (bytes < 1) return 0;
    const int8_t size = static_cast<int8_t>(in[0]);
    if (size < 0 || size > 8 || static_cast<size_t>(size) + 1 > bytes) return 0;
    uint64_t r = 0;
    for (int i = 0; i < size; ++i) r |= static_cast<uint64_t>(in[1 + i]) << (8 * i);
    *v = r;
    return 1 + static_cast<size_t>(size);
}

size_t series_blob_size(size_t n, size_t record)
{
    uint8_t tmp[9];
    return 1 + portable_uint_encode(n, tmp) + n * record;      // class version 0 is one byte
}

void series_encode(const void *records, size_t n, size_t record, unsigned version, uint8_t *out, size_t cap, size_t *written)
{
    if (!out || !written) throw Error(CLSIMHIP_ERR_ARGUMENT, "output pointers are (null)");
    if (n && !records) throw Error(CLSIMHIP_ERR_ARGUMENT, "records are (null)");
    uint8_t head[18];
    size_t h = portable_uint_encode(version, head);
    h += portable_uint_encode(n, head + h);
    if (h + n * record > cap) throw Error(CLSIMHIP_ERR_ARGUMENT, "output buffer too small for the series");
    std::memcpy(out, head, h);
    if (n) std::memcpy(out + h, records, n * record);
    *written = h + n * record;
}

void series_decode(const uint8_t *in, size_t bytes, size_t record, unsigned version, const char *class_name, void *out, size_t cap,
                   size_t *n_out, size_t *consumed)
{
    if (!in || !n_out) throw Error(CLSIMHIP_ERR_ARGUMENT, "pointers are (null)");
    uint64_t v = 0, n = 0;
    const size_t a = portable_uint_decode(in, bytes, &v);
    if (a == 0) throw Error(CLSIMHIP_ERR_IO, "truncated or malformed series: class version");
    // I3CLSimStep.cxx:119-120 / I3CLSimPhoton.cxx:148-149
    if (v != version)
        throw Error(CLSIMHIP_ERR_IO, "This reader can only read I3Vector<" + std::string(class_name) + "> version " + std::to_string(version) + ", but " +
                                         std::to_string(v) + " was provided.");
    const size_t b = portable_uint_decode(in + a, bytes - a, &n);
    if (b == 0) throw Error(CLSIMHIP_ERR_IO, "truncated or malformed series: number of records");
    if (n > (bytes - a - b) / record) throw Error(CLSIMHIP_ERR_IO, "truncated series: fewer bytes than the announced records");
    *n_out = static_cast<size_t>(n);
    if (consumed) *consumed = a + b + static_cast<size_t>(n) * record;
    if (out) {
        if (n > cap) throw Error(CLSIMHIP_ERR_ARGUMENT, "output buffer too small for the series");
        if (n) std::memcpy(out, in + a + b, static_cast<size_t>(n) * record);
    }
}

} // namespace clsimhip
