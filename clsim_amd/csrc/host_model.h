// Host-side model of a converter configuration (doubles, as the reference's
// function objects hold them) and of the compiled device tables.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/clsimhip.h"

namespace clsimhip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// The float an OpenCL compiler reads from the literal the reference prints for
// `v`: scientific notation with digits10+4 = 10 digits, correctly rounded
// (private/clsim/I3CLSimHelperToFloatString.h:36-59).  Every constant the
// reference kernel sees went through this text round trip.
inline float to_float_literal(double v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%.10e", v);
    return std::strtof(buf, nullptr);
}

namespace units {                       // I3Units / I3Constants
constexpr double nanometer = 1e-9;
constexpr double micrometer = 1e-6;
constexpr double deg = 3.14159265358979323846 / 180.0;
constexpr double c_light = 0.299792458; // m/ns
}

struct FunctionData {                   // I3CLSimFunctionFromTable / Constant (+ on the host: FromTable with its own wavelengths, DeltaPeak)
    int kind = CLSIMHIP_FUNCTION_CONSTANT;
    double start = 0, step = 0, value = 1.0;
    std::vector<double> values, wavelengths;
    double eval(double wlen) const;     // host GetValue()
    bool on_device() const { return kind == CLSIMHIP_FUNCTION_TABLE || kind == CLSIMHIP_FUNCTION_CONSTANT; }
};

struct RandomValueData {                // InterpolatedDistribution / Constant
    int kind = CLSIMHIP_RANDOM_CONSTANT;
    double first = 0, spacing = 0, value = 0;
    std::vector<double> y, x;           // x: INTERPOLATED_X only
};

struct MediumData {                     // I3CLSimMediumProperties (IceCube function classes)
    int num_layers = 0;
    double layers_z_start = 0, layers_height = 0, min_wlen = 0, max_wlen = 0;
    int lengths_kind = CLSIMHIP_LENGTHS_CONSTANT;
    std::vector<double> abs_length, sca_length;
    double alpha = 0, kappa = 0, A = 0, B = 0, D = 0, E = 0;
    std::vector<double> a_dust400, delta_tau, b400;
    int table_n = 0;                    // CLSIMHIP_LENGTHS_TABLE: FromTable per layer, [layer][table_n]
    double table_start = 0, table_step = 0;
    bool table_16bit = false;
    std::vector<double> abs_table, sca_table;
    double n[5] = {0, 0, 0, 0, 0}, g[5] = {0, 0, 0, 0, 0};
    int phase_kind = CLSIMHIP_REFINDEX_ICECUBE, group_kind = CLSIMHIP_REFINDEX_ICECUBE;
    FunctionData phase_table, group_table;
    int scatter_kind = CLSIMHIP_SCATTER_MIXED;
    double liu_fraction = 0, mean_cosine = 0;
    bool has_aniso = false;
    double aniso_azimuth = 0, aniso_k1 = 0, aniso_k2 = 0;
    bool has_pre = false, pre_renorm = false, has_post = false, post_renorm = false;
    double pre[9] = {0}, post[9] = {0};
    bool has_tilt = false;
    std::vector<double> tilt_distances, tilt_z, tilt_corr; // corr[nd][nz]
    double tilt_azimuth = 0;

    double phase_ref_index(double wlen) const;             // RefIndexIceCube::GetValue("phase")
    void validate() const;
};

MediumData medium_from_desc(const clsimhip_medium_desc &d);
MediumData medium_from_ppc(const std::string &dir, double detector_center_depth, bool use_tilt);
MediumData medium_from_photonics(const std::string &table_file, double detector_center_depth);
void dom_acceptance(double dom_radius, double efficiency, std::vector<double> &values, double &start, double &step);
RandomValueData make_cherenkov_generator(const FunctionData &bias, const MediumData &m);
RandomValueData make_wlen_generator(const FunctionData &spectrum, const FunctionData &bias, const MediumData &m);    // makeWavelengthGenerator

struct GeometryInput {                  // I3CLSimSimpleGeometry
    std::vector<int32_t> string_ids;
    std::vector<uint32_t> dom_ids;
    std::vector<double> x, y, z;
    std::vector<std::string> subdetectors;
    double om_radius = 0;
};

// What the reference's geometry code generator emits (GeometrySource.cxx:1153-1269,
// 619-700), as typed arrays.
struct GeoTables {
    int num_strings = 0;
    float om_radius = 0, string_max_radius = 0;
    std::vector<float> str_x, str_y, str_radius, str_minz, str_maxz;
    std::vector<uint8_t> str_set;
    int num_sets = 0, max_layers = 0;
    std::vector<uint16_t> set_nlayers;
    std::vector<float> set_startz, set_height;
    std::vector<uint16_t> layer_to_om;                  // padded to a multiple of 64 with 0xFFFF
    struct Cells { int nx = 0, ny = 0; float wx = 0, wy = 0, sx = 0, sy = 0; std::vector<uint16_t> index; };
    std::vector<Cells> cells;                           // one per subdetector (sorted by name)
    std::vector<std::string> subdetector_names;
    int max_dom_index = 0;
    float dom_mul_x = 0, dom_mul_y = 0;
    std::vector<int16_t> dom_tx, dom_ty;
    std::vector<float> dom_tz;
    std::vector<uint32_t> dom_start;
    std::vector<float> dom_meanx, dom_meany;
    std::vector<int32_t> string_index_to_id;            // host-side remap (OpenCL.cxx:1565-1600)
    std::vector<std::vector<uint32_t>> dom_index_to_id;
};
GeoTables build_geometry(const GeometryInput &in);
GeometryInput geometry_from_text_file(const std::string &filename, double om_radius, int32_t string_min, int32_t string_max,
                                      uint32_t dom_min, uint32_t dom_max);

// MWC multipliers / seeding
void mwc_multipliers(uint32_t *out, size_t count);
void seed_streams(const uint32_t *a, size_t count, uint64_t seed, uint64_t *x);
bool load_multipliers_from_file(const char *path, uint32_t *a, size_t count);

// wire.cpp: portable-binary-archive payload of I3Vector<I3CLSimStep> / I3Vector<I3CLSimPhoton>
size_t portable_uint_encode(uint64_t v, uint8_t out[9]);
size_t portable_uint_decode(const uint8_t *in, size_t bytes, uint64_t *v);
size_t series_blob_size(size_t n, size_t record);
void series_encode(const void *records, size_t n, size_t record, unsigned version, uint8_t *out, size_t cap, size_t *written);
void series_decode(const uint8_t *in, size_t bytes, size_t record, unsigned version, const char *class_name, void *out, size_t cap,
                   size_t *n_out, size_t *consumed);

} // namespace clsimhip
