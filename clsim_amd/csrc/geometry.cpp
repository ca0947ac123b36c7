// DOM acceleration structure: strings -> xy cell grid per subdetector -> z layer
// tables ("string sets") -> int16 DOM position templates.
//
// Produces, as typed arrays, exactly the constants the reference's geometry code
// generator prints into the OpenCL program
// (private/opencl/I3CLSimHelperGenerateGeometrySource.cxx:712-1275 and :499-709).
// The search strategy (grow an NxN grid until no cell holds two strings; try
// floor((maxZ-minZ+dz)/dz) layers, then +1, then 1,2,3...) decides which photon
// paths test which DOMs, so it is followed step by step; the values go through
// to_float_literal() because the reference prints them with "%.10e".
#include <algorithm>
#include <cmath>
#include <fstream>
#include <set>

#include "host_model.h"

namespace clsimhip {
namespace {

struct Dom { uint32_t id; double x, y, z; };
struct String {
    int id = 0;
    unsigned short subdet = 0;
    double mean_x = 0, mean_y = 0, max_z = NAN, min_z = NAN, mean_dz = NAN, max_r = NAN;
    std::vector<Dom> doms;
};

// interval [lo,hi] touches or lies inside [cmin,cmax] (GeometrySource.cxx:212-229, 305-313)
inline bool overlaps(double lo, double hi, double cmin, double cmax)
{
    return ((lo <= cmin) && (hi >= cmin)) || ((lo <= cmax) && (hi >= cmax)) || ((lo >= cmin) && (hi <= cmax));
}

// GeometrySource.cxx:135-271
bool divide_into_cells(const std::vector<String> &strings, unsigned short subdet, unsigned n, double &start_x,
                       double &start_y, double &width_x, double &width_y, std::vector<uint16_t> &cell_to_string)
{
    double min_x = NAN, min_y = NAN, max_x = NAN, max_y = NAN;
    for (const String &s : strings) {
        if (s.subdet != subdet) continue;
        if ((s.mean_x - s.max_r < min_x) || std::isnan(min_x)) min_x = s.mean_x - s.max_r;
        if ((s.mean_y - s.max_r < min_y) || std::isnan(min_y)) min_y = s.mean_y - s.max_r;
        if ((s.mean_x + s.max_r > max_x) || std::isnan(max_x)) max_x = s.mean_x + s.max_r;
        if ((s.mean_y + s.max_r > max_y) || std::isnan(max_y)) max_y = s.mean_y + s.max_r;
    }
    start_x = min_x;
    start_y = min_y;
    width_x = (max_x - min_x) / static_cast<double>(n);
    width_y = (max_y - min_y) / static_cast<double>(n);
    cell_to_string.assign(static_cast<size_t>(n) * n, 0xFFFF);
    for (unsigned i = 0; i < n; ++i) {
        const double x0 = start_x + static_cast<double>(i) * width_x;
        const double x1 = start_x + static_cast<double>(i + 1) * width_x;
        for (unsigned j = 0; j < n; ++j) {
            const double y0 = start_y + static_cast<double>(j) * width_y;
            const double y1 = start_y + static_cast<double>(j + 1) * width_y;
            bool found = false;
            for (size_t k = 0; k < strings.size(); ++k) {
                const String &s = strings[k];
                if (s.subdet != subdet) continue;
                if (overlaps(s.mean_x - s.max_r, s.mean_x + s.max_r, x0, x1) &&
                    overlaps(s.mean_y - s.max_r, s.mean_y + s.max_r, y0, y1)) {
                    if (found) return false;        // two strings in one cell
                    found = true;
                    cell_to_string[static_cast<size_t>(j) * n + i] = static_cast<uint16_t>(k);
                }
            }
        }
    }
    return true;
}

// GeometrySource.cxx:375-446
bool divide_into_layers(const String &s, unsigned n, double om_radius, double min_hint, double max_hint,
                        double &start_z, double &height, std::vector<uint16_t> &layer_to_dom)
{
    if (n == 0 || om_radius < 0.) return false;
    if (s.doms.size() >= 0xFFFF) throw Error(CLSIMHIP_ERR_CONFIG, "Dom numbers >= 65535 are not supported!");
    layer_to_dom.assign(n, 0xFFFF);
    double min_z = min_hint, max_z = max_hint;
    if ((s.min_z - om_radius < min_z) || std::isnan(min_z)) min_z = s.min_z - om_radius;
    if ((s.max_z + om_radius > max_z) || std::isnan(max_z)) max_z = s.max_z + om_radius;
    start_z = min_z;
    height = (max_z - min_z) / static_cast<double>(n);
    for (unsigned i = 0; i < n; ++i) {
        const double z0 = start_z + static_cast<double>(i) * height;
        const double z1 = start_z + static_cast<double>(i + 1) * height;
        for (size_t d = 0; d < s.doms.size(); ++d) {
            const double z = s.doms[d].z;
            if (!overlaps(z - om_radius, z + om_radius, z0, z1)) continue;
            if (layer_to_dom[i] != 0xFFFF) return false;   // two DOMs of one string in a layer
            layer_to_dom[i] = static_cast<uint16_t>(d);
        }
    }
    return true;
}

// GeometrySource.cxx:273-342
bool matches_layering(const String &s, double start_z, double height, unsigned n, double om_radius,
                      const std::vector<uint16_t> &layer_to_dom)
{
    if (n == 0 || om_radius < 0.) return false;
    size_t assigned = 0;
    for (unsigned i = 0; i < n; ++i) {
        const double z0 = start_z + static_cast<double>(i) * height;
        const double z1 = start_z + static_cast<double>(i + 1) * height;
        uint16_t should = 0xFFFF;
        for (size_t d = 0; d < s.doms.size(); ++d) {
            const double z = s.doms[d].z;
            if (!overlaps(z - om_radius, z + om_radius, z0, z1)) continue;
            if (should != 0xFFFF) return false;
            should = static_cast<uint16_t>(d);
            ++assigned;
        }
        if (layer_to_dom[i] != should) return false;
    }
    return assigned == s.doms.size();
}

// static_cast<short>(double) of the reference build (x86-64: cvttsd2si, truncation;
// the NaN of a perfectly straight string, 0/0, becomes 0 -- SURVEY.md H6)
inline int16_t to_short(double v)
{
    if (std::isnan(v)) return 0;
    return static_cast<int16_t>(static_cast<int>(v));
}

} // namespace

// I3CLSimSimpleGeometryTextFile.cxx:43-100
GeometryInput geometry_from_text_file(const std::string &filename, double om_radius, int32_t string_min, int32_t string_max,
                                      uint32_t dom_min, uint32_t dom_max)
{
    std::ifstream f(filename.c_str());
    if (f.fail()) throw Error(CLSIMHIP_ERR_IO, "Could not open input file");
    GeometryInput g;
    g.om_radius = om_radius;
    int64_t read_string, read_dom;
    double x, y, z;
    while (f >> read_string >> read_dom >> x >> y >> z) {
        if (read_string < INT32_MIN || read_string > INT32_MAX || read_dom < 0 || read_dom > static_cast<int64_t>(UINT32_MAX))
            throw Error(CLSIMHIP_ERR_IO, "Read error (numeric conversion)!");
        const int32_t s = static_cast<int32_t>(read_string);
        const uint32_t d = static_cast<uint32_t>(read_dom);
        if ((s < string_min) || (s > string_max) || (d < dom_min) || (d > dom_max)) continue;
        g.string_ids.push_back(s); g.dom_ids.push_back(d);
        g.x.push_back(x); g.y.push_back(y); g.z.push_back(z);
        g.subdetectors.push_back("default");
    }
    return g;
}

GeoTables build_geometry(const GeometryInput &in)
{
    const size_t n = in.string_ids.size();
    if (n == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "Empty geometry provided.");
    if (in.dom_ids.size() != n || in.x.size() != n || in.y.size() != n || in.z.size() != n || in.subdetectors.size() != n)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "geometry arrays differ in length");
    if (in.om_radius < 0.) throw Error(CLSIMHIP_ERR_ARGUMENT, "Zero or negative OM radius.");

    // strings are ordered like std::set<pair<int,string>>; subdetectors by name (:737-773)
    std::set<std::pair<int, std::string>> keys;
    std::set<std::string> subdet_set;
    for (size_t i = 0; i < n; ++i) {
        keys.insert({in.string_ids[i], in.subdetectors[i]});
        subdet_set.insert(in.subdetectors[i]);
    }
    if (keys.size() >= 0xFFFF - 1) throw Error(CLSIMHIP_ERR_CONFIG, "More than 65534 strings are not supported.");
    GeoTables g;
    g.subdetector_names.assign(subdet_set.begin(), subdet_set.end());
    std::map<std::string, unsigned short> subdet_index;
    for (size_t k = 0; k < g.subdetector_names.size(); ++k) subdet_index[g.subdetector_names[k]] = static_cast<unsigned short>(k);

    // :776-882
    std::vector<String> strings;
    strings.reserve(keys.size());
    double string_max_r = NAN;
    for (const auto &key : keys) {
        String s;
        s.id = key.first;
        s.subdet = subdet_index[key.second];
        double last_z = NAN, last_dz = NAN, sum_dz = 0.;
        unsigned num_dz = 0;
        for (size_t i = 0; i < n; ++i) {
            if (in.string_ids[i] != key.first || in.subdetectors[i] != key.second) continue;
            s.mean_x += in.x[i];
            s.mean_y += in.y[i];
            if ((in.z[i] > s.max_z) || std::isnan(s.max_z)) s.max_z = in.z[i];
            if ((in.z[i] < s.min_z) || std::isnan(s.min_z)) s.min_z = in.z[i];
            if (std::isnan(last_z)) {
                last_z = in.z[i];
            } else {
                const double dz = std::abs(last_z - in.z[i]);
                last_z = in.z[i];
                if (!std::isnan(last_dz)) {
                    // gaps above 1.75x the running mean are a missing DOM, not a spacing
                    if (dz < 1.75 * sum_dz / static_cast<double>(num_dz)) { sum_dz += dz; ++num_dz; last_dz = dz; }
                } else {
                    last_dz = dz; sum_dz += dz; ++num_dz;
                }
            }
            s.doms.push_back({in.dom_ids[i], in.x[i], in.y[i], in.z[i]});
        }
        s.mean_x /= static_cast<double>(s.doms.size());
        s.mean_y /= static_cast<double>(s.doms.size());
        s.mean_dz = sum_dz / static_cast<double>(num_dz);
        for (const Dom &d : s.doms) {
            const double dx = s.mean_x - d.x, dy = s.mean_y - d.y;
            const double r = std::sqrt(dx * dx + dy * dy) + in.om_radius;
            if ((r > s.max_r) || std::isnan(s.max_r)) s.max_r = r;
            if ((r > string_max_r) || std::isnan(string_max_r)) string_max_r = r;
        }
        strings.push_back(std::move(s));
    }
    const size_t ns = strings.size();

    // xy cells (:905-949)
    for (unsigned short sd = 0; sd < g.subdetector_names.size(); ++sd) {
        GeoTables::Cells c;
        unsigned grid = 1;
        double sx, sy, wx, wy;
        while (!divide_into_cells(strings, sd, grid, sx, sy, wx, wy, c.index)) {
            ++grid;
            if (grid >= 1000) throw Error(CLSIMHIP_ERR_CONFIG, "Could not generate a x-y cell division for subdetector " + g.subdetector_names[sd]);
        }
        c.nx = c.ny = static_cast<int>(grid);
        c.sx = to_float_literal(sx); c.sy = to_float_literal(sy);
        c.wx = to_float_literal(wx); c.wy = to_float_literal(wy);
        g.cells.push_back(std::move(c));
    }

    // z layers / string sets (:956-1112)
    std::vector<unsigned> set_n;
    std::vector<double> set_start, set_height;
    std::vector<std::vector<uint16_t>> set_table;
    g.str_set.resize(ns);
    unsigned max_layers = 0;
    for (size_t si = 0; si < ns; ++si) {
        const String &s = strings[si];
        bool matched = false;
        for (size_t k = 0; k < set_n.size(); ++k)
            if (matches_layering(s, set_start[k], set_height[k], set_n[k], in.om_radius, set_table[k])) {
                g.str_set[si] = static_cast<uint8_t>(k);
                matched = true;
                break;
            }
        if (matched) continue;
        g.str_set[si] = static_cast<uint8_t>(set_n.size());
        if (set_n.size() + 1 >= 0xFF) throw Error(CLSIMHIP_ERR_CONFIG, "Not more than 255 different string layer divisions (\"string sets\") are supported!");
        const double lo = s.min_z - s.mean_dz / 2., hi = s.max_z + s.mean_dz / 2.;
        const unsigned guess = static_cast<unsigned>((s.max_z - s.min_z + s.mean_dz) / s.mean_dz);
        double start = NAN, height = NAN;
        std::vector<uint16_t> table;
        unsigned nl = guess;
        bool ok = divide_into_layers(s, nl, in.om_radius, lo, hi, start, height, table);
        if (!ok) { nl = guess + 1; ok = divide_into_layers(s, nl, in.om_radius, lo, hi, start, height, table); }
        if (!ok) {
            for (nl = 1;; ++nl) {
                if (divide_into_layers(s, nl, in.om_radius, lo, hi, start, height, table)) break;
                if (nl + 1 >= 1000) throw Error(CLSIMHIP_ERR_CONFIG, "There does not seem to be a possible layer division for a string");
            }
        }
        set_n.push_back(nl); set_start.push_back(start); set_height.push_back(height); set_table.push_back(table);
        max_layers = std::max(max_layers, nl);
    }
    g.num_sets = static_cast<int>(set_n.size());
    g.max_layers = static_cast<int>(max_layers);
    const size_t used = static_cast<size_t>(g.num_sets) * max_layers;
    g.layer_to_om.assign((used / 64 + 1) * 64, 0xFFFF);
    for (size_t j = 0; j < set_n.size(); ++j)
        for (unsigned i = 0; i < set_n[j]; ++i) g.layer_to_om[j * max_layers + i] = set_table[j][i];
    for (size_t j = 0; j < set_n.size(); ++j) {
        g.set_nlayers.push_back(static_cast<uint16_t>(set_n[j]));
        g.set_startz.push_back(to_float_literal(set_start[j]));
        g.set_height.push_back(to_float_literal(set_height[j]));
    }

    // DOM position templates (:499-709)
    std::vector<double> mean_x(ns, 0.), mean_y(ns, 0.);
    for (size_t si = 0; si < ns; ++si) {
        for (const Dom &d : strings[si].doms) { mean_x[si] += d.x; mean_y[si] += d.y; }
        mean_x[si] /= static_cast<double>(strings[si].doms.size());
        mean_y[si] /= static_cast<double>(strings[si].doms.size());
        g.max_dom_index = std::max<int>(g.max_dom_index, static_cast<int>(strings[si].doms.size()));
    }
    struct P3 { double x, y, z; };
    std::vector<std::vector<P3>> templates;
    std::vector<size_t> in_template(ns);
    const double epsilon = 1e-1 * 1e-3;             // 0.1 mm
    for (size_t si = 0; si < ns; ++si) {
        const String &s = strings[si];
        bool found = false;
        for (size_t t = 0; t < templates.size() && !found; ++t) {
            if (templates[t].size() != s.doms.size()) continue;
            bool match = true;
            for (size_t j = 0; j < s.doms.size() && match; ++j) {
                if (std::abs(templates[t][j].x - (s.doms[j].x - mean_x[si])) > epsilon) match = false;
                else if (std::abs(templates[t][j].y - (s.doms[j].y - mean_y[si])) > epsilon) match = false;
                else if (std::abs(templates[t][j].z - (s.doms[j].z)) > epsilon) match = false;
            }
            if (match) { in_template[si] = t; found = true; }
        }
        if (found) continue;
        std::vector<P3> tpl;
        for (const Dom &d : s.doms) tpl.push_back({d.x - mean_x[si], d.y - mean_y[si], d.z});
        templates.push_back(std::move(tpl));
        in_template[si] = templates.size() - 1;
    }
    double max_abs_x = NAN, max_abs_y = NAN;
    std::vector<size_t> tpl_start(templates.size());
    std::vector<P3> flat;
    for (size_t t = 0; t < templates.size(); ++t) {
        tpl_start[t] = flat.size();
        for (const P3 &p : templates[t]) {
            flat.push_back(p);
            if ((std::abs(p.x) > max_abs_x) || std::isnan(max_abs_x)) max_abs_x = std::abs(p.x);
            if ((std::abs(p.y) > max_abs_y) || std::isnan(max_abs_y)) max_abs_y = std::abs(p.y);
        }
    }
    g.dom_mul_x = to_float_literal(max_abs_x / 32767.);
    g.dom_mul_y = to_float_literal(max_abs_y / 32767.);
    for (const P3 &p : flat) {
        g.dom_tx.push_back(to_short(p.x / (max_abs_x / 32767.)));
        g.dom_ty.push_back(to_short(p.y / (max_abs_y / 32767.)));
        g.dom_tz.push_back(to_float_literal(p.z));
    }

    g.num_strings = static_cast<int>(ns);
    g.om_radius = to_float_literal(in.om_radius);
    g.string_max_radius = to_float_literal(string_max_r);
    for (size_t si = 0; si < ns; ++si) {
        const String &s = strings[si];
        g.str_x.push_back(to_float_literal(s.mean_x));
        g.str_y.push_back(to_float_literal(s.mean_y));
        g.str_radius.push_back(to_float_literal(s.max_r));
        g.str_minz.push_back(to_float_literal(s.min_z));
        g.str_maxz.push_back(to_float_literal(s.max_z));
        g.dom_start.push_back(static_cast<uint32_t>(tpl_start[in_template[si]]));
        g.dom_meanx.push_back(to_float_literal(mean_x[si]));
        g.dom_meany.push_back(to_float_literal(mean_y[si]));
        g.string_index_to_id.push_back(s.id);
        std::vector<uint32_t> ids;
        for (const Dom &d : s.doms) ids.push_back(d.id);
        g.dom_index_to_id.push_back(std::move(ids));
    }
    return g;
}

} // namespace clsimhip
