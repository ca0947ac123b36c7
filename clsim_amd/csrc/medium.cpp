// Host configuration helpers: ice model loader, DOM acceptance, Cherenkov
// spectrum.  The reference does this in Python / C++ on the host
// (python/MakeIceCubeMediumProperties.py, python/GetIceCubeDOMAcceptance.py,
// private/clsim/I3CLSimModuleHelper.cxx); values stay in double here and are
// turned into kernel constants by tables.cpp.
#include <algorithm>
#include <cmath>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <sys/stat.h>

#include "host_model.h"

namespace clsimhip {

static bool file_exists(const std::string &p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// numpy.loadtxt semantics: '#' comments, blank lines skipped, whitespace separated
static std::vector<std::vector<double>> load_table(const std::string &path)
{
    std::ifstream f(path);
    if (!f.good()) throw Error(CLSIMHIP_ERR_IO, "cannot open " + path);
    std::vector<std::vector<double>> rows;
    std::string line;
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.resize(hash);
        std::istringstream ss(line);
        std::vector<double> row;
        std::string tok;
        while (ss >> tok) {
            char *end = nullptr;
            const double v = std::strtod(tok.c_str(), &end);
            if (end == tok.c_str() || *end != '\0') throw Error(CLSIMHIP_ERR_IO, "bad number '" + tok + "' in " + path);
            row.push_back(v);
        }
        if (!row.empty()) rows.push_back(row);
    }
    if (rows.empty()) throw Error(CLSIMHIP_ERR_IO, path + " is empty");
    return rows;
}

double FunctionData::eval(double wlen) const
{
    // I3CLSimFunctionFromTable::GetValue, equal spacing (FromTable.cxx:105-122)
    if (kind == CLSIMHIP_FUNCTION_CONSTANT) return value;
    if (kind == CLSIMHIP_FUNCTION_DELTA_PEAK) return (wlen == value) ? INFINITY : 0.;      // FunctionDeltaPeak.cxx (never sampled: makeWavelengthGenerator turns it into a constant)
    if (kind == CLSIMHIP_FUNCTION_TABLE_X) {                // FromTable.cxx:123-145
        if (wlen <= wavelengths[0]) return values[0];
        for (size_t i = 1; i < wavelengths.size(); ++i)
            if (wlen <= wavelengths[i]) {
                const double fraction = (wlen - wavelengths[i - 1]) / (wavelengths[i] - wavelengths[i - 1]);
                return values[i - 1] + (values[i] - values[i - 1]) * fraction;
            }
        return values[wavelengths.size() - 1];
    }
    double fbin;
    double fraction = std::modf((wlen - start) / step, &fbin);
    int ibin = static_cast<int>(fbin);
    if ((ibin < 0) || ((ibin == 0) && (fraction < 0))) {
        ibin = 0;
        fraction = 0.;
    } else if (static_cast<size_t>(ibin) >= values.size() - 1) {
        ibin = static_cast<int>(values.size()) - 2;
        fraction = 1.;
    }
    return values[ibin] + (values[ibin + 1] - values[ibin]) * fraction;
}

double MediumData::phase_ref_index(double wlen) const
{
    if (phase_kind == CLSIMHIP_REFINDEX_TABLE) return phase_table.eval(wlen);
    // RefIndexIceCube.cxx:84-101
    const double x = wlen / units::micrometer;
    return n[0] + x * (n[1] + x * (n[2] + x * (n[3] + x * n[4])));
}

void MediumData::validate() const
{
    if (num_layers < 1) throw Error(CLSIMHIP_ERR_ARGUMENT, "medium needs at least one layer");
    if (!(layers_height > 0)) throw Error(CLSIMHIP_ERR_ARGUMENT, "layer height must be positive");
    // A zero, negative, infinite or NaN length turns a photon's absorption budget into NaN, and then the reference's
    // photon loop never ends (propagation_kernel.c.cl:536, 681-691); on a GPU that is a hang, so it is refused here.
    auto usable = [](const std::vector<double> &v, bool strictly_positive, const char *what) {
        for (double x : v)
            if (!std::isfinite(x) || x < 0. || (strictly_positive && !(x > 0.)))
                throw Error(CLSIMHIP_ERR_ARGUMENT, std::string(what) + " must be finite and " + (strictly_positive ? "positive" : "non-negative"));
    };
    usable(abs_length, true, "absorption lengths"); usable(sca_length, true, "scattering lengths");
    usable(abs_table, true, "tabulated absorption lengths"); usable(sca_table, true, "tabulated scattering lengths");
    usable(b400, true, "b400");
    if (lengths_kind == CLSIMHIP_LENGTHS_ICECUBE && a_dust400.size() == delta_tau.size()) {
        // AbsLenIceCube.cxx:63-77: the dust term may be negative, the absorptivity as a whole must not be
        const double lo = (min_wlen > 0. && std::isfinite(min_wlen)) ? min_wlen : 265e-9;
        const double hi = (max_wlen > lo && std::isfinite(max_wlen)) ? max_wlen : 675e-9;
        for (size_t l = 0; l < a_dust400.size(); ++l)
            for (double w : {lo, 0.5 * (lo + hi), hi}) {
                const double x = w / units::nanometer;
                const double absorptivity = (D * a_dust400[l] + E) * std::pow(x, -kappa) + A * std::exp(-B / x) * (1. + 0.01 * delta_tau[l]);
                if (!std::isfinite(absorptivity) || !(absorptivity > 0.))
                    throw Error(CLSIMHIP_ERR_ARGUMENT, "ice layer " + std::to_string(l) + " has no positive, finite absorptivity");
            }
    }
    const size_t nl = static_cast<size_t>(num_layers);
    if (lengths_kind == CLSIMHIP_LENGTHS_CONSTANT) {
        if (abs_length.size() != nl || sca_length.size() != nl)
            throw Error(CLSIMHIP_ERR_ARGUMENT, "abs_length / sca_length need one entry per layer");
    } else if (lengths_kind == CLSIMHIP_LENGTHS_ICECUBE) {
        if (a_dust400.size() != nl || delta_tau.size() != nl || b400.size() != nl)
            throw Error(CLSIMHIP_ERR_ARGUMENT, "a_dust400 / delta_tau / b400 need one entry per layer");
    } else if (lengths_kind == CLSIMHIP_LENGTHS_TABLE) {
        if (table_n < 2) throw Error(CLSIMHIP_ERR_ARGUMENT, "values must contain at least 2 elements!");      // FromTable.cxx:84
        if (!(table_step > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "wlenStep must not be <= 0!");              // FromTable.cxx:83
        if (abs_table.size() != nl * static_cast<size_t>(table_n) || sca_table.size() != abs_table.size())
            throw Error(CLSIMHIP_ERR_ARGUMENT, "abs_length_table / sca_length_table need num_layers x table_num_wavelengths entries");
    } else
        throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown lengths_kind");
    for (int which = 0; which < 2; ++which) {
        const int kind = which ? group_kind : phase_kind;
        const FunctionData &f = which ? group_table : phase_table;
        if (kind == CLSIMHIP_REFINDEX_ICECUBE) continue;
        if (which == 1 && kind == CLSIMHIP_REFINDEX_DISPERSION) {
            // no override: the group velocity comes from the phase index's derivative (MediumPropertiesSource.cxx:274-300), which
            // FromTable does not have (FunctionFromTable.h:67; the reference's generated program would not compile)
            if (phase_kind != CLSIMHIP_REFINDEX_ICECUBE)
                throw Error(CLSIMHIP_ERR_ARGUMENT, "group velocity from dispersion needs a phase refractive index with a derivative (RefIndexIceCube)");
            continue;
        }
        if (kind != CLSIMHIP_REFINDEX_TABLE) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown refractive index kind");
        if (f.kind != CLSIMHIP_FUNCTION_TABLE || f.values.size() < 2 || !(f.step > 0.))
            throw Error(CLSIMHIP_ERR_ARGUMENT, "a tabulated refractive index needs at least 2 values and a positive step");
    }
    // FromTable has no derivative: the group velocity cannot come from the dispersion (MediumPropertiesSource.cxx:226-237)
    if (phase_kind == CLSIMHIP_REFINDEX_TABLE && group_kind != CLSIMHIP_REFINDEX_TABLE)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "a tabulated phase refractive index needs a tabulated group refractive index override");
    if (scatter_kind < CLSIMHIP_SCATTER_HG || scatter_kind > CLSIMHIP_SCATTER_MIXED)
        throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown scatter_kind");
    if (has_tilt) {
        if (tilt_distances.size() < 2 || tilt_z.size() < 2)
            throw Error(CLSIMHIP_ERR_ARGUMENT, "tilt needs at least 2 distances and 2 z coordinates");
        if (tilt_corr.size() != tilt_distances.size() * tilt_z.size())
            throw Error(CLSIMHIP_ERR_ARGUMENT, "tilt correction table has the wrong size");
    }
}

MediumData medium_from_desc(const clsimhip_medium_desc &d)
{
    MediumData m;
    m.num_layers = d.num_layers;
    m.layers_z_start = d.layers_z_start;
    m.layers_height = d.layers_height;
    m.min_wlen = d.min_wavelength;
    m.max_wlen = d.max_wavelength;
    m.lengths_kind = d.lengths_kind;
    const size_t nl = d.num_layers > 0 ? static_cast<size_t>(d.num_layers) : 0;
    if (d.lengths_kind == CLSIMHIP_LENGTHS_CONSTANT) {
        if (!d.abs_length || !d.sca_length) throw Error(CLSIMHIP_ERR_ARGUMENT, "abs_length / sca_length are null");
        m.abs_length.assign(d.abs_length, d.abs_length + nl);
        m.sca_length.assign(d.sca_length, d.sca_length + nl);
    } else if (d.lengths_kind == CLSIMHIP_LENGTHS_TABLE) {
        if (!d.abs_length_table || !d.sca_length_table || d.table_num_wavelengths < 0)
            throw Error(CLSIMHIP_ERR_ARGUMENT, "abs_length_table / sca_length_table are null");
        const size_t total = nl * static_cast<size_t>(d.table_num_wavelengths);
        m.table_n = d.table_num_wavelengths;
        m.table_start = d.table_start_wavelength;
        m.table_step = d.table_wavelength_step;
        m.table_16bit = d.table_store_as_16bit != 0;
        m.abs_table.assign(d.abs_length_table, d.abs_length_table + total);
        m.sca_table.assign(d.sca_length_table, d.sca_length_table + total);
    } else {
        if (!d.a_dust400 || !d.delta_tau || !d.b400) throw Error(CLSIMHIP_ERR_ARGUMENT, "ice tables are null");
        m.a_dust400.assign(d.a_dust400, d.a_dust400 + nl);
        m.delta_tau.assign(d.delta_tau, d.delta_tau + nl);
        m.b400.assign(d.b400, d.b400 + nl);
        m.alpha = d.alpha; m.kappa = d.kappa; m.A = d.A; m.B = d.B; m.D = d.D; m.E = d.E;
    }
    for (int i = 0; i < 5; ++i) { m.n[i] = d.n[i]; m.g[i] = d.g[i]; }
    m.phase_kind = d.phase_index_kind;
    m.group_kind = d.group_index_kind;
    auto copy_function = [](const clsimhip_function &f, FunctionData &out, const char *what) {
        if (f.kind != CLSIMHIP_FUNCTION_TABLE || !f.values || f.n < 0) throw Error(CLSIMHIP_ERR_ARGUMENT, std::string(what) + " is not a table");
        out.kind = CLSIMHIP_FUNCTION_TABLE;
        out.start = f.start; out.step = f.step;
        out.values.assign(f.values, f.values + f.n);
    };
    if (m.phase_kind == CLSIMHIP_REFINDEX_TABLE) copy_function(d.phase_index_table, m.phase_table, "phase_index_table");
    if (m.group_kind == CLSIMHIP_REFINDEX_TABLE) copy_function(d.group_index_table, m.group_table, "group_index_table");
    m.scatter_kind = d.scatter_kind;
    m.liu_fraction = d.liu_fraction;
    m.mean_cosine = d.mean_cosine;
    m.has_aniso = d.has_anisotropy != 0;
    m.aniso_azimuth = d.aniso_azimuth; m.aniso_k1 = d.aniso_k1; m.aniso_k2 = d.aniso_k2;
    m.has_pre = d.has_pre_transform != 0; m.pre_renorm = d.pre_renormalize != 0;
    m.has_post = d.has_post_transform != 0; m.post_renorm = d.post_renormalize != 0;
    for (int i = 0; i < 9; ++i) { m.pre[i] = d.pre_matrix[i]; m.post[i] = d.post_matrix[i]; }
    m.has_tilt = d.has_tilt != 0;
    if (m.has_tilt) {
        if (d.tilt_num_distances < 2 || d.tilt_num_z < 2 || !d.tilt_distances || !d.tilt_z_coordinates || !d.tilt_z_corrections)
            throw Error(CLSIMHIP_ERR_ARGUMENT, "incomplete tilt description");
        const size_t nd = d.tilt_num_distances, nz = d.tilt_num_z;
        m.tilt_distances.assign(d.tilt_distances, d.tilt_distances + nd);
        m.tilt_z.assign(d.tilt_z_coordinates, d.tilt_z_coordinates + nz);
        m.tilt_corr.assign(d.tilt_z_corrections, d.tilt_z_corrections + nd * nz);
        m.tilt_azimuth = d.tilt_azimuth;
    }
    m.validate();
    return m;
}

// python/MakeIceCubeMediumProperties.py:49-256
MediumData medium_from_ppc(const std::string &dir, double center_depth, bool use_tilt_if_available)
{
    bool use_tilt = false;
    if (use_tilt_if_available) {
        const bool has_par = file_exists(dir + "/tilt.par"), has_dat = file_exists(dir + "/tilt.dat");
        if (has_par && !has_dat) throw Error(CLSIMHIP_ERR_IO, "ice model directory has tilt.par but tilt.dat is missing!");
        if (has_dat && !has_par) throw Error(CLSIMHIP_ERR_IO, "ice model directory has tilt.dat but tilt.par is missing!");
        use_tilt = has_par && has_dat;
    }
    const auto dat = load_table(dir + "/icemodel.dat");
    const auto par = load_table(dir + "/icemodel.par");
    const auto cfg = load_table(dir + "/cfg.txt");

    MediumData m;
    m.lengths_kind = CLSIMHIP_LENGTHS_ICECUBE;
    if (par.size() == 6) {
        m.alpha = par[0][0]; m.kappa = par[1][0]; m.A = par[2][0]; m.B = par[3][0]; m.D = par[4][0]; m.E = par[5][0];
    } else if (par.size() == 4) {
        m.alpha = par[0][0]; m.kappa = par[1][0]; m.A = par[2][0]; m.B = par[3][0];
        m.D = std::pow(400., m.kappa);      // what ppc does (py:84-89)
        m.E = 0.;
    } else
        throw Error(CLSIMHIP_ERR_IO, dir + "/icemodel.par is not a valid Dima-icemodel file (needs 4 or 6 entries)");
    if (cfg.size() < 4) throw Error(CLSIMHIP_ERR_IO, dir + "/cfg.txt does not have enough configuration lines. It needs at least 4.");
    m.liu_fraction = cfg[2][0];
    m.mean_cosine = cfg[3][0];
    if (cfg.size() > 4 && cfg.size() < 7)
        throw Error(CLSIMHIP_ERR_IO, dir + "/cfg.txt has more than 4 lines but needs at least 7 for ice anisotropy");
    if (cfg.size() > 4) {
        m.has_aniso = true;
        m.aniso_azimuth = cfg[4][0] * units::deg;
        m.aniso_k1 = cfg[5][0];
        m.aniso_k2 = cfg[6][0];
    }
    if (m.liu_fraction < 0. || m.liu_fraction > 1.) throw Error(CLSIMHIP_ERR_IO, "Invalid Liu(SAM) scattering fraction configured in cfg.txt");
    if (m.mean_cosine < -1. || m.mean_cosine > 1.) throw Error(CLSIMHIP_ERR_IO, "Invalid <cos(theta)> configured in cfg.txt");

    const size_t nl = dat.size();
    if (nl < 2) throw Error(CLSIMHIP_ERR_IO, "There is only a single layer in your layer definition file");
    for (const auto &r : dat)
        if (r.size() < 4) throw Error(CLSIMHIP_ERR_IO, "icemodel.dat needs 4 columns");
    const double layer_height = dat[1][0] - dat[0][0];
    if (layer_height <= 0.) throw Error(CLSIMHIP_ERR_IO, "ice layer depths are not in increasing order");
    for (size_t i = 0; i + 1 < nl; ++i)
        if (std::abs((dat[i + 1][0] - dat[i][0]) - layer_height) > 1e-5) throw Error(CLSIMHIP_ERR_IO, "ice layers are not spaced evenly");

    // file order is top -> bottom; the medium wants bottom -> top (py:148-151)
    m.num_layers = static_cast<int>(nl);
    m.a_dust400.resize(nl); m.delta_tau.resize(nl); m.b400.resize(nl);
    for (size_t i = 0; i < nl; ++i) {
        const auto &row = dat[nl - 1 - i];
        m.b400[i] = row[1] / (1. - m.mean_cosine);      // b_e400 -> b_400 (py:153)
        m.a_dust400[i] = row[2];
        m.delta_tau[i] = row[3];
    }
    // depth in the file is the layer centre (ppc); bottom-most layer first (py:158-163)
    const double deepest_top = dat[nl - 1][0] - layer_height / 2.;
    m.layers_z_start = center_depth - (deepest_top + layer_height);
    m.layers_height = layer_height;
    m.min_wlen = 265. * units::nanometer;               // ForcedMinWlen / ForcedMaxWlen (py:178-179)
    m.max_wlen = 675. * units::nanometer;
    // RefIndexIceCube defaults (RefIndexIceCube.cxx:38-47); group override is always set (py:222-230)
    const double n_[5] = {1.55749, -1.57988, 3.99993, -4.68271, 2.09354};
    const double g_[5] = {1.227106, -0.954648, 1.42568, -0.711832, 0.0};
    for (int i = 0; i < 5; ++i) { m.n[i] = n_[i]; m.g[i] = g_[i]; }
    m.scatter_kind = CLSIMHIP_SCATTER_MIXED;

    if (m.has_aniso) {
        // python/util/GetSpiceLeaAnisotropyTransforms.py:39-101
        const double k1 = std::exp(m.aniso_k1), k2 = std::exp(m.aniso_k2), kz = 1. / (k1 * k2);
        const double sa = std::sin(m.aniso_azimuth), ca = std::cos(m.aniso_azimuth);
        const double T[3][3] = {{ca, sa, 0.}, {-sa, ca, 0.}, {0., 0., 1.}};
        const double diag[3] = {k1, k2, kz};
        // numpy.linalg.inv of a diagonal matrix: LAPACK getrf/getri gives exactly 1/d
        const double inv_diag[3] = {1. / k1, 1. / k2, 1. / kz};
        auto sandwich = [&](const double d[3], double out[9]) {
            // (T^T . A) . T with numpy.dot's left-to-right sums
            double ta[3][3];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0.;
                    for (int k = 0; k < 3; ++k) s += T[k][i] * ((k == j) ? d[k] : 0.);
                    ta[i][j] = s;
                }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double s = 0.;
                    for (int k = 0; k < 3; ++k) s += ta[i][k] * T[k][j];
                    out[3 * i + j] = s;
                }
        };
        sandwich(diag, m.pre);
        sandwich(inv_diag, m.post);
        m.has_pre = m.has_post = true;
        m.pre_renorm = m.post_renorm = true;
    }
    if (use_tilt) {
        // python/util/GetIceTiltZShift.py:40-62
        const auto tpar = load_table(dir + "/tilt.par");
        const auto tdat = load_table(dir + "/tilt.dat");
        const size_t nd = tpar.size(), nz = tdat.size();
        m.has_tilt = true;
        m.tilt_azimuth = 225. * units::deg;
        m.tilt_distances.resize(nd);
        for (size_t i = 0; i < nd; ++i) m.tilt_distances[i] = tpar[i][1];
        m.tilt_z.resize(nz);
        for (size_t k = 0; k < nz; ++k) m.tilt_z[k] = center_depth - tdat[nz - 1 - k][0];
        m.tilt_corr.resize(nd * nz);
        for (size_t i = 0; i < nd; ++i)
            for (size_t k = 0; k < nz; ++k) {
                if (tdat[nz - 1 - k].size() < nd + 1) throw Error(CLSIMHIP_ERR_IO, "tilt.dat has too few columns");
                m.tilt_corr[i * nz + k] = tdat[nz - 1 - k][i + 1];
            }
    }
    m.validate();
    return m;
}

// python/MakeIceCubeMediumPropertiesPhotonics.py:47-227
MediumData medium_from_photonics(const std::string &table_file, double /*detector_center_depth: unused by the reference too*/)
{
    std::ifstream f(table_file);
    if (!f.good()) throw Error(CLSIMHIP_ERR_IO, "cannot open " + table_file);
    typedef std::vector<std::string> Line;
    std::vector<Line> parsed;
    std::string text;
    while (std::getline(f, text)) {
        std::istringstream ss(text);
        Line tokens;
        std::string tok;
        while (ss >> tok) tokens.push_back(tok);
        if (tokens.empty() || tokens[0][0] == '#') continue;       // comment lines (py:57)
        for (char &c : tokens[0]) c = static_cast<char>(std::toupper(static_cast<unsigned char>(c)));
        parsed.push_back(tokens);
    }
    auto number = [&](const std::string &t) {
        char *end = nullptr;
        const double v = std::strtod(t.c_str(), &end);
        if (end == t.c_str() || *end != '\0') throw Error(CLSIMHIP_ERR_IO, "bad number '" + t + "' in " + table_file);
        return v;
    };
    const Line *nlayer = nullptr, *nwvl = nullptr;
    for (const Line &l : parsed) {
        if (l[0] == "NLAYER") {
            if (nlayer) throw Error(CLSIMHIP_ERR_IO, "There is more than one \"NLAYER\" entry in your ice table!");
            nlayer = &l;
        } else if (l[0] == "NWVL") {
            if (nwvl) throw Error(CLSIMHIP_ERR_IO, "There is more than one \"NWVL\" entry in your ice table!");
            nwvl = &l;
        }
    }
    if (!nlayer) throw Error(CLSIMHIP_ERR_IO, "There is no \"NLAYER\" entry in your ice table!");
    if (!nwvl) throw Error(CLSIMHIP_ERR_IO, "There is no \"NWVL\" entry in your ice table!");
    if (nlayer->size() < 2 || nwvl->size() < 4) throw Error(CLSIMHIP_ERR_IO, "incomplete NLAYER / NWVL entry in " + table_file);
    const long n_layers = std::strtol((*nlayer)[1].c_str(), nullptr, 10);
    const long n_wlen = std::strtol((*nwvl)[1].c_str(), nullptr, 10);
    double start = number((*nwvl)[2]) * units::nanometer;
    const double step = number((*nwvl)[3]) * units::nanometer;
    start += step / 2.;                                             // bin centres (py:79)

    struct Layer { std::map<std::string, std::vector<double>> v; };
    std::vector<Layer> layers;
    Layer cur;
    size_t rows = 0;
    bool first = true;
    for (const Line &l : parsed) {
        if (l[0] == "NLAYER" || l[0] == "NWVL") continue;
        if (first && l[0] != "LAYER") throw Error(CLSIMHIP_ERR_IO, "Layer definitions should start with the LAYER keyword (reading " + table_file + ")");
        first = false;
        ++rows;
        if (l[0] == "LAYER") {
            if (!cur.v.empty()) layers.push_back(cur);
            cur = Layer();
        } else if (cur.v.count(l[0]))
            throw Error(CLSIMHIP_ERR_IO, "Keyword " + l[0] + " is used twice for one layer (reading " + table_file + ")");
        std::vector<double> vals;
        for (size_t i = 1; i < l.size(); ++i) vals.push_back(number(l[i]) * 1.);
        cur.v[l[0]] = vals;
    }
    if (rows != static_cast<size_t>(n_layers) * 6)
        throw Error(CLSIMHIP_ERR_IO, "Expected " + std::to_string(n_layers) + "*6 lines [not counting NLAYER and NWVL] in the icetable file, found " + std::to_string(rows));
    if (!cur.v.empty()) layers.push_back(cur);
    if (layers.empty()) throw Error(CLSIMHIP_ERR_IO, "At least one layers is requried (reading " + table_file + ")");
    auto field = [&](const Layer &l, const char *key) -> const std::vector<double> & {
        auto it = l.v.find(key);
        if (it == l.v.end()) throw Error(CLSIMHIP_ERR_IO, std::string("a layer has no ") + key + " entry (reading " + table_file + ")");
        return it->second;
    };
    for (const Layer &l : layers)
        if (field(l, "LAYER").size() < 2) throw Error(CLSIMHIP_ERR_IO, "LAYER needs two z coordinates (reading " + table_file + ")");
    const double height = std::fabs(field(layers[0], "LAYER")[1] - field(layers[0], "LAYER")[0]);
    // layers sorted by their bottom z; an upside-down layer is only compensated for in the key (py:136-153)
    std::map<double, const Layer *> by_z;
    for (const Layer &l : layers) {
        double bottom = field(l, "LAYER")[0], top = field(l, "LAYER")[1];
        if (bottom > top) std::swap(bottom, top);
        if (std::fabs((top - bottom) - height) > 0.0001) throw Error(CLSIMHIP_ERR_IO, "Differing layer heights while reading " + table_file);
        by_z[bottom] = &l;
    }
    std::vector<const Layer *> sorted;
    double end_z = std::numeric_limits<double>::quiet_NaN();
    for (const auto &kv : by_z) {
        const double start_z = field(*kv.second, "LAYER")[0];
        if (!std::isnan(end_z) && std::fabs(end_z - start_z) > 0.0001) throw Error(CLSIMHIP_ERR_IO, "Your layers have holes.");
        end_z = field(*kv.second, "LAYER")[1];
        sorted.push_back(kv.second);
    }
    const size_t nw = static_cast<size_t>(n_wlen);
    const double mean_cos = field(*sorted[0], "COS").empty() ? 0. : field(*sorted[0], "COS")[0];
    const std::vector<double> &group0 = field(*sorted[0], "N_GROUP"), &phase0 = field(*sorted[0], "N_PHASE");
    for (const Layer *l : sorted) {
        for (double c : field(*l, "COS"))
            if (std::fabs(c - mean_cos) > 0.0001) throw Error(CLSIMHIP_ERR_IO, "only a constant mean cosine is supported by clsim");
        for (const char *key : {"COS", "ABS", "SCAT", "N_GROUP", "N_PHASE"})
            if (field(*l, key).size() != nw)
                throw Error(CLSIMHIP_ERR_IO, "Expected " + std::to_string(nw) + " " + key + " values, got " + std::to_string(field(*l, key).size()));
        for (size_t i = 0; i < nw; ++i) {
            if (std::fabs(field(*l, "N_GROUP")[i] - group0[i]) > 0.0001)
                throw Error(CLSIMHIP_ERR_IO, "N_GROUP may not be different for different layers in this version of clsim!");
            if (std::fabs(field(*l, "N_PHASE")[i] - phase0[i]) > 0.0001)
                throw Error(CLSIMHIP_ERR_IO, "N_PHASE may not be different for different layers in this version of clsim!");
        }
    }

    MediumData m;
    m.num_layers = static_cast<int>(sorted.size());
    m.layers_z_start = field(*sorted[0], "LAYER")[0];
    m.layers_height = height;
    m.lengths_kind = CLSIMHIP_LENGTHS_TABLE;
    m.table_n = static_cast<int>(nw);
    m.table_start = start;
    m.table_step = step;
    m.table_16bit = true;                                           // storeDataAsHalfPrecision=True (py:214-219)
    m.abs_table.reserve(sorted.size() * nw);
    m.sca_table.reserve(sorted.size() * nw);
    for (const Layer *l : sorted) {
        for (double a : field(*l, "ABS")) m.abs_table.push_back(1. / a);
        for (double b : field(*l, "SCAT")) m.sca_table.push_back((1. / b) * (1. - mean_cos));
    }
    m.phase_kind = m.group_kind = CLSIMHIP_REFINDEX_TABLE;
    m.phase_table.kind = m.group_table.kind = CLSIMHIP_FUNCTION_TABLE;
    m.phase_table.start = m.group_table.start = start;
    m.phase_table.step = m.group_table.step = step;
    m.phase_table.values = phase0;
    m.group_table.values = group0;
    m.scatter_kind = CLSIMHIP_SCATTER_HG;
    m.mean_cosine = mean_cos;
    // nothing forced: the range is what the tabulated functions cover (MediumProperties.cxx:85-153)
    m.min_wlen = start;
    m.max_wlen = start + step * static_cast<double>(nw - 1);
    m.validate();
    return m;
}

// python/GetIceCubeDOMAcceptance.py:35-115
void dom_acceptance(double dom_radius, double efficiency, std::vector<double> &values, double &start, double &step)
{
    static const double eff_area[43] = {
        0.0000064522, 0.0000064522, 0.0000064522, 0.0000064522, 0.0000021980, 0.0001339040, 0.0005556810,
        0.0016953000, 0.0035997000, 0.0061340900, 0.0074592700, 0.0090579800, 0.0099246700, 0.0105769000,
        0.0110961000, 0.0114214000, 0.0114425000, 0.0111527000, 0.0108086000, 0.0104458000, 0.0099763100,
        0.0093102500, 0.0087516600, 0.0083225800, 0.0079767200, 0.0075625100, 0.0066377000, 0.0053335800,
        0.0043789400, 0.0037583500, 0.0033279800, 0.0029212500, 0.0025334900, 0.0021115400, 0.0017363300,
        0.0013552700, 0.0010546600, 0.0007201020, 0.0004843820, 0.0002911110, 0.0001782310, 0.0001144300,
        0.0000509155};
    const double dom_area = M_PI * std::pow(dom_radius, 2.);
    values.resize(43);
    for (int i = 0; i < 43; ++i) values[i] = efficiency * ((eff_area[i] * 1.0) / dom_area);
    start = 260. * units::nanometer;
    step = 10. * units::nanometer;
}

// I3CLSimModuleHelper.cxx:175-263 (tabulated bias, dispersion on) with the yield of :52-63
RandomValueData make_cherenkov_generator(const FunctionData &bias, const MediumData &m)
{
    if (bias.kind != CLSIMHIP_FUNCTION_TABLE)
        throw Error(CLSIMHIP_ERR_CONFIG, "makeCherenkovWavelengthGenerator: only tabulated biases are supported");
    RandomValueData g;
    g.kind = CLSIMHIP_RANDOM_INTERPOLATED;
    g.first = bias.start;
    g.spacing = bias.step;
    g.y.resize(bias.values.size());
    const double beta = 1.;
    for (size_t i = 0; i < bias.values.size(); ++i) {
        const double wlen = bias.start + static_cast<double>(i) * bias.step;
        const double n_phase = m.phase_ref_index(wlen);
        const double yield = (2. * M_PI / (137. * (wlen * wlen))) * (1. - 1. / (std::pow(beta * n_phase, 2.)));
        g.y[i] = bias.values[i] * yield;
    }
    return g;
}

// I3CLSimModuleHelper.cxx:73-171 for the spectrum classes its callers pass (GetIceCubeFlasherSpectrum.py: a FromTable with the
// LED's measured wavelengths, or a DeltaPeak for the standard candles)
RandomValueData make_wlen_generator(const FunctionData &spectrum, const FunctionData &bias, const MediumData &)
{
    RandomValueData g;
    if (spectrum.kind == CLSIMHIP_FUNCTION_DELTA_PEAK) {        // :78-89
        g.kind = CLSIMHIP_RANDOM_CONSTANT;
        g.value = spectrum.value;
        return g;
    }
    if (spectrum.kind != CLSIMHIP_FUNCTION_TABLE && spectrum.kind != CLSIMHIP_FUNCTION_TABLE_X)
        throw Error(CLSIMHIP_ERR_CONFIG, "makeWavelengthGenerator: the spectrum must be a table or a delta peak");
    if (!bias.on_device()) throw Error(CLSIMHIP_ERR_CONFIG, "makeWavelengthGenerator: the bias must be a table with equal spacing or a constant");
    const size_t n = spectrum.values.size();
    const bool own_x = (spectrum.kind == CLSIMHIP_FUNCTION_TABLE_X);
    // a tabulated spectrum keeps its whole range (no clipping to the medium's, :100-106); the bias has to cover it (:111-114)
    const double min_wlen = own_x ? spectrum.wavelengths.front() : spectrum.start;
    const double max_wlen = own_x ? spectrum.wavelengths.back() : spectrum.start + spectrum.step * static_cast<double>(n - 1);
    if (max_wlen - min_wlen <= 0.) throw Error(CLSIMHIP_ERR_CONFIG, "Internal error, wavelength range <= 0!");
    if (bias.kind == CLSIMHIP_FUNCTION_TABLE) {
        const double bias_min = bias.start, bias_max = bias.start + bias.step * static_cast<double>(bias.values.size() - 1);
        if (bias_min > min_wlen || bias_max < max_wlen)
            throw Error(CLSIMHIP_ERR_CONFIG, "wavelength generation bias has to have a wavelength range larger or equal to the spectrum wavelength range!");
    }
    g.y.resize(n);
    for (size_t i = 0; i < n; ++i) {                            // :120-133
        const double wavelength = own_x ? spectrum.wavelengths[i] : spectrum.start + static_cast<double>(i) * spectrum.step;
        g.y[i] = bias.eval(wavelength) * spectrum.values[i];
    }
    if (own_x) { g.kind = CLSIMHIP_RANDOM_INTERPOLATED_X; g.x = spectrum.wavelengths; }
    else { g.kind = CLSIMHIP_RANDOM_INTERPOLATED; g.first = spectrum.start; g.spacing = spectrum.step; }
    return g;
}

} // namespace clsimhip
