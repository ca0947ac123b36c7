// extern "C" surface of libclsimhip.so (include/clsimhip.h).  Every entry point
// catches clsimhip::Error, stores the text for clsimhip_last_error() and returns
// the status code; nothing C++ crosses the boundary.
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "converter.h"
#include "lightsource.h"
#include "feeder.h"
#include "tabulator.h"
#include "step_store.h"
#include "flasher.h"

using namespace clsimhip;

struct clsimhip_converter { Converter impl; explicit clsimhip_converter(int dev) : impl(dev) {} };
struct clsimhip_medium { MediumData data; };
struct clsimhip_tabulator {
    std::unique_ptr<Tabulator> impl;
};

namespace {
// One message per calling thread (like errno): EnqueueSteps and GetConversionResult are called concurrently from
// several threads per converter (I3CLSimServer.cxx:126-135, 324-331), so a per-object string would be a data race.
thread_local std::string g_last_error;

// device scratch memory of one call, freed on every path out of it
struct DeviceBuffer {
    void *p = nullptr;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { if (p) (void)hipFree(p); }
    void alloc(size_t bytes, const char *what)
    {
        const hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) { p = nullptr; throw Error(CLSIMHIP_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); }
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

template <class F>
int guarded(clsimhip_converter *c, F &&f)
{
    try {
        f();
        return CLSIMHIP_OK;
    } catch (const Error &e) {
        (void)c;
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return CLSIMHIP_ERR_ARGUMENT;
    }
}
template <class F>
int guarded_const(const clsimhip_converter *c, F &&f) { return guarded(const_cast<clsimhip_converter *>(c), f); }

void need(const void *p, const char *what) { if (!p) throw Error(CLSIMHIP_ERR_ARGUMENT, std::string(what) + " is (null)"); }

FunctionData function_from(const clsimhip_function *f)
{
    need(f, "function");
    FunctionData d;
    d.kind = f->kind;
    if (f->kind == CLSIMHIP_FUNCTION_TABLE) {
        if (f->n < 2 || !f->values) throw Error(CLSIMHIP_ERR_ARGUMENT, "values must contain at least 2 elements!");
        d.start = f->start; d.step = f->step;
        d.values.assign(f->values, f->values + f->n);
    } else if (f->kind == CLSIMHIP_FUNCTION_CONSTANT || f->kind == CLSIMHIP_FUNCTION_DELTA_PEAK) {
        d.value = f->value;
    } else if (f->kind == CLSIMHIP_FUNCTION_TABLE_X) {              // FromTable.cxx:57-70
        if (f->n < 2 || !f->wavelengths) throw Error(CLSIMHIP_ERR_ARGUMENT, "wlens must contain at least 2 elements!");
        if (!f->values) throw Error(CLSIMHIP_ERR_ARGUMENT, "wlens and values must have the same size!");
        d.wavelengths.assign(f->wavelengths, f->wavelengths + f->n);
        d.values.assign(f->values, f->values + f->n);
    } else
        throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown function kind");
    return d;
}
} // namespace

namespace {
template <class F>
int guarded_tab(clsimhip_tabulator *t, F &&f)
{
    try {
        f();
        return CLSIMHIP_OK;
    } catch (const Error &e) {
        (void)t;
        g_last_error = e.what();
        return e.code;
    } catch (const std::exception &e) {
        g_last_error = e.what();
        return CLSIMHIP_ERR_ARGUMENT;
    }
}
} // namespace

extern "C" {

const char *clsimhip_version(void) { return "clsimhip 0.1 (gfx950)"; }

const char *clsimhip_last_error(const clsimhip_converter *c) { (void)c; return g_last_error.c_str(); }

int clsimhip_medium_create(const clsimhip_medium_desc *desc, clsimhip_medium **out)
{
    return guarded(nullptr, [&] {
        need(desc, "desc"); need(out, "out");
        *out = new clsimhip_medium{medium_from_desc(*desc)};
    });
}
int clsimhip_medium_create_from_ppc(const char *directory, double depth, int use_tilt, clsimhip_medium **out)
{
    return guarded(nullptr, [&] {
        need(directory, "directory"); need(out, "out");
        *out = new clsimhip_medium{medium_from_ppc(directory, depth, use_tilt != 0)};
    });
}
int clsimhip_medium_create_from_photonics(const char *table_file, double depth, clsimhip_medium **out)
{
    return guarded(nullptr, [&] {
        need(table_file, "table_file"); need(out, "out");
        *out = new clsimhip_medium{medium_from_photonics(table_file, depth)};
    });
}
int clsimhip_medium_describe(const clsimhip_medium *m, clsimhip_medium_desc *d)
{
    return guarded(nullptr, [&] {
        need(m, "medium"); need(d, "out");
        const MediumData &s = m->data;
        std::memset(d, 0, sizeof *d);
        d->num_layers = s.num_layers; d->layers_z_start = s.layers_z_start; d->layers_height = s.layers_height;
        d->min_wavelength = s.min_wlen; d->max_wavelength = s.max_wlen; d->lengths_kind = s.lengths_kind;
        d->abs_length = s.abs_length.data(); d->sca_length = s.sca_length.data();
        d->alpha = s.alpha; d->kappa = s.kappa; d->A = s.A; d->B = s.B; d->D = s.D; d->E = s.E;
        d->a_dust400 = s.a_dust400.data(); d->delta_tau = s.delta_tau.data(); d->b400 = s.b400.data();
        for (int i = 0; i < 5; ++i) { d->n[i] = s.n[i]; d->g[i] = s.g[i]; }
        d->scatter_kind = s.scatter_kind; d->liu_fraction = s.liu_fraction; d->mean_cosine = s.mean_cosine;
        d->has_anisotropy = s.has_aniso; d->aniso_azimuth = s.aniso_azimuth; d->aniso_k1 = s.aniso_k1; d->aniso_k2 = s.aniso_k2;
        d->has_pre_transform = s.has_pre; d->pre_renormalize = s.pre_renorm;
        d->has_post_transform = s.has_post; d->post_renormalize = s.post_renorm;
        for (int i = 0; i < 9; ++i) { d->pre_matrix[i] = s.pre[i]; d->post_matrix[i] = s.post[i]; }
        d->has_tilt = s.has_tilt;
        d->tilt_num_distances = static_cast<int32_t>(s.tilt_distances.size());
        d->tilt_num_z = static_cast<int32_t>(s.tilt_z.size());
        d->tilt_distances = s.tilt_distances.data(); d->tilt_z_coordinates = s.tilt_z.data();
        d->tilt_z_corrections = s.tilt_corr.data(); d->tilt_azimuth = s.tilt_azimuth;
        d->table_num_wavelengths = s.table_n; d->table_start_wavelength = s.table_start; d->table_wavelength_step = s.table_step;
        d->table_store_as_16bit = s.table_16bit;
        d->abs_length_table = s.abs_table.data(); d->sca_length_table = s.sca_table.data();
        d->phase_index_kind = s.phase_kind; d->group_index_kind = s.group_kind;
        auto view = [](const FunctionData &f, clsimhip_function &o) {
            o.kind = f.kind; o.n = static_cast<int32_t>(f.values.size()); o.start = f.start; o.step = f.step;
            o.values = f.values.data(); o.value = f.value;
        };
        view(s.phase_table, d->phase_index_table);
        view(s.group_table, d->group_index_table);
    });
}
void clsimhip_medium_destroy(clsimhip_medium *m) { delete m; }

int clsimhip_icecube_dom_acceptance(double dom_radius, double efficiency, double *values_out, double *start_out, double *step_out)
{
    return guarded(nullptr, [&] {
        need(values_out, "values_out");
        std::vector<double> v; double start, step;
        dom_acceptance(dom_radius, efficiency, v, start, step);
        std::memcpy(values_out, v.data(), v.size() * sizeof(double));
        if (start_out) *start_out = start;
        if (step_out) *step_out = step;
    });
}
int clsimhip_make_cherenkov_wlen_generator(const clsimhip_function *bias, const clsimhip_medium *m, double *y_out,
                                           double *first_out, double *spacing_out)
{
    return guarded(nullptr, [&] {
        need(m, "medium"); need(y_out, "y_out");
        const RandomValueData g = make_cherenkov_generator(function_from(bias), m->data);
        std::memcpy(y_out, g.y.data(), g.y.size() * sizeof(double));
        if (first_out) *first_out = g.first;
        if (spacing_out) *spacing_out = g.spacing;
    });
}
int clsimhip_make_wlen_generator(const clsimhip_function *spectrum, const clsimhip_function *bias, const clsimhip_medium *m,
                                 clsimhip_random_value *out, double *x_out, double *y_out, size_t capacity)
{
    return guarded(nullptr, [&] {
        need(m, "medium"); need(out, "out"); need(y_out, "y_out"); need(x_out, "x_out");
        const RandomValueData g = make_wlen_generator(function_from(spectrum), function_from(bias), m->data);
        if (g.y.size() > capacity) throw Error(CLSIMHIP_ERR_ARGUMENT, "clsimhip_make_wlen_generator: the output arrays are too small");
        std::memset(out, 0, sizeof *out);
        out->kind = g.kind; out->n = static_cast<int32_t>(g.y.size());
        out->first = g.first; out->spacing = g.spacing; out->value = g.value;
        std::memcpy(y_out, g.y.data(), g.y.size() * sizeof(double));
        std::memcpy(x_out, g.x.data(), g.x.size() * sizeof(double));
        out->y = g.y.empty() ? nullptr : y_out;
        out->x = g.x.empty() ? nullptr : x_out;
    });
}
struct clsimhip_ppc_converter { std::unique_ptr<clsimhip::PPCConverter> impl; };
int clsimhip_ppc_create(const clsimhip_medium *medium, const clsimhip_function *wavelength_bias, const clsimhip_ppc_config *config,
                        clsimhip_ppc_converter **out)
{
    return guarded(nullptr, [&] {
        need(medium, "medium"); need(wavelength_bias, "wavelength_bias"); need(out, "out");
        PPCConfig c;
        if (config) {
            c.photons_per_step = config->photons_per_step; c.high_photons_per_step = config->high_photons_per_step;
            c.use_high_photons_per_step_from = config->use_high_photons_per_step_from;
            c.use_cascade_extension = config->use_cascade_extension != 0;
            c.density = config->medium_density; c.seed = config->seed;
        }
        std::unique_ptr<clsimhip_ppc_converter> p(new clsimhip_ppc_converter);
        p->impl.reset(new PPCConverter(medium->data, function_from(wavelength_bias), c));
        *out = p.release();
    });
}
void clsimhip_ppc_destroy(clsimhip_ppc_converter *p) { delete p; }
int clsimhip_ppc_photons_per_meter(const clsimhip_ppc_converter *p, int layer, double *out)
{
    return guarded(nullptr, [&] { need(p, "converter"); need(out, "out"); *out = p->impl->mean_photons_per_meter(layer); });
}
int clsimhip_ppc_enqueue(const clsimhip_ppc_converter *p, const clsimhip_particle *particles, size_t n,
                         clsimhip_step_request *requests_out, size_t capacity, size_t *n_out)
{
    return guarded(nullptr, [&] {
        need(p, "converter"); need(n_out, "n_out");
        if (n) need(particles, "particles");
        std::vector<clsimhip_step_request> v;
        v.reserve(2 * n);
        for (size_t i = 0; i < n; ++i) p->impl->enqueue(particles[i], v);
        *n_out = v.size();
        if (requests_out) std::copy(v.begin(), v.begin() + static_cast<std::ptrdiff_t>(std::min(capacity, v.size())), requests_out);
    });
}
int clsimhip_flasher_correction_factor(const clsimhip_function *spectrum_no_bias, double peak_wavelength,
                                       const clsimhip_function *wavelength_bias, double from_wavelength, double to_wavelength, double *out)
{
    return guarded(nullptr, [&] {
        need(wavelength_bias, "wavelength_bias"); need(out, "out");
        const FunctionData bias = function_from(wavelength_bias);
        if (spectrum_no_bias) {
            const FunctionData spectrum = function_from(spectrum_no_bias);
            *out = flasher_correction_factor(&spectrum, 0., bias, from_wavelength, to_wavelength);
        } else {
            *out = flasher_correction_factor(nullptr, peak_wavelength, bias, from_wavelength, to_wavelength);
        }
    });
}
int clsimhip_flasher_enqueue(double correction_factor, uint64_t seed, const clsimhip_flasher_pulse *pulses, size_t n,
                             clsimhip_flasher_request *requests_out, size_t capacity, size_t *n_out)
{
    return guarded(nullptr, [&] {
        need(n_out, "n_out");
        if (n) need(pulses, "pulses");
        std::vector<clsimhip_flasher_request> v;
        flasher_enqueue(correction_factor, seed, pulses, n, v);
        *n_out = v.size();
        if (requests_out) std::copy(v.begin(), v.begin() + static_cast<std::ptrdiff_t>(std::min(capacity, v.size())), requests_out);
    });
}
int clsimhip_shower_parameters(int32_t particle_type, double energy_gev, double density_g_cm3, double out[4])
{
    return guarded(nullptr, [&] {
        need(out, "out");
        const ShowerParameters s = shower_parameters(particle_type, energy_gev, density_g_cm3);
        out[0] = s.a; out[1] = s.b; out[2] = s.em_scale; out[3] = s.em_scale_sigma;
    });
}
struct clsimhip_feeder {
    std::unique_ptr<clsimhip::Feeder> impl;
    std::mutex m;
    std::map<const clsimhip_step *, clsimhip::Feeder::Result> handed_out;
};
static const clsimhip_step g_empty_bunch{};
int clsimhip_feeder_create(const clsimhip_ppc_converter *ppc, int device, uint64_t seed, size_t max_bunch_size,
                           size_t bunch_size_granularity, size_t queue_depth, clsimhip_feeder **out)
{
    return guarded(nullptr, [&] {
        need(out, "out");
        std::unique_ptr<clsimhip_feeder> f(new clsimhip_feeder);
        f->impl.reset(new Feeder(ppc ? ppc->impl.get() : nullptr, device, seed, max_bunch_size, bunch_size_granularity, queue_depth));
        *out = f.release();
    });
}
void clsimhip_feeder_destroy(clsimhip_feeder *f) { delete f; }
int clsimhip_feeder_enqueue_light_source(clsimhip_feeder *f, const clsimhip_particle *particle)
{
    return guarded(nullptr, [&] { need(f, "feeder"); need(particle, "particle"); f->impl->enqueue_light_source(*particle); });
}
int clsimhip_feeder_enqueue_steps(clsimhip_feeder *f, uint32_t identifier, const clsimhip_step *steps, size_t n)
{
    return guarded(nullptr, [&] { need(f, "feeder"); if (n) need(steps, "steps"); f->impl->enqueue_steps(identifier, steps, n); });
}
int clsimhip_feeder_enqueue_barrier(clsimhip_feeder *f)
{
    return guarded(nullptr, [&] { need(f, "feeder"); f->impl->enqueue_barrier(); });
}
int clsimhip_feeder_barrier_active(const clsimhip_feeder *f, int *out)
{
    return guarded(nullptr, [&] { need(f, "feeder"); need(out, "out"); *out = f->impl->barrier_active() ? 1 : 0; });
}
int clsimhip_feeder_more_steps_available(const clsimhip_feeder *f, int *out)
{
    return guarded(nullptr, [&] { need(f, "feeder"); need(out, "out"); *out = f->impl->more_steps_available() ? 1 : 0; });
}
int clsimhip_feeder_get_conversion_result(clsimhip_feeder *f, double timeout_ms, int *got, const clsimhip_step **steps, size_t *n,
                                          const uint32_t **finished, size_t *n_finished, int *barrier_was_reset)
{
    return guarded(nullptr, [&] {
        need(f, "feeder"); need(got, "got"); need(steps, "steps"); need(n, "n"); need(finished, "finished"); need(n_finished, "n_finished");
        need(barrier_was_reset, "barrier_was_reset");
        Feeder::Result r;
        *got = 0; *steps = nullptr; *n = 0; *finished = nullptr; *n_finished = 0; *barrier_was_reset = 0;
        if (!f->impl->get_result(timeout_ms, r)) return;
        *got = 1;
        *n = r.steps->size();
        *n_finished = r.finished.size();
        *barrier_was_reset = r.last_before_barrier ? 1 : 0;
        std::lock_guard<std::mutex> lk(f->m);
        // the key is the step array itself; an empty bunch gets an address of its own
        if (r.steps->empty()) r.steps->push_back(g_empty_bunch);
        const clsimhip_step *key = r.steps->data();
        Feeder::Result &kept = f->handed_out[key];
        kept = std::move(r);
        *steps = key;
        *finished = kept.finished.empty() ? nullptr : kept.finished.data();
    });
}
int clsimhip_feeder_release_result(clsimhip_feeder *f, const clsimhip_step *steps)
{
    return guarded(nullptr, [&] { need(f, "feeder"); std::lock_guard<std::mutex> lk(f->m); f->handed_out.erase(steps); });
}
int clsimhip_mwc_multipliers(uint32_t *a_out, size_t count)
{
    return guarded(nullptr, [&] { need(a_out, "a_out"); mwc_multipliers(a_out, count); });
}
int clsimhip_mwc_multipliers_from_file(const char *path, uint32_t *a_out, size_t count)
{
    return guarded(nullptr, [&] {
        need(path, "path"); need(a_out, "a_out");
        if (!load_multipliers_from_file(path, a_out, count)) throw Error(CLSIMHIP_ERR_IO, std::string("could not read ") + std::to_string(count) + " multipliers from " + path);
    });
}
int clsimhip_seed_streams(const uint32_t *a, size_t count, uint64_t seed, uint64_t *x_out)
{
    return guarded(nullptr, [&] { need(a, "a"); need(x_out, "x_out"); seed_streams(a, count, seed, x_out); });
}

int clsimhip_create(int device_ordinal, clsimhip_converter **out)
{
    return guarded(nullptr, [&] { need(out, "out"); *out = new clsimhip_converter(device_ordinal); });
}
void clsimhip_destroy(clsimhip_converter *c) { delete c; }
int clsimhip_set_device(clsimhip_converter *c, int device_ordinal)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.set_device(device_ordinal); });
}
int clsimhip_set_concurrent_device_launches(clsimhip_converter *c, int k)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.set_concurrent_device_launches(k); });
}
int clsimhip_step_series_blob_size(size_t n, size_t *bytes)
{
    return guarded(nullptr, [&] { need(bytes, "bytes"); *bytes = series_blob_size(n, sizeof(clsimhip_step)); });
}
int clsimhip_encode_step_series(const clsimhip_step *steps, size_t n, uint8_t *out, size_t capacity, size_t *written)
{
    return guarded(nullptr, [&] { series_encode(steps, n, sizeof(clsimhip_step), 0u, out, capacity, written); });
}
int clsimhip_decode_step_series(const uint8_t *blob, size_t bytes, clsimhip_step *steps_out, size_t capacity, size_t *n, size_t *consumed)
{
    return guarded(nullptr, [&] { series_decode(blob, bytes, sizeof(clsimhip_step), 0u, "I3CLSimStep", steps_out, capacity, n, consumed); });
}
int clsimhip_photon_series_blob_size(size_t n, size_t *bytes)
{
    return guarded(nullptr, [&] { need(bytes, "bytes"); *bytes = series_blob_size(n, sizeof(clsimhip_photon)); });
}
int clsimhip_encode_photon_series(const clsimhip_photon *photons, size_t n, uint8_t *out, size_t capacity, size_t *written)
{
    return guarded(nullptr, [&] { series_encode(photons, n, sizeof(clsimhip_photon), 0u, out, capacity, written); });
}
int clsimhip_decode_photon_series(const uint8_t *blob, size_t bytes, clsimhip_photon *photons_out, size_t capacity, size_t *n, size_t *consumed)
{
    return guarded(nullptr, [&] { series_decode(blob, bytes, sizeof(clsimhip_photon), 0u, "I3CLSimPhoton", photons_out, capacity, n, consumed); });
}
int clsimhip_encode_portable_uint(uint64_t value, uint8_t out[9], size_t *written)
{
    return guarded(nullptr, [&] { need(out, "out"); need(written, "written"); *written = portable_uint_encode(value, out); });
}
int clsimhip_comm_get_unique_id(uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES])
{
    return guarded(nullptr, [&] { need(id, "id"); comm_unique_id(id); });
}
int clsimhip_comm_create(int device_ordinal, int rank, int world_size, const uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES], clsimhip_comm **out)
{
    return guarded(nullptr, [&] { need(out, "out"); *out = reinterpret_cast<clsimhip_comm *>(comm_create(device_ordinal, rank, world_size, id)); });
}
void clsimhip_comm_destroy(clsimhip_comm *comm) { comm_destroy(reinterpret_cast<Comm *>(comm)); }
int clsimhip_comm_info(clsimhip_comm *comm, int *rccl_ranks, int *rccl_rank, int *device_ordinal, char *pci_bus_id, size_t pci_bus_id_bytes)
{
    return guarded(nullptr, [&] { comm_info(reinterpret_cast<Comm *>(comm), rccl_ranks, rccl_rank, device_ordinal, pci_bus_id, pci_bus_id_bytes); });
}
int clsimhip_comm_statistics(clsimhip_comm *comm, uint64_t *gathers, double *gather_ms, uint64_t *records_sent, uint64_t *records_received, int reset)
{
    return guarded(nullptr, [&] { comm_statistics(reinterpret_cast<Comm *>(comm), gathers, gather_ms, records_sent, records_received, reset != 0); });
}
int clsimhip_gather_hits(clsimhip_comm *comm, const void *d_photons, const void *d_hit_count, size_t capacity, int root,
                         void *d_gathered, size_t gathered_capacity, uint64_t *counts_out, void *hip_stream)
{
    return guarded(nullptr, [&] {
        comm_gather_hits(reinterpret_cast<Comm *>(comm), d_photons, d_hit_count, capacity, root, d_gathered, gathered_capacity, counts_out,
                         static_cast<hipStream_t>(hip_stream));
    });
}
int clsimhip_uses_pooled_kernel(const clsimhip_converter *c, int *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.uses_pooled_kernel() ? 1 : 0; });
}
int clsimhip_kernel_for_bunch(const clsimhip_converter *c, size_t n_steps, int *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.pooled_for(n_steps) ? 1 : 0; });
}
int clsimhip_get_device(const clsimhip_converter *c, int *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.device(); });
}

int clsimhip_set_wlen_generators(clsimhip_converter *c, const clsimhip_random_value *gens, size_t n)
{
    return guarded(c, [&] {
        need(c, "converter"); need(gens, "generators");
        std::vector<RandomValueData> v(n);
        for (size_t i = 0; i < n; ++i) {
            v[i].kind = gens[i].kind;
            if (gens[i].kind == CLSIMHIP_RANDOM_INTERPOLATED) {
                if (gens[i].n < 2 || !gens[i].y) throw Error(CLSIMHIP_ERR_ARGUMENT, "At least two entries have to be specified in the vector passed to I3CLSimRandomValueInterpolatedDistribution().");
                v[i].first = gens[i].first; v[i].spacing = gens[i].spacing;
                v[i].y.assign(gens[i].y, gens[i].y + gens[i].n);
            } else if (gens[i].kind == CLSIMHIP_RANDOM_INTERPOLATED_X) {       // InterpolatedDistribution.cxx:40-55
                if (gens[i].n < 2 || !gens[i].y || !gens[i].x) throw Error(CLSIMHIP_ERR_ARGUMENT, "At least two entries have to be specified in the vectors passed to I3CLSimRandomValueInterpolatedDistribution().");
                v[i].x.assign(gens[i].x, gens[i].x + gens[i].n);
                v[i].y.assign(gens[i].y, gens[i].y + gens[i].n);
            } else if (gens[i].kind == CLSIMHIP_RANDOM_CONSTANT) {
                v[i].value = gens[i].value;
            } else if (gens[i].kind == CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION) {
                v[i].first = gens[i].first; v[i].spacing = gens[i].spacing;       // fromWlen, toWlen
            } else
                throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown random value kind");
        }
        c->impl.set_wlen_generators(std::move(v));
    });
}
int clsimhip_set_wlen_bias(clsimhip_converter *c, const clsimhip_function *bias)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.set_wlen_bias(function_from(bias)); });
}
int clsimhip_set_medium_properties(clsimhip_converter *c, const clsimhip_medium *m)
{
    return guarded(c, [&] { need(c, "converter"); need(m, "medium"); c->impl.set_medium(m->data); });
}
int clsimhip_set_geometry(clsimhip_converter *c, size_t n, const int32_t *string_ids, const uint32_t *dom_ids, const double *x,
                          const double *y, const double *z, const char *const *subdetectors, double om_radius)
{
    return guarded(c, [&] {
        need(c, "converter"); need(string_ids, "string_ids"); need(dom_ids, "dom_ids"); need(x, "x"); need(y, "y"); need(z, "z");
        need(subdetectors, "subdetectors");
        GeometryInput g;
        g.string_ids.assign(string_ids, string_ids + n); g.dom_ids.assign(dom_ids, dom_ids + n);
        g.x.assign(x, x + n); g.y.assign(y, y + n); g.z.assign(z, z + n);
        g.subdetectors.resize(n);
        for (size_t i = 0; i < n; ++i) { need(subdetectors[i], "subdetector name"); g.subdetectors[i] = subdetectors[i]; }
        g.om_radius = om_radius;
        c->impl.set_geometry(std::move(g));
    });
}
int clsimhip_set_geometry_from_text_file(clsimhip_converter *c, const char *filename, double om_radius, int32_t string_id_min,
                                         int32_t string_id_max, uint32_t dom_id_min, uint32_t dom_id_max)
{
    return guarded(c, [&] {
        need(c, "converter"); need(filename, "filename");
        c->impl.set_geometry(geometry_from_text_file(filename, om_radius, string_id_min, string_id_max, dom_id_min, dom_id_max));
    });
}
#define SETTER(name, type, call) \
    int name(clsimhip_converter *c, type value) { return guarded(c, [&] { need(c, "converter"); c->impl.call; }); }
SETTER(clsimhip_set_enable_double_buffering, int, set_double_buffering(value != 0))
SETTER(clsimhip_set_double_precision, int, set_double_precision(value != 0))
SETTER(clsimhip_set_stop_detected_photons, int, set_stop_detected(value != 0))
SETTER(clsimhip_set_save_all_photons, int, set_save_all(value != 0))
SETTER(clsimhip_set_save_all_photons_prescale, double, set_save_all_prescale(value))
SETTER(clsimhip_set_fixed_number_of_absorption_lengths, double, set_fixed_abs_lengths(value))
SETTER(clsimhip_set_dom_pancake_factor, double, set_pancake(value))
SETTER(clsimhip_set_photon_history_entries, uint32_t, set_history_entries(value))
SETTER(clsimhip_set_workgroup_size, size_t, set_workgroup_size(value))
SETTER(clsimhip_set_max_num_workitems, size_t, set_max_num_workitems(value))
#undef SETTER

int clsimhip_compile(clsimhip_converter *c) { return guarded(c, [&] { need(c, "converter"); c->impl.compile(); }); }
int clsimhip_get_max_workgroup_size(const clsimhip_converter *c, size_t *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.max_workgroup_size(); });
}
int clsimhip_initialize(clsimhip_converter *c, uint64_t seed) { return guarded(c, [&] { need(c, "converter"); c->impl.initialize(seed); }); }
int clsimhip_initialize_with_streams(clsimhip_converter *c, const uint64_t *x, const uint32_t *a, size_t count)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.initialize_with_streams(x, a, count); });
}
int clsimhip_is_initialized(const clsimhip_converter *c) { return (c && c->impl.initialized()) ? 1 : 0; }

int clsimhip_enqueue_steps(clsimhip_converter *c, const clsimhip_step *steps, size_t n, uint32_t identifier)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.enqueue_steps(steps, n, identifier); });
}
int clsimhip_get_conversion_result(clsimhip_converter *c, uint32_t *identifier, const clsimhip_photon **photons, size_t *n)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.get_result(identifier, photons, n); });
}
int clsimhip_get_result_histories(clsimhip_converter *c, const clsimhip_photon *photons, const float **histories, uint32_t *entries)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.result_histories(photons, histories, entries); });
}
int clsimhip_release_result(clsimhip_converter *c, const clsimhip_photon *photons)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.release_result(photons); });
}
int clsimhip_get_workgroup_size(const clsimhip_converter *c, size_t *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.workgroup_size(); });
}
int clsimhip_get_max_num_workitems(const clsimhip_converter *c, size_t *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.max_num_workitems(); });
}
int clsimhip_queue_size(const clsimhip_converter *c, size_t *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.queue_size(); });
}
int clsimhip_more_photons_available(const clsimhip_converter *c, int *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.more_photons_available() ? 1 : 0; });
}
int clsimhip_get_statistics(const clsimhip_converter *c, double out[8])
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); c->impl.statistics(out); });
}
int clsimhip_get_option(const clsimhip_converter *c, int option, double *out)
{
    return guarded_const(c, [&] { need(c, "converter"); need(out, "out"); *out = c->impl.option(option); });
}
int clsimhip_set_tuning(clsimhip_converter *c, const char *key, long long value)
{
    return guarded(c, [&] { need(c, "converter"); need(key, "key"); c->impl.set_tuning(key, value); });
}
int clsimhip_get_tuning(const clsimhip_converter *c, const char *key, long long *value)
{
    return guarded_const(c, [&] { need(c, "converter"); need(key, "key"); need(value, "value"); *value = c->impl.get_tuning(key); });
}
int clsimhip_propagate_device(clsimhip_converter *c, const void *d_steps, size_t n, size_t rng_offset, void *d_photons,
                              size_t capacity, void *d_hit_count, void *stream)
{
    return guarded(c, [&] {
        need(c, "converter");
        c->impl.propagate_device(d_steps, n, rng_offset, d_photons, capacity, d_hit_count, static_cast<hipStream_t>(stream));
    });
}
int clsimhip_replace_indices_with_ids(const clsimhip_converter *c, clsimhip_photon *photons, size_t n)
{
    return guarded_const(c, [&] { need(c, "converter"); if (n) need(photons, "photons"); c->impl.replace_indices(photons, n); });
}
int clsimhip_kernel_time_ms(clsimhip_converter *c, int reset, double *total_ms, uint64_t *launches)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.kernel_time(reset != 0, total_ms, launches); });
}
long clsimhip_get_table(const clsimhip_converter *c, const char *name, double *out, size_t cap)
{
    long n = 0;
    const int rc = guarded_const(c, [&] { need(c, "converter"); need(name, "name"); n = c->impl.get_table(name, out, cap); });
    return rc == CLSIMHIP_OK ? n : rc;
}
int clsimhip_get_rng_state(clsimhip_converter *c, uint64_t *x_out, size_t count)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.get_rng_state(x_out, count); });
}
// not part of the public header: work-queue counters of the last launch (tools/ only)
int clsimhip_debug_counters(clsimhip_converter *c, uint32_t out[4])
{
    return guarded(c, [&] { need(c, "converter"); c->impl.debug_counters(out); });
}
int clsimhip_eval_device_function(clsimhip_converter *c, int what, int layer, int fast, const float *in4, size_t n, float *out4)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.eval_device_function(what, layer, fast != 0, in4, n, out4); });
}
int clsimhip_eval_device_random(clsimhip_converter *c, int what, int generator, int fast, uint64_t *x, const uint32_t *a, size_t n_streams, size_t draws,
                                float *out)
{
    return guarded(c, [&] { need(c, "converter"); c->impl.eval_device_random(what, generator, fast != 0, x, a, n_streams, draws, out); });
}
int clsimhip_eval_math(int device_ordinal, int what, const float *x, const float *y, size_t n, float *out)
{
    return guarded(nullptr, [&] {
        need(x, "x"); need(out, "out");
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available");
        DeviceGuard on_device(device_ordinal);
        DeviceBuffer bx, by, bout;
        bx.alloc(n * 4 + 16, "hipMalloc"); bout.alloc(n * 4 + 16, "hipMalloc");
        float *dx = bx.as<float>(), *dy = nullptr, *dout = bout.as<float>();
        chk(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice), "hipMemcpy");
        if (y) {
            by.alloc(n * 4 + 16, "hipMalloc");
            dy = by.as<float>();
            chk(hipMemcpy(dy, y, n * 4, hipMemcpyHostToDevice), "hipMemcpy");
        }
        chk(launch_eval_math(what, dx, dy, static_cast<uint32_t>(n), dout, nullptr), "eval_math launch");
        chk(hipDeviceSynchronize(), "eval_math");
        chk(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    });
}
int clsimhip_check_math_exhaustive(int device_ordinal, int what, int exp_lo, int exp_hi, uint32_t *result, size_t result_cap)
{
    return guarded(nullptr, [&] {
        need(result, "result");
        if (!((what >= 11 && what <= 13) || (what >= 16 && what <= 19)) || exp_lo < -126 || exp_hi > 127 || exp_lo > exp_hi || result_cap < 1 || result_cap > 4096)
            throw Error(CLSIMHIP_ERR_CONFIG, "clsimhip_check_math_exhaustive: what in 11..13 or 16..19, -126 <= exp_lo <= exp_hi <= 127, 1 <= result_cap <= 4096");
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available");
        DeviceGuard on_device(device_ordinal);
        DeviceBuffer buf;
        buf.alloc(result_cap * 4, "hipMalloc");
        chk(hipMemset(buf.as<uint32_t>(), 0, result_cap * 4), "hipMemset");
        chk(launch_check_math(what, exp_lo, exp_hi, buf.as<uint32_t>(), static_cast<uint32_t>(result_cap), nullptr), "check_math launch");
        chk(hipDeviceSynchronize(), "check_math");
        chk(hipMemcpy(result, buf.as<uint32_t>(), result_cap * 4, hipMemcpyDeviceToHost), "hipMemcpy");
    });
}


// ---- step producer ----
namespace {
void plan_steps(const clsimhip_step_request *requests, size_t n, size_t granularity, std::vector<uint64_t> &first, uint64_t &real, uint64_t &padded)
{
    need(requests, "requests");
    if (granularity == 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "granularity must not be 0");
    first.assign(n + 1, 0);
    for (size_t i = 0; i < n; ++i) {
        const clsimhip_step_request &q = requests[i];
        if (q.kind > CLSIMHIP_STEPS_MUON) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown step request kind");
        if (q.photons_per_step == 0 && q.num_steps > 0) throw Error(CLSIMHIP_ERR_ARGUMENT, "photonsPerStep may not be <= 0!");
        if (q.kind == CLSIMHIP_STEPS_CASCADE && !(q.pa > 0.f)) throw Error(CLSIMHIP_ERR_ARGUMENT, "cascade shape parameter must be positive");
        first[i + 1] = first[i] + q.num_steps + (q.num_photons_in_last_step > 0 ? 1 : 0);
    }
    real = first[n];
    padded = ((real + granularity - 1) / granularity) * granularity;
}
} // namespace

extern "C" {
int clsimhip_count_generated_steps(const clsimhip_step_request *requests, size_t n, size_t granularity, size_t *steps_out, size_t *padded_out)
{
    return guarded(nullptr, [&] {
        std::vector<uint64_t> first; uint64_t real = 0, padded = 0;
        plan_steps(requests, n, granularity, first, real, padded);
        if (steps_out) *steps_out = static_cast<size_t>(real);
        if (padded_out) *padded_out = static_cast<size_t>(padded);
    });
}
int clsimhip_generate_steps_device(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed, size_t granularity,
                                   void *d_steps, size_t capacity, void *hip_stream, size_t *padded_out)
{
    return guarded(nullptr, [&] {
        need(d_steps, "d_steps");
        std::vector<uint64_t> first; uint64_t real = 0, padded = 0;
        plan_steps(requests, n, granularity, first, real, padded);
        if (padded > capacity) throw Error(CLSIMHIP_ERR_ARGUMENT, "the requests produce more steps than the buffer holds");
        if (padded_out) *padded_out = static_cast<size_t>(padded);
        if (padded == 0) return;
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the step producer has no CPU fallback)");
        DeviceGuard on_device(device);
        hipStream_t stream = static_cast<hipStream_t>(hip_stream);
        const size_t nreq = n ? n : 1;
        DeviceBuffer b_req, b_first;
        b_req.alloc(nreq * sizeof(clsimhip_step_request), "hipMalloc");
        b_first.alloc((n + 1) * sizeof(uint64_t), "hipMalloc");
        clsimhip_step_request *d_req = b_req.as<clsimhip_step_request>();
        uint64_t *d_first = b_first.as<uint64_t>();
        if (n) chk(hipMemcpyAsync(d_req, requests, n * sizeof(clsimhip_step_request), hipMemcpyHostToDevice, stream), "upload requests");
        chk(hipMemcpyAsync(d_first, first.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream), "upload offsets");
        chk(launch_generate_steps(d_req, d_first, static_cast<uint32_t>(n ? n : 1), real, padded, seed, d_steps, stream), "step generation kernel launch");
        // the request copies were made from pageable memory (synchronous w.r.t. the host); free after the kernel
        chk(hipStreamSynchronize(stream), "step generation kernel");
    });
}
int clsimhip_generate_steps(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed, size_t granularity,
                            clsimhip_step *steps_out, size_t capacity, size_t *padded_out)
{
    return guarded(nullptr, [&] {
        need(steps_out, "steps_out");
        std::vector<uint64_t> first; uint64_t real = 0, padded = 0;
        plan_steps(requests, n, granularity, first, real, padded);
        if (padded > capacity) throw Error(CLSIMHIP_ERR_ARGUMENT, "the requests produce more steps than the buffer holds");
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the step producer has no CPU fallback)");
        DeviceGuard on_device(device);
        DeviceBuffer b_steps;
        b_steps.alloc(padded * sizeof(clsimhip_step), "hipMalloc");
        void *d_steps = b_steps.p;
        size_t got = 0;
        const int rc = clsimhip_generate_steps_device(device, requests, n, seed, granularity, d_steps, padded, nullptr, &got);
        if (rc != CLSIMHIP_OK) throw Error(rc, g_last_error);
        chk(hipMemcpy(steps_out, d_steps, padded * sizeof(clsimhip_step), hipMemcpyDeviceToHost), "download steps");
        if (padded_out) *padded_out = static_cast<size_t>(padded);
    });
}
} // extern "C"

// ---- flasher step producer ----
extern "C" {
int clsimhip_count_flasher_steps(const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests, size_t n,
                                 size_t *steps_out, size_t *real_steps_out)
{
    return guarded(nullptr, [&] {
        need(config, "config"); if (n) need(requests, "requests");
        std::vector<FlasherPlanEntry> plan; std::vector<double> widths;
        const uint64_t total = plan_flasher_steps(*config, requests, n, plan, widths);
        uint64_t real = 0;
        for (const FlasherPlanEntry &e : plan) real += e.n_real;
        if (steps_out) *steps_out = static_cast<size_t>(total);
        if (real_steps_out) *real_steps_out = static_cast<size_t>(real);
    });
}
int clsimhip_flasher_time_profile(double pulse_width_ns, float density[240], float cumulative[240])
{
    return guarded(nullptr, [&] {
        need(density, "density"); need(cumulative, "cumulative");
        if (!(pulse_width_ns > 0.)) throw Error(CLSIMHIP_ERR_ARGUMENT, "the flasher time profile needs a positive pulse width");
        std::vector<float> d, c;
        interpolated_distribution_tables(0.5, flasher_time_profile(pulse_width_ns), d, c);
        std::copy(d.begin(), d.end(), density); std::copy(c.begin(), c.end(), cumulative);
    });
}
int clsimhip_generate_flasher_steps_device(int device, const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests,
                                           size_t n, uint64_t seed, void *d_steps, size_t capacity, void *hip_stream, size_t *steps_out)
{
    return guarded(nullptr, [&] {
        need(config, "config"); need(d_steps, "d_steps"); if (n) need(requests, "requests");
        std::vector<FlasherPlanEntry> plan; std::vector<double> widths;
        const uint64_t total = plan_flasher_steps(*config, requests, n, plan, widths);
        if (total > capacity) throw Error(CLSIMHIP_ERR_ARGUMENT, "the pulses produce more steps than the buffer holds");
        if (steps_out) *steps_out = static_cast<size_t>(total);
        if (total == 0) return;
        std::vector<float> profiles(std::max<size_t>(widths.size(), 1) * 2 * kFlasherProfilePoints, 0.f);
        for (size_t w = 0; w < widths.size(); ++w) {
            std::vector<float> d, c;
            interpolated_distribution_tables(0.5, flasher_time_profile(widths[w]), d, c);
            std::copy(d.begin(), d.end(), profiles.begin() + w * 2 * kFlasherProfilePoints);
            std::copy(c.begin(), c.end(), profiles.begin() + (w * 2 + 1) * kFlasherProfilePoints);
        }
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the step producer has no CPU fallback)");
        DeviceGuard on_device(device);
        hipStream_t stream = static_cast<hipStream_t>(hip_stream);
        DeviceBuffer b_req, b_plan, b_prof;
        b_req.alloc(n * sizeof(clsimhip_flasher_request), "hipMalloc");
        b_plan.alloc(n * sizeof(FlasherPlanEntry), "hipMalloc");
        b_prof.alloc(profiles.size() * sizeof(float), "hipMalloc");
        void *d_req = b_req.p, *d_plan = b_plan.p, *d_prof = b_prof.p;
        chk(hipMemcpyAsync(d_req, requests, n * sizeof(clsimhip_flasher_request), hipMemcpyHostToDevice, stream), "upload pulses");
        chk(hipMemcpyAsync(d_plan, plan.data(), n * sizeof(FlasherPlanEntry), hipMemcpyHostToDevice, stream), "upload plan");
        chk(hipMemcpyAsync(d_prof, profiles.data(), profiles.size() * sizeof(float), hipMemcpyHostToDevice, stream), "upload time profiles");
        chk(launch_generate_flasher_steps(*config, static_cast<const clsimhip_flasher_request *>(d_req), d_plan, static_cast<uint32_t>(n), total, seed,
                                          static_cast<const float *>(d_prof), d_steps, stream), "flasher step kernel launch");
        chk(hipStreamSynchronize(stream), "flasher step kernel");
    });
}
int clsimhip_generate_flasher_steps(int device, const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests,
                                    size_t n, uint64_t seed, clsimhip_step *steps_out, size_t capacity, size_t *count_out)
{
    return guarded(nullptr, [&] {
        need(config, "config"); need(steps_out, "steps_out");
        size_t total = 0;
        int rc = clsimhip_count_flasher_steps(config, requests, n, &total, nullptr);
        if (rc != CLSIMHIP_OK) throw Error(rc, g_last_error);
        if (total > capacity) throw Error(CLSIMHIP_ERR_ARGUMENT, "the pulses produce more steps than the buffer holds");
        auto chk = [](hipError_t e, const char *w) { if (e != hipSuccess) throw Error(CLSIMHIP_ERR_DEVICE, std::string(w) + ": " + hipGetErrorString(e)); };
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) throw Error(CLSIMHIP_ERR_DEVICE, "no HIP device available (the step producer has no CPU fallback)");
        DeviceGuard on_device(device);
        DeviceBuffer b_steps;
        b_steps.alloc(total * sizeof(clsimhip_step), "hipMalloc");
        void *d_steps = b_steps.p;
        rc = clsimhip_generate_flasher_steps_device(device, config, requests, n, seed, d_steps, total, nullptr, nullptr);
        if (rc != CLSIMHIP_OK) throw Error(rc, g_last_error);
        chk(hipMemcpy(steps_out, d_steps, total * sizeof(clsimhip_step), hipMemcpyDeviceToHost), "download steps");
        if (count_out) *count_out = total;
    });
}
} // extern "C"

// ---- step store ----
struct clsimhip_step_store {
    clsimhip::StepStore impl;
    explicit clsimhip_step_store(size_t bins) : impl(bins) {}
};
extern "C" {
int clsimhip_step_store_create(size_t initial_bins, clsimhip_step_store **out)
{
    return guarded(nullptr, [&] { need(out, "out"); *out = new clsimhip_step_store(initial_bins); });
}
void clsimhip_step_store_destroy(clsimhip_step_store *s) { delete s; }
int clsimhip_step_store_insert(clsimhip_step_store *s, const clsimhip_step *steps, size_t n)
{
    return guarded(nullptr, [&] {
        need(s, "store"); if (n) need(steps, "steps");
        s->impl.insert_many(steps, n);
    });
}
int clsimhip_step_store_size(const clsimhip_step_store *s, size_t *out)
{
    return guarded(nullptr, [&] { need(s, "store"); need(out, "out"); *out = s->impl.size(); });
}
int clsimhip_step_store_count(const clsimhip_step_store *s, uint32_t identifier, uint32_t *out)
{
    return guarded(nullptr, [&] { need(s, "store"); need(out, "out"); *out = s->impl.count(identifier); });
}
int clsimhip_step_store_pop_bunch(clsimhip_step_store *s, size_t size, clsimhip_step *out, size_t *popped)
{
    return guarded(nullptr, [&] {
        need(s, "store"); need(popped, "popped"); if (size) need(out, "out");
        *popped = s->impl.pop_bunch(size, out);
    });
}
int clsimhip_step_store_pop_bunch_filled(clsimhip_step_store *s, size_t size, clsimhip_step *out, const clsimhip_step *fill)
{
    return guarded(nullptr, [&] {
        need(s, "store"); need(fill, "fill"); if (size) need(out, "out");
        s->impl.pop_bunch_filled(size, out, *fill);
    });
}
int clsimhip_step_store_size_with_dummy_fill(const clsimhip_step_store *s, size_t granularity, size_t *out)
{
    return guarded(nullptr, [&] { need(s, "store"); need(out, "out"); *out = s->impl.size_with_dummy_fill(granularity); });
}
} // extern "C"

// ---- photon table maker ----

int clsimhip_tabulator_create(int device, int axes_kind, const clsimhip_axis *axes, size_t n_axes, int store_squared_weights,
                              const clsimhip_medium *medium, const clsimhip_function *wavelength_acceptance,
                              const clsimhip_polynomial *angular_acceptance, double reference_area, double step_length,
                              const uint64_t *x, const uint32_t *a, size_t streams, clsimhip_tabulator **out)
{
    return guarded_tab(nullptr, [&] {
        need(axes, "axes"); need(medium, "medium"); need(angular_acceptance, "angular_acceptance"); need(out, "out");
        std::vector<AxisData> ax(n_axes);
        for (size_t i = 0; i < n_axes; ++i) {
            if (axes[i].kind != CLSIMHIP_AXIS_LINEAR && axes[i].kind != CLSIMHIP_AXIS_POWER) throw Error(CLSIMHIP_ERR_ARGUMENT, "unknown axis kind");
            ax[i].kind = axes[i].kind; ax[i].min = axes[i].min; ax[i].max = axes[i].max; ax[i].n_bins = axes[i].n_bins;
            ax[i].power = (axes[i].kind == CLSIMHIP_AXIS_POWER) ? axes[i].power : 1;
        }
        PolynomialData poly;
        if (angular_acceptance->n < 0 || (angular_acceptance->n > 0 && !angular_acceptance->coefficients))
            throw Error(CLSIMHIP_ERR_ARGUMENT, "polynomial coefficients are (null)");
        poly.coefficients.assign(angular_acceptance->coefficients, angular_acceptance->coefficients + angular_acceptance->n);
        poly.range_min = angular_acceptance->range_min; poly.range_max = angular_acceptance->range_max;
        poly.underflow = angular_acceptance->underflow; poly.overflow = angular_acceptance->overflow;
        std::unique_ptr<clsimhip_tabulator> t(new clsimhip_tabulator);
        t->impl.reset(new Tabulator(device, axes_kind, std::move(ax), store_squared_weights != 0, medium->data,
                                    function_from(wavelength_acceptance), poly, reference_area, step_length, x, a, streams));
        *out = t.release();
    });
}
void clsimhip_tabulator_destroy(clsimhip_tabulator *t) { delete t; }
const char *clsimhip_tabulator_last_error(const clsimhip_tabulator *t) { (void)t; return g_last_error.c_str(); }
int clsimhip_tabulator_enqueue_steps(clsimhip_tabulator *t, const clsimhip_step *steps, size_t n, const double reference[7])
{
    return guarded_tab(t, [&] { need(t, "tabulator"); t->impl->enqueue_steps(steps, n, reference); });
}
int clsimhip_tabulator_finish(clsimhip_tabulator *t) { return guarded_tab(t, [&] { need(t, "tabulator"); t->impl->finish(); }); }
int clsimhip_tabulator_set_tuning(clsimhip_tabulator *t, const char *key, long long value)
{
    return guarded_tab(t, [&] { need(t, "tabulator"); need(key, "key"); t->impl->set_tuning(key, value); });
}
int clsimhip_tabulator_get_shape(const clsimhip_tabulator *t, size_t *n_bins, size_t *n_dim, size_t shape[5])
{
    return guarded_tab(const_cast<clsimhip_tabulator *>(t), [&] {
        need(t, "tabulator"); need(n_bins, "n_bins"); need(n_dim, "n_dim"); need(shape, "shape");
        *n_bins = t->impl->n_bins();
        *n_dim = t->impl->shape().size();
        for (size_t i = 0; i < 5; ++i) shape[i] = i < t->impl->shape().size() ? t->impl->shape()[i] : 0;
    });
}
int clsimhip_tabulator_get_bin_content(clsimhip_tabulator *t, float *out, size_t n_bins, int squared, int normalized)
{
    return guarded_tab(t, [&] { need(t, "tabulator"); need(out, "out"); t->impl->bin_content(out, n_bins, squared != 0, normalized != 0); });
}
int clsimhip_tabulator_get_bin_sums(clsimhip_tabulator *t, double *out, size_t n_bins, int squared)
{
    return guarded_tab(t, [&] { need(t, "tabulator"); t->impl->bin_content_double(out, n_bins, squared != 0); });
}
int clsimhip_tabulator_get_bin_edges(const clsimhip_tabulator *t, int axis, double *out, size_t cap)
{
    return guarded_tab(const_cast<clsimhip_tabulator *>(t), [&] {
        need(t, "tabulator"); need(out, "out");
        if (axis < 0 || static_cast<size_t>(axis) >= t->impl->axes().size()) throw Error(CLSIMHIP_ERR_ARGUMENT, "axis out of range");
        const AxisData &ax = t->impl->axes()[axis];
        if (cap < ax.n_bins + 1) throw Error(CLSIMHIP_ERR_ARGUMENT, "output buffer too small");
        for (unsigned i = 0; i <= ax.n_bins; ++i) out[i] = ax.bin_edge(i);      // Axis::GetBinEdges (Axis.cxx:62-74)
    });
}
int clsimhip_tabulator_get_statistics(clsimhip_tabulator *t, double out[8])
{
    return guarded_tab(t, [&] { need(t, "tabulator"); need(out, "out"); t->impl->statistics(out); });
}
int clsimhip_tabulator_get_rng_state(clsimhip_tabulator *t, uint64_t *x, size_t count)
{
    return guarded_tab(t, [&] { need(t, "tabulator"); t->impl->get_rng_state(x, count); });
}
int clsimhip_tabulator_write_fits_file(clsimhip_tabulator *t, const char *path, const char *const *keys, const int32_t *is_int,
                                       const int64_t *int_values, const double *double_values, size_t n_keys)
{
    return guarded_tab(t, [&] {
        need(t, "tabulator"); need(path, "path");
        std::vector<Tabulator::HeaderEntry> header;
        for (size_t i = 0; i < n_keys; ++i) {
            need(keys, "keys"); need(is_int, "is_int"); need(keys[i], "key");
            Tabulator::HeaderEntry e{keys[i], is_int[i] != 0, 0, 0.};
            if (e.is_int) { need(int_values, "int_values"); e.i = int_values[i]; }
            else { need(double_values, "double_values"); e.d = double_values[i]; }
            header.push_back(e);
        }
        t->impl->write_fits_file(path, header);
    });
}
long clsimhip_tabulator_get_table(const clsimhip_tabulator *t, const char *name, double *out, size_t cap)
{
    long n = -1;
    const int rc = guarded_tab(const_cast<clsimhip_tabulator *>(t), [&] { need(t, "tabulator"); need(name, "name"); n = t->impl->get_table(name, out, cap); });
    return rc == CLSIMHIP_OK ? n : rc;
}

} // extern "C"
