/*
 * clsimhip.h -- C ABI of the MI355X photon propagator (libclsimhip.so).
 *
 * This is the drop-in boundary behind clsim's step->photon converter: every
 * entry point replaces one member of the reference's C++ interface
 *   public/clsim/I3CLSimStepToPhotonConverter.h:67-192        (abstract interface)
 *   public/clsim/I3CLSimStepToPhotonConverterOpenCL.h:78-258  (concrete setters)
 * or one host helper the canonical caller uses to configure it
 *   private/clsim/I3CLSimModuleHelper.cxx:175-372, python/MakeIceCubeMediumProperties.py,
 *   python/GetIceCubeDOMAcceptance.py, private/opencl/mwcrng_init.h.
 * Plain pointers and sizes only; no C++ / torch types.  All functions return
 * CLSIMHIP_OK (0) or a negative status; clsimhip_last_error() gives the text
 * (the reference throws I3CLSimStepToPhotonConverter_exception with that text).
 *
 * Records are the reference's packed little-endian wire structs:
 *   step   48 B  public/clsim/I3CLSimStep.h:141-155
 *   photon 80 B  public/clsim/I3CLSimPhoton.h:194-213
 */
#ifndef CLSIMHIP_H
#define CLSIMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLSIMHIP_OK 0
#define CLSIMHIP_ERR_ARGUMENT (-1)  /* null / empty / out-of-range argument            */
#define CLSIMHIP_ERR_STATE (-2)     /* setter after Initialize, use before Initialize   */
#define CLSIMHIP_ERR_CONFIG (-3)    /* configuration this build cannot run              */
#define CLSIMHIP_ERR_DEVICE (-4)    /* HIP runtime error / no GPU                       */
#define CLSIMHIP_ERR_IO (-5)        /* file not found / malformed table                 */

typedef struct clsimhip_converter clsimhip_converter;
typedef struct clsimhip_medium clsimhip_medium;

/* I3CLSimStep (48 B) and I3CLSimPhoton (80 B), byte-compatible with the reference. */
#pragma pack(push, 1)
typedef struct {
    float x, y, z, time;
    float theta, phi, length, beta;
    uint32_t num_photons;
    float weight;
    uint32_t identifier;
    uint8_t source_type, dummy1;
    uint16_t dummy2;
} clsimhip_step;
typedef struct {
    float x, y, z, time;
    float theta, phi, wavelength, cherenkov_dist;
    uint32_t num_scatters;
    float weight;
    uint32_t identifier;
    int16_t string_id;
    uint16_t om_id;
    float start_x, start_y, start_z, start_time;
    float start_theta, start_phi, group_velocity, dist_in_abs_lens;
} clsimhip_photon;
#pragma pack(pop)

/* ---- value descriptions (doubles, like the reference's function objects) ---- */

/* I3CLSimFunction: FromTable (equal spacing) or Constant.
 * private/clsim/function/I3CLSimFunctionFromTable.cxx:70-90, ...Constant.cxx:40-44.
 * Two more kinds exist on the host only, as the emission spectra handed to clsimhip_make_wlen_generator -- like the reference,
 * whose FromTable has no device code for unequal spacing (FromTable.cxx:169-170) and whose DeltaPeak is turned into a constant
 * generator (I3CLSimModuleHelper.cxx:78-89): a converter refuses them as wavelength bias, a medium as refractive index. */
#define CLSIMHIP_FUNCTION_TABLE 0
#define CLSIMHIP_FUNCTION_CONSTANT 1
#define CLSIMHIP_FUNCTION_TABLE_X 2     /* FromTable(wlens, values), FromTable.cxx:57-70: n wavelengths (ascending) + n values */
#define CLSIMHIP_FUNCTION_DELTA_PEAK 3  /* I3CLSimFunctionDeltaPeak: value = the peak's wavelength */
typedef struct {
    int32_t kind;
    int32_t n;               /* TABLE, TABLE_X: number of entries (>=2)   */
    double start, step;      /* TABLE: first wavelength, spacing [m]      */
    const double *values;    /* TABLE, TABLE_X: n values                  */
    double value;            /* CONSTANT; DELTA_PEAK: peak position [m]   */
    const double *wavelengths; /* TABLE_X: n wavelengths [m]              */
} clsimhip_function;

/* I3CLSimRandomValue used as wavelength generator: InterpolatedDistribution
 * (constant x spacing, or with its own x values: the flasher LEDs' measured spectra) or Constant (delta peak).
 * private/clsim/random_value/I3CLSimRandomValueInterpolatedDistribution.cxx:40-74 */
#define CLSIMHIP_RANDOM_INTERPOLATED 0
#define CLSIMHIP_RANDOM_CONSTANT 1
#define CLSIMHIP_RANDOM_CHERENKOV_NO_DISPERSION 2 /* I3CLSimRandomValueWlenCherenkovNoDispersion(fromWlen, toWlen)
                                                    (random_value/…WlenCherenkovNoDispersion.cxx:40-98): first = fromWlen,
                                                    spacing = toWlen */
#define CLSIMHIP_RANDOM_INTERPOLATED_X 3 /* InterpolatedDistribution(x, y), InterpolatedDistribution.cxx:40-55 */
typedef struct {
    int32_t kind;
    int32_t n;
    double first, spacing;   /* INTERPOLATED: x of first point, x spacing  */
    const double *y;         /* INTERPOLATED, INTERPOLATED_X: n unnormalised densities */
    double value;            /* CONSTANT                                    */
    const double *x;         /* INTERPOLATED_X: n abscissae, ascending      */
} clsimhip_random_value;

/* I3CLSimMediumProperties restricted to the function classes of the IceCube
 * ice models (public/clsim/I3CLSimMediumProperties.h:54-200). */
#define CLSIMHIP_LENGTHS_CONSTANT 0 /* I3CLSimFunctionConstant per layer          */
#define CLSIMHIP_LENGTHS_ICECUBE 1  /* I3CLSimFunctionAbsLenIceCube/ScatLenIceCube */
#define CLSIMHIP_LENGTHS_TABLE 2    /* one I3CLSimFunctionFromTable per layer (photonics ice tables) */
#define CLSIMHIP_REFINDEX_ICECUBE 0 /* I3CLSimFunctionRefIndexIceCube (n[] / g[])  */
#define CLSIMHIP_REFINDEX_TABLE 1   /* I3CLSimFunctionFromTable, same function for every layer */
#define CLSIMHIP_REFINDEX_DISPERSION 2 /* group_index_kind only: NO group refractive index override is set; the group velocity comes from the
                                        * phase index and its derivative (I3CLSimHelperGenerateMediumPropertiesSource.cxx:274-300,
                                        * I3CLSimFunctionRefIndexIceCube.cxx:205-215).  Needs phase_index_kind ICECUBE: FromTable has no
                                        * derivative (I3CLSimFunctionFromTable.h:67).  The table maker refuses it as the reference does
                                        * (I3CLSimStepToTableConverter.cxx:103-104) */
#define CLSIMHIP_SCATTER_HG 0
#define CLSIMHIP_SCATTER_LIU 1
#define CLSIMHIP_SCATTER_MIXED 2    /* Mixed(SimplifiedLiu, HenyeyGreenstein, f)   */
typedef struct {
    int32_t num_layers;
    double layers_z_start, layers_height;
    double min_wavelength, max_wavelength;          /* ForcedMinWlen / ForcedMaxWlen */
    int32_t lengths_kind;
    const double *abs_length, *sca_length;          /* CONSTANT: per layer [m]        */
    double alpha, kappa, A, B, D, E;                /* ICECUBE                         */
    const double *a_dust400, *delta_tau, *b400;     /* ICECUBE: per layer, bottom->top */
    double n[5], g[5];                              /* RefIndexIceCube phase / group   */
    int32_t scatter_kind;
    double liu_fraction, mean_cosine;
    int32_t has_anisotropy;                         /* ScalarFieldAnisotropyAbsLenScaling */
    double aniso_azimuth, aniso_k1, aniso_k2;
    int32_t has_pre_transform, pre_renormalize;     /* VectorTransformMatrix            */
    double pre_matrix[9];
    int32_t has_post_transform, post_renormalize;
    double post_matrix[9];
    int32_t has_tilt;                               /* ScalarFieldIceTiltZShift          */
    int32_t tilt_num_distances, tilt_num_z;
    const double *tilt_distances, *tilt_z_coordinates, *tilt_z_corrections; /* [nd][nz] */
    double tilt_azimuth;
    /* TABLE lengths (private/clsim/function/I3CLSimFunctionFromTable.cxx:167-300): equal spacing common to
     * all layers; store_as_16bit = storeDataAsHalfPrecision (linear 16-bit quantisation between the
     * smallest and largest entry of each function) */
    int32_t table_num_wavelengths;
    double table_start_wavelength, table_wavelength_step;
    int32_t table_store_as_16bit;
    const double *abs_length_table, *sca_length_table; /* [num_layers][table_num_wavelengths], metres */
    /* refractive indices; the propagator needs a layer independent group velocity
     * (propagation_kernel.c.cl:525-527), so there is one function of each kind */
    int32_t phase_index_kind, group_index_kind;         /* CLSIMHIP_REFINDEX_* */
    clsimhip_function phase_index_table, group_index_table; /* kind TABLE */
} clsimhip_medium_desc;

/* ---- medium objects ---- */
/* deep copy of a description */
int clsimhip_medium_create(const clsimhip_medium_desc *desc, clsimhip_medium **out);
/* python/MakeIceCubeMediumProperties.py:49-256: PPC ice tables (icemodel.dat/.par,
 * cfg.txt[, tilt.dat/.par]) -> medium */
int clsimhip_medium_create_from_ppc(const char *directory, double detector_center_depth,
                                    int use_tilt_if_available, clsimhip_medium **out);
/* python/MakeIceCubeMediumPropertiesPhotonics.py:47-227: photonics ice table file (NLAYER / NWVL / per layer
 * LAYER, ABS, SCAT, COS, N_GROUP, N_PHASE) -> medium with tabulated lengths and refractive indices */
int clsimhip_medium_create_from_photonics(const char *table_file, double detector_center_depth, clsimhip_medium **out);
/* view into the object's own arrays (valid until destroy) */
int clsimhip_medium_describe(const clsimhip_medium *m, clsimhip_medium_desc *out);
void clsimhip_medium_destroy(clsimhip_medium *m);

/* python/GetIceCubeDOMAcceptance.py:35-115 (43 entries, 260..680 nm); values_out[43] */
int clsimhip_icecube_dom_acceptance(double dom_radius, double efficiency, double *values_out,
                                    double *start_out, double *step_out);
/* I3CLSimModuleHelper::makeCherenkovWavelengthGenerator, tabulated bias
 * (I3CLSimModuleHelper.cxx:175-263): y_out[bias->n] */
int clsimhip_make_cherenkov_wlen_generator(const clsimhip_function *bias, const clsimhip_medium *m,
                                           double *y_out, double *first_out, double *spacing_out);

/* I3CLSimModuleHelper::makeWavelengthGenerator (I3CLSimModuleHelper.cxx:73-171) for the spectrum classes its callers pass
 * (python/GetIceCubeFlasherSpectrum.py): DELTA_PEAK -> CONSTANT; TABLE / TABLE_X -> INTERPOLATED / INTERPOLATED_X on the
 * table's own binning, every entry multiplied by the bias at its wavelength.  `out` is filled in with out->y = y_out and
 * out->x = x_out (x_out only written for INTERPOLATED_X); both arrays hold `capacity` >= spectrum->n doubles. */
int clsimhip_make_wlen_generator(const clsimhip_function *spectrum, const clsimhip_function *bias, const clsimhip_medium *m,
                                 clsimhip_random_value *out, double *x_out, double *y_out, size_t capacity);

/* ---- RNG set-up (private/opencl/mwcrng_init.h:26-117, private/make_safeprimes/main.cxx) ---- */
/* first `count` MWC multipliers (a*2^32-1 and (a*2^32-2)/2 prime, descending from 4294967118) */
int clsimhip_mwc_multipliers(uint32_t *a_out, size_t count);
/* first `count` multipliers from a safeprimes file in one of the reference's formats (mwcrng_init.h:62-103:
 * 17-byte tag "safeprimes_base32" + little-endian int64 each, or text with the multiplier in column 1) */
int clsimhip_mwc_multipliers_from_file(const char *path, uint32_t *a_out, size_t count);
/* state words with init_MWC_RNG's validity loop; draws come from splitmix64(seed) */
int clsimhip_seed_streams(const uint32_t *a, size_t count, uint64_t seed, uint64_t *x_out);

/* ---- converter life cycle (I3CLSimStepToPhotonConverterOpenCL.cxx:68-388) ---- */
int clsimhip_create(int device_ordinal, clsimhip_converter **out);
void clsimhip_destroy(clsimhip_converter *c);
/* Text of the last failure of a clsimhip_* call made BY THE CALLING THREAD (thread-local, like errno; `c` is ignored and
 * may be NULL): the interface is driven by several threads per converter (I3CLSimServer.cxx:126-135, 324-331). */
const char *clsimhip_last_error(const clsimhip_converter *c);
/* SetDevice (public/clsim/I3CLSimStepToPhotonConverterOpenCL.h:96-103, OpenCL.cxx:1322-1331; the canonical caller's
 * first call, I3CLSimModuleHelper.cxx:321): the HIP device ordinal replaces the I3CLSimOpenCLDevice.  Before Initialize. */
int clsimhip_set_device(clsimhip_converter *c, int device_ordinal);
int clsimhip_get_device(const clsimhip_converter *c, int *out);
/* Device errors: a HIP error inside the converter's worker thread is where the reference log_fatal()s and exits
 * (OpenCL.cxx:768-774).  This library does not end its host: the worker stops, blocked callers wake up, and every
 * later EnqueueSteps / GetConversionResult returns CLSIMHIP_ERR_DEVICE with the original text. */

/* setters: CLSIMHIP_ERR_STATE once initialized (OpenCL.cxx:1322-1523) */
int clsimhip_set_wlen_generators(clsimhip_converter *c, const clsimhip_random_value *gens, size_t n);
int clsimhip_set_wlen_bias(clsimhip_converter *c, const clsimhip_function *bias);
int clsimhip_set_medium_properties(clsimhip_converter *c, const clsimhip_medium *m);
/* I3CLSimSimpleGeometry: parallel arrays, one entry per DOM (public/clsim/I3CLSimSimpleGeometry.h) */
int clsimhip_set_geometry(clsimhip_converter *c, size_t n, const int32_t *string_ids, const uint32_t *dom_ids,
                          const double *x, const double *y, const double *z,
                          const char *const *subdetectors, double om_radius);
/* I3CLSimSimpleGeometryTextFile (private/clsim/I3CLSimSimpleGeometryTextFile.cxx:43-100): whitespace separated
 * "string dom x y z" records; strings / DOMs outside [*_min, *_max] are ignored (reference defaults: 1..INT32_MAX,
 * 1..60); every DOM is in the subdetector "default" */
int clsimhip_set_geometry_from_text_file(clsimhip_converter *c, const char *filename, double om_radius,
                                         int32_t string_id_min, int32_t string_id_max, uint32_t dom_id_min, uint32_t dom_id_max);
int clsimhip_set_enable_double_buffering(clsimhip_converter *c, int value);
int clsimhip_set_double_precision(clsimhip_converter *c, int value);           /* only 0 */
int clsimhip_set_stop_detected_photons(clsimhip_converter *c, int value);      /* 0 (the default, as in the reference class, OpenCL.cxx:86; initializeOpenCL's callers pass 1): every DOM on a photon's way records it and the photon travels on (no STOP_PHOTONS_ON_DETECTION) */
int clsimhip_set_save_all_photons(clsimhip_converter *c, int value);           /* only 0 */
int clsimhip_set_save_all_photons_prescale(clsimhip_converter *c, double value);
int clsimhip_set_fixed_number_of_absorption_lengths(clsimhip_converter *c, double value); /* NaN = off */
int clsimhip_set_dom_pancake_factor(clsimhip_converter *c, double value);
int clsimhip_set_photon_history_entries(clsimhip_converter *c, uint32_t value); /* only 0 */
int clsimhip_set_workgroup_size(clsimhip_converter *c, size_t value);
int clsimhip_set_max_num_workitems(clsimhip_converter *c, size_t value);
/* Compile(): build the device tables from the configuration (OpenCL.cxx:485-533) */
int clsimhip_compile(clsimhip_converter *c);
int clsimhip_get_max_workgroup_size(const clsimhip_converter *c, size_t *out);
/* Initialize(): RNG streams a[i] = i-th multiplier, x[i] seeded from `seed` (OpenCL.cxx:217-388) */
int clsimhip_initialize(clsimhip_converter *c, uint64_t seed);
/* same with caller-supplied streams (count must equal max_num_workitems) */
int clsimhip_initialize_with_streams(clsimhip_converter *c, const uint64_t *x, const uint32_t *a, size_t count);
int clsimhip_is_initialized(const clsimhip_converter *c);

/* ---- steady state (OpenCL.cxx:1525-1640) ---- */
/* EnqueueSteps: copies `n` steps; blocks while 5 bunches are pending.  n must be
 * non-zero, <= max_num_workitems and a multiple of the workgroup size. */
int clsimhip_enqueue_steps(clsimhip_converter *c, const clsimhip_step *steps, size_t n, uint32_t identifier);
/* GetConversionResult: blocks for the next finished bunch.  String/DOM indices are
 * already replaced by IDs.  *photons stays valid until clsimhip_release_result. */
int clsimhip_get_conversion_result(clsimhip_converter *c, uint32_t *identifier,
                                   const clsimhip_photon **photons, size_t *n);
/* ConversionResult_t::photonHistories (I3CLSimStepToPhotonConverter.h:70-90) of the result that `photons` belongs to,
 * as ConvertPhotonHistories builds them (OpenCL.cxx:940-989): `*entries` float[4] records per photon {x, y, z,
 * absorption lengths travelled} in forward order (most recent scatter last); photon i has
 * min(numScatters_i, *entries) of them, the rest of its block is zero.  *histories is NULL when
 * PhotonHistoryEntries is 0 or the result is empty; valid until clsimhip_release_result(photons). */
int clsimhip_get_result_histories(clsimhip_converter *c, const clsimhip_photon *photons, const float **histories,
                                  uint32_t *entries);
int clsimhip_release_result(clsimhip_converter *c, const clsimhip_photon *photons);
int clsimhip_get_workgroup_size(const clsimhip_converter *c, size_t *out);
int clsimhip_get_max_num_workitems(const clsimhip_converter *c, size_t *out);
int clsimhip_queue_size(const clsimhip_converter *c, size_t *out);
int clsimhip_more_photons_available(const clsimhip_converter *c, int *out);
/* GetStatistics(): [0] TotalDeviceTime ns, [1] TotalHostTime ns, [2] NumKernelCalls,
 * [3] TotalNumPhotonsGenerated, [4] TotalNumPhotonsAtDOMs, [5] AverageDeviceTimePerPhoton,
 * [6] AverageHostTimePerPhoton, [7] DeviceUtilization */
int clsimhip_get_statistics(const clsimhip_converter *c, double out[8]);
/* the option getters of the concrete class (GetEnableDoubleBuffering ... GetDOMPancakeFactor, OpenCL.h:138-258): what the
 * setters stored */
#define CLSIMHIP_OPTION_ENABLE_DOUBLE_BUFFERING 0
#define CLSIMHIP_OPTION_DOUBLE_PRECISION 1
#define CLSIMHIP_OPTION_STOP_DETECTED_PHOTONS 2
#define CLSIMHIP_OPTION_SAVE_ALL_PHOTONS 3
#define CLSIMHIP_OPTION_SAVE_ALL_PHOTONS_PRESCALE 4
#define CLSIMHIP_OPTION_FIXED_NUMBER_OF_ABSORPTION_LENGTHS 5    /* NaN: not set */
#define CLSIMHIP_OPTION_DOM_PANCAKE_FACTOR 6
#define CLSIMHIP_OPTION_PHOTON_HISTORY_ENTRIES 7
int clsimhip_get_option(const clsimhip_converter *c, int option, double *out);

/* ---- tuning (no reference counterpart; the reference's one launch parameter is the work-group size it takes from
 * the device, OpenCL.cxx:520-560) ----
 * The launcher's parameters have defaults found by measurement; a caller who knows its bunches better
 * sets them here.  NO RESULT DEPENDS ON ANY OF THEM: every schedule gives the same photon records and the same final
 * RNG states (tests/test_stress_schedules_gpu.py).  The library reads NO tuning from the environment (a build with
 * -DCLSIMHIP_DEVELOPER, `make DEVELOPER=1`, honours the round 1-5 variable names for the measurement tools under
 * tools/); the only environment variables of the default build are CLSIMHIP_SAFEPRIMES_FILE (a file of multipliers
 * in the reference's format, mwcrng_init.h:44-82) and CLSIMHIP_RCCL_LIBRARY (which RCCL to dlopen).
 *
 *   key                  values           meaning (default)
 *   "kernel"             0 | 1 | 2        0: pooled kernel for bunches of at least "pool_min_steps" steps, classic below; 1: pooled,
 *                                         2: classic, for every bunch size (0)
 *   "pool_min_steps"     -1 | >= 0        smallest bunch the pooled kernel takes; -1: 524 288, or 0 when "kernel" is 1 (-1)
 *   "pool_max_steps"     -1 | >= 1        largest bunch the pooled kernel takes; can only lower the limit of its 23-bit step index (-1)
 *   "pool_ring"          0 ... 4096       pooled kernel: created photons a wave keeps ready; 0: what the LDS holds, 45 for IceCube (0)
 *   "k_new"              0 ... 4096       lanes / ring entries waiting before a wave creates photons; 0: automatic (0)
 *   "k_search"           0 ... 64         lanes parked before a wave searches for DOMs; 0: automatic (0)
 *   "slices"             0 ... 65535      work units a step is cut into; 0: 16, or 1 for bunches smaller than the grid (0)
 *   "k_pop"              0 ... 64         pooled kernel: free lanes before a wave hands out ready photons; 0: 4 (0)
 *   "k_wait"             -1 ... 255       pooled kernel: trips a parked lane waits for company; -1: 16; 0: never (-1)
 *   "k_aim"              -1 ... 64        pooled kernel: lanes at a string up to which the "aimed at the string?" level is asked;
 *                                         -1: 8; 0: the level is off (-1)
 *   "grid"               0 ... 2^20       workgroups of the propagation launch; 0: what the chip holds, cut to the work (0)
 *   "generic_kernels"    0 | 1            1: the generic instantiation also where Compile() found every proof of the fast one (0)
 *   "result_min_records" >= 1             smallest page-locked result buffer, in photon records (65 536)
 *   -- the three below shape tables of Compile(): CLSIMHIP_ERR_STATE after Compile() --
 *   "string_map_cells"   8 ... 4096       string proximity map, cells per axis (512)
 *   "dom_map_cells"      4 ... 512        DOM proximity map, largest number of cubic cells per axis (256)
 *   "named_search"       0 | 1            0: every DOM takes the full search instead of the one confined to the named DOM (1)
 * Any other key or value: CLSIMHIP_ERR_ARGUMENT.  May be called between bunches at any time; a launch reads the values
 * when it is queued. */
int clsimhip_set_tuning(clsimhip_converter *c, const char *key, long long value);
int clsimhip_get_tuning(const clsimhip_converter *c, const char *key, long long *value);

/* ---- device-resident path (no reference counterpart: the reference always
 * stages through host memory, OpenCL.cxx:824-911, 994-1086) ----
 * Propagates n steps that already live in HBM, on the caller's HIP stream
 * (hipStream_t passed as void*; NULL = default stream).  d_photons has room for
 * `capacity` records, *d_hit_count (uint32, zeroed by this call) receives the
 * number of detected photons (may exceed capacity; only the first `capacity`
 * are stored).  rng_offset selects the first RNG stream used (stream i of the
 * bunch = rng_offset + i).  String/DOM fields hold INDICES.  Asynchronous. */
int clsimhip_propagate_device(clsimhip_converter *c, const void *d_steps, size_t n, size_t rng_offset,
                              void *d_photons, size_t capacity, void *d_hit_count, void *stream);
/* Small bunches: a launch is a persistent grid sized for the whole chip, and a bunch of a few 10^5 steps leaves most lanes
 * with less than one step.  A caller that has several such bunches keeps k of them in flight instead -- k calls of
 * clsimhip_propagate_device on k different HIP streams, over disjoint RNG stream ranges [rng_offset, rng_offset + n) and
 * with output buffers of their own -- after telling the converter so: every launch then sizes its grid for 1/k of the
 * chip and they run side by side (work records and queue heads are per stream range and per launch).  Results do not
 * depend on k.  1 <= k <= 16, default 1; may be changed between launches. */
int clsimhip_set_concurrent_device_launches(clsimhip_converter *c, int k);
/* index -> ID translation on the device records' host copy (OpenCL.cxx:1565-1600) */
int clsimhip_replace_indices_with_ids(const clsimhip_converter *c, clsimhip_photon *photons, size_t n);
/* average duration (ms) of the propagation kernel over the launches recorded since
 * the last call with reset!=0, measured with HIP events on the launch stream */
int clsimhip_kernel_time_ms(clsimhip_converter *c, int reset, double *total_ms, uint64_t *launches);

/* ---- wire format of step and photon series ------------------------------------------------
 * Payload of I3Vector<I3CLSimStep>::serialize / I3Vector<I3CLSimPhoton>::serialize for the portable binary archive
 * (private/clsim/I3CLSimStep.cxx:111-147, private/clsim/I3CLSimPhoton.cxx:141-168), the messages of I3CLSimServer /
 * I3CLSimClient (I3CLSimServer.cxx:320, 339, 386, 408): class version (0), number of records, then the records as one
 * little-endian blob -- from the class version on.  The archive framing around it (stream header, object tracking
 * records of the shared_ptr and of the I3FrameObject base) belongs to icecube::serialization, which is not part of the
 * reference tree; integers use that archive's published encoding (count byte + little-endian bytes). */
int clsimhip_step_series_blob_size(size_t n, size_t *bytes);
int clsimhip_encode_step_series(const clsimhip_step *steps, size_t n, uint8_t *out, size_t capacity, size_t *written);
/* *n = number of steps in the blob; steps_out may be NULL to ask for the count; *consumed (may be NULL) = bytes read */
int clsimhip_decode_step_series(const uint8_t *blob, size_t bytes, clsimhip_step *steps_out, size_t capacity, size_t *n, size_t *consumed);
int clsimhip_photon_series_blob_size(size_t n, size_t *bytes);
int clsimhip_encode_photon_series(const clsimhip_photon *photons, size_t n, uint8_t *out, size_t capacity, size_t *written);
int clsimhip_decode_photon_series(const uint8_t *blob, size_t bytes, clsimhip_photon *photons_out, size_t capacity, size_t *n, size_t *consumed);
/* the archive's unsigned integer encoding itself (out: up to 9 bytes) */
int clsimhip_encode_portable_uint(uint64_t value, uint8_t out[9], size_t *written);

/* ---- multi-GPU: gather of detected photons over RCCL / xGMI (no reference counterpart, the reference
 * collects the results of its per-device converters with host threads, I3CLSimServer.cxx:77-137) ----
 * One process (or thread) per GPU: each propagates a contiguous shard of the steps with clsimhip_propagate_device and
 * its own streams -- no exchange -- then the ranks' photons are gathered on `root`, rank by rank.  RCCL is loaded at
 * run time (dlopen; CLSIMHIP_RCCL_LIBRARY overrides the name): no link-time dependency. */
#define CLSIMHIP_UNIQUE_ID_BYTES 128
typedef struct clsimhip_comm clsimhip_comm;
/* ncclGetUniqueId: called by one rank; the host application hands the bytes to the others (MPI, ZMQ, a file ...) */
int clsimhip_comm_get_unique_id(uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES]);
/* ncclCommInitRank on HIP device `device_ordinal`; collective: every rank of the job calls it */
int clsimhip_comm_create(int device_ordinal, int rank, int world_size, const uint8_t id[CLSIMHIP_UNIQUE_ID_BYTES], clsimhip_comm **out);
void clsimhip_comm_destroy(clsimhip_comm *comm);
/* What the communicator says about itself: ncclCommCount / ncclCommUserRank asked of the communicator after
 * ncclCommInitRank (clsimhip_comm_create fails when they differ from what the caller passed), the HIP device it lives on
 * and that device's PCI bus id (hipDeviceGetPCIBusId; "" if unknown).  A multi-GPU record quotes THESE numbers: N ranks
 * with N distinct bus ids.  The reference reports per-device statistics from its server (I3CLSimServer.cxx:355-368).
 * Any out pointer may be NULL. */
int clsimhip_comm_info(clsimhip_comm *comm, int *rccl_ranks, int *rccl_rank, int *device_ordinal, char *pci_bus_id, size_t pci_bus_id_bytes);
/* Gathers issued since creation (or the last reset): their number, the time they held the stream (HIP events around
 * each clsimhip_gather_hits on its stream; the call waits for the gathers issued so far), the photon records this rank
 * sent to a root / received as a root. */
int clsimhip_comm_statistics(clsimhip_comm *comm, uint64_t *gathers, double *gather_ms, uint64_t *records_sent, uint64_t *records_received, int reset);
/* Collective.  d_photons / d_hit_count: this rank's photon buffer (`capacity` records) and uint32 hit counter as
 * clsimhip_propagate_device left them.  On `root`, d_gathered (room for gathered_capacity records) receives rank 0's
 * photons, then rank 1's ...; counts_out[world_size] (host, every rank, may be NULL) receives the ranks' hit counters
 * (a rank transfers min(counter, capacity) records).  Ordered after the work already queued on `hip_stream`; returns
 * once the counts are known -- the transfers (ncclAllGather of the counts, then one ncclSend/ncclRecv per peer in one
 * group: every peer uses its own xGMI link to the root) may still be running on `hip_stream`.
 * A gather buffer that is too small for all ranks' photons: every rank learns the root's gathered_capacity with the counts,
 * the root receives the records that fit (rank 0's first), sends and receives still pair up, and EVERY rank returns
 * CLSIMHIP_ERR_ARGUMENT (counts_out is filled). */
int clsimhip_gather_hits(clsimhip_comm *comm, const void *d_photons, const void *d_hit_count, size_t capacity, int root,
                         void *d_gathered, size_t gathered_capacity, uint64_t *counts_out, void *hip_stream);

/* ---- introspection used by the parity tests ---- */
/* Which scheduling the propagation kernel runs with (after Initialize): 1 = per-wave photon pools
 * (prop_pool_kernel.hip), 0 = one photon per lane (prop_kernel.hip).  Chosen per launch: pools for bunches large
 * enough to fill them (>= 524288 steps), never with photon histories or a table image that leaves the pools no LDS.
 * Results do not depend on it.  CLSIMHIP_KERNEL=pool|classic in the environment forces one for every bunch size
 * (clsimhip_uses_pooled_kernel then reports 1 / 0), CLSIMHIP_POOL_MIN_STEPS moves the threshold. */
int clsimhip_kernel_for_bunch(const clsimhip_converter *c, size_t n_steps, int *out);
int clsimhip_uses_pooled_kernel(const clsimhip_converter *c, int *out);
/* copies the compiled table `name` (e.g. "geoStringPosX", "aDust400") converted to
 * double into out[0..cap); returns the entry count or a negative status */
long clsimhip_get_table(const clsimhip_converter *c, const char *name, double *out, size_t cap);
/* current RNG state words (after the bunches run so far) */
int clsimhip_get_rng_state(clsimhip_converter *c, uint64_t *x_out, size_t count);
/* evaluates the device math library on the GPU: what = 0 log,1 exp,2 sin,3 cos,4 powr(x,y),
 * 5 acos,6 atan2(x,y),7 rsqrt,8 sqrt,9 x/y,10 acos (single precision),11 rcp_,12 sqrt_near_,13 rsqrt_near_,
 * 14 powr_unit_(x,y),15 cbrt_,16 div_near_(x,y) */
int clsimhip_eval_math(int device_ordinal, int what, const float *x, const float *y, size_t n, float *out);

/* Single functions of an initialised converter evaluated ON THE DEVICE, from the table image its kernels read: what the
 * reference's tester classes do (private/test/I3CLSimFunctionTester, ...ScalarFieldTester, ...VectorTransformTester,
 * ...MediumPropertiesTester, ...RandomDistributionTester; used by resources/tests/testScalarFields.py,
 * testScalarFieldIceTiltZShift.py, testVectorTransforms.py to compare device with host).  in4 / out4: n x 4 floats.
 * fast != 0: the forms the standard configuration's kernels use (exact reciprocals; only valid when Compile() proved their
 * ranges: CLSIMHIP_ERR_STATE otherwise), 0: the IEEE sequences.  Same values either way (tests/test_device_functions_gpu.py). */
#define CLSIMHIP_EVAL_LENGTHS 0                 /* in.x = wavelength [m]; out.x = absorption, out.y = scattering length of `layer` */
#define CLSIMHIP_EVAL_REFRACTION 1              /* in.x = wavelength; out.x = phase refractive index, out.y = group velocity [m/ns] */
#define CLSIMHIP_EVAL_WAVELENGTH_BIAS 2         /* in.x = wavelength; out.x = getWavelengthBias */
#define CLSIMHIP_EVAL_TILT 3                    /* in.xyz = position; out.x = getTiltZShift */
#define CLSIMHIP_EVAL_ABS_LEN_SCALING 4         /* in.xyz = direction; out.x = getDirectionalAbsLenCorrFactor */
#define CLSIMHIP_EVAL_PRE_SCATTER_TRANSFORM 5   /* in.xyz = direction; out.xyz = transformDirectionPreScatter */
#define CLSIMHIP_EVAL_POST_SCATTER_TRANSFORM 6  /* in.xyz = direction; out.xyz = transformDirectionPostScatter */
int clsimhip_eval_device_function(clsimhip_converter *c, int what, int layer, int fast, const float *in4, size_t n, float *out4);
/* n_streams work items, each with its own MWC stream (x[i], a[i]) -- x is updated --, draw `draws` values:
 * out[i * draws + k] (RandomDistributionTester.cxx:43-199) */
#define CLSIMHIP_EVAL_RANDOM_UNIFORM 0              /* rand_MWC_co */
#define CLSIMHIP_EVAL_RANDOM_WAVELENGTH 1           /* generateWavelength_<generator> */
#define CLSIMHIP_EVAL_RANDOM_SCATTERING_COSINE 2    /* makeScatteringCosAngle */
int clsimhip_eval_device_random(clsimhip_converter *c, int what, int generator, int fast, uint64_t *x, const uint32_t *a, size_t n_streams,
                                size_t draws, float *out);
/* exhaustive device-side check of the range-restricted operations of the kernel's math library (what = 11 reciprocal,
 * 12 square root, 13 reciprocal square root) against the IEEE divide / sqrt: all 2^23 significands x every binary
 * exponent in [exp_lo, exp_hi].  result[0] = number of mismatches, result[1..cap) = bit patterns of the first ones.
 * what = 16: the range-restricted divide, all 2^23 divisor significands x the divisor exponents [exp_lo, exp_hi] x both
 * divisor signs x 40 numerators each (random and adversarial, exponents -40 ... 60); result[1..] = (numerator, divisor)
 * pairs of the first mismatches.  what = 17 / 18: the reciprocal of a reciprocal, the reciprocal root next to one (detmath.hip.h).
 * what = 19: the table maker's axis bin (floor, saturating conversion, clamp: Axis.cxx:45-60) as the kernel forms it against the
 * spelled-out conversion on ALL 2^32 bit patterns x five bin counts (exp_lo / exp_hi do not apply). */
int clsimhip_check_math_exhaustive(int device_ordinal, int what, int exp_lo, int exp_hi, uint32_t *result, size_t result_cap);

const char *clsimhip_version(void);

/* ---- step producer on the GPU (the arithmetic that is part of the reference tree) ----------
 * One request = one entry of the reference's step generation queue (CascadeStepData_t / MuonStepData_t,
 * private/clsim/I3CLSimLightSourceToStepConverterPPC.h): the particle, how many steps of how many photons it was
 * given (I3CLSimLightSourceToStepConverterPPC.cxx:284-470 computes those from sim-services / GSL / the random
 * service -- not part of this library), and the longitudinal profile parameters.  Steps are generated like
 * FillStep/GenerateStep/GenerateStepForMuon (:524-551, :785-842) do, one GPU lane per step. */
#define CLSIMHIP_STEPS_CASCADE 0        /* CascadeStepData_t: position = pb * Gamma(pa) along the axis, PPC angular profile */
#define CLSIMHIP_STEPS_MUON_CASCADE 1   /* MuonStepData_t, stepIsCascadeLike: position uniform along `length`, angular profile */
#define CLSIMHIP_STEPS_MUON 2           /* MuonStepData_t, muon-like: every step is the whole track (length, direction) */
typedef struct {
    float x, y, z, time;                /* particle vertex [m, ns] */
    float dx, dy, dz;                   /* unit direction of flight */
    float length;                       /* MUON*: track length [m] */
    float pa, pb;                       /* CASCADE: shower_params.a, shower_params.b [m] (b = 0: no cascade extension) */
    uint32_t kind;
    uint32_t identifier;                /* -> I3CLSimStep::identifier */
    uint32_t photons_per_step;
    uint32_t num_photons_in_last_step;  /* a last step with this many photons is appended when > 0 */
    uint64_t num_steps;                 /* steps of photons_per_step photons */
} clsimhip_step_request;
/* steps the requests produce; *padded_out = that number rounded up to a multiple of `granularity` (the converter's
 * workgroup size) with the reference's no-op steps (Async.cxx:240-257) */
int clsimhip_count_generated_steps(const clsimhip_step_request *requests, size_t n, size_t granularity,
                                   size_t *steps_out, size_t *padded_out);
/* generates into device memory d_steps (room for `capacity` steps) on `hip_stream`; asynchronous */
int clsimhip_generate_steps_device(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed,
                                   size_t granularity, void *d_steps, size_t capacity, void *hip_stream, size_t *padded_out);
/* the same into host memory (synchronous) */
int clsimhip_generate_steps(int device, const clsimhip_step_request *requests, size_t n, uint64_t seed,
                            size_t granularity, clsimhip_step *steps_out, size_t capacity, size_t *padded_out);

/* ---- particle -> step requests: the front end of I3CLSimLightSourceToStepConverterPPC ----------
 * private/clsim/I3CLSimLightSourceToStepConverterPPC.cxx, Initialize :94-132 (photon yield per metre of track,
 * ConverterUtils.cxx:44-105) and EnqueueLightSource :188-470: 5.21 m of Cherenkov-radiating track per GeV (scaled with the
 * density), the electromagnetic fraction of hadronic showers with its fluctuation, a Poisson (Gaussian above 1e7) number
 * of photons, the split into steps of photonsPerStep (highPhotonsPerStep above a threshold) photons; muons and taus as
 * a bare track plus the average secondary light (1 + max(0, 0.1880 + 0.0206 ln E)).  Host only.
 * NOT part of the reference tree and therefore restated from published sources, PARITY UNPINNED:
 * I3SimConstants::ShowerParameters (sim-services), the random service's Gaus / Poisson (phys-services / GSL), and
 * gsl_integration_qag.  Particle types are I3Particle::ParticleType values (dataclasses): PDG codes, and IceCube's own
 * codes for stochastic losses. */
#define CLSIMHIP_PARTICLE_GAMMA 22
#define CLSIMHIP_PARTICLE_EMINUS 11
#define CLSIMHIP_PARTICLE_EPLUS (-11)
#define CLSIMHIP_PARTICLE_MUMINUS 13
#define CLSIMHIP_PARTICLE_MUPLUS (-13)
#define CLSIMHIP_PARTICLE_TAUMINUS 15
#define CLSIMHIP_PARTICLE_TAUPLUS (-15)
#define CLSIMHIP_PARTICLE_PI0 111
#define CLSIMHIP_PARTICLE_PIPLUS 211
#define CLSIMHIP_PARTICLE_PIMINUS (-211)
#define CLSIMHIP_PARTICLE_K0_LONG 130
#define CLSIMHIP_PARTICLE_KPLUS 321
#define CLSIMHIP_PARTICLE_KMINUS (-321)
#define CLSIMHIP_PARTICLE_K0_SHORT 310
#define CLSIMHIP_PARTICLE_PPLUS 2212
#define CLSIMHIP_PARTICLE_PMINUS (-2212)
#define CLSIMHIP_PARTICLE_NEUTRON 2112
#define CLSIMHIP_PARTICLE_BREMS (-2000001001)
#define CLSIMHIP_PARTICLE_DELTAE (-2000001002)
#define CLSIMHIP_PARTICLE_PAIRPROD (-2000001003)
#define CLSIMHIP_PARTICLE_NUCLINT (-2000001004)
#define CLSIMHIP_PARTICLE_HADRONS (-2000001006)
#define CLSIMHIP_SHAPE_OTHER 0
#define CLSIMHIP_SHAPE_CASCADE_SEGMENT 1        /* I3Particle::CascadeSegment: steps spread uniformly over `length` (:342-357) */
typedef struct {
    int32_t type;                       /* CLSIMHIP_PARTICLE_* / any PDG code (unknown codes are treated as hadrons, :272-279) */
    int32_t shape;                      /* CLSIMHIP_SHAPE_* */
    double x, y, z, time;               /* [m, ns] */
    double dx, dy, dz;                  /* unit direction */
    double energy;                      /* [GeV] */
    double length;                      /* [m]; NaN: unknown (muons then get 2000 m, :377-380) */
    uint32_t identifier;                /* -> I3CLSimStep::identifier of its steps */
    uint32_t reserved;
} clsimhip_particle;
typedef struct {
    uint32_t photons_per_step;          /* photonsPerStep (200) */
    uint32_t high_photons_per_step;     /* highPhotonsPerStep (2000) */
    double use_high_photons_per_step_from;  /* useHighPhotonsPerStepStartingFromNumPhotons (1e9) */
    int32_t use_cascade_extension;      /* UseCascadeExtension (on) */
    int32_t reserved;
    double medium_density;              /* g/cm3; I3CLSimMediumProperties::GetMediumDensity(), 0.9216 for the IceCube media */
    uint64_t seed;                      /* of this library's generator (in the place of the I3RandomService) */
} clsimhip_ppc_config;
typedef struct clsimhip_ppc_converter clsimhip_ppc_converter;
/* SetWlenBias + SetMediumProperties + Initialize */
int clsimhip_ppc_create(const clsimhip_medium *medium, const clsimhip_function *wavelength_bias, const clsimhip_ppc_config *config,
                        clsimhip_ppc_converter **out);
void clsimhip_ppc_destroy(clsimhip_ppc_converter *p);
/* meanPhotonsPerMeterInLayer_[layer] (beta = 1, after the wavelength bias) */
int clsimhip_ppc_photons_per_meter(const clsimhip_ppc_converter *p, int layer, double *out);
/* EnqueueLightSource for n particles: one request per cascade, two per muon / tau (muon-like, then cascade-like steps),
 * ready for clsimhip_generate_steps[_device].  *n_out = requests the particles need; at most `capacity` are written. */
int clsimhip_ppc_enqueue(const clsimhip_ppc_converter *p, const clsimhip_particle *particles, size_t n,
                         clsimhip_step_request *requests_out, size_t capacity, size_t *n_out);
/* I3SimConstants::ShowerParameters as restated here: out = {a, b [m], emScale, emScaleSigma} */
int clsimhip_shower_parameters(int32_t particle_type, double energy_gev, double density_g_cm3, double out[4]);

/* ---- flasher step producer --------------------------------------------------------------
 * I3CLSimLightSourceToStepConverterFlasher (private/clsim/I3CLSimLightSourceToStepConverterFlasher.cxx): MakeSteps
 * :329-440 cuts a flasher pulse into steps of photons_per_step photons (zero length, beta 1, weight 1, the pulse's
 * spectrum as source type) and pads with dummy steps to the bunch granularity; FillStep :443-545 smears every step's
 * direction and time with three I3CLSimRandomValue distributions whose run-time parameter comes from the pulse.
 * One GPU lane makes one step, every step has its own MWC stream seeded from (seed, step index) -- the reference
 * samples with I3RandomService (phys-services, not part of the reference tree) in double precision, so parity is
 * product = oracle (oracle/stepgen_oracle.c) and distribution tests, as for the cascade/muon producer.
 * Kept from the reference: a pulse whose photons divide evenly into more than one step loses its last step to a
 * dummy step (:383-389 with :407-408). */
#define CLSIMHIP_DIST_CONSTANT 0                /* I3CLSimRandomValueConstant(): the run-time parameter itself */
#define CLSIMHIP_DIST_NORMAL 1                  /* FixParameter(NormalDistribution(), 0, value): value + parameter * Box-Muller
                                                   (random_value/I3CLSimRandomValueNormalDistribution.cxx:47-80) */
#define CLSIMHIP_DIST_UNIFORM 2                 /* Uniform(value, NaN): value + u * (parameter - value) (…Uniform.cxx:77-104) */
#define CLSIMHIP_DIST_FLASHER_TIME_PROFILE 3    /* python/I3CLSimRandomValueIceCubeFlasherTimeProfile.py: LED pulse shape for
                                                   the pulse width (parameter), an InterpolatedDistribution of 240 points */
typedef struct {
    int32_t kind;
    float value;                                /* NORMAL: mean; UNIFORM: from */
} clsimhip_distribution;
typedef struct {
    clsimhip_distribution polar;                /* angularProfileDistributionPolar, parameter = sigma_polar */
    clsimhip_distribution azimuthal;            /* angularProfileDistributionAzimuthal, parameter = sigma_azimuthal */
    clsimhip_distribution time_delay;           /* timeDelayDistribution, parameter = pulse_width */
    int32_t interpret_in_polar_coordinates;     /* interpretAngularDistributionsInPolarCoordinates (:497-541) */
    uint32_t photons_per_step;                  /* photonsPerStep_ (default 400, :46) */
    uint32_t max_bunch_size;                    /* maxBunchSize_ (default 512000, :47) */
    uint32_t bunch_size_granularity;            /* bunchSizeGranularity_ (default 512, :48) */
} clsimhip_flasher_config;
typedef struct {                                /* one I3CLSimFlasherPulse in the converter's queue (LightSourceData_t) */
    float x, y, z, time;
    float dx, dy, dz;                           /* GetDir(), direction of emission */
    float sigma_polar, sigma_azimuthal;         /* GetAngularEmissionSigmaPolar / Azimuthal [rad] */
    float pulse_width;                          /* GetPulseWidth [ns] */
    uint32_t identifier;
    uint32_t source_type;                       /* spectrumSourceTypeIndex_: wavelength generator of the pulse's spectrum */
    uint64_t num_photons_with_bias;             /* numPhotonsWithBias (EnqueueLightSource :243-262, computed by the caller) */
} clsimhip_flasher_request;                     /* 56 bytes */
/* The converter's front end (Initialize :120-159, EnqueueLightSource :214-265): photons after the wavelength bias =
 * GetNumberOfPhotonsNoBias() x PhotonNumberCorrectionFactorAfterBias (ConverterUtils.cxx:113-214: the bias at the peak for a
 * delta-peak spectrum -- spectrum_no_bias NULL, peak_wavelength used -- else the ratio of the spectrum's integrals with and
 * without bias over [from, to]), then Poisson, or a non-negative Gaussian above 1e6; pulses that end up without photons
 * are skipped.  The random numbers are this library's (one counter-based stream per identifier), as for clsimhip_ppc_*. */
typedef struct {
    float x, y, z, time;
    float dx, dy, dz;
    float sigma_polar, sigma_azimuthal;
    float pulse_width;
    uint32_t identifier;
    uint32_t source_type;
    double num_photons_no_bias;                 /* I3CLSimFlasherPulse::GetNumberOfPhotonsNoBias() */
} clsimhip_flasher_pulse;
int clsimhip_flasher_correction_factor(const clsimhip_function *spectrum_no_bias, double peak_wavelength,
                                       const clsimhip_function *wavelength_bias, double from_wavelength, double to_wavelength, double *out);
int clsimhip_flasher_enqueue(double correction_factor, uint64_t seed, const clsimhip_flasher_pulse *pulses, size_t n,
                             clsimhip_flasher_request *requests_out, size_t capacity, size_t *n_out);
/* number of output steps (real + dummy) and of steps that carry photons */
int clsimhip_count_flasher_steps(const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests, size_t n,
                                 size_t *steps_out, size_t *real_steps_out);
/* generates into device memory d_steps (room for `capacity` steps) on `hip_stream` */
int clsimhip_generate_flasher_steps_device(int device, const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests,
                                           size_t n, uint64_t seed, void *d_steps, size_t capacity, void *hip_stream, size_t *steps_out);
/* the same into host memory (synchronous) */
int clsimhip_generate_flasher_steps(int device, const clsimhip_flasher_config *config, const clsimhip_flasher_request *requests,
                                    size_t n, uint64_t seed, clsimhip_step *steps_out, size_t capacity, size_t *count_out);
/* the time delay distribution for one pulse width: 240 densities and cumulative values at 0.5 ns spacing (host only) */
int clsimhip_flasher_time_profile(double pulse_width_ns, float density[240], float cumulative[240]);

/* ---- step store ------------------------------------------------------------------------
 * I3CLSimStepStore (public/clsim/I3CLSimStepStore.h:44-320): steps sorted by photon count (one FIFO per count), with
 * the number of stored steps per identifier; the feeder thread of the reference cuts bunches from it
 * (I3CLSimLightSourceToStepConverterAsync.cxx:209-273).  Host memory only. */
typedef struct clsimhip_step_store clsimhip_step_store;
int clsimhip_step_store_create(size_t initial_bins, clsimhip_step_store **out);
void clsimhip_step_store_destroy(clsimhip_step_store *s);
/* insert_copy(step.GetNumPhotons(), step) for n steps (StepStore.h:266-283) */
int clsimhip_step_store_insert(clsimhip_step_store *s, const clsimhip_step *steps, size_t n);
int clsimhip_step_store_size(const clsimhip_step_store *s, size_t *out);
/* count(identifier) (StepStore.h:308-312) */
int clsimhip_step_store_count(const clsimhip_step_store *s, uint32_t identifier, uint32_t *out);
/* pop_bunch_to_vector(size, vect): ascending photon count, FIFO within a count; *popped <= size (StepStore.h:163-198) */
int clsimhip_step_store_pop_bunch(clsimhip_step_store *s, size_t size, clsimhip_step *out, size_t *popped);
/* pop_bunch_to_vector(size, vect, temp): exactly `size` steps, the remainder copies of *fill (StepStore.h:209-222) */
int clsimhip_step_store_pop_bunch_filled(clsimhip_step_store *s, size_t size, clsimhip_step *out, const clsimhip_step *fill);
/* numStepsWithDummyFill (Async.cxx:256): size of the padded last bunch before a barrier */
int clsimhip_step_store_size_with_dummy_fill(const clsimhip_step_store *s, size_t granularity, size_t *out);

/* ---- feeder: the worker thread of I3CLSimLightSourceToStepConverterAsync ----------------------
 * private/clsim/I3CLSimLightSourceToStepConverterAsync.cxx: EnqueueLightSource / EnqueueBarrier / BarrierActive /
 * MoreStepsAvailable / GetConversionResultWithBarrierInfo (:470-600) in front of WorkerThread_impl (:178-392): light sources
 * are converted one after the other (the PPC front end above + the GPU step producer take the place of the
 * parameterisation), their steps go through the step store, and bunches of max_bunch_size steps come out in ascending
 * photon count, each with the identifiers of the light sources that have completely left the store; the barrier flushes
 * the rest, padded with no-op steps to ((size / granularity) + 1) * granularity.  Queues of `queue_depth` entries (10 in
 * the reference) on both sides; one thread per feeder. */
typedef struct clsimhip_feeder clsimhip_feeder;
/* ppc may be NULL (only clsimhip_feeder_enqueue_steps is then accepted); it must outlive the feeder */
int clsimhip_feeder_create(const clsimhip_ppc_converter *ppc, int device, uint64_t seed, size_t max_bunch_size,
                           size_t bunch_size_granularity, size_t queue_depth, clsimhip_feeder **out);
void clsimhip_feeder_destroy(clsimhip_feeder *f);
int clsimhip_feeder_enqueue_light_source(clsimhip_feeder *f, const clsimhip_particle *particle);
/* a light source whose steps the caller made itself (the role of an I3CLSimLightSourcePropagator, e.g. Geant4) */
int clsimhip_feeder_enqueue_steps(clsimhip_feeder *f, uint32_t identifier, const clsimhip_step *steps, size_t n);
int clsimhip_feeder_enqueue_barrier(clsimhip_feeder *f);
int clsimhip_feeder_barrier_active(const clsimhip_feeder *f, int *out);
int clsimhip_feeder_more_steps_available(const clsimhip_feeder *f, int *out);
/* GetConversionResultWithBarrierInfoAndMarkers: waits up to timeout_ms (< 0: for ever; *got = 0 on timeout).  *steps (n
 * records) and *finished (n_finished identifiers) stay valid until clsimhip_feeder_release_result(f, *steps). */
int clsimhip_feeder_get_conversion_result(clsimhip_feeder *f, double timeout_ms, int *got, const clsimhip_step **steps, size_t *n,
                                          const uint32_t **finished, size_t *n_finished, int *barrier_was_reset);
int clsimhip_feeder_release_result(clsimhip_feeder *f, const clsimhip_step *steps);

/* ---- photon table maker -------------------------------------------------------------
 * I3CLSimStepToTableConverter (private/clsim/tabulator/I3CLSimStepToTableConverter.h:44-101): propagates steps with
 * the TABULATE variant of propKernel (propagation_kernel.c.cl:228-303, 755-785: fixed 42 absorption lengths, no
 * detector, a path sample every `step_length` metres) and fills a table over coordinates relative to a reference
 * particle.  4 axes, or 5 = TABULATE_IMPACT_ANGLE (StepToTableConverter.cxx:187-188: the fifth axis is the cosine of
 * the impact angle on the DOM, two random numbers per path sample); linear and power axes (inverse transform: sqrt, cbrt,
 * pow(x, 1/power), tabulator/Axis.cxx:150-171). */
#define CLSIMHIP_AXIS_LINEAR 0          /* clsim::tabulator::LinearAxis (tabulator/Axis.h:71-80) */
#define CLSIMHIP_AXIS_POWER 1           /* clsim::tabulator::PowerAxis  (tabulator/Axis.h:82-95) */
typedef struct {
    int32_t kind;
    double min, max;
    uint32_t n_bins;                    /* without the under-/overflow bins every axis gets (Axes.cxx:51-64) */
    uint32_t power;                     /* POWER: >= 1 (2: square-root spacing, 3: cube-root ...) */
} clsimhip_axis;
#define CLSIMHIP_AXES_SPHERICAL 0       /* SphericalAxes: r, azimuth [deg], cos(polar), delay time (spherical_coordinates.c.cl) */
#define CLSIMHIP_AXES_CYLINDRICAL 1     /* CylindricalAxes: rho, azimuth [rad], z, delay time (cylindrical_coordinates.c.cl) */
typedef struct {                        /* I3CLSimFunctionPolynomial (function/I3CLSimFunctionPolynomial.cxx:35-153) */
    int32_t n;
    const double *coefficients;         /* c0 + x*(c1 + x*(...)) */
    double range_min, range_max;        /* -inf / +inf: unbounded */
    double underflow, overflow;
} clsimhip_polynomial;
typedef struct clsimhip_tabulator clsimhip_tabulator;

/* I3CLSimStepToTableConverter::I3CLSimStepToTableConverter (StepToTableConverter.cxx:122-265).  wavelength_acceptance
 * biases the Cherenkov spectrum (generator 0) and weights nothing else; angular_acceptance is evaluated on the
 * photon's z direction cosine per path segment; reference_area and step_length enter Normalize().  (x, a): one MWC
 * stream per work item, `streams` a multiple of 256. */
int clsimhip_tabulator_create(int device, int axes_kind, const clsimhip_axis *axes, size_t n_axes, int store_squared_weights,
                              const clsimhip_medium *medium, const clsimhip_function *wavelength_acceptance,
                              const clsimhip_polynomial *angular_acceptance, double reference_area, double step_length,
                              const uint64_t *x, const uint32_t *a, size_t streams, clsimhip_tabulator **out);
void clsimhip_tabulator_destroy(clsimhip_tabulator *t);
const char *clsimhip_tabulator_last_error(const clsimhip_tabulator *t);
/* EnqueueSteps(steps, reference) (:272-285): reference = {x, y, z, time, dx, dy, dz} of the source particle.
 * Asynchronous; stream i of this bunch continues where stream i of the previous bunch stopped. */
int clsimhip_tabulator_enqueue_steps(clsimhip_tabulator *t, const clsimhip_step *steps, size_t n, const double reference[7]);
/* Finish() (:287-295): waits until every enqueued bunch is in the table */
int clsimhip_tabulator_finish(clsimhip_tabulator *t);
/* table maker tuning (see clsimhip_set_tuning; no result depends on it): "fast_kernels" 0 | 1 -- the instantiation with the
 * medium's proofs compiled in, measured slower for this kernel (0); "grid" -- workgroups of the launch, 0: automatic (0);
 * "standard_sampler" 0 | 1 -- 0: the generic path sampler also for a table of the reference's default shape (spherical, folded
 * azimuth, square-root distance and time axes, no squared weights), which otherwise runs the sampler specialised for it (1) */
int clsimhip_tabulator_set_tuning(clsimhip_tabulator *t, const char *key, long long value);
/* number of bins including under-/overflow bins, number of axes (4, or 5 with the impact angle), and the shape
 * (n_bins + 2 per axis; unused entries 0) */
int clsimhip_tabulator_get_shape(const clsimhip_tabulator *t, size_t *n_bins, size_t *n_dim, size_t shape[5]);
/* binContent_ (squared == 0) or squaredWeights_ as the float image WriteFITSFile stores (:595-686), before
 * (normalized == 0) or after Normalize() (:512-543) */
int clsimhip_tabulator_get_bin_content(clsimhip_tabulator *t, float *out, size_t n_bins, int squared, int normalized);
/* the double precision accumulators themselves */
int clsimhip_tabulator_get_bin_sums(clsimhip_tabulator *t, double *out, size_t n_bins, int squared);
/* Axis::GetBinEdges (Axis.cxx:62-74): n_bins + 1 edges of axis `axis` */
int clsimhip_tabulator_get_bin_edges(const clsimhip_tabulator *t, int axis, double *out, size_t cap);
/* [0] photons enqueued, [1] sum of photon weights (numPhotons*weight), [2] n_group, [3] n_phase of
 * GetMinimumRefractiveIndex (:96-120), [4] kernel time ms, [5] launches, [6] bins, [7] reserved */
int clsimhip_tabulator_get_statistics(clsimhip_tabulator *t, double out[8]);
/* WriteFITSFile(path, tableHeader) (:595-686): normalised bin content as the primary image (axis counts reversed, as
 * cfitsio/PyFITS store it), "HIERARCH _i3_<key>" header keywords -- n_photons (= spectralBiasFactor x sum of photon
 * weights), n_group, n_phase, then the caller's n_keys entries (is_int[i] ? int_values[i] : double_values[i]) --, the
 * squared weights as the IMAGE extension "ERRORS", one double IMAGE extension "EDGESi" per axis.  Written without cfitsio;
 * refuses to overwrite an existing file, like fits_create_diskfile. */
int clsimhip_tabulator_write_fits_file(clsimhip_tabulator *t, const char *path, const char *const *keys, const int32_t *is_int,
                                       const int64_t *int_values, const double *double_values, size_t n_keys);
int clsimhip_tabulator_get_rng_state(clsimhip_tabulator *t, uint64_t *x, size_t count);
long clsimhip_tabulator_get_table(const clsimhip_tabulator *t, const char *name, double *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
