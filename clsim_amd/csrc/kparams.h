// Kernel parameter block: everything wave-uniform the propagator needs.
//
// It is passed by value, i.e. it lives in the kernarg segment; the kernel reads
// it through a constant-address-space pointer with scalar loads (s_load) close
// to the use, phase by phase, instead of holding ~150 values in SGPRs for the
// whole kernel (which spills SGPRs into VGPR lanes on gfx950).  Arrays that
// lanes index divergently (ice layers, tilt grid, spectra, DOM cell index) live
// in one 32-bit word image that every workgroup copies from HBM/L2 into LDS
// once; `off_*` are word offsets into it.  Per-entity values are interleaved
// into records so that one wide LDS read fetches what a test needs.
//
// The reference bakes the same values into the OpenCL source as #defines and
// __constant arrays (MediumPropertiesSource.cxx:207-389, GeometrySource.cxx:1153-1269).
#pragma once
#include <stdint.h>

namespace clsimhip {

constexpr int kMaxGenerators = 8;
constexpr int kMaxSubdetectors = 9;     // sparse_collision_kernel.c.cl:455

#pragma pack(push, 1)
struct DevStep {                        // I3CLSimStep, 48 B
    float x, y, z, t;
    float theta, phi, length, beta;
    uint32_t num_photons;
    float weight;
    uint32_t identifier;
    uint32_t source_type_and_pad;       // low byte = sourceType
};
struct DevPhoton {                      // I3CLSimPhoton, 80 B = 20 words
    uint32_t w[20];
};
#pragma pack(pop)
// One 64-byte line per step: everything a lane needs to take over a slice of the step (the step itself, the state of
// its RNG stream, the multiplier, how many slices have been published).  Built by scan_steps_kernel at the start of a
// launch; a hand-off reads and writes this one line instead of four arrays.
constexpr int kSubQueues = 8;            // sub-queue q hands out the slices of the steps i with i % kSubQueues == q
constexpr int kQueueHeadStride = 32;     // words between sub-queue heads (128 B)
constexpr int kQueueWords = 512;         // per launch

struct WorkRecord {
    DevStep step;           // as photon creation needs it: theta, phi, weight hold the step's direction (scan_steps_kernel)
    uint64_t x;
    uint32_t a;
    uint32_t done;
};
static_assert(sizeof(WorkRecord) == 64, "work record");
static_assert(sizeof(DevStep) == 48, "step record");
static_assert(sizeof(DevPhoton) == 80, "photon record");

struct KParams {
    // ---- buffers ----
    const uint32_t *tables;             // LDS image (table_words words)
    uint32_t table_words;
    const DevStep *steps;
    uint32_t n_steps;
    uint64_t *rng_x;
    const uint32_t *rng_a;
    DevPhoton *out;
    uint32_t *hit_count;
    uint32_t max_hits;
    // work queue block of this launch (kQueueWords words, zeroed before the launch): [1] max numPhotons, [2] skipped
    // steps, [3] debug; the heads of the kSubQueues sub-queues sit on cache lines of their own at [kQueueHeadStride*(q+1)]
    uint32_t *queue;
    int32_t k_new;                      // lanes that must be waiting before photons are created
    int32_t k_search;                   // lanes that must be parked before the wave runs the DOM search ...
    int32_t k_wait;                     // ... or trips the first of them has waited (whichever comes first)
    int32_t k_aim;                      // segment_misses_string is asked when at most this many lanes of the wave reach a string's cylinder
    int32_t slices;                     // a step is handed out in this many slices (1 = whole steps)
    // pooled kernel (prop_pool_kernel.hip): entries of a wave's ring of ready photons, and how many lanes must be
    // without a photon before the wave services them; k_new is its creation batch there (0 = automatic everywhere)
    int32_t pool_ready, k_pop;
    int32_t chip_share;                 // launch geometry only: this launch may fill 1/chip_share of the chip (0, 1: all of it)
#ifdef CLSIMHIP_CENSUS
    unsigned long long *census;         // analysis build (make EXTRA=-DCLSIMHIP_CENSUS): [0..7] lane-state sums, [8] earliest
                                        // wave start, [16 + 3w ...] per wave: end time, time the first sub-queue was found dry, trips
#endif
    WorkRecord *work;                   // per step, filled by scan_steps_kernel (done = slices published so far)
    // TABLE lengths: one 16-byte record per (wavelength bin, layer): {abs[bin], abs[bin+1], sca[bin], sca[bin+1]}
    // at len_table[4*(bin*num_layers + layer)], already de-quantised (80 KB for a 171 x 30 photonics table: HBM/L2)
    const float *len_table;
    // SAVE_PHOTON_HISTORY (c.cl:452-455, 833-837, 387-392): ring of the last history_n scatter points per lane
    // (float4: x, y, z, absorption lengths left), copied to hist_out[slot * history_n ..] when the lane's photon is detected
    float *hist_ring;
    float *hist_out;
    int32_t history_n;
    int32_t has_fixed_abs;              // PROPAGATE_FOR_FIXED_NUMBER_OF_ABSORPTION_LENGTHS (c.cl:582-588)
    float fixed_abs;
    // String proximity map: prox_n x prox_n words over the xy bounding box of the string axes.  Bits 0-7 * 0.25 m: a proven
    // lower bound of the xy distance from anywhere in that cell to the surface of the nearest string cylinder (axis + largest
    // DOM offset + OM radius = prox_reach).  A step shorter than the bound cannot reach a DOM: the DOM search, which would find
    // nothing, is skipped.  Bits 8-15: the same bound for the SECOND nearest string; bits 16-31: the index of the nearest one
    // (0xffff: no string) -- a step shorter than the second bound can touch that string only, and only if its xy projection comes
    // within prox_reach of its axis (prop_device.hip.h: segment_misses_string).  512 x 512 words = 1 MB, L2 resident.
    const uint32_t *prox_map;
    int32_t prox_n;
    float prox_x0, prox_y0, prox_inv_cell, prox_reach;
    // DOM proximity map, the second level of the search filter: dprox_nx x dprox_ny x dprox_nz words (cubic cells, z fastest) over
    // the bounding box of the DOMs.  A word names the DOM nearest to the cell (bits 0-15: index into dom_centres, 0xffff =
    // none within 64 m) and carries in bits 16-23, in 0.25 m units, a proven lower bound of the 3D distance from anywhere
    // in the cell to the sphere of any OTHER DOM.  A step shorter than that bound can only touch the named DOM, and does
    // so only if the segment comes within its radius (prop_device.hip.h: dom_search_needed).  Consulted only by lanes whose step reaches a
    // string cylinder: most of them pass between two DOMs of the string (17 m apart, 0.8 m radius), and photons born
    // at a DOM (flashers) spend their lives within metres of it.  <= 256^3 cells: 64 MB as host words, 256 MB as the 16-byte device cells
    // below; the cells in use (the columns around the strings, a few MB) are L2 / MALL resident.
    // (round 5) On the device a cell is 16 bytes: {that word, x, y, z of the named DOM's centre} -- one load where rounds 2-4 made two
    // dependent ones (the word, then dom_centres[id]); 4 x the bytes, of which the columns around the strings (a few MB) are ever read.
    const uint4 *dom_cells;
    const float4 *dom_centres;          // x, y, z of every DOM as dom_position() reconstructs it, w = 0
    // Where the reference's search would meet that DOM (prop_device.hip.h: find_collision_named): x = string index | DOM number
    // in the string << 16; y = first cell column | first cell row << 12 | subdetector << 24 and w = last column | last row << 12 of
    // the rectangle of cells the string lies in (its bounding square may overlap several, GeometrySource.cxx:135-271);
    // z = first | last << 16 of the z layers of the string's layering that hold this DOM
    const uint4 *dom_named;
    // Host path only (null on the device path, whose records keep the kernel's indices): string index -> string ID and
    // (string index, DOM index) -> OM ID, applied to the hit records by assemble_hits_kernel (the reference converts on the
    // host, one photon after the other, OpenCL.cxx:1565-1619)
    const int16_t *id_strings;
    const uint16_t *id_doms;
    const uint32_t *id_dom_start;
    int32_t dprox_nx, dprox_ny, dprox_nz;
    float dprox_x0, dprox_y0, dprox_z0, dprox_inv_cell;
    float dprox_radius;                 // OM radius + safety
    const int16_t *dom_tx;              // DOM templates stay in HBM/L2 (41 KB for IC86)
    const int16_t *dom_ty;
    const float *dom_tz;
    int32_t dom_in_lds;                 // templates are part of the LDS image
    uint32_t off_dom_xy, off_dom_z;     // int16 pairs (x | y<<16), float z

    // ---- medium ----
    int32_t num_layers;
    float layer_bottom, layer_thickness, recip_thickness;
    // 4-word records per layer.  ICECUBE: {(D*aDust+E), (1+0.01*dTau), b400, 0}; CONSTANT: {abs, 0, sca, 0}
    uint32_t off_layers;
    float neg_kappa, abs_A, neg_B, neg_alpha, ref_wlen_recip, nanometer;
    float n[5], g[5], micrometer, c_light;
    int32_t len_tab_n;                  // TABLE lengths: common binning of the per-layer FromTable functions
    float len_tab_start, len_tab_step;
    int32_t phase_kind, group_kind;     // CLSIMHIP_REFINDEX_*; TABLE: float data in the LDS image
    int32_t phase_n, group_n;
    float phase_start, phase_step, group_start, group_step;
    uint32_t off_phase, off_group;
    float mix_frac, mix_frac_rest, liu_beta, hg_g, hg_one_minus_g2, hg_one_plus_g2, hg_two_g;
    float an_l[3], an_rl[3], an_azx, an_azy, an_mazy, an_B2, abs_corr_const;
    float pre[9], post[9];
    int32_t has_abs_corr, has_pre, has_post, pre_renorm, post_renorm;
    int32_t scatter_kind;               // CLSIMHIP_SCATTER_*
    float tilt_const;
    int32_t tilt_nd, tilt_nz;
    float tilt_first_z, tilt_dz, tilt_lnx, tilt_lny;
    uint32_t off_tilt_dist, off_tilt_zcorr;
    // 4-word records per distance bin j = 1..nd-1 at off_tilt_bins + 4*j: {dist[j], dist[j]-dist[j-1], its reciprocal, ok}
    uint32_t off_tilt_bins;
    float tilt_inner_dist[6];           // dist[1..nd-2], padded with +inf (used when nd <= 8)

    // ---- exact division by invariant divisors ----
    // For a divisor b that never changes, q = a*r; q' = fma(fma(-b,q,a), r, q) with r = RN(1/b) is the
    // correctly rounded quotient a/b for all a except for rare divisors (Brisebarre, Muller, Raina 2004).
    // Compile() proves it per divisor by trying every significand of a; bit set = proven, else the kernel
    // keeps the IEEE divide.  3 VALU instructions instead of ~11.
    float rcp_tilt_dz, rcp_layer_thickness, rcp_mix_frac, rcp_mix_frac_rest, rcp_hg_two_g;
    uint32_t div_ok;                    // bit 0 tilt_dz, 1 layer_thickness, 2 mix_frac, 3 mix_frac_rest, 4 hg_two_g, 5 lengths within [1e-15, 1e15] (dm::rcp_ allowed)

    // ---- spectra ----
    int32_t num_gen;
    int32_t gen_kind[kMaxGenerators], gen_n[kMaxGenerators];
    float gen_first[kMaxGenerators], gen_spacing[kMaxGenerators], gen_value[kMaxGenerators];
    uint32_t off_gen_yv[kMaxGenerators], off_gen_ycum[kMaxGenerators];
    uint32_t off_gen_xv[kMaxGenerators];        // kind 3 (InterpolatedDistribution with its own x values): _distXValues
    int32_t bias_kind, bias_n;
    float bias_start, bias_step, bias_value;
    uint32_t off_bias;

    // ---- TABULATE variant (propagation_kernel.c.cl:228-303, 755-785; Axes.cxx; spherical/cylindrical_coordinates.c.cl) ----
    double *tab_bins;                   // one accumulator per table bin (+ under/overflow bins), atomically added
    double *tab_sq_bins;                // squared weights, or null
    float tab_ref[12];                  // I3CLSimReferenceParticle: posAndTime, dir, perpDir (copied to LDS by the kernel)
    // 25-word record in the LDS image: scale[4], offset[4], nbins[4], stride[4], inverse[4], max0, max3,
    // min_inv_groupvel, tan_thetac, volume_step -- read from LDS inside the sampling loop (scalar loads there
    // stall on every use: the loop's atomics make the compiler reload kernel arguments each trip)
    uint32_t off_tab;
    int32_t tab_axes_kind, tab_full_azimuth;
    int32_t tab_ndim;                   // 4, or 5 = TABULATE_IMPACT_ANGLE (fifth axis: cosine of the impact angle)
    float tab_scale[5], tab_offset[5];  // Axis::GetIndexCode literals
    int32_t tab_inverse[5], tab_nbins[5];       // tab_inverse: the axis' power (0, 1: identity, 2: sqrt, 3: cbrt, above: pow(x, tab_inv_exp))
    float tab_inv_exp[5];                       // ToFloatString(1./power), Axis.cxx:168
    uint32_t tab_stride[5];
    // round 5: four-axis tables are kept TILED on the device -- eight bins (tab_tile_bits) of axes 0, 2, 3 (distance, polar angle, time) share one
    // 64-byte sector, so that a photon path's consecutive samples meet fewer sectors (tabulator.cpp: tiled_; sample_bin) -- and put into
    // the reference's order when the table is read.  tab_tiled = 0: the reference's order (five axes; CLSIMHIP_TAB_LAYOUT=linear).
    uint32_t tab_tiled, tab_tile_stride[3];     // strides of b0 >> e0, b1, b2 >> e2 (b3 >> e3 has stride 8)
    uint32_t tab_tile_bits[3];                  // e0, e2, e3: a sector holds 2^e0 x 2^e2 x 2^e3 bins of axes 0, 2, 3 (e0 + e2 + e3 = 3)
    // (round 6) the standard table: spherical axes, folded azimuth, square-root axes 0 and 3, identity axes 1 and 2, tiled 4 x 2 x 1, no squared
    // weights -- the four-axis kernel then runs the sampler specialised for it (prop_kernel.hip: sample_bin<..., STD>); set by tabulator.cpp
    uint32_t tab_std;
    float tab_max0, tab_max3, tab_min_inv_groupvel, tab_tan_thetac, tab_volume_step;
    int32_t ang_n;                      // getAngularAcceptance polynomial (coefficients in the LDS image)
    uint32_t off_ang;
    int32_t ang_has_min, ang_has_max;
    float ang_min, ang_max, ang_underflow, ang_overflow;

    // ---- detector ----
    int32_t has_pancake;
    float pancake, unpancake;           // PANCAKE_FACTOR, (PANCAKE_FACTOR-1)/PANCAKE_FACTOR
    float om_radius, om_radius_sq, string_max_radius_sq, string_max_radius;
    int32_t num_strings, num_sets, max_layers, num_subdet;
    // 8-word records per string: {x, y, maxZ+R, minZ-R, set | dom_start<<8, dom mean x, dom mean y, 0}
    uint32_t off_strings;
    // 4-word records per string set: {number of z layers, start z, layer height, 0}
    uint32_t off_sets;
    uint32_t off_layer_to_om;           // uint16 pairs
    // per subdetector, 8 words in LDS: nx, ny, width_x, width_y, start_x, start_y, offset of its
    // cell index (uint16 pairs), unused -- lanes sit in different subdetectors, so this is indexed per lane
    uint32_t off_subdet;
    float dom_mul_x, dom_mul_y;
    // pooled kernel: the four thresholds a loop trip consults, in ONE word read once before the loop (round 6: each was a scalar load
    // with its own wait in every trip): k_pop | k_search << 8 | k_aim << 16 | k_wait << 24, every field below 256 (the launcher packs them)
    uint32_t k_packed;
};

// kernel variants (the reference's #ifdef switches, OpenCL.cxx:390-442 and the
// generated *_IS_CONSTANT / NO_FLASHER hints)
struct KVariant {
    int lengths;            // CLSIMHIP_LENGTHS_*: per-layer constants, optimised IceCube abs/scat functions, per-layer tables
    bool tilt;              // ScalarFieldIceTiltZShift vs getTiltZShift_IS_CONSTANT
    bool aniso;             // anisotropy scaling + pre/post transforms present
    bool flasher;           // more than one wavelength generator (no NO_FLASHER)
    bool tabulate = false;  // TABULATE: record path samples into table bins instead of looking for DOMs
    bool keep_detected = false;     // no STOP_PHOTONS_ON_DETECTION: every DOM on a segment's way is saved, the photon travels on
    bool fast = false;      // standard configuration, every proof in hand (prop_device.hip.h: FAST): the pooled kernel runs the
                            // instantiation without the wave-uniform tests of those facts
    // launch tuning (clsimhip_set_tuning / clsimhip_tabulator_set_tuning; 0 / false = automatic).  Never part of a result.
    int grid = 0;                   // workgroups of the propagation launch ("grid")
    bool generic_only = false;      // the generic instantiation also where Compile() found every proof ("generic_kernels")
    bool tab_fast = false;          // table maker: the FAST instantiation (measured slower, prop_kernel.hip: launch_tab_kernel) ("fast_kernels")
};

} // namespace clsimhip
