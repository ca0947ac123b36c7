// Single functions of the propagator evaluated on the device: the counterpart of the reference's tester classes
// (private/test/I3CLSimFunctionTester, I3CLSimScalarFieldTester, I3CLSimVectorTransformTester,
// I3CLSimRandomDistributionTester, I3CLSimMediumPropertiesTester: each compiles the generated function behind a tiny
// kernel and hands the device's values back, resources/kernels/*_test_kernel.*), which the reference's own tests use to
// compare device with host (resources/tests/testScalarFields.py, testScalarFieldIceTiltZShift.py, testVectorTransforms.py).
// Here the functions are the ones the propagation kernels inline (prop_device.hip.h), read from the same table image, in
// both forms the kernels use them (FAST: the exact-reciprocal paths of the standard configuration; otherwise the IEEE
// sequences behind the wave-uniform range tests).  A test harness: nothing on the propagation path calls it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prop_device.hip.h"
#include "../../include/clsimhip.h"

namespace clsimhip {

template <int MED, bool FAST>
__global__ void __launch_bounds__(256) eval_function_kernel(const KParams Pvalue, int what, int layer, int has_tilt, const float4 *in, uint32_t n, float4 *out)
{
    const KP P = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Pvalue;
    {
        const uint32_t words = P->table_words;
        const uint32_t *src = P->tables;
        for (uint32_t i = threadIdx.x; i < words; i += 256) lds_words[i] = src[i];
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 v = in[i];
        float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        Vec3 d = {v.x, v.y, v.z};
        switch (what) {
        case CLSIMHIP_EVAL_LENGTHS: {            // x = wavelength: absorption length, scattering length of `layer`
            const IceFactors f = ice_factors<MED>(P, v.x);
            float sca = 0.0f, ab = 0.0f, rcp_sca = 0.0f, rcp_ab = 0.0f;
            const bool bounded = FAST || (P->div_ok & kFastLengths) != 0u;
            layer_lengths<MED, FAST>(P->off_layers, (MED == CLSIMHIP_LENGTHS_TABLE) ? P->len_table : nullptr, f, layer, sca, ab, rcp_sca, rcp_ab, bounded);
            if (!((MED == CLSIMHIP_LENGTHS_ICECUBE) && bounded)) { rcp_sca = rcp_t<FAST>(sca, bounded); rcp_ab = rcp_t<FAST>(ab, bounded); }
            r.x = ab; r.y = sca; r.z = rcp_ab; r.w = rcp_sca;        // (z, w: RN(1 / length): from the length's argument where the layer walk takes it that way)
            break;
        }
        case CLSIMHIP_EVAL_REFRACTION:           // x = wavelength: phase refractive index, group velocity
            r.x = phase_ref_index(P, v.x);
            r.y = group_velocity(P, v.x);
            break;
        case CLSIMHIP_EVAL_WAVELENGTH_BIAS:
            r.x = wavelength_bias(P, v.x);
            break;
        case CLSIMHIP_EVAL_TILT:                 // x, y, z = position
            r.x = has_tilt ? tilt_z_shift<FAST>(P, v.x, v.y, v.z) : P->tilt_const;      // ScalarFieldConstant: getTiltZShift_IS_CONSTANT
            break;
        case CLSIMHIP_EVAL_ABS_LEN_SCALING:      // x, y, z = direction
            r.x = P->has_abs_corr ? abs_len_corr<FAST>(P, d) : 1.0f;
            break;
        case CLSIMHIP_EVAL_PRE_SCATTER_TRANSFORM:
            if (P->has_pre) apply_matrix(P->pre, P->pre_renorm, d, FAST || (P->div_ok & kFastMatrices) != 0u);
            r = make_float4(d.x, d.y, d.z, 0.0f);
            break;
        case CLSIMHIP_EVAL_POST_SCATTER_TRANSFORM:
            if (P->has_post) apply_matrix(P->post, P->post_renorm, d, FAST || (P->div_ok & kFastMatrices) != 0u);
            r = make_float4(d.x, d.y, d.z, 0.0f);
            break;
        default: break;
        }
        out[i] = r;
    }
}

// one stream per work item, `draws` values each, out[stream * draws + k] (RandomDistributionTester.cxx:43-199)
template <bool FAST>
__global__ void __launch_bounds__(256) eval_random_kernel(const KParams Pvalue, int what, int generator, uint64_t *x, const uint32_t *a,
                                                          uint32_t n_streams, uint32_t draws, float *out)
{
    const KP P = (KP)__builtin_amdgcn_kernarg_segment_ptr();
    (void)Pvalue;
    {
        const uint32_t words = P->table_words;
        const uint32_t *src = P->tables;
        for (uint32_t i = threadIdx.x; i < words; i += 256) lds_words[i] = src[i];
    }
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n_streams; i += gridDim.x * 256) {
        uint64_t rx = x[i];
        const uint32_t ra = a[i];
        for (uint32_t k = 0; k < draws; ++k) {
            float v = 0.0f;
            if (what == CLSIMHIP_EVAL_RANDOM_UNIFORM) v = rng_co(rx, ra);
            else if (what == CLSIMHIP_EVAL_RANDOM_WAVELENGTH) v = generate_wavelength(P, generator, rx, ra);
            else if (what == CLSIMHIP_EVAL_RANDOM_SCATTERING_COSINE) v = scattering_cos<FAST>(P, rx, ra);
            out[(size_t)i * draws + k] = v;
        }
        x[i] = rx;
    }
}

hipError_t launch_eval_function(const KParams &P, int lengths_kind, bool has_tilt, bool fast, int what, int layer, const float4 *in, uint32_t n, float4 *out,
                                hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    const size_t lds = (size_t)P.table_words * 4;
    const dim3 grid((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), block(256);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    // (an image beyond 64 KB -- a detector of several hundred strings -- needs the attribute, like the propagation kernels' launch_variant)
#define BIG(kernel) \
    if (lds > 64u * 1024u) { const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; }
#define GO(m) \
    if (fast) { BIG((eval_function_kernel<m, true>)) hipLaunchKernelGGL((eval_function_kernel<m, true>), grid, block, lds, stream, P, what, layer, (int)has_tilt, in, n, out); } \
    else { BIG((eval_function_kernel<m, false>)) hipLaunchKernelGGL((eval_function_kernel<m, false>), grid, block, lds, stream, P, what, layer, (int)has_tilt, in, n, out); }
    switch (lengths_kind) {
    case CLSIMHIP_LENGTHS_CONSTANT: GO(CLSIMHIP_LENGTHS_CONSTANT) break;
    case CLSIMHIP_LENGTHS_ICECUBE: GO(CLSIMHIP_LENGTHS_ICECUBE) break;
    case CLSIMHIP_LENGTHS_TABLE: GO(CLSIMHIP_LENGTHS_TABLE) break;
    default: return hipErrorInvalidValue;
    }
#undef GO
    return hipGetLastError();
}

hipError_t launch_eval_random(const KParams &P, bool fast, int what, int generator, uint64_t *x, const uint32_t *a, uint32_t n_streams, uint32_t draws,
                              float *out, hipStream_t stream)
{
    if (n_streams == 0 || draws == 0) return hipSuccess;
    const size_t lds = (size_t)P.table_words * 4;
    const dim3 grid((n_streams + 255) / 256 < 1024 ? (n_streams + 255) / 256 : 1024), block(256);
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (fast) { BIG((eval_random_kernel<true>)) hipLaunchKernelGGL((eval_random_kernel<true>), grid, block, lds, stream, P, what, generator, x, a, n_streams, draws, out); }
    else { BIG((eval_random_kernel<false>)) hipLaunchKernelGGL((eval_random_kernel<false>), grid, block, lds, stream, P, what, generator, x, a, n_streams, draws, out); }
    return hipGetLastError();
#undef BIG
}

} // namespace clsimhip
