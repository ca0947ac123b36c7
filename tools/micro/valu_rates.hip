// Issue cost of single vector instructions on gfx950, wave64 (ANALYSIS TOOL; hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip).
// One kernel per opcode: 64 copies of the instruction per loop iteration on eight independent destination registers (inline assembly, so the opcode is
// what is measured), every lane active, 4 and 8 waves per SIMD on every SIMD of the chip.  Prints shader cycles per instruction and SIMD, with the
// clock taken as 2.4 GHz (rocprofv3 --pmc GRBM_GUI_ACTIVE of these kernels: 2.3-2.4).  What it is for: the price list behind the vector roofline
//  -- the guide's fp32 peak assumes two cycles per wave64 instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// eight instructions on eight destinations; `INS(d, a, b)` yields the assembly text for destination %d and two vector sources
#define EIGHT(INS)                                                                                                                  \
    asm volatile(INS("%0", "%8", "%9") "\n" INS("%1", "%8", "%9") "\n" INS("%2", "%8", "%9") "\n" INS("%3", "%8", "%9") "\n"            \
                 INS("%4", "%8", "%9") "\n" INS("%5", "%8", "%9") "\n" INS("%6", "%8", "%9") "\n" INS("%7", "%8", "%9")                 \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc")

// the same with a third input: a wave-uniform value in a scalar register (%10)
#define EIGHT_S(INS)                                                                                                                \
    asm volatile(INS("%0", "%8", "%10") "\n" INS("%1", "%8", "%10") "\n" INS("%2", "%8", "%10") "\n" INS("%3", "%8", "%10") "\n"        \
                 INS("%4", "%8", "%10") "\n" INS("%5", "%8", "%10") "\n" INS("%6", "%8", "%10") "\n" INS("%7", "%8", "%10")             \
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "s"(fa) : "vcc")
#define KERNEL_S(NAME, INS)                                                                                                         \
    __global__ void __launch_bounds__(256) NAME(float *out, int iterations, float fa, float fb)                                     \
    {                                                                                                                               \
        float x0 = (float)threadIdx.x + 1.0f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f; \
        float a = fa + (float)(threadIdx.x & 1u) * 1e-9f, b = fb + (float)(threadIdx.x & 2u) * 1e-9f;                                \
        for (int it = 0; it < iterations; ++it) { EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); EIGHT_S(INS); } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));                              \
    }

#define KERNEL(NAME, INS)                                                                                                           \
    __global__ void __launch_bounds__(256) NAME(float *out, int iterations, float fa, float fb)                                     \
    {                                                                                                                               \
        float x0 = (float)threadIdx.x + 1.0f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f; \
        float a = fa + (float)(threadIdx.x & 1u) * 1e-9f, b = fb + (float)(threadIdx.x & 2u) * 1e-9f;                                \
        for (int it = 0; it < iterations; ++it) { EIGHT(INS); EIGHT(INS); EIGHT(INS); EIGHT(INS); EIGHT(INS); EIGHT(INS); EIGHT(INS); EIGHT(INS); } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));                              \
    }

#define I_MUL_F32(d, a, b) "v_mul_f32 " d ", " d ", " a
#define I_ADD_F32(d, a, b) "v_add_f32 " d ", " d ", " b
#define I_FMA_F32(d, a, b) "v_fma_f32 " d ", " d ", " a ", " b
#define I_FMAC_F32(d, a, b) "v_fmac_f32 " d ", " a ", " b
#define I_MOV_B32(d, a, b) "v_mov_b32 " d ", " a
#define I_ADD_U32(d, a, b) "v_add_u32 " d ", " d ", " a
#define I_AND_B32(d, a, b) "v_and_b32 " d ", " d ", " a
#define I_XOR_B32(d, a, b) "v_xor_b32 " d ", " d ", " a
#define I_LSHL_ADD(d, a, b) "v_lshl_add_u32 " d ", " d ", 1, " a
#define I_CNDMASK(d, a, b) "v_cndmask_b32 " d ", " d ", " a ", vcc"
#define I_CMP_F32(d, a, b) "v_cmp_lt_f32 vcc, " d ", " a
#define I_MAX_F32(d, a, b) "v_max_f32 " d ", " d ", " a
#define I_MED3_F32(d, a, b) "v_med3_f32 " d ", " d ", " a ", " b
#define I_MUL_LO_U32(d, a, b) "v_mul_lo_u32 " d ", " d ", " a
#define I_MAD_U64(d, a, b) "v_mul_hi_u32 " d ", " d ", " a
#define I_RCP_F32(d, a, b) "v_rcp_f32 " d ", " d
#define I_SQRT_F32(d, a, b) "v_sqrt_f32 " d ", " d
#define I_CVT_F32_U32(d, a, b) "v_cvt_f32_u32 " d ", " d
#define I_CVT_I32_F32(d, a, b) "v_cvt_i32_f32 " d ", " d
#define I_LDEXP(d, a, b) "v_ldexp_f32 " d ", " d ", 1"
#define I_DPP_MOV(d, a, b) "v_mov_b32_dpp " d ", " a " row_shr:1 row_mask:0xf bank_mask:0xf"

#define I_MUL_F32_S(d, a, sreg) "v_mul_f32 " d ", " sreg ", " d
#define I_ADD_F32_S(d, a, sreg) "v_add_f32 " d ", " sreg ", " d
#define I_FMA_F32_S(d, a, sreg) "v_fma_f32 " d ", " d ", " sreg ", " a
#define I_FMAC_F32_S(d, a, sreg) "v_fmac_f32 " d ", " sreg ", " a
#define I_MUL_F32_LIT(d, a, sreg) "v_mul_f32 " d ", 0x3f800001, " d
#define I_MUL_F32_INL(d, a, sreg) "v_mul_f32 " d ", 1.0, " d
#define I_FMAAK(d, a, sreg) "v_fmaak_f32 " d ", " d ", " a ", 0x3089705f"
#define I_ADD_U32_S(d, a, sreg) "v_add_u32 " d ", " sreg ", " d
#define I_MOV_B32_S(d, a, sreg) "v_mov_b32 " d ", " sreg
#define I_CNDMASK_S(d, a, sreg) "v_cndmask_b32 " d ", " d ", " a ", vcc"
KERNEL_S(k_mul_f32_s, I_MUL_F32_S) KERNEL_S(k_add_f32_s, I_ADD_F32_S) KERNEL_S(k_fma_f32_s, I_FMA_F32_S) KERNEL_S(k_fmac_f32_s, I_FMAC_F32_S)
KERNEL_S(k_mul_f32_lit, I_MUL_F32_LIT) KERNEL_S(k_mul_f32_inl, I_MUL_F32_INL) KERNEL_S(k_fmaak, I_FMAAK) KERNEL_S(k_add_u32_s, I_ADD_U32_S) KERNEL_S(k_mov_b32_s, I_MOV_B32_S)

// v_cndmask_b32 with a mask that was written (a compare in front of every eight selects)
__global__ void __launch_bounds__(256) k_cndmask_fresh(float *out, int iterations, float fa, float fb)
{
    float x0 = (float)threadIdx.x + 1.0f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    float a = fa + (float)(threadIdx.x & 1u) * 1e-9f, b = fb + (float)(threadIdx.x & 2u) * 1e-9f;
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_cmp_lt_f32 vcc, %8, %9\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// v_cndmask_b32 under masks of different population: the select's mask is (lane < TRUE_LANES), written once before the loop (s_mov into vcc) or
// by a compare in front of every eight selects
template <bool FRESH>
__global__ void __launch_bounds__(256) k_cndmask_pop(float *out, int iterations, float fa, float fb, int true_lanes)
{
    float x0 = (float)threadIdx.x + 1.0f, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    float a = fa + (float)(threadIdx.x & 1u) * 1e-9f;
    const float lane = (float)(threadIdx.x & 63u), limit = (float)true_lanes;
    const unsigned long long m = true_lanes >= 64 ? ~0ull : ((1ull << true_lanes) - 1ull);
    asm volatile("s_mov_b64 vcc, %0" :: "s"(m) : "vcc");
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FRESH)
                asm volatile("v_cmp_lt_f32 vcc, %9, %10\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(lane), "v"(limit) : "vcc");
            else
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(lane), "v"(limit));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7)) + fb;
}

KERNEL(k_mul_f32, I_MUL_F32) KERNEL(k_add_f32, I_ADD_F32) KERNEL(k_fma_f32, I_FMA_F32) KERNEL(k_fmac_f32, I_FMAC_F32) KERNEL(k_mov_b32, I_MOV_B32)
KERNEL(k_add_u32, I_ADD_U32) KERNEL(k_and_b32, I_AND_B32) KERNEL(k_xor_b32, I_XOR_B32) KERNEL(k_lshl_add, I_LSHL_ADD) KERNEL(k_cndmask, I_CNDMASK)
KERNEL(k_cmp_f32, I_CMP_F32) KERNEL(k_max_f32, I_MAX_F32) KERNEL(k_med3_f32, I_MED3_F32) KERNEL(k_mul_lo_u32, I_MUL_LO_U32) KERNEL(k_mul_hi_u32, I_MAD_U64)
KERNEL(k_rcp_f32, I_RCP_F32) KERNEL(k_sqrt_f32, I_SQRT_F32) KERNEL(k_cvt_f32_u32, I_CVT_F32_U32) KERNEL(k_cvt_i32_f32, I_CVT_I32_F32) KERNEL(k_ldexp, I_LDEXP)
KERNEL(k_dpp_mov, I_DPP_MOV)

// packed fp32: two operations per lane and instruction on register pairs
__global__ void __launch_bounds__(256) k_pk_fma_f32(float *out, int iterations, float fa, float fb)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {(float)threadIdx.x + 1.0f, 2.0f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    f2 a = {fa + (float)(threadIdx.x & 1u) * 1e-9f, fa}, b = {fb + (float)(threadIdx.x & 2u) * 1e-9f, fb};
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\nv_pk_fma_f32 %1, %1, %8, %9\nv_pk_fma_f32 %2, %2, %8, %9\nv_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\nv_pk_fma_f32 %5, %5, %8, %9\nv_pk_fma_f32 %6, %6, %8, %9\nv_pk_fma_f32 %7, %7, %8, %9"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
    const f2 s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void __launch_bounds__(256) k_pk_mul_f32(float *out, int iterations, float fa, float fb)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x0 = {(float)threadIdx.x + 1.0f, 2.0f}, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    f2 a = {fa + (float)(threadIdx.x & 1u) * 1e-9f, fa};
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    }
    const f2 s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + fb;
}

typedef void (*kernel_t)(float *, int, float, float);

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iterations = 20000;
    struct { const char *name; kernel_t k; } list[] = {
        {"v_mul_f32", k_mul_f32}, {"v_add_f32", k_add_f32}, {"v_fma_f32 (VOP3)", k_fma_f32}, {"v_fmac_f32 (VOP2)", k_fmac_f32}, {"v_max_f32", k_max_f32}, {"v_med3_f32", k_med3_f32},
        {"v_ldexp_f32", k_ldexp}, {"v_cmp_lt_f32", k_cmp_f32}, {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_cvt_i32_f32", k_cvt_i32_f32},
        {"v_pk_fma_f32 (2 per lane)", k_pk_fma_f32}, {"v_pk_mul_f32 (2 per lane)", k_pk_mul_f32},
        {"v_mov_b32", k_mov_b32}, {"v_mov_b32 dpp row_shr:1", k_dpp_mov}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_xor_b32", k_xor_b32}, {"v_lshl_add_u32", k_lshl_add},
        {"v_cndmask_b32 (vcc never written)", k_cndmask}, {"v_cmp + 8 v_cndmask_b32 (9 instr)", k_cndmask_fresh},
        {"v_mul_f32, SGPR source", k_mul_f32_s}, {"v_add_f32, SGPR source", k_add_f32_s}, {"v_fma_f32, SGPR source", k_fma_f32_s}, {"v_fmac_f32, SGPR source", k_fmac_f32_s},
        {"v_mul_f32, 32-bit literal", k_mul_f32_lit}, {"v_mul_f32, inline constant", k_mul_f32_inl}, {"v_fmaak_f32 (literal)", k_fmaak},
        {"v_add_u32, SGPR source", k_add_u32_s}, {"v_mov_b32, SGPR source", k_mov_b32_s}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_rcp_f32", k_rcp_f32}, {"v_sqrt_f32", k_sqrt_f32}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves_per_simd = 4; waves_per_simd <= 8; waves_per_simd += 4) {
        const int blocks = cus * waves_per_simd;
        float *out = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&out), (size_t)blocks * 256 * sizeof(float)));
        for (auto &e : list) {
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iterations, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d waves per SIMD  %-28s %8.3f ms  %.2f cycles per instruction and SIMD at 2.4 GHz\n", waves_per_simd, e.name, ms,
                   ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iterations * 64.0));
        }
        for (int fresh = 0; fresh < 2; ++fresh)
            for (int lanes : {0, 1, 8, 16, 17, 32, 48, 64}) {
                auto k = fresh ? k_cndmask_pop<true> : k_cndmask_pop<false>;
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0000001f, 1e-9f, lanes);
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iterations, 1.0000001f, 1e-9f, lanes);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%d waves per SIMD  v_cndmask_b32, mask true in %2d lanes, %s %8.3f ms  %.2f cycles per instruction and SIMD\n", waves_per_simd, lanes,
                       fresh ? "a compare per 8 selects:" : "mask written once:      ", ms, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iterations * (fresh ? 72.0 : 64.0)));
            }
        CHECK(hipFree(out));
    }
    return 0;
}
