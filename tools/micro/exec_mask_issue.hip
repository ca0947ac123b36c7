// Does a wave64 vector instruction cost less when part of the wave is masked off?  (ANALYSIS TOOL; hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o exec_mask_issue exec_mask_issue.hip: no packed f32, as in the product)
// Every wave runs the same chain of dependent-free fused multiply-adds (eight independent accumulators per lane, so the chain is issue bound, not
// latency bound) under an exec mask chosen on the command line: all 64 lanes, the low 32, the low 16, one lane in four (16 lanes spread over all four
// quarter-waves), one lane.  If the hardware skipped quarter- or half-waves whose lanes are all off, the packed masks would run faster than the spread one;
// what it prints: vector instructions per second and SIMD for each mask, at 4 and 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) chain(float *out, unsigned long long mask, int iterations, float a, float b)
{
    const uint32_t lane = threadIdx.x & 63u;
    float x0 = (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iterations; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// the same with four-byte instructions (v_fmac_f32, VOP2: x += a * b; the chain above is v_fma_f32, VOP3, eight bytes each): is the rate above
// the vector unit's or the instruction fetch's?
__global__ void __launch_bounds__(256) chain_vop2(float *out, unsigned long long mask, int iterations, float a, float b)
{
    const uint32_t lane = threadIdx.x & 63u;
    float x0 = (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    float bv = b + (float)lane * 1e-12f;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iterations; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x0 = __builtin_fmaf(a, bv, x0); x1 = __builtin_fmaf(a, bv, x1); x2 = __builtin_fmaf(a, bv, x2); x3 = __builtin_fmaf(a, bv, x3);
                x4 = __builtin_fmaf(a, bv, x4); x5 = __builtin_fmaf(a, bv, x5); x6 = __builtin_fmaf(a, bv, x6); x7 = __builtin_fmaf(a, bv, x7);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));      // (keep the eight adds apart)
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// a mix closer to real code: multiplies, adds, integer adds, xors and selects on eight independent values (no fused multiply-add)
__global__ void __launch_bounds__(256) chain_mix(float *out, unsigned long long mask, int iterations, float a, float b)
{
    const uint32_t lane = threadIdx.x & 63u;
    float x0 = (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f;
    uint32_t y0 = threadIdx.x, y1 = y0 * 3u, y2 = y0 * 5u, y3 = y0 * 7u;
    const uint32_t m = __builtin_bit_cast(uint32_t, b) | 1u;
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < iterations; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b;
                y0 = y0 + m; y1 = y1 ^ m; y2 = y2 + y0; y3 = (y3 > y1) ? y3 - m : y3 + m;
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + (float)((y0 ^ y1) + (y2 ^ y3));
}

// a short masked region inside full-lane code, as in a real loop: per iteration 64 vector instructions with every lane, then `region` fused
// multiply-adds under the mask.  Does the region cost what its instructions cost, whatever the mask?
template <int REGION>
__global__ void __launch_bounds__(256) region_in_loop(float *out, unsigned long long mask, int iterations, float a, float b)
{
    const uint32_t lane = threadIdx.x & 63u;
    float x0 = (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    const bool in = (mask >> lane) & 1ull;
    for (int it = 0; it < iterations; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        if (in) {
#pragma unroll
            for (int k = 0; k < REGION / 8; ++k) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

// the same chain, 64 iterations (4 096 instructions), between two reads of s_memtime and of s_memrealtime (the constant 100 MHz counter):
// ticks per instruction for this wave by both, i.e. whether s_memtime follows the shader clock and what that clock was
__global__ void __launch_bounds__(256) chain_timed(float *out, unsigned long long mask, float a, float b, unsigned long long *stamps)
{
    const uint32_t lane = threadIdx.x & 63u;
    float x0 = (float)threadIdx.x, x1 = x0 + 1.0f, x2 = x0 + 2.0f, x3 = x0 + 3.0f, x4 = x0 + 4.0f, x5 = x0 + 5.0f, x6 = x0 + 6.0f, x7 = x0 + 7.0f;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t c0 = __builtin_amdgcn_s_memtime();
    if ((mask >> lane) & 1ull) {
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        }
    }
    asm volatile("s_nop 0" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
    const uint64_t c1 = __builtin_amdgcn_s_memtime();
    const uint64_t t1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = c1 - c0; stamps[1] = t1 - t0; }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iterations = 20000;               // x 64 fma per iteration and lane
    struct { const char *name; unsigned long long mask; } masks[] = {
        {"all 64 lanes", ~0ull}, {"low 32 lanes", 0xffffffffull}, {"low 16 lanes", 0xffffull}, {"lanes 16-31", 0xffff0000ull},
        {"one lane in four (16, spread)", 0x1111111111111111ull}, {"one lane in two (32, spread)", 0x5555555555555555ull}, {"low 24 lanes", 0xffffffull},
        {"low 17 lanes", 0x1ffffull}, {"low 8 lanes", 0xffull}, {"low 4 lanes", 0xfull}, {"one lane", 1ull}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves_per_simd = 4; waves_per_simd <= 8; waves_per_simd += 4) {
        const int blocks = cus * waves_per_simd;           // 256-thread blocks: 4 waves, one per SIMD
        float *out = nullptr;
        CHECK(hipMalloc(reinterpret_cast<void **>(&out), (size_t)blocks * 256 * sizeof(float)));
        for (auto &m : masks) {
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, m.mask, 100, 1.0000001f, 1e-9f);       // warm-up
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, m.mask, iterations, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double insts_per_simd = (double)waves_per_simd * iterations * 64.0;       // per SIMD: its waves' vector instructions
            printf("%d waves per SIMD  %-32s %8.3f ms  %.3f vector instructions per cycle and SIMD at 2.4 GHz\n", waves_per_simd, m.name, ms,
                   insts_per_simd / (ms * 1e-3) / 2.4e9);
        }
        for (auto &m : masks) {
            hipLaunchKernelGGL(chain_vop2, dim3(blocks), dim3(256), 0, 0, out, m.mask, 100, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(chain_vop2, dim3(blocks), dim3(256), 0, 0, out, m.mask, iterations, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%d waves per SIMD  %-32s %8.3f ms  %.3f four-byte vector instructions (v_fmac_f32) per cycle and SIMD at 2.4 GHz\n", waves_per_simd, m.name, ms,
                   (double)waves_per_simd * iterations * 64.0 / (ms * 1e-3) / 2.4e9);
        }
        for (auto &m : masks) {
            hipLaunchKernelGGL(chain_mix, dim3(blocks), dim3(256), 0, 0, out, m.mask, 100, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(chain_mix, dim3(blocks), dim3(256), 0, 0, out, m.mask, iterations, 1.0000001f, 1e-9f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            // (the select is a compare and two adds and a v_cndmask: 11 vector instructions per 8 statements)
            printf("%d waves per SIMD  %-32s %8.3f ms  mix of mul / add / integer / select: %.2f ns per iteration of 88 vector instructions and SIMD\n", waves_per_simd, m.name, ms,
                   ms * 1e6 / ((double)waves_per_simd * iterations));
        }
        for (int region : {0, 16, 32, 64, 128, 256, 512}) {
            for (auto &m : masks) {
                auto kernel = region == 0 ? region_in_loop<0> : region == 16 ? region_in_loop<16> : region == 32 ? region_in_loop<32> : region == 64 ? region_in_loop<64> :
                              region == 128 ? region_in_loop<128> : region == 256 ? region_in_loop<256> : region_in_loop<512>;
                if (region > 32 && m.mask != ~0ull && m.mask != 0xffull && m.mask != 0xffffffffull) continue;          // (longer regions: all, 32 and 8 lanes only)
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, m.mask, 100, 1.0000001f, 1e-9f);
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, m.mask, region > 64 ? iterations / 4 : iterations, 1.0000001f, 1e-9f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms = 0.f;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("%d waves per SIMD  %-32s %8.3f ms  64 full-lane instructions + a region of %3d under the mask: %.1f cycles per iteration and wave at 2.4 GHz\n", waves_per_simd, m.name, ms,
                       region, ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * (region > 64 ? iterations / 4 : iterations)));
                if (region == 0) break;          // (no region: the mask does not matter)
            }
        }
        unsigned long long *stamps = nullptr, host[2];
        CHECK(hipMalloc(reinterpret_cast<void **>(&stamps), 16));
        for (auto &m : masks) {
            // (after a long launch of the same mask, so that the clocks are where that load puts them)
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(256), 0, 0, out, m.mask, iterations, 1.0000001f, 1e-9f);
            hipLaunchKernelGGL(chain_timed, dim3(blocks), dim3(256), 0, 0, out, m.mask, 1.0000001f, 1e-9f, stamps);
            CHECK(hipMemcpy(host, stamps, 16, hipMemcpyDeviceToHost));
            printf("%d waves per SIMD  %-32s wave 0, 4096 instructions: s_memtime %llu ticks (%.3f per instruction), s_memrealtime %llu ticks of 10 ns (%.2f ns per instruction)\n",
                   waves_per_simd, m.name, host[0], host[0] / 4096.0, host[1], host[1] * 10.0 / 4096.0);
        }
        CHECK(hipFree(stamps));
        CHECK(hipFree(out));
    }
    return 0;
}
