// Memory-side fp64 atomic rate on MI355X for the table maker's access shape (ANALYSIS TOOL; hipcc --offload-arch=gfx950 -O3 -o atomic_rate atomic_rate.hip).
// Every wave instruction adds `lanes` active lanes into `sectors` distinct 64-byte sectors of a table of `mb` megabytes (lanes that share a
// sector take neighbouring bins of it); sectors are drawn at random per instruction.  Prints wave instructions, lane adds and sector requests
// per second.  usage: atomic_rate [table MB] [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }

// `f32`: binary32 adds into the same sectors' first words instead
template <bool F32>
__global__ void __launch_bounds__(256) adds(double *table, uint64_t n_sectors, int iterations, int lanes, int sectors, int repeat = 1)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool active = (int)lane < lanes;
    const uint32_t group = lane % (uint32_t)sectors;            // which of the instruction's sectors
    const uint32_t bin = (lane / (uint32_t)sectors) & 7u;       // which bin of it
    for (int it = 0; it < iterations; ++it) {
        const uint32_t h = mix(mix(wave * 0x9e3779b9u + (uint32_t)(it / repeat)) + group * 0x85ebca6bu);       // (the same sectors `repeat` instructions in a row)
        const uint64_t sector = ((uint64_t)h * n_sectors) >> 32;
        if (active) {
            if (F32) unsafeAtomicAdd(reinterpret_cast<float *>(table + sector * 8u + bin), 1.0f);
            else unsafeAtomicAdd(table + sector * 8u + bin, 1.0);
        }
    }
}

// the same adds behind `filler` dependent fused multiply-adds per instruction (the table maker's shape: ~3 300 cycles of arithmetic per 64 samples
// and wave, three waves per SIMD): what do the atomics cost a kernel that is far from their rate?
template <bool SETREG>
__global__ void __launch_bounds__(256) adds_beside_arithmetic(double *table, uint64_t n_sectors, int iterations, int lanes, int sectors, int filler, int with_adds, float *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const bool active = (int)lane < lanes;
    const uint32_t group = lane % (uint32_t)sectors;
    const uint32_t bin = (lane / (uint32_t)sectors) & 7u;
    float acc = (float)lane * 1.0e-3f;
    for (int it = 0; it < iterations; ++it) {
        if (SETREG) {
            // the propagation kernels' random numbers convert with round-toward-zero through the MODE register (prop_device.hip.h: rng_co):
            // does s_setreg wait for the wave's outstanding vector memory operations?
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_f32_u32_e32 %0, %0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(acc));
        }
        for (int k = 0; k < filler; ++k) acc = __builtin_fmaf(acc, 0.999f, 1.0e-3f);
        const uint32_t h = mix(mix(wave * 0x9e3779b9u + (uint32_t)it) + group * 0x85ebca6bu);
        const uint64_t sector = ((uint64_t)h * n_sectors) >> 32;
        if (active && with_adds) unsafeAtomicAdd(table + sector * 8u + bin, (double)acc);
    }
    if (acc == -1.0f) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 670;
    const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 3;
    const uint64_t n_sectors = mb * 1000000ull / 64ull;
    double *table = nullptr;
    CHECK(hipMalloc(&table, n_sectors * 64));
    CHECK(hipMemset(table, 0, n_sectors * 64));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int grid = cus * waves_per_simd;           // 256 threads = 4 waves per workgroup: one per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("table %zu MB, %d CUs, %d waves per SIMD\n", mb, cus, waves_per_simd);
    printf("%6s %8s %6s %14s %14s %14s\n", "type", "lanes", "sect.", "instr/s", "lane adds/s", "requests/s");
    const int shapes[][2] = {{64, 64}, {64, 32}, {64, 16}, {64, 8}, {32, 32}, {24, 16}, {24, 24}, {16, 16}, {8, 8}, {64, 1}};
    for (int f32 = 0; f32 < 2; ++f32)
        for (auto &s : shapes) {
            const int lanes = s[0], sectors = s[1];
            const int iterations = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipEventRecord(e0));
                if (f32) hipLaunchKernelGGL(adds<true>, dim3(grid), dim3(256), 0, 0, table, n_sectors, iterations, lanes, sectors);
                else hipLaunchKernelGGL(adds<false>, dim3(grid), dim3(256), 0, 0, table, n_sectors, iterations, lanes, sectors);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 1) {
                    const double instr = (double)grid * 4 * iterations / (ms * 1e-3);
                    printf("%6s %8d %6d %14.4g %14.4g %14.4g\n", f32 ? "f32" : "f64", lanes, sectors, instr, instr * lanes, instr * sectors);
                }
            }
        }
    // the same sectors again: a photon's next segment often lands in the sector its last one ended in
    printf("\na wave adds into the same 16 sectors R instructions in a row (24 lanes, f64):\n%6s %14s %14s\n", "R", "instr/s", "requests/s");
    for (int repeat : {1, 2, 4, 8, 64}) {
        const int iterations = 20000;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(adds<false>, dim3(grid), dim3(256), 0, 0, table, n_sectors, iterations, 24, 16, repeat);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double instr = (double)grid * 4 * iterations / (ms * 1e-3);
        printf("%6d %14.4g %14.4g\n", repeat, instr, instr * 16);
    }
    // arithmetic beside the adds
    float *sink = nullptr;
    CHECK(hipMalloc(&sink, 64));
    printf("\nadds beside arithmetic (24 lanes into 16 sectors per instruction, f64), %d waves per SIMD:\n", waves_per_simd);
    printf("%8s %14s %14s %10s %14s\n", "filler", "ms without", "ms with adds", "ratio", "requests/s");
    for (int setreg = 0; setreg < 2; ++setreg) {
        if (setreg) printf("... with an s_setreg MODE pair (round toward zero and back) behind every add instruction:\n");
        for (int filler : {0, 100, 200, 400, 800, 1600}) {
            const int iterations = filler >= 800 ? 3000 : 10000;
            float ms2[2] = {0, 0};
            for (int with_adds = 0; with_adds < 2; ++with_adds)
                for (int rep = 0; rep < 2; ++rep) {
                    CHECK(hipEventRecord(e0));
                    if (setreg) hipLaunchKernelGGL(adds_beside_arithmetic<true>, dim3(grid), dim3(256), 0, 0, table, n_sectors, iterations, 24, 16, filler, with_adds, sink);
                    else hipLaunchKernelGGL(adds_beside_arithmetic<false>, dim3(grid), dim3(256), 0, 0, table, n_sectors, iterations, 24, 16, filler, with_adds, sink);
                    CHECK(hipEventRecord(e1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms2[with_adds], e0, e1));
                }
            printf("%8d %14.2f %14.2f %10.3f %14.4g\n", filler, ms2[0], ms2[1], ms2[1] / ms2[0], (double)grid * 4 * iterations * 16 / (ms2[1] * 1e-3));
        }
    }
    return 0;
}
