// Measurement builds of the kernels -- never the shipped library.  The kernel sources carry their instrumentation through the
// macros below and hold no `#ifdef` of their own for it; in the default build every one of them expands to nothing (tools/code_hash.py:
// the machine code of the default build is the same with and without the instrumented lines).
//
//   make EXTRA=-DCLSIMHIP_CENSUS          lane-state census, per-wave clocks, visits and active lanes of the divergent regions
//                                         (tools/exp_census.py, tools/exp_pool_census.py; host side: converter.cpp allocates KParams::census)
//   make EXTRA=-DCLSIMHIP_TAB_TIMERS      table maker: shader-clock time per phase of a loop trip (tools/exp_tab_timers.py)
//   make EXTRA=-DCLSIMHIP_DEBUG_COUNTERS  classic kernel: polls of unpublished slices, sleeps (tools/exp_slices.py)
//
// Experiments that lost their A/B are not in the sources at all: tools/experiments/*.patch (applied by tools/build_variant.sh).
#pragma once

#ifdef CLSIMHIP_CENSUS
#define CENSUS(...) __VA_ARGS__
// one visit of a divergent region and the lanes that are active in it, per wave, in the census buffer behind the per-wave records
// (word 32768 + 32 x wave + 2 x region)
#define CENSUS_REGION(P, region)                                                                                                 \
    do {                                                                                                                          \
        const uint64_t census_m = __builtin_amdgcn_ballot_w64(true);                                                              \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(census_m)) {                                                         \
            unsigned long long *census_d = (P)->census + 32768u + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32u + 2u * (uint32_t)(region); \
            census_d[0] += 1ull;                                                                                                  \
            census_d[1] += (unsigned long long)__builtin_popcountll(census_m);                                                    \
        }                                                                                                                         \
    } while (0)
#else
#define CENSUS(...)
#define CENSUS_REGION(P, region) ((void)0)
#endif

#ifdef CLSIMHIP_TAB_TIMERS
#define TAB_TIMED(...) __VA_ARGS__
#define TAB_STAMP(k) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); t_acc[k] += now_ - t_last; t_last = now_; }
#else
#define TAB_TIMED(...)
#define TAB_STAMP(k)
#endif

#ifdef CLSIMHIP_DEBUG_COUNTERS
#define DEBUG_COUNTED(...) __VA_ARGS__
#else
#define DEBUG_COUNTED(...)
#endif
