/*
 * count_ops.hpp -- TEST / MEASUREMENT INFRASTRUCTURE, not product code.
 *
 * The counting build of the oracle (`make -C oracle liboracle_count.so`): clsim_oracle.c is compiled as C++ with every
 * `float` of its own text replaced by `cfloat` (a sed in the Makefile; oracle_math.h is included BEFORE the replacement and
 * keeps plain floats).  cfloat is one float with the same layout; its operators do the float operation and count it in
 * the thread's counters, attributed to the region of the kernel that is active (REGION(...) markers in clsim_oracle.c,
 * no-ops in the normal build).  What is counted is therefore the arithmetic the REFERENCE's expressions ask for, operation
 * by operation as written (no common subexpressions removed, none added), per photon and per loop trip:
 *   add (+, -), mul, div, compares, negations, conversions int<->float, calls of the math library BY NAME (their inner
 *   operations are not counted: the name is the unit), RNG draws, sqrt / rsqrt.
 * Results equal the normal build's bit for bit (the operators ARE the float operations, -ffp-contract=off): tests/test_count_ops.py.
 *
 * SURVEY.md 8(d) / BASELINE.md section 2 promise "a counted flop/transcendental figure per photon from the instrumented
 * CPU restatement" beside the HBM fraction; the metric's definition: resources/scripts/benchmark.py:326-340.
 */
#pragma once
#include <stdint.h>
#include <type_traits>
#define _Static_assert static_assert

enum oc_op { OC_ADD = 0, OC_MUL, OC_DIV, OC_CMP, OC_NEG, OC_CVT, OC_SQRT, OC_RSQRT, OC_LOG, OC_EXP, OC_POWR, OC_POWR_UNIT, OC_SIN, OC_COS,
             OC_SINCOS, OC_ACOS, OC_ATAN2, OC_RNG, OC_FLOOR_TRUNC, OC_FABS, OC_OTHER_MATH, OC_NUM_OPS };
enum oc_region { OCR_OTHER = 0, OCR_CREATE, OCR_WAVELENGTH, OCR_MEDIUM_PER_PHOTON, OCR_TILT, OCR_LAYER_LENGTHS, OCR_WALK, OCR_ANISO, OCR_SCATTER_ANGLE,
                 OCR_ROTATE, OCR_TRANSFORM, OCR_SEARCH_CELLS, OCR_SEARCH_STRING, OCR_SEARCH_DOM, OCR_HIT_RECORD, OCR_ADVANCE, OCR_RNG_INTERNAL, OCR_PER_STEP, OCR_NUM_REGIONS };
enum oc_event { OCE_PHOTONS = 0, OCE_TRIPS, OCE_SCATTERS, OCE_LAYER_CROSSINGS, OCE_LIU, OCE_HG, OCE_SEARCH_CALLS, OCE_CELLS, OCE_STRINGS, OCE_DOM_TESTS, OCE_HITS,
                OCE_LAYER_LENGTH_EVALS, OCE_STEPS, OCE_CROSSING_TRIPS, OCE_NUM_EVENTS };

struct oc_counters {
    uint64_t ops[OCR_NUM_REGIONS][OC_NUM_OPS];
    uint64_t events[OCE_NUM_EVENTS];
};
extern thread_local oc_counters oc_tl;
extern thread_local int oc_region_now;

static inline void oc_count(int op) { ++oc_tl.ops[oc_region_now][op]; }
static inline void oc_event_add(int e, uint64_t n) { oc_tl.events[e] += n; }

struct oc_region_guard {
    int saved;
    explicit oc_region_guard(int r) : saved(oc_region_now) { oc_region_now = r; }
    ~oc_region_guard() { oc_region_now = saved; }
};
#define OC_CAT2(a, b) a##b
#define OC_CAT(a, b) OC_CAT2(a, b)
#define REGION(r) oc_region_guard OC_CAT(oc_rg_, __LINE__)(r)
#define EVENT(e, n) oc_event_add((e), (uint64_t)(n))
#define COUNT_OP(op) oc_count(op)
/* scatterDirectionByAngle is also photon creation's last act: its operations stay with the creation then */
#define REGION_UNLESS_CREATING(r) oc_region_guard OC_CAT(oc_rg_, __LINE__)((oc_region_now == OCR_CREATE) ? OCR_CREATE : (r))
void oc_fold_thread_counters(void);
#define COUNT_FOLD() oc_fold_thread_counters()

struct cfloat {
    float v;
    cfloat() = default;
    cfloat(float x) : v(x) {}
    cfloat(double x) : v((float)x) {}
    cfloat(int x) : v((float)x) { oc_count(OC_CVT); }
    cfloat(unsigned x) : v((float)x) { oc_count(OC_CVT); }
    cfloat(long x) : v((float)x) { oc_count(OC_CVT); }
    cfloat(unsigned long x) : v((float)x) { oc_count(OC_CVT); }
    cfloat(unsigned short x) : v((float)x) { oc_count(OC_CVT); }
    cfloat(short x) : v((float)x) { oc_count(OC_CVT); }
    operator float() const { return v; }
    explicit operator int() const { oc_count(OC_CVT); return (int)v; }
    explicit operator unsigned() const { oc_count(OC_CVT); return (unsigned)v; }
    explicit operator double() const { return (double)v; }
    cfloat &operator+=(cfloat b) { oc_count(OC_ADD); v = v + b.v; return *this; }
    cfloat &operator-=(cfloat b) { oc_count(OC_ADD); v = v - b.v; return *this; }
    cfloat &operator*=(cfloat b) { oc_count(OC_MUL); v = v * b.v; return *this; }
    cfloat &operator/=(cfloat b) { oc_count(OC_DIV); v = v / b.v; return *this; }
};
static_assert(sizeof(cfloat) == 4 && std::is_trivially_copyable<cfloat>::value && std::is_standard_layout<cfloat>::value, "cfloat is one float");

template <class T> using oc_arith = typename std::enable_if<std::is_arithmetic<T>::value, int>::type;
static inline float oc_f(cfloat a) { return a.v; }
template <class T, oc_arith<T> = 0> static inline float oc_f(T a) { return (float)a; }

#define OC_BINARY(OP, WHAT, RET, EXPR)                                                                                         \
    static inline RET operator OP(cfloat a, cfloat b) { oc_count(WHAT); const float x = a.v, y = b.v; return EXPR; }           \
    template <class T, oc_arith<T> = 0> static inline RET operator OP(cfloat a, T b) { oc_count(WHAT); const float x = a.v, y = oc_f(b); return EXPR; } \
    template <class T, oc_arith<T> = 0> static inline RET operator OP(T a, cfloat b) { oc_count(WHAT); const float x = oc_f(a), y = b.v; return EXPR; }
OC_BINARY(+, OC_ADD, cfloat, cfloat(x + y))
OC_BINARY(-, OC_ADD, cfloat, cfloat(x - y))
OC_BINARY(*, OC_MUL, cfloat, cfloat(x * y))
OC_BINARY(/, OC_DIV, cfloat, cfloat(x / y))
OC_BINARY(<, OC_CMP, bool, x < y)
OC_BINARY(>, OC_CMP, bool, x > y)
OC_BINARY(<=, OC_CMP, bool, x <= y)
OC_BINARY(>=, OC_CMP, bool, x >= y)
OC_BINARY(==, OC_CMP, bool, x == y)
OC_BINARY(!=, OC_CMP, bool, x != y)
#undef OC_BINARY
static inline cfloat operator-(cfloat a) { oc_count(OC_NEG); return cfloat(-a.v); }
static inline cfloat operator+(cfloat a) { return a; }
static inline bool operator!(cfloat a) { oc_count(OC_CMP); return a.v == 0.0f; }

/* the math library by name: the call is the unit (oracle_math.h stays plain float; its inner operations are not counted) */
#define OC_CALL(what, expr) (oc_count(what), cfloat((float)(expr)))
