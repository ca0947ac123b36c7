/* TEST INFRASTRUCTURE: measures the ulp error of oracle_math.h against binary64 libm. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "oracle_math.h"

static double ulp_err(float got, double want)
{
    float w = (float)want;
    if (got == w) { /* still measure fractional error */ }
    int e; frexp(want, &e);
    double ulp = ldexp(1.0, e - 24);
    return fabs((double)got - want) / ulp;
}
#define SCAN(name, lo, hi, n, expr_got, expr_want) do { \
    double worst = 0, wx = 0; \
    for (long i = 0; i <= (n); ++i) { \
        float x = (float)((lo) + ((hi) - (lo)) * (double)i / (double)(n)); \
        double e = ulp_err(expr_got, expr_want); \
        if (e > worst) { worst = e; wx = x; } \
    } \
    printf("%-28s max %.3f ulp at x=%.9g\n", name, worst, wx); } while (0)

/* every float of a bit-pattern range (positive floats order like their patterns), against the x87 long double libm */
static double ulp_err_l(float got, long double want)
{
    int e; frexpl(want, &e);
    return (double)(fabsl((long double)got - want) / ldexpl(1.0L, e - 24));
}
#define SCAN_ALL(name, lo, hi, expr_got, expr_want) do { \
    double worst = 0; float wx = 0; \
    _Pragma("omp parallel") { double w = 0; float wi = 0; \
        _Pragma("omp for schedule(static)") \
        for (long long b = om_f2u(lo); b <= (long long)om_f2u(hi); ++b) { \
            const float x = om_u2f((uint32_t)b); const double e = ulp_err_l(expr_got, expr_want); if (e > w) { w = e; wi = x; } } \
        _Pragma("omp critical") if (w > worst) { worst = w; wx = wi; } } \
    printf("%-28s max %.3f ulp at x=%a  (EVERY float of the range)\n", name, worst, wx); } while (0)

static float sin_of(float x) { float s, c; om_sincos_2pi(x, &s, &c); return s; }
static float cos_of(float x) { float s, c; om_sincos_2pi(x, &s, &c); return c; }

int main(int argc, char **argv)
{
    const long N = 20000000;
    if (argc > 1 && argv[1][0] == 'e') {        /* `mathcheck exhaustive`: the round-5 table forms over their whole domains (a few core-minutes) */
        SCAN_ALL("log [2^-40, 4)", 0x1p-40f, 0x1.fffffep+1f, om_log(x), logl((long double)x));
        SCAN_ALL("sin [2^-60, RN(2pi)]", 0x1p-60f, OM_SINCOS_2PI_MAX, sin_of(x), sinl((long double)x));
        SCAN_ALL("cos [2^-60, RN(2pi)]", 0x1p-60f, OM_SINCOS_2PI_MAX, cos_of(x), cosl((long double)x));
        return 0;
    }
    SCAN("log (0,1]", 5.9604645e-8, 1.0, N, om_log(x), log((double)x));
    SCAN("log [1,1000]", 1.0, 1000.0, N, om_log(x), log((double)x));
    SCAN("exp [-30,0]", -30.0, 0.0, N, om_exp(x), exp((double)x));
    SCAN("exp [-1,1]", -1.0, 1.0, N, om_exp(x), exp((double)x));
    SCAN("powr(x,-1.0841) [265,675]", 265.0, 675.0, N, om_powr(x, -1.084106802940f), pow((double)x, (double)-1.084106802940f));
    SCAN("powr(x,-0.8986) [.6,1.7]", 0.6, 1.7, N, om_powr(x, -0.898608505726f), pow((double)x, (double)-0.898608505726f));
    SCAN("powr(x,0.0526) (0,1]", 1e-9, 1.0, N, om_powr(x, 0.0526315793f), pow((double)x, (double)0.0526315793f));
    SCAN("powr_unit(x,0.0526) (0,1]", 2.3283064e-10, 1.0, N, om_powr_unit(x, 0.0526315793f), pow((double)x, (double)0.0526315793f));
    SCAN("powr_unit(x,0.0526) (0,1e-4]", 2.3283064e-10, 1e-4, N, om_powr_unit(x, 0.0526315793f), pow((double)x, (double)0.0526315793f));
    SCAN("powr_unit(x,0.09) (0,1]", 2.3283064e-10, 1.0, N, om_powr_unit(x, 0.09f), pow((double)x, (double)0.09f));
    SCAN("powr_unit(x,0.09) (0,1e-6]", 2.3283064e-10, 1e-6, N, om_powr_unit(x, 0.09f), pow((double)x, (double)0.09f));
    SCAN("sin [0,2pi]", 0.0, 6.2831855, N, om_sin(x), sin((double)x));
    SCAN("cos [0,2pi]", 0.0, 6.2831855, N, om_cos(x), cos((double)x));
    SCAN("sin [-100,100]", -100.0, 100.0, N, om_sin(x), sin((double)x));
    SCAN("acos [-1,1]", -1.0, 1.0, N, om_acos(x), acos((double)x));
    SCAN("atan2(x,0.3) [-5,5]", -5.0, 5.0, N, om_atan2(x, 0.3f), atan2((double)x, 0.3));
    SCAN("atan2(0.3,x) [-5,5]", -5.0, 5.0, N, om_atan2(0.3f, x), atan2(0.3, (double)x));
    return 0;
}
