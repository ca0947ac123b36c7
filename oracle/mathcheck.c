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

int main(void)
{
    const long N = 20000000;
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
