// Host check of the arithmetic of exp_nonpos (isochrones_amd/csrc/iso_fast_kernel.h) against expl over 10^7 arguments <= 0:
//   gcc -O2 -ffp-contract=off -o /tmp/fast_exp_check tools/fast_exp_check.c -lm && /tmp/fast_exp_check
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static double exp_nonpos(double x) {
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.08767569878681e-09);
    p = fma(p, r, 2.505210838544172e-08);
    p = fma(p, r, 2.755731922398589e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.48015873015873e-05);
    p = fma(p, r, 1.984126984126984e-04);
    p = fma(p, r, 1.388888888888889e-03);
    p = fma(p, r, 8.333333333333333e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double v = ldexp(p, (int)k);
    return (x >= -745.2) ? v : ((x != x) ? x : 0.0);
}
int main() {
    double worst = 0, wx = 0; srand(2);
    for (long i = 0; i < 10000000; ++i) {
        double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
        double x;
        switch (i & 3) { case 0: x = -u * 745.0; break; case 1: x = -u * v * 1e-3; break; case 2: x = -u * 40.0; break; default: x = -ldexp(1.0 + u, (int)(v * 30) - 20); }
        if (x < -708.0) continue;                       /* (below: subnormal results, compared separately) */
        long double t = expl((long double)x);
        double g = exp_nonpos(x);
        double ulp = fabs((double)((long double)g - t)) / (nextafter((double)t, INFINITY) - (double)t);
        if (ulp > worst) { worst = ulp; wx = x; }
    }
    printf("worst %.3f ulp at x=%.17g (fast %.17g libm %.17g)\n", worst, wx, exp_nonpos(wx), exp(wx));
    double ws = 0;
    for (int i = 0; i < 100000; ++i) { double x = -708.0 - 37.0 * rand() / (double)RAND_MAX; double g = exp_nonpos(x), t = exp(x); double d = fabs(g - t) / 4.9406564584124654e-324; if (d > ws) ws = d; }
    printf("subnormal tail [-745, -708]: worst difference %.1f units of the smallest subnormal\n", ws);
    printf("specials: exp(0)=%g exp(-0)=%g exp(-746)=%g exp(-inf)=%g exp(nan)=%g exp(-745.1)=%g libm %g\n", exp_nonpos(0.0), exp_nonpos(-0.0), exp_nonpos(-746.0), exp_nonpos(-INFINITY), exp_nonpos(NAN), exp_nonpos(-745.1), exp(-745.1));
    return 0;
}
