// Host check of the arithmetic of fast_log (isochrones_amd/csrc/iso_fast_kernel.h) against logl over 2 x 10^7 arguments:
//   gcc -O2 -ffp-contract=off -o fast_log_check tools/fast_log_check.c -lm && ./fast_log_check   ->  worst 0.796 ulp
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static double rcp_nr(double d) { /* emulate v_rcp_f64 (~1e-8 seed) + 2 NR */
    double r = (double)(float)(1.0 / d);
    double e = fma(-d, r, 1.0); r = fma(r, e, r);
    e = fma(-d, r, 1.0); r = fma(r, e, r);
    return r;
}
static double fast_log(double x) {
    if (!(x > 0.0)) return x == 0.0 ? -INFINITY : NAN;
    if (x == INFINITY) return x;
    int e; double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double f = m - 1.0, d = 2.0 + f;
    double r = rcp_nr(d);
    double s = f * r;
    s = fma(r, fma(-d, s, f), s);
    const double z = s * s, w = z * z;
    const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)e;
    return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
int main() {
    double worst = 0; double wx = 0; srand(1);
    for (long i = 0; i < 20000000; ++i) {
        double u = rand() / (double)RAND_MAX, v = rand() / (double)RAND_MAX;
        double x;
        switch (i & 3) { case 0: x = exp((u - 0.5) * 1400); break; case 1: x = 1.0 + (u - 0.5) * 1e-3 * v; break;
                         case 2: x = u * 3000 + 1e-300; break; default: x = ldexp(1.0 + u, (int)(v * 40) - 20); }
        long double t = logl((long double)x);
        double g = fast_log(x);
        double ulp = fabs((double)t) > 0 ? fabs((double)((long double)g - t)) / (nextafter(fabs((double)t), INFINITY) - fabs((double)t)) : 0;
        if (ulp > worst) { worst = ulp; wx = x; }
    }
    printf("worst %.3f ulp at x=%.17g (fast %.17g libm %.17g)\n", worst, wx, fast_log(wx), log(wx));
    printf("specials: %g %g %g %g %g\n", fast_log(0.0), fast_log(-1.0), fast_log(INFINITY), fast_log(NAN), fast_log(4.9e-324));
    printf("log(denorm) libm %g\n", log(4.9e-324));
    return 0;
}
