// prior families evaluated in-kernel (generic kernels; reference arithmetic order)
// (textually included by iso_hip.hip inside its anonymous namespace: one translation unit, device code only)
#pragma once

// -------------------------------------------------------------------------------------------
// priors
// -------------------------------------------------------------------------------------------
#define LOG_INV_ROOT_2PI (-0.91893853320467267)   // log(1/sqrt(2 pi))
#define INV_ROOT_2PI 0.3989422804014327
#define LN10 2.302585092994046

__device__ __forceinline__ double lognormal_pdf(const DevPrior& P, double x, double mu, double sigma,
                                                double scale)
{
    const double y = x / scale;
    const double ly = log(y) / sigma;
    return INV_ROOT_2PI / (sigma * y) * exp(-0.5 * (ly * ly)) / scale;
}

__device__ __forceinline__ double lognormal_lnpdf(double x, double mu, double sigma, double scale,
                                                  double log_sigma)
{
    const double y = x / scale;
    const double l = log(y);
    const double ly = l / sigma;
    return LOG_INV_ROOT_2PI - (log_sigma + l) - 0.5 * (ly * ly) - mu;
}

__device__ __forceinline__ double feh_shape(const DevPrior& P, double feh)
{
    double disk;
    if (P.c != 0.0) {
        const double u = feh - 0.016, v = feh + 0.15;
        disk = 1.0 / 2.5066282746310007 *
               (0.8 / 0.15 * exp(-0.5 * (u * u) / (0.15 * 0.15)) + 0.2 / 0.22 * exp(-0.5 * (v * v) / (0.22 * 0.22)));
    } else {
        const double u = feh + 0.3;
        disk = INV_ROOT_2PI / 0.3 * exp(-0.5 * (u * u) / (0.3 * 0.3));
    }
    const double h = feh + 1.5;
    const double halo = P.k0 * exp(-0.5 * (h * h) / (0.4 * 0.4));   // k0 = 1/sqrt(2 pi 0.4^2)
    return P.a * halo + (1 - P.a) * disk;
}

// _pdf(x) of a family (no bounds handling)
__device__ double prior_raw(const DevPrior& P, double x)
{
    switch (P.kind) {
    case ISO_PRIOR_FLAT: return P.k0;                                  // 1/(hi-lo)
    case ISO_PRIOR_FLATLOG: return LN10 * exp10(x) / P.k0;             // k0 = 10^hi - 10^lo
    case ISO_PRIOR_POWERLAW: return P.k0 * pow(x, P.a);                // k0 = C
    case ISO_PRIOR_GAUSS: {
        const double z = (x - P.a) / P.b;
        return exp(-(z * z) / 2.0) * INV_ROOT_2PI / P.b / P.k0;        // k0 = exp(lognorm)
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_pdf(P, x, P.a, P.b, P.k0);   // k0 = exp(mu)
    case ISO_PRIOR_CHABRIER:
        if (x < P.d) {
            const double c = (x < 0) ? 0.0 : lognormal_pdf(P, x, P.a, P.b, P.k0);
            return c / P.e;
        } else {
            const double c = (x < P.g || x > P.h) ? 0.0 : P.k2 * pow(x, P.c);   // k2 = C of the power law
            return c / P.f;
        }
    case ISO_PRIOR_FEH: return feh_shape(P, x);
    }
    return d_nan();
}

// prior(x): the reference's __call__ form (pdf with its bounds tests)
__device__ double prior_call(const DevPrior& P, double x)
{
    if (P.kind == ISO_PRIOR_LOGNORMAL) {
        if (x < 0) return 0.0;
        return lognormal_pdf(P, x, P.a, P.b, P.k0);
    }
    if (x < P.lo || x > P.hi) return 0.0;
    const double r = prior_raw(P, x);
    return (P.kind == ISO_PRIOR_FEH) ? r / P.b : r;
}

__device__ double prior_lnpdf(const DevPrior& P, double x)
{
    switch (P.kind) {
    case ISO_PRIOR_FLAT:
    case ISO_PRIOR_FLATLOG: {
        if (x < P.lo || x > P.hi) return -d_inf();
        const double pdf = prior_raw(P, x);
        return pdf != 0 ? log(pdf) : -d_inf();
    }
    case ISO_PRIOR_POWERLAW:
        if (P.bounded && (x < P.lo || x > P.hi)) return -d_inf();
        return P.k1 + P.a * log(x);                                    // k1 = log(C)
    case ISO_PRIOR_GAUSS: {
        if (P.bounded && (x < P.lo || x > P.hi)) return -d_inf();
        const double z = (x - P.a) / P.b;
        return (-(z * z) / 2.0 + LOG_INV_ROOT_2PI) - P.k1 - P.c;       // k1 = log(sigma)
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_lnpdf(x, P.a, P.b, P.k0, P.k1);   // k1 = log(sigma)
    case ISO_PRIOR_CHABRIER:
        if (x < P.d) return lognormal_lnpdf(x, P.a, P.b, P.k0, P.k1) - P.k3;     // k3 = log(e)
        if (x < P.g || x > P.h) return -d_inf();
        return (P.k5 + P.c * log(x)) - P.k4;                     // k5 = log(C), k4 = log(f)
    case ISO_PRIOR_FEH: {
        const double pdf = prior_call(P, x);
        return pdf != 0 ? log(pdf) : -d_inf();
    }
    }
    return d_nan();
}

__device__ __forceinline__ double gauss_term(double val, double g0, double unc2, double model)
{
    const double r = val - model;
    return g0 - 0.5 * r * r / unc2;
}
