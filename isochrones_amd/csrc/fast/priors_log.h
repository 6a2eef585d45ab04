// prior families in log space with host-precomputed constants
// (part of iso_fast_kernel.h: included inside namespace iso::fastk)
#pragma once

// ---- priors in log space ------------------------------------------------------------------
__device__ __forceinline__ double lognormal_ln(const DevPrior& P, double lx)
{
    // lx = log(x); y = x/scale -> log(y) = lx - mu
    const double l = lx - P.a;
    const double ly = l * P.r1;
    return kLogInvRoot2Pi - (P.k1 + l) - 0.5 * (ly * ly) - P.a;
}

// LOCAL: -1 = the record says which disk (P.c), 1 = the two-Gaussian local disk (the reference's default)
template <int LOCAL = -1>
__device__ __forceinline__ double feh_pdf(const DevPrior& P, double x)
{
    double disk;
    if (LOCAL > 0 || (LOCAL < 0 && P.c != 0.0)) {
        constexpr double c1 = 0.8 / 0.15 / 2.5066282746310007, c2 = 0.2 / 0.22 / 2.5066282746310007;
        constexpr double e1 = -0.5 / (0.15 * 0.15), e2 = -0.5 / (0.22 * 0.22);
        const double u = x - 0.016, v = x + 0.15;
        disk = c1 * exp(e1 * (u * u)) + c2 * exp(e2 * (v * v));
    } else {
        constexpr double c0 = kInvRoot2Pi / 0.3, e0 = -0.5 / (0.3 * 0.3);
        const double u = x + 0.3;
        disk = c0 * exp(e0 * (u * u));
    }
    constexpr double eh = -0.5 / (0.4 * 0.4);
    const double h = x + 1.5;
    const double halo = P.k0 * exp(eh * (h * h));
    return (P.a * halo + (1 - P.a) * disk) * P.r0;   // r0 = 1/norm
}

// log of the reference's lnpdf(x).  HAS_LX: lx = log(x) supplied by the caller.
template <bool HAS_LX>
__device__ __forceinline__ double ln_pdf_rt(const DevPrior& P, double x, double lx)
{
    const bool outside = (x < P.lo) || (x > P.hi);
    switch (P.kind) {
    case ISO_PRIOR_FLAT: return outside ? -f_inf() : P.k1;
    case ISO_PRIOR_FLATLOG: return outside ? -f_inf() : fma(x, kLn10, P.k1);
    case ISO_PRIOR_POWERLAW: {
        if (P.bounded && outside) return -f_inf();
        const double l = HAS_LX ? lx : fast_log(x);
        return fma(P.a, l, P.k1);
    }
    case ISO_PRIOR_GAUSS: {
        if (P.bounded && outside) return -f_inf();
        const double z = (x - P.a) * P.r0;
        return (-0.5 * (z * z) + kLogInvRoot2Pi) - P.k1 - P.c;
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_ln(P, HAS_LX ? lx : fast_log(x));
    case ISO_PRIOR_CHABRIER: {
        const double l = HAS_LX ? lx : fast_log(x);
        if (x < P.d) return lognormal_ln(P, l) - P.k3;
        if (x < P.g || x > P.h) return -f_inf();
        return fma(P.c, l, P.k5) - P.k4;
    }
    case ISO_PRIOR_FEH: {
        if (outside) return -f_inf();
        const double pdf = feh_pdf<>(P, x);
        return pdf != 0 ? fast_log(pdf) : -f_inf();
    }
    }
    return f_nan();
}

// log of the reference's prior(x) (the __call__ / pdf form): -inf where the pdf is exactly 0
__device__ __forceinline__ double ln_call_rt(const DevPrior& P, double x)
{
    const bool outside = (x < P.lo) || (x > P.hi);
    switch (P.kind) {
    case ISO_PRIOR_FLAT: return outside ? -f_inf() : P.k1;
    case ISO_PRIOR_FLATLOG: return outside ? -f_inf() : fma(x, kLn10, P.k1);
    case ISO_PRIOR_POWERLAW: return outside ? -f_inf() : fma(P.a, fast_log(x), P.k1);
    case ISO_PRIOR_GAUSS: {
        if (outside) return -f_inf();
        const double z = (x - P.a) * P.r0;
        return (-0.5 * (z * z) + kLogInvRoot2Pi) - P.k1 - P.c;
    }
    case ISO_PRIOR_LOGNORMAL: return (x < 0) ? -f_inf() : lognormal_ln(P, fast_log(x));
    case ISO_PRIOR_CHABRIER: {
        if (outside) return -f_inf();
        if (x < P.d) return (x < 0) ? -f_inf() : lognormal_ln(P, fast_log(x)) - P.k3;
        if (x < P.g || x > P.h) return -f_inf();
        return fma(P.c, fast_log(x), P.k5) - P.k4;
    }
    case ISO_PRIOR_FEH: {
        if (outside) return -f_inf();
        const double pdf = feh_pdf<>(P, x);
        return pdf != 0 ? fast_log(pdf) : -f_inf();
    }
    }
    return f_nan();
}

// EEP prior term: log( orig_prior(value) * derivative ), reference priors.py:423-429 + :130-140
__device__ __forceinline__ double eep_term_rt(const DevModel& M, const DevPrior& orig, double eep, double value,
                                           double deriv)
{
    if (eep < M.eep_lo || eep > M.eep_hi) return -f_inf();
    const double lc = ln_call_rt(orig, value);
    if (lc == -f_inf()) return (deriv != deriv) ? f_nan() : -f_inf();   // 0 * deriv
    return lc + fast_log(deriv);   // deriv == 0 -> -inf, deriv < 0 -> NaN, NaN -> NaN
}

// The same three functions with the prior family K known at compile time (the single-model sampler on the reference's
// default priors: no switch on a value that first has to arrive from memory).  Same expressions, written as selects
// instead of early returns - a wave takes every side anyway, and straight-line code lets the record's fields be
// fetched in one batch.  (Not used for the run-time form: the batch kernels sit at their register caps, and computing
// both sides of every test costs them scratch - measured 12 -> 28 B per lane on the headline kernel.)
// K = ISO_PRIOR_FEH implies the local two-Gaussian disk.
template <bool HAS_LX, int K>
__device__ __forceinline__ double ln_pdf_ct(const DevPrior& P, double x, double lx)
{
    const bool outside = (x < P.lo) || (x > P.hi);
    switch (K) {
    case ISO_PRIOR_FLAT: return outside ? -f_inf() : P.k1;
    case ISO_PRIOR_FLATLOG: return outside ? -f_inf() : fma(x, kLn10, P.k1);
    case ISO_PRIOR_POWERLAW: {
        const double l = HAS_LX ? lx : fast_log(x);
        return (P.bounded && outside) ? -f_inf() : fma(P.a, l, P.k1);
    }
    case ISO_PRIOR_GAUSS: {
        const double z = (x - P.a) * P.r0;
        return (P.bounded && outside) ? -f_inf() : (-0.5 * (z * z) + kLogInvRoot2Pi) - P.k1 - P.c;
    }
    case ISO_PRIOR_LOGNORMAL: return lognormal_ln(P, HAS_LX ? lx : fast_log(x));
    case ISO_PRIOR_CHABRIER: {
        const double l = HAS_LX ? lx : fast_log(x);
        const double below = lognormal_ln(P, l) - P.k3, above = fma(P.c, l, P.k5) - P.k4;
        return (x < P.d) ? below : ((x < P.g || x > P.h) ? -f_inf() : above);
    }
    case ISO_PRIOR_FEH: {
        const double pdf = feh_pdf<1>(P, x);
        return (!outside && pdf != 0) ? fast_log(pdf) : -f_inf();
    }
    }
    return f_nan();
}

template <int K>
__device__ __forceinline__ double ln_call_ct(const DevPrior& P, double x)
{
    const bool outside = (x < P.lo) || (x > P.hi);
    switch (K) {
    case ISO_PRIOR_FLAT: return outside ? -f_inf() : P.k1;
    case ISO_PRIOR_FLATLOG: return outside ? -f_inf() : fma(x, kLn10, P.k1);
    case ISO_PRIOR_POWERLAW: return outside ? -f_inf() : fma(P.a, fast_log(x), P.k1);
    case ISO_PRIOR_GAUSS: {
        const double z = (x - P.a) * P.r0;
        return outside ? -f_inf() : (-0.5 * (z * z) + kLogInvRoot2Pi) - P.k1 - P.c;
    }
    case ISO_PRIOR_LOGNORMAL: return (x < 0) ? -f_inf() : lognormal_ln(P, fast_log(x));
    case ISO_PRIOR_CHABRIER: {
        const double l = fast_log(x);
        const double below = lognormal_ln(P, l) - P.k3, above = fma(P.c, l, P.k5) - P.k4;
        const double inside = (x < P.d) ? ((x < 0) ? -f_inf() : below) : ((x < P.g || x > P.h) ? -f_inf() : above);
        return outside ? -f_inf() : inside;
    }
    case ISO_PRIOR_FEH: {
        const double pdf = feh_pdf<1>(P, x);
        return (!outside && pdf != 0) ? fast_log(pdf) : -f_inf();
    }
    }
    return f_nan();
}

template <int K>
__device__ __forceinline__ double eep_term_ct(const DevModel& M, const DevPrior& orig, double eep, double value,
                                           double deriv)
{
    const double lc = ln_call_ct<K>(orig, value);
    const double t = (lc == -f_inf()) ? ((deriv != deriv) ? f_nan() : -f_inf())   // 0 * deriv
                                      : lc + fast_log(deriv);                     // deriv == 0 -> -inf, deriv < 0 -> NaN, NaN -> NaN
    return (eep < M.eep_lo || eep > M.eep_hi) ? -f_inf() : t;
}

// dispatch: K < 0 = the family the record names
template <bool HAS_LX, int K = -1>
__device__ __forceinline__ double ln_pdf(const DevPrior& P, double x, double lx)
{
    if constexpr (K >= 0) return ln_pdf_ct<HAS_LX, K>(P, x, lx);
    else return ln_pdf_rt<HAS_LX>(P, x, lx);
}
template <int K = -1>
__device__ __forceinline__ double ln_call(const DevPrior& P, double x)
{
    if constexpr (K >= 0) return ln_call_ct<K>(P, x);
    else return ln_call_rt(P, x);
}
template <int K = -1>
__device__ __forceinline__ double eep_term(const DevModel& M, const DevPrior& orig, double eep, double value, double deriv)
{
    if constexpr (K >= 0) return eep_term_ct<K>(M, orig, eep, value, deriv);
    else return eep_term_rt(M, orig, eep, value, deriv);
}
