/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement, in plain C, of the reference's (timothydmorton/isochrones @ v2.1) hot path:
 * bracket search -> multilinear interpolation -> interp_mag -> star_lnlike -> lnprior -> lnpost.
 * Every function cites the reference file:line it restates.  It exists to (a) check the HIP
 * path in tests/, __graft_entry__.smoke() and (b) serve as bench.py's timed `cpu_baseline`
 * ("port").  Nothing in isochrones_amd/ may import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden vectors
 * produced by importing the reference itself (oracle/make_golden.py, run in the authoring
 * container where /root/reference is mounted), against the reference's own data-free tests
 * (isochrones/tests/test_interp.py:11-46) and the recorded known answers in
 * docs/interpolate.ipynb (cells 3, 5, 12, 14).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared [-fopenmp]  (no FMA contraction, so the
 * arithmetic is the same IEEE-754 double sequence the Python reference executes).
 *
 * Only the struct definitions (iso_prior, iso_model_desc) are shared with the product, via
 * include/isochrones_amd.h; no product code is called from here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/isochrones_amd.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_DIM 4

typedef struct orc_table {
    int ndim;
    int64_t shape[ORC_MAX_DIM + 1];      /* n_0..n_{ndim-1}, n_col */
    const double* grid;                  /* borrowed */
    const double* axes[ORC_MAX_DIM];     /* borrowed */
} orc_table;

/* ---------------------------------------------------------------------------------------
 * reference: isochrones/interp.py:10-35  searchsorted(arr, x, N)
 * Returns L; *eq = 1 on an exact hit (then L is the hit index), else L = #elements < x.
 * ------------------------------------------------------------------------------------- */
static int64_t orc_searchsorted(const double* arr, double x, int64_t N, int* eq)
{
    int64_t L = 0, R = N - 1;
    int done = 0;
    *eq = 0;
    int64_t m = (L + R) / 2;
    while (!done) {
        double xm = arr[m];
        if (xm < x) {
            L = m + 1;
        } else if (xm > x) {
            R = m - 1;
        } else if (xm == x) {
            L = m;
            *eq = 1;
            done = 1;
        }
        /* Python's // floors; (L+R) can only be -1 when R = -1, L = 0 -> m = -1 is never read */
        m = (L + R) >= 0 ? (L + R) / 2 : -1;
        if (L > R) done = 1;
    }
    return L;
}

/* ---------------------------------------------------------------------------------------
 * reference: isochrones/interp.py:63-93 (2d), :96-143 (3d), :146-205 (4d)  find_indices_*d
 * Returns 1 if out of bounds.  The exact-upper-edge query (x == ax[n-1]) makes the reference
 * read one element past the table (numba: unchecked; pure Python: IndexError), i.e. it is
 * undefined there; this restatement (and the HIP path) define it as i = n-2, t = 1.
 * ------------------------------------------------------------------------------------- */
static int orc_find_indices(const orc_table* T, const double* x, int64_t* idx, double* t)
{
    for (int d = 0; d < T->ndim; ++d) {
        const double* ax = T->axes[d];
        int64_t n = T->shape[d];
        if (x[d] < ax[0] || x[d] > ax[n - 1]) return 1;
    }
    for (int d = 0; d < T->ndim; ++d) {
        const double* ax = T->axes[d];
        int64_t n = T->shape[d];
        int eq;
        int64_t ix = orc_searchsorted(ax, x[d], n, &eq);
        if (eq) {
            if (ix == n - 1 && n > 1) {          /* upper-edge definition, see above */
                idx[d] = n - 2;
                t[d] = 1.0;
            } else {
                idx[d] = ix;
                t[d] = 0.0;
            }
        } else {
            idx[d] = ix - 1;
            double c0 = ax[ix - 1];
            t[d] = (x[d] - c0) / (ax[ix] - c0);
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * reference: isochrones/interp.py:208-249 (2d), :252-293 (3d), :296-338 (4d)  interp_value_*d
 * NaN in -> NaN out; oob -> NaN; corner j offsets dim k by bit (ndim-1-k) of j; weight is the
 * running product over k starting from 1.0; values accumulate in corner order from 0.0; zero-
 * weight corners are still multiplied (a NaN neighbour poisons the result).
 * ------------------------------------------------------------------------------------- */
static void orc_interp_value(const orc_table* T, const double* x, const int32_t* icols, int k,
                             double* values)
{
    const int ndim = T->ndim;
    for (int d = 0; d < ndim; ++d) {
        if (x[d] != x[d]) {
            for (int c = 0; c < k; ++c) values[c] = NAN;
            return;
        }
    }
    int64_t idx[ORC_MAX_DIM];
    double t[ORC_MAX_DIM];
    if (orc_find_indices(T, x, idx, t)) {
        for (int c = 0; c < k; ++c) values[c] = NAN;
        return;
    }
    const int n_edges = 1 << ndim;
    const int64_t ncol = T->shape[ndim];
    for (int c = 0; c < k; ++c) values[c] = 0.0;
    for (int j = 0; j < n_edges; ++j) {
        double weight = 1.0;
        int64_t flat = 0;
        for (int d = 0; d < ndim; ++d) {
            int off = (j >> (ndim - 1 - d)) & 1;
            if (off) weight *= t[d];
            else weight *= 1 - t[d];
            flat = flat * T->shape[d] + (idx[d] + off);
        }
        const double* cell = T->grid + flat * ncol;
        for (int c = 0; c < k; ++c) values[c] += cell[icols[c]] * weight;
    }
}

/* batch form: reference isochrones/interp.py:341-392 interp_values_*d (serial loop) */
void orc_interp(const orc_table* T, const double* const* x, int64_t n, const int32_t* icols, int k,
                double* out, int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double xi[ORC_MAX_DIM];
        for (int d = 0; d < T->ndim; ++d) xi[d] = x[d][i];
        orc_interp_value(T, xi, icols, k, out + i * k);
    }
}

/* ---------------------------------------------------------------------------------------
 * the ModelGridInterpolator binding: reference isochrones/models.py:253-445, order tuples
 * :669 (track (2,0,1,3,4)) and :696 (iso (1,2,0,3,4)).
 * ------------------------------------------------------------------------------------- */
typedef struct orc_ic {
    orc_table model;     /* 3-D */
    orc_table bc;        /* 4-D */
    int kind;            /* ISO_KIND_TRACK / ISO_KIND_ISO */
    int32_t i_Teff, i_logg, i_feh, i_Mbol;
    int32_t i_prior_val, i_prior_deriv;   /* (age, dt_deep) or (mass, dm_deep) */
    int32_t i_numax, i_dnu;
} orc_ic;

static void orc_order(int kind, int order[5])
{
    static const int trk[5] = {2, 0, 1, 3, 4};
    static const int iso[5] = {1, 2, 0, 3, 4};
    memcpy(order, kind == ISO_KIND_TRACK ? trk : iso, sizeof(trk));
}

/* reference: isochrones/mags.py:8-61  interp_mag */
static void orc_interp_mag1(const orc_ic* ic, const double pars[5], const int32_t* bc_cols, int nb,
                            double* Teff, double* logg, double* feh, double* mags)
{
    int o[5];
    orc_order(ic->kind, o);
    double x3[3] = {pars[o[0]], pars[o[1]], pars[o[2]]};
    int32_t cols[4] = {ic->i_Teff, ic->i_logg, ic->i_feh, ic->i_Mbol};
    double star[4];
    orc_interp_value(&ic->model, x3, cols, 4, star);
    *Teff = star[0];
    *logg = star[1];
    *feh = star[2];
    double AV = pars[o[4]];
    double x4[4] = {star[0], star[1], star[2], AV};
    double bc[ISO_MAX_BANDS];
    orc_interp_value(&ic->bc, x4, bc_cols, nb, bc);
    double mBol = star[3];
    double dist_mod = 5 * log10(pars[o[3]] / 10.0);
    for (int b = 0; b < nb; ++b) mags[b] = mBol + dist_mod - bc[b];
}

/* reference: isochrones/mags.py:64-124  interp_mags (serial loop over samples) */
void orc_interp_mag(const orc_ic* ic, const double* pars, int64_t stride_n, int64_t stride_p, int64_t n,
                    const int32_t* bc_cols, int nb, double* Teff, double* logg, double* feh,
                    double* mags, int nthreads)
{
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double p[5], T, g, f, m[ISO_MAX_BANDS];
        for (int j = 0; j < 5; ++j) p[j] = pars[i * stride_n + j * stride_p];
        orc_interp_mag1(ic, p, bc_cols, nb, &T, &g, &f, m);
        if (Teff) Teff[i] = T;
        if (logg) logg[i] = g;
        if (feh) feh[i] = f;
        if (mags)
            for (int b = 0; b < nb; ++b) mags[i * nb + b] = m[b];
    }
}

/* reference: isochrones/utils.py:67-75  fast_addmags */
static double orc_fast_addmags(const double* mags, int n)
{
    double tot = 0;
    for (int i = 0; i < n; ++i) tot += pow(10.0, -0.4 * mags[i]);
    return -2.5 * log10(tot);
}

/* reference: isochrones/likelihood.py:7,10-13  gauss_lnprob (note: +log(unc), as written) */
static double orc_gauss_lnprob(double val, double unc, double model_val)
{
    const double LOG_ONE_OVER_ROOT_2PI = log(1.0 / sqrt(2 * M_PI));
    double resid = val - model_val;
    return LOG_ONE_OVER_ROOT_2PI + log(unc) - 0.5 * resid * resid / (unc * unc);
}

/* component c of an n_stars system: reference isochrones/likelihood.py:40-54 */
static void orc_component_pars(const double* pars, int n_stars, int c, double out[5])
{
    out[0] = pars[c];
    for (int j = 0; j < 4; ++j) out[1 + j] = pars[n_stars + j];
}

/* reference: isochrones/likelihood.py:16-147  star_lnlike */
static double orc_star_lnlike(const orc_ic* ic, const iso_model_desc* d, const double* pars)
{
    const int nb = d->n_bands;
    double Teff = 0, logg = 0, feh = 0;
    double mags[ISO_MAX_STARS][ISO_MAX_BANDS];
    for (int c = 0; c < d->n_stars; ++c) {
        double p[5], T, g, f;
        orc_component_pars(pars, d->n_stars, c, p);
        orc_interp_mag1(ic, p, d->bc_cols, nb, &T, &g, &f, mags[c]);
        if (c == 0) {
            Teff = T;
            logg = g;
            feh = f;
        }
    }
    double tot[ISO_MAX_BANDS];
    for (int b = 0; b < nb; ++b) {
        if (d->n_stars == 1) {
            tot[b] = mags[0][b];
        } else {
            double m[ISO_MAX_STARS];
            for (int c = 0; c < d->n_stars; ++c) m[c] = mags[c][b];
            tot[b] = orc_fast_addmags(m, d->n_stars);
        }
    }
    double lnlike = 0;
    const double model_spec[3] = {Teff, logg, feh};
    for (int q = 0; q < 3; ++q) {
        double val = d->spec_val[q];
        if (val == val) lnlike += orc_gauss_lnprob(val, d->spec_unc[q], model_spec[q]);
    }
    for (int b = 0; b < nb; ++b) lnlike += orc_gauss_lnprob(d->mag_val[b], d->mag_unc[b], tot[b]);
    return lnlike;
}

/* reference: isochrones/starmodel.py:1563-1614  BasicStarModel.lnlike (parallax + asteroseismic) */
static double orc_lnlike1(const orc_ic* ic, const iso_model_desc* d, const double* pars)
{
    double lnlike = orc_star_lnlike(ic, d, pars);
    const int i_dist = d->n_stars + 2;
    if (d->has_parallax) lnlike += orc_gauss_lnprob(d->plx_val, d->plx_unc, 1000.0 / pars[i_dist]);
    if (d->has_numax) {
        double p[5];
        orc_component_pars(pars, d->n_stars, 0, p);
        int o[5];
        orc_order(ic->kind, o);
        double x3[3] = {p[o[0]], p[o[1]], p[o[2]]};
        int32_t cols[2] = {ic->i_numax, ic->i_dnu};
        double v[2];
        orc_interp_value(&ic->model, x3, cols, 2, v);
        lnlike += orc_gauss_lnprob(d->numax_val, d->numax_unc, v[0]);
        if (d->has_dnu) lnlike += orc_gauss_lnprob(d->dnu_val, d->dnu_unc, v[1]);
    }
    return lnlike;
}

/* ---------------------------------------------------------------------------------------
 * priors — reference isochrones/priors.py.  Two evaluation modes exist in the reference:
 *   lnpdf(x)   (Prior.lnpdf :61-66, BoundedPrior.lnpdf :130-140)
 *   prior(x)   (Prior.__call__ :35-36 -> pdf :54-59; BoundedPrior.__call__ :112-117)
 * ------------------------------------------------------------------------------------- */
static const double ORC_ROOT_2PI = 2.5066282746310002; /* sqrt(2*pi), priors.py:16 */

static double orc_powerlaw_C(double alpha, double lo, double hi)
{
    /* priors.py:314-317 */
    return (1 + alpha) / (pow(hi, 1 + alpha) - pow(lo, 1 + alpha));
}

static double orc_lognormal_pdf(double x, double mu, double sigma)
{
    /* priors.py:272-275 */
    double scale = exp(mu);
    double y = x / scale;
    double ly = log(y) / sigma;
    return (1.0 / ORC_ROOT_2PI) / (sigma * y) * exp(-0.5 * (ly * ly)) / scale;
}

static double orc_lognormal_lnpdf(double x, double mu, double sigma)
{
    /* priors.py:277-280 */
    double scale = exp(mu);
    double y = x / scale;
    double ly = log(y) / sigma;
    return log(1.0 / ORC_ROOT_2PI) - (log(sigma) + log(y)) - 0.5 * (ly * ly) - mu;
}

static double orc_feh_pdf(double feh, double halo_fraction, int local)
{
    /* priors.py:359-381 */
    double disk;
    if (local) {
        const double disk_norm = 2.5066282746310007;
        disk = 1.0 / disk_norm *
               (0.8 / 0.15 * exp(-0.5 * pow(feh - 0.016, 2.0) / pow(0.15, 2.0)) +
                0.2 / 0.22 * exp(-0.5 * pow(feh + 0.15, 2.0) / pow(0.22, 2.0)));
    } else {
        const double mu = -0.3, sig = 0.3;
        disk = 1.0 / sqrt(2 * M_PI) / sig * exp(-0.5 * pow(feh - mu, 2) / pow(sig, 2));
    }
    const double halo_mu = -1.5, halo_sig = 0.4;
    double halo = 1.0 / sqrt(2 * M_PI * pow(halo_sig, 2)) * exp(-0.5 * pow(feh - halo_mu, 2) / pow(halo_sig, 2));
    return halo_fraction * halo + (1 - halo_fraction) * disk;
}

/* raw _pdf(x) of a family, before bounds / _norm handling */
static double orc_prior_raw_pdf(const iso_prior* P, double x);

/* prior(x): the __call__ form */
static double orc_prior_call(const iso_prior* P, double x)
{
    switch (P->kind) {
    case ISO_PRIOR_FLAT:
    case ISO_PRIOR_FLATLOG:
    case ISO_PRIOR_POWERLAW:
    case ISO_PRIOR_GAUSS:
        /* BoundedPrior.__call__ :112-117 then Prior.pdf :54-59 (same test; _norm == 1) */
        if (P->bounded && (x < P->lo || x > P->hi)) return 0;
        if (x < P->lo || x > P->hi) return 0;
        return orc_prior_raw_pdf(P, x) / 1.0;
    case ISO_PRIOR_LOGNORMAL:
        /* Prior.__call__ -> pdf with bounds (0, inf): priors.py:268 */
        if (x < 0 || x > INFINITY) return 0;
        return orc_lognormal_pdf(x, P->a, P->b) / 1.0;
    case ISO_PRIOR_CHABRIER:
        /* Prior.pdf :54-59 with _norm == 1, then BrokenPrior._pdf :205-207 */
        if (x < P->lo || x > P->hi) return 0;
        return orc_prior_raw_pdf(P, x) / 1.0;
    case ISO_PRIOR_FEH:
        /* Prior.pdf :54-59, _norm = quad(_pdf, lo, hi) :42-45 (passed in as P->b) */
        if (x < P->lo || x > P->hi) return 0;
        return orc_feh_pdf(x, P->a, P->c != 0.0) / P->b;
    }
    return NAN;
}

static double orc_prior_raw_pdf(const iso_prior* P, double x)
{
    switch (P->kind) {
    case ISO_PRIOR_FLAT: /* :287-289 */
        return 1.0 / (P->hi - P->lo);
    case ISO_PRIOR_FLATLOG: /* :300-302 */
        return log(10) * pow(10, x) / (pow(10, P->hi) - pow(10, P->lo));
    case ISO_PRIOR_POWERLAW: /* :314-318 */
        return orc_powerlaw_C(P->a, P->lo, P->hi) * pow(x, P->a);
    case ISO_PRIOR_GAUSS: { /* :254-255 (norm = exp(lognorm)) */
        double z = (x - P->a) / P->b;
        return exp(-(z * z) / 2.0) / ORC_ROOT_2PI / P->b / exp(P->c);
    }
    case ISO_PRIOR_LOGNORMAL:
        return orc_lognormal_pdf(x, P->a, P->b);
    case ISO_PRIOR_CHABRIER: {
        /* np.digitize(x, [bp]): 0 if x < bp else 1 (NaN -> 1) */
        int i = (x < P->d) ? 0 : 1;
        if (i == 0) {
            /* components[0](x): LogNormalPrior.__call__ */
            double c = (x < 0 || x > INFINITY) ? 0 : orc_lognormal_pdf(x, P->a, P->b) / 1.0;
            return c / P->e;
        } else {
            /* components[1](x): PowerLawPrior (BoundedPrior.__call__) on (g, h) */
            double c;
            if (x < P->g || x > P->h) c = 0;
            else c = orc_powerlaw_C(P->c, P->g, P->h) * pow(x, P->c) / 1.0;
            return c / P->f;
        }
    }
    case ISO_PRIOR_FEH:
        return orc_feh_pdf(x, P->a, P->c != 0.0);
    }
    return NAN;
}

/* lnpdf(x) */
static double orc_prior_lnpdf(const iso_prior* P, double x)
{
    switch (P->kind) {
    case ISO_PRIOR_FLAT:
    case ISO_PRIOR_FLATLOG: {
        /* BoundedPrior.lnpdf :130-140, no _lnpdf -> log(self.pdf(x)) if pdf else -inf */
        if (P->bounded && (x < P->lo || x > P->hi)) return -INFINITY;
        double pdf = (x < P->lo || x > P->hi) ? 0 : orc_prior_raw_pdf(P, x) / 1.0;
        return pdf != 0 ? log(pdf) : -INFINITY;
    }
    case ISO_PRIOR_POWERLAW: /* :320-323 */
        if (P->bounded && (x < P->lo || x > P->hi)) return -INFINITY;
        return log(orc_powerlaw_C(P->a, P->lo, P->hi)) + P->a * log(x);
    case ISO_PRIOR_GAUSS: { /* :256-257 */
        if (P->bounded && (x < P->lo || x > P->hi)) return -INFINITY;
        double z = (x - P->a) / P->b;
        return (-(z * z) / 2.0 - log(ORC_ROOT_2PI)) - log(P->b) - P->c;
    }
    case ISO_PRIOR_LOGNORMAL: /* Prior.lnpdf -> _lnpdf, no bounds test */
        return orc_lognormal_lnpdf(x, P->a, P->b);
    case ISO_PRIOR_CHABRIER: { /* BrokenPrior._lnpdf :209-211 */
        int i = (x < P->d) ? 0 : 1;
        if (i == 0) return orc_lognormal_lnpdf(x, P->a, P->b) - log(P->e);
        if (x < P->g || x > P->h) return -INFINITY - log(P->f);
        return (log(orc_powerlaw_C(P->c, P->g, P->h)) + P->c * log(x)) - log(P->f);
    }
    case ISO_PRIOR_FEH: { /* Prior.lnpdf :61-66 -> self(x) */
        double pdf = orc_prior_call(P, x);
        return pdf != 0 ? log(pdf) : -INFINITY;
    }
    }
    return NAN;
}

/* reference: isochrones/priors.py:409-429 EEP_prior via BoundedPrior.lnpdf :130-140 */
static double orc_eep_lnpdf(const orc_ic* ic, const iso_model_desc* d, double eep, double other, double feh)
{
    if (eep < d->eep_lo || eep > d->eep_hi) return -INFINITY;
    /* Prior.pdf :54-59 — same bounds, _norm == 1 */
    double pars5[5];
    const iso_prior* orig;
    if (ic->kind == ISO_KIND_TRACK) { /* pars = [mass, eep, feh] */
        pars5[0] = other; pars5[1] = eep; pars5[2] = feh;
        orig = &d->prior_age;
    } else {                          /* pars = [eep, age, feh] */
        pars5[0] = eep; pars5[1] = other; pars5[2] = feh;
        orig = &d->prior_mass;
    }
    int o[5];
    orc_order(ic->kind, o);
    double x3[3] = {pars5[o[0]], pars5[o[1]], pars5[o[2]]};
    int32_t cols[2] = {ic->i_prior_val, ic->i_prior_deriv};
    double v[2];
    orc_interp_value(&ic->model, x3, cols, 2, v);
    double pdf = orc_prior_call(orig, v[0]) * v[1];
    pdf = pdf / 1.0;
    /* `np.log(pdf) if pdf else -np.inf`: NaN is truthy -> log(NaN) = NaN; negative -> NaN */
    return pdf != 0 ? log(pdf) : -INFINITY;
}

/* reference: isochrones/starmodel.py:1616-1635  BasicStarModel.lnprior */
static double orc_lnprior1(const orc_ic* ic, const iso_model_desc* d, const double* pars)
{
    const int N = d->n_stars;
    double lnp = 0;
    if (N == 2) {
        if (pars[1] > pars[0]) return -INFINITY;
    } else if (N == 3) {
        /* operator precedence as written: (not (p0 > p1)) and (p1 > p2) */
        if (!(pars[0] > pars[1]) && (pars[1] > pars[2])) return -INFINITY;
    }
    if (ic->kind == ISO_KIND_TRACK) {
        /* param_names = (mass, eep, feh, distance, AV); mass_index 0, feh_index 2 */
        lnp += orc_prior_lnpdf(&d->prior_mass, pars[0]);
        lnp += orc_eep_lnpdf(ic, d, pars[1], pars[0], pars[2]);
        lnp += orc_prior_lnpdf(&d->prior_feh, pars[2]);
        lnp += orc_prior_lnpdf(&d->prior_distance, pars[3]);
        lnp += orc_prior_lnpdf(&d->prior_AV, pars[4]);
    } else {
        /* (eep[, eep_1[, eep_2]], age, feh, distance, AV); age_index N, feh_index N+1 */
        for (int c = 0; c < N; ++c) lnp += orc_eep_lnpdf(ic, d, pars[c], pars[N], pars[N + 1]);
        lnp += orc_prior_lnpdf(&d->prior_age, pars[N]);
        lnp += orc_prior_lnpdf(&d->prior_feh, pars[N + 1]);
        lnp += orc_prior_lnpdf(&d->prior_distance, pars[N + 2]);
        lnp += orc_prior_lnpdf(&d->prior_AV, pars[N + 3]);
    }
    return lnp;
}

/* reference: isochrones/starmodel.py:538-542  StarModel.lnpost */
void orc_lnpost(const orc_ic* ic, const iso_model_desc* d, const double* pars, int64_t stride_n,
                int64_t stride_p, int64_t n, double* lnpost, double* lnprior, double* lnlike,
                int nthreads)
{
    const int np_ = d->n_stars + 4;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double p[ISO_MAX_PARAMS];
        for (int j = 0; j < np_; ++j) p[j] = pars[i * stride_n + j * stride_p];
        double lp = orc_lnprior1(ic, d, p);
        double ll = NAN;
        int need_like = isfinite(lp) || lnlike != NULL;
        if (need_like) ll = orc_lnlike1(ic, d, p);
        if (lnprior) lnprior[i] = lp;
        if (lnlike) lnlike[i] = ll;
        if (lnpost) lnpost[i] = isfinite(lp) ? lp + ll : -INFINITY;
    }
}

/* reference: isochrones/starmodel.py:1637-1640  mnest_prior (bounds(par) per parameter) */
void orc_unit_cube(const iso_model_desc* d, int kind, double* cube, int64_t stride_n, int64_t stride_p,
                   int64_t n)
{
    (void)kind;
    const int np_ = d->n_stars + 4;
    for (int64_t i = 0; i < n; ++i)
        for (int p = 0; p < np_; ++p) {
            double* c = cube + i * stride_n + p * stride_p;
            double lo = d->bound_lo[p], hi = d->bound_hi[p];
            *c = (hi - lo) * *c + lo;
        }
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------
 * "next" row f2 — reference: isochrones/interp.py:488-558  interp_eep / interp_eeps
 * (mass, age, feh) -> EEP on the ragged per-track age arrays.
 *   ages[n0*n1][n_eep]  log10 age along every (feh, mass) track, NaN past `lengths`
 *   x = age, x0 = feh (axis ax0), x1 = mass (axis ax1)
 * The reference reads, but never uses, the dt_deep weights; the substitution chain for tracks
 * that end before `x` is sequential, exactly as written there.
 * ------------------------------------------------------------------------------------- */
static int64_t orc_eep_index(const double* arr, double x, int64_t N)
{
    int eq;
    if (N <= 0) return 0;
    return orc_searchsorted(arr, x, N, &eq);
}

double orc_interp_eep1(double x, double x0, double x1, const double* ax0, int64_t n0, const double* ax1,
                       int64_t n1, const double* ages, const int64_t* lengths, int64_t n_eep)
{
    if (x != x || x0 != x0 || x1 != x1) return NAN;
    orc_table T;
    T.ndim = 2;
    T.shape[0] = n0; T.shape[1] = n1; T.shape[2] = 1;
    T.grid = NULL;
    T.axes[0] = ax0; T.axes[1] = ax1;
    double xs[2] = {x0, x1};
    int64_t idx[ORC_MAX_DIM];
    double d[ORC_MAX_DIM];
    if (orc_find_indices(&T, xs, idx, d)) return NAN;
    const int64_t i0 = idx[0], i1 = idx[1];
    const int64_t ind[4] = {i0 * n1 + i1, i0 * n1 + (i1 + 1), (i0 + 1) * n1 + i1, (i0 + 1) * n1 + (i1 + 1)};
    int64_t ie[4];
    for (int k = 0; k < 4; ++k) ie[k] = orc_eep_index(ages + ind[k] * n_eep, x, lengths[ind[k]]);
    const int64_t max_i = n_eep - 1;
    for (int k = 0; k < 4; ++k)
        if (ie[k] > max_i) return NAN;
    double e[4];
    for (int k = 0; k < 4; ++k) e[k] = (double)(ie[k] + 1);
    if (ie[0] >= lengths[ind[0]]) e[0] = e[1];
    if (ie[1] >= lengths[ind[1]]) e[1] = e[0];
    if (ie[2] >= lengths[ind[2]]) e[2] = e[3];
    if (ie[3] >= lengths[ind[3]]) e[3] = e[2];
    const double d0 = d[0], d1 = d[1];
    const double eep_0 = (1 - d1) * e[0] + d1 * e[1];
    const double eep_1 = (1 - d1) * e[2] + d1 * e[3];
    return (1 - d0) * eep_0 + d0 * eep_1;
}

void orc_interp_eep(const double* x, const double* x0, const double* x1, int64_t n, const double* ax0, int64_t n0,
                    const double* ax1, int64_t n1, const double* ages, const int64_t* lengths, int64_t n_eep,
                    double* out)
{
    for (int64_t i = 0; i < n; ++i)
        out[i] = orc_interp_eep1(x[i], x0[i], x1[i], ax0, n0, ax1, n1, ages, lengths, n_eep);
}

/* ---------------------------------------------------------------------------------------
 * "next" row f4 — generic StarModel over an ObservationTree.
 * reference: isochrones/starmodel.py:538-613 (lnpost / lnlike / lnprior),
 *            isochrones/observation.py:464-491 (ObsNode.lnlike), :1116-1130 (p2pardict),
 *            :1181-1234 (ObservationTree.lnlike), isochrones/utils.py:43-64 (addmags).
 * The tree itself is flattened on the host (iso_tree_desc); this restates the arithmetic.
 * ------------------------------------------------------------------------------------- */
static double orc_addmags_mask(const double mags[][ISO_TREE_MAX_BANDS], uint32_t mask, int band, int n_leaves)
{
    /* utils.addmags: tot = sum 10**(-0.4 m); -2.5*log10(tot) — also for a single star */
    double tot = 0;
    for (int l = 0; l < n_leaves; ++l)
        if (mask & (1u << l)) tot += pow(10.0, -0.4 * mags[l][band]);
    return -2.5 * log10(tot);
}

static void orc_tree_leaf_pars(const iso_tree_desc* d, const double* p, int leaf, double out[5])
{
    int base = 0;
    for (int s = 0; s < d->leaf_system[leaf]; ++s) base += d->n_stars[s] + 4;
    const int N = d->n_stars[d->leaf_system[leaf]];
    out[0] = p[base + d->leaf_slot[leaf]];
    for (int j = 0; j < 4; ++j) out[1 + j] = p[base + N + j];
}

static double orc_tree_lnprior1(const orc_ic* ic, const iso_tree_desc* d, const double* p)
{
    /* starmodel.py:557-613 (isochrone grids only) */
    iso_model_desc shim;                      /* eep prior helper reads eep bounds + priors from here */
    memset(&shim, 0, sizeof(shim));
    shim.prior_mass = d->prior_mass;
    shim.prior_age = d->prior_age;
    shim.eep_lo = d->eep_lo;
    shim.eep_hi = d->eep_hi;
    const iso_prior* pri[4] = {&d->prior_age, &d->prior_feh, &d->prior_distance, &d->prior_AV};
    double lnp = 0;
    int i = 0;
    for (int s = 0; s < d->n_systems; ++s) {
        const int N = d->n_stars[s];
        for (int j = 0; j < 4; ++j) {
            const double val = p[i + N + j];
            if (val < d->bound_lo[j] || val > d->bound_hi[j]) return -INFINITY;
            lnp += orc_prior_lnpdf(pri[j], val);
            if (!isfinite(lnp)) return -INFINITY;
        }
        for (int j = 1; j < N; ++j)
            if (!(p[i + j] <= p[i + j - 1])) return -INFINITY;      /* (eeps[1:] <= eeps[:-1]).all() */
        for (int j = 0; j < N; ++j) lnp += orc_eep_lnpdf(ic, &shim, p[i + j], p[i + N], p[i + N + 1]);
        i += N + 4;
    }
    return lnp;
}

static double orc_tree_lnlike1(const orc_ic* ic, const iso_tree_desc* d, const double* p)
{
    const double L = log(1.0 / sqrt(2 * M_PI));
    double mags[ISO_TREE_MAX_LEAVES][ISO_TREE_MAX_BANDS];
    double spec[ISO_TREE_MAX_LEAVES][3];
    for (int l = 0; l < d->n_leaves; ++l) {
        double q[5];
        orc_tree_leaf_pars(d, p, l, q);
        orc_interp_mag1(ic, q, d->bc_cols, d->n_bands, &spec[l][0], &spec[l][1], &spec[l][2], mags[l]);
    }
    double lnl = 0;
    for (int t = 0; t < d->n_terms; ++t) {
        const iso_tree_term* T = &d->terms[t];
        double mag = T->mag, mod;
        if (T->relative) {
            mod = orc_addmags_mask(mags, T->mask, T->band, d->n_leaves) -
                  orc_addmags_mask(mags, T->ref_mask, T->band, d->n_leaves);
            mag -= T->ref_mag;
        } else {
            mod = orc_addmags_mask(mags, T->mask, T->band, d->n_leaves);
        }
        lnl += -0.5 * ((mag - mod) * (mag - mod)) / (T->unc * T->unc) + L + log(T->unc);
        if (!isfinite(lnl)) return -INFINITY;
    }
    for (int k = 0; k < d->n_spec; ++k) {
        const iso_tree_prop* S = &d->spec[k];
        const double mod = spec[S->leaf][S->prop];
        lnl += -0.5 * ((S->a - mod) * (S->a - mod)) / (S->b * S->b) + L + log(S->b);
        if (!isfinite(lnl)) return -INFINITY;       /* checked per label in the reference; same outcome */
    }
    for (int k = 0; k < d->n_limits; ++k) {
        const iso_tree_prop* S = &d->limits[k];
        const double mod = spec[S->leaf][S->prop];
        if (mod < S->a || mod > S->b || !isfinite(mod)) return -INFINITY;
    }
    int i = 0;
    for (int s = 0; s < d->n_systems; ++s) {         /* parallax, then AV, per system */
        const int N = d->n_stars[s];
        if (d->has_plx[s]) {
            const double mod = 1.0 / p[i + N + 2] * 1000.0;
            lnl += -0.5 * ((d->plx_val[s] - mod) * (d->plx_val[s] - mod)) / (d->plx_unc[s] * d->plx_unc[s]) + L +
                   log(d->plx_unc[s]);
        }
        i += N + 4;
    }
    i = 0;
    for (int s = 0; s < d->n_systems; ++s) {
        const int N = d->n_stars[s];
        if (d->has_av[s]) {
            const double AV = p[i + N + 3];
            lnl += -0.5 * ((d->av_val[s] - AV) * (d->av_val[s] - AV)) / (d->av_unc[s] * d->av_unc[s]) + L +
                   log(d->av_unc[s]);
        }
        i += N + 4;
    }
    if (!isfinite(lnl)) return -INFINITY;
    return lnl;
}

void orc_tree_lnpost(const orc_ic* ic, const iso_tree_desc* d, const double* pars, int64_t stride_n,
                     int64_t stride_p, int64_t n, double* lnpost, double* lnprior, double* lnlike, int nthreads)
{
    int np_ = 0;
    for (int s = 0; s < d->n_systems; ++s) np_ += d->n_stars[s] + 4;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (int64_t i = 0; i < n; ++i) {
        double p[ISO_TREE_MAX_PARAMS];
        for (int j = 0; j < np_; ++j) p[j] = pars[i * stride_n + j * stride_p];
        const double lp = orc_tree_lnprior1(ic, d, p);
        double ll = NAN;
        if (isfinite(lp) || lnlike) ll = orc_tree_lnlike1(ic, d, p);
        if (lnprior) lnprior[i] = lp;
        if (lnlike) lnlike[i] = ll;
        if (lnpost) lnpost[i] = isfinite(lp) ? lp + ll : -INFINITY;      /* starmodel.py:538-542 */
    }
}
