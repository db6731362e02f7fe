/*
 * loglike_ref.c -- plain-C CPU restatement of brutus `fitting.loglike`.
 *
 * TEST INFRASTRUCTURE ONLY (checker + `cpu_baseline` of bench.py).  The product
 * package never links or calls this file.
 *
 * Parity pin: checked against the reference-generated golden vectors in
 * tests/golden/ (tests/test_oracle_golden.py::test_c_oracle_*), which were
 * produced by importing the upstream Python (tools/gen_golden.py).
 *
 * Follows the reference statement by statement (scalar loops like the numba
 * kernels, same association order; compile with -ffp-contract=off):
 *   brutus/utils.py:330-345     _get_seds
 *   brutus/fitting.py:158-264   _optimize_fit_mag
 *   brutus/fitting.py:385-420   _optimize_fit_flux
 *   brutus/fitting.py:502-576   _get_sed_mle
 *   brutus/fitting.py:691-820   loglike
 *   brutus/utils.py:161-176     _chisquare_logpdf
 * The numpy row sums at fitting.py:745,792,807 use numpy's pairwise-8 order
 * for 8 <= n <= 128; rowsum() reproduces it.
 *
 * OpenMP parallelises over models inside one star (the reference is serial);
 * bench.py's CPU baseline instead runs one serial star per host thread
 * (brutus_ref_set_threads(1) + a thread pool), which scales better.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double avlim[2], av_gauss[2], rvlim[2], rv_gauss[2];
    double ltol, ltol_subthresh, init_thresh;
    int32_t dim_prior, max_iter;
} ref_params;

int brutus_ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Per-thread OpenMP team size for subsequent calls from the calling thread
 * (1 = serial inside a star, so that the caller can parallelise over stars). */
void brutus_ref_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* numpy pairwise summation of a contiguous row (n < 128: one block). */
static double rowsum(const double *a, int n) {
    if (n < 8) {
        double s = 0.;
        for (int i = 0; i < n; ++i) s += a[i];
        return s;
    }
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s += a[i];
    return s;
}

/* fitting.py:502-576 for one model.  F/Rf/Df/res are length-nb outputs. */
static void sed_mle(const float *mc /* (nb,3) gathered */, int nb, const double *d,
                    const double *V, double av, double rv, const ref_params *p,
                    double *F, double *Rf, double *Df, double *res, double *scale,
                    double *icov /* 9 */) {
    const double fac = -0.4 * log(10.);
    double s_num = 0., s_den = 0.;
    for (int j = 0; j < nb; ++j) {
        const double m = mc[3 * j], r0 = mc[3 * j + 1], dr = mc[3 * j + 2];
        double drv = dr;
        double rvec = r0 + rv * dr;
        double sed = m + av * rvec;
        sed = pow(10., -0.4 * sed);
        rvec *= fac * sed;
        drv *= fac * sed;
        F[j] = sed;
        Rf[j] = rvec;
        Df[j] = drv;
    }
    for (int j = 0; j < nb; ++j) {
        s_num += F[j] * d[j] / V[j];
        s_den += F[j] * F[j] / V[j];
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;
    double sr_mix = 0., sa_mix = 0., a_den = 0., r_den = 0., ar_mix = 0.;
    const double Av_varinv = 1. / (p->av_gauss[1] * p->av_gauss[1]);
    const double Rv_varinv = 1. / (p->rv_gauss[1] * p->rv_gauss[1]);
    const double a_reg = 1. / (0.05 * 0.05), r_reg = 1. / (0.1 * 0.1);
    for (int j = 0; j < nb; ++j) {
        const double models_int = pow(10., -0.4 * (double)mc[3 * j]);
        double reddening = F[j] - models_int;
        F[j] = F[j] * s;
        res[j] = d[j] - F[j];
        sr_mix += Df[j] * ((F[j] - res[j]) / V[j]);
        sa_mix += Rf[j] * ((F[j] - res[j]) / V[j]);
        Rf[j] = Rf[j] * s;
        Df[j] = Df[j] * s;
        reddening *= s;
        ar_mix += Df[j] * ((reddening - res[j]) / V[j]);
        a_den += Rf[j] * Rf[j] / V[j];
        r_den += Df[j] * Df[j] / V[j];
    }
    a_den += Av_varinv;
    r_den += Rv_varinv;
    a_den += a_reg;
    r_den += r_reg;
    *scale = s;
    icov[0] = s_den;
    icov[4] = a_den;
    icov[8] = r_den;
    icov[1] = icov[3] = sa_mix;
    icov[2] = icov[6] = sr_mix;
    icov[5] = icov[7] = ar_mix;
}

static double chi2_of(const double *res, const double *V, int nb, double *tmp) {
    for (int j = 0; j < nb; ++j) tmp[j] = res[j] * res[j] / V[j];
    return rowsum(tmp, nb);
}

/* Returns 0 on success, -4 if an iteration cap was hit. */
int brutus_ref_loglike(const float *models, int64_t nmodel, int nfilt, const double *flux,
                       const double *err, const uint8_t *mask, double parallax,
                       double parallax_err, int has_parallax, const ref_params *p, double *lnl,
                       double *chi2, double *scale, double *av, double *rv, double *icov,
                       int32_t *ndim_out, int32_t *k1_out, int32_t *k2_out,
                       int64_t *nsel_out) {
    const int max_iter = p->max_iter > 0 ? p->max_iter : 100000;
    /* fitting.py:706-725: clean mask, magnitudes */
    int sel[64];
    int nb = 0;
    for (int j = 0; j < nfilt && nb < 64; ++j)
        if (mask[j] && isfinite(flux[j]) && isfinite(err[j]) && err[j] > 0.) sel[nb++] = j;
    *ndim_out = nb;
    double d[64], V[64], g[64], W[64];
    const double kmag = 2.5 / log(10.);
    for (int j = 0; j < nb; ++j) {
        d[j] = flux[sel[j]];
        V[j] = err[sel[j]] * err[sel[j]];
        g[j] = -2.5 * log10(d[j]);
        W[j] = kmag * kmag * V[j] / (d[j] * d[j]);
        if (!isfinite(g[j])) {
            g[j] = 0.;
            W[j] = 1e50;
        }
    }
    const double avmin = p->avlim[0], avmax = p->avlim[1];
    const double rvmin = p->rvlim[0], rvmax = p->rvlim[1];
    const double Av_mean = p->av_gauss[0], Rv_mean = p->rv_gauss[0];
    const double Av_varinv = 1. / (p->av_gauss[1] * p->av_gauss[1]);
    const double Rv_varinv = 1. / (p->rv_gauss[1] * p->rv_gauss[1]);
    const double log_init = log(p->init_thresh);
    const double mtol = 2.5 * p->ltol;

    float *mc = (float *)malloc(sizeof(float) * (size_t)nmodel * nb * 3);
    double *res = (double *)malloc(sizeof(double) * (size_t)nmodel * nb);
    double *R = (double *)malloc(sizeof(double) * (size_t)nmodel * nb);
    double *dav = (double *)malloc(sizeof(double) * (size_t)nmodel);
    double *drv = (double *)malloc(sizeof(double) * (size_t)nmodel);
    double *logwt = (double *)malloc(sizeof(double) * (size_t)nmodel);
    double *lnlp = (double *)malloc(sizeof(double) * (size_t)nmodel);
    if (!mc || !res || !R || !dav || !drv || !logwt || !lnlp) return -2;

    /* fitting.py:714 gather + :728-733 initial models/residuals */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nmodel; ++i) {
        av[i] = Av_mean;
        rv[i] = Rv_mean;
        for (int j = 0; j < nb; ++j) {
            const float *src = models + ((size_t)i * nfilt + sel[j]) * 3;
            float *dst = mc + ((size_t)i * nb + j) * 3;
            dst[0] = src[0];
            dst[1] = src[1];
            dst[2] = src[2];
            const double rvec = (double)src[1] + rv[i] * (double)src[2];
            R[(size_t)i * nb + j] = rvec;
            res[(size_t)i * nb + j] = g[j] - ((double)src[0] + av[i] * rvec);
        }
    }

    /* ---- magnitude phase, fitting.py:158-264 ---- */
    int K1 = 0;
    for (;;) {
        ++K1;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nmodel; ++i) {
            double *r = res + (size_t)i * nb, *Rv = R + (size_t)i * nb;
            const float *c = mc + (size_t)i * nb * 3;
            double s_den = 0., rp_den = 0., srp_mix = 0.;
            for (int j = 0; j < nb; ++j) {
                const double D = c[3 * j + 2];
                s_den += 1. / W[j];
                rp_den += D * D / W[j];
                srp_mix += D / W[j];
            }
            double a_den = 0., sa_mix = 0., resid_s = 0., resid_a = 0.;
            for (int j = 0; j < nb; ++j) {
                a_den += Rv[j] * Rv[j] / W[j];
                sa_mix += Rv[j] / W[j];
                resid_s += r[j] / W[j];
                resid_a += r[j] * Rv[j] / W[j];
            }
            resid_a += (Av_mean - av[i]) * Av_varinv;
            a_den += Av_varinv;
            const double sa_idet = 1. / (s_den * a_den - sa_mix * sa_mix);
            double da = sa_idet * (s_den * resid_a - sa_mix * resid_s);
            da = da * 1.0;
            if (da < avmin - av[i]) da = avmin - av[i];
            if (da > avmax - av[i]) da = avmax - av[i];
            av[i] = av[i] + da;
            for (int j = 0; j < nb; ++j) r[j] = r[j] - da * Rv[j];
            double resid_r = 0.;
            resid_s = 0.;
            double r_den = rp_den * av[i] * av[i];
            const double sr_mix = srp_mix * av[i];
            for (int j = 0; j < nb; ++j) {
                const double D = c[3 * j + 2];
                resid_s += r[j] / W[j];
                resid_r += r[j] * D / W[j];
            }
            resid_r = resid_r * av[i];
            resid_r += (Rv_mean - rv[i]) * Rv_varinv;
            r_den += Rv_varinv;
            const double sr_idet = 1. / (s_den * r_den - sr_mix * sr_mix);
            double dr_ = sr_idet * (s_den * resid_r - sr_mix * resid_s);
            dr_ = dr_ * 1.0;
            if (dr_ < rvmin - rv[i]) dr_ = rvmin - rv[i];
            if (dr_ > rvmax - rv[i]) dr_ = rvmax - rv[i];
            rv[i] = rv[i] + dr_;
            double c2 = 0.;
            for (int j = 0; j < nb; ++j) {
                const double D = c[3 * j + 2];
                r[j] = r[j] - av[i] * dr_ * D;
                Rv[j] = Rv[j] + dr_ * D;
            }
            for (int j = 0; j < nb; ++j) c2 += r[j] * r[j] / W[j];
            dav[i] = da;
            drv[i] = dr_;
            logwt[i] = -0.5 * c2;
        }
        double max_logwt = -1e300;
        for (int64_t i = 0; i < nmodel; ++i)
            if (logwt[i] > max_logwt) max_logwt = logwt[i];
        double e = -1e300;
        for (int64_t i = 0; i < nmodel; ++i)
            if (logwt[i] > max_logwt + log_init) {
                const double a = fabs(dav[i]), b = fabs(drv[i]);
                if (a > e) e = a;
                if (b > e) e = b;
            }
        if (e < mtol) break;
        if (K1 >= max_iter) {
            free(mc); free(res); free(R); free(dav); free(drv); free(logwt); free(lnlp);
            return -4;
        }
    }
    *k1_out = K1;

    /* ---- MLE at converged (av, rv); cull statistic, fitting.py:267-269,743-759 ---- */
    double *Rf = R;   /* reuse storage: flux-space vectors overwrite mag-space ones */
    double max_lnlp = -INFINITY;
#pragma omp parallel for schedule(static) reduction(max : max_lnlp)
    for (int64_t i = 0; i < nmodel; ++i) {
        double F[64], Df[64], tmp[64];
        sed_mle(mc + (size_t)i * nb * 3, nb, d, V, av[i], rv[i], p, F, Rf + (size_t)i * nb, Df,
                res + (size_t)i * nb, &scale[i], icov + (size_t)i * 9);
        chi2[i] = chi2_of(res + (size_t)i * nb, V, nb, tmp);
        lnl[i] = -0.5 * chi2[i];
        double v = lnl[i];
        if (has_parallax && isfinite(parallax) && isfinite(parallax_err)) {
            const double par = sqrt(scale[i]);
            const double chi2_p = (par - parallax) * (par - parallax) / (parallax_err * parallax_err);
            v = lnl[i] - 0.5 * chi2_p;
        }
        lnlp[i] = v;
        if (v > max_lnlp) max_lnlp = v;
    }
    int64_t nsel = 0;
    int64_t *isel = (int64_t *)malloc(sizeof(int64_t) * (size_t)nmodel);
    for (int64_t i = 0; i < nmodel; ++i)
        if (lnlp[i] > max_lnlp + log(p->init_thresh)) isel[nsel++] = i;
    *nsel_out = nsel;

    /* ---- flux phase on survivors, fitting.py:778-803 ---- */
    double *s_av = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_rv = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_step = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_old = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_new = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_chi2 = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_scale = (double *)malloc(sizeof(double) * (size_t)(nsel + 1));
    double *s_icov = (double *)malloc(sizeof(double) * (size_t)(nsel + 1) * 9);
    double *s_Rf = (double *)malloc(sizeof(double) * (size_t)(nsel + 1) * nb);
    double *s_Df = (double *)malloc(sizeof(double) * (size_t)(nsel + 1) * nb);
    double *s_res = (double *)malloc(sizeof(double) * (size_t)(nsel + 1) * nb);
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < nsel; ++q) {
        const int64_t i = isel[q];
        double F[64], sc, ic[9];
        s_av[q] = av[i];
        s_rv[q] = rv[i];
        s_step[q] = 1.;
        s_old[q] = -1e300;
        /* the reference carries rvecs/drvecs/resid from the mag-phase MLE call */
        sed_mle(mc + (size_t)i * nb * 3, nb, d, V, av[i], rv[i], p, F, s_Rf + (size_t)q * nb,
                s_Df + (size_t)q * nb, s_res + (size_t)q * nb, &sc, ic);
    }
    const double ln_sub = log(p->ltol_subthresh);
    double lerr = 1e300;
    int K2 = 0;
    while (lerr > p->ltol) {
        ++K2;
        double mx = -INFINITY;
#pragma omp parallel for schedule(static) reduction(max : mx)
        for (int64_t q = 0; q < nsel; ++q) {
            const int64_t i = isel[q];
            double *rf = s_Rf + (size_t)q * nb, *df = s_Df + (size_t)q * nb;
            double *rs = s_res + (size_t)q * nb;
            double a_num = 0., a_den = 0., r_num = 0., r_den = 0.;
            for (int j = 0; j < nb; ++j) {
                a_num += rf[j] * rs[j] / V[j];
                a_den += rf[j] * rf[j] / V[j];
            }
            a_num += (Av_mean - s_av[q]) * Av_varinv;
            a_den += Av_varinv;
            double da = a_num / a_den;
            da *= s_step[q];
            for (int j = 0; j < nb; ++j) {
                r_num += df[j] * rs[j] / V[j];
                r_den += df[j] * df[j] / V[j];
            }
            r_num += (Rv_mean - s_rv[q]) * Rv_varinv;
            r_den += Rv_varinv;
            double dr_ = r_num / r_den;
            dr_ *= s_step[q];
            if (da < avmin - s_av[q]) da = avmin - s_av[q];
            if (da > avmax - s_av[q]) da = avmax - s_av[q];
            s_av[q] += da;
            if (dr_ < rvmin - s_rv[q]) dr_ = rvmin - s_rv[q];
            if (dr_ > rvmax - s_rv[q]) dr_ = rvmax - s_rv[q];
            s_rv[q] += dr_;
            double F[64], tmp[64];
            sed_mle(mc + (size_t)i * nb * 3, nb, d, V, s_av[q], s_rv[q], p, F, rf, df, rs,
                    &s_scale[q], s_icov + (size_t)q * 9);
            s_chi2[q] = chi2_of(rs, V, nb, tmp);
            s_new[q] = -0.5 * s_chi2[q];
            if (s_new[q] > mx) mx = s_new[q];
        }
        lerr = -INFINITY;
        for (int64_t q = 0; q < nsel; ++q) {
            if (s_new[q] > mx + ln_sub) {
                const double e = fabs(s_new[q] - s_old[q]);
                if (e > lerr) lerr = e;
            }
        }
        for (int64_t q = 0; q < nsel; ++q) {
            if (s_new[q] < s_old[q]) s_step[q] /= 1.2;
            s_old[q] = s_new[q];
        }
        if (K2 >= max_iter) break;
    }
    *k2_out = K2;

    /* fitting.py:806-815 */
    double lv[64];
    for (int j = 0; j < nb; ++j) lv[j] = log(V[j]);
    const double cst = -0.5 * (nb * log(2. * M_PI) + rowsum(lv, nb));
    for (int64_t q = 0; q < nsel; ++q) {
        const int64_t i = isel[q];
        lnl[i] = s_new[q] + cst;
        chi2[i] = s_chi2[q];
        scale[i] = s_scale[q];
        av[i] = s_av[q];
        rv[i] = s_rv[q];
        memcpy(icov + (size_t)i * 9, s_icov + (size_t)q * 9, sizeof(double) * 9);
    }
    if (p->dim_prior) {
        const double df = (double)(nb - 3);
        const double c0 = -log(pow(2., df / 2.) * tgamma(df / 2.));
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < nmodel; ++i) {
            const double y = chi2[i];
            lnl[i] = (y <= 0.) ? -INFINITY : c0 + (df / 2. - 1.) * log(y) - y / 2. - log(1.);
        }
    }
    free(mc); free(res); free(R); free(dav); free(drv); free(logwt); free(lnlp); free(isel);
    free(s_av); free(s_rv); free(s_step); free(s_old); free(s_new); free(s_chi2); free(s_scale);
    free(s_icov); free(s_Rf); free(s_Df); free(s_res);
    return (K2 >= max_iter && lerr > p->ltol) ? -4 : 0;
}
