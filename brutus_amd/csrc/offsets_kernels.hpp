// offsets_kernels.hpp -- utils.photometric_offsets on the device
// Part of the single translation unit brutus_kernels.hip; everything lives in that
// unit's anonymous namespace.
//
// Reference utils.py:1218-1400.  The fit of every object left Nsamps resampled models
// (grid index, Av, Rv, distance).  Per band b the reference
//   1. rebuilds the model flux of every draw (get_seds, utils.py:1089-1159),
//   2. re-weights the draws of an object by the likelihood of the OTHER bands
//      (phot_loglike, utils.py:1162-1215) when b took part in the fit,
//   3. bootstraps: Nmc rounds of n objects drawn with replacement, one draw each by the
//      weights, the median of model / data over the round,
//   4. reports median and standard deviation of the Nmc medians.
// The host keeps the random stream (2 n uniforms per round, numpy's order) and step 4;
// steps 1-3 run here:
//   k_po_flux    lane = (object, draw): flux of all bands, leave-one-out ln-likelihoods
//   k_po_cdf     workgroup = (object, band): normalised cumulative weights
//   k_po_boot    lane = (round, slot): two binary searches, one ratio
//   segmented radix sort of the rounds (rocPRIM) + k_po_median
#pragma once

namespace {

constexpr int PO_T = 256;

// flux (nfilt, nobj, nsamps) and lnl (nfilt, nobj, nsamps); `use` (nfilt, nobj) marks
// the (band, object) pairs of step 2 (band observed, object usable, enough other bands).
__global__ void __launch_bounds__(PO_T)
k_po_flux(int nobj, int nsamps, int nfilt, int64_t nmodel, const float *__restrict__ models,
          const int64_t *__restrict__ idxs, const double *__restrict__ reds,
          const double *__restrict__ dreds, const double *__restrict__ dists,
          const double *__restrict__ phot, const double *__restrict__ err,
          const uint8_t *__restrict__ mask, const double *__restrict__ old_off,
          const uint8_t *__restrict__ use, const uint8_t *__restrict__ mask_fit, int dim_prior,
          double *__restrict__ flux, double *__restrict__ lnl) {
    const int64_t t = (int64_t)blockIdx.x * PO_T + threadIdx.x;
    if (t >= (int64_t)nobj * nsamps) return;
    const int o = (int)(t / nsamps);
    int64_t i = idxs[t];
    if (i < 0) i += nmodel;                        // numpy's wrap (the -99 sentinel rows)
    i = i < 0 ? 0 : (i >= nmodel ? nmodel - 1 : i);
    const float *row = models + i * (3 * (int64_t)nfilt);
    const double av = reds[t], rv = dreds[t], d = dists[t];
    const double d2 = __dmul_rn(d, d);
    double term[NBMAX];
    bool on[NBMAX];
    for (int j = 0; j < nfilt; ++j) {
        // utils.py:286-347: rvec = R0 + Rv dR, sed = mag + Av rvec (no contraction: the
        // reference rounds every product)
        const double rvec = __dadd_rn((double)row[3 * j + 1], __dmul_rn(rv, (double)row[3 * j + 2]));
        const double sed = __dadd_rn((double)row[3 * j], __dmul_rn(av, rvec));
        const double f = exp10(__dmul_rn(-0.4, sed)) / d2;
        flux[((int64_t)j * nobj + o) * nsamps + (t - (int64_t)o * nsamps)] = f;
        on[j] = mask[(int64_t)o * nfilt + j] != 0;
        const double off = old_off[j];
        const double r = __dmul_rn(phot[(int64_t)o * nfilt + j], off) - f;
        const double e = __dmul_rn(err[(int64_t)o * nfilt + j], off);
        term[j] = on[j] ? __dmul_rn(r, r) / __dmul_rn(e, e) : 0.;
    }
    for (int b = 0; b < nfilt; ++b) {
        if (!mask_fit[b] || !use[(int64_t)b * nobj + o]) continue;
        double chi2 = 0.;
        int ndim = 0;
        for (int j = 0; j < nfilt; ++j)
            if (j != b && on[j]) {
                chi2 = __dadd_rn(chi2, term[j]);
                ++ndim;
            }
        // utils.py:1203-1213 without the terms that are the same for every draw of the
        // object (they cancel in the normalisation of the weights)
        double v = -0.5 * chi2;
        if (dim_prior) {
            const double a1 = 0.5 * (ndim - 3) - 1.;
            if (a1 != 0.) v += a1 * log(chi2);     // xlogy
        }
        lnl[((int64_t)b * nobj + o) * nsamps + (t - (int64_t)o * nsamps)] = v;
    }
}

// In place: lnl (band, object, :) -> cumulative weights, last entry exactly 1
// (utils.py:1358-1370: exp(lnl - logsumexp) * weights, normalised; `choice` then
// searches cumsum(p) / cumsum(p)[-1]).
__global__ void __launch_bounds__(PO_T)
k_po_cdf(int nobj, int nsamps, const double *__restrict__ weights,
         const uint8_t *__restrict__ use, const uint8_t *__restrict__ mask_fit,
         double *__restrict__ cdf) {
    __shared__ double s_slot[PO_T / 64 + 1];
    __shared__ double s_red[PO_T / 64];
    __shared__ double s_bc;
    const int o = blockIdx.x, b = blockIdx.y;
    if (!use[(int64_t)b * nobj + o]) return;
    double *row = cdf + ((int64_t)b * nobj + o) * nsamps;
    const double *w = weights + (int64_t)o * nsamps;
    const bool fit = mask_fit[b] != 0;
    double mx = 0.;
    if (fit) {
        double m = -INFINITY;
        for (int k = threadIdx.x; k < nsamps; k += PO_T) {
            const double v = row[k];
            m = v > m ? v : m;
        }
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int q = 1; q < PO_T / 64; ++q) m = s_red[q] > m ? s_red[q] : m;
            s_bc = m;
        }
        __syncthreads();
        mx = s_bc;
        if (!(mx > -INFINITY)) mx = 0.;
    }
    double carry = 0.;
    for (int k0 = 0; k0 < nsamps; k0 += PO_T) {
        const int k = k0 + threadIdx.x;
        double v = 0.;
        if (k < nsamps) v = fit ? exp(row[k] - mx) * w[k] : w[k];
        double tot;
        const double inc = block_inclusive_sum<double, PO_T>(v, s_slot, tot);
        if (k < nsamps) row[k] = carry + inc;
        carry += tot;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nsamps; k += PO_T) row[k] = row[k] / carry;
}

// count of entries <= u in a non-decreasing array (numpy searchsorted(side='right'))
__device__ __forceinline__ int po_upper(const double *__restrict__ a, int n, double u) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= u) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// One bootstrap draw: u (nmc, 2, n) are the uniforms of `choice(n, n, p=wt_obj)` and of
// the n `choice(Nsamps, p=wt[i])` calls that follow it in every round (utils.py:1381-1385).
__global__ void __launch_bounds__(PO_T)
k_po_boot(int band, int nobj, int nsamps, int nfilt, int n, int nmc,
          const int32_t *__restrict__ subset, const double *__restrict__ cdf_obj,
          const double *__restrict__ u, const double *__restrict__ flux,
          const double *__restrict__ cdf, const double *__restrict__ phot,
          double *__restrict__ vals) {
    const int64_t t = (int64_t)blockIdx.x * PO_T + threadIdx.x;
    if (t >= (int64_t)nmc * n) return;
    const int64_t j = t / n, k = t - j * n;
    int r = po_upper(cdf_obj, n, u[(2 * j) * n + k]);
    r = r < n ? r : n - 1;
    const int o = subset[r];
    const int64_t base = ((int64_t)band * nobj + o) * nsamps;
    int m = po_upper(cdf + base, nsamps, u[(2 * j + 1) * n + k]);
    m = m < nsamps ? m : nsamps - 1;
    vals[t] = flux[base + m] / phot[(int64_t)o * nfilt + band];
}

// np.median of every sorted round
__global__ void k_po_median(int n, int nmc, const double *__restrict__ sorted,
                            double *__restrict__ meds) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nmc) return;
    const double *s = sorted + (int64_t)j * n;
    // (a NaN sorts to one of the two ends; numpy's median of such a round is NaN)
    const bool bad = s[0] != s[0] || s[n - 1] != s[n - 1];
    meds[j] = bad ? NAN : ((n & 1) ? s[n / 2] : 0.5 * (s[n / 2 - 1] + s[n / 2]));
}

__global__ void k_po_segments(int n, int nmc, int32_t *__restrict__ seg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j <= nmc) seg[j] = j * n;
}

}  // namespace
