// cluster_kernels.hpp -- cluster.isochrone_loglike hot block behind brutus_cluster_lnl
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

namespace {

// ===========================================================================
// cluster.isochrone_loglike hot block (reference cluster.py:336-414)
// ===========================================================================
// For every object o and every isochrone point c (all secondary-mass-fraction
// slices concatenated): chi2 = nansum_b (phot_ob - flux_cb)^2 / err_ob^2 + chi2_p,
// lnl = chi2-logpdf(chi2, n_o) or -(chi2 + lnorm_o)/2, then
// lnl_o = logsumexp_c (lnl + lnw_c).  One lane = one object (its bands in
// VGPRs), isochrone points are workgroup-uniform (LDS broadcasts); the point axis is
// split over blockIdx.y and merged by k_cluster_merge (online logsumexp).
// The sum runs in the linear domain: exp(lnl + lnw) = e^c0 w chi2^(k/2 - 1) e^(-chi2/2),
// and k is an integer, so the power is a square root and up to four multiplications
// instead of a logarithm; only e^(-chi2/2) is tracked against a running maximum.
constexpr int CL_T = 256;      // objects per workgroup: four waves share one staged sub-slice
#ifndef BRUTUS_CL_U
#define BRUTUS_CL_U 4
#endif
constexpr int CL_U = BRUTUS_CL_U;   // points per step of the online sum
#ifndef BRUTUS_CL_OCC
#define BRUTUS_CL_OCC 1
#endif

// A point whose term must not count -- weight 0: dropped by the caller, without a finite band, or
// padding -- is staged with fluxes of DEAD_FLUX: its chi2 overflows and is clamped to CHI2_MAX
// like any non-finite chi2, so that x = -chi2 / 2 can never become the running maximum and
// e^x is 0.  With that every term T_u is finite and >= 0 and the loop needs no "is it finite"
// selects (cluster.py:394 drops the non-finite terms; so does a factor e^-5e17).  CHI2_MAX to the
// largest power the kernel takes (15.5: 33 measurements) stays finite.
constexpr double CL_DEAD_FLUX = 1e150, CL_CHI2_MAX = 1e18;

// MAGS: the points come as the plug-in's magnitude table + the list of kept rows + the two
// factors of the ln-weight (what brutus_cluster_points_grid takes), and the staging does that
// kernel's work -- 10^(-0.4 mag), the any-finite-band test, the weight -- for its own sub-slice:
// no flux table in between, one launch less per group of slices.
struct ClusterMags {
    const int32_t *src;           // (npts) kept table rows
    const double *mags;           // (nrow, nb)
    const double *lnw_eep, *lnw_smf;
    int neep;
};

template <int NB, bool MAGS>
__global__ void __launch_bounds__(CL_T, NB <= 12 ? BRUTUS_CL_OCC : 1)
k_cluster(int nobj, int nb, int npts, const double *__restrict__ pts_flux,
          const double *__restrict__ pts_lnw, ClusterMags mg, const double *__restrict__ phot,
          const double *__restrict__ ivar, const double *__restrict__ chi2_p,
          const double *__restrict__ lnorm, const int32_t *__restrict__ ndim, int dim_prior,
          int pts_per_block, double *__restrict__ part_m, double *__restrict__ part_s) {
    // This workgroup's slice of the point table goes through LDS in sub-slices of
    // SUB points, each staged as [point][NB fluxes (0 beyond nb) | weight | "has a NaN
    // band"]: the loop then reads it with broadcast ds_reads instead of a chain of
    // dependent scalar loads.  A sub-slice is padded to a multiple of CL_U points with
    // weight 0.
    constexpr int STRIDE = NB + 2;
    constexpr int SUB = (24576 / (STRIDE * 8)) / CL_U * CL_U;          // 24 KB per workgroup
    __shared__ double s_pts[SUB * STRIDE];
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    const int p0 = blockIdx.y * pts_per_block;
    const int p1_ = min(npts, p0 + pts_per_block);
    const int o = blockIdx.x * CL_T + threadIdx.x;
    const bool live = o < nobj;
    const int oo = live ? o : 0;
    // (Tried: (D - u flux)^2 with u = 1 / err, D = u phot -- two operations per band instead of
    // three, k_cluster -3 %.  Dropped: an object that sits EXACTLY on a point gets chi2 ~ 1e-30
    // instead of 0, and with one or three measurements the density there is singular / zero:
    // the edge-case test of tests/test_cluster.py differs from the reference's block.)
    double d[NB], iv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        d[b] = b < nb ? phot[(int64_t)oo * nb + b] : 0.;
        iv[b] = b < nb ? ivar[(int64_t)oo * nb + b] : 0.;
    }
    const double cp = chi2_p[oo], ln0 = lnorm[oo];
    const double k = (double)ndim[oo];
    // scipy.stats.chi2.logpdf(chi2, k) = c0 + (k/2 - 1) ln chi2 - chi2/2 (cluster.py:389),
    // or -(chi2 + lnorm)/2 without the dimensionality prior: power n2 / 2 of chi2, n2 = k - 2
    const double c0 = dim_prior ? -(k / 2.) * 0.69314718055994530942 - lgamma(k / 2.) : -0.5 * ln0;
    const int n2 = dim_prior ? ndim[oo] - 2 : 0;
    const bool odd = n2 & 1, neg = n2 < 0;
    const int mp = neg ? 0 : n2 >> 1;                         // whole powers of chi2
    const bool p1 = mp & 1, p2 = mp & 2, p4 = mp & 4, p8 = mp & 8;
    // wave-uniform: does any object of this wave need 1 / sqrt(chi2) (fewer than two
    // measurements) or chi2^8 (eighteen or more)?  Hardly ever; the common path skips both.
    const bool any_neg = __any(neg && odd), any_p8 = __any(p8);
    // (a finite stand-in for "no term yet": e^(M0 - anything) = 0 without a select)
    constexpr double M0 = -1e300;
    double m = M0, ssum = 0.;
    for (int q0 = p0; q0 < p1_; q0 += SUB) {
        const int np = min(SUB, p1_ - q0);
        const int npu = (np + CL_U - 1) / CL_U * CL_U;
        __syncthreads();                                   // previous sub-slice fully consumed
        if constexpr (MAGS) {
            // fluxes first, one (point, band) per thread and turn (an exponential each), ...
            for (int idx = threadIdx.x; idx < npu * NB; idx += CL_T) {
                const int c = idx / NB, b = idx - c * NB;
                double v = 0.;
                if (c < np && b < nb) {
                    // (a band "exists" where its MAGNITUDE is finite, cluster.py:358 -- the
                    // flux of mag = +inf is a perfectly finite 0; it is staged as -0.0, which
                    // the band sums cannot tell from 0 and the any-band test below can)
                    const double mag = mg.mags[(int64_t)mg.src[q0 + c] * nb + b];
                    v = mag == INFINITY ? -0. : exp10(-0.4 * mag);
                }
                s_pts[c * STRIDE + b] = v;
            }
            __syncthreads();
        }
        for (int c = threadIdx.x; c < npu; c += CL_T) {
            const bool real = c < np;
            double w = 0.;                                           // weight (0 for a dropped point)
            if (real) {
                if constexpr (MAGS) {
                    const int r = mg.src[q0 + c];
                    w = exp(mg.lnw_eep[r % mg.neep] + mg.lnw_smf[r / mg.neep]);
                } else {
                    w = exp(pts_lnw[q0 + c]);
                }
            }
            const double *src = MAGS ? s_pts + c * STRIDE : pts_flux + (int64_t)(q0 + (real ? c : 0)) * nb;
            const bool dead = !(w > 0.) || !(w < INFINITY);
            bool hole = false, any = false;
            for (int b = 0; b < NB; ++b) {
                const double v = b < nb ? src[b] : 0.;
                hole = hole || (v != v);
                // (fluxes from the caller's table: any non-NaN band; staged from magnitudes:
                // any band whose magnitude was finite -- not the -0.0 of +inf, not the inf of -inf)
                if constexpr (MAGS) any = any || (b < nb && fabs(v) < INFINITY && !(v == 0. && signbit(v)));
                else any = any || (b < nb && v == v);
                s_pts[c * STRIDE + b] = dead ? CL_DEAD_FLUX : v;
            }
            // (a point all of whose bands are NaN would have chi2 = chi2_p; the caller gives it
            // weight 0, cluster_points does, and so does this)
            if (!any && !dead)
                for (int b = 0; b < NB; ++b) s_pts[c * STRIDE + b] = CL_DEAD_FLUX;
            s_pts[c * STRIDE + NB] = dead || !any ? 0. : w;
            s_pts[c * STRIDE + NB + 1] = hole && any && !dead ? 1. : 0.;
        }
        __syncthreads();
        for (int c = 0; c < npu; c += CL_U) {
            // CL_U points at a time: their terms T_u e^{x_u}, x_u = -chi2_u / 2, join the running
            // sum against ONE new maximum -- one exponential per point, of a non-positive
            // argument, plus one per step for the old sum, and no select on which of the two
            // is the larger
            double x[CL_U], T[CL_U];
            double mn = m;
#pragma unroll
            for (int u = 0; u < CL_U; ++u) {
                const double *f = s_pts + (c + u) * STRIDE;
                double fb[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b) fb[b] = f[b];
                double chi2 = 0.;
                // nansum (cluster.py:381): d and iv are finite (masked bands carry iv = 0),
                // so a term is NaN exactly when the point lacks band b -- rare, and the
                // same for every lane
                if (f[NB + 1] == 0.) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const double t = d[b] - fb[b];
                        chi2 += t * t * iv[b];
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const double t = d[b] - fb[b];
                        const double term = t * t * iv[b];
                        chi2 += term == term ? term : 0.;
                    }
                }
                chi2 = fmin(chi2 + cp, CL_CHI2_MAX);        // (a NaN -- inf * 0 of a dead point -- too)
                // w chi2^(n2 / 2)
                double rt;
                if (any_neg) {
                    double sq, rsq;
                    fast_sqrt_rsqrt(chi2, sq, rsq);
                    rt = neg ? (chi2 > 0. ? rsq : 0.) : sq;     // (chi2^-1/2 at 0: +inf, dropped)
                } else {
                    rt = fast_sqrt(chi2);
                }
                double Tu = f[NB] * (odd ? rt : 1.);
                const double c2 = chi2 * chi2, c4 = c2 * c2;
                Tu = p1 ? Tu * chi2 : Tu;
                Tu = p2 ? Tu * c2 : Tu;
                Tu = p4 ? Tu * c4 : Tu;
                if (any_p8) Tu = p8 ? Tu * (c4 * c4) : Tu;
                T[u] = Tu;
                x[u] = -0.5 * chi2;
                // (a term that vanishes -- T = 0 with chi2 = 0 and a positive power -- may set
                // the maximum: it only rescales the others, by e^-(their chi2 / 2))
                mn = fmax(mn, x[u]);
            }
            const double scale = fast_exp_fin(m - mn, s_tbl);
            double add = 0.;
#pragma unroll
            for (int u = 0; u < CL_U; ++u) add = fma(T[u], fast_exp_fin(x[u] - mn, s_tbl), add);
            ssum = fma(ssum, scale, add);
            m = mn;
        }
    }
    m = ssum > 0. ? m + c0 : -INFINITY;          // (no point with a positive term)
    if (live) {
        part_m[(int64_t)blockIdx.y * nobj + o] = m;
        part_s[(int64_t)blockIdx.y * nobj + o] = ssum;
    }
}

// `neep` > 0: the weight of table row r is lnw_in[r % neep] + lnw_smf[r / neep] -- one
// initial-mass grid for all slices (ln d mini per EEP, ln d smf per slice) --, else lnw_in[r].
__global__ void k_cluster_points(int64_t npts, int nb, const int32_t *__restrict__ src,
                                 const double *__restrict__ mags,
                                 const double *__restrict__ lnw_in, int neep,
                                 const double *__restrict__ lnw_smf, double *__restrict__ flux,
                                 double *__restrict__ lnw) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= npts) return;
    const int64_t r = src ? (int64_t)src[c] : c;
    bool any = false;
    for (int b = 0; b < nb; ++b) {
        const double m = mags[r * nb + b];
        any = any || isfinite(m);
        flux[c * nb + b] = exp10(-0.4 * m);
    }
    const double w = neep > 0 ? lnw_in[r % neep] + lnw_smf[r / neep] : lnw_in[r];
    lnw[c] = any ? w : -INFINITY;
}

// Merge of the per-chunk (max, sum) pairs of every object.  Lane (o, j), j < CM_J, folds the
// chunks j, j + CM_J, ... of object o (consecutive lanes = consecutive objects: coalesced),
// the CM_J partials of an object meet in LDS.  (One thread per object looping over all 256
// chunks was 20 workgroups of dependent loads: 0.19 ms, two thirds of k_cluster itself.)
constexpr int CM_O = 32, CM_J = 8;      // objects x chunk lanes per 256-thread workgroup
__global__ void __launch_bounds__(CM_O * CM_J)
k_cluster_merge(int nobj, int nchunk, const double *__restrict__ part_m,
                const double *__restrict__ part_s, double *__restrict__ out) {
    __shared__ double s_m[CM_J][CM_O], s_s[CM_J][CM_O];
    const int ol = threadIdx.x % CM_O, j = threadIdx.x / CM_O;
    const int o = blockIdx.x * CM_O + ol;
    double m = -INFINITY, ssum = 0.;
    if (o < nobj) {
        for (int c = j; c < nchunk; c += CM_J) {
            const double x = part_m[(int64_t)c * nobj + o], sx = part_s[(int64_t)c * nobj + o];
            if (x > -INFINITY) {        // online merge of (x, sx) into (m, ssum)
                if (x > m) {
                    ssum = ssum * exp(m - x) + sx;      // (m = -inf: ssum is 0, exp gives 0)
                    m = x;
                } else {
                    ssum += sx * exp(x - m);
                }
            }
        }
    }
    s_m[j][ol] = m;
    s_s[j][ol] = ssum;
    __syncthreads();
    if (j == 0 && o < nobj) {
        double mm = -INFINITY;
#pragma unroll
        for (int q = 0; q < CM_J; ++q) mm = s_m[q][ol] > mm ? s_m[q][ol] : mm;
        double tot = 0.;
#pragma unroll
        for (int q = 0; q < CM_J; ++q)
            if (s_m[q][ol] > -INFINITY) tot += s_s[q][ol] * exp(s_m[q][ol] - mm);
        out[o] = mm > -INFINITY ? mm + log(tot) : -INFINITY;
    }
}

// Outlier mixture and total (cluster.py:410-414): lnl_mix = logaddexp(lnl + ln_fin,
// lnl_outlier + ln_fout) per object and their sum, one workgroup, a fixed summation order
// (thread t takes objects t, t + 1024, ...; a binary tree over the threads): the same bits
// on every run.
constexpr int CX_T = 1024;
__global__ void __launch_bounds__(CX_T)
k_cluster_mix(int nobj, const double *__restrict__ lnl, const double *__restrict__ lnl_out,
              double ln_fin, double ln_fout, double *__restrict__ mix, double *__restrict__ tot) {
    __shared__ double s_sum[CX_T];
    double acc = 0.;
    for (int o = threadIdx.x; o < nobj; o += CX_T) {
        const double a = lnl[o] + ln_fin, b = lnl_out[o] + ln_fout;
        const double d = a - b;
        double v;
        if (d > 0.) v = a + log1p(exp(-d));
        else if (d <= 0.) v = b + log1p(exp(d));
        else v = a + b;                     // NaN, or infinities of one sign (numpy: x1 + x2)
        mix[o] = v;
        acc += v;
    }
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (int h = CX_T / 2; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) s_sum[threadIdx.x] += s_sum[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) tot[0] = s_sum[0];
}

}  // namespace
