// grid_kernels.hpp -- generic per-phase kernels behind brutus_loglike_batch (full output planes)
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

namespace {

// ---- magnitude phase (fitting.py:158-243) ---------------------------------
template <int NB>
struct MagState {
    double res[NB], R[NB];
    double av, rv, P, Q;
    double dav, drv, logwt;
};

template <int NB>
__device__ __forceinline__ void mag_init(const Coef<NB> &c, const StarPrep &sp,
                                         const DevParams &p, double av0, double rv0,
                                         MagState<NB> &st) {
    st.av = av0;         // fitting.py:697-703: av_init / rv_init, by default the prior means
    st.rv = rv0;
    double P = 0., Q = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double D = (double)c.dr[j];
        const double R = (double)c.r0[j] + st.rv * D;      // utils.py:338
        const double M = (double)c.m[j] + st.av * R;       // utils.py:339
        st.R[j] = R;
        st.res[j] = sp.g[j] - M;                            // fitting.py:733
        const double Dw = D * sp.iW[j];
        P += D * Dw;                                        // fitting.py:163
        Q += Dw;                                            // fitting.py:164
    }
    st.P = P;
    st.Q = Q;
}

template <int NB>
__device__ __forceinline__ void mag_sweep(const Coef<NB> &c, const StarPrep &sp,
                                          const DevParams &p, MagState<NB> &st) {
    const double S = sp.S;
    // Av solve, fitting.py:176-204 (stepsize == 1 throughout this phase)
    double a_den = 0., sa = 0., rs = 0., ra = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double w = sp.iW[j];
        const double Rw = st.R[j] * w;
        a_den += st.R[j] * Rw;
        sa += Rw;
        rs += st.res[j] * w;
        ra += st.res[j] * Rw;
    }
    ra += (p.av_mean - st.av) * p.av_ivar;
    a_den += p.av_ivar;
    double dav = (S * ra - sa * rs) / (S * a_den - sa * sa);
    if (dav < p.avmin - st.av) dav = p.avmin - st.av;
    if (dav > p.avmax - st.av) dav = p.avmax - st.av;
    st.av = st.av + dav;
#pragma unroll
    for (int j = 0; j < NB; ++j) st.res[j] -= dav * st.R[j];

    // Rv solve, fitting.py:207-237
    double r_den = st.P * st.av * st.av;
    const double sr = st.Q * st.av;
    rs = 0.;
    double rr = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double w = sp.iW[j];
        rs += st.res[j] * w;
        rr += st.res[j] * ((double)c.dr[j] * w);
    }
    rr = rr * st.av;
    rr += (p.rv_mean - st.rv) * p.rv_ivar;
    r_den += p.rv_ivar;
    double drv = (S * rr - sr * rs) / (S * r_den - sr * sr);
    if (drv < p.rvmin - st.rv) drv = p.rvmin - st.rv;
    if (drv > p.rvmax - st.rv) drv = p.rvmax - st.rv;
    st.rv = st.rv + drv;
    const double t = st.av * drv;
    double chi2 = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double D = (double)c.dr[j];
        st.res[j] -= t * D;
        st.R[j] += drv * D;
        chi2 += st.res[j] * st.res[j] * sp.iW[j];          // fitting.py:240-242
    }
    st.dav = dav;
    st.drv = drv;
    st.logwt = -0.5 * chi2;
}

// ---- MLE quantities (fitting.py:502-576) ----------------------------------
struct Mle {
    double scale, chi2;
    double i00, i01, i02, i11, i12, i22;
    double a_num, r_num, a_ss, r_ss;   // sums the flux step needs (fitting.py:387-398)
};

template <int NB>
__device__ __forceinline__ void mle_eval(const Coef<NB> &c, const double (&F0)[NB],
                                         const StarPrep &sp, const DevParams &p,
                                         double av, double rv, Mle &o) {
    const double fac = -0.92103403719761827361;  // -0.4 ln 10 (utils.py:328)
    double F[NB];
    double s_num = 0., s_den = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double R = (double)c.r0[j] + rv * (double)c.dr[j];
        const double sed = (double)c.m[j] + av * R;
        const double f = exp10(-0.4 * sed);                  // utils.py:343
        F[j] = f;
        const double fw = f * sp.iV[j];
        s_num += sp.d[j] * fw;                                // fitting.py:514
        s_den += f * fw;                                      // fitting.py:515
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;                                // fitting.py:517-518
    double sr_mix = 0., sa_mix = 0., ar_mix = 0., a_den = 0., r_den = 0.;
    double a_num = 0., r_num = 0., chi2 = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double iv = sp.iV[j];
        const double D0 = (double)c.dr[j];
        const double R0 = (double)c.r0[j] + rv * D0;
        const double ff = fac * F[j];
        double Rf = R0 * ff;                                  // utils.py:344
        double Df = D0 * ff;                                  // utils.py:345
        double red = F[j] - F0[j];                            // fitting.py:529-530
        const double Fs = F[j] * s;                           // fitting.py:533
        const double res = sp.d[j] - Fs;                      // fitting.py:536
        const double t = (Fs - res) * iv;
        sr_mix += Df * t;                                     // fitting.py:539
        sa_mix += Rf * t;                                     // fitting.py:541
        Rf *= s;
        Df *= s;
        red *= s;
        ar_mix += Df * ((red - res) * iv);                    // fitting.py:550
        a_den += Rf * Rf * iv;                                // fitting.py:552
        r_den += Df * Df * iv;                                // fitting.py:553
        const double rw = res * iv;
        a_num += Rf * rw;                                     // fitting.py:388
        r_num += Df * rw;                                     // fitting.py:397
        chi2 += res * rw;                                     // fitting.py:745,792
    }
    o.a_ss = a_den;
    o.r_ss = r_den;
    o.a_num = a_num;
    o.r_num = r_num;
    a_den += p.av_ivar;                                       // fitting.py:556-561
    r_den += p.rv_ivar;
    a_den += p.a_reg;
    r_den += p.r_reg;
    o.scale = s;
    o.chi2 = chi2;
    o.i00 = s_den;
    o.i01 = sa_mix;
    o.i02 = sr_mix;
    o.i11 = a_den;
    o.i12 = ar_mix;
    o.i22 = r_den;
}

template <int NB>
__device__ __forceinline__ void compute_F0(const Coef<NB> &c, double (&F0)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = exp10(-0.4 * (double)c.m[j]);   // fitting.py:529
}

__device__ __forceinline__ void store_mle(const Planes &pl, int64_t o, const Mle &m) {
    pl.chi2[o] = m.chi2;
    pl.scale[o] = m.scale;
    pl.icov[0][o] = m.i00;
    pl.icov[1][o] = m.i01;
    pl.icov[2][o] = m.i02;
    pl.icov[3][o] = m.i11;
    pl.icov[4][o] = m.i12;
    pl.icov[5][o] = m.i22;
}

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// Per-star preparation (fitting.py:706-725).  One thread per star.
__global__ void __launch_bounds__(64)
k_prep(int nstar, int nfilt, const double *__restrict__ flux,
       const double *__restrict__ err, const uint8_t *__restrict__ mask,
       const double *__restrict__ par, const double *__restrict__ perr,
       int has_parallax, StarPrep *__restrict__ out, int32_t *__restrict__ ndim_out) {
    // One wave per star, lane = band: the logarithms and divisions of the bands run side by
    // side (one thread per star looping over its bands was 23 us of dependent float64 latency
    // per call); the three band sums are then added up by every lane in the reference's order,
    // j = 0, 1, ..., so the result does not depend on how the work was spread.
    static_assert(NBMAX <= 64, "one lane per band");
    __shared__ double s_w[NBMAX], s_lnv[NBMAX], s_d2[NBMAX];
    __shared__ int s_ok[NBMAX];
    const int s = blockIdx.x, j = threadIdx.x;
    if (s >= nstar) return;
    StarPrep &sp = out[s];
    const double kmag = 2.5 / log(10.);
    if (j < NBMAX) {
        double d = 0., iv = 0., g = 0., iw = 0., lnv = 0., d2 = 0.;
        bool ok = false;
        if (j < nfilt) {
            const double f = flux[(int64_t)s * nfilt + j];
            const double e = err[(int64_t)s * nfilt + j];
            ok = mask[(int64_t)s * nfilt + j] && isfinite(f) && isfinite(e) && e > 0.;
            if (ok) {
                const double v = e * e;
                d = f;
                iv = 1. / v;
                lnv = log(v);
                g = -2.5 * log10(f);
                double W = kmag * kmag * v / (f * f);
                if (!isfinite(g)) {                           // fitting.py:724-725
                    g = 0.;
                    W = 1e50;
                }
                iw = 1. / W;
                d2 = f * f * iv;
            }
        }
        sp.d[j] = d;
        sp.iV[j] = iv;
        sp.g[j] = g;
        sp.iW[j] = iw;
        s_w[j] = iw;
        s_lnv[j] = lnv;
        s_d2[j] = d2;
        s_ok[j] = ok ? 1 : 0;
    }
    __syncthreads();
    if (j != 0) return;
    int ndim = 0;
    double S = 0., sumlnv = 0., D2 = 0.;
    for (int k = 0; k < NBMAX; ++k) {
        if (!s_ok[k]) continue;
        ++ndim;
        sumlnv += s_lnv[k];
        S += s_w[k];
        D2 += s_d2[k];
    }
    sp.S = S;
    sp.D2 = D2;
    sp.ndim = ndim;
    sp.lnl_const = -0.5 * (ndim * log(2. * M_PI) + sumlnv);
    const double df = (double)(ndim - 3);
    sp.c0 = -log(exp2(df / 2.) * tgamma(df / 2.));
    sp.c1 = df / 2. - 1.;
    double p = nan(""), pe = nan("");
    if (has_parallax) {
        p = par[s];
        pe = perr[s];
    }
    const bool fin = isfinite(p) && isfinite(pe);
    sp.has_par = fin ? 1 : 0;
    sp.par = fin ? p : 0.;
    sp.par_ivar = fin ? 1. / (pe * pe) : 0.;
    sp.sp_on = (fin && p / pe > 4.) ? 1 : 0;                  // pdf.py:209
    const double pm = p > 0. ? p : 0.;                        // pdf.py:252-255
    sp.sp_mean = sp.sp_on ? pm * pm + pe * pe : 0.;
    sp.sp_var = sp.sp_on ? 2. * pe * pe * pe * pe + 4. * pm * pm * pe * pe : 0.;
    sp.pad_ = 0;
    ndim_out[s] = ndim;
}

__global__ void k_relayout(const float *__restrict__ aos, int64_t nmodel, int nfilt, int nb,
                           int64_t nmodel_pad, float *__restrict__ blob) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmodel_pad) return;
    float *rows = blob + (int64_t)3 * nb * nmodel_pad;
    double *f0_soa = reinterpret_cast<double *>(blob + (int64_t)6 * nb * nmodel_pad);
    for (int j = 0; j < nb; ++j) {
        float m = 0.f;
        for (int k = 0; k < 3; ++k) {
            float v = 0.f;
            if (i < nmodel && j < nfilt) v = aos[(i * nfilt + j) * 3 + k];
            if (k == 0) m = v;
            blob[(int64_t)(3 * j + k) * nmodel_pad + i] = v;
            rows[(i * nb + j) * 3 + k] = v;
        }
        const double f0 = fast_exp10(-0.4 * (double)m);
        f0_soa[(int64_t)j * nmodel_pad + i] = f0;
    }
}

// Phase 1: run `kmax` magnitude sweeps for every (star, model); emit per
// (tile, star) the two maxima per sweep that decide convergence.
//   part[((tile * nstar) + s) * 2*kmax + 2k]   = max logwt            (L_k)
//   part[... + 2k + 1] = max{logwt : step >= tol}                      (T_k)
template <int NB>
__global__ void __launch_bounds__(TILE)
k_mag_stats(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
            const StarPrep *__restrict__ stars, DevParams p, int kmax,
            double *__restrict__ part, const double *__restrict__ av_init,
            const double *__restrict__ rv_init) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    Coef<NB> c;
    load_coef<NB>(grid, nmodel_pad, i, c);
    // per-model starting point (the `av_init` / `rv_init` arrays of fitting.py:697-703)
    const double av0 = av_init ? av_init[live ? i : nmodel - 1] : p.av_mean;
    const double rv0 = rv_init ? rv_init[live ? i : nmodel - 1] : p.rv_mean;
    const int s0 = blockIdx.y * STAR_GROUP;
    const int s1 = min(nstar, s0 + STAR_GROUP);
    const double ninf = -INFINITY;
    for (int s = s0; s < s1; ++s) {
        const StarPrep &sp = stars[s];
        MagState<NB> st;
        mag_init<NB>(c, sp, p, av0, rv0, st);
        double *out = part + ((int64_t)blockIdx.x * nstar + s) * (2 * kmax);
        for (int k = 0; k < kmax; ++k) {
            mag_sweep<NB>(c, sp, p, st);
            const double lw = (live && st.logwt == st.logwt) ? st.logwt : ninf;
            const bool big = (fabs(st.dav) >= p.mtol) || (fabs(st.drv) >= p.mtol);
            block_max_store(lw, slot, out + 2 * k);
            block_max_store(big ? lw : ninf, slot, out + 2 * k + 1);
        }
    }
}

// Per-star reduction over tiles + decision.  One workgroup per star.
//   mode 0: K1 from (L_k, T_k), k < kmax         -> iters[s] (0 = not converged)
//   mode 1: single maximum                        -> vmax[s]
//   mode 2: flux convergence from (L, T)          -> done[s]
__global__ void k_reduce_decide(int mode, int ntile, int nstar, int nval,
                                const double *__restrict__ part, double thresh,
                                double *__restrict__ vmax, int32_t *__restrict__ iters,
                                int32_t *__restrict__ n_unconv) {
    __shared__ double sm[KCAP * 2][4];
    const int s = blockIdx.x;
    double v[KCAP * 2];
    for (int q = 0; q < nval; ++q) v[q] = -INFINITY;
    for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
        const double *pp = part + ((int64_t)t * nstar + s) * nval;
        for (int q = 0; q < nval; ++q) v[q] = pp[q] > v[q] ? pp[q] : v[q];
    }
    for (int q = 0; q < nval; ++q) {
        const double m = wave_max(v[q]);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int q = 0; q < nval; ++q) {
        double m = sm[q][0];
        for (int w = 1; w < 4; ++w) m = sm[q][w] > m ? sm[q][w] : m;
        v[q] = m;
    }
    if (mode == 0) {
        int K1 = 0;
        for (int k = 0; k < nval / 2; ++k) {
            const double L = v[2 * k] > -BIG ? v[2 * k] : -BIG;    // fitting.py:246-249
            const double T = v[2 * k + 1];
            const bool unconv = T > L + thresh;                     // fitting.py:252-263
            if (!unconv) {
                K1 = k + 1;
                break;
            }
        }
        iters[s] = K1;
        if (K1 == 0) atomicAdd(n_unconv, 1);
    } else if (mode == 1) {
        vmax[s] = v[0];
    } else {
        // fitting.py:798-799: lerr = max |dlnl| over lnl_new > max + ln(subthresh)
        const bool unconv = v[1] > v[0] + thresh;
        if (iters[s] >= 0) {          // still active; iters[s] = iterations run so far
            if (unconv) {
                iters[s] += 1;        // the next launch runs one more
                atomicAdd(n_unconv, 1);
            } else {
                iters[s] = -iters[s] - 1;   // done: encode final count as -(K2)-1
            }
        }
    }
}

// Phase 2: K1[s] sweeps + MLE for every (star, model); write the full-grid
// mag-phase results (these are final for every model the cull drops,
// fitting.py:809-810) and the cull statistic lnl_p (fitting.py:743-756).
template <int NB>
__global__ void __launch_bounds__(TILE)
k_mag_mle(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
          const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
          Planes pl, double *__restrict__ part, const double *__restrict__ av_init,
          const double *__restrict__ rv_init) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    Coef<NB> c;
    load_coef<NB>(grid, nmodel_pad, i, c);
    const double av0 = av_init ? av_init[live ? i : nmodel - 1] : p.av_mean;
    const double rv0 = rv_init ? rv_init[live ? i : nmodel - 1] : p.rv_mean;
    double F0[NB];
    compute_F0<NB>(c, F0);
    const int s0 = blockIdx.y * STAR_GROUP;
    const int s1 = min(nstar, s0 + STAR_GROUP);
    for (int s = s0; s < s1; ++s) {
        const StarPrep &sp = stars[s];
        MagState<NB> st;
        mag_init<NB>(c, sp, p, av0, rv0, st);
        const int K = k1[s];
        for (int k = 0; k < K; ++k) mag_sweep<NB>(c, sp, p, st);
        Mle m;
        mle_eval<NB>(c, F0, sp, p, st.av, st.rv, m);
        const double lnl = -0.5 * m.chi2;
        double lnlp = lnl;
        if (sp.has_par) {
            const double dp = sqrt(m.scale) - sp.par;
            lnlp = lnl - 0.5 * (dp * dp * sp.par_ivar);
        }
        if (live) {
            const int64_t o = (int64_t)s * pl.nmodel + i;
            store_mle(pl, o, m);
            pl.av[o] = st.av;
            pl.rv[o] = st.rv;
            pl.lnl[o] = lnl;
            pl.lnlp[o] = lnlp;
            pl.step[o] = 1.0;
        }
        block_max_store((live && lnlp == lnlp) ? lnlp : -INFINITY, slot,
                        part + ((int64_t)blockIdx.x * nstar + s));
    }
}

// Phase 3: flux-space iterations on the survivors of the cull
// (fitting.py:758-803).  `first` launches run two iterations from lnl_old =
// -1e300 (the reference always needs >= 2); continuation launches run one.
// Per (tile, star) emits L = max lnl_new and T = max{lnl_new : |dlnl| > ltol}
// of the LAST iteration of the launch.
template <int NB>
__global__ void __launch_bounds__(TILE)
k_flux(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
       const StarPrep *__restrict__ stars, DevParams p, const double *__restrict__ lnlp_max,
       const int32_t *__restrict__ k2state, int first, Planes pl, double *__restrict__ part) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    Coef<NB> c;
    load_coef<NB>(grid, nmodel_pad, i, c);
    double F0[NB];
    compute_F0<NB>(c, F0);
    const int s0 = blockIdx.y * STAR_GROUP;
    const int s1 = min(nstar, s0 + STAR_GROUP);
    const int niter = first ? 2 : 1;
    for (int s = s0; s < s1; ++s) {
        if (k2state[s] < 0) continue;   // star already converged (wave-uniform)
        const StarPrep &sp = stars[s];
        const int64_t o = (int64_t)s * pl.nmodel + i;
        bool surv = false;
        if (live) surv = pl.lnlp[o] > lnlp_max[s] + p.ln_init;    // fitting.py:758
        double L = -INFINITY, T = -INFINITY;
        if (__any(surv)) {
            if (surv) {
                double av = pl.av[o], rv = pl.rv[o];
                double step = first ? 1.0 : pl.step[o];
                double lnl_old = first ? -BIG : -0.5 * pl.chi2[o];
                Mle m;
                mle_eval<NB>(c, F0, sp, p, av, rv, m);   // rebuild the sums at (av, rv)
                double lnl_new = lnl_old, dl = 0.;
                for (int it = 0; it < niter; ++it) {
                    // fitting.py:385-420
                    double dav = (m.a_num + (p.av_mean - av) * p.av_ivar) /
                                 (m.a_ss + p.av_ivar) * step;
                    double drv = (m.r_num + (p.rv_mean - rv) * p.rv_ivar) /
                                 (m.r_ss + p.rv_ivar) * step;
                    if (dav < p.avmin - av) dav = p.avmin - av;
                    if (dav > p.avmax - av) dav = p.avmax - av;
                    av += dav;
                    if (drv < p.rvmin - rv) drv = p.rvmin - rv;
                    if (drv > p.rvmax - rv) drv = p.rvmax - rv;
                    rv += drv;
                    mle_eval<NB>(c, F0, sp, p, av, rv, m);
                    lnl_new = -0.5 * m.chi2;                        // fitting.py:795
                    dl = fabs(lnl_new - lnl_old);
                    if (lnl_new < lnl_old) step /= 1.2;             // fitting.py:802
                    lnl_old = lnl_new;
                }
                store_mle(pl, o, m);
                pl.av[o] = av;
                pl.rv[o] = rv;
                pl.lnl[o] = lnl_new;
                pl.step[o] = step;
                if (lnl_new == lnl_new) {
                    L = lnl_new;
                    if (dl > p.ltol) T = lnl_new;
                }
            }
        }
        double *out = part + ((int64_t)blockIdx.x * nstar + s) * 2;
        block_max_store(L, slot, out);
        block_max_store(T, slot, out + 1);
    }
}

// Phase 4: constants and dimensionality prior (elementwise), fitting.py:806-815.
__global__ void __launch_bounds__(TILE)
k_finalize(int64_t nmodel, int nstar, const StarPrep *__restrict__ stars, DevParams p,
           const double *__restrict__ lnlp_max, Planes pl) {
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    if (i >= nmodel) return;
    const int s = blockIdx.y;
    const StarPrep &sp = stars[s];
    const int64_t o = (int64_t)s * pl.nmodel + i;
    const bool surv = pl.lnlp[o] > lnlp_max[s] + p.ln_init;
    const double chi2 = pl.chi2[o];
    double lnl = pl.lnl[o];
    if (surv) lnl += sp.lnl_const;                                  // fitting.py:806-807
    if (p.dim_prior)                                                // utils.py:161-176
        lnl = chi2 > 0. ? sp.c0 + sp.c1 * log(chi2) - chi2 / 2. : -INFINITY;
    pl.lnl[o] = lnl;
}

// PMC calibration stream with the fused scan's access widths: 4-byte loads and
// 8-byte stores per lane, a known byte count (see tools/pmc_traffic.py).
__global__ void k_calib_stream(const float *__restrict__ in, double *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (double)in[i];
}

// the guide's reference stream: 16 B per lane in, 16 B per lane out.  One element per lane,
// workgroups in address order, non-temporal accesses: the shape that reaches the guide's
// 6.3 TB/s on this pool (profiles/r04_stream_sweep.txt: 6.1 - 6.5 TB/s; the grid-stride loop
// over 8 192 workgroups that rounds 1-3 measured with reaches 4.5 - 5.0 with the same bytes).
typedef float calib_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(TILE)
k_calib_copy16(const calib_f4 *__restrict__ in, calib_f4 *__restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// Measurement aid: the vector unit's issue rate, by kind of instruction.  Every lane runs
// `iters` rounds of 128 independent-enough operations of one kind (8 accumulators), so a launch
// of w x 256 workgroups of 256 threads puts w waves on every SIMD that issue nothing else:
//   kind 0  v_fmac_f32      1  v_fmac_f64      2  v_exp_f32 (transcendental)
// (the same loops as tools/ubench/dpp_rate.hip, which also has the DPP and packed forms).
#define CALIB_REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256)
k_calib_issue(float *__restrict__ out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    const float r = threadIdx.x * 0.5f, x = 1.0001f;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const double dr = r, dx = x;
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {
            CALIB_REP16(asm volatile(
                "v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n"
                "v_fmac_f32_e32 %3, %8, %9\n v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n"
                "v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(r), "v"(x));)
        } else if constexpr (KIND == 1) {
            CALIB_REP16(asm volatile(
                "v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n v_fmac_f64_e32 %2, %4, %5\n"
                "v_fmac_f64_e32 %3, %4, %5\n v_fmac_f64_e32 %0, %4, %5\n v_fmac_f64_e32 %1, %4, %5\n"
                "v_fmac_f64_e32 %2, %4, %5\n v_fmac_f64_e32 %3, %4, %5\n"
                : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dr), "v"(dx));)
        } else {
            CALIB_REP16(asm volatile(
                "v_exp_f32_e32 %0, %0\n v_exp_f32_e32 %1, %1\n v_exp_f32_e32 %2, %2\n v_exp_f32_e32 %3, %3\n"
                "v_exp_f32_e32 %4, %4\n v_exp_f32_e32 %5, %5\n v_exp_f32_e32 %6, %6\n v_exp_f32_e32 %7, %7\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
    }
    out[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] =
        a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}
#undef CALIB_REP16

__global__ void k_debug_exp10(const double *__restrict__ x, double *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fast_exp10(x[i]);
}
__global__ void k_debug_math(int which, const double *__restrict__ x, double *__restrict__ y,
                             int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double sq, rsq;
    switch (which) {
    case 1: y[i] = fast_exp(v); break;
    case 2: y[i] = fast_log(v); break;
    case 3: y[i] = fast_sqrt(v); break;
    case 4: fast_sqrt_rsqrt(v, sq, rsq); y[i] = rsq; break;
    case 5: y[i] = fast_rcp(v); break;
    case 6: y[i] = fast_exp_fin(v, kExp2Tbl); break;
    case 7: y[i] = fast_log_pos(v); break;
    case 8: y[i] = fast_log_r(v); break;
    case 9: y[i] = fast_exp_bf(v, kExp2Tbl); break;
    default: y[i] = fast_exp10(v);
    }
}

__global__ void k_set_i32(int32_t *p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace
