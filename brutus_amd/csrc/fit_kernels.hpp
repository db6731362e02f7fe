// fit_kernels.hpp -- fused scan + compact flux phase + record emit behind brutus_fit_batch
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

namespace {

// ===========================================================================
// FAST PATH (brutus_fit_batch): fused full-grid scan + compact flux phase
// ===========================================================================
// The magnitude phase is a weighted linear least-squares problem in
// (offset, Av, Av*Rv) for every (star, model).  Instead of carrying the Nb
// residuals through the sweeps as the reference does, the fast path forms the
// ten weighted inner products of {1, r0, dr, y = mag_obs - mag_model} once and
// runs every sweep (fitting.py:176-243) on those scalars: the update formulas
// are algebraically identical, the results agree to rounding (~1e-14), and a
// sweep costs ~45 flops instead of ~14*Nb.
//
// One fused kernel then does, per (star, model): Gram sums -> 2 speculative
// sweeps with convergence statistics -> MLE at the sweep-2 state -> cull
// statistic lnl_p and the "not a survivor" first-cut statistic lnprob_ns.
// K1 = 2 for >90 % of stars; stars with a different K1 are re-run (a few
// percent of the batch).  Only two full planes are written (16 B per pair).

struct Gram {   // weighted inner products of {1, a=r0, b=dr, y}; weights 1/mags_var
    double ua, ub, uy, aa, ab, bb, ay, by, yy;
};

template <int NB>
__device__ __forceinline__ void gram_init(const Coef<NB> &c, const StarPrep &sp, Gram &G) {
    double ua = 0., ub = 0., uy = 0., aa = 0., ab = 0., bb = 0., ay = 0., by = 0., yy = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double w = sp.iW[j];
        const double a = (double)c.r0[j], b = (double)c.dr[j];
        const double y = sp.g[j] - (double)c.m[j];
        const double aw = a * w, bw = b * w, yw = y * w;
        ua += aw;
        ub += bw;
        uy += yw;
        aa += a * aw;
        ab += a * bw;
        bb += b * bw;
        ay += a * yw;
        by += b * yw;
        yy += y * yw;
    }
    G.ua = ua; G.ub = ub; G.uy = uy; G.aa = aa; G.ab = ab; G.bb = bb;
    G.ay = ay; G.by = by; G.yy = yy;
}

// One sweep of fitting.py:176-243 on the Gram scalars.  res = y - av*(a + rv*b).
__device__ __forceinline__ void gram_sweep(const Gram &G, double S, const DevParams &p, double &av,
                                           double &rv, double &dav_o, double &drv_o,
                                           double &logwt) {
    const double uR = G.ua + rv * G.ub;                       // sum w R
    const double RR = G.aa + rv * (2. * G.ab + rv * G.bb);    // sum w R^2
    const double yR = G.ay + rv * G.by;                       // sum w y R
    double rs = G.uy - av * uR;                               // sum w res
    const double ra = (yR - av * RR) + (p.av_mean - av) * p.av_ivar;
    const double a_den = RR + p.av_ivar;
    double dav = (S * ra - uR * rs) / (S * a_den - uR * uR);
    if (dav < p.avmin - av) dav = p.avmin - av;
    if (dav > p.avmax - av) dav = p.avmax - av;
    av = av + dav;
    const double r_den = G.bb * av * av + p.rv_ivar;
    const double sr = G.ub * av;
    rs = G.uy - av * uR;
    const double bres = G.by - av * (G.ab + rv * G.bb);       // sum w res b
    const double rr = av * bres + (p.rv_mean - rv) * p.rv_ivar;
    double drv = (S * rr - sr * rs) / (S * r_den - sr * sr);
    if (drv < p.rvmin - rv) drv = p.rvmin - rv;
    if (drv > p.rvmax - rv) drv = p.rvmax - rv;
    rv = rv + drv;
    const double RR2 = G.aa + rv * (2. * G.ab + rv * G.bb);
    const double yR2 = G.ay + rv * G.by;
    const double chi2 = G.yy - av * (2. * yR2 - av * RR2);
    dav_o = dav;
    drv_o = drv;
    logwt = -0.5 * chi2;
}

// MLE quantities as mle_eval, with F = F0 * 10^(-0.4 av R) through fast_exp10.
template <int NB, bool TBL>
__device__ __forceinline__ void mle_fast(const Coef<NB> &c, const double (&F0)[NB],
                                         const StarPrep &sp, const DevParams &p, double av,
                                         double rv, const double *__restrict__ tbl, Mle &o) {
    const double fac = -0.92103403719761827361;
    const double mav = -0.4 * av;
    double F[NB];
    double s_num = 0., s_den = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double R = (double)c.r0[j] + rv * (double)c.dr[j];
        const double f = F0[j] * (TBL ? fast_exp10(mav * R, tbl) : poly_exp10(mav * R));
        F[j] = f;
        const double fw = f * sp.iV[j];
        s_num += sp.d[j] * fw;
        s_den += f * fw;
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;
    double sr_mix = 0., sa_mix = 0., ar_mix = 0., a_den = 0., r_den = 0.;
    double a_num = 0., r_num = 0., chi2 = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double iv = sp.iV[j];
        const double D0 = (double)c.dr[j];
        const double R0 = (double)c.r0[j] + rv * D0;
        const double ff = fac * F[j];
        double Rf = R0 * ff;
        double Df = D0 * ff;
        double red = F[j] - F0[j];
        const double Fs = F[j] * s;
        const double res = sp.d[j] - Fs;
        const double t = (Fs - res) * iv;
        sr_mix += Df * t;
        sa_mix += Rf * t;
        Rf *= s;
        Df *= s;
        red *= s;
        ar_mix += Df * ((red - res) * iv);
        a_den += Rf * Rf * iv;
        r_den += Df * Df * iv;
        const double rw = res * iv;
        a_num += Rf * rw;
        r_num += Df * rw;
        chi2 += res * rw;
    }
    o.a_ss = a_den;
    o.r_ss = r_den;
    o.a_num = a_num;
    o.r_num = r_num;
    a_den += p.av_ivar;
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

// ---- pinned Rv (rvlim[0] == rvlim[1] == rv_gauss[0], BASELINE configs[1]) ----
// The Rv step of every sweep is clamped to zero, so R_j = r0_j + rv dr_j is a
// per-model constant and the magnitude phase is a 2-parameter (offset, Av)
// problem: five weighted inner products of {1, R, y} instead of nine, and only
// the Av half of a sweep.  Same formulas as gram_init / gram_sweep with rv fixed;
// the Rv rows of the precision matrix are still reported (mle_fast_rf<FULL>).
struct GramR {
    double uR, RR, yR, uy, yy;
};

template <int NB>
__device__ __forceinline__ void coef_R(const Coef<NB> &c, double rv, double (&R)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) R[j] = (double)c.r0[j] + rv * (double)c.dr[j];
}

template <int NB>
__device__ __forceinline__ void gram_init_rf(const Coef<NB> &c, const double (&R)[NB],
                                             const StarPrep &sp, GramR &G) {
    double uR = 0., RR = 0., yR = 0., uy = 0., yy = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double w = sp.iW[j];
        const double y = sp.g[j] - (double)c.m[j];
        const double Rw = R[j] * w, yw = y * w;
        uR += Rw;
        RR += R[j] * Rw;
        yR += R[j] * yw;
        uy += yw;
        yy += y * yw;
    }
    G.uR = uR; G.RR = RR; G.yR = yR; G.uy = uy; G.yy = yy;
}

// The Av half of fitting.py:176-243 (the Rv half moves nothing when rvmin == rvmax).
__device__ __forceinline__ void gram_sweep_rf(const GramR &G, double S, const DevParams &p,
                                              double &av, double &dav_o, double &logwt) {
    const double rs = G.uy - av * G.uR;
    const double ra = (G.yR - av * G.RR) + (p.av_mean - av) * p.av_ivar;
    const double a_den = G.RR + p.av_ivar;
    double dav = (S * ra - G.uR * rs) / (S * a_den - G.uR * G.uR);
    if (dav < p.avmin - av) dav = p.avmin - av;
    if (dav > p.avmax - av) dav = p.avmax - av;
    av = av + dav;
    const double chi2 = G.yy - av * (2. * G.yR - av * G.RR);
    dav_o = dav;
    logwt = -0.5 * chi2;
}

// mle_fast with R given.  FULL = false leaves out the Rv sums (i02, i12, i22,
// r_num, r_ss), which only the reported precision matrix needs.
template <int NB, bool TBL, bool FULL>
__device__ __forceinline__ void mle_fast_rf(const Coef<NB> &c, const double (&R)[NB],
                                            const double (&F0)[NB], const StarPrep &sp,
                                            const DevParams &p, double av,
                                            const double *__restrict__ tbl, Mle &o) {
    const double fac = -0.92103403719761827361;
    const double mav = -0.4 * av;
    double F[NB];
    double s_num = 0., s_den = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double f = F0[j] * (TBL ? fast_exp10(mav * R[j], tbl) : poly_exp10(mav * R[j]));
        F[j] = f;
        const double fw = f * sp.iV[j];
        s_num += sp.d[j] * fw;
        s_den += f * fw;
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;
    double sr_mix = 0., sa_mix = 0., ar_mix = 0., a_den = 0., r_den = 0.;
    double a_num = 0., r_num = 0., chi2 = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double iv = sp.iV[j];
        const double ff = fac * F[j];
        double Rf = R[j] * ff;
        const double Fs = F[j] * s;
        const double res = sp.d[j] - Fs;
        const double t = (Fs - res) * iv;
        sa_mix += Rf * t;
        Rf *= s;
        a_den += Rf * Rf * iv;
        const double rw = res * iv;
        a_num += Rf * rw;
        chi2 += res * rw;
        if (FULL) {
            double Df = (double)c.dr[j] * ff;
            sr_mix += Df * t;
            Df *= s;
            const double red = (F[j] - F0[j]) * s;
            ar_mix += Df * ((red - res) * iv);
            r_den += Df * Df * iv;
            r_num += Df * rw;
        }
    }
    o.a_ss = a_den;
    o.r_ss = r_den;
    o.a_num = a_num;
    o.r_num = r_num;
    a_den += p.av_ivar;
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

// The three MLE quantities the fused scan needs (scale, chi2, i00 = sum F^2/V) in
// ONE pass over the bands: with s = sum(d F / V) / sum(F^2 / V),
//   chi2 = sum (d - s F)^2 / V = D2 - 2 s sum(d F / V) + s^2 sum(F^2 / V),
// D2 = sum d^2 / V being a per-star constant (StarPrep::D2).  No per-band flux
// array stays live (24 VGPRs at 12 bands) and the second band loop goes away.
// The expansion cancels ~4 digits (D2 ~ 1e4-1e5 against chi2 ~ 10): chi2 is good
// to ~1e-11 absolute -- it only feeds the cull / first-cut decisions here; every
// reported value comes from the two-pass form (mle_fast*).
template <int NB, bool RVF>
__device__ __forceinline__ void mle_scan(const Coef<NB> &c, const double (&R)[RVF ? NB : 1],
                                         const double (&F0)[NB], const StarPrep &sp, double av,
                                         double rv, const double *__restrict__ tbl, Mle &o) {
    const double mav = -0.4 * av;
    double s_num = 0., s_den = 0.;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        double Rj;
        if constexpr (RVF) Rj = R[j];
        else Rj = (double)c.r0[j] + rv * (double)c.dr[j];
        const double f = F0[j] * fast_exp10(mav * Rj, tbl);
        const double fw = f * sp.iV[j];
        s_num += sp.d[j] * fw;
        s_den += f * fw;
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;
    o.scale = s;
    o.i00 = s_den;
    o.chi2 = fma(-s, s_num, sp.D2) + s * fma(s, s_den, -s_num);
}

// F0 of model i from the band-major table (coalesced) / of one model from its row.
template <int NB>
__device__ __forceinline__ void load_F0(const float *__restrict__ grid, int64_t nmodel_pad,
                                        int64_t i, double (&F0)[NB]) {
    const double *t = reinterpret_cast<const double *>(grid + (int64_t)6 * NB * nmodel_pad);
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = t[(int64_t)j * nmodel_pad + i];
}
// The gather kernels are close to memory-bound and VGPR-limited, so they
// recompute F0 with the table-free polynomial (<= 1 ulp from the tabulated
// value) instead of reading 8*NB more bytes per model.
template <int NB>
__device__ __forceinline__ void compute_F0_fast(const Coef<NB> &c, double (&F0)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = poly_exp10(-0.4 * (double)c.m[j]);
}
// ... or with the table-driven form where registers allow: bit-identical to the
// tabulated F0 the scan reads
template <int NB>
__device__ __forceinline__ void compute_F0_tbl(const Coef<NB> &c, const double *__restrict__ tbl,
                                               double (&F0)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = fast_exp10(-0.4 * (double)c.m[j], tbl);
}

// lnl as `loglike` returns it for a model the cull dropped / kept, and the
// first-cut statistic lnprob (fitting.py:806-815, 976-985; pdf.py:209-218).
// FASTLOG: ln through fast_log_r (~45 instructions instead of ocml's ~90; a few
// ulp) -- the pinned-Rv kernels, which have the registers for it.
template <bool FASTLOG = false>
__device__ __forceinline__ double final_lnl(const StarPrep &sp, const DevParams &p, double chi2,
                                            bool survivor) {
    if (p.dim_prior)
        return chi2 > 0. ? sp.c0 + sp.c1 * (FASTLOG ? fast_log_r(chi2) : log(chi2)) - chi2 / 2.
                         : -INFINITY;
    return survivor ? -0.5 * chi2 + sp.lnl_const : -0.5 * chi2;
}
__device__ __forceinline__ double first_cut_lnprob(const StarPrep &sp, double lnl, double scale,
                                                   double i00) {
    double lnprob = lnl;
    if (sp.sp_on) {
        const double serr2 = 1. / fabs(i00);
        const double vt = sp.sp_var + serr2;
        const double ds = scale - sp.sp_mean;
        lnprob = lnl + -0.5 * (ds * ds / vt + log(2. * M_PI * vt));
    }
    if (!isfinite(lnprob)) lnprob = -BIG;
    return lnprob;
}

// cull statistic lnl_p (fitting.py:745-756)
__device__ __forceinline__ double cull_stat(const StarPrep &sp, const Mle &m) {
    double lnlp = -0.5 * m.chi2;
    if (sp.has_par) {
        const double dp = sqrt(m.scale) - sp.par;
        lnlp -= 0.5 * (dp * dp * sp.par_ivar);
    }
    return lnlp;
}

// FS_G = stars per workgroup of the fused scan (LDS: FS_G * NV * 2 KiB)

// Fused full-grid scan.  grid = (ceil(ntile / tiles_per_block), ceil(nrun / FS_G)).
//   FS_G             stars per workgroup (LDS = FS_G * NV * 2 KiB)
//   star_ids[nrun]   stars (indices into `stars`) handled by this launch
//   kfix[star]       number of magnitude sweeps before the MLE
// Per (block.x, star) emits NV = 2*KS + 2 maxima into part[(bx * nstar + star) * NV + v]:
//   v = 2k, 2k+1 : L_k, T_k for sweep k < KS   (only sweeps <= kfix are run)
//   v = 2KS      : max lnl_p;  v = 2KS+1 : max lnprob_ns
//   RVF              pinned-Rv specialisation (see GramR)
template <int NB, int KS, int FS_G, bool RVF>
__global__ void __launch_bounds__(TILE, 3)   // <=168 VGPRs: 3 waves/SIMD (LDS allows 3 blocks/CU)
k_fscan(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
        const int32_t *__restrict__ star_ids, const StarPrep *__restrict__ stars, DevParams p,
        const int32_t *__restrict__ kfix, int tiles_per_block, int ntile, Planes pl,
        double *__restrict__ part) {
    constexpr int NV = 2 * KS + 2;
    extern __shared__ double smax[];   // [FS_G][NV][TILE]
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    const int g0 = blockIdx.y * FS_G;
    const int ng = min(FS_G, nrun - g0);
    for (int q = threadIdx.x; q < FS_G * NV * TILE; q += TILE) smax[q] = -INFINITY;
    // each thread only ever touches its own column of smax: no barrier needed
    const int t0 = blockIdx.x * tiles_per_block;
    const int t1 = min(ntile, t0 + tiles_per_block);
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        const bool live = i < nmodel;
        Coef<NB> c;
        load_coef<NB>(grid, nmodel_pad, i, c);
        double F0[NB];
        load_F0<NB>(grid, nmodel_pad, i, F0);
        double R[RVF ? NB : 1];
        if constexpr (RVF) coef_R<NB>(c, p.rv_mean, R);
        for (int g = 0; g < ng; ++g) {
            const int s = star_ids[g0 + g];
            const StarPrep &sp = stars[s];
            double av = p.av_mean, rv = p.rv_mean;
            const int K = kfix[s];
            double *col = smax + (size_t)g * NV * TILE + threadIdx.x;
            Mle m;
            if constexpr (RVF) {
                GramR G;
                gram_init_rf<NB>(c, R, sp, G);
                // With Rv pinned the objective is exactly quadratic in (offset, Av)
                // and a sweep is its Newton step with the offset eliminated: the
                // first sweep lands on the (clamped) minimiser, every later one
                // moves by rounding noise (<< mtol) and leaves logwt unchanged.
                // So one sweep is computed; sweeps 2..K only enter the statistics
                // (L_k = L_1, no step above tolerance), which makes K1 <= 2 as in
                // the reference.
                double dav, lw;
                gram_sweep_rf(G, sp.S, p, av, dav, lw);
                if (live && lw == lw) {
                    if (lw > col[0]) col[0] = lw;
                    if (fabs(dav) >= p.mtol && lw > col[TILE]) col[TILE] = lw;
                    for (int k = 1; k < K && k < KS; ++k) {
                        double *c0 = col + (size_t)(2 * k) * TILE;
                        if (lw > c0[0]) c0[0] = lw;
                    }
                }
                mle_scan<NB, true>(c, R, F0, sp, av, rv, s_tbl, m);
            } else {
                Gram G;
                gram_init<NB>(c, sp, G);
                for (int k = 0; k < K; ++k) {
                    double dav, drv, lw;
                    gram_sweep(G, sp.S, p, av, rv, dav, drv, lw);
                    if (k < KS && live && lw == lw) {
                        const bool big = (fabs(dav) >= p.mtol) || (fabs(drv) >= p.mtol);
                        double *c0 = col + (size_t)(2 * k) * TILE;
                        if (lw > c0[0]) c0[0] = lw;
                        if (big && lw > c0[TILE]) c0[TILE] = lw;
                    }
                }
                mle_scan<NB, false>(c, R, F0, sp, av, rv, s_tbl, m);
            }
            const double lnl = -0.5 * m.chi2;
            double lnlp = lnl;
            if (sp.has_par) {
                const double dp = sqrt(m.scale) - sp.par;
                lnlp = lnl - 0.5 * (dp * dp * sp.par_ivar);
            }
            const double lnprob =
                first_cut_lnprob(sp, final_lnl<RVF>(sp, p, m.chi2, false), m.scale, m.i00);
            if (live) {
                const int64_t o = (int64_t)s * pl.nmodel + i;
                pl.lnlp[o] = lnlp;
                pl.lnprob[o] = lnprob;
                double *c0 = col + (size_t)(2 * KS) * TILE;
                if (lnlp > c0[0]) c0[0] = lnlp;          // NaN never wins
                if (lnprob > c0[TILE]) c0[TILE] = lnprob;
            }
        }
    }
    for (int g = 0; g < ng; ++g) {
        const int s = star_ids[g0 + g];
        for (int v = 0; v < NV; ++v)
            block_max_store(smax[((size_t)g * NV + v) * TILE + threadIdx.x], slot,
                            part + ((int64_t)blockIdx.x * nstar + s) * NV + v);
    }
}

// Reduce the fused-scan partials of the stars in `star_ids` and decide.
//   accept == 0: derive K1 from (L_k, T_k); k1[s] = K1 (0 = not converged in KS)
//   always: thr_cull[s] = max lnl_p + ln(init_thresh);  maxns[s] = max lnprob_ns
__global__ void k_fdecide(int nblkx, int nstar, int nrun, const int32_t *__restrict__ star_ids,
                          int KS, const double *__restrict__ part, DevParams p, int accept,
                          int32_t *__restrict__ k1, double *__restrict__ thr_cull,
                          double *__restrict__ maxns) {
    __shared__ double sm[KCAP * 2 + 2][4];
    const int s = star_ids[blockIdx.x];
    const int NV = 2 * KS + 2;
    double v[KCAP * 2 + 2];
    for (int q = 0; q < NV; ++q) v[q] = -INFINITY;
    for (int b = threadIdx.x; b < nblkx; b += blockDim.x) {
        const double *pp = part + ((int64_t)b * nstar + s) * NV;
        for (int q = 0; q < NV; ++q) v[q] = pp[q] > v[q] ? pp[q] : v[q];
    }
    for (int q = 0; q < NV; ++q) {
        const double m = wave_max(v[q]);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int q = 0; q < NV; ++q) {
        double m = sm[q][0];
        for (int w = 1; w < 4; ++w) m = sm[q][w] > m ? sm[q][w] : m;
        v[q] = m;
    }
    if (!accept) {
        int K1 = 0;
        for (int k = 0; k < KS; ++k) {
            const double L = v[2 * k] > -BIG ? v[2 * k] : -BIG;
            if (!(v[2 * k + 1] > L + p.ln_init)) {
                K1 = k + 1;
                break;
            }
        }
        k1[s] = K1;
    }
    thr_cull[s] = v[2 * KS] + p.ln_init;
    maxns[s] = v[2 * KS + 1];
}

// Ordered compaction of one (nstar, nmodel) plane against a per-star threshold:
// {i : plane[s][i] > thr[s]}.  grid = (NCHUNK, nstar).  Optionally also the
// maximum of `other[s][i]` over the complement (models that fail the test).
__global__ void __launch_bounds__(TILE)
k_cmp_count(int64_t nmodel, int ntile, const double *__restrict__ plane,
            const double *__restrict__ thr, const double *__restrict__ other,
            int64_t *__restrict__ counts, double *__restrict__ other_max,
            unsigned long long *__restrict__ mask) {
    __shared__ int wsum[4];
    __shared__ double slot[4];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    const double th = thr[s];
    int n = 0;
    double om = -INFINITY;
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        bool hit = false;
        if (i < nmodel) {
            const int64_t o = (int64_t)s * nmodel + i;
            if (plane[o] > th) {
                hit = true;
                ++n;
            } else if (other) {
                const double x = other[o];
                if (x > om) om = x;
            }
        }
        // one 64-bit membership word per wave: the scatter pass reads these
        // instead of the 8-byte-per-model plane
        const unsigned long long b = __ballot(hit);
        if ((threadIdx.x & 63) == 0)
            mask[(int64_t)s * (4 * ntile) + (int64_t)t * 4 + (threadIdx.x >> 6)] = b;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)s * NCHUNK + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (other) block_max_store(om, slot, other_max + (int64_t)s * NCHUNK + c);
}

// Exclusive scan of counts[(s, c)] in (s, c) order:
//   offsets[(s, c)], star_off[s] (star_off[nstar] = total).
// Work items of the list kernels (k_fflux, k_emit): the segment of star s in model chunk c,
// cut into pieces of TILE entries, numbered CHUNK-MAJOR: wbase[c * nstar + s] = first item
// of that segment, wbase[NCHUNK * nstar] = #items.  Workgroups that run at the same time
// then work on the same 1/NCHUNK of the grid for different stars, and the coefficient rows
// they gather are L2 hits instead of one fabric read per star.
// One workgroup of BRUTUS_MAX_BATCH threads, two exclusive scans over the nstar x NCHUNK
// counts -- star-major for the list offsets, chunk-major for the work items -- each lane
// taking a run of consecutive entries whose loads are all in flight at once (the serial
// per-star / per-chunk loops this replaces paid one memory round trip per entry).
__global__ void __launch_bounds__(BRUTUS_MAX_BATCH)
k_offsets(int nstar, const int64_t *__restrict__ counts, int64_t *__restrict__ offsets,
          int64_t *__restrict__ star_off, int32_t *__restrict__ wbase) {
    constexpr int NT = BRUTUS_MAX_BATCH;
    constexpr int EPT = NCHUNK;                       // entries per lane at the full batch
    typedef hipcub::BlockScan<int64_t, NT> Scan64;
    typedef hipcub::BlockScan<int32_t, NT> Scan32;
    __shared__ union {
        typename Scan64::TempStorage a;
        typename Scan32::TempStorage b;
    } tmp;
    const int total = nstar * NCHUNK;
    const int ept = (total + NT - 1) / NT;
    const int e0 = threadIdx.x * ept;
    {
        int64_t r[EPT];
        int64_t sum = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            r[k] = k < ept && e0 + k < total ? counts[e0 + k] : 0;
            sum += r[k];
        }
        int64_t pre, all;
        Scan64(tmp.a).ExclusiveSum(sum, pre, all);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if (k < ept && e0 + k < total) {
                offsets[e0 + k] = pre;
                if ((e0 + k) % NCHUNK == 0) star_off[(e0 + k) / NCHUNK] = pre;
            }
            pre += r[k];
        }
        if (threadIdx.x == 0) star_off[nstar] = all;
    }
    if (!wbase) return;
    __syncthreads();
    {
        int32_t r[EPT];
        int32_t sum = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = e0 + k;                     // = c * nstar + q
            const int c = e / nstar, q = e - c * nstar;
            r[k] = k < ept && e < total ? (int32_t)((counts[(int64_t)q * NCHUNK + c] + TILE - 1) / TILE) : 0;
            sum += r[k];
        }
        int32_t pre, all;
        Scan32(tmp.b).ExclusiveSum(sum, pre, all);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if (k < ept && e0 + k < total) wbase[e0 + k] = pre;
            pre += r[k];
        }
        if (threadIdx.x == 0) wbase[total] = all;
    }
}

// A chunk's membership words are fetched 64 tiles (256 words, one per lane) at a time and
// prefix-summed once; the per-tile loop then runs out of LDS.  (One dependent global load
// per tile made this kernel latency-bound: 46 round trips per workgroup.)
__global__ void __launch_bounds__(TILE)
k_cmp_scatter(int64_t nmodel, int ntile, const unsigned long long *__restrict__ mask,
              const int64_t *__restrict__ offsets, int64_t capacity,
              int32_t *__restrict__ out_idx) {
    typedef hipcub::BlockScan<int, TILE> Scan;
    __shared__ typename Scan::TempStorage s_scan;
    __shared__ unsigned long long s_word[TILE];
    __shared__ int s_pre[TILE];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    int64_t base = offsets[(int64_t)s * NCHUNK + c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long *__restrict__ row = mask + (int64_t)s * (4 * ntile);
    for (int tb = t0; tb < t1; tb += TILE / 4) {
        const int nt = min(TILE / 4, t1 - tb);
        const unsigned long long word =
            (int)threadIdx.x < 4 * nt ? row[(int64_t)tb * 4 + threadIdx.x] : 0ull;
        int pre, tot;
        Scan(s_scan).ExclusiveSum(__popcll(word), pre, tot);
        s_word[threadIdx.x] = word;
        s_pre[threadIdx.x] = pre;
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < nt; ++k) {
            const unsigned long long b = s_word[4 * k + w];
            if ((b >> lane) & 1ull) {
                const int64_t r = base + s_pre[4 * k + w] + __popcll(b & below);
                if (r < capacity) out_idx[r] = (int32_t)((int64_t)(tb + k) * TILE + threadIdx.x);
            }
        }
        base += tot;
        __syncthreads();
    }
}

// Second generation (fit2_kernels.hpp): the float32 lnprob~ plane doubles as the survivor
// map.  Its genuine entries are negative, -inf or NaN (the value of a survivor is not
// needed again: survivors are judged by their final float64 lnprob); k_fflux overwrites a
// survivor's entry with the bit pattern 1 + (position in the star's candidate list), a
// positive finite word, and leaves a candidate that failed the exact cull test alone.
// Every later pass (exact first-cut threshold, classification, emit) then reads ONE plane.  The flux-phase results live in
// candidate-list order ("staging": the Planes arrays indexed by list position), so they
// are written as full lines and read back densely.
__device__ __forceinline__ float surv_tag(int64_t slot) { return __int_as_float((int)slot + 1); }
__device__ __forceinline__ bool surv_is(float x) {
    const int b = __float_as_int(x);
    return b > 0 && b < 0x7F800000;
}
__device__ __forceinline__ int surv_slot(float x) { return __float_as_int(x) - 1; }

// A work item -> its star and the list positions [q0, q0 + n) it covers (k_offsets: items
// are pieces of (star, chunk) segments, numbered chunk-major).
struct ItemGeom {
    int64_t q0;
    int32_t s, n;
};
// one record per work item, written by the segments' owners (a 13-step search over the
// segment table per item costs the list kernels 10 %)
__global__ void k_items(int nstar, const int32_t *__restrict__ wbase,
                        const int64_t *__restrict__ offsets, const int64_t *__restrict__ star_off,
                        ItemGeom *__restrict__ items) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NCHUNK * nstar) return;
    const int c = e / nstar, s = e - c * nstar;
    const int64_t start = offsets[(int64_t)s * NCHUNK + c];
    const int64_t end = c + 1 < NCHUNK ? offsets[(int64_t)s * NCHUNK + c + 1] : star_off[s + 1];
    const int first = wbase[e], np = wbase[e + 1] - first;
    for (int k = 0; k < np; ++k) {
        ItemGeom g;
        g.q0 = start + (int64_t)k * TILE;
        g.s = s;
        g.n = (int)(end - g.q0 < TILE ? end - g.q0 : TILE);
        items[first + k] = g;
    }
}

// Walk of the work items that gives every XCD whole model chunks: workgroups are dispatched
// round-robin over the 8 XCDs, so workgroup b runs on XCD b % 8 and takes the chunks
// c = b % 8, b % 8 + 8, ...; inside a chunk the XCD's workers (`nw` of them, this one is
// number `me`) take the items in turn.  A chunk's coefficient rows then cross the fabric
// once, into ONE L2, and serve all stars from there (chunk-major numbering alone still
// fetched them into all eight).
struct ItemWalk {
    const int32_t *wbase;
    int nstar, c, item, me, nw;
    __device__ __forceinline__ void seek() {      // first chunk from c on with an item for `me`
        for (; c < NCHUNK; c += 8) {
            item = wbase[c * nstar] + me;
            if (item < wbase[(c + 1) * nstar]) return;
        }
        item = -1;
    }
    __device__ __forceinline__ void init(const int32_t *wb, int ns, int xcd, int me_, int nw_) {
        wbase = wb;
        nstar = ns;
        me = me_;
        nw = nw_;
        c = xcd;
        seek();
    }
    __device__ __forceinline__ bool done() const { return item < 0; }
    __device__ __forceinline__ void next() {
        item += nw;
        if (item >= wbase[(c + 1) * nstar]) {
            c += 8;
            seek();
        }
    }
};

// Coefficients of ONE model from the model-major copy: 3*NB/4 16-byte loads.
template <int NB>
__device__ __forceinline__ void gather_coef(const float *__restrict__ grid, int64_t nmodel_pad,
                                            int64_t i, Coef<NB> &c) {
    const float4 *row =
        reinterpret_cast<const float4 *>(grid + (int64_t)3 * NB * nmodel_pad + i * (3 * NB));
    float t[3 * NB];
#pragma unroll
    for (int q = 0; q < 3 * NB / 4; ++q) {
        const float4 v = row[q];
        t[4 * q] = v.x;
        t[4 * q + 1] = v.y;
        t[4 * q + 2] = v.z;
        t[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        c.m[j] = t[3 * j];
        c.r0[j] = t[3 * j + 1];
        c.dr[j] = t[3 * j + 2];
    }
}

// Flux phase on the compact survivor lists (fitting.py:758-803), persistent
// workgroups looping over work items.  First launch: rebuild (av, rv) from K1
// sweeps, two iterations from lnl_old = -1e300; continuation: one iteration from
// the state planes.  Writes the state/result planes at the survivors' positions
// and, per work item and wave, L = max lnl_new, T = max{lnl_new : |dlnl| > ltol},
// M = max final lnprob.
// Second-generation mode (surv32 != nullptr, fit2_kernels.hpp): the list holds the
// CANDIDATES (lnl_p~ >= threshold - eps); the first launch applies the exact cull test
// lnl_p > thr_cull[s] (fitting.py:758-759) and marks the outcome in the float32 plane
// (survivor tag / -inf, see surv_tag); only survivors iterate, store and enter the
// statistics, and they store at their LIST POSITION q, not at (star, model).
template <int NB, bool RVF, bool FIRST>
__global__ void __launch_bounds__(TILE, 2)
k_fflux(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
        const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
        const int32_t *__restrict__ k2state, const int32_t *__restrict__ surv_idx,
        const int64_t *__restrict__ surv_off, const int32_t *__restrict__ wbase,
        const ItemGeom *__restrict__ items, Planes pl, double *__restrict__ part,
        float *__restrict__ surv32, const double *__restrict__ thr_cull) {
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    // (the launch kind is a template parameter: the opening launch then carries no
    // state-reload path and its two iterations unroll)
    constexpr int niter = FIRST ? 2 : 1;
    // this lane's model in a work item, requested one item ahead (a dead lane reads the
    // item's last entry: no select on the loaded value, so nothing waits for it here)
    auto lane_model = [&](int item) -> int32_t {
        if (item < 0) return 0;
        const ItemGeom ig = items[item];
        const int64_t q = ig.q0 + threadIdx.x;
        const int64_t last = ig.q0 + ig.n - 1;
        return surv_idx[q < last ? q : last];
    };
    ItemWalk wk;
    wk.init(wbase, nstar, blockIdx.x & 7, blockIdx.x >> 3, (gridDim.x + 7 - (blockIdx.x & 7)) >> 3);
    int32_t i_nxt = lane_model(wk.item);
    while (!wk.done()) {
        const int item = wk.item;
        wk.next();
        const ItemGeom ig = items[item];
        const int s = ig.s;
        const int32_t i_me = i_nxt;
        i_nxt = lane_model(wk.item);
        if (k2state[s] < 0) continue;
        const StarPrep &sp = stars[s];
        const int64_t q = ig.q0 + threadIdx.x;
        const bool live = (int)threadIdx.x < ig.n;
        double L = -INFINITY, T = -INFINITY, M = -INFINITY;
        bool go = live;
        int64_t i = 0, o = 0;
        if (live) {
            i = i_me;
            o = (int64_t)s * pl.nmodel + i;
            if (surv32 && !FIRST) go = surv_is(surv32[o]);
        }
        const int64_t os = surv32 ? q : o;       // where this entry's state / results live
        if (go) {
            Coef<NB> c;
            gather_coef<NB>(grid, nmodel_pad, i, c);
            double F0[NB];
            if constexpr (RVF) compute_F0_tbl<NB>(c, s_tbl, F0);
            else compute_F0_fast<NB>(c, F0);
            double av, rv, step, lnl_old;
            double R[RVF ? NB : 1];
            if constexpr (RVF) coef_R<NB>(c, p.rv_mean, R);
            if constexpr (FIRST) {
                av = p.av_mean;
                rv = p.rv_mean;
                const int K = k1[s];
                if constexpr (RVF) {
                    GramR G;
                    gram_init_rf<NB>(c, R, sp, G);
                    // (a single sweep would do, see k_fscan; the loop form keeps this
                    // kernel's register allocation below the spill line)
                    for (int k = 0; k < K; ++k) {
                        double a_, c_;
                        gram_sweep_rf(G, sp.S, p, av, a_, c_);
                    }
                } else {
                    Gram G;
                    gram_init<NB>(c, sp, G);
                    for (int k = 0; k < K; ++k) {
                        double a_, b_, c_;
                        gram_sweep(G, sp.S, p, av, rv, a_, b_, c_);
                    }
                }
                step = 1.0;
                lnl_old = -BIG;
            } else {
                av = pl.av[os];
                rv = RVF ? p.rv_mean : pl.rv[os];     // pinned Rv is not staged
                step = pl.step[os];
                lnl_old = -0.5 * pl.chi2[os];
            }
            Mle m;
            if constexpr (RVF) mle_fast_rf<NB, true, false>(c, R, F0, sp, p, av, s_tbl, m);
            else mle_fast<NB, false>(c, F0, sp, p, av, rv, nullptr, m);
            if (surv32 && FIRST) {
                go = cull_stat(sp, m) > thr_cull[s];
                if (go) surv32[o] = surv_tag(q - surv_off[s]);     // a failed candidate keeps its lnprob~
            }
            double lnl_new = lnl_old, dl = 0.;
            for (int it = 0; go && it < niter; ++it) {
                double dav = (m.a_num + (p.av_mean - av) * p.av_ivar) / (m.a_ss + p.av_ivar) * step;
                if (dav < p.avmin - av) dav = p.avmin - av;
                if (dav > p.avmax - av) dav = p.avmax - av;
                av += dav;
                if constexpr (RVF) {
                    // the Rv step is clamped to zero; only the stored MLE needs the Rv sums
                    if (it + 1 < niter) mle_fast_rf<NB, true, false>(c, R, F0, sp, p, av, s_tbl, m);
                    else mle_fast_rf<NB, true, true>(c, R, F0, sp, p, av, s_tbl, m);
                } else {
                    double drv = (m.r_num + (p.rv_mean - rv) * p.rv_ivar) / (m.r_ss + p.rv_ivar) * step;
                    if (drv < p.rvmin - rv) drv = p.rvmin - rv;
                    if (drv > p.rvmax - rv) drv = p.rvmax - rv;
                    rv += drv;
                    mle_fast<NB, false>(c, F0, sp, p, av, rv, nullptr, m);
                }
                lnl_new = -0.5 * m.chi2;
                dl = fabs(lnl_new - lnl_old);
                if (lnl_new < lnl_old) step /= 1.2;
                lnl_old = lnl_new;
            }
            if (go) {
                store_mle(pl, os, m);
                pl.av[os] = av;
                if constexpr (!RVF) pl.rv[os] = rv;
                pl.step[os] = step;
                const double lnl = final_lnl<RVF>(sp, p, m.chi2, true);
                const double lnprob = first_cut_lnprob(sp, lnl, m.scale, m.i00);
                pl.lnl[os] = lnl;
                pl.lnprob[os] = lnprob;
                M = lnprob;
                if (lnl_new == lnl_new) {
                    L = lnl_new;
                    if (dl > p.ltol) T = lnl_new;
                }
            }
        }
        // one partial per wave: no workgroup barrier in the loop, the four waves drift
        // apart and overlap each other's gather latency
        L = wave_max(L);
        T = wave_max(T);
        M = wave_max(M);
        if ((threadIdx.x & 63) == 0) {
            double *out = part + ((int64_t)item * (TILE / 64) + (threadIdx.x >> 6)) * 3;
            out[0] = L;
            out[1] = T;
            out[2] = M;
        }
    }
}

// Per-star flux decision over the star's work items (one workgroup per star).
__global__ void k_fflux_decide(int nstar, const int32_t *__restrict__ wbase,
                               const double *__restrict__ part, double ln_sub,
                               int32_t *__restrict__ k2state, double *__restrict__ maxsurv,
                               int32_t *__restrict__ n_unconv) {
    __shared__ double sm[3][4];
    const int s = blockIdx.x;
    if (k2state[s] < 0) return;
    double v[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int per = TILE / 64;       // k_fflux leaves one partial per wave
    // the star's items: one run per model chunk.  blockDim.x / NCHUNK lanes share a run and
    // take its partials four at a time (clamped addresses: all twelve loads in flight; a
    // chunk-by-chunk loop paid two dependent round trips per chunk)
    {
        const int lpc = blockDim.x / NCHUNK;
        const int c = threadIdx.x / lpc, j = threadIdx.x - c * lpc;
        const int e = c * nstar + s;
        const int lo = wbase[e] * per, hi = wbase[e + 1] * per;
        constexpr int U = 4;
        for (int it = lo + j; it < hi; it += U * lpc) {
            double x[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int a = it + u * lpc < hi ? it + u * lpc : it;
#pragma unroll
                for (int q = 0; q < 3; ++q) x[u][q] = part[(int64_t)a * 3 + q];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < 3; ++q) v[q] = x[u][q] > v[q] ? x[u][q] : v[q];
        }
    }
    for (int q = 0; q < 3; ++q) {
        const double m = wave_max(v[q]);
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int q = 0; q < 3; ++q) {
        double m = sm[q][0];
        for (int w = 1; w < 4; ++w) m = sm[q][w] > m ? sm[q][w] : m;
        v[q] = m;
    }
    maxsurv[s] = v[2];
    if (v[1] > v[0] + ln_sub) {      // lerr > ltol (fitting.py:798-799)
        k2state[s] += 1;
        atomicAdd(n_unconv, 1);
    } else {
        k2state[s] = -k2state[s] - 1;
    }
}

__global__ void k_sel_thresh(int nstar, const double *__restrict__ maxns_part,
                             const double *__restrict__ maxsurv, double ln_wt,
                             double *__restrict__ thr_sel) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstar) return;
    // maximum of the final lnprob plane: survivors (flux phase) and the rest
    double m = maxsurv[s];
    for (int c = 0; c < NCHUNK; ++c) {
        const double x = maxns_part[(int64_t)s * NCHUNK + c];
        m = x > m ? x : m;
    }
    thr_sel[s] = m + ln_wt;
}

// Emit the records of the selected models (ordered lists from k_cmp_scatter).
// Survivors of the cull are read from the flux-phase results; the others are
// re-derived from the grid (K1 sweeps + MLE), which is cheaper than having the
// full-grid scan write eleven planes.
// A work item is TILE consecutive entries of one star's list and belongs to ONE WAVE.
// The two kinds are interleaved in runs of 10-20 models, so the wave first sorts its
// entries by kind (ballot ranks, two position lists in its private LDS) and then runs
// dense rounds of 64 lanes of one kind each.  No workgroup barrier anywhere: the waves
// of a CU sit in different phases (index / tag loads, row gathers, float64 MLE, record
// stores) and cover each other's latency.  A record row is written by 64 lanes with
// gaps that another round of the same wave fills microseconds later (merged in L2).
template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, 2)
k_emit(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
       const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
       const double *__restrict__ thr_cull, const int32_t *__restrict__ sel_idx,
       const int64_t *__restrict__ sel_off, const int32_t *__restrict__ wbase,
       const ItemGeom *__restrict__ items, Planes pl, int64_t capacity,
       double *__restrict__ sel_vals, const float *__restrict__ surv32,
       const int64_t *__restrict__ cand_off) {
    constexpr int NW = TILE / 64;
    __shared__ double s_tbl[64];
    __shared__ int32_t s_idx[NW][TILE];
    __shared__ int32_t s_slot[NW][TILE];
    __shared__ int16_t s_ps[NW][TILE];      // list positions of the survivors, then ...
    __shared__ int16_t s_pd[NW][TILE];      // ... of the re-derived entries
    stage_exp_table(s_tbl);
    __syncthreads();
    const int nitem = wbase[NCHUNK * nstar];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t below = (1ull << lane) - 1ull;
    // geometry of a work item: star, first list position, live entries (0 past the end)
    auto geom = [&](int item, int &s, int64_t &q0, int &n) {
        s = 0;
        q0 = 0;
        n = 0;
        if (item < 0) return;
        const ItemGeom ig = items[item];
        s = ig.s;
        q0 = ig.q0;
        const int64_t room = capacity - q0;          // a record buffer too small: drop the rest
        n = (int)(room < ig.n ? room : ig.n);
        n = n > 0 ? n : 0;
    };
    // the entries' models / their kind words (survivor tag, see surv_tag; path 1: 1 or -0.)
    auto load_idx = [&](int64_t q0, int n, int32_t (&iv)[NW]) {
#pragma unroll
        for (int r = 0; r < NW; ++r) iv[r] = r * 64 + lane < n ? sel_idx[q0 + r * 64 + lane] : 0;
    };
    auto load_kind = [&](int s, int n, const int32_t (&iv)[NW], float (&kd)[NW]) {
        const int64_t sb = (int64_t)s * pl.nmodel;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            kd[r] = -0.f;
            if (r * 64 + lane < n) {
                if (surv32) kd[r] = surv32[sb + iv[r]];
                else kd[r] = pl.lnlp[sb + iv[r]] > thr_cull[s] ? surv_tag(0) : -0.f;
            }
        }
    };
    ItemWalk wk;
    wk.init(wbase, nstar, blockIdx.x & 7, (blockIdx.x >> 3) * NW + w,
            ((gridDim.x + 7 - (blockIdx.x & 7)) >> 3) * NW);
    int item = wk.item;
    int s, n, s_n, n_n;
    int64_t q0, q0_n;
    int32_t iv[NW], iv_n[NW];
    float kd[NW], kd_n[NW];
    geom(item, s, q0, n);
    load_idx(q0, n, iv);
    load_kind(s, n, iv, kd);
    for (; item >= 0; item = wk.item) {
        // the next item's models are requested now, its kind words once those have
        // arrived (after the copy rounds): both latencies run under this item's work
        wk.next();
        geom(wk.item, s_n, q0_n, n_n);
        load_idx(q0_n, n_n, iv_n);
        const StarPrep &sp = stars[s];
        const int64_t sbase = (int64_t)s * pl.nmodel;
        bool sv[NW];
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int t = r * 64 + lane;
            sv[r] = false;
            if (t < n) {
                sv[r] = surv_is(kd[r]);
                s_slot[w][t] = surv_slot(kd[r]);
                s_idx[w][t] = iv[r];
            }
        }
        int nS = 0, nD = 0;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            const int t = r * 64 + lane;
            const uint64_t bs = __ballot(sv[r]), bd = __ballot(t < n && !sv[r]);
            if (sv[r]) s_ps[w][nS + __popcll(bs & below)] = (int16_t)t;
            else if (t < n) s_pd[w][nD + __popcll(bd & below)] = (int16_t)t;
            nS += __popcll(bs);
            nD += __popcll(bd);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // Survivors and re-derived entries take turns, 64 of each (both lists are in position
        // order): the two kinds of stores into a stretch of the record planes then follow
        // each other closely and the half-written lines are still in L2 when their other
        // half arrives.
        const int nmax = nS > nD ? nS : nD;
        for (int k0 = 0; k0 < nmax; k0 += 64) {
        // ---- survivors: copy ---------------------------------------------------------
        if (const int k = k0 + lane; k < nS) {
            const int mp = s_ps[w][k];
            const int64_t o = surv32 ? cand_off[s] + s_slot[w][mp] : sbase + s_idx[w][mp];
            double rec[BRUTUS_NVALS];
            rec[0] = pl.lnl[o];
            rec[1] = pl.chi2[o];
            rec[2] = pl.scale[o];
            rec[3] = pl.av[o];
            rec[4] = RVF ? p.rv_mean : pl.rv[o];
#pragma unroll
            for (int v = 0; v < 6; ++v) rec[5 + v] = pl.icov[v][o];
#pragma unroll
            for (int v = 0; v < BRUTUS_NVALS; ++v) sel_vals[(int64_t)v * capacity + q0 + mp] = rec[v];
        }
        if (k0 == 0) load_kind(s_n, n_n, iv_n, kd_n);
        // ---- the rest: K1 sweeps + full MLE from the model's row -----------------------
        if (const int k = k0 + lane; k < nD) {
            const int mp = s_pd[w][k];
            const int64_t i = s_idx[w][mp];
            Coef<NB> c;
            gather_coef<NB>(grid, nmodel_pad, i, c);
            double F0[NB];
            compute_F0_tbl<NB>(c, s_tbl, F0);
            double av = p.av_mean, rv = p.rv_mean;
            const int K = k1[s];
            Mle m;
            if constexpr (RVF) {
                double R[NB];
                coef_R<NB>(c, rv, R);
                GramR G;
                gram_init_rf<NB>(c, R, sp, G);
                double a_, c_;
                if (K > 0) gram_sweep_rf(G, sp.S, p, av, a_, c_);       // one solve is exact (see k_fscan)
                mle_fast_rf<NB, true, true>(c, R, F0, sp, p, av, s_tbl, m);
            } else {
                Gram G;
                gram_init<NB>(c, sp, G);
                for (int kk = 0; kk < K; ++kk) {
                    double a_, b_, c_;
                    gram_sweep(G, sp.S, p, av, rv, a_, b_, c_);
                }
                mle_fast<NB, true>(c, F0, sp, p, av, rv, s_tbl, m);
            }
            double *out = sel_vals + q0 + mp;
            out[0] = final_lnl<RVF>(sp, p, m.chi2, false);
            out[(int64_t)1 * capacity] = m.chi2;
            out[(int64_t)2 * capacity] = m.scale;
            out[(int64_t)3 * capacity] = av;
            out[(int64_t)4 * capacity] = rv;
            out[(int64_t)5 * capacity] = m.i00;
            out[(int64_t)6 * capacity] = m.i01;
            out[(int64_t)7 * capacity] = m.i02;
            out[(int64_t)8 * capacity] = m.i11;
            out[(int64_t)9 * capacity] = m.i12;
            out[(int64_t)10 * capacity] = m.i22;
        }
        }
        if (nmax == 0) load_kind(s_n, n_n, iv_n, kd_n);
        // the next item's LDS writes must not pass this item's reads
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        s = s_n;
        n = n_n;
        q0 = q0_n;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
            iv[r] = iv_n[r];
            kd[r] = kd_n[r];
        }
    }
}

}  // namespace
