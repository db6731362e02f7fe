// fit_kernels.hpp -- float64 building blocks of brutus_fit_batch: Gram-form magnitude
// sweeps, the MLE, the exact K1 probe, ordered compaction, the flux phase on the candidate
// lists (k_fflux), the record index (k_rec_index) and the derived records (k_derive).
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, fit2_kernels, cluster_kernels,
// mt_kernels, post_kernels, offsets_kernels); everything lives in that unit's anonymous
// namespace.  The pipeline that strings these kernels together is described at the top of
// fit2_kernels.hpp.
#pragma once

namespace {

// The magnitude phase is a weighted linear least-squares problem in
// (offset, Av, Av*Rv) for every (star, model).  Instead of carrying the Nb
// residuals through the sweeps as the reference does, the hot path forms the
// ten weighted inner products of {1, r0, dr, y = mag_obs - mag_model} once and
// runs every sweep (fitting.py:176-243) on those scalars: the update formulas
// are algebraically identical, the results agree to rounding (~1e-14), and a
// sweep costs ~45 flops instead of ~14*Nb.

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
// LV: how much of it the caller uses -- 0: scale, chi2, i00 and the Av step sums (a_num, a_ss);
// 1: + the Rv step sums (r_num, r_ss); 2: everything (the precision matrix's mixed terms are
// only ever read from the LAST evaluation of a model, the one that is stored).
template <int NB, bool TBL, int LV = 2>
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
    // fitting.py:526-561 with the constant factors taken out of the band sums: with
    // u = R F, v = dR F (unscaled), res = d - s F, t = s F - res = 2 s F - d,
    //   sa_mix = c sum u t / V           sr_mix = c sum v t / V          (c = -0.4 ln 10)
    //   a_den = (c s)^2 sum u^2 / V      r_den = (c s)^2 sum v^2 / V
    //   a_num = c s sum u res / V        r_num = c s sum v res / V
    //   ar_mix = c s sum v (s (F - F0) - res) / V = c s sum v (t - s F0) / V
    // 16 operations per band instead of 24; same quantities to rounding (~1e-16).
    double SA = 0., SR = 0., AR = 0., AA = 0., RR = 0., AN = 0., RN = 0., chi2 = 0.;
    const double s2 = s + s;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double iv = sp.iV[j];
        const double D0 = (double)c.dr[j];
        const double R0 = (double)c.r0[j] + rv * D0;
        const double u = R0 * F[j], v = D0 * F[j];
        const double res = fma(-s, F[j], sp.d[j]);
        const double rw = res * iv, uw = u * iv, vw = v * iv;
        chi2 = fma(res, rw, chi2);
        AN = fma(u, rw, AN);
        AA = fma(u, uw, AA);
        if (LV >= 1) {
            RN = fma(v, rw, RN);
            RR = fma(v, vw, RR);
        }
        if (LV >= 2) {
            const double t = fma(s2, F[j], -sp.d[j]);
            const double q = fma(-s, F0[j], t);
            SA = fma(uw, t, SA);
            SR = fma(vw, t, SR);
            AR = fma(vw, q, AR);
        }
    }
    const double cs = fac * s, cs2 = cs * cs;
    double a_den = cs2 * AA, r_den = cs2 * RR;
    const double a_num = cs * AN, r_num = cs * RN;
    const double sa_mix = fac * SA, sr_mix = fac * SR, ar_mix = cs * AR;
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

// (Round 4 tried the MLE in ONE pass over the bands -- moments of (d, F) that do not depend on
// the scale, chi2 = D2 - s (2 S1 - s S2) etc.: 8 % fewer float64 instructions, no flux array
// held for a second loop -- and dropped it: no measurable gain (k_fflux 1.63 vs 1.62-1.68 ms),
// and the moments cancel against D2 = sum (S/N)^2, which costs four-band stars with
// chi2 ~ 1e-3 six digits of ln chi2 (3e-9 relative on lnl instead of 1e-13).)

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

// mle_fast with R given.  LV < 2 leaves out the Rv sums and the mixed terms (i01, i02, i12,
// i22, r_num, r_ss), which only the reported precision matrix needs.
template <int NB, bool TBL, int LV>
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
    // (sums with the constant factors taken out, see mle_fast)
    double SA = 0., SR = 0., AR = 0., AA = 0., RR = 0., AN = 0., RN = 0., chi2 = 0.;
    const double s2 = s + s;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const double iv = sp.iV[j];
        const double u = R[j] * F[j];
        const double res = fma(-s, F[j], sp.d[j]);
        const double rw = res * iv, uw = u * iv;
        chi2 = fma(res, rw, chi2);
        AN = fma(u, rw, AN);
        AA = fma(u, uw, AA);
        if (LV >= 2) {
            const double t = fma(s2, F[j], -sp.d[j]);
            const double v = (double)c.dr[j] * F[j];
            const double q = fma(-s, F0[j], t);
            const double vw = v * iv;
            SA = fma(uw, t, SA);
            RN = fma(v, rw, RN);
            RR = fma(v, vw, RR);
            SR = fma(vw, t, SR);
            AR = fma(vw, q, AR);
        }
    }
    const double cs = fac * s, cs2 = cs * cs;
    double a_den = cs2 * AA, r_den = cs2 * RR;
    const double a_num = cs * AN, r_num = cs * RN;
    const double sa_mix = fac * SA, sr_mix = fac * SR, ar_mix = cs * AR;
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

// ---- more than 16 bands: the per-band arrays do not fit the register file ----
// With F0, R and F held per lane (2 * NB registers each) the list kernels, built for two
// workgroups per CU (256 registers), spill from 24 bands up: k_fflux<24> 256 VGPRs + 700-1000
// bytes of scratch, 10.4 ms per 128 stars against 2.2 ms at 16 bands.  Two remedies, measured
// per 128 stars on configs[1] / configs[2] (round 4, same box):
//  (1) ONE workgroup per CU -- a wave per SIMD, 512 registers: what does not fit the 256
//      architectural ones sits in accumulation registers, a one-instruction move away,
//      instead of scratch memory;
//  (2) the WIDE form, which holds only the model's coefficients (3 * NB float32): the flux
//      F_j = 10^(-0.4 (m_j + av R_j)) comes from ONE exponential of the whole magnitude (no F0
//      array), R_j = r0_j + rv dr_j is recomputed where it is used (one fma), and the fluxes go
//      through a lane-private column of LDS (`Fc[j * TILE]`: NB * 2 KiB per workgroup)
//      between the two passes over the bands.  The accesses are volatile: the compiler
//      otherwise forwards the stores to the loads and is back to NB live registers.  LV 2
//      costs one more exponential per band (F0 for ar_mix).
//   24 bands  k_fflux  narrow x2 10.4 / 4.12   wide x2 2.91 / 1.69   wide x1 3.55 / 2.02   narrow x1 2.66 / 1.75
//             k_derive narrow x2  6.5 / 4.35   wide x2 4.03 / 2.96   wide x1 2.69 / 2.21   narrow x1 2.56 / 2.04
//   32 bands  k_fflux  narrow x2 16.0 / 6.66   wide x2 4.87 / 2.03   wide x1 3.20 / 1.67   narrow x1 4.86 / 4.84
//             k_derive narrow x2 14.7 / 8.53   wide x2 9.48 / 5.77   wide x1 3.76 / 2.88   narrow x1 4.87 / 2.71
// hence: 24 bands narrow, 32 bands wide, both at one workgroup per CU (configs[1] / [2] at
// 24 bands 6.6k / 11.1k -> 16.8k / 18.9k stars/s, at 32 bands 3.6k / 5.8k -> 12.1k / 12.9k).
#ifndef BRUTUS_WIDE_FROM
#define BRUTUS_WIDE_FROM 25
#endif
constexpr bool wide_bands(int nb) { return nb >= BRUTUS_WIDE_FROM; }
#ifndef BRUTUS_LIST_OCC1_FROM
#define BRUTUS_LIST_OCC1_FROM 17
#endif
constexpr int list_waves(int nb) { return nb >= BRUTUS_LIST_OCC1_FROM ? 1 : 2; }
// (the opening flux kernel on its own: at 16 bands it is the one list kernel that does not fit
// 256 registers -- 52 / 148 bytes of scratch at two workgroups per CU)
#ifndef BRUTUS_FFLUX_OCC1_FROM
#define BRUTUS_FFLUX_OCC1_FROM BRUTUS_LIST_OCC1_FROM
#endif
constexpr int fflux_waves(int nb, bool first) {
    return nb >= (first ? BRUTUS_FFLUX_OCC1_FROM : BRUTUS_LIST_OCC1_FROM) ? 1 : 2;
}
// (the tile kernels of fit2_kernels.hpp likewise: k_sel_band from 24 bands -- 0.16 -> 0.10,
// 0.27 -> 0.13 ms --, k_top at 32 -- 2.19 -> 1.27 ms; k_top<24> is faster with two: 0.68
// against 0.87 ms)
constexpr int top_waves(int nb) { return nb > 24 ? 1 : 2; }
constexpr int band_waves(int nb) { return nb > 16 ? 1 : 2; }
// (an LDS pointer by address space: a volatile access through a generic pointer is a FLAT one)
typedef __attribute__((address_space(3))) volatile double *LdsColumn;
// float32 -> float64 where it is used, every time: left to itself the compiler converts the
// 3 * NB coefficients once and keeps them as doubles -- 6 * NB registers.  The conversion is
// an asm with a second, unused operand `after`: conversions with different `after` are
// different values, so nothing is merged across passes.  `after` can also hold the
// scheduler back -- a running sum as of the end of the previous group of WIDE_GROUP bands:
// a group's arithmetic cannot start before the previous group's sums exist -- which pays at two
// workgroups per CU and costs at one, where the wave is alone on its SIMD and wants every
// independent chain it can get (32 bands, k_fflux: groups of 4 3.70 / 1.87 ms, of 8 3.42 / 1.87,
// none 3.20 / 1.67): the default group is the whole band loop.  (A scheduling barrier instead
// pins only itself and the other ordered operations; the arithmetic floats around it.)
#ifndef BRUTUS_WIDE_GROUP
#define BRUTUS_WIDE_GROUP 32
#endif
__device__ __forceinline__ double wide_f64(float x, double after) {
    double r;
    asm("v_cvt_f64_f32_e32 %0, %1" : "=v"(r) : "v"(x), "v"(after));
    return r;
}
#define WIDE_ANCHOR(var, sum) if (j % BRUTUS_WIDE_GROUP == 0 && j) var = (sum)

template <int NB, int LV>
__device__ __forceinline__ void mle_wide(const Coef<NB> &c, const StarPrep &sp,
                                         const DevParams &p, double av, double rv,
                                         const double *__restrict__ tbl, LdsColumn Fc,
                                         Mle &o) {
    const double fac = -0.92103403719761827361;
    const double mav = -0.4 * av;
    double s_num = 0., s_den = 0.;
    double after = mav;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        WIDE_ANCHOR(after, s_den);
        const double R = fma(rv, wide_f64(c.dr[j], after), wide_f64(c.r0[j], after));
        const double f = fast_exp10(fma(mav, R, -0.4 * wide_f64(c.m[j], after)), tbl);
        Fc[j * TILE] = f;
        const double fw = f * sp.iV[j];
        s_num += sp.d[j] * fw;
        s_den += f * fw;
    }
    double s = s_num / s_den;
    if (s <= 1e-20) s = 1e-20;
    double SA = 0., SR = 0., AR = 0., AA = 0., RR = 0., AN = 0., RN = 0., chi2 = 0.;
    const double s2 = s + s;
    after = s;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        WIDE_ANCHOR(after, chi2);
        const double iv = sp.iV[j];
        const double f = Fc[j * TILE];
        const double D0 = wide_f64(c.dr[j], after);
        const double u = fma(rv, D0, wide_f64(c.r0[j], after)) * f;
        const double res = fma(-s, f, sp.d[j]);
        const double rw = res * iv, uw = u * iv;
        chi2 = fma(res, rw, chi2);
        AN = fma(u, rw, AN);
        AA = fma(u, uw, AA);
        if (LV >= 1) {
            const double v = D0 * f;
            const double vw = v * iv;
            RN = fma(v, rw, RN);
            RR = fma(v, vw, RR);
            if (LV >= 2) {
                const double t = fma(s2, f, -sp.d[j]);
                const double q = fma(-s, fast_exp10(-0.4 * wide_f64(c.m[j], after), tbl), t);
                SA = fma(uw, t, SA);
                SR = fma(vw, t, SR);
                AR = fma(vw, q, AR);
            }
        }
    }
    const double cs = fac * s, cs2 = cs * cs;
    double a_den = cs2 * AA, r_den = cs2 * RR;
    o.a_ss = a_den;
    o.r_ss = r_den;
    o.a_num = cs * AN;
    o.r_num = cs * RN;
    a_den += p.av_ivar;
    r_den += p.rv_ivar;
    a_den += p.a_reg;
    r_den += p.r_reg;
    o.scale = s;
    o.chi2 = chi2;
    o.i00 = s_den;
    o.i01 = fac * SA;
    o.i02 = fac * SR;
    o.i11 = a_den;
    o.i12 = cs * AR;
    o.i22 = r_den;
}

// gram_init_rf with R recomputed per band (no R array)
template <int NB>
__device__ __forceinline__ void gram_init_rf_wide(const Coef<NB> &c, double rv,
                                                  const StarPrep &sp, GramR &G) {
    double uR = 0., RR = 0., yR = 0., uy = 0., yy = 0.;
    double after = rv;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        WIDE_ANCHOR(after, yy);
        const double w = sp.iW[j];
        const double R = fma(rv, wide_f64(c.dr[j], after), wide_f64(c.r0[j], after));
        const double y = sp.g[j] - wide_f64(c.m[j], after);
        const double Rw = R * w, yw = y * w;
        uR += Rw;
        RR += R * Rw;
        yR += R * yw;
        uy += yw;
        yy += y * yw;
    }
    G.uR = uR; G.RR = RR; G.yR = yR; G.uy = uy; G.yy = yy;
}

// The MLE of the list kernels in whichever form the band count calls for.  NR / NF: lengths of
// the caller's R and F0 arrays (1 where the form at hand does not keep them).
// FULL: the evaluation whose results are stored; otherwise only what the next step needs.
template <int NB, bool RVF, bool FULL, int NR, int NF>
__device__ __forceinline__ void mle_list(const Coef<NB> &c, const double (&R)[NR],
                                         const double (&F0)[NF], LdsColumn Fc,
                                         const StarPrep &sp, const DevParams &p, double av,
                                         double rv, const double *__restrict__ tbl, Mle &o) {
    constexpr int LV = FULL ? 2 : RVF ? 0 : 1;
    if constexpr (wide_bands(NB)) mle_wide<NB, LV>(c, sp, p, av, rv, tbl, Fc, o);
    else if constexpr (RVF) mle_fast_rf<NB, true, LV>(c, R, F0, sp, p, av, tbl, o);
    else mle_fast<NB, true, LV>(c, F0, sp, p, av, rv, tbl, o);
}

// F0 of model i from the band-major table (coalesced) / of one model from its row.
template <int NB>
__device__ __forceinline__ void load_F0(const float *__restrict__ grid, int64_t nmodel_pad,
                                        int64_t i, double (&F0)[NB]) {
    const double *t = reinterpret_cast<const double *>(grid + (int64_t)6 * NB * nmodel_pad);
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = t[(int64_t)j * nmodel_pad + i];
}
// F0 with the table-free polynomial (<= 1 ulp from the tabulated value): for kernels where
// registers, not issue slots, are the limit (none of the hot kernels any more).
template <int NB>
__device__ __forceinline__ void compute_F0_fast(const Coef<NB> &c, double (&F0)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j) F0[j] = poly_exp10(-0.4 * (double)c.m[j]);
}
// ... or with the table-driven form where registers allow: bit-identical to the
// tabulated F0 the scans read
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

// Exact float64 probe of the number of magnitude sweeps K1 for the stars whose float32
// statistics (k_pre32) could not decide it.  One star per blockIdx.y, up to KS sweeps of
// fitting.py:176-243 on the Gram scalars; per (block.x, star) the maxima
//   part[(bx * nstar + star) * 2 KS + 2k], [+ 2k + 1] = L_k = max logwt,
//                                                       T_k = max{logwt : step >= mtol}
// that k_k1_decide turns into K1 (fitting.py:246-264).  Nothing else is computed: the
// statistics of the state after K1 sweeps come from a k_pre32 re-run.
template <int NB, int KS, bool RVF>
__global__ void __launch_bounds__(TILE)
k_k1probe(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
          const int32_t *__restrict__ star_ids, const StarPrep *__restrict__ stars, DevParams p,
          int tiles_per_block, int ntile, double *__restrict__ part,
          const int32_t *__restrict__ nrun_dev) {
    constexpr int NV = 2 * KS;
    __shared__ double slot[4];
    // (nrun_dev: the list and its length were put together on the device; row y of the launch
    // takes the entries y, y + gridDim.y, ...)
    if (nrun_dev) nrun = *nrun_dev;
    for (int y = blockIdx.y; y < nrun; y += gridDim.y) {
    const int s = star_ids[y];
    const StarPrep &sp = stars[s];
    double mx[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) mx[v] = -INFINITY;
    const int t0 = blockIdx.x * tiles_per_block;
    const int t1 = min(ntile, t0 + tiles_per_block);
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        const bool live = i < nmodel;
        Coef<NB> c;
        load_coef<NB>(grid, nmodel_pad, i, c);
        double av = p.av_mean, rv = p.rv_mean;
        if constexpr (RVF) {
            // With Rv pinned the objective is exactly quadratic in (offset, Av) and a
            // sweep is its Newton step with the offset eliminated: the first sweep lands
            // on the (clamped) minimiser, every later one moves by rounding noise
            // (<< mtol) and leaves logwt unchanged.  So one sweep is computed; sweeps
            // 2..KS only enter the statistics (L_k = L_1, no step above tolerance),
            // which makes K1 <= 2 as in the reference.
            double R[NB];
            coef_R<NB>(c, p.rv_mean, R);
            GramR G;
            gram_init_rf<NB>(c, R, sp, G);
            double dav, lw;
            gram_sweep_rf(G, sp.S, p, av, dav, lw);
            if (live && lw == lw) {
                mx[0] = lw > mx[0] ? lw : mx[0];
                if (fabs(dav) >= p.mtol) mx[1] = lw > mx[1] ? lw : mx[1];
#pragma unroll
                for (int k = 1; k < KS; ++k) mx[2 * k] = lw > mx[2 * k] ? lw : mx[2 * k];
            }
        } else {
            Gram G;
            gram_init<NB>(c, sp, G);
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                double dav, drv, lw;
                gram_sweep(G, sp.S, p, av, rv, dav, drv, lw);
                if (live && lw == lw) {
                    const bool big = (fabs(dav) >= p.mtol) || (fabs(drv) >= p.mtol);
                    mx[2 * k] = lw > mx[2 * k] ? lw : mx[2 * k];
                    if (big) mx[2 * k + 1] = lw > mx[2 * k + 1] ? lw : mx[2 * k + 1];
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
        block_max_store(mx[v], slot, part + ((int64_t)blockIdx.x * nstar + s) * NV + v);
    }
}

// K1 of the probed stars from the partials of k_k1probe (0 = not converged in KS sweeps).
// With `ctr` given (device-driven call): kfix[s] = K1; a general-Rv star whose K1 is not the 2
// the float32 planes were computed with joins redo_ids (count ctr[1]); K1 = 0 (more than KS
// sweeps) or K1 > max_iter raises ctr[3]: the host repeats the batch along its own path.
__global__ void k_k1_decide(int nblkx, int nstar, const int32_t *__restrict__ star_ids, int KS,
                            const double *__restrict__ part, double ln_init,
                            int32_t *__restrict__ k1, const int32_t *nrun_dev, int32_t *kfix,
                            int32_t *ctr, int32_t *__restrict__ redo_ids, int rvf, int max_iter) {
    __shared__ double sm[KCAP * 2][4];
    if (nrun_dev && (int)blockIdx.x >= *nrun_dev) return;
    const int s = star_ids[blockIdx.x];
    const int NV = 2 * KS;
    double v[KCAP * 2];
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
    int K1 = 0;
    for (int k = 0; k < KS; ++k) {
        double L = sm[2 * k][0], T = sm[2 * k + 1][0];
        for (int w = 1; w < 4; ++w) {
            L = sm[2 * k][w] > L ? sm[2 * k][w] : L;
            T = sm[2 * k + 1][w] > T ? sm[2 * k + 1][w] : T;
        }
        L = L > -BIG ? L : -BIG;
        if (!(T > L + ln_init)) {
            K1 = k + 1;
            break;
        }
    }
    k1[s] = K1;
    if (ctr) {
        if (K1 == 0 || K1 > max_iter) {
            atomicOr(ctr + 3, 1);
        } else {
            kfix[s] = K1;
            if (!rvf && K1 != 2) redo_ids[atomicAdd(ctr + 1, 1)] = s;
        }
    }
}

// Exclusive scans of per-(star, chunk) counts, one workgroup per job (blockIdx.x):
//   offsets[(s, c)] in (s, c) order, star_off[s] (star_off[nstar] = total) and, where wbase
//   is given, the work items of a list kernel (k_fflux, k_derive): the segment of star s in
//   model chunk c, cut into pieces of TILE entries, numbered CHUNK-MAJOR:
//   wbase[c * nstar + s] = first item of that segment, wbase[NCHUNK * nstar] = #items.
// Workgroups that run at the same time then work on the same 1/NCHUNK of the grid for
// different stars, and the coefficient rows they gather are L2 hits instead of one fabric
// read per star.  BRUTUS_MAX_BATCH threads, two exclusive scans over the nstar x NCHUNK counts
// -- star-major for the list offsets, chunk-major for the work items -- each lane taking a
// run of consecutive entries whose loads are all in flight at once.
struct OffsetsJob {
    const int64_t *counts;
    int64_t *offsets, *star_off;
    int32_t *wbase;
    int64_t *total_out;      // (optional) the grand total once more, where the host collects its results
};
// (Round 5: 1024 threads and one workgroup per SCAN -- blockIdx.x = 2 * job + scan -- instead of
// 256 threads running a job's two scans one after the other with 32-64 entries per lane:
// 30 -> ~8 us per launch; the work-item scan reads only the counts, not the first scan's output.)
constexpr int OFF_T = 1024;
__global__ void __launch_bounds__(OFF_T)
k_offsets(int nstar, OffsetsJob job0, OffsetsJob job1) {
    constexpr int NT = OFF_T;
    constexpr int EPT = (BRUTUS_MAX_BATCH * NCHUNK + NT - 1) / NT;      // entries per lane at the full batch
    __shared__ int64_t s_slot64[NT / 64 + 1];
    __shared__ int32_t s_slot32[NT / 64 + 1];
    const OffsetsJob job = (blockIdx.x >> 1) == 0 ? job0 : job1;
    const bool items = (blockIdx.x & 1) != 0;
    const int64_t *__restrict__ counts = job.counts;
    const int total = nstar * NCHUNK;
    const int ept = (total + NT - 1) / NT;
    const int e0 = threadIdx.x * ept;
    if (!items) {
        int64_t r[EPT];
        int64_t sum = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = e0 + k;
            const int64_t v = counts[e < total ? e : total - 1];      // (clamped address + select)
            r[k] = k < ept && e < total ? v : 0;
            sum += r[k];
        }
        int64_t all;
        int64_t pre = block_exclusive_sum<int64_t, NT>(sum, s_slot64, all);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if (k < ept && e0 + k < total) {
                job.offsets[e0 + k] = pre;
                if ((e0 + k) % NCHUNK == 0) job.star_off[(e0 + k) / NCHUNK] = pre;
            }
            pre += r[k];
        }
        if (threadIdx.x == 0) {
            job.star_off[nstar] = all;
            if (job.total_out) *job.total_out = all;
        }
        return;
    }
    if (!job.wbase) return;
    {
        int32_t r[EPT];
        int32_t sum = 0;
        int c = e0 / nstar, q = e0 - c * nstar;      // e = c * nstar + q, stepped without divisions
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = e0 + k;
            const bool in = k < ept && e < total;
            const int64_t v = counts[in ? (int64_t)q * NCHUNK + c : 0];
            r[k] = in ? (int32_t)((v + TILE - 1) / TILE) : 0;
            sum += r[k];
            if (++q == nstar) {
                q = 0;
                ++c;
            }
        }
        int32_t all;
        int32_t pre = block_exclusive_sum<int32_t, NT>(sum, s_slot32, all);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            if (k < ept && e0 + k < total) job.wbase[e0 + k] = pre;
            pre += r[k];
        }
        if (threadIdx.x == 0) job.wbase[total] = all;
    }
}

// Ordered list {i : bit i of the star's membership words is set} (= np.where order).
// A chunk's membership words are fetched 64 tiles (256 words, one per lane) at a time and
// prefix-summed once; the per-tile loop then runs out of LDS.  (One dependent global load
// per tile made this kernel latency-bound: 46 round trips per workgroup.)
__global__ void __launch_bounds__(TILE)
k_cmp_scatter(int64_t nmodel, int ntile, const unsigned long long *__restrict__ mask,
              const int64_t *__restrict__ offsets, int64_t capacity,
              int32_t *__restrict__ out_idx) {
    __shared__ int s_slot[TILE / 64 + 1];
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
        int tot;
        const int pre = block_exclusive_sum<int, TILE>(__popcll(word), s_slot, tot);
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

// The float32 lnprob~ plane doubles as the survivor map.  k_fflux overwrites a survivor's entry
// (its value is not needed again: survivors are judged by their final float64 lnprob) with
// the bit pattern 1 + (position in the star's candidate list) -- as a float a positive number
// below 2^-100 --, and leaves a candidate that failed the exact cull test alone.  Every later
// pass (exact first-cut threshold, classification) then reads ONE plane.  A genuine entry is
// never such a number: the float32 passes store +0 for any |lnprob~| < 2^-100 (surv_clean).
// (Until round 5 a tag was "any positive finite word", on the assumption that a log-density is
// negative -- but the scale-space parallax term contributes -ln(2 pi var)/2, which is +3 for a
// parallax at S/N 10: positive entries were read as tags, and k_sel_classify gathered from
// wherever they pointed.  Found by the sharp-posterior block of the bench.)
constexpr int SURV_TAG_END = 27 << 23;            // bit pattern of 2^-100: list positions < 2.2e8
__device__ __forceinline__ float surv_tag(int64_t slot) { return __int_as_float((int)slot + 1); }
__device__ __forceinline__ bool surv_is(float x) {
    const int b = __float_as_int(x);
    return b > 0 && b < SURV_TAG_END;
}
__device__ __forceinline__ float surv_clean(float x) { return fabsf(x) < 0x1p-100f ? 0.f : x; }
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

// Walk of a continuation launch of k_fflux: only the segments of the stars still iterating
// (`act`, nact of them: a per cent of a batch) -- workgroup (chunk c, active star a, piece p)
// takes the items p, p + CONT_P, ... of segment (act[a], c).  (Walking all items and skipping
// the finished stars' cost 0.18 ms per launch: one dependent load chain per item.)
constexpr int CONT_P = 16;
struct SegWalk {
    int item, end;
    __device__ __forceinline__ void init(const int32_t *wbase, int nstar, const int32_t *act, int nact,
                                         int unit) {
        const int c = unit % NCHUNK, a = (unit / NCHUNK) % nact;
        const int piece = unit / (NCHUNK * nact);
        const int e = c * nstar + act[a];
        end = wbase[e + 1];
        item = wbase[e] + piece;
        if (item >= end) item = -1;
    }
    __device__ __forceinline__ bool done() const { return item < 0; }
    __device__ __forceinline__ void next() {
        item += CONT_P;
        if (item >= end) item = -1;
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

// The same row kept RAW in registers (3*NB/4 float4): requested one work item ahead and turned
// into a Coef when its item comes up, so the gather's latency lies under the previous item's
// arithmetic (k_derive).
typedef float f32x4 __attribute__((ext_vector_type(4)));     // (a native vector: an asm operand)
template <int NB>
struct RowRegs {
    f32x4 q[3 * NB / 4];
};
template <int NB>
__device__ __forceinline__ void gather_row(const float *__restrict__ grid, int64_t nmodel_pad,
                                           int64_t i, RowRegs<NB> &r) {
    const f32x4 *row =
        reinterpret_cast<const f32x4 *>(grid + (int64_t)3 * NB * nmodel_pad + i * (3 * NB));
#pragma unroll
    for (int q = 0; q < 3 * NB / 4; ++q) r.q[q] = row[q];
}
template <int NB>
__device__ __forceinline__ void row_to_coef(const RowRegs<NB> &r, Coef<NB> &c) {
    float t[3 * NB];
#pragma unroll
    for (int q = 0; q < 3 * NB / 4; ++q) {
        t[4 * q] = r.q[q].x;
        t[4 * q + 1] = r.q[q].y;
        t[4 * q + 2] = r.q[q].z;
        t[4 * q + 3] = r.q[q].w;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        c.m[j] = t[3 * j];
        c.r0[j] = t[3 * j + 1];
        c.dr[j] = t[3 * j + 2];
    }
}
// "These loads have to have landed HERE": an empty asm that takes the registers as operands
// and orders memory operations.  gfx9 counts loads and stores in ONE counter (vmcnt) and the
// compiler, once both kinds are pending, can only wait for everything -- so a wait for a
// prefetched load placed BEHIND a batch of stores also waits for the stores' write
// acknowledgements (a full memory round trip per work item).  Placed in FRONT of the
// stores it finds only loads that were issued a whole work item ago.
template <int NB>
__device__ __forceinline__ void landed(RowRegs<NB> &r) {
#pragma unroll
    for (int q = 0; q < 3 * NB / 4; ++q) {
        f32x4 t = r.q[q];
        asm volatile("" : "+v"(t) : : "memory");
        r.q[q] = t;
    }
}
__device__ __forceinline__ void landed(int32_t &x) {
    int32_t t = x;
    asm volatile("" : "+v"(t) : : "memory");
    x = t;
}

// The caller's record value planes (BRUTUS_NVALS x cap float64; include/brutus_amd.h):
//   0 lnlike, 1 chi2, 2 scale, 3 av, 4 rv, 5..10 icov[00, 01, 02, 11, 12, 22].
// Slots [0, ncand) belong to the candidates of the cull in candidate-list order -- the flux
// phase writes its results there and they ARE the survivors' records (nothing copies them
// again) --, slots [ncand, ncand + nder) to the selected models the cull dropped (k_derive).
struct RecPlanes {
    double *vals;
    int64_t cap;
    __device__ __forceinline__ double *plane(int v) const { return vals + (int64_t)v * cap; }
};

// Flux phase on the ordered candidate lists (fitting.py:758-803), persistent workgroups
// looping over work items.  The list holds the CANDIDATES (lnl_p~ >= threshold - eps).
// First launch: rebuild (av, rv) from K1 sweeps, exact cull test lnl_p > thr_cull[s]
// (fitting.py:758-759) whose outcome is marked in the float32 plane (survivor tag / left
// alone, see surv_tag), two iterations from lnl_old = -1e300 for the survivors;
// continuation: one iteration from the stored state.  A survivor's results go to the
// record planes at its LIST POSITION q (full-line writes), its step size and final
// first-cut statistic lnprob to the workspace arrays step_st / lnprob_st at the same
// position.  Per work item and wave: L = max lnl_new, T = max{lnl_new : |dlnl| > ltol},
// M = max final lnprob.  Entries at or beyond the record capacity are skipped (the host
// sees ncand > capacity and reports BRUTUS_ENOMEM).
template <int NB, bool RVF, bool FIRST>
__global__ void __launch_bounds__(TILE, fflux_waves(NB, FIRST))
k_fflux(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
        const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
        const int32_t *__restrict__ k2state, const int32_t *__restrict__ cand_idx,
        const int64_t *__restrict__ cand_off, const int32_t *__restrict__ wbase,
        const ItemGeom *__restrict__ items, RecPlanes rec, double *__restrict__ step_st,
        double *__restrict__ lnprob_st, double *__restrict__ part, float *__restrict__ surv32,
        const double *__restrict__ thr_cull, const int32_t *__restrict__ act, int nact,
        const int32_t *__restrict__ nact_dev) {
    // continuation launches: the list of stars still iterating and its length may have been
    // put together on the device (k_fflux_decide); the launch has a fixed size and its
    // workgroups share the NCHUNK * nact * CONT_P segment pieces in turns
    if (!FIRST && nact_dev) nact = *nact_dev;
    if (!FIRST && nact <= 0) return;
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    constexpr bool WIDE = wide_bands(NB);
    __shared__ double s_F[WIDE ? NB * TILE : 1];        // the lanes' flux columns (mle_wide)
    const LdsColumn Fc = (LdsColumn)s_F + (WIDE ? threadIdx.x : 0);
    // (the launch kind is a template parameter: the opening launch then carries no
    // state-reload path and its two iterations unroll)
    constexpr int niter = FIRST ? 2 : 1;
    // this lane's model in a work item, requested one item ahead (a dead lane reads the
    // item's last entry: no select on the loaded value, so nothing waits for it here)
    auto lane_model = [&](int item) -> int32_t {
        const ItemGeom ig = items[item < 0 ? 0 : item];      // (past the end: item 0 again, unused)
        const int64_t q = ig.q0 + threadIdx.x;
        const int64_t last = ig.q0 + ig.n - 1;
        return cand_idx[q < last ? q : last];
    };
    double *__restrict__ r_lnl = rec.plane(0), *__restrict__ r_chi2 = rec.plane(1),
                        *__restrict__ r_scale = rec.plane(2), *__restrict__ r_av = rec.plane(3),
                        *__restrict__ r_rv = rec.plane(4);
    // (the walk as a generic lambda: the opening launch instantiates it once, without the
    // loop over work units around it -- with the loop in one body the opening kernel's
    // register allocation went from 248 VGPRs to 256 + 180 bytes of scratch, 1.65 -> 2.2 ms)
    auto walk = [&](auto &wk) {
    int32_t i_nxt = lane_model(wk.item);
    while (!wk.done()) {
        const int item = wk.item;
        wk.next();
        const ItemGeom ig = items[item];
        const int s = ig.s;
        const int32_t i_me = i_nxt;
        i_nxt = lane_model(wk.item);
        if (k2state[s] < 0) {
            landed(i_nxt);      // (every path into the loop head has taken it: no wait there)
            continue;
        }
        const StarPrep &sp = stars[s];
        const int64_t q = ig.q0 + threadIdx.x;       // list position = record slot
        const bool live = (int)threadIdx.x < ig.n && q < rec.cap;
        double L = -INFINITY, T = -INFINITY, M = -INFINITY;
        bool go = live;
        const int64_t i = i_me;
        const int64_t o = (int64_t)s * nmodel + i;
        if (!FIRST && live) go = surv_is(surv32[o]);
        Mle m;
        double o_lnl = 0., o_lnprob = 0., o_av = 0., o_rv = 0., o_step = 0.;
        bool store = false;
        if (go) {
            Coef<NB> c;
            gather_coef<NB>(grid, nmodel_pad, i, c);
            double F0[WIDE ? 1 : NB];
            // (table-driven in both instantiations: the general kernel ran the table-free
            // degree-13 form until round 3 -- 19 instead of 15 operations per exponential --
            // for the sake of registers it turned out not to need: 215 -> 223 VGPRs, still
            // two waves per SIMD, k_fflux 1.12 -> 1.00 ms per 128 stars on configs[2])
#ifdef BRUTUS_F0_FROM_TABLE        // (A/B build, profiles/r06_f0_table_ab.txt: 12 gathers instead of 12 exponentials)
            if constexpr (!WIDE) load_F0<NB>(grid, nmodel_pad, i, F0);
#else
            if constexpr (!WIDE) compute_F0_tbl<NB>(c, s_tbl, F0);
#endif
            double av, rv, step, lnl_old;
            double R[RVF && !WIDE ? NB : 1];
            if constexpr (RVF && !WIDE) coef_R<NB>(c, p.rv_mean, R);
            if constexpr (FIRST) {
                av = p.av_mean;
                rv = p.rv_mean;
                const int K = k1[s];
                if constexpr (RVF) {
                    GramR G;
                    if constexpr (WIDE) gram_init_rf_wide<NB>(c, p.rv_mean, sp, G);
                    else gram_init_rf<NB>(c, R, sp, G);
                    // (a single sweep would do, see k_k1probe; the loop form keeps this
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
                av = r_av[q];
                rv = RVF ? p.rv_mean : r_rv[q];       // pinned Rv is not stored
                step = step_st[q];
                lnl_old = -0.5 * r_chi2[q];
            }
            mle_list<NB, RVF, false>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
            if constexpr (FIRST) {
                go = cull_stat(sp, m) > thr_cull[s];
                if (go) surv32[o] = surv_tag(q - cand_off[s]);     // a failed candidate keeps its lnprob~
            }
            double lnl_new = lnl_old, dl = 0.;
            for (int it = 0; go && it < niter; ++it) {
                double dav = (m.a_num + (p.av_mean - av) * p.av_ivar) / (m.a_ss + p.av_ivar) * step;
                if (dav < p.avmin - av) dav = p.avmin - av;
                if (dav > p.avmax - av) dav = p.avmax - av;
                av += dav;
                if constexpr (RVF) {
                    // the Rv step is clamped to zero; only the stored MLE needs the Rv sums
                    if (it + 1 < niter) mle_list<NB, RVF, false>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
                    else mle_list<NB, RVF, true>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
                } else {
                    double drv = (m.r_num + (p.rv_mean - rv) * p.rv_ivar) / (m.r_ss + p.rv_ivar) * step;
                    if (drv < p.rvmin - rv) drv = p.rvmin - rv;
                    if (drv > p.rvmax - rv) drv = p.rvmax - rv;
                    rv += drv;
                    // (16 bands: the opening kernel, at the edge of the register file, spills 420
                    // bytes with a shorter evaluation of its own for the first iteration
                    // -- 1.29 -> 3.28 ms -- and keeps the one full form)
                    constexpr bool SHORT_FIRST = NB != 16;
                    if (SHORT_FIRST && it + 1 < niter) mle_list<NB, RVF, false>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
                    else mle_list<NB, RVF, true>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
                }
                lnl_new = -0.5 * m.chi2;
                dl = fabs(lnl_new - lnl_old);
                if (lnl_new < lnl_old) step /= 1.2;
                lnl_old = lnl_new;
            }
            if (go) {
                o_lnl = final_lnl<RVF>(sp, p, m.chi2, true);
                o_lnprob = first_cut_lnprob(sp, o_lnl, m.scale, m.i00);
                o_av = av;
                o_rv = rv;
                o_step = step;
                M = o_lnprob;
                if (lnl_new == lnl_new) {
                    L = lnl_new;
                    if (dl > p.ltol) T = lnl_new;
                }
            }
            store = go;
        }
        // The next item's list entry was requested a whole item ago: it is taken HERE, on every
        // path and in front of this item's stores, not at the top of the next iteration behind
        // them (see `landed`: that wait would include the stores' write acknowledgements).
        landed(i_nxt);
        if (store) {
            // (non-temporal, like k_derive's: -3 %; the few stars that iterate on read their state
            // back from memory)
#define FF_ST(p, x) __builtin_nontemporal_store((x), (p) + q)
            FF_ST(r_lnl, o_lnl);
            FF_ST(r_chi2, m.chi2);
            FF_ST(r_scale, m.scale);
            FF_ST(r_av, o_av);
            if constexpr (!RVF) FF_ST(r_rv, o_rv);
            FF_ST(rec.plane(5), m.i00);
            FF_ST(rec.plane(6), m.i01);
            FF_ST(rec.plane(7), m.i02);
            FF_ST(rec.plane(8), m.i11);
            FF_ST(rec.plane(9), m.i12);
            FF_ST(rec.plane(10), m.i22);
            FF_ST(step_st, o_step);
            FF_ST(lnprob_st, o_lnprob);
#undef FF_ST
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
    };
    if constexpr (FIRST) {
        ItemWalk wk;
        wk.init(wbase, nstar, blockIdx.x & 7, blockIdx.x >> 3, (gridDim.x + 7 - (blockIdx.x & 7)) >> 3);
        walk(wk);
    } else {
        const int nunit = NCHUNK * nact * CONT_P;
        for (int unit = blockIdx.x; unit < nunit; unit += gridDim.x) {
            SegWalk wk;
            wk.init(wbase, nstar, act, nact, unit);
            walk(wk);
        }
    }
}

// Per-star flux decision over the star's work items (one workgroup per star).  The stars that
// iterate on are counted in n_unconv and, where `act` is given, listed there (the next
// continuation launch walks their segments without a host round trip).
__global__ void k_fflux_decide(int nstar, const int32_t *__restrict__ wbase,
                               const double *__restrict__ part, double ln_sub,
                               int32_t *__restrict__ k2state, double *__restrict__ maxsurv,
                               int32_t *__restrict__ n_unconv, int32_t *__restrict__ act,
                               int32_t *__restrict__ zero_next) {
    __shared__ double sm[3][4];
    const int s = blockIdx.x;
    // (the counter the NEXT round's decision adds to: free since the launch before this one read it)
    if (zero_next && s == 0 && threadIdx.x == 0) *zero_next = 0;
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
        const int q = atomicAdd(n_unconv, 1);
        if (act) act[q] = s;
    } else {
        k2state[s] = -k2state[s] - 1;
    }
}

// The record index: for every selected model, in ascending model order per star
// (= np.where order, fitting.py:988-991), its model number and the slot of its values.
// Three membership bit-masks decide everything -- `mask` (selected), `dmask` (selected and
// NOT a survivor of the cull: values to be derived) and `cmask` (candidates of the cull) --
// so no value plane and no tag is read:
//   survivor  -> slot = its position in the candidate lists (where k_fflux left the values)
//              = coffsets[(s, c)] + rank among the chunk's candidates
//   derived   -> slot = ncand + position in the derived lists, and the model is appended to
//                der_idx (the work list of k_derive)
// A chunk's words are fetched 64 tiles at a time and the three popcounts prefix-summed at
// once (packed 20 bits each: a window holds 16 384 models).
__global__ void __launch_bounds__(TILE)
k_rec_index(int64_t nmodel, int ntile, int nstar, const unsigned long long *__restrict__ mask,
            const unsigned long long *__restrict__ dmask,
            const unsigned long long *__restrict__ cmask, const int64_t *__restrict__ offsets,
            const int64_t *__restrict__ doffsets, const int64_t *__restrict__ coffsets,
            const int64_t *__restrict__ cand_off, int64_t capacity,
            int32_t *__restrict__ rec_idx, int32_t *__restrict__ rec_slot,
            int32_t *__restrict__ der_idx) {
    __shared__ unsigned long long s_slot[TILE / 64 + 1];
    __shared__ unsigned long long s_word[3][TILE];
    __shared__ unsigned long long s_pre[TILE];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    int64_t base = offsets[(int64_t)s * NCHUNK + c];
    int64_t dbase = doffsets[(int64_t)s * NCHUNK + c] + cand_off[nstar];
    int64_t cbase = coffsets[(int64_t)s * NCHUNK + c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int64_t row = (int64_t)s * (4 * ntile);
    constexpr unsigned long long F20 = (1ull << 20) - 1ull;
    for (int tb = t0; tb < t1; tb += TILE / 4) {
        const int nt = min(TILE / 4, t1 - tb);
        const bool in = (int)threadIdx.x < 4 * nt;
        const int64_t a = row + (int64_t)tb * 4 + (in ? threadIdx.x : 0);
        unsigned long long wm = mask[a], wd = dmask[a], wc = cmask[a];
        if (!in) wm = wd = wc = 0ull;
        unsigned long long tot;
        const unsigned long long pre = block_exclusive_sum<unsigned long long, TILE>(
            (unsigned long long)__popcll(wm) | ((unsigned long long)__popcll(wd) << 20) |
                ((unsigned long long)__popcll(wc) << 40),
            s_slot, tot);
        s_word[0][threadIdx.x] = wm;
        s_word[1][threadIdx.x] = wd;
        s_word[2][threadIdx.x] = wc;
        s_pre[threadIdx.x] = pre;
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < nt; ++k) {
            const unsigned long long b = s_word[0][4 * k + w];
            if ((b >> lane) & 1ull) {
                const unsigned long long bd = s_word[1][4 * k + w], bc = s_word[2][4 * k + w];
                const unsigned long long pk = s_pre[4 * k + w];
                const int32_t i = (int32_t)((int64_t)(tb + k) * TILE + threadIdx.x);
                const int64_t r = base + (int64_t)(pk & F20) + __popcll(b & below);
                int64_t slot;
                if ((bd >> lane) & 1ull) {
                    slot = dbase + (int64_t)((pk >> 20) & F20) + __popcll(bd & below);
                    der_idx[slot - cand_off[nstar]] = i;
                } else {
                    slot = cbase + (int64_t)(pk >> 40) + __popcll(bc & below);
                }
                if (r < capacity) {
                    // (plain stores: the lanes' entries complete their lines in the L2 -- non-temporal
                    // ones cost this kernel 0.06 ms)
                    rec_idx[r] = i;
                    rec_slot[r] = (int32_t)slot;
                }
            }
        }
        base += (int64_t)(tot & F20);
        dbase += (int64_t)((tot >> 20) & F20);
        cbase += (int64_t)(tot >> 40);
        __syncthreads();
    }
}

// Values of the selected models the cull dropped: K1 magnitude sweeps + the full MLE from
// the model's row (fitting.py:176-243, 502-576; lnl as fitting.py:806-815 leaves it for a
// non-survivor), which is cheaper than having the full-grid pass write eleven planes.
// Lane = one entry of the derived lists (dense lanes, no divergence), written at slot
// ncand + list position: every record plane receives full lines.  Persistent workgroups
// over chunk-major work items like k_fflux; no barrier in the loop.
template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, list_waves(NB))
k_derive(const float *__restrict__ grid, int64_t nmodel_pad, int nstar,
         const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
         const int32_t *__restrict__ der_idx, const int32_t *__restrict__ wbase,
         const ItemGeom *__restrict__ items, const int64_t *__restrict__ cand_off,
         RecPlanes rec) {
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    constexpr bool WIDE = wide_bands(NB);
    __shared__ double s_F[WIDE ? NB * TILE : 1];
    const LdsColumn Fc = (LdsColumn)s_F + (WIDE ? threadIdx.x : 0);
    auto lane_model = [&](int item) -> int32_t {
        if (item < 0) return 0;
        const ItemGeom ig = items[item];
        const int64_t q = ig.q0 + threadIdx.x;
        const int64_t last = ig.q0 + ig.n - 1;
        return der_idx[q < last ? q : last];
    };
    const int64_t slot0 = cand_off[nstar];
    ItemWalk wk;
    wk.init(wbase, nstar, blockIdx.x & 7, blockIdx.x >> 3, (gridDim.x + 7 - (blockIdx.x & 7)) >> 3);
    if (wk.done()) return;
    // Software pipeline over the work items (two waves per SIMD cannot hide a gather AND a
    // write acknowledgement per item): while item k is computed, the coefficient rows of item
    // k + 1 (36 registers at 12 bands) and the list entry of item k + 2 are in flight; both are
    // taken in front of item k's stores (`landed`), so no wait in the loop finds a load younger
    // than a whole item or a store younger than the previous item's.
    int item = wk.item;
    wk.next();
    int item1 = wk.item;
    // (the second set of rows fits the register file up to 12 bands; beyond that the kernel
    // spills already and the rows are fetched when their item comes up)
    constexpr bool PIPE = NB <= 12;
    int32_t i0 = lane_model(item), i1 = lane_model(item1);
    RowRegs<NB> nx;
    if (PIPE) gather_row<NB>(grid, nmodel_pad, i0, nx);
    landed(i1);
    if (PIPE) landed<NB>(nx);
    while (item >= 0) {
        Coef<NB> c;
        if (PIPE) {
            row_to_coef<NB>(nx, c);
            gather_row<NB>(grid, nmodel_pad, i1, nx);    // (item1 < 0: row 0, never used)
        } else {
            gather_row<NB>(grid, nmodel_pad, i0, nx);
            row_to_coef<NB>(nx, c);
        }
        if (item1 >= 0) wk.next();
        const int item2 = item1 >= 0 ? wk.item : -1;
        int32_t i2 = lane_model(item2);
        const ItemGeom ig = items[item];
        const int s = ig.s;
        const int64_t slot = slot0 + ig.q0 + threadIdx.x;
        const bool live = (int)threadIdx.x < ig.n && slot < rec.cap;
        const bool wave_live = (int)(threadIdx.x & ~63u) < ig.n;      // wave-uniform
        double o_lnl = 0., o_av = 0., o_rv = 0.;
        Mle m;
        if (wave_live) {
            const StarPrep &sp = stars[s];
            double F0[WIDE ? 1 : NB];
#ifdef BRUTUS_F0_FROM_TABLE
            if constexpr (!WIDE) load_F0<NB>(grid, nmodel_pad, i0, F0);
#else
            if constexpr (!WIDE) compute_F0_tbl<NB>(c, s_tbl, F0);
#endif
            double av = p.av_mean, rv = p.rv_mean;
            const int K = k1[s];
            if constexpr (RVF) {
                double R[WIDE ? 1 : NB];
                GramR G;
                if constexpr (WIDE) {
                    gram_init_rf_wide<NB>(c, rv, sp, G);
                } else {
                    coef_R<NB>(c, rv, R);
                    gram_init_rf<NB>(c, R, sp, G);
                }
                double a_, c_;
                if (K > 0) gram_sweep_rf(G, sp.S, p, av, a_, c_);       // one solve is exact (see k_k1probe)
                mle_list<NB, true, true>(c, R, F0, Fc, sp, p, av, rv, s_tbl, m);
            } else {
                Gram G;
                gram_init<NB>(c, sp, G);
                for (int kk = 0; kk < K; ++kk) {
                    double a_, b_, c_;
                    gram_sweep(G, sp.S, p, av, rv, a_, b_, c_);
                }
                const double R1[1] = {0.};
                mle_list<NB, false, true>(c, R1, F0, Fc, sp, p, av, rv, s_tbl, m);
            }
            o_lnl = final_lnl<RVF>(sp, p, m.chi2, false);
            o_av = av;
            o_rv = rv;
        }
        if (PIPE) landed<NB>(nx);
        landed(i2);
        if (wave_live && live) {
            double *out = rec.vals + slot;
            // (non-temporal: nothing in this call reads a record again; as plain stores the eleven
            // planes' lines went through the L2 beside the row gathers -- k_derive 0.81 -> 0.68 ms)
#define REC_ST(k, x) __builtin_nontemporal_store((x), out + (int64_t)(k) * rec.cap)
            REC_ST(0, o_lnl);
            REC_ST(1, m.chi2);
            REC_ST(2, m.scale);
            REC_ST(3, o_av);
            if constexpr (!RVF) REC_ST(4, o_rv);      // pinned Rv is not stored
            REC_ST(5, m.i00);
            REC_ST(6, m.i01);
            REC_ST(7, m.i02);
            REC_ST(8, m.i11);
            REC_ST(9, m.i12);
            REC_ST(10, m.i22);
#undef REC_ST
        }
        item = item1;
        item1 = item2;
        i0 = i1;
        i1 = i2;
    }
}

}  // namespace
