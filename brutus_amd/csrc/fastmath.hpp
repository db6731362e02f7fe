// fastmath.hpp -- table / series forms of 10^x, e^x, ln x, 1/x, sqrt x for the hot kernels
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

namespace {

// 2^(k/64), k = 0..63, correctly rounded.
__constant__ double kExp2Tbl[64] = {
    1, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.1023825833078409, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.2021567314527031, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.2553807570246911, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.3396675240533029,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.5590044002378369, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.6457554781539649, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.7186192981224779, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.9784560263879509};

__device__ __forceinline__ double fast_exp10(double x, const double *__restrict__ tbl = kExp2Tbl) {
    // `tbl`: the 64-entry table, by default the constant-memory copy; the hot
    // kernels pass an LDS copy (stage_exp_table) so that the divergent look-up
    // is a ds_read instead of a vector-memory load.
    // 10^x = 2^e * 2^(k/64) * exp(t):  n = rint(64 x log2 10) = 64 e + k,
    // t = (x - n log10(2)/64) ln 10, |t| <= ln(2)/128.  Two-term Cody-Waite
    // reduction (the high part has 32 significant bits, so n*hi is exact), a
    // degree-5 polynomial and one table look-up: ~13 f64 ops (ocml exp10: ~40),
    // error < 1 ulp + table rounding on |x| < 300.
    // n by the 1.5 * 2^52 shift (one fma + one add; the integer sits in the low
    // word), ln 10 folded into the polynomial coefficients: 15 instructions.
    const double ns = fma(x, 212.60339807279118, 6755399441055744.0);
    const double n = ns - 6755399441055744.0;
    double r = fma(-n, 0.0047035936804604717, x);
    r = fma(-n, 1.7892345153159123e-12, r);
    double pl = 0.5393829291955817;                        // ln(10)^5 / 5!
    pl = fma(pl, r, 1.1712551489122673);                   // ln(10)^4 / 4!
    pl = fma(pl, r, 2.034678592293477);                    // ln(10)^3 / 3!
    pl = fma(pl, r, 2.6509490552391997);                   // ln(10)^2 / 2
    pl = fma(pl, r, 2.302585092994046);                    // ln(10)
    pl = fma(pl, r, 1.0);
    const int ni = __double2loint(ns);
    return ldexp(tbl[ni & 63] * pl, ni >> 6);
}

// Table-free variant (degree-13 polynomial after the same kind of reduction,
// 19 f64 ops, <= 1.5 ulp): used where VGPR pressure, not ALU, is the limit.
__device__ __forceinline__ double poly_exp10(double x) {
    const double n = rint(x * 3.3219280948873623);
    double r = fma(-n, 3.01029995663839276e-01, x);
    r = fma(-n, 1.42502325707809354e-17, r);
    const double t = r * 2.3025850929940457;
    double pl = 1.6059043836821613e-10;                    // 1/13!
    pl = fma(pl, t, 2.08767569878681e-09);
    pl = fma(pl, t, 2.505210838544172e-08);
    pl = fma(pl, t, 2.755731922398589e-07);
    pl = fma(pl, t, 2.7557319223985893e-06);
    pl = fma(pl, t, 2.48015873015873e-05);
    pl = fma(pl, t, 1.984126984126984e-04);
    pl = fma(pl, t, 1.3888888888888889e-03);
    pl = fma(pl, t, 8.333333333333333e-03);
    pl = fma(pl, t, 4.1666666666666664e-02);
    pl = fma(pl, t, 1.6666666666666666e-01);
    pl = fma(pl, t, 0.5);
    pl = fma(pl, t, 1.0);
    pl = fma(pl, t, 1.0);
    return ldexp(pl, (int)n);
}

// Copy the table to LDS; call from all threads of a >= 64-thread workgroup,
// followed by __syncthreads().
__device__ __forceinline__ void stage_exp_table(double *lds_tbl) {
    if (threadIdx.x < 64) lds_tbl[threadIdx.x] = kExp2Tbl[threadIdx.x];
}

// e^x with the same table: n = rint(64 x / ln 2), r = x - n ln2/64 (two-term
// Cody-Waite), degree-5 polynomial; ~13 f64 ops, <= 1 ulp + table rounding.
__device__ __forceinline__ double fast_exp(double x, const double *__restrict__ tbl = kExp2Tbl) {
    if (!(x > -745.)) return x == x ? 0. : x;        // underflow / -inf / NaN
    const double n = rint(x * 92.332482616893657);   // 64 / ln 2
    double r = fma(-n, 0.01083042469326756, x);      // ln2/64 head, 32 significant bits: n*hi exact
    r = fma(-n, 2.9815858269852933e-12, r);
    double pl = 8.3333333333333332e-03;
    pl = fma(pl, r, 4.1666666666666664e-02);
    pl = fma(pl, r, 1.6666666666666666e-01);
    pl = fma(pl, r, 0.5);
    pl = fma(pl, r, 1.0);
    pl = fma(pl, r, 1.0);
    const int ni = (int)n;
    return ldexp(tbl[ni & 63] * pl, ni >> 6);
}

// ln x for finite x > 0: x = 2^e m, m in [sqrt(1/2), sqrt 2); ln m = 2 atanh(s),
// s = (m - 1)/(m + 1), |s| <= 0.1716, odd series to s^21; ~30 f64 ops (ocml: ~98),
// <= 1 ulp, well conditioned at x -> 1 (m - 1 is exact).
__device__ __forceinline__ double fast_log(double x) {
    if (!(x > 0.) || !(x < INFINITY)) return log(x);          // 0, negative, inf, NaN: ocml semantics
    int e;
    double m = frexp(x, &e);                                   // m in [0.5, 1)
    if (m < 0.70710678118654752440) {
        m *= 2.;
        --e;
    }
    const double s = (m - 1.) / (m + 1.);
    const double z = s * s;
    double pl = 1. / 21.;
    pl = fma(pl, z, 1. / 19.);
    pl = fma(pl, z, 1. / 17.);
    pl = fma(pl, z, 1. / 15.);
    pl = fma(pl, z, 1. / 13.);
    pl = fma(pl, z, 1. / 11.);
    pl = fma(pl, z, 1. / 9.);
    pl = fma(pl, z, 1. / 7.);
    pl = fma(pl, z, 1. / 5.);
    pl = fma(pl, z, 1. / 3.);
    // ln m = 2 s + 2 s z pl ; ln x = e ln2_hi + (e ln2_lo + ln m)
    const double lm = fma(2. * s * z, pl, 2. * s);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

// 1/x, sqrt x, 1/sqrt x for normal-range x > 0 from the hardware seed
// (v_rcp_f64 / v_rsq_f64) plus Newton steps: ~1 ulp, no denormal / overflow
// rescue and no correct rounding, i.e. 5-9 instructions instead of the 13-17 of
// an IEEE divide / sqrt.  Used where the result feeds a prior density, never
// where a comparison must reproduce the reference bit for bit.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.), r, r);
    r = fma(fma(-x, r, 1.), r, r);
    return r;
}
__device__ __forceinline__ void fast_sqrt_rsqrt(double x, double &sq, double &rsq) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    // 1 / sqrt x = 1 / g by Newton, y <- y + y (1 - y g): after the coupled step h is good to
    // 2^-24 (the hardware seed to 2^-12.5), two steps give <= 1 ulp
    double y2 = h + h;
    y2 = fma(y2, fma(-y2, g, 1.), y2);
    y2 = fma(y2, fma(-y2, g, 1.), y2);
    sq = x == 0. ? 0. : g;
    rsq = y2;
}
// the same for x > 0 in the normal range (no select for x == 0: 0 gives NaN)
__device__ __forceinline__ void fast_sqrt_rsqrt_pos(double x, double &sq, double &rsq) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    double y2 = h + h;
    y2 = fma(y2, fma(-y2, g, 1.), y2);
    y2 = fma(y2, fma(-y2, g, 1.), y2);
    sq = g;
    rsq = y2;
}
__device__ __forceinline__ double fast_sqrt_pos(double x) {
    double g, h;
    fast_sqrt_rsqrt_pos(x, g, h);
    return g;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    double g, h;
    fast_sqrt_rsqrt(x, g, h);
    return g;
}
// fast_exp without the early-out branch (selects instead)
__device__ __forceinline__ double fast_exp_bf(double x, const double *__restrict__ tbl) {
    const double xc = fmax(x, -745.);                // also maps NaN to -745; fixed below
    const double ns = fma(xc, 92.332482616893657, 6755399441055744.0);   // 1.5 * 2^52 shift
    const double n = ns - 6755399441055744.0;
    double r = fma(-n, 0.01083042469326756, xc);
    r = fma(-n, 2.9815858269852933e-12, r);
    double pl = 8.3333333333333332e-03;
    pl = fma(pl, r, 4.1666666666666664e-02);
    pl = fma(pl, r, 1.6666666666666666e-01);
    pl = fma(pl, r, 0.5);
    pl = fma(pl, r, 1.0);
    pl = fma(pl, r, 1.0);
    const int ni = __double2loint(ns);
    const double v = ldexp(tbl[ni & 63] * pl, ni >> 6);
    return x > -745. ? v : (x == x ? 0. : x);
}
// fast_exp_bf for a finite argument: no selects for NaN / -inf, an argument below -745 gives
// the same denormal as -745 (5e-324, not 0).  For the Galactic prior's components, where a
// non-finite argument only arises from a sample that is out of bounds and discarded anyway.
__device__ __forceinline__ double fast_exp_fin(double x, const double *__restrict__ tbl) {
    const double xc = fmax(x, -745.);
    const double ns = fma(xc, 92.332482616893657, 6755399441055744.0);   // 1.5 * 2^52 shift
    const double n = ns - 6755399441055744.0;
    double r = fma(-n, 0.01083042469326756, xc);
    r = fma(-n, 2.9815858269852933e-12, r);
    double pl = 8.3333333333333332e-03;
    pl = fma(pl, r, 4.1666666666666664e-02);
    pl = fma(pl, r, 1.6666666666666666e-01);
    pl = fma(pl, r, 0.5);
    pl = fma(pl, r, 1.0);
    pl = fma(pl, r, 1.0);
    const int ni = __double2loint(ns);
    return ldexp(tbl[ni & 63] * pl, ni >> 6);
}
// fast_log with the reciprocal above (a few ulp)
__device__ __forceinline__ double fast_log_r(double x) {
    // branch-free: normal-range x > 0 takes the series; 0 -> -inf, +inf -> +inf,
    // negative / NaN -> NaN by selects (subnormal x is not rescued: ~2^-1022 only)
    int e;
    double m = frexp(x, &e);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? 2. * m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.) * fast_rcp(m + 1.);
    const double z = s * s;
    double pl = 1. / 21.;
    pl = fma(pl, z, 1. / 19.);
    pl = fma(pl, z, 1. / 17.);
    pl = fma(pl, z, 1. / 15.);
    pl = fma(pl, z, 1. / 13.);
    pl = fma(pl, z, 1. / 11.);
    pl = fma(pl, z, 1. / 9.);
    pl = fma(pl, z, 1. / 7.);
    pl = fma(pl, z, 1. / 5.);
    pl = fma(pl, z, 1. / 3.);
    const double lm = fma(2. * s * z, pl, 2. * s);
    const double ed = (double)e;
    const double v = fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
    return x > 0. ? (x < INFINITY ? v : x) : (x == 0. ? -INFINITY : nan(""));
}
// fast_log_r for a normal-range positive argument (no selects for 0 / inf / negative / NaN)
__device__ __forceinline__ double fast_log_pos(double x) {
    int e;
    double m = frexp(x, &e);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? 2. * m : m;
    e = lo ? e - 1 : e;
    const double s = (m - 1.) * fast_rcp(m + 1.);
    const double z = s * s;
    double pl = 1. / 21.;
    pl = fma(pl, z, 1. / 19.);
    pl = fma(pl, z, 1. / 17.);
    pl = fma(pl, z, 1. / 15.);
    pl = fma(pl, z, 1. / 13.);
    pl = fma(pl, z, 1. / 11.);
    pl = fma(pl, z, 1. / 9.);
    pl = fma(pl, z, 1. / 7.);
    pl = fma(pl, z, 1. / 5.);
    pl = fma(pl, z, 1. / 3.);
    const double lm = fma(2. * s * z, pl, 2. * s);
    const double ed = (double)e;
    return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));
}

}  // namespace
