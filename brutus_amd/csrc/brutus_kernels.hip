// brutus_kernels.hip -- gfx950 (MI355X, CDNA4) kernels + C ABI for the brutus
// per-star grid-likelihood path.  Written for wave64 / 256 CUs / 8 XCDs; no
// CUDA compatibility layer, no dual paths.
//
// What is computed (citations are to the upstream reference, brutus/*.py):
//   fitting.py:579-820   loglike         -- whole function, batched over stars
//   fitting.py:141-264   _optimize_fit_mag main loop      (mag_sweep)
//   fitting.py:502-576   _get_sed_mle                      (mle_eval)
//   fitting.py:385-420   _optimize_fit_flux step           (k_flux)
//   utils.py:330-345     _get_seds                         (inlined in both)
//   utils.py:161-176     _chisquare_logpdf                 (k_finalize, final_lnl)
//   fitting.py:976-991   lnpost parallax clip + first cut  (first_cut_lnprob,
//                                                           k_cmp_*, k_emit)
//   cluster.py:336-414   isochrone_loglike hot block       (k_cluster)
//
// Execution model.  One lane owns one model; a 256-lane workgroup owns a tile
// of 256 consecutive models and keeps that tile's 3*NB float32 coefficients in
// VGPRs while it loops over a group of stars, so the coefficient grid is read
// from HBM once per star *group*, not once per star.  Per-star vectors are
// wave-uniform and are fetched through the scalar cache (s_load).  All
// arithmetic is float64 on float32-rounded grid values, exactly the numeric
// type the reference computes in (numba promotes the f32 grid to f64).
//
// The reference's control flow hangs on three per-star GLOBAL decisions (number
// of magnitude sweeps K1, the init_thresh cull, number of flux iterations K2).
// Each is a max-type reduction over the grid, so every phase is a kernel that
// emits per-(tile, star) partial maxima, followed by a tiny per-star decision
// kernel.  Per-model work inside a phase is independent of every other model.
//   "not converged at sweep k"  <=>  max{logwt_i : step_i >= tol} > max_i logwt_i + ln(init_thresh)
// turns the masked max-step test (fitting.py:246-264) into two plain maxima.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/brutus_amd.h"

namespace {

constexpr int TILE = 256;        // models per workgroup (4 waves of 64)
constexpr int NBMAX = BRUTUS_MAX_FILT;
constexpr int STAR_GROUP = 16;   // stars per workgroup (grid.y = ceil(S / STAR_GROUP))
constexpr int KCAP = 16;         // max sweeps probed by one k_mag_stats launch
constexpr int NCHUNK = 64;       // model-range chunks for ordered compaction
constexpr double BIG = 1e300;

thread_local std::string g_err;
bool g_timing = false;
struct TimingEntry { std::string name; float ms; int count; };
std::vector<TimingEntry> g_last_timing;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess)                                                  \
            return fail(BRUTUS_EHIP, "%s failed: %s (%s:%d)", #expr,           \
                        hipGetErrorString(e_), __FILE__, __LINE__);            \
    } while (0)

// ---------------------------------------------------------------------------
// device-side data
// ---------------------------------------------------------------------------
struct StarPrep {
    double g[NBMAX];    // magnitudes -2.5 log10 d        (fitting.py:721)
    double iW[NBMAX];   // 1 / mags_var                   (fitting.py:722-725)
    double d[NBMAX];    // flux (0 for masked bands)
    double iV[NBMAX];   // 1 / flux variance (0 for masked bands)
    double S;           // sum_j 1/mags_var_j             (fitting.py:162)
    double lnl_const;   // -0.5 (Ndim ln 2pi + sum ln V)  (fitting.py:806-807)
    double c0, c1;      // chi-square logpdf constants    (utils.py:169-170)
    double par, par_ivar;        // parallax, 1/err^2 for the cull (fitting.py:749-756)
    double sp_mean, sp_var;      // pdf.py:252-255 scale-space parallax Gaussian
    int ndim;
    int has_par;        // finite parallax & error
    int sp_on;          // p/err > 4 (pdf.py:209)
    int pad_;
};

struct DevParams {
    double avmin, avmax, rvmin, rvmax;
    double av_mean, av_ivar, rv_mean, rv_ivar;
    double mtol;            // 2.5 * ltol
    double ltol;
    double ln_init, ln_sub, ln_wt;
    double a_reg, r_reg;    // 1/0.05^2, 1/0.1^2 (fitting.py:431,524)
    int dim_prior;
};

struct Planes {            // each (nstar, nmodel) float64, row stride = nmodel
    double *lnlp;          // cull statistic lnl_p
    double *lnprob;        // fast path: first-cut statistic
    double *lnl, *chi2, *scale, *av, *rv;
    double *icov[6];
    double *step;
    int64_t nmodel;
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// Combine one value per lane into a single per-(tile, star) maximum.  NaN lanes
// must already be mapped to -inf by the caller.  `slot` is LDS scratch (4 doubles).
__device__ __forceinline__ void block_max_store(double v, double *slot, double *out) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) slot[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = slot[0];
        m = slot[1] > m ? slot[1] : m;
        m = slot[2] > m ? slot[2] : m;
        m = slot[3] > m ? slot[3] : m;
        *out = m;
    }
    __syncthreads();
}

template <int NB>
struct Coef {
    float m[NB], r0[NB], dr[NB];
};

template <int NB>
__device__ __forceinline__ void load_coef(const float *__restrict__ grid, int64_t nmodel_pad,
                                          int64_t i, Coef<NB> &c) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float *p = grid + (int64_t)(3 * j) * nmodel_pad + i;
        c.m[j] = p[0];
        c.r0[j] = p[nmodel_pad];
        c.dr[j] = p[2 * nmodel_pad];
    }
}

// ---- magnitude phase (fitting.py:158-243) ---------------------------------
template <int NB>
struct MagState {
    double res[NB], R[NB];
    double av, rv, P, Q;
    double dav, drv, logwt;
};

template <int NB>
__device__ __forceinline__ void mag_init(const Coef<NB> &c, const StarPrep &sp,
                                         const DevParams &p, MagState<NB> &st) {
    st.av = p.av_mean;   // fitting.py:700-703
    st.rv = p.rv_mean;
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
__global__ void k_prep(int nstar, int nfilt, const double *__restrict__ flux,
                       const double *__restrict__ err, const uint8_t *__restrict__ mask,
                       const double *__restrict__ par, const double *__restrict__ perr,
                       int has_parallax, StarPrep *__restrict__ out,
                       int32_t *__restrict__ ndim_out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstar) return;
    StarPrep sp;
    int ndim = 0;
    double S = 0., sumlnv = 0.;
    const double kmag = 2.5 / log(10.);
    for (int j = 0; j < NBMAX; ++j) {
        double d = 0., iv = 0., g = 0., iw = 0.;
        if (j < nfilt) {
            const double f = flux[(int64_t)s * nfilt + j];
            const double e = err[(int64_t)s * nfilt + j];
            const bool ok = mask[(int64_t)s * nfilt + j] && isfinite(f) && isfinite(e) && e > 0.;
            if (ok) {
                ++ndim;
                const double v = e * e;
                d = f;
                iv = 1. / v;
                sumlnv += log(v);
                g = -2.5 * log10(f);
                double W = kmag * kmag * v / (f * f);
                if (!isfinite(g)) {                           // fitting.py:724-725
                    g = 0.;
                    W = 1e50;
                }
                iw = 1. / W;
                S += 1. / W;
            }
        }
        sp.d[j] = d;
        sp.iV[j] = iv;
        sp.g[j] = g;
        sp.iW[j] = iw;
    }
    sp.S = S;
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
    out[s] = sp;
    ndim_out[s] = ndim;
}

// AoS (nmodel, nfilt, 3) -> device grid blob; padded entries zero.  With
// Np = nmodel_pad and offsets in 4-byte units:
//   [0, 3*NB*Np)          f32 band-major SoA [NB][3][Np]   full-grid scans
//   [3*NB*Np, 6*NB*Np)    f32 model-major   [Np][NB][3]    single-model gathers
//   [6*NB*Np, 8*NB*Np)    f64 band-major    [NB][Np]       F0 = 10^(-0.4 mag)
// F0 (the unreddened model flux, fitting.py:529) is star-independent, so it
// is tabulated once here instead of 12 exponentials per tile of the full scan.
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
    const double n = rint(x * 212.60339807279118);
    double r = fma(-n, 0.0047035936804604717, x);
    r = fma(-n, 1.7892345153159123e-12, r);
    const double t = r * 2.3025850929940459;
    double pl = 8.3333333333333332e-03;                    // 1/5!
    pl = fma(pl, t, 4.1666666666666664e-02);               // 1/4!
    pl = fma(pl, t, 1.6666666666666666e-01);               // 1/3!
    pl = fma(pl, t, 0.5);
    pl = fma(pl, t, 1.0);
    pl = fma(pl, t, 1.0);
    const int ni = (int)n;
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
    h = fma(fma(-h, g, 0.5), h, h);        // h -> 1 / (2 sqrt x)
    sq = x == 0. ? 0. : g;
    rsq = 2. * h;
}
__device__ __forceinline__ double fast_sqrt(double x) {
    double g, h;
    fast_sqrt_rsqrt(x, g, h);
    return g;
}
// fast_exp without the early-out branch (selects instead)
__device__ __forceinline__ double fast_exp_bf(double x, const double *__restrict__ tbl) {
    const double xc = fmax(x, -745.);                // also maps NaN to -745; fixed below
    const double n = rint(xc * 92.332482616893657);
    double r = fma(-n, 0.01083042469326756, xc);
    r = fma(-n, 2.9815858269852933e-12, r);
    double pl = 8.3333333333333332e-03;
    pl = fma(pl, r, 4.1666666666666664e-02);
    pl = fma(pl, r, 1.6666666666666666e-01);
    pl = fma(pl, r, 0.5);
    pl = fma(pl, r, 1.0);
    pl = fma(pl, r, 1.0);
    const int ni = (int)n;
    const double v = ldexp(tbl[ni & 63] * pl, ni >> 6);
    return x > -745. ? v : (x == x ? 0. : x);
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
            double *__restrict__ part) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    Coef<NB> c;
    load_coef<NB>(grid, nmodel_pad, i, c);
    const int s0 = blockIdx.y * STAR_GROUP;
    const int s1 = min(nstar, s0 + STAR_GROUP);
    const double ninf = -INFINITY;
    for (int s = s0; s < s1; ++s) {
        const StarPrep &sp = stars[s];
        MagState<NB> st;
        mag_init<NB>(c, sp, p, st);
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
          Planes pl, double *__restrict__ part) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    Coef<NB> c;
    load_coef<NB>(grid, nmodel_pad, i, c);
    double F0[NB];
    compute_F0<NB>(c, F0);
    const int s0 = blockIdx.y * STAR_GROUP;
    const int s1 = min(nstar, s0 + STAR_GROUP);
    for (int s = s0; s < s1; ++s) {
        const StarPrep &sp = stars[s];
        MagState<NB> st;
        mag_init<NB>(c, sp, p, st);
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

__global__ void k_debug_exp10(const double *__restrict__ x, double *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fast_exp10(x[i]);
}
__global__ void k_debug_math(int which, const double *__restrict__ x, double *__restrict__ y,
                             int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = which == 1 ? fast_exp(x[i]) : which == 2 ? fast_log(x[i]) : fast_exp10(x[i]);
}

__global__ void k_set_i32(int32_t *p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

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

// lnl as `loglike` returns it for a model the cull dropped / kept, and the
// first-cut statistic lnprob (fitting.py:806-815, 976-985; pdf.py:209-218).
__device__ __forceinline__ double final_lnl(const StarPrep &sp, const DevParams &p, double chi2,
                                            bool survivor) {
    if (p.dim_prior) return chi2 > 0. ? sp.c0 + sp.c1 * log(chi2) - chi2 / 2. : -INFINITY;
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
                for (int k = 0; k < K; ++k) {
                    double dav, lw;
                    gram_sweep_rf(G, sp.S, p, av, dav, lw);
                    if (k < KS && live && lw == lw) {
                        double *c0 = col + (size_t)(2 * k) * TILE;
                        if (lw > c0[0]) c0[0] = lw;
                        if (fabs(dav) >= p.mtol && lw > c0[TILE]) c0[TILE] = lw;
                    }
                }
                mle_fast_rf<NB, true, false>(c, R, F0, sp, p, av, s_tbl, m);
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
                mle_fast<NB, true>(c, F0, sp, p, av, rv, s_tbl, m);
            }
            const double lnl = -0.5 * m.chi2;
            double lnlp = lnl;
            if (sp.has_par) {
                const double dp = sqrt(m.scale) - sp.par;
                lnlp = lnl - 0.5 * (dp * dp * sp.par_ivar);
            }
            const double lnprob =
                first_cut_lnprob(sp, final_lnl(sp, p, m.chi2, false), m.scale, m.i00);
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

// Exclusive scan of counts[(s, c)] in (s, c) order; one workgroup, thread s owns star s.
//   offsets[(s, c)], star_off[s] (star_off[nstar] = total), wbase[s] = first work
//   item of star s when its list is cut into TILE-sized work items (wbase[nstar] = #items).
__global__ void k_offsets(int nstar, const int64_t *__restrict__ counts,
                          int64_t *__restrict__ offsets, int64_t *__restrict__ star_off,
                          int32_t *__restrict__ wbase) {
    __shared__ int64_t tot[BRUTUS_MAX_BATCH + 1];
    const int s = threadIdx.x;
    int64_t n = 0;
    if (s < nstar)
        for (int c = 0; c < NCHUNK; ++c) n += counts[(int64_t)s * NCHUNK + c];
    if (s < nstar) tot[s] = n;
    __syncthreads();
    if (s == 0) {
        int64_t run = 0;
        int32_t w = 0;
        for (int q = 0; q < nstar; ++q) {
            const int64_t m = tot[q];
            tot[q] = run;
            star_off[q] = run;
            if (wbase) wbase[q] = w;
            run += m;
            w += (int32_t)((m + TILE - 1) / TILE);
        }
        star_off[nstar] = run;
        if (wbase) wbase[nstar] = w;
    }
    __syncthreads();
    if (s < nstar) {
        int64_t run = tot[s];
        for (int c = 0; c < NCHUNK; ++c) {
            offsets[(int64_t)s * NCHUNK + c] = run;
            run += counts[(int64_t)s * NCHUNK + c];
        }
    }
}

__global__ void __launch_bounds__(TILE)
k_cmp_scatter(int64_t nmodel, int ntile, const unsigned long long *__restrict__ mask,
              const int64_t *__restrict__ offsets, int64_t capacity,
              int32_t *__restrict__ out_idx) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    int64_t base = offsets[(int64_t)s * NCHUNK + c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        const unsigned long long b = mask[(int64_t)s * (4 * ntile) + (int64_t)t * 4 + w];
        const bool sel = (b >> lane) & 1ull;
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < w; ++q) woff += wsum[q];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (sel) {
            const int64_t r = base + woff + rank;
            if (r < capacity) out_idx[r] = (int32_t)i;
        }
        base += tot;
        __syncthreads();
    }
}

// Map a work item (TILE consecutive entries of one star's compact list) to its star.
__device__ __forceinline__ int star_of_item(const int32_t *__restrict__ wbase, int nstar, int item) {
    int lo = 0, hi = nstar;   // largest s with wbase[s] <= item
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (wbase[mid] <= item) lo = mid; else hi = mid;
    }
    return lo;
}

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
// and, per work item, L = max lnl_new, T = max{lnl_new : |dlnl| > ltol},
// M = max final lnprob.
template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, 2)
k_fflux(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
        const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
        const int32_t *__restrict__ k2state, int first, const int32_t *__restrict__ surv_idx,
        const int64_t *__restrict__ surv_off, const int32_t *__restrict__ wbase, Planes pl,
        double *__restrict__ part) {
    __shared__ double slot[4];
    const int nitem = wbase[nstar];
    const int niter = first ? 2 : 1;
    for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
        const int s = star_of_item(wbase, nstar, item);
        if (k2state[s] < 0) continue;
        const StarPrep &sp = stars[s];
        const int64_t q = surv_off[s] + (int64_t)(item - wbase[s]) * TILE + threadIdx.x;
        const bool live = q < surv_off[s + 1];
        double L = -INFINITY, T = -INFINITY, M = -INFINITY;
        if (live) {
            const int64_t i = surv_idx[q];
            const int64_t o = (int64_t)s * pl.nmodel + i;
            Coef<NB> c;
            gather_coef<NB>(grid, nmodel_pad, i, c);
            double F0[NB];
            compute_F0_fast<NB>(c, F0);
            double av, rv, step, lnl_old;
            double R[RVF ? NB : 1];
            if constexpr (RVF) coef_R<NB>(c, p.rv_mean, R);
            if (first) {
                av = p.av_mean;
                rv = p.rv_mean;
                const int K = k1[s];
                if constexpr (RVF) {
                    GramR G;
                    gram_init_rf<NB>(c, R, sp, G);
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
                av = pl.av[o];
                rv = pl.rv[o];
                step = pl.step[o];
                lnl_old = -0.5 * pl.chi2[o];
            }
            Mle m;
            if constexpr (RVF) mle_fast_rf<NB, false, false>(c, R, F0, sp, p, av, nullptr, m);
            else mle_fast<NB, false>(c, F0, sp, p, av, rv, nullptr, m);
            double lnl_new = lnl_old, dl = 0.;
            for (int it = 0; it < niter; ++it) {
                double dav = (m.a_num + (p.av_mean - av) * p.av_ivar) / (m.a_ss + p.av_ivar) * step;
                if (dav < p.avmin - av) dav = p.avmin - av;
                if (dav > p.avmax - av) dav = p.avmax - av;
                av += dav;
                if constexpr (RVF) {
                    // the Rv step is clamped to zero; only the stored MLE needs the Rv sums
                    if (it + 1 < niter) mle_fast_rf<NB, false, false>(c, R, F0, sp, p, av, nullptr, m);
                    else mle_fast_rf<NB, false, true>(c, R, F0, sp, p, av, nullptr, m);
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
            store_mle(pl, o, m);
            pl.av[o] = av;
            pl.rv[o] = rv;
            pl.step[o] = step;
            const double lnl = final_lnl(sp, p, m.chi2, true);
            const double lnprob = first_cut_lnprob(sp, lnl, m.scale, m.i00);
            pl.lnl[o] = lnl;
            pl.lnprob[o] = lnprob;
            M = lnprob;
            if (lnl_new == lnl_new) {
                L = lnl_new;
                if (dl > p.ltol) T = lnl_new;
            }
        }
        double *out = part + (int64_t)item * 3;
        block_max_store(L, slot, out);
        block_max_store(T, slot, out + 1);
        block_max_store(M, slot, out + 2);
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
    for (int it = wbase[s] + threadIdx.x; it < wbase[s + 1]; it += blockDim.x)
        for (int q = 0; q < 3; ++q) {
            const double x = part[(int64_t)it * 3 + q];
            v[q] = x > v[q] ? x : v[q];
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
// Survivors of the cull are read from the result planes; the others are
// re-derived from the grid (K1 sweeps + MLE), which is cheaper than having the
// full-grid scan write eleven planes.
template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, 2)
k_emit(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
       const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
       const double *__restrict__ thr_cull, const int32_t *__restrict__ sel_idx,
       const int64_t *__restrict__ sel_off, const int32_t *__restrict__ wbase, Planes pl,
       int64_t capacity, double *__restrict__ sel_vals) {
    const int nitem = wbase[nstar];
    for (int item = blockIdx.x; item < nitem; item += gridDim.x) {
        const int s = star_of_item(wbase, nstar, item);
        const StarPrep &sp = stars[s];
        const int64_t q = sel_off[s] + (int64_t)(item - wbase[s]) * TILE + threadIdx.x;
        if (q >= sel_off[s + 1] || q >= capacity) continue;
        const int64_t i = sel_idx[q];
        const int64_t o = (int64_t)s * pl.nmodel + i;
        double rec[BRUTUS_NVALS];
        if (pl.lnlp[o] > thr_cull[s]) {
            rec[0] = pl.lnl[o];
            rec[1] = pl.chi2[o];
            rec[2] = pl.scale[o];
            rec[3] = pl.av[o];
            rec[4] = pl.rv[o];
#pragma unroll
            for (int k = 0; k < 6; ++k) rec[5 + k] = pl.icov[k][o];
        } else {
            Coef<NB> c;
            gather_coef<NB>(grid, nmodel_pad, i, c);
            double F0[NB];
            compute_F0_fast<NB>(c, F0);
            double av = p.av_mean, rv = p.rv_mean;
            const int K = k1[s];
            Mle m;
            if constexpr (RVF) {
                double R[NB];
                coef_R<NB>(c, rv, R);
                GramR G;
                gram_init_rf<NB>(c, R, sp, G);
                for (int k = 0; k < K; ++k) {
                    double a_, c_;
                    gram_sweep_rf(G, sp.S, p, av, a_, c_);
                }
                mle_fast_rf<NB, false, true>(c, R, F0, sp, p, av, nullptr, m);
            } else {
                Gram G;
                gram_init<NB>(c, sp, G);
                for (int k = 0; k < K; ++k) {
                    double a_, b_, c_;
                    gram_sweep(G, sp.S, p, av, rv, a_, b_, c_);
                }
                mle_fast<NB, false>(c, F0, sp, p, av, rv, nullptr, m);
            }
            rec[0] = final_lnl(sp, p, m.chi2, false);
            rec[1] = m.chi2;
            rec[2] = m.scale;
            rec[3] = av;
            rec[4] = rv;
            rec[5] = m.i00;
            rec[6] = m.i01;
            rec[7] = m.i02;
            rec[8] = m.i11;
            rec[9] = m.i12;
            rec[10] = m.i22;
        }
#pragma unroll
        for (int k = 0; k < BRUTUS_NVALS; ++k) sel_vals[(int64_t)k * capacity + q] = rec[k];
    }
}

// ===========================================================================
// cluster.isochrone_loglike hot block (reference cluster.py:336-414)
// ===========================================================================
// For every object o and every isochrone point c (all secondary-mass-fraction
// slices concatenated): chi2 = nansum_b (phot_ob - flux_cb)^2 / err_ob^2 + chi2_p,
// lnl = chi2-logpdf(chi2, n_o) or -(chi2 + lnorm_o)/2, then
// lnl_o = logsumexp_c (lnl + lnw_c).  One lane = one object (its bands in
// VGPRs), isochrone points are wave-uniform (scalar loads); the point axis is
// split over blockIdx.y and merged by k_cluster_merge (online logsumexp).
template <int NB>
__global__ void __launch_bounds__(64)
k_cluster(int nobj, int nb, int npts, const double *__restrict__ pts_flux,
          const double *__restrict__ pts_lnw, const double *__restrict__ phot,
          const double *__restrict__ ivar, const double *__restrict__ chi2_p,
          const double *__restrict__ lnorm, const int32_t *__restrict__ ndim, int dim_prior,
          int pts_per_block, double *__restrict__ part_m, double *__restrict__ part_s) {
    const int o = blockIdx.x * 64 + threadIdx.x;
    const bool live = o < nobj;
    const int oo = live ? o : 0;
    double d[NB], iv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        d[b] = b < nb ? phot[(int64_t)oo * nb + b] : 0.;
        iv[b] = b < nb ? ivar[(int64_t)oo * nb + b] : 0.;
    }
    const double cp = chi2_p[oo], ln0 = lnorm[oo];
    const double k = (double)ndim[oo];
    const double c0 = -(k / 2.) * 0.69314718055994530942 - lgamma(k / 2.);
    const double c1 = k / 2. - 1.;
    const int p0 = blockIdx.y * pts_per_block;
    const int p1 = min(npts, p0 + pts_per_block);
    double m = -INFINITY, ssum = 0.;
    for (int c = p0; c < p1; ++c) {
        const double *f = pts_flux + (int64_t)c * nb;
        double chi2 = 0.;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (b < nb) {
                const double t = d[b] - f[b];
                const double term = t * t * iv[b];
                if (term == term) chi2 += term;          // nansum (cluster.py:381)
            }
        }
        chi2 += cp;
        double lnl;
        if (dim_prior)   // scipy.stats.chi2.logpdf(chi2, k) (cluster.py:389)
            lnl = c0 + (c1 == 0. ? 0. : c1 * log(chi2)) - chi2 / 2.;
        else
            lnl = -0.5 * (chi2 + ln0);
        if (!isfinite(lnl)) lnl = -INFINITY;             // cluster.py:394
        const double x = lnl + pts_lnw[c];
        if (x > m) {
            ssum = ssum * exp(m - x) + 1.;
            m = x;
        } else if (x > -INFINITY) {
            ssum += exp(x - m);
        }
    }
    if (live) {
        part_m[(int64_t)blockIdx.y * nobj + o] = m;
        part_s[(int64_t)blockIdx.y * nobj + o] = ssum;
    }
}

__global__ void k_cluster_merge(int nobj, int nchunk, const double *__restrict__ part_m,
                                const double *__restrict__ part_s, double *__restrict__ out) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nobj) return;
    double m = -INFINITY;
    for (int c = 0; c < nchunk; ++c) {
        const double x = part_m[(int64_t)c * nobj + o];
        m = x > m ? x : m;
    }
    double ssum = 0.;
    for (int c = 0; c < nchunk; ++c) {
        const double x = part_m[(int64_t)c * nobj + o];
        if (x > -INFINITY) ssum += part_s[(int64_t)c * nobj + o] * exp(x - m);
    }
    out[o] = m > -INFINITY ? m + log(ssum) : -INFINITY;
}

// ===========================================================================
// lnpost on the device (fitting.py:1000-1107 and the tail of _fit, :2021-2061)
// for the built-in priors, with the counter-based random stream specified in
// brutus_amd/rng.py (Philox4x32-7 + polar normals): any deviate is a pure
// function of (seed, index), so every selected model of every object is
// integrated in parallel and the result still equals, deviate for deviate, a
// sequential run of the reference with that `rstate` object.
// ===========================================================================
struct Philox4 {
    uint32_t w[4];
};

__device__ __forceinline__ Philox4 philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        if (r > 0) {
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;   // v_mad_u64_u32
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
    }
    Philox4 o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// q-th uniform of the uniform stream (rng.py: philox_uniform)
__device__ __forceinline__ double rng_uniform(uint64_t seed, uint64_t q) {
    const Philox4 o = philox4x32_7((uint32_t)q, (uint32_t)(q >> 32), 0u, 1u, (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
    return u53(o.w[0], o.w[1]);
}

// candidate `retry` of pair p of the normal stream (rng.py: philox_normal);
// true if the polar method accepts it
__device__ __forceinline__ bool rng_polar_candidate(uint64_t seed, uint64_t p, uint32_t retry,
                                                    double &x1, double &x2) {
#pragma clang fp contract(off)   // r2 must round like numpy's x1*x1 + x2*x2 (accept/reject!)
    const Philox4 o = philox4x32_7((uint32_t)p, (uint32_t)(p >> 32), retry, 0u, (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
    x1 = 2.0 * u53(o.w[0], o.w[1]) - 1.0;
    x2 = 2.0 * u53(o.w[2], o.w[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    return r2 < 1.0 && r2 > 0.0;
}
// the two normals of an accepted candidate
__device__ __forceinline__ void rng_polar_finish(double x1, double x2, double &z0, double &z1) {
#pragma clang fp contract(off)
    const double r2 = x1 * x1 + x2 * x2;
    // rng.py: f = sqrt(-2 ln(r2) / r2).  ln, the divide and the root are the ~1 ulp
    // Newton forms (the accept / reject decision above is what must be exact): the
    // normals agree with numpy's to a few ulp at a third of the IEEE sequences' cost.
    const double f = fast_sqrt(-2.0 * fast_log_r(r2) * fast_rcp(r2));
    z0 = f * x1;
    z1 = f * x2;
}
// pair p of the normal stream: z0 = normal 2p, z1 = normal 2p+1
__device__ __forceinline__ void rng_normal_pair(uint64_t seed, uint64_t p, double &z0, double &z1) {
    // The retry loop only draws candidates (lanes of a wave retry in lockstep:
    // ~3.5 rounds for 64 lanes at 21 % rejection); ln / sqrt / divide run once.
    double x1, x2;
    for (uint32_t retry = 0; !rng_polar_candidate(seed, p, retry, x1, x2); ++retry) {}
    rng_polar_finish(x1, x2, z0, z1);
}
__device__ __forceinline__ double rng_normal(uint64_t seed, uint64_t j) {
    double z0, z1;
    rng_normal_pair(seed, j >> 1, z0, z1);
    return (j & 1) ? z1 : z0;
}

struct PostParams {     // mirrors brutus_post_params
    int32_t nmc, ndraws, return_distreds, has_feh, has_loga, per_object;
    double wt_thresh, avlim[2], rvlim[2];
    int64_t nsel_max, object0;
    uint64_t seed, normal_base, uniform_base;
    double R_solar, Z_solar, R_thin, Z_thin, Rs_thin, R_thick, Z_thick, f_thick, Rs_thick;
    double Rs_halo, q_halo_ctr, q_halo_inf, r_q_halo, eta_halo, f_halo;
    double feh_mean[3], feh_sigma[3];
    double age_mean[3], age_sigma[3], age_lnnorm[3], min_age, max_age;
    // derived on the host side of the ABI call (not part of brutus_post_params)
    double ln_f_thick, ln_f_halo, inv_reff_solar2;
    double inv_R_thin, inv_Z_thin, inv_R_thick, inv_Z_thick, inv_r_q;
    double Rs_thin2, Rs_thick2, Rs_halo2, rq2, abs_Z_solar;
    double lnK, c0_thin, c0_thick, c0_halo;      // component constants relative to lnK
};
constexpr int POST_DERIVED = 17;

// stream key and uniform base of object s: one shared sequential stream, or
// (per_object) an own stream keyed seed + object index
__device__ __forceinline__ uint64_t star_seed(const PostParams &pp, int s) {
    return pp.per_object ? pp.seed + (uint64_t)(pp.object0 + s) : pp.seed;
}
__device__ __forceinline__ uint64_t star_ubase(const PostParams &pp, int s) {
    return pp.per_object ? 0ull
                         : pp.uniform_base + (uint64_t)s * (uint64_t)(pp.ndraws * (pp.return_distreds ? 2 : 1));
}

struct StarGeom {      // per object: sightline unit vector and parallax
    double cb_cl, cb_sl, sb;     // cos b cos l, cos b sin l, sin b
    double par, par_ivar, par_lnorm;
    int has_par;
};

__device__ __forceinline__ double lse3(double a, double b, double c) {
    double m = a > b ? a : b;
    m = c > m ? c : m;
    if (!(m > -INFINITY)) return m;          // all -inf (or NaN)
    return log(exp(a - m) + exp(b - m) + exp(c - m)) + m;
}

// per-model metallicity / age densities of the three components (pdf.py:380-473),
// as plain (not log) values: e^F_c, e^A_c
__device__ __forceinline__ void label_terms(const PostParams &pp, double feh, double loga,
                                            double (&Fc)[3], double (&Ac)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Fc[c] = 1.;
        Ac[c] = 1.;
        if (pp.has_feh) {
            const double d = pp.feh_mean[c] - feh;
            Fc[c] = exp(-0.5 * (d * d / (pp.feh_sigma[c] * pp.feh_sigma[c]) +
                                log(2. * M_PI * pp.feh_sigma[c] * pp.feh_sigma[c])));
        }
        if (pp.has_loga) {
            const double age = exp10(loga) / 1e9;
            const double xi = (age - pp.age_mean[c]) / pp.age_sigma[c];
            Ac[c] = (age < pp.min_age || age > pp.max_age)
                        ? 0.
                        : exp(-0.91893853320467274178 - 0.5 * xi * xi - pp.age_lnnorm[c]);
        }
    }
}

// gal_lnprior (brutus_amd/galprior.py, reference pdf.py:476-749) at distance d [kpc],
// as a plain density relative to e^lnK: gal_lnprior = lnK + ln(gal_prior_lin).
// With T_c = exp(comp_c - lnK) the three log-sum-exps of the reference collapse:
//   lse(comp) + [lse(F + comp) - lse(comp)] + [lse(A + comp) - lse(comp)]
//     = lnK + ln( (sum T_c e^F_c) (sum T_c e^A_c) / sum T_c )
// EF_c = e^F_c, EA_c = e^A_c are per-model constants.  lnK (fill_post_params) is
// an upper bound of every comp_c, so no T_c overflows, and the halo's power law
// keeps the sum away from underflow at any distance: no running maximum needed.
// Cost per call: 3 exp + 1 log (halo power) + 3 sqrt + 2 reciprocals.
__device__ __forceinline__ double gal_prior_lin(const PostParams &pp, const StarGeom &g, double d,
                                                const double (&EF)[3], const double (&EA)[3],
                                                const double *__restrict__ tbl) {
    const double x = pp.R_solar - d * g.cb_cl, y = d * g.cb_sl, Z = pp.Z_solar + d * g.sb;
    const double R2 = x * x + y * y;
    const double dZ = fabs(Z) - pp.abs_Z_solar;
    const double Rt = fast_sqrt(R2 + pp.Rs_thin2);
    const double Rk = pp.Rs_thick2 == pp.Rs_thin2 ? Rt : fast_sqrt(R2 + pp.Rs_thick2);
    // thin / thick disk: exp(-(R - R_sun)/R_c - (|Z| - |Z_sun|)/Z_c [+ ln f] - lnK)
    const double T0 = fast_exp_bf(pp.c0_thin - (Rt * pp.inv_R_thin + dZ * pp.inv_Z_thin), tbl);
    const double T1 = fast_exp_bf(pp.c0_thick - (Rk * pp.inv_R_thick + dZ * pp.inv_Z_thick), tbl);
    // halo: f (reff / reff_sun)^-eta, reff^2 = R^2 + (Z/q)^2 + Rs^2, q(r) (pdf.py:341-365)
    const double q = pp.q_halo_inf -
                     (pp.q_halo_inf - pp.q_halo_ctr) *
                         fast_exp_bf(1. - fast_sqrt(R2 + Z * Z + pp.rq2) * pp.inv_r_q, tbl);
    const double zq = Z * fast_rcp(q);
    const double T2 = fast_exp_bf(
        pp.c0_halo - 0.5 * pp.eta_halo * fast_log_r((R2 + zq * zq + pp.Rs_halo2) * pp.inv_reff_solar2), tbl);
    double num = d * d + 1e-300;                    // volume factor (pdf.py:626)
    if (pp.has_feh) num *= T0 * EF[0] + T1 * EF[1] + T2 * EF[2];
    if (pp.has_loga) num *= T0 * EA[0] + T1 * EA[1] + T2 * EA[2];
    const double S = T0 + T1 + T2;
    const int npow = (pp.has_feh ? 1 : 0) + (pp.has_loga ? 1 : 0);
    // divide by S^(npow - 1): one S stays for lse(comp) itself
    if (npow == 2) num *= fast_rcp(S);
    else if (npow == 0) num *= S;
    return num;
}
__device__ __forceinline__ double gal_lnprior_dev(const PostParams &pp, const StarGeom &g, double d,
                                                  const double (&EF)[3], const double (&EA)[3],
                                                  const double *__restrict__ tbl) {
    return pp.lnK + fast_log_r(gal_prior_lin(pp, g, d, EF, EA, tbl));
}

constexpr int PCH = 64;      // chunks per object for the record passes

// first membership word of object s (objects' 256-record tiles do not share words)
__device__ __forceinline__ int64_t mask_base(const int64_t *__restrict__ off, int s) {
    return ((off[s] + 63) >> 6) + 8 * (int64_t)s;
}

// record range of workgroup (chunk c, star s): 256-aligned slices of [off[s], off[s+1])
__device__ __forceinline__ void rec_range(const int64_t *__restrict__ off, int s, int c, int64_t &a,
                                          int64_t &b) {
    const int64_t lo = off[s], n = off[s + 1] - lo;
    const int64_t ntile = (n + TILE - 1) / TILE;
    a = lo + (ntile * c / PCH) * TILE;
    b = lo + (ntile * (c + 1) / PCH) * TILE;
    if (b > lo + n) b = lo + n;
    if (a > lo + n) a = lo + n;
}

__device__ __forceinline__ void rec_range_n(int64_t lo, int64_t n, int c, int64_t &a, int64_t &b) {
    const int64_t ntile = (n + TILE - 1) / TILE;
    a = lo + (ntile * c / PCH) * TILE;
    b = lo + (ntile * (c + 1) / PCH) * TILE;
    if (b > lo + n) b = lo + n;
    if (a > lo + n) a = lo + n;
}

// P1: lnp of the MLE point for the second cut (fitting.py:1000-1010)
__global__ void __launch_bounds__(TILE)
k_post_lnp1(PostParams pp, int64_t cap, const int32_t *__restrict__ sel_idx,
            const double *__restrict__ sel_vals, const int64_t *__restrict__ sel_off,
            const StarGeom *__restrict__ geom, const double *__restrict__ lnprior,
            const double *__restrict__ feh, const double *__restrict__ loga,
            double *__restrict__ lnp1, double *__restrict__ part) {
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    const int s = blockIdx.y, c = blockIdx.x;
    int64_t a, b;
    rec_range(sel_off, s, c, a, b);
    const StarGeom g = geom[s];
    double m = -INFINITY;
    for (int64_t r0 = a; r0 < b; r0 += TILE) {
        const int64_t r = r0 + threadIdx.x;
        if (r < b) {
            const int64_t i = sel_idx[r];
            double Fc[3], Ac[3];
            label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac);
            const double scale = sel_vals[2 * cap + r];
            const double v = sel_vals[r] + lnprior[i] + gal_lnprior_dev(pp, g, 1. / sqrt(scale), Fc, Ac, s_tbl);
            lnp1[r] = v;
            if (v > m) m = v;
        }
    }
    block_max_store(m, slot, part + (int64_t)s * PCH + c);
}

// P2a: second cut (fitting.py:1013-1016): count + membership words
__global__ void __launch_bounds__(TILE)
k_post_count2(double ln_wt, const int64_t *__restrict__ sel_off, const double *__restrict__ lnp1,
              const double *__restrict__ part, int64_t *__restrict__ counts,
              unsigned long long *__restrict__ mask) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    double mx = -INFINITY;
    for (int q = 0; q < PCH; ++q) {
        const double v = part[(int64_t)s * PCH + q];
        mx = v > mx ? v : mx;
    }
    const double thr = mx + ln_wt;
    int64_t a, b;
    rec_range(sel_off, s, c, a, b);
    int n = 0;
    for (int64_t r0 = a; r0 < b; r0 += TILE) {
        const int64_t r = r0 + threadIdx.x;
        const bool hit = r < b && lnp1[r] > thr;
        n += hit ? 1 : 0;
        const unsigned long long bl = __ballot(hit);
        if ((threadIdx.x & 63) == 0)
            mask[mask_base(sel_off, s) + ((r0 - sel_off[s]) >> 6) + (threadIdx.x >> 6)] = bl;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)s * PCH + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// P2b: offsets of the second-cut lists; normal-stream base of every object
// (3 * nmc normals per kept model, objects in order: exactly what a sequential
// rstate would have consumed), host-fallback flags.
__global__ void k_post_offsets(PostParams pp, int nstar, const int64_t *__restrict__ counts,
                               int64_t *__restrict__ offsets, int64_t *__restrict__ off2,
                               uint64_t *__restrict__ nbase, int32_t *__restrict__ flags,
                               int64_t *__restrict__ nsel) {
    __shared__ int64_t tot[BRUTUS_MAX_BATCH + 1];
    const int s = threadIdx.x;
    int64_t n = 0;
    if (s < nstar)
        for (int c = 0; c < PCH; ++c) n += counts[(int64_t)s * PCH + c];
    if (s < nstar) tot[s] = n;
    __syncthreads();
    if (s == 0) {
        int64_t run = 0;
        uint64_t nb = pp.normal_base;
        for (int q = 0; q < nstar; ++q) {
            const int64_t m = tot[q];
            tot[q] = run;
            off2[q] = run;
            nbase[q] = pp.per_object ? 0ull : nb;
            const int64_t used = m > pp.nsel_max ? pp.nsel_max : m;   // fitting.py:1029-1036
            flags[q] = m > pp.nsel_max ? 1 : 0;
            nsel[q] = used;
            nb += (uint64_t)(3 * (int64_t)pp.nmc * used);
            run += m;
        }
        off2[nstar] = run;
        nbase[nstar] = nb;
    }
    __syncthreads();
    if (s < nstar) {
        int64_t run = tot[s];
        for (int c = 0; c < PCH; ++c) {
            offsets[(int64_t)s * PCH + c] = run;
            run += counts[(int64_t)s * PCH + c];
        }
    }
}

// P2c: ordered scatter of the kept records + per-record preparation
// (fitting.py:1023, 1039-1065): lnp0 = lnlike + lnprior, covariance by the
// adjugate, PSD repair, Cholesky factor of cov + 1e-30 I.
struct RecPost {      // arrays over second-cut records (capacity = first-cut capacity)
    int32_t *src;     // position in the first-cut record arrays
    double *lnp;      // lnp0, later the final lnp
    double *cov;      // [6][cap]
    double *chol;     // [6][cap]  L00 L10 L11 L20 L21 L22
};

__device__ __forceinline__ bool inv3_sym(const double (&A)[6], double (&C)[6]) {
    // A, C: 00 01 02 11 12 22.  Adjugate by row cross products, determinant as
    // the mean of the three row.cofactor-row dots (utils.py:71-114).
    const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
    const double c00 = a11 * a22 - a12 * a12, c01 = a12 * a02 - a01 * a22, c02 = a01 * a12 - a11 * a02;
    const double c11 = a22 * a00 - a02 * a02, c12 = a02 * a01 - a12 * a00;
    const double c22 = a00 * a11 - a01 * a01;
    const double d0 = c00 * a00 + c01 * a01 + c02 * a02;
    const double d1 = c01 * a01 + c11 * a11 + c12 * a12;
    const double d2 = c02 * a02 + c12 * a12 + c22 * a22;
    const double det = (d0 + d1 + d2) / 3.;
    C[0] = c00 / det; C[1] = c01 / det; C[2] = c02 / det;
    C[3] = c11 / det; C[4] = c12 / det; C[5] = c22 / det;
    return true;
}

__device__ __forceinline__ bool is_pd3(const double (&C)[6]) {
    // all eigenvalues > 0 (fitting.py:1042) <=> leading principal minors > 0
    const double m2 = C[0] * C[3] - C[1] * C[1];
    const double m3 = C[0] * (C[3] * C[5] - C[4] * C[4]) - C[1] * (C[1] * C[5] - C[4] * C[2]) +
                      C[2] * (C[1] * C[4] - C[3] * C[2]);
    return C[0] > 0. && m2 > 0. && m3 > 0.;
}

__global__ void __launch_bounds__(TILE)
k_post_scatter2(int64_t cap, const int32_t *__restrict__ sel_idx, const double *__restrict__ sel_vals,
                const int64_t *__restrict__ sel_off, const double *__restrict__ lnprior,
                const unsigned long long *__restrict__ mask, const int64_t *__restrict__ offsets,
                RecPost rp) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    int64_t a, b;
    rec_range(sel_off, s, c, a, b);
    int64_t base = offsets[(int64_t)s * PCH + c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t r0 = a; r0 < b; r0 += TILE) {
        const int64_t r = r0 + threadIdx.x;
        const unsigned long long bl = mask[mask_base(sel_off, s) + ((r0 - sel_off[s]) >> 6) + w];
        const bool sel = (bl >> lane) & 1ull;
        const int rank = __popcll(bl & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(bl);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < w; ++q) woff += wsum[q];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (sel) {
            const int64_t o = base + woff + rank;
            rp.src[o] = (int32_t)(r - sel_off[s]);
            rp.lnp[o] = sel_vals[r] + lnprior[sel_idx[r]];
            double A[6], C[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) A[k] = sel_vals[(int64_t)(5 + k) * cap + r];
            inv3_sym(A, C);
            const double scale = sel_vals[2 * cap + r];
            const double width = 0.02;
            double count = 1.;
            for (int it = 0; it < 200 && !is_pd3(C); ++it) {       // fitting.py:1045-1065
                const double sf = scale * width;
                const bool i1 = C[0] <= 0., i2 = C[3] <= 0., i3 = C[5] <= 0.;
                if (i1 || (!i2 && !i3)) A[0] += count / (sf * sf);
                if (i2 || (!i1 && !i3)) A[3] += count / (width * width);
                if (i3 || (!i1 && !i2)) A[5] += count / (width * width);
                inv3_sym(A, C);
                count *= 2.;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) rp.cov[(int64_t)k * cap + o] = C[k];
            // Cholesky of cov + 1e-30 I (utils.py:892-894)
            const double l00 = sqrt(C[0] + 1e-30);
            const double l10 = C[1] / l00, l20 = C[2] / l00;
            const double l11 = sqrt(C[3] + 1e-30 - l10 * l10);
            const double l21 = (C[4] - l20 * l10) / l11;
            const double l22 = sqrt(C[5] + 1e-30 - l20 * l20 - l21 * l21);
            rp.chol[0 * cap + o] = l00;
            rp.chol[1 * cap + o] = l10;
            rp.chol[2 * cap + o] = l11;
            rp.chol[3 * cap + o] = l20;
            rp.chol[4 * cap + o] = l21;
            rp.chol[5 * cap + o] = l22;
        }
        base += tot;
        __syncthreads();
    }
}

// Sequential reader of normals j0, j0+1, ...: each Philox pair is generated once.
struct NormalReader {
    uint64_t seed, p;
    double z0, z1;
    bool have;
    __device__ __forceinline__ void init(uint64_t seed_) {
        seed = seed_;
        have = false;
        p = 0;
    }
    __device__ __forceinline__ double at(uint64_t j) {
        const uint64_t q = j >> 1;
        if (!have || q != p) {
            rng_normal_pair(seed, q, z0, z1);
            p = q;
            have = true;
        }
        return (j & 1) ? z1 : z0;
    }
};

// One Monte Carlo sample of a kept record (fitting.py:1071-1093) from its three
// normals: sample t of the record of rank n in the object's list uses normals
// nbase + (3 n + k) nmc + t, k = 0, 1, 2 (utils.py:897).  Returns (dist, av, rv),
// whether it is inside the fit bounds, and its prior in split form:
//   lnp_mc = lnK + ln(lin) + epar - par_lnorm / 2,   epar = -(par - par_obs)^2 ivar / 2 <= 0
__device__ __forceinline__ void mc_sample_lin(const PostParams &pp, const StarGeom &g, double z0,
                                              double z1, double z2, double s0, double a0, double r0,
                                              const double (&L)[6], const double (&Fc)[3],
                                              const double (&Ac)[3], const double *__restrict__ tbl,
                                              double &dist, double &a_mc, double &r_mc, bool &inb,
                                              double &lin, double &epar) {
    const double s_mc = s0 + L[0] * z0;
    a_mc = a0 + (L[1] * z0 + L[2] * z1);
    r_mc = r0 + (L[3] * z0 + L[4] * z1 + L[5] * z2);
    double par;
    fast_sqrt_rsqrt(s_mc, par, dist);                           // parallax and distance (~1 ulp)
    lin = gal_prior_lin(pp, g, dist, Fc, Ac, tbl);
    const double dp = par - g.par;                              // pdf.py:166-173
    epar = g.has_par ? -0.5 * (dp * dp * g.par_ivar) : 0.;
    inb = s_mc >= 1e-20 && a_mc >= pp.avlim[0] && a_mc <= pp.avlim[1] &&
          r_mc >= pp.rvlim[0] && r_mc <= pp.rvlim[1];
}

// the same as one log value (-BIG outside the bounds, fitting.py:1086-1090),
// drawing the normals on the fly
__device__ __forceinline__ double mc_sample(const PostParams &pp, NormalReader (&rd)[3],
                                            const StarGeom &g, uint64_t nb, int64_t n, int t, double s0, double a0, double r0,
                                            const double (&L)[6], const double (&Fc)[3],
                                            const double (&Ac)[3], const double *__restrict__ tbl,
                                            double &dist, double &a_mc, double &r_mc, bool &inb) {
    const uint64_t j0 = nb + (uint64_t)((3 * n) * (int64_t)pp.nmc + t);
    const double z0 = rd[0].at(j0);
    const double z1 = rd[1].at(j0 + (uint64_t)pp.nmc);
    const double z2 = rd[2].at(j0 + 2ull * (uint64_t)pp.nmc);
    double lin, epar;
    mc_sample_lin(pp, g, z0, z1, z2, s0, a0, r0, L, Fc, Ac, tbl, dist, a_mc, r_mc, inb, lin, epar);
    double v = pp.lnK + fast_log_r(lin);
    if (g.has_par) v += epar - 0.5 * g.par_lnorm;
    if (!inb) v = -BIG;
    return v;
}

// rows (polar pairs) of a k_post_mc staging slot: the 3 nmc normals of a record
// span at most 3 nmc / 2 + 1 pairs
__host__ __device__ inline int mc_npair_max(int nmc) { return (3 * nmc) / 2 + 2; }

// P4: Monte Carlo prior integral of every kept record (fitting.py:1068-1105)
// and chi2min (fitting.py:2025-2034).  One lane per record.
//
// The 3 nmc normals of a record are one contiguous run of the stream, i.e.
// ~3 nmc / 2 polar pairs.  A lane first walks its pairs with its own retry
// counter and stores the two normals of each accepted candidate in its column
// of `zs` (lane-interleaved rows of double2; 16-byte stores: the staging is
// bound by L2 write requests, lanes drift apart in row) -- a
// wave then spends ~1/0.785 Philox rounds per pair instead of the ~3.7 it takes
// until all 64 lanes of a lockstep retry loop have accepted -- and afterwards
// integrates, reading three normals per sample.
//
__global__ void __launch_bounds__(TILE, 3)
k_post_mc(PostParams pp, int64_t cap, int nitem, unsigned int *__restrict__ counter,
          double2 *__restrict__ zs, const int32_t *__restrict__ sel_idx,
          const double *__restrict__ sel_vals, const int64_t *__restrict__ sel_off,
          const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
          const uint64_t *__restrict__ nbase, const int32_t *__restrict__ flags,
          const StarGeom *__restrict__ geom, const double *__restrict__ feh,
          const double *__restrict__ loga, RecPost rp, double *__restrict__ part_max,
          double *__restrict__ part_chi2) {
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    __shared__ unsigned int s_item;
    stage_exp_table(s_tbl);
    double2 *const col = zs + (int64_t)blockIdx.x * mc_npair_max(pp.nmc) * TILE + threadIdx.x;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned int item = s_item;
        if (item >= (unsigned int)nitem) break;
        const int s = (int)(item / PCH), c = (int)(item % PCH);
        int64_t a, b;
        rec_range_n(off2[s], nsel[s], c, a, b);
        const StarGeom g = geom[s];
        const uint64_t nb = nbase[s];
        const uint64_t seed = star_seed(pp, s);
        double mx = -INFINITY, cmin = -INFINITY;   // cmin holds -min(chi2)
        if (!flags[s]) {
            for (int64_t o0 = a; o0 < b; o0 += TILE) {
                const int64_t o = o0 + threadIdx.x;
                const bool live = o < b;
                const int64_t n = o - off2[s];
                // normals j_lo .. j_lo + 3 nmc - 1 of the stream = pairs p_lo .. p_hi;
                // pair q - p_lo goes to row q - p_lo of the slot as one 16-byte store
                const uint64_t j_lo = nb + (uint64_t)(3 * n * (int64_t)pp.nmc);
                const uint64_t p_lo = j_lo >> 1;
                {
                    const uint64_t p_hi = (j_lo + (uint64_t)(3 * pp.nmc) - 1) >> 1;
                    uint64_t p = live ? p_lo : p_hi + 1;
                    uint32_t retry = 0;
                    while (p <= p_hi) {
                        double x1, x2;
                        if (rng_polar_candidate(seed, p, retry, x1, x2)) {
                            double z0, z1;
                            rng_polar_finish(x1, x2, z0, z1);
                            col[(int64_t)(p - p_lo) * TILE] = make_double2(z0, z1);
                            ++p;
                            retry = 0;
                        } else {
                            ++retry;
                        }
                    }
                }
                if (live) {
                    const int64_t r = sel_off[s] + rp.src[o];
                    const int64_t i = sel_idx[r];
                    double Fc[3], Ac[3], L[6];
                    label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac);
#pragma unroll
                    for (int k = 0; k < 6; ++k) L[k] = rp.chol[(int64_t)k * cap + o];
                    const double s0 = sel_vals[2 * cap + r], a0 = sel_vals[3 * cap + r],
                                 r0 = sel_vals[4 * cap + r];
                    // sum_t lin_t e^{epar_t} over the in-bounds samples, with a
                    // running maximum M of epar only (lin needs none); branch-free
                    double M = -INFINITY, acc = 0.;
                    int ninb = 0;
                    const double *const zc = (const double *)col;
                    const int jb = (int)(j_lo & 1);
                    for (int t = 0; t < pp.nmc; ++t) {
                        double d_, a_, r_, lin, epar;
                        bool inb;
                        // normal jj of the run: component jj & 1 of row jj >> 1
                        const int j0 = jb + t, j1 = j0 + pp.nmc, j2 = j1 + pp.nmc;
                        mc_sample_lin(pp, g, zc[(int64_t)(j0 >> 1) * (2 * TILE) + (j0 & 1)],
                                      zc[(int64_t)(j1 >> 1) * (2 * TILE) + (j1 & 1)],
                                      zc[(int64_t)(j2 >> 1) * (2 * TILE) + (j2 & 1)], s0, a0, r0, L, Fc, Ac,
                                      s_tbl, d_, a_, r_, inb, lin, epar);
                        ninb += inb ? 1 : 0;
                        if (g.has_par) {
                            const double dM = epar - M;
                            const double ex = fast_exp_bf(-fabs(dM), s_tbl);
                            const bool up = inb && dM > 0.;
                            const double add = inb ? lin : 0.;
                            acc = up ? fma(acc, ex, add) : (inb ? fma(add, ex, acc) : acc);
                            M = up ? epar : M;
                        } else {
                            acc += inb ? lin : 0.;
                        }
                    }
                    // logsumexp(lnp_mc) - ln(#in bounds), fitting.py:1094-1102; with no
                    // sample in bounds the reference yields +inf -> not finite -> -BIG
                    double lse = pp.lnK + log(acc);
                    if (g.has_par) lse += M - 0.5 * g.par_lnorm;
                    double lnp = ninb > 0 ? rp.lnp[o] + (lse - log((double)ninb)) : nan("");
                    if (!isfinite(lnp)) lnp = -BIG;                       // fitting.py:1103-1105
                    rp.lnp[o] = lnp;
                    if (lnp > mx) mx = lnp;
                    double chi2 = sel_vals[1 * cap + r];
                    if (g.has_par) {
                        const double dp = sqrt(s0) - g.par;
                        chi2 += dp * dp * g.par_ivar;
                    }
                    if (-chi2 > cmin) cmin = -chi2;
                }
            }
        }
        block_max_store(mx, slot, part_max + (int64_t)s * PCH + c);
        block_max_store(cmin, slot, part_chi2 + (int64_t)s * PCH + c);
    }
}

// P5: evidence and the cumulative weights of one object (fitting.py:2033-2038),
// chunk-parallel over grid (PCH, object):
//   k_post_evid_part : per-chunk sums of exp(lnp - max)          -> levid
//   k_post_wt_part   : per-chunk totals of wt = exp(lnp - levid)
//   k_post_cdf       : chunk offset + in-chunk running sum       -> cdf
// The chunk totals come from the very scan that later writes the cdf, so the
// cdf is monotone across chunk boundaries bit for bit.
__device__ __forceinline__ void post_star_max(const double *__restrict__ part_max,
                                              const double *__restrict__ part_chi2, int s, double &mx,
                                              double &cm) {
    mx = -INFINITY;
    cm = -INFINITY;
    for (int q = 0; q < PCH; ++q) {
        mx = fmax(mx, part_max[(int64_t)s * PCH + q]);
        cm = fmax(cm, part_chi2[(int64_t)s * PCH + q]);
    }
}

__global__ void __launch_bounds__(TILE)
k_post_evid_part(const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
                 const int32_t *__restrict__ flags, const double *__restrict__ part_max,
                 const double *__restrict__ part_chi2, RecPost rp, double *__restrict__ part_e) {
    __shared__ double sh[TILE];
    const int s = blockIdx.y, c = blockIdx.x;
    if (flags[s]) return;
    int64_t a, b;
    rec_range_n(off2[s], nsel[s], c, a, b);
    double mx, cm;
    post_star_max(part_max, part_chi2, s, mx, cm);
    double acc = 0.;
    for (int64_t o = a + threadIdx.x; o < b; o += TILE) acc += exp(rp.lnp[o] - mx);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = TILE / 2; st > 0; st >>= 1) {
        if (threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part_e[(int64_t)s * PCH + c] = sh[0];
}

// log-evidence of object s from the chunk sums (same order in every caller)
__device__ __forceinline__ double post_levid(const double *__restrict__ part_e, int s, double mx) {
    double tot = 0.;
    for (int q = 0; q < PCH; ++q) tot += part_e[(int64_t)s * PCH + q];
    return log(tot) + mx;
}

// running sum of wt over records [a, b) starting from `carry0`; returns the
// final carry (all threads).  WRITE: store the inclusive sums to cdf.
template <bool WRITE>
__device__ __forceinline__ double post_chunk_scan(const double *__restrict__ lnp, int64_t a, int64_t b,
                                                  double levid, double carry0, double *sh,
                                                  double *__restrict__ cdf) {
    double carry = carry0;
    for (int64_t o0 = a; o0 < b; o0 += TILE) {
        const int64_t o = o0 + threadIdx.x;
        const double w = o < b ? exp(lnp[o] - levid) : 0.;
        sh[threadIdx.x] = w;
        __syncthreads();
        for (int st = 1; st < TILE; st <<= 1) {          // Hillis-Steele inclusive scan
            const double v = threadIdx.x >= st ? sh[threadIdx.x - st] : 0.;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        if (WRITE && o < b) cdf[o] = carry + sh[threadIdx.x];
        carry += sh[TILE - 1];
        __syncthreads();
    }
    return carry;
}

__global__ void __launch_bounds__(TILE)
k_post_wt_part(const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
               const int32_t *__restrict__ flags, const double *__restrict__ part_max,
               const double *__restrict__ part_chi2, const double *__restrict__ part_e, RecPost rp,
               double *__restrict__ part_w) {
    __shared__ double sh[TILE];
    const int s = blockIdx.y, c = blockIdx.x;
    if (flags[s]) return;
    int64_t a, b;
    rec_range_n(off2[s], nsel[s], c, a, b);
    double mx, cm;
    post_star_max(part_max, part_chi2, s, mx, cm);
    const double levid = post_levid(part_e, s, mx);
    const double tot = post_chunk_scan<false>(rp.lnp, a, b, levid, 0., sh, nullptr);
    if (threadIdx.x == 0) part_w[(int64_t)s * PCH + c] = tot;
}

__global__ void __launch_bounds__(TILE)
k_post_cdf(const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
           const int32_t *__restrict__ flags, const double *__restrict__ part_max,
           const double *__restrict__ part_chi2, const double *__restrict__ part_e,
           const double *__restrict__ part_w, RecPost rp, double *__restrict__ cdf,
           double *__restrict__ star_out) {
    __shared__ double sh[TILE];
    const int s = blockIdx.y, c = blockIdx.x;
    if (flags[s]) return;
    int64_t a, b;
    rec_range_n(off2[s], nsel[s], c, a, b);
    double mx, cm;
    post_star_max(part_max, part_chi2, s, mx, cm);
    const double levid = post_levid(part_e, s, mx);
    double carry = 0.;                         // offset of this chunk: its predecessors' totals
    for (int q = 0; q < c; ++q) carry += part_w[(int64_t)s * PCH + q];
    carry = post_chunk_scan<true>(rp.lnp, a, b, levid, carry, sh, cdf);
    if (c == PCH - 1 && threadIdx.x == 0) {
        star_out[4 * s + 0] = levid;
        star_out[4 * s + 1] = -cm;            // chi2min
        star_out[4 * s + 2] = carry;          // total weight (cdf normaliser)
        star_out[4 * s + 3] = (double)nsel[s];
    }
}

// P6: resampling (fitting.py:2037-2057).  One lane per (object, draw).
constexpr int POST_NOUT = 17;   // scale av rv cov[9] lnprob dist red dred logwt
__global__ void __launch_bounds__(64)
k_post_draw(PostParams pp, int nstar, int64_t cap, const int32_t *__restrict__ sel_idx,
            const double *__restrict__ sel_vals, const int64_t *__restrict__ sel_off,
            const int64_t *__restrict__ off2, const int64_t *__restrict__ nselv,
            const uint64_t *__restrict__ nbase, const int32_t *__restrict__ flags,
            const StarGeom *__restrict__ geom, const double *__restrict__ feh,
            const double *__restrict__ loga, RecPost rp, const double *__restrict__ cdf,
            const double *__restrict__ star_out, int32_t *__restrict__ out_idx,
            double *__restrict__ out_vals) {
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    const int s = blockIdx.y;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= pp.ndraws || flags[s]) return;
    const int64_t a = off2[s], nsel = nselv[s];
    if (nsel <= 0) return;
    const StarGeom g = geom[s];
    const uint64_t ub = star_ubase(pp, s);
    const uint64_t seed = star_seed(pp, s);
    // choice(Nsel, p=wt): searchsorted(cdf / cdf[-1], u, side='right')
    const double total = star_out[4 * s + 2];
    const double u = rng_uniform(seed, ub + (uint64_t)q);
    int64_t lo = 0, hi = nsel;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[a + mid] / total <= u) lo = mid + 1; else hi = mid;
    }
    if (lo >= nsel) lo = nsel - 1;
    const int64_t o = a + lo;
    const int64_t r = sel_off[s] + rp.src[o];
    const int64_t i = sel_idx[r];
    out_idx[(int64_t)s * pp.ndraws + q] = (int32_t)i;
    double *ov = out_vals + ((int64_t)s * pp.ndraws + q) * POST_NOUT;
    const double s0 = sel_vals[2 * cap + r], a0 = sel_vals[3 * cap + r], r0 = sel_vals[4 * cap + r];
    ov[0] = s0;
    ov[1] = a0;
    ov[2] = r0;
    double C[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) C[k] = rp.cov[(int64_t)k * cap + o];
    ov[3] = C[0]; ov[4] = C[1]; ov[5] = C[2];
    ov[6] = C[1]; ov[7] = C[3]; ov[8] = C[4];
    ov[9] = C[2]; ov[10] = C[4]; ov[11] = C[5];
    ov[12] = rp.lnp[o];
    if (!pp.return_distreds) return;
    // second stage (fitting.py:2049-2057): pick one of the record's nmc samples
    double Fc[3], Ac[3], L[6];
    label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac);
#pragma unroll
    for (int k = 0; k < 6; ++k) L[k] = rp.chol[(int64_t)k * cap + o];
    const uint64_t nb = nbase[s];
    double m = -INFINITY;
    bool inb_;
    NormalReader rd[3];
    rd[0].init(seed); rd[1].init(seed); rd[2].init(seed);
    for (int t = 0; t < pp.nmc; ++t) {
        double d_, a_, r_;
        const double v = mc_sample(pp, rd, g, nb, lo, t, s0, a0, r0, L, Fc, Ac, s_tbl, d_, a_, r_, inb_);
        if (v > m) m = v;
    }
    double z = 0.;
    for (int t = 0; t < pp.nmc; ++t) {
        double d_, a_, r_;
        z += exp(mc_sample(pp, rd, g, nb, lo, t, s0, a0, r0, L, Fc, Ac, s_tbl, d_, a_, r_, inb_) - m);
    }
    // wt = softmax(logwts); imc = searchsorted(cumsum(wt) / sum, u2, side='right')
    const double u2 = rng_uniform(seed, ub + (uint64_t)pp.ndraws + (uint64_t)q);
    double run = 0., dist = 0., red = 0., dred = 0., lw = 0.;
    for (int t = 0; t < pp.nmc; ++t) {
        double d_, a_, r_;
        const double v = mc_sample(pp, rd, g, nb, lo, t, s0, a0, r0, L, Fc, Ac, s_tbl, d_, a_, r_, inb_);
        run += exp(v - m);
        dist = d_; red = a_; dred = r_; lw = v;
        if (run / z > u2) break;          // first cumulative weight above u2
    }
    ov[13] = dist;
    ov[14] = red;
    ov[15] = dred;
    ov[16] = lw;
}

// Nsel_max clipping (fitting.py:1029-1036): keep the nsel_max largest lnp in
// DESCENDING order.  Rare; a device radix sort per affected object.
__global__ void k_iota32(int32_t *p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (int32_t)i;
}
template <typename T>
__global__ void k_gather(T *__restrict__ dst, const T *__restrict__ src,
                         const int32_t *__restrict__ perm, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}

// per-object geometry / parallax constants
__global__ void k_post_geom(int nstar, const double *__restrict__ coords,
                            const double *__restrict__ par, const double *__restrict__ perr,
                            StarGeom *__restrict__ geom) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstar) return;
    const double l = coords[2 * s] * (M_PI / 180.), b = coords[2 * s + 1] * (M_PI / 180.);
    StarGeom g;
    g.cb_cl = cos(b) * cos(l);
    g.cb_sl = cos(b) * sin(l);
    g.sb = sin(b);
    const double p = par ? par[s] : nan(""), pe = perr ? perr[s] : nan("");
    g.has_par = (isfinite(p) && isfinite(pe)) ? 1 : 0;
    g.par = g.has_par ? p : 0.;
    g.par_ivar = g.has_par ? 1. / (pe * pe) : 0.;
    g.par_lnorm = g.has_par ? log(2. * M_PI * pe * pe) : 0.;
    geom[s] = g;
}

__global__ void k_debug_normals(uint64_t seed, uint64_t start, int64_t n, double *__restrict__ z,
                                double *__restrict__ u) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[i] = rng_normal(seed, start + (uint64_t)i);
    u[i] = rng_uniform(seed, start + (uint64_t)i);
}

__global__ void k_debug_galprior(PostParams pp, int n, const double *__restrict__ dist,
                                 const double *__restrict__ coords, const double *__restrict__ feh,
                                 const double *__restrict__ loga, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    StarGeom g;
    const double l = coords[0] * (M_PI / 180.), b = coords[1] * (M_PI / 180.);
    g.cb_cl = cos(b) * cos(l);
    g.cb_sl = cos(b) * sin(l);
    g.sb = sin(b);
    g.has_par = 0;
    double Fc[3], Ac[3];
    label_terms(pp, feh[i], loga[i], Fc, Ac);
    out[i] = gal_lnprior_dev(pp, g, dist[i], Fc, Ac, kExp2Tbl);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
const int kCompiledNB[] = {8, 12, 16, 24, 32};

int padded_nb(int nfilt) {
    for (int nb : kCompiledNB)
        if (nfilt <= nb) return nb;
    return -1;
}

int64_t pad_models(int64_t n) { return (n + TILE - 1) / TILE * TILE; }

struct Workspace {
    Planes pl;
    StarPrep *stars;
    double *part;       // per-(tile, star) partial maxima
    double *vmax_lnlp;  // (S,)
    int32_t *k1;        // (S,)
    int32_t *k2;        // (S,)  >=0 active iteration count, <0 done: -(K2)-1
    int32_t *n_unconv;  // (1,)
    int64_t *counts;    // (S, NCHUNK)
    int64_t *offsets;   // (S, NCHUNK)
    // fast path
    int32_t *ids;       // (S,) star list of a fused-scan launch
    int32_t *kfix;      // (S,)
    double *thr_cull, *maxns, *maxns_part, *maxsurv, *thr_sel;
    int32_t *surv_idx;  // (S * nmodel,) worst case
    int64_t *surv_off;  // (S + 1,)
    int32_t *wbase_surv, *wbase_sel;   // (S + 1,)
    unsigned long long *mask;          // (S, nmodel_pad / 64) membership words
    size_t bytes;
};

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Lay the workspace out over `base` (may be null: sizing only).  When the
// caller supplies the output planes (loglike_batch) they are used instead of
// workspace planes.
Workspace carve(char *base, int64_t nmodel, int nstar, bool own_outputs) {
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t n) {
        char *p = base ? base + off : nullptr;
        off += align_up(n);
        return p;
    };
    const size_t plane = (size_t)nstar * (size_t)nmodel * sizeof(double);
    const int64_t ntile = pad_models(nmodel) / TILE;
    w.pl.nmodel = nmodel;
    w.pl.lnlp = (double *)take(plane);
    w.pl.step = (double *)take(plane);
    if (own_outputs) {
        w.pl.lnl = (double *)take(plane);
        w.pl.chi2 = (double *)take(plane);
        w.pl.scale = (double *)take(plane);
        w.pl.av = (double *)take(plane);
        w.pl.rv = (double *)take(plane);
        for (int q = 0; q < 6; ++q) w.pl.icov[q] = (double *)take(plane);
    }
    w.stars = (StarPrep *)take(sizeof(StarPrep) * nstar);
    w.part = (double *)take(sizeof(double) * (size_t)ntile * nstar * 2 * KCAP);
    w.vmax_lnlp = (double *)take(sizeof(double) * nstar);
    w.k1 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.k2 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.n_unconv = (int32_t *)take(sizeof(int32_t) * 4);
    w.counts = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
    w.offsets = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
    if (own_outputs) {
        w.pl.lnprob = (double *)take(plane);
        w.ids = (int32_t *)take(sizeof(int32_t) * nstar);
        w.kfix = (int32_t *)take(sizeof(int32_t) * nstar);
        w.thr_cull = (double *)take(sizeof(double) * nstar);
        w.maxns = (double *)take(sizeof(double) * nstar);
        w.maxns_part = (double *)take(sizeof(double) * nstar * NCHUNK);
        w.maxsurv = (double *)take(sizeof(double) * nstar);
        w.thr_sel = (double *)take(sizeof(double) * nstar);
        w.surv_idx = (int32_t *)take(sizeof(int32_t) * (size_t)nstar * (size_t)nmodel);
        w.surv_off = (int64_t *)take(sizeof(int64_t) * (nstar + 1));
        w.wbase_surv = (int32_t *)take(sizeof(int32_t) * (nstar + 1));
        w.wbase_sel = (int32_t *)take(sizeof(int32_t) * (nstar + 1));
        w.mask = (unsigned long long *)take(sizeof(unsigned long long) * (size_t)nstar *
                                            (size_t)(pad_models(nmodel) / 64));
    }
    w.bytes = off;
    return w;
}

int make_params(const brutus_params *in, DevParams &p) {
    if (!in) return fail(BRUTUS_EINVAL, "params is NULL");
    if (!(in->init_thresh > 0.) || !(in->ltol_subthresh > 0.))
        return fail(BRUTUS_EINVAL, "thresholds must be positive");
    if (in->init_thresh > in->ltol_subthresh)   // fitting.py:691-693
        return fail(BRUTUS_EINVAL,
                    "The initial threshold must be smaller than or equal to the "
                    "final threshold applied to be useful!");
    p.avmin = in->avlim[0];
    p.avmax = in->avlim[1];
    p.rvmin = in->rvlim[0];
    p.rvmax = in->rvlim[1];
    p.av_mean = in->av_gauss[0];
    p.av_ivar = 1. / (in->av_gauss[1] * in->av_gauss[1]);
    p.rv_mean = in->rv_gauss[0];
    p.rv_ivar = 1. / (in->rv_gauss[1] * in->rv_gauss[1]);
    p.mtol = 2.5 * in->ltol;
    p.ltol = in->ltol;
    p.ln_init = log(in->init_thresh);
    p.ln_sub = log(in->ltol_subthresh);
    p.ln_wt = in->wt_thresh > 0. ? log(in->wt_thresh) : -INFINITY;
    p.a_reg = 1. / (0.05 * 0.05);
    p.r_reg = 1. / (0.1 * 0.1);
    p.dim_prior = in->dim_prior ? 1 : 0;
    return 0;
}

struct Timer {
    hipStream_t st;
    std::vector<std::pair<std::string, std::pair<hipEvent_t, hipEvent_t>>> ev;
    explicit Timer(hipStream_t s) : st(s) {}
    void begin(const char *name) {
        if (!g_timing) return;
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a, st);
        ev.push_back({name, {a, b}});
    }
    void end() {
        if (!g_timing) return;
        hipEventRecord(ev.back().second.second, st);
    }
    void collect() {
        if (!g_timing) return;
        g_last_timing.clear();
        for (auto &e : ev) {
            hipEventSynchronize(e.second.second);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e.second.first, e.second.second);
            bool found = false;
            for (auto &t : g_last_timing)
                if (t.name == e.first) {
                    t.ms += ms;
                    t.count += 1;
                    found = true;
                }
            if (!found) g_last_timing.push_back({e.first, ms, 1});
            hipEventDestroy(e.second.first);
            hipEventDestroy(e.second.second);
        }
        ev.clear();
    }
};

template <int NB>
int run_pipeline(const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                 int max_iter, Workspace &w, int32_t *h_k1, int32_t *h_k2,
                 hipStream_t st, Timer &tm) {
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    const dim3 gridA(ntile, (nstar + STAR_GROUP - 1) / STAR_GROUP);
    const dim3 blk(TILE);
    int32_t h_unconv = 0;

    // ---- phase 1: number of magnitude sweeps K1 per star --------------------
    int kmax = 2;
    for (;;) {
        HIP_TRY(hipMemsetAsync(w.n_unconv, 0, sizeof(int32_t), st));
        tm.begin("k_mag_stats");
        hipLaunchKernelGGL(k_mag_stats<NB>, gridA, blk, 0, st, grid, nmodel, nmodel_pad, nstar,
                           w.stars, p, kmax, w.part);
        tm.end();
        hipLaunchKernelGGL(k_reduce_decide, dim3(nstar), dim3(256), 0, st, 0, ntile, nstar,
                           2 * kmax, w.part, p.ln_init, (double *)nullptr, w.k1, w.n_unconv);
        HIP_TRY(hipMemcpyAsync(&h_unconv, w.n_unconv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_unconv == 0) break;
        if (kmax >= KCAP || kmax >= max_iter)
            return fail(BRUTUS_ENOCONV, "magnitude phase not converged after %d sweeps for %d star(s)",
                        kmax, h_unconv);
        kmax = kmax * 2 > KCAP ? KCAP : kmax * 2;
    }

    // ---- phase 2: MLE at the converged (Av, Rv); cull statistic -------------
    tm.begin("k_mag_mle");
    hipLaunchKernelGGL(k_mag_mle<NB>, gridA, blk, 0, st, grid, nmodel, nmodel_pad, nstar, w.stars,
                       p, w.k1, w.pl, w.part);
    tm.end();
    hipLaunchKernelGGL(k_reduce_decide, dim3(nstar), dim3(256), 0, st, 1, ntile, nstar, 1, w.part,
                       0.0, w.vmax_lnlp, (int32_t *)nullptr, (int32_t *)nullptr);

    // ---- phase 3: flux iterations on survivors ------------------------------
    hipLaunchKernelGGL(k_set_i32, dim3((nstar + 255) / 256), dim3(256), 0, st, w.k2, nstar, 2);
    int iter = 2;
    for (int first = 1;; first = 0) {
        HIP_TRY(hipMemsetAsync(w.n_unconv, 0, sizeof(int32_t), st));
        tm.begin(first ? "k_flux" : "k_flux_cont");
        hipLaunchKernelGGL(k_flux<NB>, gridA, blk, 0, st, grid, nmodel, nmodel_pad, nstar, w.stars,
                           p, w.vmax_lnlp, w.k2, first, w.pl, w.part);
        tm.end();
        hipLaunchKernelGGL(k_reduce_decide, dim3(nstar), dim3(256), 0, st, 2, ntile, nstar, 2,
                           w.part, p.ln_sub, (double *)nullptr, w.k2, w.n_unconv);
        HIP_TRY(hipMemcpyAsync(&h_unconv, w.n_unconv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_unconv == 0) break;
        if (iter >= max_iter)
            return fail(BRUTUS_ENOCONV, "flux phase not converged after %d iterations for %d star(s)",
                        iter, h_unconv);
        ++iter;
    }

    // ---- phase 4: constants, dimensionality prior, parallax clip ------------
    tm.begin("k_finalize");
    hipLaunchKernelGGL(k_finalize, dim3(ntile, nstar), blk, 0, st, nmodel, nstar, w.stars, p,
                       w.vmax_lnlp, w.pl);
    tm.end();
    if (h_k1) HIP_TRY(hipMemcpyAsync(h_k1, w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    if (h_k2) HIP_TRY(hipMemcpyAsync(h_k2, w.k2, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

int dispatch_pipeline(int nb, const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                      int max_iter, Workspace &w, int32_t *h_k1, int32_t *h_k2,
                      hipStream_t st, Timer &tm) {
    switch (nb) {
        case 8: return run_pipeline<8>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm);
        case 12: return run_pipeline<12>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm);
        case 16: return run_pipeline<16>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm);
        case 24: return run_pipeline<24>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm);
        case 32: return run_pipeline<32>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm);
    }
    return fail(BRUTUS_EINVAL, "unsupported band count %d", nb);
}

// ---- fast path host orchestration ------------------------------------------
constexpr int FS_TILES_PER_BLOCK = 8;
constexpr int PERSIST_BLOCKS = 4096;

template <int NB, int KS, int G, bool RVF>
int launch_fscan(const float *grid, int64_t nmodel, int nstar, const std::vector<int32_t> &ids,
                 const std::vector<int32_t> &kfix, const DevParams &p, Workspace &w, int accept,
                 hipStream_t st, Timer &tm) {
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    const int nblkx = (ntile + FS_TILES_PER_BLOCK - 1) / FS_TILES_PER_BLOCK;
    const int nrun = (int)ids.size();
    HIP_TRY(hipMemcpyAsync(w.ids, ids.data(), sizeof(int32_t) * nrun, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(w.kfix, kfix.data(), sizeof(int32_t) * nstar, hipMemcpyHostToDevice, st));
    constexpr int NV = 2 * KS + 2;
    const size_t shmem = (size_t)G * NV * TILE * sizeof(double);
    tm.begin("k_fscan");
    hipLaunchKernelGGL((k_fscan<NB, KS, G, RVF>), dim3(nblkx, (nrun + G - 1) / G), dim3(TILE), shmem, st,
                       grid, nmodel, nmodel_pad, nstar, nrun, w.ids, w.stars, p, w.kfix,
                       FS_TILES_PER_BLOCK, ntile, w.pl, w.part);
    tm.end();
    hipLaunchKernelGGL(k_fdecide, dim3(nrun), dim3(256), 0, st, nblkx, nstar, nrun, w.ids, KS,
                       w.part, p, accept, w.k1, w.thr_cull, w.maxns);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NB, bool RVF>
int run_select_emit(const float *grid, int64_t nmodel, int nstar, const DevParams &p, Workspace &w,
                    int64_t capacity, int32_t *d_sel_idx, double *d_sel_vals, int64_t *d_sel_off,
                    hipStream_t st, Timer &tm) {
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    tm.begin("k_select");
    hipLaunchKernelGGL(k_cmp_count, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile,
                       w.pl.lnprob, w.thr_sel, (const double *)nullptr, w.counts,
                       (double *)nullptr, w.mask);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, nstar, w.counts,
                       w.offsets, d_sel_off, w.wbase_sel);
    hipLaunchKernelGGL(k_cmp_scatter, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile,
                       w.mask, w.offsets, capacity, d_sel_idx);
    tm.end();
    tm.begin("k_emit");
    hipLaunchKernelGGL((k_emit<NB, RVF>), dim3(PERSIST_BLOCKS), dim3(TILE), 0, st, grid, nmodel,
                       nmodel_pad, nstar, w.stars, p, w.k1, w.thr_cull, d_sel_idx, d_sel_off,
                       w.wbase_sel, w.pl, capacity, d_sel_vals);
    tm.end();
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NB, bool RVF>
int run_fast(const float *grid, int64_t nmodel, int nstar, const DevParams &p, int max_iter,
             Workspace &w, int64_t capacity, int32_t *d_sel_idx, double *d_sel_vals,
             int64_t *d_sel_off, int32_t *h_k1, int32_t *h_k2, hipStream_t st, Timer &tm) {
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    std::vector<int32_t> ids(nstar), kfix(nstar, 2), k1(nstar, 0);
    for (int s = 0; s < nstar; ++s) ids[s] = s;

    // ---- fused scan, speculating K1 = 2 --------------------------------------
    if (int rc = launch_fscan<NB, 2, 4, RVF>(grid, nmodel, nstar, ids, kfix, p, w, 0, st, tm)) return rc;
    HIP_TRY(hipMemcpyAsync(k1.data(), w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int32_t> listA, listB;
    for (int s = 0; s < nstar; ++s) {
        if (k1[s] == 1) listA.push_back(s);
        if (k1[s] == 0) listB.push_back(s);
    }
    if (!listA.empty()) {   // converged after ONE sweep: redo those stars with one sweep
        for (int s : listA) kfix[s] = 1;
        if (int rc = launch_fscan<NB, 2, 4, RVF>(grid, nmodel, nstar, listA, kfix, p, w, 1, st, tm)) return rc;
    }
    if (!listB.empty()) {   // needs more than two sweeps: probe up to eight
        for (int s : listB) kfix[s] = 8;
        if (int rc = launch_fscan<NB, 8, 1, RVF>(grid, nmodel, nstar, listB, kfix, p, w, 0, st, tm)) return rc;
        HIP_TRY(hipMemcpyAsync(k1.data(), w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        std::vector<int32_t> listC;
        for (int s : listB) {
            if (k1[s] == 0 || k1[s] > max_iter)
                return fail(BRUTUS_ENOCONV, "magnitude phase of star %d not converged after 8 sweeps", s);
            if (k1[s] != 8) {
                kfix[s] = k1[s];
                listC.push_back(s);
            }
        }
        if (!listC.empty())
            if (int rc = launch_fscan<NB, 2, 4, RVF>(grid, nmodel, nstar, listC, kfix, p, w, 1, st, tm)) return rc;
    }

    // ---- cull: ordered survivor lists -----------------------------------------
    tm.begin("k_surv_compact");
    hipLaunchKernelGGL(k_cmp_count, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile, w.pl.lnlp,
                       w.thr_cull, w.pl.lnprob, w.counts, w.maxns_part, w.mask);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, nstar, w.counts, w.offsets,
                       w.surv_off, w.wbase_surv);
    hipLaunchKernelGGL(k_cmp_scatter, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile,
                       w.mask, w.offsets, (int64_t)nstar * nmodel, w.surv_idx);
    tm.end();

    // ---- flux phase on survivors ------------------------------------------------
    hipLaunchKernelGGL(k_set_i32, dim3((nstar + 255) / 256), dim3(256), 0, st, w.k2, nstar, 2);
    int32_t h_unconv = 0;
    int iter = 2;
    for (int first = 1;; first = 0) {
        HIP_TRY(hipMemsetAsync(w.n_unconv, 0, sizeof(int32_t), st));
        tm.begin(first ? "k_fflux" : "k_fflux_cont");
        hipLaunchKernelGGL((k_fflux<NB, RVF>), dim3(PERSIST_BLOCKS), dim3(TILE), 0, st, grid, nmodel,
                           nmodel_pad, nstar, w.stars, p, w.k1, w.k2, first, w.surv_idx, w.surv_off,
                           w.wbase_surv, w.pl, w.part);
        tm.end();
        hipLaunchKernelGGL(k_fflux_decide, dim3(nstar), dim3(256), 0, st, nstar, w.wbase_surv, w.part,
                           p.ln_sub, w.k2, w.maxsurv, w.n_unconv);
        HIP_TRY(hipMemcpyAsync(&h_unconv, w.n_unconv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_unconv == 0) break;
        if (iter >= max_iter)
            return fail(BRUTUS_ENOCONV, "flux phase not converged after %d iterations for %d star(s)",
                        iter, h_unconv);
        ++iter;
    }

    // ---- first cut of lnpost + records ------------------------------------------
    hipLaunchKernelGGL(k_sel_thresh, dim3((nstar + 63) / 64), dim3(64), 0, st, nstar, w.maxns_part,
                       w.maxsurv, p.ln_wt, w.thr_sel);
    if (int rc = run_select_emit<NB, RVF>(grid, nmodel, nstar, p, w, capacity, d_sel_idx, d_sel_vals,
                                     d_sel_off, st, tm))
        return rc;
    if (h_k1) HIP_TRY(hipMemcpyAsync(h_k1, w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    if (h_k2) HIP_TRY(hipMemcpyAsync(h_k2, w.k2, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

// Rv pinned by its limits at the value every fit starts from: the (offset, Av)
// specialisation computes the same thing
inline bool rv_pinned(const DevParams &p) { return p.rvmin == p.rvmax && p.rv_mean == p.rvmin; }

int dispatch_fast(int nb, const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                  int max_iter, Workspace &w, int64_t capacity, int32_t *d_sel_idx,
                  double *d_sel_vals, int64_t *d_sel_off, int32_t *h_k1, int32_t *h_k2,
                  hipStream_t st, Timer &tm) {
    const bool rvf = rv_pinned(p);
#define BRUTUS_CASE(N)                                                                             \
    case N:                                                                                        \
        return rvf ? run_fast<N, true>(grid, nmodel, nstar, p, max_iter, w, capacity, d_sel_idx,   \
                                       d_sel_vals, d_sel_off, h_k1, h_k2, st, tm)                  \
                   : run_fast<N, false>(grid, nmodel, nstar, p, max_iter, w, capacity, d_sel_idx,  \
                                        d_sel_vals, d_sel_off, h_k1, h_k2, st, tm);
    switch (nb) {
        BRUTUS_CASE(8)
        BRUTUS_CASE(12)
        BRUTUS_CASE(16)
        BRUTUS_CASE(24)
        BRUTUS_CASE(32)
    }
#undef BRUTUS_CASE
    return fail(BRUTUS_EINVAL, "unsupported band count %d", nb);
}

int dispatch_select_emit(int nb, const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                         Workspace &w, int64_t capacity, int32_t *d_sel_idx, double *d_sel_vals,
                         int64_t *d_sel_off, hipStream_t st, Timer &tm) {
    const bool rvf = rv_pinned(p);
#define BRUTUS_CASE(N)                                                                             \
    case N:                                                                                        \
        return rvf ? run_select_emit<N, true>(grid, nmodel, nstar, p, w, capacity, d_sel_idx,      \
                                              d_sel_vals, d_sel_off, st, tm)                       \
                   : run_select_emit<N, false>(grid, nmodel, nstar, p, w, capacity, d_sel_idx,     \
                                               d_sel_vals, d_sel_off, st, tm);
    switch (nb) {
        BRUTUS_CASE(8)
        BRUTUS_CASE(12)
        BRUTUS_CASE(16)
        BRUTUS_CASE(24)
        BRUTUS_CASE(32)
    }
#undef BRUTUS_CASE
    return fail(BRUTUS_EINVAL, "unsupported band count %d", nb);
}

int check_common(int64_t nmodel, int nfilt, int nstar) {
    if (nmodel <= 0 || nmodel > (int64_t)1 << 31) return fail(BRUTUS_EINVAL, "bad nmodel");
    if (padded_nb(nfilt) < 0 || nfilt < 1)
        return fail(BRUTUS_EINVAL, "nfilt=%d unsupported (max %d)", nfilt, BRUTUS_MAX_FILT);
    if (nstar < 1 || nstar > BRUTUS_MAX_BATCH)
        return fail(BRUTUS_EINVAL, "nstar=%d outside [1, %d]", nstar, BRUTUS_MAX_BATCH);
    return 0;
}

int launch_prep(int nstar, int nfilt, const double *d_flux, const double *d_err,
                const uint8_t *d_mask, const double *d_par, const double *d_perr, int has_par,
                Workspace &w, int32_t *d_ndim, hipStream_t st) {
    hipLaunchKernelGGL(k_prep, dim3((nstar + 63) / 64), dim3(64), 0, st, nstar, nfilt, d_flux,
                       d_err, d_mask, d_par, d_perr, (d_par && d_perr) ? has_par : 0, w.stars,
                       d_ndim);
    HIP_TRY(hipGetLastError());
    return 0;
}

void fix_k2(int32_t *h_k2, int nstar) {
    if (!h_k2) return;
    for (int s = 0; s < nstar; ++s)
        if (h_k2[s] < 0) h_k2[s] = -h_k2[s] - 1;
}

}  // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int brutus_abi_version(void) { return BRUTUS_ABI_VERSION; }
const char *brutus_last_error(void) { return g_err.c_str(); }
int brutus_padded_filters(int nfilt) { return padded_nb(nfilt); }

size_t brutus_grid_soa_bytes(int64_t nmodel, int nfilt) {
    const int nb = padded_nb(nfilt);
    if (nb < 0 || nmodel <= 0) return 0;
    // f32 coefficients (SoA + model-major) and the f64 F0 table (SoA)
    return 8 * (size_t)nb * (size_t)pad_models(nmodel) * sizeof(float);
}

int brutus_grid_relayout(const float *d_models_aos, int64_t nmodel, int nfilt, float *d_grid_soa,
                         void *stream) {
    const int nb = padded_nb(nfilt);
    if (nb < 0 || nmodel <= 0 || !d_models_aos || !d_grid_soa)
        return fail(BRUTUS_EINVAL, "bad grid arguments");
    const int64_t np = pad_models(nmodel);
    hipLaunchKernelGGL(k_relayout, dim3((unsigned)(np / TILE)), dim3(TILE), 0, (hipStream_t)stream,
                       d_models_aos, nmodel, nfilt, nb, np, d_grid_soa);
    HIP_TRY(hipGetLastError());
    return 0;
}

size_t brutus_workspace_bytes(int64_t nmodel, int nfilt, int nstar) {
    if (check_common(nmodel, nfilt, nstar)) return 0;
    return carve(nullptr, nmodel, nstar, true).bytes;
}

int brutus_loglike_batch(const float *d_grid_soa, int64_t nmodel, int nfilt, int nstar,
                         const double *d_flux, const double *d_err, const uint8_t *d_mask,
                         const double *d_parallax, const double *d_parallax_err, int has_parallax,
                         const brutus_params *params, void *d_workspace, size_t workspace_bytes,
                         double *d_lnl, double *d_chi2, double *d_scale, double *d_av, double *d_rv,
                         double *d_icov, int32_t *d_ndim, int32_t *h_k1, int32_t *h_k2,
                         void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    DevParams p;
    if (int rc = make_params(params, p)) return rc;
    if (!d_grid_soa || !d_flux || !d_err || !d_mask || !d_workspace || !d_lnl || !d_chi2 ||
        !d_scale || !d_av || !d_rv || !d_icov || !d_ndim)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    Workspace w = carve((char *)d_workspace, nmodel, nstar, false);
    if (w.bytes > workspace_bytes)
        return fail(BRUTUS_ENOMEM, "workspace too small: need %zu bytes, got %zu", w.bytes,
                    workspace_bytes);
    w.pl.lnl = d_lnl;
    w.pl.chi2 = d_chi2;
    w.pl.scale = d_scale;
    w.pl.av = d_av;
    w.pl.rv = d_rv;
    for (int q = 0; q < 6; ++q) w.pl.icov[q] = d_icov + (size_t)q * nstar * nmodel;
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    if (int rc = launch_prep(nstar, nfilt, d_flux, d_err, d_mask, d_parallax, d_parallax_err,
                             has_parallax, w, d_ndim, st))
        return rc;
    const int max_iter = params->max_iter > 0 ? params->max_iter : 256;
    int rc = dispatch_pipeline(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, max_iter, w, h_k1,
                               h_k2, st, tm);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    fix_k2(h_k2, nstar);
    tm.collect();
    return 0;
}

int brutus_fit_gather(const float *d_grid_soa, int64_t nmodel, int nfilt, int nstar,
                      const brutus_params *params, void *d_workspace, size_t workspace_bytes,
                      int64_t capacity, int32_t *d_sel_idx, double *d_sel_vals,
                      int64_t *d_sel_off, void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    DevParams p;
    if (int rc = make_params(params, p)) return rc;
    if (!d_grid_soa || !d_workspace || !d_sel_idx || !d_sel_vals || !d_sel_off || capacity < 0)
        return fail(BRUTUS_EINVAL, "bad gather arguments");
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes) return fail(BRUTUS_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    int rc = dispatch_select_emit(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, w, capacity,
                                  d_sel_idx, d_sel_vals, d_sel_off, st, tm);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return 0;
}

int brutus_fit_batch(const float *d_grid_soa, int64_t nmodel, int nfilt, int nstar,
                     const double *d_flux, const double *d_err, const uint8_t *d_mask,
                     const double *d_parallax, const double *d_parallax_err, int has_parallax,
                     const brutus_params *params, void *d_workspace, size_t workspace_bytes,
                     int64_t capacity, int32_t *d_sel_idx, double *d_sel_vals, int64_t *d_sel_off,
                     int32_t *d_ndim, int32_t *h_k1, int32_t *h_k2, void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    DevParams p;
    if (int rc = make_params(params, p)) return rc;
    if (!d_grid_soa || !d_flux || !d_err || !d_mask || !d_workspace || !d_sel_idx || !d_sel_vals ||
        !d_sel_off || !d_ndim || capacity < 0)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes)
        return fail(BRUTUS_ENOMEM, "workspace too small: need %zu bytes, got %zu", w.bytes,
                    workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    if (int rc = launch_prep(nstar, nfilt, d_flux, d_err, d_mask, d_parallax, d_parallax_err,
                             has_parallax, w, d_ndim, st))
        return rc;
    const int max_iter = params->max_iter > 0 ? params->max_iter : 256;
    int rc = dispatch_fast(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, max_iter, w, capacity,
                           d_sel_idx, d_sel_vals, d_sel_off, h_k1, h_k2, st, tm);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    fix_k2(h_k2, nstar);
    tm.collect();
    return 0;
}

constexpr int CLUSTER_CHUNKS = 64;

size_t brutus_cluster_workspace_bytes(int nobj) {
    if (nobj <= 0) return 0;
    return 2 * align_up(sizeof(double) * (size_t)nobj * CLUSTER_CHUNKS);
}

int brutus_cluster_lnl(int nobj, int nfilt, int npts, const double *d_pts_flux,
                       const double *d_pts_lnw, const double *d_phot, const double *d_ivar,
                       const double *d_chi2_p, const double *d_lnorm, const int32_t *d_ndim,
                       int dim_prior, void *d_workspace, size_t workspace_bytes, double *d_lnl,
                       void *stream) {
    const int nb = padded_nb(nfilt);
    if (nobj <= 0 || npts <= 0 || nb < 0)
        return fail(BRUTUS_EINVAL, "bad cluster dimensions (nobj=%d, npts=%d, nfilt=%d)", nobj,
                    npts, nfilt);
    if (!d_pts_flux || !d_pts_lnw || !d_phot || !d_ivar || !d_chi2_p || !d_lnorm || !d_ndim ||
        !d_workspace || !d_lnl)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    if (workspace_bytes < brutus_cluster_workspace_bytes(nobj))
        return fail(BRUTUS_ENOMEM, "cluster workspace too small");
    double *pm = (double *)d_workspace;
    double *ps = (double *)((char *)d_workspace + align_up(sizeof(double) * (size_t)nobj * CLUSTER_CHUNKS));
    hipStream_t st = (hipStream_t)stream;
    const int ppb = (npts + CLUSTER_CHUNKS - 1) / CLUSTER_CHUNKS;
    const int nchunk = (npts + ppb - 1) / ppb;
    const dim3 g((nobj + 63) / 64, nchunk);
#define BRUTUS_CL(N)                                                                              \
    case N:                                                                                       \
        hipLaunchKernelGGL(k_cluster<N>, g, dim3(64), 0, st, nobj, nfilt, npts, d_pts_flux,       \
                           d_pts_lnw, d_phot, d_ivar, d_chi2_p, d_lnorm, d_ndim, dim_prior, ppb,  \
                           pm, ps);                                                               \
        break;
    switch (nb) {
        BRUTUS_CL(8)
        BRUTUS_CL(12)
        BRUTUS_CL(16)
        BRUTUS_CL(24)
        BRUTUS_CL(32)
    }
#undef BRUTUS_CL
    hipLaunchKernelGGL(k_cluster_merge, dim3((nobj + 255) / 256), dim3(256), 0, st, nobj, nchunk, pm,
                       ps, d_lnl);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- lnpost on the device ---------------------------------------------------
struct PostWs {
    double *lnp1, *part, *part_w, *part_max, *part_chi2, *cdf, *star_out;
    unsigned long long *mask;
    int64_t *counts, *offsets, *off2;
    uint64_t *nbase;
    int32_t *flags;
    int64_t *nsel;
    StarGeom *geom;
    RecPost rp;
    // Nsel_max path: radix-sort scratch
    double *sort_keys;
    int32_t *sort_in, *sort_perm;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    // k_post_mc: work counter and staged normals
    unsigned int *mc_counter;
    double2 *mc_stage;
    size_t bytes;
};

constexpr int MC_SLOTS = 1024;     // persistent workgroups (= staging slots) of k_post_mc

PostWs carve_post(char *base, int nstar, int64_t cap, int nmc) {
    PostWs w{};
    size_t off = 0;
    auto take = [&](size_t n) {
        char *p = base ? base + off : nullptr;
        off += align_up(n);
        return p;
    };
    const size_t c = (size_t)cap;
    w.lnp1 = (double *)take(8 * c);
    w.mask = (unsigned long long *)take(8 * (c / 64 + 8 * (size_t)nstar + 16));
    w.counts = (int64_t *)take(8 * (size_t)nstar * PCH);
    w.offsets = (int64_t *)take(8 * (size_t)nstar * PCH);
    w.part = (double *)take(8 * (size_t)nstar * PCH);
    w.part_w = (double *)take(8 * (size_t)nstar * PCH);
    w.part_max = (double *)take(8 * (size_t)nstar * PCH);
    w.part_chi2 = (double *)take(8 * (size_t)nstar * PCH);
    w.off2 = (int64_t *)take(8 * ((size_t)nstar + 1));
    w.nbase = (uint64_t *)take(8 * ((size_t)nstar + 1));
    w.flags = (int32_t *)take(4 * (size_t)nstar);
    w.nsel = (int64_t *)take(8 * (size_t)nstar);
    w.geom = (StarGeom *)take(sizeof(StarGeom) * (size_t)nstar);
    w.star_out = (double *)take(8 * 4 * (size_t)nstar);
    w.rp.src = (int32_t *)take(4 * c);
    w.rp.lnp = (double *)take(8 * c);
    w.rp.cov = (double *)take(8 * 6 * c);
    w.rp.chol = (double *)take(8 * 6 * c);
    w.cdf = (double *)take(8 * c);
    w.sort_keys = (double *)take(8 * c);
    w.sort_in = (int32_t *)take(4 * c);
    w.sort_perm = (int32_t *)take(4 * c);
    w.sort_tmp_bytes = 16 * c + (8u << 20);
    w.sort_tmp = take(w.sort_tmp_bytes);
    w.mc_counter = (unsigned int *)take(256);
    w.mc_stage = (double2 *)take(sizeof(double2) * (size_t)MC_SLOTS * mc_npair_max(nmc) * TILE);
    w.bytes = off;
    return w;
}

// Keep the nsel_max best records of object s, best first (fitting.py:1029-1036).
int clip_to_nsel_max(PostWs &w, int64_t cap, int64_t a, int64_t n, int64_t keep, hipStream_t st) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_iota32, dim3(nb), dim3(256), 0, st, w.sort_in, n);
    size_t need = 0;
    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, need, w.rp.lnp + a, w.sort_keys,
                                                         w.sort_in, w.sort_perm, (int)n, 0, 64, st));
    if (need > w.sort_tmp_bytes)
        return fail(BRUTUS_ENOMEM, "radix-sort scratch too small (%zu > %zu)", need, w.sort_tmp_bytes);
    HIP_TRY(hipcub::DeviceRadixSort::SortPairsDescending(w.sort_tmp, need, w.rp.lnp + a, w.sort_keys,
                                                         w.sort_in, w.sort_perm, (int)n, 0, 64, st));
    const unsigned kb = (unsigned)((keep + 255) / 256);
    // permute every per-record array through the (now free) lnp1-sized scratch
    double *tmp = w.lnp1;
    auto permute64 = [&](double *arr) -> int {
        hipLaunchKernelGGL(k_gather<double>, dim3(kb), dim3(256), 0, st, tmp, arr + a, w.sort_perm, keep);
        HIP_TRY(hipMemcpyAsync(arr + a, tmp, 8 * (size_t)keep, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    if (int rc = permute64(w.rp.lnp)) return rc;
    for (int q = 0; q < 6; ++q) {
        if (int rc = permute64(w.rp.cov + (size_t)q * cap)) return rc;
        if (int rc = permute64(w.rp.chol + (size_t)q * cap)) return rc;
    }
    hipLaunchKernelGGL(k_gather<int32_t>, dim3(kb), dim3(256), 0, st, (int32_t *)tmp, w.rp.src + a,
                       w.sort_perm, keep);
    HIP_TRY(hipMemcpyAsync(w.rp.src + a, tmp, 4 * (size_t)keep, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

void fill_post_params(PostParams &pp, const brutus_post_params *params) {
    memcpy(&pp, params, sizeof(brutus_post_params));
    pp.ln_f_thick = log(pp.f_thick);
    pp.ln_f_halo = log(pp.f_halo);
    const double rq2 = pp.r_q_halo * pp.r_q_halo, Rs2 = pp.R_solar * pp.R_solar, Zs = pp.Z_solar;
    const double qs = pp.q_halo_inf -
                      (pp.q_halo_inf - pp.q_halo_ctr) * exp(1. - sqrt(Rs2 + Zs * Zs + rq2) / pp.r_q_halo);
    pp.inv_reff_solar2 = 1. / (Rs2 + (Zs / qs) * (Zs / qs) + pp.Rs_halo * pp.Rs_halo);
    pp.inv_R_thin = 1. / pp.R_thin;
    pp.inv_Z_thin = 1. / pp.Z_thin;
    pp.inv_R_thick = 1. / pp.R_thick;
    pp.inv_Z_thick = 1. / pp.Z_thick;
    pp.inv_r_q = 1. / pp.r_q_halo;
    pp.Rs_thin2 = pp.Rs_thin * pp.Rs_thin;
    pp.Rs_thick2 = pp.Rs_thick * pp.Rs_thick;
    pp.Rs_halo2 = pp.Rs_halo * pp.Rs_halo;
    pp.rq2 = rq2;
    pp.abs_Z_solar = fabs(Zs);
    // comp_c <= k_c: R >= 0, |Z| >= 0, reff^2 >= Rs_halo^2
    const double k_thin = pp.R_solar * pp.inv_R_thin + pp.abs_Z_solar * pp.inv_Z_thin;
    const double k_thick = pp.R_solar * pp.inv_R_thick + pp.abs_Z_solar * pp.inv_Z_thick + pp.ln_f_thick;
    const double k_halo =
        pp.ln_f_halo - 0.5 * pp.eta_halo * log(fmax(pp.Rs_halo2, 1e-12) * pp.inv_reff_solar2);
    pp.lnK = fmax(fmax(k_thin, k_thick), k_halo);
    pp.c0_thin = pp.R_solar * pp.inv_R_thin - pp.lnK;
    pp.c0_thick = pp.R_solar * pp.inv_R_thick + pp.ln_f_thick - pp.lnK;
    pp.c0_halo = pp.ln_f_halo - pp.lnK;
}

size_t brutus_post_workspace_bytes(int nstar, int64_t capacity, int nmc) {
    if (nstar < 1 || nstar > BRUTUS_MAX_BATCH || capacity < 1 || nmc < 1) return 0;
    return carve_post(nullptr, nstar, capacity, nmc).bytes;
}

int brutus_post_batch(int nstar, int64_t capacity, const int32_t *d_sel_idx,
                      const double *d_sel_vals, const int64_t *d_sel_off, const double *d_lnprior,
                      const double *d_feh, const double *d_loga, const double *d_coords,
                      const double *d_parallax, const double *d_parallax_err,
                      const brutus_post_params *params, void *d_workspace, size_t workspace_bytes,
                      int32_t *d_out_idx, double *d_out_vals, double *h_star_out,
                      int32_t *h_flags, uint64_t *h_nbase, void *stream) {
    static_assert(sizeof(PostParams) == sizeof(brutus_post_params) + POST_DERIVED * sizeof(double),
                  "post params layout");
    if (nstar < 1 || nstar > BRUTUS_MAX_BATCH || capacity < 1)
        return fail(BRUTUS_EINVAL, "bad post dimensions");
    if (!d_sel_idx || !d_sel_vals || !d_sel_off || !d_lnprior || !d_coords || !params ||
        !d_workspace || !d_out_idx || !d_out_vals || !h_star_out || !h_flags)
        return fail(BRUTUS_EINVAL, "NULL pointer");
    if (params->nmc < 1 || params->ndraws < 1 || !(params->wt_thresh > 0.))
        return fail(BRUTUS_EINVAL, "nmc, ndraws and wt_thresh must be positive");
    if ((params->has_feh && !d_feh) || (params->has_loga && !d_loga))
        return fail(BRUTUS_EINVAL, "label arrays missing");
    PostWs w = carve_post((char *)d_workspace, nstar, capacity, params->nmc);
    if (w.bytes > workspace_bytes)
        return fail(BRUTUS_ENOMEM, "post workspace too small: need %zu bytes, got %zu", w.bytes,
                    workspace_bytes);
    PostParams pp;
    fill_post_params(pp, params);
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    const dim3 g2(PCH, nstar), blk(TILE);
    hipLaunchKernelGGL(k_post_geom, dim3((nstar + 63) / 64), dim3(64), 0, st, nstar, d_coords,
                       d_parallax, d_parallax_err, w.geom);
    tm.begin("k_post_lnp1");
    hipLaunchKernelGGL(k_post_lnp1, g2, blk, 0, st, pp, capacity, d_sel_idx, d_sel_vals, d_sel_off,
                       w.geom, d_lnprior, d_feh, d_loga, w.lnp1, w.part);
    tm.end();
    tm.begin("k_post_cut2");
    hipLaunchKernelGGL(k_post_count2, g2, blk, 0, st, log(pp.wt_thresh), d_sel_off, w.lnp1, w.part,
                       w.counts, w.mask);
    hipLaunchKernelGGL(k_post_offsets, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, pp, nstar, w.counts,
                       w.offsets, w.off2, w.nbase, w.flags, w.nsel);
    hipLaunchKernelGGL(k_post_scatter2, g2, blk, 0, st, capacity, d_sel_idx, d_sel_vals, d_sel_off,
                       d_lnprior, w.mask, w.offsets, w.rp);
    tm.end();
    {   // objects with more than nsel_max survivors: sort + clip on the device
        std::vector<int32_t> hf(nstar);
        std::vector<int64_t> ho(nstar + 1);
        HIP_TRY(hipMemcpyAsync(hf.data(), w.flags, 4 * (size_t)nstar, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(ho.data(), w.off2, 8 * ((size_t)nstar + 1), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        bool any = false;
        for (int s = 0; s < nstar; ++s)
            if (hf[s]) {
                any = true;
                tm.begin("k_post_clip");
                int rc = clip_to_nsel_max(w, capacity, ho[s], ho[s + 1] - ho[s], pp.nsel_max, st);
                tm.end();
                if (rc) return rc;
            }
        if (any) HIP_TRY(hipMemsetAsync(w.flags, 0, 4 * (size_t)nstar, st));
    }
    tm.begin("k_post_mc");
    {
        const int nitem = PCH * nstar;
        HIP_TRY(hipMemsetAsync(w.mc_counter, 0, 4, st));
        hipLaunchKernelGGL(k_post_mc, dim3(nitem < MC_SLOTS ? nitem : MC_SLOTS), blk, 0, st, pp,
                           capacity, nitem, w.mc_counter, w.mc_stage, d_sel_idx,
                           d_sel_vals, d_sel_off, w.off2, w.nsel, w.nbase, w.flags, w.geom, d_feh,
                           d_loga, w.rp, w.part_max, w.part_chi2);
    }
    tm.end();
    tm.begin("k_post_cdf");
    hipLaunchKernelGGL(k_post_evid_part, g2, blk, 0, st, w.off2, w.nsel, w.flags, w.part_max,
                       w.part_chi2, w.rp, w.part);
    hipLaunchKernelGGL(k_post_wt_part, g2, blk, 0, st, w.off2, w.nsel, w.flags, w.part_max,
                       w.part_chi2, w.part, w.rp, w.part_w);
    hipLaunchKernelGGL(k_post_cdf, g2, blk, 0, st, w.off2, w.nsel, w.flags, w.part_max, w.part_chi2,
                       w.part, w.part_w, w.rp, w.cdf, w.star_out);
    tm.end();
    tm.begin("k_post_draw");
    hipLaunchKernelGGL(k_post_draw, dim3((pp.ndraws + 63) / 64, nstar), dim3(64), 0, st, pp, nstar,
                       capacity, d_sel_idx, d_sel_vals, d_sel_off, w.off2, w.nsel, w.nbase, w.flags,
                       w.geom, d_feh, d_loga, w.rp, w.cdf, w.star_out, d_out_idx, d_out_vals);
    tm.end();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_star_out, w.star_out, 8 * 4 * (size_t)nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_flags, w.flags, 4 * (size_t)nstar, hipMemcpyDeviceToHost, st));
    if (h_nbase)
        HIP_TRY(hipMemcpyAsync(h_nbase, w.nbase, 8 * ((size_t)nstar + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    tm.collect();
    return 0;
}

int brutus_debug_rng(uint64_t seed, uint64_t start, int64_t n, double *d_normals,
                     double *d_uniforms, void *stream) {
    if (!d_normals || !d_uniforms || n <= 0) return fail(BRUTUS_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_debug_normals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, seed, start, n, d_normals, d_uniforms);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_debug_galprior(const brutus_post_params *params, int n, const double *d_dist,
                          const double *d_coord, const double *d_feh, const double *d_loga,
                          double *d_out, void *stream) {
    if (!params || !d_dist || !d_coord || !d_feh || !d_loga || !d_out || n <= 0)
        return fail(BRUTUS_EINVAL, "bad arguments");
    PostParams pp;
    fill_post_params(pp, params);
    hipLaunchKernelGGL(k_debug_galprior, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       pp, n, d_dist, d_coord, d_feh, d_loga, d_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_calibrate_traffic(const float *d_in, double *d_out, int64_t n, void *stream) {
    if (!d_in || !d_out || n <= 0) return fail(BRUTUS_EINVAL, "bad calibration arguments");
    hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(TILE), 0, (hipStream_t)stream, d_in, d_out, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_debug_exp10(const double *d_x, double *d_y, int64_t n, void *stream) {
    if (!d_x || !d_y || n <= 0) return fail(BRUTUS_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_debug_exp10, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, d_x, d_y, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_debug_math(int which, const double *d_x, double *d_y, int64_t n, void *stream) {
    if (!d_x || !d_y || n <= 0 || which < 0 || which > 2) return fail(BRUTUS_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, which, d_x, d_y, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

void brutus_enable_timing(int on) { g_timing = on != 0; }

int brutus_last_timing(int *n_entries, const char **names, float *ms, int max_entries) {
    int n = 0;
    for (auto &t : g_last_timing) {
        if (n >= max_entries) break;
        names[n] = t.name.c_str();
        ms[n] = t.ms;
        ++n;
    }
    if (n_entries) *n_entries = n;
    return 0;
}

}  // extern "C"
