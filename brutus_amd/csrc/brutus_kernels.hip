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
//   utils.py:161-176     _chisquare_logpdf                 (k_finalize)
//   fitting.py:976-991   lnpost parallax clip + first cut  (k_finalize, k_count,
//                                                           k_scatter)
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
    double *lnlp;          // cull statistic, later lnprob of the first cut
    double *lnl, *chi2, *scale, *av, *rv;
    double *icov[6];
    double *step;
    int64_t nmodel;
};

__device__ __forceinline__ unsigned long long dkey(double x) {
    long long b = __double_as_longlong(x);
    return b < 0 ? ~(unsigned long long)b
                 : ((unsigned long long)b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
    long long b = (k & 0x8000000000000000ull) ? (long long)(k & 0x7fffffffffffffffull)
                                              : (long long)~k;
    return __longlong_as_double(b);
}

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

// AoS (nmodel, nfilt, 3) -> SoA [NB][3][nmodel_pad]; padded entries zero.
__global__ void k_relayout(const float *__restrict__ aos, int64_t nmodel, int nfilt, int nb,
                           int64_t nmodel_pad, float *__restrict__ soa) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nmodel_pad) return;
    for (int j = 0; j < nb; ++j)
        for (int k = 0; k < 3; ++k) {
            float v = 0.f;
            if (i < nmodel && j < nfilt) v = aos[(i * nfilt + j) * 3 + k];
            soa[(int64_t)(3 * j + k) * nmodel_pad + i] = v;
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

// Phase 4: constants, dimensionality prior, parallax clip (elementwise).
// fitting.py:806-815 and :976-985.  Overwrites lnlp with lnprob when
// `want_lnprob`; emits per (tile, star) max lnprob.
__global__ void __launch_bounds__(TILE)
k_finalize(int64_t nmodel, int nstar, const StarPrep *__restrict__ stars, DevParams p,
           const double *__restrict__ lnlp_max, int want_lnprob, Planes pl,
           double *__restrict__ part) {
    __shared__ double slot[4];
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    const bool live = i < nmodel;
    const int s = blockIdx.y;
    const StarPrep &sp = stars[s];
    const int64_t o = (int64_t)s * pl.nmodel + i;
    double lnprob = -INFINITY;
    if (live) {
        const bool surv = pl.lnlp[o] > lnlp_max[s] + p.ln_init;
        const double chi2 = pl.chi2[o];
        double lnl = pl.lnl[o];
        if (surv) lnl += sp.lnl_const;                              // fitting.py:806-807
        if (p.dim_prior)                                            // utils.py:161-176
            lnl = chi2 > 0. ? sp.c0 + sp.c1 * log(chi2) - chi2 / 2. : -INFINITY;
        pl.lnl[o] = lnl;
        if (want_lnprob) {
            lnprob = lnl;
            if (sp.sp_on) {                                         // pdf.py:209-218
                const double serr2 = 1. / fabs(pl.icov[0][o]);
                const double vt = sp.sp_var + serr2;
                const double ds = pl.scale[o] - sp.sp_mean;
                lnprob = lnl + -0.5 * (ds * ds / vt + log(2. * M_PI * vt));
            }
            if (!isfinite(lnprob)) lnprob = -BIG;                   // fitting.py:983-985
            pl.lnlp[o] = lnprob;
        }
    }
    if (want_lnprob)
        block_max_store(lnprob, slot, part + ((int64_t)blockIdx.x * nstar + s));
}

// Ordered compaction of {lnprob > max + ln(wt_thresh)} (fitting.py:988-991).
// grid = (NCHUNK, nstar); workgroup (c, s) owns a contiguous range of tiles.
__global__ void __launch_bounds__(TILE)
k_count(int64_t nmodel, int ntile, const double *__restrict__ lnprob,
        const double *__restrict__ pmax, double ln_wt, int64_t *__restrict__ counts) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    const double thr = pmax[s] + ln_wt;
    int n = 0;
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        if (i < nmodel && lnprob[(int64_t)s * nmodel + i] > thr) ++n;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)s * NCHUNK + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void k_scan(int nstar, const int64_t *__restrict__ counts,
                       int64_t *__restrict__ offsets, int64_t *__restrict__ star_off) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t run = 0;
    for (int s = 0; s < nstar; ++s) {
        star_off[s] = run;
        for (int c = 0; c < NCHUNK; ++c) {
            offsets[(int64_t)s * NCHUNK + c] = run;
            run += counts[(int64_t)s * NCHUNK + c];
        }
    }
    star_off[nstar] = run;
}

__global__ void __launch_bounds__(TILE)
k_scatter(int64_t nmodel, int ntile, Planes pl, const double *__restrict__ pmax, double ln_wt,
          const int64_t *__restrict__ offsets, int64_t capacity, int32_t *__restrict__ sel_idx,
          double *__restrict__ sel_vals) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    const double thr = pmax[s] + ln_wt;
    int64_t base = offsets[(int64_t)s * NCHUNK + c];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        const int64_t o = (int64_t)s * nmodel + i;
        const bool sel = i < nmodel && pl.lnlp[o] > thr;
        const unsigned long long b = __ballot(sel);
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(b);
        __syncthreads();
        int woff = 0;
        for (int q = 0; q < w; ++q) woff += wsum[q];
        const int tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (sel) {
            const int64_t r = base + woff + rank;
            if (r < capacity) {
                sel_idx[r] = (int32_t)i;
                sel_vals[0 * capacity + r] = pl.lnl[o];
                sel_vals[1 * capacity + r] = pl.chi2[o];
                sel_vals[2 * capacity + r] = pl.scale[o];
                sel_vals[3 * capacity + r] = pl.av[o];
                sel_vals[4 * capacity + r] = pl.rv[o];
#pragma unroll
                for (int q = 0; q < 6; ++q) sel_vals[(5 + q) * capacity + r] = pl.icov[q][o];
            }
        }
        base += tot;
        __syncthreads();
    }
}

__global__ void k_set_i32(int32_t *p, int n, int32_t v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
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
    double *vmax_prob;  // (S,)
    int32_t *k1;        // (S,)
    int32_t *k2;        // (S,)  >=0 active iteration count, <0 done: -(K2)-1
    int32_t *n_unconv;  // (1,)
    int64_t *counts;    // (S, NCHUNK)
    int64_t *offsets;   // (S, NCHUNK)
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
    w.vmax_prob = (double *)take(sizeof(double) * nstar);
    w.k1 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.k2 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.n_unconv = (int32_t *)take(sizeof(int32_t) * 4);
    w.counts = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
    w.offsets = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
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
                 int max_iter, Workspace &w, bool want_lnprob, int32_t *h_k1, int32_t *h_k2,
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
                       w.vmax_lnlp, want_lnprob ? 1 : 0, w.pl, w.part);
    tm.end();
    if (want_lnprob)
        hipLaunchKernelGGL(k_reduce_decide, dim3(nstar), dim3(256), 0, st, 1, ntile, nstar, 1,
                           w.part, 0.0, w.vmax_prob, (int32_t *)nullptr, (int32_t *)nullptr);
    if (h_k1) HIP_TRY(hipMemcpyAsync(h_k1, w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    if (h_k2) HIP_TRY(hipMemcpyAsync(h_k2, w.k2, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

int dispatch_pipeline(int nb, const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                      int max_iter, Workspace &w, bool want_lnprob, int32_t *h_k1, int32_t *h_k2,
                      hipStream_t st, Timer &tm) {
    switch (nb) {
        case 8: return run_pipeline<8>(grid, nmodel, nstar, p, max_iter, w, want_lnprob, h_k1, h_k2, st, tm);
        case 12: return run_pipeline<12>(grid, nmodel, nstar, p, max_iter, w, want_lnprob, h_k1, h_k2, st, tm);
        case 16: return run_pipeline<16>(grid, nmodel, nstar, p, max_iter, w, want_lnprob, h_k1, h_k2, st, tm);
        case 24: return run_pipeline<24>(grid, nmodel, nstar, p, max_iter, w, want_lnprob, h_k1, h_k2, st, tm);
        case 32: return run_pipeline<32>(grid, nmodel, nstar, p, max_iter, w, want_lnprob, h_k1, h_k2, st, tm);
    }
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
    return (size_t)nb * 3 * (size_t)pad_models(nmodel) * sizeof(float);
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
    int rc = dispatch_pipeline(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, max_iter, w, false,
                               h_k1, h_k2, st, tm);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    fix_k2(h_k2, nstar);
    tm.collect();
    return 0;
}

int brutus_fit_gather(int64_t nmodel, int nfilt, int nstar, void *d_workspace,
                      size_t workspace_bytes, double wt_thresh, int64_t capacity,
                      int32_t *d_sel_idx, double *d_sel_vals, int64_t *d_sel_off,
                      void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    if (!d_workspace || !d_sel_idx || !d_sel_vals || !d_sel_off || capacity < 0)
        return fail(BRUTUS_EINVAL, "bad gather arguments");
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes) return fail(BRUTUS_ENOMEM, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int ntile = (int)(pad_models(nmodel) / TILE);
    const double ln_wt = wt_thresh > 0. ? log(wt_thresh) : -INFINITY;
    hipLaunchKernelGGL(k_count, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile, w.pl.lnlp,
                       w.vmax_prob, ln_wt, w.counts);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), 0, st, nstar, w.counts, w.offsets, d_sel_off);
    hipLaunchKernelGGL(k_scatter, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile, w.pl,
                       w.vmax_prob, ln_wt, w.offsets, capacity, d_sel_idx, d_sel_vals);
    HIP_TRY(hipGetLastError());
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
    int rc = dispatch_pipeline(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, max_iter, w, true,
                               h_k1, h_k2, st, tm);
    if (rc) return rc;
    const int ntile = (int)(pad_models(nmodel) / TILE);
    tm.begin("k_select");
    hipLaunchKernelGGL(k_count, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile, w.pl.lnlp,
                       w.vmax_prob, p.ln_wt, w.counts);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), 0, st, nstar, w.counts, w.offsets, d_sel_off);
    hipLaunchKernelGGL(k_scatter, dim3(NCHUNK, nstar), dim3(TILE), 0, st, nmodel, ntile, w.pl,
                       w.vmax_prob, p.ln_wt, w.offsets, capacity, d_sel_idx, d_sel_vals);
    tm.end();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    fix_k2(h_k2, nstar);
    tm.collect();
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
