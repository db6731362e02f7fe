// post_kernels.hpp -- device lnpost (second cut, MC prior integral, resampling) behind brutus_post_batch
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

namespace {

// ===========================================================================
// lnpost on the device (fitting.py:1000-1107 and the tail of _fit, :2021-2061)
// for the built-in priors, with the counter-based random stream specified in
// brutus_amd/rng.py (Philox4x32-7 + ziggurat normals): any deviate is a pure
// function of (seed, index), so every selected model of every object is
// integrated in parallel and the result still equals, deviate for deviate, a
// sequential run of the reference with that `rstate` object.
// ===========================================================================
struct Philox4 {
    uint32_t w[4];
};

#define ZIG_TABLE_QUAL __device__ const
#include "zig_table.inc"      // ZIG_N, kZigX[ZIG_N + 1], kZigY[ZIG_N + 1]
#undef ZIG_TABLE_QUAL

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
    const Philox4 o = philox4x32_7((uint32_t)q, (uint32_t)(q >> 32), 0u, 1u /* STREAM_UNIFORM */,
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
    return u53(o.w[0], o.w[1]);
}

// ---- normal stream (rng.py: philox_normal): ziggurat, 1024 layers, two normals per call ----
constexpr uint32_t STREAM_NORMAL = 0u, STREAM_UNIFORM = 1u, STREAM_RETRY = 2u, STREAM_TAIL = 3u;

// 64 random bits -> layer i, sign and x = u X[i]; true inside the layer's rectangle (99.57 %:
// the normal is then +-x).  One multiplication: host and device round alike, the decision is
// bit-exact.  `X` is kZigX or an LDS copy of it (stage_zig_table).
__device__ __forceinline__ bool zig_try(uint32_t a, uint32_t b, const double *__restrict__ X,
                                        double &x, int &i, bool &neg) {
    i = (int)(b & (uint32_t)(ZIG_N - 1));
    neg = (b >> 10) & 1u;
    const double u = ldexp((double)a * 2097152.0 + (double)(b >> 11), -53);     // exact
    x = u * X[i];
    return x < X[i + 1];
}
// +-x for x >= 0 with the sign zig_try reads (bit 10 of `b`): two integer operations on the
// high word instead of a compare and two selects
__device__ __forceinline__ double zig_signed(double x, uint32_t b) {
    return __hiloint2double(__double2hiint(x) | (int)((b << 21) & 0x80000000u), __double2loint(x));
}
__device__ __forceinline__ Philox4 philox_at(uint64_t seed, uint64_t idx, uint32_t c2, uint32_t stream) {
    return philox4x32_7((uint32_t)idx, (uint32_t)(idx >> 32), c2, stream, (uint32_t)seed,
                        (uint32_t)(seed >> 32));
}
// normal j whose attempt 0 (layer i, x, sign) fell outside the rectangle: wedge test or tail,
// further attempts (rng.py).  0.43 % of the normals come here; ocml's exp / log.
__device__ __forceinline__ double zig_slow(uint64_t seed, uint64_t j, double x, int i, bool neg) {
#pragma clang fp contract(off)     // Y[i] + u2 (Y[i+1] - Y[i]) rounds like numpy's
    for (uint32_t r = 0;; ++r) {
        const Philox4 v = philox_at(seed, j, r, STREAM_RETRY);
        if (r > 0 && zig_try(v.w[0], v.w[1], kZigX, x, i, neg)) break;
        if (i == 0) {
            const double R = kZigX[1];
            for (uint32_t k = 0;; ++k) {
                const Philox4 t = philox_at(seed, j, (r << 16) | k, STREAM_TAIL);
                const double xt = -log(1.0 - u53(t.w[0], t.w[1])) / R;
                const double yt = -log(1.0 - u53(t.w[2], t.w[3]));
                if (yt + yt > xt * xt) {
                    x = R + xt;
                    break;
                }
            }
            break;
        }
        const double u2 = u53(v.w[2], v.w[3]);
        const double yl = kZigY[i] + u2 * (kZigY[i + 1] - kZigY[i]);
        if (yl < exp(-0.5 * x * x)) break;
    }
    return neg ? -x : x;
}
// the two normals 2q, 2q + 1 of call q
__device__ __forceinline__ void rng_normal_call(uint64_t seed, uint64_t q, const double *__restrict__ X,
                                                double &z0, double &z1) {
    const Philox4 o = philox_at(seed, q, 0u, STREAM_NORMAL);
    double x0, x1;
    int i0, i1;
    bool n0, n1;
    const bool f0 = zig_try(o.w[0], o.w[1], X, x0, i0, n0), f1 = zig_try(o.w[2], o.w[3], X, x1, i1, n1);
    z0 = n0 ? -x0 : x0;
    z1 = n1 ? -x1 : x1;
    if (!f0) z0 = zig_slow(seed, 2 * q, x0, i0, n0);
    if (!f1) z1 = zig_slow(seed, 2 * q + 1, x1, i1, n1);
}
__device__ __forceinline__ double rng_normal(uint64_t seed, uint64_t j) {
    double z0, z1;
    rng_normal_call(seed, j >> 1, kZigX, z0, z1);
    return (j & 1) ? z1 : z0;
}
// copy kZigX to LDS (ZIG_N + 1 doubles); follow with __syncthreads()
__device__ __forceinline__ void stage_zig_table(double *lds_x) {
    for (int k = threadIdx.x; k <= ZIG_N; k += blockDim.x) lds_x[k] = kZigX[k];
}
// The generator loop of k_post_mc reads the table as pairs P[i] = (X[i] 2^-53, X[i+1]): one
// 16-byte LDS read per normal, and the 53-bit integer U is multiplied by the scaled edge
// directly -- U (X[i] 2^-53) and (U 2^-53) X[i] are the same double (a power of two scales
// exactly), so the rectangle decision and the value stay bit-equal to zig_try's.
__device__ __forceinline__ void stage_zig_pairs(double2 *lds_p) {
    for (int k = threadIdx.x; k < ZIG_N; k += blockDim.x)
        lds_p[k] = make_double2(ldexp(kZigX[k], -53), kZigX[k + 1]);
}
__device__ __forceinline__ bool zig_try_p(uint32_t a, uint32_t b, const double2 *__restrict__ P, double &x) {
    const double2 e = P[b & (uint32_t)(ZIG_N - 1)];
    x = ((double)a * 2097152.0 + (double)(b >> 11)) * e.x;
    return x < e.y;
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
    double frame_mat[9], frame_off[3];     // Galactic -> Galactocentric (galprior.astropy_frame)
    // derived on the host side of the ABI call (not part of brutus_post_params)
    double ln_f_thick, ln_f_halo, inv_reff_solar2;
    double inv_R_thin, inv_Z_thin, inv_R_thick, inv_Z_thick, inv_r_q;
    double Rs_thin2, Rs_thick2, Rs_halo2, rq2, abs_Z_solar;
    double lnK, c0_thin, c0_thick, c0_halo;      // component constants relative to lnK
    // label_terms: -1 / (2 sigma^2), -ln(2 pi sigma^2) / 2; 1 / sigma_age, -ln(2 pi) / 2 - lnnorm
    double feh_nh_isig2[3], feh_c0[3], age_isig[3], age_c0[3];
    // halo_pow: binomial coefficients of (1 + r)^(-eta/2), r^1 .. r^7; halo_tbl != 0 when the
    // table form is valid for these parameters (fill_post_params)
    double halo_b[7], halo_tbl;
};
constexpr int POST_DERIVED = 37;

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
    double ux, uy, uz;           // frame_mat @ (cos b cos l, cos b sin l, sin b): the sightline's
                                 // direction in the Galactocentric frame
    double par, par_ivar, par_lnorm;
    int has_par;
    // line-of-sight dust prior (pdf.py:752-840 with a caller-supplied table): los points
    // at dist[nd], Av_mean[nd], Av_err[nd] of this object's sightline
    int dust_on, nd;
    const double *los;
    double d_off, d_scale, d_smooth, d_scat2;
    // constants of the Monte Carlo integrand for this object (k_post_geom, McC): read by
    // scalar loads inside the sample loop (mc_sample_c)
    double mc[32];
};

// Layout of StarGeom::mc.  With the sightline's direction u and the frame offset o,
// R^2(d) = |d u_xy + o_xy|^2 = A2 d^2 + A1 d + A0 and Z(d) = uz d + o2; the disks'
// exponents with |Z_sun| folded into the constant, exp(1 - x) = e exp(-x) folded into dq.
enum McC {
    MC_A2, MC_A1, MC_A0, MC_UZ, MC_O2, MC_RS_THIN2, MC_RS_THICK2, MC_C0T, MC_IRT, MC_IZT, MC_C0K,
    MC_IRK, MC_IZK, MC_QINF, MC_DQE, MC_RQ2, MC_IRQ, MC_RS_HALO2, MC_B1, MC_B2, MC_B3, MC_B4, MC_B5,
    MC_B6, MC_B7, MC_AV0, MC_AV1, MC_RV0, MC_RV1, MC_PAR, MC_PIVAR, MC_NC
};
static_assert(MC_NC <= 32, "StarGeom::mc");
// pointer into the constant address space: uniform loads through it are scalar loads
typedef const double __attribute__((address_space(4))) *CPtr;
// the same pointer, opaque to the optimiser: loads through the result cannot be hoisted
// above this point, so the constants are fetched (s_load) per use instead of being held in --
// and, beyond ~100, spilled from -- scalar registers for the whole loop
__device__ __forceinline__ CPtr mc_refresh(CPtr p) {
    uint64_t a = (uint64_t)p;
    asm volatile("" : "+s"(a));
    return (CPtr)a;
}

struct DustCtx {       // host side of brutus_post_set_dust
    const double *d_los;     // (nstar, 3, nd)
    const int32_t *d_ok;     // (nstar,) 0: no coverage on that sightline -> flat prior
    int nd;
    double offset, scale, smooth, scatter;
};

// ln prior of Av at distance `dist` [kpc]: Gaussian around the profile interpolated like
// numpy.interp (end values outside the table)
__device__ __forceinline__ double dust_lnp(const StarGeom &g, double dist, double av) {
    const double *xp = g.los, *fm = g.los + g.nd, *fe = g.los + 2 * g.nd;
    double m, e;
    if (!(dist > xp[0])) {
        m = dist == dist ? fm[0] : dist;
        e = fe[0];
    } else if (dist >= xp[g.nd - 1]) {
        m = fm[g.nd - 1];
        e = fe[g.nd - 1];
    } else {
        int lo = 0, hi = g.nd - 1;                 // xp[lo] <= dist < xp[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (xp[mid] <= dist) lo = mid; else hi = mid;
        }
        const double dx = xp[lo + 1] - xp[lo], t = dist - xp[lo];
        m = (fm[lo + 1] - fm[lo]) / dx * t + fm[lo];
        e = (fe[lo + 1] - fe[lo]) / dx * t + fe[lo];
    }
    const double mean = g.d_scale * m + g.d_off;
    const double er = g.d_smooth * g.d_scale * e;
    const double e2 = er * er + g.d_scat2;
    const double dv = av - mean;
    return -0.5 * (dv * dv / e2 + log(2. * M_PI * e2));
}

__device__ __forceinline__ double lse3(double a, double b, double c) {
    double m = a > b ? a : b;
    m = c > m ? c : m;
    if (!(m > -INFINITY)) return m;          // all -inf (or NaN)
    return log(exp(a - m) + exp(b - m) + exp(c - m)) + m;
}

// per-model metallicity / age densities of the three components (pdf.py:380-473),
// as plain (not log) values: e^F_c, e^A_c.  Table + polynomial forms (<= 2 ulp) with the
// per-component constants from fill_post_params: ~130 instructions instead of ~400 with
// ocml's exp / log / exp10 -- k_post_lnp1 spends half its time here, and k_post_mc_arr
// evaluates it on the eight lanes of a record.
__device__ __forceinline__ void label_terms(const PostParams &pp, double feh, double loga,
                                            double (&Fc)[3], double (&Ac)[3],
                                            const double *__restrict__ tbl = kExp2Tbl) {
    const double age = pp.has_loga ? fast_exp10(loga - 9., tbl) : 0.;       // Gyr
    const bool age_out = age < pp.min_age || age > pp.max_age;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Fc[c] = 1.;
        Ac[c] = 1.;
        if (pp.has_feh) {
            const double d = pp.feh_mean[c] - feh;
            Fc[c] = fast_exp_fin(fma(d * d, pp.feh_nh_isig2[c], pp.feh_c0[c]), tbl);
        }
        if (pp.has_loga) {
            const double xi = (age - pp.age_mean[c]) * pp.age_isig[c];
            Ac[c] = age_out ? 0. : fast_exp_fin(fma(-0.5 * xi, xi, pp.age_c0[c]), tbl);
        }
    }
}

// The halo's power law f (reff / reff_sun)^-eta without a logarithm and an exponential.
// With Y = reff^2 = 2^e c_k (1 + r), c_k = 1 + (k + 1/2) / 128 the centre of the mantissa's
// k-th 1/128 step, h = eta / 2 and X = Y / reff_sun^2:
//   e^c0_halo X^-h = [e^c0_halo reff_sun^(2h) 2^(-h e)] [c_k^-h] (1 + r)^-h,   |r| <= 1/256,
// two table entries and a degree-7 binomial series (next term < 2^-54 for the h that
// fill_post_params admits): 10 float64 operations and a handful of integer ones instead of
// the ~50 of fast_exp_fin(c0 - h fast_log_pos(X)); <= 4 ulp.  HALO_E0 bounds Y from below
// (fill_post_params checks Rs_halo^2 >= 2^-HALO_E0; an in-bounds sample has
// dist <= 1e10 kpc, far inside the 128 exponents tabulated).
// Table layout (LDS, stage_halo_table): [0, 128) e^c0_halo reff_sun^(2h) 2^(-h (i - HALO_E0)),
// [128, 256) c_k^-h, [256, 384) 1 / c_k.
constexpr int HALO_E0 = 8, HALO_TBL = 384;
__device__ __forceinline__ void stage_halo_table(const PostParams &pp, double *ht) {
    if (pp.halo_tbl == 0.) return;
    const double h = 0.5 * pp.eta_halo;
    for (int k = threadIdx.x; k < 128; k += blockDim.x) {
        const double ic = 1. / (1. + ((double)k + 0.5) * (1. / 128.));
        ht[256 + k] = ic;
        ht[128 + k] = pow(ic, h);              // (1 / ic)^-h with the ROUNDED 1 / c_k: consistent with r
        ht[k] = exp(pp.c0_halo) * pow(pp.inv_reff_solar2, -h) * pow(2., -h * (double)(k - HALO_E0));
    }
}
template <class B>
__device__ __forceinline__ double halo_pow(const B &hb, double Y, const double *__restrict__ ht) {
    const unsigned hi = (unsigned)__double2hiint(Y);
    const unsigned ie = ((hi >> 20) - (unsigned)(1023 - HALO_E0)) & 127u;
    const unsigned k = (hi >> 13) & 127u;
    const double m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(Y));
    const double r = fma(m, ht[256 + k], -1.);
    double pl = hb[6];
    pl = fma(pl, r, hb[5]);
    pl = fma(pl, r, hb[4]);
    pl = fma(pl, r, hb[3]);
    pl = fma(pl, r, hb[2]);
    pl = fma(pl, r, hb[1]);
    pl = fma(pl, r, hb[0]);
    pl = fma(pl, r, 1.);
    return (ht[ie] * ht[128 + k]) * pl;
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
                                                const double *__restrict__ tbl,
                                                const double *__restrict__ ht = nullptr) {
    // Galactocentric position (reference pdf.py:631-635): frame offset + d * direction
    const double x = fma(d, g.ux, pp.frame_off[0]), y = fma(d, g.uy, pp.frame_off[1]),
                 Z = fma(d, g.uz, pp.frame_off[2]);
    const double R2 = x * x + y * y;
    const double dZ = fabs(Z) - pp.abs_Z_solar;
    const double Rt = fast_sqrt(R2 + pp.Rs_thin2);
    const double Rk = pp.Rs_thick2 == pp.Rs_thin2 ? Rt : fast_sqrt(R2 + pp.Rs_thick2);
    // thin / thick disk: exp(-(R - R_sun)/R_c - (|Z| - |Z_sun|)/Z_c [+ ln f] - lnK)
    const double T0 = fast_exp_fin(pp.c0_thin - (Rt * pp.inv_R_thin + dZ * pp.inv_Z_thin), tbl);
    const double T1 = fast_exp_fin(pp.c0_thick - (Rk * pp.inv_R_thick + dZ * pp.inv_Z_thick), tbl);
    // halo: f (reff / reff_sun)^-eta, reff^2 = R^2 + (Z/q)^2 + Rs^2, q(r) (pdf.py:341-365)
    const double q = pp.q_halo_inf -
                     (pp.q_halo_inf - pp.q_halo_ctr) *
                         fast_exp_fin(1. - fast_sqrt(R2 + Z * Z + pp.rq2) * pp.inv_r_q, tbl);
    const double zq = Z * fast_rcp(q);
    const double Y = R2 + zq * zq + pp.Rs_halo2;
    const double T2 = ht ? halo_pow(pp.halo_b, Y, ht)
                         : fast_exp_fin(pp.c0_halo - 0.5 * pp.eta_halo * fast_log_pos(Y * pp.inv_reff_solar2), tbl);
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
k_post_lnp1(PostParams pp, int64_t cap, const int32_t *__restrict__ sel_idx, const int32_t *__restrict__ rec_slot,
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
            const int64_t i = sel_idx[r], vs = rec_slot[r];       // vs: slot of the record's values
            double Fc[3], Ac[3];
            label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac, s_tbl);
            const double scale = sel_vals[2 * cap + vs];
            const double dist = 1. / sqrt(scale);
            double v = sel_vals[vs] + lnprior[i] + gal_lnprior_dev(pp, g, dist, Fc, Ac, s_tbl);
            if (g.dust_on) v += dust_lnp(g, dist, sel_vals[3 * cap + vs]);      // fitting.py:1009-1010
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
    double *chol;     // [6][cap]  L00 L10 L11 L20 L21 L22 of the record's covariance
    // (the covariance itself is not stored: only the Ndraws resampled records of an object
    // report it, and k_post_draw derives it again with rec_cov -- 48 bytes per kept record
    // less to write, permute and hold)
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

// covariance of the record at value slot `vs`: inverse of its precision matrix with the
// reference's repair loop for matrices that are not positive definite (fitting.py:1039-1065)
__device__ __forceinline__ void rec_cov(const double *__restrict__ sel_vals, int64_t cap, int64_t vs,
                                        double (&C)[6]) {
    double A[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) A[k] = sel_vals[(int64_t)(5 + k) * cap + vs];
    inv3_sym(A, C);
    const double scale = sel_vals[2 * cap + vs];
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
}

__global__ void __launch_bounds__(TILE)
k_post_scatter2(int64_t cap, const int32_t *__restrict__ sel_idx, const int32_t *__restrict__ rec_slot, const double *__restrict__ sel_vals,
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
            const int64_t vs = rec_slot[r];
            rp.lnp[o] = sel_vals[vs] + lnprior[sel_idx[r]];
            double C[6];
            rec_cov(sel_vals, cap, vs, C);
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

// Sequential reader of normals j0, j0+1, ...: each Philox call is evaluated once.
struct NormalReader {
    uint64_t seed, p;
    double z0, z1;
    bool have;
    const double *zs;      // non-null: the object's normals are in memory (k_mt_stream)
    const ZMap *zm;        // non-null: ... where pass 1 of the stream walk left them (object `obj`)
    int obj;
    __device__ __forceinline__ void init(uint64_t seed_, const double *zs_ = nullptr,
                                         const ZMap *zm_ = nullptr, int obj_ = 0) {
        seed = seed_;
        have = false;
        p = 0;
        zs = zs_;
        zm = zm_;
        obj = obj_;
    }
    __device__ __forceinline__ double at(uint64_t j) {
        if (zm) return zmap_at(*zm, obj, (int64_t)j);
        if (zs) return zs[j];
        const uint64_t q = j >> 1;
        if (!have || q != p) {
            rng_normal_call(seed, q, kZigX, z0, z1);
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
                                              double &lin, double &epar,
                                              const double *__restrict__ ht = nullptr) {
    const double s_mc = s0 + L[0] * z0;
    a_mc = a0 + (L[1] * z0 + L[2] * z1);
    r_mc = r0 + (L[3] * z0 + L[4] * z1 + L[5] * z2);
    double par;
    fast_sqrt_rsqrt(s_mc, par, dist);                           // parallax and distance (~1 ulp)
    lin = gal_prior_lin(pp, g, dist, Fc, Ac, tbl, ht);
    const double dp = par - g.par;                              // pdf.py:166-173
    epar = g.has_par ? -0.5 * (dp * dp * g.par_ivar) : 0.;
    if (g.dust_on) epar += dust_lnp(g, dist, a_mc);             // fitting.py:1084-1085
    inb = s_mc >= 1e-20 && a_mc >= pp.avlim[0] && a_mc <= pp.avlim[1] &&
          r_mc >= pp.rvlim[0] && r_mc <= pp.rvlim[1];
}

// the three density components along the object's sightline at distance d, relative to e^lnK
// (`aZ` = |Z|, or sigma Z with a fixed sign for the smooth continuation the sightline table fits)
__device__ __forceinline__ void sightline_T(CPtr cb, bool one_rs, double d, double Z, double aZ,
                                            const double *__restrict__ tbl, const double *__restrict__ ht,
                                            double &T0, double &T1, double &T2) {
    const double R2 = fmax(fma(fma(cb[MC_A2], d, cb[MC_A1]), d, cb[MC_A0]), 0.);
    const double Rt = fast_sqrt_pos(R2 + cb[MC_RS_THIN2]);      // (Rs^2, rq^2 > 0: fill_post_params)
    const double Rk = one_rs ? Rt : fast_sqrt_pos(R2 + cb[MC_RS_THICK2]);
    T0 = fast_exp_fin(cb[MC_C0T] - fma(Rt, cb[MC_IRT], aZ * cb[MC_IZT]), tbl);
    T1 = fast_exp_fin(cb[MC_C0K] - fma(Rk, cb[MC_IRK], aZ * cb[MC_IZK]), tbl);
    const double q = cb[MC_QINF] -
                     cb[MC_DQE] * fast_exp_fin(-(fast_sqrt_pos(fma(Z, Z, R2) + cb[MC_RQ2]) * cb[MC_IRQ]), tbl);
    const double zq = Z * fast_rcp(q);
    struct HB {
        CPtr c;
        __device__ __forceinline__ double operator[](int k) const { return c[MC_B1 + k]; }
    };
    T2 = halo_pow(HB{cb}, fma(zq, zq, R2) + cb[MC_RS_HALO2], ht);
}
// volume factor x mixture of the components with the record's label weights (gal_prior_lin's tail)
__device__ __forceinline__ double mc_mix(const PostParams &pp, double d, double T0, double T1, double T2,
                                         const double (&EF)[3], const double (&EA)[3]) {
    double num = d * d + 1e-300;                    // volume factor (pdf.py:626)
    if (pp.has_feh) num *= T0 * EF[0] + T1 * EF[1] + T2 * EF[2];
    if (pp.has_loga) num *= T0 * EA[0] + T1 * EA[1] + T2 * EA[2];
    const double S = T0 + T1 + T2;
    const int npow = (pp.has_feh ? 1 : 0) + (pp.has_loga ? 1 : 0);
    if (npow == 2) num *= fast_rcp(S);
    else if (npow == 0) num *= S;
    return num;
}

// ---- sightline table -------------------------------------------------------------------
// T0, T1, T2 depend on the sample only through its distance, and the samples of one work item
// (object, chunk) all lie on one sightline: the item's workgroup tabulates the three functions
// once, in LDS, and a sample evaluates three degree-7 polynomials (21 fused multiply-adds, 12
// 16-byte LDS reads) in place of three square roots, three exponentials, a reciprocal and the
// halo's power law (~100 float64 operations).
//   * abscissa: s = 1 / d^2, the quantity a sample is drawn in -- the interval of a sample is
//     bits 62..48 of s (exponent and four mantissa bits: SL_K = 16 equal intervals per octave of
//     s), the position t in [-1, 1) inside it is the rest of the mantissa: no logarithm, no
//     division, exact;
//   * window: the SL_NI = 144 intervals (nine octaves of s, a factor 22.6 in distance) ending
//     with the one that holds max(s0 + 4 sigma_s) over the item's records; a sample outside it
//     takes the closed form (wave-divergent, rare);
//   * fit: interpolation at the eight Chebyshev nodes of the interval, turned into monomial
//     coefficients with the inverse Vandermonde matrix of tools/gen_sl_vinv.py.  Mixture error
//     <= 2e-12 relative over 32 sightlines x 2^-7 .. 2^8 kpc (tools/dev/sightline_fit.py; the
//     GPU test bounds it on 10^6 distances against the closed form);
//   * certified: interpolation at Chebyshev nodes errs most at t = 0, where the polynomial is its
//     constant coefficient: the builder evaluates the closed form there once more and an interval
//     where any COMPONENT is off by more than SL_TOL of itself (the disks far above the plane,
//     where an interval spans several scale heights; a component of e^-50 still decides the label
//     terms of a model whose age or metallicity the other components exclude) is left to the
//     closed form together with every interval beyond it (`lo`);
//   * |Z|: T0, T1 have a kink where the sightline crosses the plane Z = 0.  The (at most one)
//     interval with the crossing holds two fits, each of the smooth continuation with the sign
//     of Z fixed (row SL_NI for Z < 0); the sample picks by the sign of its own Z.
#ifndef BRUTUS_SL_NOCT
#define BRUTUS_SL_NOCT 9
#endif
constexpr int SL_K_LOG2 = 4, SL_NOCT = BRUTUS_SL_NOCT, SL_NI = SL_NOCT << SL_K_LOG2, SL_ROW = 3 * 8;
#ifdef BRUTUS_NO_SIGHTLINE_TABLE          // (A/B builds: the closed form for every sample)
constexpr bool SL_ON = false;
#else
constexpr bool SL_ON = true;
#endif
constexpr double SL_TOL = 2.5e-12;
#ifndef BRUTUS_SL_MIN_REC
#define BRUTUS_SL_MIN_REC 256
#endif
constexpr int SL_MIN_REC = BRUTUS_SL_MIN_REC;    // records of a work item from which it gets a table
#include "sl_vinv.inc"
static_assert(SL_NODES == 8, "degree-7 fits");
struct SlTab {
    const double2 *c;     // LDS: (SL_NI + 1) rows of SL_ROW doubles, function-major, t^0 .. t^7
    int base, kink;       // bits 62..48 of s of the first interval; interval with the Z = 0 crossing or -1
    int ni;               // intervals (row ni: the second fit of the crossing's interval)
    int lo;               // intervals [lo, ni) are certified
};
// window of an item from the largest s its samples are expected at
__device__ __forceinline__ int sl_window(double top, int ni) {
    const int p = top > 0x1p-900 && top < 0x1p900 ? (__double2hiint(top) >> 16) : 0x3ff0;   // (else: around s = 1)
    return p + 1 - ni;
}
// all threads of the workgroup; follow with __syncthreads().  `kink` is an LDS int set to -1
// (and synchronised) by the caller.
__device__ __forceinline__ void sl_build(CPtr cb, bool one_rs, const double *__restrict__ tbl,
                                         const double *__restrict__ ht, int base, int ni,
                                         double *__restrict__ sl, int *__restrict__ kink, int *__restrict__ lo) {
    for (int i = threadIdx.x; i < ni; i += blockDim.x) {
        const double s_lo = __hiloint2double((base + i) << 16, 0), s_hi = __hiloint2double((base + i + 1) << 16, 0);
        const double h = 0.5 * (s_hi - s_lo), mid = s_lo + h;
        double pa, da, pb, db;
        fast_sqrt_rsqrt_pos(s_lo, pa, da);
        fast_sqrt_rsqrt_pos(s_hi, pb, db);
        const bool na = fma(da, cb[MC_UZ], cb[MC_O2]) < 0., nb = fma(db, cb[MC_UZ], cb[MC_O2]) < 0.;
        const int nfit = na != nb ? 2 : 1;
        if (nfit == 2) *kink = i;
#pragma unroll 1
        for (int f = 0; f < nfit; ++f) {
            const double sg = nfit == 2 ? (f == 0 ? 1. : -1.) : (na ? -1. : 1.);
            double *const row = sl + (f == 0 ? i : ni) * SL_ROW;
#pragma unroll 1
            for (int j = 0; j < SL_NODES; ++j) {
                double par, d, T0, T1, T2;
                fast_sqrt_rsqrt_pos(fma(h, kSlNode[j], mid), par, d);
                const double Z = fma(d, cb[MC_UZ], cb[MC_O2]);
                sightline_T(cb, one_rs, d, Z, sg * Z, tbl, ht, T0, T1, T2);
                row[j] = T0;
                row[8 + j] = T1;
                row[16 + j] = T2;
            }
#pragma unroll 1
            for (int fn = 0; fn < 3; ++fn) {
                double v[SL_NODES], c[SL_NODES];
#pragma unroll
                for (int j = 0; j < SL_NODES; ++j) v[j] = row[fn * 8 + j];
#pragma unroll
                for (int m = 0; m < SL_NODES; ++m) {
                    double a = 0.;
#pragma unroll
                    for (int j = 0; j < SL_NODES; ++j) a = fma(kSlVinv[m][j], v[j], a);
                    c[m] = a;
                }
#pragma unroll
                for (int m = 0; m < SL_NODES; ++m) row[fn * 8 + m] = c[m];
            }
            {
                double par, d, T0, T1, T2;
                fast_sqrt_rsqrt_pos(mid, par, d);
                const double Z = fma(d, cb[MC_UZ], cb[MC_O2]);
                sightline_T(cb, one_rs, d, Z, sg * Z, tbl, ht, T0, T1, T2);
                const bool ok = fabs(row[0] - T0) <= fma(SL_TOL, T0, 1e-290) && fabs(row[8] - T1) <= fma(SL_TOL, T1, 1e-290) &&
                                fabs(row[16] - T2) <= fma(SL_TOL, T2, 1e-290);
                if (!ok) atomicMax(lo, i + 1);
            }
        }
    }
}
__device__ __forceinline__ double sl_poly(const double2 *__restrict__ c, double t) {
    const double2 c01 = c[0], c23 = c[1], c45 = c[2], c67 = c[3];
    double p = fma(c67.y, t, c67.x);
    p = fma(p, t, c45.y);
    p = fma(p, t, c45.x);
    p = fma(p, t, c23.y);
    p = fma(p, t, c23.x);
    p = fma(p, t, c01.y);
    return fma(p, t, c01.x);
}

// LDS of a workgroup that keeps a sightline table, besides the rows themselves
struct SlCtl {
    double top[4];
    int kink, lo;
};
// The table of one work item: records [a, b) of the list `rp` of the object whose selected
// rows start at `row0`.  All threads of the workgroup (barriers inside).
__device__ __forceinline__ SlTab sl_item(CPtr cb, bool one_rs, const double *__restrict__ tbl,
                                         const double *__restrict__ ht, int ni, double2 *__restrict__ rows,
                                         SlCtl *__restrict__ ctl, int64_t a, int64_t b, bool on,
                                         const int32_t *__restrict__ rec_slot, int64_t row0, const RecPost &rp,
                                         const double *__restrict__ sel_vals, int64_t cap) {
    // (building the table costs about what 50 records' samples save: a short item -- sharp
    // posteriors keep ~10^3 models per object, 15 per item -- goes without, uniformly for the workgroup)
    if (b - a < SL_MIN_REC) {
        SlTab none;
        none.c = rows;
        none.base = 0;
        none.kink = -1;
        none.ni = none.lo = 0;
        return none;
    }
    // window: the ni intervals ending with the one of max(s0 + 4 sigma_s) over the records (centring
    // it on the records' mean octave instead made no difference: r05_sightline_table_ab.txt)
    double top = 0.;
    if (threadIdx.x == 0) ctl->kink = -1, ctl->lo = 0;
    __syncthreads();
    if (on)
        for (int64_t o = a + threadIdx.x; o < b; o += blockDim.x)
            top = fmax(top, fma(4., rp.chol[o], sel_vals[2 * cap + rec_slot[row0 + rp.src[o]]]));
    top = wave_max(top);
    if ((threadIdx.x & 63) == 0) ctl->top[threadIdx.x >> 6] = top;
    __syncthreads();
    SlTab sl;
    sl.ni = ni;
    sl.base = __builtin_amdgcn_readfirstlane(
        sl_window(fmax(fmax(ctl->top[0], ctl->top[1]), fmax(ctl->top[2], ctl->top[3])), ni));
    sl_build(cb, one_rs, tbl, ht, sl.base, ni, reinterpret_cast<double *>(rows), &ctl->kink, &ctl->lo);
    __syncthreads();
    sl.c = rows;
    sl.kink = __builtin_amdgcn_readfirstlane(ctl->kink);
    sl.lo = __builtin_amdgcn_readfirstlane(ctl->lo);
    return sl;
}

// mc_sample_lin with the object's constants read through `cb` (StarGeom::mc, McC) and the
// halo table: the form the two Monte Carlo kernels run when the table is valid.  Same
// quantities; R^2 from the quadratic in d (rounding differs from x^2 + y^2 by ~1e-15
// relative to R^2 + Rs^2), clamped at 0 against cancellation on a sightline through the
// Galactic centre's axis.  With a sightline table (`sl`) the components come from it when the
// sample lies inside its window (`tab` says so).
template <bool SL = false>
__device__ __forceinline__ void mc_sample_c(CPtr cb, const PostParams &pp, const StarGeom &g, bool has_par,
                                            bool dust_on, bool one_rs, double z0, double z1, double z2,
                                            double s0, double a0, double r0, const double (&L)[6],
                                            const double (&EF)[3], const double (&EA)[3],
                                            const double *__restrict__ tbl, const double *__restrict__ ht,
                                            bool &inb, double &lin, double &epar, SlTab sl = SlTab{},
                                            bool *tab = nullptr) {
    const double s_mc = s0 + L[0] * z0;
    const double a_mc = a0 + (L[1] * z0 + L[2] * z1);
    const double r_mc = r0 + (L[3] * z0 + L[4] * z1 + L[5] * z2);
    double par, d;
    fast_sqrt_rsqrt_pos(s_mc, par, d);       // (s_mc <= 0: NaN, and the sample is out of bounds)
    const double Z = fma(d, cb[MC_UZ], cb[MC_O2]);
    double T0, T1, T2;
    if constexpr (SL) {
        const int hi = __double2hiint(s_mc), lo = __double2loint(s_mc);
        const int idx = (hi >> 16) - sl.base;
        const bool in = (unsigned)(idx - sl.lo) < (unsigned)(sl.ni - sl.lo);
        if (tab) *tab = in;
        if (in) {
            const int row = (idx == sl.kink && Z < 0.) ? sl.ni : idx;
            // mantissa bits 47..0 of s as a number in [2, 4), minus 3
            const unsigned mh = (((unsigned)hi << SL_K_LOG2) | ((unsigned)lo >> (32 - SL_K_LOG2))) & 0x000fffffu;
            const double t = __hiloint2double((int)(mh | 0x40000000u), (int)((unsigned)lo << SL_K_LOG2)) - 3.;
            const double2 *const c = sl.c + row * (SL_ROW / 2);
            T0 = sl_poly(c, t);
            T1 = sl_poly(c + 4, t);
            T2 = sl_poly(c + 8, t);
        } else if (s_mc >= 1e-20) {
            sightline_T(cb, one_rs, d, Z, fabs(Z), tbl, ht, T0, T1, T2);
        } else {
            T0 = T1 = T2 = 1.;               // (out of bounds: the value is not used)
        }
    } else {
        sightline_T(cb, one_rs, d, Z, fabs(Z), tbl, ht, T0, T1, T2);
    }
    lin = mc_mix(pp, d, T0, T1, T2, EF, EA);
    epar = 0.;
    if (has_par) {
        const double dp = par - cb[MC_PAR];                     // pdf.py:166-173
        epar = -0.5 * (dp * dp * cb[MC_PIVAR]);
    }
    if (dust_on) epar += dust_lnp(g, d, a_mc);                  // fitting.py:1084-1085
    inb = s_mc >= 1e-20 && a_mc >= cb[MC_AV0] && a_mc <= cb[MC_AV1] && r_mc >= cb[MC_RV0] &&
          r_mc <= cb[MC_RV1];
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
    if (g.has_par || g.dust_on) v += epar - (g.has_par ? 0.5 * g.par_lnorm : 0.);
    if (!inb) v = -BIG;
    return v;
}

// rows (Philox calls = pairs of normals) of a k_post_mc staging slot: the 3 nmc normals of a
// record span at most 3 nmc / 2 + 1 calls
__host__ __device__ inline int mc_npair_max(int nmc) { return (3 * nmc) / 2 + 2; }

// P4: Monte Carlo prior integral of every kept record (fitting.py:1068-1105)
// and chi2min (fitting.py:2025-2034).  One lane per record.
//
// The 3 nmc normals of a record are one contiguous run of the stream, i.e.
// ~3 nmc / 2 Philox calls of two normals each.  A lane first walks its calls: 99.57 % of
// the normals are the ziggurat's rectangle case (a table look-up in LDS, one
// multiplication, one comparison); a call with a normal outside is noted in the lane's
// pending list (LDS, MC_PEND rows per lane) and redone by the slow path afterwards, so
// that the wedge / tail code runs max-over-lanes(pending) ~ 3 times per 64 records and
// not whenever one of 64 lanes needs it (every third call).  The two normals of a call
// go to the lane's column of `zs` as one 16-byte store (lane-interleaved rows of
// double2); afterwards the lane integrates, reading three normals per sample.
constexpr int MC_PEND = 8;
// > |z| of any normal of either stream: the ziggurat's tail returns R + xt with xt^2 < 2 yt <=
// 2 x 53 ln 2, i.e. |z| < 4.04 + 8.58 = 12.62; numpy's polar method sqrt(-2 ln r2) <= 12.01 at the
// smallest r2 = 2^-104 two 53-bit uniforms can form
constexpr double MC_ZMAX = 13.;

// Objects s0 .. s1 - 1 by falling number of kept records: the Monte Carlo kernels take the work
// items (object, chunk) in this order, largest first, so that the items left for the end of the
// launch are the short ones (items are 1 / PCH of an object: 0.5 - 3.5 ms each at the bench's
// posteriors against 18 ms for the launch -- taken in index order the last ones idled half the
// device for the length of an average item).
__global__ void k_post_order(int s0, int s1, const int64_t *__restrict__ nsel, int32_t *__restrict__ ord) {
    const int s = s0 + (int)threadIdx.x;
    if (s >= s1) return;
    const int64_t n = nsel[s];
    int rank = 0;
    for (int j = s0; j < s1; ++j) {
        const int64_t m = nsel[j];
        rank += (m > n || (m == n && j < s)) ? 1 : 0;
    }
    ord[rank] = s;
}

template <bool HT>
__global__ void __launch_bounds__(TILE, 3)
k_post_mc(PostParams pp, int64_t cap, int item_base, int nitem, unsigned int *__restrict__ counter,
          const double *__restrict__ zarr, const int64_t *__restrict__ zoff,
          double2 *__restrict__ zs, const int32_t *__restrict__ sel_idx, const int32_t *__restrict__ rec_slot,
          const double *__restrict__ sel_vals, const int64_t *__restrict__ sel_off,
          const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
          const uint64_t *__restrict__ nbase, const int32_t *__restrict__ flags,
          const StarGeom *__restrict__ geom, const double *__restrict__ feh,
          const double *__restrict__ loga, RecPost rp, double *__restrict__ part_max,
          double *__restrict__ part_chi2, const int32_t *__restrict__ ord) {
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    __shared__ unsigned int s_item;
    __shared__ double2 s_zp[ZIG_N];
    __shared__ unsigned short s_pend[MC_PEND][TILE];
    __shared__ double s_halo[HALO_TBL];
    __shared__ double2 s_sl[HT && SL_ON ? (SL_NI + 1) * (SL_ROW / 2) : 1];
    __shared__ SlCtl s_ctl;
    stage_exp_table(s_tbl);
    stage_zig_pairs(s_zp);
    if constexpr (HT) stage_halo_table(pp, s_halo);
    const double *const ht = HT ? s_halo : nullptr;
    double2 *const col = zs + (int64_t)blockIdx.x * mc_npair_max(pp.nmc) * TILE + threadIdx.x;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_item = (unsigned int)item_base + atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned int item = s_item;
        if (item >= (unsigned int)nitem) break;
        // (k-th item of the launch -> chunk k % PCH of the object with the (k / PCH)-th most records)
        const int s = __builtin_amdgcn_readfirstlane(ord[(item - (unsigned int)item_base) / PCH]),
                  c = (int)(item % PCH);
        int64_t a, b;
        rec_range_n(off2[s], nsel[s], c, a, b);
        const StarGeom g = geom[s];
        const CPtr cb0 = (CPtr)(uintptr_t)geom[s].mc;
        const bool one_rs = pp.Rs_thick2 == pp.Rs_thin2;
        // array source (numpy stream reproduced by k_mt_stream): the object's normals are
        // numbered from 0 in its own slice of zarr
        const double2 *const zsrc = zarr ? reinterpret_cast<const double2 *>(zarr + zoff[s]) : nullptr;
        const uint64_t nb = zarr ? 0ull : nbase[s];
        const uint64_t seed = star_seed(pp, s);
        double mx = -INFINITY, cmin = -INFINITY;   // cmin holds -min(chi2)
        SlTab sl{};
        if constexpr (HT && SL_ON)
            sl = sl_item(cb0, one_rs, s_tbl, ht, SL_NI, s_sl, &s_ctl, a, b, !flags[s], rec_slot, sel_off[s], rp,
                         sel_vals, cap);
        if (!flags[s]) {
            for (int64_t o0 = a; o0 < b; o0 += TILE) {
                const int64_t o = o0 + threadIdx.x;
                const bool live = o < b;
                const int64_t n = o - off2[s];
                // normals j_lo .. j_lo + 3 nmc - 1 of the stream = pairs p_lo .. p_hi;
                // pair q - p_lo goes to row q - p_lo of the slot as one 16-byte store
                const uint64_t j_lo = nb + (uint64_t)(3 * n * (int64_t)pp.nmc);
                const uint64_t p_lo = j_lo >> 1;
                // The record: its mean (s0, a0, r0) and Cholesky factor L.  Of its three runs of
                // normals, the integrand sees the second (z1) and the third (z2) only through the
                // bounds on Av and Rv (and z1 through the dust prior): a_mc = a0 + L1 z0 + L2 z1,
                // r_mc = r0 + L3 z0 + L4 z1 + L5 z2.  |z| < MC_ZMAX for every normal either stream can
                // produce, so a record whose mean is MC_ZMAX sum|L| away from the limits is in bounds whatever
                // the normals are: a wave whose records all are (an Rv pinned by its prior: every
                // record; NOT one pinned by rvlim = (x, x): there a sample is in bounds only where
                // L5 z2 rounds to nothing, as in the reference) skips the third run -- a third of the generator's work --, and the
                // second too where Av is as safe.  Same sums, bit for bit.
                int64_t i = 0, vs = 0;
                double L[6] = {0., 0., 0., 0., 0., 0.}, s0 = 0., a0 = 0., r0 = 0.;
                bool need1 = false, need2 = false;
                if (live) {
                    const int64_t r = sel_off[s] + rp.src[o];
                    i = sel_idx[r];
                    vs = rec_slot[r];
#pragma unroll
                    for (int k = 0; k < 6; ++k) L[k] = rp.chol[(int64_t)k * cap + o];
                    s0 = sel_vals[2 * cap + vs];
                    a0 = sel_vals[3 * cap + vs];
                    r0 = sel_vals[4 * cap + vs];
                    const double mr = MC_ZMAX * (fabs(L[3]) + fabs(L[4]) + fabs(L[5])),
                                 ma = MC_ZMAX * (fabs(L[1]) + fabs(L[2]));
                    // (NaN anywhere: not safe)
                    const bool rv_safe = r0 - mr >= pp.rvlim[0] && r0 + mr <= pp.rvlim[1];
                    const bool av_safe = a0 - ma >= pp.avlim[0] && a0 + ma <= pp.avlim[1];
                    need2 = !rv_safe;
                    need1 = need2 || !av_safe || g.dust_on;
#ifdef BRUTUS_MC_NO_RUN_SKIP                  // (A/B builds: all three runs, always)
                    need1 = need2 = true;
#endif
                }
                const bool w2 = __any(need2), w1 = w2 || __any(need1);      // wave-uniform
                const int nnorm = (w2 ? 3 : w1 ? 2 : 1) * pp.nmc;            // normals of the run in use
                if (zsrc) {
                    // copy the record's run of the stream into the lane's staging column
                    // (the last pair may reach one normal past the object's slice: the
                    // buffer is padded, the value is never used)
                    const uint64_t p_hi = (j_lo + (uint64_t)nnorm - 1) >> 1;
                    if (live)
                        for (uint64_t p = p_lo; p <= p_hi; ++p) col[(int64_t)(p - p_lo) * TILE] = zsrc[p];
                } else {
                    const int nrow = live ? (int)(((j_lo + (uint64_t)nnorm - 1) >> 1) - p_lo) + 1 : 0;
                    // both normals of row `row` by the full algorithm (slow path where needed)
                    auto redo = [&](int row) {
                        double z0, z1;
                        rng_normal_call(seed, p_lo + (uint64_t)row, kZigX, z0, z1);
                        col[(int64_t)row * TILE] = make_double2(z0, z1);
                    };
                    int npend = 0;
                    for (int row = 0; row < nrow; ++row) {
                        const Philox4 w4 = philox_at(seed, p_lo + (uint64_t)row, 0u, STREAM_NORMAL);
                        double x0, x1;
                        const bool f0 = zig_try_p(w4.w[0], w4.w[1], s_zp, x0),
                                   f1 = zig_try_p(w4.w[2], w4.w[3], s_zp, x1);
                        if (f0 && f1) {
                            col[(int64_t)row * TILE] = make_double2(zig_signed(x0, w4.w[1]), zig_signed(x1, w4.w[3]));
                        } else if (npend < MC_PEND) {
                            s_pend[npend++][threadIdx.x] = (unsigned short)row;
                        } else {
                            redo(row);
                        }
                    }
                    while (npend > 0) redo((int)s_pend[--npend][threadIdx.x]);
                }
                if (live) {
                    double Fc[3], Ac[3];
                    label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac, s_tbl);
                    // sum_t lin_t e^{epar_t} over the in-bounds samples, with a
                    // running maximum M of epar only (lin needs none); branch-free
                    double M = -INFINITY, acc = 0.;
                    int ninb = 0;
                    const double *const zc = (const double *)col;
                    const int jb = (int)(j_lo & 1);
                    // (the normals of sample t + 1 are requested while sample t is integrated)
                    auto zat = [&](int jj) { return zc[(int64_t)(jj >> 1) * (2 * TILE) + (jj & 1)]; };
                    // (a run that was not generated reads as zeros: in bounds, like the real ones)
                    double zn0 = zat(jb), zn1 = w1 ? zat(jb + pp.nmc) : 0., zn2 = w2 ? zat(jb + 2 * pp.nmc) : 0.;
                    for (int t = 0; t < pp.nmc; ++t) {
                        double d_, a_, r_, lin, epar;
                        bool inb;
                        // normal jj of the run: component jj & 1 of row jj >> 1
                        const double z0 = zn0, z1 = zn1, z2 = zn2;
                        {
                            const int tn = t + 1 < pp.nmc ? t + 1 : t;
                            const int j0 = jb + tn, j1 = j0 + pp.nmc, j2 = j1 + pp.nmc;
                            zn0 = zat(j0);
                            zn1 = w1 ? zat(j1) : 0.;
                            zn2 = w2 ? zat(j2) : 0.;
                        }
                        if constexpr (HT)
                            mc_sample_c<SL_ON>(mc_refresh(cb0), pp, g, g.has_par, g.dust_on, one_rs, z0, z1, z2,
                                              s0, a0, r0, L, Fc, Ac, s_tbl, ht, inb, lin, epar, sl);
                        else
                            mc_sample_lin(pp, g, z0, z1, z2, s0, a0, r0, L, Fc, Ac, s_tbl, d_, a_, r_, inb, lin,
                                          epar);
                        ninb += inb ? 1 : 0;
                        if (g.has_par || g.dust_on) {
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
                    double lse = pp.lnK + fast_log_r(acc);
                    if (g.has_par || g.dust_on) lse += M - (g.has_par ? 0.5 * g.par_lnorm : 0.);
                    double lnp = ninb > 0 ? rp.lnp[o] + (lse - fast_log_r((double)ninb)) : nan("");
                    if (!isfinite(lnp)) lnp = -BIG;                       // fitting.py:1103-1105
                    rp.lnp[o] = lnp;
                    if (lnp > mx) mx = lnp;
                    double chi2 = sel_vals[1 * cap + vs];
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

// P4 with the normals in memory (numpy's stream reproduced by mt_kernels.hpp), nmc <=
// MCA_NMC.  The 3 nmc normals of a record are one contiguous run of the object's slice and
// consecutive records follow each other, so a wave takes MCA_R records at a time: their
// MCA_R * 3 nmc normals come in as ONE coalesced copy into the wave's LDS tile, and the 64
// lanes are MCA_R records x MCA_G sample groups (group g integrates samples g, g + MCA_G,
// ...); the partial (max, sum, count) triples of a record's lanes are merged with three
// shuffle steps.  The lane-per-record form (k_post_mc with `zarr`) copies every run into
// a lane-interleaved staging column first: 16-byte loads at a 1.2 KB stride, then the same
// bytes written and read once more -- three times the HBM traffic of this kernel.
// (No sightline table here: with the tiles of normals there is room for three octaves of s beside
// three workgroups per CU -- no gain, most waves then run both forms -- and nine octaves at two
// workgroups per CU cost 20 %: profiles/r05_sightline_table_ab.txt.)
constexpr int MCA_R = 8, MCA_G = 8, MCA_NMC = 64;
constexpr int MCA_U = 8;          // loads of the tile copy a lane keeps in flight

template <bool HT>
__global__ void __launch_bounds__(TILE, 3)
k_post_mc_arr(PostParams pp, int64_t cap, int item_base, int nitem, unsigned int *__restrict__ counter,
              const double *__restrict__ zarr, const int64_t *__restrict__ zoff,
              const int32_t *__restrict__ sel_idx, const int32_t *__restrict__ rec_slot, const double *__restrict__ sel_vals,
              const int64_t *__restrict__ sel_off, const int64_t *__restrict__ off2,
              const int64_t *__restrict__ nsel, const int32_t *__restrict__ flags,
              const StarGeom *__restrict__ geom, const double *__restrict__ feh,
              const double *__restrict__ loga, RecPost rp, double *__restrict__ part_max,
              double *__restrict__ part_chi2, ZMap zm, const int32_t *__restrict__ ord) {
    static_assert(MCA_R * MCA_G == 64, "one wave = records x sample groups");
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    __shared__ unsigned int s_item;
    extern __shared__ double s_z[];       // (TILE / 64) tiles of MCA_R * 3 * nmc normals: as
                                          // little LDS as the call needs, so that the stream
                                          // walkers of the next batch fit beside this kernel
    __shared__ double s_halo[HALO_TBL];
    stage_exp_table(s_tbl);
    if constexpr (HT) stage_halo_table(pp, s_halo);
    const double *const ht = HT ? s_halo : nullptr;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rec8 = lane / MCA_G, grp = lane % MCA_G;
    const int run = 3 * pp.nmc;
    double *const zt = s_z + (size_t)w * MCA_R * run;
    for (;;) {
        __syncthreads();
        // counter == nullptr: one item per workgroup (grid = number of items).  Short
        // workgroups instead of persistent ones let the kernels of the next batch's
        // high-priority stream in between (BruteForce's two-phase pipeline).
        if (counter && threadIdx.x == 0) s_item = (unsigned int)item_base + atomicAdd(counter, 1u);
        __syncthreads();
        const unsigned int item = counter ? s_item : (unsigned int)item_base + blockIdx.x;
        if (item >= (unsigned int)nitem) break;
        const int s = __builtin_amdgcn_readfirstlane(ord[(item - (unsigned int)item_base) / PCH]),
                  c = (int)(item % PCH);
        int64_t a, b;
        rec_range_n(off2[s], nsel[s], c, a, b);
        const StarGeom g = geom[s];
        const CPtr cb0 = (CPtr)(uintptr_t)geom[s].mc;
        const bool one_rs = pp.Rs_thick2 == pp.Rs_thin2;
        // the object's normals, numbered from 0: a flat array, or (zm.zloc) the pairs where
        // pass 1 of the stream walk left them, through the object's segment list
        const bool mapped = zm.zloc != nullptr;
        const double *const zsrc = mapped ? nullptr : zarr + zoff[s];
        const int c_o = mapped ? zm.c[s] : 0;
        const int64_t slo = mapped ? zm.seg_lo[s] : 0;
        const int nsg = mapped ? zm.nseg[s] : 0;
        int sg = 0;                                     // this wave's segment cursor
        double mx = -INFINITY, cmin = -INFINITY;        // cmin holds -min(chi2)
        if (!flags[s]) {
            for (int64_t ob = a + (int64_t)w * 64; ob < b; ob += TILE) {
                for (int pass = 0; pass < 64 / MCA_R; ++pass) {
                    const int64_t o0 = ob + pass * MCA_R;
                    if (o0 >= b) break;                                  // wave-uniform
                    const int nrec = (int)(b - o0 < MCA_R ? b - o0 : MCA_R);
                    const int64_t j0 = (o0 - off2[s]) * run;
                    if (mapped) {
                        // (a tile of MCA_R records = 600 pairs spans at most two segments)
                        const int64_t pf = j0 > c_o ? (j0 - c_o) >> 1 : 0;
                        while (sg + 1 < nsg && zm.seg_pair0[slo + sg + 1] <= pf) ++sg;
                        const int64_t p0 = zm.seg_pair0[slo + sg], a0 = zm.seg_addr[slo + sg];
                        const bool two = sg + 1 < nsg;
                        const int64_t p1 = two ? zm.seg_pair0[slo + sg + 1] : INT64_MAX;
                        const int64_t a1 = two ? zm.seg_addr[slo + sg + 1] : 0;
                        const double *const zd = (const double *)zm.zloc;
                        // (MCA_U loads in flight per lane: one at a time, the copy of a tile
                        // was 19 dependent round trips to memory, 10 of the kernel's 27 ms)
                        for (int kb = lane; kb < nrec * run; kb += 64 * MCA_U) {
                            double v[MCA_U];
#pragma unroll
                            for (int u = 0; u < MCA_U; ++u) {
                                const int k = kb + 64 * u;
                                const int64_t jj = j0 + (k < nrec * run ? k : 0), pq = jj - c_o, pr = pq >> 1;
                                const int64_t ad = pr >= p1 ? a1 + (pr - p1) : a0 + (pr - p0);
                                v[u] = jj < c_o ? zm.cached[s] : zd[2 * ad + (pq & 1)];
                            }
#pragma unroll
                            for (int u = 0; u < MCA_U; ++u)
                                if (kb + 64 * u < nrec * run) zt[kb + 64 * u] = v[u];
                        }
                    } else {
                        for (int kb = lane; kb < nrec * run; kb += 64 * MCA_U) {
                            double v[MCA_U];
#pragma unroll
                            for (int u = 0; u < MCA_U; ++u) {
                                const int k = kb + 64 * u;
                                v[u] = zsrc[j0 + (k < nrec * run ? k : 0)];
                            }
#pragma unroll
                            for (int u = 0; u < MCA_U; ++u)
                                if (kb + 64 * u < nrec * run) zt[kb + 64 * u] = v[u];
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const bool live = rec8 < nrec;
                    const int64_t o = o0 + (live ? rec8 : 0);
                    const int64_t r = sel_off[s] + rp.src[o];
                    const int64_t i = sel_idx[r], vs = rec_slot[r];
                    double Fc[3], Ac[3], L[6];
                    label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac, s_tbl);
#pragma unroll
                    for (int k = 0; k < 6; ++k) L[k] = rp.chol[(int64_t)k * cap + o];
                    const double s0 = sel_vals[2 * cap + vs], a0 = sel_vals[3 * cap + vs],
                                 r0 = sel_vals[4 * cap + vs];
                    double M = -INFINITY, acc = 0.;
                    int ninb = 0;
                    const double *const zr = zt + (live ? rec8 : 0) * run;
                    if (live) {
                        for (int t = grp; t < pp.nmc; t += MCA_G) {
                            double d_, a_, r_, lin, epar;
                            bool inb;
                            if constexpr (HT)
                                mc_sample_c(mc_refresh(cb0), pp, g, g.has_par, g.dust_on, one_rs, zr[t],
                                            zr[pp.nmc + t], zr[2 * pp.nmc + t], s0, a0, r0, L, Fc, Ac, s_tbl,
                                            ht, inb, lin, epar);
                            else
                                mc_sample_lin(pp, g, zr[t], zr[pp.nmc + t], zr[2 * pp.nmc + t], s0, a0, r0, L,
                                              Fc, Ac, s_tbl, d_, a_, r_, inb, lin, epar);
                            ninb += inb ? 1 : 0;
                            if (g.has_par || g.dust_on) {
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
                    }
                    // merge the MCA_G partial triples of a record (lanes differing in the low bits)
#pragma unroll
                    for (int off = 1; off < MCA_G; off <<= 1) {
                        const double Mo = __shfl_xor(M, off, 64), acco = __shfl_xor(acc, off, 64);
                        ninb += __shfl_xor(ninb, off, 64);
                        if (g.has_par || g.dust_on) {
                            if (Mo > -INFINITY) {
                                if (M > -INFINITY) {
                                    const double ex = fast_exp_bf(-fabs(Mo - M), s_tbl);
                                    acc = M >= Mo ? fma(acco, ex, acc) : fma(acc, ex, acco);
                                    M = M >= Mo ? M : Mo;
                                } else {
                                    acc = acco;
                                    M = Mo;
                                }
                            }
                        } else {
                            acc += acco;
                        }
                    }
                    if (live && grp == 0) {
                        // logsumexp(lnp_mc) - ln(#in bounds), fitting.py:1094-1102; with no
                        // sample in bounds the reference yields +inf -> not finite -> -BIG
                        double lse = pp.lnK + fast_log_r(acc);
                        if (g.has_par || g.dust_on) lse += M - (g.has_par ? 0.5 * g.par_lnorm : 0.);
                        double lnp = ninb > 0 ? rp.lnp[o] + (lse - fast_log_r((double)ninb)) : nan("");
                        if (!isfinite(lnp)) lnp = -BIG;                       // fitting.py:1103-1105
                        rp.lnp[o] = lnp;
                        if (lnp > mx) mx = lnp;
                        double chi2 = sel_vals[1 * cap + vs];
                        if (g.has_par) {
                            const double dp = sqrt(s0) - g.par;
                            chi2 += dp * dp * g.par_ivar;
                        }
                        if (-chi2 > cmin) cmin = -chi2;
                    }
                    // the next pass overwrites the tile
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        block_max_store(mx, slot, part_max + (int64_t)s * PCH + c);
        block_max_store(cmin, slot, part_chi2 + (int64_t)s * PCH + c);
        if (!counter) break;
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
k_post_evid_part(int s0, const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
                 const int32_t *__restrict__ flags, const double *__restrict__ part_max,
                 const double *__restrict__ part_chi2, RecPost rp, double *__restrict__ part_e) {
    __shared__ double sh[TILE];
    const int s = s0 + blockIdx.y, c = blockIdx.x;
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
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
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
            const double v = (int)threadIdx.x >= st ? sh[threadIdx.x - st] : 0.;
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
k_post_wt_part(int s0, const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
               const int32_t *__restrict__ flags, const double *__restrict__ part_max,
               const double *__restrict__ part_chi2, const double *__restrict__ part_e, RecPost rp,
               double *__restrict__ part_w) {
    __shared__ double sh[TILE];
    const int s = s0 + blockIdx.y, c = blockIdx.x;
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
k_post_cdf(int s0, const int64_t *__restrict__ off2, const int64_t *__restrict__ nsel,
           const int32_t *__restrict__ flags, const double *__restrict__ part_max,
           const double *__restrict__ part_chi2, const double *__restrict__ part_e,
           const double *__restrict__ part_w, RecPost rp, double *__restrict__ cdf,
           double *__restrict__ star_out) {
    __shared__ double sh[TILE];
    const int s = s0 + blockIdx.y, c = blockIdx.x;
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
k_post_draw(PostParams pp, int sbase, const double *__restrict__ zarr, const int64_t *__restrict__ zoff,
            const double *__restrict__ uarr, int64_t cap, const int32_t *__restrict__ sel_idx, const int32_t *__restrict__ rec_slot,
            const double *__restrict__ sel_vals, const int64_t *__restrict__ sel_off,
            const int64_t *__restrict__ off2, const int64_t *__restrict__ nselv,
            const uint64_t *__restrict__ nbase, const int32_t *__restrict__ flags,
            const StarGeom *__restrict__ geom, const double *__restrict__ feh,
            const double *__restrict__ loga, RecPost rp, const double *__restrict__ cdf,
            const double *__restrict__ star_out, int32_t *__restrict__ out_idx,
            double *__restrict__ out_vals, ZMap zm) {
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    const int s = sbase + blockIdx.y;
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= pp.ndraws || flags[s]) return;
    const int nuni = pp.ndraws * (pp.return_distreds ? 2 : 1);
    const int64_t a = off2[s], nsel = nselv[s];
    if (nsel <= 0) return;
    const StarGeom g = geom[s];
    const uint64_t ub = star_ubase(pp, s);
    const uint64_t seed = star_seed(pp, s);
    // choice(Nsel, p=wt): searchsorted(cdf / cdf[-1], u, side='right')
    const double total = star_out[4 * s + 2];
    const double u = uarr ? uarr[(int64_t)s * nuni + q] : rng_uniform(seed, ub + (uint64_t)q);
    int64_t lo = 0, hi = nsel;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[a + mid] / total <= u) lo = mid + 1; else hi = mid;
    }
    if (lo >= nsel) lo = nsel - 1;
    const int64_t o = a + lo;
    const int64_t r = sel_off[s] + rp.src[o];
    const int64_t i = sel_idx[r], vs = rec_slot[r];
    out_idx[(int64_t)s * pp.ndraws + q] = (int32_t)i;
    double *ov = out_vals + ((int64_t)s * pp.ndraws + q) * POST_NOUT;
    const double s0 = sel_vals[2 * cap + vs], a0 = sel_vals[3 * cap + vs], r0 = sel_vals[4 * cap + vs];
    ov[0] = s0;
    ov[1] = a0;
    ov[2] = r0;
    double C[6];
    rec_cov(sel_vals, cap, vs, C);
    ov[3] = C[0]; ov[4] = C[1]; ov[5] = C[2];
    ov[6] = C[1]; ov[7] = C[3]; ov[8] = C[4];
    ov[9] = C[2]; ov[10] = C[4]; ov[11] = C[5];
    ov[12] = rp.lnp[o];
    if (!pp.return_distreds) return;
    // second stage (fitting.py:2049-2057): pick one of the record's nmc samples
    double Fc[3], Ac[3], L[6];
    label_terms(pp, pp.has_feh ? feh[i] : 0., pp.has_loga ? loga[i] : 0., Fc, Ac, s_tbl);
#pragma unroll
    for (int k = 0; k < 6; ++k) L[k] = rp.chol[(int64_t)k * cap + o];
    const uint64_t nb = zarr ? 0ull : nbase[s];
    const double *zo = zarr ? zarr + zoff[s] : nullptr;
    double m = -INFINITY;
    bool inb_;
    NormalReader rd[3];
    const ZMap *zmp = zm.zloc ? &zm : nullptr;      // (zarr is then only a flag: nb = 0)
    if (zmp) zo = nullptr;
    rd[0].init(seed, zo, zmp, s); rd[1].init(seed, zo, zmp, s); rd[2].init(seed, zo, zmp, s);
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
    const double u2 = uarr ? uarr[(int64_t)s * nuni + pp.ndraws + q]
                           : rng_uniform(seed, ub + (uint64_t)pp.ndraws + (uint64_t)q);
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

// Nsel_max clip: the kept records of one object, best first.  k_clip_gather pulls lnp, the
// six Cholesky planes and the source index through `perm` into a dense scratch (7 planes of
// `keep` doubles + `keep` ints), k_clip_store writes them back in place.
__global__ void k_clip_gather(RecPost rp, int64_t cap, int64_t a, const int32_t *__restrict__ perm,
                              int64_t keep, double *__restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= keep) return;
    const int64_t j = a + perm[i];
    tmp[i] = rp.lnp[j];
#pragma unroll
    for (int q = 0; q < 6; ++q) tmp[(1 + q) * keep + i] = rp.chol[(int64_t)q * cap + j];
    reinterpret_cast<int32_t *>(tmp + 7 * keep)[i] = rp.src[j];
}
__global__ void k_clip_store(RecPost rp, int64_t cap, int64_t a, int64_t keep,
                             const double *__restrict__ tmp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= keep) return;
    const int64_t j = a + i;
    rp.lnp[j] = tmp[i];
#pragma unroll
    for (int q = 0; q < 6; ++q) rp.chol[(int64_t)q * cap + j] = tmp[(1 + q) * keep + i];
    rp.src[j] = reinterpret_cast<const int32_t *>(tmp + 7 * keep)[i];
}

// direction of the sightline (l, b) [rad] in the Galactocentric frame of `pp`
__device__ __forceinline__ void sightline(const PostParams &pp, double l, double b, StarGeom &g) {
    const double n0 = cos(b) * cos(l), n1 = cos(b) * sin(l), n2 = sin(b);
    g.ux = pp.frame_mat[0] * n0 + pp.frame_mat[1] * n1 + pp.frame_mat[2] * n2;
    g.uy = pp.frame_mat[3] * n0 + pp.frame_mat[4] * n1 + pp.frame_mat[5] * n2;
    g.uz = pp.frame_mat[6] * n0 + pp.frame_mat[7] * n1 + pp.frame_mat[8] * n2;
}

// per-object geometry / parallax constants
__global__ void k_post_geom(PostParams pp, int nstar, const double *__restrict__ coords,
                            const double *__restrict__ par, const double *__restrict__ perr,
                            DustCtx dc, StarGeom *__restrict__ geom) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nstar) return;
    const double l = coords[2 * s] * (M_PI / 180.), b = coords[2 * s + 1] * (M_PI / 180.);
    StarGeom g;
    sightline(pp, l, b, g);
    const double p = par ? par[s] : nan(""), pe = perr ? perr[s] : nan("");
    g.has_par = (isfinite(p) && isfinite(pe)) ? 1 : 0;
    g.par = g.has_par ? p : 0.;
    g.par_ivar = g.has_par ? 1. / (pe * pe) : 0.;
    g.par_lnorm = g.has_par ? log(2. * M_PI * pe * pe) : 0.;
    g.dust_on = (dc.d_los && dc.d_ok[s]) ? 1 : 0;
    g.nd = dc.nd;
    g.los = dc.d_los ? dc.d_los + (int64_t)s * 3 * dc.nd : nullptr;
    g.d_off = dc.offset;
    g.d_scale = dc.scale;
    g.d_smooth = dc.smooth;
    g.d_scat2 = dc.scatter * dc.scatter;
    const double o0 = pp.frame_off[0], o1 = pp.frame_off[1];
    for (int k = 0; k < 32; ++k) g.mc[k] = 0.;
    g.mc[MC_A2] = g.ux * g.ux + g.uy * g.uy;
    g.mc[MC_A1] = 2. * (g.ux * o0 + g.uy * o1);
    g.mc[MC_A0] = o0 * o0 + o1 * o1;
    g.mc[MC_UZ] = g.uz;
    g.mc[MC_O2] = pp.frame_off[2];
    g.mc[MC_RS_THIN2] = pp.Rs_thin2;
    g.mc[MC_RS_THICK2] = pp.Rs_thick2;
    g.mc[MC_C0T] = pp.c0_thin + pp.abs_Z_solar * pp.inv_Z_thin;
    g.mc[MC_IRT] = pp.inv_R_thin;
    g.mc[MC_IZT] = pp.inv_Z_thin;
    g.mc[MC_C0K] = pp.c0_thick + pp.abs_Z_solar * pp.inv_Z_thick;
    g.mc[MC_IRK] = pp.inv_R_thick;
    g.mc[MC_IZK] = pp.inv_Z_thick;
    g.mc[MC_QINF] = pp.q_halo_inf;
    g.mc[MC_DQE] = (pp.q_halo_inf - pp.q_halo_ctr) * 2.71828182845904523536;
    g.mc[MC_RQ2] = pp.rq2;
    g.mc[MC_IRQ] = pp.inv_r_q;
    g.mc[MC_RS_HALO2] = pp.Rs_halo2;
    for (int k = 0; k < 7; ++k) g.mc[MC_B1 + k] = pp.halo_b[k];
    g.mc[MC_AV0] = pp.avlim[0];
    g.mc[MC_AV1] = pp.avlim[1];
    g.mc[MC_RV0] = pp.rvlim[0];
    g.mc[MC_RV1] = pp.rvlim[1];
    g.mc[MC_PAR] = g.par;
    g.mc[MC_PIVAR] = g.par_ivar;
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
    sightline(pp, l, b, g);
    g.has_par = 0;
    g.dust_on = 0;
    double Fc[3], Ac[3];
    label_terms(pp, feh[i], loga[i], Fc, Ac);
    out[i] = gal_lnprior_dev(pp, g, dist[i], Fc, Ac, kExp2Tbl);
}

// the same ln prior as the sample loop of k_post_mc / k_post_mc_arr evaluates it: the object's
// constant block (geom[0], from k_post_geom) through scalar loads and the halo table when the
// parameters admit it, the plain form otherwise -- a sample with z = 0 at s0 = 1 / d^2
template <bool HT>
__global__ void k_debug_galprior_mc(PostParams pp, int n, const double *__restrict__ dist,
                                    const StarGeom *__restrict__ geom, const double *__restrict__ feh,
                                    const double *__restrict__ loga, double *__restrict__ out) {
    __shared__ double s_tbl[64];
    __shared__ double s_halo[HALO_TBL];
    stage_exp_table(s_tbl);
    if constexpr (HT) stage_halo_table(pp, s_halo);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const StarGeom g = geom[0];
    double Fc[3], Ac[3];
    label_terms(pp, feh[i], loga[i], Fc, Ac, s_tbl);
    const double L[6] = {0., 0., 0., 0., 0., 0.};
    const double s0 = 1. / (dist[i] * dist[i]), a0 = 0.5 * (pp.avlim[0] + pp.avlim[1]),
                 r0 = 0.5 * (pp.rvlim[0] + pp.rvlim[1]);
    double lin, epar, d_, a_, r_;
    bool inb;
    if constexpr (HT)
        mc_sample_c(mc_refresh((CPtr)(uintptr_t)geom[0].mc), pp, g, false, false,
                    pp.Rs_thick2 == pp.Rs_thin2, 0., 0., 0., s0, a0, r0, L, Fc, Ac, s_tbl, s_halo, inb, lin,
                    epar);
    else
        mc_sample_lin(pp, g, 0., 0., 0., s0, a0, r0, L, Fc, Ac, s_tbl, d_, a_, r_, inb, lin, epar);
    out[i] = inb ? pp.lnK + fast_log_r(lin) : nan("");
}

// ... and through the sightline table of the Monte Carlo kernels: every workgroup builds the table
// for the window of ITS 256 distances (sl_window of the largest s among them) and evaluates them
// like samples; used[i] = 1 where the table was used, 0 where the sample fell outside the window
// and took the closed form
__global__ void __launch_bounds__(TILE)
k_debug_galprior_sl(PostParams pp, int n, const double *__restrict__ dist, const StarGeom *__restrict__ geom,
                    const double *__restrict__ feh, const double *__restrict__ loga,
                    double *__restrict__ out, int32_t *__restrict__ used) {
    __shared__ double s_tbl[64];
    __shared__ double s_halo[HALO_TBL];
    __shared__ double2 s_sl[(SL_NI + 1) * (SL_ROW / 2)];
    __shared__ double s_top[4];
    __shared__ int s_kink, s_lo;
    stage_exp_table(s_tbl);
    stage_halo_table(pp, s_halo);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const StarGeom g = geom[0];
    const CPtr cb = (CPtr)(uintptr_t)geom[0].mc;
    const bool one_rs = pp.Rs_thick2 == pp.Rs_thin2;
    const double s0 = i < n ? 1. / (dist[i] * dist[i]) : 0.;
    const double top = wave_max(s0);
    if ((threadIdx.x & 63) == 0) s_top[threadIdx.x >> 6] = top;
    if (threadIdx.x == 0) s_kink = -1, s_lo = 0;
    __syncthreads();
    SlTab sl;
    sl.ni = SL_NI;
    sl.base = __builtin_amdgcn_readfirstlane(sl_window(fmax(fmax(s_top[0], s_top[1]), fmax(s_top[2], s_top[3])), SL_NI));
    sl_build(cb, one_rs, s_tbl, s_halo, sl.base, SL_NI, reinterpret_cast<double *>(s_sl), &s_kink, &s_lo);
    __syncthreads();
    sl.c = s_sl;
    sl.kink = __builtin_amdgcn_readfirstlane(s_kink);
    sl.lo = __builtin_amdgcn_readfirstlane(s_lo);
    if (i >= n) return;
    double Fc[3], Ac[3];
    label_terms(pp, feh[i], loga[i], Fc, Ac, s_tbl);
    const double L[6] = {0., 0., 0., 0., 0., 0.};
    const double a0 = 0.5 * (pp.avlim[0] + pp.avlim[1]), r0 = 0.5 * (pp.rvlim[0] + pp.rvlim[1]);
    double lin, epar;
    bool inb, tab = false;
    mc_sample_c<true>(mc_refresh(cb), pp, g, false, false, one_rs, 0., 0., 0., s0, a0, r0, L, Fc, Ac, s_tbl,
                      s_halo, inb, lin, epar, sl, &tab);
    out[i] = inb ? pp.lnK + fast_log_r(lin) : nan("");
    used[i] = tab ? 1 : 0;
}

}  // namespace
