// fit2_kernels.hpp -- the hot path behind brutus_fit_batch: float32 proof pass, exact
// thresholds, selection.  Part of the single translation unit brutus_kernels.hip (included
// after fit_kernels.hpp); everything lives in that unit's anonymous namespace.
//
// Why.  The reference's per-star control flow hangs on maxima over the whole grid
// (fitting.py:246-264 K1, :758-759 cull, :798-799 K2, :988-991 first cut), but
// only the models within a few units of those maxima take part in any decision or
// reach the output; for everything else it is enough to PROVE that it is below
// the thresholds.  So:
//
//   k_pre32    every (star, model) in float32 (centred magnitudes, fluxes scaled to
//              O(1)): approximate cull statistic lnl_p~ and first-cut statistic
//              lnprob~ (4 + 4 B per pair), float32 maxima per 2048-model block, and the
//              sweep statistics that decide K1 when they are clear of the tolerances
//              (else k_k1probe, exact).
//   k_top      float64 re-evaluation of the models within 2 eps of the float32 maximum
//              (only the blocks whose float32 maximum is that high are touched)
//              ->  EXACT max lnl_p, i.e. the exact cull threshold.
//   k_cmp_count32 + k_offsets + k_items + k_cmp_scatter
//              ordered list of the models with lnl_p~ >= threshold - eps, cut into work
//              items by (star, model chunk) segment and numbered chunk-major.
//   k_fflux    (fit_kernels.hpp) exact cull test in float64, flux iterations for the
//              survivors; their results are written ONCE, at the candidate's list position,
//              straight into the caller's record planes (full-line writes) -- that is the
//              survivors' final storage; the list position is also left as a tag in the
//              lnprob~ plane (surv_tag).
//   k_top (B)  exact maximum of lnprob over the non-survivors that could exceed the
//              survivors' maximum  ->  EXACT first-cut threshold.
//   k_sel_classify + k_sel_band
//              two bit-masks: selected, and selected-but-not-a-survivor ("derived").
//              float32 decides when |lnprob~ - thr| > eps, the models inside the band are
//              queued and re-evaluated in float64 with dense lanes; survivors by their
//              final value.
//   k_offsets + k_items + k_rec_index
//              ordered record index from the bit-masks alone: (model, slot of its values)
//              per selected model in np.where order; derived models get the slots behind
//              the candidates' and form the work list of ...
//   k_derive   (fit_kernels.hpp) ... K1 sweeps + full MLE for the selected models the cull
//              dropped, dense lanes, full-line writes.
//
// float32 never produces an output value or a decision: a lane whose float32 value
// is NaN or inside the error band is re-evaluated in float64.  `eps` is a per-star
// bound on |float32 - float64| (Star32::eps, checked at run time with BRUTUS_AUDIT=1
// and in tests/test_gpu_fit2.py).
#pragma once

#include "pre32_types.hpp"

namespace {

// pre32_types.hpp names the tile geometry on its own (it is also compiled into pre32s_unit.hip,
// which does not see common.hpp): one definition in effect
static_assert(PS_TILE == TILE && PS_NBMAX == NBMAX, "pre32_types.hpp and common.hpp disagree");

// ---------------------------------------------------------------------------
// per-star float32 companion of StarPrep
// ---------------------------------------------------------------------------
// `init` (brutus_fit_batch): the call's small device state starts here too -- the identity star
// list, two sweeps / two flux iterations per star, every counter zero -- instead of as two
// uploads, a fill kernel and a memset in front of it.
struct CallInit {
    int32_t *ids_all, *kfix, *k2, *counters;
    int ncounters;
};
__global__ void __launch_bounds__(64)
k_prep32(int nstar, const StarPrep *__restrict__ stars, float eps_scale, int dim_prior,
         Star32 *__restrict__ out, CallInit init) {
    // one wave per star, lane = band (like k_prep); the band sums are added up in band order
    static_assert(NBMAX <= 64, "one lane per band");
    __shared__ double s_a[NBMAX], s_b[NBMAX], s_c[NBMAX];
    __shared__ double s_gbar, s_D;
    __shared__ int s_bad;
    const int s = blockIdx.x, j = threadIdx.x;
    if (s >= nstar) return;
    if (init.ids_all && j == 0) {
        init.ids_all[s] = s;
        init.kfix[s] = 2;
        init.k2[s] = 2;
    }
    if (init.counters && s == 0 && j < init.ncounters) init.counters[j] = 0;
    const StarPrep &sp = stars[s];
    Star32 &o = out[s];
    // bands with a non-positive flux carry mags_var = 1e50: no weight
    double w = 0.;
    if (j < NBMAX) {
        w = sp.iW[j] > 1e-40 ? sp.iW[j] : 0.;
        s_a[j] = w;
        s_b[j] = w * sp.g[j];
    }
    if (j == 0) s_bad = 0;
    __syncthreads();
    if (j == 0) {
        double sw = 0., swg = 0.;
        for (int k = 0; k < NBMAX; ++k) {
            sw += s_a[k];
            swg += s_b[k];
        }
        const double gbar = sw > 0. ? swg / sw : 0.;
        const double D = exp10(-0.4 * gbar);
        s_gbar = gbar;
        s_D = D;
        if (!(sw > 0. && isfinite(gbar) && isfinite(D) && D > 1e-30 && D < 1e30)) s_bad = 1;
    }
    __syncthreads();
    const double gbar = s_gbar, D = s_D;
    if (j < NBMAX) {
        const double dd = sp.d[j] / D, iv = sp.iV[j] * D * D;
        o.gc[j] = (float)(w > 0. ? sp.g[j] - gbar : 0.);
        o.w[j] = (float)w;
        o.dd[j] = (float)dd;
        o.iv[j] = (float)iv;
        s_a[j] = dd * dd * iv;
        s_b[j] = w;
        s_c[j] = fabs(dd) * sqrt(iv);
        if (!(fabs(dd) < 1e15) || !(iv < 1e30) || !(w < 1e30)) atomicOr(&s_bad, 1);
    }
    __syncthreads();
    if (j != 0) return;
    double DD2 = 0., wy = 0., snmax = 0.;
    for (int k = 0; k < NBMAX; ++k) {
        DD2 += s_a[k];
        wy += s_b[k];
        snmax = s_c[k] > snmax ? s_c[k] : snmax;
    }
    o.S = (float)sp.S;
    o.DD2 = (float)DD2;
    o.gbar = (float)gbar;
    o.par = (float)sp.par;
    o.par_ivar = (float)sp.par_ivar;
    o.sp_mean = (float)sp.sp_mean;
    o.sp_var = (float)sp.sp_var;
    o.c0 = (float)sp.c0;
    o.c1 = (float)sp.c1;
    o.has_par = sp.has_par;
    o.sp_on = sp.sp_on;
    // |f32 - f64|: the single-pass chi2 cancels against DD2 = sum (S/N)^2, the
    // magnitude-space chi2 against sum w y^2 (y up to a few mag for the models that
    // matter); both scale with 2^-23 times those sums.  The parallax terms add
    // 2^-23 times the parallax S/N.  eps_scale (default 1) multiplies the lot.
    const double u = 1.1920929e-07;
    const double psn = sp.has_par ? fabs(sp.par) * sqrt(sp.par_ivar) + 4. : 0.;
    o.eps = (float)(eps_scale * (0.02 + 16. * u * (DD2 + 64. * snmax) + 64. * u * psn * psn));
    o.epsw = (float)(eps_scale * (0.02 + 64. * u * wy));
    // ln(chi2) of the dimensionality prior amplifies the chi2 error by c1 / chi2
    o.chi2_lo = fmaxf(4.f * o.eps, dim_prior ? 0.5f * fabsf(o.c1) + 0.5f : 0.f);
    o.ok = s_bad ? 0 : 1;
}

// ---------------------------------------------------------------------------
// k_pre32: the whole grid in float32
// ---------------------------------------------------------------------------
// part[(bx * nstar + s) * NV32 + v]:
//   v = 0, 1, 2 : sweep 1: L = max logwt, max{logwt : step >= mtol + slack},
//                 max{logwt : step >= mtol - slack}
//   v = 3, 4, 5 : the same for sweep 2 (general kernels only)
//   v = 6, 7    : max lnl_p~, max lnprob~ (state after kfix[s] sweeps)
//   v = 8       : > 0 if a live lane produced a non-finite logwt
//   v = 9       : > 0 if a live lane's lnl_p~ or lnprob~ is NaN (= "ask float64")
// logwt is the reference's sweep statistic -chi2/2 with the distance-modulus offset
// LEFT IN the residuals (fitting.py:240-243).  With centred
// magnitudes y_c = y - Delta (Delta = gbar - mbar) and o = sum w (y_c - Av R) / S,
//   chi2 = [chi2_c - S o^2] + S (Delta + o)^2 :
// the bracket is the offset-free chi2, and Delta + o is small exactly for the models
// that can be inside the ln(init_thresh) window, so float32 resolves it there.
// General kernels: band pairs (j, j + 1) through packed float32 arithmetic (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32: two results per issue slot): the model's values of two adjacent bands are
// an adjacent register pair, the star's constants of two adjacent bands an adjacent SGPR
// pair as they lie in Star32 -- no shuffling.  Every band sum becomes an (even, odd) pair of
// partial sums, added at the end.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 ld2(const float *p) { return f32x2{p[0], p[1]}; }
__device__ __forceinline__ float hsum(f32x2 v) { return v.x + v.y; }

template <int NB, bool RVF, int G>
__global__ void __launch_bounds__(TILE)
k_pre32(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
        const int32_t *__restrict__ star_ids, const Star32 *__restrict__ stars, P32 p,
        const int32_t *__restrict__ kfix, int ntile, float *__restrict__ lnlp32,
        float *__restrict__ lnpr32, float *__restrict__ part,
        const int32_t *__restrict__ nrun_dev) {
    __shared__ float slot[4][G * NV32];
    // (re-run over a star list that was put together on the device: its length lives there
    // too; the launch is sized for the whole batch and the surplus workgroups leave here)
    if (nrun_dev) nrun = *nrun_dev;
    if (nrun <= 0) return;
    const float C10 = -1.32877123795494494f;     // -0.4 log2(10)
    const float NINF = -INFINITY;
    // 1-D launch of 8 * ceil(nchunk / 8) * ngroup workgroups.  Workgroup L runs on XCD L % 8
    // (round-robin dispatch) and takes, in turn, the star groups of the tile chunks
    // L % 8, L % 8 + 8, ...: a tile chunk is fetched from the fabric ONCE, into one L2, and
    // serves all nrun / G star groups from there (108 MB per call instead of 108 MB per group)
    const int ngroup = (nrun + G - 1) / G;
    const int bx = ((int)(blockIdx.x >> 3) / ngroup) * 8 + (int)(blockIdx.x & 7);     // tile chunk
    if (bx * F2_T >= ntile) return;
    const int g0 = ((int)(blockIdx.x >> 3) % ngroup) * G;
    const int ng = min(G, nrun - g0);
    float mx[G][NV32];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int v = 0; v < NV32; ++v) mx[g][v] = NINF;
    const int t0 = bx * F2_T;
    const int t1 = min(ntile, t0 + F2_T);
    const float inv_nf = 1.f / (float)p.nfilt;
    for (int t = t0; t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        const bool live = i < nmodel;
        float mc[NB], R[NB], dr[RVF ? 1 : NB];
        float mbar = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float *q = grid + (int64_t)(3 * j) * nmodel_pad + i;
            mc[j] = q[0];
            R[j] = q[nmodel_pad];
            const float d = q[2 * nmodel_pad];
            if constexpr (RVF) R[j] = fmaf(p.rv_mean, d, R[j]);
            else dr[j] = d;
            if (j < p.nfilt) mbar += mc[j];
        }
        mbar *= inv_nf;
#pragma unroll
        for (int j = 0; j < NB; ++j) mc[j] -= mbar;
        // per-tile products the star loop would otherwise redo for every star (pinned Rv):
        // R^2 for sum w R^2, and the centred magnitudes pre-multiplied by -0.4 log2(10)
        float R2[RVF ? NB : 1], mcC[RVF ? NB : 1];
        if constexpr (RVF) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                R2[j] = R[j] * R[j];
                mcC[j] = C10 * mc[j];
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g >= ng) break;
            const int s = star_ids[g0 + g];
            const Star32 &sp = stars[s];
            const int K = kfix[s];
            float av = p.av_mean, rv = p.rv_mean;
            if constexpr (RVF) {
                // (scalar here: the packed form of this kernel needs 151 VGPRs instead of 98
                // and loses the fourth wave per SIMD, 0.73 -> 0.78 ms)
                float uR = 0.f, RR = 0.f, yR = 0.f, uy = 0.f, yy = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float w = sp.w[j];
                    const float y = sp.gc[j] - mc[j];
                    const float yw = y * w;
                    uR = fmaf(R[j], w, uR);
                    RR = fmaf(R2[j], w, RR);
                    yR = fmaf(R[j], yw, yR);
                    uy += yw;
                    yy = fmaf(y, yw, yy);
                }
                const float rs = uy - av * uR;
                const float ra = (yR - av * RR) + (p.av_mean - av) * p.av_ivar;
                const float a_den = RR + p.av_ivar;
                float dav = (sp.S * ra - uR * rs) * __builtin_amdgcn_rcpf(sp.S * a_den - uR * uR);
                dav = fmaxf(dav, p.avmin - av);
                dav = fminf(dav, p.avmax - av);
                av += dav;
                const float oc = (uy - av * uR) * __builtin_amdgcn_rcpf(sp.S);
                const float tt0 = (sp.gbar - mbar) + oc;
                const float lw = -0.5f * ((yy - av * (2.f * yR - av * RR)) + sp.S * (tt0 * tt0 - oc * oc));
                if (live) {
                    if (lw == lw) {
                        mx[g][0] = fmaxf(mx[g][0], lw);
                        const float st = fabsf(dav);
                        if (st >= p.mtol_hi) mx[g][1] = fmaxf(mx[g][1], lw);
                        if (st >= p.mtol_lo) mx[g][2] = fmaxf(mx[g][2], lw);
                    } else {
                        mx[g][8] = 1.f;
                    }
                }
            } else {
                f32x2 ua2 = 0.f, ub2 = 0.f, uy2 = 0.f, aa2 = 0.f, ab2 = 0.f, bb2 = 0.f, ay2 = 0.f,
                      by2 = 0.f, yy2 = 0.f;
#pragma unroll
                for (int j = 0; j < NB; j += 2) {
                    const f32x2 w = ld2(sp.w + j);
                    const f32x2 a = ld2(R + j), b = ld2(dr + j);
                    const f32x2 y = ld2(sp.gc + j) - ld2(mc + j);
                    const f32x2 aw = a * w, bw = b * w, yw = y * w;
                    ua2 += aw;
                    ub2 += bw;
                    uy2 += yw;
                    aa2 = pk_fma(a, aw, aa2);
                    ab2 = pk_fma(a, bw, ab2);
                    bb2 = pk_fma(b, bw, bb2);
                    ay2 = pk_fma(a, yw, ay2);
                    by2 = pk_fma(b, yw, by2);
                    yy2 = pk_fma(y, yw, yy2);
                }
                const float ua = hsum(ua2), ub = hsum(ub2), uy = hsum(uy2), aa = hsum(aa2),
                            ab = hsum(ab2), bb = hsum(bb2), ay = hsum(ay2), by = hsum(by2),
                            yy = hsum(yy2);
                auto sweep = [&](float &dav_o, float &drv_o) -> float {
                    const float uR = ua + rv * ub;
                    const float RR = aa + rv * (2.f * ab + rv * bb);
                    const float yR = ay + rv * by;
                    float rs = uy - av * uR;
                    const float ra = (yR - av * RR) + (p.av_mean - av) * p.av_ivar;
                    const float a_den = RR + p.av_ivar;
                    float dav = (sp.S * ra - uR * rs) * __builtin_amdgcn_rcpf(sp.S * a_den - uR * uR);
                    dav = fmaxf(dav, p.avmin - av);
                    dav = fminf(dav, p.avmax - av);
                    av += dav;
                    const float r_den = bb * av * av + p.rv_ivar;
                    const float sr = ub * av;
                    rs = uy - av * uR;
                    const float bres = by - av * (ab + rv * bb);
                    const float rr = av * bres + (p.rv_mean - rv) * p.rv_ivar;
                    float drv = (sp.S * rr - sr * rs) * __builtin_amdgcn_rcpf(sp.S * r_den - sr * sr);
                    drv = fmaxf(drv, p.rvmin - rv);
                    drv = fminf(drv, p.rvmax - rv);
                    rv += drv;
                    const float RR2 = aa + rv * (2.f * ab + rv * bb);
                    const float yR2 = ay + rv * by;
                    dav_o = dav;
                    drv_o = drv;
                    const float uR2 = ua + rv * ub;
                    const float oc = (uy - av * uR2) * __builtin_amdgcn_rcpf(sp.S);
                    const float tt0 = (sp.gbar - mbar) + oc;
                    return -0.5f * ((yy - av * (2.f * yR2 - av * RR2)) + sp.S * (tt0 * tt0 - oc * oc));
                };
                float d1, d2;
                if (K >= 1) {
                    const float lw = sweep(d1, d2);
                    if (live) {
                        if (lw == lw) {
                            const float st = fmaxf(fabsf(d1), fabsf(d2));
                            mx[g][0] = fmaxf(mx[g][0], lw);
                            if (st >= p.mtol_hi) mx[g][1] = fmaxf(mx[g][1], lw);
                            if (st >= p.mtol_lo) mx[g][2] = fmaxf(mx[g][2], lw);
                        } else {
                            mx[g][8] = 1.f;
                        }
                    }
                }
                if (K >= 2) {
                    const float lw = sweep(d1, d2);
                    if (live) {
                        if (lw == lw) {
                            const float st = fmaxf(fabsf(d1), fabsf(d2));
                            mx[g][3] = fmaxf(mx[g][3], lw);
                            if (st >= p.mtol_hi) mx[g][4] = fmaxf(mx[g][4], lw);
                            if (st >= p.mtol_lo) mx[g][5] = fmaxf(mx[g][5], lw);
                        } else {
                            mx[g][8] = 1.f;
                        }
                    }
                }
                for (int k = 2; k < K; ++k) sweep(d1, d2);
            }
            // MLE in scaled units: F = A f, A = 10^(-0.4 mbar), d = D dd
            float num, den;
            if constexpr (RVF) {
                num = 0.f;
                den = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(C10 * av, R[j], mcC[j]));
                    const float fw = e * sp.iv[j];
                    num = fmaf(sp.dd[j], fw, num);
                    den = fmaf(e, fw, den);
                }
            } else {
                f32x2 num2 = 0.f, den2 = 0.f;
#pragma unroll
                for (int j = 0; j < NB; j += 2) {
                    const f32x2 Rj = pk_fma((f32x2)rv, ld2(dr + j), ld2(R + j));
                    const f32x2 arg = C10 * pk_fma((f32x2)av, Rj, ld2(mc + j));
                    const f32x2 e = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
                    const f32x2 fw = e * ld2(sp.iv + j);
                    num2 = pk_fma(ld2(sp.dd + j), fw, num2);
                    den2 = pk_fma(e, fw, den2);
                }
                num = hsum(num2);
                den = hsum(den2);
            }
            const float q = __builtin_amdgcn_exp2f(C10 * (sp.gbar - mbar));       // D / A
            float tt = num * __builtin_amdgcn_rcpf(den);
            float sc = tt * q;
            if (sc <= 1e-20f) {
                sc = 1e-20f;
                tt = sc * __builtin_amdgcn_rcpf(q);
            }
            const float chi2 = fmaf(tt, fmaf(tt, den, -2.f * num), sp.DD2);
            const float lnl = -0.5f * chi2;
            float lnlp = lnl;
            if (sp.has_par) {
                const float dp = __builtin_amdgcn_sqrtf(sc) - sp.par;
                lnlp = lnl - 0.5f * (dp * dp * sp.par_ivar);
            }
            float lnpr = lnl;
            if (p.dim_prior) lnpr = sp.c0 + sp.c1 * __logf(chi2) - 0.5f * chi2;
            if (sp.sp_on) {
                const float vt = sp.sp_var + q * q * __builtin_amdgcn_rcpf(den);
                const float ds = sc - sp.sp_mean;
                lnpr += -0.5f * (ds * ds * __builtin_amdgcn_rcpf(vt) + __logf(6.2831853071795865f * vt));
            }
            // a chi2 the cancellation cannot resolve, or a star float32 cannot
            // represent: NaN = "re-evaluate in float64"
            if (!(chi2 > 4.f * sp.eps) || !sp.ok) lnlp = NAN;
            lnpr = surv_clean(lnpr);                  // (the plane's tag range stays free)
            if (!(chi2 > sp.chi2_lo) || !sp.ok) lnpr = NAN;
            if (live) {
                const int64_t o = (int64_t)s * nmodel + i;
                lnlp32[o] = lnlp;
                lnpr32[o] = lnpr;
                if (lnlp != lnlp || lnpr != lnpr) mx[g][9] = 1.f;
                if (lnlp > mx[g][6]) mx[g][6] = lnlp;
                if (lnpr > mx[g][7]) mx[g][7] = lnpr;
            }
        }
    }
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int v = 0; v < NV32; ++v) {
            float x = mx[g][v];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
            if (lane == 0) slot[wv][g * NV32 + v] = x;
        }
    __syncthreads();
    if (threadIdx.x < G * NV32) {
        const int g = threadIdx.x / NV32, v = threadIdx.x % NV32;
        if (g < ng) {
            const int s = star_ids[g0 + g];
            const float x = fmaxf(fmaxf(slot[0][threadIdx.x], slot[1][threadIdx.x]),
                                  fmaxf(slot[2][threadIdx.x], slot[3][threadIdx.x]));
            part[((int64_t)bx * nstar + s) * NV32 + v] = x;
        }
    }
}

// Per-star reduction of the float32 partials and the K1 decision they allow.
//   status[s] = 0: K1 = k1[s] decided (== kfix used)      1: K1 = 1 decided, planes were
//   computed after 2 sweeps -> redo k_pre32 with kfix = 1  2: undecided -> exact probe
// With `ctr` given the follow-up lists are put together here, on the device (no host round
// trip): status 1 -> kfix[s] = K1 and s appended to redo_ids (count ctr[1]); status 2 -> s
// appended to probe_ids (count ctr[0]).  `nrun_dev`: the launch covers the whole batch, only
// the first *nrun_dev entries of star_ids exist.
__global__ void k_pre_decide(int nblkx, int nstar, const int32_t *__restrict__ star_ids,
                             const float *__restrict__ part, const Star32 *__restrict__ stars,
                             float ln_init, int rvf, int32_t *kfix,
                             int accept, float *__restrict__ st32, int32_t *__restrict__ k1,
                             int32_t *__restrict__ status, double *__restrict__ nomA,
                             int32_t *ctr, int32_t *__restrict__ probe_ids,
                             int32_t *__restrict__ redo_ids, const int32_t *nrun_dev) {
    __shared__ float sm[NV32][4];
    if (nrun_dev && (int)blockIdx.x >= *nrun_dev) return;
    const int s = star_ids[blockIdx.x];
    float v[NV32];
    for (int q = 0; q < NV32; ++q) v[q] = -INFINITY;
    for (int b = threadIdx.x; b < nblkx; b += blockDim.x) {
        const float *pp = part + ((int64_t)b * nstar + s) * NV32;
        for (int q = 0; q < NV32; ++q) v[q] = fmaxf(v[q], pp[q]);
    }
    for (int q = 0; q < NV32; ++q) {
        float x = v[q];
        for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
        if ((threadIdx.x & 63) == 0) sm[q][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int q = 0; q < NV32; ++q)
        v[q] = fmaxf(fmaxf(sm[q][0], sm[q][1]), fmaxf(sm[q][2], sm[q][3]));
    for (int q = 0; q < NV32; ++q) st32[(int64_t)s * NV32 + q] = v[q];
    const Star32 &sp = stars[s];
    const float ew = 2.f * sp.epsw;
    int st = 2, K1 = 0;
    const int K = kfix[s];
    if (sp.ok && !(v[8] > 0.f) && v[0] > -INFINITY) {
        const bool cont1 = v[1] > v[0] + ln_init + ew;      // surely not converged after sweep 1
        const bool may1 = v[2] > v[0] + ln_init - ew;       // possibly not converged
        if (!may1) {
            K1 = 1;
            st = (rvf || K == 1) ? 0 : 1;
        } else if (cont1) {
            if (rvf) {            // pinned Rv: sweep 2 never moves (see k_k1probe)
                K1 = 2;
                st = 0;
            } else if (K >= 2) {
                const bool cont2 = v[4] > v[3] + ln_init + ew;
                const bool may2 = v[5] > v[3] + ln_init - ew;
                if (!may2) {
                    K1 = 2;
                    st = (K == 2) ? 0 : 2;
                } else if (cont2) {
                    st = 2;       // K1 >= 3: exact probe
                }
            }
        }
    }
    if (!accept) {      // accept: K1 already known exactly (exact probe / redo), stats only
        k1[s] = K1;
        status[s] = st;
        if (ctr) {
            if (st == 1) {
                kfix[s] = K1;
                redo_ids[atomicAdd(ctr + 1, 1)] = s;
            } else if (st == 2) {
                probe_ids[atomicAdd(ctr + 0, 1)] = s;
            }
        }
    }
    // nominees for the exact maximum of lnl_p: within 2 eps of the float32 maximum
    nomA[s] = (double)v[6] - 2. * (double)sp.eps;
}

// ---------------------------------------------------------------------------
// float64 building blocks on a register-resident tile
// ---------------------------------------------------------------------------
template <int NB, bool RVF>
struct Tile64 {
    Coef<NB> c;
    double F0[NB];
    double R[RVF ? NB : 1];
};

template <int NB, bool RVF>
__device__ __forceinline__ void tile_load(const float *__restrict__ grid, int64_t nmodel_pad,
                                          int64_t i, double rv_mean, Tile64<NB, RVF> &t) {
    load_coef<NB>(grid, nmodel_pad, i, t.c);
    load_F0<NB>(grid, nmodel_pad, i, t.F0);
    if constexpr (RVF) coef_R<NB>(t.c, rv_mean, t.R);
}

// K sweeps of the magnitude phase from (av_mean, rv_mean) (fitting.py:176-243)
template <int NB, bool RVF>
__device__ __forceinline__ void mag_phase(const Tile64<NB, RVF> &t, const StarPrep &sp,
                                          const DevParams &p, int K, double &av, double &rv) {
    av = p.av_mean;
    rv = p.rv_mean;
    if constexpr (RVF) {
        // pinned Rv: the first sweep lands on the (clamped) minimiser, later ones move by
        // rounding noise only (see k_k1probe); one step, like k_derive
        GramR Gm;
        gram_init_rf<NB>(t.c, t.R, sp, Gm);
        double a_, c_;
        if (K > 0) gram_sweep_rf(Gm, sp.S, p, av, a_, c_);
    } else {
        Gram Gm;
        gram_init<NB>(t.c, sp, Gm);
        for (int k = 0; k < K; ++k) {
            double a_, b_, c_;
            gram_sweep(Gm, sp.S, p, av, rv, a_, b_, c_);
        }
    }
}

template <int NB, bool RVF, bool FULL>
__device__ __forceinline__ void mle_at(const Tile64<NB, RVF> &t, const StarPrep &sp,
                                       const DevParams &p, double av, double rv,
                                       const double *__restrict__ tbl, Mle &m) {
    if constexpr (RVF) mle_fast_rf<NB, true, FULL ? 2 : 0>(t.c, t.R, t.F0, sp, p, av, tbl, m);
    else mle_fast<NB, true, FULL ? 2 : 0>(t.c, t.F0, sp, p, av, rv, tbl, m);
}

__device__ __forceinline__ void audit(float *__restrict__ aud, int s, float f32v, double f64v,
                                      double thr) {
    // largest |f32 - f64| among lanes within 12 of the threshold (run-time check of eps)
    if (aud && f64v > thr - 12. && f32v == f32v) {
        const float d = fabsf((float)(f64v - (double)f32v));
        atomicMax(reinterpret_cast<int *>(aud + s), __float_as_int(d));
    }
}

// mask word of (star s, tile t, wave w)
__device__ __forceinline__ int64_t mword(int s, int ntile, int t, int w) {
    return (int64_t)s * (4 * ntile) + (int64_t)t * 4 + w;
}

// ---------------------------------------------------------------------------
// k_top: exact maxima over nominees
// ---------------------------------------------------------------------------
// mode 0: nominees = !(lnlp32 < nom[s])                          -> max lnl_p
// mode 1: nominees = non-survivors with !(lnpr32 < nom[s])       -> max lnprob (mag-phase value)
//         (survivors carry a tag in that plane, see surv_tag)
// A (block, star) whose float32 block maximum (part32 column 6 + mode) is below nom[s]
// and that holds no NaN lane is skipped outright.
// part[(bx * nstar + s)] = block maximum (-inf if no nominee)
template <int NB, bool RVF, int G>
__global__ void __launch_bounds__(TILE, top_waves(NB))
k_top(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
      const int32_t *__restrict__ star_ids, const StarPrep *__restrict__ stars, DevParams p,
      const int32_t *__restrict__ k1, int ntile, int mode, const float *__restrict__ plane32,
      const double *__restrict__ nom, const float *__restrict__ surv32,
      const float *__restrict__ part32, double *__restrict__ part, float *__restrict__ aud) {
    __shared__ double slot[4][G];
    __shared__ double s_tbl[64];
    const int g0 = blockIdx.y * G;
    const int ng = min(G, nrun - g0);
    const int t0 = blockIdx.x * F2_T;
    const int t1 = min(ntile, t0 + F2_T);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bool hot[G];
    bool anyhot = false;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        hot[g] = false;
        if (g < ng) {
            const int s = star_ids[g0 + g];
            const float *pp = part32 + ((int64_t)blockIdx.x * nstar + s) * NV32;
            hot[g] = !((double)pp[6 + mode] < nom[s]) || pp[9] > 0.f;
            anyhot = anyhot || hot[g];
        }
    }
    // (all but a few per cent of the workgroups end here: before the table staging, its
    // barrier and everything else -- a launch is ~12 000 workgroups for some hundred hot ones)
    if (!anyhot) {
        if ((int)threadIdx.x < ng) part[(int64_t)blockIdx.x * nstar + star_ids[g0 + threadIdx.x]] = -INFINITY;
        return;
    }
    stage_exp_table(s_tbl);
    __syncthreads();
    if (lane < G) slot[wv][lane] = -INFINITY;      // wave-private row: no barrier needed
    // nominee masks of the block's tiles for its hot stars: all float32 values are requested
    // first (clamped addresses, no guards: every load of the batch is in flight at once;
    // one guarded load per (tile, star) made each a round trip of its own), the masks wait
    // in a wave-private LDS row
    __shared__ unsigned long long s_need[4][G][F2_T];
    const float *__restrict__ tagp = mode == 1 ? surv32 : plane32;     // (mode 0: value unused)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (g < ng && hot[g]) {
            const int s = star_ids[g0 + g];
            const double nm_s = nom[s];
            float v[F2_T], tg[F2_T];
#pragma unroll
            for (int u = 0; u < F2_T; ++u) {
                const int64_t i = (int64_t)(t0 + u) * TILE + threadIdx.x;
                const int64_t ic = i < nmodel ? i : nmodel - 1;
                v[u] = plane32[(int64_t)s * nmodel + ic];
                tg[u] = tagp[(int64_t)s * nmodel + ic];
            }
#pragma unroll
            for (int u = 0; u < F2_T; ++u) {
                const int64_t i = (int64_t)(t0 + u) * TILE + threadIdx.x;
                bool nm = t0 + u < t1 && i < nmodel && !(v[u] < nm_s);
                if (mode == 1 && nm) nm = !surv_is(tg[u]);
                const unsigned long long b = __ballot(nm);
                if (lane == 0) s_need[wv][g][u] = b;
            }
        } else if (lane < F2_T) {
            s_need[wv][g][lane] = 0ull;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int t = t0; anyhot && t < t1; ++t) {
        const int64_t i = (int64_t)t * TILE + threadIdx.x;
        unsigned long long need[G];
        bool any = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned long long b = s_need[wv][g][t - t0];
            need[g] = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                      (unsigned int)__builtin_amdgcn_readfirstlane((int)b);
            any = any || need[g] != 0ull;
        }
        if (!any) continue;
        Tile64<NB, RVF> tl;
        tile_load<NB, RVF>(grid, nmodel_pad, i, p.rv_mean, tl);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (g >= ng || need[g] == 0ull) continue;
            const int s = star_ids[g0 + g];
            const StarPrep &sp = stars[s];
            double av, rv;
            mag_phase<NB, RVF>(tl, sp, p, k1[s], av, rv);
            Mle m;
            mle_at<NB, RVF, false>(tl, sp, p, av, rv, s_tbl, m);
            double val;
            if (mode == 0) val = cull_stat(sp, m);
            else val = first_cut_lnprob(sp, final_lnl<false>(sp, p, m.chi2, false), m.scale, m.i00);
            const bool mine = (need[g] >> lane) & 1ull;
            if (mine) audit(aud, s, plane32[(int64_t)s * nmodel + i], val, nom[s]);
            const double x = wave_max((mine && val == val) ? val : -INFINITY);
            if (lane == 0 && x > slot[wv][g]) slot[wv][g] = x;
        }
    }
    __syncthreads();
    if (threadIdx.x < ng) {
        const int g = threadIdx.x;
        double x = slot[0][g];
        x = slot[1][g] > x ? slot[1][g] : x;
        x = slot[2][g] > x ? slot[2][g] : x;
        x = slot[3][g] > x ? slot[3][g] : x;
        part[(int64_t)blockIdx.x * nstar + star_ids[g0 + g]] = x;
    }
}

// ---------------------------------------------------------------------------
// k_hot_list + k_top1: the same exact maxima from a LIST of the hot (block, star) pairs
// ---------------------------------------------------------------------------
// k_top is launched over every (block, star group) and all but a few per cent of its ~12 000
// workgroups find nothing to do: 0.08 ms per launch, twice per call, for some hundred hot
// pairs.  k_hot_list (one workgroup per star) lists the pairs whose float32 block maximum
// reaches the nominee level (or that hold a NaN lane) and marks the others' partials -inf;
// k_top1 walks the list with a fixed launch, one (block, star) per workgroup and turn.
//   mode 0: nom = nomA (k_pre_decide);   mode 1: nomB[s] = maxsurv[s] - eps is formed here.
// An entry is  block * BRUTUS_MAX_BATCH + star.
__global__ void __launch_bounds__(256)
k_hot_list(int nblkx, int nstar, int mode, const float *__restrict__ part32,
           const double *__restrict__ nomA, const double *__restrict__ maxsurv,
           const Star32 *__restrict__ s32, double *__restrict__ nomB, double *__restrict__ part,
           int32_t *__restrict__ hot, int32_t *__restrict__ nhot) {
    const int s = blockIdx.x;
    double nm;
    if (mode == 0) {
        nm = nomA[s];
    } else {
        nm = maxsurv[s] - (double)s32[s].eps;
        if (threadIdx.x == 0) nomB[s] = nm;
    }
    for (int b = threadIdx.x; b < nblkx; b += blockDim.x) {
        const float *pp = part32 + ((int64_t)b * nstar + s) * NV32;
        const bool is_hot = !((double)pp[6 + mode] < nm) || pp[9] > 0.f;
        if (is_hot) hot[atomicAdd(nhot, 1)] = b * BRUTUS_MAX_BATCH + s;
        else part[(int64_t)b * nstar + s] = -INFINITY;
    }
}

template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, top_waves(NB))
k_top1(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nstar,
       const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1, int ntile,
       int mode, const float *__restrict__ plane32, const double *__restrict__ nom,
       const int32_t *__restrict__ hot, const int32_t *__restrict__ nhot,
       double *__restrict__ part, float *__restrict__ aud) {
    __shared__ double slot[4];
    __shared__ double s_tbl[64];
    const int n = *nhot;
    if ((int)blockIdx.x >= n) return;
    stage_exp_table(s_tbl);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int e = blockIdx.x; e < n; e += gridDim.x) {
        const int entry = hot[e];
        const int b = entry / BRUTUS_MAX_BATCH, s = entry - b * BRUTUS_MAX_BATCH;
        const int t0 = b * F2_T, t1 = min(ntile, t0 + F2_T);
        const StarPrep &sp = stars[s];
        const double nm_s = nom[s];
        const int K = k1[s];
        // the block's float32 values first, all in flight at once (clamped addresses)
        float v[F2_T];
#pragma unroll
        for (int u = 0; u < F2_T; ++u) {
            const int64_t i = (int64_t)(t0 + u) * TILE + threadIdx.x;
            v[u] = plane32[(int64_t)s * nmodel + (i < nmodel ? i : nmodel - 1)];
        }
        double mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < F2_T; ++u) {
            const int64_t i = (int64_t)(t0 + u) * TILE + threadIdx.x;
            // nominee: not below the level (NaN counts); mode 1: and not a survivor (tagged)
            bool mine = t0 + u < t1 && i < nmodel && !((double)v[u] < nm_s);
            if (mode == 1 && mine) mine = !surv_is(v[u]);
            if (__ballot(mine) == 0ull) continue;          // (wave-uniform)
            Tile64<NB, RVF> tl;
            tile_load<NB, RVF>(grid, nmodel_pad, i, p.rv_mean, tl);
            double av, rv;
            mag_phase<NB, RVF>(tl, sp, p, K, av, rv);
            Mle m;
            mle_at<NB, RVF, false>(tl, sp, p, av, rv, s_tbl, m);
            double val;
            if (mode == 0) val = cull_stat(sp, m);
            else val = first_cut_lnprob(sp, final_lnl<false>(sp, p, m.chi2, false), m.scale, m.i00);
            if (mine) audit(aud, s, v[u], val, nm_s);
            if (mine && val == val) mx = val > mx ? val : mx;
        }
        block_max_store(mx, slot, part + (int64_t)b * nstar + s);
    }
}

// thresholds from k_top's partials.
//   mode 0: thr_cull[s] = max + ln_init;  candS[s] = thr_cull - eps
//   mode 1: thr_sel[s] = max(maxsurv[s], max) + ln_wt
__global__ void k_top_decide(int nblkx, int nstar, const int32_t *__restrict__ star_ids, int mode,
                             const double *__restrict__ part, const Star32 *__restrict__ s32,
                             double ln_thr, const double *__restrict__ maxsurv,
                             double *__restrict__ thr, double *__restrict__ cand) {
    __shared__ double sm[4];
    const int s = star_ids[blockIdx.x];
    double v = -INFINITY;
    for (int b = threadIdx.x; b < nblkx; b += blockDim.x) {
        const double x = part[(int64_t)b * nstar + s];
        v = x > v ? x : v;
    }
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x != 0) return;
    v = sm[0];
    for (int w = 1; w < 4; ++w) v = sm[w] > v ? sm[w] : v;
    if (mode == 0) {
        thr[s] = v + ln_thr;
        cand[s] = v + ln_thr - (double)s32[s].eps;
    } else {
        const double m = maxsurv[s] > v ? maxsurv[s] : v;
        thr[s] = m + ln_thr;
    }
}

// nomB[s] = maxsurv[s] - eps: the non-survivors that could exceed the survivors' maximum
__global__ void k_nomB(int nstar, const double *__restrict__ maxsurv, const Star32 *__restrict__ s32,
                       double *__restrict__ nomB) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nstar) nomB[s] = maxsurv[s] - (double)s32[s].eps;
}

// Ordered compaction of a float32 plane: {i : !(plane[s][i] < thr[s])} (NaN counts as a
// hit): one membership word per wave and tile, one count per (star, model chunk).
__global__ void __launch_bounds__(TILE)
k_cmp_count32(int64_t nmodel, int ntile, const float *__restrict__ plane,
              const double *__restrict__ thr, int64_t *__restrict__ counts,
              unsigned long long *__restrict__ mask) {
    __shared__ int wsum[4];
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    const double th = thr[s];
    int n = 0;
#ifndef CNT_U
#define CNT_U 16
#endif
    constexpr int U = CNT_U;      // tiles in flight per lane (one 4-byte load each: latency-bound otherwise)
    for (int tb = t0; tb < t1; tb += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // (clamped address + select: a guarded load compiles to a branch and a full
            // wait per load, which serialises the batch)
            const int64_t i = (int64_t)(tb + u) * TILE + threadIdx.x;
            const bool in = tb + u < t1 && i < nmodel;
            const float x = plane[(int64_t)s * nmodel + (in ? i : nmodel - 1)];
            v[u] = in ? x : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (tb + u >= t1) break;
            const bool hit = !((double)v[u] < th);
            n += hit ? 1 : 0;
            const unsigned long long b = __ballot(hit);
            if ((threadIdx.x & 63) == 0)
                mask[(int64_t)s * (4 * ntile) + (int64_t)(tb + u) * 4 + (threadIdx.x >> 6)] = b;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[(int64_t)s * NCHUNK + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// ---------------------------------------------------------------------------
// k_sel_classify + k_sel_band: the first cut of lnpost as a bit-mask (fitting.py:976-991)
// ---------------------------------------------------------------------------
// grid = (NCHUNK, nstar), one star per workgroup.  Survivors (tagged in the lnprob~ plane)
// are tested on their final float64 lnprob (staged in candidate order); the rest on
// lnprob~ with the margin eps.
// Models inside the band |lnprob~ - thr| <= eps (or NaN) go to the block's region of
// `bandq` (starts at s * nmodel + first model of the chunk: cannot overflow) and are
// re-evaluated in float64, with dense lanes, by k_sel_band, which ORs the outcome into
// the membership words.  Outputs: membership words + per-chunk counts of the selected
// models (mask, counts) and of those among them that are not survivors of the cull, whose
// values k_derive has to compute (dmask, dcounts).
__global__ void __launch_bounds__(TILE)
k_sel_classify(int64_t nmodel, int ntile, const Star32 *__restrict__ s32,
               const float *__restrict__ lnpr32,
               const double *__restrict__ lnprob_st, const int64_t *__restrict__ cand_off,
               const double *__restrict__ thr_sel,
               int64_t *__restrict__ counts, unsigned long long *__restrict__ mask,
               int64_t *__restrict__ dcounts, unsigned long long *__restrict__ dmask,
               int32_t *__restrict__ bandq, int32_t *__restrict__ bandn) {
    __shared__ int qn;
    __shared__ int wsum[4], dsum[4];
    if (threadIdx.x == 0) qn = 0;
    __syncthreads();
    const int s = blockIdx.y, c = blockIdx.x;
    const int t0 = (int)((int64_t)ntile * c / NCHUNK), t1 = (int)((int64_t)ntile * (c + 1) / NCHUNK);
    const double th = thr_sel[s];
    const double e = (double)s32[s].eps;
    const int64_t cbase = cand_off[s];
    int32_t *queue = bandq + (int64_t)s * nmodel + (int64_t)t0 * TILE;
    int n = 0, nd = 0;
#ifndef CLS_U
#define CLS_U 12
#endif
    constexpr int U = CLS_U;      // tiles in flight per lane
    for (int tb = t0; tb < t1; tb += U) {
        float v32[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = (int64_t)(tb + u) * TILE + threadIdx.x;
            const bool in = tb + u < t1 && i < nmodel;
            const float x = lnpr32[(int64_t)s * nmodel + (in ? i : nmodel - 1)];   // see k_cmp_count32
            v32[u] = in ? x : -INFINITY;
        }
        // (the survivors' staged values: second round of loads, again all in flight)
        double vst[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            vst[u] = surv_is(v32[u]) ? lnprob_st[cbase + surv_slot(v32[u])] : 0.;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tb + u;
            if (t >= t1) break;
            const int64_t i = (int64_t)t * TILE + threadIdx.x;
            bool yes = false, der = false, bd = false;
            if (i < nmodel) {
                if (surv_is(v32[u])) {
                    yes = vst[u] > th;
                } else {
                    const double v = (double)v32[u];
                    yes = der = v >= th + e;
                    bd = !yes && !(v < th - e);
                }
            }
            n += yes ? 1 : 0;
            nd += der ? 1 : 0;
            const unsigned long long b = __ballot(yes), bdr = __ballot(der);
            if ((threadIdx.x & 63) == 0) {
                const int64_t a = (int64_t)s * (4 * ntile) + (int64_t)t * 4 + (threadIdx.x >> 6);
                mask[a] = b;
                dmask[a] = bdr;
            }
            if (bd) queue[atomicAdd(&qn, 1)] = (int32_t)i;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_xor(n, off, 64);
        nd += __shfl_xor(nd, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        wsum[threadIdx.x >> 6] = n;
        dsum[threadIdx.x >> 6] = nd;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[(int64_t)s * NCHUNK + c] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        dcounts[(int64_t)s * NCHUNK + c] = dsum[0] + dsum[1] + dsum[2] + dsum[3];
        bandn[s * NCHUNK + c] = qn;
    }
}

// grid = (NCHUNK / SB_C, nstar, SB_Z): the queues of SB_C neighbouring chunks form one list
// (a single chunk queues ~50 models on the Av-only bench: a quarter of the lanes, and 8 192
// workgroups that each pay the full index -> row -> MLE latency chain); a list longer than a
// workgroup is shared in turns by the SB_Z workgroups of its column.
constexpr int SB_C = 8, SB_Z = 8;
template <int NB, bool RVF>
__global__ void __launch_bounds__(TILE, band_waves(NB))
k_sel_band(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int ntile,
           const StarPrep *__restrict__ stars, DevParams p, const int32_t *__restrict__ k1,
           const float *__restrict__ lnpr32, const double *__restrict__ thr_sel,
           const int32_t *__restrict__ bandq, const int32_t *__restrict__ bandn,
           int64_t *__restrict__ counts, unsigned long long *__restrict__ mask,
           int64_t *__restrict__ dcounts, unsigned long long *__restrict__ dmask,
           float *__restrict__ aud) {
    const int s = blockIdx.y, c0 = blockIdx.x * SB_C;
    int pre[SB_C + 1];
    pre[0] = 0;
#pragma unroll
    for (int k = 0; k < SB_C; ++k) pre[k + 1] = pre[k] + bandn[s * NCHUNK + c0 + k];
    const int n = pre[SB_C];
    if ((int)(blockIdx.z * TILE) >= n) return;      // (long lists are shared by gridDim.z workgroups)
    __shared__ double s_tbl[64];
    stage_exp_table(s_tbl);
    __syncthreads();
    const StarPrep &sp = stars[s];
    const double th = thr_sel[s];
    const int K = k1[s];
    for (int q = blockIdx.z * TILE + threadIdx.x; q < n; q += TILE * gridDim.z) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < SB_C; ++j) k += q >= pre[j] ? 1 : 0;
        int qk = q;
#pragma unroll
        for (int j = 1; j < SB_C; ++j) qk = k == j ? q - pre[j] : qk;
        const int c = c0 + k;
        const int t0 = (int)((int64_t)ntile * c / NCHUNK);
        const int64_t i = bandq[(int64_t)s * nmodel + (int64_t)t0 * TILE + qk];
        Tile64<NB, RVF> tl;
        gather_coef<NB>(grid, nmodel_pad, i, tl.c);
        compute_F0_tbl<NB>(tl.c, s_tbl, tl.F0);
        if constexpr (RVF) coef_R<NB>(tl.c, p.rv_mean, tl.R);
        double av, rv;
        mag_phase<NB, RVF>(tl, sp, p, K, av, rv);
        Mle m;
        mle_at<NB, RVF, false>(tl, sp, p, av, rv, s_tbl, m);
        const double lnprob =
            first_cut_lnprob(sp, final_lnl<false>(sp, p, m.chi2, false), m.scale, m.i00);
        audit(aud, s, lnpr32[(int64_t)s * nmodel + i], lnprob, th);
        if (lnprob > th) {       // (a band model is never a survivor: selected = derived)
            atomicOr(mask + (int64_t)s * (4 * ntile) + (i >> 6), 1ull << (i & 63));
            atomicOr(dmask + (int64_t)s * (4 * ntile) + (i >> 6), 1ull << (i & 63));
            atomicAdd(reinterpret_cast<unsigned long long *>(counts + (int64_t)s * NCHUNK + c), 1ull);
            atomicAdd(reinterpret_cast<unsigned long long *>(dcounts + (int64_t)s * NCHUNK + c), 1ull);
        }
    }
}


// Deep K1 probe for ONE star (k_mag_stats partials with nstar = 1): the first sweep
// k at which max{logwt : step >= tol} <= max logwt + ln(init_thresh)
// (fitting.py:246-264); 0 if none of the kmax sweeps converged.
__global__ void k_k1_deep_decide(int ntile, int kmax, const double *__restrict__ part,
                                 double ln_init, int32_t *__restrict__ out_k1) {
    __shared__ double sm[2][4];
    __shared__ int done;
    if (threadIdx.x == 0) done = 0;
    __syncthreads();
    for (int k = 0; k < kmax; ++k) {
        double L = -INFINITY, T = -INFINITY;
        for (int t = threadIdx.x; t < ntile; t += blockDim.x) {
            const double *pp = part + (int64_t)t * (2 * kmax) + 2 * k;
            L = pp[0] > L ? pp[0] : L;
            T = pp[1] > T ? pp[1] : T;
        }
        L = wave_max(L);
        T = wave_max(T);
        if ((threadIdx.x & 63) == 0) {
            sm[0][threadIdx.x >> 6] = L;
            sm[1][threadIdx.x >> 6] = T;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) {
                L = sm[0][w] > L ? sm[0][w] : L;
                T = sm[1][w] > T ? sm[1][w] : T;
            }
            L = L > -BIG ? L : -BIG;
            if (!(T > L + ln_init)) {
                *out_k1 = k + 1;
                done = 1;
            }
        }
        __syncthreads();
        if (done) return;
    }
    if (threadIdx.x == 0) *out_k1 = 0;
}

}  // namespace
