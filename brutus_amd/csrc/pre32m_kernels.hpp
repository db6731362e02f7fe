// pre32m_kernels.hpp -- the float32 proof pass with its band contractions on the matrix pipe.
// Compiled in pre32s_unit.hip beside k_pre32s (same statistics, same outputs, same launch
// arguments); k_pre32s stays as the A/B partner behind BRUTUS_PRE32_MFMA=0.
//
// Why, and what came of it (round 6; profiles/r06_pre32m_ab.txt, profiles/r06_mfma_valu.txt).
// Of k_pre32s' 221 (pinned Rv) / ~330 (general) vector instructions per (star, model) pair-wave,
// 84 / 132 form the sweep's inner products over the bands,
//     uR = sum_j w_j R_j,  RR = sum_j w_j R_j^2,  yR = sum_j w_j R_j y_j,  uy = sum_j w_j y_j,
//     yy = sum_j w_j y_j^2,      y_j = gcC_j - mcC_j        (fitting.py:176-243 as Gram sums),
// and every one of them is  sum_j (star vector)_j x (model feature)_j : a K = NB GEMM between
// the stars' {w, w gcC} and the models' {R, R^2, mcC, R mcC, mcC^2}.  v_mfma_f32_16x16x4_f32
// computes it in exact float32 (an fmaf chain): 21 (pinned) / 36 (general) MFMAs per 16 models x
// 16 stars, 2.6 / 4.5 matrix cycles per pair.  The hope was that they run BESIDE the vector
// instructions that remain.  They do not: on gfx950 a float32 MFMA and float32 vector
// instructions of one SIMD take the SUM of their times, interleaved in one wave or issued from
// different waves (tools/ubench/mfma_valu.hip: 64 v_fmac_f32 0.61 ms, 4 MFMAs 0.47 ms, both 1.07
// ms) -- the float32 matrix rate equals the vector rate because it is the same multipliers.
// What is left is the MFMA's better packing (no y = g - m, no y w, 1024 multiply-adds per
// issue slot): the pinned kernel 0.594 -> 0.541 ms alone, 0.72 -> 0.65 ms inside a call (-9.5 %),
// the general one -4 %; whole job +1.3 % (configs[1]) / -0.8 % (configs[2]).  Below the 10 % the
// review set as the bar, and the expanded sums below cancel where the vector form subtracts
// first: NOT the default (BRUTUS_PRE32_MFMA=1 selects it; tests/test_gpu_fit2.py keeps it honest).
// The 16-star granularity, the XCD-aware block numbering and the 16-byte stores of this form
// are sound on their own.
//
// Expanded sums.  With y = gcC - mcC,
//     yR = sum (w gcC)_j R_j - sum w_j (R mcC)_j,
//     uy = U0 - sum w_j mcC_j,                         U0 = sum w gcC   (~ 0: gc is w-centred)
//     yy = G2 - 2 sum (w gcC)_j mcC_j + sum w_j mcC_j^2,   G2 = sum w gcC^2.
// The two-term planes run as ONE chain with the terms of a band adjacent -- k = (w gcC)_j x
// feature, k + 1 = w_j x feature' -- so the partial sums stay near the running value of the
// direct form plus one band's term; U0 and G2 are the chains' initial values.  The cancellation
// costs accuracy where |y| << |gcC|: |error| <~ 2 NB u sum_j w_j (|gcC_j| + |mcC_j|)^2 instead of
// ~ u sum w |y| (|gcC| + |mcC|); Star32::eps / epsw carry the term (k_prep32, fit2_kernels.hpp),
// BRUTUS_AUDIT and tests/test_gpu_fit2.py hold it to account.
//
// Layout.  v_mfma_f32_16x16x4_f32: A operand lane l = A[i = l & 15][k = l >> 4], B operand lane
// l = B[k = l >> 4][j = l & 15], result register r of lane l = D[4 (l >> 4) + r][l & 15].  Rows
// i = 16 models of a tile, columns j = the wave's 16 stars: a lane ends up with ITS star (lane &
// 15) and four consecutive models (4 (lane >> 4) + r) -- the per-pair code of k_pre32s runs
// unchanged with lane = (star, model quarter), reads the model-only values (mcC_j, R_j) from the
// tile's LDS rows with quarter-wave-uniform addresses, and stores its four models as ONE 16-byte
// word per plane (no LDS transposition).
// Staging: lane (i, k) loads model i's raw coefficients for the NB / 2 bands of parity k >> 1,
// builds the features and hands the MFMAs their A operands straight from registers:
//     two-term planes, MFMA t: k -> band 2 t + (k >> 1), (k & 1) ? second term : first term
//     one-term planes, MFMA t: k -> band (k >> 1) + 4 t + 2 (k & 1)
// (any assignment of bands to k slots works as long as both operands use the same one).
// Semantics to hold: /root/reference/brutus/fitting.py:173-264, :743-759, :976-985 +
// pdf.py:209-218; float32 only ever classifies.
#pragma once

#include "pre32s_kernels.hpp"

namespace {

typedef float pm_f32x4 __attribute__((ext_vector_type(4)));

constexpr int pm_rs(int nb, bool rvf) { return ((rvf ? 2 : 3) * nb + 1 + 3) & ~3; }   // floats per LDS row
// four rows (one per result register) per lane quarter; the quarters 8 banks apart, so that the
// four 16-byte words of a quarter-uniform ds_read_b128 come from disjoint banks
constexpr int pm_gs(int rs) { return 4 * rs + ((8 - (4 * rs) % 32 + 32) % 32); }
constexpr bool pm_bands(int nb) { return nb == 8 || nb == 12; }

// a where the mask is all ones, b where it is zero: v_bfi_b32 (written with bit operations so that
// the optimiser does not turn "odd ? x[2 t + 1] : x[2 t]" into an indexed access of a scratch copy)
__device__ __forceinline__ float pm_pick(unsigned m, float a, float b) {
    return __uint_as_float((__float_as_uint(a) & m) | (__float_as_uint(b) & ~m));
}

// DIAG (development builds only; the library launches DIAG = 0): 1 = the MFMAs left out, their
// operands still computed -- what the vector side alone costs
template <int DIAG>
__device__ __forceinline__ pm_f32x4 pm_mfma(float a, float b, pm_f32x4 c) {
    if constexpr (DIAG == 1) {
        asm volatile("" ::"v"(a), "v"(b));
        return c;
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
}

template <int NB, bool RVF, int DIAG = 0>
__global__ void __launch_bounds__(PS_TILE, RVF ? 3 : 2)
k_pre32m(const float *__restrict__ grid, int64_t nmodel, int64_t nmodel_pad, int nblkx, int nstar, int nrun,
         const int32_t *__restrict__ star_ids, const Star32 *__restrict__ stars, P32 p,
         float *__restrict__ lnlp32, float *__restrict__ lnpr32, float *__restrict__ part) {
    constexpr int NQ = NB / 2;                    // bands per staging lane (one parity class)
    constexpr int NT = NB / 4;                    // MFMAs per one-term plane
    constexpr int RS = pm_rs(NB, RVF), GS = pm_gs(RS);
    constexpr int OFF_A = NB, OFF_B = 2 * NB, OFF_MBAR = (RVF ? 2 : 3) * NB;
    __shared__ __attribute__((aligned(16))) float s_row[4][4 * GS];
    __shared__ float s_mx[4][NV32][16];
    const float C10 = -1.32877123795494494f;      // c = -0.4 log2(10)
    const float NINF = -INFINITY;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sj = lane & 15, g = lane >> 4;      // pair code: star column, model quarter; operands: k = g
    const int par = g >> 1, odd = g & 1;          // staging: band parity of this lane, term of a pair
    // 1-D launch of 8 ceil(nblkx / 8) ngroup workgroups.  Workgroup L runs on XCD L % 8 (round-robin
    // dispatch) and takes, in turn, the star groups of the model blocks L % 8, L % 8 + 8, ...: the
    // ngroup workgroups that share a block's coefficients follow each other on ONE XCD and find
    // them in its L2 (dealt out in launch order they would land on 8 XCDs and fetch them 8 times)
    const int ngroup = (nrun + 15) >> 4;
    const int q_ = (int)(blockIdx.x >> 3);
    const int bx = (q_ / ngroup) * 8 + (int)(blockIdx.x & 7), sg = q_ % ngroup;
    if (bx >= nblkx) return;
    const int sl = sg * 16 + sj;
    const bool slive = sl < nrun;
    const int s = star_ids[slive ? sl : nrun - 1];
    const int64_t sbase = slive ? (int64_t)s * nmodel : (int64_t)-1;
    const Star32 &sp = stars[s];
    // ---- star side: B operands, chain starts, the constants of the per-pair code ---------------
    float bw2[NQ], bw1[NT];
    float U0 = 0.f, G2 = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float gcC = C10 * sp.gc[j], w = sp.w[j];
        U0 = fmaf(w, gcC, U0);
        G2 = fmaf(w * gcC, gcC, G2);
    }
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        const int b = 2 * t + par;
        const float w = sp.w[b];
        bw2[t] = odd ? w : w * (C10 * sp.gc[b]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) bw1[t] = sp.w[par + 4 * t + 2 * odd];
    float dd[NB], iv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        dd[j] = sp.dd[j];
        iv[j] = sp.iv[j];
    }
    const float S = sp.S, rS = __builtin_amdgcn_rcpf(sp.S), gbarC = C10 * sp.gbar, DD2 = sp.DD2;
    const bool has_par = sp.has_par != 0, sp_on = sp.sp_on != 0;
    const float eps4 = sp.ok ? 4.f * sp.eps : INFINITY, chi2_lo = sp.ok ? sp.chi2_lo : INFINITY;
    const float parx = has_par ? sp.par : 0.f, par_hiv = has_par ? 0.5f * sp.par_ivar : 0.f;
    const float sp_mean = sp.sp_mean, sp_var = sp.sp_var, c0 = sp.c0, c1 = sp.c1;
    const bool any_par = __ballot(has_par) != 0ull, any_sp = __ballot(sp_on) != 0ull;     // wave-uniform
    const float avm = C10 * p.av_mean, av_lo = C10 * p.avmax, av_hi = C10 * p.avmin;
    const float tol_hi = -C10 * p.mtol_hi, tol_lo = -C10 * p.mtol_lo;
    const float lw_scale = -0.5f / (C10 * C10);
    float mx0 = NINF, mx1 = NINF, mx2 = NINF, mx3 = NINF, mx4 = NINF, mx5 = NINF, mx6 = NINF, mx7 = NINF,
          mx8 = NINF, mx9 = NINF;
    const int64_t i0 = (int64_t)bx * (F2_T * PS_TILE) + (int64_t)wv * PS_WM;
    const int64_t i1 = i0 + PS_WM < nmodel ? i0 + PS_WM : nmodel;
    // ---- model side ----------------------------------------------------------------------------
    const float inv_nf = 1.f / (float)p.nfilt;
    const float *gpar = grid + (int64_t)(3 * par) * nmodel_pad;       // this lane's parity class
    float cm[NQ], cr0[NQ], cdr[NQ];
    auto fetch = [&](int64_t ib) {
        // (a ragged last tile repeats its last model: maxima unchanged, stores guarded)
        int64_t i = ib + sj;
        i = i < i1 ? i : i1 - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float *c = gpar + (int64_t)(6 * q) * nmodel_pad + i;
            cm[q] = c[0];
            cr0[q] = c[nmodel_pad];
            cdr[q] = c[2 * nmodel_pad];
        }
    };
    const unsigned oddm = odd ? 0xffffffffu : 0u;
    const int wsel = odd ? OFF_A : 0;
    float *rowbase = &s_row[wv][0];
    float *wrow = rowbase + (sj >> 2) * GS + (sj & 3) * RS;           // staging: the row of model sj
    const float *rrow = rowbase + g * GS;                             // pair code: the quarter's four rows
    if (i0 < i1) fetch(i0);
    for (int64_t ib = i0; ib < i1; ib += 16) {
        const int nm = (int)(i1 - ib < 16 ? i1 - ib : 16);
        pm_f32x4 a_uR = {0.f, 0.f, 0.f, 0.f}, a_RR = a_uR, a_yR = a_uR, a_ub = a_uR, a_ab = a_uR, a_bb = a_uR,
                 a_by = a_uR;
        pm_f32x4 a_uy = {U0, U0, U0, U0}, a_yy = {G2, G2, G2, G2};
        {
            // (mean over the padded bands too: the blob holds zeros there, k_relayout)
            float ps = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) ps += cm[q];
            ps += __shfl_xor(ps, 32, 64);
            const float mbar = ps * inv_nf;
            float mc[NQ], A[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int b = 2 * q + par;
                mc[q] = C10 * (cm[q] - mbar);
                if constexpr (RVF) A[q] = fmaf(p.rv_mean, cdr[q], cr0[q]);
                else A[q] = cr0[q];
                // the tile's rows for the per-pair code: even lanes the magnitudes, odd ones R
                // (no branches: a lane pair (k, k ^ 1) holds the same model and bands, so what both
                // write -- dr, mbar -- lands twice with the same value)
                wrow[wsel + b] = odd ? A[q] : mc[q];
                if constexpr (!RVF) wrow[OFF_B + b] = cdr[q];
            }
            wrow[OFF_MBAR] = C10 * mbar;
            // two-term planes: band 2 t + par, first term from the even lanes, second from the odd
#pragma unroll
            for (int t = 0; t < NQ; ++t) {
                const float tm = odd ? -mc[t] : 1.f;
                a_yR = pm_mfma<DIAG>(A[t] * tm, bw2[t], a_yR);
                if constexpr (!RVF)
                    a_by = pm_mfma<DIAG>(cdr[t] * tm, bw2[t], a_by);
                a_yy = pm_mfma<DIAG>(mc[t] * (odd ? mc[t] : -2.f), bw2[t], a_yy);
                // (keep the chains alternating as written: left to itself the scheduler lines up six
                // dependent MFMAs of one accumulator, each waiting out the previous one's latency)
                __builtin_amdgcn_sched_barrier(0);
            }
            // one-term planes: band par + 4 t + 2 odd
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float xa = pm_pick(oddm, A[2 * t + 1], A[2 * t]);
                const float xm = pm_pick(oddm, mc[2 * t + 1], mc[2 * t]);
                a_uR = pm_mfma<DIAG>(xa, bw1[t], a_uR);
                a_RR = pm_mfma<DIAG>(xa * xa, bw1[t], a_RR);
                a_uy = pm_mfma<DIAG>(-xm, bw1[t], a_uy);
                if constexpr (!RVF) {
                    const float xb = pm_pick(oddm, cdr[2 * t + 1], cdr[2 * t]);
                    a_ub = pm_mfma<DIAG>(xb, bw1[t], a_ub);
                    a_ab = pm_mfma<DIAG>(xa * xb, bw1[t], a_ab);
                    a_bb = pm_mfma<DIAG>(xb * xb, bw1[t], a_bb);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (ib + 16 < i1) fetch(ib + 16);            // the next tile's coefficients: a whole tile of latency cover
        ps_wave_sync();
        float o_lnlp[4], o_lnpr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 *__restrict__ r4 = reinterpret_cast<const float4 *>(rrow + r * RS);
            float mcC[NB], A[NB], B[RVF ? 1 : NB];
#pragma unroll
            for (int k = 0; k < NB / 4; ++k) {
                const float4 a = r4[k], c = r4[NB / 4 + k];
                mcC[4 * k] = a.x; mcC[4 * k + 1] = a.y; mcC[4 * k + 2] = a.z; mcC[4 * k + 3] = a.w;
                A[4 * k] = c.x; A[4 * k + 1] = c.y; A[4 * k + 2] = c.z; A[4 * k + 3] = c.w;
                if constexpr (!RVF) {
                    const float4 d = r4[2 * (NB / 4) + k];
                    B[4 * k] = d.x; B[4 * k + 1] = d.y; B[4 * k + 2] = d.z; B[4 * k + 3] = d.w;
                }
            }
            const float dbar = gbarC - rrow[r * RS + OFF_MBAR];      // c (gbar - mbar)
            float av = avm, rv = p.rv_mean;                          // av: c Av throughout
            const float uy = a_uy[r], yy = a_yy[r];
            if constexpr (RVF) {
                const float uR = a_uR[r], RR = a_RR[r], yR = a_yR[r];
                const float rs = uy - av * uR;
                const float ra = (yR - av * RR) + (avm - av) * p.av_ivar;
                const float a_den = RR + p.av_ivar;
                float dav = (S * ra - uR * rs) * __builtin_amdgcn_rcpf(S * a_den - uR * uR);
                dav = fminf(dav, av_hi - av);
                dav = fmaxf(dav, av_lo - av);
                av += dav;
                const float oc = (uy - av * uR) * rS;
                const float tt0 = dbar + oc;
                const float lw = lw_scale * ((yy - av * (2.f * yR - av * RR)) + S * (tt0 * tt0 - oc * oc));
                const float st = fabsf(dav);
                mx0 = vmaxf(mx0, lw);
                mx1 = vmaxf(mx1, st >= tol_hi ? lw : NINF);
                mx2 = vmaxf(mx2, st >= tol_lo ? lw : NINF);
                mx8 = lw != lw ? 1.f : mx8;
            } else {
                const float ua = a_uR[r], ub = a_ub[r], aa = a_RR[r], ab = a_ab[r], bb = a_bb[r], ay = a_yR[r],
                            by = a_by[r];
                const float c2 = C10 * C10;
                auto sweep = [&](float &dav_o, float &drv_o) -> float {
                    const float uR = ua + rv * ub;
                    const float RR = aa + rv * (2.f * ab + rv * bb);
                    const float yR = ay + rv * by;
                    float rs = uy - av * uR;
                    const float ra = (yR - av * RR) + (avm - av) * p.av_ivar;
                    const float a_den = RR + p.av_ivar;
                    float dav = (S * ra - uR * rs) * __builtin_amdgcn_rcpf(S * a_den - uR * uR);
                    dav = fminf(dav, av_hi - av);
                    dav = fmaxf(dav, av_lo - av);
                    av += dav;
                    const float r_den = bb * av * av + c2 * p.rv_ivar;
                    const float sr = ub * av;
                    rs = uy - av * uR;
                    const float bres = by - av * (ab + rv * bb);
                    const float rr = av * bres + c2 * ((p.rv_mean - rv) * p.rv_ivar);
                    float drv = (S * rr - sr * rs) * __builtin_amdgcn_rcpf(S * r_den - sr * sr);
                    drv = fmaxf(drv, p.rvmin - rv);
                    drv = fminf(drv, p.rvmax - rv);
                    rv += drv;
                    const float RR2 = aa + rv * (2.f * ab + rv * bb);
                    const float yR2 = ay + rv * by;
                    dav_o = dav;
                    drv_o = drv;
                    const float uR2 = ua + rv * ub;
                    const float oc = (uy - av * uR2) * rS;
                    const float tt0 = dbar + oc;
                    return lw_scale * ((yy - av * (2.f * yR2 - av * RR2)) + S * (tt0 * tt0 - oc * oc));
                };
                float d1, d2;
                {
                    const float lw = sweep(d1, d2);
                    const float st = fmaxf(fabsf(d1) * (-1.f / C10), fabsf(d2));
                    mx0 = vmaxf(mx0, lw);
                    mx1 = vmaxf(mx1, st >= p.mtol_hi ? lw : NINF);
                    mx2 = vmaxf(mx2, st >= p.mtol_lo ? lw : NINF);
                    mx8 = lw != lw ? 1.f : mx8;
                }
                {
                    const float lw = sweep(d1, d2);
                    const float st = fmaxf(fabsf(d1) * (-1.f / C10), fabsf(d2));
                    mx3 = vmaxf(mx3, lw);
                    mx4 = vmaxf(mx4, st >= p.mtol_hi ? lw : NINF);
                    mx5 = vmaxf(mx5, st >= p.mtol_lo ? lw : NINF);
                    mx8 = lw != lw ? 1.f : mx8;
                }
            }
            // MLE in scaled units: F = A f, A = 10^(-0.4 mbar), d = D dd
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                float Rj;
                if constexpr (RVF) Rj = A[j];
                else Rj = fmaf(rv, B[j], A[j]);
                const float e = __builtin_amdgcn_exp2f(fmaf(av, Rj, mcC[j]));
                const float fw = e * iv[j];
                num = fmaf(dd[j], fw, num);
                den = fmaf(e, fw, den);
            }
            const float q = __builtin_amdgcn_exp2f(dbar);       // D / A
            const float rden = __builtin_amdgcn_rcpf(den);
            float tt = num * rden;
            float sc = tt * q;
            {
                const bool tiny = sc <= 1e-20f;
                const float tt_lo = 1e-20f * __builtin_amdgcn_rcpf(q);
                tt = tiny ? tt_lo : tt;
                sc = tiny ? 1e-20f : sc;
            }
            const float chi2 = fmaf(tt, fmaf(tt, den, -2.f * num), DD2);
            const float lnl = -0.5f * chi2;
            float lnlp = lnl;
            if (any_par) {
                const float dp = __builtin_amdgcn_sqrtf(sc) - parx;
                lnlp = has_par ? lnl - dp * dp * par_hiv : lnl;
            }
            float lnpr = lnl;
            if (p.dim_prior) lnpr = c0 + c1 * ln_pos(chi2) - 0.5f * chi2;
            if (any_sp) {
                const float vt = sp_var + q * q * rden;
                const float ds = sc - sp_mean;
                const float t = -0.5f * (ds * ds * __builtin_amdgcn_rcpf(vt) + ln_pos(6.2831853071795865f * vt));
                lnpr = sp_on ? lnpr + t : lnpr;
            }
            lnlp = chi2 > eps4 ? lnlp : NAN;
            lnpr = fabsf(lnpr) < 0x1p-100f ? 0.f : lnpr;
            lnpr = chi2 > chi2_lo ? lnpr : NAN;
            o_lnlp[r] = lnlp;
            o_lnpr[r] = lnpr;
            mx9 = (lnlp != lnlp || lnpr != lnpr) ? 1.f : mx9;
            mx6 = vmaxf(mx6, lnlp);
            mx7 = vmaxf(mx7, lnpr);
        }
        // four consecutive models of one star per lane: one 16-byte word per plane where the row
        // is aligned (s nmodel + ib + 4 g: nmodel a multiple of 4), scalars otherwise
        if (sbase >= 0) {
            const int64_t o = sbase + ib + 4 * g;
            if (4 * g + 4 <= nm && (o & 3) == 0) {
                *reinterpret_cast<float4 *>(lnlp32 + o) = make_float4(o_lnlp[0], o_lnlp[1], o_lnlp[2], o_lnlp[3]);
                *reinterpret_cast<float4 *>(lnpr32 + o) = make_float4(o_lnpr[0], o_lnpr[1], o_lnpr[2], o_lnpr[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < nm) {
                        lnlp32[o + r] = o_lnlp[r];
                        lnpr32[o + r] = o_lnpr[r];
                    }
            }
        }
        ps_wave_sync();
    }
    // a star's four lane quarters, then the four waves
    {
        float v[NV32] = {mx0, mx1, mx2, mx3, mx4, mx5, mx6, mx7, mx8, mx9};
#pragma unroll
        for (int k = 0; k < NV32; ++k) {
            v[k] = vmaxf(v[k], __shfl_xor(v[k], 16, 64));
            v[k] = vmaxf(v[k], __shfl_xor(v[k], 32, 64));
            if (g == 0) s_mx[wv][k][sj] = v[k];
        }
    }
    __syncthreads();
    if (wv == 0 && g == 0 && slive) {
#pragma unroll
        for (int k = 0; k < NV32; ++k) {
            float x = s_mx[0][k][sj];
            x = vmaxf(x, s_mx[1][k][sj]);
            x = vmaxf(x, s_mx[2][k][sj]);
            x = vmaxf(x, s_mx[3][k][sj]);
            part[((int64_t)bx * nstar + s) * NV32 + k] = x;
        }
    }
}

}  // namespace
