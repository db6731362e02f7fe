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
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <stdlib.h>

#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/brutus_amd.h"
#include "../../include/brutus_amd_debug.h"

#include "common.hpp"
#include "fastmath.hpp"
#include "grid_kernels.hpp"
#include "fit_kernels.hpp"
#include "fit2_kernels.hpp"
#include "cluster_kernels.hpp"
#include "mt_kernels.hpp"
#include "post_kernels.hpp"
#include "offsets_kernels.hpp"

namespace {

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
// (48, 64: the full-grid pipeline only -- brutus_loglike_batch; the hot path's list kernels hold
// 5 NB values per lane and stop at BRUTUS_MAX_FILT_FIT = 32, where they already run one wave per SIMD)
const int kCompiledNB[] = {8, 12, 16, 24, 32, 48, 64};

int padded_nb(int nfilt) {
    for (int nb : kCompiledNB)
        if (nfilt <= nb) return nb;
    return -1;
}

int64_t pad_models(int64_t n) { return (n + TILE - 1) / TILE * TILE; }

struct Workspace {
    Planes pl;          // brutus_loglike_batch only: the caller's output planes + lnlp, step
    StarPrep *stars;
    double *part;       // per-(tile, star) partial maxima
    double *vmax_lnlp;  // (S,)
    int32_t *k1;        // (S,)
    int32_t *k2;        // (S,)  >=0 active iteration count, <0 done: -(K2)-1
    int32_t *n_unconv;  // (1,)
    int64_t *counts;    // (S, NCHUNK)
    int64_t *offsets;   // (S, NCHUNK)
    // brutus_fit_batch
    int32_t *ids;       // (S,) star list of a launch
    int32_t *ids2;      // (S,) second list (device-driven call: probe list beside the redo list)
    int32_t *ctr;       // (8,) device-driven call: [0] stars to probe, [1] stars to redo, [3] "host path needed";
                        //      both drivers: [4], [5] lengths of the hot (block, star) lists of the two k_top1 launches
    int32_t *hot;       // (nblk2 * S,) hot (block, star) pairs of a k_top1 launch
    int64_t *res;       // brutus_fit_batch: what the host reads at the end of a call, in ONE copy --
                        // totals[4] = selected, derived, candidates, -; then k1 (S), k2 (S),
                        // n_unconv (4), ctr (8) as int32 (w.k1 / w.k2 / w.n_unconv / w.ctr point here)
    int32_t *kfix;      // (S,)
    double *thr_cull, *maxsurv, *thr_sel;
    int32_t *surv_idx;  // (S * nmodel,) worst case: candidate lists, then band queues, then derived lists
    int64_t *surv_off;  // (S + 1,) candidate list offsets; [S] = candidates of the batch
    int64_t *coffsets;  // (S, NCHUNK) candidate list offsets per chunk (kept for k_rec_index)
    int64_t *dcounts, *doffsets;       // (S, NCHUNK) derived lists
    int64_t *der_off;                  // (S + 1,)
    int32_t *wbase_surv, *wbase_der;   // (NCHUNK * S + 1,): chunk-major work items
    ItemGeom *items_surv, *items_der;  // one record per work item
    int32_t *bandn;                    // (S * NCHUNK,) band-queue fill of k_sel_classify
    unsigned long long *mask, *dmask;  // (S, nmodel_pad / 64) selected / selected-and-derived
    unsigned long long *smask;         // candidate bit-mask, same layout
    double *step_st, *lnprob_st;       // (S * nmodel,) worst case: flux-phase step size and final
                                       // first-cut statistic, by candidate-list position
    Star32 *s32;                       // (S,)
    float *lnlp32, *lnpr32;            // (S, nmodel) float32 statistics
    float *part32, *st32;              // (nblk2, S, NV32), (S, NV32)
    int32_t *status, *ids_all;         // (S,)
    double *nomA, *nomB, *candS;       // (S,)
    float *aud;                        // (S,) run-time audit of eps
    StarPrep *stars_tmp;               // (1,) scratch for the deep K1 probe
    size_t part_doubles;
    size_t bytes;
};

// (2 MiB: the granule of the device's large pages -- every array the list kernels stream
// through starts on a page boundary of its own)
size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
size_t align_big(size_t x) { return (x + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1); }

// Lay the workspace out over `base` (may be null: sizing only).  fit = false:
// brutus_loglike_batch (the caller supplies the output planes); fit = true: brutus_fit_batch.
Workspace carve(char *base, int64_t nmodel, int nstar, bool fit) {
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t n) {
        char *p = base ? base + off : nullptr;
        off += align_up(n);
        return p;
    };
    auto take_big = [&](size_t n) {       // plane-sized arrays: absolute 2 MiB alignment
        off = align_big((size_t)base + off) - (size_t)base;
        char *p = base ? base + off : nullptr;
        off += n;
        return p;
    };
    const size_t pairs = (size_t)nstar * (size_t)nmodel;
    const int64_t ntile = pad_models(nmodel) / TILE;
    w.pl.nmodel = nmodel;
    w.stars = (StarPrep *)take(sizeof(StarPrep) * nstar);
    w.part_doubles = (size_t)ntile * (size_t)(nstar * 2 * KCAP > 1024 ? nstar * 2 * KCAP : 1024);
    w.part = (double *)take(sizeof(double) * w.part_doubles);
    w.stars_tmp = (StarPrep *)take(sizeof(StarPrep));
    w.vmax_lnlp = (double *)take(sizeof(double) * nstar);
    w.k1 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.k2 = (int32_t *)take(sizeof(int32_t) * nstar);
    w.n_unconv = (int32_t *)take(sizeof(int32_t) * 4);
    w.counts = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
    w.offsets = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
    if (!fit) {
        w.pl.lnlp = (double *)take_big(sizeof(double) * pairs);
        w.pl.step = (double *)take_big(sizeof(double) * pairs);
    } else {
        w.ids = (int32_t *)take(sizeof(int32_t) * nstar);
        w.ids2 = (int32_t *)take(sizeof(int32_t) * nstar);
        w.res = (int64_t *)take(sizeof(int64_t) * 4 + sizeof(int32_t) * (2 * (size_t)nstar + 12));
        if (base) {
            int32_t *r32 = (int32_t *)(w.res + 4);
            w.k1 = r32;
            w.k2 = r32 + nstar;
            w.n_unconv = r32 + 2 * nstar;
            w.ctr = r32 + 2 * nstar + 4;
        } else {
            w.ctr = nullptr;
        }
        w.kfix = (int32_t *)take(sizeof(int32_t) * nstar);
        w.thr_cull = (double *)take(sizeof(double) * nstar);
        w.maxsurv = (double *)take(sizeof(double) * nstar);
        w.thr_sel = (double *)take(sizeof(double) * nstar);
        w.surv_off = (int64_t *)take(sizeof(int64_t) * (nstar + 1));
        w.der_off = (int64_t *)take(sizeof(int64_t) * (nstar + 1));
        w.coffsets = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
        w.dcounts = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
        w.doffsets = (int64_t *)take(sizeof(int64_t) * nstar * NCHUNK);
        w.wbase_surv = (int32_t *)take(sizeof(int32_t) * ((size_t)NCHUNK * nstar + 1));
        w.wbase_der = (int32_t *)take(sizeof(int32_t) * ((size_t)NCHUNK * nstar + 1));
        {
            const size_t nit = (size_t)nstar * ((size_t)(pad_models(nmodel) / TILE) + NCHUNK);
            w.items_surv = (ItemGeom *)take(sizeof(ItemGeom) * nit);
            w.items_der = (ItemGeom *)take(sizeof(ItemGeom) * nit);
        }
        w.bandn = (int32_t *)take(sizeof(int32_t) * (size_t)NCHUNK * nstar);
        const size_t words = (size_t)nstar * (size_t)(pad_models(nmodel) / 64);
        w.mask = (unsigned long long *)take_big(sizeof(unsigned long long) * words);
        w.dmask = (unsigned long long *)take_big(sizeof(unsigned long long) * words);
        w.smask = (unsigned long long *)take_big(sizeof(unsigned long long) * words);
        const size_t nblk2 = (size_t)(ntile + F2_T - 1) / F2_T;
        w.s32 = (Star32 *)take(sizeof(Star32) * nstar);
        w.part32 = (float *)take(sizeof(float) * nblk2 * nstar * NV32);
        w.hot = (int32_t *)take(sizeof(int32_t) * nblk2 * nstar);
        w.st32 = (float *)take(sizeof(float) * nstar * NV32);
        w.status = (int32_t *)take(sizeof(int32_t) * nstar);
        w.ids_all = (int32_t *)take(sizeof(int32_t) * nstar);
        w.nomA = (double *)take(sizeof(double) * nstar);
        w.nomB = (double *)take(sizeof(double) * nstar);
        w.candS = (double *)take(sizeof(double) * nstar);
        w.aud = (float *)take(sizeof(float) * nstar * 4);
        w.lnlp32 = (float *)take_big(sizeof(float) * pairs);
        w.lnpr32 = (float *)take_big(sizeof(float) * pairs);
        w.surv_idx = (int32_t *)take_big(sizeof(int32_t) * pairs);
        w.step_st = (double *)take_big(sizeof(double) * pairs);
        w.lnprob_st = (double *)take_big(sizeof(double) * pairs);
    }
    w.bytes = align_big(off) + (base ? 0 : (size_t)4 << 20);     // (sizing: room for the base's own offset)
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
    // (development aid: BRUTUS_TRACE_KERNELS=1 waits for every timed section and names it on
    // stderr -- a memory fault then says which kernel it was)
    const bool trace = getenv("BRUTUS_TRACE_KERNELS") != nullptr;
    const char *cur = "";
    void begin(const char *name) {
        cur = name;
        if (trace) fprintf(stderr, "[brutus] %s ...\n", name);
        if (!g_timing) return;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, st);
        ev.push_back({name, {a, b}});
    }
    void end() {
        if (trace) {
            const hipError_t e = hipStreamSynchronize(st);
            fprintf(stderr, "[brutus] %s done (%s)\n", cur, hipGetErrorString(e));
        }
        if (!g_timing) return;
        (void)hipEventRecord(ev.back().second.second, st);
    }
    void collect() {
        if (!g_timing) return;
        g_last_timing.clear();
        for (auto &e : ev) {
            (void)hipEventSynchronize(e.second.second);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e.second.first, e.second.second);
            bool found = false;
            for (auto &t : g_last_timing)
                if (t.name == e.first) {
                    t.ms += ms;
                    t.count += 1;
                    found = true;
                }
            if (!found) g_last_timing.push_back({e.first, ms, 1});
            (void)hipEventDestroy(e.second.first);
            (void)hipEventDestroy(e.second.second);
        }
        ev.clear();
    }
};

struct Workspace;
template <int NB>
int probe_k1_deep(const float *grid, int64_t nmodel, int star, const DevParams &p, int max_iter,
                  Workspace &w, int32_t *k1_out, hipStream_t st, const double *av_init = nullptr,
                  const double *rv_init = nullptr);

template <int NB>
int run_pipeline(const float *grid, int64_t nmodel, int nstar, const DevParams &p,
                 int max_iter, Workspace &w, int32_t *h_k1, int32_t *h_k2,
                 hipStream_t st, Timer &tm, const double *av_init, const double *rv_init) {
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
                           w.stars, p, kmax, w.part, av_init, rv_init);
        tm.end();
        hipLaunchKernelGGL(k_reduce_decide, dim3(nstar), dim3(256), 0, st, 0, ntile, nstar,
                           2 * kmax, w.part, p.ln_init, (double *)nullptr, w.k1, w.n_unconv);
        HIP_TRY(hipMemcpyAsync(&h_unconv, w.n_unconv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (h_unconv == 0) break;
        if (kmax >= max_iter)
            return fail(BRUTUS_ENOCONV, "magnitude phase not converged after %d sweeps for %d star(s)",
                        kmax, h_unconv);
        if (kmax >= KCAP) {     // the few stars that need more: one by one, no cap but max_iter
            std::vector<int32_t> hk(nstar);
            HIP_TRY(hipMemcpyAsync(hk.data(), w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int s = 0; s < nstar; ++s)
                if (hk[s] == 0)
                    if (int rc = probe_k1_deep<NB>(grid, nmodel, s, p, max_iter, w, &hk[s], st, av_init, rv_init))
                        return rc;
            break;
        }
        kmax = kmax * 2 > KCAP ? KCAP : kmax * 2;
    }

    // ---- phase 2: MLE at the converged (Av, Rv); cull statistic -------------
    tm.begin("k_mag_mle");
    hipLaunchKernelGGL(k_mag_mle<NB>, gridA, blk, 0, st, grid, nmodel, nmodel_pad, nstar, w.stars,
                       p, w.k1, w.pl, w.part, av_init, rv_init);
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
                      hipStream_t st, Timer &tm, const double *av_init, const double *rv_init) {
    switch (nb) {
        case 12: return run_pipeline<12>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
#ifndef BRUTUS_DEV_NB12_ONLY      // (tools/ab/build.sh: kernel A/B builds in seconds; never set for the product)
        case 8: return run_pipeline<8>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
        case 16: return run_pipeline<16>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
        case 24: return run_pipeline<24>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
        case 32: return run_pipeline<32>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
        case 48: return run_pipeline<48>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
        case 64: return run_pipeline<64>(grid, nmodel, nstar, p, max_iter, w, h_k1, h_k2, st, tm, av_init, rv_init);
#endif
    }
    return fail(BRUTUS_EINVAL, "unsupported band count %d", nb);
}

// ---- hot path host orchestration (brutus_fit_batch) ----------------------------------
thread_local int t_audit_call = 0;        // dispatch_fit: this call's float32 bound is audited and enforced
constexpr int BRUTUS_RETRY_HOSTDRIVEN = -1000;     // internal: never leaves dispatch_fit
std::atomic<long long> g_fit_calls{0}, g_fit_retries{0};     // device-driven calls / repeated host-driven
constexpr int FS_TILES_PER_BLOCK = 8;
constexpr int PERSIST_BLOCKS = 4096;

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
double env_double(const char *name, double dflt) {
    const char *v = getenv(name);
    return v && *v ? atof(v) : dflt;
}

// nact == 0: the opening launch (all stars); else a continuation over the `nact` stars listed
// in w.ids (device) that are still iterating.
// nact < 0: a continuation whose list (w.ids) and length (*nact_dev) were written by
// k_fflux_decide on the device -- a launch of fixed size.
constexpr int CONT_BLOCKS = 2048;
template <int NB, bool RVF>
void launch_fflux(hipStream_t st, int nact, const float *grid, int64_t nmodel, int64_t nmodel_pad,
                  int nstar, const DevParams &p, const Workspace &w, const RecPlanes &rec,
                  const int32_t *nact_dev = nullptr) {
    if (nact == 0)
        hipLaunchKernelGGL((k_fflux<NB, RVF, true>), dim3(PERSIST_BLOCKS), dim3(TILE), 0, st, grid,
                           nmodel, nmodel_pad, nstar, w.stars, p, w.k1, w.k2, w.surv_idx, w.surv_off,
                           w.wbase_surv, w.items_surv, rec, w.step_st, w.lnprob_st, w.part, w.lnpr32,
                           w.thr_cull, (const int32_t *)nullptr, 0, (const int32_t *)nullptr);
    else if (nact > 0)
        hipLaunchKernelGGL((k_fflux<NB, RVF, false>), dim3(NCHUNK * nact * CONT_P), dim3(TILE), 0, st,
                           grid, nmodel, nmodel_pad, nstar, w.stars, p, w.k1, w.k2, w.surv_idx,
                           w.surv_off, w.wbase_surv, w.items_surv, rec, w.step_st, w.lnprob_st, w.part,
                           w.lnpr32, w.thr_cull, w.ids, nact, (const int32_t *)nullptr);
    else       // (nact = -r, the r-th continuation: a per cent of the stars reach the first, fewer every round)
        hipLaunchKernelGGL((k_fflux<NB, RVF, false>), dim3(nact == -1 ? CONT_BLOCKS : CONT_BLOCKS / 8), dim3(TILE), 0, st,
                           grid, nmodel, nmodel_pad, nstar, w.stars, p, w.k1, w.k2, w.surv_idx,
                           w.surv_off, w.wbase_surv, w.items_surv, rec, w.step_st, w.lnprob_st, w.part,
                           w.lnpr32, w.thr_cull, w.ids, 0, nact_dev);
}

// Exact K1 of the stars in `ids` by probing KS = 8 sweeps in float64 (k1 = 0: more needed).
// `ids` = nullptr: the probe list (w.ids2, length w.ctr[0]) was put together on the device by
// k_pre_decide; k_k1_decide then appends to the redo list (w.ids, length w.ctr[1]) itself.
template <int NB, bool RVF>
int launch_k1probe(const float *grid, int64_t nmodel, int nstar, const std::vector<int32_t> *ids,
                   const DevParams &p, int max_iter, Workspace &w, hipStream_t st, Timer &tm) {
    constexpr int KS = 8;
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    const int nblkx = (ntile + FS_TILES_PER_BLOCK - 1) / FS_TILES_PER_BLOCK;
    if (ids) {
        const int nrun = (int)ids->size();
        HIP_TRY(hipMemcpyAsync(w.ids, ids->data(), sizeof(int32_t) * nrun, hipMemcpyHostToDevice, st));
        tm.begin("k_k1probe");
        hipLaunchKernelGGL((k_k1probe<NB, KS, RVF>), dim3(nblkx, nrun), dim3(TILE), 0, st, grid, nmodel,
                           nmodel_pad, nstar, nrun, w.ids, w.stars, p, FS_TILES_PER_BLOCK, ntile, w.part,
                           (const int32_t *)nullptr);
        tm.end();
        hipLaunchKernelGGL(k_k1_decide, dim3(nrun), dim3(256), 0, st, nblkx, nstar, w.ids, KS, w.part,
                           p.ln_init, w.k1, (const int32_t *)nullptr, (int32_t *)nullptr,
                           (int32_t *)nullptr, (int32_t *)nullptr, RVF ? 1 : 0, max_iter);
    } else {
        tm.begin("k_k1probe");
        hipLaunchKernelGGL((k_k1probe<NB, KS, RVF>), dim3(nblkx, nstar < 8 ? nstar : 8), dim3(TILE), 0, st,
                           grid, nmodel, nmodel_pad, nstar, 0, w.ids2, w.stars, p, FS_TILES_PER_BLOCK,
                           ntile, w.part, w.ctr + 0);
        tm.end();
        hipLaunchKernelGGL(k_k1_decide, dim3(nstar), dim3(256), 0, st, nblkx, nstar, w.ids2, KS, w.part,
                           p.ln_init, w.k1, w.ctr + 0, w.kfix, w.ctr, w.ids, RVF ? 1 : 0, max_iter);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Exact number of magnitude sweeps of ONE star by probing kmax = 16, 32, ... sweeps
// with the residual-carrying kernels (no cap but max_iter; fitting.py:173-264).
template <int NB>
int probe_k1_deep(const float *grid, int64_t nmodel, int star, const DevParams &p, int max_iter,
                  Workspace &w, int32_t *k1_out, hipStream_t st, const double *av_init,
                  const double *rv_init) {
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    HIP_TRY(hipMemcpyAsync(w.stars_tmp, w.stars + star, sizeof(StarPrep), hipMemcpyDeviceToDevice, st));
    const int cap = (int)(w.part_doubles / ((size_t)ntile * 2));
    for (int kmax = 16;; kmax *= 2) {
        if (kmax > max_iter) kmax = max_iter;
        if (kmax > cap) kmax = cap;
        hipLaunchKernelGGL(k_mag_stats<NB>, dim3(ntile, 1), dim3(TILE), 0, st, grid, nmodel,
                           nmodel_pad, 1, w.stars_tmp, p, kmax, w.part, av_init, rv_init);
        hipLaunchKernelGGL(k_k1_deep_decide, dim3(1), dim3(256), 0, st, ntile, kmax, w.part,
                           p.ln_init, w.k1 + star);
        HIP_TRY(hipMemcpyAsync(k1_out, w.k1 + star, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (*k1_out > 0) return 0;
        if (kmax >= max_iter || kmax >= cap)
            return fail(BRUTUS_ENOCONV, "magnitude phase of star %d not converged after %d sweeps",
                        star, kmax);
    }
}

// mode 0: host-driven (`ids`, `kfix` from the host; k_pre_decide leaves k1 / status for the host)
// mode 1: device-driven opening pass over all stars (`ids`, `kfix` uploaded; k_pre_decide puts
//         the probe list w.ids2 / w.ctr[0] and the redo list w.ids / w.ctr[1] together)
// mode 2: device-driven re-run over the redo list as it stands on the device (accept = 1)
template <int NB, bool RVF>
int launch_pre32(const float *grid, int64_t nmodel, int nfilt, int nstar,
                 const std::vector<int32_t> &ids, const std::vector<int32_t> &kfix,
                 const DevParams &p, Workspace &w, int accept, hipStream_t st, Timer &tm,
                 int mode = 0) {
    constexpr int G = 4;
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    const int nblkx = (ntile + F2_T - 1) / F2_T;
    const int nrun = mode == 2 ? nstar : (int)ids.size();
    const int32_t *list = mode == 1 ? w.ids_all : w.ids;
    if (mode == 0) HIP_TRY(hipMemcpyAsync(w.ids, ids.data(), sizeof(int32_t) * nrun, hipMemcpyHostToDevice, st));
    // (mode 1: kfix = 2 for every star, set on the device by k_prep32)
    if (mode == 0) HIP_TRY(hipMemcpyAsync(w.kfix, kfix.data(), sizeof(int32_t) * nstar, hipMemcpyHostToDevice, st));
    const int32_t *nrun_dev = mode == 2 ? w.ctr + 1 : nullptr;
    P32 q;
    q.avmin = (float)p.avmin;
    q.avmax = (float)p.avmax;
    q.rvmin = (float)p.rvmin;
    q.rvmax = (float)p.rvmax;
    q.av_mean = (float)p.av_mean;
    q.av_ivar = (float)p.av_ivar;
    q.rv_mean = (float)p.rv_mean;
    q.rv_ivar = (float)p.rv_ivar;
    q.mtol_hi = (float)(p.mtol * 1.002 + 1e-4);
    q.mtol_lo = (float)(p.mtol * 0.998 - 1e-4);
    q.dim_prior = p.dim_prior;
    q.nfilt = nfilt;
    // Long star lists take the star-lane pass (pre32s_kernels.hpp: lane = star, the models' rows
    // broadcast from LDS); short ones -- the re-run over a handful of stars, lists with other
    // sweep counts than the opening pass's two -- the tile pass.
    // (development / test switches, read per call: the tests flip them inside one process)
    const int use_mfma = env_int("BRUTUS_PRE32_MFMA", 0);     // (measured, not the default: pre32m_kernels.hpp)
    const int min_stars = env_int("BRUTUS_PRE32_STAR_LANES_MIN", 32);
    bool lanes_are_stars = brutus_i_pre32s_bands(NB) && mode != 2 && nrun >= min_stars &&
                           env_int("BRUTUS_PRE32_STAR_LANES", 1) != 0;
    // (two translation units, one definition of the shared layout: pre32_types.hpp)
    if (brutus_i_pre32_layout(0) != (int)sizeof(Star32) || brutus_i_pre32_layout(1) != (int)sizeof(P32) ||
        brutus_i_pre32_layout(2) != F2_T || brutus_i_pre32_layout(3) != TILE || brutus_i_pre32_layout(4) != NV32)
        return fail(BRUTUS_EINVAL, "float32 pass: the library's translation units disagree on the Star32 / "
                                   "tile layout (built with different flags?)");
    if (lanes_are_stars && !RVF)
        for (int k = 0; k < nrun && lanes_are_stars; ++k) lanes_are_stars = kfix[ids[k]] == 2;
    if (lanes_are_stars) {
        tm.begin("k_pre32");
        if (brutus_i_pre32s_launch(NB, use_mfma, RVF ? 1 : 0, grid, nmodel, nmodel_pad, nstar, nrun, list, w.s32, &q,
                                   w.lnlp32, w.lnpr32, w.part32, st))
            return fail(BRUTUS_EHIP, "star-lane float32 pass: launch failed");
        tm.end();
    } else {
    tm.begin("k_pre32");
    hipLaunchKernelGGL((k_pre32<NB, RVF, G>), dim3(8 * ((nblkx + 7) / 8) * ((nrun + G - 1) / G)),
                       dim3(TILE), 0, st, grid,
                       nmodel, nmodel_pad, nstar, nrun, list, w.s32, q, w.kfix, ntile, w.lnlp32,
                       w.lnpr32, w.part32, nrun_dev);
    tm.end();
    }
    hipLaunchKernelGGL(k_pre_decide, dim3(nrun), dim3(256), 0, st, nblkx, nstar, list, w.part32,
                       w.s32, (float)p.ln_init, RVF ? 1 : 0, w.kfix, accept, w.st32, w.k1, w.status,
                       w.nomA, mode == 1 ? w.ctr : (int32_t *)nullptr, w.ids2, w.ids, nrun_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// h_counts[0] = selected models of the batch (= d_rec_off[nstar]), [1] = candidates of the
// cull (the record slots the flux phase owns), [2] = record slots needed in all.
template <int NB, bool RVF>
int run_fit(const float *grid, int64_t nmodel, int nfilt, int nstar, const DevParams &p,
            int max_iter, Workspace &w, int64_t capacity, int32_t *d_rec_idx, int32_t *d_rec_slot,
            double *d_rec_vals, int64_t *d_rec_off, int32_t *h_k1, int32_t *h_k2, int64_t *h_counts,
            hipStream_t st, Timer &tm, bool device_driven) {
    constexpr int G = 4;
    const int64_t nmodel_pad = pad_models(nmodel);
    const int ntile = (int)(nmodel_pad / TILE);
    const int nblkx = (ntile + F2_T - 1) / F2_T;
    const dim3 blk(TILE);
    const RecPlanes rec{d_rec_vals, capacity};
    std::vector<int32_t> ids(nstar), kfix(nstar, 2), k1(nstar, 0), status(nstar, 0);
    for (int s = 0; s < nstar; ++s) ids[s] = s;
    // the run-time audit of the float32 bound: every call with BRUTUS_AUDIT=1 (recorded for the
    // caller to read: tests, tools/fuzz_fit.py), and ENFORCED on the calls dispatch_fit picks
    // (the first of the process and every BRUTUS_AUDIT_EVERY-th after it, audit_verdict below)
    const int audit_on = env_int("BRUTUS_AUDIT", 0) != 0 || t_audit_call;
    float *aud = audit_on ? w.aud : nullptr;
    if (aud) HIP_TRY(hipMemsetAsync(w.aud, 0, sizeof(float) * nstar * 4, st));
    h_counts[0] = h_counts[1] = h_counts[2] = 0;

    // ---- float32 pass over the whole grid; K1 where float32 can decide it ----------
    // (k_prep32 also starts the call's small device state: w.ids_all = 0, 1, ..., kfix = k2 = 2,
    // n_unconv and ctr zero -- they lie side by side in the result block)
    hipLaunchKernelGGL(k_prep32, dim3(nstar), dim3(64), 0, st, nstar, w.stars,
                       (float)env_double("BRUTUS_EPS_SCALE", 1.0), p.dim_prior, w.s32,
                       CallInit{w.ids_all, w.kfix, w.k2, w.n_unconv, 12});
    // Two drivers for the same kernels.  DEVICE-DRIVEN (default): which stars need the exact K1
    // probe, which need their float32 planes redone and which iterate on in the flux phase is
    // decided and listed ON THE DEVICE (k_pre_decide, k_k1_decide, k_fflux_decide), the
    // follow-up launches are issued unconditionally with a size that fits any list (their
    // surplus workgroups leave at once), and the host sees the call once, at its end.  What
    // that cannot express -- a star that needs more than the eight probed sweeps, a flux phase
    // longer than FLUX_ROUNDS continuations -- raises a flag, and the batch is done again by
    // the HOST-DRIVEN driver below (round 3's: a host decision after the float32 pass and after
    // every flux launch; each one idles the stream for a round trip).
    const int FLUX_ROUNDS = env_int("BRUTUS_FLUX_ROUNDS", 4);       // (development switch)
    if (device_driven) {
        if (int rc = launch_pre32<NB, RVF>(grid, nmodel, nfilt, nstar, ids, kfix, p, w, 0, st, tm, 1)) return rc;
        if (int rc = launch_k1probe<NB, RVF>(grid, nmodel, nstar, nullptr, p, max_iter, w, st, tm)) return rc;
        if (!RVF)       // (pinned Rv: the planes never depend on the sweep count)
            if (int rc = launch_pre32<NB, RVF>(grid, nmodel, nfilt, nstar, ids, kfix, p, w, 1, st, tm, 2)) return rc;
    } else {
    if (int rc = launch_pre32<NB, RVF>(grid, nmodel, nfilt, nstar, ids, kfix, p, w, 0, st, tm)) return rc;
    HIP_TRY(hipMemcpyAsync(k1.data(), w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(status.data(), w.status, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    std::vector<int32_t> probe, redo;
    for (int s = 0; s < nstar; ++s) {
        if (status[s] == 1) redo.push_back(s);
        if (status[s] == 2) probe.push_back(s);
    }
    if (!probe.empty()) {   // float32 could not decide: exact probe (float64, up to 8 sweeps, then deeper)
        if (int rc = launch_k1probe<NB, RVF>(grid, nmodel, nstar, &probe, p, max_iter, w, st, tm)) return rc;
        HIP_TRY(hipMemcpyAsync(k1.data(), w.k1, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int s : probe) {
            if (k1[s] == 0)
                if (int rc = probe_k1_deep<NB>(grid, nmodel, s, p, max_iter, w, &k1[s], st)) return rc;
            if (k1[s] > max_iter)
                return fail(BRUTUS_ENOCONV, "magnitude phase of star %d needs %d sweeps (max_iter %d)", s,
                            k1[s], max_iter);
            kfix[s] = k1[s];
            if (!RVF && k1[s] != 2) redo.push_back(s);
        }
    }
    for (int s : redo) kfix[s] = k1[s];
    if (!redo.empty())      // float32 statistics at the state after K1 sweeps
        if (int rc = launch_pre32<NB, RVF>(grid, nmodel, nfilt, nstar, redo, kfix, p, w, 1, st, tm)) return rc;
    }

    // ---- exact cull threshold ---------------------------------------------------------
    // (k_hot_list + k_top1: the hot (block, star) pairs as a list; k_top itself -- every pair
    // gets a workgroup, nearly all of which leave at once -- stays behind BRUTUS_TOP_LIST=0)
    const bool top_list = env_int("BRUTUS_TOP_LIST", 1) != 0;
    constexpr int TOP_BLOCKS = 1024;
    tm.begin("k_top");
    if (top_list) {
        hipLaunchKernelGGL(k_hot_list, dim3(nstar), dim3(256), 0, st, nblkx, nstar, 0, w.part32, w.nomA,
                           (const double *)nullptr, w.s32, (double *)nullptr, w.part, w.hot, w.ctr + 4);
        hipLaunchKernelGGL((k_top1<NB, RVF>), dim3(TOP_BLOCKS), blk, 0, st, grid, nmodel, nmodel_pad, nstar,
                           w.stars, p, w.k1, ntile, 0, w.lnlp32, w.nomA, w.hot, w.ctr + 4, w.part, aud);
    } else
    hipLaunchKernelGGL((k_top<NB, RVF, G>), dim3(nblkx, (nstar + G - 1) / G), blk, 0, st, grid, nmodel,
                       nmodel_pad, nstar, nstar, w.ids_all, w.stars, p, w.k1, ntile, 0, w.lnlp32,
                       w.nomA, (const float *)nullptr, w.part32, w.part, aud);
    tm.end();
    hipLaunchKernelGGL(k_top_decide, dim3(nstar), dim3(256), 0, st, nblkx, nstar, w.ids_all, 0, w.part,
                       w.s32, p.ln_init, (const double *)nullptr, w.thr_cull, w.candS);

    // ---- candidates (lnl_p~ >= thr_cull - eps) as ordered lists --------------------------
    tm.begin("k_surv_compact");
    hipLaunchKernelGGL(k_cmp_count32, dim3(NCHUNK, nstar), blk, 0, st, nmodel, ntile, w.lnlp32, w.candS,
                       w.counts, w.smask);
    {
        const OffsetsJob job{w.counts, w.coffsets, w.surv_off, w.wbase_surv, w.res + 2};
        hipLaunchKernelGGL(k_offsets, dim3(2), dim3(OFF_T), 0, st, nstar, job, job);
    }
    hipLaunchKernelGGL(k_items, dim3((NCHUNK * nstar + 255) / 256), dim3(256), 0, st, nstar, w.wbase_surv,
                       w.coffsets, w.surv_off, w.items_surv);
    hipLaunchKernelGGL(k_cmp_scatter, dim3(NCHUNK, nstar), blk, 0, st, nmodel, ntile, w.smask,
                       w.coffsets, (int64_t)nstar * nmodel, w.surv_idx);
    tm.end();
    int64_t h_ncand = 0;
    if (!device_driven)
        HIP_TRY(hipMemcpyAsync(&h_ncand, w.surv_off + nstar, sizeof(int64_t), hipMemcpyDeviceToHost, st));

    // ---- exact cull test + flux phase on the candidates; results into the record planes ------
    int32_t h_unconv = 0;
    int iter = 2;
    std::vector<int32_t> k2s(nstar), act;
    if (device_driven) {
        // opening launch + FLUX_ROUNDS continuations; round r's decision counts and lists the
        // stars that iterate on in w.n_unconv[r & 1] / w.ids, the next launch reads them there
        // (both counters start at zero, k_prep32; round r's decision zeroes the one round r + 1 adds to)
        for (int r = 0; r <= FLUX_ROUNDS; ++r) {
            tm.begin(r == 0 ? "k_fflux" : "k_fflux_cont");
            launch_fflux<NB, RVF>(st, r == 0 ? 0 : -r, grid, nmodel, nmodel_pad, nstar, p, w, rec,
                                  w.n_unconv + ((r - 1) & 1));
            tm.end();
            hipLaunchKernelGGL(k_fflux_decide, dim3(nstar), dim3(256), 0, st, nstar, w.wbase_surv, w.part,
                               p.ln_sub, w.k2, w.maxsurv, w.n_unconv + (r & 1), w.ids,
                               w.n_unconv + ((r + 1) & 1));
        }
        // (stars still iterating at the end: w.n_unconv[FLUX_ROUNDS & 1], read with the result block)
    } else
    for (int first = 1;; first = 0) {
        HIP_TRY(hipMemsetAsync(w.n_unconv, 0, sizeof(int32_t), st));
        tm.begin(first ? "k_fflux" : "k_fflux_cont");
        launch_fflux<NB, RVF>(st, (int)act.size(), grid, nmodel, nmodel_pad, nstar, p, w, rec);
        tm.end();
        hipLaunchKernelGGL(k_fflux_decide, dim3(nstar), dim3(256), 0, st, nstar, w.wbase_surv, w.part,
                           p.ln_sub, w.k2, w.maxsurv, w.n_unconv, (int32_t *)nullptr, (int32_t *)nullptr);
        HIP_TRY(hipMemcpyAsync(&h_unconv, w.n_unconv, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(k2s.data(), w.k2, sizeof(int32_t) * nstar, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (first && h_ncand > capacity) {
            // the flux phase keeps its results in the record planes: nothing to go on with
            h_counts[1] = h_ncand;
            h_counts[2] = 2 * h_ncand;       // (the derived records come on top; a guess)
            return fail(BRUTUS_ENOMEM, "record buffer too small: %lld candidate slots, capacity %lld",
                        (long long)h_ncand, (long long)capacity);
        }
        if (h_unconv == 0) break;
        if (iter >= max_iter)
            return fail(BRUTUS_ENOCONV, "flux phase not converged after %d iterations for %d star(s)",
                        iter, h_unconv);
        ++iter;
        act.clear();                 // the stars still iterating: the next launch walks their segments only
        for (int s = 0; s < nstar; ++s)
            if (k2s[s] >= 0) act.push_back(s);
        HIP_TRY(hipMemcpyAsync(w.ids, act.data(), sizeof(int32_t) * act.size(), hipMemcpyHostToDevice, st));
    }

    // ---- exact first-cut threshold, selection masks ---------------------------------------
    tm.begin("k_top");
    if (top_list) {
        hipLaunchKernelGGL(k_hot_list, dim3(nstar), dim3(256), 0, st, nblkx, nstar, 1, w.part32,
                           (const double *)nullptr, w.maxsurv, w.s32, w.nomB, w.part, w.hot, w.ctr + 5);
        hipLaunchKernelGGL((k_top1<NB, RVF>), dim3(TOP_BLOCKS), blk, 0, st, grid, nmodel, nmodel_pad, nstar,
                           w.stars, p, w.k1, ntile, 1, w.lnpr32, w.nomB, w.hot, w.ctr + 5, w.part,
                           aud ? aud + nstar : nullptr);
    } else {
    hipLaunchKernelGGL(k_nomB, dim3((nstar + 63) / 64), dim3(64), 0, st, nstar, w.maxsurv, w.s32, w.nomB);
    hipLaunchKernelGGL((k_top<NB, RVF, G>), dim3(nblkx, (nstar + G - 1) / G), blk, 0, st, grid, nmodel,
                       nmodel_pad, nstar, nstar, w.ids_all, w.stars, p, w.k1, ntile, 1, w.lnpr32,
                       w.nomB, w.lnpr32, w.part32, w.part, aud ? aud + nstar : nullptr);
    }
    tm.end();
    hipLaunchKernelGGL(k_top_decide, dim3(nstar), dim3(256), 0, st, nblkx, nstar, w.ids_all, 1, w.part,
                       w.s32, p.ln_wt, w.maxsurv, w.thr_sel, (double *)nullptr);
    tm.begin("k_sel_classify");
    hipLaunchKernelGGL(k_sel_classify, dim3(NCHUNK, nstar), blk, 0, st, nmodel, ntile, w.s32,
                       w.lnpr32, w.lnprob_st, w.surv_off, w.thr_sel, w.counts, w.mask, w.dcounts,
                       w.dmask, w.surv_idx, w.bandn);
    tm.end();
    tm.begin("k_sel_band");      // (the candidate lists in surv_idx are no longer needed)
    hipLaunchKernelGGL((k_sel_band<NB, RVF>), dim3(NCHUNK / SB_C, nstar, SB_Z), blk, 0, st, grid, nmodel, nmodel_pad,
                       ntile, w.stars, p, w.k1, w.lnpr32, w.thr_sel, w.surv_idx, w.bandn, w.counts,
                       w.mask, w.dcounts, w.dmask, aud ? aud + 2 * nstar : nullptr);
    tm.end();

    // ---- record index (model, slot) in np.where order; values of the derived records -------
    tm.begin("k_select");
    {
        const OffsetsJob sel{w.counts, w.offsets, d_rec_off, nullptr, w.res + 0};
        const OffsetsJob der{w.dcounts, w.doffsets, w.der_off, w.wbase_der, w.res + 1};
        hipLaunchKernelGGL(k_offsets, dim3(4), dim3(OFF_T), 0, st, nstar, sel, der);
    }
    hipLaunchKernelGGL(k_items, dim3((NCHUNK * nstar + 255) / 256), dim3(256), 0, st, nstar, w.wbase_der,
                       w.doffsets, w.der_off, w.items_der);
    // (the band queues in surv_idx are no longer needed either: it now takes the derived lists)
    hipLaunchKernelGGL(k_rec_index, dim3(NCHUNK, nstar), blk, 0, st, nmodel, ntile, nstar, w.mask, w.dmask,
                       w.smask, w.offsets, w.doffsets, w.coffsets, w.surv_off, capacity, d_rec_idx,
                       d_rec_slot, w.surv_idx);
    tm.end();
    tm.begin("k_derive");
    hipLaunchKernelGGL((k_derive<NB, RVF>), dim3(PERSIST_BLOCKS), blk, 0, st, grid, nmodel_pad, nstar,
                       w.stars, p, w.k1, w.surv_idx, w.wbase_der, w.items_der, w.surv_off, rec);
    tm.end();
    // everything the host wants to know, in one copy: totals, K1, K2, the counters
    int64_t h_nder = 0;
    int32_t h_ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h_left = 0;
    std::vector<int64_t> h_res(4 + (2 * (size_t)nstar + 12 + 1) / 2);
    HIP_TRY(hipMemcpyAsync(h_res.data(), w.res, sizeof(int64_t) * 4 + sizeof(int32_t) * (2 * (size_t)nstar + 12),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    {
        const int32_t *r32 = reinterpret_cast<const int32_t *>(h_res.data() + 4);
        h_counts[0] = h_res[0];
        h_nder = h_res[1];
        if (device_driven) h_ncand = h_res[2];
        if (h_k1) memcpy(h_k1, r32, sizeof(int32_t) * nstar);
        if (h_k2) memcpy(h_k2, r32 + nstar, sizeof(int32_t) * nstar);
        if (device_driven) {
            memcpy(h_ctr, r32 + 2 * nstar + 4, sizeof(h_ctr));
            h_left = r32[2 * nstar + (FLUX_ROUNDS & 1)];
        }
    }
    if (device_driven && h_ncand > capacity) {      // (every write was bounded by the capacity)
        h_counts[1] = h_ncand;
        h_counts[2] = h_ncand + (h_nder > 0 ? h_nder : h_ncand);
        return fail(BRUTUS_ENOMEM, "record buffer too small: %lld candidate slots, capacity %lld",
                    (long long)h_ncand, (long long)capacity);
    }
    if (device_driven && (h_ctr[3] != 0 || h_left != 0)) return BRUTUS_RETRY_HOSTDRIVEN;
    h_counts[1] = h_ncand;
    h_counts[2] = h_ncand + h_nder;
    if (h_counts[2] > capacity)
        return fail(BRUTUS_ENOMEM, "record buffer too small: %lld slots needed, capacity %lld",
                    (long long)h_counts[2], (long long)capacity);
    return 0;
}


// ---- numpy stream on many workgroups (mt_kernels.hpp, second half) ----------------
std::mutex g_mt_mu;
std::vector<uint32_t> g_mt_polys;        // (1 + R) x 624 words: strides MT_J, MT_L1 * MT_J * 2^r

struct MtPlanStream {
    int o0, o1;            // objects (global indices)
    int64_t T, K, base, bit_base;
    int pos0;
};

// slots to generate for a stream whose objects need `a` accepted pairs in total
int64_t mt_slots_for(int64_t pairs, int64_t nobj, int nuni) {
    const double need = 1.2733 * (double)pairs * 1.01 + 50. * sqrt((double)pairs + 1.) + 4096.;
    int64_t T = (int64_t)need + nobj * (int64_t)(nuni / 2 + 8);
    return (T + MT_SB - 1) / MT_SB * MT_SB;
}

// device bytes the parallel walk of the given streams needs besides Z
size_t mt_scratch_bytes(const std::vector<MtPlanStream> &ps, int nobj_total) {
    size_t b = 4096;
    int64_t K = 0, T = 0;
    for (const auto &p : ps) {
        K += p.K;
        T += p.T;
    }
    const size_t ns = ps.size();
    b += 16 * MT_N * 4 + 256;
    b += (size_t)K * MT_N * 4 + 256;                 // windows
    b += (size_t)K * sizeof(MtSub) + 256;
    b += (size_t)(K + ns) * (8 + 8 + 4) * 2 + 1024;  // chain arrays (two levels)
    b += (size_t)T / 8 + 256;                        // bitmap
    b += (size_t)T / MT_SB * 4 + 256;                // cnt
    b += ((size_t)T / MT_SB + ns + 1) * 8 + 256;     // pre
    b += ns * 128 + 4096;                            // per-stream arrays
    b += (size_t)nobj_total * sizeof(MtObj) + 256;
    b += ((size_t)K + nobj_total) * 16 + 512;        // segment lists (k_mt_segments)
    b += (size_t)nobj_total * 24 + 1024 + (ns + 1) * 8;
    return b;
}

// Walk the streams of one group with many workgroups.  Returns 0, a negative error, or 1
// if the generated slots did not suffice / the shape is not supported (caller then uses
// the sequential k_mt_stream; nothing has been modified).
// Page-locked staging for the plan arrays of a walk: one host -> device copy instead of a
// dozen small ones from pageable vectors (each of which queues behind whatever long kernel
// another stream has on the device when the phases of two batches overlap).
struct PinnedBuf {
    char *p = nullptr;
    size_t cap = 0;
    char *get(size_t n) {
        if (n > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = (n + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
            if (hipHostMalloc((void **)&p, want, hipHostMallocDefault) != hipSuccess) return nullptr;
            cap = want;
        }
        return p;
    }
    // (never freed at thread / process exit: the HIP runtime may be gone by then)
};
thread_local PinnedBuf g_plan_pin;

// brutus_post_set_after_jump: a caller's hook, fired once from the calling thread when the
// jump-ahead windows of its next numpy-stream call are complete (the stream is drained
// first) -- or, if that call takes no jump, at the latest before it returns.
struct AfterJump {
    void (*fn)(void *);
    void *arg;
};
thread_local AfterJump g_after_jump{nullptr, nullptr};
void fire_after_jump(hipStream_t st) {
    if (!g_after_jump.fn) return;
    const AfterJump h = g_after_jump;
    g_after_jump = AfterJump{nullptr, nullptr};
    (void)hipStreamSynchronize(st);
    h.fn(h.arg);
}

// k_mt_emit of a walk whose caller asked for it to be deferred (phase 1 of
// brutus_post_batch_numpy_phase): everything it reads stays in the caller's buffers.
struct MtEmitLaunch {
    int Ktot;
    const MtSub *subs;
    const uint32_t *win;
    const unsigned long long *bits;
    const int64_t *bitbase, *sblo, *pre;
    const int32_t *seg;
    const MtObj *objs;
    const int64_t *nnorm, *zoff;
    double *Z;
    int nuni;
    double *U, *endgauss;
    int uni_only;        // the normals were written by pass 1: only the uniform slots are left
    ZMap zm;             // ... and this is how the consumers find them (zm.zloc == nullptr: flat Z)
    // k_mt_segments (runs with the emit: the consumers' half of the call)
    int nstream, nobj;
    const double *gauss0;
    const int64_t *subbase;
};
std::mutex g_emit_mu;
std::map<const void *, MtEmitLaunch> g_emit;      // key: the scratch base

void launch_mt_emit(const MtEmitLaunch &e, hipStream_t st, Timer &tm) {
    tm.begin("k_mt_emit");
    if (e.uni_only)
        hipLaunchKernelGGL(k_mt_segments, dim3((unsigned)e.nobj), dim3(64), 0, st, e.nstream, e.seg, e.nnorm,
                           e.gauss0, e.subs, e.subbase, e.bitbase, e.sblo, e.bits, e.pre, e.objs,
                           e.zm.zloc, const_cast<int64_t *>(e.zm.seg_pair0),
                           const_cast<int64_t *>(e.zm.seg_addr), const_cast<int64_t *>(e.zm.seg_lo),
                           const_cast<int32_t *>(e.zm.nseg), const_cast<double *>(e.zm.cached),
                           const_cast<int32_t *>(e.zm.c));
    hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)e.Ktot), dim3(MT_PT), 0, st, e.Ktot, e.subs, e.win,
                       e.bits, e.bitbase, e.sblo, e.pre, e.seg, e.objs, e.nnorm, e.zoff, e.Z, e.nuni,
                       e.U, e.endgauss, e.uni_only);
    tm.end();
}

int mt_walk_parallel(int nstream, const std::vector<int32_t> &seg, uint32_t *d_states,
                     const std::vector<int> &pos0, const std::vector<int64_t> &nnorm,
                     const int32_t *d_seg, const int64_t *d_nnorm, const int64_t *d_zoff, double *d_Z,
                     int nuni, double *d_U, char *scratch, size_t scratch_bytes, int nobj_total,
                     hipStream_t st, Timer &tm, bool defer_emit, double2 *d_zloc, size_t zloc_pairs,
                     MtEmitLaunch *out) {
    if (nuni & 1) return 1;                      // slot grid needs an even number of uniforms
    std::vector<uint32_t> polys;
    {
        std::lock_guard<std::mutex> lk(g_mt_mu);
        polys = g_mt_polys;
    }
    if (polys.size() < 2 * MT_N) return 1;
    const int nlev = (int)(polys.size() / MT_N) - 1;       // first-level strides 128 J 2^r, r < nlev
    std::vector<MtPlanStream> ps(nstream);
    int64_t Ktot = 0, Ttot = 0;
    for (int g = 0; g < nstream; ++g) {
        MtPlanStream &p = ps[g];
        p.o0 = seg[g];
        p.o1 = seg[g + 1];
        p.pos0 = pos0[g];
        int64_t pairs = 0;
        for (int o = p.o0; o < p.o1; ++o) pairs += (nnorm[o] + 1) / 2;
        p.T = mt_slots_for(pairs, p.o1 - p.o0, nuni);
        p.K = (p.pos0 + 4 * p.T + MT_J - 1) / MT_J;
        if (p.K < 1) p.K = 1;
        p.base = Ktot;
        p.bit_base = Ttot;
        Ktot += p.K;
        Ttot += p.T;
    }
    if (mt_scratch_bytes(ps, nobj_total) > scratch_bytes) return 1;
    // one walk: pass 1 also writes the accepted candidates' normals (16 bytes per slot)
    const bool mapped = d_zloc && (size_t)Ttot <= zloc_pairs;
    // ---- carve ---------------------------------------------------------------------
    size_t off = 0;
    auto take = [&](size_t n) {
        char *q = scratch + off;
        off += (n + 255) & ~(size_t)255;
        return q;
    };
    uint32_t *d_win = (uint32_t *)take((size_t)Ktot * MT_N * 4);
    unsigned long long *d_bits = (unsigned long long *)take((size_t)Ttot / 8);
    const int64_t nsb = Ttot / MT_SB;
    uint32_t *d_cnt = (uint32_t *)take((size_t)nsb * 4);
    int64_t *d_pre = (int64_t *)take((size_t)(nsb + nstream + 1) * 8);
    // read back together after k_mt_resolve: [fail | end slots]
    int32_t *d_fail = (int32_t *)take(256);
    int64_t *d_endslot = (int64_t *)take(8 * (size_t)nstream);
    int32_t *d_endhasg = (int32_t *)take(4 * (size_t)nstream);
    int32_t *d_endnew = (int32_t *)take(4 * (size_t)nstream);
    double *d_endgauss = (double *)take(8 * (size_t)nstream);
    // uploaded together before k_mt_advance: [window index | skip | skipc]
    const size_t adv_stride = (8 * (size_t)nstream + 255) & ~(size_t)255;
    int64_t *d_widx = (int64_t *)take(8 * (size_t)nstream);
    int64_t *d_skip = (int64_t *)take(8 * (size_t)nstream);
    int64_t *d_skipc = (int64_t *)take(8 * (size_t)nstream);
    MtObj *d_objs = (MtObj *)take((size_t)nobj_total * sizeof(MtObj));
    int64_t *d_segp0 = (int64_t *)take(8 * ((size_t)Ktot + nobj_total));
    int64_t *d_sega = (int64_t *)take(8 * ((size_t)Ktot + nobj_total));
    int64_t *d_seglo = (int64_t *)take(8 * (size_t)nobj_total);
    int32_t *d_nseg = (int32_t *)take(4 * (size_t)nobj_total);
    double *d_cached = (double *)take(8 * (size_t)nobj_total);
    int32_t *d_cflag = (int32_t *)take(4 * (size_t)nobj_total);
    double *d_gauss0 = (double *)take(8 * (size_t)nstream);
    // chains
    std::vector<int64_t> c2s, c2d;
    std::vector<int32_t> c2n;
    std::vector<std::vector<int64_t>> r1s(16), r1d(16);      // chains of first-level round r
    std::vector<MtSub> subs(Ktot);
    std::vector<int64_t> hbase(nstream + 1), hbit(nstream), hsblo(nstream + 1), hT(nstream);
    hbase[nstream] = Ktot;
    for (int g = 0; g < nstream; ++g) {
        const MtPlanStream &p = ps[g];
        hbase[g] = p.base;
        hbit[g] = p.bit_base;
        hsblo[g] = p.bit_base / MT_SB;
        hT[g] = p.T;
        {
            // first-level windows (every MT_L1-th sub-stream) by a doubling tree: round r makes
            // windows 2^r .. 2^(r+1) - 1 from windows 0 .. 2^r - 1 with the stride 128 J 2^r
            const int64_t n1 = (p.K - 1) / MT_L1;          // first-level windows besides window 0
            for (int r = 0; ((int64_t)1 << r) <= n1; ++r) {
                if (r >= nlev) return 1;                   // stream longer than the polynomials reach
                for (int64_t m = 0; m < ((int64_t)1 << r) && m + ((int64_t)1 << r) <= n1; ++m) {
                    r1s[r].push_back(p.base + MT_L1 * m);
                    r1d[r].push_back(p.base + MT_L1 * (m + ((int64_t)1 << r)));
                }
            }
        }
        for (int64_t m = 0; m < p.K; m += MT_L1) {
            const int64_t cnt = std::min<int64_t>(MT_L1 - 1, p.K - m - 1);
            if (cnt > 0) {
                c2s.push_back(p.base + m);
                c2d.push_back(p.base + m + 1);
                c2n.push_back((int32_t)cnt);
            }
        }
        for (int64_t k = 0; k < p.K; ++k) {
            MtSub &sb = subs[p.base + k];
            auto qk = [&](int64_t kk) -> int64_t {
                if (kk <= 0) return 0;
                int64_t q = (kk * MT_J - p.pos0 + 3) / 4;
                q = (q + 63) / 64 * 64;
                return q;
            };
            sb.q0 = std::min(qk(k), p.T);
            sb.q1 = std::min(qk(k + 1), p.T);
            if (k == p.K - 1) sb.q1 = p.T;
            sb.bit0 = p.bit_base + sb.q0;
            sb.stream = g;
            sb.skip = (int32_t)(p.pos0 + 4 * sb.q0 - k * MT_J);
        }
    }
    hsblo[nstream] = Ttot / MT_SB;
    const size_t n2 = c2s.size();
    size_t n1tot = 0;
    for (int r = 0; r < 16; ++r) n1tot += r1s[r].size();
    // ---- the plan block: one contiguous device region, one page-locked mirror, one copy --------
    const size_t plan0 = off;
    uint32_t *d_polys = (uint32_t *)take(polys.size() * 4);
    MtSub *d_subs = (MtSub *)take((size_t)Ktot * sizeof(MtSub));
    int64_t *d_base = (int64_t *)take(8 * ((size_t)nstream + 1));
    int64_t *d_bitbase = (int64_t *)take(8 * (size_t)nstream);
    int64_t *d_sblo = (int64_t *)take(8 * ((size_t)nstream + 1));
    int64_t *d_tslots = (int64_t *)take(8 * (size_t)nstream);
    int64_t *d_c1s = (int64_t *)take(8 * (n1tot + 1)), *d_c1d = (int64_t *)take(8 * (n1tot + 1));
    int32_t *d_c1n = (int32_t *)take(4 * (n1tot + 1));
    int64_t *d_c2s = (int64_t *)take(8 * (n2 + 1)), *d_c2d = (int64_t *)take(8 * (n2 + 1));
    int32_t *d_c2n = (int32_t *)take(4 * (n2 + 1));
    const size_t plan_bytes = off - plan0;
    if (off > scratch_bytes) return 1;
    char *hp = g_plan_pin.get(plan_bytes);
    if (!hp) return fail(BRUTUS_ENOMEM, "page-locked staging for the stream plan (%zu bytes)", plan_bytes);
    auto at = [&](const void *d) { return hp + ((const char *)d - (scratch + plan0)); };
    memcpy(at(d_polys), polys.data(), polys.size() * 4);
    memcpy(at(d_subs), subs.data(), sizeof(MtSub) * (size_t)Ktot);
    memcpy(at(d_base), hbase.data(), 8 * ((size_t)nstream + 1));
    memcpy(at(d_bitbase), hbit.data(), 8 * (size_t)nstream);
    memcpy(at(d_sblo), hsblo.data(), 8 * ((size_t)nstream + 1));
    memcpy(at(d_tslots), hT.data(), 8 * (size_t)nstream);
    {
        int64_t *fs = (int64_t *)at(d_c1s), *fd = (int64_t *)at(d_c1d);
        int32_t *fn = (int32_t *)at(d_c1n);
        size_t k = 0;
        for (int r = 0; r < 16; ++r)
            for (size_t q = 0; q < r1s[r].size(); ++q, ++k) {
                fs[k] = r1s[r][q];
                fd[k] = r1d[r][q];
                fn[k] = 1;
            }
    }
    if (n2) {
        memcpy(at(d_c2s), c2s.data(), 8 * n2);
        memcpy(at(d_c2d), c2d.data(), 8 * n2);
        memcpy(at(d_c2n), c2n.data(), 4 * n2);
    }
    HIP_TRY(hipMemcpyAsync(scratch + plan0, hp, plan_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_fail, 0, 4, st));
    // ---- sub-stream windows by jump-ahead ------------------------------------------------
    tm.begin("k_mt_jump");
    hipLaunchKernelGGL(k_mt_keys, dim3(nstream), dim3(256), 0, st, nstream, d_states, d_base, d_win);
    const size_t jlds = (size_t)MT_JX * 4 + 19968 * 2;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_TRY(hipFuncSetAttribute((const void *)k_mt_jump, hipFuncAttributeMaxDynamicSharedMemorySize, (int)jlds));
        attr_set = true;
    }
    {
        size_t o1 = 0;
        for (int r = 0; r < 16; ++r) {
            const size_t nr = r1s[r].size();
            if (!nr) continue;
            hipLaunchKernelGGL(k_mt_jump, dim3((unsigned)nr), dim3(MT_NT), jlds, st,
                               d_polys + (size_t)(1 + r) * MT_N, d_win, d_c1s + o1, d_c1d + o1,
                               (int64_t)1, d_c1n + o1);
            o1 += nr;
        }
    }
    if (n2)
        hipLaunchKernelGGL(k_mt_jump, dim3((unsigned)n2), dim3(MT_NT), jlds, st, d_polys, d_win, d_c2s,
                           d_c2d, (int64_t)1, d_c2n);
    tm.end();
    fire_after_jump(st);
    // ---- pass 1, prefix, boundaries ----------------------------------------------------------
    tm.begin("k_mt_bits");
    if (mapped)
        hipLaunchKernelGGL(k_mt_bits<true>, dim3((unsigned)Ktot), dim3(MT_PT), 0, st, (int)Ktot, d_subs,
                           d_win, d_bits, d_zloc);
    else
        hipLaunchKernelGGL(k_mt_bits<false>, dim3((unsigned)Ktot), dim3(MT_PT), 0, st, (int)Ktot, d_subs,
                           d_win, d_bits, (double2 *)nullptr);
    tm.end();
    tm.begin("k_mt_resolve");
    hipLaunchKernelGGL(k_mt_sbcount, dim3((unsigned)((nsb + 3) / 4)), dim3(256), 0, st, nsb, d_bits, d_cnt);
    hipLaunchKernelGGL(k_mt_sbscan, dim3(nstream), dim3(1024), 0, st, d_sblo, d_cnt, d_pre);
    hipLaunchKernelGGL(k_mt_resolve, dim3(nstream), dim3(64), 0, st, d_seg, d_nnorm, nuni, d_states,
                       d_bitbase, d_sblo, d_tslots, d_bits, d_pre, d_zoff, mapped ? (double *)nullptr : d_Z,
                       d_objs, d_endslot, d_endhasg, d_endnew, d_fail, d_gauss0);
    tm.end();
    // (the plan's page-locked mirror is free again once the stream has passed the copy; it
    // doubles as the landing zone of [fail | end slots], which are adjacent on the device)
    const size_t back_bytes = 256 + 8 * (size_t)nstream;
    char *hb = g_plan_pin.get(back_bytes > plan_bytes ? back_bytes : plan_bytes);
    if (!hb) return fail(BRUTUS_ENOMEM, "page-locked staging");
    HIP_TRY(hipMemcpyAsync(hb, d_fail, back_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const int32_t hfail = *(const int32_t *)hb;
    std::vector<int64_t> hend(nstream);
    memcpy(hend.data(), hb + 256, 8 * (size_t)nstream);
    if (hfail) return 1;        // not enough slots generated (the resolve wrote only scratch and
                                // possibly a cached deviate the sequential walk rewrites)
    // ---- states after the last consumed word ---------------------------------------------------
    // (before pass 2: the boundaries fix the state; a new cached deviate is the f * x1 of the
    // candidate slot that ends 2 nuni words before the end, which k_mt_advance meets on its
    // way when it starts from the window holding that slot)
    char *ha = g_plan_pin.get(3 * adv_stride);
    if (!ha) return fail(BRUTUS_ENOMEM, "page-locked staging");
    int64_t *hw = (int64_t *)ha, *hs = (int64_t *)(ha + adv_stride), *hc = (int64_t *)(ha + 2 * adv_stride);
    for (int g = 0; g < nstream; ++g) {
        const int64_t e = ps[g].pos0 + 4 * hend[g];
        const int64_t ec = e - 2 * (int64_t)nuni;
        int64_t k = (ec - 4 >= 0 ? ec - 4 : 0) / MT_J;
        if (k >= ps[g].K) k = ps[g].K - 1;
        hw[g] = ps[g].base + k;
        hs[g] = e - k * MT_J;
        hc[g] = ec - k * MT_J;
    }
    HIP_TRY(hipMemcpyAsync(d_widx, ha, 3 * adv_stride, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_mt_advance, dim3(nstream), dim3(MT_PT), 0, st, nstream, d_win, d_widx, d_skip,
                       d_skipc, d_endhasg, d_endnew, d_endgauss, d_states);
    // ---- pass 2 -----------------------------------------------------------------------------
    MtEmitLaunch el{(int)Ktot, d_subs, d_win, d_bits, d_bitbase, d_sblo, d_pre, d_seg, d_objs,
                    d_nnorm, d_zoff, d_Z, nuni, d_U, d_endgauss, mapped ? 1 : 0, ZMap{},
                    nstream, seg[nstream] - seg[0], d_gauss0, d_base};
    if (mapped) el.zm = ZMap{d_zloc, d_segp0, d_sega, d_seglo, d_nseg, d_cached, d_cflag};
    if (out) *out = el;
    if (defer_emit) {
        std::lock_guard<std::mutex> lk(g_emit_mu);
        g_emit[(const void *)scratch] = el;
    } else {
        launch_mt_emit(el, st, tm);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));        // host vectors go out of scope
    return 0;
}


// Walk the streams of one group: many workgroups per stream when the jump polynomials are
// loaded and the shape allows it, else one workgroup per stream (k_mt_stream).
int mt_walk(int nstream, const std::vector<int32_t> &seg, uint32_t *d_states, std::vector<int> &pos0,
            const std::vector<int64_t> &nnorm, const int32_t *d_seg, const int64_t *d_nnorm,
            const int64_t *d_zoff, double *d_Z, int nuni, double *d_U, char *scratch,
            size_t scratch_bytes, int nobj_total, hipStream_t st, Timer &tm, bool defer_emit = false,
            double2 *d_zloc = nullptr, size_t zloc_pairs = 0, MtEmitLaunch *out = nullptr) {
    int rc = 1;
    if (out) *out = MtEmitLaunch{};
    if (scratch) {
        std::lock_guard<std::mutex> lk(g_emit_mu);
        g_emit.erase((const void *)scratch);
    }
    if (env_int("BRUTUS_MT_PARALLEL", 1) && scratch)
        rc = mt_walk_parallel(nstream, seg, d_states, pos0, nnorm, d_seg, d_nnorm, d_zoff, d_Z, nuni,
                              d_U, scratch, scratch_bytes, nobj_total, st, tm, defer_emit, d_zloc,
                              zloc_pairs, out);
    if (rc < 0) return rc;
    if (rc == 1) {
        if (out) *out = MtEmitLaunch{};           // flat normals from the sequential walker
        tm.begin("k_mt_stream");
        hipLaunchKernelGGL(k_mt_stream, dim3(nstream), dim3(MT_NT), 0, st, nstream, d_seg, d_states,
                           d_nnorm, d_zoff, d_Z, nuni, d_U);
        tm.end();
        HIP_TRY(hipGetLastError());
    }
    // where the streams stand now (the next group of a shared stream starts there)
    std::vector<uint32_t> hp(nstream);
    for (int g = 0; g < nstream; ++g)
        HIP_TRY(hipMemcpyAsync(&hp[g], d_states + (size_t)g * MT_STATE_WORDS + MT_N, 4,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int g = 0; g < nstream; ++g) pos0[g] = (int)hp[g];
    return 0;
}

// Rv pinned by its limits at the value every fit starts from: the (offset, Av)
// specialisation computes the same thing
inline bool rv_pinned(const DevParams &p) { return p.rvmin == p.rvmax && p.rv_mean == p.rvmin; }

// The float32 pass only classifies, and what it "proves" below a threshold never reaches the
// output: Star32::eps has to bound |float32 - float64| for that to be sound.  Every pair the call
// re-evaluates in float64 anyway (the nominees of both exact maxima, the pairs inside the
// first-cut band) is compared with its float32 value on an audited call; one at or above eps
// fails the call -- loudly, instead of a model silently missing from a posterior.
int audit_verdict(const Workspace &w, int nstar, hipStream_t st) {
    std::vector<float> aud(4 * (size_t)nstar);
    std::vector<Star32> s32(nstar);
    HIP_TRY(hipMemcpyAsync(aud.data(), w.aud, sizeof(float) * aud.size(), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(s32.data(), w.s32, sizeof(Star32) * (size_t)nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    static const char *what[3] = {"cull statistic", "first-cut statistic (maximum)", "first-cut statistic (band)"};
    for (int r = 0; r < 3; ++r)
        for (int s = 0; s < nstar; ++s)
            if (!(aud[(size_t)r * nstar + s] < s32[s].eps))
                return fail(BRUTUS_EPRECISION,
                            "float32 proof bound violated: star %d of the batch, %s: |float32 - float64| = %.3g "
                            "against eps = %.3g (BRUTUS_EPS_SCALE raises the bound; please report the input)",
                            s, what[r], (double)aud[(size_t)r * nstar + s], (double)s32[s].eps);
    return 0;
}

int dispatch_fit(int nb, int nfilt, const float *grid, int64_t nmodel, int nstar,
                 const DevParams &p, int max_iter, Workspace &w, int64_t capacity,
                 int32_t *d_rec_idx, int32_t *d_rec_slot, double *d_rec_vals, int64_t *d_rec_off,
                 int32_t *h_k1, int32_t *h_k2, int64_t *h_counts, hipStream_t st, Timer &tm) {
    const bool rvf = rv_pinned(p);
    // BRUTUS_FIT_HOSTDRIVEN=1: round 3's driver for every batch (A/B timing, and the tests
    // that compare the two drivers record for record)
    const bool hostdriven = env_int("BRUTUS_FIT_HOSTDRIVEN", 0) != 0;
    {
        const long long call_no = g_fit_calls.fetch_add(1);
        const int every = env_int("BRUTUS_AUDIT_EVERY", 64);
        t_audit_call = every > 0 && call_no % every == 0;
    }
#define BRUTUS_CASE(N)                                                                             \
    case N: {                                                                                      \
        int rc = hostdriven ? BRUTUS_RETRY_HOSTDRIVEN : 0;                                         \
        if (!hostdriven)                                                                           \
            rc = rvf ? run_fit<N, true>(grid, nmodel, nfilt, nstar, p, max_iter, w, capacity,      \
                                        d_rec_idx, d_rec_slot, d_rec_vals, d_rec_off, h_k1, h_k2,  \
                                        h_counts, st, tm, true)                                    \
                     : run_fit<N, false>(grid, nmodel, nfilt, nstar, p, max_iter, w, capacity,     \
                                         d_rec_idx, d_rec_slot, d_rec_vals, d_rec_off, h_k1, h_k2, \
                                         h_counts, st, tm, true);                                  \
        if (rc == BRUTUS_RETRY_HOSTDRIVEN) {                                                       \
            if (!hostdriven) g_fit_retries.fetch_add(1);                                           \
            rc = rvf ? run_fit<N, true>(grid, nmodel, nfilt, nstar, p, max_iter, w, capacity,      \
                                        d_rec_idx, d_rec_slot, d_rec_vals, d_rec_off, h_k1, h_k2,  \
                                        h_counts, st, tm, false)                                   \
                     : run_fit<N, false>(grid, nmodel, nfilt, nstar, p, max_iter, w, capacity,     \
                                         d_rec_idx, d_rec_slot, d_rec_vals, d_rec_off, h_k1, h_k2, \
                                         h_counts, st, tm, false);                                 \
        }                                                                                          \
        if (rc == 0 && t_audit_call) rc = audit_verdict(w, nstar, st);                             \
        t_audit_call = 0;                                                                          \
        return rc;                                                                                 \
    }
    switch (nb) {
        BRUTUS_CASE(12)
#ifndef BRUTUS_DEV_NB12_ONLY
        BRUTUS_CASE(8)
        BRUTUS_CASE(16)
        BRUTUS_CASE(24)
        BRUTUS_CASE(32)
#endif
    }
#undef BRUTUS_CASE
    if (nb > BRUTUS_MAX_FILT_FIT)
        return fail(BRUTUS_EINVAL, "brutus_fit_batch fits at most %d bands at once (%d given): take the full-grid "
                                   "outputs of brutus_loglike_batch and cut on them", BRUTUS_MAX_FILT_FIT, nfilt);
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
    hipLaunchKernelGGL(k_prep, dim3(nstar), dim3(64), 0, st, nstar, nfilt, d_flux,
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
    const size_t a = carve(nullptr, nmodel, nstar, true).bytes, b = carve(nullptr, nmodel, nstar, false).bytes;
    return a > b ? a : b;
}

int brutus_loglike_batch(const float *d_grid_soa, int64_t nmodel, int nfilt, int nstar,
                         const double *d_flux, const double *d_err, const uint8_t *d_mask,
                         const double *d_parallax, const double *d_parallax_err, int has_parallax,
                         const brutus_params *params, void *d_workspace, size_t workspace_bytes,
                         double *d_lnl, double *d_chi2, double *d_scale, double *d_av, double *d_rv,
                         double *d_icov, int32_t *d_ndim, int32_t *h_k1, int32_t *h_k2,
                         const double *d_av_init, const double *d_rv_init, void *stream) {
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
    const int max_iter = params->max_iter > 0 ? params->max_iter : 65536;
    int rc = dispatch_pipeline(padded_nb(nfilt), d_grid_soa, nmodel, nstar, p, max_iter, w, h_k1,
                               h_k2, st, tm, d_av_init, d_rv_init);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    fix_k2(h_k2, nstar);
    tm.collect();
    return 0;
}

int brutus_fit_batch(const float *d_grid_soa, int64_t nmodel, int nfilt, int nstar,
                     const double *d_flux, const double *d_err, const uint8_t *d_mask,
                     const double *d_parallax, const double *d_parallax_err, int has_parallax,
                     const brutus_params *params, void *d_workspace, size_t workspace_bytes,
                     int64_t capacity, int32_t *d_rec_idx, int32_t *d_rec_slot, double *d_rec_vals,
                     int64_t *d_rec_off, int32_t *d_ndim, int32_t *h_k1, int32_t *h_k2,
                     int64_t *h_counts, void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    DevParams p;
    if (int rc = make_params(params, p)) return rc;
    if (!d_grid_soa || !d_flux || !d_err || !d_mask || !d_workspace || !d_rec_idx || !d_rec_slot ||
        !d_rec_vals || !d_rec_off || !d_ndim || !h_counts || capacity < 0 || capacity > INT32_MAX)
        return fail(BRUTUS_EINVAL, "NULL pointer or capacity outside [0, 2^31)");
    if (nmodel >= SURV_TAG_END)       // (candidate-list positions are kept as float32 bit patterns below 2^-100)
        return fail(BRUTUS_EINVAL, "brutus_fit_batch takes grids of fewer than %d models", SURV_TAG_END);
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes)
        return fail(BRUTUS_ENOMEM, "workspace too small: need %zu bytes, got %zu", w.bytes,
                    workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    if (int rc = launch_prep(nstar, nfilt, d_flux, d_err, d_mask, d_parallax, d_parallax_err,
                             has_parallax, w, d_ndim, st))
        return rc;
    const int max_iter = params->max_iter > 0 ? params->max_iter : 65536;
    int rc = dispatch_fit(padded_nb(nfilt), nfilt, d_grid_soa, nmodel, nstar, p, max_iter, w,
                          capacity, d_rec_idx, d_rec_slot, d_rec_vals, d_rec_off, h_k1, h_k2,
                          h_counts, st, tm);
    (void)hipStreamSynchronize(st);
    if (rc) return rc;
    fix_k2(h_k2, nstar);
    tm.collect();
    return 0;
}

constexpr int CLUSTER_CHUNKS = 256;

size_t brutus_cluster_workspace_bytes(int nobj) {
    if (nobj <= 0) return 0;
    return 2 * align_up(sizeof(double) * (size_t)nobj * CLUSTER_CHUNKS);
}

int brutus_cluster_chunks(void) { return CLUSTER_CHUNKS; }

extern "C++" {
template <bool MAGS>
static int cluster_part(int nobj, int nfilt, int npts, const double *d_pts_flux,
                        const double *d_pts_lnw, ClusterMags mg, const double *d_phot,
                        const double *d_ivar, const double *d_chi2_p, const double *d_lnorm,
                        const int32_t *d_ndim, int dim_prior, void *d_workspace,
                        size_t workspace_bytes, int chunk_lo, int chunk_n, void *stream) {
    const int nb = padded_nb(nfilt);
    if (nobj <= 0 || npts < 0 || nb < 0)
        return fail(BRUTUS_EINVAL, "bad cluster dimensions (nobj=%d, npts=%d, nfilt=%d)", nobj,
                    npts, nfilt);
    if (chunk_lo < 0 || chunk_n < 1 || chunk_lo + chunk_n > CLUSTER_CHUNKS)
        return fail(BRUTUS_EINVAL, "bad chunk range [%d, %d) of %d", chunk_lo, chunk_lo + chunk_n,
                    CLUSTER_CHUNKS);
    if (!d_phot || !d_ivar || !d_chi2_p || !d_lnorm || !d_ndim || !d_workspace)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    if (npts > 0 && (MAGS ? (!mg.src || !mg.mags || !mg.lnw_eep || !mg.lnw_smf || mg.neep <= 0)
                          : (!d_pts_flux || !d_pts_lnw)))
        return fail(BRUTUS_EINVAL, "NULL device pointer (isochrone points)");
    if (workspace_bytes < brutus_cluster_workspace_bytes(nobj))
        return fail(BRUTUS_ENOMEM, "cluster workspace too small");
    double *pm = (double *)d_workspace + (size_t)chunk_lo * nobj;
    double *ps = (double *)((char *)d_workspace + align_up(sizeof(double) * (size_t)nobj * CLUSTER_CHUNKS)) +
                 (size_t)chunk_lo * nobj;
    hipStream_t st = (hipStream_t)stream;
    // every chunk of the range is written: one without points holds (-inf, 0)
    const int ppb = npts > 0 ? (npts + chunk_n - 1) / chunk_n : 1;
    const dim3 g((nobj + CL_T - 1) / CL_T, chunk_n);
    Timer tm(st);
    tm.begin("k_cluster");
#define BRUTUS_CL(N)                                                                              \
    case N:                                                                                       \
        hipLaunchKernelGGL((k_cluster<N, MAGS>), g, dim3(CL_T), 0, st, nobj, nfilt, npts,         \
                           d_pts_flux, d_pts_lnw, mg, d_phot, d_ivar, d_chi2_p, d_lnorm, d_ndim,  \
                           dim_prior, ppb, pm, ps);                                               \
        break;
    switch (nb) {
        BRUTUS_CL(12)
#ifndef BRUTUS_DEV_NB12_ONLY
        BRUTUS_CL(8)
        BRUTUS_CL(16)
        BRUTUS_CL(24)
        BRUTUS_CL(32)
#endif
        default:
            return fail(BRUTUS_EINVAL, "cluster likelihood: at most %d bands (%d given)", BRUTUS_MAX_FILT_FIT, nfilt);
    }
#undef BRUTUS_CL
    tm.end();
    HIP_TRY(hipGetLastError());
    tm.collect();
    return 0;
}
}  // extern "C++"

int brutus_cluster_lnl_part(int nobj, int nfilt, int npts, const double *d_pts_flux,
                            const double *d_pts_lnw, const double *d_phot, const double *d_ivar,
                            const double *d_chi2_p, const double *d_lnorm, const int32_t *d_ndim,
                            int dim_prior, void *d_workspace, size_t workspace_bytes, int chunk_lo,
                            int chunk_n, void *stream) {
    return cluster_part<false>(nobj, nfilt, npts, d_pts_flux, d_pts_lnw, ClusterMags{}, d_phot,
                               d_ivar, d_chi2_p, d_lnorm, d_ndim, dim_prior, d_workspace,
                               workspace_bytes, chunk_lo, chunk_n, stream);
}

int brutus_cluster_lnl_part_mags(int nobj, int nfilt, int npts, int neep, const int32_t *d_src,
                                 const double *d_mags, const double *d_lnw_eep,
                                 const double *d_lnw_smf, const double *d_phot,
                                 const double *d_ivar, const double *d_chi2_p,
                                 const double *d_lnorm, const int32_t *d_ndim, int dim_prior,
                                 void *d_workspace, size_t workspace_bytes, int chunk_lo,
                                 int chunk_n, void *stream) {
    ClusterMags mg;
    mg.src = d_src;
    mg.mags = d_mags;
    mg.lnw_eep = d_lnw_eep;
    mg.lnw_smf = d_lnw_smf;
    mg.neep = neep;
    return cluster_part<true>(nobj, nfilt, npts, nullptr, nullptr, mg, d_phot, d_ivar, d_chi2_p,
                              d_lnorm, d_ndim, dim_prior, d_workspace, workspace_bytes, chunk_lo,
                              chunk_n, stream);
}

int brutus_cluster_lnl_merge(int nobj, int nchunk, void *d_workspace, size_t workspace_bytes,
                             double *d_lnl, void *stream) {
    if (nobj <= 0 || nchunk < 1 || nchunk > CLUSTER_CHUNKS || !d_workspace || !d_lnl)
        return fail(BRUTUS_EINVAL, "bad cluster merge (nobj=%d, nchunk=%d)", nobj, nchunk);
    if (workspace_bytes < brutus_cluster_workspace_bytes(nobj))
        return fail(BRUTUS_ENOMEM, "cluster workspace too small");
    double *pm = (double *)d_workspace;
    double *ps = (double *)((char *)d_workspace + align_up(sizeof(double) * (size_t)nobj * CLUSTER_CHUNKS));
    hipLaunchKernelGGL(k_cluster_merge, dim3((nobj + CM_O - 1) / CM_O), dim3(CM_O * CM_J), 0,
                       (hipStream_t)stream, nobj, nchunk, pm, ps, d_lnl);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_cluster_mix(int nobj, const double *d_lnl, const double *d_lnl_outlier, double ln_fin,
                       double ln_fout, double *d_lnl_mix, double *d_lnl_tot, void *stream) {
    if (nobj <= 0 || !d_lnl || !d_lnl_outlier || !d_lnl_mix || !d_lnl_tot)
        return fail(BRUTUS_EINVAL, "bad cluster mixture arguments (nobj=%d)", nobj);
    hipLaunchKernelGGL(k_cluster_mix, dim3(1), dim3(CX_T), 0, (hipStream_t)stream, nobj, d_lnl,
                       d_lnl_outlier, ln_fin, ln_fout, d_lnl_mix, d_lnl_tot);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_cluster_lnl(int nobj, int nfilt, int npts, const double *d_pts_flux,
                       const double *d_pts_lnw, const double *d_phot, const double *d_ivar,
                       const double *d_chi2_p, const double *d_lnorm, const int32_t *d_ndim,
                       int dim_prior, void *d_workspace, size_t workspace_bytes, double *d_lnl,
                       void *stream) {
    if (npts <= 0) return fail(BRUTUS_EINVAL, "bad cluster dimensions (npts=%d)", npts);
    if (!d_lnl) return fail(BRUTUS_EINVAL, "NULL device pointer");
    static const int want_chunks = env_int("BRUTUS_CLUSTER_CHUNKS", CLUSTER_CHUNKS);
    const int use_chunks = want_chunks < 1 ? 1 : (want_chunks > CLUSTER_CHUNKS ? CLUSTER_CHUNKS : want_chunks);
    const int rc = brutus_cluster_lnl_part(nobj, nfilt, npts, d_pts_flux, d_pts_lnw, d_phot, d_ivar,
                                           d_chi2_p, d_lnorm, d_ndim, dim_prior, d_workspace,
                                           workspace_bytes, 0, use_chunks, stream);
    if (rc) return rc;
    return brutus_cluster_lnl_merge(nobj, use_chunks, d_workspace, workspace_bytes, d_lnl, stream);
}

int brutus_cluster_points(int64_t npts, int nfilt, const int32_t *d_src, const double *d_mags,
                          const double *d_lnw_in, double *d_pts_flux, double *d_pts_lnw,
                          void *stream) {
    if (npts <= 0 || nfilt <= 0) return fail(BRUTUS_EINVAL, "bad point-table dimensions");
    if (!d_mags || !d_lnw_in || !d_pts_flux || !d_pts_lnw) return fail(BRUTUS_EINVAL, "NULL device pointer");
    hipLaunchKernelGGL(k_cluster_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, npts, nfilt, d_src, d_mags, d_lnw_in, 0,
                       (const double *)nullptr, d_pts_flux, d_pts_lnw);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_cluster_points_grid(int64_t npts, int nfilt, int neep, const int32_t *d_src,
                               const double *d_mags, const double *d_lnw_eep,
                               const double *d_lnw_smf, double *d_pts_flux, double *d_pts_lnw,
                               void *stream) {
    if (npts <= 0 || nfilt <= 0 || neep <= 0) return fail(BRUTUS_EINVAL, "bad point-table dimensions");
    if (!d_mags || !d_lnw_eep || !d_lnw_smf || !d_pts_flux || !d_pts_lnw)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    hipLaunchKernelGGL(k_cluster_points, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, npts, nfilt, d_src, d_mags, d_lnw_eep, neep, d_lnw_smf,
                       d_pts_flux, d_pts_lnw);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- utils.photometric_offsets on the device (offsets_kernels.hpp) ----------
int brutus_offsets_weights(int nobj, int nsamps, int nfilt, int64_t nmodel, const float *d_models,
                           const int64_t *d_idxs, const double *d_reds, const double *d_dreds,
                           const double *d_dists, const double *d_phot, const double *d_err,
                           const uint8_t *d_mask, const double *d_weights,
                           const double *d_old_offsets, const uint8_t *d_use,
                           const uint8_t *d_mask_fit, int dim_prior, double *d_flux, double *d_cdf,
                           void *stream) {
    if (nobj <= 0 || nsamps <= 0 || nfilt <= 0 || nfilt > NBMAX || nmodel <= 0)
        return fail(BRUTUS_EINVAL, "bad photometric-offset dimensions (nobj=%d, nsamps=%d, nfilt=%d)",
                    nobj, nsamps, nfilt);
    if (!d_models || !d_idxs || !d_reds || !d_dreds || !d_dists || !d_phot || !d_err || !d_mask ||
        !d_weights || !d_old_offsets || !d_use || !d_mask_fit || !d_flux || !d_cdf)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nt = (int64_t)nobj * nsamps;
    hipLaunchKernelGGL(k_po_flux, dim3((unsigned)((nt + PO_T - 1) / PO_T)), dim3(PO_T), 0, st, nobj,
                       nsamps, nfilt, nmodel, d_models, d_idxs, d_reds, d_dreds, d_dists, d_phot,
                       d_err, d_mask, d_old_offsets, d_use, d_mask_fit, dim_prior, d_flux, d_cdf);
    hipLaunchKernelGGL(k_po_cdf, dim3(nobj, nfilt), dim3(PO_T), 0, st, nobj, nsamps, d_weights,
                       d_use, d_mask_fit, d_cdf);
    HIP_TRY(hipGetLastError());
    return 0;
}

namespace {
// [vals | sorted | segment offsets | rocPRIM scratch]
struct OffsetsWs {
    double *vals, *sorted;
    int32_t *seg;
    void *tmp;
    size_t tmp_bytes, bytes;
};
static int carve_offsets(char *base, int n, int nmc, OffsetsWs &w) {
    const size_t nv = (size_t)n * nmc;
    size_t off = 0;
    w.vals = (double *)(base + off);
    off += align_up(sizeof(double) * nv);
    w.sorted = (double *)(base + off);
    off += align_up(sizeof(double) * nv);
    w.seg = (int32_t *)(base + off);
    off += align_up(sizeof(int32_t) * ((size_t)nmc + 1));
    w.tmp = base + off;
    w.tmp_bytes = 0;
    hipError_t e = rocprim::segmented_radix_sort_keys(
        nullptr, w.tmp_bytes, (const double *)nullptr, (double *)nullptr, (unsigned int)nv,
        (unsigned int)nmc, (const int32_t *)nullptr, (const int32_t *)nullptr);
    if (e != hipSuccess) return -1;
    off += align_up(w.tmp_bytes);
    w.bytes = off;
    return 0;
}
}  // namespace

size_t brutus_offsets_workspace_bytes(int n, int nmc) {
    if (n <= 0 || nmc <= 0 || (int64_t)n * nmc >= (int64_t)1 << 31) return 0;
    OffsetsWs w;
    if (carve_offsets(nullptr, n, nmc, w)) return 0;
    return w.bytes;
}

int brutus_offsets_bootstrap(int band, int nobj, int nsamps, int nfilt, int n, int nmc,
                             const int32_t *d_subset, const double *d_cdf_obj, const double *d_u,
                             const double *d_flux, const double *d_cdf, const double *d_phot,
                             void *d_workspace, size_t workspace_bytes, double *d_meds,
                             void *stream) {
    if (band < 0 || band >= nfilt || nobj <= 0 || nsamps <= 0 || n <= 0 || n > nobj || nmc <= 0 ||
        (int64_t)n * nmc >= (int64_t)1 << 31)
        return fail(BRUTUS_EINVAL, "bad bootstrap dimensions (band=%d, n=%d, nmc=%d)", band, n, nmc);
    if (!d_subset || !d_cdf_obj || !d_u || !d_flux || !d_cdf || !d_phot || !d_workspace || !d_meds)
        return fail(BRUTUS_EINVAL, "NULL device pointer");
    OffsetsWs w;
    if (carve_offsets((char *)d_workspace, n, nmc, w)) return fail(BRUTUS_EHIP, "rocPRIM sizing failed");
    if (workspace_bytes < w.bytes) return fail(BRUTUS_ENOMEM, "photometric-offset workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t nv = (int64_t)n * nmc;
    hipLaunchKernelGGL(k_po_segments, dim3((nmc + 1 + 255) / 256), dim3(256), 0, st, n, nmc, w.seg);
    hipLaunchKernelGGL(k_po_boot, dim3((unsigned)((nv + PO_T - 1) / PO_T)), dim3(PO_T), 0, st, band,
                       nobj, nsamps, nfilt, n, nmc, d_subset, d_cdf_obj, d_u, d_flux, d_cdf, d_phot,
                       w.vals);
    HIP_TRY(rocprim::segmented_radix_sort_keys(w.tmp, w.tmp_bytes, (const double *)w.vals, w.sorted,
                                               (unsigned int)nv, (unsigned int)nmc, w.seg, w.seg + 1,
                                               0, 64, st));
    hipLaunchKernelGGL(k_po_median, dim3((nmc + 255) / 256), dim3(256), 0, st, n, nmc, w.sorted,
                       d_meds);
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
    int32_t *mc_order;      // objects by falling number of kept records (k_post_order)
    double2 *mc_stage;
    // numpy-stream mode (mt_kernels.hpp)
    uint32_t *mt_states;      // (nstar, MT_STATE_WORDS)
    int64_t *mt_nnorm, *mt_zoff;   // (nstar,)
    int32_t *mt_seg;          // (nstar + 1,)
    double *mt_uni;           // (nstar, 2 * ndraws)
    size_t bytes;
};

constexpr int MC_SLOTS = 1024;     // persistent workgroups (= staging slots) of k_post_mc

static PostWs carve_post(char *base, int nstar, int64_t cap, int nmc, int ndraws = 0) {
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
    w.rp.chol = (double *)take(8 * 6 * c);
    w.cdf = (double *)take(8 * c);
    w.sort_keys = (double *)take(8 * c);
    w.sort_in = (int32_t *)take(4 * c);
    w.sort_perm = (int32_t *)take(4 * c);
    w.sort_tmp_bytes = 16 * c + (8u << 20);
    w.sort_tmp = take(w.sort_tmp_bytes);
    w.mc_counter = (unsigned int *)take(256);
    w.mc_order = (int32_t *)take(4 * (size_t)BRUTUS_MAX_BATCH);
    w.mc_stage = (double2 *)take(sizeof(double2) * (size_t)MC_SLOTS * mc_npair_max(nmc) * TILE);
    w.mt_states = (uint32_t *)take(sizeof(uint32_t) * (size_t)nstar * MT_STATE_WORDS);
    w.mt_nnorm = (int64_t *)take(8 * (size_t)nstar);
    w.mt_zoff = (int64_t *)take(8 * (size_t)nstar);
    w.mt_seg = (int32_t *)take(4 * ((size_t)nstar + 1));
    w.mt_uni = (double *)take(8 * (size_t)nstar * 2 * (size_t)(ndraws > 0 ? ndraws : 1));
    w.bytes = off;
    return w;
}

thread_local DustCtx g_dust{};

struct MtArgs {            // numpy-stream mode of post_batch_impl
    int nstream;           // 1: one stream serves all objects in order; nstar: one per object
    uint32_t *h_states;    // (nstream, MT_STATE_WORDS) in / out
    double *d_zbuf;        // normals of one group of objects
    size_t zbuf_doubles;
    int phase;             // 0: whole call; 1: up to and including the stream walk (states
                           // advanced, normals + uniforms left in the buffers); 2: the rest
};

// Keep the nsel_max best records of object s, best first (fitting.py:1029-1036).
static int clip_to_nsel_max(PostWs &w, int64_t cap, int64_t a, int64_t n, int64_t keep, hipStream_t st) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_iota32, dim3(nb), dim3(256), 0, st, w.sort_in, n);
    size_t need = 0;
    HIP_TRY(rocprim::radix_sort_pairs_desc(nullptr, need, w.rp.lnp + a, w.sort_keys, w.sort_in,
                                           w.sort_perm, (size_t)n, 0, 64, st));
    if (need > w.sort_tmp_bytes)
        return fail(BRUTUS_ENOMEM, "radix-sort scratch too small (%zu > %zu)", need, w.sort_tmp_bytes);
    HIP_TRY(rocprim::radix_sort_pairs_desc(w.sort_tmp, need, w.rp.lnp + a, w.sort_keys, w.sort_in,
                                           w.sort_perm, (size_t)n, 0, 64, st));
    const unsigned kb = (unsigned)((keep + 255) / 256);
    // permute every per-record array through the (now free) lnp1-sized scratch
    double *tmp = w.lnp1;
    if (8 * keep <= cap) {           // all planes at once: two launches instead of 16
        hipLaunchKernelGGL(k_clip_gather, dim3(kb), dim3(256), 0, st, w.rp, cap, a, w.sort_perm, keep, tmp);
        hipLaunchKernelGGL(k_clip_store, dim3(kb), dim3(256), 0, st, w.rp, cap, a, keep, tmp);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    auto permute64 = [&](double *arr) -> int {
        hipLaunchKernelGGL(k_gather<double>, dim3(kb), dim3(256), 0, st, tmp, arr + a, w.sort_perm, keep);
        HIP_TRY(hipMemcpyAsync(arr + a, tmp, 8 * (size_t)keep, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    if (int rc = permute64(w.rp.lnp)) return rc;
    for (int q = 0; q < 6; ++q)
        if (int rc = permute64(w.rp.chol + (size_t)q * cap)) return rc;
    hipLaunchKernelGGL(k_gather<int32_t>, dim3(kb), dim3(256), 0, st, (int32_t *)tmp, w.rp.src + a,
                       w.sort_perm, keep);
    HIP_TRY(hipMemcpyAsync(w.rp.src + a, tmp, 4 * (size_t)keep, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipGetLastError());
    return 0;
}

static void fill_post_params(PostParams &pp, const brutus_post_params *params) {
    memcpy(&pp, params, sizeof(brutus_post_params));
    pp.ln_f_thick = log(pp.f_thick);
    pp.ln_f_halo = log(pp.f_halo);
    const double rq2 = pp.r_q_halo * pp.r_q_halo, Rs2 = pp.R_solar * pp.R_solar, Zs = pp.Z_solar;
    const double qs = pp.q_halo_inf -
                      (pp.q_halo_inf - pp.q_halo_ctr) * exp(1. - sqrt(Rs2 + Zs * Zs + rq2) / pp.r_q_halo);
    pp.inv_reff_solar2 = 1. / (Rs2 + (Zs / qs) * (Zs / qs) + pp.Rs_halo * pp.Rs_halo);
    for (int c = 0; c < 3; ++c) {
        const double s2 = pp.feh_sigma[c] * pp.feh_sigma[c];
        pp.feh_nh_isig2[c] = -0.5 / s2;
        pp.feh_c0[c] = -0.5 * log(2. * M_PI * s2);
        pp.age_isig[c] = 1. / pp.age_sigma[c];
        pp.age_c0[c] = -0.91893853320467274178 - pp.age_lnnorm[c];
    }
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
    // halo_pow (post_kernels.hpp): (1 + r)^-h = sum_n b_n r^n, b_n = b_(n-1) (-h - n + 1) / n.
    // The table form needs the series' first dropped term below 2^-54 at |r| = 1/256 and
    // reff^2 >= Rs_halo^2 >= 2^-HALO_E0 for every distance.
    const double h = 0.5 * pp.eta_halo;
    double b = 1.;
    for (int n = 1; n <= 8; ++n) {
        b *= (-h - (double)n + 1.) / (double)n;
        if (n <= 7) pp.halo_b[n - 1] = b;
    }
    const bool ok = std::isfinite(h) && fabs(b) * ldexp(1., -64) < ldexp(1., -54) &&
                    pp.Rs_halo2 >= ldexp(1., -HALO_E0) && std::isfinite(pp.c0_halo) &&
                    std::isfinite(pow(pp.inv_reff_solar2, -h)) &&
                    // (reff^2 stays finite and in the tabulated range: 0 < q(r) between q_ctr and q_inf)
                    pp.q_halo_ctr > 0. && pp.q_halo_inf > 0. && pp.r_q_halo > 0. &&
                    // (mc_sample_c takes its square roots without the x == 0 select)
                    pp.Rs_thin2 >= ldexp(1., -HALO_E0) && pp.Rs_thick2 >= ldexp(1., -HALO_E0);
    pp.halo_tbl = ok ? 1. : 0.;
}

size_t brutus_post_workspace_bytes(int nstar, int64_t capacity, int nmc) {
    if (nstar < 1 || nstar > BRUTUS_MAX_BATCH || capacity < 1 || nmc < 1) return 0;
    // sized for up to 4096 draws per object in the numpy-stream mode
    return carve_post(nullptr, nstar, capacity, nmc, 4096).bytes;
}

}  // extern "C"

namespace {
int post_batch_impl(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                      const double *d_sel_vals, const int64_t *d_sel_off, const double *d_lnprior,
                      const double *d_feh, const double *d_loga, const double *d_coords,
                      const double *d_parallax, const double *d_parallax_err,
                      const brutus_post_params *params, void *d_workspace, size_t workspace_bytes,
                      int32_t *d_out_idx, double *d_out_vals, double *h_star_out,
                      int32_t *h_flags, uint64_t *h_nbase, void *stream, const MtArgs *mt) {
    static_assert(sizeof(PostParams) == sizeof(brutus_post_params) + POST_DERIVED * sizeof(double),
                  "post params layout");
    if (nstar < 1 || nstar > BRUTUS_MAX_BATCH || capacity < 1)
        return fail(BRUTUS_EINVAL, "bad post dimensions");
    if (!d_sel_idx || !d_rec_slot || !d_sel_vals || !d_sel_off || !d_lnprior || !d_coords || !params ||
        !d_workspace || !d_out_idx || !d_out_vals || !h_star_out || !h_flags)
        return fail(BRUTUS_EINVAL, "NULL pointer");
    if (params->nmc < 1 || params->ndraws < 1 || !(params->wt_thresh > 0.))
        return fail(BRUTUS_EINVAL, "nmc, ndraws and wt_thresh must be positive");
    if ((params->has_feh && !d_feh) || (params->has_loga && !d_loga))
        return fail(BRUTUS_EINVAL, "label arrays missing");
    if (params->ndraws > 4096) return fail(BRUTUS_EINVAL, "at most 4096 draws per object");
    PostWs w = carve_post((char *)d_workspace, nstar, capacity, params->nmc, 4096);
    if (w.bytes > workspace_bytes)
        return fail(BRUTUS_ENOMEM, "post workspace too small: need %zu bytes, got %zu", w.bytes,
                    workspace_bytes);
    PostParams pp;
    fill_post_params(pp, params);
    hipStream_t st = (hipStream_t)stream;
    Timer tm(st);
    const dim3 g2(PCH, nstar), blk(TILE);
    const int phase = mt ? mt->phase : 0;
    if (phase != 2) {
    DustCtx dc = g_dust;                 // one-shot: set by brutus_post_set_dust on this thread
    g_dust = DustCtx{};
    if (dc.d_los && (dc.nd < 2 || dc.nd > 4096)) return fail(BRUTUS_EINVAL, "bad dust table");
    hipLaunchKernelGGL(k_post_geom, dim3((nstar + 63) / 64), dim3(64), 0, st, pp, nstar, d_coords,
                       d_parallax, d_parallax_err, dc, w.geom);
    tm.begin("k_post_lnp1");
    hipLaunchKernelGGL(k_post_lnp1, g2, blk, 0, st, pp, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off,
                       w.geom, d_lnprior, d_feh, d_loga, w.lnp1, w.part);
    tm.end();
    tm.begin("k_post_cut2");
    hipLaunchKernelGGL(k_post_count2, g2, blk, 0, st, log(pp.wt_thresh), d_sel_off, w.lnp1, w.part,
                       w.counts, w.mask);
    hipLaunchKernelGGL(k_post_offsets, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, pp, nstar, w.counts,
                       w.offsets, w.off2, w.nbase, w.flags, w.nsel);
    hipLaunchKernelGGL(k_post_scatter2, g2, blk, 0, st, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off,
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
    }      // phase != 2
    const dim3 gdraw((pp.ndraws + 63) / 64, nstar);
    if (!mt) {
        tm.begin("k_post_mc");
        {
            const int nitem = PCH * nstar;
            HIP_TRY(hipMemsetAsync(w.mc_counter, 0, 4, st));
            hipLaunchKernelGGL(k_post_order, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, 0, nstar, w.nsel, w.mc_order);
            hipLaunchKernelGGL(pp.halo_tbl != 0. ? k_post_mc<true> : k_post_mc<false>,
                               dim3(nitem < MC_SLOTS ? nitem : MC_SLOTS), blk, 0, st, pp,
                               capacity, 0, nitem, w.mc_counter, (const double *)nullptr,
                               (const int64_t *)nullptr, w.mc_stage, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off,
                               w.off2, w.nsel, w.nbase, w.flags, w.geom, d_feh, d_loga, w.rp,
                               w.part_max, w.part_chi2, (const int32_t *)w.mc_order);
        }
        tm.end();
        tm.begin("k_post_cdf");
        hipLaunchKernelGGL(k_post_evid_part, g2, blk, 0, st, 0, w.off2, w.nsel, w.flags, w.part_max,
                           w.part_chi2, w.rp, w.part);
        hipLaunchKernelGGL(k_post_wt_part, g2, blk, 0, st, 0, w.off2, w.nsel, w.flags, w.part_max,
                           w.part_chi2, w.part, w.rp, w.part_w);
        hipLaunchKernelGGL(k_post_cdf, g2, blk, 0, st, 0, w.off2, w.nsel, w.flags, w.part_max,
                           w.part_chi2, w.part, w.part_w, w.rp, w.cdf, w.star_out);
        tm.end();
        tm.begin("k_post_draw");
        hipLaunchKernelGGL(k_post_draw, gdraw, dim3(64), 0, st, pp, 0, (const double *)nullptr,
                           (const int64_t *)nullptr, (const double *)nullptr, capacity, d_sel_idx, d_rec_slot,
                           d_sel_vals, d_sel_off, w.off2, w.nsel, w.nbase, w.flags, w.geom, d_feh,
                           d_loga, w.rp, w.cdf, w.star_out, d_out_idx, d_out_vals, ZMap{});
        tm.end();
    } else {
        // numpy's own stream (mt_kernels.hpp): objects are served in groups whose normals fit
        // the caller's buffer; a group's stream walk, Monte Carlo integral, cdf and draws run
        // before the next group overwrites the buffer.
        // Phases (brutus_post_batch_numpy_phase): 1 stops after the stream walk of the ONE
        // group that must hold all objects, 2 picks up from the buffers phase 1 left --
        // the caller runs phase 2 of batch k beside phase 1 of batch k + 1 (second
        // workspace and buffer), since the generator state is final after the walk.
        std::vector<int64_t> hn(nstar), nnorm(nstar), zoff(nstar);
        std::vector<int> hpos(mt->nstream);
        const int nuni = pp.ndraws * (pp.return_distreds ? 2 : 1);
        if (phase != 2) {
            HIP_TRY(hipMemcpyAsync(hn.data(), w.nsel, 8 * (size_t)nstar, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(w.mt_states, mt->h_states,
                                   sizeof(uint32_t) * (size_t)mt->nstream * MT_STATE_WORDS,
                                   hipMemcpyHostToDevice, st));
            HIP_TRY(hipStreamSynchronize(st));
            for (int s = 0; s < nstar; ++s) nnorm[s] = 3 * (int64_t)pp.nmc * hn[s];
            for (int g = 0; g < mt->nstream; ++g)
                hpos[g] = (int)mt->h_states[(size_t)g * MT_STATE_WORDS + MT_N];
        }
        // the caller's buffer: the first eighth (at least 64 MB) is scratch of the parallel
        // stream walk (bitmap, sub-stream windows ...), the rest holds the normals
        size_t zscratch = (mt->zbuf_doubles * 8 / 8 + 255) & ~(size_t)255;
        if (zscratch < ((size_t)64 << 20)) zscratch = (size_t)64 << 20;
        if (zscratch > mt->zbuf_doubles * 8 / 2) zscratch = 0;
        double *zbase = mt->d_zbuf + zscratch / 8;
        const size_t zdoubles = mt->zbuf_doubles - zscratch / 8;
        std::vector<int32_t> seg(nstar + 1);
        for (int s0 = 0; s0 < nstar;) {
            int s1 = s0;
            int64_t used = 0;
            while (phase != 2 && s1 < nstar) {
                const int64_t need = ((nnorm[s1] + 1) & ~(int64_t)1) + 2;      // even, padded
                if (used + need > (int64_t)zdoubles) break;
                zoff[s1] = used;
                used += need;
                ++s1;
            }
            if (phase == 2) s1 = nstar;
            if (s1 == s0)
                return fail(BRUTUS_ENOMEM, "normal buffer too small: object %d needs %lld doubles, "
                            "buffer holds %zu", s0, (long long)nnorm[s0] + 3, zdoubles);
            if (phase == 1 && s1 < nstar)
                return fail(BRUTUS_ENOMEM, "normal buffer too small for one group (%d of %d objects "
                            "fit): use the whole-call form", s1, nstar);
            const int ng = s1 - s0;
            int nseg;
            uint32_t *d_states;
            if (mt->nstream == 1) {
                nseg = 1;
                seg[0] = s0;
                seg[1] = s1;
                d_states = w.mt_states;
            } else {
                nseg = ng;
                for (int q = 0; q <= ng; ++q) seg[q] = s0 + q;
                d_states = w.mt_states + (size_t)s0 * MT_STATE_WORDS;
            }
            // One walk over the stream (pass 1 leaves the normals in the buffer as pairs per
            // sub-stream, 16 bytes per generated slot, read through segment lists) when the
            // whole call is one group, the 8 x 8 integrator applies and the buffer holds the
            // slots; otherwise the flat layout of two walks.
            static const int use_mapped = env_int("BRUTUS_MT_ONE_WALK", 1);
            static const int use_arr_ = env_int("BRUTUS_POST_MC_ARR", 1);
            const bool try_mapped = use_mapped && use_arr_ && pp.nmc <= MCA_NMC && s0 == 0 && s1 == nstar;
            MtEmitLaunch el{};
            hipEvent_t uni_event = nullptr;
            if (phase != 2) {
            HIP_TRY(hipMemcpyAsync(w.mt_nnorm, nnorm.data(), 8 * (size_t)nstar, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(w.mt_zoff, zoff.data(), 8 * (size_t)nstar, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(w.mt_seg, seg.data(), 4 * (size_t)(nseg + 1), hipMemcpyHostToDevice, st));
            {
                std::vector<int32_t> segv(seg.begin(), seg.begin() + nseg + 1);
                std::vector<int> p0(nseg);
                for (int q = 0; q < nseg; ++q) p0[q] = hpos[mt->nstream == 1 ? 0 : s0 + q];
                if (int rc = mt_walk(nseg, segv, d_states, p0, nnorm, w.mt_seg, w.mt_nnorm, w.mt_zoff, zbase,
                                     nuni, w.mt_uni, (char *)mt->d_zbuf, zscratch, nstar, st, tm,
                                     phase == 1, try_mapped ? (double2 *)zbase : (double2 *)nullptr,
                                     try_mapped ? zdoubles / 2 : 0, &el)) {
                    fire_after_jump(st);
                    return rc;
                }
                fire_after_jump(st);          // (no jump taken: the sequential walker)
                for (int q = 0; q < nseg; ++q) hpos[mt->nstream == 1 ? 0 : s0 + q] = p0[q];
            }
            }      // phase != 2
            if (phase == 1) {
                HIP_TRY(hipMemcpyAsync(mt->h_states, w.mt_states,
                                       sizeof(uint32_t) * (size_t)mt->nstream * MT_STATE_WORDS,
                                       hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                tm.collect();
                return 0;
            }
            if (phase == 2) {       // the pass phase 1 left for us: normals / uniforms to their places
                bool have = false;
                {
                    std::lock_guard<std::mutex> lk(g_emit_mu);
                    auto it = g_emit.find((const void *)mt->d_zbuf);
                    if (it != g_emit.end()) {
                        el = it->second;
                        g_emit.erase(it);
                        have = true;
                    }
                }
                // The uniform slots (few workgroups, each walking a sub-stream: latency, not
                // work) go to a side stream beside the Monte Carlo integral; the draws wait
                // for them.  (With kernel timing on, everything stays on the one stream.)
                if (have && el.uni_only && !g_timing) {
                    thread_local hipStream_t side = nullptr;
                    thread_local hipEvent_t ev_in = nullptr, ev_out = nullptr;
                    if (!side) {
                        HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
                        HIP_TRY(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
                        HIP_TRY(hipEventCreateWithFlags(&ev_out, hipEventDisableTiming));
                    }
                    HIP_TRY(hipEventRecord(ev_in, st));             // (whatever the caller queued)
                    HIP_TRY(hipStreamWaitEvent(side, ev_in, 0));
                    // segment lists on the main stream (the integral needs them) ...
                    hipLaunchKernelGGL(k_mt_segments, dim3((unsigned)el.nobj), dim3(64), 0, st, el.nstream,
                                       el.seg, el.nnorm, el.gauss0, el.subs, el.subbase, el.bitbase,
                                       el.sblo, el.bits, el.pre, el.objs, el.zm.zloc,
                                       const_cast<int64_t *>(el.zm.seg_pair0),
                                       const_cast<int64_t *>(el.zm.seg_addr),
                                       const_cast<int64_t *>(el.zm.seg_lo), const_cast<int32_t *>(el.zm.nseg),
                                       const_cast<double *>(el.zm.cached), const_cast<int32_t *>(el.zm.c));
                    // ... the uniforms on the side stream
                    hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)el.Ktot), dim3(MT_PT), 0, side, el.Ktot,
                                       el.subs, el.win, el.bits, el.bitbase, el.sblo, el.pre, el.seg,
                                       el.objs, el.nnorm, el.zoff, el.Z, el.nuni, el.U, el.endgauss, 1);
                    HIP_TRY(hipEventRecord(ev_out, side));
                    uni_event = ev_out;
                } else if (have) {
                    launch_mt_emit(el, st, tm);
                }
            }
            tm.begin("k_post_mc");
            {
                const int nitem = PCH * ng;
                HIP_TRY(hipMemsetAsync(w.mc_counter, 0, 4, st));
                hipLaunchKernelGGL(k_post_order, dim3(1), dim3(BRUTUS_MAX_BATCH), 0, st, s0, s1, w.nsel, w.mc_order);
                static const int use_arr = env_int("BRUTUS_POST_MC_ARR", 1);
                static const int arr_persistent = env_int("BRUTUS_POST_MC_ARR_PERSISTENT", 0);
                if (use_arr && pp.nmc <= MCA_NMC)
                    hipLaunchKernelGGL(pp.halo_tbl != 0. ? k_post_mc_arr<true> : k_post_mc_arr<false>,
                                       dim3(arr_persistent ? (nitem < MC_SLOTS ? nitem : MC_SLOTS) : nitem), blk,
                                       sizeof(double) * (TILE / 64) * MCA_R * 3 * pp.nmc,
                                       st, pp, capacity, PCH * s0, PCH * s1,
                                       arr_persistent ? w.mc_counter : (unsigned int *)nullptr,
                                       (const double *)zbase, (const int64_t *)w.mt_zoff, d_sel_idx, d_rec_slot,
                                       d_sel_vals, d_sel_off, w.off2, w.nsel, w.flags, w.geom, d_feh,
                                       d_loga, w.rp, w.part_max, w.part_chi2, el.zm, (const int32_t *)w.mc_order);
                else
                    hipLaunchKernelGGL(pp.halo_tbl != 0. ? k_post_mc<true> : k_post_mc<false>,
                                       dim3(nitem < MC_SLOTS ? nitem : MC_SLOTS), blk, 0, st,
                                       pp, capacity, PCH * s0, PCH * s1, w.mc_counter,
                                       (const double *)zbase, (const int64_t *)w.mt_zoff, w.mc_stage,
                                       d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off, w.off2, w.nsel, w.nbase, w.flags,
                                       w.geom, d_feh, d_loga, w.rp, w.part_max, w.part_chi2,
                                       (const int32_t *)w.mc_order);
            }
            tm.end();
            const dim3 gg(PCH, ng);
            tm.begin("k_post_cdf");
            hipLaunchKernelGGL(k_post_evid_part, gg, blk, 0, st, s0, w.off2, w.nsel, w.flags, w.part_max,
                               w.part_chi2, w.rp, w.part);
            hipLaunchKernelGGL(k_post_wt_part, gg, blk, 0, st, s0, w.off2, w.nsel, w.flags, w.part_max,
                               w.part_chi2, w.part, w.rp, w.part_w);
            hipLaunchKernelGGL(k_post_cdf, gg, blk, 0, st, s0, w.off2, w.nsel, w.flags, w.part_max,
                               w.part_chi2, w.part, w.part_w, w.rp, w.cdf, w.star_out);
            tm.end();
            tm.begin("k_post_draw");
            if (uni_event) HIP_TRY(hipStreamWaitEvent(st, uni_event, 0));
            hipLaunchKernelGGL(k_post_draw, dim3((pp.ndraws + 63) / 64, ng), dim3(64), 0, st, pp, s0,
                               (const double *)zbase, (const int64_t *)w.mt_zoff,
                               (const double *)w.mt_uni, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off,
                               w.off2, w.nsel, w.nbase, w.flags, w.geom, d_feh, d_loga, w.rp, w.cdf,
                               w.star_out, d_out_idx, d_out_vals, el.zm);
            tm.end();
            HIP_TRY(hipStreamSynchronize(st));     // the host arrays of this group are reused
            s0 = s1;
        }
        if (phase == 0)
            HIP_TRY(hipMemcpyAsync(mt->h_states, w.mt_states,
                                   sizeof(uint32_t) * (size_t)mt->nstream * MT_STATE_WORDS,
                                   hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_star_out, w.star_out, 8 * 4 * (size_t)nstar, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h_flags, w.flags, 4 * (size_t)nstar, hipMemcpyDeviceToHost, st));
    if (h_nbase)
        HIP_TRY(hipMemcpyAsync(h_nbase, w.nbase, 8 * ((size_t)nstar + 1), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    tm.collect();
    return 0;
}

}  // namespace

extern "C" {

int brutus_post_batch(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                      const double *d_sel_vals, const int64_t *d_sel_off, const double *d_lnprior,
                      const double *d_feh, const double *d_loga, const double *d_coords,
                      const double *d_parallax, const double *d_parallax_err,
                      const brutus_post_params *params, void *d_workspace, size_t workspace_bytes,
                      int32_t *d_out_idx, double *d_out_vals, double *h_star_out,
                      int32_t *h_flags, uint64_t *h_nbase, void *stream) {
    return post_batch_impl(nstar, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off, d_lnprior, d_feh, d_loga,
                           d_coords, d_parallax, d_parallax_err, params, d_workspace, workspace_bytes,
                           d_out_idx, d_out_vals, h_star_out, h_flags, h_nbase, stream, nullptr);
}

int brutus_post_batch_numpy(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                            const double *d_sel_vals, const int64_t *d_sel_off,
                            const double *d_lnprior, const double *d_feh, const double *d_loga,
                            const double *d_coords, const double *d_parallax,
                            const double *d_parallax_err, const brutus_post_params *params,
                            void *d_workspace, size_t workspace_bytes, int32_t *d_out_idx,
                            double *d_out_vals, double *h_star_out, int32_t *h_flags,
                            int nstream, uint32_t *h_states, double *d_zbuf, size_t zbuf_doubles,
                            void *stream) {
    if ((nstream != 1 && nstream != nstar) || !h_states || !d_zbuf || zbuf_doubles < 1024)
        return fail(BRUTUS_EINVAL, "bad numpy-stream arguments");
    MtArgs mt{nstream, h_states, d_zbuf, zbuf_doubles, 0};
    return post_batch_impl(nstar, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off, d_lnprior, d_feh, d_loga,
                           d_coords, d_parallax, d_parallax_err, params, d_workspace, workspace_bytes,
                           d_out_idx, d_out_vals, h_star_out, h_flags, nullptr, stream, &mt);
}

int brutus_post_batch_numpy_phase(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                                  const double *d_sel_vals, const int64_t *d_sel_off,
                                  const double *d_lnprior, const double *d_feh, const double *d_loga,
                                  const double *d_coords, const double *d_parallax,
                                  const double *d_parallax_err, const brutus_post_params *params,
                                  void *d_workspace, size_t workspace_bytes, int32_t *d_out_idx,
                                  double *d_out_vals, double *h_star_out, int32_t *h_flags,
                                  int nstream, uint32_t *h_states, double *d_zbuf,
                                  size_t zbuf_doubles, int phase, void *stream) {
    if ((nstream != 1 && nstream != nstar) || !h_states || !d_zbuf || zbuf_doubles < 1024 ||
        phase < 0 || phase > 2)
        return fail(BRUTUS_EINVAL, "bad numpy-stream arguments");
    MtArgs mt{nstream, h_states, d_zbuf, zbuf_doubles, phase};
    return post_batch_impl(nstar, capacity, d_sel_idx, d_rec_slot, d_sel_vals, d_sel_off, d_lnprior, d_feh, d_loga,
                           d_coords, d_parallax, d_parallax_err, params, d_workspace, workspace_bytes,
                           d_out_idx, d_out_vals, h_star_out, h_flags, nullptr, stream, &mt);
}

int brutus_post_set_after_jump(void (*fn)(void *), void *arg) {
    g_after_jump = AfterJump{fn, arg};
    return 0;
}

int brutus_post_set_dust(const double *d_los, const int32_t *d_ok, int nd, double offset,
                         double scale, double smooth, double scatter) {
    g_dust = DustCtx{d_los, d_ok, nd, offset, scale, smooth, scatter};
    return 0;
}

int brutus_set_mt_jump(const uint32_t *h_polys, int npoly, int64_t stride0, int64_t stride1) {
    if (!h_polys || npoly < 2 || npoly > 16 || stride0 != MT_J || stride1 != MT_J * MT_L1)
        return fail(BRUTUS_EINVAL, "jump polynomials must be for strides %lld, %lld * 2^r words",
                    (long long)MT_J, (long long)(MT_J * MT_L1));
    std::lock_guard<std::mutex> lk(g_mt_mu);
    g_mt_polys.assign(h_polys, h_polys + (size_t)npoly * MT_N);
    return 0;
}

int brutus_debug_mt_stream(int nobj, int nstream, uint32_t *h_states, const int64_t *h_nnorm,
                           int nuni, double *d_z, double *d_u, void *stream) {
    // test hook: walk the stream(s) for objects that need h_nnorm[o] normals and nuni
    // uniforms each; normals of object o at d_z + sum of the (even-rounded + 2) counts before it
    if (nobj < 1 || (nstream != 1 && nstream != nobj) || !h_states || !h_nnorm || !d_z || !d_u)
        return fail(BRUTUS_EINVAL, "bad arguments");
    hipStream_t st = (hipStream_t)stream;
    std::vector<int64_t> zoff(nobj);
    std::vector<int32_t> seg(nobj + 1);
    int64_t used = 0;
    for (int o = 0; o < nobj; ++o) {
        zoff[o] = used;
        used += ((h_nnorm[o] + 1) & ~(int64_t)1) + 2;
    }
    const int nseg = nstream == 1 ? 1 : nobj;
    if (nstream == 1) {
        seg[0] = 0;
        seg[1] = nobj;
    } else {
        for (int q = 0; q <= nobj; ++q) seg[q] = q;
    }
    uint32_t *d_states;
    int64_t *d_nn, *d_zo;
    int32_t *d_seg;
    HIP_TRY(hipMalloc(&d_states, sizeof(uint32_t) * (size_t)nstream * MT_STATE_WORDS));
    HIP_TRY(hipMalloc(&d_nn, 8 * (size_t)nobj));
    HIP_TRY(hipMalloc(&d_zo, 8 * (size_t)nobj));
    HIP_TRY(hipMalloc(&d_seg, 4 * ((size_t)nobj + 1)));
    HIP_TRY(hipMemcpyAsync(d_states, h_states, sizeof(uint32_t) * (size_t)nstream * MT_STATE_WORDS, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_nn, h_nnorm, 8 * (size_t)nobj, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_zo, zoff.data(), 8 * (size_t)nobj, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_seg, seg.data(), 4 * ((size_t)nseg + 1), hipMemcpyHostToDevice, st));
    {
        int64_t tot = 0;
        for (int o = 0; o < nobj; ++o) tot += h_nnorm[o];
        const size_t sbytes = ((size_t)256 << 20) + (size_t)tot / 4 + (size_t)nobj * 65536;
        char *scratch = nullptr;
        HIP_TRY(hipMalloc(&scratch, sbytes));
        std::vector<int32_t> segv(seg.begin(), seg.begin() + nseg + 1);
        std::vector<int> p0(nseg);
        for (int g = 0; g < nseg; ++g) p0[g] = (int)h_states[(size_t)g * MT_STATE_WORDS + MT_N];
        std::vector<int64_t> nn(h_nnorm, h_nnorm + nobj);
        Timer tm(st);
        int rc = mt_walk(nseg, segv, d_states, p0, nn, d_seg, d_nn, d_zo, d_z, nuni, d_u, scratch, sbytes,
                         nobj, st, tm);
        HIP_TRY(hipStreamSynchronize(st));
        tm.collect();
        (void)hipFree(scratch);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(h_states, d_states, sizeof(uint32_t) * (size_t)nstream * MT_STATE_WORDS, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(d_states);
    (void)hipFree(d_nn);
    (void)hipFree(d_zo);
    (void)hipFree(d_seg);
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

namespace zig_host {
#define ZIG_TABLE_QUAL static const
#include "zig_table.inc"
#undef ZIG_TABLE_QUAL
}   // namespace zig_host

int brutus_debug_fit_stats(int64_t *calls, int64_t *repeated) {
    if (calls) *calls = g_fit_calls.load();
    if (repeated) *repeated = g_fit_retries.load();
    return 0;
}

int brutus_debug_zig_table(double *h_x, double *h_y, int n) {
    if (!h_x || !h_y || n != zig_host::ZIG_N + 1) return fail(BRUTUS_EINVAL, "bad arguments");
    for (int k = 0; k < n; ++k) {
        h_x[k] = zig_host::kZigX[k];
        h_y[k] = zig_host::kZigY[k];
    }
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

int brutus_debug_galprior_mc(const brutus_post_params *params, int n, const double *d_dist,
                             const double *d_coord, const double *d_feh, const double *d_loga,
                             double *d_out, void *stream) {
    if (!params || !d_dist || !d_coord || !d_feh || !d_loga || !d_out || n <= 0)
        return fail(BRUTUS_EINVAL, "bad arguments");
    PostParams pp;
    fill_post_params(pp, params);
    hipStream_t st = (hipStream_t)stream;
    StarGeom *geom = nullptr;
    HIP_TRY(hipMalloc(&geom, sizeof(StarGeom)));
    DustCtx dc{};
    hipLaunchKernelGGL(k_post_geom, dim3(1), dim3(64), 0, st, pp, 1, d_coord, (const double *)nullptr,
                       (const double *)nullptr, dc, geom);
    hipLaunchKernelGGL(pp.halo_tbl != 0. ? k_debug_galprior_mc<true> : k_debug_galprior_mc<false>,
                       dim3((n + 255) / 256), dim3(256), 0, st, pp, n, d_dist, geom, d_feh, d_loga, d_out);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(geom);
    HIP_TRY(e);
    HIP_TRY(e2);
    return 0;
}

int brutus_debug_galprior_sl(const brutus_post_params *params, int n, const double *d_dist,
                             const double *d_coord, const double *d_feh, const double *d_loga,
                             double *d_out, int32_t *d_used, void *stream) {
    if (!params || !d_dist || !d_coord || !d_feh || !d_loga || !d_out || !d_used || n <= 0)
        return fail(BRUTUS_EINVAL, "bad arguments");
    PostParams pp;
    fill_post_params(pp, params);
    if (pp.halo_tbl == 0.) return fail(BRUTUS_EINVAL, "these parameters do not admit the halo table: no sightline table");
    hipStream_t st = (hipStream_t)stream;
    StarGeom *geom = nullptr;
    HIP_TRY(hipMalloc(&geom, sizeof(StarGeom)));
    DustCtx dc{};
    hipLaunchKernelGGL(k_post_geom, dim3(1), dim3(64), 0, st, pp, 1, d_coord, (const double *)nullptr,
                       (const double *)nullptr, dc, geom);
    hipLaunchKernelGGL(k_debug_galprior_sl, dim3((n + TILE - 1) / TILE), dim3(TILE), 0, st, pp, n, d_dist, geom,
                       d_feh, d_loga, d_out, d_used);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(st);
    hipFree(geom);
    HIP_TRY(e);
    HIP_TRY(e2);
    return 0;
}

int brutus_calibrate_traffic(const float *d_in, double *d_out, int64_t n, void *stream) {
    if (!d_in || !d_out || n <= 0) return fail(BRUTUS_EINVAL, "bad calibration arguments");
    hipLaunchKernelGGL(k_calib_stream, dim3(4096), dim3(TILE), 0, (hipStream_t)stream, d_in, d_out, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_calibrate_copy16(const void *d_in, void *d_out, int64_t nbytes, void *stream) {
    if (!d_in || !d_out || nbytes < 16 || (nbytes & 15)) return fail(BRUTUS_EINVAL, "bad calibration arguments");
    const int64_t n = nbytes / 16;
    if ((n + TILE - 1) / TILE > 0x7fffffff) return fail(BRUTUS_EINVAL, "calibration buffer too large");
    hipLaunchKernelGGL(k_calib_copy16, dim3((unsigned)((n + TILE - 1) / TILE)), dim3(TILE), 0,
                       (hipStream_t)stream, (const calib_f4 *)d_in, (calib_f4 *)d_out, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int brutus_calibrate_issue(int kind, int iters, int waves_per_simd, float *d_scratch,
                           int64_t scratch_floats, void *stream) {
    if (kind < 0 || kind > 2 || iters <= 0 || waves_per_simd < 1 || waves_per_simd > 8 || !d_scratch)
        return fail(BRUTUS_EINVAL, "bad calibration arguments");
    int dev = 0, ncu = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = ncu * waves_per_simd;       // 256 threads = one wave on each of a CU's four SIMDs
    if (scratch_floats < (int64_t)blocks * 256) return fail(BRUTUS_EINVAL, "calibration scratch too small");
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(k_calib_issue<0>, dim3(blocks), dim3(256), 0, st, d_scratch, iters);
    else if (kind == 1) hipLaunchKernelGGL(k_calib_issue<1>, dim3(blocks), dim3(256), 0, st, d_scratch, iters);
    else hipLaunchKernelGGL(k_calib_issue<2>, dim3(blocks), dim3(256), 0, st, d_scratch, iters);
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
    if (!d_x || !d_y || n <= 0 || which < 0 || which > 9) return fail(BRUTUS_EINVAL, "bad arguments");
    hipLaunchKernelGGL(k_debug_math, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, which, d_x, d_y, n);
    HIP_TRY(hipGetLastError());
    return 0;
}


int brutus_debug_copy(void *d_workspace, size_t workspace_bytes, int64_t nmodel, int nfilt, int nstar,
                      int which, void *d_dst, size_t nbytes, void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    if (!d_workspace || !d_dst) return fail(BRUTUS_EINVAL, "NULL pointer");
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes) return fail(BRUTUS_ENOMEM, "workspace too small");
    const size_t plane = (size_t)nstar * (size_t)nmodel;
    const void *src = nullptr;
    size_t have = 0;
    switch (which) {
        case 2: src = w.lnlp32; have = 4 * plane; break;        // float32 statistics
        case 3: src = w.lnpr32; have = 4 * plane; break;
        case 4: src = w.aud; have = 4 * (size_t)nstar * 4; break;
        case 5: src = w.s32; have = sizeof(Star32) * (size_t)nstar; break;
        case 6: src = w.thr_cull; have = 8 * (size_t)nstar; break;
        case 7: src = w.thr_sel; have = 8 * (size_t)nstar; break;
        case 8: src = w.st32; have = 4 * (size_t)nstar * NV32; break;
        case 9: src = w.status; have = 4 * (size_t)nstar; break;
        default: return fail(BRUTUS_EINVAL, "unknown array %d", which);
    }
    if (nbytes > have) return fail(BRUTUS_EINVAL, "array %d holds %zu bytes, %zu requested", which, have, nbytes);
    HIP_TRY(hipMemcpyAsync(d_dst, src, nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

int brutus_debug_sizeof_star32(void) { return (int)sizeof(Star32); }

int brutus_debug_pre32_time(void *d_workspace, size_t workspace_bytes, const float *d_grid_soa,
                            int64_t nmodel, int nfilt, int nstar, const brutus_params *params,
                            int form, int reps, float *h_ms, void *stream) {
    if (int rc = check_common(nmodel, nfilt, nstar)) return rc;
    if (!d_workspace || !d_grid_soa || !h_ms || reps < 1) return fail(BRUTUS_EINVAL, "bad arguments");
    DevParams p;
    if (int rc = make_params(params, p)) return rc;
    Workspace w = carve((char *)d_workspace, nmodel, nstar, true);
    if (w.bytes > workspace_bytes) return fail(BRUTUS_ENOMEM, "workspace too small");
    const int nb = brutus_padded_filters(nfilt);
    if (!brutus_i_pre32s_bands(nb)) return fail(BRUTUS_EINVAL, "no star-lane pass for %d bands", nb);
    const bool rvf = p.rvmin == p.rvmax && p.rvmin == p.rv_mean;
    P32 q;
    q.avmin = (float)p.avmin; q.avmax = (float)p.avmax; q.rvmin = (float)p.rvmin; q.rvmax = (float)p.rvmax;
    q.av_mean = (float)p.av_mean; q.av_ivar = (float)p.av_ivar; q.rv_mean = (float)p.rv_mean;
    q.rv_ivar = (float)p.rv_ivar;
    q.mtol_hi = (float)(p.mtol * 1.002 + 1e-4);
    q.mtol_lo = (float)(p.mtol * 0.998 - 1e-4);
    q.dim_prior = p.dim_prior;
    q.nfilt = nfilt;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    for (int k = 0; k < reps + 1; ++k) {
        if (k == 1) HIP_TRY(hipEventRecord(a, st));
        if (brutus_i_pre32s_launch(nb, form, rvf ? 1 : 0, d_grid_soa, nmodel, pad_models(nmodel), nstar, nstar,
                                   w.ids_all, w.s32, &q, w.lnlp32, w.lnpr32, w.part32, st))
            return fail(BRUTUS_EHIP, "star-lane float32 pass: launch failed");
    }
    HIP_TRY(hipEventRecord(b, st));
    HIP_TRY(hipEventSynchronize(b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    *h_ms = ms / reps;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
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
