/*
 * brutus_amd.h -- C ABI of the MI355X (gfx950) brute-force photometric fitter.
 *
 * The upstream reference (joshspeagle/brutus) is pure Python; its "native" layer
 * is four numba-jitted loops.  This library replaces exactly that layer plus
 * the full-grid part of `loglike`/`lnpost`:
 *
 *   brutus/utils.py:286-347     _get_seds
 *   brutus/fitting.py:34-271    _optimize_fit_mag
 *   brutus/fitting.py:274-427   _optimize_fit_flux
 *   brutus/fitting.py:430-576   _get_sed_mle
 *   brutus/fitting.py:579-820   loglike            (whole function)
 *   brutus/fitting.py:976-991   lnpost: parallax clip + first `wt_thresh` cut
 *   brutus/utils.py:130-176     _chisquare_logpdf
 *   brutus/cluster.py:336-414   isochrone_loglike hot block (brutus_cluster_lnl)
 *
 * Conventions
 *   - every pointer prefixed d_ is a DEVICE pointer (HBM), h_ is a host pointer;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - all functions return 0 on success, a negative BRUTUS_E* code on failure;
 *     brutus_last_error() returns a human-readable message for the calling thread;
 *   - no allocation happens inside the hot calls: the caller owns every buffer
 *     (sizes from the *_bytes() queries) -- PyTorch tensors in the Python host.
 *
 * The Python host (brutus_amd/fitting.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 */
#ifndef BRUTUS_AMD_H
#define BRUTUS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRUTUS_ABI_VERSION 4
#ifndef BRUTUS_MAX_FILT      /* (overridable for A/B builds of the library only: it sizes the per-star blocks) */
#define BRUTUS_MAX_FILT 64   /* bands per call: brutus_loglike_batch (full-grid outputs), offsets */
#endif
#define BRUTUS_MAX_FILT_FIT 32 /* bands per brutus_fit_batch / cluster call (register budget of the list kernels) */
#define BRUTUS_MAX_BATCH 256 /* stars per brutus_*_batch call                     */
#define BRUTUS_NVALS 11      /* lnlike, chi2, scale, av, rv, icov[00,01,02,11,12,22] */

#define BRUTUS_OK 0
#define BRUTUS_EINVAL (-1)    /* bad argument                                      */
#define BRUTUS_ENOMEM (-2)    /* workspace / record buffer too small               */
#define BRUTUS_EHIP (-3)      /* HIP runtime error (message in brutus_last_error)  */
#define BRUTUS_ENOCONV (-4)   /* iteration cap hit (the reference would spin)      */
#define BRUTUS_EPRECISION (-5) /* brutus_fit_batch: the audited float32 error bound does not hold (see there) */

/* Keyword arguments of fitting.loglike (fitting.py:579-585) plus lnpost's
 * wt_thresh (fitting.py:823-827).  av_gauss=None maps to (0, 1e6) on the host
 * exactly like fitting.py:695-696. */
typedef struct brutus_params {
    double avlim[2];
    double av_gauss[2];
    double rvlim[2];
    double rv_gauss[2];
    double ltol;           /* flux-phase tolerance; mag-phase tol = 2.5*ltol     */
    double ltol_subthresh;
    double init_thresh;    /* logl_initthresh                                     */
    double wt_thresh;      /* lnpost first cut; only used by brutus_fit_batch     */
    int32_t dim_prior;     /* logl_dim_prior                                      */
    int32_t max_iter;      /* safety cap on mag sweeps / flux iterations (0 = 65536: the reference has none) */
} brutus_params;

int brutus_abi_version(void);
const char *brutus_last_error(void);

/* ---- model grid ------------------------------------------------------------
 * `utils.load_models` (utils.py:588-591) returns models as (Nmodel, Nfilt, 3)
 * float32 = (mag, R, dR/dRv) per band.  The device grid blob holds the
 * coefficients twice -- a band-major structure-of-arrays
 * [nfilt_pad][3][nmodel_pad] that the full-grid scans stream with coalesced
 * loads, and a model-major [nmodel_pad][nfilt_pad][3] copy for the kernels
 * that gather single models -- followed by the band-major float64 table of
 * unreddened model fluxes 10^(-0.4 mag) (fitting.py:529).
 * nfilt_pad = the compiled band count >= nfilt (see brutus_padded_filters: 8, 12, 16, 24, 32 and,
 * for brutus_loglike_batch only, 48 and 64),
 * nmodel_pad = nmodel rounded up to 256.  Padded entries are zero. */
int brutus_padded_filters(int nfilt);              /* <0 if nfilt unsupported     */
size_t brutus_grid_soa_bytes(int64_t nmodel, int nfilt);
int brutus_grid_relayout(const float *d_models_aos, int64_t nmodel, int nfilt,
                         float *d_grid_soa, void *stream);

/* ---- per-star grid likelihood ---------------------------------------------
 * Inputs for a batch of `nstar` stars (row-major, C order):
 *   d_flux, d_err  (nstar, nfilt) float64 maggies;  d_mask (nstar, nfilt) uint8;
 *   d_parallax, d_parallax_err (nstar,) float64 mas, NaN = no measurement
 *   (the notebooks' convention; `None` in the Python API maps to NaN with
 *   has_parallax=0, see fitting.py:749-756 vs :976-982).
 */
size_t brutus_workspace_bytes(int64_t nmodel, int nfilt, int nstar);

/* fitting.loglike(..., return_vals=True) for every star of the batch, full-grid
 * outputs, each plane (nstar, nmodel) float64; d_icov is (6, nstar, nmodel)
 * holding the unique entries [00, 01, 02, 11, 12, 22] of icov_sar
 * (fitting.py:563-574).  d_ndim (nstar,) int32.  h_k1/h_k2 (optional, host,
 * (nstar,) int32) receive the number of magnitude sweeps / flux iterations.
 * d_av_init / d_rv_init (optional, (nmodel,) float64): per-model starting values of the
 * magnitude phase, `av_init` / `rv_init` of fitting.py:697-703, shared by the stars of the
 * batch; NULL = the prior means av_gauss[0] / rv_gauss[0] (the reference's default). */
int brutus_loglike_batch(const float *d_grid_soa, int64_t nmodel, int nfilt,
                         int nstar, const double *d_flux, const double *d_err,
                         const uint8_t *d_mask, const double *d_parallax,
                         const double *d_parallax_err, int has_parallax,
                         const brutus_params *params, void *d_workspace,
                         size_t workspace_bytes, double *d_lnl, double *d_chi2,
                         double *d_scale, double *d_av, double *d_rv,
                         double *d_icov, int32_t *d_ndim, int32_t *h_k1,
                         int32_t *h_k2, const double *d_av_init,
                         const double *d_rv_init, void *stream);

/* The fit() hot path: loglike + lnpost's parallax clip + first wt_thresh cut
 * (fitting.py:976-991), emitting only the selected models as INDEXED RECORDS:
 *   d_rec_idx  (capacity,) int32   model index, ascending per star (= np.where order)
 *   d_rec_slot (capacity,) int32   column of that record's values in d_rec_vals
 *   d_rec_vals (BRUTUS_NVALS, capacity) float64
 *   d_rec_off  (nstar + 1,) int64  record range of star s is [off[s], off[s+1])
 * i.e. value v of record r is d_rec_vals[v * capacity + d_rec_slot[r]].  The indirection
 * exists because a value is written exactly once, where it is computed: columns
 * [0, ncand) belong to the candidates of the likelihood cull (fitting.py:758-759) in
 * candidate-list order and receive the flux-phase results (fitting.py:778-803) of those
 * that survive it; columns [ncand, ncand + nder) receive the selected models the cull
 * dropped.  Columns of candidates that fail the cull or the first cut are never referenced.
 * When Rv is pinned (rvlim[0] == rvlim[1] == rv_gauss[0]) plane 4 (rv) is NOT written:
 * every record's rv is rv_gauss[0].
 * h_counts (host, 3 x int64): [0] selected models of the batch (= off[nstar]),
 * [1] ncand, [2] columns needed = ncand + nder.  BRUTUS_ENOMEM if [2] > capacity (or
 * already [1] > capacity: then [2] is an estimate): call again with larger buffers.
 * Float32 never produces an output value or a decision here, it only proves models to be below
 * the thresholds; the error bound that rests on is AUDITED on the first call of a process and
 * every BRUTUS_AUDIT_EVERY-th (environment, default 64; 0 = never) after it -- every pair the
 * call re-evaluates in float64 anyway is compared with its float32 value -- and a call whose
 * audit finds |float32 - float64| >= the bound fails with BRUTUS_EPRECISION. */
int brutus_fit_batch(const float *d_grid_soa, int64_t nmodel, int nfilt,
                     int nstar, const double *d_flux, const double *d_err,
                     const uint8_t *d_mask, const double *d_parallax,
                     const double *d_parallax_err, int has_parallax,
                     const brutus_params *params, void *d_workspace,
                     size_t workspace_bytes, int64_t capacity,
                     int32_t *d_rec_idx, int32_t *d_rec_slot, double *d_rec_vals,
                     int64_t *d_rec_off, int32_t *d_ndim, int32_t *h_k1, int32_t *h_k2,
                     int64_t *h_counts, void *stream);

/* ---- lnpost on the device ------------------------------------------------------
 * Everything of fitting.lnpost after the first cut (fitting.py:1000-1107) and
 * the resampling tail of BruteForce._fit (fitting.py:2021-2061), for the
 * built-in priors: static lnprior + the Galactic model of pdf.gal_lnprior
 * (pdf.py:476-749; Galactocentric frame passed in, see frame_mat) + the parallax
 * likelihood.  Random numbers follow brutus_amd/rng.py (PhiloxRandomState:
 * Philox4x32-7; 53-bit uniforms; normals by a 1024-layer ziggurat, two per call):
 * normal j / uniform q are functions of (seed, j) / (seed, q), so the result is
 * what the reference produces when it is handed that object as `rstate`.
 *
 * Input = the device-resident indexed records brutus_fit_batch emitted (d_sel_idx = its
 * d_rec_idx, d_rec_slot, d_sel_vals = d_rec_vals, d_sel_off = d_rec_off, same capacity).  Output per
 * object s and draw q < ndraws:
 *   d_out_idx  (nstar, ndraws) i32      resampled model index
 *   d_out_vals (nstar, ndraws, 17) f64  scale, av, rv, cov_sar[9], lnprob,
 *                                       dist, red, dred, logwt
 *   h_star_out (nstar, 4) f64 (host)    levid, chi2min, sum of weights, Nsel
 *   h_flags    (nstar,) i32 (host)      0; an object with more than nsel_max
 *                                       survivors of the second cut is clipped to
 *                                       the nsel_max best, best first, by a device
 *                                       radix sort (fitting.py:1029-1036); a
 *                                       non-zero flag asks the caller to redo that
 *                                       object on the host (not used at present)
 *   h_nbase (nstar + 1,) u64 (host, optional)  normal-stream position at which
 *                                       each object starts; [nstar] = after the batch
 * Object s consumes 3*nmc*min(Nsel_s, nsel_max) normals starting where object
 * s-1 stopped (first at normal_base) and uniforms
 * [uniform_base + s*K, uniform_base + (s+1)*K), K = ndraws * (1 + return_distreds). */
typedef struct brutus_post_params {
    int32_t nmc, ndraws, return_distreds, has_feh, has_loga;
    int32_t per_object;   /* 1: object s draws from its own stream keyed
                             seed + object0 + s, positions from 0 (order- and
                             sharding-independent); 0: one shared stream */
    double wt_thresh, avlim[2], rvlim[2];
    int64_t nsel_max, object0;
    uint64_t seed, normal_base, uniform_base;
    double R_solar, Z_solar, R_thin, Z_thin, Rs_thin, R_thick, Z_thick, f_thick, Rs_thick;
    double Rs_halo, q_halo_ctr, q_halo_inf, r_q_halo, eta_halo, f_halo;
    double feh_mean[3], feh_sigma[3];
    double age_mean[3], age_sigma[3], age_lnnorm[3], min_age, max_age;
    /* Galactic -> Galactocentric frame of the prior (reference pdf.py:631-635 goes through
     * astropy's `Galactocentric`): x_gc [kpc] = frame_mat (row-major 3x3) @ d (cos b cos l,
     * cos b sin l, sin b) + frame_off; R = hypot(x, y), Z = z.  brutus_amd/galprior.py
     * (`astropy_frame`, `simple_frame`) builds the two frames the host knows. */
    double frame_mat[9], frame_off[3];
} brutus_post_params;

size_t brutus_post_workspace_bytes(int nstar, int64_t capacity, int nmc);
int brutus_post_batch(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                      const double *d_sel_vals, const int64_t *d_sel_off,
                      const double *d_lnprior, const double *d_feh,
                      const double *d_loga, const double *d_coords,
                      const double *d_parallax, const double *d_parallax_err,
                      const brutus_post_params *params, void *d_workspace,
                      size_t workspace_bytes, int32_t *d_out_idx,
                      double *d_out_vals, double *h_star_out, int32_t *h_flags,
                      uint64_t *h_nbase, void *stream);

/* The same with NUMPY'S OWN random stream (legacy `numpy.random.RandomState`: MT19937,
 * polar Box-Muller with the cached second deviate, `choice` = searchsorted of
 * random_sample), i.e. what the reference consumes when `rstate` is a RandomState or
 * None (fitting.py:937-944, utils.py:892-905, fitting.py:2037-2053).  h_states holds
 * `nstream` generator states in numpy's representation, 628 words each:
 * key[624], pos, has_gauss, cached_gaussian (2 words, little endian double) --
 * `RandomState.get_state()`; they are advanced in place, so `set_state` makes the
 * caller's generator continue exactly where the reference's would.
 *   nstream == 1     one stream serves the objects in order (the reference's semantics);
 *                    a stream is sequential by nature: one workgroup walks it
 *   nstream == nstar object s has its own stream (per-object seeds: sharded runs)
 * d_zbuf (zbuf_doubles float64) receives the normals of a group of objects; an object
 * needs 3 * nmc * Nsel + 3 doubles; BRUTUS_ENOMEM if a single object does not fit.  The
 * first eighth of the buffer is scratch of the stream walk.  When all objects of the call
 * fit as one group, nmc <= 64 and the rest of the buffer also holds 16 bytes per generated
 * slot (about 1.3 x the flat normals), the stream is walked once: the accepted candidates'
 * normals stay where the walk produced them and the consumers read them through per-object
 * segment lists; otherwise a second walk writes the flat array.  Same results either way. */
int brutus_post_batch_numpy(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                            const double *d_sel_vals, const int64_t *d_sel_off,
                            const double *d_lnprior, const double *d_feh,
                            const double *d_loga, const double *d_coords,
                            const double *d_parallax, const double *d_parallax_err,
                            const brutus_post_params *params, void *d_workspace,
                            size_t workspace_bytes, int32_t *d_out_idx,
                            double *d_out_vals, double *h_star_out, int32_t *h_flags,
                            int nstream, uint32_t *h_states, double *d_zbuf,
                            size_t zbuf_doubles, void *stream);

/* The same call in two halves, so that a caller can overlap them across batches: the
 * generator state of a numpy stream is final once the stream has been walked, before
 * the Monte Carlo integral that consumes the normals.  phase 1 = first and second cut,
 * covariances, stream walk (h_states advanced; normals and uniforms stay in d_zbuf /
 * the workspace; fails with BRUTUS_ENOMEM if the objects do not fit d_zbuf as ONE group
 * -- use the whole-call form then); phase 2 = Monte Carlo integral, evidence, draws,
 * outputs, with the SAME arguments, workspace and buffer, on any stream / thread, after
 * phase 1 returned.  phase 0 = brutus_post_batch_numpy. */
int brutus_post_batch_numpy_phase(int nstar, int64_t capacity, const int32_t *d_sel_idx, const int32_t *d_rec_slot,
                                  const double *d_sel_vals, const int64_t *d_sel_off,
                                  const double *d_lnprior, const double *d_feh, const double *d_loga,
                                  const double *d_coords, const double *d_parallax,
                                  const double *d_parallax_err, const brutus_post_params *params,
                                  void *d_workspace, size_t workspace_bytes, int32_t *d_out_idx,
                                  double *d_out_vals, double *h_star_out, int32_t *h_flags,
                                  int nstream, uint32_t *h_states, double *d_zbuf,
                                  size_t zbuf_doubles, int phase, void *stream);

/* Scheduling hook for callers that pipeline the two phases: `fn(arg)` is called once, on the
 * calling thread, from inside the NEXT brutus_post_batch_numpy[_phase] call of this thread,
 * as soon as the jump-ahead windows of its stream walk are complete (the call's stream is
 * drained first), at the latest before that call returns or fails after its walk.  The
 * jump-ahead kernels need whole compute units (125 KB of LDS per workgroup) and starve
 * behind a long kernel of another stream: a caller holds back phase 2 of the previous batch
 * until the hook fires, and the two then overlap where they can (stream walk: LDS-bound,
 * Monte Carlo integral: float64-issue-bound).  fn == NULL clears a pending hook. */
int brutus_post_set_after_jump(void (*fn)(void *), void *arg);

/* Line-of-sight dust prior for the NEXT brutus_post_batch / brutus_post_batch_numpy call
 * of the calling thread (one-shot): the reference's `dust_lnprior` (pdf.py:752-840,
 * Gaussian in Av around the profile interpolated at the distance) with the profile of every
 * object's sightline supplied by the caller instead of the Bayestar map:
 *   d_los (nstar, 3, nd) f64: dist [kpc], Av_mean, Av_err;  d_ok (nstar,) i32: 0 = no
 *   coverage (flat prior, like the reference).  Applied at the MLE point (fitting.py:1009-
 *   1010) and to every Monte Carlo sample (fitting.py:1084-1085). */
int brutus_post_set_dust(const double *d_los, const int32_t *d_ok, int nd,
                         double offset, double scale, double smooth, double scatter);

/* Jump-ahead polynomials of MT19937 (brutus_amd/mt_jump.npz, made and checked against
 * numpy by tools/gen_mt_jump.py): h_polys = uint32 (npoly, 624), x^(stride - 1) mod phi
 * for stride0 = 2 096 640 words and 128 * stride0 * 2^r, r = 0 .. npoly - 2.  With them
 * loaded one stream is walked by many workgroups (sub-streams of stride0 words, their
 * start windows from a doubling tree of jumps); without them by one. */
int brutus_set_mt_jump(const uint32_t *h_polys, int npoly, int64_t stride0,
                       int64_t stride1);

/* ---- cluster mode ------------------------------------------------------------
 * Hot block of cluster.isochrone_loglike (cluster.py:336-414): for nobj objects
 * and npts isochrone points (all secondary-mass-fraction slices concatenated,
 * masked points removed by the caller),
 *   d_pts_flux (npts, nfilt) f64  model fluxes 10^(-0.4 cmd_sed), NaN = no model
 *   d_pts_lnw  (npts,)       f64  ln(grad_mini) + ln(grad_smf)   (cluster.py:397-403)
 *   d_phot, d_ivar (nobj, nfilt)  offset-scaled fluxes and 1/err^2 (0 for missing bands)
 *   d_chi2_p, d_lnorm (nobj,)     parallax chi2; ln-normalisation (dim_prior == 0)
 *   d_ndim (nobj,) i32            phot_n, the chi-square degrees of freedom
 * writes d_lnl (nobj,) = logsumexp over points of (lnl + lnw), i.e. the
 * `lnl` of cluster.py:407 before the outlier mixture. */
/* Isochrone points of all secondary-mass-fraction slices from the plug-in's apparent
 * magnitudes (reference cluster.py:346-366, the `10**(-0.4 * seds)` and the
 * any-finite-band test of every slice): d_mags (nrow, nfilt) float64 and d_lnw_in (nrow)
 * = ln(d mini) + ln(d smf) over the whole table, d_src (npts) int32 the rows the host
 * kept, in order (NULL: all nrow = npts rows) -> d_pts_flux (npts, nfilt), d_pts_lnw
 * (npts), the inputs of brutus_cluster_lnl; a kept row without a finite band gets
 * weight -inf. */
int brutus_cluster_points(int64_t npts, int nfilt, const int32_t *d_src, const double *d_mags,
                          const double *d_lnw_in, double *d_pts_flux, double *d_pts_lnw,
                          void *stream);
/* The same where all slices share one initial-mass grid (the reference's isochrones do:
 * `mini` depends on EEP, [Fe/H] and age, not on the mass fraction): the weight of table row
 * r is d_lnw_eep[r % neep] + d_lnw_smf[r / neep] = ln(d mini) + ln(d smf) (-inf for an EEP
 * the caller drops), formed here instead of in a host array of nrow values per call. */
int brutus_cluster_points_grid(int64_t npts, int nfilt, int neep, const int32_t *d_src,
                               const double *d_mags, const double *d_lnw_eep,
                               const double *d_lnw_smf, double *d_pts_flux,
                               double *d_pts_lnw, void *stream);
size_t brutus_cluster_workspace_bytes(int nobj);
int brutus_cluster_lnl(int nobj, int nfilt, int npts, const double *d_pts_flux,
                       const double *d_pts_lnw, const double *d_phot,
                       const double *d_ivar, const double *d_chi2_p,
                       const double *d_lnorm, const int32_t *d_ndim,
                       int dim_prior, void *d_workspace, size_t workspace_bytes,
                       double *d_lnl, void *stream);
/* The same in pieces, for a caller that gets the isochrone points a few slices at a time
 * (the reference asks its plug-in once per secondary-mass-fraction slice, cluster.py:346-366)
 * and wants the device to work on one piece while the host prepares the next: the sum over
 * points is kept as brutus_cluster_chunks() partial (max, sum) pairs per object in the
 * workspace; `_part` fills the chunks [chunk_lo, chunk_lo + chunk_n) from ITS npts points
 * (npts = 0: they hold "no point"), `_merge` folds the first nchunk chunks into d_lnl.
 * brutus_cluster_lnl = one `_part` over all chunks + `_merge`.  Every chunk below nchunk
 * must have been filled by a `_part` call on the same stream. */
int brutus_cluster_chunks(void);
int brutus_cluster_lnl_part(int nobj, int nfilt, int npts, const double *d_pts_flux,
                            const double *d_pts_lnw, const double *d_phot,
                            const double *d_ivar, const double *d_chi2_p,
                            const double *d_lnorm, const int32_t *d_ndim, int dim_prior,
                            void *d_workspace, size_t workspace_bytes, int chunk_lo,
                            int chunk_n, void *stream);
/* `_part` straight from the plug-in's magnitude table: the inputs of
 * brutus_cluster_points_grid (kept rows d_src (npts), d_mags (nrow, nfilt), the two factors of
 * the ln-weight, neep) instead of a flux table -- the kernel forms fluxes and weights of its
 * own sub-slices, one launch and one table less per piece. */
int brutus_cluster_lnl_part_mags(int nobj, int nfilt, int npts, int neep, const int32_t *d_src,
                                 const double *d_mags, const double *d_lnw_eep,
                                 const double *d_lnw_smf, const double *d_phot,
                                 const double *d_ivar, const double *d_chi2_p,
                                 const double *d_lnorm, const int32_t *d_ndim, int dim_prior,
                                 void *d_workspace, size_t workspace_bytes, int chunk_lo,
                                 int chunk_n, void *stream);
int brutus_cluster_lnl_merge(int nobj, int nchunk, void *d_workspace, size_t workspace_bytes,
                             double *d_lnl, void *stream);
/* Outlier mixture and total of cluster.py:410-414 on the device: d_lnl_mix (nobj) =
 * logaddexp(d_lnl + ln_fin, d_lnl_outlier + ln_fout) (numpy's logaddexp, NaN and infinities
 * included) and d_lnl_tot (1) = their sum in a fixed order (the same bits on every run). */
int brutus_cluster_mix(int nobj, const double *d_lnl, const double *d_lnl_outlier, double ln_fin,
                       double ln_fout, double *d_lnl_mix, double *d_lnl_tot, void *stream);

/* ---- utils.photometric_offsets (reference utils.py:1218-1400) ------------------------
 * The per-band bootstrap of model / data flux ratios over the resampled fits of many
 * objects.  The caller keeps numpy's random stream and the final median / std over the
 * rounds; these two calls replace the reference's get_seds + phot_loglike + per-object
 * `choice` loop.
 *
 * brutus_offsets_weights: for every object o and resampled draw k
 *   d_models (nmodel, nfilt, 3) f32   the grid as utils.load_models returns it
 *   d_idxs, d_reds, d_dreds, d_dists (nobj, nsamps)   the fit's draws (negative idx wraps)
 *   d_phot, d_err (nobj, nfilt) f64, d_mask (nobj, nfilt) u8, d_weights (nobj, nsamps)
 *   d_old_offsets (nfilt), d_mask_fit (nfilt) u8
 *   d_use (nfilt, nobj) u8            the objects that enter band b (utils.py:1337-1350)
 * writes d_flux (nfilt, nobj, nsamps) = 10^(-0.4 sed) / dist^2 (utils.py:1327-1331) and
 * d_cdf (nfilt, nobj, nsamps) = cumulative normalised weights of the draws: leave-one-band-
 * out likelihood x weights where mask_fit[b], plain weights otherwise (utils.py:1355-1372);
 * rows with d_use == 0 are left untouched. */
int brutus_offsets_weights(int nobj, int nsamps, int nfilt, int64_t nmodel, const float *d_models,
                           const int64_t *d_idxs, const double *d_reds, const double *d_dreds,
                           const double *d_dists, const double *d_phot, const double *d_err,
                           const uint8_t *d_mask, const double *d_weights,
                           const double *d_old_offsets, const uint8_t *d_use,
                           const uint8_t *d_mask_fit, int dim_prior, double *d_flux, double *d_cdf,
                           void *stream);
/* brutus_offsets_bootstrap: the nmc bootstrap rounds of ONE band (utils.py:1374-1387):
 *   d_subset (n) i32   the objects of this band, d_cdf_obj (n) their cumulative weights
 *   d_u (nmc, 2, n)    the 2 n uniforms of every round in numpy's order: the n of
 *                      `choice(n, size=n, p=wt_obj)`, then one per `choice(Nsamps, p=wt[i])`
 * both `choice`s are numpy's legacy searchsorted(cdf, u, 'right'); writes d_meds (nmc) =
 * np.median of the round's model / data ratios.  0 bytes from the size query = bad sizes. */
size_t brutus_offsets_workspace_bytes(int n, int nmc);
int brutus_offsets_bootstrap(int band, int nobj, int nsamps, int nfilt, int n, int nmc,
                             const int32_t *d_subset, const double *d_cdf_obj, const double *d_u,
                             const double *d_flux, const double *d_cdf, const double *d_phot,
                             void *d_workspace, size_t workspace_bytes, double *d_meds,
                             void *stream);


/* Name and average duration (HIP events on `stream`) of the kernels launched
 * by the last *_batch call; used by bench.py for the roofline line. */
int brutus_last_timing(int *n_entries, const char **names, float *ms,
                       int max_entries);
void brutus_enable_timing(int on);

#ifdef __cplusplus
}
#endif
#endif /* BRUTUS_AMD_H */
