/*
 * brutus_amd_debug.h -- test hooks and measurement aids exported by libbrutus_amd.so.
 * NOT part of the product ABI (include/brutus_amd.h): nothing in brutus_amd/fitting.py's
 * product path calls these; tests/ and bench.py's calibration stream do.
 */
#ifndef BRUTUS_AMD_DEBUG_H
#define BRUTUS_AMD_DEBUG_H

#include "brutus_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: walk numpy stream(s) for nobj objects needing h_nnorm[o] normals and nuni
 * uniforms each (normals of object o at d_z + sum over earlier objects of
 * (h_nnorm rounded up to even) + 2; uniforms at d_u + o * nuni). */
int brutus_debug_mt_stream(int nobj, int nstream, uint32_t *h_states,
                           const int64_t *h_nnorm, int nuni, double *d_z, double *d_u,
                           void *stream);

/* Test hooks for the two building blocks above. */
int brutus_debug_rng(uint64_t seed, uint64_t start, int64_t n, double *d_normals,
                     double *d_uniforms, void *stream);
/* The ziggurat layer table of the normal stream as compiled into the library (host
 * copy of brutus_amd/csrc/zig_table.inc; n must be 1025): tests compare it with
 * brutus_amd/_zigtab.py.  Needs no GPU. */
int brutus_debug_zig_table(double *h_x, double *h_y, int n);
int brutus_debug_galprior(const brutus_post_params *params, int n,
                          const double *d_dist, const double *d_coord,
                          const double *d_feh, const double *d_loga, double *d_out,
                          void *stream);

/* The same ln prior in the form the Monte Carlo sample loop evaluates it (per-object
 * constant block read by scalar loads, table-driven halo power law when the parameters
 * admit it, the plain form otherwise).
 * Synchronises the stream. */
int brutus_debug_galprior_mc(const brutus_post_params *params, int n,
                             const double *d_dist, const double *d_coord,
                             const double *d_feh, const double *d_loga, double *d_out,
                             void *stream);

/* ... and through the sightline table the Monte Carlo kernels tabulate per work item
 * (post_kernels.hpp, "sightline table"): every 256 consecutive distances share one table whose
 * window is set by the nearest of them; d_used[i] = 1 where the table served the distance, 0
 * where it lay outside the window and the closed form was evaluated.  Fails with BRUTUS_EINVAL
 * for parameters that do not admit the halo table.  Synchronises the stream. */
int brutus_debug_galprior_sl(const brutus_post_params *params, int n,
                             const double *d_dist, const double *d_coord,
                             const double *d_feh, const double *d_loga, double *d_out,
                             int32_t *d_used, void *stream);

/* Measurement aid: brutus_fit_batch calls of this process so far and how many of them had to
 * be repeated by the host-driven driver (a star with more than eight magnitude sweeps, a flux
 * phase longer than the device-driven call's continuation rounds).  Needs no GPU. */
int brutus_debug_fit_stats(int64_t *calls, int64_t *repeated);

/* Measurement aid: out[i] = (double)in[i] for n elements, i.e. exactly 4n bytes
 * read (4 B/lane) and 8n bytes written (8 B/lane) -- the access widths of the
 * fused scan -- so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated
 * on a known byte count (MI355X_MICROARCH.md, HBM section). */
int brutus_calibrate_traffic(const float *d_in, double *d_out, int64_t n,
                             void *stream);
/* Plain device copy with 16 B per lane (nbytes a multiple of 16): the streaming
 * ceiling MI355X_MICROARCH.md quotes (6.29 TB/s) is measured with this access. */
int brutus_calibrate_copy16(const void *d_in, void *d_out, int64_t nbytes,
                            void *stream);

/* Measurement aid: the vector unit's issue rate.  Launches waves_per_simd x (number of CUs)
 * workgroups of 256 threads, every lane running `iters` x 128 back-to-back operations of one
 * kind (0 v_fmac_f32, 1 v_fmac_f64, 2 v_exp_f32) and nothing else; timed by the caller with
 * events on `stream`, (elapsed) / (iters x 128 x waves_per_simd) is the time one SIMD needs
 * per wave-instruction at the clock the device holds under that load -- the unit of
 * bench.py's `roofline.valu`.  d_scratch: waves_per_simd x CUs x 256 floats. */
int brutus_calibrate_issue(int kind, int iters, int waves_per_simd, float *d_scratch,
                           int64_t scratch_floats, void *stream);

/* Test hooks: y[i] = the kernels' own elementary functions for n inputs.  `which` in
 * brutus_debug_math: 0 10^x, 1 e^x, 2 ln x (the general forms); 3 sqrt x, 4 1/sqrt x, 5 1/x
 * (hardware seed + Newton); 6 e^x for finite x, 7 ln x for normal positive x (the
 * select-free forms of the Galactic prior), 8 / 9 the branch-free ln x / e^x. */
int brutus_debug_exp10(const double *d_x, double *d_y, int64_t n, void *stream);
int brutus_debug_math(int which, const double *d_x, double *d_y, int64_t n,
                      void *stream);

/* Test hook: copy one internal array of the workspace of the last brutus_fit_batch
 * (same nmodel / nfilt / nstar) into a caller-owned device buffer.  which =
 * 2, 3: the float32 statistics (nstar, nmodel) of the cull / first cut; 4: run-time audit max|f32 - f64| (3, nstar)
 * (BRUTUS_AUDIT=1); 5: per-star float32 block; 6, 7: exact cull / first-cut
 * thresholds (nstar,) f64; 8: float32 maxima (nstar, 10); 9: K1 status (nstar,) i32. */
int brutus_debug_copy(void *d_workspace, size_t workspace_bytes, int64_t nmodel,
                      int nfilt, int nstar, int which, void *d_dst, size_t nbytes,
                      void *stream);
int brutus_debug_sizeof_star32(void);

/* Measurement aid: the float32 pass alone.  Re-launches it `reps` (+ 1 untimed) times over the
 * star constants the last brutus_fit_batch left in the workspace (same nmodel / nfilt / nstar /
 * params; the float32 planes are overwritten with the same values) and reports the average
 * duration by HIP events on `stream`.  form: 0 = k_pre32s (all-vector), 1 = k_pre32m (band
 * contractions on the matrix pipe), other values = development variants where built. */
int brutus_debug_pre32_time(void *d_workspace, size_t workspace_bytes, const float *d_grid_soa,
                            int64_t nmodel, int nfilt, int nstar, const brutus_params *params,
                            int form, int reps, float *h_ms, void *stream);


#ifdef __cplusplus
}
#endif
#endif /* BRUTUS_AMD_DEBUG_H */
