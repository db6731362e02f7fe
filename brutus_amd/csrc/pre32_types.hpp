// pre32_types.hpp -- what the float32 proof pass shares between the two translation units of the
// library: brutus_kernels.hip (everything else) and pre32s_unit.hip (the star-lane pass, built
// with other compiler flags; see there).
#pragma once

#include <stdint.h>

#include "../../include/brutus_amd.h"

namespace {

#ifndef BRUTUS_PRE32_TYPES_TILE
#define BRUTUS_PRE32_TYPES_TILE
constexpr int PS_TILE = 256;                 // = TILE of common.hpp
constexpr int PS_NBMAX = BRUTUS_MAX_FILT;    // = NBMAX
#endif

#ifndef BRUTUS_F2_T
#define BRUTUS_F2_T 8
#endif
constexpr int F2_T = BRUTUS_F2_T;   // tiles per block of the float32 pass and its partial maxima (2048 models)
constexpr int NV32 = 10;       // float32 partial maxima per (block, star)

struct Star32 {
    float gc[PS_NBMAX];    // magnitude - weighted mean magnitude
    float w[PS_NBMAX];     // 1 / mags_var
    float dd[PS_NBMAX];    // flux / D,  D = 10^(-0.4 gbar)
    float iv[PS_NBMAX];    // D^2 / flux variance
    float S, DD2, gbar;
    float par, par_ivar, sp_mean, sp_var;
    float c0, c1;
    float eps, epsw;    // bounds on |f32 - f64| of lnl_p / lnprob and of logwt
    float chi2_lo;      // below this chi2 float32 is not trusted (re-evaluated in float64)
    int has_par, sp_on, ok;
};

struct P32 {
    float avmin, avmax, rvmin, rvmax, av_mean, av_ivar, rv_mean, rv_ivar;
    float mtol_hi, mtol_lo;     // mtol +- slack for the step test
    int dim_prior, nfilt;
};

}  // namespace

// The star-lane pass (pre32s_kernels.hpp), launched from pre32s_unit.hip.  nb: padded band
// count (8 or 12; brutus_pre32s_bands says which are built), rvf: Rv pinned.  Returns 0, or -1
// for a band count that is not built.  Not part of the C ABI (hidden visibility).
extern "C" __attribute__((visibility("hidden"))) int brutus_i_pre32s_bands(int nb);
extern "C" __attribute__((visibility("hidden"))) int brutus_i_pre32_layout(int what);
// mfma: 1 = k_pre32m (pre32m_kernels.hpp: the band contractions on the matrix pipe, 16 stars per
// wave), 0 = k_pre32s (all-vector, 64 stars per wave)
extern "C" __attribute__((visibility("hidden"))) int brutus_i_pre32s_launch(
    int nb, int mfma, int rvf, const float *grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
    const int32_t *star_ids, const void *stars32, const void *p32, float *lnlp32, float *lnpr32,
    float *part32, void *stream);
