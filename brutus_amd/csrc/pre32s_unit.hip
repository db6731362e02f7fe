// pre32s_unit.hip -- second translation unit of libbrutus_amd.so: the star-lane float32 pass
// (pre32s_kernels.hpp) and its launcher, compiled with -fno-slp-vectorize.
//
// Why a unit of its own.  hipcc's SLP vectoriser pairs neighbouring float32 operations of the
// unrolled band loops into v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.  On gfx950 a packed
// operation issues at HALF the rate of a plain one (tools/ubench/dpp_rate.hip: 2.3 against
// 1.16 ns per wave-instruction and SIMD), so packing buys nothing, and the moves that line the
// operands up cost k_pre32s 31 registers (128 instead of 97 before the rows went to LDS) and
// 0.15 ms per 128 stars (0.77 against 0.62).  The flag cannot be given per kernel, and applied
// to the whole library it pushes k_fflux<24> from 128 to 236 bytes of scratch -- so the one
// kernel that needs it lives here.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "pre32s_kernels.hpp"
#include "pre32m_kernels.hpp"

namespace {

template <int NB>
int launch_nb(int mfma, int rvf, const float *grid, int64_t nmodel, int64_t nmodel_pad, int nstar, int nrun,
              const int32_t *star_ids, const Star32 *stars, const P32 &p, float *lnlp32,
              float *lnpr32, float *part32, hipStream_t st) {
    const int ntile = (int)(nmodel_pad / PS_TILE);
    const int nblkx = (ntile + F2_T - 1) / F2_T;
#ifdef BRUTUS_DEV_PRE32M_DIAG
    if (mfma == 2) {
        const dim3 gm(8 * ((nblkx + 7) / 8) * ((nrun + 15) / 16)), bm(PS_TILE);
        if (rvf)
            hipLaunchKernelGGL((k_pre32m<NB, true, 1>), gm, bm, 0, st, grid, nmodel, nmodel_pad, nblkx, nstar, nrun,
                               star_ids, stars, p, lnlp32, lnpr32, part32);
        else
            hipLaunchKernelGGL((k_pre32m<NB, false, 1>), gm, bm, 0, st, grid, nmodel, nmodel_pad, nblkx, nstar, nrun,
                               star_ids, stars, p, lnlp32, lnpr32, part32);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
#endif
    if (mfma) {
        // (k_pre32m: 16 stars per wave, the band contractions on the matrix pipe)
        const dim3 gm(8 * ((nblkx + 7) / 8) * ((nrun + 15) / 16)), bm(PS_TILE);
        if (rvf)
            hipLaunchKernelGGL((k_pre32m<NB, true>), gm, bm, 0, st, grid, nmodel, nmodel_pad, nblkx, nstar, nrun,
                               star_ids, stars, p, lnlp32, lnpr32, part32);
        else
            hipLaunchKernelGGL((k_pre32m<NB, false>), gm, bm, 0, st, grid, nmodel, nmodel_pad, nblkx, nstar, nrun,
                               star_ids, stars, p, lnlp32, lnpr32, part32);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const dim3 g(nblkx * ((nrun + 63) / 64)), b(PS_TILE);
    if (rvf)
        hipLaunchKernelGGL((k_pre32s<NB, true>), g, b, 0, st, grid, nmodel, nmodel_pad, nstar, nrun,
                           star_ids, stars, p, lnlp32, lnpr32, part32);
    else
        hipLaunchKernelGGL((k_pre32s<NB, false>), g, b, 0, st, grid, nmodel, nmodel_pad, nstar, nrun,
                           star_ids, stars, p, lnlp32, lnpr32, part32);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" int brutus_i_pre32s_bands(int nb) {
#ifndef BRUTUS_DEV_NB12_ONLY
    if (nb == 8) return 1;
#endif
    return nb == 12 && ps_bands(12) ? 1 : 0;
}

extern "C" int brutus_i_pre32_layout(int what) {
    // what the two translation units must agree on (pre32_types.hpp is compiled into both)
    switch (what) {
        case 0: return (int)sizeof(Star32);
        case 1: return (int)sizeof(P32);
        case 2: return F2_T;
        case 3: return PS_TILE;
        case 4: return NV32;
        default: return -1;
    }
}

extern "C" int brutus_i_pre32s_launch(int nb, int mfma, int rvf, const float *grid, int64_t nmodel,
                                      int64_t nmodel_pad, int nstar, int nrun,
                                      const int32_t *star_ids, const void *stars32,
                                      const void *p32, float *lnlp32, float *lnpr32,
                                      float *part32, void *stream) {
    const Star32 *stars = static_cast<const Star32 *>(stars32);
    const P32 &p = *static_cast<const P32 *>(p32);
    hipStream_t st = (hipStream_t)stream;
    switch (nb) {
#ifndef BRUTUS_DEV_NB12_ONLY
        case 8: return launch_nb<8>(mfma, rvf, grid, nmodel, nmodel_pad, nstar, nrun, star_ids, stars, p, lnlp32, lnpr32, part32, st);
#endif
        case 12: return launch_nb<12>(mfma, rvf, grid, nmodel, nmodel_pad, nstar, nrun, star_ids, stars, p, lnlp32, lnpr32, part32, st);
        default: return -1;
    }
}
