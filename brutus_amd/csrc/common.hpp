// common.hpp -- constants, error helper, per-star / per-model device structs, reductions
// Part of the single translation unit brutus_kernels.hip (included there, in
// this order: common, fastmath, grid_kernels, fit_kernels, cluster_kernels,
// post_kernels); everything lives in that unit's anonymous namespace.
#pragma once

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
thread_local std::vector<TimingEntry> g_last_timing;   // per calling thread (scan-ahead + lnpost threads time concurrently)

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
    double D2;          // sum_j d_j^2 / V_j (single-pass chi2 of the fused scan)
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

// Three maxima at once (one barrier pair instead of three); `slot` = 12 doubles,
// out[0..2].  Double-buffered by the caller's loop parity is not needed: the
// trailing barrier protects the slots before the next use.
__device__ __forceinline__ void block_max_store3(double a, double b, double c, double *slot,
                                                 double *out) {
    a = wave_max(a);
    b = wave_max(b);
    c = wave_max(c);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        slot[w] = a;
        slot[4 + w] = b;
        slot[8 + w] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double *q = slot + 4 * threadIdx.x;
        double m = q[0];
        m = q[1] > m ? q[1] : m;
        m = q[2] > m ? q[2] : m;
        m = q[3] > m ? q[3] : m;
        out[threadIdx.x] = m;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------
// wave64 / workgroup prefix sums on the DPP data path (no LDS inside a wave)
// ---------------------------------------------------------------------------
// One step of a wave-wide inclusive scan: v + (v of the lane `ctrl` names), lanes the DPP
// pattern leaves without a source add 0.  gfx9 DPP controls: row_shr:n = 0x110 + n (within
// rows of 16 lanes), row_bcast:15 = 0x142 (lane 15 of a row to the next row, row mask 0xa),
// row_bcast:31 = 0x143 (lane 31 to rows 2 and 3, row mask 0xc).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_i32(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROWMASK, 0xf, false);
}
template <class T, int CTRL, int ROWMASK>
__device__ __forceinline__ T dpp_take(T v) {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit values");
    if constexpr (sizeof(T) == 4) {
        const int r = dpp_i32<CTRL, ROWMASK>(__builtin_bit_cast(int, v));
        return __builtin_bit_cast(T, r);
    } else {
        const long long b = __builtin_bit_cast(long long, v);
        const int lo = dpp_i32<CTRL, ROWMASK>((int)b), hi = dpp_i32<CTRL, ROWMASK>((int)(b >> 32));
        const long long r = ((long long)hi << 32) | (unsigned int)lo;
        return __builtin_bit_cast(T, r);      // (all-zero bits = 0 for integers and for +0.0)
    }
}
template <class T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
    v += dpp_take<T, 0x111, 0xf>(v);
    v += dpp_take<T, 0x112, 0xf>(v);
    v += dpp_take<T, 0x114, 0xf>(v);
    v += dpp_take<T, 0x118, 0xf>(v);
    v += dpp_take<T, 0x142, 0xa>(v);
    v += dpp_take<T, 0x143, 0xc>(v);
    return v;
}
// Exclusive prefix sum over a workgroup of NT threads (NT / 64 waves); `slot` is LDS scratch
// of NT / 64 + 1 values.  Returns this thread's exclusive prefix, `total` = the sum over the
// workgroup.  Two barriers; `slot` may be reused after the call returns on every thread of
// the NEXT call's first barrier (callers in a loop are safe: the second barrier here orders it).
template <class T, int NT>
__device__ __forceinline__ T block_exclusive_sum(T v, T *slot, T &total) {
    constexpr int NW = NT / 64;
    const T inc = wave_inclusive_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();                       // (slot free: everybody is past the previous use)
    if (lane == 63) slot[w] = inc;
    __syncthreads();
    T pre = T(0), tot = T(0);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const T x = slot[q];
        if (q < w) pre += x;
        tot += x;
    }
    total = tot;
    return pre + (inc - v);
}
// The INCLUSIVE sum of the same scan (for floating point `exclusive + v` is not it:
// (inc - v) + v need not round back to inc).
template <class T, int NT>
__device__ __forceinline__ T block_inclusive_sum(T v, T *slot, T &total) {
    constexpr int NW = NT / 64;
    const T inc = wave_inclusive_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) slot[w] = inc;
    __syncthreads();
    T pre = T(0), tot = T(0);
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        const T x = slot[q];
        if (q < w) pre += x;
        tot += x;
    }
    total = tot;
    return pre + inc;
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

}  // namespace
